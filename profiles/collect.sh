#!/bin/bash
# Collect the round's evidence on the GPU box (run from the repo root through gpurun):
#   profiles/collect.sh <tag>            e.g. r01_c
# For each workload: the plain bench line, rocprofv3 --kernel-trace --stats of the same command, and the two
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: 3 + 2 TCC counters of 4).  Every GPU step has
# its own time limit and the chain stops at the first failure.  Raw output goes to gpurun_out/<tag>/; summarise with
# profiles/summarise.py.
set -e -o pipefail
tag=${1:-r01_c}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for wl in c3 c4 c5; do
  timeout -k 10 240 python3 bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "bench $wl done"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 bench.py --workload $wl --steps 10 --no-cpu-baseline > $out/stats_$wl.json 2> $out/stats_$wl.log
  echo "stats $wl done"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_fetch_$wl.json 2> $out/pmc_fetch_$wl.log
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_write_$wl.json 2> $out/pmc_write_$wl.log
  echo "pmc $wl done"
done
