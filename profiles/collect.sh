#!/bin/bash
# Collect the round's evidence on the GPU box (run from the repo root through gpurun):
#   profiles/collect.sh <tag> [workloads]          e.g. profiles/collect.sh r03_a "c3 c4 c5"
# For each workload: the plain bench line (with the end_to_end sample), rocprofv3 --kernel-trace --stats of the same
# command, the two HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: 3 + 2 TCC counters of 4) and
# two SQ passes (8 SQ counters each).  Every GPU step has its own time limit and the chain stops at the first failure.
# Raw output goes to gpurun_out/<tag>/; summarise with profiles/summarise.py <tag>.
set -e -o pipefail
tag=${1:-r03_f}
wls=${2:-"c3 c4 c5"}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for wl in $wls; do
  timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 5 > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "bench $wl done"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python3 bench.py --workload $wl --steps 10 --no-cpu-baseline --no-e2e --no-two-streams > $out/stats_$wl.json 2> $out/stats_$wl.log
  echo "stats $wl done"
  short="--steps 3 --warmup 1 --repeat 1 --no-cpu-baseline --no-e2e --no-two-streams"
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$wl -- python3 bench.py --workload $wl $short > $out/pmc_fetch_$wl.json 2> $out/pmc_fetch_$wl.log
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$wl -- python3 bench.py --workload $wl $short > $out/pmc_write_$wl.json 2> $out/pmc_write_$wl.log
  echo "pmc $wl done"
  timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq1_$wl -- python3 bench.py --workload $wl $short > $out/sq1_$wl.json 2> $out/sq1_$wl.log
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $out/sq2_$wl -- python3 bench.py --workload $wl $short > $out/sq2_$wl.json 2> $out/sq2_$wl.log
  echo "sq $wl done"
done
