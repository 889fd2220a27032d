#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by profiles/collect.sh) into the committed summaries under profiles/:
   <tag>_<wl>_kernel_stats.csv, <tag>_<wl>_bench.json, <tag>_<wl>_bench_under_rocprof.json and <tag>_traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies
128-byte fabric reads at 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section), so it is doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01_c"
root = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(root, "..", "gpurun_out", tag)


def per_kernel(path):
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            if not name.startswith("k"):
                continue
            a = acc.setdefault(name, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


traffic = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "workloads": {}}
for wl in ("c3", "c4", "c5"):
    st = glob.glob(os.path.join(src, "stats_" + wl, "*", "*_kernel_stats.csv"))
    if st:
        shutil.copy(st[0], os.path.join(root, "%s_%s_kernel_stats.csv" % (tag, wl)))
    for a, b in (("bench_%s.json", "%s_%s_bench.json"), ("stats_%s.json", "%s_%s_bench_under_rocprof.json")):
        p = os.path.join(src, a % wl)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(root, b % (tag, wl)))
    fe = glob.glob(os.path.join(src, "pmc_fetch_" + wl, "*", "*_counter_collection.csv"))
    wr = glob.glob(os.path.join(src, "pmc_write_" + wl, "*", "*_counter_collection.csv"))
    if not (fe and wr):
        continue
    f, w = per_kernel(fe[0]), per_kernel(wr[0])
    bench = json.load(open(os.path.join(src, "pmc_fetch_%s.json" % wl)))
    traffic["workloads"][wl] = {
        "batch": bench["config"]["images_per_gpu_per_step"],
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "kernels": {k: {"FETCH_SIZE_KiB": round(f[k]), "WRITE_SIZE_KiB": round(w.get(k, 0)),
                        "hbm_bytes_per_launch": int((2 * f[k] + w.get(k, 0)) * 1024)} for k in sorted(f)}}
json.dump(traffic, open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
