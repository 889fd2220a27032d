#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by profiles/collect.sh) into the committed summaries under profiles/:
   <tag>_<wl>_kernel_stats.csv, <tag>_<wl>_bench.json, <tag>_<wl>_bench_under_rocprof.json, <tag>_traffic.json
   (also copied to rNN_traffic.json, NN from the tag: bench.py reads the newest) and <tag>_sq_counters.json.

HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE tallies 128-byte
fabric reads at 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section), so it is doubled; WRITE_SIZE is exact.  Counters
are averaged per launch and multiplied by the launches of that kernel in one step (k1_sync runs three times a step).
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


def per_kernel(path, by_counter=False):
    """-> {kernel: mean counter value per launch} (or {kernel: {counter: mean}}) and {kernel: launches seen}"""
    acc, launches = {}, {}
    with open(path) as f:
        for row in csv.DictReader(f):
            # "void k2_mcu_lanes<4, true>(JdKernelArgs, ...)" -> "k2_mcu_lanes": the template instances of one kernel (sampling
            # mode, coefficient form) are one row; a batch of one kind of image launches exactly one of them
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
            if not name.startswith("k"):
                continue
            key = (name, row["Counter_Name"]) if by_counter else name
            a = acc.setdefault(key, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            launches.setdefault(name, set()).add(row["Dispatch_Id"])
    mean = {k: v[0] / v[1] for k, v in acc.items()}
    if by_counter:
        out = {}
        for (name, c), v in mean.items():
            out.setdefault(name, {})[c] = v
        mean = out
    return mean, {k: len(v) for k, v in launches.items()}


traffic = {"note": __doc__.split("\n\n")[1].replace("\n", " "), "workloads": {}}
sq = {"note": "SQ counters per launch (mean over the launches of the profiled run; rocprofv3 sums over the XCDs). "
              "SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); "
              "SQ_BUSY_CYCLES comes out as 128 x the kernel's duration (checked against the kernel trace at ~2.1 GHz), i.e. it "
              "is summed over 128 counter instances, and the chip has 1024 SIMDs, so valu_busy = SQ_ACTIVE_INST_VALU / "
              "(8 * SQ_BUSY_CYCLES) is the share of SIMD-cycles in which a vector instruction was executing.", "workloads": {}}
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
    (f, nl), (w, _) = per_kernel(fe[0]), per_kernel(wr[0])
    bench = json.load(open(os.path.join(src, "pmc_fetch_%s.json" % wl)))
    # passes the profiled process made: three while it chose `steps_internal`, then (warmup + steps) x steps_internal
    steps = 3 + (bench["steps"] + bench["warmup"]) * bench.get("steps_internal", 1)
    per_step = {k: max(1, round(nl[k] / steps)) for k in nl}
    traffic["workloads"][wl] = {
        "images_per_step": bench["config"]["items_per_gpu_per_pass"],
        "algorithmic_bytes_per_step": bench["roofline"]["algorithmic_bytes_per_launch"],
        "kernels": {k: {"FETCH_SIZE_KiB_per_launch": round(f[k]), "WRITE_SIZE_KiB_per_launch": round(w.get(k, 0)),
                        "launches_per_step": per_step[k],
                        "hbm_bytes_per_step": int((2 * f[k] + w.get(k, 0)) * 1024 * per_step[k])} for k in sorted(f)}}
    # ... and per stage of jd_batch_run (what bench.py's stage times are): the kernels each stage launches
    stages = {"unstuff": ["k0_unstuff"], "huffman_intervals": ["k1_write_intervals"],
              "huffman_sync": ["k1_sync", "k1_sync_spine", "k1f_cold"],
              "huffman_write": ["k1_write_subseq", "k1_dc_scan", "k1f_write", "k1f_check", "k1f_scan", "k1f_index"],
              "idct_colour": ["k2_mcu_lanes"]}
    kk = traffic["workloads"][wl]["kernels"]
    traffic["workloads"][wl]["stages"] = {s: {"kernels": [k for k in ks if k in kk], "hbm_bytes_per_step": sum(kk[k]["hbm_bytes_per_step"] for k in ks if k in kk)}
                                          for s, ks in stages.items() if any(k in kk for k in ks)}
    traffic["workloads"][wl]["hbm_bytes_per_step_all_kernels"] = sum(v["hbm_bytes_per_step"] for v in traffic["workloads"][wl]["kernels"].values())
    cs = {}
    for part in ("sq1_", "sq2_"):
        g = glob.glob(os.path.join(src, part + wl, "*", "*_counter_collection.csv"))
        if g:
            m, _ = per_kernel(g[0], by_counter=True)
            for k, d in m.items():
                cs.setdefault(k, {}).update({c: round(v) for c, v in d.items()})
    for k, d in cs.items():
        if d.get("SQ_BUSY_CYCLES"):
            d["valu_busy"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) / (8.0 * d["SQ_BUSY_CYCLES"]), 3)
            d["kernel_cycles"] = round(d["SQ_BUSY_CYCLES"] * 4 / 128)
        if d.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = round(d.get("SQ_INSTS_VALU", 0) / d["SQ_WAVES"], 1)
    if cs:
        sq["workloads"][wl] = cs
json.dump(traffic, open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(root, tag.split("_")[0] + "_traffic.json"), "w"), indent=1)
json.dump(sq, open(os.path.join(root, tag + "_sq_counters.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps(sq, indent=1))
