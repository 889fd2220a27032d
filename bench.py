#!/usr/bin/env python3
"""bench.py -- decoded megapixels/s of the MI355X JPEG hot path (BASELINE.json metric).

One "pass" = the whole hot path (unstuff + restart index, Huffman decode, dequant + IDCT, chroma spread + colour) over
the rank's synthetic JPEGs, whose entropy-coded segments are already resident in HBM; outputs (RGB8) stay in HBM.  One
"step" = `steps_internal` passes back to back (chosen so that the K timed steps last about a second: an outside sampler
can then see the GPU busy; `--repeat 1` makes a step one pass).  Workloads:

  c3   8192x8192 4:2:0, Annex-K tables at quality 85, DRI every 8 MCUs; `--images` per GPU per pass (16, all distinct:
       474 MB of entropy-coded data, more than the 256 MB Infinity Cache); weak scaling.
       `--shard bands`: the SAME `--images` pictures for the whole job, each cut by jd_split_bands into one band of MCU
       rows per rank (restart-interval-range sharding); strong scaling
  c4   the same images without restart markers (self-synchronising Huffman decode), 64 per GPU per pass (the 16 distinct
       pictures four times over, every image with its own copy of its data in HBM: the sync pass's launches amortise
       their tails -- 245k / 260k / 268k MP/s at 16 / 32 / 64 images per pass); weak scaling
  c5   BASELINE's "batch of 10k 1920x1080 4:2:0": `--total` images (10000; `--distinct` 64 seeds cycled, every image
       with its own copy of its data in HBM) dealt over the ranks by jpeg_decoder_amd.shard_images and decoded in
       per-rank batches of `--batch` (5000: measured 236k / 243k / 244k / 249k / 248k MP/s at 1000 / 2000 / 2500 / 5000 /
       10000 per batch on one GPU -- fewer, larger launches); strong scaling
  prog the reference's four progressive (SOF2) files and the 14 libjpeg twins through jd_decode_scans_batch (SURVEY
       row N4), host memory in and out, at batch sizes 1 / 64 / 1024 / 4096; one rank

With no `--workload` the headline is c3 and the same process then measures c4 and c5 (and, on one GPU, prog; on several,
c3 band-sharded) and attaches them as `"workloads": {...}`, each with its own value / ms_per_step / roofline / verified.

Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks time); images or
bands shard with no data-path collective.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
STARTS the N ranks itself (child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; the parent never touches
the GPU), relays rank 0's line and exits with the worst child's code; under torch.distributed.run it is a rank.
`--dry` replaces the GPU work by nothing at all (gloo backend, planning and rendezvous only): the launcher test on CPU.
`--share-gpu` rehearses the N-rank path on a one-GPU box (all ranks on device 0, gloo): correctness, not speed.

Rank 0 prints ONE JSON line.  `value` is metric (i) of SURVEY 8(d): inputs and outputs resident in HBM.  `roofline` is
for the dominant stage (device time from hipEvents recorded inside the timed steps, on the stream the kernels run on);
`cpu_baseline` is the CPU oracle (a port of the reference's algorithm, single thread) on a bounded sample, N=1 only;
`two_streams` (workloads of several batches, one rank): the same batches run alternately on two HIP streams -- an
extra figure, `value` stays single-stream so that the stage times add up to the pass.
`verified`: one image of every batch of the timed configuration equals the oracle's pixels; `end_to_end` (N=1, after
the timed region, on a bounded sample) holds metrics (ii) and (iii): pinned host memory -> H2D -> kernels -> D2H ->
pinned host memory with three batches in flight, without and with the host's header parsing.
`--mode e2e|host` makes (ii) / (iii) the timed quantity of the whole run instead.
"""
import argparse
import ctypes
import glob
import json
import math
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
GEOM = {"c3": (8192, 8192, 8), "c4": (8192, 8192, 0), "c5": (1920, 1080, 0)}
LABEL = {"c3": "C3: synthetic 8192x8192 baseline 4:2:0, DRI every 8 MCUs",
         "c4": "C4: synthetic 8192x8192 baseline 4:2:0, no restart markers",
         "c5": "C5: batch of synthetic 1920x1080 baseline 4:2:0, no restart markers",
         "prog": "N4: the reference's progressive test files + libjpeg twins, jd_decode_scans_batch"}
MODE_LABEL = {"resident": "(i) inputs and outputs resident in HBM",
              "e2e": "(ii) pinned host -> H2D -> kernels -> D2H -> pinned host, 3 batches in flight",
              "host": "(iii) as (ii) + header parsing and ECS staging from whole files"}
STAGE_KERNEL = {"unstuff": "k0_unstuff", "huffman_intervals": "k1_write_intervals", "huffman_sync": "k1_sync",
                "huffman_write": "k1_write_subseq", "idct_colour": "k2_mcu_lanes"}
# streams without restart markers take the single-decode path unless JD_NODRI_FUSED=0 (jd_internal.h): its stages
STAGE_KERNEL_FUSED = dict(STAGE_KERNEL, huffman_sync="k1f_cold", huffman_write="k1f_write (+ k1f_check, k1f_scan, k1f_index)")


def stage_kernel(stage, ri):
    fused = ri == 0 and os.environ.get("JD_NODRI_FUSED", "1") != "0"
    return (STAGE_KERNEL_FUSED if fused else STAGE_KERNEL).get(stage, stage)
DTYPE = "f64 IDCT / f32 colour / int16 coefficients (bytes between the kernels)"
TARGET_TIMED_S = 1.2     # what `--repeat 0` aims the timed region at


# ---------------------------------------------------------------------------------------------------------------
# launcher: `--gpus N` without a rendezvous in the environment starts the N ranks (the reference's file loop,
# main.go:1509-1512, dealt over processes).  Nothing here imports torch or touches HIP.
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    for line in out0.decode().splitlines():                    # rank 0's JSON line (gloo prints a connection notice to stdout)
        if line.startswith("{"):
            print(line, flush=True)
    worst = max(codes, key=lambda c: (c != 0, abs(c)))
    return worst if 0 <= worst < 256 else 1


# ---------------------------------------------------------------------------------------------------------------
class Rank:
    """what a rank knows about the job: its place, its device, the two collectives the bench uses"""

    def __init__(self, dry, share_gpu=False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dry = dry
        import torch
        self.torch = torch
        self.dist = None
        self.gpu = 0 if share_gpu else self.local_rank          # HIP device this rank decodes on
        if dry:
            self.dev = self.cdev = torch.device("cpu")
            if self.world > 1:
                import torch.distributed as dist
                dist.init_process_group("gloo")
                self.dist = dist
        else:
            torch.cuda.set_device(self.gpu)
            self.dev = torch.device("cuda", self.gpu)
            self.cdev = torch.device("cpu") if share_gpu else self.dev   # where the barrier's / reductions' tensors live
            if self.world > 1:
                import torch.distributed as dist
                if share_gpu:                                   # rehearsal on a one-GPU box: RCCL refuses two ranks on one device
                    dist.init_process_group("gloo")
                else:
                    dist.init_process_group("nccl", device_id=self.dev)
                self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize()

    def max_f(self, v):
        if self.dist is None:
            return float(v)
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_i(self, v):
        if self.dist is None:
            return int(v)
        t = self.torch.tensor([v], dtype=self.torch.int64, device=self.cdev)
        self.dist.all_reduce(t)
        return int(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


_SYNTH = {}   # (w, h, seed, quality, detail) -> SynthImage: c3 and c4 are twins of the same pictures


def synth(jd, w, h, seed, quality, detail):
    key = (w, h, seed, quality, detail)
    if key not in _SYNTH:
        _SYNTH[key] = jd.SynthImage(w, h, 2, 2, 3, quality=quality, seed=seed, detail=detail)
    return _SYNTH[key]


def make_distinct(jd, w, h, ri, seeds, quality, detail):
    """-> {seed: (file bytes, SynthImage)} (host threads: generation is host-bound)"""
    def one(seed):
        im = synth(jd, w, h, seed, quality, detail)
        return seed, (im.encode(ri), im)
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        return dict(ex.map(one, seeds))


_LAST_ORACLE = [None, None]


def oracle_rgb(synth_image):
    """CPU oracle's pixels for one synthetic image (its no-DRI twin: the reference cannot decode restart markers, F3)"""
    if _LAST_ORACLE[0] is synth_image:
        return _LAST_ORACLE[1]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    o = orclib.OracleImage(synth_image.encode(0))
    _LAST_ORACLE[0], _LAST_ORACLE[1] = synth_image, (o.rgb, o.t_entropy + o.t_pixels)
    return _LAST_ORACLE[1]


_CPU_BASE = {}


def cpu_baseline(jd, w, h, quality, detail, first):
    """CPU oracle (restatement of the reference's algorithm, 1 thread) on a bounded sample of the workload's geometry.
    The reference cannot decode restart markers, so c3 is timed on its no-DRI twin (= c4's files).
    `first` = hot-path seconds already measured on one image by the verification leg."""
    if (w, h) in _CPU_BASE:
        return _CPU_BASE[(w, h)]
    n = 6 if w * h > 4e6 else 120
    px, t_hot = w * h, first
    for i in range(1, n):
        _, t = oracle_rgb(jd.SynthImage(w, h, 2, 2, 3, quality=quality, seed=1000 + i, detail=detail))
        t_hot += t                                            # main.go:1032 + 1054-1057 only
        px += w * h
    _CPU_BASE[(w, h)] = {
        "value": round(px / t_hot / 1e6, 3), "unit": "MP/s", "cores": 1, "cores_available": os.cpu_count(), "kind": "port",
        "sample": "%d x %dx%d 4:2:0 q%d no-DRI twin; in-memory hot path only (entropy decode + dequant/IDCT/"
                  "spread/colour), %.1f s on 1 of %d host cores" % (n, w, h, quality, t_hot, os.cpu_count())}
    return _CPU_BASE[(w, h)]


def recorded_traffic(wl, n_mine, stage):
    """HBM bytes per pass moved by the kernels of `stage` (all their launches) from the committed PMC passes (newest
    profiles/rNN_traffic.json), if they were taken on this workload and batch.  PMC counters cannot be read from inside
    this process: the figure is RECORDED, not live (`traffic_source`)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path))["workloads"][wl]
            if rec["images_per_step"] != n_mine:
                continue
            return rec["stages"][stage]["hbm_bytes_per_step"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def end_to_end(jd, torch, files, frames, w, h, device, host_parse, passes, per_slot, slots=3, threads=16, want=None):
    """Metrics (ii) / (iii): `slots` pinned-I/O batches of `per_slot` images in flight.  Every submit copies the slot's
    ECS staging to the device, decodes, and copies all RGB back into pinned host memory.  host_parse: each group is
    first parsed and staged from whole JPEG files in ordinary host memory (jd_batch_stage_jpegs)."""
    cap = max(len(f) for f in files)
    batches = [jd.Batch(frames[:per_slot], [cap] * per_slot, device=device, pinned_io=True) for _ in range(slots)]
    groups = [[files[(g * per_slot + i) % len(files)] for i in range(per_slot)] for g in range(slots)]
    arrs = [[np.frombuffer(f, np.uint8) for f in grp] for grp in groups]
    ptrs = [(ctypes.c_void_p * per_slot)(*[a.ctypes.data for a in grp]) for grp in arrs]
    lens = [(ctypes.c_size_t * per_slot)(*[a.size for a in grp]) for grp in arrs]
    for s, b in enumerate(batches):                         # warm-up: stage + one full trip per slot
        b.stage_jpegs_raw(ptrs[s], lens[s], per_slot, threads=threads)
        b.submit(); b.wait()
    n_sub = passes * slots
    pending = [False] * slots
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_sub):
        s = k % slots
        if pending[s]:
            batches[s].wait()
        if host_parse:
            batches[s].stage_jpegs_raw(ptrs[s], lens[s], per_slot, threads=threads)
        batches[s].submit()
        pending[s] = True
    for s in range(slots):
        if pending[s]:
            batches[s].wait()
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(batches[0].host_rgb(0), want)) if want is not None else None
    for b in batches:
        b.close()
    return n_sub * per_slot * w * h / dt / 1e6, n_sub * per_slot, ok


def pcie_rates(torch, dev, mb=512):
    """plain pinned-memory copies, for the PCIe roofline of the end-to-end metrics"""
    n = mb << 20
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    out = {}
    for name, (dst, src) in {"h2d": (d, host), "d2h": (host, d)}.items():
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        out[name] = 3 * n / (time.perf_counter() - t0) / 1e9
    return out


# ---------------------------------------------------------------------------------------------------------------
def plan_workload(jd, R, wl, shard, args):
    """Which pictures this rank decodes.  -> dict(seeds, items, per_batch, scaling, parallelism); an item is
    (seed, band index or None).  Host-only: `--dry` runs exactly this."""
    if wl == "c5":
        per_batch = args.batch or 5000
        distinct = args.distinct or 64
        plan = jd.plan_job(args.total, R.world, R.rank, per_batch, distinct)   # strong scaling: the job is `total` images
        items = [(s, None) for grp in plan for _, s in grp]
        return {"seeds": sorted(set(s for s, _ in items)), "items": items, "per_batch": per_batch, "scaling": "strong",
                "parallelism": "whole images dealt over the GPUs (shard_images), no collective"}
    n_images = args.images or (64 if wl == "c4" else 16)
    per_batch = args.batch or n_images
    distinct = min(args.distinct or 16, n_images)
    if shard == "bands":
        seeds = [i for i in range(distinct)]                    # the same pictures on every rank: the job is `images` pictures
        mine = jd.shard_bands(R.world, R.world, R.rank)         # band k of every picture -> rank k % world
        items = [(seeds[i % distinct], k) for i in range(n_images) for k in mine]
        return {"seeds": seeds, "items": items, "per_batch": per_batch * max(1, len(mine)), "scaling": "strong",
                "parallelism": "every picture cut into %d bands of MCU rows (jd_split_bands), band k on rank k (shard_bands), "
                               "no collective" % R.world}
    seeds = [R.rank * 100003 + i for i in range(distinct)]     # weak scaling: every rank its own pictures
    items = [(seeds[i % distinct], None) for i in range(n_images)]
    return {"seeds": seeds, "items": items, "per_batch": per_batch, "scaling": "weak",
            "parallelism": "whole images dealt over the GPUs (shard_images), no collective"}


def cut_item(jd, R, parsed, file, band):
    """-> (frame, ecs offset in the file, ecs length, first pixel row, pixel rows) of an item"""
    fr, off, ln = parsed
    if band is None:
        return fr, off, ln, 0, fr.height
    bands = jd.split_bands(fr, np.frombuffer(file, np.uint8, ln, off), R.world)
    if band >= len(bands):
        return None
    b = bands[band]
    return b.frame, off + b.ecs_off, b.ecs_len, b.y0, b.frame.height


def run_resident(jd, R, wl, shard, args, headline):
    """metric (i) for one workload -> the result dict (rank 0) or None"""
    torch = R.torch
    w, h, ri = GEOM[wl]
    plan = plan_workload(jd, R, wl, shard, args)
    made = make_distinct(jd, w, h, ri, plan["seeds"], args.quality, args.detail)
    parsed = {s: jd.parse(made[s][0]) for s in plan["seeds"]}
    cuts = {}
    for s, band in set(plan["items"]):
        cuts[(s, band)] = cut_item(jd, R, parsed[s], made[s][0], band)
    items = [it for it in plan["items"] if cuts[it] is not None]
    n_mine = len(items)
    rows_mine = sum(cuts[it][4] for it in items)               # pixel rows this rank produces per pass
    if R.dry:
        n_job = R.sum_i(rows_mine * w)
        R.barrier()
        return {"dry": True, "pixels_per_pass": n_job, "items_rank0": n_mine, "scaling": plan["scaling"],
                "parallelism": plan["parallelism"]} if R.rank == 0 else None

    src = {it: torch.from_numpy(np.frombuffer(made[it[0]][0], np.uint8, cuts[it][2], cuts[it][1]).copy()).to(R.dev)
           for it in set(items)}
    # explicit streams: a NULL handle would mean "each batch's own stream" to jd_batch_run, and batches would overlap
    tstreams = [torch.cuda.Stream(device=R.dev) for _ in range(max(1, args.streams))]
    per_batch = plan["per_batch"]
    # every item owns its entropy-coded bytes in HBM (no two share cache lines) and its RGB output
    batches, keep = [], []
    for b0 in range(0, n_mine, per_batch):
        idx = range(b0, min(b0 + per_batch, n_mine))
        frames = [cuts[items[i]][0] for i in idx]
        lens = [cuts[items[i]][2] for i in idx]
        rgbs = [torch.empty((cuts[items[i]][4], w, 3), dtype=torch.uint8, device=R.dev) for i in idx]
        b = jd.Batch(frames, lens, device=R.gpu, external_io=True)
        for k, i in enumerate(idx):
            e = src[items[i]].clone()
            assert e.data_ptr() % 16 == 0
            keep.append(e)
            b.bind_ecs(k, e.data_ptr(), lens[k])
            b.bind_rgb(k, rgbs[k].data_ptr(), w * 3)
        batches.append((b, rgbs, list(idx)))
    torch.cuda.synchronize()                                   # the copies above ran on torch's default stream

    def one_pass():
        for k, (b, _, _) in enumerate(batches):
            b.run(tstreams[k % len(tstreams)].cuda_stream)

    # ---- passes per step: aim the K timed steps at about a second --------------------------------------------
    one_pass(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        one_pass()
    torch.cuda.synchronize()
    t_pass = R.max_f((time.perf_counter() - t0) / 2)
    rep = args.repeat if args.repeat > 0 else max(1, int(math.ceil(TARGET_TIMED_S / (max(args.steps, 1) * t_pass))))

    def step():
        for _ in range(rep):
            one_pass()
    for _ in range(args.warmup):
        step()
    for b, _, _ in batches:
        b.wait()
        b.timing(True)
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    R.barrier()
    dt = time.perf_counter() - t0
    stages, nruns = {}, 0
    for b, _, _ in batches:
        b.wait()                                               # surfaces any decode error
        nruns, st = b.stage_times()
        for k, v in st.items():
            stages[k] = stages.get(k, 0.0) + v
    # ---- the same batches dealt over two streams (extra, one rank: what a host with several batches in flight gets; the
    # figures above stay single-stream so that the stage times add up to the pass) -----------------------------------
    two = None
    if len(batches) >= 2 and len(tstreams) == 1 and R.world == 1 and not args.no_two_streams:
        s2 = [torch.cuda.Stream(device=R.dev) for _ in range(2)]

        def pass2():
            for k, (b, _, _) in enumerate(batches):
                b.run(s2[k % 2].cuda_stream)
        pass2(); torch.cuda.synchronize()
        n2 = max(2, (args.steps * rep) // 2)
        t2 = time.perf_counter()
        for _ in range(n2):
            pass2()
        torch.cuda.synchronize()
        two = (time.perf_counter() - t2) / n2
        for b, _, _ in batches:
            b.wait()
    ecs_bytes = sum(cuts[it][2] for it in items)
    dt = R.max_f(dt)
    px_job = R.sum_i(rows_mine * w)                            # pixels the whole job decodes per pass
    items_job = R.sum_i(n_mine)

    # ---- one image of EVERY batch against the oracle (outside the timed region; every rank checks its own) ------
    ok, t_first = True, 0.0
    for b, rgbs, idx in batches:
        seed, band = items[idx[0]]
        want, t1 = oracle_rgb(made[seed][1])
        t_first = t_first or t1
        y0, rows = cuts[items[idx[0]]][3], cuts[items[idx[0]]][4]
        ok = ok and bool(np.array_equal(rgbs[0].cpu().numpy(), want[y0:y0 + rows]))
        if wl != "c5" and not headline:
            break
    ok_all = R.sum_i(0 if ok else 1) == 0
    want0 = oracle_rgb(made[items[0][0]][1])[0] if (headline and R.world == 1 and not args.no_e2e) else None

    out = None
    if R.rank == 0:
        ms_step = dt / args.steps * 1e3
        out = {"value": round(px_job * rep / (dt / args.steps) / 1e6, 1), "unit": "MP/s", "ms_per_step": round(ms_step, 4),
               "steps_internal": rep, "ms_per_pass": round(ms_step / rep, 4), "timed_region_s": round(dt, 3),
               "scaling": plan["scaling"],
               "config": {"workload": LABEL[wl] + ("; band-sharded" if shard == "bands" else ""), "mode": MODE_LABEL["resident"],
                          "images_per_step": (px_job // (w * h)) * rep, "images_per_pass": px_job // (w * h),
                          "items_per_gpu_per_pass": n_mine, "items_per_pass": items_job, "images_per_batch": per_batch,
                          "streams": len(tstreams), "distinct_images_per_gpu": len(plan["seeds"]), "quality": args.quality,
                          "restart_interval": ri, "parallelism": plan["parallelism"],
                          "ecs_bytes_per_pixel": round(ecs_bytes / max(1, rows_mine * w), 4)}}
        dom = max((k for k in stages if k != "clear"), key=lambda k: stages[k])
        alg_bytes = ecs_bytes + 3 * w * rows_mine               # stuffed ECS in + RGB8 out, per pass (this rank)
        achieved = alg_bytes / (stages[dom] * 1e-6) / 1e9
        traffic, tsrc = recorded_traffic(wl, n_mine, dom) if shard != "bands" else (None, None)
        out["roofline"] = {"bound": "hbm", "kernel": stage_kernel(dom, ri), "stage": dom, "achieved": round(achieved, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "traffic_source": ("recorded: " + tsrc) if tsrc else None,
                           "launch_us": round(stages[dom], 1), "algorithmic_bytes_per_launch": alg_bytes,
                           "note": "per pass of one rank: all launches of the kernel in a pass together",
                           "whole_path_frac": round(alg_bytes / (sum(stages.values()) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                           "stage_us": {k: round(v, 1) for k, v in stages.items()}, "runs_averaged": nruns}
        if two:
            out["two_streams"] = {"value": round(rows_mine * w / two / 1e6, 1), "unit": "MP/s", "ms_per_pass": round(two * 1e3, 4),
                                  "note": "the same batches run alternately on two HIP streams (batches in flight overlap "
                                          "each other's latency-bound tails); `value` above is single-stream"}
        out["verified"] = ok_all
        out["verified_images"] = len(batches) if (wl == "c5" or headline) else 1
        if R.world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(jd, w, h, args.quality, args.detail, t_first)
    for b, _, _ in batches:
        b.close()
    del batches, keep, src
    torch.cuda.empty_cache()
    if out is not None and want0 is not None and shard != "bands":
        per_slot = 2 if wl != "c5" else 128
        files = [made[it[0]][0] for it in items[:min(n_mine, 384)]]
        files = [files[i % len(files)] for i in range(3 * per_slot)]
        frames = [parsed[items[0][0]][0]] * per_slot
        passes = 3 if wl != "c5" else 6
        pc = pcie_rates(torch, R.dev)
        ii, n_ii, ok_ii = end_to_end(jd, torch, files, frames, w, h, R.gpu, False, passes, per_slot, want=want0)
        iii, n_iii, ok_iii = end_to_end(jd, torch, files, frames, w, h, R.gpu, True, passes, per_slot, want=want0)
        roof = pc["d2h"] * 1e9 / 3.0 / 1e6                      # MP/s if nothing but the RGB copy-back took time
        out["end_to_end"] = {"device_incl_pcie_MPs": round(ii, 1), "host_incl_parse_MPs": round(iii, 1),
                             "images_timed": n_ii, "batches_in_flight": 3, "images_per_batch": per_slot,
                             "pcie_h2d_GBs": round(pc["h2d"], 1), "pcie_d2h_GBs": round(pc["d2h"], 1),
                             "pcie_roofline_MPs": round(roof, 1), "frac_of_pcie_roofline": round(ii / roof, 3),
                             "verified": bool(ok_ii and ok_iii)}
    return out


def run_pipeline(jd, R, wl, args):
    """--mode e2e | host: metric (ii) / (iii) as the timed quantity"""
    w, h, ri = GEOM[wl]
    plan = plan_workload(jd, R, wl, "images", args)
    made = make_distinct(jd, w, h, ri, plan["seeds"], args.quality, args.detail)
    parsed = {s: jd.parse(made[s][0]) for s in plan["seeds"]}
    n_mine = len(plan["items"])
    per_slot = args.batch or (4 if wl != "c5" else 250)
    files = [made[plan["items"][i % n_mine][0]][0] for i in range(min(n_mine, 3 * per_slot))]
    frames = [parsed[plan["items"][0][0]][0]] * per_slot
    passes = max(1, -(-n_mine // (3 * per_slot))) * args.steps
    R.barrier()
    mps, n_img, _ = end_to_end(jd, R.torch, files, frames, w, h, R.gpu, args.mode == "host", passes, per_slot)
    dt = R.max_f(n_img * w * h / (mps * 1e6))
    R.barrier()
    n_job = R.sum_i(n_img)
    if R.rank != 0:
        return None
    return {"value": round(n_job * w * h / dt / 1e6, 1), "unit": "MP/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
            "steps_internal": 1, "scaling": plan["scaling"],
            "config": {"workload": LABEL[wl], "mode": MODE_LABEL[args.mode], "images_per_step": n_job // args.steps,
                       "images_per_batch": per_slot, "quality": args.quality, "restart_interval": ri,
                       "parallelism": plan["parallelism"]}}


def run_progressive(jd, R, args):
    """SURVEY row N4 measured like the others: the reference's four progressive files + the 14 libjpeg twins through
    jd_decode_scans_batch (host memory in, host memory out: parse, table packing, upload, kp_scan levels, K2, download),
    at batch sizes 1 / 64 / 1024 / 4096, with the CPU oracle on the same files beside it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    gold = os.path.join(ROOT, "tests", "golden")
    paths = sorted(glob.glob(os.path.join(gold, "ref_test", "p", "*.jpg"))) + sorted(glob.glob(os.path.join(gold, "prog_twins", "*_prog.jpg")))
    files = [open(p, "rb").read() for p in paths]
    px = [f.width * f.height for f in (jd.parse_scans(d)[0] for d in files)]
    if R.dry:
        return {"dry": True, "files": len(files)} if R.rank == 0 else None
    jd.decode_progressive_batch(files[:2], device=R.gpu)             # warm-up (first launch of kp_scan)
    res, ok = {}, True
    for nb in (1, 64, 1024, 4096):
        grp = [files[i % len(files)] for i in range(nb)]
        gpx = sum(px[i % len(files)] for i in range(nb))
        reps = 5 if nb == 1 else (2 if nb == 64 else 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            outs, status = jd.decode_progressive_batch(grp, device=R.gpu)
        dt = (time.perf_counter() - t0) / reps
        ok = ok and all(s == 0 for s in status)
        res[str(nb)] = {"files": nb, "ms_per_call": round(dt * 1e3, 2), "MPs": round(gpx / dt / 1e6, 1)}
    # parity + CPU baseline on every distinct file (oracle: entropy + pixel time of the same files, 1 thread)
    outs, _ = jd.decode_progressive_batch(files, device=R.gpu)
    t_cpu = 0.0
    for d, got in zip(files, outs):
        o = orclib.OracleImage(d)
        t_cpu += o.t_entropy + o.t_pixels
        ok = ok and bool(np.array_equal(got, o.rgb))
    best = max(res.values(), key=lambda r: r["MPs"])
    out = {"value": best["MPs"], "unit": "MP/s", "ms_per_step": best["ms_per_call"], "steps_internal": 1, "scaling": "none",
           "config": {"workload": LABEL["prog"], "mode": "host memory in, host memory out (parse, upload, scans, K2, download)",
                      "distinct_files": len(files), "megapixels_distinct": round(sum(px) / 1e6, 2)},
           "by_batch": res, "verified": ok, "verified_images": len(files),
           "cpu_baseline": {"value": round(sum(px) / t_cpu / 1e6, 3), "unit": "MP/s", "cores": 1, "kind": "port",
                            "sample": "the %d distinct progressive files once each, in-memory hot path only, %.2f s" % (len(files), t_cpu)},
           "single_file_vs_cpu": round(res["1"]["MPs"] / (sum(px) / t_cpu / 1e6), 3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "c3", "c4", "c5", "prog"],
                    help="all: c3 is the headline, c4 / c5 (and prog or band-sharded c3) ride along under `workloads`")
    ap.add_argument("--shard", default="images", choices=["images", "bands"],
                    help="c3: whole images per rank (weak scaling) or one band of every image per rank (strong scaling)")
    ap.add_argument("--mode", default="resident", choices=["resident", "e2e", "host"],
                    help="what the timed steps measure: (i) HBM-resident kernels, (ii) + PCIe both ways, (iii) + header parsing")
    ap.add_argument("--images", type=int, default=0, help="c3/c4: images per GPU per pass (default: c3 16, c4 64; bands: images of the whole job)")
    ap.add_argument("--batch", type=int, default=0, help="images per jd_batch (default: c3/c4 all of the pass's, c5 5000)")
    ap.add_argument("--repeat", type=int, default=0, help="passes per counted step (0: as many as make the timed region about a second)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the rank's batches are dealt over (1: one after another)")
    ap.add_argument("--total", type=int, default=10000, help="c5: images in the whole job, dealt over the ranks")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic images per rank (0 = workload default)")
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--detail", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end sample of the default mode")
    ap.add_argument("--no-extra", action="store_true", help="--workload all: measure the headline only")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the two_streams extra (profiles: its overlapping launches would blur the per-kernel figures)")
    ap.add_argument("--dry", action="store_true", help="no GPU work: launcher, rendezvous (gloo) and planning only")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank decodes on device 0, gloo for the barrier "
                                                             "(RCCL refuses two ranks on one device); the value then measures nothing")
    ap.add_argument("--dry-size", default=None, help="picture size WxH used for c3 / c4 / c5 instead of the workloads' own "
                                                      "(--dry: 256x128 unless given; otherwise a rehearsal, not a measurement)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # not under a launcher: be one (before anything touches the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    if args.dry or args.dry_size:
        dw, dh = (int(v) for v in (args.dry_size or "256x128").split("x"))
        for k in ("c3", "c4", "c5"):
            GEOM[k] = (dw, dh, GEOM[k][2])
            LABEL[k] += " -- REHEARSAL at %dx%d" % (dw, dh)
    R = Rank(args.dry, args.share_gpu)
    import jpeg_decoder_amd as jd
    if not args.dry:
        jd.init([R.gpu])

    headline_wl = "c3" if args.workload == "all" else args.workload
    extra = {}
    if headline_wl == "prog":
        head = run_progressive(jd, R, args) if R.rank == 0 else None
    elif args.mode != "resident":
        head = run_pipeline(jd, R, headline_wl, args)
    else:
        head = run_resident(jd, R, headline_wl, args.shard if headline_wl == "c3" else "images", args, True)
        if args.workload == "all" and not args.no_extra:
            for wl in ("c4", "c5"):
                extra[wl] = run_resident(jd, R, wl, "images", args, False)
            if R.world > 1 and args.shard != "bands":
                extra["c3_bands"] = run_resident(jd, R, "c3", "bands", args, False)
            if R.world == 1:
                extra["prog"] = run_progressive(jd, R, args)
    if R.rank == 0:
        out = {"metric": "decoded megapixels/sec", "value": head.get("value"), "unit": "MP/s", "n_gpus": R.world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": head.get("ms_per_step"), "higher_is_better": True,
               "scaling": head.get("scaling"), "vs_baseline": None, "dtype": DTYPE, "data": "synthetic"}
        out.update({k: v for k, v in head.items() if k not in out})
        if extra:
            out["workloads"] = extra
        print(json.dumps(out), flush=True)
    R.close()


if __name__ == "__main__":
    main()
