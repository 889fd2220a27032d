#!/usr/bin/env python3
"""bench.py -- decoded megapixels/s of the MI355X baseline-JPEG hot path (BASELINE.json metric).

One "step" = one pass of the whole hot path (unstuff + restart index, Huffman decode, dequant + IDCT,
chroma spread + colour) over the rank's synthetic JPEGs, whose entropy-coded segments are already resident
in HBM; outputs (RGB8) stay in HBM.  Workloads:

  c3 (default)  8192x8192 4:2:0, Annex-K tables at quality 85, DRI every 8 MCUs; `--batch` images per GPU per step
                (16, all distinct: 474 MB of entropy-coded data, more than the 256 MB Infinity Cache); weak scaling
  c4            the same images without restart markers (self-synchronising Huffman decode); weak scaling
  c5            BASELINE's "batch of 10k 1920x1080 4:2:0": `--total` images (10000; `--distinct` 64 seeds cycled,
                every image with its own copy of its data in HBM) dealt over the ranks by
                jpeg_decoder_amd.shard_images and decoded in per-rank batches of `--batch` (1000); strong scaling

Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks
time); whole images shard with no data-path collective.

Rank 0 prints ONE JSON line.  `value` is metric (i) of SURVEY 8(d): inputs and outputs resident in HBM.
`roofline` is for the dominant stage (device time from hipEvents recorded inside the timed steps, on the stream the
kernels run on); `cpu_baseline` is the CPU oracle (a port of the reference's algorithm, single thread) on a bounded
sample, N=1 only; `verified` says that an image decoded by the timed configuration equals the oracle's pixels;
`end_to_end` (N=1, after the timed region, on a bounded sample) holds metrics (ii) and (iii): pinned host memory ->
H2D -> kernels -> D2H -> pinned host memory with three batches in flight, without and with the host's header parsing.
`--mode e2e|host` makes (ii) / (iii) the timed quantity of the whole run instead.
"""
import argparse
import ctypes
import json
import os
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
         "c5": "C5: batch of synthetic 1920x1080 baseline 4:2:0, no restart markers"}


def make_distinct(workload, seeds, quality, detail):
    """-> {seed: (file bytes, SynthImage)} for the distinct images of this rank (host threads: generation is host-bound)"""
    import jpeg_decoder_amd as jd
    w, h, ri = GEOM[workload]

    def one(seed):
        im = jd.SynthImage(w, h, 2, 2, 3, quality=quality, seed=seed, detail=detail)
        return seed, (im.encode(ri), im)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return dict(ex.map(one, seeds))


def oracle_rgb(synth_image):
    """CPU oracle's pixels for one synthetic image (its no-DRI twin: the reference cannot decode restart markers, F3)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    o = orclib.OracleImage(synth_image.encode(0))
    return o.rgb, o.t_entropy + o.t_pixels


def cpu_baseline(workload, quality, detail, first):
    """CPU oracle (restatement of the reference's algorithm, 1 thread) on a bounded sample of the workload.
    The reference cannot decode restart markers, so c3 is timed on its no-DRI twin.  `first` = hot-path
    seconds already measured on one image by the verification leg."""
    import jpeg_decoder_amd as jd
    w, h, _ = GEOM[workload]
    n = 6 if workload != "c5" else 120
    px, t_hot = w * h, first
    for i in range(1, n):
        _, t = oracle_rgb(jd.SynthImage(w, h, 2, 2, 3, quality=quality, seed=1000 + i, detail=detail))
        t_hot += t                                            # main.go:1032 + 1054-1057 only
        px += w * h
    return {"value": round(px / t_hot / 1e6, 3), "unit": "MP/s", "cores": 1, "cores_available": os.cpu_count(),
            "kind": "port",
            "sample": "%d x %dx%d 4:2:0 q%d no-DRI twin; in-memory hot path only (entropy decode + dequant/IDCT/"
                      "spread/colour), %.1f s on 1 of %d host cores" % (n, w, h, quality, t_hot, os.cpu_count())}


def end_to_end(jd, torch, workload, files, frames, w, h, device, host_parse, passes, per_slot, slots=3, threads=16, want=None):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"])
    ap.add_argument("--mode", default="resident", choices=["resident", "e2e", "host"],
                    help="what the timed steps measure: (i) HBM-resident kernels, (ii) + PCIe both ways, (iii) + header parsing")
    ap.add_argument("--images", type=int, default=16, help="c3/c4: images per GPU per step")
    ap.add_argument("--batch", type=int, default=0, help="images per jd_batch (default: c3/c4 all of the step's, c5 1000)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the rank's batches are dealt over (1: one after another)")
    ap.add_argument("--total", type=int, default=10000, help="c5: images in the whole job, dealt over the ranks")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic images per rank (0 = workload default)")
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--detail", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end sample of the default mode")
    args = ap.parse_args()

    import torch
    import jpeg_decoder_amd as jd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    dev = torch.device("cuda", local_rank)
    wl = args.workload
    w, h, ri = GEOM[wl]

    # ---- the rank's images ---------------------------------------------------------------------------------
    if wl == "c5":
        per_batch = args.batch or 1000
        distinct = args.distinct or 64
        plan = jd.plan_job(args.total, world, rank, per_batch, distinct)   # strong scaling: the job is `total` images
        mine = [g for grp in plan for g, _ in grp]
        seed_of = [s for grp in plan for _, s in grp]
        seeds = sorted(set(seed_of))
    else:
        per_batch = args.batch or args.images
        distinct = min(args.distinct or 16, args.images)
        mine = list(range(args.images))                         # weak scaling: every rank its own images
        seeds = [rank * 100003 + i for i in range(distinct)]
        seed_of = [seeds[i % distinct] for i in mine]
    made = make_distinct(wl, seeds, args.quality, args.detail)
    n_mine = len(mine)

    jd.init([local_rank])
    parsed = {s: jd.parse(made[s][0]) for s in seeds}
    src = {s: torch.from_numpy(np.frombuffer(made[s][0], np.uint8, parsed[s][2], parsed[s][1]).copy()).to(dev) for s in seeds}
    # explicit streams: a NULL handle would mean "each batch's own stream" to jd_batch_run, and batches would overlap
    tstreams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    if args.mode == "resident":
        # every image owns its entropy-coded bytes in HBM (no two images share cache lines) and its RGB output
        batches, keep = [], []
        for b0 in range(0, n_mine, per_batch):
            idx = range(b0, min(b0 + per_batch, n_mine))
            frames = [parsed[seed_of[i]][0] for i in idx]
            lens = [parsed[seed_of[i]][2] for i in idx]
            rgb = torch.empty((len(idx), h, w, 3), dtype=torch.uint8, device=dev)
            b = jd.Batch(frames, lens, device=local_rank, external_io=True)
            for k, i in enumerate(idx):
                e = src[seed_of[i]].clone()
                assert e.data_ptr() % 16 == 0
                keep.append(e)
                b.bind_ecs(k, e.data_ptr(), lens[k])
                b.bind_rgb(k, rgb[k].data_ptr(), w * 3)
            batches.append((b, rgb))
        torch.cuda.synchronize()                               # the copies above ran on torch's default stream

        def step():
            for k, (b, _) in enumerate(batches):
                b.run(tstreams[k % len(tstreams)].cuda_stream)
        for _ in range(args.warmup):
            step()
        for b, _ in batches:
            b.wait() if args.warmup else None
            b.timing(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        stages, nruns = {}, 0
        for b, _ in batches:
            b.wait()                                           # surfaces any decode error
            nruns, st = b.stage_times()
            for k, v in st.items():
                stages[k] = stages.get(k, 0.0) + v
        got0 = batches[0][1][0].cpu().numpy()
        ecs_bytes = sum(parsed[seed_of[i]][2] for i in range(n_mine))
    else:
        host_parse = args.mode == "host"
        per_slot = args.batch or (4 if wl != "c5" else 250)
        files = [made[seed_of[i % n_mine]][0] for i in range(min(n_mine, 3 * per_slot))]
        frames = [parsed[seed_of[0]][0]] * per_slot
        passes = max(1, -(-n_mine // (3 * per_slot))) * args.steps
        barrier()
        mps, n_img, _ = end_to_end(jd, torch, wl, files, frames, w, h, local_rank, host_parse, passes, per_slot)
        dt = n_img * w * h / (mps * 1e6)
        if dist is not None:
            barrier()
        n_mine = n_img // args.steps
        stages, nruns, got0, ecs_bytes = {}, 0, None, 0

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cnt = torch.tensor([n_mine], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt)
        n_job = int(cnt.item())
    else:
        n_job = n_mine

    STAGE_KERNEL = {"unstuff": "k0_unstuff", "huffman_intervals": "k1_write_intervals", "huffman_sync": "k1_sync",
                    "huffman_write": "k1_write_subseq", "idct_colour": "k2_mcu_lanes"}

    def recorded_traffic(kernel):
        """HBM bytes per step moved by `kernel` (all its launches) from the committed PMC passes
        (profiles/r02_traffic.json), if they were taken on this workload and batch; PMC counters cannot be read from
        inside this process."""
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))["workloads"][wl]
            if rec["images_per_step"] != n_mine:
                return None
            return rec["kernels"][kernel]["hbm_bytes_per_step"]
        except (OSError, KeyError, ValueError):
            return None

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": "decoded megapixels/sec", "value": round(n_job * w * h / (dt / args.steps) / 1e6, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "strong" if wl == "c5" else "weak", "vs_baseline": None,
            "dtype": "f64 IDCT / f32 colour / int16 coefficients (bytes between the kernels)", "data": "synthetic",
            "config": {"workload": LABEL[wl], "mode": {"resident": "(i) inputs and outputs resident in HBM",
                                                       "e2e": "(ii) pinned host -> H2D -> kernels -> D2H -> pinned host, 3 batches in flight",
                                                       "host": "(iii) as (ii) + header parsing and ECS staging from whole files"}[args.mode],
                       "images_per_step": n_job, "images_per_gpu_per_step": n_mine, "images_per_batch": per_batch if args.mode == "resident" else None,
                       "streams": len(tstreams), "distinct_images_per_gpu": len(seeds), "quality": args.quality, "restart_interval": ri,
                       "parallelism": "whole images dealt over the GPUs (shard_images), no collective"},
        }
        if args.mode == "resident":
            dom = max((k for k in stages if k != "clear"), key=lambda k: stages[k])
            alg_bytes = ecs_bytes + 3 * w * h * n_mine          # stuffed ECS in + RGB8 out, per step (this rank)
            achieved = alg_bytes / (stages[dom] * 1e-6) / 1e9
            out["config"]["ecs_bytes_per_pixel"] = round(ecs_bytes / (n_mine * w * h), 4)
            out["roofline"] = {"bound": "hbm", "kernel": STAGE_KERNEL.get(dom, dom), "stage": dom, "achieved": round(achieved, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                               "traffic": recorded_traffic(STAGE_KERNEL.get(dom, dom)),
                               "launch_us": round(stages[dom], 1), "algorithmic_bytes_per_launch": alg_bytes,
                               "note": "per step of one rank: all launches of the kernel in a step together",
                               "whole_path_frac": round(alg_bytes / (sum(stages.values()) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "stage_us": {k: round(v, 1) for k, v in stages.items()}, "runs_averaged": nruns}
            # ---- the timed configuration's own output against the oracle (outside the timed region) ----
            want, t_first = oracle_rgb(made[seed_of[0]][1])
            out["verified"] = bool(np.array_equal(got0, want))
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(wl, args.quality, args.detail, t_first)
            if world == 1 and not args.no_e2e:
                for b, _ in batches:
                    b.close()
                del batches, keep
                torch.cuda.empty_cache()
                per_slot = 2 if wl != "c5" else 128
                files = [made[seed_of[i % n_mine]][0] for i in range(3 * per_slot)]
                frames = [parsed[seed_of[0]][0]] * per_slot
                passes = 3 if wl != "c5" else 6
                pc = pcie_rates(torch, dev)
                ii, n_ii, ok_ii = end_to_end(jd, torch, wl, files, frames, w, h, local_rank, False, passes, per_slot, want=want)
                iii, n_iii, ok_iii = end_to_end(jd, torch, wl, files, frames, w, h, local_rank, True, passes, per_slot, want=want)
                roof = pc["d2h"] * 1e9 / 3.0 / 1e6              # MP/s if nothing but the RGB copy-back took time
                out["end_to_end"] = {"device_incl_pcie_MPs": round(ii, 1), "host_incl_parse_MPs": round(iii, 1),
                                     "images_timed": n_ii, "batches_in_flight": 3, "images_per_batch": per_slot,
                                     "pcie_h2d_GBs": round(pc["h2d"], 1), "pcie_d2h_GBs": round(pc["d2h"], 1),
                                     "pcie_roofline_MPs": round(roof, 1), "frac_of_pcie_roofline": round(ii / roof, 3),
                                     "verified": bool(ok_ii and ok_iii)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
