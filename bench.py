#!/usr/bin/env python3
"""bench.py -- decoded megapixels/s of the MI355X baseline-JPEG hot path (BASELINE.json metric).

One "step" = one pass of the whole hot path (unstuff + restart index, Huffman decode, dequant + IDCT,
chroma spread + colour) over one batch of synthetic JPEGs whose entropy-coded segments are already resident
in HBM; outputs (RGB8) stay in HBM.  Workloads:

  c3 (default)  8192x8192 4:2:0, Annex-K tables at quality 85, DRI every 8 MCUs, `--batch` images per GPU (16)
  c4            the same images without restart markers (self-synchronising Huffman decode) (16)
  c5            `--batch` 1920x1080 4:2:0 images without restart markers (1024)

Multi-GPU: one process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks
time); every rank decodes its own images -- whole images shard with no data-path collective ("weak").

Rank 0 prints ONE JSON line.  `roofline` is for the dominant stage (its launch time measured with hipEvents
inside the timed steps, on the stream the kernels run on); `cpu_baseline` is the CPU oracle (a port of the
reference's algorithm, single thread) on a bounded sample, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_images(workload, batch, rank, quality, detail):
    import jpeg_decoder_amd as jd
    if workload in ("c3", "c4"):
        w = h = 8192
        ri = 8 if workload == "c3" else 0
    elif workload == "c5":
        w, h, ri = 1920, 1080, 0
    else:
        raise SystemExit("unknown workload " + workload)
    files = []
    distinct = min(batch, 4 if workload != "c5" else 64)      # distinct seeds, cycled (generation is host-bound)
    cache = {}
    for i in range(batch):
        seed = rank * 100003 + (i % distinct)
        if seed not in cache:
            cache[seed] = jd.SynthImage(w, h, 2, 2, 3, quality=quality, seed=seed, detail=detail).encode(ri)
        files.append(cache[seed])
    return w, h, ri, files, distinct


def cpu_baseline(workload, quality, detail):
    """CPU oracle (restatement of the reference's algorithm, 1 thread) on a bounded sample of the workload.
    The reference cannot decode restart markers, so c3 is timed on its no-DRI twin."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    import jpeg_decoder_amd as jd
    if workload in ("c3", "c4"):
        w = h = 8192
        n = 6
    else:
        w, h, n = 1920, 1080, 120
    px, t_hot, t_wall = 0, 0.0, 0.0
    for i in range(n):
        data = jd.SynthImage(w, h, 2, 2, 3, quality=quality, seed=i, detail=detail).encode(0)
        t0 = time.perf_counter()
        o = orclib.OracleImage(data)
        t_wall += time.perf_counter() - t0
        t_hot += o.t_entropy + o.t_pixels                    # main.go:1032 + 1054-1057 only
        px += w * h
        del o
    return {"value": round(px / t_hot / 1e6, 3), "unit": "MP/s", "cores": 1, "kind": "port",
            "sample": "%d x %dx%d 4:2:0 q%d no-DRI twin; in-memory hot path only (entropy decode + dequant/IDCT/"
                      "spread/colour), %.1f s; whole oracle call incl. parse %.1f s" % (n, w, h, quality, t_hot, t_wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"])
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (0 = workload default)")
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--detail", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="verify coefficients of image 0 against the encoder's")
    args = ap.parse_args()

    import torch
    import jpeg_decoder_amd as jd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    batch = args.batch or {"c3": 16, "c4": 16, "c5": 1024}[args.workload]
    w, h, ri, files, distinct = make_images(args.workload, batch, rank, args.quality, args.detail)

    jd.init([local_rank])
    frames, ecs_t, lens = [], [], []
    for data in files:
        fr, off, ln = jd.parse(data)
        frames.append(fr)
        t = torch.from_numpy(np.frombuffer(data, np.uint8, ln, off).copy()).to(dev)   # ECS resident in HBM
        ecs_t.append(t)
        lens.append(ln)
    rgb = torch.empty((batch, h, w, 3), dtype=torch.uint8, device=dev)
    b = jd.Batch(frames, lens, device=local_rank)
    for i in range(batch):
        assert ecs_t[i].data_ptr() % 16 == 0
        b.bind_ecs(i, ecs_t[i].data_ptr(), lens[i])
        b.bind_rgb(i, rgb[i].data_ptr(), w * 3)
    stream = torch.cuda.current_stream().cuda_stream

    for _ in range(args.warmup):
        b.run(stream)
    b.wait() if args.warmup else None
    if args.check:
        got = rgb[0].cpu().numpy()
        b2 = jd.Batch(frames[:1], lens[:1], device=local_rank)
        b2.bind_ecs(0, ecs_t[0].data_ptr(), lens[0])
        b2.run(); b2.wait()
        assert np.array_equal(b2.download(0), got), "batched output differs from single-image output"
        b2.close()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    b.timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.run(stream)
    barrier()
    dt = time.perf_counter() - t0
    b.wait()                                               # surfaces any decode error
    nruns, stages = b.stage_times()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    STAGE_KERNEL = {"unstuff": "k0_unstuff", "huffman_intervals": "k1_write_intervals", "huffman_sync": "k1_sync",
                    "huffman_write": "k1_write_subseq", "idct_colour": "k2_mcu_lanes"}

    def recorded_traffic(kernel):
        """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_g_traffic.json), if they were
        taken on this workload and batch; PMC counters cannot be read from inside this process."""
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "r01_g_traffic.json")))["workloads"][args.workload]
            if rec["batch"] != batch:
                return None
            return rec["kernels"][kernel]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            return None

    if rank == 0:
        px_step = batch * w * h * world
        ms = dt / args.steps * 1e3
        dom = max((k for k in stages if k != "clear"), key=lambda k: stages[k])
        alg_bytes = sum(lens) + 3 * w * h * batch           # stuffed ECS in + RGB8 out, per launch
        achieved = alg_bytes / (stages[dom] * 1e-6) / 1e9
        out = {
            "metric": "decoded megapixels/sec", "value": round(px_step / (dt / args.steps) / 1e6, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 IDCT / f32 colour / int16 coefficients",
            "data": "synthetic",
            "config": {"workload": {"c3": "C3: synthetic 8192x8192 baseline 4:2:0, DRI every 8 MCUs",
                                    "c4": "C4: synthetic 8192x8192 baseline 4:2:0, no restart markers",
                                    "c5": "C5: synthetic 1920x1080 baseline 4:2:0, no restart markers"}[args.workload],
                       "images_per_gpu_per_step": batch, "distinct_images_per_gpu": distinct, "quality": args.quality,
                       "ecs_bytes_per_pixel": round(sum(lens) / (batch * w * h), 4), "restart_interval": ri,
                       "parallelism": "whole images per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": STAGE_KERNEL.get(dom, dom), "stage": dom, "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": recorded_traffic(STAGE_KERNEL.get(dom, dom)),
                         "launch_us": round(stages[dom], 1), "algorithmic_bytes_per_launch": alg_bytes,
                         "whole_path_frac": round(alg_bytes / (sum(stages.values()) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                         "stage_us": {k: round(v, 1) for k, v in stages.items()}, "runs_averaged": nruns},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, args.quality, args.detail)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
