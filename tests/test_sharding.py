"""N>1 path on CPU: two gloo ranks plan and shard a job the way bench.py does for BASELINE config 5 (whole images dealt
by shard_images / plan_job, per-rank jd_batch-sized groups, no data-path collective), build the jd_frame lists a rank
hands to jd_batch_create, and agree on the max-over-ranks time.  The CPU oracle stands in for the device as the checker
(this test has no GPU); the product path is exercised by the gpu-marked tests."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import jpeg_decoder_amd as jd
import orclib

W, H, DISTINCT, PER_BATCH = 48, 32, 5, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _image(seed):
    return jd.SynthImage(W, H, 2, 2, 3, quality=85, seed=seed, detail=2).encode(2 if seed % 2 else 0)


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = jd.plan_job(total, world, rank, PER_BATCH, DISTINCT)          # the call bench.py makes
        rows = []
        for grp in plan:
            assert 1 <= len(grp) <= PER_BATCH
            files = [_image(seed) for _, seed in grp]
            parsed = [jd.parse(f) for f in files]
            frames = (jd.Frame * len(grp))(*[p[0] for p in parsed])          # what jd_batch_create receives
            for k, (g, seed) in enumerate(grp):
                assert seed == g % DISTINCT
                assert jd._lib().jd_frame_check(ctypes.byref(frames[k])) == 0
                assert (frames[k].width, frames[k].height, frames[k].restart_interval) == (W, H, 2 if seed % 2 else 0)
                mode = orclib.MODE_RESTART_AWARE if seed % 2 else orclib.MODE_REFERENCE
                rows.append((g, int(orclib.OracleImage(files[k], mode).rgb.astype(np.int64).sum())))
        np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array(rows, np.int64).reshape(-1, 2))
        # no data-path collective: only the barrier, the max-over-ranks time and the image count of the job
        dist.barrier()
        t = jd.max_over_ranks(0.25 * (rank + 1), dist)
        assert abs(t - 0.25 * world) < 1e-12
        counts = torch.tensor([len(rows)], dtype=torch.int64)
        dist.all_reduce(counts)
        assert int(counts.item()) == total
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [7, 16])
def test_two_ranks_partition_the_job(tmp_path, total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    seen = {}
    for r in range(world):
        for g, s in np.load(tmp_path / ("r%d.npy" % r)):
            assert int(g) not in seen
            seen[int(g)] = int(s)
    assert sorted(seen) == list(range(total))
    # each image decoded by exactly one rank gives what a single-rank job gives
    for g in (0, total - 1):
        seed = g % DISTINCT
        mode = orclib.MODE_RESTART_AWARE if seed % 2 else orclib.MODE_REFERENCE
        assert seen[g] == int(orclib.OracleImage(_image(seed), mode).rgb.astype(np.int64).sum())


def test_shard_and_plan_rules():
    for n in (0, 1, 7, 10000):
        for world in (1, 2, 4, 8):
            parts = [jd.shard_images(n, world, r) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            plans = [jd.plan_job(n, world, r, 1000, 64) for r in range(world)]
            assert sorted(g for pl in plans for grp in pl for g, _ in grp) == list(range(n))
            assert all(len(grp) <= 1000 and all(s == g % 64 for g, s in grp) for pl in plans for grp in pl)
    assert [len(g) for g in jd.plan_job(10000, 8, 0, 1000, 64)] == [1000, 250]
    with pytest.raises(ValueError):
        jd.shard_images(4, 2, 2)
    with pytest.raises(ValueError):
        jd.plan_job(4, 2, 0, 0, 64)
