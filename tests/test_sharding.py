"""N>1 path on CPU: two gloo ranks shard a list of images the way bench.py / INTEGRATION.md section 4 do
(whole images, round-robin, no data-path collective) and agree on the max-over-ranks time."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import jpeg_decoder_amd as jd
import orclib


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = jd.shard_images(n_images, world, rank)
        # every rank builds ITS images only (seed = global image index) and parses them with the host twin;
        # the CPU oracle stands in for the device here (this test has no GPU) -- it is the checker, the
        # product path is exercised by the gpu-marked tests
        sums = []
        for i in mine:
            data = jd.SynthImage(48, 32, 2, 2, 3, quality=85, seed=i, detail=2).encode(2 if i % 2 else 0)
            fr, off, ln = jd.parse(data)
            assert (fr.width, fr.height) == (48, 32) and fr.restart_interval == (2 if i % 2 else 0)
            mode = orclib.MODE_RESTART_AWARE if i % 2 else orclib.MODE_REFERENCE
            sums.append(int(orclib.OracleImage(data, mode).rgb.astype(np.int64).sum()))
        np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([mine, sums], np.int64))
        # no data-path collective: only the barrier and the max-over-ranks time
        dist.barrier()
        t = jd.max_over_ranks(0.25 * (rank + 1), dist)
        assert abs(t - 0.25 * world) < 1e-12
        counts = torch.tensor([len(mine)], dtype=torch.int64)
        dist.all_reduce(counts)
        assert int(counts.item()) == n_images
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [5, 8])
def test_two_ranks_partition_the_images(tmp_path, n_images):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_images, str(tmp_path)), nprocs=world, join=True)
    seen = {}
    for r in range(world):
        idx, sums = np.load(tmp_path / ("r%d.npy" % r))
        for i, s in zip(idx, sums):
            assert int(i) not in seen
            seen[int(i)] = int(s)
    assert sorted(seen) == list(range(n_images))
    # each image decoded by exactly one rank gives what a single-rank job gives
    for i in (0, n_images - 1):
        data = jd.SynthImage(48, 32, 2, 2, 3, quality=85, seed=i, detail=2).encode(2 if i % 2 else 0)
        mode = orclib.MODE_RESTART_AWARE if i % 2 else orclib.MODE_REFERENCE
        assert seen[i] == int(orclib.OracleImage(data, mode).rgb.astype(np.int64).sum())


def test_shard_rule():
    for n in (0, 1, 7, 10000):
        for world in (1, 2, 4, 8):
            parts = [jd.shard_images(n, world, r) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        jd.shard_images(4, 2, 2)
