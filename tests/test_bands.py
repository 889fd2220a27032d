"""Restart-interval-range sharding: jd_split_bands cuts one DRI image into bands of whole MCU rows that
decode independently.  CPU: the cut rules, and that the bands -- each decoded as an image of its own by the CPU oracle in
its restart-aware mode -- tile the picture the oracle gives for the whole file.  Two gloo ranks deal the bands with
shard_bands and write disjoint rows.  The GPU form of the same check is tests/test_gpu_bands.py."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import jpeg_decoder_amd as jd
import orclib


def band_file(file_bytes, ecs_off, band, ecs):
    """a JPEG file of its own for one band: the original headers with the band's height, the band's segment, EOI"""
    head = bytearray(file_bytes[:ecs_off])
    i = head.index(b"\xff\xc0")
    head[i + 5] = band.frame.height >> 8
    head[i + 6] = band.frame.height & 255
    return bytes(head) + ecs[band.ecs_off:band.ecs_off + band.ecs_len].tobytes() + b"\xff\xd9"


def make(w, h, hs, vs, nc, ri, seed=5):
    f = jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=seed, detail=2).encode(ri)
    fr, off, ln = jd.parse(f)
    return f, fr, off, np.frombuffer(f, np.uint8)[off:off + ln]


@pytest.mark.parametrize("geom", [(256, 200, 2, 2, 3, 4), (256, 200, 2, 2, 3, 16), (200, 136, 1, 1, 3, 5), (96, 90, 2, 1, 3, 3),
                                  (128, 100, 1, 1, 1, 8)])
@pytest.mark.parametrize("parts", [1, 2, 3, 7])
def test_bands_tile_the_picture(geom, parts):
    w, h, hs, vs, nc, ri = geom
    f, fr, off, ecs = make(w, h, hs, vs, nc, ri)
    bands = jd.split_bands(fr, ecs, parts)
    assert 1 <= len(bands) <= parts
    whole = orclib.OracleImage(f, orclib.MODE_RESTART_AWARE).rgb
    mcu_h, mx = 8 * vs, (w + 8 * hs - 1) // (8 * hs)
    ipr = mx // ri
    y = 0
    for k, b in enumerate(bands):
        assert b.y0 == y and b.frame.width == w and b.frame.restart_interval == ri
        assert b.y0 % mcu_h == 0 and ((b.y0 // mcu_h) * ipr) % 8 == 0          # every band's first marker is RST0 again
        if k + 1 < len(bands):
            assert bytes(ecs[b.ecs_off + b.ecs_len:b.ecs_off + b.ecs_len + 2]) in [bytes([0xFF, 0xD0 + i]) for i in range(8)]
            assert bands[k + 1].ecs_off == b.ecs_off + b.ecs_len + 2
        part = orclib.OracleImage(band_file(f, off, b, ecs), orclib.MODE_RESTART_AWARE).rgb
        assert part.shape[0] == b.frame.height
        assert np.array_equal(part, whole[y:y + b.frame.height])
        y += b.frame.height
    assert y == h and bands[-1].ecs_off + bands[-1].ecs_len == ecs.size


def test_grey_frame_with_2x2_sampling_byte():
    """A single-component frame has a 1x1 MCU whatever its SOF sampling byte says (main.go:753-767; jd_frame_geometry): the
    bands of a grey DRI file whose SOF says h = v = 2 are cut in 8-pixel rows, not 16 (round-2 advisor finding)."""
    w, h, ri = 128, 100, 8
    f = bytearray(jd.SynthImage(w, h, 1, 1, 1, quality=85, seed=5, detail=2).encode(ri))
    i = f.index(b"\xff\xc0")
    assert f[i + 11] == 0x11
    f[i + 11] = 0x22
    f = bytes(f)
    fr, off, ln = jd.parse(f)
    assert (fr.ncomp, fr.comp[0].h, fr.comp[0].v) == (1, 2, 2) and jd.geometry(fr) == (16, 13, 1)
    ecs = np.frombuffer(f, np.uint8)[off:off + ln]
    whole = orclib.OracleImage(f, orclib.MODE_RESTART_AWARE).rgb
    bands = jd.split_bands(fr, ecs, 3)
    assert [b.y0 for b in bands] == [0, 32, 64] and [b.frame.height for b in bands] == [32, 32, 36]
    for b in bands:
        part = orclib.OracleImage(band_file(f, off, b, ecs), orclib.MODE_RESTART_AWARE).rgb
        assert np.array_equal(part, whole[b.y0:b.y0 + b.frame.height])


def test_cut_rules():
    f, fr, off, ecs = make(256, 64, 2, 2, 3, 0)                                 # no restart markers: nothing to cut at
    with pytest.raises(jd.JpegHipError) as e:
        jd.split_bands(fr, ecs, 2)
    assert e.value.code == -2
    f, fr, off, ecs = make(256, 64, 2, 2, 3, 3)                                 # 16 MCUs per row, DRI 3: rows do not start intervals
    with pytest.raises(jd.JpegHipError) as e:
        jd.split_bands(fr, ecs, 2)
    assert e.value.code == -2
    f, fr, off, ecs = make(256, 400, 2, 2, 3, 16)                               # one interval per row: boundaries every 8 rows
    bands = jd.split_bands(fr, ecs, 64)
    assert len(bands) == 4 and [b.y0 for b in bands] == [0, 128, 256, 384]      # 25 MCU rows -> 4 units of 8 rows
    f, fr, off, ecs = make(256, 64, 2, 2, 3, 4)
    cut = ecs[:ecs.size // 4].copy()                                            # markers missing
    with pytest.raises(jd.JpegHipError) as e:
        jd.split_bands(fr, cut, 4)
    assert e.value.code == -7
    bad = ecs.copy()
    i = bytes(bad).index(b"\xff\xd1")
    bad[i + 1] = 0xD3                                                           # wrongly numbered
    with pytest.raises(jd.JpegHipError) as e:
        jd.split_bands(fr, bad, 4)
    assert e.value.code == -7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f, fr, off, ecs = make(256, 200, 2, 2, 3, 4)
        bands = jd.split_bands(fr, ecs, 5)
        picture = np.zeros((200, 256, 3), np.uint8)
        rows = np.zeros(200, np.int64)
        for k in jd.shard_bands(len(bands), world, rank):                       # this rank's bands; the oracle stands in for the device
            b = bands[k]
            picture[b.y0:b.y0 + b.frame.height] = orclib.OracleImage(band_file(f, off, b, ecs), orclib.MODE_RESTART_AWARE).rgb
            rows[b.y0:b.y0 + b.frame.height] += 1
        np.save(os.path.join(out_dir, "p%d.npy" % rank), picture)
        np.save(os.path.join(out_dir, "r%d.npy" % rank), rows)
        dist.barrier()                                                          # no data-path collective
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_image(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rows = sum(np.load(tmp_path / ("r%d.npy" % r)) for r in range(world))
    assert np.all(rows == 1)                                                    # every pixel row written by exactly one rank
    picture = sum(np.load(tmp_path / ("p%d.npy" % r)).astype(np.int64) for r in range(world)).astype(np.uint8)
    f, fr, off, ecs = make(256, 200, 2, 2, 3, 4)
    assert np.array_equal(picture, orclib.OracleImage(f, orclib.MODE_RESTART_AWARE).rgb)
