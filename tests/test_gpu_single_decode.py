"""GPU: the single-decode path of streams without restart markers (JD_IMG_FUSED: k1f_cold / k1f_write / k1f_check /
k1f_scan / k1f_index, K2 reading coefficients through the block index) against the oracle, and against the two-pass path
it replaces (JD_NODRI_FUSED=0) -- which stays the fallback and must stay correct.

The library picks the path per run by the batch's size (the single-decode path is the throughput path: jd_api.cpp,
refresh()); the images here are small, so the tests force it.
Knobs (read when a batch is created): JD_NODRI_FUSED=0 switches the path off, =2 takes it whatever the size; JD_DEBUG_FUSED_ROUNDS=n enqueues n blind
re-decode rounds instead of 6, so that jd_batch_wait's loop has work; JD_DEBUG_FUSED_LONG_FROM=0 takes the path's long
(8192-bit) sub-sequences whatever the batch's size; JD_DEBUG_FUSED_CAP_PERCENT=p gives the staging p % of
the frame's blocks, so that it overflows and the batch falls back to the two-pass path."""
import contextlib
import os

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def knobs(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _check(data):
    o = orclib.OracleImage(data)
    assert np.array_equal(jd.decode_coefficients(data), o.scan_order_coef())
    assert np.array_equal(jd.decode_jpeg(data), o.rgb)


GEOMETRIES = [(64, 48, 2, 2, 3), (333, 129, 2, 1, 3), (1040, 520, 1, 2, 3), (640, 360, 1, 1, 3), (2048, 1024, 2, 2, 3),
              (1920, 1080, 2, 2, 3), (801, 601, 1, 1, 1), (16, 16, 2, 2, 3), (8, 8, 1, 1, 1)]


@pytest.mark.parametrize("fused,long_from", [(2, 1), (2, 0), (0, 1)])
@pytest.mark.parametrize("w,h,hs,vs,nc", GEOMETRIES)
def test_both_paths_against_the_oracle(w, h, hs, vs, nc, fused, long_from):
    """the single-decode path with sub-sequences of 4096 and of 8192 bits (JD_DEBUG_FUSED_LONG_FROM=0: the long ones
    whatever the batch's size), and the two-pass path"""
    with knobs(JD_NODRI_FUSED=fused, JD_DEBUG_FUSED_LONG_FROM=long_from):
        for q, detail in ((85, 2), (50, 1), (92, 3)):
            _check(jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=w + q, detail=detail).encode(0))


@pytest.mark.parametrize("rounds", [0, 1])
def test_host_loop_finishes_what_blind_rounds_left(rounds):
    """no (or one) blind re-decode round: one sub-sequence in twelve starts from a wrong guess, the last check sees it and
    jd_batch_wait decodes, checks and re-indexes until nothing moves"""
    with knobs(JD_DEBUG_FUSED_ROUNDS=rounds, JD_NODRI_FUSED=2, JD_DEBUG_FUSED_LONG_FROM=rounds):
        for seed, (w, h) in enumerate([(2048, 1024), (1920, 1080), (640, 360)]):
            _check(jd.SynthImage(w, h, 2, 2, 3, quality=90, seed=seed, detail=3).encode(0))


def test_staging_overflow_falls_back_to_the_two_pass_path():
    with knobs(JD_DEBUG_FUSED_CAP_PERCENT=40, JD_NODRI_FUSED=2):
        data = jd.SynthImage(1920, 1080, 2, 2, 3, quality=85, seed=5, detail=2).encode(0)
        _check(data)
        # ... and a batch that fell back stays usable: run it again with other data of the same frames
        ims = [jd.SynthImage(640, 360, 2, 2, 3, quality=85, seed=s, detail=2) for s in (1, 2, 3)]
        files = [im.encode(0) for im in ims]
        parsed = [jd.parse(f) for f in files]
        b = jd.Batch([p[0] for p in parsed], [int(1.5 * max(p[2] for p in parsed))] * 3, keep_coef=True)
        for rnd in range(2):
            order = [0, 1, 2] if rnd == 0 else [2, 0, 1]
            for i, k in enumerate(order):
                b.upload(i, files[k][parsed[k][1]:parsed[k][1] + parsed[k][2]])
            b.run(); b.wait()
            for i, k in enumerate(order):
                o = orclib.OracleImage(files[k])
                assert np.array_equal(b.coefficients(i), o.scan_order_coef()), (rnd, i)
                assert np.array_equal(b.download(i), o.rgb), (rnd, i)
        b.close()


def test_many_small_images_and_reuse_of_a_batch():
    """one workgroup per image, lists of a few sub-sequences; the second run reuses records, chunk tables and lists"""
    ims = [jd.SynthImage(96 + 16 * (s % 5), 64 + 8 * (s % 7), 2, 2, 3, quality=60 + 3 * (s % 11), seed=s, detail=1 + s % 3) for s in range(40)]
    files = [im.encode(0) for im in ims]
    parsed = [jd.parse(f) for f in files]
    with knobs(JD_NODRI_FUSED=2):
        b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for rnd in range(2):
        for i, (f, p) in enumerate(zip(files, parsed)):
            b.upload(i, f[p[1]:p[1] + p[2]])
        b.run(); b.wait()
        for i, f in enumerate(files):
            o = orclib.OracleImage(f)
            assert np.array_equal(b.coefficients(i), o.scan_order_coef()), (rnd, i)
            assert np.array_equal(b.download(i), o.rgb), (rnd, i)
    b.close()


def test_the_path_is_chosen_by_size_and_both_choices_are_right():
    """default knobs: a batch of two images stays on the two-pass path, one with enough stream for the machine (96 images of
    ~2970 sub-sequences: above the rule's 250,000) takes the single-decode path; either way the pictures are the oracle's"""
    im = jd.SynthImage(1920, 1080, 2, 2, 3, quality=92, seed=77, detail=3)
    data = im.encode(0)
    fr, off, ln = jd.parse(data)
    want = orclib.OracleImage(data)
    for n in (2, 96):
        assert n == 2 or n * ((ln * 8 + 4095) // 4096) >= 250000, ln
        b = jd.Batch([fr] * n, [ln] * n, keep_coef=True)
        for i in range(n):
            b.upload(i, data[off:off + ln])
        b.run(); b.wait()
        for i in (0, n // 2, n - 1):
            assert np.array_equal(b.coefficients(i), want.scan_order_coef()), (n, i)
            assert np.array_equal(b.download(i), want.rgb), (n, i)
        b.close()


def test_truncated_and_damaged_streams_report_as_the_two_pass_path_does():
    im = jd.SynthImage(640, 360, 2, 2, 3, quality=85, seed=9, detail=2)
    data = im.encode(0)
    fr, off, ln = jd.parse(data)
    cut = data[:off + ln // 2] + data[off + ln:]
    flip = bytearray(data); flip[off + ln // 3] ^= 0x5A; flip = bytes(flip)
    for bad in (cut, flip):
        codes = []
        for fused in (2, 0):
            with knobs(JD_NODRI_FUSED=fused):
                try:
                    jd.decode_jpeg(bad)
                    codes.append(0)
                except jd.JpegHipError as e:
                    codes.append(e.code)
        assert codes[0] == codes[1], codes
    with pytest.raises(jd.JpegHipError):
        jd.decode_jpeg(cut)


@pytest.mark.parametrize("cap_percent", [0, 35])
def test_pinned_pipeline_on_the_single_decode_path(cap_percent):
    """jd_batch_stage_jpegs / submit / wait with the pictures coming back through the pinned staging: the single-decode
    path, and its fall-back to the two-pass path when the staging overflows -- the copy-back is enqueued again behind
    the second run"""
    env = {"JD_NODRI_FUSED": 2}
    if cap_percent:
        env["JD_DEBUG_FUSED_CAP_PERCENT"] = cap_percent
    ims = [jd.SynthImage(640, 360, 2, 2, 3, quality=85, seed=40 + s, detail=2) for s in range(6)]
    files = [im.encode(0) for im in ims]
    fr, _, _ = jd.parse(files[0])
    cap = max(jd.parse(f)[2] for f in files) + 256
    with knobs(**env):
        b = jd.Batch([fr] * 3, [cap] * 3, pinned_io=True)
    for group in (files[:3], files[3:], files[1:4]):
        b.stage_jpegs(group, threads=2)
        b.submit(); b.wait()
        for i, f in enumerate(group):
            assert np.array_equal(b.host_rgb(i), orclib.OracleImage(f).rgb), i
    b.close()


def test_two_pass_images_survive_the_single_decode_images_host_loop():
    """A batch with both kinds of no-DRI image: int16 ones (quality 98, or all components on one table pair) stay on the
    two-pass path, byte-form ones take the single-decode path.  With no blind rounds only the latter need jd_batch_wait's
    loop -- and the recomputation after it used to run the two-pass images' spine scan a second time over its own output
    (it turns totals into prefixes in place): Error! EOF or wrong pictures for them (found by tests/fuzz_batches.py)."""
    specs = [(640, 360, 2, 2, 3, 98, 0), (640, 360, 2, 2, 3, 85, 0), (8, 200, 1, 1, 3, 100, 0), (1023, 33, 1, 1, 3, 85, jd.SynthImage.SHARED_TABLES),
             (1920, 33, 2, 1, 3, 60, 0), (17, 9, 2, 2, 3, 100, 0), (250, 200, 1, 2, 3, 90, 0), (64, 64, 1, 1, 1, 99, 0)]
    ims = [jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=10 + k, detail=2) for k, (w, h, hs, vs, nc, q, _) in enumerate(specs)]
    files = [im.encode(0, s[6]) for im, s in zip(ims, specs)]
    parsed = [jd.parse(f) for f in files]
    want = [orclib.OracleImage(f) for f in files]
    for rounds in (0, 6):
        with knobs(JD_NODRI_FUSED=2, JD_DEBUG_FUSED_ROUNDS=rounds):
            b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
        for rnd in range(2):
            for i, (f, p) in enumerate(zip(files, parsed)):
                b.upload(i, f[p[1]:p[1] + p[2]])
            b.run(); b.wait()
            for i in range(len(files)):
                assert b.status(i) == 0, (rounds, rnd, i, b.status(i))
                assert np.array_equal(b.coefficients(i), want[i].scan_order_coef()), (rounds, rnd, i)
                assert np.array_equal(b.download(i), want[i].rgb), (rounds, rnd, i)
        b.close()


@pytest.mark.parametrize("rounds", [None, 0])
def test_mixed_batch_at_the_size_where_the_rule_takes_the_path(rounds):
    """default path choice: 96 1080p images (the rule's 250,000 sub-sequences) plus the other kinds in the same batch -- int16
    form (quality 98), one table pair for all components, tiny frames, restart markers, grey; with the blind rounds the
    library plans, and with none (every single-decode image then waits for jd_batch_wait's loop, the others must not care)."""
    big = [jd.SynthImage(1920, 1080, 2, 2, 3, quality=92, seed=300 + k, detail=3) for k in range(3)]
    extra = [((640, 360, 2, 2, 3, 98, 2), 0, 0), ((1023, 200, 1, 1, 3, 85, 2), 0, jd.SynthImage.SHARED_TABLES), ((8, 8, 2, 2, 3, 85, 1), 0, 0),
             ((17, 200, 2, 1, 3, 100, 1), 0, 0), ((640, 360, 2, 2, 3, 85, 2), 8, 0), ((801, 601, 1, 1, 1, 90, 2), 0, 0), ((250, 9, 1, 2, 3, 60, 3), 0, 0)]
    ims = [big[k % 3] for k in range(96)]
    enc = [(0, 0)] * 96
    for k, (g, ri, fl) in enumerate(extra):
        at = 5 + 13 * k
        ims.insert(at, jd.SynthImage(g[0], g[1], g[2], g[3], g[4], quality=g[5], seed=400 + k, detail=g[6])); enc.insert(at, (ri, fl))
    cache = {}
    def file_of(i):
        key = (id(ims[i]), enc[i])
        if key not in cache: cache[key] = ims[i].encode(*enc[i])
        return cache[key]
    files = [file_of(i) for i in range(len(ims))]
    parsed = [jd.parse(f) for f in files]
    assert sum((p[2] * 8 + 4095) // 4096 for p in parsed[:5]) * 19 >= 250000
    with knobs(**({} if rounds is None else {"JD_DEBUG_FUSED_ROUNDS": rounds})):
        b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for i, (f, p) in enumerate(zip(files, parsed)):
        b.upload(i, f[p[1]:p[1] + p[2]])
    b.run(); b.wait()
    coef, rgb = {}, {}
    for i in range(len(ims)):
        key = (id(ims[i]), enc[i])
        assert b.status(i) == 0, (i, b.status(i))
        if i < 40 or key not in coef:
            if key not in coef: coef[key] = ims[i].coefficients()
            assert np.array_equal(b.coefficients(i), coef[key]), i
        if key not in rgb:
            rgb[key] = orclib.OracleImage(ims[i].encode(0, enc[i][1])).rgb
            assert np.array_equal(b.download(i), rgb[key]), i
    assert np.array_equal(b.download(len(ims) - 1), rgb[(id(ims[-1]), enc[-1])])
    b.close()


@pytest.mark.parametrize("long_from", [1, 0])
def test_a_large_image_among_small_ones(long_from):
    """one image with more than 8192 sub-sequences switches the whole batch to the tiled prefix scan (two launches on the
    first decode's grid); the small images, the two-pass ones and the DRI one in the same batch go through it too"""
    specs = [(250, 200, 2, 2, 3, 85, 0, 0), (4096, 4096, 2, 2, 3, 95, 0, 0), (8, 8, 1, 1, 1, 85, 0, 0), (1023, 360, 1, 1, 3, 98, 0, 0),
             (640, 360, 2, 1, 3, 90, 0, jd.SynthImage.SHARED_TABLES), (1920, 1080, 2, 2, 3, 85, 0, 0), (333, 129, 1, 2, 3, 85, 5, 0), (17, 777, 2, 2, 3, 60, 0, 0)]
    ims = [jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=500 + k, detail=3) for k, (w, h, hs, vs, nc, q, _, _) in enumerate(specs)]
    files = [im.encode(s[6], s[7]) for im, s in zip(ims, specs)]
    parsed = [jd.parse(f) for f in files]
    assert (parsed[1][2] * 8 + 8191) // 8192 > 8192, parsed[1][2]
    for rounds in (None, 0):
        kn = {"JD_NODRI_FUSED": 2, "JD_DEBUG_FUSED_LONG_FROM": long_from}
        if rounds is not None: kn["JD_DEBUG_FUSED_ROUNDS"] = rounds
        with knobs(**kn):
            b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
        for i, (f, p) in enumerate(zip(files, parsed)):
            b.upload(i, f[p[1]:p[1] + p[2]])
        b.run(); b.wait()
        for i, im in enumerate(ims):
            assert b.status(i) == 0, (rounds, i, b.status(i))
            assert np.array_equal(b.coefficients(i), im.coefficients()), (rounds, i)
            if rounds is None:
                assert np.array_equal(b.download(i), orclib.OracleImage(im.encode(0, specs[i][7])).rgb), (rounds, i)
        b.close()


def test_one_batch_changes_path_from_run_to_run():
    """the path is chosen per RUN: the same batch of 96 1080p slots decodes short streams (187,000 sub-sequences: two-pass
    path), then long ones (285,000: single-decode path, which lays the coefficient region out differently), then short
    ones again"""
    small = jd.SynthImage(1920, 1080, 2, 2, 3, quality=92, seed=5, detail=1)
    large = jd.SynthImage(1920, 1080, 2, 2, 3, quality=92, seed=6, detail=3)
    fs, fl = small.encode(0), large.encode(0)
    ps, pl = jd.parse(fs), jd.parse(fl)
    assert bytes(ps[0]) == bytes(pl[0])
    n = 96
    assert n * ((ps[2] * 8 + 4095) // 4096) < 250000 <= n * ((pl[2] * 8 + 4095) // 4096)
    want = {0: (small.coefficients(), orclib.OracleImage(fs).rgb), 1: (large.coefficients(), orclib.OracleImage(fl).rgb)}
    b = jd.Batch([ps[0]] * n, [pl[2] + 64] * n, keep_coef=True)
    for rnd, which in enumerate((0, 1, 0, 1)):
        f, p = (fs, ps) if which == 0 else (fl, pl)
        for i in range(n):
            # (the last run mixes the two kinds; its total stays above the rule's threshold)
            g, q, w = (fs, ps, 0) if (rnd == 3 and i % 8 == 0) else (f, p, which)
            b.upload(i, g[q[1]:q[1] + q[2]])
        b.run(); b.wait()
        for i in (0, 1, 47, 88, n - 1):
            w = 0 if (rnd == 3 and i % 8 == 0) else which
            assert b.status(i) == 0
            assert np.array_equal(b.coefficients(i), want[w][0]), (rnd, i)
            assert np.array_equal(b.download(i), want[w][1]), (rnd, i)
    b.close()


@pytest.mark.parametrize("rounds,long_from", [(None, 0), (None, 1), (0, 0), (1, 1)])
def test_redecode_launches_that_share_one_copy_of_the_tables(rounds, long_from):
    """a batch whose single-decode images all use one table set re-decodes with eight waves behind one copy of the tables, a
    wave or two per image (k1f_write's share_w; the library does so from 1200 images, JD_DEBUG_FUSED_SHARE_FROM=1 here):
    images with lists of a few entries and of a few tiles, images without any list (one sub-sequence), a DRI image and a
    two-pass image in between, every round count incl. the host's loop"""
    specs = [(640, 360, 2, 2, 3, 85, 0), (8, 8, 1, 1, 1, 85, 0), (1920, 1080, 2, 2, 3, 92, 0), (250, 200, 1, 2, 3, 60, 0), (640, 360, 2, 2, 3, 98, 0),
             (333, 129, 2, 1, 3, 85, 7), (1023, 777, 1, 1, 3, 85, 0), (17, 9, 2, 2, 3, 85, 0), (801, 601, 1, 1, 1, 90, 0)] * 3
    ims = [jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=700 + k, detail=1 + k % 3) for k, (w, h, hs, vs, nc, q, _) in enumerate(specs)]
    files = [im.encode(s[6]) for im, s in zip(ims, specs)]
    parsed = [jd.parse(f) for f in files]
    kn = {"JD_NODRI_FUSED": 2, "JD_DEBUG_FUSED_SHARE_FROM": 1, "JD_DEBUG_FUSED_LONG_FROM": long_from}
    if rounds is not None: kn["JD_DEBUG_FUSED_ROUNDS"] = rounds
    with knobs(**kn):
        b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for rnd in range(2):
        for i, (f, p) in enumerate(zip(files, parsed)):
            b.upload(i, f[p[1]:p[1] + p[2]])
        b.run(); b.wait()
        for i, im in enumerate(ims):
            assert b.status(i) == 0, (rnd, i, b.status(i))
            assert np.array_equal(b.coefficients(i), im.coefficients()), (rnd, i)
            if rnd == 0 and i < 9:
                assert np.array_equal(b.download(i), orclib.OracleImage(im.encode(0)).rgb), (rnd, i)
    b.close()


@pytest.mark.parametrize("biggest,rounds", [((160, 120), None), ((320, 240), None), ((640, 360), None), ((320, 240), 0)])
def test_first_decode_of_many_small_images_on_one_copy_of_the_tables(biggest, rounds):
    """batches of thumbnails: when no image has more than 256 sub-sequences the FIRST decode, too, runs in workgroups of
    eight waves on one copy of the tables -- one, two or four waves per image by the largest image (here about 17, 66 and 203
    sub-sequences) -- instead of a 512-lane workgroup per image (20,000 x 320x240: 61,600 -> 142,500 MP/s)"""
    sizes = [biggest, (8, 8), (96, 64), (64, 9), biggest, (17, 33), (120, 90), biggest]
    specs = [(w, h) + m for k, (w, h) in enumerate(sizes * 3) for m in [[(2, 2, 3), (1, 1, 3), (2, 1, 3), (1, 1, 1), (1, 2, 3)][k % 5]]]
    ims = [jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=900 + k, detail=1 + k % 3) for k, (w, h, hs, vs, nc) in enumerate(specs)]
    files = [im.encode(0) for im in ims]
    parsed = [jd.parse(f) for f in files]
    kn = {"JD_NODRI_FUSED": 2, "JD_DEBUG_FUSED_SHARE_FROM": 1}
    if rounds is not None: kn["JD_DEBUG_FUSED_ROUNDS"] = rounds
    with knobs(**kn):
        b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for rnd in range(2):
        for i, (f, p) in enumerate(zip(files, parsed)):
            b.upload(i, f[p[1]:p[1] + p[2]])
        b.run(); b.wait()
        for i, im in enumerate(ims):
            assert b.status(i) == 0, (rnd, i, b.status(i))
            assert np.array_equal(b.coefficients(i), im.coefficients()), (rnd, i)
            if rnd == 0:
                assert np.array_equal(b.download(i), orclib.OracleImage(files[i]).rgb), (rnd, i)
    b.close()


def test_many_small_images_on_one_table_set_take_the_path_by_default():
    """default knobs: 640 small images that share their Huffman tables (no image above 256 sub-sequences) -- far below the
    rule's 250,000 sub-sequences, but the class for which the single-decode path runs eight waves per copy of the tables
    and wins from ~600 images (jd_api.cpp, refresh()); 320 of them stay on the two-pass path.  Either way: the oracle's pictures"""
    kinds = [(96, 64, 2, 2, 3), (64, 48, 1, 1, 3), (160, 120, 2, 1, 3), (8, 8, 2, 2, 3), (120, 90, 1, 2, 3), (320, 240, 2, 2, 3), (17, 9, 1, 1, 3), (250, 33, 2, 2, 3)]
    ims = [jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=1200 + k, detail=1 + k % 3) for k, (w, h, hs, vs, nc) in enumerate(kinds)]
    files = [im.encode(0) for im in ims]
    parsed = [jd.parse(f) for f in files]
    assert len({(bytes(p[0].huff[0][0].bits), bytes(p[0].huff[1][1].bits)) for p in parsed}) == 1
    want = [(im.coefficients(), orclib.OracleImage(f).rgb) for im, f in zip(ims, files)]
    for n in (640, 320):
        order = [(7 * i + i // 8) % len(kinds) for i in range(n)]
        b = jd.Batch([parsed[k][0] for k in order], [parsed[k][2] for k in order], keep_coef=True)
        for rnd in range(2):
            for i, k in enumerate(order):
                b.upload(i, files[k][parsed[k][1]:parsed[k][1] + parsed[k][2]])
            b.run(); b.wait()
            for i, k in enumerate(order):
                assert b.status(i) == 0, (n, rnd, i)
                if i % 5 == 0 or i >= n - 8:
                    assert np.array_equal(b.coefficients(i), want[k][0]), (n, rnd, i)
                    assert np.array_equal(b.download(i), want[k][1]), (n, rnd, i)
        b.close()
