"""GPU: every BASELINE configuration against the CPU oracle at its own geometry.

C1/C2 (the reference's cat0* files) are in test_gpu_parity.py; here: C3/C4 (8192 x 8192 4:2:0, DRI 8 / no DRI) with the
RGB of the whole frame compared with the oracle's, C5 (1920 x 1080 4:2:0 -- blockHeight 135, a ragged last MCU row) as one
jd_batch of distinct images, and the BMP writer fed from GPU-decoded pixels."""
import os

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(__file__)
REF = os.path.join(HERE, "golden", "ref_test")


@pytest.mark.parametrize("nodri_path,long_from", [(2, 1), (2, 0), (0, 1)])
def test_c5_batch_1080p_against_oracle(nodri_path, long_from, monkeypatch):
    """One batch: 64 distinct 1920x1080 4:2:0 no-DRI images plus DRI and 4:2:2 / 4:4:0 frames of the same size; every
    image's coefficients AND pixels bit-exact against the oracle (DRI files: the oracle decodes the no-DRI twin, F3).
    Once through each no-DRI path (JD_NODRI_FUSED: 2 = single decode, 0 = two passes; by size this batch would take the
    second)."""
    monkeypatch.setenv("JD_NODRI_FUSED", str(nodri_path))
    monkeypatch.setenv("JD_DEBUG_FUSED_LONG_FROM", str(long_from))
    specs = [(2, 2, 0, s) for s in range(64)] + [(2, 2, 8, 64), (2, 2, 120, 65), (2, 1, 0, 66), (1, 2, 0, 67), (2, 1, 5, 68)]
    files, twins = [], []
    for hs, vs, ri, seed in specs:
        im = jd.SynthImage(1920, 1080, hs, vs, 3, quality=85, seed=seed, detail=2)
        files.append(im.encode(ri))
        twins.append(im.encode(0) if ri else None)
    parsed = [jd.parse(f) for f in files]
    b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for i, (f, p) in enumerate(zip(files, parsed)):
        b.upload(i, f[p[1]:p[1] + p[2]])
    b.run(); b.wait()
    for i, f in enumerate(files):
        o = orclib.OracleImage(twins[i] if twins[i] is not None else f)
        assert b.status(i) == 0
        assert np.array_equal(b.coefficients(i), o.scan_order_coef()), "coefficients, image %d %s" % (i, specs[i])
        assert np.array_equal(b.download(i), o.rgb), "pixels, image %d %s" % (i, specs[i])
        del o
    b.close()


def test_c3_c4_full_size_against_oracle(monkeypatch):
    """BASELINE size (8192 x 8192, 4:2:0): coefficients equal the encoder's own; the RGB of the DRI-8 file (C3) and of its
    no-DRI twin (C4) both equal the oracle's decode of the twin, all 201 MB of it -- the twin through both no-DRI paths
    (a single image takes the two-pass path by size; the bench's 64-image batches take the single-decode path)."""
    im = jd.SynthImage(8192, 8192, 2, 2, 3, quality=85, seed=0, detail=2)
    dri, twin = im.encode(8), im.encode(0)
    want_c = im.coefficients()
    assert np.array_equal(jd.decode_coefficients(dri), want_c)
    assert np.array_equal(jd.decode_coefficients(twin), want_c)
    monkeypatch.setenv("JD_NODRI_FUSED", "2")
    assert np.array_equal(jd.decode_coefficients(twin), want_c), "C4, single-decode path: coefficients"
    monkeypatch.setenv("JD_DEBUG_FUSED_LONG_FROM", "0")
    assert np.array_equal(jd.decode_coefficients(twin), want_c), "C4, single-decode path, 8192-bit sub-sequences: coefficients"
    monkeypatch.delenv("JD_NODRI_FUSED"); monkeypatch.delenv("JD_DEBUG_FUSED_LONG_FROM")
    del want_c
    want = orclib.OracleImage(twin).rgb
    got = jd.decode_jpeg(dri)
    assert np.array_equal(got, want), "C3: %d pixels differ" % int((got != want).any(axis=2).sum())
    got = jd.decode_jpeg(twin)
    assert np.array_equal(got, want), "C4: %d pixels differ" % int((got != want).any(axis=2).sum())
    monkeypatch.setenv("JD_NODRI_FUSED", "2")
    for long_from in ("1", "0"):
        monkeypatch.setenv("JD_DEBUG_FUSED_LONG_FROM", long_from)
        got = jd.decode_jpeg(twin)
        assert np.array_equal(got, want), "C4, single-decode path: %d pixels differ" % int((got != want).any(axis=2).sum())


@pytest.mark.parametrize("name", ["cat0-q.jpg", "cat1.jpg", "cat0-h.jpg"])
def test_bmp_from_gpu_pixels(name):
    """N1: GPU decode -> jd_bmp_encode is the file writeBitMap would write (main.go:1171-1244), byte for byte (widths
    295 and 800: with and without row padding)."""
    data = open(os.path.join(REF, name), "rb").read()
    assert jd.write_bitmap(jd.decode_jpeg(data)) == orclib.OracleImage(data).bmp()


@pytest.mark.parametrize("flags", [jd.SynthImage.ZERO_BASED_IDS, jd.SynthImage.DQT_16BIT,
                                   jd.SynthImage.ZERO_BASED_IDS | jd.SynthImage.DQT_16BIT])
def test_parser_corner_cases_on_gpu(flags):
    """zero-based component ids (main.go:133-135, 153-157, 961-963) and 16-bit DQT entries cut to their low byte
    (main.go:60-68): same pixels as the plain file, and as the oracle's decode of the odd file itself."""
    im = jd.SynthImage(200, 120, 2, 2, 3, quality=85, seed=4, detail=2)
    odd, plain = im.encode(0, flags), im.encode(0)
    want = orclib.OracleImage(odd).rgb
    assert np.array_equal(want, orclib.OracleImage(plain).rgb)
    assert np.array_equal(jd.decode_jpeg(odd), want)
