"""GPU: progressive (SOF2) files through jd_parse_scans + jd_decode_scans (kp_scan, then K2) against the CPU oracle:
accumulated coefficients and pixels bit-exact, for the reference's own four progressive files and for the libjpeg-made
twins (whose baseline halves must give the same pixels through the baseline path)."""
import glob
import os

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(__file__)
TWINS = sorted(glob.glob(os.path.join(HERE, "golden", "prog_twins", "*_prog.jpg")))
REF_P = sorted(glob.glob(os.path.join(HERE, "golden", "ref_test", "p", "*.jpg")))


@pytest.mark.parametrize("path", REF_P + TWINS, ids=lambda p: os.path.basename(p))
def test_progressive_against_oracle(path):
    data = open(path, "rb").read()
    o = orclib.OracleImage(data)
    got_c = jd.decode_progressive(data, coefficients=True)
    want_c = o.scan_order_coef()
    bad = np.argwhere(got_c != want_c)
    assert bad.size == 0, "first coefficient mismatch at (mcu, blk, zz) %s: got %d want %d (%d mismatches)" % (
        bad[0], got_c[tuple(bad[0])], want_c[tuple(bad[0])], len(bad))
    assert np.array_equal(jd.decode_progressive(data), o.rgb)
    assert np.array_equal(jd.decode_jpeg(data), o.rgb)                 # decode_jpeg routes SOF2 files here


@pytest.mark.parametrize("path", TWINS[::3], ids=lambda p: os.path.basename(p))
def test_progressive_twin_equals_baseline_path(path):
    prog = open(path, "rb").read()
    base = open(path.replace("_prog.jpg", "_base.jpg"), "rb").read()
    assert np.array_equal(jd.decode_progressive(prog), jd.decode_jpeg(base))
    assert np.array_equal(jd.decode_progressive(prog, coefficients=True), jd.decode_coefficients(base))
    assert np.array_equal(jd.decode_progressive(base), jd.decode_jpeg(base))   # an SOF0 file through the scans entry


def test_progressive_errors():
    data = bytearray(open(REF_P[0], "rb").read())
    fr, scans, n = jd.parse_scans(bytes(data))
    cut = bytes(data[:scans[4].ecs_off + scans[4].ecs_len // 2]) + b"\xff\xd9"   # file ends in the middle of scan 5
    with pytest.raises(jd.JpegHipError) as e:
        jd.decode_progressive(cut)
    assert e.value.code in (-8, -9, -10)
    out = jd.decode_progressive(bytes(data))                                     # still fine afterwards
    assert np.array_equal(out, orclib.OracleImage(bytes(data)).rgb)


def test_progressive_batch():
    """jd_decode_scans_batch: files of different geometry and scan scripts in one call, one of them damaged: every good
    image equals the oracle's pixels, the damaged one reports its own error."""
    paths = REF_P + TWINS[:6] + REF_P[:2]
    files = [open(p, "rb").read() for p in paths]
    fr, scans, n = jd.parse_scans(files[1])
    bad = bytearray(files[1])
    s = scans[5]
    for k in range(s.ecs_off + 100, s.ecs_off + 400):          # wreck a stretch of a refinement scan
        if bad[k] != 0xFF and bad[k - 1] != 0xFF:
            bad[k] ^= 0x5A
            if bad[k] == 0xFF:
                bad[k] = 0x7F
    files.append(bytes(bad))
    outs, status = jd.decode_progressive_batch(files)
    for i, f in enumerate(files[:-1]):
        assert status[i] == 0, (i, status[i])
        assert np.array_equal(outs[i], orclib.OracleImage(f).rgb), paths[i]
    assert status[-1] in (0, -8, -9, -10)                       # damaged data: an error, or garbage pixels, never a hang


def test_progressive_batch_of_many_files():
    """A call with enough pictures that the copy-out is shared over several host threads (32 pictures a thread), some
    with padded output rows (the strided copy), one damaged file in the middle: every good picture equals the
    oracle's, the padding stays the caller's."""
    paths = (REF_P + TWINS) * 8
    paths = paths[:150]
    want = {p: orclib.OracleImage(open(p, "rb").read()).rgb for p in set(paths)}
    files = [open(p, "rb").read() for p in paths]
    fr, scans, n = jd.parse_scans(files[1])
    bad = bytearray(files[1])
    s = scans[5]
    for k in range(s.ecs_off + 100, s.ecs_off + 400):
        if bad[k] != 0xFF and bad[k - 1] != 0xFF:
            bad[k] ^= 0x5A
            if bad[k] == 0xFF:
                bad[k] = 0x7F
    mid = len(files) // 2
    files[mid] = bytes(bad)
    pad = [(0, 5, 0, 64)[i % 4] for i in range(len(files))]
    for _ in range(2):                                          # the second call reuses nothing of the first
        outs, status = jd.decode_progressive_batch(files, row_pad=pad)
        for i, p in enumerate(paths):
            if i == mid:
                assert status[i] in (0, -8, -9, -10)
                continue
            assert status[i] == 0, (i, status[i])
            assert np.array_equal(outs[i], want[p]), (i, p)
