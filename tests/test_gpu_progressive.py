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
