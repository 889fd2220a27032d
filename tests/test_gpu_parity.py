"""GPU parity: HIP path (through the C-ABI) vs the CPU oracle.  Bar: coefficients AND RGB bit-exact."""
import json
import os

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(__file__)
REF = os.path.join(HERE, "golden", "ref_test")
GOLD = json.load(open(os.path.join(HERE, "golden", "appendix_b.json")))["files"]


def _cmp(data, mode=orclib.MODE_REFERENCE, twin=None):
    """decode `data` on the GPU; oracle decodes `twin` (or data) on the CPU; compare coef + rgb."""
    o = orclib.OracleImage(twin if twin is not None else data, mode)
    want_c = o.scan_order_coef()
    got_c = jd.decode_coefficients(data)
    assert got_c.shape == want_c.shape
    bad = np.argwhere(got_c != want_c)
    assert bad.size == 0, "first coefficient mismatch at (mcu, blk, zz) %s: got %d want %d (%d mismatches)" % (
        bad[0], got_c[tuple(bad[0])], want_c[tuple(bad[0])], len(bad))
    got = jd.decode_jpeg(data)
    want = o.rgb
    assert got.shape == want.shape
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() == 0, "rgb max abs diff %d, %d pixels differ, first at %s" % (d.max(), (d.max(axis=2) > 0).sum(),
                                                                              np.argwhere(d.max(axis=2) > 0)[0])
    return o


@pytest.mark.parametrize("name", list(GOLD))
def test_reference_fixtures(name):
    """Every baseline image of the reference's test/ directory (configs C1, C2 and the camera files)."""
    _cmp(open(os.path.join(REF, name), "rb").read())


MODES = [(1, 1, 3), (2, 1, 3), (1, 2, 3), (2, 2, 3), (1, 1, 1)]
SIZES = [(8, 8), (16, 16), (17, 9), (64, 48), (100, 75), (333, 129), (1040, 520)]


@pytest.mark.parametrize("hs,vs,nc", MODES)
@pytest.mark.parametrize("w,h", SIZES)
def test_synthetic_no_dri(w, h, hs, vs, nc):
    im = jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=w * 131 + h, detail=2)
    data = im.encode(0)
    assert np.array_equal(jd.decode_coefficients(data), im.coefficients())
    _cmp(data)


@pytest.mark.parametrize("hs,vs,nc", MODES)
@pytest.mark.parametrize("w,h,ri", [(64, 48, 1), (100, 75, 3), (333, 129, 8), (1040, 520, 8), (1040, 520, 65), (16, 16, 7)])
def test_synthetic_dri_twin(w, h, ri, hs, vs, nc):
    """the reference cannot decode DRI files; the oracle for them is the no-DRI twin."""
    im = jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=w + 7 * h + ri, detail=2)
    dri, twin = im.encode(ri), im.encode(0)
    assert np.array_equal(jd.decode_coefficients(dri), im.coefficients())
    _cmp(dri, twin=twin)
    _cmp(dri, mode=orclib.MODE_RESTART_AWARE)      # and against the oracle's restart-aware extension


@pytest.mark.parametrize("q,detail", [(10, 0), (50, 1), (95, 3), (100, 3)])
def test_quality_range(q, detail):
    im = jd.SynthImage(256, 256, 2, 2, 3, quality=q, seed=q, detail=detail)
    _cmp(im.encode(0))
    _cmp(im.encode(4), twin=im.encode(0))
