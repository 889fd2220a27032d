"""Progressive (SOF2) files, SURVEY row N4 -- CPU side: the oracle's SOF2 branch (main.go:241-428, 1035-1062) and the
parser twin jd_parse_scans.

Pin of the oracle's SOF2 branch: tests/golden/prog_twins/ holds the same pixels saved by libjpeg as a baseline and as a
progressive file (tests/golden/make_prog_twins.py); both carry the same quantised coefficients, and the oracle's SOF0
branch is pinned by SURVEY Appendix B -- so the progressive twin must decode to the baseline twin's coefficients and
pixels.  The reference's own four progressive files (test/p/*.jpg) have no counterpart to compare with: for them parity
is unpinned beyond "decodes, and looks like the libjpeg decode" (recorded in SURVEY Appendix B's style below)."""
import ctypes
import glob
import hashlib
import os

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

HERE = os.path.dirname(__file__)
TWINS = sorted(glob.glob(os.path.join(HERE, "golden", "prog_twins", "*_prog.jpg")))
REF_P = sorted(glob.glob(os.path.join(HERE, "golden", "ref_test", "p", "*.jpg")))

# sha256 of the oracle's RGB for the reference's progressive files, recorded when the SOF2 branch was written (the twin
# test above had passed): a regression pin, not a reference-made golden
REF_P_RGB_SHA256 = {
    "cat0-h.jpg": "ca7f2f9dd47cc6bded385b938de97445c7924b3f17d3971bf1784c7364d17f24",
    "cat0-q.jpg": "4bafb048321f2c900b2235d937a9607c8989c2728d4d2a4022d489374bb0745a",
    "cat0-v.jpg": "c8a37903e7c22508f33c799c25a7c1a7b772ce5793d3c1c006949c526fc194f0",
    "huey.jpg": "f4d2f9f935e090039897a6cda0a6fa576e6eed41d2e22203e5916489d4f68103",
}


def test_fixture_sets_present():
    assert len(TWINS) == 14 and len(REF_P) == 4


@pytest.mark.parametrize("path", TWINS, ids=[os.path.basename(p) for p in TWINS])
def test_oracle_progressive_equals_baseline_twin(path):
    prog = orclib.OracleImage(open(path, "rb").read())
    base = orclib.OracleImage(open(path.replace("_prog.jpg", "_base.jpg"), "rb").read())
    assert np.array_equal(prog.coef, base.coef)
    assert np.array_equal(prog.rgb, base.rgb)


@pytest.mark.parametrize("path", REF_P, ids=[os.path.basename(p) for p in REF_P])
def test_oracle_decodes_reference_progressive_files(path):
    o = orclib.OracleImage(open(path, "rb").read())
    assert o.rgb.shape == (o.height, o.width, 3)
    assert 40 < o.rgb.mean() < 220 and o.rgb.std() > 20            # a picture, not a constant or noise-saturated frame
    name = os.path.basename(path)
    if name in REF_P_RGB_SHA256:
        assert hashlib.sha256(o.rgb.tobytes()).hexdigest() == REF_P_RGB_SHA256[name]


@pytest.mark.parametrize("path", REF_P + TWINS[:3], ids=lambda p: os.path.basename(p))
def test_parse_scans(path):
    data = open(path, "rb").read()
    fr, scans, n = jd.parse_scans(data)
    o = orclib.OracleImage(data)
    assert (fr.width, fr.height, fr.ncomp) == (o.width, o.height, o.ncomp)
    assert n >= 2 and all(scans[k].progressive == 1 for k in range(n))
    assert ctypes.sizeof(jd.Scan) == 2240
    total = 0
    for k in range(n):
        s = scans[k]
        assert 1 <= s.ncomp <= fr.ncomp and s.ss <= s.se <= 63
        end = s.ecs_off + s.ecs_len
        assert data[end] == 0xFF and data[end + 1] in (0xD9, 0xC4, 0xDA, 0xDD)   # what ends a scan (main.go:1001-1008)
        total += s.ecs_len
    assert total < len(data)
    with pytest.raises(jd.JpegHipError) as e:
        jd.parse(data)                                              # the baseline parser leaves SOF2 alone
    assert e.value.code == -2


def test_parse_scans_on_a_baseline_file():
    data = open(os.path.join(HERE, "golden", "ref_test", "cat1.jpg"), "rb").read()
    fr, scans, n = jd.parse_scans(data)
    f0, off, ln = jd.parse(data)
    assert n == 1 and scans[0].progressive == 0 and (scans[0].ecs_off, scans[0].ecs_len) == (off, ln)
    assert bytes(fr) == bytes(f0)
