"""Pins the CPU oracle: SURVEY.md Appendix B hashes for all nine baseline files of the reference's test/
directory (the reference itself ships no goldens -- 'parity unpinned by the reference', see oracle/oracle.h)."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import orclib

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_b.json")))["files"]
REF = os.path.join(os.path.dirname(__file__), "golden", "ref_test")
SMALL = [n for n in GOLD if not n.startswith("cam/2022")]


def _bits(x):
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


def test_idct_constants_bit_patterns():
    # SURVEY Appendix A.2 (main.go:1339-1354)
    names = ["M0", "M1", "M2", "M3", "M4", "M5", "S0", "S1", "S2", "S3", "S4", "S5", "S6", "S7"]
    want = dict(M0=0x3ffd906bcf328d46, M1=0x3ff6a09e667f3bcd, M3=0x3ff6a09e667f3bcd, M5=0x3fe87de2a6aea964,
                M2=0x3ff1517a7bdb3894, M4=0x4004e7ae9144f0fc, S0=0x3fd6a09e667f3bcc, S1=0x3fdf6297cff75cb0,
                S2=0x3fdd906bcf328d46, S4=0x3fd6a09e667f3bcd, S5=0x3fd1c73b39ae68c9, S6=0x3fc87de2a6aea964)
    go = dict(zip(names, orclib.idct_constants(orclib.COS_GO)))
    libm = dict(zip(names, orclib.idct_constants(orclib.COS_LIBM)))
    for k, v in want.items():
        assert _bits(go[k]) == v, k
        assert _bits(libm[k]) == v, k
    assert _bits(go["S3"]) == 0x3fda9b66290ea1a4 and _bits(libm["S3"]) == 0x3fda9b66290ea1a3
    assert _bits(go["S7"]) == 0x3fb8f8b83c69a60c and _bits(libm["S7"]) == 0x3fb8f8b83c69a60d


@pytest.mark.parametrize("name", list(GOLD))
def test_appendix_b(name):
    g = GOLD[name]
    img = orclib.OracleImage(open(os.path.join(REF, name), "rb").read())
    assert (img.width, img.height, img.H, img.V) == (g["width"], g["height"], g["luma_h"], g["luma_v"])
    assert img.bitstream_len == g["bitstream_len"]          # len(bitstream), main.go:1028
    coef, ycc, rgb = img.coef, img.ycc, img.rgb
    assert hashlib.sha256(coef.tobytes()).hexdigest() == g["coef_sha256"]
    assert hashlib.sha256(ycc.tobytes()).hexdigest() == g["ycc_sha256"]
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == g["rgb_sha256"]
    assert int(np.abs(coef).max()) == g["max_abs_coef"]
    assert int(rgb.astype(np.int64).sum()) == g["rgb_byte_sum"]
    h, w = rgb.shape[:2]
    assert [list(map(int, rgb[0, 0])), list(map(int, rgb[h // 2, w // 2])), list(map(int, rgb[h - 1, w - 1]))] == g["px_first_mid_last"]


@pytest.mark.parametrize("name", SMALL)
def test_cos_flavour_is_immaterial(name):
    """S3/S7 differ by 1 ulp between Go's math.Cos and glibc (Appendix A.2): no output sample changes."""
    data = open(os.path.join(REF, name), "rb").read()
    a = orclib.OracleImage(data)
    try:
        orclib.LIB.orc_set_cos_flavour(orclib.COS_LIBM)
        b = orclib.OracleImage(data)
        assert np.array_equal(a.ycc, b.ycc) and np.array_equal(a.rgb, b.rgb)
    finally:
        orclib.LIB.orc_set_cos_flavour(orclib.COS_GO)


@pytest.mark.parametrize("name", ["cat1.jpg", "cat0-q.jpg"])
def test_bmp_layout(name):
    """writeBitMap (main.go:1171-1218): 26-byte header, bottom-up BGR rows, width%4 pad bytes."""
    img = orclib.OracleImage(open(os.path.join(REF, name), "rb").read())
    bmp, rgb = img.bmp(), img.rgb
    w, h, pad = img.width, img.height, img.width % 4
    assert bmp[:2] == b"BM" and struct.unpack_from("<IIII", bmp, 2) == (len(bmp), 0, 26, 12)
    assert struct.unpack_from("<HHHH", bmp, 18) == (w, h, 1, 24)
    assert len(bmp) == 26 + h * (3 * w + pad)
    rows = np.frombuffer(bmp, np.uint8, offset=26).reshape(h, 3 * w + pad)
    assert np.array_equal(rows[::-1, :3 * w].reshape(h, w, 3)[:, :, ::-1], rgb)
    assert not rows[:, 3 * w:].any()


def test_pixels_from_coef_roundtrip():
    img = orclib.OracleImage(open(os.path.join(REF, "cat0-q.jpg"), "rb").read())
    assert np.array_equal(img.pixels_from_coef(img.coef), img.rgb)


def test_errors():
    with pytest.raises(orclib.OracleError):
        orclib.OracleImage(b"\xff\xd8\xff")                  # EOF inside the marker loop (main.go:22-25)
    with pytest.raises(orclib.OracleError):
        orclib.OracleImage(b"\x00\x00\xff\xe0\x00\x02")      # not a JPEG (main.go:1095)
    data = open(os.path.join(REF, "cat1.jpg"), "rb").read()
    with pytest.raises(orclib.OracleError):
        orclib.OracleImage(data[:5000])                      # truncated ECS: EOF before EOI


def test_oracle_against_the_generators_coefficients_under_random_huffman_tables():
    """second pin of the entropy stage, widened: the synthetic encoder keeps the quantised coefficients it
    codes, and the oracle must read them back whatever valid Huffman tables the file carries -- Annex K.3's or random ones
    with codes of 1..16 bits (JDS_RANDOM_TABLES), with and without restart markers, shared table pairs included"""
    import jpeg_decoder_amd as jd
    S = jd.SynthImage
    lens = set()
    for seed in range(60):
        im = S(64 + seed % 50, 40 + seed % 9, *[(1, 1, 1), (1, 1, 3), (2, 1, 3), (1, 2, 3), (2, 2, 3)][seed % 5],
               quality=[30, 85, 98, 100][seed % 4], seed=seed, detail=seed % 3 + 1)
        flags = (S.RANDOM_TABLES | (seed << 8) if seed % 6 else 0) | (S.SHARED_TABLES if seed % 5 == 0 else 0)
        data = im.encode(seed % 3 * 4, flags)
        fr = jd.parse(data)[0]
        for cls in range(2):
            for i in range(2):
                if fr.huff[cls][i].present:
                    lens.update(l + 1 for l in range(16) if fr.huff[cls][i].bits[l])
        o = orclib.OracleImage(data, orclib.MODE_RESTART_AWARE)
        assert np.array_equal(o.scan_order_coef(), im.coefficients()), seed
    assert min(lens) == 1 and max(lens) == 16, lens
