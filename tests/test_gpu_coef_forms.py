"""GPU: the two forms coefficients take between the Huffman writer and K2.

* bytes (write_segments8 / k2_mcu_lanes<.., true>): 64 bytes per block in a low plane, wide blocks (an AC coefficient
  outside int8, |AC 63| > 7, a DC outside 12 bits) marked and completed by a second, per-lane decode into a high plane;
* int16 (write_segments / k2_mcu_lanes<.., false>): images whose quantisers make wide blocks common, uniform-table
  images, progressive files.
jd_batch_create picks the form per image; JD_COEF_FORM=8 / 16 (read when a batch is created) forces it, which is how
these tests drive wide blocks through the byte form and narrow images through the int16 form."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu


@contextmanager
def coef_form(v):
    old = os.environ.get("JD_COEF_FORM")
    if v is None:
        os.environ.pop("JD_COEF_FORM", None)
    else:
        os.environ["JD_COEF_FORM"] = str(v)
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("JD_COEF_FORM", None)
        else:
            os.environ["JD_COEF_FORM"] = old


def _check(data, twin=None):
    o = orclib.OracleImage(twin if twin is not None else data)
    assert np.array_equal(jd.decode_coefficients(data), o.scan_order_coef())
    assert np.array_equal(jd.decode_jpeg(data), o.rgb)
    return o


def _wide_fraction(coef):
    c = coef.reshape(-1, 64).astype(np.int32)
    return float(((np.abs(c[:, 1:63]) > 127).any(1) | (np.abs(c[:, 63]) > 7)).mean())


@pytest.mark.parametrize("w,h,hs,vs,nc,ri", [(640, 360, 2, 2, 3, 0), (640, 360, 2, 2, 3, 5), (333, 129, 2, 1, 3, 0),
                                              (321, 200, 1, 1, 3, 1), (321, 200, 1, 1, 3, 3), (200, 120, 1, 2, 3, 7),
                                              (500, 300, 1, 1, 1, 0), (500, 300, 1, 1, 1, 3), (2048, 1024, 2, 2, 3, 0)])
def test_wide_blocks_through_the_byte_form(w, h, hs, vs, nc, ri):
    """quality 100 (every quantiser 1): most blocks are wide; bpm = 3 or 1 with odd restart intervals also starts
    segments on odd block indices (the half-line stores)"""
    im = jd.SynthImage(w, h, hs, vs, nc, quality=100, seed=w + ri, detail=3)
    assert _wide_fraction(im.coefficients()) > 0.2
    data, twin = im.encode(ri), im.encode(0)
    for form in (8, None, 16):
        with coef_form(form):
            assert np.array_equal(jd.decode_coefficients(data), im.coefficients()), form
            _check(data, twin)


@pytest.mark.parametrize("quality", [90, 95, 98])
def test_some_wide_blocks(quality):
    """a few wide blocks among narrow ones: mixed waves in the writer and in K2"""
    im = jd.SynthImage(1024, 768, 2, 2, 3, quality=quality, seed=quality, detail=3)
    f = _wide_fraction(im.coefficients())
    for ri in (0, 4):
        data = im.encode(ri)
        with coef_form(8):
            assert np.array_equal(jd.decode_coefficients(data), im.coefficients())
            _check(data, im.encode(0))
    print("quality", quality, "wide fraction", f)


def test_narrow_images_through_the_int16_form():
    im = jd.SynthImage(1024, 512, 2, 2, 3, quality=85, seed=3, detail=2)
    assert _wide_fraction(im.coefficients()) == 0.0
    for ri in (0, 8):
        with coef_form(16):
            _check(im.encode(ri), im.encode(0))


def test_reference_photos_forced_bytes():
    """the reference's own baseline photographs (tests/golden/ref_test), whatever their quantisers, through the byte form"""
    root = os.path.join(os.path.dirname(__file__), "golden", "ref_test")
    names = [n for n in sorted(os.listdir(root)) if n.endswith(".jpg")][:6]
    assert names
    with coef_form(8):
        for n in names:
            data = open(os.path.join(root, n), "rb").read()
            _check(data)


def test_both_forms_in_one_batch():
    """a batch that holds byte-form and int16-form images of several sampling modes: one launch per (mode, form)"""
    ims = [jd.SynthImage(640, 360, 2, 2, 3, quality=85, seed=1, detail=2), jd.SynthImage(640, 360, 2, 2, 3, quality=100, seed=2, detail=3),
           jd.SynthImage(320, 200, 1, 1, 3, quality=85, seed=3, detail=2), jd.SynthImage(320, 200, 1, 1, 1, quality=100, seed=4, detail=3),
           jd.SynthImage(330, 130, 2, 1, 3, quality=97, seed=5, detail=3)]
    files = [im.encode(ri) for im, ri in zip(ims, (0, 4, 3, 0, 2))]
    twins = [im.encode(0) for im in ims]
    parsed = [jd.parse(f) for f in files]
    with coef_form(None):
        b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for i, (f, p) in enumerate(zip(files, parsed)):
        b.upload(i, f[p[1]:p[1] + p[2]])
    b.run(); b.wait()
    for i, im in enumerate(ims):
        assert np.array_equal(b.coefficients(i), im.coefficients()), i
        assert np.array_equal(b.download(i), orclib.OracleImage(twins[i]).rgb), i
    b.close()


# ---- hand-made streams: coefficient values no 8-bit encoder produces (DC outside 12 bits, AC categories 11 .. 15) ------
def _flat_tables():
    """DC: categories 0..15 as sixteen 5-bit codes; AC: all 256 run/size symbols, 128 of length 8 then 128 of length 9"""
    dc_bits = [0] * 16; dc_bits[4] = 16
    ac_bits = [0] * 16; ac_bits[7] = 128; ac_bits[8] = 128
    return (dc_bits, list(range(16))), (ac_bits, list(range(256)))


def _codes(bits, vals):
    out, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            out[vals[k]] = (code, ln); code += 1; k += 1
        code <<= 1
    return out


def _grey_jpeg(blocks):
    """blocks: list of 64-entry zig-zag coefficient lists (DC absolute) -> a baseline grey JPEG, 8 pixels high"""
    (dcb, dcv), (acb, acv) = _flat_tables()
    dcc, acc = _codes(dcb, dcv), _codes(acb, acv)
    bits = []

    def put(v, n):
        for i in range(n - 1, -1, -1):
            bits.append((v >> i) & 1)

    def mag(v):
        n = 0 if v == 0 else int(abs(v)).bit_length()
        return n, (v if v >= 0 else v + (1 << n) - 1)

    pred = 0
    for blk in blocks:
        n, m = mag(blk[0] - pred); pred = blk[0]
        put(*dcc[n]); put(m, n)
        run = 0
        for k in range(1, 64):
            if blk[k] == 0:
                run += 1
                continue
            while run > 15:
                put(*acc[0xF0]); run -= 16
            n, m = mag(blk[k])
            put(*acc[(run << 4) | n]); put(m, n); run = 0
        if run:
            put(*acc[0x00])
    while len(bits) % 8:
        bits.append(1)
    ecs = bytearray()
    for i in range(0, len(bits), 8):
        b = int("".join(map(str, bits[i:i + 8])), 2)
        ecs.append(b)
        if b == 0xFF:
            ecs.append(0)
    w = 8 * len(blocks)
    seg = lambda m, p: bytes([0xFF, m]) + (len(p) + 2).to_bytes(2, "big") + bytes(p)
    out = b"\xFF\xD8" + seg(0xDB, [0] + [1] * 64) + seg(0xC0, [8, 0, 8, w >> 8, w & 255, 1, 1, 0x11, 0])
    out += seg(0xC4, [0x00] + dcb + dcv) + seg(0xC4, [0x10] + acb + acv) + seg(0xDA, [1, 1, 0x00, 0, 63, 0])
    return out + bytes(ecs) + b"\xFF\xD9"


def _blocks(spec, n=70):
    """n blocks (more than one wave's lanes see in a step), block i built by spec(i)"""
    out = []
    for i in range(n):
        b = [0] * 64
        b[0] = (i * 37) % 200 - 100
        b[1 + i % 5] = (-1) ** i * (1 + i % 90)
        spec(i, b)
        out.append(b)
    return out


def _put(b, **kv):
    for k, v in kv.items():
        b[int(k[1:])] = v


@pytest.mark.parametrize("name,spec", [
    ("dc_outside_12_bits", lambda i, b: _put(b, k0=(3000 if i % 7 == 3 else -2500 if i % 7 == 5 else b[0]))),
    ("dc_near_16_bits", lambda i, b: _put(b, k0=(32000 if i == 9 else -32000 if i == 40 else b[0]))),
    ("ac63_up_to_12_bits", lambda i, b: _put(b, k63=(0, 7, -7, 8, -8, 127, -128, 300, -1023, 2047, -2048)[i % 11])),
    ("ac_categories_11_to_15", lambda i, b: _put(b, k5=(0, 1500, -3000, 6000, -12000, 32767, -32767)[i % 7],
                                                  k62=(0, 0, 200, -129, 128)[i % 5])),
    ("everything_wide", lambda i, b: _put(b, k0=5000 - 77 * i, k1=-200 - i, k63=-900 + 20 * i, k31=1000)),
])
@pytest.mark.parametrize("form", [8, 16, None])
def test_hand_made_wide_values(name, spec, form):
    blocks = _blocks(spec)
    data = _grey_jpeg(blocks)
    want = np.array(blocks, np.int32).reshape(-1, 1, 64)
    o = orclib.OracleImage(data)
    assert np.array_equal(o.scan_order_coef().reshape(want.shape), want)
    with coef_form(form):
        got = jd.decode_coefficients(data)
        assert np.array_equal(got.reshape(want.shape), want), (name, form)
        assert np.array_equal(jd.decode_jpeg(data), o.rgb), (name, form)


@pytest.mark.parametrize("form,ok", [(8, False), (16, True)])
def test_ac63_beyond_12_bits(form, ok):
    """the byte form's one narrower limit: AC 63 has 12 bits (the reference: int); int16 holds it"""
    data = _grey_jpeg(_blocks(lambda i, b: _put(b, k63=(5000 if i == 33 else 0))))
    with coef_form(form):
        if ok:
            assert np.array_equal(jd.decode_jpeg(data), orclib.OracleImage(data).rgb)
        else:
            with pytest.raises(jd.JpegHipError) as e:
                jd.decode_jpeg(data)
            assert e.value.code == -9 or "coefficient" in str(e.value).lower()
