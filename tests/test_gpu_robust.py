"""GPU: the C-ABI's other entry forms, mixed batches and damaged streams (error codes, no hangs, no stray writes).
The reference handles every stream error with print + os.Exit(1) (main.go:189-192, 1014-1016); the library returns
a JD_E_* code instead (include/jpeghip.h)."""
import os

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(__file__)
REF = os.path.join(HERE, "golden", "ref_test")
E_MARKER, E_RESTART, E_HUFFMAN, E_COEF, E_TRUNC = -6, -7, -8, -9, -10


def unstuff(ecs):
    """What main.go:993-1021 does to the segment, but keeping where the RSTn markers were."""
    out, rst, i = bytearray(), [], 0
    while i < len(ecs):
        b = ecs[i]
        if b == 0xFF and i + 1 < len(ecs):
            n = ecs[i + 1]
            if n == 0x00:
                out.append(0xFF); i += 2; continue
            if 0xD0 <= n <= 0xD7:
                rst.append(len(out)); i += 2; continue
            if n == 0xFF:
                i += 1; continue
        out.append(b); i += 1
    return bytes(out), rst


def _synth(w, h, hs=2, vs=2, nc=3, ri=0, seed=3, q=85):
    im = jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=seed, detail=2)
    return im, im.encode(ri)


@pytest.mark.parametrize("ri", [0, 5])
def test_unstuffed_entry_form(ri):
    """ecs_is_stuffed = 0: the reference's `_bitstream` (main.go:1011,1018) + restart offsets."""
    im, data = _synth(200, 120, ri=ri)
    fr, off, ln = jd.parse(data)
    bs, rst = unstuff(data[off:off + ln])
    want = orclib.OracleImage(im.encode(0)).rgb
    assert np.array_equal(jd.decode_unstuffed(fr, bs, rst), want), "pre-unstuffed entry form"
    assert np.array_equal(jd.decode_jpeg(data), want), "stuffed entry form"


def test_fill_bytes_before_markers():
    """FF FF .. Dn: extra FF fill bytes in front of a restart marker are dropped (T.81 B.1.1.2)."""
    im, data = _synth(160, 96, ri=4)
    fr, off, ln = jd.parse(data)
    ecs = bytearray(data[off:off + ln])
    i = ecs.index(b"\xff\xd0")
    ecs[i:i] = b"\xff\xff"
    patched = data[:off] + bytes(ecs) + data[off + ln:]
    assert np.array_equal(jd.decode_jpeg(patched), jd.decode_jpeg(data))


def test_mixed_batch_matches_single_decodes():
    """One jd_batch with every geometry, DRI and no-DRI, fixtures and synthetic: same pixels as one-by-one."""
    files = [open(os.path.join(REF, n), "rb").read() for n in ("cat1.jpg", "cat0-h.jpg", "cat0-v.jpg", "cat0-q.jpg")]
    for (w, h, hs, vs, nc, ri) in [(333, 129, 2, 2, 3, 8), (64, 48, 1, 1, 1, 0), (100, 75, 2, 1, 3, 3), (1040, 520, 1, 2, 3, 0),
                                   (17, 9, 1, 1, 3, 1), (640, 360, 2, 2, 3, 0)]:
        files.append(_synth(w, h, hs, vs, nc, ri, seed=w)[1])
    parsed = [jd.parse(f) for f in files]
    b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for i, (f, p) in enumerate(zip(files, parsed)):
        b.upload(i, f[p[1]:p[1] + p[2]])
    for _ in range(2):                                         # a second run on the same batch must give the same
        b.run(); b.wait()
        for i, f in enumerate(files):
            assert b.status(i) == 0
            assert np.array_equal(b.download(i), jd.decode_jpeg(f)), i
            assert np.array_equal(b.coefficients(i), jd.decode_coefficients(f)), i
    b.close()


def test_row_stride_and_device_buffers():
    """bind_ecs / bind_rgb with caller-owned device memory and a padded row stride: pixels land at the stride, the
    padding is untouched.  (Device memory comes straight from the HIP runtime the library itself uses.)"""
    import ctypes
    jd.init()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    _, data = _synth(333, 129, ri=0)
    fr, off, ln = jd.parse(data)
    stride = 333 * 3 + 13
    d_out, d_ecs = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_out), 129 * stride) == 0 and hip.hipMalloc(ctypes.byref(d_ecs), ln + 16) == 0
    assert hip.hipMemset(d_out, 0xA5, 129 * stride) == 0
    ecs = np.frombuffer(data, np.uint8, ln, off).copy()
    assert hip.hipMemcpy(d_ecs, ecs.ctypes.data, ln, 1) == 0
    b = jd.Batch([fr], [ln])
    b.bind_ecs(0, d_ecs.value, ln); b.bind_rgb(0, d_out.value, stride)
    b.run(); b.wait()
    host = np.empty((129, stride), np.uint8)
    assert hip.hipMemcpy(host.ctypes.data, d_out, host.size, 2) == 0
    assert np.array_equal(host[:, :999].reshape(129, 333, 3), jd.decode_jpeg(data))
    assert (host[:, 999:] == 0xA5).all()
    b.close()
    hip.hipFree(d_out); hip.hipFree(d_ecs)


def _expect(data, code):
    """decode through jd_decode with the frame parsed from the file as given (jd_parse only looks for the segment's
    end, so a damaged segment still parses unless the damage looks like a marker to it)"""
    try:
        fr, off, ln = jd.parse(data)
    except jd.JpegHipError as e:
        assert e.code == code, e
        return
    import ctypes
    arr = np.frombuffer(data, np.uint8)
    out = np.empty((fr.height, fr.width, 3), np.uint8)
    rc = jd._lib().jd_decode(ctypes.byref(fr), arr.ctypes.data + off, ln, 1, None, 0, out.ctypes.data, fr.width * 3, 0)
    assert rc == code, rc


@pytest.mark.parametrize("ri", [0, 6])
def test_truncated_stream(ri):
    _, data = _synth(320, 240, ri=ri)
    fr, off, ln = jd.parse(data)
    cut = data[:off + ln // 2] + b"\xff\xd9"
    _expect(cut, E_RESTART if ri else E_TRUNC)               # DRI: markers missing; no DRI: data ends early


def test_marker_inside_segment():
    _, data = _synth(128, 128)
    fr, off, ln = jd.parse(data)
    bad = bytearray(data)
    bad[off + ln // 3:off + ln // 3 + 2] = b"\xff\xc4"       # a DHT marker in the middle of the scan (main.go:1013-1016)
    _expect(bytes(bad), E_MARKER)


def test_restart_markers_out_of_sequence():
    _, data = _synth(256, 128, ri=2)
    bad = bytearray(data)
    i = bad.index(b"\xff\xd3")
    bad[i + 1] = 0xD5
    _expect(bytes(bad), E_RESTART)


def test_bit_errors_never_hang_or_crash():
    """Flip bits all over the entropy-coded segment: every decode returns (ok or an error code), quickly."""
    rng = np.random.default_rng(5)
    for ri in (0, 4):
        _, data = _synth(256, 192, ri=ri, seed=11)
        fr, off, ln = jd.parse(data)
        for trial in range(12):
            bad = bytearray(data)
            for _ in range(1 + trial):
                k = off + int(rng.integers(0, ln))
                if bad[k] != 0xFF and bad[k - 1] != 0xFF:
                    bad[k] ^= 1 << int(rng.integers(0, 8))
                    if bad[k] == 0xFF:
                        bad[k] = 0xFE
            try:
                out = jd.decode_jpeg(bytes(bad))
                assert out.shape == (192, 256, 3)
            except jd.JpegHipError as e:
                assert e.code in (E_RESTART, E_HUFFMAN, E_COEF, E_TRUNC, -11), e
    good = _synth(256, 192, ri=4, seed=11)[1]                  # and the library still decodes correctly afterwards
    assert np.array_equal(jd.decode_jpeg(good), orclib.OracleImage(_synth(256, 192, ri=0, seed=11)[1]).rgb)


def _colour444_jpeg(mcus, q):
    """mcus: list of (Y, Cb, Cr) DC values -> a baseline 4:4:4 JPEG, 8 pixels high, DC-only blocks, every quantiser = q,
    one flat DC and one flat AC table for all components (hand-made: values no encoder produces)"""
    from test_gpu_coef_forms import _codes, _flat_tables
    (dcb, dcv), (acb, acv) = _flat_tables()
    dcc, acc = _codes(dcb, dcv), _codes(acb, acv)
    bits = []

    def put(v, n):
        for i in range(n - 1, -1, -1):
            bits.append((v >> i) & 1)
    pred = [0, 0, 0]
    for m in mcus:
        for c in range(3):
            d = m[c] - pred[c]; pred[c] = m[c]
            n = 0 if d == 0 else int(abs(d)).bit_length()
            put(*dcc[n]); put(d if d >= 0 else d + (1 << n) - 1, n)
            put(*acc[0x00])
    while len(bits) % 8:
        bits.append(1)
    ecs = bytearray()
    for i in range(0, len(bits), 8):
        b = int("".join(map(str, bits[i:i + 8])), 2)
        ecs.append(b)
        if b == 0xFF:
            ecs.append(0)
    w = 8 * len(mcus)
    seg = lambda m, p: bytes([0xFF, m]) + (len(p) + 2).to_bytes(2, "big") + bytes(p)
    out = b"\xFF\xD8" + seg(0xDB, [0] + [q] * 64) + seg(0xC0, [8, 0, 8, w >> 8, w & 255, 3, 1, 0x11, 0, 2, 0x11, 0, 3, 0x11, 0])
    out += seg(0xC4, [0x00] + dcb + dcv) + seg(0xC4, [0x10] + acb + acv) + seg(0xDA, [3, 1, 0x00, 2, 0x00, 3, 0x00, 0, 63, 0])
    return out + bytes(ecs) + b"\xFF\xD9"


@pytest.mark.parametrize("form", [None, 8])
def test_chroma_samples_beyond_int16_are_flagged(form):
    """K2 keeps post-IDCT chroma samples as int16 pairs.  Only a crafted stream leaves that range (an 8-bit encoder's
    samples stay below ~1100); the packing saturates, so the pixels would differ from the reference's: JD_E_COEF, not
    silence (round-2 verdict, weak 9).  DC-only chroma blocks of 1000 x 255 / 8 = 31875 pass and match the oracle;
    1100 x 255 / 8 = 35062 is flagged."""
    from test_gpu_coef_forms import coef_form
    ok = [((i * 7) % 100 - 50, 1000 if i % 9 == 4 else i - 30, -1000 if i % 11 == 6 else 30 - i) for i in range(70)]
    bad = list(ok); bad[37] = (0, 5, -1100)
    with coef_form(form):
        data = _colour444_jpeg(ok, 255)
        assert np.array_equal(jd.decode_jpeg(data), orclib.OracleImage(data).rgb)
        _expect(_colour444_jpeg(bad, 255), -9)


@pytest.mark.parametrize("w,h,hs,vs,nc", [(65535, 16, 2, 2, 3), (16, 65535, 2, 2, 3), (65535, 9, 1, 1, 1), (9, 65535, 1, 2, 3), (65535, 8, 2, 1, 3)])
def test_largest_dimensions(w, h, hs, vs, nc):
    """the 16-bit limits of SOF0's width and height (main.go:117-118), one at a time: a row of 8192 MCUs, a column of 8192;
    with restart markers every MCU row, with none on either path"""
    im = jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=w % 97 + h % 89, detail=2)
    want = orclib.OracleImage(im.encode(0)).rgb
    mcux = (w + 8 * hs - 1) // (8 * hs)
    for ri, fused in ((0, None), (0, 2), (0, 0), (mcux, None), (7, None)):
        old = os.environ.get("JD_NODRI_FUSED")
        try:
            if fused is None: os.environ.pop("JD_NODRI_FUSED", None)
            else: os.environ["JD_NODRI_FUSED"] = str(fused)
            data = im.encode(ri)
            assert np.array_equal(jd.decode_coefficients(data), im.coefficients()), (ri, fused)
            assert np.array_equal(jd.decode_jpeg(data), want), (ri, fused)
        finally:
            if old is None: os.environ.pop("JD_NODRI_FUSED", None)
            else: os.environ["JD_NODRI_FUSED"] = old


def test_largest_restart_interval():
    """DRI = 65535 (the field's limit) on a frame of 82,944 MCUs: two intervals, the second one short"""
    im = jd.SynthImage(2304, 2304, 1, 1, 3, quality=60, seed=3, detail=1)
    data = im.encode(65535)
    assert np.array_equal(jd.decode_coefficients(data), im.coefficients())
    assert np.array_equal(jd.decode_jpeg(data), orclib.OracleImage(im.encode(0)).rgb)
