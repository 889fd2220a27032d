"""jpeg-decoder_amd -- Python plumbing over the C-ABI of libjpeghip.so (include/jpeghip.h).

The product is the HIP library; this module only marshals arguments with ctypes for the tests and
bench.py.  There is NO CPU fallback: importing works without a GPU (so the symbol checks can run), but every
decode call goes through the HIP kernels and raises if the library or a device is missing.

Function names mirror the reference's Go host (decodeJPEG -> decode_jpeg, writeBitMap -> write_bitmap).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# JPEGHIP_LIB: experiments only (A/B builds of the same HIP library); the default is the in-tree build
LIB_PATH = os.environ.get("JPEGHIP_LIB") or os.path.join(_HERE, "libjpeghip.so")
SYNTH_PATH = os.path.join(_HERE, "libjdsynth.so")

_u8, _u16, _u32 = ctypes.c_uint8, ctypes.c_uint16, ctypes.c_uint32
_vp, _ci, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


class Component(ctypes.Structure):
    _fields_ = [("h", _u8), ("v", _u8), ("tq", _u8), ("td", _u8), ("ta", _u8)]


class Huff(ctypes.Structure):
    _fields_ = [("bits", _u8 * 16), ("vals", _u8 * 256), ("nvals", _u16), ("present", _u8)]


class Frame(ctypes.Structure):
    """jd_frame: what the Go Header holds at cut 1 (main.go:1467-1491)."""
    _fields_ = [("width", _u16), ("height", _u16), ("ncomp", _u8), ("comp", Component * 3),
                ("restart_interval", _u16), ("qt", (_u8 * 64) * 4), ("qt_present", _u8),
                ("huff", (Huff * 4) * 2)]


class Scan(ctypes.Structure):
    """jd_scan: one SOS of a progressive file -- scan header, tables in force, extent of its entropy-coded bytes."""
    _fields_ = [("progressive", _u8), ("ncomp", _u8), ("comp", _u8 * 3), ("td", _u8 * 3), ("ta", _u8 * 3),
                ("ss", _u8), ("se", _u8), ("ah", _u8), ("al", _u8), ("ecs_off", ctypes.c_uint64), ("ecs_len", ctypes.c_uint64),
                ("huff", (Huff * 4) * 2)]


class Band(ctypes.Structure):
    """jd_band: a run of whole MCU rows of a DRI image that decodes on its own (jd_split_bands)."""
    _fields_ = [("frame", Frame), ("ecs_off", ctypes.c_size_t), ("ecs_len", ctypes.c_size_t), ("y0", ctypes.c_uint32)]


# every symbol include/jpeghip.h declares (tests/test_abi.py checks the library exports exactly these)
ABI_SYMBOLS = [
    "jd_strerror", "jd_last_hip_error", "jd_init", "jd_shutdown", "jd_device_count", "jd_frame_check",
    "jd_frame_geometry", "jd_decode", "jd_decode_coefficients", "jd_batch_create", "jd_batch_destroy",
    "jd_batch_upload", "jd_batch_upload_unstuffed", "jd_batch_bind_ecs", "jd_batch_bind_rgb", "jd_batch_run",
    "jd_batch_wait", "jd_batch_status", "jd_batch_download", "jd_batch_coefficients", "jd_batch_profile",
    "jd_batch_timing", "jd_batch_stage_times", "jd_batch_workspace_bytes", "jd_batch_unstuffed_len", "jd_parse", "jd_bmp_encode",
    "jd_batch_stage_reserve", "jd_batch_stage_ecs", "jd_batch_stage_jpegs", "jd_batch_submit", "jd_batch_host_rgb",
    "jd_host_alloc", "jd_host_free", "jd_decode_resident", "jd_parse_scans", "jd_decode_scans", "jd_decode_scans_coefficients",
    "jd_decode_scans_batch", "jd_split_bands",
]


class JpegHipError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = _lib().jd_strerror(code).decode() if _LIB is not None else "library not loaded"
        super().__init__("%s: %s (code %d, hip %d)" % (where, msg, code, _lib().jd_last_hip_error()))


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libjpeghip.so is not built (run __graft_entry__.build() or make -C "
                               "jpeg-decoder_amd/csrc); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        PF = ctypes.POINTER(Frame)
        L.jd_strerror.restype = ctypes.c_char_p
        L.jd_strerror.argtypes = [_ci]
        L.jd_init.argtypes = [ctypes.POINTER(_ci), _ci]
        L.jd_frame_check.argtypes = [PF]
        L.jd_frame_geometry.argtypes = [PF, ctypes.POINTER(_u32), ctypes.POINTER(_u32), ctypes.POINTER(_u32)]
        L.jd_decode.argtypes = [PF, _vp, _sz, _ci, _vp, _sz, _vp, _sz, _ci]
        L.jd_decode_coefficients.argtypes = [PF, _vp, _sz, _ci, _vp, _sz, _vp, _sz, _ci]
        L.jd_batch_create.argtypes = [_ci, _ci, PF, ctypes.POINTER(_sz), ctypes.c_uint, ctypes.POINTER(_vp)]
        L.jd_batch_destroy.argtypes = [_vp]
        L.jd_batch_destroy.restype = None
        L.jd_batch_upload.argtypes = [_vp, _ci, _vp, _sz]
        L.jd_batch_upload_unstuffed.argtypes = [_vp, _ci, _vp, _sz, _vp, _sz]
        L.jd_batch_bind_ecs.argtypes = [_vp, _ci, _vp, _sz]
        L.jd_batch_bind_rgb.argtypes = [_vp, _ci, _vp, _sz]
        L.jd_batch_run.argtypes = [_vp, _vp]
        L.jd_batch_wait.argtypes = [_vp]
        L.jd_batch_status.argtypes = [_vp, _ci]
        L.jd_batch_download.argtypes = [_vp, _ci, _vp, _sz]
        L.jd_batch_coefficients.argtypes = [_vp, _ci, _vp, _sz]
        L.jd_batch_profile.argtypes = [_vp, _ci, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p), _ci]
        L.jd_batch_timing.argtypes = [_vp, _ci]
        L.jd_batch_stage_times.argtypes = [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_char_p), _ci]
        L.jd_batch_workspace_bytes.argtypes = [_vp]
        L.jd_batch_workspace_bytes.restype = _sz
        L.jd_batch_unstuffed_len.argtypes = [_vp, _ci, ctypes.POINTER(_sz)]
        L.jd_batch_stage_reserve.argtypes = [_vp, _ci, _sz, ctypes.POINTER(_vp)]
        L.jd_batch_stage_ecs.argtypes = [_vp, _ci, _vp, _sz]
        L.jd_batch_stage_jpegs.argtypes = [_vp, _ci, _ci, ctypes.POINTER(_vp), ctypes.POINTER(_sz), _ci]
        L.jd_batch_submit.argtypes = [_vp, _vp]
        L.jd_batch_host_rgb.argtypes = [_vp, _ci, ctypes.POINTER(_vp), ctypes.POINTER(_sz)]
        L.jd_decode_resident.argtypes = [PF, _vp, _sz, _vp, _sz, _ci, _vp]
        PS = ctypes.POINTER(Scan)
        L.jd_parse_scans.argtypes = [_vp, _sz, PF, PS, _ci, ctypes.POINTER(_ci)]
        L.jd_decode_scans.argtypes = [PF, PS, _ci, _vp, _sz, _vp, _sz, _ci]
        L.jd_decode_scans_coefficients.argtypes = [PF, PS, _ci, _vp, _sz, _vp, _sz, _ci]
        L.jd_decode_scans_batch.argtypes = [_ci, PF, ctypes.POINTER(PS), ctypes.POINTER(_ci), ctypes.POINTER(_vp),
                                            ctypes.POINTER(_sz), ctypes.POINTER(_vp), ctypes.POINTER(_sz), ctypes.POINTER(_ci), _ci]
        L.jd_host_alloc.argtypes = [_sz, ctypes.POINTER(_vp)]
        L.jd_host_free.argtypes = [_vp]
        L.jd_host_free.restype = None
        L.jd_parse.argtypes = [_vp, _sz, PF, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]
        L.jd_bmp_encode.argtypes = [_vp, _sz, _ci, _ci, _vp, _sz]
        L.jd_bmp_encode.restype = ctypes.c_long
        L.jd_split_bands.argtypes = [PF, _vp, _sz, _ci, ctypes.POINTER(Band), ctypes.POINTER(_ci)]
        _LIB = L
    return _LIB


def _check(rc, where):
    if rc != 0:
        raise JpegHipError(rc, where)


_inited = False


def init(devices=None):
    """jd_init: bind the library to HIP devices (all visible ones by default)."""
    global _inited
    if devices is None:
        rc = _lib().jd_init(None, 0)
    else:
        arr = (_ci * len(devices))(*devices)
        rc = _lib().jd_init(arr, len(devices))
    _check(rc, "jd_init")
    _inited = True


def _ensure_init():
    if not _inited:
        init()


def _as_u8(data):
    a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    return a


def parse(data):
    """jd_parse: twin of the Go marker parser (decodeJPEG..., main.go:1076-1154).  -> (Frame, ecs_off, ecs_len)"""
    a = _as_u8(data)
    fr, off, ln = Frame(), _sz(), _sz()
    _check(_lib().jd_parse(a.ctypes.data, a.size, ctypes.byref(fr), ctypes.byref(off), ctypes.byref(ln)), "jd_parse")
    return fr, off.value, ln.value


def split_bands(frame, ecs, parts):
    """jd_split_bands: restart-interval-range sharding: cut a DRI image's stuffed segment into at most `parts`
    bands of whole MCU rows.  -> [Band]; band k decodes like an image (frame = band.frame, segment =
    ecs[band.ecs_off : band.ecs_off + band.ecs_len]) into pixel rows band.y0 .. band.y0 + band.frame.height."""
    a = _as_u8(ecs)
    bands, n = (Band * parts)(), _ci()
    _check(_lib().jd_split_bands(ctypes.byref(frame), a.ctypes.data, a.size, parts, bands, ctypes.byref(n)), "jd_split_bands")
    return [bands[i] for i in range(n.value)]


def parse_scans(data, cap=64):
    """jd_parse_scans: frame + one Scan per SOS of an SOF0 or SOF2 file (main.go:942-1021, 1035-1052).  -> (Frame, [Scan])"""
    a = _as_u8(data)
    fr, scans, n = Frame(), (Scan * cap)(), _ci()
    _check(_lib().jd_parse_scans(a.ctypes.data, a.size, ctypes.byref(fr), scans, cap, ctypes.byref(n)), "jd_parse_scans")
    return fr, scans, n.value


def geometry(frame):
    mx, my, bpm = _u32(), _u32(), _u32()
    _check(_lib().jd_frame_geometry(ctypes.byref(frame), ctypes.byref(mx), ctypes.byref(my), ctypes.byref(bpm)),
           "jd_frame_geometry")
    return mx.value, my.value, bpm.value


def decode_progressive(data, device=0, coefficients=False):
    """jd_parse_scans + jd_decode_scans: an SOF2 (or SOF0) file scan by scan.  -> RGB, or the accumulated coefficients."""
    _ensure_init()
    a = _as_u8(data)
    fr, scans, n = parse_scans(a)
    if coefficients:
        mx, my, bpm = geometry(fr)
        out = np.empty((mx * my, bpm, 64), np.int16)
        _check(_lib().jd_decode_scans_coefficients(ctypes.byref(fr), scans, n, a.ctypes.data, a.size, out.ctypes.data, out.size,
                                                   device), "jd_decode_scans_coefficients")
        return out
    out = np.empty((fr.height, fr.width, 3), np.uint8)
    _check(_lib().jd_decode_scans(ctypes.byref(fr), scans, n, a.ctypes.data, a.size, out.ctypes.data, fr.width * 3, device),
           "jd_decode_scans")
    return out


def decode_progressive_batch(files, device=0, row_pad=None):
    """jd_decode_scans_batch: many progressive files in one call.  -> (list of RGB arrays, list of per-image status codes).
    row_pad[i] extra bytes end every output row of picture i (the padding is returned untouched: the caller's bytes)."""
    _ensure_init()
    n = len(files)
    arrs = [_as_u8(f) for f in files]
    parsed = [parse_scans(a) for a in arrs]
    frames = (Frame * n)(*[p[0] for p in parsed])
    PS = ctypes.POINTER(Scan)
    scans = (PS * n)(*[ctypes.cast(p[1], PS) for p in parsed])
    counts = (_ci * n)(*[p[2] for p in parsed])
    ptrs = (_vp * n)(*[a.ctypes.data for a in arrs])
    lens = (_sz * n)(*[a.size for a in arrs])
    pad = list(row_pad) if row_pad is not None else [0] * n
    bufs = [np.full((p[0].height, p[0].width * 3 + pad[i]), 0xA5, np.uint8) if pad[i] else
            np.empty((p[0].height, p[0].width * 3), np.uint8) for i, p in enumerate(parsed)]
    optrs = (_vp * n)(*[o.ctypes.data for o in bufs])
    strides = (_sz * n)(*[o.shape[1] for o in bufs]) if row_pad is not None else None
    status = (_ci * n)()
    rc = _lib().jd_decode_scans_batch(n, frames, scans, counts, ptrs, lens, optrs, strides, status, device)
    if rc != 0 and all(s == 0 for s in status):
        raise JpegHipError(rc, "jd_decode_scans_batch")
    outs = []
    for i, (o, p) in enumerate(zip(bufs, parsed)):
        if pad[i] and not (o[:, p[0].width * 3:] == 0xA5).all():
            raise AssertionError("jd_decode_scans_batch wrote into the row padding of picture %d" % i)
        outs.append(o[:, :p[0].width * 3].reshape(p[0].height, p[0].width, 3))
    return outs, list(status)


def decode_jpeg(data, device=0):
    """Parse + jd_decode: the whole of decodeJPEG up to (not including) writeBitMap.  -> H x W x 3 uint8 RGB.
    Progressive files (which jd_parse leaves to jd_parse_scans) go through decode_progressive."""
    _ensure_init()
    a = _as_u8(data)
    try:
        fr, off, ln = parse(a)
    except JpegHipError as e:
        if e.code != -2:
            raise
        return decode_progressive(a, device)
    out = np.empty((fr.height, fr.width, 3), np.uint8)
    _check(_lib().jd_decode(ctypes.byref(fr), a.ctypes.data + off, ln, 1, None, 0, out.ctypes.data, fr.width * 3, device),
           "jd_decode")
    return out


def decode_unstuffed(frame, bitstream, rst_offsets=(), device=0):
    """jd_decode with the reference's `_bitstream` (main.go:1011,1018) + restart offsets."""
    _ensure_init()
    a = _as_u8(bitstream)
    r = np.ascontiguousarray(rst_offsets, np.uint32)
    out = np.empty((frame.height, frame.width, 3), np.uint8)
    _check(_lib().jd_decode(ctypes.byref(frame), a.ctypes.data, a.size, 0, r.ctypes.data if r.size else None, r.size,
                            out.ctypes.data, frame.width * 3, device), "jd_decode")
    return out


def decode_coefficients(data, device=0):
    """jd_decode_coefficients: int16 [mcu][block-in-MCU][64], zig-zag order, absolute DC (cut 1 parity hook)."""
    _ensure_init()
    a = _as_u8(data)
    fr, off, ln = parse(a)
    mx, my, bpm = geometry(fr)
    out = np.empty((mx * my, bpm, 64), np.int16)
    _check(_lib().jd_decode_coefficients(ctypes.byref(fr), a.ctypes.data + off, ln, 1, None, 0, out.ctypes.data, out.size,
                                         device), "jd_decode_coefficients")
    return out


def decode_resident(frame, d_ecs, ecs_len, d_rgb, row_stride, device=0, stream=None):
    """jd_decode_resident: device-resident in and out, workspace = the library's fixed scratch, no allocation."""
    _ensure_init()
    _check(_lib().jd_decode_resident(ctypes.byref(frame), d_ecs, ecs_len, d_rgb, row_stride, device, stream),
           "jd_decode_resident")


def write_bitmap(rgb):
    """jd_bmp_encode: the bytes writeBitMap (main.go:1171-1218) would put in <name>.bmp."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    n = _lib().jd_bmp_encode(None, 0, w, h, None, 0)
    out = np.empty(n, np.uint8)
    got = _lib().jd_bmp_encode(rgb.ctypes.data, w * 3, w, h, out.ctypes.data, n)
    if got != n:
        raise JpegHipError(int(got), "jd_bmp_encode")
    return out.tobytes()


class Batch:
    """jd_batch: n images decoded by one fixed sequence of batched kernels on one device."""

    def __init__(self, frames, ecs_caps, device=0, keep_coef=False, pinned_io=False, external_io=False):
        _ensure_init()
        n = len(frames)
        self.n, self.device = n, device
        self.frames = (Frame * n)(*frames)
        caps = (_sz * n)(*[int(c) for c in ecs_caps])
        h = _vp()
        flags = (1 if keep_coef else 0) | (2 if pinned_io else 0) | (4 if external_io else 0)
        _check(_lib().jd_batch_create(device, n, self.frames, caps, flags, ctypes.byref(h)), "jd_batch_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            _lib().jd_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def upload(self, i, ecs):
        a = _as_u8(ecs)
        _check(_lib().jd_batch_upload(self._h, i, a.ctypes.data, a.size), "jd_batch_upload")

    def upload_unstuffed(self, i, bitstream, rst_offsets=()):
        a = _as_u8(bitstream)
        r = np.ascontiguousarray(rst_offsets, np.uint32)
        _check(_lib().jd_batch_upload_unstuffed(self._h, i, a.ctypes.data, a.size, r.ctypes.data if r.size else None, r.size),
               "jd_batch_upload_unstuffed")

    def bind_ecs(self, i, dev_ptr, length):
        _check(_lib().jd_batch_bind_ecs(self._h, i, dev_ptr, length), "jd_batch_bind_ecs")

    def bind_rgb(self, i, dev_ptr, row_stride):
        _check(_lib().jd_batch_bind_rgb(self._h, i, dev_ptr, row_stride), "jd_batch_bind_rgb")

    # ---- pinned-memory pipeline (JD_BATCH_PINNED_IO) ----
    def stage_ecs(self, i, ecs):
        a = _as_u8(ecs)
        _check(_lib().jd_batch_stage_ecs(self._h, i, a.ctypes.data, a.size), "jd_batch_stage_ecs")

    def stage_jpegs(self, files, first=0, threads=8):
        """files: list of bytes / uint8 arrays holding whole JPEG files (kept alive by the caller during the call)"""
        arrs = [_as_u8(f) for f in files]
        ptrs = (_vp * len(arrs))(*[a.ctypes.data for a in arrs])
        lens = (_sz * len(arrs))(*[a.size for a in arrs])
        _check(_lib().jd_batch_stage_jpegs(self._h, first, len(arrs), ptrs, lens, threads), "jd_batch_stage_jpegs")

    def stage_jpegs_raw(self, ptrs, lens, count, first=0, threads=8):
        """the same with prepared ctypes arrays of file pointers / lengths (bench.py's timed host loop)"""
        _check(_lib().jd_batch_stage_jpegs(self._h, first, count, ptrs, lens, threads), "jd_batch_stage_jpegs")

    def submit(self, stream=None):
        _check(_lib().jd_batch_submit(self._h, stream), "jd_batch_submit")

    def host_rgb(self, i):
        """-> H x W x 3 uint8 view of the pinned RGB staging (valid after wait(), until the next submit)"""
        p, st = _vp(), _sz()
        _check(_lib().jd_batch_host_rgb(self._h, i, ctypes.byref(p), ctypes.byref(st)), "jd_batch_host_rgb")
        f = self.frames[i]
        buf = (ctypes.c_uint8 * (st.value * f.height)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(f.height, f.width, 3)

    def run(self, stream=None):
        _check(_lib().jd_batch_run(self._h, stream), "jd_batch_run")

    def wait(self):
        _check(_lib().jd_batch_wait(self._h), "jd_batch_wait")

    def status(self, i):
        return _lib().jd_batch_status(self._h, i)

    def download(self, i):
        f = self.frames[i]
        out = np.empty((f.height, f.width, 3), np.uint8)
        _check(_lib().jd_batch_download(self._h, i, out.ctypes.data, f.width * 3), "jd_batch_download")
        return out

    def coefficients(self, i):
        mx, my, bpm = geometry(self.frames[i])
        out = np.empty((mx * my, bpm, 64), np.int16)
        _check(_lib().jd_batch_coefficients(self._h, i, out.ctypes.data, out.size), "jd_batch_coefficients")
        return out

    def profile(self, iters=5):
        us = (ctypes.c_float * 16)()
        names = (ctypes.c_char_p * 16)()
        k = _lib().jd_batch_profile(self._h, iters, us, names, 16)
        if k < 0:
            raise JpegHipError(k, "jd_batch_profile")
        return {names[i].decode(): float(us[i]) for i in range(k)}

    def timing(self, enable=True):
        _check(_lib().jd_batch_timing(self._h, 1 if enable else 0), "jd_batch_timing")

    def stage_times(self):
        """-> (runs averaged, {stage: microseconds}) measured with hipEvents inside the timed runs."""
        us = (ctypes.c_float * 16)()
        names = (ctypes.c_char_p * 16)()
        k = _lib().jd_batch_stage_times(self._h, us, names, 16)
        if k < 0:
            raise JpegHipError(k, "jd_batch_stage_times")
        return k, {names[i].decode(): float(us[i]) for i in range(6)}

    def workspace_bytes(self):
        return _lib().jd_batch_workspace_bytes(self._h)

    def unstuffed_len(self, i):
        v = _sz()
        _check(_lib().jd_batch_unstuffed_len(self._h, i, ctypes.byref(v)), "jd_batch_unstuffed_len")
        return v.value


def shard_images(n_images, world, rank):
    """Multi-GPU rule: whole images are independent, so image i goes to rank i % world -- a static
    round-robin with no data-path collective.  The reference decodes argv[1:] one after another
    (main.go:1509-1512); this is the same loop dealt over ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank %d / world %d" % (rank, world))
    return list(range(rank, n_images, world))


def shard_bands(n_bands, world, rank):
    """Restart-interval-range sharding of ONE image: band k of split_bands goes to rank k % world, as
    shard_images deals whole images.  The ranks write disjoint pixel rows of the picture; nothing is exchanged."""
    return shard_images(n_bands, world, rank)


def plan_job(total, world, rank, per_batch, distinct):
    """A rank's share of a job of `total` images (BASELINE config 5: "batch of 10k 1080p sharded across 1/2/4/8"), cut
    into jd_batch-sized groups: -> list of batches, each a list of (global image index, seed).  Image g of the job is the
    synthetic image with seed g % distinct; shard_images deals the indices; nothing is exchanged between ranks."""
    if per_batch < 1 or distinct < 1:
        raise ValueError("per_batch and distinct must be positive")
    mine = shard_images(total, world, rank)
    return [[(g, g % distinct) for g in mine[b0:b0 + per_batch]] for b0 in range(0, len(mine), per_batch)]


def max_over_ranks(seconds, dist=None, device=None):
    """bench.py's timing rule: the job takes as long as its slowest rank."""
    if dist is None:
        return float(seconds)
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- synthetic workload maker (libjdsynth.so; host only) ---------------------------------------------------
_SYN = None


def _syn():
    global _SYN
    if _SYN is None:
        if not os.path.exists(SYNTH_PATH):
            raise RuntimeError("libjdsynth.so is not built")
        S = ctypes.CDLL(SYNTH_PATH)
        S.jds_create.argtypes = [_ci] * 6 + [ctypes.c_uint64, _ci, ctypes.POINTER(_vp)]
        S.jds_encode.argtypes = [_vp, _ci, ctypes.POINTER(ctypes.POINTER(_u8)), ctypes.POINTER(_sz)]
        S.jds_encode_ex.argtypes = [_vp, _ci, ctypes.c_uint, ctypes.POINTER(ctypes.POINTER(_u8)), ctypes.POINTER(_sz)]
        S.jds_destroy.argtypes = [_vp]
        S.jds_destroy.restype = None
        S.jds_free_buffer.argtypes = [ctypes.POINTER(_u8)]
        S.jds_free_buffer.restype = None
        S.jds_mcus.argtypes = [_vp]
        S.jds_blocks_per_mcu.argtypes = [_vp]
        S.jds_coefficients.argtypes = [_vp, _vp]
        _SYN = S
    return _SYN


class SynthImage:
    """Deterministic noise image -> quantised coefficients; encode() entropy-codes it (DRI twins)."""

    def __init__(self, width, height, hs=2, vs=2, ncomp=3, quality=85, seed=0, detail=2):
        h = _vp()
        rc = _syn().jds_create(width, height, hs, vs, ncomp, quality, seed, detail, ctypes.byref(h))
        if rc:
            raise ValueError("jds_create failed: %d" % rc)
        self._h = h
        self.width, self.height = width, height

    def __del__(self):
        if getattr(self, "_h", None) and _syn is not None:       # (module globals are gone at interpreter shutdown)
            _syn().jds_destroy(self._h)
            self._h = None

    def coefficients(self):
        S = _syn()
        out = np.empty((S.jds_mcus(self._h), S.jds_blocks_per_mcu(self._h), 64), np.int16)
        S.jds_coefficients(self._h, out.ctypes.data)
        return out

    SHARED_TABLES, ZERO_BASED_IDS, DQT_16BIT, RANDOM_TABLES = 1, 2, 4, 8   # RANDOM_TABLES | (seed << 8)

    def encode(self, restart_interval=0, flags=0):
        """flags: SHARED_TABLES (all components on Huffman tables 0/0), ZERO_BASED_IDS, DQT_16BIT, RANDOM_TABLES | seed << 8 (jd_synth.h)"""
        p, n = ctypes.POINTER(_u8)(), _sz()
        rc = _syn().jds_encode_ex(self._h, restart_interval, flags, ctypes.byref(p), ctypes.byref(n))
        if rc:
            raise ValueError("jds_encode failed: %d" % rc)
        data = ctypes.string_at(p, n.value)
        _syn().jds_free_buffer(p)
        return data
