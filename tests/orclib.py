"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")

MODE_REFERENCE = 0
MODE_RESTART_AWARE = 1
COS_GO = 0
COS_LIBM = 1


def _load():
    src = os.path.join(ROOT, "oracle", "oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(_SO)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.orc_decode_mem.argtypes = [vp, ctypes.c_size_t, ci, ctypes.POINTER(vp)]
    L.orc_decode_file.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(vp)]
    L.orc_last_message.restype = ctypes.c_char_p
    L.orc_strerror.restype = ctypes.c_char_p
    L.orc_strerror.argtypes = [ci]
    for n in ("width", "height", "components", "block_width", "block_height", "block_width_real",
              "block_height_real", "restart_interval"):
        getattr(L, "orc_" + n).argtypes = [vp]
    L.orc_hfactor.argtypes = [vp, ci]
    L.orc_vfactor.argtypes = [vp, ci]
    L.orc_bitstream_len.argtypes = [vp]
    L.orc_bitstream_len.restype = ctypes.c_size_t
    for n in ("coef", "ycc", "rgb"):
        getattr(L, "orc_get_" + n).argtypes = [vp, vp]
    L.orc_bmp_bytes.argtypes = [vp, vp, ctypes.c_size_t]
    L.orc_bmp_bytes.restype = ctypes.c_long
    L.orc_pixels_from_coef.argtypes = [vp, vp, vp]
    L.orc_last_timing.argtypes = [vp, vp]
    L.orc_free.argtypes = [vp]
    L.orc_get_idct_constants.argtypes = [vp]
    L.orc_set_cos_flavour.argtypes = [ci]
    return L


LIB = _load()


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oracle error %d (%s): %s" % (code, LIB.orc_strerror(code).decode(), msg))
        self.code = code


class OracleImage:
    """One decoded image with every stage snapshot (coef / ycc / rgb)."""

    def __init__(self, data: bytes, mode: int = MODE_REFERENCE):
        h = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * len(data)).from_buffer_copy(data)
        rc = LIB.orc_decode_mem(buf, len(data), mode, ctypes.byref(h))
        if rc:
            raise OracleError(rc, LIB.orc_last_message().decode())
        self._h = h
        g = lambda n: getattr(LIB, "orc_" + n)(h)
        self.width, self.height, self.ncomp = g("width"), g("height"), g("components")
        self.bw, self.bh, self.bwr, self.bhr = g("block_width"), g("block_height"), g("block_width_real"), g("block_height_real")
        self.H, self.V = LIB.orc_hfactor(h, 0), LIB.orc_vfactor(h, 0)
        self.restart_interval = g("restart_interval")
        self.bitstream_len = LIB.orc_bitstream_len(h)
        t = (ctypes.c_double * 2)()
        LIB.orc_last_timing(h, t)
        self.t_entropy, self.t_pixels = t[0], t[1]

    def __del__(self):
        if getattr(self, "_h", None):
            LIB.orc_free(self._h)
            self._h = None

    def _stage(self, name):
        a = np.zeros((self.bwr * self.bhr, 3, 64), np.int16)
        rc = getattr(LIB, "orc_get_" + name)(self._h, a.ctypes.data)
        assert rc == 0
        return a

    @property
    def coef(self):
        return self._stage("coef")

    @property
    def ycc(self):
        return self._stage("ycc")

    @property
    def rgb(self):
        a = np.zeros((self.height, self.width, 3), np.uint8)
        LIB.orc_get_rgb(self._h, a.ctypes.data)
        return a

    def bmp(self) -> bytes:
        n = LIB.orc_bmp_bytes(self._h, None, 0)
        b = (ctypes.c_ubyte * n)()
        assert LIB.orc_bmp_bytes(self._h, b, n) == n
        return bytes(b)

    def pixels_from_coef(self, coef: np.ndarray) -> np.ndarray:
        """Reference pixel pipeline (main.go:1054-1057) on given Header.blocks-layout coefficients."""
        coef = np.ascontiguousarray(coef, np.int16)
        assert coef.shape == (self.bwr * self.bhr, 3, 64)
        out = np.zeros((self.height, self.width, 3), np.uint8)
        rc = LIB.orc_pixels_from_coef(self._h, coef.ctypes.data, out.ctypes.data)
        if rc:
            raise OracleError(rc, LIB.orc_last_message().decode())
        return out

    # ---- layout helpers shared by the parity tests -------------------------------------------
    def scan_order_coef(self, zigzag=True) -> np.ndarray:
        """Coefficients re-laid in scan order: [mcu][block-in-MCU][64] (Y blocks row-major inside the MCU,
        then Cb, Cr), optionally back in zig-zag order -- the device layout of jd_decode_coefficients."""
        return blocks_to_scan_order(self.coef, self.bw, self.bh, self.bwr, self.H, self.V, self.ncomp, zigzag)


def zigzag_table():
    zz, r, c = [], 0, 0
    for _ in range(64):
        zz.append(r * 8 + c)
        if (r + c) % 2 == 0:
            if c == 7:
                r += 1
            elif r == 0:
                c += 1
            else:
                r, c = r - 1, c + 1
        else:
            if r == 7:
                c += 1
            elif c == 0:
                r += 1
            else:
                r, c = r + 1, c - 1
    return np.array(zz)


def blocks_to_scan_order(blocks, bw, bh, bwr, H, V, ncomp, zigzag=True):
    mx, my = (bw + H - 1) // H, (bh + V - 1) // V
    bpm = H * V + (ncomp - 1)
    out = np.zeros((my * mx, bpm, 64), np.int16)
    zz = zigzag_table()
    ys, xs = np.meshgrid(np.arange(my) * V, np.arange(mx) * H, indexing="ij")
    k = 0
    for u in range(V):
        for v in range(H):
            idx = ((xs + v) + (ys + u) * bwr).reshape(-1)
            out[:, k] = blocks[idx, 0]
            k += 1
    base = (xs + ys * bwr).reshape(-1)
    for c in range(1, ncomp):
        out[:, k] = blocks[base, c]
        k += 1
    if zigzag:
        out = out[:, :, zz]
    return out


def idct_constants(flavour=COS_GO):
    LIB.orc_set_cos_flavour(flavour)
    c = (ctypes.c_double * 14)()
    LIB.orc_get_idct_constants(c)
    LIB.orc_set_cos_flavour(COS_GO)
    return np.array(list(c), np.float64)
