"""What would a pair-symbol writer table buy K1a?  (round-2 verdict, item 3)

For a C3-shaped image (4:2:0, Annex-K tables at quality 85, DRI 8) rebuild every block's AC symbol sequence from the
encoder's quantised coefficients, price each symbol at code length + magnitude bits, and replay K1a's block-synchronous
walk: lane = restart interval, 64 consecutive intervals per wave, step b costs the wave the LARGEST lookup count among
its 64 lanes' blocks b.  A pair table (P primary bits) resolves two symbols in one lookup when both fit in P bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeg_decoder_amd as jd

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
im = jd.SynthImage(W, H, 2, 2, 3, quality=85, seed=0, detail=2)
fr, _, _ = jd.parse(im.encode(8))
coef = im.coefficients()                     # [mcu][6][64] zig-zag, absolute DC
nm = coef.shape[0]


def lengths(h):
    out, k = {}, 0
    for ln in range(1, 17):
        for _ in range(h.bits[ln - 1]):
            out[h.vals[k]] = ln
            k += 1
    return out


ac = [lengths(fr.huff[1][fr.comp[c].ta]) for c in range(3)]
res = {}
for P in (0, 9, 10, 11, 12):
    looks = np.zeros((nm, 6), np.int32)
    syms = np.zeros((nm, 6), np.int32)
    for b in range(6):
        tab = ac[0 if b < 4 else b - 3]
        for m in range(nm):
            blk = coef[m, b]
            nzi = np.nonzero(blk[1:])[0] + 1
            bits, prev = [], 0
            for i in nzi:
                run = i - prev - 1
                while run > 15:
                    bits.append(tab[0xF0]); run -= 16
                size = int(abs(int(blk[i]))).bit_length()
                bits.append(tab[(run << 4) | size] + size)
                prev = i
            if prev != 63:
                bits.append(tab[0])
            n, j = 0, 0
            while j < len(bits):
                if P and j + 1 < len(bits) and bits[j] + bits[j + 1] <= P:
                    j += 2
                else:
                    j += 1
                n += 1
            looks[m, b], syms[m, b] = n, len(bits)
    # K1a: interval = 8 MCUs = 48 blocks; wave = 64 consecutive intervals; step b = block b of each lane's interval
    iv = looks[: nm // 8 * 8].reshape(-1, 48)
    iv = iv[: iv.shape[0] // 64 * 64].reshape(-1, 64, 48)
    res[P] = (syms.mean(), looks.mean(), iv.max(axis=1).mean())
    print("primary bits %2d: AC symbols/block %.2f, lookups/block %.2f (%.2f symbols per lookup), wave step = max over 64 lanes: %.2f lookups"
          % (P, res[P][0], res[P][1], res[P][0] / res[P][1], res[P][2]))
base = res[0][2]
for P in (9, 10, 11, 12):
    print("P=%d: slowest-lane lookups x %.3f; break-even cost of a pair step = %.1f VALU (single step: 28)" % (P, res[P][2] / base, 28 * base / res[P][2]))
