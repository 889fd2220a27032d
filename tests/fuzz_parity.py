"""one-off randomised parity sweep: GPU vs oracle over random geometry / sampling / quality / DRI / coefficient form"""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import jpeg_decoder_amd as jd, orclib
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
t0 = time.time(); n = 0; bad = 0; t_say = t0
modes = [(1, 1, 1), (1, 1, 3), (2, 1, 3), (1, 2, 3), (2, 2, 3)]
while time.time() - t0 < budget:
    hs, vs, nc = rng.choice(modes)
    w = rng.choice([8, 17, 64, 250, 640, 1023, 1920, 2500]); h = rng.choice([8, 9, 33, 200, 360, 777, 1080, 1500])
    q = rng.choice([30, 60, 85, 90, 93, 95, 98, 100]); detail = rng.choice([1, 2, 3])
    mcus = ((w + 8 * hs - 1) // (8 * hs)) * ((h + 8 * vs - 1) // (8 * vs))
    ri = rng.choice([0, 0, 1, 2, 3, 7, 8, 16, max(1, mcus // 3), mcus + 5])
    if ri > 65535: ri = 0
    form = rng.choice([None, 8, 16])
    flags = rng.choice([0, 0, 0, jd.SynthImage.SHARED_TABLES])
    im = jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=rng.randrange(1 << 30), detail=detail)
    data, twin = im.encode(ri, flags), im.encode(0, flags)
    if form is None: os.environ.pop("JD_COEF_FORM", None)
    else: os.environ["JD_COEF_FORM"] = str(form)
    try:
        c = jd.decode_coefficients(data); rgb = jd.decode_jpeg(data)
        o = orclib.OracleImage(twin)
        ok = np.array_equal(c, im.coefficients()) and np.array_equal(rgb, o.rgb)
    except Exception as e:
        ok = False; print("EXC", e)
    n += 1
    if time.time() - t_say > 30:                      # (a silent GPU command is taken to be hung)
        t_say = time.time(); print("...", n, "cases,", bad, "bad", flush=True)
    if not ok:
        bad += 1; print("MISMATCH", w, h, hs, vs, nc, q, detail, ri, form, flags)
print("cases", n, "bad", bad)
