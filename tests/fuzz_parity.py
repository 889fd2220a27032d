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
    if os.environ.get("FUZZ_CLASSIC"):                # the two-pass path alone
        ri = 0; os.environ["JD_NODRI_FUSED"] = "0"
    elif os.environ.get("FUZZ_NODRI"):                  # stress the single-decode path: no restart markers, random knobs
        ri = 0
        for k, vals in (("JD_DEBUG_FUSED_ROUNDS", [None, None, 0, 1, 2]), ("JD_DEBUG_FUSED_CH0", [None, None, 4, 8, 24]),
                        ("JD_DEBUG_FUSED_CAP_PERCENT", [None, None, None, 30]), ("JD_NODRI_FUSED", [2, 2, 2, 2, None, 0]), ("JD_DEBUG_FUSED_LONG_FROM", [None, 0, 0])):
            v = rng.choice(vals)
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = str(v)
    form = rng.choice([None, 8, 16])
    flags = rng.choice([0, 0, 0, jd.SynthImage.SHARED_TABLES])
    if rng.random() < 0.35: flags |= jd.SynthImage.RANDOM_TABLES | (rng.randrange(1 << 20) << 8)   # random valid Huffman tables, codes of 1..16 bits
    seed = rng.randrange(1 << 30)
    im = jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=seed, detail=detail)
    data, twin = im.encode(ri, flags), im.encode(0, flags)
    lo_hi = os.environ.get("FUZZ_WINDOW")              # "a:b": draw every case, run only cases a .. b - 1 (a quick way back to a failure)
    if lo_hi:
        a_, b_ = (int(x) for x in lo_hi.split(":"))
        if n >= b_: break
        if n < a_:
            n += 1; continue
        print("case", n, (w, h, hs, vs, nc, q, detail, ri, form, flags, seed), flush=True)
    if form is None: os.environ.pop("JD_COEF_FORM", None)
    else: os.environ["JD_COEF_FORM"] = str(form)
    try:
        c = jd.decode_coefficients(data); rgb = jd.decode_jpeg(data)
        o = orclib.OracleImage(twin)
        okc, okp = np.array_equal(c, im.coefficients()), np.array_equal(rgb, o.rgb)
        ok = okc and okp
        if not ok: print("coef ok", okc, "pixels ok", okp)
    except Exception as e:
        ok = False; print("EXC", e)
    n += 1
    if time.time() - t_say > 30:                      # (a silent GPU command is taken to be hung)
        t_say = time.time(); print("...", n, "cases,", bad, "bad", flush=True)
    if not ok:
        again = []
        for _ in range(3):                            # does it fail again?  (a fault that comes and goes is a race or stale memory)
            try:
                again.append(bool(np.array_equal(jd.decode_coefficients(data), im.coefficients())) and bool(np.array_equal(jd.decode_jpeg(data), o.rgb)))
            except Exception as e:
                again.append(str(e)[-40:])
        print("again:", again)
        for knob in ({"JD_DEBUG_POISON": "0"}, {"JD_DEBUG_POISON": "165"}, {"JD_SYNC_FIXUPS": "8"}, {"JD_COEF_FORM": "16"}, {"JD_NODRI_FUSED": "1"}, {}):
            saved = {k: os.environ.get(k) for k in knob}
            os.environ.update(knob)
            try:
                r = bool(np.array_equal(jd.decode_coefficients(data), im.coefficients()))
            except Exception as e:
                r = str(e)[-30:]
            print("   with", knob, "->", r)
            for k, v in saved.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        bad += 1; print("MISMATCH", w, h, hs, vs, nc, q, detail, ri, form, flags, "seed", seed, {k: v for k, v in os.environ.items() if k.startswith("JD_")}, flush=True)
print("cases", n, "bad", bad)
