import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jpeg_decoder_amd as jd, orclib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w, h = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "320x240").split("x"))
ims = [jd.SynthImage(w, h, 2, 2, 3, quality=85, seed=s, detail=2) for s in range(32)]
files = [im.encode(0) for im in ims]
parsed = [jd.parse(f) for f in files]
cap = max(p[2] for p in parsed) + 64
print("stream bytes", parsed[0][2], "sub-sequences of 4096 bits", (parsed[0][2] * 8 + 4095) // 4096, flush=True)
for share in ("1", "0"):
    os.environ["JD_DEBUG_FUSED_SHARE"] = share
    b = jd.Batch([parsed[0][0]] * n, [cap] * n)
    for i in range(n):
        f, p = files[i % 32], parsed[i % 32]
        b.upload(i, f[p[1]:p[1] + p[2]])
    for _ in range(2): b.run(); b.wait()
    b.timing(True)
    t0 = time.perf_counter()
    for _ in range(5): b.run(); b.wait()
    dt = (time.perf_counter() - t0) / 5
    k, st = b.stage_times()
    ok = all(np.array_equal(b.download(i), orclib.OracleImage(files[i % 32]).rgb) for i in (0, 1, 31, 32, n // 2, n - 1))
    print("share", share, "%d x %dx%d: %.2f ms per run, %.0f MP/s" % (n, w, h, dt * 1e3, n * w * h / dt / 1e6), {a: round(v) for a, v in st.items()}, "ok" if ok else "WRONG", flush=True)
    b.close()
