import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import jpeg_decoder_amd as jd, orclib
w, h = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "320x240").split("x"))
ims = [jd.SynthImage(w, h, 2, 2, 3, quality=85, seed=s, detail=2) for s in range(32)]
files = [im.encode(0) for im in ims]
parsed = [jd.parse(f) for f in files]
cap = max(p[2] for p in parsed) + 64
nsub = (parsed[0][2] * 8 + 4095) // 4096
for n in [int(x) for x in sys.argv[2].split(",")]:
    for name, env in (("default", {}), ("two-pass", {"JD_NODRI_FUSED": "0"}), ("single-decode, shared", {"JD_NODRI_FUSED": "2", "JD_DEBUG_FUSED_SHARE_FROM": "1"}), ("single-decode, not shared", {"JD_NODRI_FUSED": "2", "JD_DEBUG_FUSED_SHARE": "0"})):
        for k in ("JD_NODRI_FUSED", "JD_DEBUG_FUSED_SHARE_FROM", "JD_DEBUG_FUSED_SHARE"): os.environ.pop(k, None)
        os.environ.update(env)
        b = jd.Batch([parsed[0][0]] * n, [cap] * n)
        for i in range(n):
            f, p = files[i % 32], parsed[i % 32]
            b.upload(i, f[p[1]:p[1] + p[2]])
        for _ in range(3): b.run(); b.wait()
        t0 = time.perf_counter()
        for _ in range(8): b.run(); b.wait()
        dt = (time.perf_counter() - t0) / 8
        ok = all(np.array_equal(b.download(i), orclib.OracleImage(files[i % 32]).rgb) for i in (0, n // 2, n - 1))
        print("%5d x %dx%d (%d sub-sequences, %d in all) %-26s %.3f ms  %.0f MP/s %s" % (n, w, h, nsub, n * nsub, name, dt * 1e3, n * w * h / dt / 1e6, "ok" if ok else "WRONG"), flush=True)
        b.close()
