# stage times per sampling mode (K2 kernels other than the benchmark's 4:2:0): 8 images of 4096 x 4096, DRI 8
import sys, os, json
sys.path.insert(0, '.')
import jpeg_decoder_amd as jd
for (hs, vs, nc, name) in [(1, 1, 1, "grey"), (1, 1, 3, "4:4:4"), (2, 1, 3, "4:2:2"), (1, 2, 3, "4:4:0"), (2, 2, 3, "4:2:0")]:
    files = [jd.SynthImage(4096, 4096, hs, vs, nc, quality=85, seed=s, detail=2).encode(8) for s in range(8)]
    parsed = [jd.parse(f) for f in files]
    b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed])
    for i, (f, p) in enumerate(zip(files, parsed)):
        b.upload(i, f[p[1]:p[1] + p[2]])
    b.profile(2)
    t = b.profile(6)
    print(name, "K2 %.0f us = %.0f MP/s in K2;" % (t["idct_colour"], 8 * 4096 * 4096 / t["idct_colour"]), {k: round(v) for k, v in t.items()}, flush=True)
    del b
