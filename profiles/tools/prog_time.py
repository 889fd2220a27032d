import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import jpeg_decoder_amd as jd
jd.init([0])
files = [open(p, "rb").read() for p in sorted(glob.glob("tests/golden/ref_test/p/*.jpg"))]
for p, d in zip(sorted(glob.glob("tests/golden/ref_test/p/*.jpg")), files):
    jd.decode_progressive(d)
    t0 = time.perf_counter()
    for _ in range(3):
        out = jd.decode_progressive(d)
    dt = (time.perf_counter() - t0) / 3
    print(os.path.basename(p), out.shape, "%.1f ms per file (one-shot, PCIe and allocation included) = %.1f MP/s" % (dt * 1e3, out.shape[0] * out.shape[1] / dt / 1e6))
for n in (16, 64, 256, 1024):
    batch = [files[i % 3] for i in range(n)]                    # the three 800x533 files
    jd.decode_progressive_batch(batch[:4])
    t0 = time.perf_counter()
    outs, st = jd.decode_progressive_batch(batch)
    dt = time.perf_counter() - t0
    assert not any(st)
    print("batch of %d 800x533 progressive files: %.1f ms = %.2f ms per file = %.1f MP/s (host parse, unstuff, PCIe and allocation included)" % (
        n, dt * 1e3, dt * 1e3 / n, n * 800 * 533 / dt / 1e6))
