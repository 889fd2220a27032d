"""jd_decode_resident: host + device time per call for a 1920x1080 4:2:0 image, first call for a frame (plans: table
packing, table upload) against repeated calls with the same frame (plan reused)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import jpeg_decoder_amd as jd
jd.init([0])
for ri in (0, 8):
    files = [jd.SynthImage(1920, 1080, 2, 2, 3, quality=q, seed=3, detail=2).encode(ri) for q in (85, 75)]
    ins = []
    for f in files:
        fr, off, ln = jd.parse(f)
        ins.append((fr, torch.from_numpy(np.frombuffer(f, np.uint8, ln, off).copy()).cuda(), ln))
    out = torch.empty((1080, 5760), dtype=torch.uint8, device="cuda")
    jd.decode_resident(ins[0][0], ins[0][1].data_ptr(), ins[0][2], out.data_ptr(), 5760)
    first, again = [], []
    for it in range(20):
        fr, e, ln = ins[1 - it % 2]                      # alternate frames: every call re-plans
        torch.cuda.synchronize(); t0 = time.perf_counter()
        jd.decode_resident(fr, e.data_ptr(), ln, out.data_ptr(), 5760)
        first.append(time.perf_counter() - t0)
    fr, e, ln = ins[0]
    jd.decode_resident(fr, e.data_ptr(), ln, out.data_ptr(), 5760)
    for it in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        jd.decode_resident(fr, e.data_ptr(), ln, out.data_ptr(), 5760)
        again.append(time.perf_counter() - t0)
    print("restart interval %d: new frame %.0f us per call (median), same frame %.0f us per call (median)" % (ri, np.median(first) * 1e6, np.median(again) * 1e6))
