"""one progressive file, a few one-shot decodes: run under rocprofv3 --kernel-trace --stats to see what kp_scan costs per level"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeg_decoder_amd as jd
jd.init([0])
name = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/ref_test/p/cat0-q.jpg"
d = open(name, "rb").read()
fr, scans, n = jd.parse_scans(d)
print(name, fr.width, fr.height, "scans:", [(scans[k].ncomp, scans[k].ss, scans[k].se, scans[k].ah, scans[k].al, scans[k].ecs_len) for k in range(n)])
jd.decode_progressive(d)
for _ in range(5):
    t0 = time.perf_counter()
    jd.decode_progressive(d)
    print("%.2f ms" % ((time.perf_counter() - t0) * 1e3))
