import sys, os, json
sys.path.insert(0, '.')
import bench, jpeg_decoder_amd as jd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
made = bench.make_distinct(os.environ.get("WL","c4"), list(range(n)), 85, 2)
files = [made[s][0] for s in range(n)]
parsed = [jd.parse(f) for f in files]
b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed])
for i, (f, p) in enumerate(zip(files, parsed)):
    b.upload(i, f[p[1]:p[1] + p[2]])
b.profile(2)
print(json.dumps(b.profile(8)))
