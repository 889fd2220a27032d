"""one-off randomised sweep over jd_split_bands: a DRI image of random geometry cut into a random number of bands, the bands
decoded as the images of one batch, stitched and compared with the oracle's picture.  Restart intervals that do not tile
MCU rows must be refused (an error code), never mis-cut.
usage: python tests/fuzz_bands.py SEED SECONDS"""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import jpeg_decoder_amd as jd, orclib

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
modes = [(1, 1, 1), (1, 1, 3), (2, 1, 3), (1, 2, 3), (2, 2, 3)]
t0 = time.time(); t_say = t0; n = 0; bad = 0; refused = 0; cut = 0
while time.time() - t0 < budget:
    hs, vs, nc = rng.choice(modes)
    w = rng.choice([8, 17, 64, 250, 640, 1023, 1920]); h = rng.choice([8, 9, 33, 200, 360, 777, 1080])
    mcux = (w + 8 * hs - 1) // (8 * hs); mcuy = (h + 8 * vs - 1) // (8 * vs)
    ri = rng.choice([mcux, mcux, 2 * mcux, 3 * mcux, max(1, mcux // 2), 1, 2, 8, rng.randrange(1, mcux * mcuy + 2)])
    if ri > 65535: ri = mcux
    parts = rng.choice([1, 2, 3, 4, 5, 8, 13, 64])
    spec = (w, h, hs, vs, nc, ri, parts)
    im = jd.SynthImage(w, h, hs, vs, nc, quality=rng.choice([30, 85, 95, 100]), seed=rng.randrange(1 << 30), detail=rng.choice([1, 2, 3]))
    f = im.encode(ri, rng.choice([0, 0, jd.SynthImage.SHARED_TABLES]))
    fr, off, ln = jd.parse(f)
    ecs = np.frombuffer(f, np.uint8)[off:off + ln]
    try:
        bands = jd.split_bands(fr, ecs, parts)
    except jd.JpegHipError as e:
        refused += 1; n += 1
        rows_tile = mcux % ri == 0                           # (jd_split_bands wants every MCU row to begin an interval)
        if rows_tile: bad += 1; print("REFUSED a tiling interval:", e, spec, flush=True)
        continue
    try:
        assert 1 <= len(bands) <= parts
        assert bands[0].y0 == 0 and sum(x.frame.height for x in bands) == h
        b = jd.Batch([x.frame for x in bands], [x.ecs_len for x in bands])
        for k, x in enumerate(bands):
            b.upload(k, ecs[x.ecs_off:x.ecs_off + x.ecs_len])
        b.run(); b.wait()
        picture = np.concatenate([b.download(k) for k in range(len(bands))], axis=0)
        b.close()
        ok = np.array_equal(picture, orclib.OracleImage(f, orclib.MODE_RESTART_AWARE).rgb)
        cut += len(bands)
    except Exception as e:
        ok = False; print("EXC", repr(e), flush=True)
    if not ok: bad += 1; print("MISMATCH", spec, [(x.y0, x.frame.height, x.ecs_off, x.ecs_len) for x in bands], flush=True)
    n += 1
    if time.time() - t_say > 30:
        t_say = time.time(); print("...", n, "images,", bad, "bad", flush=True)
print("images", n, "bands", cut, "refused", refused, "bad", bad)
