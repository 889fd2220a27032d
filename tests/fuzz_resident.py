"""one-off randomised sweep over jd_decode_resident (the allocation-free entry point): calls in a row on one device, each
with a frame drawn at random -- or the previous frame again with other content (the plan is reused), or damaged data --
against the oracle's pixels.  A frame beyond the fixed scratch must be refused with JD_E_TOO_LARGE and leave the next call alone.
usage: python tests/fuzz_resident.py SEED SECONDS"""
import os, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import jpeg_decoder_amd as jd, orclib

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
jd.init([0])
modes = [(1, 1, 1), (1, 1, 3), (2, 1, 3), (1, 2, 3), (2, 2, 3)]
t0 = time.time(); t_say = t0; n = 0; bad = 0; refused = 0; hurt_n = 0
spec = None
while time.time() - t0 < budget:
    if spec is None or rng.random() < 0.6:
        hs, vs, nc = rng.choice(modes)
        w = rng.choice([8, 17, 64, 250, 640, 1023, 1920, 2500]); h = rng.choice([8, 9, 33, 200, 360, 777, 1080])
        mcus = ((w + 8 * hs - 1) // (8 * hs)) * ((h + 8 * vs - 1) // (8 * vs))
        spec = (w, h, hs, vs, nc, rng.choice([30, 60, 85, 90, 95, 98, 100]), rng.choice([1, 2, 3]),
                min(65535, rng.choice([0, 0, 0, 1, 3, 8, max(1, mcus // 3)])), rng.choice([0, 0, 0, jd.SynthImage.SHARED_TABLES]))
    w, h, hs, vs, nc, q, detail, ri, fl = spec
    im = jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=rng.randrange(1 << 30), detail=detail)
    data = im.encode(ri, fl)
    fr, off, ln = jd.parse(data)
    body = bytearray(data[off:off + ln])
    hurt = rng.random() < 0.15 and ln > 8
    if hurt:
        body = body[:rng.randrange(1, ln)]
        while body and body[-1] == 0xFF: body.pop()
        if not body: body = bytearray(b"\x00")
    ecs = torch.from_numpy(np.frombuffer(bytes(body), np.uint8).copy()).cuda()
    pad = rng.choice([0, 0, 7, 64])
    out = torch.full((h, w * 3 + pad), 0xA5, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    try:
        jd.decode_resident(fr, ecs.data_ptr(), len(body), out.data_ptr(), w * 3 + pad, 0, rng.choice([None, torch.cuda.current_stream().cuda_stream]))
        got = out.cpu().numpy()
        if hurt: hurt_n += 1
        else:
            ok = np.array_equal(got[:, :w * 3].reshape(h, w, 3), orclib.OracleImage(im.encode(0, fl)).rgb) and (got[:, w * 3:] == 0xA5).all()
            if not ok: bad += 1; print("MISMATCH call", n, spec, flush=True)
    except jd.JpegHipError as e:
        if hurt: hurt_n += 1
        elif e.code == -14: refused += 1; spec = None
        else: bad += 1; print("EXC call", n, e, spec, flush=True)
    n += 1
    if time.time() - t_say > 30:
        t_say = time.time(); print("...", n, "calls,", bad, "bad", flush=True)
print("calls", n, "refused as too large", refused, "damaged", hurt_n, "bad", bad)
