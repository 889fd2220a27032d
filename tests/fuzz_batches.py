"""one-off randomised parity sweep over BATCHES: images of mixed geometry / sampling / quality / restart interval in one
jd_batch, run two or three times with other content in the same slots (what a run leaves behind must not reach the next:
lists, chunk tables, guesses, checkpoints), on either no-DRI path and with the debug knobs that shrink the staging.
Expected coefficients are the generator's, expected pixels the oracle's (a sample per run: the oracle is the slow side).
usage: python tests/fuzz_batches.py SEED SECONDS"""
import os, re, sys, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import jpeg_decoder_amd as jd, orclib

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
t0 = time.time(); n = 0; bad = 0; t_say = t0; n_img = 0
modes = [(1, 1, 1), (1, 1, 3), (2, 1, 3), (1, 2, 3), (2, 2, 3)]
KNOBS = (("JD_NODRI_FUSED", [None, 2, 2, 2, 0]), ("JD_DEBUG_FUSED_ROUNDS", [None, None, 0, 1, 2]),
         ("JD_DEBUG_FUSED_CH0", [None, None, 4, 8, 24]), ("JD_DEBUG_FUSED_CAP_PERCENT", [None, None, None, 30, 60]),
         ("JD_DEBUG_FUSED_LONG_FROM", [None, 0, 0]), ("JD_COEF_FORM", [None, None, 8, 16]), ("JD_DEBUG_POISON", [None, 0, 165, 255]),
         ("JD_DEBUG_FUSED_SHARE_FROM", [None, 1, 1]))           # 1: re-decode launches with eight waves per copy of the tables whatever the batch's size


def unstuff(ecs):
    """the reference's `_bitstream` (main.go:993-1021) and where the restart markers were (the encoder writes no fill bytes)"""
    out, rst = bytearray(), []
    for k, piece in enumerate(re.split(rb"\xff[\xd0-\xd7]", ecs)):
        if k: rst.append(len(out))
        out += piece.replace(b"\xff\x00", b"\xff")
    return bytes(out), rst


def slot(rng_):
    hs, vs, nc = rng_.choice(modes)
    w = rng_.choice([8, 17, 64, 250, 640, 1023, 1920]); h = rng_.choice([8, 9, 33, 200, 360, 777])
    q = rng_.choice([30, 60, 85, 90, 95, 98, 100]); detail = rng_.choice([1, 2, 3])
    mcus = ((w + 8 * hs - 1) // (8 * hs)) * ((h + 8 * vs - 1) // (8 * vs))
    ri = rng_.choice([0, 0, 0, 0, 1, 3, 8, max(1, mcus // 3)])
    flags = rng_.choice([0, 0, 0, jd.SynthImage.SHARED_TABLES])
    if rng_.random() < 0.3: flags |= jd.SynthImage.RANDOM_TABLES | (rng_.randrange(1 << 20) << 8)   # random valid Huffman tables, codes of 1..16 bits
    return (w, h, hs, vs, nc, q, detail, min(ri, 65535), flags)


BIG = os.environ.get("FUZZ_BIG")                          # batches large enough for the size rule to take the single-decode path
_pool = {}


def make_image(s, rng_):
    """a SynthImage for slot s; the large ones come from a small pool (making one costs more than decoding a batch)"""
    if s[0] * s[1] < 1000000:
        return jd.SynthImage(s[0], s[1], s[2], s[3], s[4], quality=s[5], seed=rng_.randrange(1 << 30), detail=s[6])
    key = (s[:7], rng_.randrange(3))
    if key not in _pool:
        _pool[key] = jd.SynthImage(s[0], s[1], s[2], s[3], s[4], quality=s[5], seed=1000 + len(_pool), detail=s[6])
    return _pool[key]


only = os.environ.get("FUZZ_BATCH_SEED")                # replay one batch: every draw of a batch comes from its own seed
while time.time() - t0 < budget:
    bseed = int(only) if only else rng.randrange(1 << 62)
    rng_b = random.Random(bseed)
    for k, vals in KNOBS:
        v = rng_b.choice(vals)
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    for k in os.environ.get("FUZZ_DROP", "").split(","):     # (with FUZZ_BATCH_SEED: the same batch without some of its knobs)
        os.environ.pop(k, None)
    for kv in os.environ.get("FUZZ_SET", "").split(","):
        if "=" in kv: os.environ[kv.split("=")[0]] = kv.split("=")[1]
    kind = rng_b.choice(["mixed", "mixed", "nodri", "same"])
    count = rng_b.choice([1, 2, 3, 5, 9, 17, 40])
    slots = [slot(rng_b) for _ in range(count)]
    if BIG:
        for k in ("JD_NODRI_FUSED", "JD_DEBUG_FUSED_CH0", "JD_DEBUG_FUSED_CAP_PERCENT", "JD_DEBUG_FUSED_LONG_FROM", "JD_COEF_FORM"): os.environ.pop(k, None)
        kind = "mixed"; count = rng_b.choice([100, 120, 140])
        bigs = [(1920, 1080, 2, 2, 3, q, 3, 0, 0) for q in (85, 90, 92)] + [(1920, 1080, 2, 1, 3, 90, 3, 0, 0)]
        slots = [rng_b.choice(bigs) if rng_b.random() < 0.8 else slot(rng_b) for _ in range(count)]
    if kind == "nodri": slots = [s[:7] + (0,) + s[8:] for s in slots]
    if kind == "same": slots = [slots[0][:7] + (0,) + slots[0][8:]] * count
    runs = rng_b.choice([1, 2, 3])
    b = None; frames = None; last_status = []
    pinned = rng_b.random() < 0.3                               # the pinned-memory pipeline instead of upload / run / download
    pre_unstuffed = not pinned and rng_b.random() < 0.4
    desc = ("seed", bseed, kind, count, runs, "pinned" if pinned else "upload+unstuffed" if pre_unstuffed else "upload", {k: v for k, v in os.environ.items() if k.startswith("JD_")})
    r = -1
    try:
        for r in range(runs):
            ims = [make_image(s, rng_b) for s in slots]
            enc = {}
            files = [enc.setdefault((id(im), s[7], s[8]), im.encode(s[7], s[8])) for im, s in zip(ims, slots)]
            parsed = [jd.parse(f) for f in files]
            # one image in some runs is damaged (cut short, or a stretch of its entropy-coded bytes wrecked): it may report
            # anything, its neighbours must not notice
            hurt = rng_b.randrange(count) if rng_b.random() < 0.3 else -1
            if hurt >= 0:
                fr_, off, ln = parsed[hurt]
                body = bytearray(files[hurt][off:off + ln])
                if rng_b.random() < 0.5 and ln > 8:
                    body = body[:rng_b.randrange(1, ln)]
                    while body and body[-1] == 0xFF: body.pop()
                    if not body: body = bytearray(b"\x00")
                else:
                    at = rng_b.randrange(ln)
                    for k in range(at, min(ln, at + rng_b.choice([1, 4, 64]))):
                        if body[k] != 0xFF and (k == 0 or body[k - 1] != 0xFF):
                            body[k] ^= rng_b.randrange(1, 255)
                            if body[k] == 0xFF: body[k] = 0x7F
                files[hurt] = files[hurt][:off] + bytes(body) + files[hurt][off + ln:]
                parsed[hurt] = (fr_, off, len(body))
            if b is None:
                frames = [bytes(p[0]) for p in parsed]
                b = jd.Batch([p[0] for p in parsed], [int(1.3 * jd.parse(enc[(id(im), s[7], s[8])])[2]) + 64 for im, s in zip(ims, slots)],
                             keep_coef=True, pinned_io=pinned)
            elif frames != [bytes(p[0]) for p in parsed]:
                break                                             # (tables that depend on the content: a new batch's business)
            try:
                if pinned:
                    for i, (f, p) in enumerate(zip(files, parsed)):
                        b.stage_ecs(i, f[p[1]:p[1] + p[2]])
                    b.submit()
                else:
                    for i, (f, p) in enumerate(zip(files, parsed)):
                        if pre_unstuffed and i != hurt and rng_b.random() < 0.5:   # the other entry form, per image and per run
                            bs, rst = unstuff(f[p[1]:p[1] + p[2]])
                            b.upload_unstuffed(i, bs, rst)
                        else: b.upload(i, f[p[1]:p[1] + p[2]])
                    b.run()
                b.wait()
            except jd.JpegHipError:
                if hurt < 0: raise
            finally: last_status = [b.status(i) for i in range(count)]
            look = set(rng_b.sample(range(count), min(count, 3)))
            for i in range(count):
                if i == hurt: continue
                if BIG and i not in look and rng_b.random() < 0.8: continue      # (reading 140 big pictures back takes longer than decoding them)
                ok = b.status(i) == 0 and np.array_equal(b.coefficients(i), ims[i].coefficients())
                if ok and i in look:
                    ok = np.array_equal(b.host_rgb(i) if pinned else b.download(i), orclib.OracleImage(ims[i].encode(0, slots[i][8])).rgb)
                n_img += 1
                if not ok:
                    bad += 1; print("MISMATCH batch", n, "run", r, "image", i, "status", b.status(i), "damaged:", hurt, slots[i], desc, flush=True)
    except Exception as e:
        bad += 1; print("EXC", repr(e), desc, flush=True)
        print("    run", r, "images with a status:", [(i, st, slots[i]) for i, st in enumerate(last_status) if st], flush=True)
    if b is not None: b.close()
    n += 1
    if only:
        print("statuses of the last run:", [b_status for b_status in last_status]); break
    if time.time() - t_say > 30:                      # (a silent GPU command is taken to be hung)
        t_say = time.time(); print("...", n, "batches,", n_img, "images,", bad, "bad", flush=True)
print("batches", n, "images", n_img, "bad", bad)
