"""GPU: INTEGRATION.md's threading contract -- one host thread per batch, different batches from different threads, the
one-shot entry points (jd_decode, jd_decode_coefficients, jd_decode_scans) from any thread: four threads decode their own
images at the same time, every picture equals the oracle's.  (ctypes drops the GIL for the length of a call.)"""
import glob
import os
import threading

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)


def test_batches_and_one_shot_calls_from_four_threads():
    specs = [(640, 360, 2, 2, 3, 85, 0), (333, 129, 2, 1, 3, 92, 8), (1023, 200, 1, 1, 3, 98, 0), (250, 777, 1, 2, 3, 60, 3)]
    prog = open(sorted(glob.glob(os.path.join(HERE, "golden", "ref_test", "p", "*.jpg")))[0], "rb").read()
    want_prog = orclib.OracleImage(prog).rgb
    work = []
    for t, (w, h, hs, vs, nc, q, ri) in enumerate(specs):
        ims = [jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=100 * t + k, detail=1 + k % 3) for k in range(6)]
        files = [im.encode(ri) for im in ims]
        work.append((files, [orclib.OracleImage(im.encode(0)).rgb for im in ims], [im.coefficients() for im in ims]))
    errors = []
    start = threading.Barrier(len(specs))

    def worker(t):
        try:
            files, rgb, coef = work[t]
            parsed = [jd.parse(f) for f in files]
            b = jd.Batch([p[0] for p in parsed], [max(p[2] for p in parsed) + 64] * len(parsed), keep_coef=True)
            start.wait()
            for rnd in range(6):
                order = [(k + rnd) % len(files) for k in range(len(files))]
                for i, k in enumerate(order):
                    if bytes(parsed[k][0]) != bytes(parsed[i][0]):
                        order = list(range(len(files))); break      # (content-dependent frames: keep every image in its slot)
                for i, k in enumerate(order):
                    b.upload(i, files[k][parsed[k][1]:parsed[k][1] + parsed[k][2]])
                b.run(); b.wait()
                for i, k in enumerate(order):
                    assert np.array_equal(b.coefficients(i), coef[k]), ("batch", t, rnd, i)
                    assert np.array_equal(b.download(i), rgb[k]), ("batch", t, rnd, i)
                k = rnd % len(files)
                assert np.array_equal(jd.decode_jpeg(files[k]), rgb[k]), ("jd_decode", t, rnd)
                assert np.array_equal(jd.decode_coefficients(files[k]), coef[k]), ("jd_decode_coefficients", t, rnd)
                if t == rnd % len(specs):
                    assert np.array_equal(jd.decode_progressive(prog), want_prog), ("jd_decode_scans", t, rnd)
            b.close()
        except BaseException as e:                                  # (an assertion in a thread is not the test's otherwise)
            errors.append((t, repr(e)))
            try: start.abort()
            except Exception: pass

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(specs))]
    for th in threads: th.start()
    for th in threads: th.join(timeout=150)
    assert not any(th.is_alive() for th in threads), "a worker thread hangs"
    assert not errors, errors
