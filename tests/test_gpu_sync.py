"""GPU: the corners of the self-synchronising (no-DRI) path.

* every component on ONE Huffman table pair: the decoder's path through the stream does not depend on the block-in-MCU
  phase, so that phase never self-synchronises; the library runs the sync pass without it and predicts DC in a pass of
  its own (k1_dc_scan);
* the host-driven extra fix-up loop of jd_batch_wait, forced by enqueueing no blind fix-ups and switching the warm-up
  sub-sequences off (JD_SYNC_FIXUPS=0, JD_DEBUG_SYNC_NO_WARM=1: read when a batch is created)."""
import os

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu


def _check(data, twin=None):
    o = orclib.OracleImage(twin if twin is not None else data)
    assert np.array_equal(jd.decode_coefficients(data), o.scan_order_coef())
    assert np.array_equal(jd.decode_jpeg(data), o.rgb)


@pytest.mark.parametrize("w,h,hs,vs", [(64, 48, 2, 2), (333, 129, 2, 1), (1040, 520, 1, 2), (640, 360, 1, 1), (2048, 1024, 2, 2)])
def test_shared_tables_no_dri(w, h, hs, vs):
    im = jd.SynthImage(w, h, hs, vs, 3, quality=85, seed=w + h, detail=2)
    data = im.encode(0, jd.SynthImage.SHARED_TABLES)
    assert np.array_equal(jd.decode_coefficients(data), im.coefficients())
    _check(data)


def test_shared_tables_with_dri_and_in_a_mixed_batch():
    im = jd.SynthImage(640, 360, 2, 2, 3, quality=85, seed=21, detail=2)
    shared, shared_dri, plain = (im.encode(0, jd.SynthImage.SHARED_TABLES), im.encode(6, jd.SynthImage.SHARED_TABLES),
                                 im.encode(0))
    _check(shared_dri, twin=shared)
    files = [plain, shared, shared_dri, shared]
    parsed = [jd.parse(f) for f in files]
    b = jd.Batch([p[0] for p in parsed], [p[2] for p in parsed], keep_coef=True)
    for i, (f, p) in enumerate(zip(files, parsed)):
        b.upload(i, f[p[1]:p[1] + p[2]])
    b.run(); b.wait()
    want = orclib.OracleImage(plain)
    for i in range(len(files)):
        assert np.array_equal(b.coefficients(i), want.scan_order_coef()), i
        assert np.array_equal(b.download(i), want.rgb), i
    b.close()


@pytest.mark.parametrize("knobs", [{"JD_SYNC_FIXUPS": "0"}, {"JD_SYNC_FIXUPS": "0", "JD_DEBUG_SYNC_NO_WARM": "1"},
                                   {"JD_SYNC_FIXUPS": "1", "JD_DEBUG_SYNC_NO_WARM": "1"}])
def test_host_fixup_loop(knobs):
    """jd_batch_wait's 'more fix-ups needed' branch: the sync pass is left incomplete on purpose (several sync
    workgroups per image, no blind fix-ups and / or no warm-up), the host loop must finish it and re-run the consumers.
    (The two-pass path: these images would otherwise take the single-decode path, tests/test_gpu_single_decode.py.)"""
    knobs = dict(knobs, JD_NODRI_FUSED="0")
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        for seed, (w, h) in enumerate([(2048, 1024), (1920, 1080)]):
            im = jd.SynthImage(w, h, 2, 2, 3, quality=90, seed=seed, detail=3)
            data = im.encode(0)
            fr, off, ln = jd.parse(data)
            assert ln > 3 * 253 * 512, "needs several sync workgroups"
            assert np.array_equal(jd.decode_coefficients(data), im.coefficients())
            assert np.array_equal(jd.decode_jpeg(data), orclib.OracleImage(data).rgb)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_stale_checkpoints_are_not_trusted():
    """Round 3, found by tests/fuzz_parity.py (FUZZ_CLASSIC=1, case 1073 of seed 11): sub-sequence 1 of an image gets its
    first full decode in round 1 of k1_sync -- a MODE 2 decode, which stops at the first checkpoint whose state it
    reproduces -- but its checkpoints are only written by a MODE 1 decode.  It compared its state with whatever the
    checkpoint memory held from an earlier batch, and after this big image and this small one the third image's state
    matched a stale record: wrong block count, 'Error! EOF'.  Checkpoints are now trusted only when JdSub::pad says this run
    wrote them.  (The sequence depends on the allocator handing the third decode the first one's memory: it did, every
    time, in a fresh process and in the fuzz run.)"""
    old = {k: os.environ.get(k) for k in ("JD_NODRI_FUSED", "JD_COEF_FORM")}
    os.environ["JD_NODRI_FUSED"] = "0"
    try:
        for (w, h, hs, vs, q, detail, form, seed) in [(1920, 1500, 2, 1, 98, 3, None, 662586496), (8, 777, 1, 1, 100, 1, 8, 380998917),
                                                      (1023, 200, 2, 1, 60, 1, 8, 786934824)]:
            if form is None:
                os.environ.pop("JD_COEF_FORM", None)
            else:
                os.environ["JD_COEF_FORM"] = str(form)
            im = jd.SynthImage(w, h, hs, vs, 3, quality=q, seed=seed, detail=detail)
            data = im.encode(0)
            assert np.array_equal(jd.decode_coefficients(data), im.coefficients()), (w, h)
            assert np.array_equal(jd.decode_jpeg(data), orclib.OracleImage(data).rgb), (w, h)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
