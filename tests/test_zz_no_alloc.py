"""GPU: jd_decode_resident (the allocation-free entry point: workspace = the library's file-scope device array) against
the existing device-resident entry points jd_batch_{create, bind_ecs, bind_rgb, run, wait} -- bit-exact, on calls in a row
that share the array (a large image, then smaller ones of other kinds), and its size limits."""
import numpy as np
import pytest
import torch

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu


def _dev_inputs(data):
    fr, off, ln = jd.parse(data)
    ecs = torch.from_numpy(np.frombuffer(data, np.uint8, ln, off).copy()).cuda()
    assert ecs.data_ptr() % 16 == 0
    return fr, ecs, ln


def _batch_rgb(fr, ecs, ln, stride):
    out = torch.full((fr.height, stride), 0xA5, dtype=torch.uint8, device="cuda")
    b = jd.Batch([fr], [ln])
    b.bind_ecs(0, ecs.data_ptr(), ln)
    b.bind_rgb(0, out.data_ptr(), stride)
    b.run(); b.wait(); b.close()
    return out.cpu().numpy()


def _resident_rgb(fr, ecs, ln, stride, stream=None):
    out = torch.full((fr.height, stride), 0xA5, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    jd.decode_resident(fr, ecs.data_ptr(), ln, out.data_ptr(), stride, 0, stream)
    return out.cpu().numpy()


def test_resident_matches_batch_on_calls_in_a_row():
    jd.init([0])
    cases = [  # (w, h, hs, vs, ncomp, restart interval, synth flags): the first is the largest
        (1920, 1080, 2, 2, 3, 0, 0), (640, 360, 2, 2, 3, 4, 0), (333, 129, 2, 1, 3, 0, 0), (1040, 520, 1, 2, 3, 0, jd.SynthImage.SHARED_TABLES),
        (64, 48, 1, 1, 1, 0, 0), (1920, 1080, 2, 2, 3, 8, 0), (100, 75, 1, 1, 3, 3, 0)]
    for k, (w, h, hs, vs, nc, ri, fl) in enumerate(cases):
        im = jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=40 + k, detail=2)
        data = im.encode(ri, fl)
        fr, ecs, ln = _dev_inputs(data)
        stride = w * 3 + (0 if k % 2 == 0 else 20)
        want = _batch_rgb(fr, ecs, ln, stride)
        got = _resident_rgb(fr, ecs, ln, stride, torch.cuda.current_stream().cuda_stream if k % 3 == 1 else None)
        assert np.array_equal(got, want), "case %d %s" % (k, cases[k])
        assert (got[:, w * 3:] == 0xA5).all()
        oracle = orclib.OracleImage(im.encode(0, fl)).rgb
        assert np.array_equal(got[:, :w * 3].reshape(h, w, 3), oracle), "case %d vs oracle" % k


def test_resident_reuses_its_plan_for_the_same_frame():
    """Calls with one jd_frame share the plan of the first (no table packing, no table upload): different pictures of the
    same frame, one whose segment outgrows the planned capacity (re-planned), a smaller one again, each equal to the batch path."""
    jd.init([0])
    made = [jd.SynthImage(640, 480, 2, 2, 3, quality=85, seed=s, detail=d).encode(0) for s, d in ((3, 1), (6, 1), (1, 2), (2, 2), (3, 1), (5, 6))]
    frames = [jd.parse(m)[0] for m in made]
    assert all(bytes(f) == bytes(frames[0]) for f in frames)              # one frame: same geometry, same tables
    lens = [jd.parse(m)[2] for m in made]
    assert lens[2] > 1.25 * lens[0] + 65536 - 1 or lens[2] > 131072           # the third picture's segment outgrows the first plan's capacity
    for m in made:
        fr, ecs, ln = _dev_inputs(m)
        assert np.array_equal(_resident_rgb(fr, ecs, ln, 640 * 3), _batch_rgb(fr, ecs, ln, 640 * 3))


def test_resident_allocates_no_device_memory():
    jd.init([0])
    im = jd.SynthImage(800, 600, 2, 2, 3, quality=85, seed=5, detail=2)
    fr, ecs, ln = _dev_inputs(im.encode(0))
    out = torch.empty((600, 2400), dtype=torch.uint8, device="cuda")
    jd.decode_resident(fr, ecs.data_ptr(), ln, out.data_ptr(), 2400)          # first call loads what loads lazily
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        jd.decode_resident(fr, ecs.data_ptr(), ln, out.data_ptr(), 2400)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0


def test_resident_limits_and_errors():
    jd.init([0])
    big = jd.SynthImage(4096, 2048, 2, 2, 3, quality=85, seed=1, detail=2).encode(0)       # 12.6 M coefficients
    fr, ecs, ln = _dev_inputs(big)
    out = torch.empty((2048, 4096 * 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(jd.JpegHipError) as e:
        jd.decode_resident(fr, ecs.data_ptr(), ln, out.data_ptr(), 4096 * 3)
    assert e.value.code == -14
    # a damaged stream reports its error and leaves the entry point usable
    im = jd.SynthImage(320, 240, 2, 2, 3, quality=85, seed=2, detail=2)
    data = im.encode(0)
    fr, ecs, ln = _dev_inputs(data)
    out = torch.empty((240, 960), dtype=torch.uint8, device="cuda")
    with pytest.raises(jd.JpegHipError) as e:
        jd.decode_resident(fr, ecs.data_ptr(), ln // 2, out.data_ptr(), 960)
    assert e.value.code == -10
    jd.decode_resident(fr, ecs.data_ptr(), ln, out.data_ptr(), 960)
    assert np.array_equal(out.cpu().numpy().reshape(240, 320, 3), orclib.OracleImage(data).rgb)
