"""GPU: the pinned-memory I/O pipeline: library-owned pinned staging, H2D -> kernels -> D2H chained by
events, several batches in flight; and the ordering of uploads against batch creation."""
import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib
from test_gpu_robust import unstuff

pytestmark = pytest.mark.gpu


def _files(n, w=640, h=360, ri=0, seed0=0):
    return [jd.SynthImage(w, h, 2, 2, 3, quality=85, seed=seed0 + i, detail=2).encode(ri) for i in range(n)]


def test_three_batches_in_flight():
    per, slots, rounds = 6, 3, 3
    files = _files(per * slots * rounds)
    want = [orclib.OracleImage(f).rgb for f in files]
    fr, _, ln = jd.parse(files[0])
    cap = max(jd.parse(f)[2] for f in files) + 64
    batches = [jd.Batch([fr] * per, [cap] * per, pinned_io=True) for _ in range(slots)]
    owner = [None] * slots
    checked = 0
    for k in range(slots * rounds + slots):
        s = k % slots
        b = batches[s]
        if owner[s] is not None:                        # finish what this slot did last ...
            b.wait()
            for i in range(per):
                assert np.array_equal(b.host_rgb(i), want[owner[s] * per + i]), (owner[s], i)
                checked += 1
            owner[s] = None
        if k < slots * rounds:                          # ... and give it the next group of files
            group = files[k * per:(k + 1) * per]
            if k % 2:
                b.stage_jpegs(group, threads=3)
            else:
                for i, f in enumerate(group):
                    _, off, n = jd.parse(f)
                    b.stage_ecs(i, f[off:off + n])
            b.submit()
            owner[s] = k
    assert checked == len(files)
    # a finished staging may be submitted again as it stands
    batches[0].submit(); batches[0].wait()
    assert np.array_equal(batches[0].host_rgb(per - 1), want[(slots * rounds - slots) * per + per - 1])
    for b in batches:
        b.close()


def test_stage_errors():
    files = _files(2)
    fr, _, ln = jd.parse(files[0])
    b = jd.Batch([fr, fr], [ln + 4096] * 2, pinned_io=True)
    with pytest.raises(jd.JpegHipError):
        b.submit()                                      # nothing staged
    other = jd.SynthImage(320, 200, 2, 2, 3, quality=85, seed=1, detail=2).encode(0)
    with pytest.raises(jd.JpegHipError):
        b.stage_jpegs([files[0], other])                # second file is not the frame the slot was planned for
    plain = jd.Batch([fr], [ln + 4096])
    with pytest.raises(jd.JpegHipError):
        plain.stage_ecs(0, b"\x00" * 16)                # not a pinned-I/O batch
    plain.close(); b.close()


@pytest.mark.parametrize("ri", [0, 5])
def test_upload_right_after_create(ri):
    """Round 1 saw stale pre-unstuffed uploads about once in 100 decodes: jd_batch_create cleared the unstuffed-stream
    buffer with a legacy-stream hipMemset that the batch's non-blocking stream never waited for.  The clear now runs on
    the batch's stream; this is the sequence that raced: create, upload_unstuffed at once, run on the batch's own stream."""
    for k in range(8):
        w, h = 200 + 64 * k, 120 + 8 * k
        im = jd.SynthImage(w, h, 2, 2, 3, quality=85, seed=k, detail=2)
        data = im.encode(ri)
        fr, off, ln = jd.parse(data)
        bs, rst = unstuff(data[off:off + ln])
        b = jd.Batch([fr], [ln])
        b.upload_unstuffed(0, bs, rst)
        b.run(); b.wait()
        assert np.array_equal(b.download(0), orclib.OracleImage(im.encode(0)).rgb), k
        b.close()
