"""GPU: the bands of jd_split_bands, decoded as the images of one jd_batch (as they would be on different GPUs), tile the
picture the whole image decodes to, and that picture is the oracle's."""
import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("geom", [(1024, 600, 2, 2, 3, 8), (640, 360, 1, 1, 3, 80), (512, 520, 2, 1, 3, 4), (4096, 2048, 2, 2, 3, 8)])
@pytest.mark.parametrize("parts", [2, 5])
def test_bands_on_device(geom, parts):
    w, h, hs, vs, nc, ri = geom
    im = jd.SynthImage(w, h, hs, vs, nc, quality=85, seed=11, detail=2)
    f = im.encode(ri)
    fr, off, ln = jd.parse(f)
    ecs = np.frombuffer(f, np.uint8)[off:off + ln]
    bands = jd.split_bands(fr, ecs, parts)
    assert len(bands) == parts
    b = jd.Batch([x.frame for x in bands], [x.ecs_len for x in bands])
    for k, x in enumerate(bands):
        b.upload(k, ecs[x.ecs_off:x.ecs_off + x.ecs_len])
    b.run()
    b.wait()
    picture = np.concatenate([b.download(k) for k in range(len(bands))], axis=0)
    whole = jd.decode_jpeg(f)
    assert np.array_equal(picture, whole)
    assert np.array_equal(picture, orclib.OracleImage(im.encode(0)).rgb)        # the reference's pixels (no-DRI twin, F3)


def test_grey_2x2_sampling_byte_on_device():
    """grey DRI file whose SOF says h = v = 2: bands are 8-pixel rows (1x1 MCU), and they tile the whole decode"""
    w, h, ri = 256, 200, 8
    f = bytearray(jd.SynthImage(w, h, 1, 1, 1, quality=85, seed=7, detail=2).encode(ri))
    i = f.index(b"\xff\xc0")
    f[i + 11] = 0x22
    f = bytes(f)
    fr, off, ln = jd.parse(f)
    ecs = np.frombuffer(f, np.uint8)[off:off + ln]
    bands = jd.split_bands(fr, ecs, 3)
    assert len(bands) == 3 and [b.y0 for b in bands] == [0, 64, 128]                # 25 rows of 8 pixels, cut every 2 rows
    b = jd.Batch([x.frame for x in bands], [x.ecs_len for x in bands])
    for k, x in enumerate(bands):
        b.upload(k, ecs[x.ecs_off:x.ecs_off + x.ecs_len])
    b.run()
    b.wait()
    picture = np.concatenate([b.download(k) for k in range(len(bands))], axis=0)
    assert np.array_equal(picture, jd.decode_jpeg(f))
    assert np.array_equal(picture, orclib.OracleImage(f, orclib.MODE_RESTART_AWARE).rgb)
