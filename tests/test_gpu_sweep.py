"""GPU: seeded random sweep over geometry x sampling x quality x restart interval x coefficient form x table sharing,
coefficients against the encoder's own and pixels against the oracle (the no-DRI twin for DRI files).
tests/fuzz_parity.py is the open-ended form of the same loop (2,938 cases without a mismatch on the round-2 build)."""
import os
import random

import numpy as np
import pytest

import jpeg_decoder_amd as jd
import orclib

pytestmark = pytest.mark.gpu

MODES = [(1, 1, 1), (1, 1, 3), (2, 1, 3), (1, 2, 3), (2, 2, 3)]


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_sweep(seed):
    rng = random.Random(seed)
    old = os.environ.get("JD_COEF_FORM")
    try:
        for _ in range(40):
            hs, vs, nc = rng.choice(MODES)
            w = rng.choice([8, 17, 64, 250, 640, 1023, 1920])
            h = rng.choice([8, 9, 33, 200, 360, 777, 1080])
            q = rng.choice([30, 60, 85, 90, 93, 95, 98, 100])
            mcus = ((w + 8 * hs - 1) // (8 * hs)) * ((h + 8 * vs - 1) // (8 * vs))
            ri = rng.choice([0, 0, 1, 2, 3, 7, 8, 16, max(1, mcus // 3), mcus + 5])
            form = rng.choice([None, 8, 16])
            flags = rng.choice([0, 0, 0, jd.SynthImage.SHARED_TABLES])
            im = jd.SynthImage(w, h, hs, vs, nc, quality=q, seed=rng.randrange(1 << 30), detail=rng.choice([1, 2, 3]))
            data, twin = im.encode(ri, flags), im.encode(0, flags)
            if form is None:
                os.environ.pop("JD_COEF_FORM", None)
            else:
                os.environ["JD_COEF_FORM"] = str(form)
            case = (w, h, hs, vs, nc, q, ri, form, flags)
            assert np.array_equal(jd.decode_coefficients(data), im.coefficients()), case
            assert np.array_equal(jd.decode_jpeg(data), orclib.OracleImage(twin).rgb), case
    finally:
        if old is None:
            os.environ.pop("JD_COEF_FORM", None)
        else:
            os.environ["JD_COEF_FORM"] = old


@pytest.mark.parametrize("fused", [None, 2, 0])
def test_random_huffman_tables(fused):
    """files coded with random valid Huffman tables (JDS_RANDOM_TABLES: code lengths 1..16 for any symbol, symbols in
    random order) instead of Annex K.3's: what the table packers make of arbitrary tables -- primary / secondary splits,
    pair entries, the bound on blocks per sub-sequence -- through every Huffman kernel"""
    rng = random.Random(1000 + (fused or 7))
    S = jd.SynthImage
    old = {k: os.environ.get(k) for k in ("JD_NODRI_FUSED", "JD_COEF_FORM")}
    try:
        if fused is None: os.environ.pop("JD_NODRI_FUSED", None)
        else: os.environ["JD_NODRI_FUSED"] = str(fused)
        for k in range(36):
            hs, vs, nc = rng.choice(MODES)
            w = rng.choice([17, 64, 250, 640, 1023, 1920]); h = rng.choice([9, 33, 200, 360, 777])
            q = rng.choice([30, 85, 93, 98, 100])
            ri = rng.choice([0, 0, 0, 1, 8, 50])
            form = rng.choice([None, 8, 16])
            flags = S.RANDOM_TABLES | (rng.randrange(1 << 20) << 8) | rng.choice([0, 0, S.SHARED_TABLES])
            im = S(w, h, hs, vs, nc, quality=q, seed=rng.randrange(1 << 30), detail=rng.choice([1, 2, 3]))
            data, twin = im.encode(ri, flags), im.encode(0, flags)
            if form is None: os.environ.pop("JD_COEF_FORM", None)
            else: os.environ["JD_COEF_FORM"] = str(form)
            case = (k, w, h, hs, vs, nc, q, ri, form, flags)
            assert np.array_equal(jd.decode_coefficients(data), im.coefficients()), case
            assert np.array_equal(jd.decode_jpeg(data), orclib.OracleImage(twin).rgb), case
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
