"""Makes tests/golden/prog_twins/: the same pixels saved by Pillow (libjpeg) as a baseline and as a progressive JPEG.
libjpeg computes the same quantised coefficients for both; only the entropy coding differs (one interleaved scan against
ten spectral-selection / successive-approximation scans).  The oracle's SOF0 branch is pinned by SURVEY Appendix B, so
"progressive twin decodes to the baseline twin's coefficients" pins its SOF2 branch (tests/test_oracle_golden.py).
Source pixels: crops of the reference's own test/cat0.jpg (MIT, LICENSE.reference) and a synthetic gradient + noise.
Run from the repo root with Pillow installed; the outputs are committed."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "prog_twins")
os.makedirs(OUT, exist_ok=True)
cat = Image.open(os.path.join(HERE, "ref_test", "cat0.jpg")).convert("RGB")
rng = np.random.default_rng(7)
yy, xx = np.mgrid[0:97, 0:141]
noise = np.clip(np.stack([xx * 1.7, yy * 2.3, (xx + yy) * 0.9], -1) + rng.normal(0, 25, (97, 141, 3)), 0, 255).astype(np.uint8)
sources = {"cat_crop": cat.crop((300, 150, 300 + 200, 150 + 136)), "noise": Image.fromarray(noise)}
for name, im in sources.items():
    for sub, tag in ((0, "444"), (1, "422"), (2, "420")):
        for q in (60, 92):
            kw = dict(quality=q, subsampling=sub, optimize=(q == 92))
            im.save(os.path.join(OUT, "%s_%s_q%d_base.jpg" % (name, tag, q)), "JPEG", **kw)
            im.save(os.path.join(OUT, "%s_%s_q%d_prog.jpg" % (name, tag, q)), "JPEG", progressive=True, **kw)
    g = im.convert("L")
    g.save(os.path.join(OUT, "%s_gray_base.jpg" % name), "JPEG", quality=85)
    g.save(os.path.join(OUT, "%s_gray_prog.jpg" % name), "JPEG", quality=85, progressive=True)
