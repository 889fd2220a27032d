"""GPU: bench.py's N-rank path on real kernels.  The build box has one GPU, so both ranks decode on device 0 (`--share-gpu`:
gloo for the barrier, because RCCL refuses two ranks on one device) and the pictures are small (`--dry-size`): a
rehearsal of the launcher, the sharding (whole images, 10k-job dealing, restart-interval bands of one picture on
different ranks) and the per-rank verification against the oracle -- not a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_decode_and_verify_every_workload():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dry-size", "512x384",
                        "--images", "4", "--total", "48", "--batch", "8", "--steps", "2", "--warmup", "1", "--repeat", "1",
                        "--no-cpu-baseline", "--no-e2e"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["scaling"] == "weak"
    assert d["config"]["images_per_pass"] == 8 and d["value"] > 0              # 4 pictures per rank
    w = d["workloads"]
    assert set(w) == {"c4", "c5", "c3_bands"}
    for k in w:
        assert w[k]["verified"] is True and w[k]["value"] > 0, k               # sum over both ranks' own checks
    assert w["c5"]["config"]["images_per_pass"] == 48 and w["c5"]["scaling"] == "strong"
    assert w["c3_bands"]["config"]["images_per_pass"] == 4 and w["c3_bands"]["config"]["items_per_pass"] == 8   # 2 bands per picture
