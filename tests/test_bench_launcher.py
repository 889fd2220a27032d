"""bench.py's multi-GPU entry on CPU.  `--gpus N` without a rendezvous in the environment must START N ranks (round 2
parsed the flag and never used it); under torch.distributed.run (the driver's command for N > 1) bench.py is a rank.
`--dry` keeps the launcher, the gloo rendezvous, the sharding plans (shard_images / plan_job / jd_split_bands +
shard_bands) and the one-line JSON, and does no decode work at all -- the GPU path is covered by the gpu-marked tests."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _run(cmd, **kw):
    p = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, **kw)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr.decode()


def _check_two_rank_line(line):
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["dry"] is True and d["metric"] == "decoded megapixels/sec"
    assert d["scaling"] == "weak" and d["items_rank0"] == 16                 # c3: every rank its own 16 pictures
    assert d["pixels_per_pass"] == 2 * 16 * 256 * 128
    w = d["workloads"]
    assert w["c5"]["scaling"] == "strong" and w["c5"]["items_rank0"] == 5000  # 10,000 images dealt over two ranks
    assert w["c5"]["pixels_per_pass"] == 10000 * 256 * 128
    # band sharding: the job is 16 pictures whatever the rank count; rank 0 holds one band of each
    assert w["c3_bands"]["scaling"] == "strong" and w["c3_bands"]["items_rank0"] == 16
    assert w["c3_bands"]["pixels_per_pass"] == 16 * 256 * 128
    assert "prog" not in w                                                    # one-rank extra


def test_gpus_flag_starts_the_ranks():
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2", "--dry", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err
    assert len(lines) == 1, lines                                             # rank 0 alone prints
    _check_two_rank_line(lines[0])


def test_under_torch_distributed_run():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rc, lines, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry"])
    assert rc == 0, err
    assert len(lines) == 1, lines
    _check_two_rank_line(lines[0])


def test_band_sharded_headline_and_single_rank():
    rc, lines, err = _run([sys.executable, BENCH, "--gpus", "2", "--dry", "--workload", "c3", "--shard", "bands", "--images", "4"])
    assert rc == 0, err
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["pixels_per_pass"] == 4 * 256 * 128 and "workloads" not in d
    rc, lines, err = _run([sys.executable, BENCH, "--dry"])
    assert rc == 0, err
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and set(d["workloads"]) == {"c4", "c5", "prog"}


def test_a_failing_rank_fails_the_launcher():
    rc, lines, _ = _run([sys.executable, BENCH, "--gpus", "2", "--dry", "--dry-size", "nonsense"])
    assert rc != 0 and not lines
