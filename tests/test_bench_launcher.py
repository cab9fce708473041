"""`python bench.py --gpus N` must bring up its N ranks itself: the launcher, without a GPU (--dry-launch)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout     # ONE json line, from rank 0
    return json.loads(lines[0])


def test_plain_command_launches_two_ranks():
    d = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-launch"])
    assert d["dry_launch"] and d["n_gpus"] == 2
    assert d["ranks"] == [0, 1] and d["local_ranks"] == [0, 1] and d["pids"] == 2


def test_under_a_launcher_it_is_one_rank():
    # the driver's form: python -m torch.distributed.run ... bench.py --gpus N: no second level of launching
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-launch"],
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["local_ranks"] == [0, 1, 2] and d["pids"] == 3


def test_single_rank_dry_launch():
    d = run(["--dry-launch"])
    assert d["n_gpus"] == 1 and d["ranks"] == [0]
