"""bench.py's launch contract (no GPU): `python bench.py --gpus N` starts its own N ranks when no launcher did, the
`python -m torch.distributed.run ... bench.py --gpus N` form the driver uses keeps working, and a rank count that does not match --gpus is
a clear error, not an assert.  GC_BENCH_DRY=1 stops after the rendez-vous (gloo) -- the measurement itself needs MI355X GPUs."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(GC_BENCH_DRY="1", GC_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    return env


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_bare_gpus_2_self_launches_its_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _last_json(r.stdout) == {"dry_run": True, "world": 2, "sum_of_rank_plus_1": 3.0}


def test_torchrun_form_still_works():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _last_json(r.stdout)["world"] == 2


def test_mismatched_rank_count_is_a_clear_error():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr) and "AssertionError" not in r.stderr
