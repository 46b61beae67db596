"""bench.py --gpus N must really start N ranks (the flag was parsed and ignored).  CPU rehearsal: the
launcher's children rendezvous over gloo (--dry-run) instead of touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_starts_that_many_ranks():
    for n in (1, 2, 3):
        r = _run(["--gpus", str(n), "--dry-run"])
        assert r.returncode == 0, r.stderr
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["ranks"] == list(range(n)) and out["local_ranks"] == list(range(n))


def test_launcher_env_is_respected_and_mismatch_is_an_error():
    # started by torch.distributed.run: WORLD_SIZE set -> no second launcher level
    r = _run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout)["n_gpus"] == 1
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_a_failing_rank_fails_the_launch_quickly():
    # no GPU here: every rank exits with the "needs an MI355X" message; the parent must report failure, not hang
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "1"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""
