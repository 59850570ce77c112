"""`python bench.py --gpus N` from a plain shell must start its own rank processes (missing #1): the reference
gets N ranks from one command through Lightning's strategy='ddp' (src/hardware_utils.py:86-95).  CPU-only: the children run
bench.py's `--launch-check` leg (gloo group, no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


def test_bench_self_launch_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["world"] == 2 and out["n_gpus"] == 2 and out["rank_sum"] == 1.0 and out["backend"] == "gloo"


def test_bench_under_torchrun_env_is_a_rank():
    """With the torchrun variables present bench.py must NOT spawn: it is rank 0 of a world of 1 here."""
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["world"] == 1


def test_launcher_propagates_failure(tmp_path):
    from vit_amd.launch import launch_ranks

    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if r == 1:\n    sys.exit(7)\n"
                      "time.sleep(30)\n")  # rank 0 would hang: the launcher must stop it
    import time

    t0 = time.time()
    rc = launch_ranks(2, str(script), [])
    assert rc == 7
    assert time.time() - t0 < 20


def test_launcher_world_mismatch_fails():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=300,
                       env=dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert r.returncode != 0
