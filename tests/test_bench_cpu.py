"""bench.py's launcher logic that needs no GPU: `--gpus N` outside torchrun starts the ranks itself -- and refuses cleanly when the
node has fewer devices (on this CPU box: zero)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_spawn_refuses_without_devices():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    assert "--gpus 64 but this node exposes" in r.stderr.decode()
    assert r.stdout.decode().strip() == ""      # no half-written JSON line


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in r.stderr.decode()
