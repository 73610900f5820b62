"""The RCCL leg of the data-parallel path on a 1-GPU box: VTS_DDP_FORCE=1 creates the process group (backend nccl = RCCL) with
one rank and runs the gradient-bucket all-reduces between the captured segments of the step.  It caught a real failure: the
process group's watchdog thread queries events while the step is being captured, which the global capture mode forbids
(the capture now uses the thread-local mode).  The bench contract is checked on the way: exactly ONE line on stdout (RCCL prints
a version banner to fd 1 when the communicator is created)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env_extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "3"] + (["--no_cpu_baseline"] if env_extra else []) + extra,
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [["--size", "256", "--batch", "2"], ["--size", "256", "--batch", "2", "--no_graph"],
                                   ["--model", "pix2pixHD", "--batch", "8"]])
def test_step_with_rccl_process_group_single_rank(extra):
    port = {"--no_graph": "29561", "--model": "29562"}.get(extra[-1] if extra[-1] == "--no_graph" else extra[0], "29560")
    out = _bench(extra, {"VTS_DDP_FORCE": "1", "MASTER_PORT": port})
    assert out["config"]["losses_finite"] and out["value"] > 0 and out["n_gpus"] == 1
    assert out["config"]["hip_graph"] == ("--no_graph" not in extra)


def test_bench_line_schema():
    """the contract of the driver: one JSON line with the agreed keys, roofline and (at N=1) cpu_baseline objects"""
    out = _bench(["--size", "256", "--batch", "1"], {})
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline"}
    assert need <= set(out), need - set(out)
    assert out["metric"] == "train_images_per_sec" and out["unit"] == "images/s" and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["dtype"] == "f32" and out["data"] == "synthetic" and "workload" in out["config"] and out["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"]) and out["roofline"]["bound"] in ("hbm", "mfma")
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-9
    assert abs(out["value"] - out["n_gpus"] * 1 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
