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


def _bench(extra, env_extra, want_stderr=False):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "3"] + (["--no_cpu_baseline"] if env_extra else []) + extra,
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return (json.loads(lines[0]), r.stderr.decode()) if want_stderr else json.loads(lines[0])


def nranks_seen(log):
    """rank counts RCCL reports in its NCCL_DEBUG=INFO lines ("... rank 0 nranks 2 cudaDev 0 ...")"""
    import re
    return sorted({int(m) for m in re.findall(r"nranks (\d+)", log)})


def test_rccl_debug_log_reports_the_rank_count():
    """the check tests/test_rccl2_gpu.py applies on two devices ("RCCL saw both ranks"), validated here on the one-rank group: bench.py
    sends the communicator's INFO lines to a file of its own (NCCL_DEBUG_FILE, chosen before torch loads RCCL), parses `nranks N` out of
    them and reports it -- with every rank's clock and device identity gathered over the backend -- in the line's `comm` block"""
    out, err = _bench(["--size", "256", "--batch", "1"], {"VTS_DDP_FORCE": "1", "MASTER_PORT": "29564"}, want_stderr=True)
    c = out["comm"]
    assert out["n_gpus"] == 1 and c["nranks_seen"] == 1 and c["ranks_gathered"] == 1 and len(c["devices_seen"]) == 1, (c, err[-2000:])
    assert c["ms_per_step_per_rank"]["min"] == c["ms_per_step_per_rank"]["max"] > 0


def test_bench_collective_ab_runs_on_the_one_rank_group():
    """the warm-up A/B of bench.py (torch.distributed all_reduce vs the library's reduce-scatter + all-gather; automatic at world > 1) forced
    on the one-rank group: the direct path is checked against all_reduce on a 1000003-element buffer, both are timed, one is chosen"""
    out, err = _bench(["--size", "256", "--batch", "1", "--steps", "12"], {"VTS_DDP_FORCE": "1", "VTS_DDP_AB": "force", "MASTER_PORT": "29566"}, want_stderr=True)
    ab = out["comm"]["collective_ab"]
    assert ab is not None and ab["direct_checked"] is True and "direct_error" not in ab, (ab, err[-2000:])
    assert ab["torch_all_reduce_ms_per_step"] > 0 and ab["direct_ms_per_step"] > 0 and ab["chosen"] in ("torch", "direct")
    assert out["comm"]["collective"].startswith("reduce-scatter" if ab["chosen"] == "direct" else "torch.distributed")


def test_direct_reduce_scatter_all_gather_collective_single_rank(tmp_path):
    """VTS_DDP_DIRECT=1: the buckets go through the library's own RCCL communicator (reduce-scatter + all-gather + tail all-reduce on a
    side stream, include/vts.h: vts_allreduce_flat_async / _wait) instead of torch.distributed's all_reduce -- one rank: the sum over
    ranks is the buffer itself, and the real step still runs (eager, capture, replay) with finite losses"""
    script = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from vts import ddp
rank, world = ddp.init_from_env("cuda")
assert ddp.active() and ddp.DIRECT
for n in (1, 5, 4096, (1 << 20) + 3):
    t = torch.randn(n, device="cuda")
    ref = t.clone()
    b = ddp.GradBucket(t)
    b.start()
    b.wait()
    torch.cuda.synchronize()
    assert torch.equal(t, ref), n
print("direct collective ok")
dist.destroy_process_group()
""" % os.path.join(ROOT, "visual-tactile-synthesis_amd")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29565", VTS_DDP_FORCE="1", VTS_DDP_DIRECT="1")
    r = subprocess.run([sys.executable, "-c", script], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and "direct collective ok" in r.stdout.decode(), r.stderr.decode()[-3000:]
    out = _bench(["--size", "256", "--batch", "2"], {"VTS_DDP_FORCE": "1", "VTS_DDP_DIRECT": "1", "MASTER_PORT": "29566"})
    assert out["config"]["losses_finite"] and out["value"] > 0


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
