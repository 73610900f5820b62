"""RCCL on TWO DEVICES (skipped on a 1-GPU box): the real training step with one process per GPU, backend nccl (= RCCL over xGMI).

What only this test can see (tests/test_ddp_step_gpu.py shares one GPU over gloo, tests/test_rccl_gpu.py has one rank):
  * every rank binds cuda:LOCAL_RANK -- model tensors, the C library's stream and the RCCL communicator on the same device;
  * the collectives really ran between two ranks (dist.get_world_size() == 2, backend nccl);
  * replicas stay bit-identical through an eager step, the HIP-graph capture (thread-local capture mode with RCCL's watchdog alive
    on two ranks) and a graph replay, while every rank trains on its own sample;
  * `python bench.py --gpus 2` starts its ranks itself and prints one JSON line with the `comm` block.
Replaces what the reference gets from nn.DataParallel (/root/reference/models/base_model.py:104-108)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "visual-tactile-synthesis_amd")

needs2 = pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs")

WORKER = r'''
import os, sys, random
sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from vts import ddp, lib as L
rank, world = ddp.init_from_env("cuda")
local = int(os.environ["LOCAL_RANK"])
assert world == 2 and dist.get_world_size() == 2 and dist.get_backend() == "nccl"
assert torch.cuda.current_device() == local
from tests.test_ddp_step_gpu import build, sample_and_draws, SEED
from tests.test_step_gpu import load_test_weights
out_dir = sys.argv[1]
model, opt = build()
assert model.device.index == local, (model.device, local)
load_test_weights(model, SEED if rank == 0 else SEED + 10)
model.parallelize()
assert ddp.active() and set(model.ddp.buckets) == {"D", "D2", "G_dec", "G_enc"}
for b in model.ddp.buckets.values():
    assert b.buf.device.index == local
batch, _ = sample_and_draws(rank)
torch.manual_seed(100 + rank)
random.seed(100 + rank)
snaps = []
for it in range(3):               # eager, capture, replay
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    assert model.fake_I.device.index == local
    snaps.append({n: getattr(model, "flat" + n).flat.cpu().clone() for n in ("G", "D", "D2")})
assert model._graphs is not None and len(model._graphs) == 5
# the collective itself: a tensor that differs per rank must come back as the sum on both
t = torch.full((1 << 16,), float(rank + 1), device="cuda")
dist.all_reduce(t)
assert float(t[0]) == 3.0 and float(t[-1]) == 3.0
torch.save({"snaps": snaps, "losses": model.get_current_losses(), "device": local}, os.path.join(out_dir, "rank%%d.pt" %% rank))
dist.barrier()
dist.destroy_process_group()
'''


@needs2
def test_two_devices_real_step_replicas_identical(tmp_path):
    script = WORKER % dict(pkg=PKG, root=ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), NCCL_DEBUG="INFO")
        env.pop("VTS_DDP_BACKEND", None)
        procs.append(subprocess.Popen([sys.executable, "-c", script, str(tmp_path)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    logs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err.decode()[-4000:]
        logs.append(out.decode() + err.decode())
    # RCCL itself saw both ranks: its NCCL_DEBUG=INFO init lines say "nranks 2" (the parse is validated on one rank by
    # tests/test_rccl_gpu.py::test_rccl_debug_log_reports_the_rank_count)
    from tests.test_rccl_gpu import nranks_seen
    for log in logs:
        assert nranks_seen(log) == [2], log[-2000:]
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(2))
    assert (r0["device"], r1["device"]) == (0, 1)
    for it in range(3):
        for n in ("G", "D", "D2"):
            assert torch.equal(r0["snaps"][it][n], r1["snaps"][it][n]), "replicas differ after step %d (net%s)" % (it, n)
            assert torch.isfinite(r0["snaps"][it][n]).all()
    assert r0["losses"] != r1["losses"]      # different samples per rank


@needs2
def test_bench_spawns_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--size", "256", "--batch", "2"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["losses_finite"]
    assert "comm" in out and "exposed_allreduce_ms_per_step" in out["comm"] and set(out["comm"]["buckets"]) == {"D", "D2", "G_dec", "G_enc"}
    c = out["comm"]      # the record proves its rank count by itself: RCCL's own init log, one clock and one device per rank
    assert c["nranks_seen"] == 2 and c["ranks_gathered"] == 2 and len(c["devices_seen"]) == 2
    assert 0 < c["ms_per_step_per_rank"]["min"] <= c["ms_per_step_per_rank"]["max"]
    ab = c["collective_ab"]    # warm-up A/B of all_reduce vs the library's reduce-scatter + all-gather, the latter checked first
    assert ab is not None and ab["torch_all_reduce_ms_per_step"] > 0 and ab["chosen"] in ("torch", "direct")
    assert ab["direct_checked"] or "direct_error" in ab or ab["direct_ms_per_step"] is None


def test_bench_refuses_more_gpus_than_the_node_has():
    """the self-spawn path on any box: asking for more GPUs than exist is a clear error, not a hang"""
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and ("exposes %d GPU" % (n - 1)) in r.stderr.decode()
