"""pix2pixHD under data parallelism with the generator's gradient cut into buckets (the 730 MB bucket of the reference's
ngf-64 generator in >= 8 pieces, each all-reduced as soon as the backward has written it).  Two ranks share the test box's one GPU
(gloo through VTS_DDP_BACKEND, as tests/test_ddp_step_gpu.py): the step with K buckets / K backward stages must leave exactly the
weights of the step with ONE bucket behind the whole backward -- eagerly and with the stages captured as HIP graphs -- and both ranks
must hold identical replicas."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "visual-tactile-synthesis_amd")

WORKER = r'''
import os, sys
sys.path.insert(0, %(pkg)r); sys.path.insert(0, %(root)r)
import torch
from vts import ddp
rank, world = ddp.init_from_env("cuda")
from tests.test_pix2pixHD_gpu import FLAGS, p2p_batch, load_weights
from models import create_model
from options.train_options import TrainOptions
mode, out_dir = sys.argv[1], sys.argv[2]
opt = TrainOptions(cmd_line=FLAGS).parse()
model = create_model(opt)
model.setup(opt)
load_weights(model, 60 if rank == 0 else 70)      # rank 1 starts from other weights: parallelize() replaces them
model.parallelize()
model.train()
K = len(model.flatG.cuts) + 1
assert ddp.active() and sorted(model.ddp.buckets) == sorted(["D", "D2"] + (["G_%%d" %% j for j in range(K)] if K > 1 else ["G"])), model.ddp.buckets.keys()
opt.use_hip_graph = mode == "graph"
batch = p2p_batch(4, 32, 90 + rank)                 # every rank its own samples
steps = 3 if mode == "graph" else 1                # eager, capture, replay
for _ in range(steps):
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
torch.cuda.synchronize()
if mode == "graph":
    assert model._graphs is not None and len(model._graphs) == (K + 2 if K > 1 else 3), len(model._graphs)
torch.save({"K": K, "flat": {n: getattr(model, "flat" + n).flat.cpu() for n in ("G", "D", "D2")}, "grad": model.flatG.grad.cpu(),
            "sizes": [int(b.buf.numel()) for k, b in sorted(model.ddp.buckets.items()) if k.startswith("G")]},
           os.path.join(out_dir, "rank%%d.pt" %% rank))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def _run(mode, out_dir, port, buckets):
    os.makedirs(str(out_dir), exist_ok=True)
    script = WORKER % dict(pkg=PKG, root=ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   VTS_DDP_BACKEND="gloo", VTS_G_BUCKETS=str(buckets), VTS_G_BUCKET_MIN_MB="0.001")
        procs.append(subprocess.Popen([sys.executable, "-c", script, mode, str(out_dir)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        try:
            _, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err.decode()[-4000:]
    return [torch.load(os.path.join(str(out_dir), "rank%d.pt" % r)) for r in range(2)]


@pytest.mark.parametrize("mode,port", [("eager", 29641), ("graph", 29651)])
def test_chunked_generator_bucket_equals_the_single_bucket(tmp_path, mode, port):
    one = _run(mode, tmp_path / "one", port, 1)
    many = _run(mode, tmp_path / "many", port + 2, 8)
    assert one[0]["K"] == 1 and many[0]["K"] >= 4, (one[0]["K"], many[0]["K"])
    assert sum(many[0]["sizes"]) == one[0]["sizes"][0]                      # the buckets tile the flat gradient
    for n in ("G", "D", "D2"):
        assert torch.equal(many[0]["flat"][n], many[1]["flat"][n])           # replicas identical
        assert torch.equal(many[0]["flat"][n], one[0]["flat"][n]), n         # ... and equal to the single-bucket step, bit for bit
    assert torch.equal(many[0]["grad"], one[0]["grad"])
