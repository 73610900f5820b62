"""world_size-2 CPU (gloo) checks of the data-parallel glue: replica broadcast, bucketed gradient
all-reduce with the 1/world scale folded into the optimiser, per-rank data sharding."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "visual-tactile-synthesis_amd")


def _worker(rank, world, port, out):
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import torch.nn as nn

    from vts import ddp
    from vts.optim import FlatParams

    r, w = ddp.init_from_env("cpu")
    assert (r, w) == (rank, world)

    class M:  # minimal stand-in for a BaseModel with one network
        model_names = ["G"]

    torch.manual_seed(rank)  # replicas start different on purpose
    m = M()
    m.netG = nn.Sequential(nn.Conv2d(2, 3, 3), nn.BatchNorm2d(3))
    m.flatG = FlatParams(m.netG)
    state = ddp.attach(m)
    w0 = m.flatG.flat.clone()
    m.flatG.grad.fill_(float(rank + 1))
    state.buckets["G"].start()
    state.buckets["G"].wait()
    from data.synthetic_dataset import SyntheticDataset
    from types import SimpleNamespace

    ds = SyntheticDataset(SimpleNamespace(crop_size=32, synthetic_size=0, data_len=2, batch_size_G2=4, batch_size_G2_val=4,
                                          isTrain=True, rank=rank, data_seed=1, cache_samples=False))
    out[rank] = dict(w0=w0, grad=m.flatG.grad.clone(), scale=state.grad_scale, s=ds[0]["S"].clone(),
                     bn=list(m.netG.buffers())[0].clone())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucket_allreduce_and_broadcast():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, 29611, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["w0"], b["w0"])                     # rank 0's replica everywhere
    assert torch.equal(a["grad"], b["grad"]) and float(a["grad"][0]) == 3.0   # 1 + 2 summed
    assert a["scale"] == 0.5                                  # mean = sum * 1/world, applied in the Adam kernel
    assert not torch.equal(a["s"], b["s"])                    # ranks draw different samples
