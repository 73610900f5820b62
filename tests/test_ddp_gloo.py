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


def test_flat_params_chunk_cuts_at_convolution_weights():
    """FlatParams.chunk (vts/optim.py): bucket boundaries are starts of 4-d parameters, in order, the buckets tile the gradient"""
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.nn as nn

    from vts.optim import FlatParams

    layers = []
    for i in range(12):
        layers += [nn.Conv2d(8, 8, 3), nn.BatchNorm2d(8)]
    net = nn.Sequential(*layers)
    flat = FlatParams(net)
    cut_params = flat.chunk(8, min_floats=1)
    assert 4 <= len(flat.cuts) + 1 <= 8 and flat.cuts == sorted(set(flat.cuts))
    starts, o = {}, 0
    for p in flat.params:
        starts[o] = p
        o += p.numel()
    assert all(starts[c].dim() == 4 and starts[c] is q for c, q in zip(flat.cuts, cut_params))
    b = flat.buckets("G")
    assert list(b) == ["G_%d" % k for k in range(len(flat.cuts) + 1)] and sum(v.numel() for v in b.values()) == flat.numel
    assert b["G_0"].data_ptr() == flat.grad.data_ptr()
    sizes = [v.numel() for v in b.values()]
    assert max(sizes) <= 2 * min(sizes) + 8 * 8 * 9          # roughly equal
    # too small to be worth several collectives: one bucket
    flat2 = FlatParams(nn.Sequential(nn.Conv2d(2, 2, 3), nn.Conv2d(2, 2, 3)))
    assert flat2.chunk(8) == [] and list(flat2.buckets("G")) == ["G"]


def _chunk_worker(rank, world, port, out):
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import torch.nn as nn

    from vts import ddp
    from vts.optim import FlatParams

    ddp.init_from_env("cpu")

    class M:
        model_names = ["G"]

    res = {}
    for tag, k in (("one", 1), ("many", 8)):
        torch.manual_seed(5)
        m = M()
        m.netG = nn.Sequential(*[nn.Conv2d(4, 4, 3) for _ in range(10)])
        m.flatG = FlatParams(m.netG)
        m.flatG.chunk(k, min_floats=1)
        state = ddp.attach(m)
        g = torch.Generator().manual_seed(100 + rank)
        m.flatG.grad.copy_(torch.randn(m.flatG.numel, generator=g))
        names = sorted(state.buckets, key=lambda s: -int(s.split("_")[1]) if "_" in s else 0)     # backward order: last bucket first
        for nme in names:
            state.buckets[nme].start()
        for nme in names:
            state.buckets[nme].wait()
        res[tag] = (m.flatG.grad.clone(), len(names))
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_chunked_buckets_equal_the_monolithic_allreduce():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_chunk_worker, args=(world, 29621, out), nprocs=world, join=True)
    for r in (0, 1):
        assert out[r]["many"][1] >= 4 and out[r]["one"][1] == 1
        assert torch.equal(out[r]["many"][0], out[r]["one"][0])
    assert torch.equal(out[0]["many"][0], out[1]["many"][0])


# ---- world 8 / world 4 (BASELINE configs 2 and 3): no box here has that many GPUs, so the bucket logic -- cuts at layer boundaries,
# buckets whose length is not a multiple of the world size, the chunks started in backward order, the reduce-scatter + all-gather slice
# plan of the C library with its ragged tail -- runs over gloo on the real generator classes built on the CPU -------------------------

def _wide_worker(rank, world, port, out, kind):
    for p in (PKG, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import torch.distributed as dist

    from models import networks
    from vts import ddp
    from vts.optim import FlatParams

    ddp.init_from_env("cpu")

    class M:
        pass

    torch.manual_seed(100 + rank)          # every rank builds other weights: attach() must leave rank 0's everywhere
    m = M()
    if kind == "pix2pixHD":                # config 3: the coarse generator of pix2pixHD (reference networks.py:1952-1980), narrow so that the CPU holds 4 of them
        m.model_names = ["G"]
        m.netG = networks.GlobalGenerator(3, 3, ngf=6, n_downsampling=4, n_blocks=9, norm="batch")
        m.flatG = FlatParams(m.netG)
        m.flatG.chunk(8, min_floats=1)     # as models/pix2pixHD_model.py does (VTS_G_BUCKETS 8)
    else:                                  # config 2: the U-Net generator with its decoder / encoder buckets, D and D2 as one bucket each
        m.model_names = ["G", "D", "D2"]
        m.netG = networks.define_G(4, 5, 10, "unet256_custom", norm="instance", opt=None)
        m.netD = networks.MultiscaleDiscriminator(4, ndf=8)
        m.netD2 = networks.MultiscaleDiscriminator(7, ndf=8)
        m.flatG = FlatParams(m.netG, first=lambda k: k.startswith("up"))       # decoder first, as models/sinskitG_model.py lays it out
        m.flatD, m.flatD2 = FlatParams(m.netD), FlatParams(m.netD2)
    state = ddp.attach(m)
    res = {"w0": {n: getattr(m, "flat" + n).flat.clone() for n in m.model_names}, "scale": state.grad_scale}
    # per-rank gradients: integers, so that any summation order gives the same bits and the expected sum is exact
    for i, n in enumerate(m.model_names):
        f = getattr(m, "flat" + n)
        g = torch.Generator().manual_seed(1000 * i + rank)
        f.grad.copy_(torch.randint(-64, 64, (f.numel,), generator=g).float())
    # the order the captured step uses: the generator's chunks from the LAST one (written first by the backward) down to the first,
    # every one started before the previous has finished; the discriminators' buckets first (they are complete before the G backward)
    def order(k):
        tail = k.split("_")[-1]
        return -int(tail) if tail.isdigit() else {"dec": 0, "enc": 1}.get(tail, 0)      # G_dec is complete before G_enc
    names = sorted(state.buckets, key=lambda k: (k.startswith("G_") or k == "G", order(k)))
    for nme in names:
        state.buckets[nme].start()
    for nme in reversed(names):            # waits in another order than the starts: the handles are independent
        state.buckets[nme].wait()
    res["names"] = names
    res["sizes"] = {k: int(b.buf.numel()) for k, b in state.buckets.items()}
    res["grad"] = {n: getattr(m, "flat" + n).grad.clone() for n in m.model_names}
    # the direct collective's plan (vts_allreduce_slice_plan: the arithmetic vts_allreduce_flat_async runs on RCCL) replayed over gloo:
    # rank r reduces its slice, the slices are gathered back, the ragged tail is all-reduced
    direct = {}
    for i, n in enumerate(m.model_names):
        f = getattr(m, "flat" + n)
        g = torch.Generator().manual_seed(1000 * i + rank)
        buf = torch.randint(-64, 64, (f.numel,), generator=g).float()
        plans = [ddp.slice_plan(f.numel, world, r) for r in range(world)]
        for r, (off, chunk, _, _) in enumerate(plans):
            if chunk:
                dist.reduce(buf[off:off + chunk], dst=r)             # reduce-scatter: slice r ends up summed on rank r
        for r, (off, chunk, _, _) in enumerate(plans):
            if chunk:
                dist.broadcast(buf[off:off + chunk], src=r)          # all-gather
        _, _, toff, tail = plans[rank]
        if tail:
            dist.all_reduce(buf[toff:toff + tail])
        direct[n] = buf
        res.setdefault("plans", {})[n] = plans
    res["direct"] = direct
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def _expected_sum(numel, world, i):
    tot = torch.zeros(numel)
    for r in range(world):
        g = torch.Generator().manual_seed(1000 * i + r)
        tot += torch.randint(-64, 64, (numel,), generator=g).float()
    return tot


def _check_wide(out, world, names_expected_prefix):
    r0 = out[0]
    for n, w in r0["w0"].items():
        for r in range(1, world):
            assert torch.equal(out[r]["w0"][n], w), ("replica", n, r)
    assert r0["scale"] == 1.0 / world
    for i, n in enumerate(r0["grad"]):
        want = _expected_sum(r0["grad"][n].numel(), world, i)
        for r in range(world):
            assert torch.equal(out[r]["grad"][n], want), ("bucketed all-reduce", n, r)
            assert torch.equal(out[r]["direct"][n], want), ("reduce-scatter + all-gather plan", n, r)
        plans = r0["plans"][n]
        numel = r0["grad"][n].numel()
        chunk = numel // world
        assert [p[0] for p in plans] == [r * chunk for r in range(world)] and all(p[1] == chunk for p in plans)
        assert all(p[2] == chunk * world and p[3] == numel - chunk * world for p in plans)
    return r0


def test_eight_rank_buckets_of_the_headline_generator_with_ragged_tails():
    """BASELINE config 2's exchange (8 ranks) over gloo: D / D2 buckets and the generator's decoder / encoder buckets of the REAL network
    classes; none of the bucket lengths is a multiple of 8, so the slice plan has a tail everywhere"""
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_wide_worker, args=(world, 29631, out, "skitG"), nprocs=world, join=True)
    r0 = _check_wide(out, world, "G")
    assert set(r0["sizes"]) == {"D", "D2", "G_dec", "G_enc"} and r0["names"][-2:] == ["G_dec", "G_enc"]
    assert r0["sizes"]["G_dec"] + r0["sizes"]["G_enc"] == r0["grad"]["G"].numel()
    assert any(v % world for v in r0["sizes"].values())                       # ragged buckets
    assert any(r0["grad"][n].numel() % world for n in r0["grad"])             # ... and ragged tails of the slice plan


def test_four_rank_chunked_pix2pixHD_generator_buckets_in_backward_order():
    """BASELINE config 3's exchange (4 ranks): the pix2pixHD coarse generator's gradient as up to 8 chunks cut at convolution weights,
    started from the last chunk down (the order its backward completes them)"""
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_wide_worker, args=(world, 29641, out, "pix2pixHD"), nprocs=world, join=True)
    r0 = _check_wide(out, world, "G")
    gnames = [k for k in r0["names"] if k.startswith("G")]
    assert len(gnames) >= 4 and gnames == ["G_%d" % j for j in range(len(gnames) - 1, -1, -1)]       # backward order
    assert sum(r0["sizes"][k] for k in gnames) == r0["grad"]["G"].numel()
