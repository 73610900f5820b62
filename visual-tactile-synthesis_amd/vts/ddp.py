"""Data-parallel glue: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The hot path shards by image: every rank holds full replicas of G, D and D2
and draws its own samples.  The only exchange is a sum-all-reduce of ONE flat fp32 gradient
bucket per network (G 5.9 MB, D and D2 0.54 MB each); the division by world size is folded
into the fused Adam kernel (`grad_scale`).  Each all-reduce is issued asynchronously at the end of the
captured segment that completes its bucket and waited for just before that network's Adam step
(models/sinskitG_model.py:_segments): the D / D2 buckets travel under the generator's L1 terms, the
generator's DECODER bucket (its gradients are complete halfway through the backward) under the
encoder's backward; what stays exposed is the encoder bucket (bench.py reports the exposed time).  BatchNorm statistics stay per rank, like per-replica BN under the
reference's nn.DataParallel (base_model.py:104-108).
"""
import os

import torch
import torch.distributed as dist


# VTS_DDP_FORCE=1: create the process group and run the bucket all-reduces even with ONE rank, so that the RCCL path
# (async all-reduce on RCCL's stream between the captured segments of the step) can be exercised on a 1-GPU box.
FORCE = os.environ.get("VTS_DDP_FORCE", "0") == "1"


# measurement switch (bench.py): skip the all-reduces so that (time with) - (time without) = the exposed communication per step
COMM_OFF = False


def active():
    return _active()


def _active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def init_from_env(device_type="cuda"):
    """Initialise the default process group from torchrun-style env vars (no-op for 1 rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world <= 1 and not FORCE) or dist.is_initialized():
        return int(os.environ.get("RANK", "0")), world
    if world <= 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
    rank = int(os.environ["RANK"])
    if device_type == "cuda":
        # one GPU per rank; VTS_DDP_BACKEND=gloo (ranks sharing a device) exists to exercise this path on a 1-GPU box
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group(backend=os.environ.get("VTS_DDP_BACKEND", "nccl"), rank=rank, world_size=world)
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return rank, world


# VTS_DDP_DIRECT=1: the buckets travel as an explicit reduce-scatter + all-gather on the C library's own RCCL communicator and side stream
# (include/vts.h: vts_allreduce_flat_async / _wait) instead of torch.distributed's all_reduce.  Same GradBucket interface; off by default
# (never compared on a multi-GPU node).
DIRECT = os.environ.get("VTS_DDP_DIRECT", "0") == "1"
_comm = None


def direct_comm():
    """the library's communicator (created once per process; the 128-byte id travels over the default process group)"""
    global _comm
    if _comm is None:
        import ctypes as C

        from . import lib as L
        lib = L.load()
        ident = torch.zeros(128, dtype=torch.uint8)
        if dist.get_rank() == 0:
            raw = (C.c_ubyte * 128)()
            L.check(lib.vts_comm_unique_id(C.cast(raw, C.c_void_p)), "vts_comm_unique_id")
            ident = torch.tensor(list(raw), dtype=torch.uint8)
        dev = torch.device("cuda", torch.cuda.current_device())
        t = ident.to(dev) if dist.get_backend() == "nccl" else ident
        dist.broadcast(t, 0)
        raw = (C.c_ubyte * 128)(*t.cpu().tolist())
        handle = C.c_void_p()
        L.check(lib.vts_comm_init(C.cast(raw, C.c_void_p), dist.get_rank(), dist.get_world_size(), C.byref(handle)), "vts_comm_init")
        _comm = handle
    return _comm


def slice_plan(n, world, rank):
    """(offset, chunk, tail_offset, tail) of the direct collective for an n-element bucket: rank `rank` reduces [offset, offset + chunk),
    everyone all-reduces the remainder [tail_offset, tail_offset + tail).  The C library's own arithmetic (vts_allreduce_slice_plan:
    host only, no GPU), so that CPU tests can replay the plan over gloo at world sizes no test box has GPUs for."""
    import ctypes as C

    from . import lib as L
    o, c, to, t = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    L.check(L.load().vts_allreduce_slice_plan(int(n), int(world), int(rank), C.byref(o), C.byref(c), C.byref(to), C.byref(t)), "vts_allreduce_slice_plan")
    return o.value, c.value, to.value, t.value


class GradBucket:
    """Asynchronous all-reduce of one flat gradient buffer."""

    def __init__(self, flat_grad):
        self.buf = flat_grad
        self.work = None
        self.direct = False

    def start(self):
        if _active() and not COMM_OFF:
            if DIRECT and self.buf.is_cuda:
                from . import lib as L
                L.check(L.load().vts_allreduce_flat_async(direct_comm(), self.buf.data_ptr(), self.buf.numel(), L.stream()), "vts_allreduce_flat_async")
                self.direct = True
            else:
                self.work = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, async_op=True)

    def wait(self):
        if self.direct:
            from . import lib as L
            L.check(L.load().vts_allreduce_flat_wait(direct_comm(), L.stream()), "vts_allreduce_flat_wait")
            self.direct = False
        if self.work is not None:
            self.work.wait()
            self.work = None


class DDPState:
    def __init__(self, world, buckets):
        self.world = world
        self.buckets = buckets
        self.grad_scale = 1.0 / world


def broadcast_module(module, src=0):
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def attach(model):
    """Called by BaseModel.parallelize(): replicate rank 0's weights, create gradient buckets."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    buckets = {}
    for name in model.model_names:
        net = getattr(model, "net" + name)
        if world > 1 or _active():
            flat = getattr(model, "flat" + name, None)
            if flat is not None:
                dist.broadcast(flat.flat, 0)
                for b in net.buffers():
                    dist.broadcast(b, 0)
            else:
                broadcast_module(net, 0)
        flat = getattr(model, "flat" + name, None)
        if flat is not None:
            for bname, view in flat.buckets(name).items():
                buckets[bname] = GradBucket(view)
    return DDPState(world, buckets)
