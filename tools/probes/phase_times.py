"""Where the replayed step's time is, phase by phase: the step is captured as SEVEN graphs instead of three (cuts between the generator
forward + stacks, the discriminator lanes, Adam(D) + the generator step's discriminator passes, the decoder half of the generator backward,
the encoder half, Adam(G)) and HIP events between the replays time each phase in place.  The cuts cost a graph launch each (the sum is
~0.1 - 0.2 ms above the three-graph step).  python tools/probes/phase_times.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
from vts import engine as _engine  # noqa: E402

_engine.D_CHAINS = False      # the phases below are those of the JOINED schedule (round 6's default chains the phases: no cut points between them)
model, opt = bench.build_model(1024, batch, "skitG")
opt.use_hip_graph = False
data = bench.make_batch(1024, batch, 0, opt.style_code_dim)
model.set_input(data, phase="train")
for _ in range(3):
    model.optimize_parameters(epoch=1)
torch.cuda.synchronize()

fwd = model._forward_and_stacks
g_bwd = model._g_backward


def seg_fwd():
    fwd()


def seg_d():
    model._forward_and_stacks = lambda *a, **k: None
    try:
        model._seg_d_updates()
    finally:
        model._forward_and_stacks = fwd


def seg_gd():
    model._g_backward = lambda part="all": None
    try:
        if not getattr(model, "_g_pre_done", False):
            model._seg_g_pre()
        model._g_pre_done = False
        model._seg_g_main()
    finally:
        model._g_backward = g_bwd


segs = [("generator forward + post-processing + patch stacks", seg_fwd), ("discriminator updates (lanes) [+ generator L1 terms]", seg_d),
        ("Adam(D, D2) + discriminator passes of the generator step", seg_gd), ("generator backward: decoder half (all up blocks)", lambda: g_bwd("decoder")),
        ("generator backward: encoder half", model._seg_g_enc), ("Adam(G)", model._seg_adam_g)]
from vts import ops  # noqa: E402

pool = torch.cuda.graph_pool_handle()
stream = torch.cuda.Stream()
graphs = []
ops.freeze_ws(("probe", "train"))
ops.step_begin(model._loss_buf, model._step_counters)
for name, fn in segs:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, pool=pool, stream=stream, capture_error_mode="thread_local"):
        fn()
    graphs.append(g)
torch.cuda.synchronize()
reps = 100
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(segs) + 1)] for _ in range(reps)]
for r in range(-5, reps):
    ops.step_begin(model._loss_buf, model._step_counters)
    if r >= 0:
        ev[r][0].record()
    for i, g in enumerate(graphs):
        g.replay()
        if r >= 0:
            ev[r][i + 1].record()
torch.cuda.synchronize()
tot = 0.0
for i, (name, _) in enumerate(segs):
    ts = sorted(ev[r][i].elapsed_time(ev[r][i + 1]) for r in range(reps))
    med = ts[len(ts) // 2]
    tot += med
    print("%-62s %.3f ms (min %.3f)" % (name, med, ts[0]))
print("%-62s %.3f ms" % ("sum", tot))
