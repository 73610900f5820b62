"""Micro-benchmark of vts_norm_stats / vts_norm_bwd / vts_channel_sum (HIP events): achieved GB/s vs algorithmic bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from vts import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


G8, G640 = [0, 4], [0, 256, 384]   # pass groups of the batched discriminator launches (fake | real; fake | more fake | real)
for shape, mode, groups in [((4, 10, 512, 512), 0, None), ((4, 20, 256, 256), 0, None), ((4, 40, 128, 128), 0, None), ((4, 80, 64, 64), 0, None),
                            ((4, 80, 32, 32), 0, None), ((4, 80, 16, 16), 0, None), ((4, 80, 8, 8), 0, None),
                            ((8, 16, 257, 257), 1, G8), ((8, 32, 129, 129), 1, G8), ((8, 64, 130, 130), 1, G8), ((8, 16, 129, 129), 1, G8),
                            ((8, 32, 65, 65), 1, G8), ((8, 64, 66, 66), 1, G8), ((8, 16, 65, 65), 1, G8), ((8, 32, 33, 33), 1, G8),
                            ((8, 64, 34, 34), 1, G8), ((4, 16, 257, 257), 1, None),
                            ((640, 16, 9, 9), 1, G640), ((640, 32, 5, 5), 1, G640), ((640, 64, 6, 6), 1, G640), ((640, 16, 5, 5), 1, G640),
                            ((640, 32, 3, 3), 1, G640), ((640, 64, 4, 4), 1, G640), ((640, 16, 3, 3), 1, G640), ((640, 32, 2, 2), 1, G640),
                            ((640, 64, 3, 3), 1, G640)]:
    x = torch.randn(shape, device=dev)
    n, c, h, w = shape
    gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    kw = dict(gamma=gamma, beta=beta, groups=groups) if mode else {}
    us = timeit(lambda: ops.norm_stats(x, mode, **kw))
    a = ops.norm_stats(x, mode, **kw)
    dy = torch.randn(shape, device=dev)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    kb = dict(gamma=gamma, dgamma=dg, dbeta=db, groups=groups) if mode else {}
    ub = timeit(lambda: ops.norm_bwd(dy, a, mode, **kb))
    out = torch.zeros(c, device=dev)
    uc = timeit(lambda: ops.channel_sum(x, out))
    by = x.numel() * 4.0
    print("%-22s mode %d : stats %7.1f us %7.1f GB/s | bwd %7.1f us %7.1f GB/s (5 passes) | chsum %7.1f us %7.1f GB/s" % (
        shape, mode, us, by / us / 1e3, ub, 5 * by / ub / 1e3, uc, by / uc / 1e3))
