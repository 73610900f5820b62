"""Winograd F(2x2,3x3) kernel against the direct GEMM-class kernel on the VGG stack's shapes: relative error against a float64 reference (small
cases) / against the direct kernel, and time per launch (20 launches in one HIP graph).  python tools/mb_wino.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vts import ops  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=20):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = [(1, 32, 64, 24, 40, True), (2, 64, 64, 72, 88, True), (4, 64, 64, 1024, 1024, False), (4, 64, 128, 512, 512, False), (4, 128, 128, 512, 512, False),
         (4, 128, 256, 256, 256, False), (4, 256, 256, 256, 256, False), (4, 256, 512, 128, 128, False), (4, 512, 512, 128, 128, False), (4, 512, 512, 64, 64, False),
         (256, 64, 64, 32, 32, False), (256, 128, 128, 16, 16, False), (1024, 256, 256, 8, 8, False), (1024, 128, 256, 8, 8, False),
         (1024, 512, 512, 4, 4, False), (1024, 512, 512, 2, 2, False)]
for n, ci, co, h, w, check64 in cases:
    gen = torch.Generator().manual_seed(ci * 7 + co)
    x = (torch.rand(n, ci, h, w, generator=gen) * 2 - 1).to(dev)
    wt = ((torch.rand(co, ci, 3, 3, generator=gen) * 2 - 1) / (ci * 9) ** 0.5).to(dev)
    b = (torch.rand(co, generator=gen) - 0.5).to(dev)
    p = ops.pad_affine(x, (1, 1, 1, 1), 0)
    packed = ops.w3x3_pack(wt, "conv_fwd", tag="mbw")
    z = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_wide(p, packed, b, z)
    ok = ops.conv3x3_wino_ok(n, ci, co, h, w)
    line = "N%d %d->%d %dx%d: " % (n, ci, co, h, w)
    if not ok:
        print(line + "not taken by the Winograd kernel")
        continue
    U = ops.w3x3_wino_pack(wt, "conv_fwd", tag="mbw")
    y = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_wino(p, U, b, y)
    torch.cuda.synchronize()
    rel = ((y - z).norm() / z.norm()).item()
    line += "rel vs direct %.2e" % rel
    if check64:
        ref = F.conv2d(x.cpu().double(), wt.cpu().double(), b.cpu().double(), padding=1)
        line += ", vs float64: winograd %.2e direct %.2e" % (((y.cpu().double() - ref).norm() / ref.norm()).item(), ((z.cpu().double() - ref).norm() / ref.norm()).item())
    # padded-layout epilogues
    yp = torch.full((n, co, h + 2, w + 2), 3.0, device=dev)
    ops.conv3x3_wino(p, U, b, yp, ep_mode=1)
    okp = torch.equal(yp, F.pad(torch.relu(y), (1,) * 4))
    flops = 2.0 * n * h * w * co * ci * 9
    td = timed(lambda: ops.conv3x3_wide(p, packed, b, z))
    tw = timed(lambda: ops.conv3x3_wino(p, U, b, y))
    print(line + ", relu+pad == relu(plain): %s; direct %.0f us (%.0f TF) winograd %.0f us (%.0f TF-equivalent)  x%.2f" % (okp, td, flops / td / 1e6, tw, flops / tw / 1e6, td / tw))
