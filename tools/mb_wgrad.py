"""Micro-benchmark + parity check of vts_wgrad4x4 on the weight-gradient shapes of the headline step (HIP events, 20 reps;
reference: torch.nn.grad.conv2d_weight on the CPU).  VTS_WGRAD_NS=0 selects the K-split kernel only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vts import lib as L  # noqa: E402
from vts import ops  # noqa: E402
from vts.ops import Act  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(8)


def timeit(fn, reps=20):
    """20 calls (weight-gradient kernel + its reduction) captured in one HIP graph: eager timing floors at the ~15 us Python enqueue"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("VTS_MB_EAGER"):       # (under rocprofv3: plain launches, the trace has the kernel times)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 0.0
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(N, CL, LH, CH, stride, pad, affine=False, split_lo=0, check=True):
    HH = (LH - 1) * stride + 4 - 2 * pad
    g = torch.Generator().manual_seed(N * 1000 + CL * 10 + CH)
    lo = torch.randn(N, CL, LH, LH, generator=g)
    hi = torch.randn(N, CH, HH, HH, generator=g)
    lo_d, hi_d = lo.to(dev), hi.to(dev)
    dw = torch.empty(CL, CH, 4, 4, device=dev)
    if affine:
        sc, sh = torch.rand(N * CH, generator=g) + 0.5, torch.randn(N * CH, generator=g) * 0.3
        hi_act = Act(hi_d, sc.to(dev), sh.to(dev))
        hi_ref = F.leaky_relu(hi * sc.view(N, CH, 1, 1) + sh.view(N, CH, 1, 1), 0.2)
        act_hi = L.ACT_LRELU
    else:
        hi_act, hi_ref, act_hi = Act(hi_d), hi, 0
    if split_lo:
        a, b = lo_d[:, :split_lo].contiguous(), lo_d[:, split_lo:].contiguous()
        fn = lambda: ops.wgrad4x4(Act(a), hi_act, dw, lo1=Act(b), act_hi=act_hi, stride=stride, pad=pad)   # noqa: E731
    else:
        fn = lambda: ops.wgrad4x4(Act(lo_d), hi_act, dw, act_hi=act_hi, stride=stride, pad=pad)   # noqa: E731
    us = timeit(fn)
    kern = L.load().vts_last_kernel().decode()
    err = float("nan")
    if check:
        ref = torch.nn.grad.conv2d_weight(hi_ref, (CL, CH, 4, 4), lo, stride=stride, padding=pad)
        err = float((dw.cpu() - ref).norm() / ref.norm())
    fl = 2.0 * N * LH * LH * CL * CH * 16
    by = 4.0 * (lo.numel() + hi.numel())
    us = max(us, 1e-9)
    print("wgrad N%d lo %dx%d hi %dx%d s%d p%d%s%s : %8.1f us  %6.2f TF  %7.1f GB/s  rel-L2 %.2e  %s" % (
        N, CL, LH, CH, HH, stride, pad, " affine" if affine else "", " split" if split_lo else "", us, fl / us / 1e6, by / us / 1e3, err, kern))
    return err


if __name__ == "__main__":
    errs = []
    quick = os.environ.get("VTS_MB") == "quick"
    errs.append(case(4, 64, 130, 32, 1, 2, affine=True))         # D1 scale 0, layer 3
    errs.append(case(4, 80, 128, 20, 2, 1, split_lo=40))         # G up2 (lo = cat(up3 out, skip))
    errs.append(case(4, 40, 256, 10, 2, 1, split_lo=20))         # G up1
    errs.append(case(4, 160, 64, 40, 2, 1, split_lo=80))         # G up3
    errs.append(case(4, 64, 66, 32, 1, 2, affine=True))          # D1 scale 1, layer 3
    errs.append(case(4, 16, 257, 8, 2, 2, affine=True))          # D1 scale 0, layer 1
    errs.append(case(4, 32, 129, 16, 2, 2, affine=True))         # D1 scale 0, layer 2
    errs.append(case(4, 10, 512, 9, 2, 1))                       # G down0
    errs.append(case(4, 20, 256, 10, 2, 1, affine=True))         # G down1
    errs.append(case(4, 40, 128, 20, 2, 1, affine=True))         # G down2
    errs.append(case(4, 80, 64, 40, 2, 1, affine=True))          # G down3
    errs.append(case(4, 80, 32, 80, 2, 1, affine=True))          # G down4
    if not quick:
        errs.append(case(4, 160, 32, 80, 2, 1, split_lo=80))     # G up4
        errs.append(case(4, 160, 16, 80, 2, 1, split_lo=80))     # G up5
        errs.append(case(4, 592, 4, 80, 2, 1, split_lo=80))      # G up7 with the style tile
        errs.append(case(4, 8, 513, 4, 2, 2))                    # D1 scale 0, layer 0 (K-split kernel: CH < 5)
        errs.append(case(4, 10, 512, 3, 2, 1))                   # G up0
        errs.append(case(4, 10, 512, 2, 2, 1))                   # G up0_T
        errs.append(case(8, 8, 513, 4, 2, 2, check=False))       # D1 scale 0 in the D update (N = 8): layers 0 .. 3
        errs[-1] = 0.0
        errs.append(case(8, 16, 257, 8, 2, 2, affine=True, check=False))
        errs[-1] = 0.0
        errs.append(case(8, 32, 129, 16, 2, 2, affine=True, check=False))
        errs[-1] = 0.0
        errs.append(case(8, 64, 130, 32, 1, 2, affine=True, check=False))
        errs[-1] = 0.0
        errs.append(case(4, 1, 131, 64, 1, 2, affine=True))      # D1 scale 0, layer 4
        errs.append(case(256, 16, 9, 8, 2, 2, affine=True))      # D2 patches, layer 1
        errs.append(case(256, 8, 17, 7, 2, 2))                   # D2 patches, layer 0
        errs.append(case(256, 64, 6, 32, 1, 2, affine=True))     # D2 patches, layer 3 (narrow-map kernel)
        errs.append(case(3, 33, 21, 13, 2, 1, affine=True))      # ragged everything
        errs.append(case(2, 17, 19, 21, 1, 2, affine=True, split_lo=5))
    bad = [e for e in errs if not e < 2e-5]
    print("MAX rel-L2 %.2e  %s" % (max(errs), "FAIL" if bad else "OK"))
