"""Micro-benchmark of the thin-layer shapes of the headline step, one launch each (HIP events): run twice, with and without VTS_NO_PX=1
(resp. VTS_NO_PXT=1), to compare the lane = pixel members (vts_conv_px.hip) with the kernels they replace.   python tools/mb_px.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from vts import lib as L, ops  # noqa: E402
from vts.ops import Act  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    """20 launches captured in one HIP graph (the Python / ctypes enqueue costs ~15 us per call: eager timing floors there)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
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


ONLY = [int(t) for t in os.environ["VTS_MB_ONLY"].split(",")] if os.environ.get("VTS_MB_ONLY") else None
COUNT = [0]


def case(*a, **k):
    COUNT[0] += 1
    if ONLY is None or COUNT[0] - 1 in ONLY:
        case_(*a, **k)


def case_(N, Cin, H, Cout, pad, transposed=False, act=0, dmask=False, acc=False, affine=False):
    x = torch.randn(N, Cin, H, H, device=dev)
    if transposed:
        OH = (H - 1) * 2 - 2 * pad + 4
        if pad == 2:
            OH += 1
        w = torch.randn(Cin, Cout, 4, 4, device=dev) * 0.1
        wsco, wsci = 16, Cout * 16
    else:
        OH = (H + 2 * pad - 4) // 2 + 1
        w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.1
        wsco, wsci = Cin * 16, 16
    out = torch.zeros(N, Cout, OH, OH, device=dev)
    a = Act(x, torch.ones(N * Cin, device=dev), torch.zeros(N * Cin, device=dev)) if affine else Act(x)
    dm = Act(torch.randn(N, Cout, OH, OH, device=dev)) if dmask else None
    us = timeit(lambda: ops.conv4x4(a, w, wsco, wsci, Cout, out, stride=2, pad=pad, transposed=transposed, act_in=act, dmask=dm,
                                    dmask_act=L.ACT_RELU if dmask else 0, accumulate=acc))
    kern = L.load().vts_last_kernel().decode()
    taps = 4 if transposed else 16
    fl = 2.0 * N * OH * OH * Cout * Cin * taps
    by = 4.0 * (x.numel() + out.numel() * (1 + int(dmask) + int(acc)))
    roof = max(fl / 157.3e6, by / 8e6)
    print("%s N%d %3dx%4d^2 -> %3dx%4d^2 p%d%s%s%s : %7.1f us %6.2f TF %7.1f GB/s  frac %.2f  %s" % (
        "convT" if transposed else "conv ", N, Cin, H, Cout, OH, pad, " act" if act else "", " dmask" if dmask else "", " acc" if acc else "",
        us, fl / us / 1e6, by / us / 1e3, roof / us, kern))


if __name__ == "__main__":
    print("VTS_NO_PX=%s VTS_NO_PXT=%s VTS_ABLATE=%s" % (os.environ.get("VTS_NO_PX"), os.environ.get("VTS_NO_PXT"), os.environ.get("VTS_ABLATE")))
    case(4, 9, 1024, 10, 1)                       # G down0
    case(4, 10, 512, 20, 1, act=1)                # G down1
    case(8, 4, 1024, 8, 2)                        # D1 layer 0, both passes
    case(4, 7, 1024, 8, 2)                        # D2 full-resolution layer 0
    case(4, 4, 1024, 8, 2)
    case(8, 8, 513, 16, 2, act=1)                 # D layer 1
    case(4, 8, 513, 16, 2, act=1)
    case(4, 3, 1024, 10, 1, dmask=True)           # backward-data of up0
    case(4, 2, 1024, 10, 1, dmask=True)           # ... of up0_T
    case(4, 10, 512, 20, 1, dmask=True)           # ... of up1
    case(4, 10, 512, 20, 1, dmask=True, acc=True)
    # transposed stride 2
    case(4, 10, 512, 3, 1, transposed=True, act=2)       # up0
    case(4, 10, 512, 2, 1, transposed=True, act=2)       # up0_T
    case(4, 40, 256, 10, 1, transposed=True, act=2, affine=True)      # up1
    case(4, 80, 128, 20, 1, transposed=True, act=2, affine=True)      # up2
    case(4, 10, 512, 9, 1, transposed=True)                # backward-data of down0
    case(4, 20, 256, 10, 1, transposed=True, dmask=True)   # ... of down1
    case(4, 40, 128, 20, 1, transposed=True, dmask=True)   # ... of down2
    case(8, 16, 257, 8, 2, transposed=True, dmask=True)    # ... of D layer 1
    case(4, 16, 257, 8, 2, transposed=True, dmask=True)
    case(8, 8, 513, 4, 2, transposed=True)                 # ... of D layer 0 (image gradient)
    case(8, 32, 129, 16, 2, transposed=True, dmask=True)   # ... of D layer 2
