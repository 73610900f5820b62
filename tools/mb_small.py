"""Micro-benchmark of the D2 patch-stack shapes (640 / 256 maps of 2^2 .. 32^2), one launch each in a HIP graph.   python tools/mb_small.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from vts import lib as L, ops  # noqa: E402
from vts.ops import Act  # noqa: E402
from mb_px import timeit  # noqa: E402

dev = torch.device("cuda:0")


def case(N, Cin, H, Cout, stride, pad, transposed=False, act=1, dmask=False, OH=None):
    x = torch.randn(N, Cin, H, H, device=dev)
    if transposed:
        OH = OH or (H - 1) * stride - 2 * pad + 4
        w = torch.randn(Cin, Cout, 4, 4, device=dev) * 0.1
        wsco, wsci = 16, Cout * 16
    else:
        OH = (H + 2 * pad - 4) // stride + 1
        w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.1
        wsco, wsci = Cin * 16, 16
    out = torch.zeros(N, Cout, OH, OH, device=dev)
    a = Act(x, torch.ones(N * Cin, device=dev), torch.zeros(N * Cin, device=dev))
    dm = Act(torch.randn(N, Cout, OH, OH, device=dev)) if dmask else None
    us = timeit(lambda: ops.conv4x4(a, w, wsco, wsci, Cout, out, stride=stride, pad=pad, transposed=transposed, act_in=0 if dmask else act, dmask=dm,
                                    dmask_act=L.ACT_LRELU if dmask else 0))
    kern = L.load().vts_last_kernel().decode()
    taps = 4 if (transposed and stride == 2) else 16
    fl = 2.0 * N * OH * OH * Cout * Cin * taps
    by = 4.0 * (x.numel() + out.numel() * (1 + int(dmask)) + w.numel())
    roof = max(fl / 157.3e6, by / 8e6)
    print("%s N%d %3dx%3d^2 -> %3dx%3d^2 s%d p%d%s : %7.1f us %6.2f TF %7.1f GB/s  roof %5.1f us frac %.2f  %s" % (
        "convT" if transposed else "conv ", N, Cin, H, Cout, OH, stride, pad, " dmask" if dmask else "", us, fl / us / 1e6, by / us / 1e3, roof, roof / us, kern))


if __name__ == "__main__":
    print({k: v for k, v in os.environ.items() if k.startswith("VTS_")})
    for N in (640, 256):
        case(N, 7, 32, 8, 2, 2, act=0)
        case(N, 8, 17, 16, 2, 2)
        case(N, 16, 9, 32, 2, 2)
        case(N, 32, 5, 64, 1, 2)
        case(N, 64, 6, 1, 1, 2)
    case(640, 64, 6, 32, 1, 2, transposed=True, dmask=True)
    case(640, 32, 5, 16, 2, 2, transposed=True, dmask=True, OH=9)
    case(640, 16, 9, 8, 2, 2, transposed=True, dmask=True, OH=17)
    case(256, 8, 17, 7, 2, 2, transposed=True, OH=32)
