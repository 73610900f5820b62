"""Micro-benchmark of the conv4x4_kernel shapes that carry the headline step (one launch each, 20 launches in one HIP graph).
   python tools/mb_conv_big.py      (env knobs of vts_conv.hip apply: VTS_TARGET_WGS, VTS_SMALL_WGS, ...)"""
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


def case(N, Cin, H, Cout, stride, pad, transposed=False, act=0, dmask=False, acc=False, affine=True, W=None):
    W = W or H
    x = torch.randn(N, Cin, H, W, device=dev)
    if transposed:
        OH, OW = (H - 1) * stride - 2 * pad + 4, (W - 1) * stride - 2 * pad + 4
        if stride == 2 and pad == 2:
            OH, OW = OH + 1, OW + 1
        w = torch.randn(Cin, Cout, 4, 4, device=dev) * 0.1
        wsco, wsci = 16, Cout * 16
    else:
        OH, OW = (H + 2 * pad - 4) // stride + 1, (W + 2 * pad - 4) // stride + 1
        w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.1
        wsco, wsci = Cin * 16, 16
    out = torch.zeros(N, Cout, OH, OW, device=dev)
    a = Act(x, torch.ones(N * Cin, device=dev), torch.zeros(N * Cin, device=dev)) if affine else Act(x)
    dm = Act(torch.randn(N, Cout, OH, OW, device=dev)) if dmask else None
    us = timeit(lambda: ops.conv4x4(a, w, wsco, wsci, Cout, out, stride=stride, pad=pad, transposed=transposed, act_in=act, dmask=dm,
                                    dmask_act=L.ACT_RELU if dmask else 0, accumulate=acc))
    kern = L.load().vts_last_kernel().decode()
    taps = 4 if (transposed and stride == 2) else 16
    fl = 2.0 * N * OH * OW * Cout * Cin * taps
    by = 4.0 * (x.numel() + out.numel() * (1 + int(dmask) + int(acc)))
    roof = max(fl / 157.3e6, by / 8e6)
    print("%s N%d %3dx%4d^2 -> %3dx%4d^2 s%d p%d%s%s%s : %7.1f us %6.2f TF %7.1f GB/s  frac %.2f  %s" % (
        "convT" if transposed else "conv ", N, Cin, H, Cout, OH, stride, pad, " act" if act else "", " dmask" if dmask else "", " acc" if acc else "",
        us, fl / us / 1e6, by / us / 1e3, roof / us, kern))


if __name__ == "__main__":
    print({k: v for k, v in os.environ.items() if k.startswith("VTS_")})
    if os.environ.get("VTS_MB_INNER"):
        case(4, 80, 64, 80, 2, 1, act=1)                        # down4
        case(4, 80, 32, 80, 2, 1, act=1)                        # down5
        case(4, 80, 16, 80, 2, 1, act=1)                        # down6
        case(4, 80, 8, 80, 2, 1, act=1)                         # down7
        case(4, 592, 4, 80, 2, 1, transposed=True, act=2)       # up7
        case(4, 160, 8, 80, 2, 1, transposed=True, act=2)       # up6
        case(4, 160, 16, 80, 2, 1, transposed=True, act=2)      # up5
        case(4, 160, 32, 80, 2, 1, transposed=True, act=2)      # up4
        case(4, 80, 8, 80, 2, 1, dmask=True, affine=False)      # backward-data of up6's input half
        case(4, 80, 16, 80, 2, 1, dmask=True, affine=False)
        case(4, 80, 32, 80, 2, 1, dmask=True, affine=False)
        case(4, 80, 4, 80, 2, 1, transposed=True, dmask=True, affine=False)    # backward-data of down7
        case(4, 80, 8, 80, 2, 1, transposed=True, dmask=True, affine=False)
        case(4, 80, 16, 80, 2, 1, transposed=True, dmask=True, affine=False)
        sys.exit(0)
    case(4, 160, 64, 40, 2, 1, transposed=True, act=2)      # up3
    case(4, 80, 128, 20, 2, 1, transposed=True, act=2)      # up2
    case(4, 40, 256, 10, 2, 1, transposed=True, act=2)      # up1
    case(4, 160, 32, 80, 2, 1, transposed=True, act=2)      # up4
    case(4, 10, 512, 20, 2, 1, act=1)                       # down1
    case(4, 20, 256, 40, 2, 1, act=1)                       # down2
    case(4, 40, 128, 80, 2, 1, act=1)                       # down3
    case(4, 80, 64, 80, 2, 1, act=1)                        # down4
    case(4, 10, 512, 20, 2, 1, dmask=True, affine=False)    # backward-data of up1
    case(4, 20, 256, 40, 2, 1, dmask=True, affine=False)    # ... of up2
    case(4, 40, 128, 80, 2, 1, dmask=True, affine=False)    # ... of up3
    case(4, 20, 256, 10, 2, 1, transposed=True, dmask=True, affine=False)   # ... of down1
    case(4, 40, 128, 20, 2, 1, transposed=True, dmask=True, affine=False)   # ... of down2
    case(4, 80, 64, 40, 2, 1, transposed=True, dmask=True, affine=False)    # ... of down3
    case(8, 32, 129, 64, 1, 2, act=1)                       # D layer 3 (stride 1), both passes
    case(4, 32, 129, 64, 1, 2, act=1)
    case(8, 64, 130, 32, 1, 2, transposed=True, dmask=True, affine=False)   # its backward-data
    case(8, 16, 257, 32, 2, 2, act=1)                       # D layer 2
    case(8, 32, 129, 16, 2, 2, transposed=True, dmask=True, affine=False)
