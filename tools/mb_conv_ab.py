"""A/B micro-benchmark of vts_conv4x4 on the generator / discriminator shapes of the headline step (HIP events, 30 back-to-back launches each).
    VTS_LIB_PATH=<library> python tools/mb_conv_ab.py [tag]        one line per shape: us, achieved TFLOP/s of the algorithmic work
tools/ab_libs.sh runs it for two libraries on the same box and prints the ratio."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from vts import ops  # noqa: E402
from vts.ops import Act  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=30):
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


def case(name, N, Cin, H, W, Cout, stride, pad, transposed, affine):
    x = torch.randn(N, Cin, H, W, device=dev)
    if transposed:
        OH = (H - 1) * stride - 2 * pad + 4
        w = torch.randn(Cin, Cout, 4, 4, device=dev) * 0.1
        wsco, wsci = 16, Cout * 16
    else:
        OH = (H + 2 * pad - 4) // stride + 1
        w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.1
        wsco, wsci = Cin * 16, 16
    out = torch.empty(N, Cout, OH, OH, device=dev)
    if affine:
        a = Act(x, torch.rand(N * Cin, device=dev) + 0.5, torch.randn(N * Cin, device=dev) * 0.1)
        us = timeit(lambda: ops.conv4x4(a, w, wsco, wsci, Cout, out, stride=stride, pad=pad, transposed=transposed, act_in=1))
    else:
        a = Act(x)
        us = timeit(lambda: ops.conv4x4(a, w, wsco, wsci, Cout, out, stride=stride, pad=pad, transposed=transposed, act_in=0))
    taps = 4 if (transposed and stride == 2) else 16
    fl = 2.0 * N * OH * OH * Cout * Cin * taps
    print("%-34s %8.1f us %7.2f TF" % (name, us, fl / us / 1e6), flush=True)


CASES = [
    # generator forward (normalise-on-load) ...
    ("G down1 fwd 10->20 @512", 4, 10, 512, 512, 20, 2, 1, False, True),
    ("G down2 fwd 20->40 @256", 4, 20, 256, 256, 40, 2, 1, False, True),
    ("G down3 fwd 40->80 @128", 4, 40, 128, 128, 80, 2, 1, False, True),
    ("G down4 fwd 80->80 @64", 4, 80, 64, 64, 80, 2, 1, False, True),
    ("G up4 fwd 160->80 @32", 4, 160, 32, 32, 80, 2, 1, True, True),
    ("G up3 fwd 160->40 @64", 4, 160, 64, 64, 40, 2, 1, True, True),
    ("G up2 fwd 80->20 @128", 4, 80, 128, 128, 20, 2, 1, True, True),
    ("G up1 fwd 40->10 @256", 4, 40, 256, 256, 10, 2, 1, True, True),
    # ... and the input adjoints (raw gradients: no affine, no activation)
    ("G down1 bwd 20->10 @256", 4, 20, 256, 256, 10, 2, 1, True, False),
    ("G down2 bwd 40->20 @128", 4, 40, 128, 128, 20, 2, 1, True, False),
    ("G down3 bwd 80->40 @64", 4, 80, 64, 64, 40, 2, 1, True, False),
    ("G up3 bwd 40->160 @128", 4, 40, 128, 128, 160, 2, 1, False, False),
    ("G up2 bwd 20->80 @256", 4, 20, 256, 256, 80, 2, 1, False, False),
    ("G up1 bwd 10->40 @512", 4, 10, 512, 512, 40, 2, 1, False, False),
    # discriminator D1 (full resolution scale), forward and adjoint
    ("D 8->16 @513 s2", 4, 8, 513, 513, 16, 2, 2, False, True),
    ("D 16->32 @257 s2", 4, 16, 257, 257, 32, 2, 2, False, True),
    ("D 32->64 @129 s1", 4, 32, 129, 129, 64, 1, 2, False, True),
    ("D 32->64 @129 s1 N8", 8, 32, 129, 129, 64, 1, 2, False, True),
    ("D bwd 64->32 @130 s1", 4, 64, 130, 130, 32, 1, 2, True, False),
    ("D bwd 32->16 @129 s2", 4, 32, 129, 129, 16, 2, 2, True, False),
]

if __name__ == "__main__":
    print("# lib %s %s" % (os.environ.get("VTS_LIB_PATH", "(default)"), sys.argv[1] if len(sys.argv) > 1 else ""))
    sel = os.environ.get("VTS_MB_SEL")
    for c in CASES:
        if sel and sel not in c[0]:
            continue
        case(*c)
