"""Micro-benchmark of single vts_conv4x4 / vts_wgrad4x4 launches (HIP events, 20 reps)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from vts import ops  # noqa: E402
from vts.ops import Act  # noqa: E402

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


def conv_case(N, Cin, H, W, Cout, stride, pad, transposed):
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
    sc = torch.ones(N * Cin, device=dev)
    sh = torch.zeros(N * Cin, device=dev)
    a = Act(x, sc, sh)
    us = timeit(lambda: ops.conv4x4(a, w, wsco, wsci, Cout, out, stride=stride, pad=pad, transposed=transposed, act_in=1))
    taps = 4 if (transposed and stride == 2) else 16
    fl = 2.0 * N * OH * OH * Cout * Cin * taps
    by = 4.0 * (x.numel() + out.numel())
    print("%s N%d %dx%dx%d -> %dx%dx%d s%d p%d : %8.1f us  %6.2f TF  %7.1f GB/s" % (
        "convT" if transposed else "conv ", N, Cin, H, W, Cout, OH, OH, stride, pad, us, fl / us / 1e6, by / us / 1e3))


def wgrad_case(N, CL, LH, CH, stride, pad):
    HH = (LH - 1) * stride + 4 - 2 * pad
    lo = torch.randn(N, CL, LH, LH, device=dev)
    hi = torch.randn(N, CH, HH, HH, device=dev)
    dw = torch.empty(CL, CH, 4, 4, device=dev)
    us = timeit(lambda: ops.wgrad4x4(Act(lo), Act(hi), dw, stride=stride, pad=pad))
    fl = 2.0 * N * LH * LH * CL * CH * 16
    by = 4.0 * (lo.numel() + hi.numel())
    print("wgrad N%d lo %dx%d hi %dx%d s%d : %8.1f us  %6.2f TF  %7.1f GB/s" % (N, CL, LH, CH, HH, stride, us, fl / us / 1e6, by / us / 1e3))


def wide_case(N, C, H, W):
    p = torch.randn(N, C, H + 2, W + 2, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.01
    out = torch.empty(N, C, H, W, device=dev)
    wt = ops.w3x3_pack(w, "conv_fwd")
    us = timeit(lambda: ops.conv3x3_wide(p, wt, None, out), reps=5)
    fl = 2.0 * N * H * W * C * C * 9
    print("wide3x3 N%d %dx%dx%d : %8.1f us  %6.2f TF (%.1f%% of 157.3)" % (N, C, H, W, us, fl / us / 1e6, fl / us / 1e6 / 1.573))
    dw = torch.empty_like(w)
    us = timeit(lambda: ops.wgrad3x3_wide(out, p, dw), reps=5)
    print("wgrad3x3_wide N%d %dx%dx%d : %8.1f us  %6.2f TF (%.1f%% of 157.3)" % (N, C, H, W, us, fl / us / 1e6, fl / us / 1e6 / 1.573))


if __name__ == "__main__":
    print("ABLATE=%s SMALL=%s" % (os.environ.get("VTS_ABLATE"), os.environ.get("VTS_SMALL_WGS")))
    if os.environ.get("VTS_MB") == "pmc":
        conv_case(4, 9, 1024, 1024, 10, 2, 1, False)
        wgrad_case(4, 10, 512, 9, 2, 1)
        sys.exit(0)
    if os.environ.get("VTS_MB", "").startswith("one:"):
        a = [int(v) for v in os.environ["VTS_MB"][4:].split(",")]
        conv_case(a[0], a[1], a[2], a[3], a[4], a[5], a[6], bool(a[7]))
        sys.exit(0)
    if os.environ.get("VTS_MB", "").startswith("wideone:"):     # VTS_MB=wideone:N,C,H,W  (with VTS_WIDE_KS=k to force the k-split)
        wide_case(*[int(v) for v in os.environ["VTS_MB"].split(":")[1].split(",")])
        sys.exit(0)
    if os.environ.get("VTS_MB") == "wide":
        wide_case(1, 1024, 64, 128)
        wide_case(1, 1024, 16, 32)
        wide_case(4, 512, 64, 64)
        wide_case(4, 256, 128, 128)
        sys.exit(0)
    if os.environ.get("VTS_MB") == "top":
        conv_case(4, 4, 1024, 1024, 8, 2, 2, False)
        conv_case(4, 8, 513, 513, 16, 2, 2, False)
        conv_case(4, 16, 257, 257, 8, 2, 2, True)
        conv_case(4, 32, 129, 129, 16, 2, 2, True)
        conv_case(4, 40, 256, 256, 10, 2, 1, True)
        conv_case(4, 20, 512, 512, 3, 2, 1, True)
        conv_case(4, 32, 129, 129, 64, 1, 2, False)
        conv_case(4, 64, 130, 130, 32, 1, 2, True)
        conv_case(4, 64, 130, 130, 1, 1, 2, False)
        conv_case(4, 160, 64, 64, 40, 2, 1, True)
        sys.exit(0)
    if os.environ.get("VTS_MB") == "small":
        conv_case(256, 64, 6, 6, 1, 1, 2, False)
        conv_case(256, 32, 5, 5, 64, 1, 2, False)
        conv_case(256, 7, 32, 32, 8, 2, 2, False)
        conv_case(256, 64, 6, 6, 32, 1, 2, True)
        sys.exit(0)
    conv_case(4, 9, 1024, 1024, 10, 2, 1, False)
    conv_case(4, 10, 512, 512, 20, 2, 1, False)
    conv_case(4, 40, 128, 128, 80, 2, 1, False)
    conv_case(4, 32, 129, 129, 64, 1, 2, False)
    conv_case(4, 10, 512, 512, 3, 2, 1, True)
    conv_case(4, 40, 256, 256, 10, 2, 1, True)
    conv_case(4, 160, 64, 64, 40, 2, 1, True)
    if not os.environ.get("VTS_ABLATE"):
        wgrad_case(4, 10, 512, 9, 2, 1)
        wgrad_case(4, 40, 256, 10, 2, 1)
        wgrad_case(4, 64, 130, 32, 1, 2)
