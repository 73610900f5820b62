"""Time composition of the weight-gradient kernel ALONE (deferred: no reduction launch) on the generator / D1 shapes of the headline
step.  Run once per environment setting (the kernel's knobs are read once per process): VTS_ABLATE (1 no global loads, 2 no MFMA
phase, 4 no LDS staging), VTS_WGRAD_NS_WGS, VTS_WGRAD_CAP_MB.  Prints one line per shape: us, workgroups, partial copies."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from vts import lib as L  # noqa: E402
from vts import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # name, N, CL, LH, CH, stride, pad
    ("down0", 4, 10, 512, 9, 2, 1), ("down1", 4, 20, 256, 10, 2, 1), ("down2", 4, 40, 128, 20, 2, 1), ("down3", 4, 80, 64, 40, 2, 1),
    ("down4", 4, 80, 32, 80, 2, 1), ("down5", 4, 80, 16, 80, 2, 1), ("up5", 4, 160, 16, 80, 2, 1), ("up4", 4, 160, 32, 80, 2, 1),
    ("up3", 4, 160, 64, 40, 2, 1), ("up2", 4, 80, 128, 20, 2, 1), ("up1", 4, 40, 256, 10, 2, 1), ("up0", 4, 20, 512, 3, 2, 1),
    ("d1s0l3", 4, 64, 130, 32, 1, 2), ("d1s0l2", 4, 32, 129, 16, 2, 2), ("d1s0l1", 4, 16, 257, 8, 2, 2), ("d1s0l1x8", 8, 16, 257, 8, 2, 2),
]


def run(name, N, CL, LH, CH, stride, pad, reps=30):
    HH = (LH - 1) * stride + 4 - 2 * pad
    lo = torch.randn(N, CL, LH, LH, device=dev)
    hi = torch.randn(N, CH, HH, HH, device=dev)
    dw = torch.empty(CL, CH, 4, 4, device=dev)
    lib = L.load()
    d = L.WgradDesc()
    d.lo0, d.lo1, d.hi0, d.hi1 = ops._op(lo), ops._op(None), ops._op(hi), ops._op(None)
    d.N, d.LH, d.LW, d.HH, d.HW = N, LH, LH, HH, HH
    d.stride, d.pad, d.pad_dx = stride, pad, 0
    d.dw = dw.data_ptr()
    d.defer = 1
    n = lib.vts_wgrad4x4_ws_floats(C.byref(d))
    ws = torch.empty(n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.vts_wgrad4x4(C.byref(d), ws.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.vts_wgrad4x4(C.byref(d), ws.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    fl = 2.0 * N * LH * LH * CL * CH * 16
    by = 4.0 * (lo.numel() + hi.numel())
    roof = max(fl / 157.3e6, by / 8e6)
    print("%-9s %7.1f us  roof %5.1f us  frac %.2f  copies %4d (%5.1f MB)  %s" % (
        name, us, roof, roof / us, n // (CL * CH * 16), n * 4 / 1e6, lib.vts_last_kernel().decode()), flush=True)


if __name__ == "__main__":
    only = os.environ.get("SHAPES")
    print("# env", {k: v for k, v in os.environ.items() if k.startswith("VTS_")})
    for s in SHAPES:
        if only is None or s[0] in only.split(","):
            run(*s)
