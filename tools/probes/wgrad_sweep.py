"""Plan sweep of the N-split weight-gradient kernel (VTS_WGRAD_TUNE=1: "cl_groups,ch_groups,copies" per call) on the shapes of the
headline step.  Per shape: the default plan, then every (cl_groups, ch_groups) with <= 5 x 5 accumulator tiles per wave at several
workgroup counts; cost = kernel time + copies * nel * 4 B / 2.2 TB/s (what the batched reduction adds)."""
import ctypes as C
import os
import sys

os.environ["VTS_WGRAD_TUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from vts import lib as L  # noqa: E402
from vts import ops  # noqa: E402
from vts.ops import Act  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # name, N, CL, LH, CH, stride, pad, affine
    ("down0", 4, 10, 512, 9, 2, 1, 0), ("down1", 4, 20, 256, 10, 2, 1, 1), ("down2", 4, 40, 128, 20, 2, 1, 1), ("down3", 4, 80, 64, 40, 2, 1, 1),
    ("down4", 4, 80, 32, 80, 2, 1, 1), ("down5", 4, 80, 16, 80, 2, 1, 1), ("up5", 4, 160, 16, 80, 2, 1, 0), ("up4", 4, 160, 32, 80, 2, 1, 0),
    ("up3", 4, 160, 64, 40, 2, 1, 0), ("up2", 4, 80, 128, 20, 2, 1, 0), ("up1", 4, 40, 256, 10, 2, 1, 0),
    ("d1s0l4", 8, 1, 131, 64, 1, 2, 1), ("d1s0l3", 8, 64, 130, 32, 1, 2, 1), ("d1s0l2", 8, 32, 129, 16, 2, 2, 1), ("d1s0l1", 8, 16, 257, 8, 2, 2, 1),
    ("d1s1l3", 8, 64, 66, 32, 1, 2, 1), ("d1s1l2", 8, 32, 65, 16, 2, 2, 1), ("d1s1l1", 8, 16, 129, 8, 2, 2, 1),
    ("up0", 4, 10, 512, 3, 2, 1, 0), ("up0T", 4, 10, 512, 2, 2, 1, 0), ("d1s0l0", 8, 8, 513, 4, 2, 2, 0), ("d1s1l0", 8, 8, 257, 4, 2, 2, 0),
    ("d1s2l3", 8, 64, 34, 32, 1, 2, 1), ("d1s2l2", 8, 32, 33, 16, 2, 2, 1), ("d1s2l1", 8, 16, 65, 8, 2, 2, 1),
]
cdiv = lambda a, b: (a + b - 1) // b  # noqa: E731


def one(d, ws, reps=20):
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        lib.vts_wgrad4x4(C.byref(d), ws.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.vts_wgrad4x4(C.byref(d), ws.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def sweep(name, N, CL, LH, CH, stride, pad, affine):
    HH = (LH - 1) * stride + 4 - 2 * pad
    lo = torch.randn(N, CL, LH, LH, device=dev)
    hi = torch.randn(N, CH, HH, HH, device=dev)
    dw = torch.empty(CL, CH, 4, 4, device=dev)
    lib = L.load()
    d = L.WgradDesc()
    hi_op = Act(hi, torch.rand(N * CH, device=dev) + 0.5, torch.randn(N * CH, device=dev) * 0.3) if affine else hi
    d.lo0, d.lo1, d.hi0, d.hi1 = ops._op(lo), ops._op(None), ops._op(hi_op), ops._op(None)
    d.act_hi = L.ACT_LRELU if affine else 0
    d.N, d.LH, d.LW, d.HH, d.HW = N, LH, LH, HH, HH
    d.stride, d.pad, d.pad_dx = stride, pad, 0
    d.dw = dw.data_ptr()
    d.defer = 1
    nel = CL * CH * 16
    ws = torch.empty(max(1 << 24, 2048 * nel // 4), device=dev)     # >= 64 MB
    os.environ.pop("VTS_WGRAD_PLAN", None)
    n0 = lib.vts_wgrad4x4_ws_floats(C.byref(d))
    t0 = one(d, ws)
    red = lambda copies: copies * nel * 4 / 2.2e6   # noqa: E731  us
    fl = 2.0 * N * LH * LH * CL * CH * 16
    by = 4.0 * (lo.numel() + hi.numel())
    roof = max(fl / 157.3e6, by / 8e6)
    print("%s N%d lo %dx%d hi %dx%d s%d: roof %.1f us | default %s copies %d: %.1f us + reduce %.1f = %.1f" % (
        name, N, CL, LH, CH, HH, stride, roof, lib.vts_last_kernel().decode(), n0 // nel, t0, red(n0 // nel), t0 + red(n0 // nel)), flush=True)
    if os.environ.get("SWEEP_DEFAULT_ONLY"):
        return t0 + red(n0 // nel)
    ty = 2 if stride == 2 else 4
    ntiles = N * cdiv(LH, ty) * cdiv(LH, 28)
    res = []
    for clg in range(1, cdiv(CL, 16) + 1):
        clt = cdiv(CL, 16 * clg)
        if clt > 5 or (clg > 1 and cdiv(CL, 16 * (clg - 1)) == clt):
            continue
        for chg in range(1, cdiv(CH, 4) + 1):
            cht = cdiv(CH, 4 * chg)
            if cht > 5 or (chg > 1 and cdiv(CH, 4 * (chg - 1)) == cht):
                continue
            groups = clg * chg
            pws = set()
            for target in (128, 192, 256, 384, 512, 768, 1024, 1536):
                pw = max(1, min(target // groups, ntiles))
                pws.add(pw)
                for q in range(pw, max(pw // 2, 0), -1):      # the nearest count below that divides the tiles evenly
                    if ntiles % q == 0:
                        pws.add(q)
                        break
            for pw in sorted(pws):
                if pw * nel * 4 > ws.numel() * 4 or pw * groups > 2048:
                    continue
                os.environ["VTS_WGRAD_PLAN"] = "%d,%d,%d" % (clg, chg, pw)
                t = one(d, ws)
                res.append((t + red(pw), t, clg, chg, clt, cht, pw, pw * groups))
                if CSV is not None:
                    CSV.write("%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.2f\n" % (name, N, CL, LH, CH, stride, affine, clg, chg, clt, cht, pw, ntiles, t))
    res.sort()
    for tot, t, clg, chg, clt, cht, pw, wgs in res[:6]:
        print("    clg %d chg %d (clt %d cht %d) copies %4d wgs %4d tiles/wg %.2f: %6.1f us + reduce %5.1f = %6.1f" % (
            clg, chg, clt, cht, pw, wgs, ntiles / pw, t, red(pw), tot), flush=True)
    bt = min(res, key=lambda r: r[1])
    print("    fastest kernel alone: clg %d chg %d copies %d wgs %d: %.1f us" % (bt[2], bt[3], bt[6], bt[7], bt[1]), flush=True)


CSV = open(os.environ["SWEEP_CSV"], "w") if os.environ.get("SWEEP_CSV") else None

if __name__ == "__main__":
    only = os.environ.get("SHAPES")
    tot = 0.0
    for s in SHAPES:
        if only is None or s[0] in only.split(","):
            tot += sweep(*s) or 0.0
    print("sum of default plans: %.1f us" % tot)
