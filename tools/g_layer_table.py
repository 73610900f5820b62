"""Per-layer roofline table of the generator (SURVEY.md 8d's list: the 17 conv / convT layers of CustomUnetGenerator at 1024 x 1024,
4 images): forward, backward-data and weight-gradient launches timed one by one (HIP events, 20 reps, operands with the
normalise-on-load affine + activation the network uses), against  t_roof = max(flops / 157.3 TFLOP/s, bytes / 8 TB/s)  with the
ALGORITHMIC flops (2 MAC) and bytes (4 (in + out + w)) of the layer.      python tools/g_layer_table.py [out.md]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from vts import lib as L  # noqa: E402
from vts import ops  # noqa: E402
from vts.ops import Act  # noqa: E402

dev = torch.device("cuda:0")
PEAK_TF, PEAK_GBS = 157.3, 8000.0
N = 4


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


def act(c, h, affine=True):
    x = torch.randn(N, c, h, h, device=dev)
    return Act(x, torch.rand(N * c, device=dev) + 0.5, torch.randn(N * c, device=dev) * 0.1) if affine else Act(x)


def layer(name, cin, cout, hin, transposed, dual=0, copies=1):
    """one generator layer: Conv2d(4, s2, p1) hin -> hin/2 or ConvTranspose2d(4, s2, p1) hin -> 2 hin; dual: channels of the second
    (concat-on-load) source"""
    hout = hin * 2 if transposed else hin // 2
    a0 = act(cin - dual, hin, affine=name != "down0")
    a1 = act(dual, hin, affine=name != "down0") if dual else None
    w = (torch.randn(cin, cout, 4, 4, device=dev) if transposed else torch.randn(cout, cin, 4, 4, device=dev)) * 0.05
    out = torch.empty(N, cout, hout, hout, device=dev)
    g = torch.randn_like(out)
    dx = torch.empty(N, cin - dual, hin, hin, device=dev)
    dw = torch.empty_like(w)
    act_in = 0 if name == "down0" else (L.ACT_RELU if transposed else L.ACT_LRELU)
    if transposed:
        fwd = lambda: ops.conv4x4(a0, w, 16, cout * 16, cout, out, in1=a1, stride=2, pad=1, transposed=True, act_in=act_in)   # noqa: E731
        bwd = lambda: ops.conv4x4(Act(g), w, cout * 16, 16, cin - dual, dx, stride=2, pad=1, dmask=a0, dmask_act=act_in)      # noqa: E731
        wg = lambda: ops.wgrad4x4(a0, Act(g), dw, lo1=a1, act_lo=act_in, stride=2, pad=1)                                    # noqa: E731
    else:
        fwd = lambda: ops.conv4x4(a0, w, cin * 16, 16, cout, out, in1=a1, stride=2, pad=1, act_in=act_in)                     # noqa: E731
        bwd = lambda: ops.conv4x4(Act(g), w, 16, cin * 16, cin - dual, dx, stride=2, pad=1, transposed=True, dmask=a0, dmask_act=act_in)   # noqa: E731
        wg = lambda: ops.wgrad4x4(Act(g), a0, dw, hi1=a1, act_hi=act_in, stride=2, pad=1)                                    # noqa: E731
    taps = 4 if transposed else 16     # MACs per output element and input channel
    flops = 2.0 * N * hout * hout * cout * cin * taps
    nbytes = 4.0 * (N * cin * hin * hin + N * cout * hout * hout + cin * cout * 16)
    t_roof = max(flops / (PEAK_TF * 1e12), nbytes / (PEAK_GBS * 1e9)) * 1e6
    row = [name + (" x%d" % copies if copies > 1 else ""), "%d->%d @%d->%d" % (cin, cout, hin, hout), flops / 1e9, nbytes / 1e6, flops / nbytes, t_roof]
    for fn in (fwd, bwd, wg):
        us = timeit(fn)
        row += [us, t_roof / us, L.load().vts_last_kernel().decode()]
    return row


def main():
    ngf, style = 10, 512
    ch = [ngf * min(2 ** i, 8) for i in range(8)]
    rows = [layer("down0", 9, ch[0], 1024, False, dual=8)]
    for i in range(1, 8):
        rows.append(layer("down%d" % i, ch[i - 1], ch[i], 1024 >> i, False))
    rows.append(layer("up7", ch[7] + style, ch[6], 4, True, dual=style))
    for i in range(6, 0, -1):
        rows.append(layer("up%d" % i, 2 * ch[i], ch[i - 1], 1024 >> (i + 1), True, dual=ch[i], copies=2 if i <= 3 else 1))
    rows.append(layer("up0", ch[0], 3, 512, True))
    rows.append(layer("up0_T", ch[0], 2, 512, True))
    lines = ["| layer | shape (N=4) | GFLOP | MB | flop/B | t_roof us | fwd us | fwd frac | bwd-data us | frac | wgrad us | frac |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    tot = [0.0, 0.0, 0.0, 0.0]
    for r in rows:
        mult = 2 if " x2" in r[0] else 1
        lines.append("| %s | %s | %.3f | %.1f | %.1f | %.1f | %.1f | **%.2f** | %.1f | %.2f | %.1f | %.2f |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[9], r[10], r[12], r[13]))
        tot[0] += r[5] * mult; tot[1] += r[6] * mult; tot[2] += r[9] * mult; tot[3] += r[12] * mult
    lines.append("| **sum** | | | | | %.1f | %.1f | **%.2f** | %.1f | %.2f | %.1f | %.2f |" % (tot[0], tot[1], tot[0] / tot[1], tot[2], tot[0] / tot[2], tot[3], tot[0] / tot[3]))
    kern = ["", "kernels (fwd / bwd-data / wgrad):"] + ["* %s: %s / %s / %s" % (r[0], r[8], r[11], r[14]) for r in rows]
    text = "\n".join(lines + kern)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
