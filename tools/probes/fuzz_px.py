"""Random-shape check of the lane = pixel members (vts_conv_px.hip) against PyTorch on the GPU box: stride-2 convolutions / transposed
convolutions with few output channels on maps >= 128 x 128, dual sources, per-channel affine + activation on load, bias, derivative
mask, accumulation, tanh (transposed), odd / even sizes and paddings.   python tools/probes/fuzz_px.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from vts import lib as L, ops  # noqa: E402
from vts.ops import Act  # noqa: E402

dev = torch.device("cuda:0")


def apply(x, sc, sh, act):
    n, c = x.shape[:2]
    v = x * sc.view(n, c, 1, 1) + sh.view(n, c, 1, 1)
    return F.leaky_relu(v, 0.2) if act == 1 else (F.relu(v) if act == 2 else v)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst, used = 0.0, {}
    for it in range(cases):
        transposed = rng.random() < 0.5
        N = rng.choice([1, 2, 3])
        C0, C1 = rng.choice([1, 2, 3, 5, 8, 10]), rng.choice([0, 0, 1, 4])
        Cout = rng.choice([1, 2, 3, 4, 7, 8, 9, 12]) if not transposed else rng.choice([1, 2, 3, 4, 5, 8])
        pad = rng.choice([1, 2]) if transposed else rng.choice([0, 1, 2, 3])
        act = rng.choice([0, 1, 2])
        affine = rng.random() < 0.6 or act != 0
        g = torch.Generator().manual_seed(it)
        if transposed:
            IH, IW = rng.randint(64, 150), rng.randint(64, 150)
            OH, OW = (IH - 1) * 2 - 2 * pad + 4, (IW - 1) * 2 - 2 * pad + 4
            op = 0
            if pad == 2 and rng.random() < 0.5:       # the odd output size a stride-2 pad-2 convolution maps back (output_padding 1)
                OH, OW, op = OH + 1, OW + 1, 1
            w = torch.randn(C0 + C1, Cout, 4, 4, generator=g) * 0.2
            wsco, wsci = 16, Cout * 16
        else:
            IH, IW = rng.randint(256, 300), rng.randint(256, 300)
            OH, OW = (IH + 2 * pad - 4) // 2 + 1, (IW + 2 * pad - 4) // 2 + 1
            w = torch.randn(Cout, C0 + C1, 4, 4, generator=g) * 0.2
            wsco, wsci = (C0 + C1) * 16, 16
        x0, x1 = torch.randn(N, C0, IH, IW, generator=g), torch.randn(N, max(C1, 1), IH, IW, generator=g)
        sc0, sh0 = 1 + 0.3 * torch.randn(N * C0, generator=g), 0.2 * torch.randn(N * C0, generator=g)
        sc1, sh1 = 1 + 0.3 * torch.randn(N * max(C1, 1), generator=g), 0.2 * torch.randn(N * max(C1, 1), generator=g)
        if not affine:
            sc0, sh0, sc1, sh1 = torch.ones_like(sc0), torch.zeros_like(sh0), torch.ones_like(sc1), torch.zeros_like(sh1)
        xs = [apply(x0, sc0, sh0, act)] + ([apply(x1, sc1, sh1, act)] if C1 else [])
        bias = torch.randn(Cout, generator=g) if rng.random() < 0.5 else None
        tanh = transposed and rng.random() < 0.3
        ref = F.conv_transpose2d(torch.cat(xs, 1), w, bias, stride=2, padding=pad, output_padding=op) if transposed else F.conv2d(torch.cat(xs, 1), w, bias, stride=2, padding=pad)
        assert tuple(ref.shape[2:]) == (OH, OW), (ref.shape, OH, OW)
        if tanh:
            ref = torch.tanh(ref)
        dm = torch.randn(ref.shape, generator=g) if (rng.random() < 0.4 and not tanh) else None
        dact = rng.choice([1, 2])
        if dm is not None:
            ref = ref * torch.where(dm > 0, torch.ones_like(dm), torch.full_like(dm, 0.2 if dact == 1 else 0.0))
        base = torch.randn(ref.shape, generator=g) if rng.random() < 0.4 else None
        want = ref + base if base is not None else ref
        out = base.clone().to(dev) if base is not None else torch.full(ref.shape, float("nan"), device=dev)
        a0 = Act(x0.to(dev), sc0.to(dev), sh0.to(dev)) if affine else Act(x0.to(dev))
        a1 = (Act(x1.to(dev), sc1.to(dev), sh1.to(dev)) if affine else Act(x1.to(dev))) if C1 else None
        ops.conv4x4(a0, w.to(dev), wsco, wsci, Cout, out, in1=a1, bias=bias.to(dev) if bias is not None else None, stride=2, pad=pad,
                    transposed=transposed, act_in=act, act_out=3 if tanh else 0, dmask=Act(dm.to(dev)) if dm is not None else None, dmask_act=dact,
                    accumulate=base is not None)
        kern = L.load().vts_last_kernel().decode().split("<")[0]
        used[kern] = used.get(kern, 0) + 1
        err = ((out.cpu().double() - want.double()).norm() / want.double().norm().clamp_min(1e-30)).item()
        worst = max(worst, err)
        if not err < 2e-5:
            print("MISMATCH case %d: %s N%d C%d+%d -> %d in %dx%d out %dx%d pad %d act %d affine %s bias %s tanh %s dmask %s acc %s: rel-L2 %.3e (%s)" % (
                it, "convT" if transposed else "conv", N, C0, C1, Cout, IH, IW, OH, OW, pad, act, affine, bias is not None, tanh, dm is not None, base is not None, err,
                L.load().vts_last_kernel().decode()))
            sys.exit(1)
    print("fuzz ok: %d cases, worst rel-L2 %.2e, kernels %s" % (cases, worst, used))


if __name__ == "__main__":
    main()
