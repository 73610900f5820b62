"""Kernel-level parity: every vts_* entry point (through the C ABI) vs a plain PyTorch fp32
CPU evaluation of the same op on the same seeded inputs.  Tolerances: rel-L2 <= 1e-5 for
single ops (fp32 accumulation order differs), exact for pure data movement."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import detrand, nets  # noqa: E402  (checker only)


def _dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _affine(n, c, seed, name):
    sc = 1.0 + 0.3 * detrand.uniform((n * c,), seed, name + "sc")
    sh = 0.2 * detrand.uniform((n * c,), seed, name + "sh")
    return sc, sh


def _apply(x, sc, sh, act):
    n, c = x.shape[:2]
    v = x * sc.view(n, c, 1, 1) + sh.view(n, c, 1, 1)
    if act == 1:
        v = F.leaky_relu(v, 0.2)
    elif act == 2:
        v = F.relu(v)
    return v


CONV_CASES = [
    # (N, C0, C1, Cout, H, W, stride, pad, act_in)
    (2, 9, 0, 10, 64, 64, 2, 1, 0),
    (1, 10, 0, 20, 50, 70, 2, 1, 1),
    (2, 20, 0, 40, 32, 32, 2, 1, 1),
    (1, 40, 0, 80, 16, 16, 2, 1, 1),
    (1, 80, 0, 80, 4, 4, 2, 1, 1),
    (2, 1, 3, 8, 64, 64, 2, 2, 0),      # D layer 0 with dual source (S ++ I)
    (1, 8, 0, 16, 33, 33, 2, 2, 1),
    (1, 32, 0, 64, 9, 9, 1, 2, 1),
    (2, 64, 0, 1, 10, 10, 1, 2, 1),
    (1, 7, 0, 8, 32, 32, 2, 2, 0),
    (1, 5, 0, 33, 20, 24, 2, 1, 2),
    # small-map (flattened batch) path: N >= 8, maps <= 34
    (9, 7, 0, 8, 32, 32, 2, 2, 0),
    (12, 8, 0, 16, 17, 17, 2, 2, 1),
    (33, 16, 0, 32, 9, 9, 2, 2, 1),
    (16, 32, 0, 64, 5, 5, 1, 2, 1),
    (8, 64, 0, 1, 6, 6, 1, 2, 1),
    (20, 32, 0, 64, 2, 2, 1, 2, 1),
    (10, 3, 4, 8, 16, 12, 2, 2, 0),
    (64, 80, 0, 80, 4, 4, 2, 1, 1),
    # thin full-size path (stride 2, Cin x Cout <= 128, output >= 128 x 128): direct packed-FMA kernel
    (2, 1, 3, 8, 256, 256, 2, 2, 0),
    (1, 9, 0, 10, 300, 260, 2, 1, 0),
    (1, 8, 0, 16, 257, 259, 2, 2, 1),
    (1, 3, 0, 2, 301, 277, 2, 1, 2),
    (1, 5, 2, 13, 270, 256, 2, 2, 1),
    # lane = pixel member (vts_conv_px.hip): stride 2, Cout <= 12, output >= 128 x 128; ragged rows / columns, odd widths, every block count
    (1, 4, 0, 3, 258, 262, 2, 1, 0),
    (2, 3, 2, 12, 257, 300, 2, 2, 1),
    (1, 1, 0, 7, 400, 256, 2, 1, 2),
    (1, 10, 0, 9, 262, 259, 2, 1, 1),
    (1, 6, 0, 5, 261, 257, 2, 2, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward(case):
    from vts import ops
    from vts.ops import Act

    N, C0, C1, Cout, H, W, s, p, act = case
    dev = _dev()
    x0 = detrand.uniform((N, C0, H, W), 1, "x0")
    sc0, sh0 = _affine(N, C0, 1, "a0")
    xs = [_apply(x0, sc0, sh0, act)]
    in1 = None
    if C1:
        x1 = detrand.uniform((N, C1, H, W), 1, "x1")
        sc1, sh1 = _affine(N, C1, 1, "a1")
        xs.append(_apply(x1, sc1, sh1, act))
        in1 = Act(x1.to(dev), sc1.to(dev), sh1.to(dev))
    w = detrand.uniform((Cout, C0 + C1, 4, 4), 2, "w") * 0.2
    b = detrand.uniform((Cout,), 2, "b")
    ref = F.conv2d(torch.cat(xs, 1), w, b, stride=s, padding=p)
    out = torch.full(ref.shape, float("nan"), device=dev)
    ops.conv4x4(Act(x0.to(dev), sc0.to(dev), sh0.to(dev)), w.to(dev), (C0 + C1) * 16, 16, Cout, out, in1=in1, bias=b.to(dev),
                stride=s, pad=p, act_in=act)
    assert rel(out, ref) < 1e-5


def test_thin_stride2_convolutions_take_the_lane_pixel_member():
    """the dispatch of vts_conv4x4: thin stride-2 layers on full-size maps run on conv_px_s2_kernel (4x4x1 MFMA, lane = pixel), with
    dual sources, per-channel affine + LeakyReLU on load, derivative mask and accumulation; wide or small ones do not"""
    from vts import lib as L, ops
    from vts.ops import Act

    dev = _dev()
    N, C0, C1, Cout, H, W = 2, 5, 4, 10, 260, 300
    x0, x1 = detrand.uniform((N, C0, H, W), 7, "x0"), detrand.uniform((N, C1, H, W), 7, "x1")
    (sc0, sh0), (sc1, sh1) = _affine(N, C0, 7, "a0"), _affine(N, C1, 7, "a1")
    w = detrand.uniform((Cout, C0 + C1, 4, 4), 7, "w") * 0.2
    ref = F.conv2d(torch.cat([_apply(x0, sc0, sh0, 1), _apply(x1, sc1, sh1, 1)], 1), w, None, stride=2, padding=1)
    m = detrand.uniform(ref.shape, 7, "m")
    base = detrand.uniform(ref.shape, 7, "base")
    want = base + ref * (m > 0).float()
    out = base.clone().to(dev)
    ops.conv4x4(Act(x0.to(dev), sc0.to(dev), sh0.to(dev)), w.to(dev), (C0 + C1) * 16, 16, Cout, out, in1=Act(x1.to(dev), sc1.to(dev), sh1.to(dev)),
                stride=2, pad=1, act_in=1, dmask=Act(m.to(dev)), dmask_act=L.ACT_RELU, accumulate=True)
    assert L.load().vts_last_kernel().decode().startswith("conv_px_s2_kernel<3,")
    assert rel(out, want) < 1e-5
    # plain operand (no affine, no activation: the gradient case) on an odd input width: the last column pair straddles the row end
    xo = detrand.uniform((1, 3, 261, 259), 7, "xo")
    ref = F.conv2d(xo, w[:, :3], None, stride=2, padding=2)
    out = torch.empty(ref.shape, device=dev)
    ops.conv4x4(Act(xo.to(dev)), w[:, :3].contiguous().to(dev), 3 * 16, 16, Cout, out, stride=2, pad=2)
    assert L.load().vts_last_kernel().decode().startswith("conv_px_s2_kernel<3,") and rel(out, ref) < 1e-5
    for cout, hw in ((40, 260), (10, 64)):     # too wide / too small for it
        out = torch.empty(1, cout, hw // 2, hw // 2, device=dev)
        ops.conv4x4(Act(x0[:1, :, :hw, :hw].contiguous().to(dev)), w[:1].repeat(cout, 1, 1, 1)[:, :C0].contiguous().to(dev), C0 * 16, 16, cout, out, stride=2, pad=1)
        assert not L.load().vts_last_kernel().decode().startswith("conv_px")


CONVT_CASES = [
    # (N, C0, C1, Cout, H, W, stride, pad, act_in, tanh)
    (1, 80, 0, 80, 2, 2, 2, 1, 2, 0),
    (2, 80, 80, 80, 8, 8, 2, 1, 2, 0),
    (1, 80, 80, 40, 16, 16, 2, 1, 2, 0),
    (1, 40, 40, 20, 24, 40, 2, 1, 2, 0),
    (2, 20, 20, 10, 32, 32, 2, 1, 2, 0),
    (2, 10, 0, 3, 64, 64, 2, 1, 2, 1),
    (1, 10, 0, 2, 48, 80, 2, 1, 2, 1),
    (1, 16, 0, 8, 17, 17, 2, 2, 0, 0),   # backward-data geometry of a s2 p2 conv (odd -> even size)
    (1, 64, 0, 32, 10, 10, 1, 2, 0, 0),  # backward-data geometry of a s1 p2 conv
    (1, 1, 0, 64, 11, 11, 1, 2, 0, 0),
    # small-map path
    (16, 64, 0, 32, 6, 6, 1, 2, 0, 0),
    (9, 1, 0, 64, 7, 7, 1, 2, 0, 0),
    (12, 32, 0, 16, 5, 5, 2, 2, 0, 0),
    (10, 16, 0, 8, 9, 9, 2, 2, 0, 0),
    (40, 8, 0, 7, 17, 17, 2, 2, 0, 0),
    (8, 80, 80, 80, 2, 2, 2, 1, 2, 0),
    (8, 10, 0, 3, 16, 16, 2, 1, 2, 1),
    # thin full-size path (Cout <= 16, output >= 128 x 128): direct packed-FMA kernel
    (1, 10, 10, 3, 70, 90, 2, 1, 2, 1),
    (2, 16, 0, 8, 65, 65, 2, 2, 0, 0),
    (1, 10, 0, 10, 64, 72, 2, 1, 2, 0),
    (1, 5, 3, 16, 64, 64, 2, 1, 1, 0),
    (1, 8, 0, 2, 129, 67, 2, 1, 0, 0),
    (2, 5, 0, 13, 80, 66, 2, 2, 1, 0),
    (1, 40, 0, 10, 64, 72, 2, 1, 2, 0),      # same geometry, too many channels for it: MFMA kernel
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transposed_forward(case):
    from vts import ops
    from vts.ops import Act

    N, C0, C1, Cout, H, W, s, p, act, tanh = case
    dev = _dev()
    x0 = detrand.uniform((N, C0, H, W), 3, "x0")
    sc0, sh0 = _affine(N, C0, 3, "a0")
    xs = [_apply(x0, sc0, sh0, act)]
    in1 = None
    if C1:
        x1 = detrand.uniform((N, C1, H, W), 3, "x1")
        sc1, sh1 = _affine(N, C1, 3, "a1")
        xs.append(_apply(x1, sc1, sh1, act))
        in1 = Act(x1.to(dev), sc1.to(dev), sh1.to(dev))
    Cin = C0 + C1
    w = detrand.uniform((Cin, Cout, 4, 4), 4, "w") * 0.2
    b = detrand.uniform((Cout,), 4, "b")
    if s == 2 and p == 2:
        OH, OW = 2 * H - 2, 2 * W - 2  # the size whose s2/p2 conv gives H (even input sizes only)
        ref = F.conv_transpose2d(torch.cat(xs, 1), w, b, stride=2, padding=2)
        assert ref.shape[2] == 2 * H - 2
    elif s == 2:
        ref = F.conv_transpose2d(torch.cat(xs, 1), w, b, stride=2, padding=1)
    else:
        ref = F.conv_transpose2d(torch.cat(xs, 1), w, b, stride=1, padding=p)
    if tanh:
        ref = torch.tanh(ref)
    out = torch.full(ref.shape, float("nan"), device=dev)
    ops.conv4x4(Act(x0.to(dev), sc0.to(dev), sh0.to(dev)), w.to(dev), 16, Cout * 16, Cout, out, in1=in1, bias=b.to(dev), stride=s,
                pad=p, transposed=True, act_in=act, act_out=3 if tanh else 0)
    assert rel(out, ref) < 1e-5


@pytest.mark.parametrize("H", [34, 130, 257])
def test_conv_transposed_odd_output_and_dmask_accumulate(H):
    """backward-data of Conv2d(4->8, s2, p2) at H=34: output 34 from input 18, channel sub-range,
    derivative mask and accumulation -- the G-step path into fake_I (H >= 128: the thin full-size kernel)."""
    from vts import ops
    from vts.ops import Act

    dev = _dev()
    N, Cin, Cout = 2, 4, 8
    x = detrand.uniform((N, Cin, H, H), 5, "x").requires_grad_(True)
    w = detrand.uniform((Cout, Cin, 4, 4), 5, "w") * 0.3
    y = F.conv2d(F.leaky_relu(x, 0.2), w, None, stride=2, padding=2)
    g = detrand.uniform(tuple(y.shape), 5, "g")
    (y * g).sum().backward()
    ref_full = x.grad  # includes the LeakyReLU derivative mask
    prev = detrand.uniform((N, 3, H, H), 5, "prev")
    out = prev.clone().to(dev)
    xin = x.detach().to(dev)
    # channels 1..3 only: offset the weight view by one input channel (16 floats)
    wd = w.to(dev)
    dm = L_operand_slice(xin, 1, 3)
    ops.conv4x4(Act(g.to(dev)), wd.view(-1)[16:], 16, Cin * 16, 3, out, stride=2, pad=2, transposed=True, dmask=dm, dmask_act=1,
                accumulate=True)
    assert rel(out, prev + ref_full[:, 1:4]) < 1e-5


def test_small_map_dmask_accumulate_odd_sizes():
    """small-map path: backward-data of Conv2d(8->16, s2, p2) on 17x17 maps (odd size: 17 -> 9 -> 17),
    with derivative mask and accumulation, batch 24."""
    from vts import ops
    from vts.ops import Act

    dev = _dev()
    N, Cin, Cout, H = 24, 8, 16, 17
    x = detrand.uniform((N, Cin, H, H), 25, "x").requires_grad_(True)
    sc, sh = _affine(N, Cin, 25, "a")
    w = detrand.uniform((Cout, Cin, 4, 4), 25, "w") * 0.3
    xn = x * sc.view(N, Cin, 1, 1) + sh.view(N, Cin, 1, 1)
    y = F.conv2d(F.leaky_relu(xn, 0.2), w, None, stride=2, padding=2)
    g = detrand.uniform(tuple(y.shape), 25, "g")
    (y * g).sum().backward()
    ref = x.grad / sc.view(N, Cin, 1, 1)   # gradient wrt the normalised tensor, incl. the LeakyReLU mask
    prev = detrand.uniform((N, Cin, H, H), 25, "prev")
    out = prev.clone().to(dev)
    ops.conv4x4(Act(g.to(dev)), w.to(dev), 16, Cin * 16, Cin, out, stride=2, pad=2, transposed=True,
                dmask=Act(x.detach().to(dev), sc.to(dev), sh.to(dev)), dmask_act=1, accumulate=True)
    assert rel(out, prev + ref) < 1e-5


def L_operand_slice(t, c0, c):
    from vts import lib as L

    return L.Operand(t[:, c0:].data_ptr(), None, None, c, t.stride(0))


WGRAD_CASES = [
    # (N, CL0, CL1, CH, LH, LW, stride, pad, act_lo, act_hi, convT?)
    (2, 10, 0, 9, 32, 32, 2, 1, 0, 0, False),
    (1, 20, 0, 10, 25, 35, 2, 1, 0, 1, False),
    (2, 80, 0, 80, 4, 4, 2, 1, 0, 1, False),
    (1, 8, 0, 4, 33, 33, 2, 2, 0, 0, False),
    (1, 64, 0, 32, 10, 10, 1, 2, 0, 1, False),
    (2, 1, 0, 64, 11, 11, 1, 2, 0, 1, False),
    (1, 16, 0, 7, 17, 17, 2, 2, 0, 0, False),
    (2, 20, 20, 10, 16, 16, 2, 1, 2, 0, True),
    (1, 80, 80, 40, 8, 8, 2, 1, 2, 0, True),
    (2, 10, 0, 3, 32, 32, 2, 1, 2, 0, True),
    # narrow-tile path (maps at most 8 wide), D2-on-patches shapes
    (32, 64, 0, 32, 6, 6, 1, 2, 0, 1, False),
    (16, 1, 0, 64, 7, 7, 1, 2, 0, 1, False),
    (9, 32, 0, 16, 5, 5, 2, 2, 0, 1, False),
    (12, 16, 0, 8, 3, 3, 2, 2, 0, 1, False),
    (8, 40, 40, 20, 8, 8, 2, 1, 2, 0, True),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad(case):
    from vts import ops
    from vts.ops import Act

    N, CL0, CL1, CH, LH, LW, s, p, act_lo, act_hi, convT = case
    dev = _dev()
    HH, HW = (LH - 1) * s + 4 - 2 * p, (LW - 1) * s + 4 - 2 * p
    lo0 = detrand.uniform((N, CL0, LH, LW), 6, "lo0")
    a0 = _affine(N, CL0, 6, "l0")
    los = [_apply(lo0, a0[0], a0[1], act_lo)]
    lo1_act = None
    if CL1:
        lo1 = detrand.uniform((N, CL1, LH, LW), 6, "lo1")
        a1 = _affine(N, CL1, 6, "l1")
        los.append(_apply(lo1, a1[0], a1[1], act_lo))
        lo1_act = Act(lo1.to(dev), a1[0].to(dev), a1[1].to(dev))
    hi = detrand.uniform((N, CH, HH, HW), 6, "hi")
    ah = _affine(N, CH, 6, "h")
    hiv = _apply(hi, ah[0], ah[1], act_hi)
    lov = torch.cat(los, 1)
    CL = CL0 + CL1
    w = torch.zeros(CL, CH, 4, 4, requires_grad=True)
    # dw[cl][ch][ky][kx] = sum lo * hi(shifted): the weight gradient of conv(hi -> lo) with cotangent lo
    y = F.conv2d(hiv, w, None, stride=s, padding=p)
    assert y.shape[2:] == lov.shape[2:]
    (y * lov).sum().backward()
    ref = w.grad
    dw = torch.full((CL, CH, 4, 4), float("nan"), device=dev)
    ops.wgrad4x4(Act(lo0.to(dev), a0[0].to(dev), a0[1].to(dev)), Act(hi.to(dev), ah[0].to(dev), ah[1].to(dev)), dw, lo1=lo1_act,
                 act_lo=act_lo, act_hi=act_hi, stride=s, pad=p)
    assert rel(dw, ref) < 2e-5
    dw2 = dw.clone()
    ops.wgrad4x4(Act(lo0.to(dev), a0[0].to(dev), a0[1].to(dev)), Act(hi.to(dev), ah[0].to(dev), ah[1].to(dev)), dw2, lo1=lo1_act,
                 act_lo=act_lo, act_hi=act_hi, stride=s, pad=p, accumulate=True)
    assert rel(dw2, 2 * ref) < 2e-5


@pytest.fixture(params=[False, True], ids=["two_launches", "last_block_finalizes"])
def fuse_finalize(request):
    from vts import ops
    keep, ops.FUSE_FINALIZE = ops.FUSE_FINALIZE, request.param
    yield request.param
    ops.FUSE_FINALIZE = keep


@pytest.mark.parametrize("shape", [(2, 10, 64, 64), (1, 80, 2, 2), (3, 20, 37, 53), (1, 3, 128, 128), (2, 4, 300, 300)])
@pytest.mark.parametrize("mode", [0, 1])
def test_norm_forward_backward(shape, mode, fuse_finalize):
    from vts import ops

    dev = _dev()
    n, c, h, w = shape
    x = (detrand.uniform(shape, 7, "x") * 2 + 0.7).requires_grad_(True)
    gamma = (1 + 0.2 * detrand.uniform((c,), 7, "g")).requires_grad_(True)
    beta = (0.1 * detrand.uniform((c,), 7, "b")).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    if mode == 0:
        y = F.instance_norm(x, eps=1e-5)
    else:
        y = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    cot = detrand.uniform(shape, 7, "cot")
    (y * cot).sum().backward()
    xd = x.detach().to(dev)
    rmd, rvd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    nbt = torch.zeros((), dtype=torch.long, device=dev)
    a = ops.norm_stats(xd, mode, gamma=gamma.detach().to(dev) if mode else None, beta=beta.detach().to(dev) if mode else None,
                       running_mean=rmd if mode else None, running_var=rvd if mode else None, nbt=nbt if mode else None)
    yk = xd * a.scale.view(n, c, 1, 1) + a.shift.view(n, c, 1, 1)
    assert rel(yk, y) < 1e-5
    if mode:
        if n * h * w > 1:
            assert rel(rmd, rm) < 1e-5 and rel(rvd, rv) < 1e-5
        assert int(nbt) == 1
    dy = cot.to(dev).clone()
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.norm_bwd(dy, a, mode, gamma=gamma.detach().to(dev) if mode else None, dgamma=dg if mode else None, dbeta=db if mode else None)
    assert rel(dy, x.grad) < 2e-4
    if mode:
        assert rel(dg, gamma.grad) < 1e-4 and rel(db, beta.grad) < 1e-4


FUSED_NORM_CASES = [
    # (N, Cin, Cout, H, W, stride, pad, transposed, mode, groups, offset)
    (2, 10, 20, 256, 256, 2, 1, False, 0, None, 0.0),      # down1-like: tiled kernel, NR 2
    (1, 20, 40, 200, 136, 2, 1, False, 0, None, 0.0),      # ragged edges in both directions, NR 3
    (2, 40, 10, 130, 150, 2, 1, True, 0, None, 0.0),       # up1-like: four parity phases, NR 1
    (1, 80, 20, 96, 100, 2, 1, True, 0, None, 0.0),        # up2-like, NR 2
    (4, 8, 16, 259, 257, 2, 2, False, 1, [0, 1, 3], 0.0),  # D layer 1: BatchNorm over pass groups, pad 2 (odd output size 130 x 129)
    (2, 16, 32, 131, 129, 2, 2, False, 1, None, 0.0),      # D layer 2
    (2, 32, 64, 129, 130, 1, 2, False, 1, [0, 1], 0.0),    # D layer 3: stride 1, NR 4
    (2, 10, 20, 256, 256, 2, 1, False, 0, None, 40.0),     # |mean| >> sigma: the per-wave two-pass partials must stay robust
    (1, 160, 80, 32, 32, 2, 1, True, 0, None, 0.0),        # inner layer: k-split epilogue fusion (InstanceNorm only)
    (3, 32, 64, 20, 20, 1, 2, False, 1, None, 0.0),        # small grid + BatchNorm: must NOT take the InstanceNorm k-split fusion
    (2, 9, 10, 260, 262, 2, 1, False, 0, None, 0.0),       # lane = pixel member (vts_conv_px.hip): statistics per wave of 2 rows x 62 pixels
    (3, 4, 8, 259, 257, 2, 2, False, 1, [0, 1], 0.0),      # ... even padding (63 pixels per wave), BatchNorm pass groups, odd sizes
    (1, 5, 12, 256, 256, 2, 1, False, 0, None, 40.0),      # ... |mean| >> sigma
    (2, 12, 7, 129, 131, 2, 2, True, 1, None, 0.0),        # transposed lane = pixel member, even padding, odd output width (dword stores)
]


@pytest.mark.parametrize("case", FUSED_NORM_CASES)
def test_conv_with_statistics_from_the_epilogue(case):
    """vts_conv4x4_norm: the statistics the convolution's epilogue emits (per-wave partials merged by the second stage of
    vts_norm_stats) against the stand-alone statistics pass on the same output and against F.instance_norm / F.batch_norm of a
    PyTorch evaluation of the convolution: scale / shift, saved mean / rstd, running statistics, num_batches_tracked"""
    from vts import lib as L, ops

    dev = _dev()
    N, Cin, Cout, H, W, stride, pad, transposed, mode, groups, offset = case
    x = detrand.uniform((N, Cin, H, W), 31, "x")
    wshape = (Cin, Cout, 4, 4) if transposed else (Cout, Cin, 4, 4)
    w = detrand.uniform(wshape, 31, "w") * (1.0 / (Cin * 16) ** 0.5)
    b = 0.1 * detrand.uniform((Cout,), 31, "b") + offset
    if transposed:
        ref = F.conv_transpose2d(F.leaky_relu(x, 0.2), w, b, stride=stride, padding=pad)
    else:
        ref = F.conv2d(F.leaky_relu(x, 0.2), w, b, stride=stride, padding=pad)
    OH, OW = ref.shape[2:]
    gamma, beta = 1 + 0.2 * detrand.uniform((Cout,), 31, "g"), 0.1 * detrand.uniform((Cout,), 31, "bt")
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)

    def run(fuse):
        import os
        out = torch.empty(N, Cout, OH, OW, device=dev)
        rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
        nbt = torch.zeros((), dtype=torch.long, device=dev)
        kw = dict(gamma=gamma.to(dev), beta=beta.to(dev), running_mean=rm, running_var=rv, nbt=nbt, groups=groups) if mode else None
        args = (16, Cout * 16) if transposed else (Cin * 16, 16)
        if fuse:
            calls, keep = [], ops.norm_stats
            ops.norm_stats = lambda *a_, **k_: (calls.append(1), keep(*a_, **k_))[1]
            try:
                a = ops.conv4x4(xd, wd, args[0], args[1], Cout, out, bias=bd, stride=stride, pad=pad, transposed=transposed, act_in=L.ACT_LRELU,
                                instance_norm=mode == 0, batch_norm=kw)
            finally:
                ops.norm_stats = keep
            kern = "standalone" if calls else "fused"
        else:
            ops.conv4x4(xd, wd, args[0], args[1], Cout, out, bias=bd, stride=stride, pad=pad, transposed=transposed, act_in=L.ACT_LRELU)
            a = ops.norm_stats(out, mode, **(kw or {}))
            kern = ""
        return out, a, rm, rv, nbt, kern

    out_f, a_f, rm_f, rv_f, nbt_f, kern = run(True)
    out_u, a_u, rm_u, rv_u, nbt_u, _ = run(False)
    assert torch.equal(out_f, out_u)                        # the convolution itself is untouched by the fusion
    assert rel(out_f, ref) < 1e-5
    # against the stand-alone pass: same Chan merge, other partition of the plane -> equal to rounding
    for name in ("scale", "shift", "mean", "rstd"):
        assert rel(getattr(a_f, name), getattr(a_u, name)) < 2e-6, name
    # against PyTorch
    bounds = (list(groups) if groups else [0]) + [N]
    if mode == 0:
        y = F.instance_norm(ref, eps=1e-5)
    else:
        rm, rv = torch.zeros(Cout), torch.ones(Cout)
        y = torch.cat([F.batch_norm(ref[bounds[i]:bounds[i + 1]], rm, rv, gamma, beta, True, 0.1, 1e-5) for i in range(len(bounds) - 1)])
        assert rel(rm_f, rm) < 1e-5 and rel(rv_f, rv) < 1e-5 and int(nbt_f) == len(bounds) - 1 == int(nbt_u)
    yk = out_f * a_f.scale.view(N, Cout, 1, 1) + a_f.shift.view(N, Cout, 1, 1)
    assert rel(yk, y) < (2e-4 if offset else 1e-5)
    # no stand-alone statistics pass followed, except where a BatchNorm layer runs on the small-grid (k-split) path
    assert kern == ("standalone" if (mode == 1 and OH * OW <= 4096) else "fused"), kern


BWD_SUM_CASES = [
    # (N, Cg, Cx, GH, GW, stride, pad, transposed, mode, groups, accumulate)   g [N,Cg,GH,GW] -> dx [N,Cx,..] masked by the normalised x
    (2, 10, 40, 128, 128, 2, 1, False, 0, None, False),    # up-block input gradient: conv s2 of the output gradient, ReLU mask, InstanceNorm below
    (1, 20, 10, 96, 136, 2, 1, True, 0, None, True),       # encoder: transposed s2, LeakyReLU mask, accumulated onto the skip contribution
    (4, 32, 16, 66, 65, 2, 2, True, 1, [0, 1, 3], False),  # PatchGAN: transposed s2 pad 2, BatchNorm with pass groups below
    (2, 64, 32, 130, 131, 1, 2, True, 1, None, False),     # PatchGAN stride-1 layer
    (2, 80, 80, 32, 32, 2, 1, False, 0, None, False),      # inner up-block: small-grid k-split path, InstanceNorm backward applied by its epilogue
    (3, 80, 80, 8, 8, 2, 1, True, 0, None, True),          # inner down-block: transposed, k-split, accumulated
    (2, 3, 10, 256, 260, 2, 1, False, 0, None, False),     # outermost up-block (lane = pixel member): 3 -> 10, ReLU mask
    (1, 7, 12, 262, 300, 2, 1, False, 0, None, True),      # ... 7 -> 12 accumulated onto the skip contribution, ragged tiles
]


@pytest.mark.parametrize("case", BWD_SUM_CASES)
def test_backward_data_conv_emits_the_norm_backward_sums(case):
    """vts_conv4x4_bsums + vts_norm_bwd_from_partials (the sums of the normalisation backward taken in the epilogue of the
    convolution that produces its input gradient) against the plain sequence vts_conv4x4 + vts_norm_bwd on the same operands:
    gradient w.r.t. the raw tensor, dgamma / dbeta"""
    from vts import lib as L, ops

    dev = _dev()
    N, Cg, Cx, GH, GW, stride, pad, transposed, mode, groups, accumulate = case
    g = detrand.uniform((N, Cg, GH, GW), 51, "g").to(dev)
    if transposed:
        XH, XW = (GH - 1) * stride + 4 - 2 * pad, (GW - 1) * stride + 4 - 2 * pad
        if stride == 2:       # the forward conv must map the size back (output_padding ambiguity)
            XH += (XH + 2 * pad - 4) % 2
            XW += (XW + 2 * pad - 4) % 2
        w = (detrand.uniform((Cg, Cx, 4, 4), 51, "w") * 0.05).to(dev)           # Conv2d weight [Cout = Cg, Cin = Cx]
        wargs = (16, Cx * 16)
    else:
        XH, XW = (GH + 2 * pad - 4) // stride + 1, (GW + 2 * pad - 4) // stride + 1
        w = (detrand.uniform((Cx, Cg, 4, 4), 51, "w") * 0.05).to(dev)           # ConvTranspose2d weight [Cin = Cx, Cout = Cg]
        wargs = (Cg * 16, 16)
    x = (detrand.uniform((N, Cx, XH, XW), 51, "x") * 2 + 0.4).to(dev)
    gamma, beta = (1 + 0.2 * detrand.uniform((Cx,), 51, "ga")).to(dev), (0.1 * detrand.uniform((Cx,), 51, "be")).to(dev)
    a = ops.norm_stats(x, mode, gamma=gamma if mode else None, beta=beta if mode else None, groups=groups)
    base = detrand.uniform((N, Cx, XH, XW), 51, "acc").to(dev)
    act = L.ACT_LRELU if transposed else L.ACT_RELU

    def run(fused):
        keep, ops.BWD_SUMS = ops.BWD_SUMS, fused
        try:
            dx = base.clone() if accumulate else torch.empty_like(base)
            ops.conv4x4(ops.Act(g), w, wargs[0], wargs[1], Cx, dx, stride=stride, pad=pad, transposed=transposed, dmask=a, dmask_act=act,
                        accumulate=accumulate, bwd_sums=True if mode else "in")
            had = dx.data_ptr() in ops.BSUMS
            dg, db = torch.zeros(Cx, device=dev), torch.zeros(Cx, device=dev)
            ops.norm_bwd(dx, a, mode, gamma=gamma if mode else None, dgamma=dg if mode else None, dbeta=db if mode else None, groups=groups,
                         beta=beta if mode else None)
            return dx, dg, db, had
        finally:
            ops.BWD_SUMS = keep

    dx_f, dg_f, db_f, had = run(True)
    dx_p, dg_p, db_p, had_p = run(False)
    assert had and not had_p and not ops.BSUMS                # the fused run really took the epilogue sums, and consumed them
    assert rel(dx_f, dx_p) < 2e-5
    if mode:
        assert rel(dg_f, dg_p) < 2e-5 and rel(db_f, db_p) < 2e-5


@pytest.mark.parametrize("shape,groups", [((7, 6, 9, 9), [0, 3, 5]), ((5, 4, 70, 61), [0, 2]), ((640, 8, 5, 5), [0, 256, 384]),
                                          ((6, 3, 150, 140), [0, 1, 4])])
def test_batchnorm_batched_passes_equal_sequential_calls(shape, groups):
    """several discriminator passes in ONE launch (vts_norm_desc.ngroups): per-pass batch statistics, running statistics advanced
    in pass order with a separately launched pass spliced in after pass 0 -- equal to sequential F.batch_norm calls"""
    from vts import ops

    dev = _dev()
    n, c, h, w = shape
    x = (detrand.uniform(shape, 17, "x") * 2 + 0.3).requires_grad_(True)
    xe = detrand.uniform((2, c, h + 3, w + 1), 17, "xe") * 3 - 0.4          # the spliced pass (its own launch, other spatial size)
    gamma = (1 + 0.2 * detrand.uniform((c,), 17, "g")).requires_grad_(True)
    beta = (0.1 * detrand.uniform((c,), 17, "b")).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    bounds = list(groups) + [n]
    ys = []
    for gi in range(len(groups)):
        ys.append(F.batch_norm(x[bounds[gi]:bounds[gi + 1]], rm, rv, gamma, beta, True, 0.1, 1e-5))
        if gi == 0:
            F.batch_norm(xe, rm, rv, gamma, beta, True, 0.1, 1e-5)
    y = torch.cat(ys)
    cot = detrand.uniform(shape, 17, "cot")
    (y * cot).sum().backward()

    gd, bd = gamma.detach().to(dev), beta.detach().to(dev)
    rmd, rvd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    nbt = torch.zeros((), dtype=torch.long, device=dev)
    stat = (torch.empty(c, device=dev), torch.empty(c, device=dev))
    ops.norm_stats(xe.to(dev), 1, gamma=gd, beta=bd, stat_out=stat)          # records its statistics, no running update
    xd = x.detach().to(dev)
    a = ops.norm_stats(xd, 1, gamma=gd, beta=bd, running_mean=rmd, running_var=rvd, nbt=nbt, groups=groups, ext=(stat[0], stat[1], 0))
    yk = xd * a.scale.view(n, c, 1, 1) + a.shift.view(n, c, 1, 1)
    assert rel(yk, y) < 1e-5
    assert rel(rmd, rm) < 1e-5 and rel(rvd, rv) < 1e-5
    assert int(nbt) == len(groups) + 1
    dy = cot.to(dev).clone()
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    ops.norm_bwd(dy, a, 1, gamma=gd, dgamma=dg, dbeta=db, groups=groups)
    assert rel(dy, x.grad) < 2e-4
    assert rel(dg, gamma.grad) < 1e-4 and rel(db, beta.grad) < 1e-4


def test_norm_large_mean_is_robust():
    from vts import ops

    dev = _dev()
    x = detrand.uniform((1, 2, 96, 96), 8, "x") * 0.01 + 100.0
    y = F.instance_norm(x.double(), eps=1e-5).float()
    a = ops.norm_stats(x.to(dev), 0)
    yk = x.to(dev) * a.scale.view(1, 2, 1, 1) + a.shift.view(1, 2, 1, 1)
    # scale*x + shift form loses digits at |mean| >> std; statistics themselves must be right
    m = x.double().mean(dim=(2, 3)).flatten()
    v = x.double().var(dim=(2, 3), unbiased=False).flatten()
    assert rel(a.mean, m) < 1e-6
    assert rel(a.rstd, 1 / torch.sqrt(v + 1e-5)) < 1e-4
    assert (yk.cpu() - y).abs().max() < 0.1


def test_channel_sum_and_act_bwd(fuse_finalize):
    from vts import ops
    from vts.ops import Act

    dev = _dev()
    x = detrand.uniform((3, 5, 40, 33), 9, "x")
    out = torch.zeros(5, device=dev)
    ops.channel_sum(x.to(dev), out)
    assert rel(out, x.sum(dim=(0, 2, 3))) < 1e-5
    sc, sh = _affine(3, 5, 9, "a")
    g = detrand.uniform((3, 5, 40, 33), 9, "g")
    v = (x * sc.view(3, 5, 1, 1) + sh.view(3, 5, 1, 1))
    for kind, ref in ((1, g * torch.where(v > 0, 1.0, 0.2)), (2, g * (v > 0).float())):
        dy = torch.ones(3, 5, 40, 33, device=dev)
        ops.act_bwd(g.to(dev), Act(x.to(dev), sc.to(dev), sh.to(dev)), kind, dy, accumulate=True)
        assert rel(dy, ref + 1) < 1e-6


@pytest.mark.parametrize("hw", [(64, 64), (33, 47), (2, 2), (513, 17), (66, 130), (33, 46), (5, 4), (131, 258)])
def test_avgpool(hw):
    from vts import ops

    dev = _dev()
    x = detrand.uniform((2, 4, hw[0], hw[1]), 10, "x").requires_grad_(True)
    y = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)
    cot = detrand.uniform(tuple(y.shape), 10, "c")
    (y * cot).sum().backward()
    yk = ops.avgpool(x.detach().to(dev))
    assert rel(yk, y) < 1e-6
    dx = torch.ones(2, 4, hw[0], hw[1], device=dev)
    ops.avgpool_bwd(cot.to(dev), dx, accumulate=True)
    assert rel(dx, x.grad + 1) < 1e-6


@pytest.mark.parametrize("mode", ["nonsaturating", "lsgan", "vanilla", "wgan", "hinge"])
@pytest.mark.parametrize("real", [True, False])
def test_ganloss(mode, real):
    from vts import ops

    dev = _dev()
    p = (detrand.uniform((3, 1, 9, 11), 11, "p") * 4).requires_grad_(True)
    ref = nets.gan_loss_single(p, real, mode, 0.8, 0.0)
    ref = ref.mean() * 2.5
    ref.backward()
    slot = ops.loss_slots(1, dev)
    dp = torch.empty(3, 1, 9, 11, device=dev)
    ops.ganloss(p.detach().to(dev), mode, real, 2.5, slot, dp, label=0.8 if real else 0.0)
    assert abs(ops.loss_values(slot)[0] - ref.item()) < 1e-5 * max(1, abs(ref.item()))
    assert rel(dp, p.grad) < 1e-5


def test_l1_and_adam():
    from vts import ops

    dev = _dev()
    a = detrand.uniform((2, 3, 17, 19), 12, "a").requires_grad_(True)
    b = detrand.uniform((2, 3, 17, 19), 12, "b")
    ref = F.l1_loss(a, b) * 100
    ref.backward()
    slot = ops.loss_slots(1, dev)
    g = torch.empty(2, 3, 17, 19, device=dev)
    ops.l1(a.detach().to(dev), b.to(dev), 100.0 / a.numel(), slot, g)
    assert abs(ops.loss_values(slot)[0] - ref.item()) < 1e-4 * abs(ref.item())
    assert rel(g, a.grad) < 1e-6
    # Adam, 3 steps, beta1 = 0 like the reference
    p = detrand.uniform((1000,), 13, "p")
    m, v = torch.zeros(1000), torch.zeros(1000)
    pd, md, vd = p.clone().to(dev), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    for step in range(1, 4):
        gr = detrand.uniform((1000,), 13, "g%d" % step) * 10 ** (-step * 2)
        nets.adam_update(p, gr, m, v, step, 1e-3, 0.0, 0.99)
        ops.adam_flat(pd, gr.to(dev), md, vd, 1e-3, 0.0, 0.99, 1e-8, step)
    assert rel(pd, p) < 1e-6 and rel(vd, v) < 1e-5


def test_patch_gather_scatter():
    from vts import ops

    dev = _dev()
    N, C, H, W, PPI = 2, 2, 96, 80, 7
    src = detrand.uniform((N, C, H, W), 14, "src").requires_grad_(True)
    ox = torch.tensor([0, 10, 60, 70, 33, 5, 10, -3, 12, 50, 60, 1, 2, 12], dtype=torch.int32)  # duplicates + border clamps
    oy = torch.tensor([0, 20, 80, 5, 64, 90, 20, -5, 30, 70, 1, 2, 3, 30], dtype=torch.int32)
    img = torch.arange(N).repeat_interleave(PPI).int()
    outs = [nets.gather_patches(src[n:n + 1], ox[n * PPI:(n + 1) * PPI], oy[n * PPI:(n + 1) * PPI], 32) for n in range(N)]
    ref = torch.cat(outs, 0)
    cot = detrand.uniform(tuple(ref.shape), 14, "cot")
    (ref * cot).sum().backward()
    out = torch.zeros(N * PPI, 5, 32, 32, device=dev)
    ops.patch_gather(src.detach().to(dev), img.to(dev), ox.to(dev), oy.to(dev), 32, out, c0=2)
    assert torch.equal(out[:, 2:4].cpu(), ref.detach())
    assert float(out[:, :2].abs().sum()) == 0
    dp = torch.zeros(N * PPI, 5, 32, 32)
    dp[:, 2:4] = cot
    dsrc = torch.full((N, C, H, W), 7.0, device=dev)
    ops.patch_scatter_bwd(dp.to(dev), 2, C, ox.to(dev), oy.to(dev), PPI, 32, dsrc)
    assert rel(dsrc, src.grad) < 1e-6


def test_patch_jobs_equal_single_gathers_copies_and_fill():
    """vts_patch_jobs (all channel runs of the D2 stacks in one launch) against vts_patch_gather / copy_ / fill_ one by one, including a
    channel-slice source with a batch stride (the augmented image lives inside the 7-channel full-resolution stack) and border clamps"""
    from vts import ops

    dev = _dev()
    n, h, w, P = 2, 70, 90, 10
    full = detrand.uniform((n, 7, h, w), 41, "full").to(dev)
    img = torch.tensor([0, 1] * 5, dtype=torch.int32, device=dev)
    offx = torch.tensor([-3, 0, 10, 60, 70, 85, 5, 33, 58, 1], dtype=torch.int32, device=dev)
    offy = torch.tensor([0, -5, 40, 50, 3, 60, 38, 39, 12, 69], dtype=torch.int32, device=dev)
    pt = detrand.uniform((P, 2, 32, 32), 41, "pt").to(dev)
    msk = (detrand.uniform((P, 1, 32, 32), 41, "m") > 0).float().to(dev)
    a, b = torch.zeros(P, 7, 32, 32, device=dev), torch.zeros(P, 7, 32, 32, device=dev)
    g = dict(img=img, offx=offx, offy=offy)
    ops.patch_jobs([dict(dst=a, c0=0, src=full[:, 0:2], channels=2, **g), dict(dst=a, c0=2, src=full[:, 2:3], **g),
                    dict(dst=a, c0=3, src=full[:, 3:6], **g), dict(dst=a, c0=6, src=msk)])
    ops.patch_gather(full[:, 0:2], img, offx, offy, 32, b, c0=0, channels=2)
    ops.patch_gather(full[:, 2:3], img, offx, offy, 32, b, c0=2, channels=1)
    ops.patch_gather(full[:, 3:6], img, offx, offy, 32, b, c0=3, channels=3)
    b[:, 6:7].copy_(msk)
    assert torch.equal(a, b)
    c = torch.zeros(P, 3, 32, 32, device=dev)
    ops.patch_jobs([dict(dst=c, c0=0, src=pt), dict(dst=c, c0=2, channels=1, fill=1.0)])
    assert torch.equal(c[:, 0:2], pt) and bool((c[:, 2] == 1).all())


def test_g_post_diffaug_outgrad_spe():
    from vts import ops

    dev = _dev()
    N, H, W = 2, 40, 56
    g = torch.tanh(detrand.uniform((N, 5, H, W), 15, "g") * 2)
    M = (detrand.uniform((N, 1, H, W), 15, "m") > -0.5).float()
    rb, rs = torch.tensor([0.3, 0.9]), torch.tensor([0.1, 0.7])
    fI, fT = g[:, :3] * M, g[:, 3:] * M
    outs = [torch.empty(N, c, H, W, device=dev) for c in (3, 2, 3, 3)]
    ops.g_post(g.to(dev), M.to(dev), 0.25, rb.to(dev), rs.to(dev), *outs)
    assert rel(outs[0], fI) < 1e-6 and rel(outs[1], fT) < 1e-6
    assert rel(outs[2], nets.compute_normal(fT, 0.25)) < 1e-6
    assert rel(outs[3], nets.diffaug_bs(fI, rb, rs) * M) < 1e-6
    aug = torch.empty(N, 3, H, W, device=dev)
    ops.diffaug_bs_mask(fI.to(dev), M.to(dev), rb.to(dev), rs.to(dev), aug)
    assert rel(aug, nets.diffaug_bs(fI, rb, rs) * M) < 1e-6
    dI, dT = detrand.uniform((N, 3, H, W), 15, "dI"), detrand.uniform((N, 2, H, W), 15, "dT")
    d = torch.empty(N, 5, H, W, device=dev)
    ops.g_out_grad(dI.to(dev), dT.to(dev), M.to(dev), g.to(dev), d)
    assert rel(d, torch.cat([dI, dT], 1) * M * (1 - g * g)) < 1e-6
    buf = torch.zeros(N, 9, H, W, device=dev)
    ops.spe_grid(buf, 4, c0=1)
    assert (buf[:, 1:].cpu() - nets.spe_grid(N, H, W)).abs().max() < 2e-6
    big = torch.zeros(1, 8, 1100, 1030, device=dev)
    ops.spe_grid(big, 4)
    assert (big.cpu() - nets.spe_grid(1, 1100, 1030)).abs().max() < 1e-4
    y = ops.mask_mul(dI.to(dev), M.to(dev))
    assert rel(y, dI * M) < 1e-7


@pytest.mark.parametrize("hw", [(64, 96), (65, 97), (33, 50)])
def test_g_out_grad_with_the_pyramid_merge_equals_the_two_launch_form_bit_for_bit(hw):
    """vts_g_out_grad_pool: the adjoint of the pyramid's average pool (AvgPool2d(3, 2, 1, count_include_pad=False)) of the coarse image
    gradient added on the fly == vts_avgpool3s2_bwd(accumulate) + vts_g_out_grad, bitwise (even and odd map sizes: border windows hold
    4 / 6 / 9 taps), and == the autograd adjoint of torch's pool"""
    import torch.nn.functional as F
    from vts import ops

    dev = _dev()
    N, (H, W) = 2, hw
    OH, OW = (H + 1) // 2, (W + 1) // 2
    dI, dT = detrand.uniform((N, 3, H, W), 25, "dI").to(dev), detrand.uniform((N, 2, H, W), 25, "dT").to(dev)
    dc = detrand.uniform((N, 3, OH, OW), 25, "dIc").to(dev)
    M = (detrand.uniform((N, 1, H, W), 25, "M") > 0).float().to(dev)
    g = torch.tanh(detrand.uniform((N, 5, H, W), 25, "g")).to(dev)
    fused = ops.g_out_grad(dI, dT, M, g, torch.empty(N, 5, H, W, device=dev), coarse=dc)
    dI2 = dI.clone()
    ops.avgpool_bwd(dc, dI2, accumulate=True)
    two = ops.g_out_grad(dI2, dT, M, g, torch.empty(N, 5, H, W, device=dev))
    assert torch.equal(fused, two)
    x = torch.zeros(N, 3, H, W, requires_grad=True)
    F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False).backward(dc.cpu())
    want = torch.cat([dI.cpu() + x.grad, dT.cpu()], 1) * M.cpu() * (1 - g.cpu() * g.cpu())
    assert rel(fused, want) < 1e-6


def test_mask_candidates_and_select():
    import random

    from vts import ops

    dev = _dev()
    M = torch.zeros(2, 1, 80, 96)
    M[0, 0, 20:40, 25:50] = 1
    M[1, 0, 5:70, 60:90] = 1
    M[1, 0, 0, 0] = 1
    cand, prefix = ops.mask_candidates(M.to(dev))
    ranks = []
    exp_x, exp_y = [], []
    random.seed(3)
    for n in range(2):
        pos = nets.dilated_mask_positions(M[n:n + 1])
        assert int(prefix[n, -1]) == pos.shape[0]
        r = random.sample(range(pos.shape[0]), 9)
        r[0], r[1] = 0, pos.shape[0] - 1
        ranks.append(r)
        exp_y += pos[r][:, 0].tolist()
        exp_x += pos[r][:, 1].tolist()
    ox, oy = ops.mask_select(cand, prefix, torch.tensor(ranks, dtype=torch.int64, device=dev), 80, 96)
    assert ox.cpu().tolist() == exp_x and oy.cpu().tolist() == exp_y


def test_device_side_rank_sampler_draws_distinct_uniform_ranks():
    """vts_mask_sample_ranks = random.sample(range(candidates), K) (models/model_utils.py:217) on the device: K distinct ranks below
    every image's own candidate count, a function of the seed only, every rank equally likely; fewer candidates than K wrap"""
    from vts import ops

    dev = _dev()
    N, H, W, K = 3, 64, 80, 32
    M = torch.zeros(N, 1, H, W)
    M[0, 0, 20:40, 10:50] = 1
    M[1, 0, 5:8, 5:9] = 1
    M[2, 0, 30, 40] = 1
    cand, prefix = ops.mask_candidates(M.to(dev))
    counts = prefix[:, -1].tolist()
    assert counts == [int(c) for c in cand.view(N, -1).sum(1).tolist()] and min(counts) > K
    r1 = ops.mask_sample_ranks(prefix, H, K, 1234, torch.empty(N, K, dtype=torch.int64, device=dev)).cpu()
    r2 = ops.mask_sample_ranks(prefix, H, K, 1234, torch.empty(N, K, dtype=torch.int64, device=dev)).cpu()
    r3 = ops.mask_sample_ranks(prefix, H, K, 1235, torch.empty(N, K, dtype=torch.int64, device=dev)).cpu()
    assert torch.equal(r1, r2) and not torch.equal(r1, r3)
    for n in range(N):
        assert len(set(r1[n].tolist())) == K and 0 <= int(r1[n].min()) and int(r1[n].max()) < counts[n]
    # the ranks resolve to candidate positions (the consumer of the ranks)
    offx, offy = ops.mask_select(cand, prefix, r1.to(dev), H, W)
    assert bool(cand[torch.arange(N).repeat_interleave(K), offy.long().cpu(), offx.long().cpu()].all())
    # uniformity: image 2 has 17 x 17 = 289 candidates; over 400 seeds every rank is drawn about 400 * 32 / 289 = 44 times
    hist = torch.zeros(counts[2])
    for seed in range(400):
        r = ops.mask_sample_ranks(prefix, H, K, seed * 7919 + 1, torch.empty(N, K, dtype=torch.int64, device=dev)).cpu()
        hist += torch.bincount(r[2], minlength=counts[2])
    assert float(hist.min()) > 15 and float(hist.max()) < 80 and abs(float(hist.mean()) - 400 * K / counts[2]) < 1e-3
    # fewer candidates than K (the reference raises there): ranks wrap, nothing out of range
    M1 = torch.zeros(1, 1, H, W)
    M1[0, 0, 0, 0] = 1
    c1, p1 = ops.mask_candidates(M1.to(dev))
    rw = ops.mask_sample_ranks(p1, H, 16, 5, torch.empty(1, 16, dtype=torch.int64, device=dev)).cpu()
    assert int(p1[0, -1]) < 16 and int(rw.max()) < int(p1[0, -1])


@pytest.mark.parametrize("allneg", [False, True])
def test_patchnce(allneg):
    from vts import ops

    dev = _dev()
    B, P, D = 2, 48, 40
    q0 = detrand.uniform((B * P, D), 16, "q")
    k0 = detrand.uniform((B * P, D), 16, "k")
    qn = ops.l2norm_rows(q0.to(dev))
    assert rel(qn, nets.l2_normalize(q0)) < 1e-6
    q = nets.l2_normalize(q0).requires_grad_(True)
    k = nets.l2_normalize(k0)
    ref = nets.patchnce_loss(q, k, B, 0.07, allneg)
    ref.sum().backward()
    loss, dq = ops.patchnce(q.detach().to(dev), k.to(dev), 1 if allneg else B, 0.07)
    assert rel(loss, ref) < 1e-5 and rel(dq, q.grad) < 1e-5


def test_patchsample_f_and_patchnce_at_reference_size(golden_dir):
    """PatchSampleF (gather, MLP on the MFMA tile routine, L2 norm) and PatchNCE at the reference's 256 patches x 256 dims (both
    negative modes: per image on the MFMA kernel, whole minibatch on the one-query-per-workgroup kernel) against the vectors the
    REFERENCE modules produced (tests/golden/patchsample.npz) and against the oracle for the full gradient"""
    import os

    from models.networks import PatchNCELoss, PatchSampleF
    from types import SimpleNamespace
    from vts import lib as L

    dev = _dev()
    g = np.load(os.path.join(golden_dir, "patchsample.npz"))
    feats = [detrand.uniform((2, 24, 20, 18), 41, "f0"), detrand.uniform((2, 40, 9, 11), 41, "f1")]
    ids = [g["ids0"], g["ids1"]]
    fd = [f.to(dev) for f in feats]
    plain, rid = PatchSampleF(use_mlp=False)(fd, 256, ids)
    assert rel(plain[0], torch.from_numpy(g["plain0"])) < 1e-6 and rel(plain[1], torch.from_numpy(g["plain1"])) < 1e-6
    assert rid[0].cpu().tolist() == ids[0].tolist()
    f = PatchSampleF(use_mlp=True, nc=256)
    f.create_mlp(fd)
    assert sorted(f.state_dict().keys()) == [str(k) for k in g["mlp_keys"]]
    for i in range(2):
        m = getattr(f, "mlp_%d" % i)
        assert abs(float(m[0].weight.std()) - 0.02) < 0.004 and float(m[0].bias.abs().max()) == 0.0      # init_net normal / 0.02
        for idx, (kw, kb) in ((0, ("w0", "b0")), (2, ("w2", "b2"))):
            m[idx].weight.data.copy_(torch.from_numpy(g["mlp%d_%s" % (i, kw)].astype(np.float32)))
            m[idx].bias.data.copy_(torch.from_numpy(g["mlp%d_%s" % (i, kb)]))
    fm, _ = f(fd, 256, ids)
    assert rel(fm[0][::8], torch.from_numpy(g["mlp0_sub"])) < 1e-5 and rel(fm[1][::8], torch.from_numpy(g["mlp1_sub"])) < 1e-5
    fk, _ = f([detrand.uniform((2, 24, 20, 18), 43, "k0").to(dev), fd[1]], 256, ids)
    for allneg in (False, True):
        crit = PatchNCELoss(SimpleNamespace(nce_includes_all_negatives_from_minibatch=allneg, batch_size=2, nce_T=0.07))
        loss, dq = crit(fm[0], fk[0], want_grad=True)
        assert L.load().vts_last_kernel().decode() == ("patchnce_kernel" if allneg else "patchnce_mfma_kernel")
        assert rel(loss, torch.from_numpy(g["nce_loss_%d" % allneg])) < 2e-5
        assert rel(dq[::8, ::4], torch.from_numpy(g["nce_dq_sub_%d" % allneg])) < 1e-4
        q = fm[0].cpu().clone().requires_grad_(True)
        ref = nets.patchnce_loss(q, fk[0].cpu(), 2, 0.07, allneg)
        gref, = torch.autograd.grad(ref.sum(), q)
        assert rel(dq, gref) < 1e-4 and rel(loss, ref) < 2e-5


@pytest.mark.parametrize("B,P,D", [(3, 48, 40), (1, 200, 100), (2, 17, 256), (2, 256, 19)])
def test_patchnce_mfma_ragged_sizes(B, P, D):
    from vts import ops

    dev = _dev()
    q = nets.l2_normalize(detrand.uniform((B * P, D), 5, "q") - 0.5).requires_grad_(True)
    k = nets.l2_normalize(detrand.uniform((B * P, D), 5, "k") - 0.5)
    ref = nets.patchnce_loss(q, k, B, 0.07, False)
    gref, = torch.autograd.grad((ref * 0.5).sum(), q)
    loss, dq = ops.patchnce(q.detach().to(dev), k.to(dev), B, 0.07, gscale=0.5)
    assert rel(loss, ref) < 2e-5 and rel(dq, gref) < 1e-4


def test_linear_rows_matches_torch():
    from vts import ops

    dev = _dev()
    for r, i, o in ((512, 24, 256), (70, 300, 33), (64, 4, 600)):
        x, w, b = detrand.uniform((r, i), 6, "x") - 0.5, detrand.uniform((o, i), 6, "w") - 0.5, detrand.uniform((o,), 6, "b")
        for relu in (False, True):
            y = ops.linear_rows(x.to(dev), w.to(dev), b.to(dev), relu=relu)
            ref = F.linear(x, w, b)
            assert rel(y, F.relu(ref) if relu else ref) < 1e-5


@pytest.mark.parametrize("geom", [(32, 32, 64, 64), (40, 40, 32, 32), (37, 53, 32, 32), (32, 32, 128, 128), (96, 80, 24, 20), (32, 32, 32, 32)])
def test_bicubic_antialias_resampler_and_adjoint(geom):
    """vts_resample_table with PyTorch's anti-aliased bicubic tables vs F.interpolate(mode="bicubic", align_corners=False,
    antialias=True) -- the resampling the reference applies to patches / images when T_resolution_multiplier is 2 or 4 or a patch
    cut-out is not 32 px (sinskitG_model.py:1440-1476, 1531-1557) -- and its adjoint vs autograd; equal sizes are the identity"""
    from vts import ops
    ih, iw, oh, ow = geom
    x = detrand.uniform((3, 2, ih, iw), 77, "rs_x")
    xo = x.clone().requires_grad_(True)
    ref = F.interpolate(xo, (oh, ow), mode="bicubic", align_corners=False, antialias=True)
    got = ops.bicubic_aa(x.to(_dev()), (oh, ow))
    assert rel(got, ref) < 2e-6
    cot = detrand.uniform(tuple(ref.shape), 77, "rs_c")
    (ref * cot).sum().backward()
    din = ops.bicubic_aa_bwd(cot.to(_dev()), (ih, iw))
    assert rel(din, xo.grad) < 2e-6
    acc = ops.bicubic_aa_bwd(cot.to(_dev()), (ih, iw), din=din.clone(), accumulate=True)
    assert rel(acc, 2 * xo.grad) < 2e-6
    if (ih, iw) == (oh, ow):
        assert torch.equal(got.cpu(), x)


@pytest.mark.parametrize("shape", [(8, 16, 257, 8), (4, 10, 200, 9), (2, 3, 97, 5), (4, 80, 32, 40)])
def test_deferred_weight_gradient_reduction_equals_immediate(shape):
    """vts_wgrad_reduce_batch (both its 256-element form and the 64-element form for > 256 copies of a small dw, with 16-byte and
    ragged 4-byte rows) against the single-job reduction of the same partial copies, incl. a second accumulated contribution"""
    from vts import ops

    n, cl, lh, ch = shape
    dev = _dev()
    hh = (lh - 1) * 2 + 4 - 2
    lo = detrand.uniform((n, cl, lh, lh), 5, "lo").to(dev)
    hi = detrand.uniform((n, ch, hh, hh), 5, "hi").to(dev)
    lo2 = detrand.uniform((n, cl, lh, lh), 5, "lo2").to(dev)
    ref = torch.empty(cl, ch, 4, 4, device=dev)
    ops.wgrad4x4(lo, hi, ref, stride=2, pad=1, defer=False)
    ops.wgrad4x4(lo2, hi, ref, stride=2, pad=1, defer=False, accumulate=True)
    got = torch.full((cl, ch, 4, 4), 7.0, device=dev)
    with ops.deferred_wgrad():
        ops.wgrad4x4(lo, hi, got, stride=2, pad=1)
        ops.wgrad4x4(lo2, hi, got, stride=2, pad=1, accumulate=True)
    assert rel(got, ref) < 2e-6


@pytest.mark.parametrize("shape", [(2, 64, 35, 35, 2, True), (3, 20, 67, 40, 2, True), (2, 5, 16, 33, 1, False), (8, 64, 131, 131, 2, True)])
def test_single_channel_weight_gradient(shape):
    """the PatchGAN head's weight gradient (one low-resolution channel) on the vector-ALU member (wgrad_head_kernel) against
    torch.nn.grad.conv2d_weight, incl. normalise + LeakyReLU on the high-resolution operand and ragged tiles"""
    from vts import lib as L
    from vts import ops
    from vts.ops import Act

    n, ch, lh, lw, pad, affine = shape
    dev = _dev()
    hh, hw = lh - 1 + 4 - 2 * pad, lw - 1 + 4 - 2 * pad
    lo = detrand.uniform((n, 1, lh, lw), 4, "lo")
    hi = detrand.uniform((n, ch, hh, hw), 4, "hi")
    if affine:
        sc, sh = _affine(n, ch, 4, "h")
        hiv = F.leaky_relu(hi * sc.view(n, ch, 1, 1) + sh.view(n, ch, 1, 1), 0.2)
        hi_op, act = Act(hi.to(dev), sc.to(dev), sh.to(dev)), L.ACT_LRELU
    else:
        hiv, hi_op, act = hi, Act(hi.to(dev)), 0
    ref = torch.nn.grad.conv2d_weight(hiv.double(), (1, ch, 4, 4), lo.double(), stride=1, padding=pad).float()
    dw = torch.empty(1, ch, 4, 4, device=dev)
    ops.wgrad4x4(Act(lo.to(dev)), hi_op, dw, act_hi=act, stride=1, pad=pad, defer=False)
    assert L.load().vts_last_kernel().decode() == "wgrad_head_kernel"
    assert rel(dw, ref) < 3e-6


@pytest.mark.parametrize("n,h,w,with_I,with_M", [(2, 64, 64, True, True), (3, 33, 47, True, True), (1, 17, 19, False, True), (2, 40, 24, True, False)])
def test_input_images_from_bytes_equals_expand_then_mask(n, h, w, with_I, with_M):
    """vts_input_images_u8 (set_input's image part in one launch, dword-load and generic instances) is bit-identical to
    vts_u8_expand + vts_mask_mul, i.e. to Normalize(ToTensor(bytes)) * (mask bytes / 255)"""
    from vts import ops

    dev = _dev()
    g = torch.Generator().manual_seed(n * 1000 + h)
    S = torch.randint(0, 256, (n, 1, h, w), generator=g, dtype=torch.uint8)
    I = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8) if with_I else None
    M = (torch.randint(0, 2, (n, 1, h, w), generator=g, dtype=torch.uint8) * 255) if with_M else None
    if with_M:
        M[0, 0, 0, :5] = torch.tensor([0, 1, 127, 200, 255], dtype=torch.uint8)      # (non-binary mask bytes take the same path)
    Mo = torch.empty(n, 1, h, w, device=dev) if with_M else None
    S2 = torch.full((2 * n, 1, h, w), 9.0, device=dev)
    I2 = torch.full((2 * n, 3, h, w), 9.0, device=dev) if with_I else None
    ops.input_images_u8(S.to(dev), I.to(dev) if with_I else None, M.to(dev) if with_M else None, Mo, S2[:n], S2[n:], I2[n:] if with_I else None)
    m = M.float().div(255) if with_M else torch.ones(n, 1, h, w)
    s_ref = (S.float().div(255) - 0.5) / 0.5 * m
    assert torch.equal(S2[:n].cpu(), s_ref) and torch.equal(S2[n:].cpu(), s_ref)
    if with_M:
        assert torch.equal(Mo.cpu(), m)
        assert torch.equal(ops.mask_mul(ops.u8_expand(S.to(dev), True), Mo).cpu(), s_ref)
    if with_I:
        assert torch.equal(I2[n:].cpu(), (I.float().div(255) - 0.5) / 0.5 * m) and bool((I2[:n] == 9).all())


def test_patchsample_whole_map_matches_reference(golden_dir):
    """PatchSampleF(num_patches=0) on the HIP path against the reference module's outputs (tests/golden/patchsample_whole.npz): every
    position gathered, the MLP on the rows, the norm over the positions of each (image, channel), NCHW result, empty id lists"""
    import os

    from models.networks import PatchSampleF

    dev = _dev()
    g = np.load(os.path.join(golden_dir, "patchsample_whole.npz"))
    fd = [detrand.uniform((2, 6, 5, 7), 51, "f0").to(dev), detrand.uniform((3, 10, 4, 4), 51, "f1").to(dev)]
    plain, ids = PatchSampleF(use_mlp=False)(fd, 0, None)
    assert ids == [[], []] and plain[0].shape == (2, 6, 5, 7) and plain[1].shape == (3, 10, 4, 4)
    assert rel(plain[0], torch.from_numpy(g["plain0"])) < 1e-6 and rel(plain[1], torch.from_numpy(g["plain1"])) < 1e-6
    f = PatchSampleF(use_mlp=True, nc=12)
    f.create_mlp(fd)
    for i in range(2):
        m = getattr(f, "mlp_%d" % i)
        for idx, (kw, kb) in ((0, ("w0", "b0")), (2, ("w2", "b2"))):
            m[idx].weight.data.copy_(torch.from_numpy(g["mlp%d_%s" % (i, kw)]))
            m[idx].bias.data.copy_(torch.from_numpy(g["mlp%d_%s" % (i, kb)]))
    fm, _ = f(fd, 0, None)
    assert fm[0].shape == (2, 12, 5, 7) and rel(fm[0], torch.from_numpy(g["mlp0"])) < 1e-5 and rel(fm[1], torch.from_numpy(g["mlp1"])) < 1e-5
