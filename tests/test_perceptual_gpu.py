"""Perceptual terms on the HIP path: LPIPS-VGG16 (the reference's default lambda_G1_lpips / lambda_G2_lpips terms and the *_LPIPS
metrics) and the pix2pixHD VGG19 feature loss.

Checker = oracle/perceptual.py (restatement of lpips.LPIPS's published algorithm and of the reference's own Vgg19 / VGGLoss) with the
product's stand-in weights, plus the fixtures made by running the REFERENCE's call sites on that restatement
(tests/golden/sinskitG_lpips_step_256.npz, pix2pixHD_vgg_step_32.npz).  Tolerances: kernels 1e-5 rel-L2; network values 1e-4; input
gradients 1e-3 (fp32 through 13 convolutions); step losses 1e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch.utils.data import default_collate

pytestmark = pytest.mark.gpu

from oracle import detrand, nets, perceptual as chk, step  # noqa: E402  (checker only)


def _dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("shape", [(2, 5, 8, 12), (1, 64, 32, 32), (3, 7, 2, 2)])
def test_maxpool_relu_pad_and_adjoint(shape):
    from vts import ops

    n, c, h, w = shape
    z = detrand.uniform(shape, 3, "z").requires_grad_(True)
    z.data[0, 0, 0:2, 0:2] = 0.37          # a tie inside one window: the first element takes the gradient (PyTorch's rule)
    z.data[0, 1, 0:2, 0:2] = -0.5          # a window without a positive element: no gradient at all
    y = F.max_pool2d(F.relu(z), 2, 2)
    cot = detrand.uniform(tuple(y.shape), 3, "cot")
    (y * cot).sum().backward()
    zd = z.detach().to(_dev())
    for pad in (0, 1):
        out = ops.maxpool2_relu_pad(zd, pad)
        assert torch.equal(out.cpu(), F.pad(y.detach(), (pad,) * 4))
    gz = ops.maxpool2_relu_bwd(cot.to(_dev()), zd)
    # through relu as well: the product applies the ReLU mask in the next kernel (relu_mask_pad), PyTorch's z.grad has it already
    masked = ops.relu_mask_pad(gz, None, zd, pad=0)
    assert torch.equal(masked.cpu(), z.grad)


@pytest.mark.parametrize("shape,pad,zpad,tap", [((2, 5, 8, 12), 1, 1, True), ((1, 3, 2, 2), 2, 0, True), ((2, 4, 2, 10), 1, 2, False),
                                                ((2, 4, 10, 2), 2, 1, True), ((1, 64, 32, 32), 1, 1, True), ((3, 2, 6, 6), 0, 0, True),
                                                ((1, 2, 5, 7), 1, 0, True)])
def test_maxpool_adjoint_window_kernel_against_autograd(shape, pad, zpad, tap):
    """vts_maxpool2_relu_bwd (one thread per 2 x 2 window for even sizes, per element otherwise): routed gradient + tap gradient where
    z > 0, written into a zero-bordered map -- against autograd of max_pool2d(relu(z)) + <tap, relu(z)>; border zeros on a poisoned buffer"""
    from vts import lib as L
    from vts import ops

    n, c, h, w = shape
    dev = _dev()
    z = detrand.uniform(shape, 31, "z").requires_grad_(True)
    z.data[0, 0, 0:2, 0:2] = 0.41
    z.data[0, 1, 0:2, 0:2] = -0.3
    y = F.max_pool2d(F.relu(z), 2, 2)
    cot = detrand.uniform(tuple(y.shape), 31, "cot")
    t = detrand.uniform(shape, 31, "tap") if tap else None
    loss = (y * cot).sum() + ((F.relu(z) * t).sum() if tap else 0)
    loss.backward()
    zp = F.pad(z.detach(), (zpad,) * 4).to(dev)
    tp = F.pad(t, (zpad,) * 4).to(dev) if tap else None
    out = torch.full((n, c, h + 2 * pad, w + 2 * pad), 9.0, device=dev)
    lib = L.load()
    L.check(lib.vts_maxpool2_relu_bwd(cot.to(dev).data_ptr(), zp.data_ptr(), n * c, h, w, out.data_ptr(), zpad, L.ptr(tp), pad, L.stream()), "bwd")
    # (the kernel leaves the ReLU mask of the ROUTED part to the routing rule itself: a maximum is only routed where it is positive)
    assert torch.equal(out.cpu(), F.pad(z.grad, (pad,) * 4)), (out.cpu() - F.pad(z.grad, (pad,) * 4)).abs().max()


def test_relu_mask_pad_and_l1_relu():
    from vts import ops

    dev = _dev()
    shape = (2, 6, 9, 11)
    z, g, g2 = detrand.uniform(shape, 5, "z"), detrand.uniform(shape, 5, "g"), detrand.uniform(shape, 5, "g2")
    out = ops.relu_mask_pad(g.to(dev), g2.to(dev), z.to(dev), pad=1)
    assert torch.equal(out.cpu(), F.pad((g + g2) * (z > 0), (1, 1, 1, 1)))
    out = ops.relu_mask_pad(None, g2.to(dev), z.to(dev), pad=0)
    assert torch.equal(out.cpu(), g2 * (z > 0))
    za = detrand.uniform(shape, 6, "za").requires_grad_(True)
    zb = detrand.uniform(shape, 6, "zb")
    ra = F.relu(za)
    ra.retain_grad()
    loss = 0.7 * F.l1_loss(ra, F.relu(zb))
    loss.backward()
    slot = ops.loss_slots(1, dev)
    grad = torch.empty(shape, device=dev)
    ops.l1_relu(za.detach().to(dev), zb.to(dev), 0.7 / za.numel(), slot, grad=grad)
    assert abs(ops.loss_values(slot)[0] - float(loss)) < 1e-6
    assert rel(grad, ra.grad) < 1e-6


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (3, 512, 2, 2), (1, 128, 9, 7)])
def test_lpips_layer_value_and_gradient(shape):
    from vts import ops

    dev = _dev()
    n, c, h, w = shape
    z0 = detrand.uniform(shape, 7, "z0").requires_grad_(True)
    z1 = detrand.uniform(shape, 7, "z1")
    wl = torch.rand(c, generator=torch.Generator().manual_seed(1)) * 0.1
    f0 = F.relu(z0)
    f0.retain_grad()
    d = (chk.LPIPS.normalize_tensor(f0) - chk.LPIPS.normalize_tensor(F.relu(z1))) ** 2
    val = F.conv2d(d, wl.view(1, c, 1, 1)).mean([2, 3]).sum() * 0.3
    val.backward()
    slot = ops.loss_slots(1, dev)
    dz = torch.empty(shape, device=dev)
    ops.lpips_layer(z0.detach().to(dev), z1.to(dev), wl.to(dev), 0.3, slot, dz0=dz, grad_coeff=0.3)
    assert abs(ops.loss_values(slot)[0] - float(val)) <= 1e-5 * max(1e-3, abs(float(val)))
    assert rel(dz, z0.grad) < 1e-5                 # (the gradient w.r.t. z0 itself: the ReLU mask is applied by the kernel)
    # the padded layout of the VGG stacks (relu(z) inside a zero border, as ops.conv3x3_wide_relu_pad leaves it): same value, and the
    # gradient in the same layout
    p0, p1 = F.pad(F.relu(z0.detach()), (1,) * 4).to(dev), F.pad(F.relu(z1), (1,) * 4).to(dev)
    slot2 = ops.loss_slots(1, dev)
    dzp = ops.zero_border(torch.full(p0.shape, 7.0, device=dev))
    ops.lpips_layer(p0, p1, wl.to(dev), 0.3, slot2, dz0=dzp, grad_coeff=0.3, zpad=1)
    assert ops.loss_values(slot2)[0] == ops.loss_values(slot)[0]
    assert torch.equal(dzp, F.pad(dz, (1,) * 4))


@pytest.mark.parametrize("shape", [(2, 8, 40, 72), (4, 64, 65, 97), (4, 136, 64, 128)])
def test_padded_layout_convolution_epilogues(shape):
    """ops.conv3x3_wide_relu_pad / conv3x3_wide_mask_pad against the dense convolution followed by the separate pass they replace
    (bit-identical: same kernel, same accumulation order), and the readers of the padded layout against their dense forms"""
    from vts import ops

    dev = _dev()
    n, ci, h, w = shape
    co = 64 if ci != 64 else 128
    x = detrand.uniform(shape, 21, "x").to(dev)
    wt = (detrand.uniform((co, ci, 3, 3), 21, "w") * 0.2).to(dev)
    bias = detrand.uniform((co,), 21, "b").to(dev)
    p = ops.pad_affine(x, (1, 1, 1, 1), 0)
    packed = ops.w3x3_pack(wt, "conv_fwd", tag="t_epi%d" % ci)
    z = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_wide(p, packed, bias, z)
    out = torch.full((n, co, h + 2, w + 2), 3.0, device=dev)
    assert ops.conv3x3_wide_relu_pad(p, packed, bias, out)
    assert torch.equal(out, F.pad(F.relu(z), (1,) * 4))
    ref = F.conv2d(x.cpu().double(), wt.cpu().double(), bias.cpu().double(), padding=1)
    assert rel(z.cpu().double(), ref) < 1e-5
    # adjoint epilogue: (conv + add) where mask > 0
    mask = F.pad(F.relu(detrand.uniform((n, co, h, w), 22, "m")), (1,) * 4).to(dev)
    add = ops.zero_border(detrand.uniform((n, co, h + 2, w + 2), 22, "a").to(dev))
    z0 = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_wide(p, packed, None, z0)
    for a in (add, None):
        out = torch.full((n, co, h + 2, w + 2), 3.0, device=dev)
        assert ops.conv3x3_wide_mask_pad(p, packed, out, mask, add=a)
        want = z0 + a[:, :, 1:-1, 1:-1] if a is not None else z0
        assert torch.equal(out, F.pad(want * (mask[:, :, 1:-1, 1:-1] > 0), (1,) * 4))
    # readers: pooling, its adjoint (+ tap gradient, padded output), the ReLU mask
    zr = F.pad(F.relu(z), (1,) * 4)
    assert torch.equal(ops.maxpool2_relu_pad(zr, 1, zpad=1), ops.maxpool2_relu_pad(z, 1))
    g = detrand.uniform((n, co, h // 2, w // 2), 23, "g").to(dev)
    tap = detrand.uniform((n, co, h, w), 23, "t").to(dev)
    dense = ops.relu_mask_pad(ops.maxpool2_relu_bwd(g, z), tap, z, pad=1)
    assert torch.equal(ops.maxpool2_relu_bwd(g, zr, zpad=1, g2=F.pad(tap, (1,) * 4), pad=1), dense)
    gd = detrand.uniform((n, co, h, w), 24, "gd").to(dev)
    assert torch.equal(ops.relu_mask_pad(gd, F.pad(tap, (1,) * 4), zr, pad=1, zpad=1), ops.relu_mask_pad(gd, tap, z, pad=1))


@pytest.mark.parametrize("case", [(16, 64, 64, 64, 64), (9, 36, 128, 70, 90), (4, 128, 64, 130, 126), (256, 64, 128, 8, 8), (600, 64, 128, 6, 10),
                                  (1024, 64, 512, 2, 2), (1000, 40, 128, 5, 3)])
def test_winograd_convolution_against_float64_and_the_direct_kernel(case):
    """ops.conv3x3_wino (F(2x2, 3x3), fp32): against F.conv2d in float64 -- as close as the direct GEMM-class kernel --, odd map sizes and a
    channel count that is not a multiple of the 8-channel chunk; the padded-layout epilogues (ReLU / mask + tap gradient, zero border
    stored by the kernel) against the plain result; the input adjoint through the flipped / transposed transform"""
    from vts import ops

    dev = _dev()
    n, ci, co, h, w = case
    assert ops.conv3x3_wino_ok(n, ci, co, h, w)
    x = detrand.uniform((n, ci, h, w), 41, "x")
    wt = detrand.uniform((co, ci, 3, 3), 41, "w") * (1.0 / (ci * 9) ** 0.5)
    b = 0.1 * detrand.uniform((co,), 41, "b")
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    xd, wd, bd = x.to(dev), wt.to(dev), b.to(dev)
    p = ops.pad_affine(xd, (1, 1, 1, 1), 0)
    U = ops.w3x3_wino_pack(wd, "conv_fwd", tag="t_wino%d" % ci)
    y = torch.full((n, co, h, w), 7.0, device=dev)
    ops.conv3x3_wino(p, U, bd, y)
    z = torch.empty_like(y)
    ops.conv3x3_wide(p, ops.w3x3_pack(wd, "conv_fwd", tag="t_wino%d" % ci), bd, z)
    e_w, e_d = rel(y.cpu().double(), ref), rel(z.cpu().double(), ref)
    assert e_w < 2e-6 and e_w < 4 * e_d + 1e-7, (e_w, e_d)        # observed 3e-7 .. 1e-6 for both forms
    # padded layout: ReLU, and mask + tap gradient (no bias: the adjoint's form)
    yp = torch.full((n, co, h + 2, w + 2), 3.0, device=dev)
    ops.conv3x3_wino(p, U, bd, yp, ep_mode=1)
    assert torch.equal(yp, F.pad(torch.relu(y), (1,) * 4))
    y0 = torch.empty_like(y)
    ops.conv3x3_wino(p, U, None, y0)
    mask = F.pad(torch.relu(detrand.uniform((n, co, h, w), 42, "m")), (1,) * 4).to(dev)
    add = ops.zero_border(detrand.uniform((n, co, h + 2, w + 2), 42, "a").to(dev))
    for a in (add, None):
        out = torch.full((n, co, h + 2, w + 2), 3.0, device=dev)
        ops.conv3x3_wino(p, U, None, out, ep_mode=2, add=a, mask=mask)
        want = y0 + a[:, :, 1:-1, 1:-1] if a is not None else y0
        assert torch.equal(out, F.pad(want * (mask[:, :, 1:-1, 1:-1] > 0), (1,) * 4))
    # input adjoint: gradient of sum(conv(x) * g) w.r.t. x = conv of the padded g with the flipped, transposed weight
    if ops.conv3x3_wino_ok(n, co, ci, h, w):
        g = detrand.uniform((n, co, h, w), 43, "g")
        gref = F.conv_transpose2d(g.double(), wt.double(), padding=1)
        gin = torch.empty(n, ci, h, w, device=dev)
        ops.conv3x3_wino(ops.pad_affine(g.to(dev), (1, 1, 1, 1), 0), ops.w3x3_wino_pack(wd, "conv_adj", tag="t_wino%d" % ci), None, gin)
        assert rel(gin.cpu().double(), gref) < 2e-6


def test_winograd_declines_what_it_does_not_take():
    from vts import ops

    assert not ops.conv3x3_wino_ok(4, 3, 64, 1024, 1024)      # the 3-channel stem
    assert not ops.conv3x3_wino_ok(4, 64, 3, 1024, 1024)      # its adjoint (output channels not a multiple of 64)
    assert not ops.conv3x3_wino_ok(1, 64, 64, 32, 32)         # too few workgroups for the chip
    p = torch.zeros(1, 64, 34, 34, device=_dev())
    with pytest.raises(RuntimeError):
        ops.conv3x3_wino(p, torch.zeros(64 * 16 * 64, device=_dev()), None, torch.empty(1, 64, 32, 32, device=_dev()))


def test_small_maps_decline_the_padded_epilogue():
    from vts import ops

    dev = _dev()
    p = torch.zeros(256, 256, 10, 10, device=dev)
    packed = ops.w3x3_pack(torch.zeros(256, 256, 3, 3, device=dev), "conv_fwd", tag="t_small")
    assert not ops.conv3x3_wide_relu_pad(p, packed, None, torch.empty(256, 256, 10, 10, device=dev))


def test_padded_and_dense_vgg_schedules_agree_bit_for_bit(monkeypatch):
    """VTS_VGG_PADDED=0 (round 3: dense raw outputs + separate ReLU / padding passes) and the padded-layout schedule run the same
    arithmetic in the same order"""
    from vts import ops, perceptual as P

    dev = _dev()
    net, _ = _lpips_pair()
    a = detrand.uniform((2, 3, 64, 96), 31, "a").to(dev)
    b = detrand.uniform((2, 3, 64, 96), 31, "b").to(dev)
    res = {}
    for padded in (True, False):
        monkeypatch.setattr(P, "PADDED", padded)
        slot = ops.loss_slots(1, dev)
        grad = torch.empty_like(a)
        P.lpips_term(net, a, b, 1.0, slot, grad_into=grad)
        res[padded] = (ops.loss_values(slot)[0], grad)
    assert res[True][0] == res[False][0]
    assert torch.equal(res[True][1], res[False][1])


def _lpips_pair():
    from models import perceptual

    net = perceptual.LpipsVgg16().to(_dev())
    return net, chk.LPIPS(sd=net.own_state())


def test_lpips_network_matches_checker_and_reference_fixture(golden_dir):
    """values per sample and the input gradient: 3-channel images and broadcast 1-channel tactile patches; the fixture values come
    from the module the reference's own step ran on (oracle/make_golden.py lpips)"""
    from vts import ops, perceptual as P

    dev = _dev()
    g = np.load(golden_dir + "/sinskitG_lpips_step_256.npz")
    seed = int(g["seed"])
    net, lp = _lpips_pair()
    for shape, sc, ta, tb, key in (((2, 3, 64, 64), 1.0, "lp_a", "lp_b", "3"), ((5, 1, 32, 32), 0.3, "lp_a1", "lp_b1", "1")):
        a, b = sc * detrand.uniform(shape, seed, ta), sc * detrand.uniform(shape, seed, tb)
        ar = a.clone().requires_grad_(True)
        vref = lp(ar, b)
        vref.sum().backward()
        np.testing.assert_allclose(vref.detach().flatten().double().numpy(), g["module/val" + key], rtol=1e-5)
        ad, bd = a.to(dev), b.to(dev)
        grad = torch.zeros(shape, device=dev)
        per = []
        for i in range(shape[0]):            # per-sample values through the slot
            slot = ops.loss_slots(1, dev)
            P.lpips_term(net, ad[i:i + 1], bd[i:i + 1], 1.0, slot)
            per.append(ops.loss_values(slot)[0])
        np.testing.assert_allclose(per, g["module/val" + key], rtol=2e-4)
        slot = ops.loss_slots(1, dev)
        P.lpips_term(net, ad, bd, 1.0, slot, grad_into=grad)
        assert abs(ops.loss_values(slot)[0] - float(vref.sum())) <= 2e-4 * float(vref.sum())
        assert rel(grad, ar.grad) < 1e-3
        p = detrand.probe(grad.cpu(), "lp_ga" if key == "3" else "lp_ga1")
        rp = g["module/grad%s_probe" % key]
        assert abs(p[1] - rp[1]) <= 1e-3 * abs(rp[1])


def test_lpips_channel_view_of_a_patch_stack():
    """the tactile term passes 1-channel VIEWS of [P, 2, 32, 32] tensors and accumulates into 1-channel views of the gradient"""
    from vts import ops, perceptual as P

    dev = _dev()
    net, lp = _lpips_pair()
    fake = (0.3 * detrand.uniform((6, 2, 32, 32), 9, "f")).requires_grad_(True)
    real = 0.3 * detrand.uniform((6, 2, 32, 32), 9, "r")
    ref = chk.touch_lpips(lp, fake, real, 3, 10.0)
    ref.backward()
    fd, rd = fake.detach().to(dev), real.to(dev)
    grad = torch.full((6, 2, 32, 32), 0.25, device=dev)      # pre-filled: the term accumulates
    slot = ops.loss_slots(1, dev)
    for c in (0, 1):
        P.lpips_term(net, fd[:, c:c + 1], rd[:, c:c + 1], 10.0 / 2, slot, grad_into=grad[:, c:c + 1], grad_accumulate=True)
    assert abs(ops.loss_values(slot)[0] - float(ref)) <= 2e-4 * float(ref)
    assert rel(grad - 0.25, fake.grad) < 1e-3


def test_vgg19_feature_loss_matches_checker():
    from models import perceptual
    from vts import ops, perceptual as P

    dev = _dev()
    net = perceptual.Vgg19Features().to(dev)
    vl = chk.VGGLoss(chk.Vgg19(sd=net.own_state()))
    x = detrand.uniform((2, 3, 32, 32), 11, "x").requires_grad_(True)
    y = detrand.uniform((2, 3, 32, 32), 11, "y")
    ref = vl(x, y) * 10.0
    ref.backward()
    slot = ops.loss_slots(1, dev)
    gx = P.vgg_feature_l1(net, x.detach().to(dev), y.to(dev), 10.0, slot)
    assert abs(ops.loss_values(slot)[0] - float(ref)) <= 2e-4 * float(ref)
    assert rel(gx, x.grad) < 2e-3       # L1 subgradients: sign flips of near-zero feature differences


def _make_lpips_model(size):
    from models import create_model
    from options.train_options import TrainOptions

    flags = ("--model sinskitG --gpu_ids 0 --lambda_G1_lpips 1 --lambda_G2_lpips 10 --use_vision_aided_loss False --lambda_G2_GAN_feat 0 "
             "--checkpoints_dir /tmp/vts_test_ckpt --name tl --crop_size %d --batch_size 1" % size)
    opt = TrainOptions(cmd_line=flags).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    return model, opt


def test_step_with_lpips_terms_matches_reference_golden(golden_dir):
    """the HIP training step with the reference's default LPIPS lambdas against the REFERENCE's own step (run on the restated LPIPS
    module): all logged losses incl. G_lpips / G2_lpips, fake_I, every generator gradient"""
    from data.synthetic_dataset import make_sample
    from tests.test_step_gpu import load_test_weights, null_grad_bias, probe_close

    g = np.load(golden_dir + "/sinskitG_lpips_step_256.npz")
    size, seed, nt = int(g["size"]), int(g["seed"]), int(g["nt"])
    model, opt = _make_lpips_model(size)
    assert model.loss_lpips_pretrained is False and "G_lpips" in model.loss_names and "G2_lpips" in model.loss_names
    load_test_weights(model, seed)
    model._draws = {"aug": torch.from_numpy(g["s0/aug"]), "more_idx": torch.from_numpy(g["s0/more_idx"])}
    model.set_input(default_collate([make_sample(size, nt, nt, seed)]), phase="train")
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    losses = model.get_current_losses()
    ref = dict(zip([str(s) for s in g["s0/loss_names"]], g["s0/loss_values"]))
    for k in ("l_G_lpips", "l_G2_lpips", "l_G_L1", "l_G2_L1", "l_G_GAN", "l_D_real_I", "l_D_fake_I", "l_D_fake_T_concat", "l_D_real_T_concat"):
        assert abs(losses[k] - ref[k]) <= 1e-3 * max(1.0, abs(ref[k])), (k, losses[k], ref[k])
    probe_close(model.fake_I, g["s0/fake_I_probe"], "fake_I", 1e-3)
    for k, p in model.netG.named_parameters():
        if null_grad_bias("G", k):
            continue
        probe_close(p.grad, g["s0/grad_G/%s" % k], k, 2e-3)


def test_lpips_step_graph_replay_and_metrics():
    """three steps (eager, capture, replay) with the LPIPS terms inside the captured segments; then the *_LPIPS metrics of the
    validation patches against the checker"""
    import random

    from data.synthetic_dataset import make_sample
    from tests.test_step_gpu import load_test_weights
    from vts import ops

    model, opt = _make_lpips_model(256)
    load_test_weights(model, 31)
    random.seed(5)
    torch.manual_seed(5)
    batch = default_collate([make_sample(256, 64, 16, 31)])
    vals = []
    for _ in range(3):
        model.set_input(batch, phase="train")
        model.optimize_parameters(epoch=1)
        vals.append(model.get_current_losses()["l_G_lpips"])
    assert model._graphs is not None and all(np.isfinite(vals)) and vals[0] > 0
    model.eval()
    model.set_input(batch, phase="val")
    model.test()
    m = model.compute_metrics()
    assert m["I_LPIPS"] > 0 and m["T_LPIPS"] > 0 and model.metric_lpips_pretrained is False
    lp = chk.LPIPS(sd=model.netLPIPS.own_state())
    with torch.no_grad():
        ref_I = float(lp(model.real_I.cpu(), model.fake_I.cpu()).mean())
        pset = model.val_set
        P = pset["real_T"].shape[0]
        fake_T = torch.empty(P, 2, 32, 32, device=model.device)
        model._gather(model.fake_T, pset, fake_T, 0, channels=2)
        rT = F.interpolate(pset["real_T"].cpu(), (224, 224))
        fT = F.interpolate(fake_T.cpu().clamp(0, 1), (224, 224))
        ref_T = float(lp(rT[:, 0:1], fT[:, 0:1]).mean() + lp(rT[:, 1:2], fT[:, 1:2]).mean())     # compute_touch_lpips_loss, batch_size_G2 None
    assert abs(m["I_LPIPS"] - ref_I) <= 1e-3 * ref_I and abs(m["T_LPIPS"] - ref_T) <= 1e-3 * ref_T


def test_pix2pixHD_step_with_vgg_term_matches_reference_golden(golden_dir):
    from models import create_model
    from options.train_options import TrainOptions
    from tests.test_oracle_golden import p2p_batch
    from tests.test_step_gpu import probe_close

    g = np.load(golden_dir + "/pix2pixHD_vgg_step_32.npz")
    size, seed, n = int(g["size"]), int(g["seed"]), int(g["n"])
    flags = " ".join(str(f) for f in g["flags"]) + " --gpu_ids 0 --checkpoints_dir /tmp/vts_test_ckpt --name p2pv --dataset_mode patchskit"
    opt = TrainOptions(cmd_line=flags).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    shG = nets.resnet_param_shapes(1, 5, 8, 2, 3, norm="batch", down="stride", up="convT", conv_bias=True)
    model.netG.load_state_dict(detrand.test_weights(shG, seed))
    model.netD.load_state_dict(detrand.test_weights(nets.d_if_param_shapes(4, 8, 2), seed + 1))
    model.netD2.load_state_dict(detrand.test_weights(nets.d_if_param_shapes(3, 8, 2), seed + 2))
    model.set_input(p2p_batch(n, size, seed), phase="train")
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    losses = model.get_current_losses()
    ref = dict(zip([str(k) for k in g["s0/loss_names"]], g["s0/loss_values"]))
    for k in ("l_G_VGG", "l_G_VGG_I", "l_G_VGG_T", "l_G_GAN", "l_D_real", "l_D_fake"):
        assert abs(losses[k] - ref[k]) <= 1e-3 * max(1.0, abs(ref[k])), (k, losses[k], ref[k])
    np.testing.assert_allclose(model.fake_I.cpu().numpy(), g["s0/fake_I"], rtol=1e-3, atol=1e-4)
    for k, p in model.netG.named_parameters():
        rp = g["s0/grad_G/%s" % k]
        if k.endswith(".bias") and abs(rp[1]) < 1e-4:
            continue
        probe_close(p.grad, rp, k, 3e-3)


def test_lpips_on_a_full_patch_set_equals_its_halves():
    """256 patches x 512 channels exceeds the 16-bit grid dimension of the padding kernel (sample chunks): one call on the whole
    set = the sum of two calls on its halves, values and gradients (the size of the headline step's tactile term)"""
    from vts import ops, perceptual as P

    dev = _dev()
    net, _ = _lpips_pair()
    a, b = (0.3 * detrand.uniform((256, 1, 32, 32), 13, "a")).to(dev), (0.3 * detrand.uniform((256, 1, 32, 32), 13, "b")).to(dev)
    s_all, s_half = ops.loss_slots(1, dev), ops.loss_slots(1, dev)
    g_all, g_half = torch.empty_like(a), torch.empty_like(a)
    P.lpips_term(net, a, b, 1.0, s_all, grad_into=g_all)
    for h in (slice(0, 128), slice(128, 256)):
        P.lpips_term(net, a[h], b[h], 1.0, s_half, grad_into=g_half[h])
    va, vh = ops.loss_values(s_all)[0], ops.loss_values(s_half)[0]
    assert va > 0 and abs(va - vh) <= 1e-5 * va
    assert rel(g_all, g_half) < 3e-4      # other k-split plans at the two batch sizes (fp32 summation order through 13 layers + ReLU masks)


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 128), (8, 72, 64, 100, 90), (8, 128, 48, 96, 96)])
def test_conv3x3_wide_64_channel_tile(case):
    """layers of <= 64 output channels on >= 256 tiles take the 64-channel x 8-row tile (conv3x3_wide64_kernel): against F.conv2d,
    and bit-identical to the 128-channel tile's result (same per-output summation order)"""
    import os

    from vts import lib as L, ops

    dev = _dev()
    n, ci, co, h, w = case
    x = detrand.uniform((n, ci, h, w), 17, "x")
    wt = detrand.uniform((co, ci, 3, 3), 17, "w") * (1.0 / (ci * 9) ** 0.5)
    b = 0.1 * detrand.uniform((co,), 17, "b")
    ref = F.conv2d(x, wt, b, padding=1)
    p = F.pad(x, (1, 1, 1, 1)).to(dev)
    packed = ops.w3x3_pack(wt.to(dev), "conv_fwd", tag="t64")
    out = torch.empty(n, co, h, w, device=dev)
    ops.conv3x3_wide(p, packed, b.to(dev), out)
    assert L.load().vts_last_kernel().decode() == "conv3x3_wide64_kernel"
    assert rel(out, ref) < 1e-5
    # the same layer on the generic wide kernel: a dispatch switch, i.e. only in the instrumented library (make PROFILING=1, csrc/vts_internal.h:vts_tune)
    if os.environ.get("VTS_LIB_PATH", "").endswith("_prof.so"):
        os.environ["VTS_NO_WIDE64"] = "1"
        try:
            out2 = torch.empty_like(out)
            ops.conv3x3_wide(p, packed, b.to(dev), out2)
            assert L.load().vts_last_kernel().decode().startswith("conv3x3_wide_kernel")
        finally:
            del os.environ["VTS_NO_WIDE64"]
        assert torch.equal(out, out2)


def test_lpips_alex_network_and_test_phase_metrics_match_reference_fixture(golden_dir):
    """lpips.LPIPS(net="alex"), the reference's test-phase eval_LPIPS (models/sinskitG_model.py:501), on the HIP kernels (11 x 11 stride-4
    stem as space-to-depth + GEMM-class 3 x 3, MaxPool2d(3, 2), 5 x 5 on 4 x 4 tap blocks): module values and I_LPIPS / T_LPIPS against
    tests/golden/lpips_metrics.npz -- the reference's compute_evaluation_metric run on the restated module -- and the VGG backbone's
    values of the same fixture through the training-phase path."""
    from models import perceptual as MP
    from oracle.make_golden import lpips_metric_inputs
    from vts import ops, perceptual as P

    dev = _dev()
    g = np.load(golden_dir + "/lpips_metrics.npz")
    seed = int(g["seed"])
    alex = MP.build_lpips_alex(None, dev)
    sd = chk.standin_state_alex(MP.LpipsAlex.SEED)
    for k in range(5):      # product and checker draw the same stand-in numbers
        assert torch.equal(alex.convs[k].weight.cpu(), sd["conv%d.weight" % k]) and torch.equal(alex.lins[k].cpu(), sd["lin%d.weight" % k])
    a, b = detrand.uniform((2, 3, 80, 96), seed, "lp_a").to(dev), detrand.uniform((2, 3, 80, 96), seed, "lp_b").to(dev)
    per = []
    for i in range(2):
        slot = ops.loss_slots(1, dev)
        P.lpips_alex_value(alex, a[i:i + 1], b[i:i + 1], 1.0, slot)
        per.append(ops.loss_values(slot)[0])
    np.testing.assert_allclose(per, g["alex/module_val"], rtol=2e-4)
    # feature maps against the checker, layer by layer
    lp = chk.LPIPS(net="alex")
    x = detrand.uniform((1, 3, 75, 91), seed, "alex_x")
    with torch.no_grad():
        ref = lp.net(x)
    zs = P.alex_forward(alex, x.to(dev))
    for k in range(5):
        assert rel(torch.relu(zs[k]), ref[k]) < 2e-5, k
    # the metric glue of both phases
    real_I, fake_I, real_T, fake_T = (t.to(dev) for t in lpips_metric_inputs(seed))
    vgg = MP.build_lpips(None, dev)
    for name, term in (("alex", lambda x, y, c, s: P.lpips_alex_value(alex, x, y, c, s)), ("vgg", lambda x, y, c, s: P.lpips_term(vgg, x, y, c, s))):
        buf = ops.loss_slots(2, dev)
        term(real_I, fake_I, 1.0 / real_I.shape[0], buf[0:1])
        for c in (0, 1):
            ra = ops.sifid_input(real_T, c, 1, size=(224, 224))
            fb = ops.sifid_input(fake_T, c, 1, size=(224, 224), clamp01=True)
            term(ra, fb, 1.0 / real_T.shape[0], buf[1:2])
        lv = ops.loss_values(buf)
        assert abs(lv[0] - float(g[name + "/I_LPIPS"])) <= 3e-4 * lv[0], (name, lv, float(g[name + "/I_LPIPS"]))
        assert abs(lv[1] - float(g[name + "/T_LPIPS"])) <= 3e-4 * lv[1], (name, lv, float(g[name + "/T_LPIPS"]))
