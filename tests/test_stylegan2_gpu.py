"""StyleGAN2 blocks on the HIP path: operators against plain PyTorch fp32 CPU evaluations and the committed
reference vectors (tests/golden/stylegan2_32.npz: the reference's stylegan_networks.py run on CPU), the discriminator forward /
backward against the reference output and oracle autograd.  Tolerances: rel-L2 1e-5 per operator, 2e-4 for network gradients."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import detrand, stylegan2 as sg  # noqa: E402  (checker only)


def _dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "stylegan2_32.npz"), allow_pickle=False)


def test_upfirdn2d_matches_reference_and_adjoint(gold):
    from vts import ops
    from models.stylegan2_blocks import make_kernel
    dev = _dev()
    seed = int(gold["seed"])
    u = detrand.uniform((2, 3, 9, 11), seed, "ufd_in")
    for i, (up, down, pad) in enumerate(sg.UPFIRDN_CASES):
        k = make_kernel() * (up ** 2)
        kl = [[float(v) for v in row] for row in k]
        out = ops.upfirdn2d(u.to(dev), kl, up, down, pad)
        assert out.shape == gold["ufd/%d" % i].shape, i
        assert rel(out, torch.from_numpy(gold["ufd/%d" % i])) < 1e-6, i
        ur = u.clone().requires_grad_(True)
        ref = sg.upfirdn2d(ur, k, up, down, pad)
        cot = detrand.uniform(tuple(ref.shape), seed, "ufd_cot%d" % i)
        (ref * cot).sum().backward()
        din = torch.full(u.shape, float("nan"), device=dev)
        ops.upfirdn2d_bwd(cot.to(dev), din, kl, up, down, pad)
        assert rel(din, ur.grad) < 1e-6, i
        ops.upfirdn2d_bwd(cot.to(dev), din, kl, up, down, pad, accumulate=True)
        assert rel(din, 2 * ur.grad) < 1e-6, i
    # a non-square, asymmetric kernel
    k = torch.tensor([[1.0, 2.0, -1.0], [0.5, 0.0, 3.0]])
    x = detrand.uniform((1, 2, 7, 8), 3, "x")
    out = ops.upfirdn2d(x.to(dev), [[float(v) for v in r] for r in k], 2, 1, (1, 2))
    assert rel(out, sg.upfirdn2d(x, k, 2, 1, (1, 2))) < 1e-6


def test_bias_act_forward_backward(gold):
    from vts import ops
    dev = _dev()
    seed = int(gold["seed"])
    u = detrand.uniform((2, 3, 9, 11), seed, "ufd_in")
    b = detrand.uniform((1, 3, 1, 1), seed, "flb")
    out = ops.bias_act(u.to(dev), b.to(dev))
    assert rel(out, torch.from_numpy(gold["flrelu"])) < 1e-6
    res = detrand.uniform(tuple(u.shape), seed, "res")
    xr, br = u.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.leaky_relu(xr + br, 0.2) * 0.7 + res
    out = ops.bias_act(u.to(dev), b.to(dev), 0.2, 0.7, res=res.to(dev))
    assert rel(out, ref) < 1e-6
    cot = detrand.uniform(tuple(u.shape), seed, "cot")
    (ref * cot).sum().backward()
    dx = ops.bias_act_bwd(cot.to(dev), u.to(dev), b.to(dev), 0.2, 0.7)
    assert rel(dx, xr.grad) < 1e-6
    db = torch.zeros(3, device=dev)
    ops.channel_sum(dx, db)
    assert rel(db, br.grad.view(-1)) < 1e-5
    assert rel(ops.bias_act(u.to(dev), None, 0.2, 2 ** 0.5), F.leaky_relu(u, 0.2) * 2 ** 0.5) < 1e-6     # ScaledLeakyReLU


@pytest.mark.parametrize("K,shape", [(3, (2, 20, 36, 33, 33)), (3, (1, 8, 12, 16, 18)), (1, (2, 24, 40, 31, 31)), (1, (1, 6, 10, 12, 14)),
                                     (3, (3, 96, 130, 9, 9))])
def test_stride2_small_kernel_conv_family(K, shape):
    """Conv2d(K, stride 2, padding 0) for K in {1, 3} on the stride-2 4x4 kernels: forward, input adjoint, weight gradient"""
    from vts import ops
    from vts.ops import Act
    n, ci, co, h, w = shape
    dev = _dev()
    x = detrand.uniform((n, ci, h, w), 51, "x").requires_grad_(True)
    wt = (detrand.uniform((co, ci, K, K), 51, "w") * 0.3).requires_grad_(True)
    b = detrand.uniform((co,), 51, "b")
    ref = F.conv2d(x * 0.5, wt, b, stride=2)
    cot = detrand.uniform(tuple(ref.shape), 51, "cot")
    (ref * cot).sum().backward()
    xd, wd, cd = x.detach().to(dev), wt.detach().to(dev), cot.to(dev)
    half = Act(xd, torch.full((n * ci,), 0.5, device=dev), torch.zeros(n * ci, device=dev))
    out = torch.full(ref.shape, float("nan"), device=dev)
    ops.convk_s2(half, wd, out, bias=b.to(dev))
    assert rel(out, ref) < 1e-5
    dx = torch.full(x.shape, float("nan"), device=dev)
    ops.convk_s2_bwd_data(Act(cd, torch.full((n * co,), 0.5, device=dev), torch.zeros(n * co, device=dev)), wd, dx)
    assert rel(dx, x.grad) < 1e-5
    dw = torch.full(wt.shape, float("nan"), device=dev)
    ops.wgradk_s2(cd, half, dw)
    assert rel(dw, wt.grad) < 2e-5
    ops.wgradk_s2(cd, half, dw, accumulate=True)
    assert rel(dw, 2 * wt.grad) < 2e-5


def _build_D(gold, dev):
    from models.stylegan2_blocks import StyleGAN2Discriminator
    from vts.optim import FlatParams
    size, seed, ndf, cin = (int(gold[k]) for k in ("size", "seed", "ndf", "input_nc"))
    D = StyleGAN2Discriminator(cin, ndf, size).to(dev)
    shapes = sg.d_param_shapes(cin, ndf, size)
    assert {k: tuple(v.shape) for k, v in D.named_parameters()} == {k: tuple(v) for k, v in shapes.items()}
    assert sorted(D.state_dict().keys()) == sorted(gold["ref_keys"].tolist())          # checkpoint compatible with the reference
    sd = sg.test_weights(shapes, seed)
    D.load_state_dict(sd, strict=False)
    return D, FlatParams(D), sd


def test_stylegan2_discriminator_matches_reference_and_oracle(gold):
    from vts import engine
    dev = _dev()
    size, seed, cin, n = (int(gold[k]) for k in ("size", "seed", "input_nc", "n"))
    D, flat, sd = _build_D(gold, dev)
    x = detrand.uniform((n, cin, size, size), seed, "d_in")
    y, ctx = engine.sg2d_forward(D, x.to(dev))
    assert rel(y, torch.from_numpy(gold["D_out"])) < 2e-5
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    yo = sg.discriminator_forward(sdo, xo, size)
    cot = detrand.uniform(tuple(yo.shape), seed, "d_cot")
    (yo * cot).sum().backward()
    flat.grad.zero_()
    dx = engine.sg2d_backward(D, ctx, cot.to(dev), input_grad=True)
    assert rel(dx, xo.grad) < 2e-4
    assert rel(dx[:, :, ::4, ::4], torch.from_numpy(gold["D_dx_sub"])) < 2e-4           # the reference's own input gradient
    named = dict(D.named_parameters())
    for k, v in sdo.items():
        assert rel(named[k].grad, v.grad) < 2e-4, (k, rel(named[k].grad, v.grad))
        p, rp = detrand.probe(named[k].grad.cpu(), k), gold["D_grad/" + k]
        assert abs(p[1] - rp[1]) <= 5e-4 * max(abs(rp[1]), 1e-12), (k, p, rp)
    # accumulate: a second backward doubles the gradients
    engine.sg2d_backward(D, ctx, cot.to(dev), accumulate=True)
    for k, v in sdo.items():
        assert rel(named[k].grad, 2 * v.grad) < 2e-4, k


def test_modulated_conv2d_forward_matches_reference(gold):
    from vts import engine
    dev = _dev()
    seed = int(gold["seed"])
    shapes = {"weight": (1, 20, 12, 3, 3), "modulation.weight": (12, 16), "modulation.bias": (12,)}
    w = {k: v.to(dev) for k, v in sg.test_weights(shapes, seed + 1).items()}
    xi = detrand.uniform((2, 12, 10, 10), seed, "mod_in").to(dev)
    st = detrand.uniform((2, 16), seed, "mod_style").to(dev)
    for tag, kw in (("plain", {}), ("nodemod", {"demodulate": False}), ("down", {"downsample": True})):
        out = engine.modulated_conv2d(xi, st, w["weight"], w["modulation.weight"], w["modulation.bias"], **kw)
        assert rel(out, torch.from_numpy(gold["mod/%s/out" % tag])) < 2e-5, tag


def test_train_step_with_stylegan2_discriminator_matches_oracle():
    """sinskitG step through create_model with --netD stylegan2 vs the CPU oracle step with the same discriminator (the oracle's
    blocks are pinned to the reference by stylegan2_32.npz; the step-level composition follows sinskitG_model.py compute_D1_loss /
    compute_G1_loss with networks.define_D's stylegan2 branch)"""
    import random

    from torch.utils.data import default_collate

    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions
    from oracle import nets, step

    size, nt, seed, n = 256, 64, 61, 2      # the U-Net generator (8 downsamplings) needs >= 256
    flags = ("--model sinskitG --gpu_ids 0 --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False "
             "--lambda_G2_GAN_feat 0 --checkpoints_dir /tmp/vts_test_ckpt --name tsg --crop_size %d --load_size %d --batch_size %d "
             "--netD stylegan2" % (size, size, n))
    opt = TrainOptions(cmd_line=flags).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    assert getattr(model.netD, "is_stylegan2_d", False) and model.netD.size == size
    sdG = detrand.test_weights(nets.g_param_shapes(), seed)
    sdD = sg.test_weights(sg.d_param_shapes(4, opt.ndf, size), seed + 1)
    sdD2 = detrand.test_weights(nets.d_param_shapes(7), seed + 2)
    model.netG.load_state_dict(sdG)
    model.netD.load_state_dict(sdD, strict=False)
    model.netD2.load_state_dict(sdD2)
    batch = default_collate([make_sample(size, nt, nt, seed + i) for i in range(n)])
    random.seed(5)
    counts = [int(nets.dilated_mask_positions(batch["M"][i:i + 1].float()).shape[0]) for i in range(n)]
    draws = {"aug": detrand.uniform((4, n), 3, "aug") * 0.5 + 0.5, "more_idx": torch.tensor([random.sample(range(c), 32) for c in counts])}
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    ref = step.train_step(sdG, sdD, sdD2, adam, batch, draws, opt=step.hp(netD="stylegan2"))
    model._draws = draws
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3
    for k, p in model.netD.named_parameters():
        assert rel(p.grad, ref["grad_D"][k]) < 2e-3, k
    for k, p in model.netG.named_parameters():        # the generator's gradient came through the StyleGAN2 discriminator
        if k.endswith("bias") and not any(k.startswith(q) for q in ("down0.", "down7.", "up0.", "up0_T.")):
            continue
        assert rel(p.grad, ref["grad_G"][k]) < 2e-3, k
    # keeps training, also captured as HIP graphs
    model._draws = None
    for _ in range(3):
        model.set_input(batch, phase="train")
        model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in model.get_current_losses().values())


def test_train_step_with_stylegan2_discriminator_matches_reference_golden(golden_dir):
    """the HIP step with --netD stylegan2 against the REFERENCE's own SinSKITGModel.optimize_parameters run with that flag
    (tests/golden/sinskitG_sg2d_step_256.npz): losses, outputs, gradient probes of all three networks"""
    from torch.utils.data import default_collate

    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions
    from oracle import nets

    g = np.load(os.path.join(golden_dir, "sinskitG_sg2d_step_256.npz"), allow_pickle=False)
    size, seed, nt = int(g["size"]), int(g["seed"]), int(g["nt"])
    flags = ("--model sinskitG --gpu_ids 0 --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False "
             "--lambda_G2_GAN_feat 0 --checkpoints_dir /tmp/vts_test_ckpt --name tsg2 --crop_size %d --load_size %d --batch_size 1 "
             "--netD stylegan2" % (size, size))
    opt = TrainOptions(cmd_line=flags).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    model.netG.load_state_dict(detrand.test_weights(nets.g_param_shapes(), seed))
    model.netD.load_state_dict(sg.test_weights(sg.d_param_shapes(4, opt.ndf, size), seed + 1), strict=False)
    model.netD2.load_state_dict(detrand.test_weights(nets.d_param_shapes(7), seed + 2))
    batch = default_collate([make_sample(size, nt, nt, seed)])
    model._draws = {"aug": torch.from_numpy(g["aug"]), "more_idx": torch.from_numpy(g["more_idx"])}
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    ref = dict(zip([str(s) for s in g["loss_names"]], g["loss_values"]))
    for k, v in ref.items():
        assert abs(losses[k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses[k], v)
    assert rel(model.fake_I[:, :, ::4, ::4], torch.from_numpy(g["fake_I_sub"])) < 1e-3
    for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
        for k, p in net.named_parameters():
            if k.endswith("bias") and ((nm == "G" and not any(k.startswith(q) for q in ("down0.", "down7.", "up0.", "up0_T.")))
                                       or (nm == "D2" and k.split(".")[1] in ("2", "5", "8"))):
                continue                                   # bias in front of a norm: analytically zero gradient
            pr, rp = detrand.probe(p.grad.cpu(), k), g["grad_%s/%s" % (nm, k)]
            scale = max(abs(rp[1]), 1e-12)
            assert abs(pr[1] - rp[1]) <= 2e-3 * scale and abs(pr[2] - rp[2]) <= 8e-3 * scale, (nm, k, pr, rp)


def test_stylegan2_generator_matches_reference_and_oracle(golden_dir):
    """`--netG smallstylegan2` (StyleGAN2Encoder + StyleGAN2Decoder: ConvLayer, ResBlocks with and without downsampling, StyledConv with
    the upsampling ModulatedConv2d and its demodulation backward) on the HIP engine: output, input gradient and every parameter gradient
    vs the REFERENCE module run on CPU (tests/golden/stylegan2_g_32.npz) and vs the oracle; then the noise-injecting variant (`stylegan2`)
    against the oracle with the noise draws passed in."""
    import os

    import numpy as np

    from models.stylegan2_blocks import StyleGAN2Generator
    from oracle.make_golden import SG2G_CFG as cfg
    from vts import engine
    from vts.optim import FlatParams
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "stylegan2_g_32.npz"))
    seed, cin, n = int(g["seed"]), int(g["input_nc"]), int(g["n"])
    shapes = sg.g_param_shapes(cin, **cfg)
    for noise in (False, True):
        G = StyleGAN2Generator(cin, 3, ngf=cfg["ngf"], n_blocks=cfg["n_blocks"], size=cfg["size"], num_downsampling=cfg["num_downsampling"],
                               inject_noise=noise).to(dev)
        assert {k: tuple(v.shape) for k, v in G.named_parameters()} == shapes
        assert sorted(G.state_dict().keys()) == sorted(g["ref_keys"].tolist())          # checkpoint compatible with the reference
        flat = FlatParams(G)
        sd = sg.test_weights(shapes, seed)
        G.load_state_dict(sd, strict=False)
        x = detrand.uniform((n, cin, cfg["size"], cfg["size"]), seed, "g_in")
        noises = None
        if noise:
            noises = [detrand.uniform((n, 1, cfg["size"] >> (cfg["num_downsampling"] - 1 - i), cfg["size"] >> (cfg["num_downsampling"] - 1 - i)), seed, "nz%d" % i)
                      for i in range(cfg["num_downsampling"])]
        y, ctx = engine.sg2g_forward(G, x.to(dev), noises=None if noises is None else [t.to(dev) for t in noises])
        sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        yo = sg.generator_forward(sdo, xo, noises=noises, **cfg)
        assert rel(y, yo) < 2e-5
        cot = detrand.uniform(tuple(yo.shape), seed, "g_cot")
        (yo * cot).sum().backward()
        flat.grad.zero_()
        dx = engine.sg2g_backward(G, ctx, cot.to(dev), input_grad=True)
        assert rel(dx, xo.grad) < 2e-4
        named = dict(G.named_parameters())
        for k, v in sdo.items():
            if k.endswith("noise.weight") and not noise:
                continue
            assert rel(named[k].grad, v.grad) < 3e-4, (k, rel(named[k].grad, v.grad))
        if not noise:      # the reference's own numbers
            assert rel(y, torch.from_numpy(g["G_out"])) < 2e-5
            assert rel(dx[:, :, ::4, ::4], torch.from_numpy(g["G_dx_sub"])) < 2e-4
            for k in sdo:
                if "G_grad/" + k in g.files and not k.endswith("noise.weight"):
                    p, rp = detrand.probe(named[k].grad.cpu(), k), g["G_grad/" + k]
                    assert abs(p[1] - rp[1]) <= 5e-4 * max(abs(rp[1]), 1e-12), (k, p, rp)
            engine.sg2g_backward(G, ctx, cot.to(dev), accumulate=True)     # a second backward doubles the gradients
            for k, v in sdo.items():
                if not k.endswith("noise.weight"):
                    assert rel(named[k].grad, 2 * v.grad) < 3e-4, k
