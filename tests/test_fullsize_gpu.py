"""BASELINE-size checks (1024 x 1024): one full oracle comparison at batch 1, and size-independent properties of the
HIP path at the benchmark configuration (skitG, 4 images): adjointness of the convolution family (forward vs
backward-data vs weight gradient must describe the same bilinear map), per-sample independence of the generator,
run-to-run bitwise determinism of a whole training step."""
import random

import numpy as np
import pytest
import torch
from torch.utils.data import default_collate

pytestmark = pytest.mark.gpu

# gradient comparisons against the oracle: every (relative L2 error, network, parameter) is recorded, the worst one is printed at the
# end of the module (pytest -s) and the bound is ~2x the worst value observed on the MI355X (round 4: see the fixture below)
GRAD_WORST = []
# north_star's bar: 1e-3 relative L2.  It holds for every parameter TENSOR.  The one-element gradients -- the biases of the three prediction
# heads, each a plain sum over N x H x W signed values -- get 1.5e-3: their "relative L2" is the relative error of ONE cancelling sum, and
# the fp32 CPU oracle's own value of it moves by several 1e-4 with its thread count (bench.py / __graft_entry__.smoke pin it to <= 16
# threads for that reason); observed worst on the MI355X 9.6e-4 (D layer2.11.bias, 4 x 35 x 35 terms), worst multi-element tensor 6e-4.
GRAD_TOL = 1.0e-3
GRAD_TOL_SCALAR = 1.5e-3


def grad_tol(t):
    return GRAD_TOL_SCALAR if t.numel() == 1 else GRAD_TOL


@pytest.fixture(scope="module", autouse=True)
def _report_worst_gradient():
    yield
    if GRAD_WORST:
        top = sorted(GRAD_WORST, reverse=True)[:6]
        print("\n[%s] gradient rel-L2 vs the oracle over %d comparisons, bound %.1e (one-element tensors %.1e); worst: %s"
              % (__name__, len(GRAD_WORST), GRAD_TOL, GRAD_TOL_SCALAR, ", ".join("%.2e %s %s" % w for w in top)))

from oracle import detrand, nets, step  # noqa: E402  (checker only)

SIZE = 1024
FLAGS = ("--model %s --gpu_ids 0 --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False "
         "--lambda_G2_GAN_feat 0 --checkpoints_dir /tmp/vts_test_ckpt --name full --crop_size %d --batch_size %d")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def make_model(model_name, n, extra=""):
    from models import create_model
    from options.train_options import TrainOptions
    opt = TrainOptions(cmd_line=FLAGS % (model_name, SIZE, n) + extra).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    return model, opt


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("case", [
    # (N, C0, C1, Cout, H, W, stride, pad): the outer layers of G / D1 at the benchmark size
    (4, 1, 8, 10, 1024, 1024, 2, 1), (4, 10, 0, 20, 512, 512, 2, 1), (4, 1, 3, 8, 1024, 1024, 2, 2), (4, 32, 0, 64, 129, 129, 1, 2)])
def test_conv_family_is_one_bilinear_map_at_full_size(case):
    """<conv(x; w), y> == <x, conv_bwd_data(y; w)> == <w, wgrad(x, y)>  (no oracle involved)"""
    from vts import ops
    n, c0, c1, co, h, w, s, p = case
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    x0 = torch.randn(n, c0, h, w, generator=g).to(dev)
    x1 = torch.randn(n, c1, h, w, generator=g).to(dev) if c1 else None
    wt = (torch.randn(co, c0 + c1, 4, 4, generator=g) * 0.1).to(dev)
    oh, ow = (h + 2 * p - 4) // s + 1, (w + 2 * p - 4) // s + 1
    y = torch.randn(n, co, oh, ow, generator=g).to(dev)
    out = torch.empty(n, co, oh, ow, device=dev)
    ops.conv4x4(x0, wt, (c0 + c1) * 16, 16, co, out, in1=x1, stride=s, pad=p)
    lhs = dot(out, y)
    terms = float((out.double().abs() * y.double().abs()).sum())   # fp32 kernels: errors scale with the summed magnitudes
    dx0 = torch.empty_like(x0)
    ops.conv4x4(y, wt, 16, (c0 + c1) * 16, c0, dx0, stride=s, pad=p, transposed=True)
    rhs = dot(dx0, x0)
    if c1:
        dx1 = torch.empty_like(x1)
        ops.conv4x4(y, wt.view(-1)[c0 * 16:], 16, (c0 + c1) * 16, c1, dx1, stride=s, pad=p, transposed=True)
        rhs += dot(dx1, x1)
    dw = torch.empty_like(wt)
    ops.wgrad4x4(y, x0, dw, hi1=x1, stride=s, pad=p)
    third = dot(dw, wt)
    assert abs(lhs - rhs) <= 1e-6 * terms and abs(lhs - third) <= 1e-6 * terms, (lhs, rhs, third, terms)


def test_generator_is_per_sample_independent_at_full_size():
    """InstanceNorm generator: G(batch)[i] == G(sample i), at 1024 x 1024 with 4 images"""
    from data.synthetic_dataset import make_sample
    model, opt = make_model("skitG", 4)
    batch = default_collate([make_sample(SIZE, 64, 64, 300 + i, style_dim=opt.style_code_dim) for i in range(4)])
    model.set_input(batch, phase="train")
    model.test()
    full_I, full_T = model.fake_I.clone(), model.fake_T.clone()
    one = default_collate([make_sample(SIZE, 64, 64, 302, style_dim=opt.style_code_dim)])
    model.set_input(one, phase="train")
    model.test()
    # (batch 1 and batch 4 take different tile / k-split plans for the inner layers: equal up to fp32 summation order)
    assert rel(model.fake_I, full_I[2:3]) < 1e-5 and rel(model.fake_T, full_T[2:3]) < 1e-5


def test_step_is_bitwise_deterministic_at_full_size():
    """two models from the same seed, the same 4-image batch: identical losses / weights after 3 steps (eager + graphs)"""
    from data.synthetic_dataset import make_sample
    res = []
    for _ in range(2):
        torch.manual_seed(5)
        random.seed(5)
        model, opt = make_model("skitG", 4)
        sd = detrand.test_weights(nets.g_param_shapes(style_nc=opt.style_code_dim), 9)
        model.netG.load_state_dict(sd)
        model.netD.load_state_dict(detrand.test_weights(nets.d_param_shapes(4), 10))
        model.netD2.load_state_dict(detrand.test_weights(nets.d_param_shapes(7), 11))
        batch = default_collate([make_sample(SIZE, 64, 64, 400 + i, style_dim=opt.style_code_dim) for i in range(4)])
        model.set_input(batch, phase="train")
        random.seed(77)
        for _ in range(3):
            model.optimize_parameters(epoch=1)
        torch.cuda.synchronize()
        res.append((model.get_current_losses(), model.flatG.flat.clone(), model.flatD.flat.clone(), model.flatD2.flat.clone()))
        del model
    assert all(np.isfinite(v) for v in res[0][0].values())
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)                     # weights: bitwise
    for k, v in res[0][0].items():                   # logged losses: 64-bit fixed-point accumulation, order independent
        assert v == res[1][0][k], k


def test_full_size_step_matches_oracle_batch1():
    """sinskitG, 1024 x 1024, batch 1: losses, outputs and gradients vs the CPU oracle (one step)"""
    from data.synthetic_dataset import make_sample
    model, opt = make_model("sinskitG", 1)
    seed = 123
    sds = (detrand.test_weights(nets.g_param_shapes(), seed), detrand.test_weights(nets.d_param_shapes(4), seed + 1),
           detrand.test_weights(nets.d_param_shapes(7), seed + 2))
    for net, sd in zip((model.netG, model.netD, model.netD2), sds):
        net.load_state_dict(sd)
    batch = default_collate([make_sample(SIZE, 64, 64, seed)])
    random.seed(5)
    cnt = int(nets.dilated_mask_positions(batch["M"].float()).shape[0])
    draws = {"aug": detrand.uniform((4, 1), 3, "aug") * 0.5 + 0.5, "more_idx": torch.tensor([random.sample(range(cnt), 32)])}
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    ref = step.train_step(sds[0], sds[1], sds[2], adam, batch, draws)
    model._draws = draws
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3 and rel(model.fake_T, ref["fake_T"]) < 1e-3
    for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
        for k, p in net.named_parameters():
            r = ref["grad_" + nm][k]
            if k.endswith("bias") and ((nm == "G" and not k.startswith(("down0.", "down7.", "up0.", "up0_T."))) or
                                       (nm != "G" and k.split(".")[1] in ("2", "5", "8"))):
                continue    # bias in front of a normalisation: mathematically zero gradient (rounding noise in autograd)
            GRAD_WORST.append((rel(p.grad, r), nm, k))
            assert GRAD_WORST[-1][0] < grad_tol(r), GRAD_WORST[-1]


def test_headline_config_step_matches_oracle():
    """BASELINE config 1 itself -- skitG (512-d style code tiled into the innermost up block), 4 images of 1024 x 1024, 64 tactile
    patches each -- one whole training step against the CPU oracle: all logged losses, both outputs, the gradient of every parameter
    of G / D / D2 at its backward point, BatchNorm running statistics (the batched D passes must advance them in the reference's
    order).  The oracle needs ~10 s for this step."""
    from data.synthetic_dataset import make_sample
    model, opt = make_model("skitG", 4)
    assert opt.use_style_code and opt.style_code_dim == 512
    seed, n = 321, 4
    sds = (detrand.test_weights(nets.g_param_shapes(style_nc=opt.style_code_dim), seed), detrand.test_weights(nets.d_param_shapes(4), seed + 1),
           detrand.test_weights(nets.d_param_shapes(7), seed + 2))
    for net, sd in zip((model.netG, model.netD, model.netD2), sds):
        net.load_state_dict(sd)
    batch = default_collate([make_sample(SIZE, 64, 64, seed + i, style_dim=opt.style_code_dim) for i in range(n)])
    random.seed(6)
    counts = [int(nets.dilated_mask_positions(batch["M"][i:i + 1].float()).shape[0]) for i in range(n)]
    draws = {"aug": detrand.uniform((4, n), 4, "aug") * 0.5 + 0.5, "more_idx": torch.tensor([random.sample(range(c), 32) for c in counts])}
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    ref = step.train_step(sds[0], sds[1], sds[2], adam, batch, draws, style_code=batch["style_code"].float())
    model._draws = draws
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3 and rel(model.fake_T, ref["fake_T"]) < 1e-3
    worst = (0.0, None)
    for nm, net, sd in (("G", model.netG, sds[0]), ("D", model.netD, sds[1]), ("D2", model.netD2, sds[2])):
        for k, p in net.named_parameters():
            if k.endswith("bias") and ((nm == "G" and not k.startswith(("down0.", "down7.", "up0.", "up0_T."))) or
                                       (nm != "G" and k.split(".")[1] in ("2", "5", "8"))):
                continue    # bias in front of a normalisation: mathematically zero gradient
            e = rel(p.grad, ref["grad_" + nm][k])
            worst = max(worst, (e, nm + "." + k))
            GRAD_WORST.append((e, nm, k))
            assert e < grad_tol(p.grad), (nm, k, e)
        for k, b in net.named_buffers():    # the oracle updated its state dicts in place: running statistics after the step
            if k.endswith("running_mean"):
                scale = float(sd[k.replace("running_mean", "running_var")].max().sqrt())
                assert (b.cpu() - sd[k]).abs().max().item() < 1e-3 * scale, (nm, k)
            elif b.dtype.is_floating_point:
                assert rel(b, sd[k]) < 1e-3, (nm, k)
            else:
                assert int(b) == int(sd[k]), (nm, k)
    print("worst gradient rel-L2", worst)


def test_inference_batch16_matches_oracle():
    """BASELINE config 4's shape: generator forward (test()) on 16 images of 1024 x 1024, every image against the CPU oracle"""
    from data.synthetic_dataset import make_sample
    model, opt = make_model("skitG", 16)
    sdG = detrand.test_weights(nets.g_param_shapes(style_nc=opt.style_code_dim), 55)
    model.netG.load_state_dict(sdG)
    model.eval()
    batch = default_collate([make_sample(SIZE, 64, 64, 700 + i, style_dim=opt.style_code_dim) for i in range(16)])
    model.set_input(batch, phase="test")
    for _ in range(3):          # eager, capture, replay: the captured forward must give the same images
        model.test()
    torch.cuda.synchronize()
    fake_I, fake_T = step.inference(sdG, batch, style_code=batch["style_code"].float())
    for i in range(16):
        assert rel(model.fake_I[i], fake_I[i]) < 1e-3 and rel(model.fake_T[i], fake_T[i]) < 1e-3, i
