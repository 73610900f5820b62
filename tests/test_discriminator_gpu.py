"""The multiscale PatchGAN discriminator off its defaults, and DiffAugment beyond 'bs', at operator level on the GPU.

  * depth (reference NLayerDiscriminator(n_layers), networks.py:1696-1737) 1 / 2 / 4 / 5 and the trailing Sigmoid the reference adds for
    gan_mode 'vanilla' (:1659, 1731-1732; BCEWithLogits still follows, :507-509): forward maps, loss, every parameter gradient, the
    gradient into the second concat source and the BatchNorm running statistics against the oracle's autograd (oracle/nets.py, pinned to
    the reference module by tests/test_oracle_golden.py at depths 2 / 3 / 4);
  * every DiffAugment letter and multi-letter policies against the REFERENCE's outputs (tests/golden/diffaug.npz).
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, nets  # noqa: E402  (checker only)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("n_layers,mode,cin,n,hw,ndf", [(2, "vanilla", 4, 2, 96, 8), (4, "vanilla", 7, 6, 32, 8), (1, "hinge", 4, 1, 64, 8),
                                                         (5, "nonsaturating", 4, 2, 128, 8), (4, "lsgan", 4, 1, 80, 64)])
def test_patchgan_depths_and_the_vanilla_sigmoid(n_layers, mode, cin, n, hw, ndf):
    from models import networks
    from vts import engine, ops

    dev = torch.device("cuda:0")
    seed = 77 + n_layers
    opt = SimpleNamespace(gan_mode=mode)
    D = networks.define_D(cin, ndf, "multiscale", n_layers, "batch", "xavier", 0.02, False, num_D=3, gpu_ids=[0], opt=opt)
    shapes = nets.d_param_shapes(cin, ndf=ndf, n_layers=n_layers)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
    assert D.use_sigmoid == (mode == "vanilla")
    sd = detrand.test_weights(shapes, seed)
    D.load_state_dict(sd)
    D.train()
    sd = {k: v.clone() for k, v in sd.items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    c0 = 1
    x0 = detrand.uniform((n, c0, hw, hw), seed, "x0")
    x1 = detrand.uniform((n, cin - c0, hw, hw), seed, "x1").requires_grad_(True)
    # ---- oracle: forward, GANLoss(mode, 0.8)(preds, real=True).mean() * 2.5, autograd
    preds_ref = nets.msd_forward(sd, torch.cat([x0, x1], 1), 3, n_layers=n_layers, use_sigmoid=mode == "vanilla")
    loss_ref = (nets.gan_loss(preds_ref, True, mode, real_label=0.8) * 1.0).mean() * 2.5
    loss_ref.backward()
    # ---- product
    for p in D.parameters():
        p.grad = torch.zeros_like(p)
    crit = networks.GANLoss(mode, target_real_label=0.8, target_fake_label=0.0)
    preds, ctx = engine.msd_forward(D, x0.to(dev), x1.detach().to(dev), keep=True)
    slot = ops.loss_slots(1, dev)
    dp = crit.accumulate(preds, True, 2.5, slot, pre_sigmoid=D.use_sigmoid)
    din = torch.zeros(n, cin - c0, hw, hw, device=dev)
    engine.msd_backward(D, ctx, dp, param_grads=True, accumulate=False, input_grad=(din, False))
    torch.cuda.synchronize()
    tol = 1e-3
    for k, b in D.named_buffers():
        if b.dtype.is_floating_point:
            if k.endswith("running_mean"):
                scale = float(sd[k.replace("running_mean", "running_var")].max().sqrt())
                assert (b.cpu() - sd[k]).abs().max().item() < tol * scale, k
            else:
                assert rel(b, sd[k]) < tol, k
        else:
            assert int(b) == int(sd[k]), k
    out = D(torch.cat([x0, x1.detach()], 1).to(dev))          # the module's own forward: the maps behind the Sigmoid where there is one
    for s in range(3):
        raw_ref = torch.logit(preds_ref[s][-1].detach().double().clamp(1e-12, 1 - 1e-12)).float() if mode == "vanilla" else preds_ref[s][-1].detach()
        assert preds[s].shape == preds_ref[s][-1].shape
        assert rel(preds[s], raw_ref) < tol, (s, rel(preds[s], raw_ref))
        assert rel(out[s][-1], preds_ref[s][-1]) < tol
    assert abs(ops.loss_values(slot)[0] - loss_ref.item()) <= tol * max(1.0, abs(loss_ref.item()))
    conv_idx, bn_idx, _ = nets.d_layout(n_layers)
    for k, p in D.named_parameters():
        if k.endswith("bias") and int(k.split(".")[1]) in bn_idx:      # conv bias in front of a BatchNorm: identically zero gradient
            continue
        assert rel(p.grad, sd[k].grad) < 2e-3, (k, rel(p.grad, sd[k].grad))
    assert rel(din, x1.grad) < 2e-3


def test_ganloss_on_sigmoid_outputs_kernel():
    """vts_ganloss mode 5 against autograd through sigmoid + BCEWithLogits, both labels"""
    import torch.nn.functional as F
    from vts import ops

    dev = torch.device("cuda:0")
    for real, label in ((True, 0.8), (False, 0.0), (True, 1.0)):
        p = (detrand.uniform((3, 1, 9, 11), 11, "p") * 6).requires_grad_(True)
        ref = F.binary_cross_entropy_with_logits(torch.sigmoid(p), torch.full_like(p, label)) * 2.5
        ref.backward()
        slot = ops.loss_slots(1, dev)
        dp = torch.empty(3, 1, 9, 11, device=dev)
        ops.ganloss(p.detach().to(dev), "vanilla_sigmoid", real, 2.5, slot, dp, label=label)
        assert abs(ops.loss_values(slot)[0] - ref.item()) < 1e-5 * max(1, abs(ref.item()))
        assert rel(dp, p.grad) < 1e-5


def test_diffaugment_policies_match_reference(golden_dir):
    from vts import ops

    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "diffaug.npz"))
    for pol in (str(v) for v in g["policies"]):
        for si, shape in enumerate(g["shapes"]):
            shape = tuple(int(v) for v in shape)
            x = detrand.uniform(shape, 11 + si, "diffaug_" + pol)
            torch.manual_seed(50 + si)
            d = nets.diffaug_draws(pol, shape)
            tag = "%s/%d" % (pol, si)
            # into a channel slice of a wider stack, as the step does (the D2 full-resolution stack)
            stack = torch.full((shape[0], shape[1] + 4, shape[2], shape[3]), 7.0, device=dev)
            y = ops.diffaug_policy(x.to(dev), pol, d, None, stack[:, 2:2 + shape[1]])
            torch.cuda.synchronize()
            assert bool((stack[:, :2] == 7).all()) and bool((stack[:, 2 + shape[1]:] == 7).all())
            if si == 0:
                assert (y.cpu() - torch.from_numpy(g[tag + "/out"])).abs().max().item() < 2e-6, tag
            else:
                assert (y[:, :, ::3, ::3].cpu() - torch.from_numpy(g[tag + "/out_sub"])).abs().max().item() < 2e-6, tag
                pr, ref = detrand.probe(y.contiguous().cpu(), "out"), g[tag + "/out_probe"]
                assert abs(pr[1] - ref[1]) <= 1e-5 * max(abs(ref[1]), 1e-12) and abs(pr[2] - ref[2]) <= 4e-5 * max(abs(ref[1]), 1e-12), tag
    # the mask rides on the last operation; draws made on the device have the reference's ranges
    shape = (3, 3, 40, 24)
    x = detrand.uniform(shape, 5, "x")
    M = (detrand.uniform((3, 1, 40, 24), 5, "m") > -0.3).float()
    torch.manual_seed(9)
    d = nets.diffaug_draws("tcbon", shape)
    y = ops.diffaug_policy(x.to(dev), "tcbon", d, M.to(dev), torch.empty(shape, device=dev))
    assert (y.cpu() - nets.diffaug(x, "tcbon", d) * M).abs().max().item() < 2e-6
    dd = ops.diffaug_draws("bsctno", (64, 3, 40, 24), dev)
    assert 0 <= float(dd[0]["r"].min()) and float(dd[2]["r"].max()) < 1
    assert int(dd[3]["tx"].min()) >= -5 and int(dd[3]["tx"].max()) <= 5 and int(dd[3]["ty"].min()) >= -3 and int(dd[3]["ty"].max()) <= 3
    assert int(dd[5]["ox"].min()) >= 0 and int(dd[5]["ox"].max()) <= 40 and int(dd[5]["oy"].max()) <= 24
    assert float(dd[4]["sigma"].max()) < 0.1 and bool((dd[4]["sigma"] == 0).any()) and dd[4]["noise"].shape == (64, 3, 40, 24)
    with pytest.raises(KeyError):
        ops.diffaug_draws("bx", (1, 3, 8, 8), dev)
