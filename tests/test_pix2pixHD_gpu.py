"""pix2pixHD baseline step on the HIP path (models/pix2pixHD_model.py) through create_model: against the golden vectors
produced by RUNNING THE REFERENCE Pix2PixHDModel (tests/golden/pix2pixHD_step_32.npz: small G / D, patch batch of 4)
and against the CPU oracle; HIP-graph replay; checkpoint keys.  Tolerances as in test_step_gpu.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import detrand, nets, step  # noqa: E402  (checker only)

FLAGS = ("--model pix2pixHD --gpu_ids 0 --no_vgg_loss True --ngf 8 --ndf 8 --n_downsample_global 3 --n_blocks_global 2 "
         "--batch_size 4 --checkpoints_dir /tmp/vts_test_ckpt --name p2p --dataset_mode patchskit")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def p2p_batch(n, size, seed):
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    M = (((yy - size / 2) / (0.45 * size)) ** 2 + ((xx - size / 2) / (0.4 * size)) ** 2 <= 1).float()[None, None].repeat(n, 1, 1, 1)
    return {"S_images": detrand.uniform((n, 1, size, size), seed, "S"), "M_images": M,
            "I_images": detrand.uniform((n, 3, size, size), seed, "I"), "T_images": 0.3 * detrand.uniform((n, 2, size, size), seed, "T"),
            "I_masks": torch.ones(n, size, size, dtype=torch.float64), "name": ["synthetic"] * n, "S_paths": ["synthetic.png"] * n,
            "augmentation_params": {}}


def make_model(extra=""):
    from models import create_model
    from options.train_options import TrainOptions
    opt = TrainOptions(cmd_line=FLAGS + extra).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    return model, opt


def load_weights(model, seed):
    sds = (detrand.test_weights(nets.resnet_param_shapes(1, 5, 8, 2, 3, norm="batch", down="stride", up="convT", conv_bias=True), seed),
           detrand.test_weights(nets.d_if_param_shapes(4, 8, 2), seed + 1), detrand.test_weights(nets.d_if_param_shapes(3, 8, 2), seed + 2))
    for net, sd in zip((model.netG, model.netD, model.netD2), sds):
        assert sorted(net.state_dict().keys()) == sorted(sd.keys())
        net.load_state_dict(sd)
    return sds


def test_step_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "pix2pixHD_step_32.npz"))
    size, seed, n = int(g["size"]), int(g["seed"]), int(g["n"])
    model, opt = make_model(" --use_hip_graph False")
    sdG, sdD, sdD2 = load_weights(model, seed)
    batch = p2p_batch(n, size, seed)
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    ref = dict(zip([str(k) for k in g["s0/loss_names"]], g["s0/loss_values"]))
    for k, v in ref.items():
        assert abs(losses[k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses[k], v)
    assert rel(model.fake_I, torch.from_numpy(g["s0/fake_I"])) < 1e-3 and rel(model.fake_T, torch.from_numpy(g["s0/fake_T"])) < 1e-3
    for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
        for k, p in net.named_parameters():
            rp = g["s0/grad_%s/%s" % (nm, k)]
            if k.endswith(".bias") and abs(rp[1]) < 1e-4:
                continue    # conv bias in front of a BatchNorm: rounding noise in the reference, exactly 0 here
            pr = detrand.probe(p.grad.cpu(), k)
            assert abs(pr[1] - rp[1]) <= 2e-3 * max(abs(rp[1]), 1e-12), (nm, k, pr, rp)
        for k, b in net.named_buffers():
            if b.dtype.is_floating_point:
                ref_b = torch.from_numpy(g["s0/buf_%s/%s" % (nm, k)])
                assert (b.cpu().double() - ref_b).abs().max().item() <= 1e-3 * max(1.0, float(ref_b.abs().max())), (nm, k)


def test_two_steps_match_oracle_and_graph_replay():
    size, seed, n = 32, 61, 4
    model, opt = make_model("")
    sdG, sdD, sdD2 = load_weights(model, seed)
    batch = p2p_batch(n, size, seed)
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    hp = step.p2p_hp(n_blocks_global=2, n_downsample_global=3)
    ref = step.p2p_train_step(sdG, sdD, sdD2, adam, batch, hp)
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)                      # eager
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3 and rel(model.fake_T, ref["fake_T"]) < 1e-3
    for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
        named = dict(net.named_parameters())
        for k, gr in ref["grad_" + nm].items():
            if k.endswith(".bias") and gr.norm() < 1e-4:
                continue
            assert rel(named[k].grad, gr) < 2e-3, (nm, k)
    # second step from the oracle's state (Adam beta1 = 0.5 keeps sign noise small, but compare from synced weights)
    for net, sd in zip((model.netG, model.netD, model.netD2), (sdG, sdD, sdD2)):
        net.load_state_dict({k: v.detach() for k, v in sd.items()})
    for nm, o in (("G", model.optimizer_G), ("D", model.optimizer_D), ("D2", model.optimizer_D2)):
        sdx = {"G": sdG, "D": sdD, "D2": sdD2}[nm]
        o.load_named_state({"G": model.netG, "D": model.netD, "D2": model.netD2}[nm], adam[nm]["m"], adam[nm]["v"], adam[nm]["step"])
    ref2 = step.p2p_train_step(sdG, sdD, sdD2, adam, batch, hp)
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)                      # captured HIP graphs from here on
    assert model._graphs is not None
    losses = model.get_current_losses()
    for k, v in ref2["losses"].items():
        assert abs(losses["l_" + k] - v) <= 2e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref2["fake_I"]) < 2e-3
    model.optimize_parameters(epoch=1)
    assert all(np.isfinite(v) for v in model.get_current_losses().values())


def test_inference_and_checkpoint_keys(tmp_path):
    from models import create_model
    from options.test_options import TestOptions
    model, opt = make_model("")
    sdG, _, _ = load_weights(model, 71)
    model.save_dir = str(tmp_path)
    model.save_networks("latest")
    saved = torch.load(os.path.join(str(tmp_path), "latest_net_D.pth"))
    assert sorted(saved.keys()) == sorted(nets.d_if_param_shapes(4, 8, 2).keys())
    topt = TestOptions(cmd_line="--model pix2pixHD --gpu_ids 0 --ngf 8 --n_downsample_global 3 --n_blocks_global 2 --return_patch True "
                                "--checkpoints_dir %s --name x --dataset_mode patchskit" % tmp_path).parse()
    tm = create_model(topt)
    tm.save_dir = str(tmp_path)
    tm.setup(topt)
    tm.parallelize()
    tm.eval()
    batch = p2p_batch(2, 32, 71)
    tm.set_input(batch, phase="test")
    tm.test()
    inp = step.p2p_prepare(batch)
    with torch.no_grad():
        _, fi, ft = step.p2p_generator({k: v.detach() for k, v in sdG.items()}, inp, step.p2p_hp(n_blocks_global=2, n_downsample_global=3), training=False)
    assert rel(tm.fake_I, fi) < 1e-3 and rel(tm.fake_T, ft) < 1e-3
