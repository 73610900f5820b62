"""End-to-end parity of the HIP training step (through models.create_model, i.e. the drop-in
boundary) against (a) the golden vectors produced by RUNNING THE REFERENCE on BASELINE
config 0 (sinskitG, 256x256, batch 1) and (b) the CPU oracle at batch 2.

Tolerance (north_star): outputs within 1e-3 rel-L2 of the reference.  Gradients are compared
at 2e-3 on their l2 norm / fixed random projection.  Biases of convolutions that feed an
Instance/BatchNorm have a mathematically-zero gradient (pure rounding noise in the reference
as well), so they are excluded from gradient and parameter comparisons.
"""
import json
import os

import numpy as np
import pytest
import torch
from torch.utils.data import default_collate

pytestmark = pytest.mark.gpu

# gradient comparisons against the oracle: every (relative L2 error, network, parameter) is recorded, the worst one is printed at the
# end of the module (pytest -s) and the bound is ~2x the worst value observed on the MI355X (round 4: see the fixture below)
GRAD_WORST = []
# north_star's 1e-3 on EVERY gradient tensor, by true relative L2.  Two judges: the oracle evaluated in FLOAT64 (oracle/step.py runs at
# torch's default dtype) and the fp32 CPU oracle.  Almost every tensor is within 1e-4 of float64.  The exceptions are not rounding noise
# but KINKS: a LeakyReLU / ReLU pre-activation within rounding distance of zero falls on different sides in fp32 and float64, which
# changes one element of the gradient map by a factor of 5 (or switches it off) -- 2e-3 .. 1e-2 of the tensor's norm on the 34 x 34 .. 129 x 129
# discriminator maps, carried down to every layer below (measured: tools/probes are not needed, `ref32 vs f64` in the report below shows
# it for the CPU oracle alone: 5.2e-3 on D layer2.* at batch 2, 1e-3 .. 2.5e-3 on D2 layer2.* in the second step, 1.5e-3 on G up3).
# The gradient is discontinuous there, so neither side is "the" answer: such a tensor must then agree within 1e-3 with the fp32 CPU
# oracle, which took the same side as the device.  (Rounds 4 / 5 compared with the fp32 oracle only, at 2e-3, worst 1.27e-3.)
GRAD_TOL = 1e-3


def f64_state(sds, adam=None):
    """deep copies of state dicts (and Adam states) in float64"""
    out = [{k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()} for sd in sds]
    if adam is None:
        return out
    a64 = {nm: {"step": st["step"], "m": {k: v.clone().double() for k, v in st["m"].items()}, "v": {k: v.clone().double() for k, v in st["v"].items()}}
           for nm, st in adam.items()}
    return out, a64


def train_step_f64(sds64, adam64, batch, draws, **kw):
    """oracle.step.train_step in float64 (the judge of both fp32 results)"""
    torch.set_default_dtype(torch.float64)
    try:
        return step.train_step(sds64[0], sds64[1], sds64[2], adam64, batch, draws, **kw)
    finally:
        torch.set_default_dtype(torch.float32)


def check_grad(p_grad, ref32, ref64, nm, k):
    e64, e32, e_cpu = rel(p_grad, ref64), rel(p_grad, ref32), rel(ref32, ref64)
    GRAD_WORST.append((min(e64, e32), nm, k, e64, e32, e_cpu))
    assert min(e64, e32) < GRAD_TOL, (nm, k, "HIP vs float64 %.3e" % e64, "HIP vs fp32 CPU %.3e" % e32, "fp32 CPU vs float64 %.3e" % e_cpu)


@pytest.fixture(scope="module", autouse=True)
def _report_worst_gradient():
    yield
    if GRAD_WORST:
        w = max(GRAD_WORST)
        smooth = [g for g in GRAD_WORST if g[5] < GRAD_TOL]            # tensors where fp32 CPU and float64 are on the same side of every kink
        kinks = [g for g in GRAD_WORST if g[5] >= GRAD_TOL]
        print("\n[%s] %d gradient tensors, bound %.1e.  Worst distance to the nearer judge: %.3e (%s %s).  Worst HIP vs float64 where the "
              "fp32 CPU oracle agrees with float64: %.3e.  Tensors with a kink between fp32 and float64 (fp32 CPU vs float64 >= bound): %d, "
              "there HIP vs fp32 CPU at most %.3e: %s"
              % (__name__, len(GRAD_WORST), GRAD_TOL, w[0], w[1], w[2], max(g[3] for g in smooth) if smooth else 0.0, len(kinks),
                 max(g[4] for g in kinks) if kinks else 0.0, [(g[1], g[2], "cpu32|f64 %.1e" % g[5]) for g in kinks]))

from oracle import detrand, nets, step  # noqa: E402  (checker only)

FLAGS = ("--model sinskitG --gpu_ids 0 --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False "
         "--lambda_G2_GAN_feat 0 --checkpoints_dir /tmp/vts_test_ckpt --name t --crop_size %d --batch_size %d")


def make_model(size, n):
    from models import create_model
    from options.train_options import TrainOptions

    opt = TrainOptions(cmd_line=FLAGS % (size, n)).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    return model, opt


def load_test_weights(model, seed):
    sds = (detrand.test_weights(nets.g_param_shapes(), seed), detrand.test_weights(nets.d_param_shapes(4), seed + 1),
           detrand.test_weights(nets.d_param_shapes(7), seed + 2))
    for net, sd in zip((model.netG, model.netD, model.netD2), sds):
        assert list(net.state_dict().keys()) == list(sd.keys())
        net.load_state_dict(sd)
    return sds


def null_grad_bias(net_name, key):
    if not key.endswith("bias"):
        return False
    if net_name == "G":
        return not any(key.startswith(p) for p in ("down0.", "down7.", "up0.", "up0_T."))
    return key.split(".")[1] in ("2", "5", "8")


def probe_close(t, ref, name, rtol):
    p = detrand.probe(t.detach().cpu(), name)
    scale = max(abs(ref[1]), 1e-12)
    assert abs(p[1] - ref[1]) <= rtol * scale, (name, p, ref)
    assert abs(p[2] - ref[2]) <= 4 * rtol * scale, (name, p, ref)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_step_matches_reference_golden(golden_dir):
    from data.synthetic_dataset import make_sample

    g = np.load(os.path.join(golden_dir, "sinskitG_step_256.npz"))
    size, seed, steps, nt = int(g["size"]), int(g["seed"]), int(g["steps"]), int(g["nt"])
    model, opt = make_model(size, 1)
    load_test_weights(model, seed)
    batch = default_collate([make_sample(size, nt, nt, seed)])
    for it in range(1):
        tag = "s%d" % it
        tol = 1e-3
        model._draws = {"aug": torch.from_numpy(g[tag + "/aug"]), "more_idx": torch.from_numpy(g[tag + "/more_idx"])}
        model.set_input(batch, phase="train")
        model.optimize_parameters(epoch=1)
        assert model.fake_sample_offset_x.cpu().tolist() == g[tag + "/more_ox"].astype(int).tolist()
        assert model.fake_sample_offset_y.cpu().tolist() == g[tag + "/more_oy"].astype(int).tolist()
        losses = model.get_current_losses()
        ref = dict(zip([str(s) for s in g[tag + "/loss_names"]], g[tag + "/loss_values"]))
        for k, v in losses.items():
            assert abs(v - ref[k]) <= tol * max(1.0, abs(ref[k])), (k, v, ref[k])
        assert rel(model.fake_I[:, :, ::4, ::4], torch.from_numpy(g[tag + "/fake_I_sub"])) < tol
        assert rel(model.fake_T[:, :, ::4, ::4], torch.from_numpy(g[tag + "/fake_T_sub"])) < tol
        for nm in ("fake_N", "aug_fake_I", "aug_real_I", "pred_fake_T_full", "pred_fake_I"):
            key = {"pred_fake_T_full": "pftf", "pred_fake_I": "pfi"}.get(nm, nm)
            probe_close(getattr(model, nm).contiguous(), g["%s/%s_probe" % (tag, nm)], key, 2 * tol)
        for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
            for k, p in net.named_parameters():
                if null_grad_bias(nm, k):
                    continue
                # (gradient PROBES of the reference's fp32 run at 2e-3 (norm) / 8e-3 (projection): that run is itself 7e-4 / 5.2e-3 away from
                #  its own float64 run on these probes for down0 ... down2 -- tests/golden/sinskitG_step_grads_256.npz, computed in
                #  DESIGN.md section 8 item 8 -- so they cannot be tightened; the 1e-3 bound on every gradient TENSOR is held against the
                #  float64 reference in test_generator_gradients_against_the_reference_in_float64 and against float64 oracles below)
                probe_close(p.grad, g["%s/grad_%s/%s" % (tag, nm, k)], k, 2 * tol)
                probe_close(p.data, g["%s/param_%s/%s" % (tag, nm, k)], k, tol)
            for k, b in net.named_buffers():
                refb = torch.from_numpy(g["%s/buf_%s/%s" % (tag, nm, k)])
                if k.endswith("running_mean"):
                    # a near-zero mean of O(1) activations: compare on the activation scale sqrt(running_var)
                    scale = float(np.sqrt(g["%s/buf_%s/%s" % (tag, nm, k.replace("running_mean", "running_var"))].max()))
                    assert (b.double().cpu() - refb).abs().max().item() < tol * scale, k
                elif b.dtype.is_floating_point:
                    assert rel(b, refb) < tol, k
                else:
                    assert int(b) == int(refb), k


def test_generator_gradients_against_the_reference_in_float64(golden_dir):
    """Every generator weight gradient of the reference step by TRUE relative L2 (not probes), against the reference run in FLOAT64
    (tests/golden/sinskitG_step_grads_256.npz, oracle/make_golden.py:golden_step_full_grads).  Bound: north_star's 1e-3 on every tensor.
    The fixture also records how far the reference's OWN fp32 CPU run is from its float64 run (`ref32_vs_64`): 5e-3 .. 6e-3 on
    down0 ... down2 -- so a comparison against the fp32 reference cannot hold 1e-3 there whatever the device computes; the probes of
    test_step_matches_reference_golden pass at 2e-3 only because a norm and one projection do not see most of that difference."""
    from data.synthetic_dataset import make_sample

    g = np.load(os.path.join(golden_dir, "sinskitG_step_grads_256.npz"))
    s = np.load(os.path.join(golden_dir, "sinskitG_step_256.npz"))
    size, seed, nt = int(g["size"]), int(g["seed"]), int(g["nt"])
    model, opt = make_model(size, 1)
    load_test_weights(model, seed)
    model._draws = {"aug": torch.from_numpy(s["s0/aug"]), "more_idx": torch.from_numpy(s["s0/more_idx"])}
    model.set_input(default_collate([make_sample(size, nt, nt, seed)]), phase="train")
    model.optimize_parameters(epoch=1)
    grads = dict(model.netG.named_parameters())
    keys = [f[len("g64/"):] for f in g.files if f.startswith("g64/")]
    assert len(keys) == 20
    rows = []
    for k in keys:
        err = rel(grads[k].grad, torch.from_numpy(g["g64/" + k]))
        rows.append((err, float(g["ref32_vs_64/" + k]), k))
        assert err < 1e-3, (k, err)
    worst = max(rows)
    print("\n[generator gradients vs the float64 reference] worst %.3e (%s; the reference's own fp32 run: %.3e); tensors where the HIP step is "
          "closer to float64 than the reference's fp32 run: %d of %d" % (worst[0], worst[2], worst[1], sum(e <= r for e, r, _ in rows), len(rows)))


def test_step_conditioning_ablations_match_reference_golden(golden_dir):
    """--use_cGAN False (D1 on the image alone: its input gradient is then the gradient w.r.t. the ONLY concat source), --use_cGAN_G2_S False
    and --use_cGAN_G2_I False (D2 stacks of 6 / 3 channels) and all three (D2 on the tactile patches alone): one REFERENCE step each
    (tests/golden/sinskitG_cond_step_256.npz, oracle/make_golden.py:COND_VARIANTS).  The two conditioning flags the reference cannot run
    itself (--use_cGAN_G2 False, --use_bg_mask False: probed) raise with that finding."""
    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions
    from oracle.make_golden import cond_channels

    for bad in ("--use_cGAN_G2 False", "--use_bg_mask False"):
        with pytest.raises(NotImplementedError, match="reference"):
            create_model(TrainOptions(cmd_line=(FLAGS % (256, 1)) + " " + bad).parse())
    g = np.load(os.path.join(golden_dir, "sinskitG_cond_step_256.npz"))
    size, seed, nt = int(g["size"]), int(g["seed"]), int(g["nt"])
    tol = 1e-3
    for vi, name in enumerate(str(v) for v in g["variants"]):
        extra = json.loads(str(g[name + "/flags"]))
        opt = TrainOptions(cmd_line=(FLAGS % (size, 1)) + " " + " ".join(extra)).parse()
        model = create_model(opt)
        model.setup(opt)
        model.parallelize()
        model.train()
        c1, c2 = cond_channels(extra)
        sds = (detrand.test_weights(nets.g_param_shapes(), seed + 10 * vi), detrand.test_weights(nets.d_param_shapes(c1), seed + 10 * vi + 1),
               detrand.test_weights(nets.d_param_shapes(c2), seed + 10 * vi + 2))
        for net, sd in zip((model.netG, model.netD, model.netD2), sds):
            assert sorted(net.state_dict().keys()) == sorted(sd.keys()), name
            net.load_state_dict(sd)
        model._draws = {"more_idx": torch.from_numpy(g[name + "/more_idx"]), "aug": torch.from_numpy(g[name + "/aug"])}
        model.set_input(default_collate([make_sample(size, nt, nt, seed + 10 * vi)]), phase="train")
        model.optimize_parameters(epoch=1)
        ref = dict(zip([str(s) for s in g[name + "/loss_names"]], g[name + "/loss_values"]))
        for k, v in model.get_current_losses().items():
            assert abs(v - ref[k]) <= tol * max(1.0, abs(ref[k])), (name, k, v, ref[k])
        for nm in ("fake_I", "fake_T", "pred_fake_T_full", "pred_fake_I"):
            key = {"pred_fake_T_full": "pftf", "pred_fake_I": "pfi"}.get(nm, nm)
            probe_close(getattr(model, nm).contiguous(), g["%s/%s_probe" % (name, nm)], key, 2 * tol)
        for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
            bn_convs = () if nm == "G" else tuple(str(c) for c in net.BN_IDX)
            for k, p in net.named_parameters():
                if (nm == "G" and null_grad_bias(nm, k)) or (nm != "G" and k.endswith("bias") and k.split(".")[1] in bn_convs):
                    continue
                probe_close(p.grad, g["%s/grad_%s/%s" % (name, nm, k)], k, 2 * tol)
                probe_close(p.data, g["%s/param_%s/%s" % (name, nm, k)], k, tol)
            for k, b in net.named_buffers():
                refb = torch.from_numpy(g["%s/buf_%s/%s" % (name, nm, k)])
                if k.endswith("running_mean"):
                    scale = float(np.sqrt(g["%s/buf_%s/%s" % (name, nm, k.replace("running_mean", "running_var"))].max()))
                    assert (b.double().cpu() - refb).abs().max().item() < tol * scale, (name, k)
                elif b.dtype.is_floating_point:
                    assert rel(b, refb) < tol, (name, k)
                else:
                    assert int(b) == int(refb), (name, k)
        # the same step captured and replayed (default graph path): losses stay finite and the replicas of the step's outputs keep their shapes
        assert model._full_stack.shape[1] == c2 and model._stack_all.shape[1] == c2


def test_step_variants_match_reference_golden(golden_dir):
    """One reference step per variant of oracle/make_golden.py:VARIANTS: PatchGAN depths 2 / 4 (--n_layers_D, --n_layers_D2), hinge,
    and the six-letter DiffAugment policy 'bsctno' (its draws regenerate from torch's CPU generator in the reference's order)."""
    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions

    g = np.load(os.path.join(golden_dir, "sinskitG_variants_step_256.npz"))
    size, seed, nt = int(g["size"]), int(g["seed"]), int(g["nt"])
    tol = 1e-3
    for vi, name in enumerate(str(v) for v in g["variants"]):
        extra = json.loads(str(g[name + "/flags"]))
        opt = TrainOptions(cmd_line=(FLAGS % (size, 1)) + " " + " ".join(extra)).parse()
        model = create_model(opt)
        model.setup(opt)
        model.parallelize()
        model.train()
        resnet = opt.netG.startswith("resnet_")
        gshapes = nets.resnet_param_shapes(n_blocks=int(opt.netG[len("resnet_")]), use_dropout=not opt.no_dropout) if resnet else nets.g_param_shapes()
        sds = (detrand.test_weights(gshapes, seed + 10 * vi),
               detrand.test_weights(nets.d_param_shapes(4, n_layers=opt.n_layers_D), seed + 10 * vi + 1),
               detrand.test_weights(nets.d_param_shapes(7, n_layers=opt.n_layers_D2), seed + 10 * vi + 2))
        for net, sd in zip((model.netG, model.netD, model.netD2), sds):
            assert sorted(k for k in net.state_dict().keys() if "filt" not in k) == sorted(sd.keys()), name
            net.load_state_dict(sd, strict=not (resnet and net is model.netG))
        model._draws = {"more_idx": torch.from_numpy(g[name + "/more_idx"])}
        torch.manual_seed(seed + vi)
        if not opt.no_dropout:
            model._draws["dropout"] = (nets.resnet_dropout_draws((1, size, size), n_blocks=int(opt.netG[len("resnet_")])) if resnet
                                       else nets.dropout_draws((1, size, size)))
        if opt.diffaugment == "bs":
            model._draws["aug"] = torch.from_numpy(g[name + "/aug"])
        else:
            model._draws["aug_policy"] = (nets.diffaug_draws(opt.diffaugment, (1, 3, size, size)), nets.diffaug_draws(opt.diffaugment, (1, 3, size, size)))
        model.set_input(default_collate([make_sample(size, nt, nt, seed + 10 * vi)]), phase="train")
        model.optimize_parameters(epoch=1)
        ref = dict(zip([str(s) for s in g[name + "/loss_names"]], g[name + "/loss_values"]))
        for k, v in model.get_current_losses().items():
            assert abs(v - ref[k]) <= tol * max(1.0, abs(ref[k])), (name, k, v, ref[k])
        assert rel(model.aug_fake_I[:, :, ::8, ::8], torch.from_numpy(g[name + "/aug_fake_I_sub"])) < tol, name
        for nm in ("fake_I", "fake_T", "aug_fake_I", "aug_real_I", "pred_fake_T_full", "pred_fake_I"):
            key = {"pred_fake_T_full": "pftf", "pred_fake_I": "pfi"}.get(nm, nm)
            probe_close(getattr(model, nm).contiguous(), g["%s/%s_probe" % (name, nm)], key, 2 * tol)
        for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
            bn_convs = () if nm == "G" else tuple(str(c) for c in net.BN_IDX)
            for k, p in net.named_parameters():
                if nm == "G" and resnet:
                    if k.endswith("bias") and not k.startswith("model.%d." % max(int(kk.split(".")[1]) for kk in dict(net.named_parameters()))):
                        continue        # every conv bias but the output conv's feeds an InstanceNorm: identically zero gradient
                elif (nm == "G" and null_grad_bias(nm, k)) or (nm != "G" and k.endswith("bias") and k.split(".")[1] in bn_convs):
                    continue
                probe_close(p.grad, g["%s/grad_%s/%s" % (name, nm, k)], k, 2 * tol)
                probe_close(p.data, g["%s/param_%s/%s" % (name, nm, k)], k, tol)
            for k, b in net.named_buffers():
                refb = torch.from_numpy(g["%s/buf_%s/%s" % (name, nm, k)])
                if k.endswith("running_mean"):
                    scale = float(np.sqrt(g["%s/buf_%s/%s" % (name, nm, k.replace("running_mean", "running_var"))].max()))
                    assert (b.double().cpu() - refb).abs().max().item() < tol * scale, (name, k)
                elif b.dtype.is_floating_point:
                    assert rel(b, refb) < tol, (name, k)
                else:
                    assert int(b) == int(refb), (name, k)


def test_second_step_from_synced_state(golden_dir):
    """Step 2 (Adam bias correction at step_count=2, BN running stats continuing) from a state
    synchronised with the oracle after step 1.  With beta1=0 the first Adam update is
    ~lr*sign(g), so weights whose gradient is rounding noise may differ by 2*lr between any two
    correct implementations; synchronising isolates the per-step arithmetic.  The oracle's own
    step 2 is pinned to the reference by tests/test_oracle_golden.py."""
    from data.synthetic_dataset import make_sample

    g = np.load(os.path.join(golden_dir, "sinskitG_step_256.npz"))
    size, seed, nt = int(g["size"]), int(g["seed"]), int(g["nt"])
    model, opt = make_model(size, 1)
    sds = load_test_weights(model, seed)
    batch = default_collate([make_sample(size, nt, nt, seed)])
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    d0 = {"aug": torch.from_numpy(g["s0/aug"]), "more_idx": torch.from_numpy(g["s0/more_idx"])}
    step.train_step(sds[0], sds[1], sds[2], adam, batch, d0, record=False)
    model._draws = d0
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)          # advances step counters / BN counters on the HIP side
    for nm, net, sd, optim in (("G", model.netG, sds[0], model.optimizer_G), ("D", model.netD, sds[1], model.optimizer_D),
                               ("D2", model.netD2, sds[2], model.optimizer_D2)):
        net.load_state_dict({k: v.detach() for k, v in sd.items()})
        optim.load_named_state(net, adam[nm]["m"], adam[nm]["v"], adam[nm]["step"])
    d1 = {"aug": torch.from_numpy(g["s1/aug"]), "more_idx": torch.from_numpy(g["s1/more_idx"])}
    sds64, adam64 = f64_state(sds, adam)                      # the same state, for the float64 evaluation of step 2
    ref = step.train_step(sds[0], sds[1], sds[2], adam, batch, d1)
    ref64 = train_step_f64(sds64, adam64, batch, d1)
    model._draws = d1
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3 and rel(model.fake_T, ref["fake_T"]) < 1e-3
    for nm, net, sd in (("G", model.netG, sds[0]), ("D", model.netD, sds[1]), ("D2", model.netD2, sds[2])):
        for k, p in net.named_parameters():
            if null_grad_bias(nm, k):
                continue
            check_grad(p.grad, ref["grad_" + nm][k], ref64["grad_" + nm][k], nm, k)
            assert rel(p.data, sd[k]) < 1e-3, (nm, k)


def test_step_batch2_matches_oracle():
    from data.synthetic_dataset import make_sample

    size, nt, seed, n = 256, 64, 77, 2
    model, opt = make_model(size, n)
    sdG, sdD, sdD2 = load_test_weights(model, seed)
    batch = default_collate([make_sample(size, nt, nt, seed + i) for i in range(n)])
    import random
    random.seed(5)
    counts = [int(nets.dilated_mask_positions(batch["M"][i:i + 1].float()).shape[0]) for i in range(n)]
    draws = {"aug": detrand.uniform((4, n), 3, "aug") * 0.5 + 0.5,
             "more_idx": torch.tensor([random.sample(range(c), 32) for c in counts])}
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    sds64 = f64_state((sdG, sdD, sdD2))
    ref64 = train_step_f64(sds64, {k: step.new_adam_state() for k in ("G", "D", "D2")}, batch, draws)
    ref = step.train_step(sdG, sdD, sdD2, adam, batch, draws)
    model._draws = draws
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3 and rel(model.fake_T, ref["fake_T"]) < 1e-3
    for nm, net, sd in (("G", model.netG, sdG), ("D", model.netD, sdD), ("D2", model.netD2, sdD2)):
        for k, p in net.named_parameters():
            if null_grad_bias(nm, k):
                continue
            check_grad(p.grad, ref["grad_" + nm][k], ref64["grad_" + nm][k], nm, k)
            # beta1 = 0: the first Adam update is ~lr*sign(g); elements whose gradient is rounding noise may
            # move by 2*lr either way, so post-step weights are compared at 3e-3 (the gradients above at 2e-3)
            assert rel(p.data, sd[k]) < 3e-3, (nm, k)
        for k, b in net.named_buffers():
            if k.endswith("running_mean"):
                scale = float(sd[k.replace("running_mean", "running_var")].max().sqrt())
                assert (b.cpu() - sd[k]).abs().max().item() < 1e-3 * scale, (nm, k)
            elif b.dtype.is_floating_point:
                assert rel(b, sd[k]) < 1e-3, (nm, k)


@pytest.mark.parametrize("name", ["G2_no_S", "G2_no_I"])
def test_step_batch2_conditioning_ablations_match_oracle(name):
    """The D2 conditioning ablations at batch 2 against the oracle.  The reference-golden test of these flags runs batch 1 (upstream
    asserts it), where the batch stride of the full-resolution D2 stack never matters: with --use_cGAN_G2_S False the mask channel of
    samples >= 1 was written into sample 0 (ops.g_post took the stride from the absent sketch slice) -- seen in the full-resolution
    prediction map and in the BatchNorm running statistics that pass advances (sinskitG_model.py:1490-1501)."""
    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions
    from oracle.make_golden import COND_VARIANTS, cond_channels
    import random

    size, nt, seed, n = 256, 64, 91, 2
    extra = COND_VARIANTS[name]
    opt = TrainOptions(cmd_line=(FLAGS % (size, n)) + " " + " ".join(extra)).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    c1, c2 = cond_channels(extra)
    sdG, sdD, sdD2 = (detrand.test_weights(nets.g_param_shapes(), seed), detrand.test_weights(nets.d_param_shapes(c1), seed + 1),
                      detrand.test_weights(nets.d_param_shapes(c2), seed + 2))
    for net, sd in zip((model.netG, model.netD, model.netD2), (sdG, sdD, sdD2)):
        net.load_state_dict(sd)
    batch = default_collate([make_sample(size, nt, nt, seed + i) for i in range(n)])
    random.seed(6)
    counts = [int(nets.dilated_mask_positions(batch["M"][i:i + 1].float()).shape[0]) for i in range(n)]
    draws = {"aug": detrand.uniform((4, n), 4, "aug") * 0.5 + 0.5,
             "more_idx": torch.tensor([random.sample(range(c), 32) for c in counts])}
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    oopt = step.hp()
    for k, v in zip(extra[::2], extra[1::2]):
        setattr(oopt, k.lstrip("-"), v == "True")
    ref64 = train_step_f64(f64_state((sdG, sdD, sdD2)), {k: step.new_adam_state() for k in ("G", "D", "D2")}, batch, draws, opt=oopt)
    ref = step.train_step(sdG, sdD, sdD2, adam, batch, draws, opt=oopt)
    model._draws = draws
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    assert model._full_stack.shape == (n, c2, size, size)
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.pred_fake_T_full, ref["pred_fake_T_full"]) < 1e-3
    for nm, net, sd in (("D", model.netD, sdD), ("D2", model.netD2, sdD2)):
        for k, p in net.named_parameters():
            if null_grad_bias(nm, k):
                continue
            check_grad(p.grad, ref["grad_" + nm][k], ref64["grad_" + nm][k], nm, k)
        for k, b in net.named_buffers():
            if k.endswith("running_mean"):
                scale = float(sd[k.replace("running_mean", "running_var")].max().sqrt())
                assert (b.cpu() - sd[k]).abs().max().item() < 1e-3 * scale, (nm, k)
            elif b.dtype.is_floating_point:
                assert rel(b, sd[k]) < 1e-3, (nm, k)


def test_inference_forward_matches_oracle_and_checkpoint_roundtrip(tmp_path):
    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.test_options import TestOptions

    size, seed = 256, 31
    model, opt = make_model(size, 1)
    sdG, _, _ = load_test_weights(model, seed)
    model.save_dir = str(tmp_path)
    model.save_networks("latest")
    saved = torch.load(os.path.join(str(tmp_path), "latest_net_G.pth"))
    assert list(saved.keys()) == list(nets.g_param_shapes().keys())
    topt = TestOptions(cmd_line="--model sinskitG --gpu_ids 0 --checkpoints_dir %s --name x --crop_size %d" % (tmp_path, size)).parse()
    tm = create_model(topt)
    tm.save_dir = str(tmp_path)
    tm.setup(topt)          # loads latest_net_G.pth
    tm.parallelize()
    tm.eval()
    batch = default_collate([make_sample(size, 8, 8, seed)])
    tm.set_input(batch, phase="test")
    tm.test()
    fi, ft = step.inference(sdG, batch)
    assert rel(tm.fake_I, fi) < 1e-3 and rel(tm.fake_T, ft) < 1e-3
    assert set(tm.get_current_visuals().keys()) >= {"real_S", "fake_I", "fake_gx", "fake_gy", "fake_N"}


def test_hip_graph_replay_equals_eager():
    """Steps replayed from the captured HIP graphs (step 2 onwards) equal eager execution;
    inputs are re-uploaded into the persistent buffers between steps."""
    import random

    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions

    size, n, seed = 256, 2, 55
    models_ = []
    for graph in (False, True):
        flags = (FLAGS % (size, n)) + " --use_diffaug False --use_hip_graph %s" % graph
        opt = TrainOptions(cmd_line=flags).parse()
        m = create_model(opt)
        m.setup(opt)
        m.parallelize()
        m.train()
        load_test_weights(m, seed)
        models_.append(m)
    batches = [default_collate([make_sample(size, 64, 64, seed + 10 * s + i) for i in range(n)]) for s in range(3)]
    for m in models_:
        random.seed(11)
        for b in batches + [batches[0]]:
            m.set_input(b, phase="train")
            m.optimize_parameters(epoch=1)
    eager, graph = models_
    assert graph._graphs is not None and eager._graphs is None
    le, lg = eager.get_current_losses(), graph.get_current_losses()
    for k in le:
        assert abs(le[k] - lg[k]) <= 1e-5 * max(1.0, abs(le[k])), (k, le[k], lg[k])
    for nm in ("G", "D", "D2"):
        a, b = getattr(eager, "flat" + nm).flat, getattr(graph, "flat" + nm).flat
        assert rel(b, a) < 1e-6, nm
        assert getattr(eager, "optimizer_" + nm).step_count == getattr(graph, "optimizer_" + nm).step_count == 4
        assert int(getattr(graph, "optimizer_" + nm).step_dev) == 4
    assert rel(graph.fake_I, eager.fake_I) < 1e-6


def test_discriminator_chains_equal_the_joined_schedule_bit_for_bit(monkeypatch):
    """engine.msd_chain (per-discriminator chains: update -> Adam -> its passes of the generator step, no join between D1 and D2; input
    pyramids pooled inside the lanes) launches exactly the kernels of the joined schedule (msd_multi per stage, pyramids pooled up front)
    on the same operands in the same per-tensor order: four steps -- eager, capture, two replays -- end in bit-identical weights, Adam
    moments, BatchNorm buffers and logged losses."""
    import random

    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions
    from vts import engine

    size, n, seed = 256, 2, 58
    batches = [default_collate([make_sample(size, 64, 64, seed + 10 * s + i) for i in range(n)]) for s in range(2)]
    res = {}
    for chains in (True, False):
        monkeypatch.setattr(engine, "D_CHAINS", chains)
        monkeypatch.setattr(engine, "LAZY_PYRAMID", chains)
        opt = TrainOptions(cmd_line=(FLAGS % (size, n)) + " --use_hip_graph True").parse()
        m = create_model(opt)
        m.setup(opt)
        m.parallelize()
        m.train()
        load_test_weights(m, seed)
        random.seed(12)
        torch.manual_seed(12)
        for b in batches + batches:
            m.set_input(b, phase="train")
            m.optimize_parameters(epoch=1)
        torch.cuda.synchronize()
        assert m._graphs is not None
        res[chains] = dict(flat={nm: getattr(m, "flat" + nm).flat.clone() for nm in ("G", "D", "D2")},
                           m={nm: getattr(m, "optimizer_" + nm).m.clone() for nm in ("G", "D", "D2")},
                           bufs={nm + "." + k: b.clone() for nm in ("D", "D2") for k, b in getattr(m, "net" + nm).named_buffers()},
                           losses=m.get_current_losses(), nodes=sum(k for _, k in m.graph_nodes))
    a, b = res[True], res[False]
    for nm in ("G", "D", "D2"):
        assert torch.equal(a["flat"][nm], b["flat"][nm]), nm
        assert torch.equal(a["m"][nm], b["m"][nm]), nm
    for k in a["bufs"]:
        assert torch.equal(a["bufs"][k], b["bufs"][k]), k
    assert a["losses"] == b["losses"]
    assert a["nodes"] > 0 and b["nodes"] > 0


def test_train_and_test_scripts_end_to_end(tmp_path):
    """The headless train.py / test.py entry points run against the synthetic dataset, write the
    reference checkpoint file set and loss log, and test.py reloads the generator."""
    import subprocess
    import sys

    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-tactile-synthesis_amd")
    common = ["--model", "sinskitG", "--gpu_ids", "0", "--dataset_mode", "synthetic", "--crop_size", "256", "--checkpoints_dir",
              str(tmp_path), "--name", "e2e"]
    train = [sys.executable, os.path.join(pkg, "train.py")] + common + [
        "--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0", "--use_vision_aided_loss", "False", "--data_len", "3", "--n_epochs", "1",
        "--n_epochs_decay", "1", "--print_freq", "1", "--save_latest_freq", "2", "--save_epoch_freq", "1", "--batch_size", "1"]
    out = subprocess.run(train, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = os.path.join(str(tmp_path), "e2e")
    for f in ("latest_net_G.pth", "latest_net_D.pth", "latest_net_D2.pth", "best_net_G.pth", "1_net_G.pth", "loss_log.txt", "train_opt.txt"):
        assert os.path.exists(os.path.join(d, f)), f
    log = open(os.path.join(d, "loss_log.txt")).read()
    assert "l_G_GAN" in log and "l_G2_L1" in log and "nan" not in log.lower()
    assert "learning rate = 0.0005000" in out.stdout      # LambdaLR: 1 - 1/(n_epochs_decay+1) after epoch 1
    test = [sys.executable, os.path.join(pkg, "test.py")] + common + ["--epoch", "latest", "--eval", "--results_dir", str(tmp_path / "res"),
                                                                       "--num_test", "2", "--data_len", "2"]
    out = subprocess.run(test, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("processed synthetic_") == 2
    saved = torch.load(os.path.join(str(tmp_path), "res", "e2e", "test_latest", "synthetic_1234.pt"))
    assert saved["fake_I"].shape == (1, 3, 256, 256) and saved["fake_gx"].shape == (1, 1, 256, 256)


def test_train_script_runs_with_the_reference_default_loss_flags(tmp_path):
    """train.py with NO loss flag touched -- the published command's defaults: both LPIPS terms on (stand-in VGG16 weights here) and
    --use_vision_aided_loss True -- trains and logs the reference's loss names, the three D3 entries as 0 before the warm-up epoch; with
    the warm-up epoch moved to 1 the run stops with the message that names the flag (the CLIP term is not built)."""
    import subprocess
    import sys

    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-tactile-synthesis_amd")
    common = [sys.executable, os.path.join(pkg, "train.py"), "--model", "sinskitG", "--gpu_ids", "0", "--dataset_mode", "synthetic", "--crop_size", "256",
              "--checkpoints_dir", str(tmp_path), "--data_len", "2", "--n_epochs", "1", "--n_epochs_decay", "0", "--print_freq", "1",
              "--save_latest_freq", "100", "--save_epoch_freq", "100", "--batch_size", "1"]
    out = subprocess.run(common + ["--name", "dflt"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    log = open(os.path.join(str(tmp_path), "dflt", "loss_log.txt")).read()
    for name in ("l_G_GAN", "l_G_D3", "l_D3_real_I", "l_D3_fake_I", "l_G_L1", "l_G_lpips", "l_G2_lpips"):
        assert name in log, name
    assert "l_G_D3: 0.000" in log and "nan" not in log.lower()
    out = subprocess.run(common + ["--name", "warm", "--vision_aided_warmup_epoch", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode != 0 and "vision_aided_warmup_epoch" in out.stderr and "use_vision_aided_loss False" in out.stderr, out.stderr[-1500:]
    # the cutoff is announced at construction, and what was trained is on disk before the run stops
    assert "WARNING: --use_vision_aided_loss True" in out.stdout and "STOPS" in out.stdout, out.stdout[-1500:]
    for net in ("G", "D", "D2"):
        assert os.path.exists(os.path.join(str(tmp_path), "warm", "latest_net_%s.pth" % net)), net


def test_skitG_trains_from_the_multi_material_dataset(tmp_path):
    """train.py --model skitG --dataset_mode skit: two seeded materials at the reference's relative place ./datasets/singleskit_<m>_padded_<size>_x1/
    (data/skit_dataset.py), each with a precomputed style code; batches alternate between the materials, checkpoints and the loss log appear"""
    import subprocess
    import sys

    from data.synthetic_material import write_material

    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-tactile-synthesis_amd")
    for m, ms in (("matA", 31), ("matB", 32)):
        root = write_material(str(tmp_path / "datasets" / ("singleskit_%s_padded_400_x1" % m)), seed=ms, phase="train")
        code = np.random.RandomState(ms).randn(512).astype(np.float32)
        np.save(os.path.join(root, "style_code.npy"), code / np.linalg.norm(code))
    train = [sys.executable, os.path.join(pkg, "train.py"), "--model", "skitG", "--gpu_ids", "0", "--dataset_mode", "skit", "--material_list", "matA",
             "matB", "--padded_size", "400", "--dataroot", "unused", "--preprocess", "zoom_crop", "--random_scale_max", "1.04", "--crop_size", "320",
             "--center_w", "200", "--center_h", "160", "--batch_size_G2", "8", "--batch_size_G2_val", "6", "--add_fake_T_sample_size", "4",
             "--w_resampling", "False", "--checkpoints_dir", str(tmp_path), "--name", "skit", "--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0",
             "--use_vision_aided_loss", "False", "--data_len", "4", "--n_epochs", "1", "--n_epochs_decay", "0", "--print_freq", "1",
             "--save_latest_freq", "2", "--save_epoch_freq", "1", "--batch_size", "1"]
    out = subprocess.run(train, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    d = os.path.join(str(tmp_path), "skit")
    for f in ("latest_net_G.pth", "latest_net_D.pth", "latest_net_D2.pth", "loss_log.txt"):
        assert os.path.exists(os.path.join(d, f)), f
    log = open(os.path.join(d, "loss_log.txt")).read()
    assert "l_G_GAN" in log and "l_G2_L1" in log and "nan" not in log.lower()
    assert "material_list is ['matA', 'matB']" in out.stdout


def test_inference_graph_replay_equals_eager_and_follows_new_inputs():
    from data.synthetic_dataset import make_sample
    model, opt = make_model(256, 1)
    load_test_weights(model, 5)
    model.eval()
    outs = []
    for seed in (1, 2, 3, 2):
        model.set_input(default_collate([make_sample(256, 8, 8, seed)]), phase="test")
        model.test()
        outs.append((model.fake_I.clone(), model.fake_T.clone()))
    assert model._infer_graph is not None               # calls 3 and 4 were graph replays
    opt.use_hip_graph = False
    model.set_input(default_collate([make_sample(256, 8, 8, 2)]), phase="test")
    model.test()
    for o in (outs[1], outs[3]):
        assert torch.equal(o[0], model.fake_I) and torch.equal(o[1], model.fake_T)
    assert not torch.equal(outs[2][0], outs[1][0])


def test_validation_replay_between_training_replays_reads_its_own_outputs():
    """train.py's loop: training steps replayed from HIP graphs, a validation test() in between (eager, then captured, then replayed).
    After every test() the model's outputs / metrics must be the VALIDATION batch's -- a replayed training step rebinds fake_I / fake_T
    to the training graphs' tensors, so the inference replay has to re-attach its own (round-2 advisor finding)."""
    import random

    from data.synthetic_dataset import make_sample
    model, opt = make_model(256, 1)
    load_test_weights(model, 21)
    random.seed(3)
    train_b = [default_collate([make_sample(256, 64, 64, 300 + i)]) for i in range(2)]
    val_b = default_collate([make_sample(256, 16, 24, 77)])
    for it in range(5):
        model.train()
        model.set_input(train_b[it % 2], phase="train")
        model.optimize_parameters(epoch=1)
        model.eval()
        model.set_input(val_b, phase="val")
        model.test()
        got_I, got_T = model.fake_I.clone(), model.fake_T.clone()
        got_m = model.compute_metrics()
        # the same weights through an eager forward
        keep, opt.use_hip_graph = opt.use_hip_graph, False
        model.test()
        opt.use_hip_graph = keep
        assert torch.equal(got_I, model.fake_I) and torch.equal(got_T, model.fake_T), "validation %d read another batch's outputs" % it
        ref_m = model.compute_metrics()
        for k in ref_m:
            assert got_m[k] == ref_m[k] or (got_m[k] != got_m[k] and ref_m[k] != ref_m[k]), (it, k, got_m[k], ref_m[k])
    assert model._graphs is not None and model._infer_graph is not None     # both kinds of replay really happened


def test_step_on_a_singleskit_dataset_batch_matches_oracle(tmp_path):
    """the REAL dataset front-end in front of the HIP step: a TouchClothing-format material on disk (data/synthetic_material.py) ->
    data/singleskit_dataset.py (bit-identical to the reference class, tests/test_dataset.py) -> default_collate -> set_input ->
    optimize_parameters, against the oracle on the same collated batch (random crop, 8 sampled tactile squares with their contact
    masks and crop offsets, real augmentation parameters)"""
    import random

    from data.singleskit_dataset import SingleSkitDataset
    from data.synthetic_material import write_material
    from models import create_model
    from options.train_options import TrainOptions
    from oracle.make_dataset_golden import dataset_opt

    root = write_material(str(tmp_path / "m"), seed=3)
    random.seed(1)
    np.random.seed(1)
    ds = SingleSkitDataset(dataset_opt(root, "train", w_resampling=True))
    batch = default_collate([ds[0]])
    nt = batch["T_images"].shape[1]
    assert nt == 8 and batch["S"].shape == (1, 1, 256, 256)
    opt = TrainOptions(cmd_line=(FLAGS % (256, 1)) + " --batch_size_G2 %d" % nt).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    sds = load_test_weights(model, 91)
    random.seed(2)
    cnt = int(nets.dilated_mask_positions(batch["M"].float()).shape[0])
    draws = {"aug": detrand.uniform((4, 1), 5, "aug") * 0.5 + 0.5, "more_idx": torch.tensor([random.sample(range(cnt), 32)])}
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    ref64 = train_step_f64(f64_state(sds), {k: step.new_adam_state() for k in ("G", "D", "D2")}, batch, draws, opt=step.hp(batch_size_G2=nt))
    ref = step.train_step(sds[0], sds[1], sds[2], adam, batch, draws, opt=step.hp(batch_size_G2=nt))
    model._draws = draws
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3 and rel(model.fake_T, ref["fake_T"]) < 1e-3
    assert rel(model.fake_T_concat, ref["fake_T_concat"]) < 1e-3
    for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
        named = dict(net.named_parameters())
        for k, gr in ref["grad_" + nm].items():
            if null_grad_bias(nm, k):
                continue
            check_grad(named[k].grad, gr, ref64["grad_" + nm][k], nm, k)


def test_eval_metrics_match_oracle():
    """I_PSNR / T_AE / T_MSE kernels vs the oracle (pinned to the reference's compute_evaluation_metric), and through the model"""
    from data.synthetic_dataset import make_sample
    from vts import ops
    dev = torch.device("cuda:0")
    real_I, fake_I = detrand.uniform((2, 3, 70, 90), 8, "rI"), 1.2 * detrand.uniform((2, 3, 70, 90), 8, "fI")
    real_T, fake_T = 0.3 * detrand.uniform((9, 2, 32, 32), 8, "rT"), 0.6 * detrand.uniform((9, 2, 32, 32), 8, "fT")
    got = ops.eval_metrics(real_I.to(dev), fake_I.to(dev), real_T.to(dev), fake_T.to(dev)).cpu().tolist()
    ref = nets.eval_metrics(real_I, fake_I, real_T, fake_T)
    for v, k in zip(got, ("I_PSNR", "T_AE", "T_MSE", "I_SSIM")):
        assert abs(v - ref[k]) <= 1e-4 * max(1.0, abs(ref[k])), (k, v, ref[k])
    # SSIM sanity that does not depend on the restatement: identical images give exactly 1, and it is symmetric in its arguments
    same = ops.eval_metrics(real_I.to(dev), real_I.to(dev), real_T.to(dev), fake_T.to(dev)).cpu().tolist()
    assert abs(same[3] - 1.0) < 1e-6
    model, opt = make_model(256, 1)
    sdG, _, _ = load_test_weights(model, 13)
    batch = default_collate([make_sample(256, 16, 24, 13)])
    model.set_input(batch, phase="val")
    model.test()
    m = model.compute_metrics()
    fi, ft = step.inference(sdG, batch)
    coords = batch["val_T_coords"][0]
    ox, oy, _ = nets.find_coords_for_patch(coords)
    fake_T_concat = nets.gather_patches(ft, ox, oy, 32)
    inp = step.prepare_input(batch)
    real_T_val = batch["val_T_images"][0].float() * batch["val_I_masks"][0].float()[:, None]
    ref = nets.eval_metrics(inp.real_I, fi, real_T_val, fake_T_concat)
    for k in ("I_PSNR", "T_AE", "T_MSE", "I_SSIM"):
        assert abs(m[k] - ref[k]) <= 2e-3 * max(1.0, abs(ref[k])), (k, m[k], ref[k])
    assert set(model.get_current_metrics().keys()) == {"m_I_PSNR", "m_T_AE", "m_T_MSE", "m_I_SSIM"}


def test_ssim_known_answers():
    """I_SSIM (torchmetrics' structural_similarity_index_measure, model_utils.py:496-499) is a third-party dependency that is absent
    here: its kernel stays PARITY-UNPINNED against torchmetrics itself.  What CAN be pinned without it are the closed forms of Wang et
    al.'s definition that any implementation must reproduce: two constant images a, b give (2ab + C1) / (a^2 + b^2 + C1) in every
    window (variances and covariance vanish, the contrast-structure factor is C2 / C2), with C1 = (0.01 * data_range)^2; identical
    images give exactly 1; the measure is symmetric."""
    from vts import lib as L, ops
    dev = torch.device("cuda:0")
    lib = L.load()
    ws = ops.workspace(lib.vts_metric_ws_floats(), dev)
    lohi = torch.tensor([-1.0, 1.0], device=dev)        # the normalisation range: x -> (x + 1) / 2, data_range 1
    out = torch.zeros(1, device=dev)

    def ssim(a, b):
        L.check(lib.vts_metric_ssim(a.data_ptr(), b.data_ptr(), a.shape[0] * a.shape[1], a.shape[2], a.shape[3], lohi.data_ptr(), out.data_ptr(), ws.data_ptr(), L.stream()),
                "vts_metric_ssim")
        return float(out.item())

    for va, vb in ((0.2, -0.4), (-0.9, 0.9), (0.0, 0.5)):
        a, b = torch.full((1, 3, 40, 52), va, device=dev), torch.full((1, 3, 40, 52), vb, device=dev)
        x, y = (va + 1) / 2, (vb + 1) / 2
        want = (2 * x * y + 1e-4) / (x * x + y * y + 1e-4)
        # 5e-4: the windowed moments are E[x^2] - mu^2 in fp32 (as in torchmetrics), whose rounding residue ~1e-8 sits beside C2 = 9e-4
        assert abs(ssim(a, b) - want) < 5e-4, (va, vb, ssim(a, b), want)
    g = torch.Generator().manual_seed(4)
    p, q = (torch.rand(2, 3, 37, 61, generator=g) * 2 - 1).to(dev), (torch.rand(2, 3, 37, 61, generator=g) * 2 - 1).to(dev)
    assert abs(ssim(p, p) - 1.0) < 1e-6 and abs(ssim(p, q) - ssim(q, p)) < 1e-6 and ssim(p, q) < 0.2


def test_sifid_chain_matches_oracle_and_reference_golden(golden_dir, monkeypatch):
    """I_SIFID / T_SIFID on the HIP path (vts_sifid_input, Inception block 0 on the conv kernels, device Frechet distance) vs (i) the
    values the REFERENCE's compute_evaluation_metric chain produced for the same seeded inputs and stand-in weights
    (tests/golden/sifid.npz), (ii) the oracle on a second input set, feature maps included, (iii) through the model's compute_metrics"""
    from data.synthetic_dataset import make_sample
    from models import inception
    from oracle.make_golden import sifid_inputs
    from vts import engine, ops
    dev = torch.device("cuda:0")
    net = inception.InceptionBlock0()
    sd = net.state_dict()
    net = net.to(dev)
    g = np.load(os.path.join(golden_dir, "sifid.npz"))
    real_I, fake_I, real_T, fake_T = sifid_inputs(int(g["seed"]))
    got_i = float(engine.sifid_images(net, real_I.to(dev), fake_I.to(dev)))
    got_t = float(engine.sifid_tactile(net, real_T.to(dev), fake_T.to(dev)))
    assert abs(got_i - float(g["I_SIFID"])) <= 2e-3 * abs(float(g["I_SIFID"])), (got_i, float(g["I_SIFID"]))
    assert abs(got_t - float(g["T_SIFID"])) <= 2e-3 * abs(float(g["T_SIFID"])), (got_t, float(g["T_SIFID"]))
    # feature maps and the input preparation against the oracle (odd sizes, two images)
    x = detrand.uniform((2, 3, 75, 101), 31, "x")
    f = engine.inception_block0(net, x.to(dev))
    ref = nets.inception_block0(x, sd)
    assert f.shape == ref.shape and rel(f, ref) < 1e-5
    lohi = ops.minmax(x.to(dev))
    prep = ops.sifid_input((1.2 * x).to(dev), 0, 3, lohi=lohi, clamp01=True)
    assert rel(prep, 2 * torch.clamp((1.2 * x - x.min()) / (x.max() - x.min()), 0, 1) - 1) < 1e-6
    t = detrand.uniform((3, 2, 32, 32), 32, "t")
    up = ops.sifid_input(t.to(dev), 1, 1, size=(299, 299), clamp01=True)
    want = torch.nn.functional.interpolate(torch.clamp(t[:, 1:2], 0, 1), (299, 299)).repeat(1, 3, 1, 1)
    assert torch.equal(up.cpu(), want)
    # through the model: the two names appear beside the four weight-free metrics
    monkeypatch.setenv("VTS_SIFID", "1")
    model, opt = make_model(256, 1)
    sdG, _, _ = load_test_weights(model, 13)
    batch = default_collate([make_sample(256, 16, 24, 13)])
    model.set_input(batch, phase="val")
    model.test()
    m = model.compute_metrics()
    fi, ft = step.inference(sdG, batch)
    ox, oy, _ = nets.find_coords_for_patch(batch["val_T_coords"][0])
    fake_T_concat = nets.gather_patches(ft, ox, oy, 32)
    real_T_val = batch["val_T_images"][0].float() * batch["val_I_masks"][0].float()[:, None]
    inp = step.prepare_input(batch)
    assert abs(m["I_SIFID"] - nets.sifid_images(inp.real_I, fi, sd)) <= 5e-3 * abs(m["I_SIFID"]) + 1e-6
    assert abs(m["T_SIFID"] - nets.sifid_tactile(real_T_val, fake_T_concat, sd)) <= 5e-3 * abs(m["T_SIFID"]) + 1e-6
    assert {"m_I_SIFID", "m_T_SIFID"} <= set(model.get_current_metrics().keys()) and model.metric_sifid_pretrained is False


def test_style_code_generator_matches_reference_golden(golden_dir):
    """the skitG generator (CustomUnetGenerator + style code, the network of the headline configuration) on the HIP engine vs the
    REFERENCE module run on CPU (tests/golden/nets_style_256.npz): outputs and every parameter gradient"""
    from models import create_model
    from options.train_options import TrainOptions
    from vts import engine

    g = np.load(os.path.join(golden_dir, "nets_style_256.npz"))
    size, seed, n, sd_dim, nl = (int(g[k]) for k in ("size", "seed", "n", "style_code_dim", "num_layer_style_code"))
    flags = (FLAGS % (size, n)).replace("--model sinskitG", "--model skitG")
    opt = TrainOptions(cmd_line=flags).parse()
    assert opt.use_style_code and opt.style_code_dim == sd_dim and opt.num_layer_style_code == nl
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    G = model.netG
    G.load_state_dict(detrand.test_weights(nets.g_param_shapes(style_nc=sd_dim, num_layer_style_code=nl), seed))
    dev = torch.device("cuda:0")
    x = detrand.uniform((n, 9, size, size), seed, "g_in").to(dev)
    sc = detrand.uniform((n, sd_dim), seed, "style")
    sc = (sc / sc.norm(dim=1, keepdim=True)).to(dev)
    y, ctx = engine.unet_forward(G, x, style_code=sc)
    assert rel(y[:, :, ::4, ::4], torch.from_numpy(g["G_out_sub"])) < 1e-4
    probe_close(y, g["G_out_probe"], "g_out", 2e-4)
    cot = detrand.uniform(tuple(y.shape), seed, "g_cot").to(dev)
    engine.unet_backward(G, ctx, (cot * (1.0 - y * y)).contiguous())
    for k, p in G.named_parameters():
        if null_grad_bias("G", k):
            continue
        probe_close(p.grad, g["G_grad/" + k], k, 1e-3)


def test_style_code_project_and_adain_modes_match_reference_golden(golden_dir):
    """--style_code_mapping_mode project with --style_code_mode concat (BatchNorm1d, batch 2) and adain (InstanceNorm1d, batch 1) on the
    HIP engine at the reference's 1536-pixel design size vs the REFERENCE module run on CPU (tests/golden/nets_style_modes_1536.npz):
    outputs, style-code gradient, every parameter gradient incl. style_code_mapping0, BatchNorm1d running statistics; AdaIN operator vs
    the oracle"""
    from models import create_model
    from options.train_options import TrainOptions
    from oracle.make_golden import STYLE_MODE_CASES, style_mode_shapes
    from vts import engine, ops

    g = np.load(os.path.join(golden_dir, "nets_style_modes_1536.npz"))
    size, seed = int(g["size"]), int(g["seed"])
    dev = torch.device("cuda:0")
    xa, sa = detrand.uniform((2, 7, 6, 6), 3, "ax"), 0.5 + detrand.uniform((2, 7, 6, 6), 3, "as")
    xo, so = xa.clone().requires_grad_(True), sa.clone().requires_grad_(True)
    yo = nets.adain(xo, so)
    cot = detrand.uniform(tuple(yo.shape), 3, "ac")
    (yo * cot).sum().backward()
    assert rel(ops.adain(xa.to(dev), sa.to(dev)), yo) < 1e-6
    dxa, dsa = ops.adain_bwd(cot.to(dev), xa.to(dev), sa.to(dev))
    assert rel(dxa, xo.grad) < 1e-5 and rel(dsa, so.grad) < 1e-5
    for mode, mapping, n in STYLE_MODE_CASES:
        flags = (FLAGS % (size, n)).replace("--model sinskitG", "--model skitG") + " --style_code_mode %s --style_code_mapping_mode %s" % (mode, mapping)
        opt = TrainOptions(cmd_line=flags).parse()
        model = create_model(opt)
        model.setup(opt)
        model.parallelize()
        G = model.netG
        t = mode + "/"
        assert sorted(G.state_dict().keys()) == sorted(g[t + "keys"].tolist())          # checkpoint compatible with the reference
        G.load_state_dict(detrand.test_weights(style_mode_shapes(mode, n), seed), strict=False)
        G.train()
        x = detrand.uniform((n, 9, size, size), seed, "g_in").to(dev)
        sc = detrand.uniform((n, 512), seed, "style")
        sc = (sc / sc.norm(dim=1, keepdim=True)).to(dev)
        y, ctx = engine.unet_forward(G, x, style_code=sc)
        assert rel(y[:, :, ::16, ::16], torch.from_numpy(g[t + "G_out_sub"])) < 1e-4
        probe_close(y, g[t + "G_out_probe"], "g_out", 2e-4)
        cot = detrand.uniform(tuple(y.shape), seed, "g_cot").to(dev)
        model.flatG.grad.zero_()
        engine.unet_backward(G, ctx, (cot * (1.0 - y * y)).contiguous())
        assert rel(ctx.dstyle, torch.from_numpy(g[t + "G_dstyle"])) < 5e-3
        for k, p in G.named_parameters():
            if null_grad_bias("G", k) and not k.startswith("style_code_mapping"):
                continue
            if mode == "adain" and k == "down7.model.1.bias":
                continue
            probe_close(p.grad, g[t + "G_grad/" + k], k, 1e-3)
        if n > 1:
            sdict = G.state_dict()
            assert rel(sdict["style_code_mapping0.1.running_mean"], torch.from_numpy(g[t + "bn_running_mean"])) < 1e-5
            assert rel(sdict["style_code_mapping0.1.running_var"], torch.from_numpy(g[t + "bn_running_var"])) < 1e-5
        del model, G, ctx, y
        torch.cuda.empty_cache()


def test_frechet_distance_matches_reference(golden_dir):
    """vts_frechet_distance (float64 moments + Newton-Schulz matrix square root) vs the REFERENCE's calculate_frechet_distance on
    synthetic features (tests/golden/metrics.npz) and vs the oracle restatement (scipy sqrtm)"""
    from vts import ops
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    seed = int(g["seed"])
    dev = torch.device("cuda:0")
    for i in range(len(nets.FRECHET_CASES)):
        f1, f2 = nets.frechet_case(i, seed)
        got = float(ops.frechet_distance(f1.to(dev), f2.to(dev)).cpu())
        ref = float(g["fd/%d" % i])
        assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (i, got, ref, nets.frechet_distance(f1, f2))
    f1, _ = nets.frechet_case(0, seed)
    assert abs(float(ops.frechet_distance(f1.to(dev), f1.to(dev)).cpu())) < 1e-4      # identical sets: 0


@pytest.mark.parametrize("size", [256, 512])
def test_generator_gradients_serial_schedule_equals_lane_schedule(size, monkeypatch):
    """VTS_PARALLEL_SCALES=0 (one stream: up{i} and up{i}_T accumulate into the split-point gradient by two calls) against the
    two-lane schedule, at the crop sizes whose inner maps take the k-split path (N < 8, maps <= 32x32): the fused InstanceNorm
    backward of the k-split epilogue must not run on a partial split-point gradient (round-3 advisor finding)."""
    from models import create_model
    from options.train_options import TrainOptions
    from vts import engine

    opt = TrainOptions(cmd_line=FLAGS % (size, 1)).parse()
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    G = model.netG
    G.load_state_dict(detrand.test_weights(nets.g_param_shapes(), 77))
    dev = torch.device("cuda:0")
    x = detrand.uniform((1, 9, size, size), 77, "g_in").to(dev)
    cot = detrand.uniform((1, 5, size, size), 77, "g_cot").to(dev)
    grads = {}
    for par in (True, False):
        monkeypatch.setattr(engine, "PARALLEL_SCALES", par)
        for p in G.parameters():
            p.grad.zero_()
        y, ctx = engine.unet_forward(G, x)
        engine.unet_backward(G, ctx, (cot * (1.0 - y * y)).contiguous())
        torch.cuda.synchronize()
        grads[par] = {k: p.grad.clone() for k, p in G.named_parameters()}
    worst = 0.0
    for k in grads[True]:
        if null_grad_bias("G", k):
            continue
        worst = max(worst, rel(grads[False][k], grads[True][k]))
    assert worst < 2e-5, worst


def test_test_phase_lpips_metrics_use_the_alexnet_backbone(tmp_path, monkeypatch):
    """the reference evaluates I_LPIPS / T_LPIPS with lpips.LPIPS(net="alex") in the test phase (models/sinskitG_model.py:501): a model
    built from TestOptions reports that backbone and the checker's AlexNet restatement reproduces both values (stand-in weights)"""
    import torch.nn.functional as F

    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.test_options import TestOptions
    from oracle import perceptual as chk

    monkeypatch.setenv("VTS_LPIPS_METRICS", "1")
    size, seed = 256, 37
    topt = TestOptions(cmd_line="--model sinskitG --gpu_ids 0 --checkpoints_dir %s --name x --crop_size %d" % (tmp_path, size)).parse()
    tm = create_model(topt)
    tm.setup(topt)
    tm.parallelize()
    tm.netG.load_state_dict(detrand.test_weights(nets.g_param_shapes(), seed))
    tm.eval()
    batch = default_collate([make_sample(size, 8, 8, seed)])
    tm.set_input(batch, phase="test")
    tm.test()
    m = tm.compute_metrics()
    assert tm.metric_lpips_backbone == "alex" and tm.metric_lpips_pretrained is False and tm.metric_ssim_pinned == "restatement"
    lp = chk.LPIPS(net="alex")
    with torch.no_grad():
        ref_I = float(lp(tm.real_I.cpu(), tm.fake_I.cpu()).mean())
        pset = tm.val_set
        P = pset["real_T"].shape[0]
        fake_T = torch.empty(P, 2, 32, 32, device=tm.device)
        tm._gather(tm.fake_T, pset, fake_T, 0, channels=2)
        rT, fT = F.interpolate(pset["real_T"].cpu(), (224, 224)), F.interpolate(fake_T.cpu().clamp(0, 1), (224, 224))
        ref_T = float(lp(rT[:, 0:1], fT[:, 0:1]).mean() + lp(rT[:, 1:2], fT[:, 1:2]).mean())
    assert abs(m["I_LPIPS"] - ref_I) <= 1e-3 * ref_I and abs(m["T_LPIPS"] - ref_T) <= 1e-3 * ref_T


def test_graph_replayed_step_with_device_draws_matches_oracle():
    """The execution mode bench.py times -- HIP-graph replay, 'more fake T' ranks drawn on the device (vts_mask_sample_ranks), DiffAugment
    draws from torch.rand inside the captured graph -- against the oracle, not only against the eager path: steps 1-2 run eager / capture
    on the model's own draws, then model and oracle are put in the same state (weights, BatchNorm buffers, Adam moments and step counts
    of an oracle that took two steps), step 3 is a pure replay, and the oracle repeats it with the draws READ BACK from the device."""
    from data.synthetic_dataset import make_sample

    size, nt, seed = 256, 64, 71
    model, opt = make_model(size, 1)
    assert opt.use_hip_graph and model._draws is None
    sds = load_test_weights(model, seed)
    batch = default_collate([make_sample(size, nt, nt, seed)])
    import random

    random.seed(3)
    torch.manual_seed(3)
    cnt = int(nets.dilated_mask_positions(batch["M"].float()).shape[0])
    adam = {k: step.new_adam_state() for k in ("G", "D", "D2")}
    for s in range(2):      # the oracle's two steps (any draws: only the state they leave matters)
        d = {"aug": detrand.uniform((4, 1), seed + s, "aug") * 0.5 + 0.5, "more_idx": torch.tensor([random.sample(range(cnt), 32)])}
        step.train_step(sds[0], sds[1], sds[2], adam, batch, d, record=False)
    for s in range(2):      # the model's: eager, then capture (+ first replay)
        model.set_input(batch, phase="train")
        model.optimize_parameters(epoch=1)
    assert model._graphs is not None
    for nm, net, sd, optim in (("G", model.netG, sds[0], model.optimizer_G), ("D", model.netD, sds[1], model.optimizer_D),
                               ("D2", model.netD2, sds[2], model.optimizer_D2)):
        net.load_state_dict({k: v.detach() for k, v in sd.items()})
        optim.load_named_state(net, adam[nm]["m"], adam[nm]["v"], adam[nm]["step"])
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)          # replay
    torch.cuda.synchronize()
    drawn = {"aug": model._aug.detach().cpu().clone(), "more_idx": model._ranks.detach().cpu().clone()}
    assert drawn["aug"].shape == (4, 1) and float(drawn["aug"].min()) >= 0.0 and float(drawn["aug"].max()) < 1.0
    assert drawn["more_idx"].shape == (1, 32) and len(set(drawn["more_idx"][0].tolist())) == 32 and int(drawn["more_idx"].max()) < cnt
    ref = step.train_step(sds[0], sds[1], sds[2], adam, batch, drawn)
    losses = model.get_current_losses()
    for k, v in ref["losses"].items():
        assert abs(losses["l_" + k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses["l_" + k], v)
    assert rel(model.fake_I, ref["fake_I"]) < 1e-3 and rel(model.fake_T, ref["fake_T"]) < 1e-3
    worst = 0.0
    for nm, net, sd in (("G", model.netG, sds[0]), ("D", model.netD, sds[1]), ("D2", model.netD2, sds[2])):
        for k, p in net.named_parameters():
            if null_grad_bias(nm, k):
                continue
            worst = max(worst, rel(p.grad, ref["grad_" + nm][k]))
            assert rel(p.data, sd[k]) < 1e-3, (nm, k)
    print("graph-mode step: worst gradient rel-L2 vs the oracle %.2e" % worst)
    assert worst < 2e-3, worst


def test_uint8_batch_upload_is_bit_identical_to_the_float_batch():
    """the optional S_u8 / I_u8 / M_u8 batch keys (a quarter of the PCIe bytes): vts_u8_expand reproduces ToTensor [+ Normalize(0.5, 0.5)]
    bit for bit for all 256 levels, and a training step fed from the bytes leaves exactly the weights of a step fed from the float tensors"""
    from data.synthetic_dataset import make_sample
    from vts import ops

    dev = torch.device("cuda:0")
    lv = torch.arange(256, dtype=torch.uint8)
    f = lv.to(torch.float32).div(255)
    assert torch.equal(ops.u8_expand(lv.to(dev), False).cpu(), f) and torch.equal(ops.u8_expand(lv.to(dev), True).cpu(), (f - 0.5) / 0.5)
    batch = default_collate([make_sample(256, 64, 64, 91 + i, quantize8=True) for i in range(2)])
    assert batch["S_u8"].dtype == torch.uint8 and torch.equal((batch["S_u8"].float().div(255) - 0.5) / 0.5, batch["S"])
    plain = {k: v for k, v in batch.items() if not k.endswith("_u8")}
    flats = []
    for b in (batch, plain):
        import random
        random.seed(7)
        torch.manual_seed(7)
        model, opt = make_model(256, 2)
        load_test_weights(model, 91)
        for _ in range(2):
            model.set_input(b, phase="train")
            model.optimize_parameters(epoch=1)
        torch.cuda.synchronize()
        flats.append({n: getattr(model, "flat" + n).flat.cpu().clone() for n in ("G", "D", "D2")})
        assert torch.equal(model.real_I.cpu(), (batch["I"] * batch["M"]))
    for n in ("G", "D", "D2"):
        assert torch.equal(flats[0][n], flats[1][n]), n


def test_fresh_batch_staging_paths_agree_bit_for_bit(monkeypatch):
    """set_input's fast paths -- the one-launch image preparation (vts_input_images_u8) and the patch sets travelling on the copy stream
    into double-buffered staging blocks -- against the plain sequence (u8_expand + mask_mul launches, patch block read on the launch
    stream): six replayed steps over alternating batches leave identical weights and losses (a staging race would show here)"""
    from data.synthetic_dataset import make_sample

    batches = [default_collate([make_sample(256, 64, 64, 191 + 2 * k + i, quantize8=True) for i in range(2)]) for k in range(3)]
    batches = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
    results = []
    for fused, cs in (("1", "1"), ("0", "0"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("VTS_TUNING", "1")        # experiment switches are honoured only under it (vts/tune.py)
        monkeypatch.setenv("VTS_FUSED_INPUT", fused)
        monkeypatch.setenv("VTS_PATCH_COPY_STREAM", cs)
        import random
        random.seed(7)
        torch.manual_seed(7)
        model, opt = make_model(256, 2)
        load_test_weights(model, 91)
        for it in range(6):
            model.set_input(batches[it % 3], phase="train")
            model.optimize_parameters(epoch=1)
        torch.cuda.synchronize()
        results.append(({n: getattr(model, "flat" + n).flat.cpu().clone() for n in ("G", "D", "D2")}, model.get_current_losses(),
                        model.real_I.cpu().clone(), model.train_set["real_T"].cpu().clone()))
        b = batches[5 % 3]
        assert torch.equal(results[-1][2], b["I"] * b["M"])
    for r in results[1:]:
        for n in ("G", "D", "D2"):
            assert torch.equal(results[0][0][n], r[0][n]), n
        assert results[0][1] == r[1]
        assert torch.equal(results[0][3], r[3])


def test_default_vision_aided_flag_before_the_warmup_epoch_matches_reference(golden_dir):
    """The reference's DEFAULT --use_vision_aided_loss True: before --vision_aided_warmup_epoch (100) its step never calls netD3 and logs the
    three D3 entries as 0.0 (tests/golden/sinskitG_d3_warmup_step_256.npz: the reference run with a stand-in netD3 whose forward raises).
    The HIP path accepts the flag, reports the same names in the same order and the same values, and raises once the epoch reaches the
    warm-up epoch (CLIP + the package's head are not built)."""
    from data.synthetic_dataset import make_sample
    from models import create_model
    from options.train_options import TrainOptions

    g = np.load(os.path.join(golden_dir, "sinskitG_d3_warmup_step_256.npz"))
    size, seed, nt = int(g["size"]), int(g["seed"]), int(g["nt"])
    opt = TrainOptions(cmd_line=FLAGS.replace("--use_vision_aided_loss False ", "") % (size, 1)).parse()
    assert opt.use_vision_aided_loss is True and opt.vision_aided_warmup_epoch == int(g["warmup_epoch"])
    model = create_model(opt)
    model.setup(opt)
    model.parallelize()
    model.train()
    load_test_weights(model, seed)
    batch = default_collate([make_sample(size, nt, nt, seed)])
    model._draws = {"aug": torch.from_numpy(g["aug"]), "more_idx": torch.from_numpy(g["more_idx"])}
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    assert list(losses.keys()) == [str(k) for k in g["loss_names"]]
    for k, v in zip(losses.keys(), g["loss_values"]):
        assert abs(losses[k] - v) <= 1e-3 * max(1.0, abs(v)), (k, losses[k], v)
    assert losses["l_G_D3"] == 0.0 and losses["l_D3_real_I"] == 0.0 and losses["l_D3_fake_I"] == 0.0
    probe_close(model.fake_I.contiguous(), g["fake_I_probe"], "fake_I", 1e-3)
    for k, p in model.netG.named_parameters():
        if not null_grad_bias("G", k):
            probe_close(p.data, g["param_G/" + k], k, 1e-3)
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=int(g["warmup_epoch"]) - 1)
    with pytest.raises(NotImplementedError, match="vision_aided_warmup_epoch"):
        model.optimize_parameters(epoch=int(g["warmup_epoch"]))
