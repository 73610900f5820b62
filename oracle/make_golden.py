"""Generate tests/golden/*.npz by RUNNING THE REFERENCE on CPU (build container only).

    python -m oracle.make_golden            # from /root/repo, needs /root/reference

The reference is imported through oracle/ref_import.py (stubs for absent pip
packages; LPIPS / vision-aided / CLIP terms disabled by lambda=0 flags).  Inputs and
weights are regenerated from seeds by committed code (oracle/detrand.py and the
product's synthetic batch generator), so fixtures hold only seeds, configuration,
losses, sub-sampled outputs and per-tensor probes (sum, l2, fixed random projection).
"""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "visual-tactile-synthesis_amd")
GOLD = os.path.join(ROOT, "tests", "golden")


def _synthetic_batch(size, nt, seed, n=1):
    """Collated batch from the product's generator (imported by path to avoid clashing
    with the reference's own `data` package)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("vts_synth", os.path.join(PKG, "data", "synthetic_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules.setdefault("util", __import__("util"))  # reference util has str2bool too
    spec.loader.exec_module(mod)
    from torch.utils.data import default_collate

    return default_collate([mod.make_sample(size, nt, nt, seed + i) for i in range(n)])


def golden_step_sg2d(size=256, seed=909, nt=64):
    """One SinSKITGModel.optimize_parameters of the REFERENCE with --netD stylegan2 (networks.py:437-442): pins the step-level use of the
    StyleGAN2 discriminator (losses, gradients of G / D / D2, outputs)."""
    from oracle import detrand, nets, ref_import, stylegan2 as sg

    ref_import.load()
    from models.sinskitG_model import SinSKITGModel

    flags = ["--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0", "--use_vision_aided_loss", "False", "--lambda_G2_GAN_feat", "0",
             "--checkpoints_dir", "/tmp/vts_golden_ckpt", "--name", "golden_sg2d", "--netD", "stylegan2", "--load_size", str(size),
             "--crop_size", str(size)]
    opt = _ref_opt("sinskitG", True, flags)
    model = SinSKITGModel(opt)
    model.setup(opt)
    shapesD = sg.d_param_shapes(4, opt.ndf, size)
    assert {k: tuple(v.shape) for k, v in model.netD.named_parameters()} == {k: tuple(v) for k, v in shapesD.items()}
    model.netG.load_state_dict(detrand.test_weights(nets.g_param_shapes(), seed))
    model.netD.load_state_dict(sg.test_weights(shapesD, seed + 1), strict=False)
    model.netD2.load_state_dict(detrand.test_weights(nets.d_param_shapes(7), seed + 2))
    model.train()
    batch = _synthetic_batch(size, nt, seed)
    out = {"size": size, "seed": seed, "nt": nt, "ndf": opt.ndf, "flags": json.dumps(flags)}
    model.set_input(batch, phase="train")
    k = int(nets.dilated_mask_positions(model.M).shape[0])
    torch.manual_seed(seed)
    out["aug"] = torch.stack([torch.rand(1, 1, 1, 1).flatten() for _ in range(4)]).numpy()
    random.seed(seed)
    out["more_idx"] = np.array(random.sample(range(k), opt.add_fake_T_sample_size), dtype=np.int64)[None]
    torch.manual_seed(seed)
    random.seed(seed)
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    out["loss_names"] = np.array(list(losses.keys()))
    out["loss_values"] = np.array(list(losses.values()), dtype=np.float64)
    for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
        for kk, p in net.named_parameters():
            out["grad_%s/%s" % (nm, kk)] = detrand.probe(p.grad, kk)
    out["fake_I_sub"] = model.fake_I.detach()[:, :, ::4, ::4].numpy()
    out["pred_fake_I"] = model.pred_fake_I.detach().numpy() if torch.is_tensor(model.pred_fake_I) else np.asarray(model.pred_fake_I[-1].detach())
    np.savez_compressed(os.path.join(GOLD, "sinskitG_sg2d_step_%d.npz" % size), **out)
    print("wrote sinskitG_sg2d_step_%d.npz (%d entries)" % (size, len(out)))
    print({k: float(v) for k, v in losses.items()})


def _ref_opt(model, is_train, extra):
    from options.test_options import TestOptions
    from options.train_options import TrainOptions
    import models

    o = (TrainOptions if is_train else TestOptions)()
    parser = argparse.ArgumentParser()
    parser = o.initialize(parser)
    parser = models.get_option_setter(model)(parser, is_train)
    opt, _ = parser.parse_known_args(extra)
    opt.isTrain = is_train
    opt.gpu_ids = []
    return opt


def probes(sd, tag):
    from oracle import detrand

    return {"%s/%s" % (tag, k): detrand.probe(v, k) for k, v in sd.items() if v.dtype.is_floating_point}


def golden_patchsample():
    """PatchSampleF (with and without its MLP) + PatchNCELoss at the reference's working size: 256 patches x 256 dims, batch 2"""
    from oracle import detrand, ref_import

    ref_import.load()
    from models import networks
    from models.patchnce import PatchNCELoss
    from types import SimpleNamespace

    out = {"meta": np.array("reference PatchSampleF / PatchNCELoss; torch %s" % torch.__version__)}
    feats = [detrand.uniform((2, 24, 20, 18), 41, "f0"), detrand.uniform((2, 40, 9, 11), 41, "f1")]
    ids = [np.random.RandomState(3).permutation(20 * 18)[:256], np.random.RandomState(4).permutation(9 * 11)[:256]]
    out["ids0"], out["ids1"] = ids[0], ids[1]
    plain = networks.PatchSampleF(use_mlp=False, gpu_ids=[])
    fo, _ = plain(feats, 256, ids)
    out["plain0"], out["plain1"] = fo[0].numpy(), fo[1].numpy()
    torch.manual_seed(17)
    mlp = networks.PatchSampleF(use_mlp=True, init_type="normal", init_gain=0.02, nc=256, gpu_ids=[])
    with torch.no_grad():
        fm, _ = mlp(feats, 256, ids)
    for i in range(2):
        m = getattr(mlp, "mlp_%d" % i)
        # the reference initialises N(0, 0.02): stored so that the build loads exactly these weights
        out["mlp%d_w0" % i], out["mlp%d_b0" % i] = m[0].weight.detach().numpy(), m[0].bias.detach().numpy()
        out["mlp%d_w2" % i], out["mlp%d_b2" % i] = m[2].weight.detach().numpy().astype(np.float16), m[2].bias.detach().numpy()
        m[2].weight.data = torch.from_numpy(out["mlp%d_w2" % i].astype(np.float32))     # fp16-representable second layer: small fixture
    with torch.no_grad():
        fm, _ = mlp(feats, 256, ids)
    out["mlp0_sub"], out["mlp1_sub"] = fm[0][::8].numpy(), fm[1][::8].numpy()
    out["mlp_keys"] = np.array(sorted(mlp.state_dict().keys()))
    # PatchNCE at 2 x 256 patches x 256 dims on the MLP features of map 0 (q) and a second sampler call on other features (k)
    fk, _ = mlp([detrand.uniform((2, 24, 20, 18), 43, "k0"), feats[1]], 256, ids)
    q, k = fm[0].detach().clone().requires_grad_(True), fk[0].detach()
    for allneg in (False, True):
        nce = PatchNCELoss(SimpleNamespace(nce_includes_all_negatives_from_minibatch=allneg, batch_size=2, nce_T=0.07))
        loss = nce(q, k)
        g, = torch.autograd.grad(loss.sum(), q)
        out["nce_loss_%d" % allneg] = loss.detach().numpy()
        out["nce_dq_sub_%d" % allneg] = g[::8, ::4].numpy()
        out["nce_dq_probe_%d" % allneg] = np.array(detrand.probe(g, "dq"))
    np.savez_compressed(os.path.join(GOLD, "patchsample.npz"), **out)
    print("wrote patchsample.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


def golden_patchsample_whole():
    """PatchSampleF with num_patches = 0 (the whole map; its Normalize then runs over the positions): with and without the MLP"""
    from oracle import detrand, ref_import

    ref_import.load()
    from models import networks

    out = {}
    feats = [detrand.uniform((2, 6, 5, 7), 51, "f0"), detrand.uniform((3, 10, 4, 4), 51, "f1")]
    fo, ids = networks.PatchSampleF(use_mlp=False, gpu_ids=[])(feats, 0, None)
    assert ids == [[], []]
    out["plain0"], out["plain1"] = fo[0].numpy(), fo[1].numpy()
    torch.manual_seed(19)
    mlp = networks.PatchSampleF(use_mlp=True, init_type="normal", init_gain=0.02, nc=12, gpu_ids=[])
    with torch.no_grad():
        fm, _ = mlp(feats, 0, None)
    for i in range(2):
        m = getattr(mlp, "mlp_%d" % i)
        out["mlp%d_w0" % i], out["mlp%d_b0" % i] = m[0].weight.detach().numpy(), m[0].bias.detach().numpy()
        out["mlp%d_w2" % i], out["mlp%d_b2" % i] = m[2].weight.detach().numpy(), m[2].bias.detach().numpy()
    out["mlp0"], out["mlp1"] = fm[0].numpy(), fm[1].numpy()
    np.savez_compressed(os.path.join(GOLD, "patchsample_whole.npz"), **out)
    print("wrote patchsample_whole.npz", {k: v.shape for k, v in out.items()})


def golden_ops():
    """Operator-level vectors: SPE, DiffAugment, GANLoss (all modes), PatchNCE, patch gather, normals."""
    from oracle import detrand, ref_import

    ref_import.load()
    from models import networks
    from models.model_utils import compute_normal, get_patch_in_input
    from models.patchnce import PatchNCELoss
    from thirdparty.DiffAugment import DiffAugment
    from thirdparty.mmgeneration.positional_encoding import SinusoidalPositionalEmbedding as SPE
    from types import SimpleNamespace

    out = {}
    x = torch.zeros(2, 1, 24, 40)
    out["spe_24x40"] = SPE(4, 0, 1024)(x).numpy()
    big = SPE(4, 0, 1024)(torch.zeros(1, 1, 1100, 1030))  # exceeds init_size: table is rebuilt
    out["spe_1100x1030_sub"] = big[:, :, ::50, ::47].numpy()

    img = detrand.uniform((2, 3, 20, 28), 11, "diffaug")
    torch.manual_seed(5)
    out["diffaug_out"] = DiffAugment(img, policy="bs").numpy()
    torch.manual_seed(5)
    out["diffaug_draws"] = torch.stack([torch.rand(2, 1, 1, 1).flatten() for _ in range(2)]).numpy()

    preds = [[detrand.uniform((3, 1, 9, 9), 3, "p0") * 3], [detrand.uniform((3, 1, 5, 5), 3, "p1") * 3]]
    for mode in ["nonsaturating", "lsgan", "vanilla", "wgan", "hinge"]:
        crit = networks.GANLoss(mode, target_real_label=0.8, target_fake_label=0.0)
        for real in (True, False):
            out["gan_%s_%d" % (mode, real)] = np.atleast_1d(crit(preds, real).detach().numpy())

    fq = detrand.uniform((2 * 16, 24), 21, "fq")
    fk = detrand.uniform((2 * 16, 24), 21, "fk")
    fq = fq / fq.norm(dim=1, keepdim=True)
    fk = fk / fk.norm(dim=1, keepdim=True)
    for allneg in (False, True):
        nce = PatchNCELoss(SimpleNamespace(nce_includes_all_negatives_from_minibatch=allneg, batch_size=2, nce_T=0.07))
        out["patchnce_%d" % allneg] = nce(fq, fk).numpy()
    out["normalize"] = networks.Normalize(2)(detrand.uniform((5, 7), 2, "nrm")).numpy()

    im = detrand.uniform((1, 3, 96, 80), 31, "gather")
    coords = np.zeros((1, 6, 8))
    coords[0, :, 0] = [0, 10, 60, 70, 33, 5]   # x (70+32 > 80: clamps at the border)
    coords[0, :, 1] = [0, 20, 80, 5, 64, 90]   # y (80+32 > 96, 90+32 > 96)
    coords[0, :, 2:4] = 40
    coords[0, :, 4] = 32
    coords[0, :, 5] = 1.0
    coords[0, :, 6] = [0, 3, 7, 1, 2, 4]
    coords[0, :, 7] = [5, 0, 2, 6, 1, 3]
    out["gather_coords"] = coords
    out["gather_out"] = get_patch_in_input(im, coords).numpy()
    out["normal_out"] = compute_normal(im[:, :2], scale_nz=0.25).numpy()

    # random ("more fake T") mode: dilated-mask positions + sampled offsets
    M = torch.zeros(1, 1, 64, 72)
    M[0, 0, 20:40, 25:50] = 1
    random.seed(9)
    samples, ox, oy, cs = get_patch_in_input(im[:, :2, :64, :72].contiguous(), coords=None, sample_size=5,
                                             return_offset=True, M=M, center_h=None, center_w=None)
    out["more_M"] = M.numpy()
    out["more_samples"] = samples.numpy()
    out["more_ox"] = ox.numpy().reshape(-1)
    out["more_oy"] = oy.numpy().reshape(-1)
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)
    print("wrote ops.npz", {k: v.shape for k, v in out.items()})


def golden_nets(size=256, seed=101):
    """G and D forward + input-gradient vectors from the reference modules with seeded test weights."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models import networks

    opt = _ref_opt("sinskitG", True, [])
    out = {"size": size, "seed": seed}
    G = networks.define_G(9, 5, 10, "unet256_custom", "instance", False, "xavier", 0.02, False, False, [], opt,
                          num_layer_separate=4)
    ref_keys = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    mine = nets.g_param_shapes()
    assert ref_keys == {k: tuple(v) for k, v in mine.items()}, "G key/shape mismatch"
    out["G_init_std"] = np.array([G.state_dict()["down3.model.1.weight"].std().item(),
                                  G.state_dict()["up3.model.1.weight"].std().item()])
    G.load_state_dict(detrand.test_weights(mine, seed))
    x = detrand.uniform((1, 9, size, size), seed, "g_in").requires_grad_(True)
    y = G(x)
    (y * detrand.uniform(tuple(y.shape), seed, "g_cot")).sum().backward()
    out["G_out_sub"] = y.detach()[:, :, ::4, ::4].numpy()
    out["G_out_probe"] = detrand.probe(y, "g_out")
    out["G_dx_probe"] = detrand.probe(x.grad, "g_dx")
    for k, p in G.named_parameters():
        out["G_grad/" + k] = detrand.probe(p.grad, k)

    for name, cin, n, hw in (("D", 4, 1, size), ("D2", 7, 6, 32)):
        D = networks.define_D(cin, 8, "multiscale", 3, "batch", "xavier", 0.02, False, num_D=3, gpu_ids=[], opt=opt)
        shapes = nets.d_param_shapes(cin)
        assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
        D.load_state_dict(detrand.test_weights(shapes, seed + 1))
        D.train()
        x = detrand.uniform((n, cin, hw, hw), seed, name + "_in").requires_grad_(True)
        preds = D(x)
        tot = 0
        for s, p in enumerate(preds):
            out["%s_pred%d" % (name, s)] = p[-1].detach().numpy()
            tot = tot + (p[-1] * detrand.uniform(tuple(p[-1].shape), seed, "%s_cot%d" % (name, s))).sum()
        tot.backward()
        out[name + "_dx_probe"] = detrand.probe(x.grad, name + "_dx")
        for k, p in D.named_parameters():
            out["%s_grad/%s" % (name, k)] = detrand.probe(p.grad, k)
        for k, b in D.named_buffers():
            if b.dtype.is_floating_point:
                out["%s_buf/%s" % (name, k)] = b.numpy()
    np.savez_compressed(os.path.join(GOLD, "nets_%d.npz" % size), **out)
    print("wrote nets_%d.npz (%d entries)" % (size, len(out)))


def golden_nets_style(size=256, seed=111, n=2):
    """skitG generator: CustomUnetGenerator with a style code (use_style_code, style_code_mode concat, mapping tile; reference
    networks.py:1436-1466, 1595-1640) -- the network of the headline configuration.  Forward + gradients (incl. d/d style_code)."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models import networks

    opt = _ref_opt("skitG", True, ["--use_style_code", "True", "--batch_size", str(n)])
    out = {"size": size, "seed": seed, "n": n, "style_code_dim": opt.style_code_dim, "num_layer_style_code": opt.num_layer_style_code}
    G = networks.define_G(9, 5, 10, "unet256_custom", "instance", False, "xavier", 0.02, False, False, [], opt, num_layer_separate=4)
    ref = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    mine = nets.g_param_shapes(style_nc=opt.style_code_dim, num_layer_style_code=opt.num_layer_style_code)
    dead = {k: v for k, v in ref.items() if k.startswith("style_code_mapping")}      # created upstream, unused in tile mode
    assert {k: v for k, v in ref.items() if k not in dead} == {k: tuple(v) for k, v in mine.items()}, "style G key/shape mismatch"
    out["dead_keys"] = np.array(sorted(dead.keys()))
    G.load_state_dict(detrand.test_weights(mine, seed), strict=False)
    x = detrand.uniform((n, 9, size, size), seed, "g_in").requires_grad_(True)
    sc = detrand.uniform((n, opt.style_code_dim), seed, "style")
    sc = (sc / sc.norm(dim=1, keepdim=True)).requires_grad_(True)
    y = G(x, style_code=sc)
    (y * detrand.uniform(tuple(y.shape), seed, "g_cot")).sum().backward()
    out["G_out_sub"] = y.detach()[:, :, ::4, ::4].numpy()
    out["G_out_probe"] = detrand.probe(y, "g_out")
    out["G_dx_probe"] = detrand.probe(x.grad, "g_dx")
    out["G_dstyle"] = sc.grad.numpy()
    for k, p in G.named_parameters():
        if k not in dead:
            out["G_grad/" + k] = detrand.probe(p.grad, k)
    np.savez_compressed(os.path.join(GOLD, "nets_style_%d.npz" % size), **out)
    print("wrote nets_style_%d.npz (%d entries)" % (size, len(out)))


STYLE_MODE_CASES = (("concat", "project", 2), ("adain", "project", 1))     # (style_code_mode, style_code_mapping_mode, batch)


def style_mode_shapes(mode, n):
    from oracle import nets
    nc = 80 if mode == "adain" else 5      # ngf * 8 | ngf // 2 at ngf = 10 (networks.py:1446-1457)
    return nets.g_param_shapes(style_nc=0 if mode == "adain" else nc, num_layer_style_code=1, style_map_nc=nc, style_bn=n > 1)


def golden_nets_style_modes(size=1536, seed=117):
    """CustomUnetGenerator with the projected style code: style_code_mapping0 (Linear -> BatchNorm1d at batch 2 | InstanceNorm1d at
    batch 1 -> ReLU) concatenated into up7, and the adain mode (thirdparty/AdaIN) -- reference networks.py:1444-1465, 1608-1632.  The
    reference builds the mapping for a 1536-pixel input (input_size=1536, :1432), so this runs at 1536 x 1536."""
    from oracle import detrand, ref_import

    ref_import.load()
    from models import networks

    out = {"size": size, "seed": seed}
    for mode, mapping, n in STYLE_MODE_CASES:
        opt = _ref_opt("skitG", True, ["--use_style_code", "True", "--batch_size", str(n), "--style_code_mode", mode, "--style_code_mapping_mode", mapping])
        G = networks.define_G(9, 5, 10, "unet256_custom", "instance", False, "xavier", 0.02, False, False, [], opt, num_layer_separate=4)
        mine = style_mode_shapes(mode, n)
        assert {k: tuple(v.shape) for k, v in G.named_parameters()} == {k: tuple(v) for k, v in mine.items()}, "style-mode G key/shape mismatch"
        G.load_state_dict(detrand.test_weights(mine, seed), strict=False)
        G.train()
        x = detrand.uniform((n, 9, size, size), seed, "g_in").requires_grad_(True)
        sc = detrand.uniform((n, opt.style_code_dim), seed, "style")
        sc = (sc / sc.norm(dim=1, keepdim=True)).requires_grad_(True)
        y = G(x, style_code=sc)
        (y * detrand.uniform(tuple(y.shape), seed, "g_cot")).sum().backward()
        t = "%s/" % mode
        out[t + "keys"] = np.array(sorted(G.state_dict().keys()))
        out[t + "G_out_sub"] = y.detach()[:, :, ::16, ::16].numpy()
        out[t + "G_out_probe"] = detrand.probe(y, "g_out")
        out[t + "G_dx_probe"] = detrand.probe(x.grad, "g_dx")
        out[t + "G_dstyle"] = sc.grad.numpy()
        for k, p in G.named_parameters():
            out[t + "G_grad/" + k] = detrand.probe(p.grad, k)
        if n > 1:
            out[t + "bn_running_mean"] = G.state_dict()["style_code_mapping0.1.running_mean"].numpy()
            out[t + "bn_running_var"] = G.state_dict()["style_code_mapping0.1.running_var"].numpy()
    np.savez_compressed(os.path.join(GOLD, "nets_style_modes_%d.npz" % size), **out)
    print("wrote nets_style_modes_%d.npz (%d entries)" % (size, len(out)))


def golden_resnet(size=64, seed=303, n_blocks=9, ngf=10):
    """ResnetGenerator (--netG resnet_9blocks, reference defaults) forward + gradients with seeded test weights."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models import networks

    opt = _ref_opt("sinskitG", True, [])
    out = {"size": size, "seed": seed, "n_blocks": n_blocks, "ngf": ngf}
    G = networks.define_G(9, 5, ngf, "resnet_%dblocks" % n_blocks, "instance", False, "xavier", 0.02, False, False, [], opt)
    ref = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    mine = nets.resnet_param_shapes(9, 5, ngf, n_blocks)
    learn = {k: v for k, v in ref.items() if not k.endswith(".filt")}
    assert learn == {k: tuple(v) for k, v in mine.items()}, "resnet G key/shape mismatch"
    out["ref_keys"] = np.array(sorted(ref.keys()))
    out["ref_filt_down"] = G.state_dict()["model.7.filt"][0, 0].numpy()
    out["ref_filt_up"] = G.state_dict()["model.%d.filt" % (12 + n_blocks)][0, 0].numpy()
    G.load_state_dict(detrand.test_weights(mine, seed), strict=False)
    x = detrand.uniform((2, 9, size, size), seed, "g_in").requires_grad_(True)
    y = G(x)
    (y * detrand.uniform(tuple(y.shape), seed, "g_cot")).sum().backward()
    out["G_out"] = y.detach().numpy()
    out["G_dx_probe"] = detrand.probe(x.grad, "g_dx")
    for k, p in G.named_parameters():
        out["G_grad/" + k] = detrand.probe(p.grad, k)
    np.savez_compressed(os.path.join(GOLD, "resnet_%d.npz" % size), **out)
    print("wrote resnet_%d.npz (%d entries)" % (size, len(out)))


def golden_global(h=64, w=32, seed=404, ngf=8, n_down=3, n_blocks=3):
    """pix2pixHD GlobalGenerator (define_G netG='global', BatchNorm, train mode): forward + gradients + BN buffers."""
    import copy

    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models import networks

    opt = copy.copy(_ref_opt("sinskitG", True, []))
    opt.n_downsample_global, opt.n_blocks_global = n_down, n_blocks
    out = {"h": h, "w": w, "seed": seed, "ngf": ngf, "n_down": n_down, "n_blocks": n_blocks}
    G = networks.define_G(1, 5, ngf, "global", "batch", False, "xavier", 0.02, False, False, [], opt)
    ref = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    mine = nets.resnet_param_shapes(1, 5, ngf, n_blocks, n_down, norm="batch", down="stride", up="convT", conv_bias=True)
    assert ref == {k: tuple(v) for k, v in mine.items()}, "global G key/shape mismatch"
    out["ref_keys"] = np.array(sorted(ref.keys()))
    G.load_state_dict(detrand.test_weights(mine, seed))
    G.train()
    x = detrand.uniform((2, 1, h, w), seed, "g_in").requires_grad_(True)
    y = G(x)
    (y * detrand.uniform(tuple(y.shape), seed, "g_cot")).sum().backward()
    out["G_out"] = y.detach().numpy()
    out["G_dx_probe"] = detrand.probe(x.grad, "g_dx")
    for k, p in G.named_parameters():
        out["G_grad/" + k] = detrand.probe(p.grad, k)
    for k, b in G.named_buffers():
        if b.dtype.is_floating_point:
            out["G_buf/" + k] = b.numpy()
    np.savez_compressed(os.path.join(GOLD, "global_%dx%d.npz" % (h, w)), **out)
    print("wrote global_%dx%d.npz (%d entries)" % (h, w, len(out)))


def golden_local(h=64, w=32, seed=707, ngf=4, n_down=2, n_blocks_global=2, n_blocks_local=2, n_local=1):
    """pix2pixHD LocalEnhancer (define_G netG='local', BatchNorm, train mode): forward + gradients + BN buffers."""
    import copy

    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models import networks

    opt = copy.copy(_ref_opt("sinskitG", True, []))
    opt.n_downsample_global, opt.n_blocks_global, opt.n_local_enhancers, opt.n_blocks_local = n_down, n_blocks_global, n_local, n_blocks_local
    out = {"h": h, "w": w, "seed": seed, "ngf": ngf, "n_down": n_down, "n_blocks_global": n_blocks_global, "n_blocks_local": n_blocks_local,
           "n_local": n_local}
    G = networks.define_G(1, 5, ngf, "local", "batch", False, "xavier", 0.02, False, False, [], opt)
    ref = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    mine = nets.local_enhancer_param_shapes(1, 5, ngf, n_down, n_blocks_global, n_blocks_local, n_local)
    assert ref == {k: tuple(v) for k, v in mine.items()}, sorted(set(ref) ^ set(mine))[:8]
    out["ref_keys"] = np.array(sorted(ref.keys()))
    G.load_state_dict(detrand.test_weights(mine, seed))
    G.train()
    x = detrand.uniform((2, 1, h, w), seed, "g_in").requires_grad_(True)
    y = G(x)
    (y * detrand.uniform(tuple(y.shape), seed, "g_cot")).sum().backward()
    out["G_out"] = y.detach().numpy()
    for k, p in G.named_parameters():
        out["G_grad/" + k] = detrand.probe(p.grad, k)
    for k, b in G.named_buffers():
        if b.dtype.is_floating_point:
            out["G_buf/" + k] = b.numpy()
    fname = ("local_%dx%d.npz" if n_local == 1 else "local%d_%%dx%%d.npz" % n_local) % (h, w)
    np.savez_compressed(os.path.join(GOLD, fname), **out)
    print("wrote %s (%d entries)" % (fname, len(out)))


def p2p_batch(n, size, seed):
    """synthetic patch batch with the patchskit contract (data/patchskit_dataset.py:277-333; return_patch=True)"""
    from oracle import detrand
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    M = (((yy - size / 2) / (0.45 * size)) ** 2 + ((xx - size / 2) / (0.4 * size)) ** 2 <= 1).float()[None, None].repeat(n, 1, 1, 1)
    return {"S_images": detrand.uniform((n, 1, size, size), seed, "S"), "M_images": M,
            "I_images": detrand.uniform((n, 3, size, size), seed, "I"), "T_images": 0.3 * detrand.uniform((n, 2, size, size), seed, "T"),
            "I_masks": torch.ones(n, size, size, dtype=torch.float64), "name": ["synthetic"] * n, "S_paths": ["synthetic.png"] * n,
            "augmentation_params": {}}


P2P_FLAGS = ["--model", "pix2pixHD", "--no_vgg_loss", "True", "--ngf", "8", "--ndf", "8", "--n_downsample_global", "3", "--n_blocks_global", "2", "--batch_size", "4"]


def golden_p2p_step(size=32, seed=505, n=4, steps=2, extra=(), fname="pix2pixHD_step_%d.npz", n_layers=3):
    """Pix2PixHDModel.optimize_parameters x `steps` on one synthetic patch batch (small G / D so the fixture stays small)."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    import models
    opt = _ref_opt("pix2pixHD", True, P2P_FLAGS + list(extra))
    opt.checkpoints_dir, opt.name = "/tmp/vts_golden_ckpt", "p2p"
    os.makedirs(os.path.join(opt.checkpoints_dir, opt.name), exist_ok=True)
    model = models.create_model(opt)
    model.setup(opt)
    shG = nets.resnet_param_shapes(1, 5, 8, 2, 3, norm="batch", down="stride", up="convT", conv_bias=True)
    shD, shD2 = nets.d_if_param_shapes(4, 8, 2, n_layers), nets.d_if_param_shapes(3, 8, 2, n_layers)
    for net, sh in ((model.netG, shG), (model.netD, shD), (model.netD2, shD2)):
        ref = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert ref == {k: tuple(v) for k, v in sh.items()}, (sorted(set(ref) ^ set(sh))[:6])
    model.netG.load_state_dict(detrand.test_weights(shG, seed))
    model.netD.load_state_dict(detrand.test_weights(shD, seed + 1))
    model.netD2.load_state_dict(detrand.test_weights(shD2, seed + 2))
    model.train()
    batch = p2p_batch(n, size, seed)
    out = {"size": size, "seed": seed, "n": n, "steps": steps, "flags": np.array(P2P_FLAGS + list(extra)), "n_layers_D": n_layers,
           "lr": opt.lr, "beta1": opt.beta1, "gan_mode": np.array(opt.gan_mode)}
    for it in range(steps):
        model.set_input(batch, phase="train")
        model.optimize_parameters(epoch=1)
        losses = model.get_current_losses()
        tag = "s%d" % it
        out[tag + "/loss_names"] = np.array(list(losses.keys()))
        out[tag + "/loss_values"] = np.array([float(v) for v in losses.values()], dtype=np.float64)
        for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
            for kk, p in net.named_parameters():
                out["%s/grad_%s/%s" % (tag, nm, kk)] = detrand.probe(p.grad, kk)
                out["%s/param_%s/%s" % (tag, nm, kk)] = detrand.probe(p, kk)
            for kk, b in net.named_buffers():
                out["%s/buf_%s/%s" % (tag, nm, kk)] = b.detach().double().numpy()
        out[tag + "/fake_I"] = model.fake_I.detach().numpy()
        out[tag + "/fake_T"] = model.fake_T.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, fname % size), **out)
    print("wrote " + fname % size + " (%d entries)" % len(out))
    print({k: float(v) for k, v in losses.items()})


def golden_image_pool(seed=616, pool_size=3, n=4, batches=6):
    """util/image_pool.py:ImagePool.query of the reference on `batches` batches of n constant images (image i of batch b is filled with
    the id b * n + i): the ids it returns, under random.seed(seed)."""
    import random

    from oracle import ref_import

    ref_import.load()
    from util.image_pool import ImagePool
    pool = ImagePool(pool_size)
    random.seed(seed)
    ids = []
    for b in range(batches):
        imgs = torch.stack([torch.full((2, 3, 5), float(b * n + i)) for i in range(n)])
        out = pool.query(imgs)
        assert all(float(o.min()) == float(o.max()) for o in out)
        ids.append([int(o[0, 0, 0]) for o in out])
    np.savez_compressed(os.path.join(GOLD, "image_pool.npz"), seed=seed, pool_size=pool_size, n=n, batches=batches, returned=np.array(ids))
    print("wrote image_pool.npz", ids)


def _p2p_model(extra, g_shapes, seed, n_layers=3, override=None):
    """override: option attributes set after parsing (the reference's parser rejects --netG local / global -- `choices` lists neither,
    options/base_options.py; 'global' arrives through set_defaults)"""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    import models
    opt = _ref_opt("pix2pixHD", True, P2P_FLAGS + list(extra))
    for k, v in (override or {}).items():
        setattr(opt, k, v)
    opt.checkpoints_dir, opt.name = "/tmp/vts_golden_ckpt", "p2p"
    os.makedirs(os.path.join(opt.checkpoints_dir, opt.name), exist_ok=True)
    model = models.create_model(opt)
    model.setup(opt)
    shD, shD2 = nets.d_if_param_shapes(4, 8, 2, n_layers), nets.d_if_param_shapes(3, 8, 2, n_layers)
    for net, sh in ((model.netG, g_shapes), (model.netD, shD), (model.netD2, shD2)):
        ref = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert ref == {k: tuple(v) for k, v in sh.items()}, (sorted(set(ref) ^ set(sh))[:6])
    model.netG.load_state_dict(detrand.test_weights(g_shapes, seed))
    model.netD.load_state_dict(detrand.test_weights(shD, seed + 1))
    model.netD2.load_state_dict(detrand.test_weights(shD2, seed + 2))
    model.train()
    return model, opt


def golden_p2p_pool_step(size=32, seed=535, n=4, steps=2, pool_size=3, rseed=548):
    """Pix2PixHDModel.optimize_parameters x 2 with --pool_size 3 (fake_pool.query in backward_D, pix2pixHD_model.py:582, 626) under
    random.seed(rseed): the first batch fills the pool and its fourth image draws (548: it swaps with slot 0), the second batch draws four
    times (548: four swaps, the last with the slot the second image of the same batch has just written)."""
    import random

    from oracle import nets
    extra = ["--pool_size", str(pool_size)]
    shG = nets.resnet_param_shapes(1, 5, 8, 2, 3, norm="batch", down="stride", up="convT", conv_bias=True)
    model, opt = _p2p_model(extra, shG, seed)
    batch = p2p_batch(n, size, seed)
    out = {"size": size, "seed": seed, "rseed": rseed, "n": n, "steps": steps, "pool_size": pool_size, "flags": np.array(P2P_FLAGS + extra)}
    random.seed(rseed)
    state = random.getstate()
    plan = []
    for it in range(steps):
        model.set_input(batch, phase="train")
        model.optimize_parameters(epoch=1)
        losses = model.get_current_losses()
        out["s%d/loss_names" % it] = np.array(list(losses.keys()))
        out["s%d/loss_values" % it] = np.array([float(v) for v in losses.values()], dtype=np.float64)
        out["s%d/fake_I" % it] = model.fake_I.detach().numpy()
        for kk, p in model.netD.named_parameters():
            out["s%d/grad_D/%s" % (it, kk)] = p.grad.detach().numpy()
    # the decisions the pool made (replayed from the same generator state: the step makes no other use of `random`)
    random.setstate(state)
    num = 0
    for it in range(steps):
        for i in range(n):
            if num < pool_size:
                plan.append((-1, num))
                num += 1
            elif random.uniform(0, 1) > 0.5:
                k = random.randint(0, pool_size - 1)
                plan.append((k, k))
            else:
                plan.append((-1, -1))
    out["plan"] = np.array(plan).reshape(steps, n, 2)
    np.savez_compressed(os.path.join(GOLD, "pix2pixHD_pool_step_%d.npz" % size), **out)
    print("wrote pix2pixHD_pool_step_%d.npz" % size, out["plan"].tolist(), {k: float(v) for k, v in losses.items()})


def golden_p2p_fix_global(size=32, seed=545, n=4):
    """--netG local --niter_fix_global 1 (pix2pixHD_model.py:403-421, 942-949; train.py:209-211): one step with optimizer_G over the local
    enhancer only, update_learning_rate + update_fixed_params as train.py calls them at the end of epoch 1, one more step."""
    from oracle import nets
    extra = ["--ngf", "4", "--n_downsample_global", "2", "--n_blocks_local", "2", "--niter_fix_global", "1"]
    shG = nets.local_enhancer_param_shapes(1, 5, 4, 2, 2, 2)
    model, opt = _p2p_model(extra, shG, seed, override={"netG": "local"})
    batch = p2p_batch(n, size, seed)
    keys = ["model.4.weight", "model.2.weight", "model1_1.4.weight", "model1_2.0.conv_block.1.weight", "model1_2.6.weight", "model1_2.6.bias"]
    named = dict(model.netG.named_parameters())
    assert all(k in named for k in keys), [k for k in keys if k not in named]
    out = {"size": size, "seed": seed, "n": n, "flags": np.array(P2P_FLAGS + extra), "keys": np.array(keys), "lr": opt.lr,
           "niter_decay": opt.niter_decay}
    for k in keys:
        out["init/" + k] = named[k].detach().numpy().copy()
    for it in range(2):
        model.set_input(batch, phase="train")
        model.optimize_parameters(epoch=1)
        losses = model.get_current_losses()
        out["s%d/loss_names" % it] = np.array(list(losses.keys()))
        out["s%d/loss_values" % it] = np.array([float(v) for v in losses.values()], dtype=np.float64)
        for k in keys:
            out["s%d/param/%s" % (it, k)] = named[k].detach().numpy().copy()
        if it == 0:
            model.update_learning_rate()
            model.update_fixed_params()
            out["lr_after"] = float(model.optimizer_D.param_groups[0]["lr"])
            out["lr_G_after"] = float(model.optimizer_G.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(GOLD, "pix2pixHD_fix_global_%d.npz" % size), **out)
    print("wrote pix2pixHD_fix_global_%d.npz" % size, out["lr_after"], out["lr_G_after"], {k: float(v) for k, v in losses.items()})


def golden_lpips_step(size=256, seed=212, nt=64):
    """One SinSKITGModel.optimize_parameters of the REFERENCE with the LPIPS terms ON (the reference's default flags:
    lambda_G1_lpips 1, lambda_G2_lpips 10).  The `lpips` package is third-party and absent here: `lpips.LPIPS` is replaced by the
    restatement of its published algorithm (oracle/perceptual.py) on seeded stand-in weights, so what this fixture pins is the
    reference's own call sites and reductions around the network (sinskitG_model.py:1709-1716, 1619-1658, 1819-1838) and their
    gradient path into the generator."""
    from oracle import detrand, nets, perceptual, ref_import

    ref_import.load()
    sys.modules["lpips"].LPIPS = perceptual.LPIPS
    import models.sinskitG_model as ref_mod
    ref_mod.lpips.LPIPS = perceptual.LPIPS
    flags = ["--lambda_G1_lpips", "1", "--lambda_G2_lpips", "10", "--use_vision_aided_loss", "False",
             "--lambda_G2_GAN_feat", "0", "--checkpoints_dir", "/tmp/vts_golden_ckpt", "--name", "golden_lpips"]
    opt = _ref_opt("sinskitG", True, flags)
    model = ref_mod.SinSKITGModel(opt)
    model.setup(opt)
    model.netG.load_state_dict(detrand.test_weights(nets.g_param_shapes(), seed))
    model.netD.load_state_dict(detrand.test_weights(nets.d_param_shapes(4), seed + 1))
    model.netD2.load_state_dict(detrand.test_weights(nets.d_param_shapes(7), seed + 2))
    model.train()
    batch = _synthetic_batch(size, nt, seed)
    out = {"size": size, "seed": seed, "nt": nt, "flags": json.dumps(flags)}
    model.set_input(batch, phase="train")
    k = int(nets.dilated_mask_positions(model.M).shape[0])
    torch.manual_seed(seed)
    aug = torch.stack([torch.rand(1, 1, 1, 1).flatten() for _ in range(4)])
    random.seed(seed)
    more = np.array(random.sample(range(k), opt.add_fake_T_sample_size), dtype=np.int64)[None]
    torch.manual_seed(seed)
    random.seed(seed)
    model.optimize_parameters(epoch=1)
    out["s0/aug"], out["s0/more_idx"] = aug.numpy(), more
    losses = model.get_current_losses()
    out["s0/loss_names"] = np.array(list(losses.keys()))
    out["s0/loss_values"] = np.array(list(losses.values()), dtype=np.float64)
    for kk, p in model.netG.named_parameters():
        out["s0/grad_G/%s" % kk] = detrand.probe(p.grad, kk)
    out["s0/fake_I_probe"] = detrand.probe(model.fake_I, "fake_I")
    # the module alone on seeded inputs: values per sample (3-channel and 1-channel inputs) and the input gradient
    lp = perceptual.LPIPS()
    a = (detrand.uniform((2, 3, 64, 64), seed, "lp_a")).requires_grad_(True)
    b = detrand.uniform((2, 3, 64, 64), seed, "lp_b")
    v = lp(a, b)
    v.sum().backward()
    out["module/val3"] = v.detach().flatten().double().numpy()
    out["module/grad3_probe"] = detrand.probe(a.grad, "lp_ga")
    a1 = (0.3 * detrand.uniform((5, 1, 32, 32), seed, "lp_a1")).requires_grad_(True)
    b1 = 0.3 * detrand.uniform((5, 1, 32, 32), seed, "lp_b1")
    v1 = lp(a1, b1)
    v1.sum().backward()
    out["module/val1"] = v1.detach().flatten().double().numpy()
    out["module/grad1_probe"] = detrand.probe(a1.grad, "lp_ga1")
    np.savez_compressed(os.path.join(GOLD, "sinskitG_lpips_step_%d.npz" % size), **out)
    print("wrote sinskitG_lpips_step_%d.npz" % size, {k: float(v) for k, v in losses.items()})


def golden_p2p_vgg_step(size=32, seed=515, n=4):
    """One Pix2PixHDModel.optimize_parameters of the REFERENCE with its VGG feature term ON (lambda_vgg 10, the default).  torchvision
    is absent: `networks.VGGLoss` is replaced by the restatement of the reference's own VGGLoss / Vgg19 (oracle/perceptual.py,
    networks.py:2021-2067) on seeded stand-in weights; pinned are the call sites (pix2pixHD_model.py:680-693) and the gradient into G."""
    from oracle import detrand, nets, perceptual, ref_import

    ref_import.load()
    import models
    from models import networks as ref_networks
    ref_networks.VGGLoss = lambda gpu_ids=None: perceptual.VGGLoss()
    flags = [f for f in P2P_FLAGS]
    flags[flags.index("--no_vgg_loss") + 1] = "False"
    opt = _ref_opt("pix2pixHD", True, flags)
    opt.checkpoints_dir, opt.name = "/tmp/vts_golden_ckpt", "p2p_vgg"
    os.makedirs(os.path.join(opt.checkpoints_dir, opt.name), exist_ok=True)
    model = models.create_model(opt)
    model.setup(opt)
    shG = nets.resnet_param_shapes(1, 5, 8, 2, 3, norm="batch", down="stride", up="convT", conv_bias=True)
    shD, shD2 = nets.d_if_param_shapes(4, 8, 2), nets.d_if_param_shapes(3, 8, 2)
    model.netG.load_state_dict(detrand.test_weights(shG, seed))
    model.netD.load_state_dict(detrand.test_weights(shD, seed + 1))
    model.netD2.load_state_dict(detrand.test_weights(shD2, seed + 2))
    model.train()
    batch = p2p_batch(n, size, seed)
    out = {"size": size, "seed": seed, "n": n, "flags": np.array(flags), "lambda_vgg": float(opt.lambda_vgg)}
    model.set_input(batch, phase="train")
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    out["s0/loss_names"] = np.array(list(losses.keys()))
    out["s0/loss_values"] = np.array([float(v) for v in losses.values()], dtype=np.float64)
    for kk, p in model.netG.named_parameters():
        out["s0/grad_G/%s" % kk] = detrand.probe(p.grad, kk)
    out["s0/fake_I"] = model.fake_I.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "pix2pixHD_vgg_step_%d.npz" % size), **out)
    print("wrote pix2pixHD_vgg_step_%d.npz" % size, {k: float(v) for k, v in losses.items()})


def golden_metrics(seed=606):
    """T_AE / T_MSE from the reference's compute_evaluation_metric (I_PSNR needs torchmetrics, which is not installed:
    the oracle restates torchmetrics' published formula for it)"""
    from oracle import detrand, ref_import

    ref_import.load()
    from models.model_utils import compute_evaluation_metric

    real_I, fake_I = detrand.uniform((1, 3, 64, 64), seed, "rI"), 1.2 * detrand.uniform((1, 3, 64, 64), seed, "fI")
    real_T, fake_T = 0.3 * detrand.uniform((6, 2, 32, 32), seed, "rT"), 0.6 * detrand.uniform((6, 2, 32, 32), seed, "fT")
    m = compute_evaluation_metric(["G"], real_I, fake_I, real_T_concat=real_T, fake_T_concat=fake_T, eval_metrics=["T_AE", "T_MSE"])
    out = {"seed": seed, "T_AE": float(m["metric_T_AE"]), "T_MSE": float(m["metric_T_MSE"])}
    # Frechet distance of the reference (models/sifid.py:102-176) on synthetic channel-major features [D, P]
    from models import sifid
    from oracle import nets
    for i, (d, p1, p2) in enumerate(nets.FRECHET_CASES):
        f1, f2 = nets.frechet_case(i, seed)
        a1, a2 = f1.numpy().T.astype(np.float64), f2.numpy().T.astype(np.float64)       # (positions, dims) as get_activations returns
        out["fd/%d" % i] = float(sifid.calculate_frechet_distance(np.mean(a1, axis=0), np.cov(a1, rowvar=False), np.mean(a2, axis=0),
                                                                  np.cov(a2, rowvar=False)))
    np.savez_compressed(os.path.join(GOLD, "metrics.npz"), **out)
    print("wrote metrics.npz", out)


def sifid_inputs(seed=4242):
    from oracle import detrand
    real_I, fake_I = detrand.uniform((1, 3, 72, 88), seed, "rI"), 1.2 * detrand.uniform((1, 3, 72, 88), seed, "fI")
    real_T, fake_T = 0.3 * detrand.uniform((5, 2, 32, 32), seed, "rT"), 0.6 * detrand.uniform((5, 2, 32, 32), seed, "fT")
    return real_I, fake_I, real_T, fake_T


def golden_sifid(seed=4242):
    """I_SIFID / T_SIFID through the REFERENCE's compute_evaluation_metric -> calculate_sifid_given_arrays -> get_activations ->
    calculate_activation_statistics -> calculate_frechet_distance (models/model_utils.py:481-488, 541-555; models/sifid.py).  The only
    stand-in is the network object: torchvision (and its pretrained weights) is absent, so sifid.InceptionV3 is replaced by a module
    that applies the wrapper's `2 * x - 1` and then oracle.nets.inception_block0 on the seeded stand-in weights of
    visual-tactile-synthesis_amd/models/inception.py -- everything around the three BasicConv2d layers is the reference's own code."""
    from oracle import nets, ref_import

    ref_import.load()
    sys.path.insert(1, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-tactile-synthesis_amd", "models"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("vts_inception", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                "visual-tactile-synthesis_amd", "models", "inception.py"))
    inc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(inc)
    sd = inc.InceptionBlock0().state_dict()
    from models import model_utils, sifid

    class StandIn(torch.nn.Module):
        BLOCK_INDEX_BY_DIM = {64: 0}

        def __init__(self, blocks):
            super().__init__()

        def forward(self, x):
            return [nets.inception_block0(2 * x - 1, sd)]

    sifid.InceptionV3 = StandIn
    model_utils.calculate_sifid_given_arrays.__globals__["InceptionV3"] = StandIn
    real_I, fake_I, real_T, fake_T = sifid_inputs(seed)
    m = model_utils.compute_evaluation_metric(["G"], real_I, fake_I, real_T_concat=real_T, fake_T_concat=fake_T,
                                              eval_metrics=["I_SIFID", "T_SIFID"], device=None)
    out = {"seed": seed, "I_SIFID": float(m["metric_I_SIFID"]), "T_SIFID": float(m["metric_T_SIFID"])}
    np.savez_compressed(os.path.join(GOLD, "sifid.npz"), **out)
    print("wrote sifid.npz", out)


def lpips_metric_inputs(seed=5151):
    from oracle import detrand
    real_I, fake_I = detrand.uniform((2, 3, 72, 88), seed, "rI"), 1.2 * detrand.uniform((2, 3, 72, 88), seed, "fI")
    real_T, fake_T = 0.3 * detrand.uniform((5, 2, 32, 32), seed, "rT"), 0.6 * detrand.uniform((5, 2, 32, 32), seed, "fT")
    return real_I, fake_I, real_T, fake_T


def golden_lpips_metrics(seed=5151):
    """I_LPIPS / T_LPIPS through the REFERENCE's compute_evaluation_metric (models/model_utils.py:475-478, 521-527 ->
    models/tactile_patch_fid.py:compute_touch_lpips_loss) for both backbones the reference evaluates with: `eval_LPIPS` =
    lpips.LPIPS(net="vgg") while training / validating and lpips.LPIPS(net="alex") in the test phase (models/sinskitG_model.py:497-501).
    The only stand-in is the network object (pip package `lpips` and its weights are absent): oracle/perceptual.py:LPIPS, the
    restatement of the package's published algorithm, on seeded stand-in weights.  Also the module's own values on a seeded pair."""
    from oracle import detrand, perceptual, ref_import

    ref_import.load()
    from models import model_utils

    real_I, fake_I, real_T, fake_T = lpips_metric_inputs(seed)
    out = {"seed": seed}
    for net in ("vgg", "alex"):
        lp = perceptual.LPIPS(net=net)
        with torch.no_grad():
            m = model_utils.compute_evaluation_metric(["G"], real_I, fake_I, real_T_concat=real_T, fake_T_concat=fake_T,
                                                      eval_metrics=["I_LPIPS", "T_LPIPS"], eval_LPIPS=lp, device=None)
            a, b = detrand.uniform((2, 3, 80, 96), seed, "lp_a"), detrand.uniform((2, 3, 80, 96), seed, "lp_b")
            out["%s/module_val" % net] = lp(a, b).flatten().numpy()
        out["%s/I_LPIPS" % net], out["%s/T_LPIPS" % net] = float(m["metric_I_LPIPS"]), float(m["metric_T_LPIPS"])
    np.savez_compressed(os.path.join(GOLD, "lpips_metrics.npz"), **out)
    print("wrote lpips_metrics.npz", {k: v for k, v in out.items()})


def golden_io():
    """util.tensor2im / tensor2arr of the reference (util/util.py:58-122) on a ramp tensor"""
    from oracle import ref_import

    ref_import.load()
    from util import util as ru

    x = torch.linspace(-1.5, 1.5, 2 * 3 * 4 * 5).reshape(2, 3, 4, 5)
    out = {"im_rgb": ru.tensor2im(x), "im_gray": ru.tensor2im(x[:, :1]), "im_2d": ru.tensor2im(x[0, 0]),
           "arr_gray": ru.tensor2arr(x[:1, :1], imtype=np.float32), "arr_rgb": ru.tensor2arr(x[:1])}
    np.savez_compressed(os.path.join(GOLD, "image_io.npz"), **out)
    print("wrote image_io.npz", {k: v.shape for k, v in out.items()})


def friction_inputs(seed=515, h=40, w=56):
    """seeded stand-ins for the saved outputs the rendering post-processing reads: raw gx / gy arrays in (-1, 1), a uint8 image and mask"""
    from oracle import detrand
    gx = (0.4 * detrand.uniform((h, w), seed, "gx")).numpy().astype(np.float64)
    gy = (0.4 * detrand.uniform((h, w), seed, "gy")).numpy().astype(np.float64)
    img = ((detrand.uniform((h, w, 3), seed, "I").numpy() + 1) * 127.5).astype(np.uint8)
    m = np.where(detrand.uniform((h, w), seed, "M").numpy() > -0.5, 255, 127).astype(np.uint8)
    return gx, gy, img, m


def golden_friction():
    """postprocess_gz of the reference (Step2_Postprocessing_for_Rendering.py:18-140), mappings log10 / exp2, raw-array and PNG inputs"""
    from PIL import Image

    from oracle import ref_import

    # The upstream script does not import (IndentationError at its line 340, and it needs cv2 / skimage at module level): run the text
    # of its postprocess_gz function only, read from the reference tree at generation time (nothing of it is kept here).
    src = open(os.path.join(ref_import.REF_ROOT, "Step2_Postprocessing_for_Rendering.py")).read()
    body = src[src.index("def postprocess_gz("):src.index("def generate_Tanvas_images(")]
    ns = {"np": np, "Image": Image}
    exec(compile(body, "Step2_Postprocessing_for_Rendering.py:postprocess_gz", "exec"), ns)
    step2 = type("step2", (), {"postprocess_gz": staticmethod(ns["postprocess_gz"])})

    gx, gy, img, m = friction_inputs()
    out = {}
    for tag, kw in (("log10_raw", dict(method="log10", use_raw_arr=True)),
                    ("exp2_png_thr", dict(method="exp2", use_raw_arr=False, thresholding=True, threshold_quantile=0.8, change_bg_color=True, bg_color=(1, 2, 3)))):
        a, b = (gx, gy) if kw["use_raw_arr"] else (np.round((gx + 1) * 127.5), np.round((gy + 1) * 127.5))
        res = step2.postprocess_gz(img.copy(), m, a.copy(), b.copy(), Tanvas_width=48, Tanvas_height=32, **kw)
        for k, v in zip(("gz", "I", "post", "gz_T", "I_T", "post_T"), res):
            out["%s/%s" % (tag, k)] = v
    np.savez_compressed(os.path.join(GOLD, "friction.npz"), **out)
    print("wrote friction.npz", {k: v.shape for k, v in out.items()})


def golden_sg2(size=32, seed=808, ndf=8, input_nc=4, n=3):
    """StyleGAN2 blocks: the reference's StyleGAN2Discriminator forward + gradients, upfirdn2d in several
    up / down / pad configurations, fused_leaky_relu, ModulatedConv2d (plain / upsample / downsample) with a style vector."""
    import argparse

    from oracle import detrand, ref_import, stylegan2 as sg

    ref_import.load()
    from models import stylegan_networks as R

    out = {"size": size, "seed": seed, "ndf": ndf, "input_nc": input_nc, "n": n}
    opt = argparse.Namespace(netD="stylegan2", D_patch_size=None, load_size=size, crop_size=size)
    D = R.StyleGAN2Discriminator(input_nc, ndf, 3, False, size=size, opt=opt)
    ref = {k: tuple(v.shape) for k, v in D.named_parameters()}
    mine = sg.d_param_shapes(input_nc, ndf, size)
    assert ref == {k: tuple(v) for k, v in mine.items()}, "StyleGAN2 D key/shape mismatch"
    for k, v in sg.d_buffers(input_nc, ndf, size).items():
        assert torch.equal(D.state_dict()[k], v), k
    out["ref_keys"] = np.array(sorted(D.state_dict().keys()))
    D.load_state_dict(sg.test_weights(mine, seed), strict=False)
    x = detrand.uniform((n, input_nc, size, size), seed, "d_in").requires_grad_(True)
    y = D(x)
    (y * detrand.uniform(tuple(y.shape), seed, "d_cot")).sum().backward()
    out["D_out"] = y.detach().numpy()
    out["D_dx_probe"] = detrand.probe(x.grad, "d_dx")
    out["D_dx_sub"] = x.grad[:, :, ::4, ::4].numpy()
    for k, p in D.named_parameters():
        out["D_grad/" + k] = detrand.probe(p.grad, k)
    # upfirdn2d
    u = detrand.uniform((2, 3, 9, 11), seed, "ufd_in")
    k4 = R.make_kernel([1, 3, 3, 1])
    for i, (up, down, pad) in enumerate(sg.UPFIRDN_CASES):
        out["ufd/%d" % i] = R.upfirdn2d(u, k4 * (up ** 2), up=up, down=down, pad=pad).numpy()
    out["flrelu"] = R.fused_leaky_relu(u, detrand.uniform((1, 3, 1, 1), seed, "flb")).numpy()
    # ModulatedConv2d with a style vector
    for tag, kw in (("plain", {}), ("up", {"upsample": True}), ("down", {"downsample": True}), ("nodemod", {"demodulate": False})):
        M = R.ModulatedConv2d(12, 20, 3, 16, **kw)
        shapes = {k: tuple(v.shape) for k, v in M.named_parameters()}
        M.load_state_dict(sg.test_weights(shapes, seed + 1), strict=False)
        xi = detrand.uniform((2, 12, 10, 10), seed, "mod_in").requires_grad_(True)
        st = detrand.uniform((2, 16), seed, "mod_style").requires_grad_(True)
        yo = M(xi, st)
        (yo * detrand.uniform(tuple(yo.shape), seed, "mod_cot" + tag)).sum().backward()
        out["mod/%s/out" % tag] = yo.detach().numpy()
        out["mod/%s/dx" % tag] = detrand.probe(xi.grad, "mdx")
        out["mod/%s/dstyle" % tag] = st.grad.numpy()
        for k, p in M.named_parameters():
            out["mod/%s/grad/%s" % (tag, k)] = detrand.probe(p.grad, k)
    np.savez_compressed(os.path.join(GOLD, "stylegan2_%d.npz" % size), **out)
    print("wrote stylegan2_%d.npz (%d entries)" % (size, len(out)))


SG2G_CFG = dict(ngf=2, size=32, n_blocks=2, num_downsampling=2)


def golden_sg2g(seed=909, input_nc=4, n=2):
    """StyleGAN2Generator of the reference (`--netG smallstylegan2`: encoder + decoder without noise injection, stylegan_networks.py:
    800-930) on CPU with seeded weights: output, input gradient, every parameter gradient.  Its ModulatedConv2d builds the style-free
    modulation with `.cuda()` (:310); Tensor.cuda is made a no-op for this run -- nothing else is touched."""
    import argparse

    from oracle import detrand, ref_import, stylegan2 as sg

    ref_import.load()
    from models import stylegan_networks as R

    cfg = SG2G_CFG
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        opt = argparse.Namespace(load_size=cfg["size"], crop_size=cfg["size"], stylegan2_G_num_downsampling=cfg["num_downsampling"], netG="smallstylegan2")
        G = R.StyleGAN2Generator(input_nc, 3, ngf=cfg["ngf"], n_blocks=cfg["n_blocks"], opt=opt)
        shapes = sg.g_param_shapes(input_nc, **cfg)
        assert {k: tuple(v.shape) for k, v in G.named_parameters()} == shapes, "StyleGAN2 G key/shape mismatch"
        G.load_state_dict(sg.test_weights(shapes, seed), strict=False)
        x = detrand.uniform((n, input_nc, cfg["size"], cfg["size"]), seed, "g_in").requires_grad_(True)
        y = G(x)
        (y * detrand.uniform(tuple(y.shape), seed, "g_cot")).sum().backward()
    finally:
        torch.Tensor.cuda = cuda
    out = {"seed": seed, "input_nc": input_nc, "n": n, "ref_keys": np.array(sorted(G.state_dict().keys())), "G_out": y.detach().numpy(),
           "G_dx_sub": x.grad[:, :, ::4, ::4].numpy(), "G_dx_probe": detrand.probe(x.grad, "g_dx")}
    for k, p in G.named_parameters():
        if p.grad is not None:
            out["G_grad/" + k] = detrand.probe(p.grad, k)
    np.savez_compressed(os.path.join(GOLD, "stylegan2_g_32.npz"), **out)
    print("wrote stylegan2_g_32.npz (%d entries)" % len(out))


def golden_diffaug():
    """DiffAugment beyond 'bs' (thirdparty/DiffAugment.py:25-96): every letter on its own and three multi-letter policies, on odd and
    even extents.  Inputs regenerate from detrand; the draws regenerate from torch's global generator (nets.diffaug_draws after
    torch.manual_seed); the fixture keeps the reference's outputs sub-sampled + probed and the small draws as a cross-check."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from thirdparty.DiffAugment import DiffAugment

    out = {"policies": np.array(["b", "s", "c", "t", "o", "n", "bsctno", "onctsb", "ttcc"]), "shapes": np.array([[2, 3, 20, 28], [3, 3, 33, 17], [1, 3, 64, 64]])}
    for pol in out["policies"]:
        pol = str(pol)
        for si, shape in enumerate(out["shapes"]):
            shape = tuple(int(v) for v in shape)
            x = detrand.uniform(shape, 11 + si, "diffaug_" + pol)
            torch.manual_seed(50 + si)
            ref = DiffAugment(x, policy=pol)
            torch.manual_seed(50 + si)
            d = nets.diffaug_draws(pol, shape)
            assert float((nets.diffaug(x, pol, d) - ref).abs().max()) <= 2e-7      # (exact but for the summation order of the contrast mean)
            tag = "%s/%d" % (pol, si)
            if si == 0:
                out[tag + "/out"] = ref.numpy()
            else:
                out[tag + "/out_sub"] = ref[:, :, ::3, ::3].numpy()
                out[tag + "/out_probe"] = detrand.probe(ref, "out")
            for k, dd in enumerate(d):
                for nm, v in dd.items():
                    if nm != "noise":
                        out["%s/draw%d_%s" % (tag, k, nm)] = v.numpy()
                    else:
                        out["%s/draw%d_noise_probe" % (tag, k)] = detrand.probe(v, "noise")
    np.savez_compressed(os.path.join(GOLD, "diffaug.npz"), **out)
    print("wrote diffaug.npz (%d entries)" % len(out))


VARIANTS = {
    # PatchGAN depth per discriminator off its default (row a7).  (gan_mode lsgan / vanilla / wgan cannot run the reference's sinskitG
    # step at all: compute_G2_loss takes len() of the scalar those modes return, sinskitG_model.py:1783 -- so the variants use the two
    # per-sample modes.)
    "depth_2_4": ["--n_layers_D", "2", "--n_layers_D2", "4"],
    "hinge_depth_4_2": ["--gan_mode", "hinge", "--n_layers_D", "4", "--n_layers_D2", "2"],
    # DiffAugment policy with every letter (row a15)
    "diffaug_all": ["--diffaugment", "bsctno"],
    # Dropout(0.5) in the intermediate Up blocks of the generator (row a4; the draws replay: nets.dropout_draws, then DiffAugment's)
    "dropout": ["--no_dropout", "False"],
    # the same flag with the ResNet generator (row a19): Dropout(0.5) inside every block
    "resnet_dropout": ["--netG", "resnet_6blocks", "--no_dropout", "False"],
}


# conditioning ablations the reference CAN run (probed round 5: --use_cGAN_G2 False dies in define_D -- networks.py:1658 dereferences an opt
# that sinskitG_model.py:575 does not pass -- and --use_bg_mask False in optimize_parameters -- sinskitG_model.py:638 reads a self.M that
# set_input only creates under use_bg_mask, :721)
COND_VARIANTS = {
    "no_cGAN": ["--use_cGAN", "False"],                                         # D1 on the image alone (3 channels)
    "G2_no_S": ["--use_cGAN_G2_S", "False"],                                    # D2 stacks [T, I, mask] (6 channels)
    "G2_no_I": ["--use_cGAN_G2_I", "False"],                                    # D2 stacks [T, S] (3 channels)
    "no_cGAN_G2_T_only": ["--use_cGAN", "False", "--use_cGAN_G2_S", "False", "--use_cGAN_G2_I", "False"],     # D2 on the tactile patches alone
}


def cond_channels(extra):
    kw = dict(zip((k.lstrip("-") for k in extra[::2]), (v == "True" for v in extra[1::2])))
    c1 = 3 + (1 if kw.get("use_cGAN", True) else 0)
    c2 = 2 + (1 if kw.get("use_cGAN_G2_S", True) else 0) + (4 if kw.get("use_cGAN_G2_I", True) else 0)
    return c1, c2


def golden_step_conditioning(size=256, seed=515, nt=64):
    """One SinSKITGModel.optimize_parameters of the REFERENCE per entry of COND_VARIANTS -> tests/golden/sinskitG_cond_step_256.npz"""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models.sinskitG_model import SinSKITGModel

    out = {"size": size, "seed": seed, "nt": nt, "variants": np.array(list(COND_VARIANTS))}
    for vi, (name, extra) in enumerate(COND_VARIANTS.items()):
        flags = ["--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0", "--use_vision_aided_loss", "False", "--lambda_G2_GAN_feat", "0",
                 "--checkpoints_dir", "/tmp/vts_golden_ckpt", "--name", "golden_cond"] + extra
        opt = _ref_opt("sinskitG", True, flags)
        model = SinSKITGModel(opt)
        model.setup(opt)
        c1, c2 = cond_channels(extra)
        shapesG, shapesD, shapesD2 = nets.g_param_shapes(), nets.d_param_shapes(c1), nets.d_param_shapes(c2)
        for net, sh in ((model.netD, shapesD), (model.netD2, shapesD2)):
            assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in sh.items()}, name
        model.netG.load_state_dict(detrand.test_weights(shapesG, seed + 10 * vi))
        model.netD.load_state_dict(detrand.test_weights(shapesD, seed + 10 * vi + 1))
        model.netD2.load_state_dict(detrand.test_weights(shapesD2, seed + 10 * vi + 2))
        model.train()
        batch = _synthetic_batch(size, nt, seed + 10 * vi)
        model.set_input(batch, phase="train")
        k = int(nets.dilated_mask_positions(model.M).shape[0])
        torch.manual_seed(seed + vi)
        aug = torch.stack([torch.rand(1, 1, 1, 1).flatten() for _ in range(4)])
        random.seed(seed + vi)
        more = np.array(random.sample(range(k), opt.add_fake_T_sample_size), dtype=np.int64)[None]
        torch.manual_seed(seed + vi)
        random.seed(seed + vi)
        model.optimize_parameters(epoch=1)
        tag = name
        out[tag + "/flags"] = json.dumps(extra)
        out[tag + "/aug"] = aug.numpy()
        out[tag + "/more_idx"] = more
        losses = model.get_current_losses()
        out[tag + "/loss_names"] = np.array(list(losses.keys()))
        out[tag + "/loss_values"] = np.array(list(losses.values()), dtype=np.float64)
        for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
            for kk, p in net.named_parameters():
                out["%s/grad_%s/%s" % (tag, nm, kk)] = detrand.probe(p.grad, kk)
                out["%s/param_%s/%s" % (tag, nm, kk)] = detrand.probe(p, kk)
            for kk, b in net.named_buffers():
                out["%s/buf_%s/%s" % (tag, nm, kk)] = b.detach().double().numpy()
        out[tag + "/fake_I_probe"] = detrand.probe(model.fake_I, "fake_I")
        out[tag + "/fake_T_probe"] = detrand.probe(model.fake_T, "fake_T")
        out[tag + "/pred_fake_T_full_probe"] = detrand.probe(model.pred_fake_T_full, "pftf")
        out[tag + "/pred_fake_I_probe"] = detrand.probe(model.pred_fake_I, "pfi")
        print(name, c1, c2, {k: round(float(v), 5) for k, v in losses.items()})
    np.savez_compressed(os.path.join(GOLD, "sinskitG_cond_step_%d.npz" % size), **out)
    print("wrote sinskitG_cond_step_%d.npz (%d entries)" % (size, len(out)))


def golden_step_d3_warmup(size=256, seed=626, nt=64):
    """One SinSKITGModel.optimize_parameters of the REFERENCE with its DEFAULT --use_vision_aided_loss True at epoch 1, i.e. before
    --vision_aided_warmup_epoch (100): the model constructs vision_aided_loss.Discriminator (sinskitG_model.py:546-551; the package is
    absent here, so a stand-in class whose forward RAISES is injected -- the fixture proves it is never called) and reports the three D3
    entries as 0.0 (:1399-1402, 1721-1722).  Pinned: the order of the loss names and that the step is the flag-off step."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    import vision_aided_loss

    class _NeverCalled(torch.nn.Module):
        def __init__(self, cv_type=None, loss_type=None, device=None, **kw):
            super().__init__()
            assert cv_type == "clip" and loss_type == "multilevel_sigmoid_s", (cv_type, loss_type)
            self.cv_ensemble = torch.nn.Linear(1, 1)

        def forward(self, *a, **k):
            raise AssertionError("netD3 called before the warm-up epoch")

    vision_aided_loss.Discriminator = _NeverCalled
    from models.sinskitG_model import SinSKITGModel

    flags = ["--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0", "--lambda_G2_GAN_feat", "0", "--checkpoints_dir", "/tmp/vts_golden_ckpt",
             "--name", "golden_d3"]
    opt = _ref_opt("sinskitG", True, flags)
    assert opt.use_vision_aided_loss is True and opt.vision_aided_warmup_epoch == 100
    model = SinSKITGModel(opt)
    model.setup(opt)
    model.netG.load_state_dict(detrand.test_weights(nets.g_param_shapes(), seed))
    model.netD.load_state_dict(detrand.test_weights(nets.d_param_shapes(4), seed + 1))
    model.netD2.load_state_dict(detrand.test_weights(nets.d_param_shapes(7), seed + 2))
    model.train()
    batch = _synthetic_batch(size, nt, seed)
    model.set_input(batch, phase="train")
    k = int(nets.dilated_mask_positions(model.M).shape[0])
    torch.manual_seed(seed)
    aug = torch.stack([torch.rand(1, 1, 1, 1).flatten() for _ in range(4)])
    random.seed(seed)
    more = np.array(random.sample(range(k), opt.add_fake_T_sample_size), dtype=np.int64)[None]
    torch.manual_seed(seed)
    random.seed(seed)
    model.optimize_parameters(epoch=1)
    losses = model.get_current_losses()
    out = {"size": size, "seed": seed, "nt": nt, "warmup_epoch": opt.vision_aided_warmup_epoch, "aug": aug.numpy(), "more_idx": more,
           "loss_names": np.array(list(losses.keys())), "loss_values": np.array(list(losses.values()), dtype=np.float64),
           "fake_I_probe": detrand.probe(model.fake_I, "fake_I"), "fake_T_probe": detrand.probe(model.fake_T, "fake_T")}
    for kk, p in model.netG.named_parameters():
        out["param_G/" + kk] = detrand.probe(p, kk)
    np.savez_compressed(os.path.join(GOLD, "sinskitG_d3_warmup_step_%d.npz" % size), **out)
    print("wrote sinskitG_d3_warmup_step_%d.npz" % size, {k: round(float(v), 5) for k, v in losses.items()})


def variant_g_shapes(opt):
    from oracle import nets
    if opt.netG.startswith("resnet_"):
        return nets.resnet_param_shapes(n_blocks=int(opt.netG[len("resnet_")]), use_dropout=not opt.no_dropout)
    return nets.g_param_shapes()


def variant_dropout_draws(opt, size):
    from oracle import nets
    if opt.netG.startswith("resnet_"):
        return nets.resnet_dropout_draws((1, size, size), n_blocks=int(opt.netG[len("resnet_")]))
    return nets.dropout_draws((1, size, size))


def golden_step_variants(size=256, seed=232, nt=64):
    """One SinSKITGModel.optimize_parameters of the REFERENCE per entry of VARIANTS."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models.sinskitG_model import SinSKITGModel

    out = {"size": size, "seed": seed, "nt": nt, "variants": np.array(list(VARIANTS))}
    for vi, (name, extra) in enumerate(VARIANTS.items()):
        flags = ["--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0", "--use_vision_aided_loss", "False", "--lambda_G2_GAN_feat", "0",
                 "--checkpoints_dir", "/tmp/vts_golden_ckpt", "--name", "golden_var"] + extra
        opt = _ref_opt("sinskitG", True, flags)
        model = SinSKITGModel(opt)
        model.setup(opt)
        shapesG = variant_g_shapes(opt)
        shapesD = nets.d_param_shapes(4, n_layers=opt.n_layers_D)
        shapesD2 = nets.d_param_shapes(7, n_layers=opt.n_layers_D2)
        for net, sh in ((model.netD, shapesD), (model.netD2, shapesD2)):
            assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in sh.items()}, name
        model.netG.load_state_dict(detrand.test_weights(shapesG, seed + 10 * vi), strict=not opt.netG.startswith("resnet_"))   # (blur `filt` buffers)
        model.netD.load_state_dict(detrand.test_weights(shapesD, seed + 10 * vi + 1))
        model.netD2.load_state_dict(detrand.test_weights(shapesD2, seed + 10 * vi + 2))
        model.train()
        batch = _synthetic_batch(size, nt, seed + 10 * vi)
        model.set_input(batch, phase="train")
        k = int(nets.dilated_mask_positions(model.M).shape[0])
        torch.manual_seed(seed + vi)
        if not opt.no_dropout:
            variant_dropout_draws(opt, size)      # the generator forward consumes its dropout masks first
        if opt.diffaugment == "bs":
            aug = torch.stack([torch.rand(1, 1, 1, 1).flatten() for _ in range(4)])
        else:
            aug = None
        random.seed(seed + vi)
        more = np.array(random.sample(range(k), opt.add_fake_T_sample_size), dtype=np.int64)[None]
        torch.manual_seed(seed + vi)
        random.seed(seed + vi)
        model.optimize_parameters(epoch=1)
        tag = name
        out[tag + "/flags"] = json.dumps(extra)
        if aug is not None:
            out[tag + "/aug"] = aug.numpy()
        out[tag + "/more_idx"] = more
        losses = model.get_current_losses()
        out[tag + "/loss_names"] = np.array(list(losses.keys()))
        out[tag + "/loss_values"] = np.array(list(losses.values()), dtype=np.float64)
        for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
            for kk, p in net.named_parameters():
                out["%s/grad_%s/%s" % (tag, nm, kk)] = detrand.probe(p.grad, kk)
                out["%s/param_%s/%s" % (tag, nm, kk)] = detrand.probe(p, kk)
            for kk, b in net.named_buffers():
                out["%s/buf_%s/%s" % (tag, nm, kk)] = b.detach().double().numpy()
        out[tag + "/fake_I_probe"] = detrand.probe(model.fake_I, "fake_I")
        out[tag + "/fake_T_probe"] = detrand.probe(model.fake_T, "fake_T")
        out[tag + "/aug_fake_I_probe"] = detrand.probe(model.aug_fake_I, "aug_fake_I")
        out[tag + "/aug_real_I_probe"] = detrand.probe(model.aug_real_I, "aug_real_I")
        out[tag + "/aug_fake_I_sub"] = model.aug_fake_I.detach()[:, :, ::8, ::8].numpy()
        out[tag + "/pred_fake_T_full_probe"] = detrand.probe(model.pred_fake_T_full, "pftf")
        out[tag + "/pred_fake_I_probe"] = detrand.probe(model.pred_fake_I, "pfi")
        print(name, {k: round(float(v), 5) for k, v in losses.items()})
    np.savez_compressed(os.path.join(GOLD, "sinskitG_variants_step_%d.npz" % size), **out)
    print("wrote sinskitG_variants_step_%d.npz (%d entries)" % (size, len(out)))


def golden_step(size=256, seed=202, steps=2, nt=64):
    """Full SinSKITGModel.optimize_parameters x `steps` on one synthetic sample (BASELINE config 0)."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models.sinskitG_model import SinSKITGModel

    flags = ["--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0", "--use_vision_aided_loss", "False",
             "--lambda_G2_GAN_feat", "0", "--checkpoints_dir", "/tmp/vts_golden_ckpt", "--name", "golden"]
    opt = _ref_opt("sinskitG", True, flags)
    model = SinSKITGModel(opt)
    model.setup(opt)
    shapesG, shapesD, shapesD2 = nets.g_param_shapes(), nets.d_param_shapes(4), nets.d_param_shapes(7)
    model.netG.load_state_dict(detrand.test_weights(shapesG, seed))
    model.netD.load_state_dict(detrand.test_weights(shapesD, seed + 1))
    model.netD2.load_state_dict(detrand.test_weights(shapesD2, seed + 2))
    model.train()
    batch = _synthetic_batch(size, nt, seed)
    out = {"size": size, "seed": seed, "steps": steps, "nt": nt, "flags": json.dumps(flags)}
    for it in range(steps):
        model.set_input(batch, phase="train")
        k = int(nets.dilated_mask_positions(model.M).shape[0])
        torch.manual_seed(seed + it)
        aug = torch.stack([torch.rand(1, 1, 1, 1).flatten() for _ in range(4)])
        random.seed(seed + it)
        more = np.array(random.sample(range(k), opt.add_fake_T_sample_size), dtype=np.int64)[None]
        torch.manual_seed(seed + it)
        random.seed(seed + it)
        model.optimize_parameters(epoch=1)
        tag = "s%d" % it
        out[tag + "/aug"] = aug.numpy()
        out[tag + "/more_idx"] = more
        out[tag + "/more_ox"] = np.asarray(model.fake_sample_offset_x).reshape(-1)
        out[tag + "/more_oy"] = np.asarray(model.fake_sample_offset_y).reshape(-1)
        losses = model.get_current_losses()
        out[tag + "/loss_names"] = np.array(list(losses.keys()))
        out[tag + "/loss_values"] = np.array(list(losses.values()), dtype=np.float64)
        for nm, net in (("G", model.netG), ("D", model.netD), ("D2", model.netD2)):
            for kk, p in net.named_parameters():
                out["%s/grad_%s/%s" % (tag, nm, kk)] = detrand.probe(p.grad, kk)
                out["%s/param_%s/%s" % (tag, nm, kk)] = detrand.probe(p, kk)
            for kk, b in net.named_buffers():
                out["%s/buf_%s/%s" % (tag, nm, kk)] = b.detach().double().numpy()
        out[tag + "/fake_I_sub"] = model.fake_I.detach()[:, :, ::4, ::4].numpy()
        out[tag + "/fake_T_sub"] = model.fake_T.detach()[:, :, ::4, ::4].numpy()
        out[tag + "/fake_I_probe"] = detrand.probe(model.fake_I, "fake_I")
        out[tag + "/fake_T_probe"] = detrand.probe(model.fake_T, "fake_T")
        out[tag + "/fake_N_probe"] = detrand.probe(model.fake_N, "fake_N")
        out[tag + "/aug_fake_I_probe"] = detrand.probe(model.aug_fake_I, "aug_fake_I")
        out[tag + "/aug_real_I_probe"] = detrand.probe(model.aug_real_I, "aug_real_I")
        out[tag + "/pred_fake_T_full_probe"] = detrand.probe(model.pred_fake_T_full, "pftf")
        out[tag + "/pred_fake_I_probe"] = detrand.probe(model.pred_fake_I, "pfi")
    np.savez_compressed(os.path.join(GOLD, "sinskitG_step_%d.npz" % size), **out)
    print("wrote sinskitG_step_%d.npz (%d entries)" % (size, len(out)))
    print({k: float(v) for k, v in losses.items()})


FULL_GRAD_LAYERS = ("down4", "down5", "down6", "down7", "up7", "up6", "up5", "up4")


def golden_step_full_grads(size=256, seed=202, nt=64):
    """Step 0 of golden_step() once more, twice: the REFERENCE in its own fp32 and the same reference code in float64 (default dtype
    float64, every network / input cast) -> tests/golden/sinskitG_step_grads_256.npz:
      * g64/<key>: the float64 gradient (stored rounded to fp32) of every convolution weight of the generator (1.46 M values, of which
        the inner layers down4 ... up4 -- `layers` -- hold 1.1 M): the HIP step is judged against these by TRUE relative L2
        (tests/test_step_gpu.py);
      * ref32_vs_64/<key>: relative L2 distance between the reference's fp32 gradient and its float64 gradient for EVERY generator
        tensor -- how far the reference's own CPU path is from exact arithmetic (down0 ... down2: ~5e-3);
      * g32_probe/<key>: the probes of the fp32 run (equal to sinskitG_step_256.npz: the two files belong to one run).
    The generator's gradient does not depend on the DiffAugment draws (its loss is the D1 term on the un-augmented image plus the L1
    terms, sinskitG_model.py:1671-1716), which is why the float64 run's different random stream does not matter here."""
    from oracle import detrand, nets, ref_import

    ref_import.load()
    from models.sinskitG_model import SinSKITGModel

    flags = ["--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0", "--use_vision_aided_loss", "False",
             "--lambda_G2_GAN_feat", "0", "--checkpoints_dir", "/tmp/vts_golden_ckpt", "--name", "golden"]

    def run(dtype):
        torch.set_default_dtype(dtype)
        try:
            opt = _ref_opt("sinskitG", True, flags)
            model = SinSKITGModel(opt)
            model.setup(opt)
            cast = lambda sd: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}      # noqa: E731
            model.netG.load_state_dict(cast(detrand.test_weights(nets.g_param_shapes(), seed)))
            model.netD.load_state_dict(cast(detrand.test_weights(nets.d_param_shapes(4), seed + 1)))
            model.netD2.load_state_dict(cast(detrand.test_weights(nets.d_param_shapes(7), seed + 2)))
            for net in (model.netG, model.netD, model.netD2):
                net.to(dtype)
            model.train()
            batch = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in _synthetic_batch(size, nt, seed).items()}
            model.set_input(batch, phase="train")
            torch.manual_seed(seed)
            random.seed(seed)
            model.optimize_parameters(epoch=1)
            assert model.fake_I.dtype == dtype and next(model.netG.parameters()).grad.dtype == dtype
            return ({k: p.grad.detach().clone() for k, p in model.netG.named_parameters()},
                    {k: float(v) for k, v in model.get_current_losses().items()})
        finally:
            torch.set_default_dtype(torch.float32)

    g32, l32 = run(torch.float32)
    g64, l64 = run(torch.float64)
    out = {"size": size, "seed": seed, "nt": nt, "layers": np.array(FULL_GRAD_LAYERS),
           "loss_names": np.array(list(l32)), "loss32": np.array(list(l32.values())), "loss64": np.array([l64[k] for k in l32])}
    for k in g32:
        out["ref32_vs_64/" + k] = ((g32[k].double() - g64[k]).norm() / g64[k].norm().clamp_min(1e-300)).item()
        out["g32_probe/" + k] = detrand.probe(g32[k], k)
        if k.endswith("weight"):
            out["g64/" + k] = g64[k].float().numpy()
    np.savez_compressed(os.path.join(GOLD, "sinskitG_step_grads_%d.npz" % size), **out)
    print("wrote sinskitG_step_grads_%d.npz" % size)
    for k in g32:
        if k.endswith("weight"):
            print("  %-24s reference fp32 vs float64: %.3e" % (k, out["ref32_vs_64/" + k]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["ops", "nets", "step", "resnet", "global", "local", "p2p", "metrics", "sg2", "sg2step", "style", "io"]
    if "cond" in which:
        golden_step_conditioning()
    if "patchsample" in which:
        golden_patchsample()
    if "patchsamplewhole" in which:
        golden_patchsample_whole()
    if "ops" in which:
        golden_ops()
    if "nets" in which:
        golden_nets()
    if "step" in which:
        golden_step()
    if "resnet" in which:
        golden_resnet()
    if "global" in which:
        golden_global()
    if "local" in which:
        golden_local()
    if "local2" in which:      # two local enhancers (--n_local_enhancers 2): three pyramid levels
        golden_local(seed=717, n_local=2)
    if "p2p" in which:
        golden_p2p_step()
    if "metrics" in which:
        golden_metrics()
    if "sg2" in which:
        golden_sg2()
    if "sg2step" in which:
        golden_step_sg2d()
    if "style" in which:
        golden_nets_style()
    if "io" in which:
        golden_io()
    if "friction" in which:
        golden_friction()
    if "sifid" in which:
        golden_sifid()
    if "sg2g" in which:
        golden_sg2g()
    if "stylemodes" in which:
        golden_nets_style_modes()
    if "lpips" in which:
        golden_lpips_step()
    if "lpipsmetrics" in which:
        golden_lpips_metrics()
    if "p2pvgg" in which:
        golden_p2p_vgg_step()
    if "p2pvanilla" in which:
        # gan_mode 'vanilla' (the discriminators end in a Sigmoid and BCEWithLogits follows, networks.py:1659, 507-509) at PatchGAN depth 2
        golden_p2p_step(seed=525, steps=1, extra=("--gan_mode", "vanilla", "--n_layers_D", "2"), fname="pix2pixHD_vanilla_step_%d.npz", n_layers=2)
    if "d3warmup" in which:
        golden_step_d3_warmup()
    if "pool" in which:
        golden_image_pool()
    if "p2ppool" in which:
        golden_p2p_pool_step()
    if "p2pfix" in which:
        golden_p2p_fix_global()
    if "diffaug" in which:
        golden_diffaug()
    if "variants" in which:
        golden_step_variants()
    if "fullgrads" in which:
        golden_step_full_grads()
