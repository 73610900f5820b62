"""CPU restatement of one SinSKITGModel training step (TEST INFRASTRUCTURE ONLY).

Restates /root/reference/models/sinskitG_model.py:
  set_input :702-793, forward :1293-1344, compute_additional_output :1268-1291,
  optimize_parameters :601-700, compute_D1_loss :1346-1407, compute_D2_loss :1409-1617,
  compute_G1_loss :1660-1726, compute_G2_loss :1728-1842
with LPIPS / vision-aided / CLIP terms off (lambda 0; "parity unpinned" third-party
terms) and generalised from the reference's hard N=1 to N>=1 by
looping the patch gather per sample (SURVEY.md §7 "Batch > 1").

Random draws are *inputs* (`draws`), so the oracle, the reference run that made
the golden vectors, and the HIP path all consume identical numbers:
  draws["aug"]      float32 [4, N]  = (brightness real, saturation real, brightness fake, saturation fake)
                    in the order DiffAugment consumes torch.rand (:1330-1333)
  draws["more_idx"] int64 [N, n_more] indices into the row-major nonzero list of the
                    dilated mask (model_utils.py:212-222, random.sample)

Pinned by tests/test_oracle_golden.py against tests/golden/sinskitG_step_*.npz.
Only tests/, smoke() and bench.py's cpu_baseline leg may import this module.
"""
import copy
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import nets

DEFAULT_HP = dict(
    lambda_G1_GAN=1.0, lambda_G1_L1=100.0, lambda_G2_GAN=5.0, lambda_G2_L1=10.0,
    lr=1e-3, lr_G2=5e-4, beta1=0.0, beta2=0.99, gan_mode="nonsaturating",
    batch_size_G2=64, add_fake_T_sample_size=32, scale_nz=0.25, num_D=3,
    use_more_fakeT=True, use_diffaug=True, lr_scale=1.0, netG="unet256_custom",
    lambda_G1_lpips=0.0, lambda_G2_lpips=0.0,
    n_layers_D=3, n_layers_D2=3, smooth_GAN_label=True, diffaugment="bs",
    # conditioning of the discriminators (sinskitG_model.py:525-559, 1359, 1372, 1481-1486, 1521-1560, 1578-1582, 1671, 1775-1779): D1 sees
    # cat(S, I) or I alone; the D2 stacks are [T] + [S] + [I, mask].  (use_cGAN_G2 False and use_bg_mask False cannot run upstream:
    # define_D is called without opt (networks.py:1658) / set_input never sets self.M (:638) -- probed, not restated.)
    use_cGAN=True, use_cGAN_G2_S=True, use_cGAN_G2_I=True,
)


def hp(**kw):
    d = dict(DEFAULT_HP)
    d.update(kw)
    return SimpleNamespace(**d)


def _f(t):
    """to the working precision: fp32 (the reference's), or float64 when a test evaluates the oracle under torch.set_default_dtype(float64)
    to judge fp32 results (the HIP path's and this oracle's own) against exact arithmetic"""
    return t.to(torch.get_default_dtype())


def prepare_input(batch):
    """set_input: mask multiply, SPE, patch reshape."""
    S = _f(batch["S"])
    M = _f(batch["M"])
    I = _f(batch["I"])
    n, _, h, w = S.shape
    real_S = S * M
    real_I = I * M
    S_pe = _f(nets.spe_grid(n, h, w, 4))      # (evaluated in fp32 like the reference's buffer, then an INPUT at the working precision)
    T = _f(torch.as_tensor(batch["T_images"]))
    K = _f(torch.as_tensor(batch["I_masks"]))
    nt = T.shape[1]
    masks = K.reshape(-1, 1, 32, 32)
    real_T = T.reshape(-1, 2, 32, 32) * masks
    coords = torch.as_tensor(batch["T_coords"]).numpy()
    return SimpleNamespace(real_S=real_S, real_I=real_I, M=M, S_pe=S_pe, real_T=real_T, masks=masks,
                           coords=coords, N=n, NT=nt)


def _gather_all(img, coords):
    outs = []
    for n in range(img.shape[0]):
        ox, oy, _ = nets.find_coords_for_patch(coords[n])
        outs.append(nets.gather_patches(img[n:n + 1], ox, oy, 32))
    return torch.cat(outs, 0)


def generator_forward(sdG, inp, opt, style_code=None, dropout_masks=None):
    netG = getattr(opt, "netG", "unet256_custom")
    if netG.startswith("resnet_"):   # --netG resnet_{4,6,9}blocks (sinskitG_model.py:509-520 -> networks.define_G)
        out = nets.resnet_forward(sdG, torch.cat((inp.real_S, inp.S_pe), 1), n_blocks=int(netG[len("resnet_")]), dropout_masks=dropout_masks)
    else:
        out = nets.unet_forward(sdG, torch.cat((inp.real_S, inp.S_pe), 1), style_code=style_code, dropout_masks=dropout_masks)
    fake_I = out[:, 0:3] * inp.M
    fake_T = out[:, -2:] * inp.M
    return out, fake_I, fake_T


def _req(sd, flag):
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(flag)
            v.grad = None


def _grads(sd):
    return {k: v.grad.detach().clone() for k, v in sd.items() if v.requires_grad and v.grad is not None}


def _adam(sd, state, lr, opt):
    state["step"] += 1
    for k, p in sd.items():
        if not p.requires_grad or p.grad is None:
            continue
        if k not in state["m"]:
            state["m"][k] = torch.zeros_like(p)
            state["v"][k] = torch.zeros_like(p)
        with torch.no_grad():
            nets.adam_update(p, p.grad, state["m"][k], state["v"][k], state["step"], lr, opt.beta1, opt.beta2)


def new_adam_state():
    return {"step": 0, "m": {}, "v": {}}


def _d1(sdD, x, opt):
    """netD: the multiscale PatchGAN (default) or the StyleGAN2 discriminator (--netD stylegan2, networks.py:437-442)"""
    if getattr(opt, "netD", "multiscale") == "stylegan2":
        from oracle import stylegan2
        return [[stylegan2.discriminator_forward(sdD, x, x.shape[-1])]]
    return nets.msd_forward(sdD, x, opt.num_D, n_layers=getattr(opt, "n_layers_D", 3), use_sigmoid=opt.gan_mode == "vanilla")


def _d2(sdD2, x, opt):
    """netD2: MultiscaleDiscriminator(n_layers_D2), with the trailing Sigmoid the reference adds for gan_mode 'vanilla' (networks.py:1659)"""
    return nets.msd_forward(sdD2, x, opt.num_D, n_layers=getattr(opt, "n_layers_D2", 3), use_sigmoid=opt.gan_mode == "vanilla")


def _exchange(sd, name, exchange):
    """data-parallel runs replace every local gradient by the mean over the ranks between backward and Adam; `exchange(name, grads)`
    returns that mean for network `name` given this rank's gradients (the test harness supplies it: tests/test_ddp_step_gpu.py)"""
    if exchange is None:
        return
    new = exchange(name, _grads(sd))
    for k, g in new.items():
        sd[k].grad = g.clone()


def train_step(sdG, sdD, sdD2, adam, batch, draws, opt=None, style_code=None, record=True, exchange=None, lpips=None):
    """One G+D1+D2 update, in place on the three state dicts and `adam` (dict of 3 Adam states).
    exchange: optional hook emulating the data-parallel gradient exchange (see _exchange); recorded gradients are the LOCAL ones.
    lpips: the LPIPS module (oracle.perceptual.LPIPS) when opt.lambda_G1_lpips / lambda_G2_lpips > 0 (compute_G1_loss :1709-1716,
    compute_G2_loss :1819-1838 -> _compute_touch_lpips_loss :1619-1658).

    Returns a dict with losses, outputs and (if record) the gradients taken at each of
    the three backward points.
    """
    opt = opt or hp()
    out = {}
    inp = prepare_input(batch)
    N, NT = inp.N, inp.NT
    lamD1, lamD2 = opt.lambda_G1_GAN, opt.lambda_G2_GAN
    # label smoothing: GANLoss(gan_mode, target_real_label=0.8) (sinskitG_model.py:485-488); only lsgan / vanilla read the labels
    real_label = 0.8 if getattr(opt, "smooth_GAN_label", True) else 1.0
    gl = lambda p, real: nets.gan_loss(p, real, opt.gan_mode, real_label=real_label)

    # ---- forward (G requires grad for the later G step) ----
    _req(sdG, True)
    _req(sdD, False)
    _req(sdD2, False)
    g_out, fake_I, fake_T = generator_forward(sdG, inp, opt, style_code, dropout_masks=draws.get("dropout"))
    if opt.use_diffaug and "aug_policy" in draws:
        # any policy over b / s / c / t / o / n: draws["aug_policy"] = (nets.diffaug_draws for real_I, then for fake_I) (:1330-1333)
        pol = getattr(opt, "diffaugment", "bs")
        aug_real_I = nets.diffaug(inp.real_I, pol, draws["aug_policy"][0]) * inp.M
        aug_fake_I = nets.diffaug(fake_I, pol, draws["aug_policy"][1]) * inp.M
    elif opt.use_diffaug:
        aug = _f(draws["aug"])
        aug_real_I = nets.diffaug_bs(inp.real_I, aug[0], aug[1]) * inp.M
        aug_fake_I = nets.diffaug_bs(fake_I, aug[2], aug[3]) * inp.M
    else:
        aug_real_I, aug_fake_I = inp.real_I * inp.M, fake_I * inp.M

    # ---- patches (compute_additional_output) ----
    fake_T_concat = _gather_all(fake_T, inp.coords)
    S_concat = _gather_all(inp.real_S, inp.coords).detach()
    real_I_concat = torch.cat([_gather_all(aug_real_I, inp.coords).detach(), inp.masks], 1)
    fake_I_concat = torch.cat([_gather_all(aug_fake_I, inp.coords).detach(), inp.masks], 1)
    fake_I_full = torch.cat([aug_fake_I.detach(), inp.M], 1)

    cS, cI = bool(getattr(opt, "use_cGAN_G2_S", True)), bool(getattr(opt, "use_cGAN_G2_I", True))
    d1_in = (lambda img: torch.cat((inp.real_S, img), 1)) if getattr(opt, "use_cGAN", True) else (lambda img: img)
    stack = lambda t, s_, i_: torch.cat([t] + ([s_] if cS else []) + ([i_] if cI else []), 1)      # noqa: E731

    # ---- D1 step ----
    _req(sdD, True)
    pred_fake = _d1(sdD, d1_in(fake_I.detach()), opt)
    loss_D_fake_I = gl(pred_fake, False).mean() * lamD1
    pred_real = _d1(sdD, d1_in(inp.real_I), opt)
    loss_D_real_I = gl(pred_real, True).mean() * lamD1
    loss_D1 = (loss_D_fake_I + loss_D_real_I) * 0.5
    loss_D1.backward()
    if record:
        out["grad_D"] = _grads(sdD)
        out["pred_fake_I"] = [p[-1].detach().clone() for p in pred_fake]
    _exchange(sdD, "D", exchange)
    _adam(sdD, adam["D"], opt.lr * opt.lr_scale, opt)
    _req(sdD, False)

    # ---- D2 step ----
    _req(sdD2, True)
    fake_stack = stack(fake_T_concat.detach(), S_concat, fake_I_concat)
    pred_fake_T = _d2(sdD2, fake_stack, opt)
    loss_D_fake_T = gl(pred_fake_T, False).mean() * lamD2
    full_stack = stack(fake_T.detach(), inp.real_S, fake_I_full)
    pred_full = _d2(sdD2, full_stack, opt)  # visualisation only; still updates BN stats
    loss_D_more = torch.zeros(())
    if opt.use_more_fakeT:
        stacks = []
        for n in range(N):
            pos = nets.dilated_mask_positions(inp.M[n:n + 1])
            sel = pos[draws["more_idx"][n].long()]
            oy, ox = sel[:, 0].to(torch.int32), sel[:, 1].to(torch.int32)
            t = nets.gather_patches(fake_T[n:n + 1].detach(), ox, oy, 32)
            s = nets.gather_patches(inp.real_S[n:n + 1], ox, oy, 32)
            i = nets.gather_patches(fake_I[n:n + 1].detach(), ox, oy, 32)
            stacks.append(stack(t, s, torch.cat((i, torch.ones_like(s)), 1)))
        more_stack = torch.cat(stacks, 0)
        pred_more = _d2(sdD2, more_stack, opt)
        loss_D_more = gl(pred_more, False).mean() * lamD2
    real_stack = stack(inp.real_T, S_concat, real_I_concat)
    pred_real_T = _d2(sdD2, real_stack, opt)
    loss_D_real_T = gl(pred_real_T, True).mean() * lamD2
    loss_D2 = (loss_D_fake_T + loss_D_more + loss_D_real_T) * 0.5
    loss_D2.backward()
    if record:
        out["grad_D2"] = _grads(sdD2)
        out["pred_fake_T_full"] = pred_full[-1][-1].detach().clone()
    _exchange(sdD2, "D2", exchange)
    _adam(sdD2, adam["D2"], opt.lr_G2 * opt.lr_scale, opt)
    _req(sdD2, False)

    # ---- G step ----
    pred_g = _d1(sdD, d1_in(fake_I), opt)
    loss_G_GAN = gl(pred_g, True).mean() * lamD1
    loss_G_L1 = F.l1_loss(fake_I, inp.real_I) * opt.lambda_G1_L1
    g2_stack = stack(fake_T_concat.clone().detach(), S_concat, fake_I_concat)
    pred_g2 = _d2(sdD2, g2_stack, opt)
    loss_G2_GAN = (gl(pred_g2, True) * lamD2).view(-1, NT).mean(dim=0).sum()  # logged only: no gradient path
    l1 = (fake_T_concat - inp.real_T).abs() * opt.lambda_G2_L1
    loss_G2_L1 = l1.view(-1, NT, 2, 32, 32).sum(dim=1).mean()
    loss_G_lpips = loss_G2_lpips = torch.zeros(())
    if getattr(opt, "lambda_G1_lpips", 0.0) > 0.0:
        loss_G_lpips = lpips(fake_I, inp.real_I).mean() * opt.lambda_G1_lpips
    if getattr(opt, "lambda_G2_lpips", 0.0) > 0.0:
        from oracle import perceptual
        loss_G2_lpips = perceptual.touch_lpips(lpips, fake_T_concat, inp.real_T, NT, opt.lambda_G2_lpips)
    loss_G = loss_G_GAN + loss_G_L1 + loss_G2_L1 + loss_G2_GAN.detach() * 0 + loss_G_lpips + loss_G2_lpips
    loss_G.backward()
    if record:
        out["grad_G"] = _grads(sdG)
    _exchange(sdG, "G", exchange)
    _adam(sdG, adam["G"], opt.lr * opt.lr_scale, opt)
    _req(sdG, False)

    out["losses"] = {
        "G_GAN": float(loss_G_GAN.detach()), "D_real_I": float(loss_D_real_I.detach()), "D_fake_I": float(loss_D_fake_I.detach()),
        "G_L1": float(loss_G_L1.detach()), "G2_GAN": float(loss_G2_GAN.detach()), "D_real_T_concat": float(loss_D_real_T.detach()),
        "D_fake_T_concat": float(loss_D_fake_T.detach()), "D_more_fake_T": float(loss_D_more.detach()), "G2_L1": float(loss_G2_L1.detach()),
    }
    if getattr(opt, "lambda_G1_lpips", 0.0) > 0.0:
        out["losses"]["G_lpips"] = float(loss_G_lpips.detach())
    if getattr(opt, "lambda_G2_lpips", 0.0) > 0.0:
        out["losses"]["G2_lpips"] = float(loss_G2_lpips.detach())
    if record:
        out["g_out"] = g_out.detach().clone()
        out["fake_I"] = fake_I.detach().clone()
        out["fake_T"] = fake_T.detach().clone()
        out["fake_N"] = nets.compute_normal(fake_T.detach(), opt.scale_nz)
        out["aug_real_I"] = aug_real_I.detach().clone()
        out["aug_fake_I"] = aug_fake_I.detach().clone()
        out["fake_T_concat"] = fake_T_concat.detach().clone()
    return out


def clone_sd(sd):
    return {k: v.detach().clone() for k, v in sd.items()}


def inference(sdG, batch, opt=None, style_code=None):
    """test(): no_grad forward (sinskitG_model.py:795-807)."""
    opt = opt or hp()
    inp = prepare_input(batch) if "T_images" in batch else None
    with torch.no_grad():
        _, fake_I, fake_T = generator_forward(sdG, inp, opt, style_code)
    return fake_I, fake_T


# ----------------------------------------------------------------------------
# pix2pixHD baseline step  (models/pix2pixHD_model.py:431-509 set_input, :587-619 forward, :621-644 backward_D,
# :646-700 backward_G, :702-722 optimize_parameters).  Patch training: S/M/I/T_images are [N,*,32,32].
# The GAN feature-matching term compares every discriminator feature WITH ITSELF (.detach()) (:662-680): its
# value and its gradient are identically zero, so it is reported as 0 and contributes nothing; VGG / LPIPS terms
# need pretrained weights (off: --no_vgg_loss True).
# ----------------------------------------------------------------------------
P2P_HP = dict(lr=2e-4, beta1=0.5, beta2=0.999, gan_mode="lsgan", n_blocks_global=9, n_downsample_global=4, num_D_D1=2, num_D_D2=2,
              scale_nz=0.25, n_layers_D=3)


def p2p_hp(**kw):
    d = dict(P2P_HP)
    d.update(kw)
    return SimpleNamespace(**d)


def p2p_prepare(batch):
    S, M, I = batch["S_images"].float(), batch["M_images"].float(), batch["I_images"].float()
    h, w = S.shape[-2:]
    T = batch["T_images"].reshape(-1, 2, h, w).float()
    masks = batch["I_masks"].reshape(-1, 1, h, w).float()
    return SimpleNamespace(real_S=S * M, real_I=I * M, M=M, real_T=T * masks)


def p2p_generator(sdG, inp, opt, training=True):
    out = nets.resnet_forward(sdG, inp.real_S, opt.n_blocks_global, opt.n_downsample_global, norm="batch", down="stride", up="convT",
                              training=training)
    return out, out[:, :3] * inp.M, out[:, -2:] * inp.M


def p2p_train_step(sdG, sdD, sdD2, adam, batch, opt=None, record=True, vgg_loss=None, lambda_vgg=10.0):
    """One D + D2 + G update in place on the three state dicts (D / D2 in the getIntermFeat key style) and `adam`.
    vgg_loss: oracle.perceptual.VGGLoss for the VGG feature term (pix2pixHD_model.py:680-693: the image, and gx / gy tiled to three
    channels), None = --no_vgg_loss True."""
    opt = opt or p2p_hp()
    inp = p2p_prepare(batch)
    nl = getattr(opt, "n_layers_D", 3)
    pD, pD2 = nets.d_if_to_plain(sdD, nl), nets.d_if_to_plain(sdD2, nl)     # views onto the same tensors
    gl = lambda p, real: nets.gan_loss(p, real, opt.gan_mode)
    # both discriminators: depth n_layers_D.  No Sigmoid even for gan_mode 'vanilla': with getIntermFeat (this model's key style) the
    # reference registers only the first n_layers + 2 blocks of NLayerDiscriminator's sequence (networks.py:1664-1666) -- the trailing
    # [Sigmoid] block is dropped, unlike in the fused `layer<i>` form (:1668) the sinskitG discriminators use
    dkw = dict(n_layers=nl, use_sigmoid=False)
    _req(sdG, True)
    g_out, fake_I, fake_T = p2p_generator(sdG, inp, opt)
    # ---- D, D2 (one backward for both) ----
    _req(sdD, True)
    _req(sdD2, True)
    loss_D_fake = gl(nets.msd_forward(pD, torch.cat((inp.real_S, fake_I.detach()), 1), opt.num_D_D1, **dkw), False)
    loss_D_real = gl(nets.msd_forward(pD, torch.cat((inp.real_S, inp.real_I), 1), opt.num_D_D1, **dkw), True)
    loss_D2_fake = gl(nets.msd_forward(pD2, torch.cat((inp.real_S, fake_T.detach()), 1), opt.num_D_D2, **dkw), False)
    loss_D2_real = gl(nets.msd_forward(pD2, torch.cat((inp.real_S, inp.real_T), 1), opt.num_D_D2, **dkw), True)
    ((loss_D_fake + loss_D_real) * 0.5 + (loss_D2_fake + loss_D2_real) * 0.5).backward()
    out = {}
    if record:
        out["grad_D"], out["grad_D2"] = _grads(sdD), _grads(sdD2)
    _adam(sdD, adam["D"], opt.lr, opt)
    _adam(sdD2, adam["D2"], opt.lr, opt)
    _req(sdD, False)
    _req(sdD2, False)
    # ---- G ----
    loss_G_I = gl(nets.msd_forward(pD, torch.cat((inp.real_S, fake_I), 1), opt.num_D_D1, **dkw), True)
    loss_G_T = gl(nets.msd_forward(pD2, torch.cat((inp.real_S, fake_T), 1), opt.num_D_D2, **dkw), True)
    loss_vgg_I = loss_vgg_T = torch.zeros(())
    if vgg_loss is not None:
        loss_vgg_I = vgg_loss(fake_I, inp.real_I) * lambda_vgg
        t3 = lambda t, c: t[:, c:c + 1].expand(-1, 3, -1, -1)
        loss_vgg_T = (vgg_loss(t3(fake_T, 0), t3(inp.real_T, 0)) + vgg_loss(t3(fake_T, 1), t3(inp.real_T, 1))) * lambda_vgg
    (loss_G_I + loss_G_T + loss_vgg_I + loss_vgg_T).backward()
    if record:
        out["grad_G"] = _grads(sdG)
    _adam(sdG, adam["G"], opt.lr, opt)
    _req(sdG, False)
    out["losses"] = {"G_GAN_I": float(loss_G_I.detach()), "G_GAN_T": float(loss_G_T.detach()), "G_GAN": float((loss_G_I + loss_G_T).detach()),
                     "D_real": float(loss_D_real.detach()), "D_fake": float(loss_D_fake.detach()), "D2_real": float(loss_D2_real.detach()),
                     "D2_fake": float(loss_D2_fake.detach()), "G_GAN_Feat": 0.0, "G_GAN_Feat_I": 0.0, "G_GAN_Feat_T": 0.0}
    if vgg_loss is not None:
        out["losses"].update({"G_VGG_I": float(loss_vgg_I.detach()), "G_VGG_T": float(loss_vgg_T.detach()), "G_VGG": float((loss_vgg_I + loss_vgg_T).detach())})
    if record:
        out["fake_I"], out["fake_T"] = fake_I.detach().clone(), fake_T.detach().clone()
        out["fake_N"] = nets.compute_normal(fake_T.detach(), opt.scale_nz)
    return out
