"""Network definitions of the hot path: parameter containers + factories.

Mirrors the constructor surface of the reference (`define_G`, `define_D`, `init_weights`,
`get_scheduler`, `GANLoss`; /root/reference/models/networks.py:148-252, 255-325, 392-442,
448-542) and emits exactly the reference's `state_dict` keys and shapes
(SURVEY.md §8b "Checkpoint format"), so released checkpoints load.  The modules hold
parameters only: all arithmetic is done by vts.engine over libvts_hip.so -- calling
`forward` runs the HIP path (there is no eager fallback).
"""
import math

import torch
import torch.nn as nn
from torch.optim import lr_scheduler

from vts import engine


class _Holder(nn.Module):
    """A numbered-children container (state_dict keys '<idx>.<param>')."""

    def __init__(self, children):
        super().__init__()
        for name, mod in children.items():
            self.add_module(str(name), mod)


class _ConvParams(nn.Module):
    def __init__(self, shape, bias_n, transposed=False):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(shape))
        self.bias = nn.Parameter(torch.zeros(bias_n)) if bias_n else None
        self.transposed = transposed


class _BNParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Block(nn.Module):
    """`downN` / `upN` wrapper whose only child is `model` (keys 'downN.model.K.weight')."""

    def __init__(self, idx, conv):
        super().__init__()
        self.model = _Holder({idx: conv})
        self.conv_idx = str(idx)

    @property
    def conv(self):
        return getattr(self.model, self.conv_idx)


class CustomUnetGenerator(nn.Module):
    """U-Net with dual visual / tactile decoders (reference: networks.py:1430-1645,
    thirdparty/unet/unet_parts_custom.py:9-79).  InstanceNorm only (the hot-path default)."""

    def __init__(self, input_nc, output_nc, num_downs=8, ngf=64, num_layer_separate=0, opt=None):
        super().__init__()
        assert output_nc == 5, "current architecture is designed specifically for 5 output channels, 3 - RGB, 2 - touch"
        assert 0 <= num_layer_separate <= num_downs
        self.input_nc, self.num_downs, self.ngf, self.num_layer_separate = input_nc, num_downs, ngf, num_layer_separate
        self.opt = opt
        use_style = bool(opt is not None and getattr(opt, "use_style_code", False))
        self.use_style = use_style
        self.num_layer_style_code = 0
        self.style_nc = 0
        if use_style:
            if getattr(opt, "style_code_mode", "concat") != "concat" or getattr(opt, "style_code_mapping_mode", "tile") != "tile":
                raise NotImplementedError("style code: only mode=concat / mapping=tile is built (SURVEY.md §8 a5)")
            nl = getattr(opt, "num_layer_style_code", -1)
            self.num_layer_style_code = num_downs if nl == -1 else nl
            self.style_nc = opt.style_code_dim
            # the reference also constructs (never uses) Linear style_code_mapping<i> layers in tile mode
            # (networks.py:1446-1465); they are dead parameters and are not created here.
        ch = [ngf * min(2 ** i, 8) for i in range(num_downs)]
        self.channels = ch

        def style_extra(i):
            return self.style_nc if (use_style and i >= num_downs - self.num_layer_style_code) else 0

        def up(i, outer):
            inner = ch[i] * (1 if i in (0, num_downs - 1) else 2) + style_extra(i)
            return _Block(1, _ConvParams((inner, outer, 4, 4), outer, transposed=True))

        self.down0 = _Block(0, _ConvParams((ch[0], input_nc, 4, 4), ch[0]))
        self.up0 = _Block(1, _ConvParams((ch[0] + style_extra(0), 3 if num_layer_separate > 0 else 5, 4, 4),
                                         3 if num_layer_separate > 0 else 5, transposed=True))
        if num_layer_separate >= 1:
            self.up0_T = _Block(1, _ConvParams((ch[0] + style_extra(0), 2, 4, 4), 2, transposed=True))
        for i in range(1, num_downs):
            setattr(self, "down%d" % i, _Block(1, _ConvParams((ch[i], ch[i - 1], 4, 4), ch[i])))
            setattr(self, "up%d" % i, up(i, ch[i - 1]))
            if num_layer_separate >= i + 1:
                setattr(self, "up%d_T" % i, up(i, ch[i - 1]))

    def forward(self, x, style_code=None, verbose=False):
        """Inference forward on the HIP path; returns [N,5,H,W]."""
        out, _ = engine.unet_forward(self, x, style_code=style_code, keep=False)
        return out


class MultiscaleDiscriminator(nn.Module):
    """num_D PatchGANs over an average-pooled pyramid (reference: networks.py:1649-1750).
    BatchNorm2d(affine, running stats) as in the hot-path default (normD=batch)."""

    CONV_IDX = (0, 2, 5, 8, 11)
    BN_IDX = {2: 3, 5: 6, 8: 9}
    STRIDE = {0: 2, 2: 2, 5: 2, 8: 1, 11: 1}

    def __init__(self, input_nc, ndf=64, n_layers=3, num_D=3, opt=None):
        super().__init__()
        if n_layers != 3:
            raise NotImplementedError("MultiscaleDiscriminator: only n_layers=3 is built")
        self.input_nc, self.ndf, self.num_D = input_nc, ndf, num_D
        chans = [input_nc, ndf, min(ndf * 2, 512), min(ndf * 4, 512), min(ndf * 8, 512), 1]
        self.chans = chans
        for d in range(num_D):
            children = {}
            for j, ci in enumerate(self.CONV_IDX):
                children[ci] = _ConvParams((chans[j + 1], chans[j], 4, 4), chans[j + 1])
                if ci in self.BN_IDX:
                    children[self.BN_IDX[ci]] = _BNParams(chans[j + 1])
            setattr(self, "layer%d" % d, _Holder(dict(sorted(children.items()))))

    def forward(self, x):
        """Returns [[pred_scale0], [pred_scale1], ...] like the reference."""
        preds, _ = engine.msd_forward(self, x, None, keep=False)
        return [[p] for p in preds]


def init_weights(net, init_type="normal", init_gain=0.02):
    """networks.py:191-231: conv weights by `init_type`, biases 0, BatchNorm weight ~ N(1, gain)."""
    for m in net.modules():
        if isinstance(m, _ConvParams):
            if init_type == "normal":
                nn.init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                nn.init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                nn.init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                nn.init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif isinstance(m, _BNParams):
            nn.init.normal_(m.weight.data, 1.0, init_gain)
            nn.init.constant_(m.bias.data, 0.0)


def init_net(net, init_type="normal", init_gain=0.02, gpu_ids=(), initialize_weights=True):
    if initialize_weights:
        init_weights(net, init_type, init_gain)
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(torch.device("cuda", gpu_ids[0]))
    return net


def define_G(input_nc, output_nc, ngf, netG, norm="batch", use_dropout=False, init_type="normal", init_gain=0.02,
             no_antialias=False, no_antialias_up=False, gpu_ids=(), opt=None, generate_T_imgs=False, num_layer_separate=0):
    if netG != "unet256_custom":
        raise NotImplementedError("Generator model name [%s] is not recognized (built: unet256_custom)" % netG)
    if norm != "instance":
        raise NotImplementedError("unet256_custom is built for normG=instance only")
    net = CustomUnetGenerator(input_nc, output_nc, num_downs=8, ngf=ngf, num_layer_separate=num_layer_separate, opt=opt)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm="batch", init_type="normal", init_gain=0.02, no_antialias=False,
             num_D=1, gpu_ids=(), opt=None):
    if netD != "multiscale":
        raise NotImplementedError("Discriminator model name [%s] is not recognized (built: multiscale)" % netD)
    if norm != "batch":
        raise NotImplementedError("multiscale discriminator is built for normD=batch only")
    net = MultiscaleDiscriminator(input_nc, ndf, n_layers_D, num_D=num_D, opt=opt)
    return init_net(net, init_type, init_gain, gpu_ids)


def get_scheduler(optimizer, opt):
    """networks.py:148-174."""
    if opt.lr_policy == "linear":
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.n_epochs) / float(opt.n_epochs_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == "cosine":
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.n_epochs, eta_min=0)
    raise NotImplementedError("learning rate policy [%s] is not implemented" % opt.lr_policy)


class GANLoss:
    """GAN objectives over (lists of) discriminator predictions (networks.py:448-542).

    `__call__` returns the per-call loss as a 1-element device tensor (forward only);
    `accumulate` is what the training step uses: value and d/dpred in one kernel pass per scale.
    """

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        from vts import lib as L

        if gan_mode not in L.GAN_MODES:
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)
        self.gan_mode, self.real_label, self.fake_label = gan_mode, target_real_label, target_fake_label

    def to(self, device):
        return self

    def accumulate(self, preds, target_is_real, coeff, slot, grad_coeff=None, want_grad=True):
        """slot += coeff * sum_scales mean_batch(loss); returns [dpred per scale] scaled by grad_coeff."""
        from vts import ops

        grads = []
        label = self.real_label if target_is_real else self.fake_label
        for p in preds:
            p = p[-1] if isinstance(p, (list, tuple)) else p
            g = torch.empty_like(p) if want_grad else None
            ops.ganloss(p, self.gan_mode, target_is_real, coeff, slot, g, label=label, grad_coeff=grad_coeff)
            grads.append(g)
        return grads

    def __call__(self, preds, target_is_real):
        first = preds[0][-1] if isinstance(preds[0], (list, tuple)) else preds[-1]
        slot = torch.zeros(1, device=first.device)
        plist = preds if isinstance(preds[0], (list, tuple)) else [preds[-1]]
        self.accumulate(plist, target_is_real, 1.0, slot, want_grad=False)
        return slot
