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


class _LinearParams(nn.Module):
    """nn.Linear(.., bias=False): initialised like the convolutions by init_weights (networks.py:200: 'Conv' or 'Linear' in the class name)"""

    def __init__(self, out_f, in_f):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(out_f, in_f))
        self.bias = None


class _BN1dParams(_BNParams):
    """nn.BatchNorm1d: init_weights leaves it at weight 1 / bias 0 (networks.py:218 matches 'BatchNorm2d' only)"""


STYLE_INPUT_SIZE = 1536    # CustomUnetGenerator(input_size=1536) (networks.py:1432); define_G never passes another value (:307)


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

    def __init__(self, input_nc, output_nc, num_downs=8, ngf=64, num_layer_separate=0, opt=None, use_dropout=False):
        super().__init__()
        assert output_nc == 5, "current architecture is designed specifically for 5 output channels, 3 - RGB, 2 - touch"
        assert 0 <= num_layer_separate <= num_downs
        self.input_nc, self.num_downs, self.ngf, self.num_layer_separate = input_nc, num_downs, ngf, num_layer_separate
        # Dropout(0.5) behind the InstanceNorm of the intermediate Up blocks up<num_downs // 2> .. up<num_downs - 2> (reference
        # networks.py:1508-1519 -> unet_parts_custom.py:66-67), active in train() mode only.  Built for the shared trunk (the blocks
        # that have no `_T` twin: num_layer_separate <= num_downs // 2, the reference's default 4 of 8).
        self.use_dropout = bool(use_dropout)
        if self.use_dropout and num_layer_separate > num_downs // 2:
            raise NotImplementedError("Up-block dropout with num_layer_separate > num_downs // 2 (dropout inside the separate decoders)")
        self.opt = opt
        use_style = bool(opt is not None and getattr(opt, "use_style_code", False))
        self.use_style = use_style
        self.num_layer_style_code = 0
        self.style_nc = 0
        self.style_mode, self.style_mapping = "concat", "tile"
        if use_style:
            self.style_mode = getattr(opt, "style_code_mode", "concat")
            self.style_mapping = getattr(opt, "style_code_mapping_mode", "tile")
            if self.style_mode not in ("concat", "adain") or self.style_mapping not in ("tile", "project"):
                raise NotImplementedError("style code mode %s / mapping %s (networks.py:1608-1632)" % (self.style_mode, self.style_mapping))
            if self.style_mode == "adain" and self.style_mapping == "tile":
                raise ValueError("style_code_mode adain needs style_code_mapping_mode project: the tiled 512-channel code cannot match the "
                                 "content's channels (thirdparty/AdaIN/function.py:16)")
            nl = getattr(opt, "num_layer_style_code", -1)
            self.num_layer_style_code = num_downs if nl == -1 else nl
            # channels of the style map of layer i (networks.py:1446-1457): the code itself (tile), ngf // 2 (project), ngf * 8 (adain)
            self.style_nc = ngf * 8 if self.style_mode == "adain" else (opt.style_code_dim if self.style_mapping == "tile" else ngf // 2)
            if self.style_mapping == "project":
                # style_code_mapping<j> = Linear(style_dim, out_size^2 * nc, bias=False) -> BatchNorm1d | InstanceNorm1d -> ReLU for
                # up layer i = num_downs - 1 - j (networks.py:1444-1465).  In tile mode the reference creates the same Linear layers
                # and never uses them (dead parameters: not created here).
                for j in range(self.num_layer_style_code):
                    side = STYLE_INPUT_SIZE // (2 ** (num_downs - j))
                    mods = {0: _LinearParams(side * side * self.style_nc, opt.style_code_dim)}
                    if getattr(opt, "batch_size", 1) > 1:
                        mods[1] = _BN1dParams(side * side * self.style_nc)
                    setattr(self, "style_code_mapping%d" % j, _Holder(mods))
        ch = [ngf * min(2 ** i, 8) for i in range(num_downs)]
        self.channels = ch

        def style_extra(i):     # extra input channels of up layer i: none in adain mode (networks.py:1477-1480)
            return self.style_nc if (use_style and self.style_mode != "adain" and i >= num_downs - self.num_layer_style_code) else 0

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


class _Filt(nn.Module):
    """Downsample / Upsample of the reference register their fixed blur kernel as a buffer `filt`
    (networks.py:62,99); it is part of the state_dict, so it is kept here (the HIP kernels have it built in)."""

    def __init__(self, channels, taps, scale):
        super().__init__()
        a = torch.tensor(taps, dtype=torch.float32)
        f = a[:, None] * a[None, :]
        self.register_buffer("filt", (f / f.sum() * scale)[None, None].repeat(channels, 1, 1, 1))


class ResnetGenerator(nn.Module):
    """ResNet-style generators as one parameter container (reference: ResnetGenerator networks.py:1051-1154,
    ResnetBlock :1267-1324, Downsample :51-74, Upsample :87-107; pix2pixHD GlobalGenerator :1952-1980).

      norm  'instance' | 'batch'        down  'blur' (conv3 s1 + anti-aliased Downsample) | 'stride' (conv3 s2)
      up    'blur' (Upsample + conv3) | 'convT' (ConvTranspose2d 3x3 s2 p1 op1)

    `self.layout` lists the reference's nn.Sequential so that the state_dict keys are `model.<idx>...`;
    reflect padding; optional dropout in the blocks."""

    def __init__(self, input_nc, output_nc, ngf=64, n_blocks=6, n_downsampling=2, norm="instance", down="blur", up="blur",
                 conv_bias=None, opt=None, use_dropout=False):
        super().__init__()
        assert norm in ("instance", "batch") and down in ("blur", "stride") and up in ("blur", "convT")
        self.input_nc, self.output_nc, self.ngf, self.n_blocks, self.n_down = input_nc, output_nc, ngf, n_blocks, n_downsampling
        self.norm = norm
        # use_dropout: Dropout(0.5) behind the first conv / norm / ReLU of every block (ResnetBlock.build_conv_block, networks.py:1305-1306;
        # train() mode only) -- it also shifts the second conv / norm of the block from conv_block.5 / .6 to .6 / .7 in the state dict
        self.use_dropout = bool(use_dropout)
        kb = 6 if self.use_dropout else 5
        self._block_keys = ("1", "2", str(kb), str(kb + 1))
        if conv_bias is None:          # ResnetGenerator: use_bias = (norm_layer == InstanceNorm2d)  (networks.py:1069-1073)
            conv_bias = norm == "instance"
        mods, layout = {}, []
        idx = 0

        def add(kind, mod=None, **kw):
            nonlocal idx
            if mod is not None:
                mods[idx] = mod
            layout.append(dict(kind=kind, idx=idx, **kw))
            idx += 1

        def add_norm(c):
            add("norm", _BNParams(c) if norm == "batch" else None)

        def block(c):
            kids = {1: _ConvParams((c, c, 3, 3), c if conv_bias else 0), kb: _ConvParams((c, c, 3, 3), c if conv_bias else 0)}
            if norm == "batch":
                kids[2], kids[kb + 1] = _BNParams(c), _BNParams(c)
            return _Holder({"conv_block": _Holder(kids)})

        add("pad")
        add("conv7", _ConvParams((ngf, input_nc, 7, 7), ngf if conv_bias else 0))
        add_norm(ngf); add("relu")
        for i in range(n_downsampling):
            c = ngf * 2 ** i
            add("conv3", _ConvParams((2 * c, c, 3, 3), 2 * c if conv_bias else 0), stride=2 if down == "stride" else 1)
            add_norm(2 * c); add("relu")
            if down == "blur":
                add("down", _Filt(2 * c, [1.0, 2.0, 1.0], 1.0))
        c = ngf * 2 ** n_downsampling
        for _ in range(n_blocks):
            add("block", block(c))
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            if up == "blur":
                add("up", _Filt(c, [1.0, 3.0, 3.0, 1.0], 4.0))
                add("conv3", _ConvParams((c // 2, c, 3, 3), c // 2 if conv_bias else 0), stride=1)
            else:
                add("convT3", _ConvParams((c, c // 2, 3, 3), c // 2 if conv_bias else 0, transposed=True))
            add_norm(c // 2); add("relu")
        add("pad")
        add("conv7", _ConvParams((output_nc, ngf, 7, 7), output_nc))
        add("tanh")
        self.model = _Holder(mods)
        self.layout = layout

    def mod(self, idx):
        return getattr(self.model, str(idx), None)

    def block_mods(self, idx):
        cb = getattr(self.model, str(idx)).conv_block
        return [getattr(cb, k, None) for k in self._block_keys]   # conv a, norm a, conv b, norm b

    def forward(self, x, style_code=None, verbose=False):
        """Inference forward on the HIP path; returns [N,output_nc,H,W]."""
        out, _ = engine.resnet_forward(self, x, keep=False)
        return out


class _SeqBuilder:
    """collects (module index -> parameter container) and the layout list of one nn.Sequential of the reference"""

    def __init__(self, norm, conv_bias=True):
        self.mods, self.layout, self.idx, self.norm, self.conv_bias = {}, [], 0, norm, conv_bias

    def add(self, kind, mod=None, **kw):
        if mod is not None:
            self.mods[self.idx] = mod
        self.layout.append(dict(kind=kind, idx=self.idx, **kw))
        self.idx += 1

    def conv7(self, cin, cout, final=False):
        self.add("pad")
        self.add("conv7", _ConvParams((cout, cin, 7, 7), cout if (self.conv_bias or final) else 0))
        if final:
            self.add("tanh")
        else:
            self.norm_relu(cout)

    def norm_relu(self, c):
        self.add("norm", _BNParams(c) if self.norm == "batch" else None)
        self.add("relu")

    def conv3(self, cin, cout, stride):
        self.add("conv3", _ConvParams((cout, cin, 3, 3), cout if self.conv_bias else 0), stride=stride)
        self.norm_relu(cout)

    def convT3(self, cin, cout):
        self.add("convT3", _ConvParams((cin, cout, 3, 3), cout if self.conv_bias else 0, transposed=True))
        self.norm_relu(cout)

    def block(self, c):
        kids = {1: _ConvParams((c, c, 3, 3), c if self.conv_bias else 0), 5: _ConvParams((c, c, 3, 3), c if self.conv_bias else 0)}
        if self.norm == "batch":
            kids[2], kids[6] = _BNParams(c), _BNParams(c)
        self.add("block", _Holder({"conv_block": _Holder(kids)}))


class _SeqView:
    """layout + module lookup over one _Holder (what engine._seq_forward walks); not an nn.Module"""

    def __init__(self, holder, layout):
        self._holder, self.layout = holder, layout

    def mod(self, idx):
        return getattr(self._holder, str(idx), None)

    def block_mods(self, idx):
        cb = getattr(self._holder, str(idx)).conv_block
        return [getattr(cb, k, None) for k in ("1", "2", "5", "6")]


class LocalEnhancer(nn.Module):
    """pix2pixHD LocalEnhancer (reference: networks.py:1897-1949; `define_G(netG='local')` :311-313) with L = n_local_enhancers:
    `model` = GlobalGenerator(ngf * 2^L).model without its last three layers, applied to the input average-pooled L times;
    enhancer n = 1 .. L at pyramid level L - n with f = ngf * 2^(L - n) filters: `model<n>_1` = 7x7 conv + stride-2 3x3 conv on that
    level, `model<n>_2` = ResnetBlocks + ConvTranspose2d on (its output + the output below); the last one ends in the 7x7 conv + tanh."""
    is_local_enhancer = True

    def __init__(self, input_nc, output_nc, ngf=32, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1, n_blocks_local=3,
                 norm="batch", opt=None):
        super().__init__()
        if n_local_enhancers < 1:
            raise ValueError("LocalEnhancer: n_local_enhancers must be >= 1")
        self.norm = norm
        self.n_local_enhancers = L = n_local_enhancers
        ngf_g = ngf * 2 ** L
        g = _SeqBuilder(norm)
        g.conv7(input_nc, ngf_g)
        for i in range(n_downsample_global):
            g.conv3(ngf_g * 2 ** i, ngf_g * 2 ** (i + 1), 2)
        for _ in range(n_blocks_global):
            g.block(ngf_g * 2 ** n_downsample_global)
        for i in range(n_downsample_global):
            c = ngf_g * 2 ** (n_downsample_global - i)
            g.convT3(c, c // 2)
        self.model = _Holder(g.mods)
        self.seq_global = _SeqView(self.model, g.layout)
        self.seq_down, self.seq_up = [], []          # [n - 1] -> the views of model<n>_1 / model<n>_2
        for n in range(1, L + 1):
            f = ngf * 2 ** (L - n)
            d = _SeqBuilder(norm)
            d.conv7(input_nc, f)
            d.conv3(f, f * 2, 2)
            u = _SeqBuilder(norm)
            for _ in range(n_blocks_local):
                u.block(f * 2)
            u.convT3(f * 2, f)
            if n == L:
                u.conv7(f, output_nc, final=True)
            hd, hu = _Holder(d.mods), _Holder(u.mods)
            setattr(self, "model%d_1" % n, hd)
            setattr(self, "model%d_2" % n, hu)
            self.seq_down.append(_SeqView(hd, d.layout))
            self.seq_up.append(_SeqView(hu, u.layout))
        self.seq_11, self.seq_12 = self.seq_down[0], self.seq_up[0]

    def forward(self, x, style_code=None, verbose=False):
        out, _ = engine.resnet_forward(self, x, keep=False)
        return out


class GlobalGenerator(ResnetGenerator):
    """pix2pixHD coarse generator (reference: networks.py:1952-1980; `define_G(netG='global')` :309-310):
    BatchNorm, stride-2 3x3 downsampling, ConvTranspose2d upsampling, every convolution with a bias."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm="batch", opt=None):
        super().__init__(input_nc, output_nc, ngf=ngf, n_blocks=n_blocks, n_downsampling=n_downsampling, norm=norm,
                         down="stride", up="convT", conv_bias=True, opt=opt)


def _patchgan_layout(n_layers):
    """Sequential indices of one NLayerDiscriminator (reference networks.py:1696-1737): conv 0 (stride 2); n_layers - 1 blocks
    [conv stride 2, norm, LeakyReLU]; one block [conv stride 1, norm, LeakyReLU]; conv stride 1 -> 1 channel."""
    conv = [0] + [2 + 3 * k for k in range(n_layers)] + [2 + 3 * n_layers]
    return tuple(conv), {c: c + 1 for c in conv[1:-1]}, {c: (2 if j < n_layers else 1) for j, c in enumerate(conv)}


def _patchgan_channels(input_nc, ndf, n_layers):
    ch = [input_nc, ndf]
    for _ in range(n_layers):
        ch.append(min(ch[-1] * 2, 512))
    return ch + [1]


class MultiscaleDiscriminator(nn.Module):
    """num_D PatchGANs of depth n_layers over an average-pooled pyramid (reference: networks.py:1649-1750).
    BatchNorm2d(affine, running stats) as in the hot-path default (normD=batch).  With gan_mode 'vanilla' the reference appends a
    Sigmoid to every PatchGAN (`use_sigmoid`, :1659, 1731-1732) and still feeds BCEWithLogits (:507-509): `use_sigmoid` is kept as an
    attribute, forward() returns the squashed maps and the loss kernel takes the same detour (GANLoss.accumulate(pre_sigmoid=True))."""

    CONV_IDX, BN_IDX, STRIDE = _patchgan_layout(3)      # (0, 2, 5, 8, 11) / {2: 3, 5: 6, 8: 9} / strides 2 2 2 1 1

    def __init__(self, input_nc, ndf=64, n_layers=3, num_D=3, opt=None):
        super().__init__()
        if n_layers < 1:
            raise ValueError("MultiscaleDiscriminator: n_layers must be >= 1")
        self.input_nc, self.ndf, self.num_D, self.n_layers = input_nc, ndf, num_D, n_layers
        self.use_sigmoid = getattr(opt, "gan_mode", None) == "vanilla"
        self.CONV_IDX, self.BN_IDX, self.STRIDE = _patchgan_layout(n_layers)
        chans = _patchgan_channels(input_nc, ndf, n_layers)
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
        return [[torch.sigmoid(p) if self.use_sigmoid else p] for p in preds]


class _LayerView:
    """`layer<d>`-style accessor over the scale<d>_layer<j> modules of MultiscaleDiscriminatorIF (not a Module)."""

    def __init__(self, owner, d):
        self._owner, self._d = owner, d

    def __getattr__(self, name):
        ci = int(name)
        conv_idx = self._owner.CONV_IDX
        if ci in conv_idx:
            return getattr(getattr(self._owner, "scale%d_layer%d" % (self._d, conv_idx.index(ci))), "0")
        j = {v: conv_idx.index(k) for k, v in self._owner.BN_IDX.items()}[ci]
        return getattr(getattr(self._owner, "scale%d_layer%d" % (self._d, j)), "1")


class MultiscaleDiscriminatorIF(MultiscaleDiscriminator):
    """The same discriminator with `getIntermFeat=True` (pix2pixHD's `--getIntermFeat_D` default; reference
    networks.py:1661-1667): every layer is its own module `scale<d>_layer<j>` (state_dict keys
    `scale<d>_layer<j>.{0 conv | 1 norm}.*`) and forward returns the per-layer features of every scale."""

    def __init__(self, input_nc, ndf=64, n_layers=3, num_D=3, opt=None):
        nn.Module.__init__(self)
        if n_layers < 1:
            raise ValueError("MultiscaleDiscriminator: n_layers must be >= 1")
        self.input_nc, self.ndf, self.num_D, self.n_layers = input_nc, ndf, num_D, n_layers
        # no Sigmoid in this form, whatever gan_mode: the reference registers only the first n_layers + 2 blocks of the PatchGAN's
        # sequence as scale<d>_layer<j> (networks.py:1664-1666), which leaves the trailing [Sigmoid] block of 'vanilla' mode out
        self.use_sigmoid = False
        self.CONV_IDX, self.BN_IDX, self.STRIDE = _patchgan_layout(n_layers)
        chans = _patchgan_channels(input_nc, ndf, n_layers)
        self.chans = chans
        for d in range(num_D):
            for j in range(n_layers + 2):
                kids = {0: _ConvParams((chans[j + 1], chans[j], 4, 4), chans[j + 1])}
                if 1 <= j <= n_layers:
                    kids[1] = _BNParams(chans[j + 1])
                setattr(self, "scale%d_layer%d" % (d, j), _Holder(kids))

    def __getattr__(self, name):
        if name.startswith("layer") and name[5:].isdigit():
            return _LayerView(self, int(name[5:]))
        return nn.Module.__getattr__(self, name)

    def forward(self, x):
        """[[feat_0, ..., pred] per scale]: post-activation layer outputs like the reference's singleD_forward."""
        from vts import ops as _ops
        _, ctx = engine.msd_forward(self, x, None, keep=True)
        res = []
        for (_, _, acts) in ctx.scales:
            feats = [_ops.pad_affine(a, (0, 0, 0, 0), 0, act=engine.LRELU) for a in acts[:-1]]
            res.append(feats + [torch.sigmoid(acts[-1].data) if self.use_sigmoid else acts[-1].data])
        return res


def init_weights(net, init_type="normal", init_gain=0.02):
    """networks.py:191-231: conv weights by `init_type`, biases 0, BatchNorm weight ~ N(1, gain)."""
    for m in net.modules():
        if isinstance(m, (_ConvParams, _LinearParams)):
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
        elif isinstance(m, _BNParams) and not isinstance(m, _BN1dParams):
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
    resnet_blocks = {"resnet_9blocks": 9, "resnet_6blocks": 6, "resnet_4blocks": 4}
    if netG in ("stylegan2", "smallstylegan2"):
        # networks.py:297-300: StyleGAN2Generator(n_blocks 6 | 2); noise injection unless 'small' is in the name (stylegan_networks.py:886);
        # no init_weights for StyleGAN2 nets (:325).  The module emits THREE channels whatever output_nc is (:892).
        import math

        from .stylegan2_blocks import StyleGAN2Generator
        size = 2 ** int(round(math.log2(min(opt.load_size, opt.crop_size))))            # stylegan_networks.py:820
        net = StyleGAN2Generator(input_nc, output_nc, ngf, n_blocks=2 if netG == "smallstylegan2" else 6, size=size,
                                 num_downsampling=getattr(opt, "stylegan2_G_num_downsampling", 1), inject_noise="small" not in netG)
        return init_net(net, init_type, init_gain, gpu_ids, initialize_weights=False)
    if netG not in resnet_blocks and netG not in ("unet256_custom", "global", "local"):
        raise NotImplementedError("Generator model name [%s] is not recognized (built: unet256_custom, resnet_{4,6,9}blocks, global, local, "
                                  "stylegan2, smallstylegan2)" % netG)
    if netG == "local":    # pix2pixHD LocalEnhancer (networks.py:311-313)
        if norm not in ("batch", "instance"):
            raise NotImplementedError("local enhancer: norm %s is not built" % norm)
        net = LocalEnhancer(input_nc, output_nc, ngf, getattr(opt, "n_downsample_global", 4), getattr(opt, "n_blocks_global", 9),
                            getattr(opt, "n_local_enhancers", 1), getattr(opt, "n_blocks_local", 3), norm, opt=opt)
        return init_net(net, init_type, init_gain, gpu_ids)
    if netG == "global":   # pix2pixHD coarse generator (networks.py:309-310)
        if norm not in ("batch", "instance"):
            raise NotImplementedError("global generator: norm %s is not built" % norm)
        net = GlobalGenerator(input_nc, output_nc, ngf, getattr(opt, "n_downsample_global", 4), getattr(opt, "n_blocks_global", 9), norm, opt=opt)
        return init_net(net, init_type, init_gain, gpu_ids)
    if netG in resnet_blocks:
        if norm not in ("batch", "instance"):
            raise NotImplementedError("resnet generator: norm %s is not built" % norm)
        if generate_T_imgs:
            raise NotImplementedError("resnet generator: generate_T_imgs is not built")
        net = ResnetGenerator(input_nc, output_nc, ngf=ngf, n_blocks=resnet_blocks[netG], norm=norm,
                              down="stride" if no_antialias else "blur", up="convT" if no_antialias_up else "blur", opt=opt,
                              use_dropout=use_dropout)
    elif norm != "instance":
        raise NotImplementedError("unet256_custom is built for normG=instance only")
    else:
        net = CustomUnetGenerator(input_nc, output_nc, num_downs=8, ngf=ngf, num_layer_separate=num_layer_separate, opt=opt,
                                  use_dropout=use_dropout)
    return init_net(net, init_type, init_gain, gpu_ids)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm="batch", init_type="normal", init_gain=0.02, no_antialias=False,
             num_D=1, gpu_ids=(), opt=None):
    if netD == "stylegan2":   # networks.py:437-442: no init_weights for StyleGAN2 nets (randn weights, runtime 1/sqrt(fan_in) scale)
        import math

        from .stylegan2_blocks import StyleGAN2Discriminator
        size = 2 ** int(round(math.log2(min(opt.load_size, opt.crop_size))))            # stylegan_networks.py:701-702
        return init_net(StyleGAN2Discriminator(input_nc, ndf, size), init_type, init_gain, gpu_ids, initialize_weights=False)
    if netD != "multiscale":
        raise NotImplementedError("Discriminator model name [%s] is not recognized (built: multiscale, stylegan2)" % netD)
    if norm != "batch":
        raise NotImplementedError("multiscale discriminator is built for normD=batch only")
    cls = MultiscaleDiscriminatorIF if getattr(opt, "getIntermFeat_D", False) else MultiscaleDiscriminator   # networks.py:1661
    net = cls(input_nc, ndf, n_layers_D, num_D=num_D, opt=opt)
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

        if gan_mode not in L.GAN_MODES or gan_mode == "vanilla_sigmoid":
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)
        self.gan_mode, self.real_label, self.fake_label = gan_mode, target_real_label, target_fake_label

    def to(self, device):
        return self

    def accumulate(self, preds, target_is_real, coeff, slot, grad_coeff=None, want_grad=True, out_grads=None, pre_sigmoid=False):
        """slot += coeff * sum_scales mean_batch(loss); returns [dpred per scale] scaled by grad_coeff (written into out_grads
        -- e.g. batch slices of one gradient buffer -- when given).  `preds` are the discriminator's RAW maps; pre_sigmoid: the
        discriminator ends in a Sigmoid (MultiscaleDiscriminator.use_sigmoid, gan_mode 'vanilla'), applied inside the loss kernel."""
        from vts import ops

        grads = []
        label = self.real_label if target_is_real else self.fake_label
        mode = self.gan_mode
        if pre_sigmoid:
            if mode != "vanilla":
                raise NotImplementedError("a discriminator ending in a Sigmoid is only known with gan_mode 'vanilla'")
            mode = "vanilla_sigmoid"
        for i, p in enumerate(preds):
            p = p[-1] if isinstance(p, (list, tuple)) else p
            g = (out_grads[i] if out_grads is not None else torch.empty_like(p)) if want_grad else None
            ops.ganloss(p, mode, target_is_real, coeff, slot, g, label=label, grad_coeff=grad_coeff)
            grads.append(g)
        return grads

    def __call__(self, preds, target_is_real):
        first = preds[0][-1] if isinstance(preds[0], (list, tuple)) else preds[-1]
        from vts import ops
        slot = ops.loss_slots(1, first.device)
        plist = preds if isinstance(preds[0], (list, tuple)) else [preds[-1]]
        self.accumulate(plist, target_is_real, 1.0, slot, want_grad=False)
        return (slot.double() / ops.LOSS_SCALE).float()


class PatchSampleF(nn.Module):
    """PatchSampleF (reference networks.py:667-719) on the HIP path, forward: per feature map pick `num_patches` spatial positions
    (given `patch_ids`, else a numpy permutation like the reference -- the same ids for every image of the batch), gather them as rows
    [B * P, C] (vts_patch_sample), optionally run the 2-layer MLP Linear(C, nc) - ReLU - Linear(nc, nc) (vts_linear_rows; created on
    first use and initialised like the reference: init_net normal / 0.02, bias 0), L2-normalise the rows (vts_l2norm_rows).
    state_dict keys match the reference's (mlp_<i>.0.weight, mlp_<i>.0.bias, mlp_<i>.2.weight, mlp_<i>.2.bias).
    num_patches == 0: the whole map, normalised over the positions of each (image, channel) and reshaped back to NCHW like the reference."""

    def __init__(self, use_mlp=False, init_type="normal", init_gain=0.02, nc=256, gpu_ids=()):
        super().__init__()
        self.use_mlp, self.nc, self.mlp_init, self.init_type, self.init_gain, self.gpu_ids = use_mlp, nc, False, init_type, init_gain, gpu_ids

    def create_mlp(self, feats):
        for mlp_id, feat in enumerate(feats):
            mlp = nn.Sequential(nn.Linear(feat.shape[1], self.nc), nn.ReLU(), nn.Linear(self.nc, self.nc)).to(feat.device)
            for m in mlp:
                if isinstance(m, nn.Linear):
                    if self.init_type == "normal":
                        nn.init.normal_(m.weight.data, 0.0, self.init_gain)
                    elif self.init_type == "xavier":
                        nn.init.xavier_normal_(m.weight.data, gain=self.init_gain)
                    else:
                        raise NotImplementedError("PatchSampleF: init_type %s is not built" % self.init_type)
                    nn.init.constant_(m.bias.data, 0.0)
            setattr(self, "mlp_%d" % mlp_id, mlp)
        self.mlp_init = True

    @torch.no_grad()
    def forward(self, feats, num_patches=64, patch_ids=None):
        import numpy as np

        from vts import ops

        if num_patches < 0:
            raise ValueError("PatchSampleF: num_patches must be >= 0")
        if self.use_mlp and not self.mlp_init:
            self.create_mlp(feats)
        return_ids, return_feats = [], []
        for feat_id, feat in enumerate(feats):
            b, _, fh, fw = feat.shape
            hw = fh * fw
            if num_patches == 0:        # the whole map (networks.py:704-706): every position, no ids returned
                patch_id = torch.arange(hw, dtype=torch.long, device=feat.device)
            elif patch_ids is not None:
                patch_id = patch_ids[feat_id]
            else:
                patch_id = np.random.permutation(hw)
                patch_id = patch_id[:int(min(num_patches, patch_id.shape[0]))]
            patch_id = torch.as_tensor(patch_id, dtype=torch.long, device=feat.device)
            x = ops.patch_sample(feat.contiguous(), patch_id)
            if self.use_mlp:
                mlp = getattr(self, "mlp_%d" % feat_id)
                x = ops.linear_rows(x, mlp[0].weight, mlp[0].bias, relu=True)
                x = ops.linear_rows(x, mlp[2].weight, mlp[2].bias)
            if num_patches == 0:
                # the reference keeps the rows 3-D here ([B, HW, C]), so its Normalize(2) sums over dim 1 = the POSITIONS of each
                # (image, channel), and then reshapes back to [B, C, H, W] (networks.py:713-717): rows = (image, channel) over HW
                c = x.shape[1]
                rows = x.view(b, hw, c).permute(0, 2, 1).contiguous().view(b * c, hw)
                return_ids.append([])
                return_feats.append(ops.l2norm_rows(rows).view(b, c, fh, fw))
                continue
            return_ids.append(patch_id)
            return_feats.append(ops.l2norm_rows(x))
        return return_feats, return_ids


class PatchNCELoss:
    """PatchNCELoss (reference models/patchnce.py:6-55): per-patch loss [B * P] and, with want_grad, d sum(loss) / d feat_q"""

    def __init__(self, opt):
        self.opt = opt

    def __call__(self, feat_q, feat_k, want_grad=False):
        from vts import ops

        groups = 1 if self.opt.nce_includes_all_negatives_from_minibatch else self.opt.batch_size
        loss, dq = ops.patchnce(feat_q.contiguous(), feat_k.contiguous(), groups, self.opt.nce_T, want_grad=want_grad)
        return (loss, dq) if want_grad else loss
