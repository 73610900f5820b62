"""Frozen VGG feature stacks of the perceptual terms: parameter holders, weight loaders, seeded stand-ins.

  LpipsVgg16   lpips.LPIPS(net="vgg") as the reference builds it (models/sinskitG_model.py:495; pip package `lpips`, requirements.txt:12):
               ScalingLayer, torchvision vgg16.features cut after relu1_2 / 2_2 / 3_3 / 4_3 / 5_3, five 1x1 "lin" layers.
  Vgg19Features the reference's own Vgg19 (models/networks.py:2036-2067): torchvision vgg19.features cut after relu1_1 / 2_1 / 3_1 / 4_1 / 5_1,
               used by VGGLoss (:2021-2033) of the pix2pixHD baseline.

The modules only HOLD parameters; the forward / input-gradient schedules run on the HIP kernels (vts/perceptual.py: GEMM-class 3x3
convolutions + vts_perceptual.hip).  The pretrained weights (torchvision downloads, lpips/weights/v0.1/vgg.pth) cannot exist offline:
`--lpips_weights` / `--vgg_weights` (or $VTS_LPIPS_WEIGHTS / $VTS_VGG_WEIGHTS) load them from state-dict files; without a file the
stack is initialised from a fixed seed and `pretrained` stays False -- losses / metrics then compare builds on the same seed only,
and every report says so (the same treatment as models/inception.py)."""
import os

import torch
import torch.nn as nn

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512)
LPIPS_TAPS = (1, 3, 6, 9, 12)      # convolution indices whose ReLU output is a feature tap
VGG19_TAPS = (0, 2, 4, 8, 12)
LPIPS_SHIFT = (-.030, -.088, -.188)
LPIPS_SCALE = (.458, .448, .450)
VGG_LOSS_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)


def conv_shapes(cfg):
    out, cin = [], 3
    for v in cfg:
        if v != "M":
            out.append((v, cin))
            cin = v
    return out


def feature_indices(cfg):
    """torchvision `features` index of every convolution (conv, relu pairs; 'M' is one module)"""
    idx, out = 0, []
    for v in cfg:
        if v == "M":
            idx += 1
        else:
            out.append(idx)
            idx += 2
    return out


def standin_state(cfg, taps, seed, lin=True):
    """seeded stand-in weights (He-scaled convolutions, small biases, non-negative lin weights); the same draw order as the checker's
    oracle/perceptual.py:standin_state, so both sides hold identical numbers"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    shapes = conv_shapes(cfg)
    for k, (co, ci) in enumerate(shapes):
        sd["conv%d.weight" % k] = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5
        sd["conv%d.bias" % k] = 0.05 * torch.randn(co, generator=g)
    if lin:
        for i, k in enumerate(taps):
            sd["lin%d.weight" % i] = torch.rand(1, shapes[k][0], 1, 1, generator=g) * (2.0 / shapes[k][0])
    return sd


class _VggStack(nn.Module):
    CFG, TAPS, SEED, HAS_LIN = None, None, 0, False

    def __init__(self):
        super().__init__()
        self.cfg, self.taps = self.CFG, self.TAPS
        shapes = conv_shapes(self.CFG)
        self.convs = nn.ModuleList([nn.Conv2d(ci, co, 3, padding=1) for co, ci in shapes])
        if self.HAS_LIN:
            self.lins = nn.ParameterList([nn.Parameter(torch.zeros(1, shapes[k][0], 1, 1)) for k in self.TAPS])
        self.pretrained = False
        self._load_own(standin_state(self.CFG, self.TAPS, self.SEED, lin=self.HAS_LIN))
        for p in self.parameters():
            p.requires_grad = False
        self._packed = {}     # packed weights of the GEMM-class kernels: built once (the stack is frozen), see vts/perceptual.py

    def _load_own(self, sd):
        with torch.no_grad():
            for k, m in enumerate(self.convs):
                m.weight.copy_(sd["conv%d.weight" % k])
                m.bias.copy_(sd["conv%d.bias" % k])
            if self.HAS_LIN:
                for i, p in enumerate(self.lins):
                    p.copy_(sd["lin%d.weight" % i].reshape(p.shape))
        self._packed = {}

    def own_state(self):
        sd = {}
        for k, m in enumerate(self.convs):
            sd["conv%d.weight" % k], sd["conv%d.bias" % k] = m.weight.detach().cpu(), m.bias.detach().cpu()
        if self.HAS_LIN:
            for i, p in enumerate(self.lins):
                sd["lin%d.weight" % i] = p.detach().cpu()
        return sd

    def load_weights(self, paths):
        """state-dict file(s), comma separated: torchvision vgg (`features.<i>.*`), the lpips / reference wrappers (`net.slice<j>.<i>.*`,
        `slice<j>.<i>.*`), this module's own names (`conv<k>.*`), and for LPIPS the lin layers (`lin<i>.model.1.weight`,
        `lins.<i>.model.1.weight`, `lin<i>.weight`)"""
        merged = {}
        for path in str(paths).split(","):
            sd = torch.load(path.strip(), map_location="cpu", weights_only=True)      # (state dicts only: no pickled code from a user path)
            merged.update(sd.get("state_dict", sd))
        fidx = feature_indices(self.CFG)
        own = {}
        for key, v in merged.items():
            parts = key.split(".")
            name = None
            if parts[0].startswith("conv") and parts[0][4:].isdigit():
                name = key
            elif len(parts) >= 3 and parts[-1] in ("weight", "bias") and parts[-2].isdigit() and ("features" in parts or any(p.startswith("slice") for p in parts)):
                i = int(parts[-2])
                if i in fidx:
                    name = "conv%d.%s" % (fidx.index(i), parts[-1])
            elif self.HAS_LIN and parts[-1] == "weight" and (parts[0].startswith("lin") or parts[0] == "lins"):
                digits = parts[1] if parts[0] == "lins" else parts[0][3:]
                if digits.isdigit():
                    name = "lin%d.weight" % int(digits)
            if name is not None:
                own[name] = v.float()
        need = self.own_state().keys()
        missing = [k for k in need if k not in own]
        if missing:
            raise KeyError("%s weights %s lack %s" % (type(self).__name__, paths, missing[:4]))
        self._load_own(own)
        self.pretrained = True
        return self


class LpipsVgg16(_VggStack):
    CFG, TAPS, SEED, HAS_LIN = VGG16_CFG, LPIPS_TAPS, 20180111, True
    shift, scale = LPIPS_SHIFT, LPIPS_SCALE


class Vgg19Features(_VggStack):
    CFG, TAPS, SEED, HAS_LIN = VGG19_CFG, VGG19_TAPS, 20140904, False
    weights = VGG_LOSS_WEIGHTS


# ---- AlexNet variant (the reference's test-phase eval_LPIPS = lpips.LPIPS(net="alex"), models/sinskitG_model.py:501) ----
ALEX_CONVS = ((64, 3, 11, 4, 2), (192, 64, 5, 1, 2), (384, 192, 3, 1, 1), (256, 384, 3, 1, 1), (256, 256, 3, 1, 1))   # (cout, cin, k, stride, pad)
ALEX_POOL_BEFORE = (False, True, True, False, False)        # MaxPool2d(3, 2) in front of the convolution
ALEX_FEATURE_INDEX = (0, 3, 6, 8, 10)                       # torchvision alexnet.features indices of the convolutions


def standin_state_alex(seed):
    """seeded stand-in weights (the same draw order as the checker's oracle/perceptual.py:standin_state_alex)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, (co, ci, ks, _, _) in enumerate(ALEX_CONVS):
        sd["conv%d.weight" % k] = torch.randn(co, ci, ks, ks, generator=g) * (2.0 / (ci * ks * ks)) ** 0.5
        sd["conv%d.bias" % k] = 0.05 * torch.randn(co, generator=g)
    for i, (co, _, _, _, _) in enumerate(ALEX_CONVS):
        sd["lin%d.weight" % i] = torch.rand(1, co, 1, 1, generator=g) * (2.0 / co)
    return sd


class LpipsAlex(nn.Module):
    """lpips.LPIPS(net="alex"): ScalingLayer, torchvision alexnet.features cut after its five ReLUs, five 1x1 lin layers.  Parameter
    holder; the forward runs in vts/perceptual.py:alex_forward (forward only: the reference uses it as an evaluation metric)."""
    SEED = 20180112
    shift, scale = LPIPS_SHIFT, LPIPS_SCALE
    taps = (0, 1, 2, 3, 4)

    def __init__(self):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(ci, co, ks, stride=st, padding=pd) for co, ci, ks, st, pd in ALEX_CONVS])
        self.lins = nn.ParameterList([nn.Parameter(torch.zeros(1, co, 1, 1)) for co, _, _, _, _ in ALEX_CONVS])
        self.pretrained = False
        self._load_own(standin_state_alex(self.SEED))
        for p in self.parameters():
            p.requires_grad = False
        self._packed = {}

    def _load_own(self, sd):
        with torch.no_grad():
            for k, m in enumerate(self.convs):
                m.weight.copy_(sd["conv%d.weight" % k])
                m.bias.copy_(sd["conv%d.bias" % k])
            for i, p in enumerate(self.lins):
                p.copy_(sd["lin%d.weight" % i].reshape(p.shape))
        self._packed = {}

    def stem_weight(self):
        """the 11 x 11 stride-4 stem as a 3 x 3 convolution over the 48 space-to-depth channels: w'[o][(c, i, j)][a][b] = w[o][c][4a + i][4b + j]"""
        w = self.convs[0].weight
        co, ci = w.shape[:2]
        w12 = torch.zeros(co, ci, 12, 12, dtype=w.dtype, device=w.device)
        w12[:, :, :11, :11] = w
        return w12.view(co, ci, 3, 4, 3, 4).permute(0, 1, 3, 5, 2, 4).reshape(co, ci * 16, 3, 3).contiguous()

    def load_weights(self, paths):
        """state-dict file(s), comma separated: torchvision alexnet (`features.<i>.*`), the lpips wrapper (`net.slice<j>.<i>.*`), own names
        (`conv<k>.*`) and the lin layers (`lin<i>.model.1.weight`, `lins.<i>.model.1.weight`, `lin<i>.weight`)"""
        merged = {}
        for path in str(paths).split(","):
            sd = torch.load(path.strip(), map_location="cpu", weights_only=True)
            merged.update(sd.get("state_dict", sd))
        own = {}
        for key, v in merged.items():
            parts = key.split(".")
            name = None
            if parts[0].startswith("conv") and parts[0][4:].isdigit():
                name = key
            elif len(parts) >= 3 and parts[-1] in ("weight", "bias") and parts[-2].isdigit() and ("features" in parts or any(p.startswith("slice") for p in parts)):
                i = int(parts[-2])
                if i in ALEX_FEATURE_INDEX:
                    name = "conv%d.%s" % (ALEX_FEATURE_INDEX.index(i), parts[-1])
            elif parts[-1] == "weight" and (parts[0].startswith("lin") or parts[0] == "lins"):
                digits = parts[1] if parts[0] == "lins" else parts[0][3:]
                if digits.isdigit():
                    name = "lin%d.weight" % int(digits)
            if name is not None:
                own[name] = v.float()
        need = ["conv%d.%s" % (k, t) for k in range(5) for t in ("weight", "bias")] + ["lin%d.weight" % i for i in range(5)]
        missing = [k for k in need if k not in own]
        if missing:
            raise KeyError("LpipsAlex weights %s lack %s" % (paths, missing[:4]))
        self._load_own(own)
        self.pretrained = True
        return self


def build_lpips_alex(opt=None, device=None):
    net = LpipsAlex()
    path = getattr(opt, "lpips_alex_weights", None) or os.environ.get("VTS_LPIPS_ALEX_WEIGHTS")
    if path:
        net.load_weights(path)
    return net.to(device) if device is not None else net


def build_lpips(opt=None, device=None):
    net = LpipsVgg16()
    path = getattr(opt, "lpips_weights", None) or os.environ.get("VTS_LPIPS_WEIGHTS")
    if path:
        net.load_weights(path)
    return net.to(device) if device is not None else net


def build_vgg19(opt=None, device=None):
    net = Vgg19Features()
    path = getattr(opt, "vgg_weights", None) or os.environ.get("VTS_VGG_WEIGHTS")
    if path:
        net.load_weights(path)
    return net.to(device) if device is not None else net
