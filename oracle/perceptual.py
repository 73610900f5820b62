"""CPU restatement of the perceptual networks the reference calls (TEST INFRASTRUCTURE ONLY: only tests/, smoke() and
bench.py's cpu_baseline leg may import anything under oracle/).

Two third-party pieces, both absent from /root/reference and from this image:

* `lpips.LPIPS(net="vgg")` (and net="alex", the reference's test-phase `eval_LPIPS`, models/sinskitG_model.py:501: the same head on
  torchvision alexnet.features cut after its five ReLUs) -- pip package `lpips` (requirements.txt:12, unpinned; current release 0.1.4, model version "0.1").
  Call sites: models/sinskitG_model.py:495 (construction), :1711 (I term), :1639-1646 (gx / gy terms), models/model_utils.py:477,
  523-527 (I_LPIPS / T_LPIPS metrics).  Its published algorithm (lpips/lpips.py, lpips/pretrained_networks.py of that release):
      x -> (x - shift) / scale,  shift = (-.030, -.088, -.188), scale = (.458, .448, .450)  (a 1-channel input broadcasts to 3)
      torchvision vgg16.features cut after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (indices 4, 9, 16, 23, 30)
      per tap:  f / (sqrt(sum_c f^2) + 1e-10)  for both images,  (f0n - f1n)^2,  a 1x1 convolution C -> 1 without bias (the learned
      "lin" layer; its Dropout is inactive: the module is built in eval mode),  mean over H x W;  the five values are summed: [N,1,1,1].
* torchvision `vgg19(pretrained=True).features` inside the reference's own `Vgg19` / `VGGLoss` (models/networks.py:2021-2067): cuts
  after relu1_1, relu2_1, relu3_1, relu4_1, relu5_1 (indices 2, 7, 12, 21, 30), loss = sum_i w_i * L1(f_i(x), f_i(y).detach()),
  w = (1/32, 1/16, 1/8, 1/4, 1).

PARITY: the ARITHMETIC is pinned -- the reference's own call sites run on these modules with seeded stand-in weights
(oracle/make_golden.py `lpips` / tests/golden/lpips_step_256.npz), and the HIP path is compared with the same modules.  The
pretrained WEIGHTS (torchvision VGG16 / VGG19, LPIPS v0.1 lin layers) cannot exist offline: values with the stand-in weights compare
builds on the same seed only ("parity unpinned" for the published numbers; `--lpips_weights` / `--vgg_weights` load real ones).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)          # torchvision cfg "D" up to relu5_3
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512)         # cfg "E" up to relu5_1
LPIPS_TAPS = (1, 3, 6, 9, 12)      # index (0-based, counting convolutions) of the conv whose ReLU output is tapped: relu1_2 ... relu5_3
VGG19_TAPS = (0, 2, 4, 8, 12)      # relu1_1, relu2_1, relu3_1, relu4_1, relu5_1
LPIPS_SHIFT = (-.030, -.088, -.188)
LPIPS_SCALE = (.458, .448, .450)
VGG_LOSS_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)


def conv_shapes(cfg):
    """[(cout, cin)] of the 3x3 convolutions of a torchvision VGG feature stack"""
    out, cin = [], 3
    for v in cfg:
        if v != "M":
            out.append((v, cin))
            cin = v
    return out


def feature_index(cfg, k):
    """index inside torchvision's `features` Sequential of the k-th convolution (state-dict key `features.<idx>.weight`)"""
    idx, seen = 0, 0
    for v in cfg:
        if v == "M":
            idx += 1
            continue
        if seen == k:
            return idx
        seen += 1
        idx += 2
    raise IndexError(k)


def standin_state(cfg, taps, seed, lin=True):
    """seeded stand-in weights with the statistics of a trained VGG: He-scaled convolutions (activations keep their scale through the
    stack), small biases, non-negative lin weights.  Keys: conv<k>.weight / conv<k>.bias, lin<i>.weight [1, C, 1, 1]."""
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


class VggFeatures(nn.Module):
    """ReLU outputs of the tapped convolutions of a VGG feature stack (3x3, padding 1, MaxPool2d(2, 2) where cfg says 'M')"""

    def __init__(self, cfg, taps, sd):
        super().__init__()
        self.cfg, self.taps = cfg, tuple(taps)
        self.convs = nn.ModuleList([nn.Conv2d(ci, co, 3, padding=1) for co, ci in conv_shapes(cfg)])
        with torch.no_grad():
            for k, m in enumerate(self.convs):
                m.weight.copy_(sd["conv%d.weight" % k])
                m.bias.copy_(sd["conv%d.bias" % k])
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        feats, k = [], 0
        for v in self.cfg:
            if v == "M":
                x = F.max_pool2d(x, 2, 2)
                continue
            x = F.relu(self.convs[k](x))
            if k in self.taps:
                feats.append(x)
            k += 1
        return feats


# torchvision alexnet.features as lpips/pretrained_networks.py:alexnet slices it (relu1 .. relu5 are the five taps):
#   Conv2d(3, 64, 11, stride 4, padding 2) ReLU | MaxPool2d(3, 2) Conv2d(64, 192, 5, padding 2) ReLU | MaxPool2d(3, 2) Conv2d(192, 384, 3,
#   padding 1) ReLU | Conv2d(384, 256, 3, padding 1) ReLU | Conv2d(256, 256, 3, padding 1) ReLU
ALEX_CONVS = ((64, 3, 11, 4, 2), (192, 64, 5, 1, 2), (384, 192, 3, 1, 1), (256, 384, 3, 1, 1), (256, 256, 3, 1, 1))   # (cout, cin, k, stride, pad)
ALEX_POOL_BEFORE = (False, True, True, False, False)        # MaxPool2d(kernel 3, stride 2) in front of the convolution
ALEX_FEATURE_INDEX = (0, 3, 6, 8, 10)                       # state-dict key `features.<idx>.weight` of torchvision's alexnet


def standin_state_alex(seed):
    """seeded stand-in weights of the AlexNet variant (He-scaled convolutions, small biases, non-negative lin weights)"""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, (co, ci, ks, _, _) in enumerate(ALEX_CONVS):
        sd["conv%d.weight" % k] = torch.randn(co, ci, ks, ks, generator=g) * (2.0 / (ci * ks * ks)) ** 0.5
        sd["conv%d.bias" % k] = 0.05 * torch.randn(co, generator=g)
    for i, (co, _, _, _, _) in enumerate(ALEX_CONVS):
        sd["lin%d.weight" % i] = torch.rand(1, co, 1, 1, generator=g) * (2.0 / co)
    return sd


class AlexFeatures(nn.Module):
    """ReLU outputs of the five convolutions of torchvision's AlexNet feature stack"""

    def __init__(self, sd):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(ci, co, ks, stride=st, padding=pd) for co, ci, ks, st, pd in ALEX_CONVS])
        with torch.no_grad():
            for k, m in enumerate(self.convs):
                m.weight.copy_(sd["conv%d.weight" % k])
                m.bias.copy_(sd["conv%d.bias" % k])
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        feats = []
        for k, m in enumerate(self.convs):
            if ALEX_POOL_BEFORE[k]:
                x = F.max_pool2d(x, 3, 2)
            x = F.relu(m(x))
            feats.append(x)
        return feats


class LPIPS(nn.Module):
    """lpips.LPIPS(net='vgg' | 'alex', version='0.1', lpips=True, spatial=False) in eval mode.  The reference builds the VGG variant for
    training / validation (models/sinskitG_model.py:495-499) and the AlexNet variant as `eval_LPIPS` of the test phase (:501)."""

    def __init__(self, net="vgg", seed=None, sd=None, **kw):
        super().__init__()
        if net == "alex":
            sd = sd if sd is not None else standin_state_alex(20180112 if seed is None else seed)
            self.net = AlexFeatures(sd)
            ntaps = len(ALEX_CONVS)
        elif net == "vgg":
            sd = sd if sd is not None else standin_state(VGG16_CFG, LPIPS_TAPS, 20180111 if seed is None else seed)
            self.net = VggFeatures(VGG16_CFG, LPIPS_TAPS, sd)
            ntaps = len(LPIPS_TAPS)
        else:
            raise NotImplementedError("lpips.LPIPS(net=%r): the reference uses 'vgg' and 'alex' only" % net)
        self.lins = nn.ParameterList([nn.Parameter(sd["lin%d.weight" % i].clone(), requires_grad=False) for i in range(ntaps)])
        self.register_buffer("shift", torch.tensor(LPIPS_SHIFT)[None, :, None, None])
        self.register_buffer("scale", torch.tensor(LPIPS_SCALE)[None, :, None, None])
        self.eval()

    @staticmethod
    def normalize_tensor(f, eps=1e-10):
        return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + eps)

    def forward(self, in0, in1, retPerLayer=False, normalize=False):
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        f0 = self.net((in0 - self.shift) / self.scale)
        f1 = self.net((in1 - self.shift) / self.scale)
        val = 0
        for a, b, w in zip(f0, f1, self.lins):
            d = (self.normalize_tensor(a) - self.normalize_tensor(b)) ** 2
            val = val + F.conv2d(d, w).mean([2, 3], keepdim=True)
        return val


class Vgg19(nn.Module):
    """the reference's Vgg19 (models/networks.py:2036-2067) on given weights"""

    def __init__(self, seed=20140904, sd=None):
        super().__init__()
        sd = sd if sd is not None else standin_state(VGG19_CFG, VGG19_TAPS, seed, lin=False)
        self.net = VggFeatures(VGG19_CFG, VGG19_TAPS, sd)

    def forward(self, x):
        return self.net(x)


class VGGLoss(nn.Module):
    """models/networks.py:2021-2033"""

    def __init__(self, vgg=None):
        super().__init__()
        self.vgg = vgg if vgg is not None else Vgg19()

    def forward(self, x, y):
        fx, fy = self.vgg(x), self.vgg(y)
        loss = 0
        for w, a, b in zip(VGG_LOSS_WEIGHTS, fx, fy):
            loss = loss + w * F.l1_loss(a, b.detach())
        return loss


def touch_lpips(lp, fake_T, real_T, nt, lam):
    """_compute_touch_lpips_loss (models/sinskitG_model.py:1619-1658): gx and gy as 1-channel images, per-sample sum over the NT patches"""
    gx = lp(fake_T[:, 0:1], real_T[:, 0:1]).view(-1, nt, 1, 1, 1).sum(1).mean()
    gy = lp(fake_T[:, 1:2], real_T[:, 1:2]).view(-1, nt, 1, 1, 1).sum(1).mean()
    return lam * (gx + gy)
