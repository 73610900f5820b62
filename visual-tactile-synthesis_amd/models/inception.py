"""First block of torchvision's Inception-v3 as the reference's SIFID uses it (models/inception.py:57-67: `block0 =
[Conv2d_1a_3x3, Conv2d_2a_3x3, Conv2d_2b_3x3]`, NO max-pool in this fork; models/sifid.py:205-233 feeds it one image at a time and
takes the 64-channel map as `positions x 64` activations).  A BasicConv2d of torchvision is Conv2d(bias=False) -> BatchNorm2d(eps
0.001) -> ReLU; the three layers are 3 -> 32 (3x3, stride 2), 32 -> 32 (3x3) and 32 -> 64 (3x3, padding 1).

The module only HOLDS parameters (state-dict keys of the reference wrapper: `blocks.0.{0,1,2}.{conv.weight, bn.*}`; torchvision's own
names `Conv2d_1a_3x3.*` ... are accepted by `load_weights`): the forward runs on the HIP kernels (vts.engine.inception_block0).
The pretrained weights (torchvision `inception_v3(pretrained=True)`, models/inception.py:58) cannot exist offline; without a weight
file the block is initialised from a fixed seed and `pretrained` stays False -- SIFID values are then comparable between builds on
the same seed only, and every report says so."""
import os

import torch
import torch.nn as nn

_TV_NAMES = ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3")
SPEC = ((3, 32, 2, 0), (32, 32, 1, 0), (32, 64, 1, 1))   # (cin, cout, stride, padding) of the three 3x3 BasicConv2d layers


class BasicConv2d(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)


class InceptionBlock0(nn.Module):
    def __init__(self, seed=20150512):
        super().__init__()
        self.blocks = nn.ModuleList([nn.Sequential(*[BasicConv2d(ci, co) for ci, co, _, _ in SPEC])])
        self.pretrained = False
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():       # stand-in statistics of a trained network: unit-gain weights, non-trivial BatchNorm buffers
            for m in self.blocks[0]:
                fan_in = m.conv.weight[0].numel()
                m.conv.weight.copy_(torch.randn(m.conv.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                m.bn.weight.copy_(1.0 + 0.1 * torch.randn(m.bn.weight.shape, generator=g))
                m.bn.bias.copy_(0.1 * torch.randn(m.bn.bias.shape, generator=g))
                m.bn.running_mean.copy_(0.1 * torch.randn(m.bn.running_mean.shape, generator=g))
                m.bn.running_var.copy_(1.0 + 0.2 * torch.rand(m.bn.running_var.shape, generator=g))
        for p in self.parameters():
            p.requires_grad = False

    def load_weights(self, path):
        """state dict of the reference's InceptionV3 wrapper, of torchvision's inception_v3, or of pytorch-fid's FID network"""
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        own = self.state_dict()
        got = {}
        for k, v in sd.items():
            for i, tv in enumerate(_TV_NAMES):
                if k.startswith(tv + "."):
                    k = "blocks.0.%d.%s" % (i, k[len(tv) + 1:])
            if k in own and own[k].shape == v.shape:
                got[k] = v
        missing = [k for k in own if k not in got and not k.endswith("num_batches_tracked")]
        if missing:
            raise KeyError("inception weights %s lack %s" % (path, missing[:4]))
        self.load_state_dict(got, strict=False)
        self.pretrained = True
        return self


def build(opt=None, device=None):
    """the block for SIFID: weights from --inception_weights / $VTS_INCEPTION_WEIGHTS when given, else the seeded stand-in"""
    net = InceptionBlock0()
    path = getattr(opt, "inception_weights", None) or os.environ.get("VTS_INCEPTION_WEIGHTS")
    if path:
        net.load_weights(path)
    return net.to(device) if device is not None else net
