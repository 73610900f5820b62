"""StyleGAN2 discriminator of the reference (`--netD stylegan2`, models/stylegan_networks.py:696-786) as a parameter container with
the reference's state-dict keys (convs.N.conv1.0.weight, convs.N.conv2.0.kernel, final_linear.0.weight ...).  The arithmetic is
in vts/engine.py (sg2d_forward / sg2d_backward) on the HIP kernels; these modules define no forward of their own.

Layer plan (reference lines): ConvLayer :622-668 = [Blur] -> EqualConv2d -> [FusedLeakyReLU]; ResBlock :671-693 =
conv1 (3x3) -> conv2 (Blur, 3x3 stride 2) + skip (Blur, 1x1 stride 2, linear), (out + skip) / sqrt 2; the discriminator :696-786 =
ConvLayer 1x1 -> ResBlocks down to 4x4 -> ConvLayer 3x3 -> EqualLinear(C*16, C, fused_lrelu) -> EqualLinear(C, 1).
"""
import math

import torch
import torch.nn as nn

BLUR_TAPS = (1.0, 3.0, 3.0, 1.0)


def make_kernel(k=BLUR_TAPS):
    """models/stylegan_networks.py:87-95"""
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


BLUR_KERNEL = [[float(v) for v in row] for row in make_kernel()]


class _Holder(nn.Module):
    """one Sequential slot of the reference: owns `weight` / `bias` parameters or the `kernel` buffer"""


class ConvLayer(nn.Module):
    def __init__(self, cin, cout, k, downsample=False, bias=True, activate=True):
        super().__init__()
        self.cin, self.cout, self.k, self.downsample, self.activate = cin, cout, k, downsample, activate
        i = 0
        if downsample:
            blur = _Holder()
            blur.register_buffer("kernel", make_kernel())
            self.add_module("0", blur)
            i = 1
        conv = _Holder()
        conv.weight = nn.Parameter(torch.randn(cout, cin, k, k))
        if bias and not activate:
            conv.bias = nn.Parameter(torch.zeros(cout))
        self.add_module(str(i), conv)
        self.conv = [conv]             # in a list: not registered twice
        self.act = [None]
        if activate and bias:
            act = _Holder()
            act.bias = nn.Parameter(torch.zeros(1, cout, 1, 1))
            self.add_module(str(i + 1), act)
            self.act = [act]
        p = 2 + (k - 1)
        self.blur_pad = ((p + 1) // 2, p // 2)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, downsample=True):
        super().__init__()
        self.conv1 = ConvLayer(cin, cin, 3)
        self.conv2 = ConvLayer(cin, cout, 3, downsample=downsample)
        # stylegan_networks.py:679-684: a 1x1 skip convolution only when the shape changes, nn.Identity otherwise
        self.skip = ConvLayer(cin, cout, 1, downsample=downsample, activate=False, bias=False) if (cin != cout or downsample) else None


class _Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))


def d_channels(ndf):
    m = ndf / 64
    return {4: min(384, int(4096 * m)), 8: min(384, int(2048 * m)), 16: min(384, int(1024 * m)), 32: min(384, int(512 * m)),
            64: int(256 * m), 128: int(128 * m), 256: int(64 * m), 512: int(32 * m), 1024: int(16 * m)}


class StyleGAN2Discriminator(nn.Module):
    is_stylegan2_d = True

    def __init__(self, input_nc, ndf=64, size=256):
        super().__init__()
        assert size >= 8 and size & (size - 1) == 0, "size must be a power of two >= 8"
        ch = d_channels(ndf)
        self.size, self.input_nc = size, input_nc
        convs = [ConvLayer(input_nc, ch[size], 1)]
        cin = ch[size]
        for i in range(int(math.log2(size)), 2, -1):
            convs.append(ResBlock(cin, ch[2 ** (i - 1)]))
            cin = ch[2 ** (i - 1)]
        self.convs = nn.Sequential(*convs)
        self.final_conv = ConvLayer(cin, ch[4], 3)
        self.final_linear = nn.Sequential(_Linear(ch[4] * 16, ch[4]), _Linear(ch[4], 1))

    def forward(self, x):
        from vts import engine
        return engine.sg2d_forward(self, x, keep=False)[0]



# ---- generator side (`--netG stylegan2 | smallstylegan2`, stylegan_networks.py:800-930): parameter containers with the reference's keys
# (encoder.convs.N.*, decoder.convs.N.{conv.weight, conv.blur.kernel, noise.weight, activate.bias}); arithmetic in vts/engine.py
# (sg2g_forward / sg2g_backward).
class StyledConvUp(nn.Module):
    """StyledConv(cin, cout, 3, upsample=True) :378-407 as the decoder uses it (style = None)"""

    def __init__(self, cin, cout, inject_noise):
        super().__init__()
        self.cin, self.cout, self.inject_noise = cin, cout, inject_noise
        self.conv = _Holder()
        self.conv.weight = nn.Parameter(torch.randn(1, cout, cin, 3, 3))
        self.conv.blur = _Holder()
        self.conv.blur.register_buffer("kernel", make_kernel() * 4)
        self.noise = _Holder()
        self.noise.weight = nn.Parameter(torch.zeros(1))
        self.activate = _Holder()
        self.activate.bias = nn.Parameter(torch.zeros(1, cout, 1, 1))


def g_channels(ngf):
    """stylegan_networks.py:805-816"""
    m = ngf / 32
    return {4: min(512, int(round(4096 * m))), 8: min(512, int(round(2048 * m))), 16: min(512, int(round(1024 * m))),
            32: min(512, int(round(512 * m))), 64: int(round(256 * m)), 128: int(round(128 * m)), 256: int(round(64 * m)),
            512: int(round(32 * m)), 1024: int(round(16 * m))}


class _Convs(nn.Module):
    def __init__(self, convs):
        super().__init__()
        self.convs = nn.Sequential(*convs)


class StyleGAN2Generator(nn.Module):
    """StyleGAN2Generator :915-930 = StyleGAN2Encoder :800-851 (Identity, ConvLayer 1x1, ResBlock(down) x num_downsampling, ResBlock x
    n_blocks / 2) + StyleGAN2Decoder :854-912 (ResBlock x n_blocks / 2, StyledConv(up) x num_downsampling, ConvLayer -> 3 channels).
    The reference emits THREE channels whatever output_nc says (:892)."""
    is_stylegan2_g = True

    def __init__(self, input_nc, output_nc, ngf=64, n_blocks=6, size=256, num_downsampling=1, inject_noise=True):
        super().__init__()
        ch = g_channels(ngf)
        res = 2 ** int(round(math.log2(size)))
        self.n_blocks, self.num_downsampling, self.inject_noise = n_blocks, num_downsampling, inject_noise
        enc = [nn.Identity(), ConvLayer(input_nc, ch[res], 1)]
        for _ in range(num_downsampling):
            enc.append(ResBlock(ch[res], ch[res // 2], downsample=True))
            res //= 2
        for _ in range(n_blocks // 2):
            enc.append(ResBlock(ch[res], ch[res], downsample=False))
        dec = [ResBlock(ch[res], ch[res], downsample=False) for _ in range(n_blocks // 2)]
        for _ in range(num_downsampling):
            dec.append(StyledConvUp(ch[res], ch[res * 2], inject_noise))
            res *= 2
        dec.append(ConvLayer(ch[res], 3, 1))
        self.encoder, self.decoder = _Convs(enc), _Convs(dec)

    def forward(self, x):
        from vts import engine
        return engine.sg2g_forward(self, x, keep=False)[0]
