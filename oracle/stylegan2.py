"""CPU restatement of the reference's StyleGAN2 building blocks (TEST INFRASTRUCTURE ONLY: imported by tests/, smoke() and
the cpu_baseline leg of bench.py, never by the product path).

Follows /root/reference/models/stylegan_networks.py:
  fused_leaky_relu :18-19, upfirdn2d_native :38-72, make_kernel :87-95, Blur :140-156, EqualConv2d :159-190,
  EqualLinear :199-227, ScaledLeakyReLU :236-245, ModulatedConv2d :248-348, ConvLayer :622-668, ResBlock :671-693,
  StyleGAN2Discriminator :696-786.
Pinned to the reference by tests/golden/stylegan2_32.npz (oracle/make_golden.py sg2: the reference modules run on CPU with the
same seeded weights), checked by tests/test_oracle_golden.py.  Functional form over a state dict with the reference's keys.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import detrand

SQRT2 = math.sqrt(2.0)

# (up, down, (pad0, pad1)) configurations shared by the golden generator and the tests: Blur of a downsampling ConvLayer with a
# 3x3 / 1x1 kernel, Blur after the transposed conv of an upsampling ModulatedConv2d, Upsample / Downsample modules, a crop
UPFIRDN_CASES = [(1, 1, (2, 2)), (1, 1, (1, 1)), (1, 1, (1, 1 - 0)), (2, 1, (2, 1)), (1, 2, (1, 1)), (1, 1, (-1, 2)), (2, 2, (0, 3))]


def make_kernel(k=(1, 3, 3, 1)):
    """stylegan_networks.py:87-95 -- outer product of the 1-D taps, normalised to sum 1"""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """stylegan_networks.py:38-76 -- zero-insertion upsampling by `up`, padding (negative = crop) by pad = (before, after) on both
    axes, correlation with the FLIPPED kernel, decimation by `down`"""
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    out = x.reshape(n * c, 1, h, 1, w, 1)
    out = F.pad(out, [0, up - 1, 0, 0, 0, up - 1, 0, 0]).reshape(n * c, 1, h * up, w * up)
    out = F.pad(out, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    out = out[:, :, max(-p0, 0): out.shape[2] - max(-p1, 0), max(-p0, 0): out.shape[3] - max(-p1, 0)]
    out = F.conv2d(out, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw))
    out = out.reshape(n, c, out.shape[2], out.shape[3])
    return out[:, :, ::down, ::down]


def fused_leaky_relu(x, bias, slope=0.2, scale=SQRT2):
    """stylegan_networks.py:18-19; bias is [1,C,1,1] (FusedLeakyReLU) or [C] (EqualLinear activation)"""
    if bias.ndim == 1:
        bias = bias.view(1, -1, *([1] * (x.ndim - 2)))
    return F.leaky_relu(x + bias, slope) * scale


def equal_conv2d(x, weight, bias=None, stride=1, padding=0):
    """stylegan_networks.py:159-190 -- runtime weight scale 1/sqrt(fan_in)"""
    scale = 1.0 / math.sqrt(weight.shape[1] * weight.shape[2] * weight.shape[3])
    return F.conv2d(x, weight * scale, bias=bias, stride=stride, padding=padding)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """stylegan_networks.py:199-227"""
    scale = (1.0 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, bias=bias * lr_mul)


def conv_layer(sd, prefix, x, kernel_size, downsample=False, bias=True, activate=True):
    """stylegan_networks.py:622-668 -- [Blur] -> EqualConv2d -> [FusedLeakyReLU | ScaledLeakyReLU]; Sequential indices as upstream"""
    i = 0
    if downsample:
        p = (4 - 2) + (kernel_size - 1)
        x = upfirdn2d(x, sd[prefix + "0.kernel"], pad=((p + 1) // 2, p // 2))
        i, stride, padding = 1, 2, 0
    else:
        stride, padding = 1, kernel_size // 2
    conv_bias = sd.get(prefix + "%d.bias" % i) if (bias and not activate) else None
    x = equal_conv2d(x, sd[prefix + "%d.weight" % i], conv_bias, stride, padding)
    if activate:
        if bias:
            x = fused_leaky_relu(x, sd[prefix + "%d.bias" % (i + 1)])
        else:
            x = F.leaky_relu(x, 0.2) * SQRT2
    return x


def res_block(sd, prefix, x, downsample=True, has_skip=True, skip_gain=1.0):
    """stylegan_networks.py:671-693"""
    out = conv_layer(sd, prefix + "conv1.", x, 3)
    out = conv_layer(sd, prefix + "conv2.", out, 3, downsample=downsample)
    skip = conv_layer(sd, prefix + "skip.", x, 1, downsample=downsample, activate=False, bias=False) if has_skip else x
    return (out * skip_gain + skip) / math.sqrt(skip_gain ** 2 + 1.0)


def d_channels(ndf):
    """stylegan_networks.py:707-719"""
    m = ndf / 64
    return {4: min(384, int(4096 * m)), 8: min(384, int(2048 * m)), 16: min(384, int(1024 * m)), 32: min(384, int(512 * m)),
            64: int(256 * m), 128: int(128 * m), 256: int(64 * m), 512: int(32 * m), 1024: int(16 * m)}


def d_param_shapes(input_nc, ndf, size):
    """learnable parameters of StyleGAN2Discriminator(netD='stylegan2') (stylegan_networks.py:696-753), reference key names"""
    ch = d_channels(ndf)
    shapes = {"convs.0.0.weight": (ch[size], input_nc, 1, 1), "convs.0.1.bias": (1, ch[size], 1, 1)}
    cin = ch[size]
    log_size = int(math.log2(size))
    for n, i in enumerate(range(log_size, 2, -1)):
        cout = ch[2 ** (i - 1)]
        p = "convs.%d." % (n + 1)
        shapes[p + "conv1.0.weight"] = (cin, cin, 3, 3)
        shapes[p + "conv1.1.bias"] = (1, cin, 1, 1)
        shapes[p + "conv2.1.weight"] = (cout, cin, 3, 3)
        shapes[p + "conv2.2.bias"] = (1, cout, 1, 1)
        shapes[p + "skip.1.weight"] = (cout, cin, 1, 1)
        cin = cout
    shapes["final_conv.0.weight"] = (ch[4], cin, 3, 3)
    shapes["final_conv.1.bias"] = (1, ch[4], 1, 1)
    shapes["final_linear.0.weight"] = (ch[4], ch[4] * 16)
    shapes["final_linear.0.bias"] = (ch[4],)
    shapes["final_linear.1.weight"] = (1, ch[4])
    shapes["final_linear.1.bias"] = (1,)
    return shapes


def d_buffers(input_nc, ndf, size):
    """the registered Blur kernels (stylegan_networks.py:147-149)"""
    out = {}
    for n in range(int(math.log2(size)) - 2):
        out["convs.%d.conv2.0.kernel" % (n + 1)] = make_kernel()
        out["convs.%d.skip.0.kernel" % (n + 1)] = make_kernel()
    return out


def test_weights(shapes, seed):
    """unit-variance weights (the blocks carry their own 1/sqrt(fan_in) runtime scale), O(0.3) biases"""
    sd = {}
    for k, shp in shapes.items():
        u = detrand.uniform(tuple(shp), seed, k)
        sd[k] = u * (math.sqrt(3.0) if k.endswith("weight") else 0.3)
    return sd


def discriminator_forward(sd, x, size):
    """StyleGAN2Discriminator.forward (stylegan_networks.py:755-786), netD='stylegan2' (no patch crop, no minibatch-stddev:
    the upstream code disables it with `if False and ...`)"""
    sd = dict(sd)
    for k, v in d_buffers(0, 0, size).items():
        sd.setdefault(k, v)
    out = conv_layer(sd, "convs.0.", x, 1)
    for n in range(int(math.log2(size)) - 2):
        out = res_block(sd, "convs.%d." % (n + 1), out)
    out = conv_layer(sd, "final_conv.", out, 3)
    out = out.reshape(out.shape[0], -1)
    out = equal_linear(out, sd["final_linear.0.weight"], sd["final_linear.0.bias"], activation=True)
    return equal_linear(out, sd["final_linear.1.weight"], sd["final_linear.1.bias"])


def modulated_conv2d(x, style, weight, mod_weight, mod_bias, demodulate=True, upsample=False, downsample=False):
    """ModulatedConv2d.forward (stylegan_networks.py:304-348) with a style vector; weight [1,Co,Ci,K,K]; the modulation is an
    EqualLinear(style_dim, Ci, bias_init=1).  Grouped-conv formulation as upstream (batch folded into groups)."""
    n, ci, h, w = x.shape
    _, co, _, k, _ = weight.shape
    s = equal_linear(style, mod_weight, mod_bias).view(n, 1, ci, 1, 1)
    wgt = (1.0 / math.sqrt(ci * k * k)) * weight * s
    if demodulate:
        wgt = wgt * torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8).view(n, co, 1, 1, 1)
    if upsample:
        p = (4 - 2) - (k - 1)
        wt = wgt.transpose(1, 2).reshape(n * ci, co, k, k)
        out = F.conv_transpose2d(x.reshape(1, n * ci, h, w), wt, padding=0, stride=2, groups=n)
        out = out.view(n, co, out.shape[2], out.shape[3])
        return upfirdn2d(out, make_kernel() * 4, pad=((p + 1) // 2 + 1, p // 2 + 1))
    if downsample:
        p = (4 - 2) + (k - 1)
        x = upfirdn2d(x, make_kernel(), pad=((p + 1) // 2, p // 2))
        h, w = x.shape[2:]
        out = F.conv2d(x.reshape(1, n * ci, h, w), wgt.view(n * co, ci, k, k), padding=0, stride=2, groups=n)
        return out.view(n, co, out.shape[2], out.shape[3])
    out = F.conv2d(x.reshape(1, n * ci, h, w), wgt.view(n * co, ci, k, k), padding=k // 2, groups=n)
    return out.view(n, co, out.shape[2], out.shape[3])



# ---- generator side (stylegan_networks.py:351-407, 800-930) -------------------------------------------------------------------------
def g_channels(ngf):
    """stylegan_networks.py:805-816"""
    m = ngf / 32
    return {4: min(512, int(round(4096 * m))), 8: min(512, int(round(2048 * m))), 16: min(512, int(round(1024 * m))),
            32: min(512, int(round(512 * m))), 64: int(round(256 * m)), 128: int(round(128 * m)), 256: int(round(64 * m)),
            512: int(round(32 * m)), 1024: int(round(16 * m))}


def styled_conv_up(sd, prefix, x, noise=None):
    """StyledConv(upsample=True).forward with style None (:399-407): ModulatedConv2d with s = 1 (the reference builds the ones tensor
    with .cuda(), :309-310), Blur(kernel * 4, pad (1, 1)), optional NoiseInjection (:351-363), FusedLeakyReLU"""
    w = sd[prefix + "conv.weight"]
    n, ci = x.shape[0], x.shape[1]
    out = modulated_conv2d_nostyle_up(x, w)
    if noise is not None:
        out = out + sd[prefix + "noise.weight"] * noise
    return fused_leaky_relu(out, sd[prefix + "activate.bias"])


def modulated_conv2d_nostyle_up(x, weight):
    n, ci, h, w = x.shape
    _, co, _, k, _ = weight.shape
    wgt = (1.0 / math.sqrt(ci * k * k)) * weight * torch.ones(n, 1, ci, 1, 1)
    wgt = wgt * torch.rsqrt(wgt.pow(2).sum([2, 3, 4]) + 1e-8).view(n, co, 1, 1, 1)
    p = (4 - 2) - (k - 1)
    wt = wgt.transpose(1, 2).reshape(n * ci, co, k, k)
    out = F.conv_transpose2d(x.reshape(1, n * ci, h, w), wt, padding=0, stride=2, groups=n)
    out = out.view(n, co, out.shape[2], out.shape[3])
    return upfirdn2d(out, make_kernel() * 4, pad=((p + 1) // 2 + 1, p // 2 + 1))


def g_layout(ngf, size, n_blocks, num_downsampling):
    """(encoder entries, decoder entries): ('conv', cin, cout, k) | ('res', cin, cout, downsample) | ('up', cin, cout)"""
    ch = g_channels(ngf)
    res = 2 ** int(round(math.log2(size)))
    enc, dec = [], []
    for _ in range(num_downsampling):
        enc.append(("res", ch[res], ch[res // 2], True))
        res //= 2
    for _ in range(n_blocks // 2):
        enc.append(("res", ch[res], ch[res], False))
    for _ in range(n_blocks // 2):
        dec.append(("res", ch[res], ch[res], False))
    for _ in range(num_downsampling):
        dec.append(("up", ch[res], ch[res * 2]))
        res *= 2
    return ch[2 ** int(round(math.log2(size)))], enc, dec, ch[res]


def g_param_shapes(input_nc, ngf, size, n_blocks, num_downsampling):
    """learnable parameters of StyleGAN2Generator, reference key names (probe: state_dict of the reference module)"""
    c0, enc, dec, clast = g_layout(ngf, size, n_blocks, num_downsampling)
    shapes = {"encoder.convs.1.0.weight": (c0, input_nc, 1, 1), "encoder.convs.1.1.bias": (1, c0, 1, 1)}

    def res(p, cin, cout, down):
        i = 1 if down else 0
        shapes[p + "conv1.0.weight"] = (cin, cin, 3, 3)
        shapes[p + "conv1.1.bias"] = (1, cin, 1, 1)
        shapes[p + "conv2.%d.weight" % i] = (cout, cin, 3, 3)
        shapes[p + "conv2.%d.bias" % (i + 1)] = (1, cout, 1, 1)
        if down or cin != cout:
            shapes[p + "skip.%d.weight" % i] = (cout, cin, 1, 1)

    for j, (_, cin, cout, down) in enumerate(enc):
        res("encoder.convs.%d." % (j + 2), cin, cout, down)
    for j, e in enumerate(dec):
        p = "decoder.convs.%d." % j
        if e[0] == "res":
            res(p, e[1], e[2], e[3])
        else:
            shapes[p + "conv.weight"] = (1, e[2], e[1], 3, 3)
            shapes[p + "noise.weight"] = (1,)
            shapes[p + "activate.bias"] = (1, e[2], 1, 1)
    p = "decoder.convs.%d." % len(dec)
    shapes[p + "0.weight"] = (3, clast, 1, 1)
    shapes[p + "1.bias"] = (1, 3, 1, 1)
    return shapes


def generator_forward(sd, x, ngf, size, n_blocks, num_downsampling, noises=None):
    """StyleGAN2Generator.forward (:922-930) = encoder (:838-851) then decoder (:897-912)"""
    sd = dict(sd)
    _, enc, dec, _ = g_layout(ngf, size, n_blocks, num_downsampling)
    k = make_kernel()
    out = conv_layer(sd, "encoder.convs.1.", x, 1)
    for j, (_, cin, cout, down) in enumerate(enc):
        p = "encoder.convs.%d." % (j + 2)
        sd.setdefault(p + "conv2.0.kernel", k)
        sd.setdefault(p + "skip.0.kernel", k)
        out = res_block(sd, p, out, downsample=down, has_skip=down or cin != cout)
    u = 0
    for j, e in enumerate(dec):
        p = "decoder.convs.%d." % j
        if e[0] == "res":
            out = res_block(sd, p, out, downsample=False, has_skip=e[1] != e[2])
        else:
            out = styled_conv_up(sd, p, out, None if noises is None else noises[u])
            u += 1
    return conv_layer(sd, "decoder.convs.%d." % len(dec), out, 1)
