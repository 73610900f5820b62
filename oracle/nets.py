"""CPU restatement of the reference networks and operators (TEST INFRASTRUCTURE ONLY).

This file is the *oracle*: a plain PyTorch-CPU fp32 restatement of the reference's
hot-path arithmetic, written functionally over `state_dict`s that use the
reference's key names.  It is pinned against golden vectors produced by running
the reference itself (oracle/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product path never does.

Each function cites the reference lines it restates (paths relative to
/root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# positional encoding, DiffAugment, normals
# ----------------------------------------------------------------------------


def spe_grid(n, h, w, dim=4, device="cpu"):
    """SinusoidalPositionalEmbedding(dim, padding_idx=0)(x) for a 4-D x.

    thirdparty/mmgeneration/positional_encoding.py:54-82 (get_embedding),
    :118-160 (make_grid2d): positions run 1..W / 1..H, emb[p] = [sin(p f_i)..., cos(p f_i)...],
    f_i = exp(-i ln(1e4)/(dim/2-1)); grid = cat(x-embedding over rows, y-embedding over cols).
    """
    half = dim // 2
    f = torch.exp(torch.arange(half, dtype=torch.float) * -(np.log(10000) / (half - 1)))

    def axis(length):
        pos = torch.arange(length + 1, dtype=torch.float).unsqueeze(1) * f.unsqueeze(0)
        emb = torch.cat([torch.sin(pos), torch.cos(pos)], dim=1)  # [L+1, dim]
        return emb[1:].t()  # [dim, L]; row 0 is the padding row

    xe = axis(w)[:, None, :].expand(dim, h, w)
    ye = axis(h)[:, :, None].expand(dim, h, w)
    grid = torch.cat([xe, ye], 0)[None].expand(n, 2 * dim, h, w)
    return grid.contiguous().to(device)


def diffaug_bs(x, r_b, r_s):
    """DiffAugment policy 'bs' with the drawn uniforms passed in.

    thirdparty/DiffAugment.py:25-33: brightness x + (r_b - 0.5); saturation
    (x - mean_c x) * (2 r_s) + mean_c x.  r_b, r_s: [N] tensors in [0,1).
    """
    x = x + (r_b.view(-1, 1, 1, 1) - 0.5)
    m = x.mean(dim=1, keepdim=True)
    return ((x - m) * (r_s.view(-1, 1, 1, 1) * 2) + m).contiguous()


def diffaug_draws(policy, shape, dtype=torch.float32):
    """The random numbers DiffAugment(x, policy) consumes for x of `shape` [N, C, H, W], drawn from torch's global generator with the
    reference's own calls in the reference's order (thirdparty/DiffAugment.py:25-80, AUGMENT_FNS :89-96), so that after
    torch.manual_seed(s) this returns exactly what the reference draws after the same seed.  One dict per policy letter:
      b / s / c : r  [N] uniform            t : tx, ty [N] int64 row / column translation
      o : ox, oy [N] int64 cutout centre     n : sigma [N] (gate already applied), noise [N, C, H, W]"""
    n, c, h, w = shape
    out = []
    for letter in policy:
        if letter in "bsc":
            out.append({"r": torch.rand(n, 1, 1, 1, dtype=dtype).flatten()})
        elif letter == "t":
            sx, sy = int(h * 0.125 + 0.5), int(w * 0.125 + 0.5)
            tx = torch.randint(-sx, sx + 1, size=[n, 1, 1])
            ty = torch.randint(-sy, sy + 1, size=[n, 1, 1])
            out.append({"tx": tx.flatten(), "ty": ty.flatten()})
        elif letter == "o":
            cs = int(h * 0.5 + 0.5), int(w * 0.5 + 0.5)
            ox = torch.randint(0, h + (1 - cs[0] % 2), size=[n, 1, 1])
            oy = torch.randint(0, w + (1 - cs[1] % 2), size=[n, 1, 1])
            out.append({"ox": ox.flatten(), "oy": oy.flatten()})
        elif letter == "n":
            sigma = torch.rand(n, 1, 1, 1, dtype=dtype).abs() * 0.1
            sigma = torch.where(torch.rand(n, 1, 1, 1, dtype=dtype) < 0.5, sigma, torch.zeros_like(sigma))
            out.append({"sigma": sigma.flatten(), "noise": torch.randn(n, c, h, w, dtype=dtype)})
        else:
            raise KeyError(letter)
    return out


def diffaug(x, policy, draws):
    """DiffAugment(x, policy) (thirdparty/DiffAugment.py:9-80) with the drawn numbers passed in (diffaug_draws); letters apply in order.
      b  x + (r - 0.5)                                   s  (x - mean_c x) * 2 r + mean_c x
      c  (x - mean_chw x) * (r + 0.5) + mean_chw x       t  out[i, j] = x[i + tx, j + ty], zero outside (zero pad 1 + clamped gather)
      o  rows / columns clamp(o - size // 2 + [0, size)) zeroed, size = int(0.5 * extent + 0.5)
      n  x + sigma * noise"""
    n, c, h, w = x.shape
    v = lambda t: t.to(x.dtype).view(-1, 1, 1, 1)
    for letter, d in zip(policy, draws):
        if letter == "b":
            x = x + (v(d["r"]) - 0.5)
        elif letter == "s":
            m = x.mean(dim=1, keepdim=True)
            x = (x - m) * (v(d["r"]) * 2) + m
        elif letter == "c":
            m = x.mean(dim=[1, 2, 3], keepdim=True)
            x = (x - m) * (v(d["r"]) + 0.5) + m
        elif letter == "t":
            rows = torch.arange(h).view(1, h, 1) + d["tx"].view(-1, 1, 1)
            cols = torch.arange(w).view(1, 1, w) + d["ty"].view(-1, 1, 1)
            ok = ((rows >= 0) & (rows < h) & (cols >= 0) & (cols < w)).unsqueeze(1)
            b = torch.arange(n).view(-1, 1, 1)
            g = x.permute(0, 2, 3, 1)[b, rows.clamp(0, h - 1).expand(n, h, w), cols.clamp(0, w - 1).expand(n, h, w)].permute(0, 3, 1, 2)
            x = g * ok.to(x.dtype)
        elif letter == "o":
            ch, cw = int(h * 0.5 + 0.5), int(w * 0.5 + 0.5)
            mask = torch.ones(n, 1, h, w, dtype=x.dtype)
            for i in range(n):
                r0, r1 = int(d["ox"][i]) - ch // 2, int(d["ox"][i]) - ch // 2 + ch - 1
                c0, c1 = int(d["oy"][i]) - cw // 2, int(d["oy"][i]) - cw // 2 + cw - 1
                r0, r1 = min(max(r0, 0), h - 1), min(max(r1, 0), h - 1)
                c0, c1 = min(max(c0, 0), w - 1), min(max(c1, 0), w - 1)
                mask[i, 0, r0:r1 + 1, c0:c1 + 1] = 0
            x = x * mask
        elif letter == "n":
            x = x + v(d["sigma"]) * d["noise"]
        else:
            raise KeyError(letter)
    return x.contiguous()


def compute_normal(t, scale_nz):
    """models/model_utils.py:408-428."""
    gx, gy = t[:, 0:1], t[:, 1:2]
    return F.normalize(torch.cat([gx, gy, scale_nz * torch.ones_like(gx)], 1), dim=1)


def eval_metrics(real_I, fake_I, real_T, fake_T):
    """compute_evaluation_metric (models/model_utils.py:431-561), the metrics that need no pretrained network:
    I_PSNR (:481-496; torchmetrics' peak_signal_noise_ratio(data_range=1) = 10 log10(1 / mse)), T_AE (:531-536 with
    normal_losses.py:10-33 mode 'evaluate'), T_MSE (:557).  The fake tactile patches are clamped to [0, 1] (:521)."""
    lo, hi = real_I.min(), real_I.max()
    r = (real_I - lo) / (hi - lo)
    f = torch.clamp((fake_I - lo) / (hi - lo), 0, 1)
    psnr = 10.0 * torch.log10(1.0 / torch.mean((r - f) ** 2))
    fT = torch.clamp(fake_T, 0, 1)
    cos = torch.clamp(torch.cosine_similarity(compute_normal(fT, 1), compute_normal(real_T, 1), dim=1, eps=1e-6), -1.0, 1.0)
    out = {"I_PSNR": float(psnr), "T_AE": float((torch.acos(cos) * 180.0 / np.pi).mean()), "T_MSE": float(torch.mean((real_T - fT) ** 2))}
    if min(real_I.shape[2:]) >= 11:
        out["I_SSIM"] = float(ssim(r, f))
    return out


# (dims, positions of set 1, positions of set 2) of the synthetic Frechet-distance cases shared by the golden generator and the tests
FRECHET_CASES = [(64, 4096, 4096), (64, 1000, 777), (16, 300, 300), (64, 200, 200), (3, 50, 64)]


def frechet_case(i, seed):
    """channel-major features [D, P]: correlated channels with different means / scales for the two sets"""
    from oracle import detrand
    d, p1, p2 = FRECHET_CASES[i]
    mix = detrand.uniform((d, d), seed, "fd_mix%d" % i) * (1.0 / d ** 0.5) + torch.eye(d)
    f1 = mix @ detrand.uniform((d, p1), seed, "fd_a%d" % i) + 0.3 * detrand.uniform((d, 1), seed, "fd_m%d" % i)
    f2 = 1.3 * (mix.t() @ detrand.uniform((d, p2), seed, "fd_b%d" % i)) + 0.1
    return f1.contiguous(), f2.contiguous()


def frechet_distance(f1, f2):
    """models/sifid.py:102-176 on channel-major features [D, P] (float64 numpy + scipy.linalg.sqrtm, as the reference)"""
    from scipy import linalg
    a1, a2 = f1.numpy().T.astype(np.float64), f2.numpy().T.astype(np.float64)
    mu1, mu2, s1, s2 = np.mean(a1, axis=0), np.mean(a2, axis=0), np.atleast_2d(np.cov(a1, rowvar=False)), np.atleast_2d(np.cov(a2, rowvar=False))
    covmean, _ = linalg.sqrtm(s1.dot(s2), disp=False)
    diff = mu1 - mu2
    return float(diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean.real))


def ssim(target, preds, data_range=1.0, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """I_SSIM (models/model_utils.py:498-499) = torchmetrics.functional.structural_similarity_index_measure(data_range=1).
    torchmetrics is an un-pinned pip dependency of the reference (requirements.txt:18) that is absent from this image, so this is a
    restatement of its published algorithm (torchmetrics/functional/image/ssim.py, _ssim_update: Gaussian window from
    exp(-(d / sigma)^2 / 2) normalised to sum 1, reflect padding by (k-1)/2, five grouped-convolution moments, variances clamped at
    0, the padded border cropped, per-image mean then batch mean) -- PARITY UNPINNED for this one metric (no reference run possible)."""
    n, c, h, w = preds.shape
    pad = (kernel_size - 1) // 2
    d = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1.0, dtype=preds.dtype)
    g = torch.exp(-((d / sigma) ** 2) / 2)
    g = (g / g.sum()).unsqueeze(0)
    kernel = (g.t() @ g).expand(c, 1, kernel_size, kernel_size)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    p = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    outs = F.conv2d(torch.cat((p, t, p * p, t * t, p * t)), kernel, groups=c).split(n)
    mu_p2, mu_t2, mu_pt = outs[0] ** 2, outs[1] ** 2, outs[0] * outs[1]
    s_p, s_t, s_pt = torch.clamp(outs[2] - mu_p2, min=0.0), torch.clamp(outs[3] - mu_t2, min=0.0), outs[4] - mu_pt
    full = ((2 * mu_pt + c1) * (2 * s_pt + c2)) / ((mu_p2 + mu_t2 + c1) * (s_p + s_t + c2))
    return full[..., pad:-pad, pad:-pad].reshape(n, -1).mean(-1).mean()


# ----------------------------------------------------------------------------
# patch gather
# ----------------------------------------------------------------------------


def find_coords_for_patch(coords, scale_multiplier=1):
    """models/model_utils.py:23-69.  coords [NT,8] float64 -> int32 (offx, offy, cutout)."""
    c = np.asarray(coords, dtype=np.float64).reshape(-1, 8)
    ox = np.round((c[:, 0] + c[:, -2] / c[:, -3]) * scale_multiplier)
    oy = np.round((c[:, 1] + c[:, -1] / c[:, -3]) * scale_multiplier)
    cs = np.round(c[:, -4] / c[:, -3] * scale_multiplier)
    to_i = lambda a: torch.FloatTensor(a).to(torch.int32)
    return to_i(ox), to_i(oy), to_i(cs)


def gather_patches(img, offx, offy, size):
    """models/model_utils.py:252-333: clamp-to-border crop of `size` x `size` windows.

    img [1,C,H,W]; offx/offy int tensors [P].  Returns [P,C,size,size].
    (The reference materialises img.repeat(P,...) and fancy-indexes; same values.)
    """
    assert img.shape[0] == 1
    H, W = img.shape[-2:]
    ar = torch.arange(size, device=img.device)
    ys = (offy.to(img.device).long()[:, None] + ar[None]).clamp(0, H - 1)  # [P,size]
    xs = (offx.to(img.device).long()[:, None] + ar[None]).clamp(0, W - 1)
    out = img[0][:, ys[:, :, None], xs[:, None, :]]  # [C,P,size,size]
    return out.permute(1, 0, 2, 3).contiguous()


def dilated_mask_positions(M):
    """models/model_utils.py:212-216: 17x17 all-ones conv (padding 1) + clamp, nonzero().

    M [1,1,H,W] -> LongTensor [K,2] of (y, x) in row-major order.  The map is
    (H-14) x (W-14) and its indices are used un-shifted as patch offsets (quirk kept).
    """
    k = torch.ones(1, 1, 17, 17, dtype=M.dtype, device=M.device)
    e = torch.clamp(F.conv2d(M, k, padding=(1, 1)), 0, 1)
    return torch.nonzero(e, as_tuple=False)[:, -2:]


# ----------------------------------------------------------------------------
# generator  (models/networks.py:1430-1645, thirdparty/unet/unet_parts_custom.py:9-79)
# ----------------------------------------------------------------------------


def _inorm(x):
    return F.instance_norm(x, eps=1e-5)


def style_map(sd, j, style_code, h, w, bn_train=True):
    """style_code_mapping<j> (networks.py:1459-1465, 1611-1615): Linear(bias=False) -> BatchNorm1d (keys present) | InstanceNorm1d -> ReLU,
    reshaped to [N, -1, h, w].  BatchNorm1d in training mode (batch statistics; the running buffers are not tracked here)."""
    p = "style_code_mapping%d." % j
    z = F.linear(style_code.to(torch.float32), sd[p + "0.weight"])
    if p + "1.weight" in sd:
        if bn_train:
            z = F.batch_norm(z, None, None, sd[p + "1.weight"], sd[p + "1.bias"], True, 0.1, 1e-5)
        else:
            z = F.batch_norm(z, sd[p + "1.running_mean"], sd[p + "1.running_var"], sd[p + "1.weight"], sd[p + "1.bias"], False, 0.1, 1e-5)
    else:
        z = F.instance_norm(z[None], eps=1e-5)[0]     # InstanceNorm1d on a 2-D tensor: an unbatched (C = N, L = P) input
    return F.relu(z).reshape(style_code.shape[0], -1, h, w)


def adain(content, style, eps=1e-5):
    """adaptive_instance_normalization (thirdparty/AdaIN/function.py:4-23): unbiased variance + eps under the square root"""
    n, c = content.shape[:2]

    def ms(t):
        v = t.reshape(n, c, -1)
        return v.mean(2).view(n, c, 1, 1), (v.var(dim=2) + eps).sqrt().view(n, c, 1, 1)

    sm, ss = ms(style)
    cm, cs = ms(content)
    return (content - cm) / cs * ss + sm


def dropout_draws(shape_n_h_w, ngf=10, num_downs=8):
    """keep masks of the Up-block dropouts (--no_dropout False: Dropout(0.5) behind the InstanceNorm of up<num_downs-2> .. up<num_downs//2>,
    networks.py:1508-1519, unet_parts_custom.py:66-67) drawn from torch's global generator as F.dropout draws them on the CPU
    (`empty_like(x).bernoulli_(1 - p)`), in forward order: {layer: [N, 8 ngf, h, w] of 0 / 1}"""
    n, h, w = shape_n_h_w
    out = {}
    for i in range(num_downs - 2, num_downs // 2 - 1, -1):
        out[i] = torch.empty(n, ngf * 8, h >> i, w >> i).bernoulli_(0.5)
    return out


def unet_forward(sd, x, num_downs=8, num_layer_separate=4, style_code=None, num_layer_style_code=1,
                 return_feats=False, style_mode="concat", style_mapping="tile", dropout_masks=None):
    """CustomUnetGenerator.forward (networks.py:1576-1645), instance norm; style code concat / adain x tile / project (:1600-1632);
    dropout_masks: {layer: keep mask} of the train-mode Dropout(0.5) behind the norm of the intermediate Up blocks (dropout_draws)"""
    feats = []
    for i in range(num_downs):
        key = "down%d.model.%d" % (i, 0 if i == 0 else 1)
        if i > 0:
            x = F.leaky_relu(x, 0.2)
        x = F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=2, padding=1)
        if 0 < i < num_downs - 1:
            x = _inorm(x)
        feats.append(x)
    ups = {}
    x_t = None
    for i in range(num_downs - 1, -1, -1):
        skip = feats[i]
        if style_code is not None and i >= num_downs - num_layer_style_code:
            if style_mapping == "tile":
                sc = style_code.to(torch.float32)[..., None, None].expand(-1, -1, skip.shape[2], skip.shape[3])
            else:
                sc = style_map(sd, num_downs - i - 1, style_code, skip.shape[2], skip.shape[3])
            if style_mode == "concat":
                x = torch.cat([x, sc], 1)
                if x_t is not None:
                    x_t = torch.cat([x_t, sc], 1)
            else:
                x = adain(x, sc)
                if x_t is not None:
                    x_t = adain(x_t, sc)

        def up(name, inp):
            if i == 0 or i == num_downs - 1:  # outermost / innermost: no skip concat
                z = inp
            else:
                z = torch.cat([inp, skip], 1)
            z = F.relu(z)
            z = F.conv_transpose2d(z, sd[name + ".model.1.weight"], sd[name + ".model.1.bias"], stride=2, padding=1)
            if i == 0:
                return torch.tanh(z)
            z = _inorm(z)
            if dropout_masks is not None and i in dropout_masks and not name.endswith("_T"):
                z = z * dropout_masks[i] / 0.5
            return z

        if num_layer_separate >= i + 1:
            if x_t is None:
                x_t = x
            x_t = up("up%d_T" % i, x_t)
            ups["up%d_T" % i] = x_t
        x = up("up%d" % i, x)
        ups["up%d" % i] = x
    out = torch.cat([x, x_t], 1) if x_t is not None else x
    if return_feats:
        return out, feats, ups
    return out


# ----------------------------------------------------------------------------
# ResNet generator with anti-aliased resampling  (models/networks.py:1051-1154; ResnetBlock :1267-1324;
# Downsample :51-74 (filt 3, stride 2, reflect pad); Upsample :87-107 (filt 4, stride 2, replicate pad))
# reference defaults: InstanceNorm, reflect padding, no dropout, anti-aliased down- and up-sampling
# ----------------------------------------------------------------------------
def blur_down(x):
    """Downsample.forward (networks.py:66-74): reflect pad [1,1,1,1], depthwise [1,2,1]x[1,2,1]/16, stride 2."""
    c = x.shape[1]
    a = torch.tensor([1.0, 2.0, 1.0], dtype=x.dtype, device=x.device)
    f = (a[:, None] * a[None, :]) / 16.0
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), f[None, None].repeat(c, 1, 1, 1), stride=2, groups=c)


def blur_up(x):
    """Upsample.forward (networks.py:101-107): replicate pad 1, depthwise conv_transpose [1,3,3,1]x[1,3,3,1]*4/64,
    stride 2, padding 2, then [1:, 1:] and (even filter) [:-1, :-1]."""
    c = x.shape[1]
    a = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=x.dtype, device=x.device)
    f = (a[:, None] * a[None, :]) / 64.0 * 4.0
    y = F.conv_transpose2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), f[None, None].repeat(c, 1, 1, 1), stride=2, padding=2, groups=c)
    return y[:, :, 1:, 1:][:, :, :-1, :-1]


def resnet_layout(n_blocks=9, n_down=2, down="blur", up="blur"):
    """nn.Sequential indices of ResnetGenerator.model (networks.py:1075-1147) / GlobalGenerator.model
    (networks.py:1959-1975) -> [(kind, idx)]; idx = index of the group's first module."""
    lay, idx = [], 0

    def add(kind, n=1):
        nonlocal idx
        lay.append((kind, idx))
        idx += n

    add("conv7_in", 4)                 # pad, conv, norm, relu
    for _ in range(n_down):
        add("conv3_down", 4 if down == "blur" else 3)   # conv, norm, relu(, Downsample)
    for _ in range(n_blocks):
        add("block", 1)
    for _ in range(n_down):
        add("up_conv3", 4 if up == "blur" else 3)       # (Upsample,) conv | ConvTranspose, norm, relu
    add("conv7_out", 3)                # pad, conv, tanh
    return lay


def _gnorm(sd, key, x, norm, training):
    """InstanceNorm2d(affine=False) or BatchNorm2d at `key` (training: batch stats + running update)."""
    if norm == "instance":
        return _inorm(x)
    y = F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"], training, 0.1, 1e-5)
    if training and key + ".num_batches_tracked" in sd:
        sd[key + ".num_batches_tracked"] += 1
    return y


def resnet_dropout_draws(shape_n_h_w, ngf=10, n_blocks=9, n_down=2):
    """keep masks of the blocks' Dropout(0.5) (--no_dropout False; ResnetBlock.build_conv_block, networks.py:1305-1306) drawn from torch's
    global generator as F.dropout draws them on the CPU, in block order: a list of [N, ngf * 2^n_down, h / 2^n_down, w / 2^n_down] of 0 / 1"""
    n, h, w = shape_n_h_w
    return [torch.empty(n, ngf * 2 ** n_down, h >> n_down, w >> n_down).bernoulli_(0.5) for _ in range(n_blocks)]


def resnet_forward(sd, x, n_blocks=9, n_down=2, norm="instance", down="blur", up="blur", training=True, dropout_masks=None):
    """ResnetGenerator.forward (reference defaults: instance / blur / blur) and, with norm='batch', down='stride',
    up='convT', pix2pixHD's GlobalGenerator.forward.  Missing biases (`use_bias=False` with BatchNorm) are None.
    dropout_masks (resnet_dropout_draws): the generator was built with use_dropout -- Dropout(0.5) behind the first ReLU of every block,
    which also moves the block's second conv / norm from conv_block.5 / .6 to .6 / .7 in the state dict."""
    masks = list(dropout_masks) if dropout_masks is not None else None
    kb = 6 if masks is not None else 5
    def wb(i):
        return sd["model.%d.weight" % i], sd.get("model.%d.bias" % i)

    for kind, i in resnet_layout(n_blocks, n_down, down, up):
        if kind == "conv7_in":
            w, b = wb(i + 1)
            x = F.relu(_gnorm(sd, "model.%d" % (i + 2), F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w, b), norm, training))
        elif kind == "conv3_down":
            w, b = wb(i)
            x = F.relu(_gnorm(sd, "model.%d" % (i + 1), F.conv2d(x, w, b, stride=1 if down == "blur" else 2, padding=1), norm, training))
            if down == "blur":
                x = blur_down(x)
        elif kind == "block":
            k = "model.%d.conv_block." % i
            y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[k + "1.weight"], sd.get(k + "1.bias"))
            y = F.relu(_gnorm(sd, k + "2", y, norm, training))
            if masks is not None and training:
                y = y * masks.pop(0) / 0.5
            y = F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), sd[k + "%d.weight" % kb], sd.get(k + "%d.bias" % kb))
            x = x + _gnorm(sd, k + "%d" % (kb + 1), y, norm, training)
        elif kind == "up_conv3":
            if up == "blur":
                w, b = wb(i + 1)
                x = F.relu(_gnorm(sd, "model.%d" % (i + 2), F.conv2d(blur_up(x), w, b, padding=1), norm, training))
            else:
                w, b = wb(i)
                x = F.relu(_gnorm(sd, "model.%d" % (i + 1), F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1), norm, training))
        else:
            w, b = wb(i + 1)
            x = torch.tanh(F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), w, b))
    return x


def local_enhancer_forward(sd, x, n_down=3, n_blocks_global=9, n_blocks_local=3, norm="batch", training=True, n_local=1):
    """pix2pixHD LocalEnhancer.forward with n_local enhancers (models/networks.py:1897-1949): the global trunk on the input average-pooled
    n_local times, then enhancer n = 1 .. n_local on the pyramid level n_local - n: model<n>_1 (7x7 conv, stride-2 3x3 conv) + the output
    below, through model<n>_2 (blocks, ConvTranspose2d; the last one ends in the 7x7 conv + tanh)"""
    def c7(pre, i, t):
        return F.conv2d(F.pad(t, (3, 3, 3, 3), mode="reflect"), sd["%s.%d.weight" % (pre, i)], sd["%s.%d.bias" % (pre, i)])

    def nr(pre, i, t):
        return F.relu(_gnorm(sd, "%s.%d" % (pre, i), t, norm, training))

    def block(pre, i, t):
        k = "%s.%d.conv_block." % (pre, i)
        y = F.conv2d(F.pad(t, (1, 1, 1, 1), mode="reflect"), sd[k + "1.weight"], sd[k + "1.bias"])
        y = F.relu(_gnorm(sd, k + "2", y, norm, training))
        y = F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), sd[k + "5.weight"], sd[k + "5.bias"])
        return t + _gnorm(sd, k + "6", y, norm, training)

    def convT(pre, i, t):
        return F.conv_transpose2d(t, sd["%s.%d.weight" % (pre, i)], sd["%s.%d.bias" % (pre, i)], stride=2, padding=1, output_padding=1)

    pyramid = [x]
    for _ in range(n_local):
        pyramid.append(F.avg_pool2d(pyramid[-1], 3, stride=2, padding=1, count_include_pad=False))
    # global trunk on the coarsest level (GlobalGenerator.model[:-3])
    g = nr("model", 2, c7("model", 1, pyramid[-1]))
    i = 4
    for _ in range(n_down):
        g = nr("model", i + 1, F.conv2d(g, sd["model.%d.weight" % i], sd["model.%d.bias" % i], stride=2, padding=1))
        i += 3
    for _ in range(n_blocks_global):
        g = block("model", i, g)
        i += 1
    for _ in range(n_down):
        g = nr("model", i + 1, convT("model", i, g))
        i += 3
    # local enhancers, coarse to fine
    for n in range(1, n_local + 1):
        m1, m2 = "model%d_1" % n, "model%d_2" % n
        d = nr(m1, 2, c7(m1, 1, pyramid[n_local - n]))
        d = nr(m1, 5, F.conv2d(d, sd[m1 + ".4.weight"], sd[m1 + ".4.bias"], stride=2, padding=1))
        u = d + g
        for j in range(n_blocks_local):
            u = block(m2, j, u)
        j = n_blocks_local
        g = nr(m2, j + 1, convT(m2, j, u))
    return torch.tanh(c7("model%d_2" % n_local, n_blocks_local + 4, g))


def local_enhancer_param_shapes(input_nc=1, output_nc=5, ngf=32, n_down=3, n_blocks_global=9, n_blocks_local=3, n_local=1):
    """state_dict of LocalEnhancer (BatchNorm, every conv with bias)"""
    sh = {}

    def conv(key, shape, bias_n):
        sh[key + ".weight"] = shape
        sh[key + ".bias"] = (bias_n,)

    def bn(key, c):
        for nm, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,)), ("num_batches_tracked", ())):
            sh[key + "." + nm] = shp

    def block(key, c):
        for j in (1, 5):
            conv("%s.conv_block.%d" % (key, j), (c, c, 3, 3), c)
            bn("%s.conv_block.%d" % (key, j + 1), c)

    c = ngf * 2 ** n_local
    conv("model.1", (c, input_nc, 7, 7), c); bn("model.2", c)
    i = 4
    for _ in range(n_down):
        conv("model.%d" % i, (2 * c, c, 3, 3), 2 * c); bn("model.%d" % (i + 1), 2 * c)
        c *= 2
        i += 3
    for _ in range(n_blocks_global):
        block("model.%d" % i, c)
        i += 1
    for _ in range(n_down):
        conv("model.%d" % i, (c, c // 2, 3, 3), c // 2); bn("model.%d" % (i + 1), c // 2)
        c //= 2
        i += 3
    for n in range(1, n_local + 1):
        f = ngf * 2 ** (n_local - n)
        m1, m2 = "model%d_1" % n, "model%d_2" % n
        conv(m1 + ".1", (f, input_nc, 7, 7), f); bn(m1 + ".2", f)
        conv(m1 + ".4", (2 * f, f, 3, 3), 2 * f); bn(m1 + ".5", 2 * f)
        for j in range(n_blocks_local):
            block("%s.%d" % (m2, j), 2 * f)
        j = n_blocks_local
        conv("%s.%d" % (m2, j), (2 * f, f, 3, 3), f); bn("%s.%d" % (m2, j + 1), f)
        if n == n_local:
            conv("%s.%d" % (m2, j + 4), (output_nc, f, 7, 7), output_nc)
    return sh


def resnet_param_shapes(input_nc=9, output_nc=5, ngf=10, n_blocks=9, n_down=2, norm="instance", down="blur", up="blur", conv_bias=None,
                        use_dropout=False):
    """state_dict entries of ResnetGenerator / GlobalGenerator (the `filt` buffers of the blur modules are constants
    and not listed).  conv_bias None: the reference's rule use_bias = (norm == instance)."""
    if conv_bias is None:
        conv_bias = norm == "instance"
    sh = {}

    def conv(key, shape, bias=conv_bias):
        sh[key + ".weight"] = shape
        if bias:
            sh[key + ".bias"] = (shape[0],)

    def nrm(key, c):
        if norm == "batch":
            sh[key + ".weight"] = (c,)
            sh[key + ".bias"] = (c,)
            sh[key + ".running_mean"] = (c,)
            sh[key + ".running_var"] = (c,)
            sh[key + ".num_batches_tracked"] = ()

    for kind, i in resnet_layout(n_blocks, n_down, down, up):
        if kind == "conv7_in":
            conv("model.%d" % (i + 1), (ngf, input_nc, 7, 7))
            nrm("model.%d" % (i + 2), ngf)
            c = ngf
        elif kind == "conv3_down":
            conv("model.%d" % i, (2 * c, c, 3, 3))
            nrm("model.%d" % (i + 1), 2 * c)
            c *= 2
        elif kind == "block":
            for j in (1, 6 if use_dropout else 5):      # (a Dropout module behind the first ReLU shifts the second conv / norm)
                conv("model.%d.conv_block.%d" % (i, j), (c, c, 3, 3))
                nrm("model.%d.conv_block.%d" % (i, j + 1), c)
        elif kind == "up_conv3":
            if up == "blur":
                conv("model.%d" % (i + 1), (c // 2, c, 3, 3))
                nrm("model.%d" % (i + 2), c // 2)
            else:
                sh["model.%d.weight" % i] = (c, c // 2, 3, 3)       # ConvTranspose2d layout [Cin, Cout, 3, 3]
                if conv_bias:
                    sh["model.%d.bias" % i] = (c // 2,)
                nrm("model.%d" % (i + 1), c // 2)
            c //= 2
        else:
            conv("model.%d" % (i + 1), (output_nc, c, 7, 7), bias=True)
    return sh


# ----------------------------------------------------------------------------
# multiscale PatchGAN discriminator  (models/networks.py:1649-1750)
# ----------------------------------------------------------------------------

D_CONV_IDX = (0, 2, 5, 8, 11)
D_BN_IDX = {2: 3, 5: 6, 8: 9}
D_STRIDE = {0: 2, 2: 2, 5: 2, 8: 1, 11: 1}


def d_layout(n_layers=3):
    """Sequential indices of one NLayerDiscriminator (networks.py:1696-1737): conv 0 (stride 2) + LeakyReLU; n_layers - 1 blocks
    [conv stride 2, norm, LeakyReLU]; one block [conv stride 1, norm, LeakyReLU]; conv stride 1 -> 1 channel.
    Returns (conv indices, {conv: norm index}, {conv: stride}); n_layers = 3 gives D_CONV_IDX / D_BN_IDX / D_STRIDE."""
    conv = [0] + [2 + 3 * k for k in range(n_layers)] + [2 + 3 * n_layers]
    bn = {c: c + 1 for c in conv[1:-1]}
    stride = {c: (2 if j < n_layers else 1) for j, c in enumerate(conv)}
    return tuple(conv), bn, stride


def d_channels(input_nc, ndf, n_layers=3):
    """channel counts along one PatchGAN: nf doubles per block, capped at 512 (networks.py:1712-1727)"""
    ch = [input_nc, ndf]
    for _ in range(n_layers):
        ch.append(min(ch[-1] * 2, 512))
    return ch + [1]


def nlayer_forward(sd, prefix, x, training=True, update_stats=True, momentum=0.1, feats=None, n_layers=3, use_sigmoid=False):
    """One NLayerDiscriminator (BatchNorm2d affine + running stats); use_sigmoid: the trailing nn.Sigmoid the reference appends for
    gan_mode 'vanilla' (networks.py:1659, 1731-1732)."""
    conv_idx, bn_idx, stride = d_layout(n_layers)
    for ci in conv_idx:
        k = "%s.%d" % (prefix, ci)
        x = F.conv2d(x, sd[k + ".weight"], sd[k + ".bias"], stride=stride[ci], padding=2)
        if ci in bn_idx:
            b = "%s.%d" % (prefix, bn_idx[ci])
            if training and update_stats:
                x = F.batch_norm(x, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"],
                                 sd[b + ".bias"], True, momentum, 1e-5)
                sd[b + ".num_batches_tracked"] += 1
            elif training:
                x = F.batch_norm(x, None, None, sd[b + ".weight"], sd[b + ".bias"], True, momentum, 1e-5)
            else:
                x = F.batch_norm(x, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"],
                                 sd[b + ".bias"], False, momentum, 1e-5)
        if feats is not None:
            feats.append(x)
        if ci != conv_idx[-1]:
            x = F.leaky_relu(x, 0.2)
    return torch.sigmoid(x) if use_sigmoid else x


def msd_forward(sd, x, num_D=3, training=True, update_stats=True, n_layers=3, use_sigmoid=False):
    """MultiscaleDiscriminator.forward: layer{num_D-1} sees full resolution first;
    pyramid by AvgPool2d(3, 2, padding 1, count_include_pad=False).  Returns [[pred_s0],...]."""
    res = []
    for i in range(num_D):
        res.append([nlayer_forward(sd, "layer%d" % (num_D - 1 - i), x, training, update_stats, n_layers=n_layers, use_sigmoid=use_sigmoid)])
        if i != num_D - 1:
            x = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)
    return res


# ----------------------------------------------------------------------------
# GAN loss (models/networks.py:448-542)
# ----------------------------------------------------------------------------


def gan_loss_single(pred, target_is_real, mode="nonsaturating", real_label=1.0, fake_label=0.0):
    bs = pred.size(0)
    if mode == "lsgan":
        t = torch.full_like(pred, real_label if target_is_real else fake_label)
        return F.mse_loss(pred, t)
    if mode == "vanilla":
        t = torch.full_like(pred, real_label if target_is_real else fake_label)
        return F.binary_cross_entropy_with_logits(pred, t)
    if mode in ("wgan", "wgangp"):
        return -pred.mean() if target_is_real else pred.mean()
    if mode == "nonsaturating":
        return F.softplus(-pred if target_is_real else pred).view(bs, -1).mean(dim=1)
    if mode == "hinge":
        return F.relu(1.0 - pred if target_is_real else 1.0 + pred).view(bs, -1).mean(dim=1)
    raise NotImplementedError(mode)


def gan_loss(preds, target_is_real, mode="nonsaturating", real_label=1.0, fake_label=0.0):
    """Sum over scales of the per-scale loss (networks.py:536-540)."""
    if isinstance(preds[0], list):
        loss = 0
        for p in preds:
            loss = loss + gan_loss_single(p[-1], target_is_real, mode, real_label, fake_label)
        return loss
    return gan_loss_single(preds[-1], target_is_real, mode, real_label, fake_label)


# ----------------------------------------------------------------------------
# PatchNCE (models/patchnce.py:13-55, networks.py:585-594 Normalize)
# ----------------------------------------------------------------------------


def l2_normalize(x):
    norm = x.pow(2).sum(1, keepdim=True).pow(0.5)
    return x.div(norm + 1e-7)


def patchnce_loss(feat_q, feat_k, batch_size, nce_T=0.07, all_negatives_from_minibatch=False):
    n, dim = feat_q.shape
    feat_k = feat_k.detach()
    l_pos = torch.bmm(feat_q.view(n, 1, -1), feat_k.view(n, -1, 1)).view(n, 1)
    b = 1 if all_negatives_from_minibatch else batch_size
    q = feat_q.view(b, -1, dim)
    k = feat_k.view(b, -1, dim)
    npatches = q.size(1)
    l_neg = torch.bmm(q, k.transpose(2, 1))
    eye = torch.eye(npatches, dtype=torch.bool, device=q.device)[None]
    l_neg = l_neg.masked_fill(eye, -10.0).view(-1, npatches)
    out = torch.cat((l_pos, l_neg), 1) / nce_T
    return F.cross_entropy(out, torch.zeros(out.size(0), dtype=torch.long, device=q.device), reduction="none")


def patch_sample_f(feats, patch_ids, mlps=None):
    """PatchSampleF.forward with given ids (networks.py:687-719): feats list of [B, C, H, W]; mlps: None or a list of
    (w0 [nc, C], b0, w2 [nc, nc], b2) per feature map.  Returns the list of L2-normalised [B * P, C | nc] rows.
    patch_ids None = num_patches 0: the whole map stays [B, HW, C] through the MLP, Normalize(2) then divides by the norm over dim 1
    -- the positions -- and the result is reshaped to [B, C | nc, H, W] (:704-706, 713-717)."""
    outs = []
    for i, feat in enumerate(feats):
        if patch_ids is None:
            b, _, h, w = feat.shape
            x = feat.permute(0, 2, 3, 1).flatten(1, 2)
            if mlps is not None:
                w0, b0, w2, b2 = mlps[i]
                x = F.linear(F.relu(F.linear(x, w0, b0)), w2, b2)
            x = l2_normalize(x)
            outs.append(x.permute(0, 2, 1).reshape(b, x.shape[-1], h, w))
            continue
        x = feat.permute(0, 2, 3, 1).flatten(1, 2)[:, torch.as_tensor(patch_ids[i], dtype=torch.long), :].flatten(0, 1)
        if mlps is not None:
            w0, b0, w2, b2 = mlps[i]
            x = F.linear(F.relu(F.linear(x, w0, b0)), w2, b2)
        outs.append(l2_normalize(x))
    return outs


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam semantics used at sinskitG_model.py:589-599)
# ----------------------------------------------------------------------------


def adam_update(p, g, m, v, step, lr, beta1, beta2, eps=1e-8):
    """In-place single-tensor Adam, torch.optim.Adam defaults (no amsgrad, no weight decay)."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------
# initialisation (models/networks.py:191-231) -- distribution-level restatement;
# parity tests load reference-initialised weights instead of matching RNG streams.
# ----------------------------------------------------------------------------


def g_param_shapes(input_nc=9, ngf=10, num_downs=8, num_layer_separate=4, style_nc=0, num_layer_style_code=1, style_map_nc=0,
                   style_dim=512, style_bn=False):
    """Ordered {key: shape} of CustomUnetGenerator (instance norm => conv bias present).  style_nc: extra input channels of the styled
    up layers (concat mode); style_map_nc > 0: style_code_mapping<j> layers emitting that many channels on the (1536 >> (num_downs - j))^2
    map (project mapping, networks.py:1444-1465), with BatchNorm1d parameters when style_bn (batch_size > 1)."""
    ch = [ngf * min(2 ** i, 8) for i in range(num_downs)]  # out channels of down_i
    shapes = {}
    for j in range(num_layer_style_code if style_map_nc else 0):
        side = 1536 // (2 ** (num_downs - j))
        shapes["style_code_mapping%d.0.weight" % j] = (side * side * style_map_nc, style_dim)
        if style_bn:
            shapes["style_code_mapping%d.1.weight" % j] = (side * side * style_map_nc,)
            shapes["style_code_mapping%d.1.bias" % j] = (side * side * style_map_nc,)
    shapes["down0.model.0.weight"] = (ch[0], input_nc, 4, 4)
    shapes["down0.model.0.bias"] = (ch[0],)

    def add_up(i, suffix=""):
        outer = 3 if i == 0 else ch[i - 1]
        if suffix and i == 0:
            outer = 2
        inner = ch[i] * (1 if i in (0, num_downs - 1) else 2)
        if style_nc and i >= num_downs - num_layer_style_code:
            inner += style_nc
        shapes["up%d%s.model.1.weight" % (i, suffix)] = (inner, outer, 4, 4)
        shapes["up%d%s.model.1.bias" % (i, suffix)] = (outer,)

    add_up(0)
    if num_layer_separate >= 1:
        add_up(0, "_T")
    for i in range(1, num_downs):
        shapes["down%d.model.1.weight" % i] = (ch[i], ch[i - 1], 4, 4)
        shapes["down%d.model.1.bias" % i] = (ch[i],)
        add_up(i)
        if num_layer_separate >= i + 1:
            add_up(i, "_T")
    return shapes


def d_if_param_shapes(input_nc, ndf=64, num_D=2, n_layers=3):
    """MultiscaleDiscriminator with getIntermFeat=True (pix2pixHD default `getIntermFeat_D`, networks.py:1661-1667):
    the same layers as d_param_shapes under the keys scale{i}_layer{j}.{0 conv | 1 norm}.*"""
    ch = d_channels(input_nc, ndf, n_layers)
    sh = {}
    for i in range(num_D):
        for j in range(n_layers + 2):
            k = "scale%d_layer%d." % (i, j)
            sh[k + "0.weight"] = (ch[j + 1], ch[j], 4, 4)
            sh[k + "0.bias"] = (ch[j + 1],)
            if 1 <= j <= n_layers:
                for nm, shape in (("weight", (ch[j + 1],)), ("bias", (ch[j + 1],)), ("running_mean", (ch[j + 1],)),
                                  ("running_var", (ch[j + 1],)), ("num_batches_tracked", ())):
                    sh[k + "1." + nm] = shape
    return sh


def d_if_to_plain(sd, n_layers=3):
    """scale{i}_layer{j}.{0|1}.x  ->  layer{i}.{conv / norm index of the fused Sequential}.x  (same network)"""
    conv_idx, bn_of_conv, _ = d_layout(n_layers)
    out = {}
    for k, v in sd.items():
        head, sub, name = k.split(".", 2)
        i, j = int(head[5:head.index("_")]), int(head[head.index("layer") + 5:])
        out["layer%d.%d.%s" % (i, conv_idx[j] if sub == "0" else bn_of_conv[conv_idx[j]], name)] = v
    return out


def d_param_shapes(input_nc, ndf=8, num_D=3, n_layers=3):
    """Ordered {key: shape} of MultiscaleDiscriminator (BatchNorm2d), incl. buffers."""
    chans = d_channels(input_nc, ndf, n_layers)
    conv_idx, bn_idx, _ = d_layout(n_layers)
    shapes = {}
    for d in range(num_D):
        for j, ci in enumerate(conv_idx):
            shapes["layer%d.%d.weight" % (d, ci)] = (chans[j + 1], chans[j], 4, 4)
            shapes["layer%d.%d.bias" % (d, ci)] = (chans[j + 1],)
            if ci in bn_idx:
                b = "layer%d.%d" % (d, bn_idx[ci])
                c = chans[j + 1]
                shapes[b + ".weight"] = (c,)
                shapes[b + ".bias"] = (c,)
                shapes[b + ".running_mean"] = (c,)
                shapes[b + ".running_var"] = (c,)
                shapes[b + ".num_batches_tracked"] = ()
    return shapes


# ---- SIFID chain (models/sifid.py:38-101, 156-233; models/inception.py:57-67, 113-147; models/model_utils.py:481-488, 541-555) -------
def inception_block0(x, sd):
    """the reference's InceptionV3(output_blocks=[0]) forward AFTER its `2 * x - 1` on a state dict with the wrapper's keys
    (blocks.0.k.conv.weight / bn.*): three torchvision BasicConv2d = Conv2d(3x3, bias=False) -> BatchNorm2d(eval, eps 0.001) -> ReLU with
    (stride, padding) = (2, 0), (1, 0), (1, 1).  torchvision is absent from this image: this restates its published layer definition
    (parity unpinned for that third-party architecture; the statistics / Frechet arithmetic behind it IS pinned, tests/golden/metrics.npz)"""
    import torch.nn.functional as F
    for k, (stride, pad) in enumerate(((2, 0), (1, 0), (1, 1))):
        pre = "blocks.0.%d." % k
        x = F.conv2d(x, sd[pre + "conv.weight"], None, stride=stride, padding=pad)
        x = F.batch_norm(x, sd[pre + "bn.running_mean"], sd[pre + "bn.running_var"], sd[pre + "bn.weight"], sd[pre + "bn.bias"], False, 0.0, 0.001)
        x = F.relu(x)
    return x


def sifid_pairs(a, b, sd):
    """calculate_sifid_given_arrays (sifid.py:205-233) on network inputs a, b [N,3,H,W]: per image, activations = positions x 64
    (get_activations :96), mean / np.cov, Frechet distance"""
    out = []
    for i in range(a.shape[0]):
        fa, fb = inception_block0(a[i:i + 1], sd)[0], inception_block0(b[i:i + 1], sd)[0]
        out.append(frechet_distance(fa.reshape(fa.shape[0], -1), fb.reshape(fb.shape[0], -1)))
    return out


def sifid_images(real_I, fake_I, sd):
    """I_SIFID (model_utils.py:481-488)"""
    lo, hi = real_I.min(), real_I.max()
    r = (real_I - lo) / (hi - lo)
    f = torch.clamp((fake_I - lo) / (hi - lo), 0, 1)
    v = sifid_pairs(2 * r - 1, 2 * f - 1, sd)
    return v[0] if len(v) == 1 else float(np.mean(v))


def sifid_tactile(real_T, fake_T, sd, size=299):
    """T_SIFID (model_utils.py:519, 541-555); convert2tensor's (x + 1) / 2 and the network's 2x - 1 cancel"""
    import torch.nn.functional as F
    fake_T = torch.clamp(fake_T, 0, 1)
    r, f = F.interpolate(real_T, (size, size)), F.interpolate(fake_T, (size, size))
    vals = []
    for c in (0, 1):
        vals.append(np.array(sifid_pairs(r[:, c:c + 1].repeat(1, 3, 1, 1), f[:, c:c + 1].repeat(1, 3, 1, 1), sd)))
    return float(np.mean((vals[0] + vals[1]) / 2))
