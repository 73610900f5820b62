"""Forward / input-gradient schedules of the frozen VGG feature stacks and the perceptual terms built on them.

  lpips_term      lpips.LPIPS(net="vgg")(fake, real) reduced the way the reference reduces it (models/sinskitG_model.py:1711: mean over
                  samples; :1648-1657: per-image sum over the NT patches), value into a fixed-point loss slot, gradient w.r.t. `fake`
  vgg_feature_l1  VGGLoss (models/networks.py:2021-2033): sum_i w_i L1(relu_i(x), relu_i(y)), gradient w.r.t. x

Network layout: torchvision VGG features = 3x3 convolutions (padding 1) + ReLU, MaxPool2d(2, 2) between blocks.  Round 4: a
convolution stores relu(z) straight into the next convolution's pre-padded operand (zero one-pixel border), so between two convolutions
of a block there is NO pass at all, and behind a block only the pooling pass; taps, pooling and the backward's ReLU mask read that padded
tensor (relu(z) > 0 <=> z > 0).  The backward mirrors it: an input adjoint's epilogue applies the ReLU mask of the layer in front and
adds that layer's tap gradient.  Layers whose maps are too small for a tiled launch keep the round-3 form: the RAW output
z, relu applied by the reader, padding as one fused pass (pad_affine with ReLU / vts_relu_mask_pad).  Every convolution runs on the
GEMM-class MFMA kernels (ops.conv3x3_wide; the 3 -> 64 stem as one 8-channel chunk) with weights packed ONCE (the stacks are frozen).
The backward is the input adjoint only (no weight gradients): the same GEMM kernel on flipped / transposed packing,
vts_maxpool2_relu_bwd behind a block, and the stem's 64 -> 3 adjoint as 4x4 tap blocks on the generator's kernel (ops.convk_bwd_data).
"""
import os
from . import tune

import torch

from . import lib as L
from . import ops

RELU = L.ACT_RELU
BATCH_PAIR = tune.get("VTS_LPIPS_BATCH_PAIR", "1") != "0"     # fake | real images as one batch through the VGG stack (0: two forwards, round 3)
PADDED = tune.get("VTS_VGG_PADDED", "1") != "0"    # 0: every activation as a dense raw output + separate ReLU / padding passes (round 3)


def _packed(net, k, mode):
    key = (k, mode)
    buf = net._packed.get(key)
    if buf is None:
        w = net.convs[k].weight
        buf = ops.w3x3_pack(w, mode, tag="frozen%d" % id(net))
        net._packed[key] = buf
    return buf


def _wino(net, k, mode):
    """transform-domain weights of convolution k for the Winograd kernel (frozen stack: transformed once)"""
    key = (k, mode, "wino")
    buf = net._packed.get(key)
    if buf is None:
        buf = net._packed[key] = ops.w3x3_wino_pack(net.convs[k].weight, mode, tag="frozen%d" % id(net))
    return buf


def _layout(net):
    """[(conv index, pooled_before?)] in forward order"""
    out, k, pool = [], 0, False
    for v in net.cfg:
        if v == "M":
            pool = True
        else:
            out.append((k, pool))
            pool = False
            k += 1
    return out


def vgg_forward(net, x, keep_all=True, last_tap_only_needed=True):
    """x [N, 3, H, W] (already in the network's input space).  Returns {conv index: (activation, zpad)}: every convolution when keep_all
    (a backward follows), else the tapped ones only.  Stops after the deepest tap.
      zpad 1  relu(z) in the interior of a zero-bordered [N, C, H + 2, W + 2] tensor -- written by the convolution itself
              (ops.conv3x3_wide_relu_pad) and read as it is by the next convolution: no pass in between
      zpad 0  the raw output z, dense (maps too small for a tiled launch; VTS_VGG_PADDED=0)"""
    n, _, h, w = x.shape
    dev = x.device
    zs = {}
    prev = None
    last = max(net.taps)
    for k, pooled in _layout(net):
        if k > last:
            break
        conv = net.convs[k]
        co, ci = conv.weight.shape[:2]
        if k == 0:
            p = ops.pad_affine(x, (1, 1, 1, 1), 0)      # (3 channels: the stem runs on the GEMM-class kernel too, one 8-channel chunk)
        else:
            pt, pz = prev
            if pooled:
                p = ops.maxpool2_relu_pad(pt, 1, zpad=pz)
            else:
                p = pt if pz else ops.pad_affine(pt, (1, 1, 1, 1), 0, act=RELU)
        hh, ww = p.shape[2] - 2, p.shape[3] - 2
        cur = None
        if ops.conv3x3_wino_ok(n, ci, co, hh, ww):       # Winograd F(2x2, 3x3): the layers with >= 32 input channels on maps that fill the chip
            out = torch.empty(n, co, hh + 2 * PADDED, ww + 2 * PADDED, dtype=torch.float32, device=dev)
            ops.conv3x3_wino(p, _wino(net, k, "conv_fwd"), conv.bias, out, ep_mode=1 if PADDED else 0)
            cur = (out, 1 if PADDED else 0)
            del out
        elif PADDED:
            out = torch.empty(n, co, hh + 2, ww + 2, dtype=torch.float32, device=dev)
            if ops.conv3x3_wide_relu_pad(p, _packed(net, k, "conv_fwd"), conv.bias, out):
                cur = (out, 1)
            del out
        if cur is None:
            z = torch.empty(n, co, hh, ww, dtype=torch.float32, device=dev)
            ops.conv3x3_wide(p, _packed(net, k, "conv_fwd"), conv.bias, z)
            cur = (z, 0)
        if keep_all or k in net.taps:
            zs[k] = cur
        prev = cur
    return zs


def tap_gradient_buffer(feat):
    """an uninitialised tap gradient in the activation's layout (zero border where it has one)"""
    t, zp = feat
    g = torch.empty_like(t)
    if zp:
        ops.zero_border(g, zp)
    return g


def vgg_backward(net, zs, tap_grads, x_shape, masked_taps=False):
    """gradient w.r.t. the network input given {tap conv index: gradient w.r.t. relu(z_tap), in that activation's layout}; zs from
    vgg_forward(keep_all=True).  masked_taps: the tap gradients already carry the ReLU mask (ops.lpips_layer writes them so).
    G is the gradient w.r.t. z_k in the input adjoint's pre-padded layout; where the layer in front is kept padded its ReLU mask and tap
    gradient ride in the adjoint's epilogue (ops.conv3x3_wide_mask_pad) or in the pooling adjoint (ops.maxpool2_relu_bwd)."""
    lay = [e for e in _layout(net) if e[0] <= max(net.taps)]
    n = x_shape[0]
    top = max(k for k in tap_grads)
    G = None
    for idx in range(len(lay) - 1, 0, -1):
        k, pooled = lay[idx]
        if k > top:
            continue
        if k == top:
            zt, zp = zs[k]
            G = tap_grads[k] if (masked_taps and zp == 1) else ops.relu_mask_pad(None, tap_grads[k], zt, pad=1, zpad=zp)
        conv = net.convs[k]
        ci = conv.weight.shape[1]
        kf = lay[idx - 1][0]
        ft, fp = zs[kf]
        T = tap_grads.get(kf)
        hh, ww = G.shape[2] - 2, G.shape[3] - 2
        co = conv.weight.shape[0]
        wino = ops.conv3x3_wino_ok(n, co, ci, hh, ww)        # the input adjoint: a convolution from the layer's output to its input channels
        if PADDED and not pooled and fp == 1:
            out = torch.empty_like(ft)
            if wino:
                ops.conv3x3_wino(G, _wino(net, k, "conv_adj"), None, out, ep_mode=2, add=T, mask=ft)
                G = out
                continue
            if ops.conv3x3_wide_mask_pad(G, _packed(net, k, "conv_adj"), out, ft, add=T):
                G = out
                continue
            del out
        gin = torch.empty(n, ci, hh, ww, dtype=torch.float32, device=G.device)
        if wino:
            ops.conv3x3_wino(G, _wino(net, k, "conv_adj"), None, gin)
        else:
            ops.conv3x3_wide(G, _packed(net, k, "conv_adj"), None, gin)
        G = ops.maxpool2_relu_bwd(gin, ft, zpad=fp, g2=T, pad=1) if pooled else ops.relu_mask_pad(gin, T, ft, pad=1, zpad=fp)
    if G is None:
        raise RuntimeError("vgg_backward: no tap gradient given")
    # the stem's input adjoint (64 -> 3) on the generator's kernel: G carries a zero border, so the padding-1 adjoint of the dense
    # gradient is the padding-2 adjoint of G (din[y] = sum_k G[y + 2 - k] w[k])
    dx = torch.empty(x_shape, dtype=torch.float32, device=G.device)
    ops.convk_bwd_data(G, net.convs[0].weight, dx, pad=2)
    return dx


def alex_forward(net, x):
    """AlexNet feature stack of models/perceptual.py:LpipsAlex on the HIP kernels; x [N, 3, H, W] in the network's input space.
    Returns {k: raw output z of convolution k} (the taps read relu(z) on load, like the VGG taps).
      conv0  11 x 11, stride 4, padding 2: space-to-depth by 4 of the padded input (vts_s2d4_pad) + a valid 3 x 3 convolution over the 48
             phase channels on the GEMM-class kernel (weights rearranged once: LpipsAlex.stem_weight)
      conv1  5 x 5, padding 2 on relu -> MaxPool2d(3, 2): four 4 x 4 tap blocks on the generator's kernel (ops.convk)
      conv2-4  3 x 3, padding 1: GEMM-class kernel on the pre-padded relu'd input"""
    n, _, h, w = x.shape
    dev = x.device
    oh, ow = (h + 4 - 11) // 4 + 1, (w + 4 - 11) // 4 + 1
    zs = {}
    stem = net._packed.get("stem")
    if stem is None:
        net._stem_w = net.stem_weight()
        stem = net._packed["stem"] = ops.w3x3_pack(net._stem_w, "conv_fwd", tag="alexstem%d" % id(net))
    p = ops.s2d4_pad(x.contiguous(), 2, oh + 2, ow + 2)
    z = torch.empty(n, 64, oh, ow, dtype=torch.float32, device=dev)
    ops.conv3x3_wide(p, stem, net.convs[0].bias, z)
    zs[0] = z
    q = ops.maxpool3s2_relu_pad(z, 0)
    z = torch.empty(n, 192, q.shape[2], q.shape[3], dtype=torch.float32, device=dev)
    ops.convk(q, net.convs[1].weight, z, bias=net.convs[1].bias, pad=2)
    zs[1] = z
    prev = z
    for k in (2, 3, 4):
        p = ops.maxpool3s2_relu_pad(prev, 1) if k == 2 else ops.pad_affine(prev, (1, 1, 1, 1), 0, act=RELU)
        z = torch.empty(n, net.convs[k].weight.shape[0], p.shape[2] - 2, p.shape[3] - 2, dtype=torch.float32, device=dev)
        ops.conv3x3_wide(p, _packed(net, k, "conv_fwd"), net.convs[k].bias, z)
        zs[k] = z
        prev = z
    return zs


def lpips_alex_value(net, a, b, coeff, loss_slot, channels=None):
    """loss_slot += coeff * sum_n lpips.LPIPS(net="alex")(a_n, b_n) (the metric is symmetric in its arguments; no gradient)"""
    cx = a.shape[1] if channels is None else channels
    z1 = alex_forward(net, ops.lpips_input(b, net.shift, net.scale, channels=cx))
    z0 = alex_forward(net, ops.lpips_input(a, net.shift, net.scale, channels=cx))
    for i in range(5):
        ops.lpips_layer(z0[i], z1[i], net.lins[i].view(-1), coeff, loss_slot, dz0=None, grad_coeff=coeff)


def lpips_term(net, fake, real, coeff, loss_slot, grad_into=None, grad_accumulate=False, channels=None, nstride_fake=None, nstride_real=None,
               grad_nstride=None):
    """loss_slot += coeff * sum_n LPIPS(fake_n, real_n); grad_into (+)= coeff * d(.)/d fake when given.
    fake / real: [N, 3, H, W], or 1-channel VIEWS (channels=1, nstride_* = batch stride of the tensor they are a channel of):
    the ScalingLayer broadcasts the single channel to three, as lpips does for the tactile gx / gy images."""
    cx = fake.shape[1] if channels is None else channels
    want = grad_into is not None
    n = fake.shape[0]
    if BATCH_PAIR and want:
        # fake and real images through the stack as ONE batch of 2 N (rows [0, N) fake, [N, 2 N) real): half the launches, twice the
        # workgroups per launch on the deep layers' small maps; the backward runs on the fake rows (contiguous batch slices)
        y = torch.empty(2 * n, 3, fake.shape[2], fake.shape[3], dtype=torch.float32, device=fake.device)
        y0 = ops.lpips_input(fake, net.shift, net.scale, nstride=nstride_fake, channels=cx, out=y[:n])
        ops.lpips_input(real, net.shift, net.scale, nstride=nstride_real, channels=cx, out=y[n:])
        zz = vgg_forward(net, y, keep_all=True)
        del y
        z0 = {k: (t[:n], zp) for k, (t, zp) in zz.items()}
        z1 = {k: (zz[k][0][n:], zz[k][1]) for k in net.taps}
        del zz
    else:
        y1 = ops.lpips_input(real, net.shift, net.scale, nstride=nstride_real, channels=cx)
        z1 = vgg_forward(net, y1, keep_all=False)
        del y1
        y0 = ops.lpips_input(fake, net.shift, net.scale, nstride=nstride_fake, channels=cx)
        z0 = vgg_forward(net, y0, keep_all=want)
    tap_grads = {}
    for i, k in enumerate(net.taps):
        assert z0[k][1] == z1[k][1]
        dz = tap_gradient_buffer(z0[k]) if want else None
        ops.lpips_layer(z0[k][0], z1[k][0], net.lins[i].view(-1), coeff, loss_slot, dz0=dz, grad_coeff=coeff, zpad=z0[k][1])
        if want:
            tap_grads[k] = dz
    if not want:
        return None
    del z1
    gy = vgg_backward(net, z0, tap_grads, tuple(y0.shape), masked_taps=True)
    return ops.lpips_input_bwd(gy, net.scale, grad_into, cx, accumulate=grad_accumulate, nstride=grad_nstride)


def vgg_feature_l1(net, x, y, coeff, loss_slot, want_grad=True):
    """VGGLoss: loss_slot += coeff * sum_i w_i mean|relu_i(x) - relu_i(y)|; returns d(.)/dx (or None).  x, y [N, 3, H, W]."""
    n = x.shape[0]
    if BATCH_PAIR and want_grad and x.shape == y.shape:
        zz = vgg_forward(net, torch.cat([x, y], 0), keep_all=True)        # rows [0, N): x (the backward's side), [N, 2 N): y
        zx = {k: (t[:n], zp) for k, (t, zp) in zz.items()}
        zy = {k: (zz[k][0][n:], zz[k][1]) for k in net.taps}
        del zz
    else:
        zy = vgg_forward(net, y, keep_all=False)
        zx = vgg_forward(net, x, keep_all=want_grad)
    tap_grads = {}
    for wi, k in zip(net.weights, net.taps):
        (tx, zp), (ty, zq) = zx[k], zy[k]
        assert zp == zq
        # (padded activations: the two zero borders agree, so the flat pass over the whole buffers adds nothing there and leaves a zero
        # border in the gradient; the mean is over the interior)
        count = tx.shape[0] * tx.shape[1] * (tx.shape[2] - 2 * zp) * (tx.shape[3] - 2 * zp)
        g = torch.empty_like(tx) if want_grad else None
        ops.l1_relu(tx, ty, coeff * wi / count, loss_slot, grad=g)
        if want_grad:
            tap_grads[k] = g
    if not want_grad:
        return None
    return vgg_backward(net, zx, tap_grads, tuple(x.shape))
