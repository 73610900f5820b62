"""Explicit forward / backward schedules of the two network families over libvts_hip.so.

There is no autograd on the hot path: the U-Net generator and the multiscale PatchGAN
discriminator are fixed graphs, so their backward passes are written out as kernel
sequences.  Activations travel as `ops.Act` (raw tensor + per-(n,c) scale/shift): a
normalisation layer is a statistics pass only, and every consumer (next conv, skip
concat, derivative mask, weight gradient) normalises on load inside the conv kernels.

Reference graphs reproduced here:
  CustomUnetGenerator.forward     /root/reference/models/networks.py:1576-1645
  Down / Up blocks                /root/reference/thirdparty/unet/unet_parts_custom.py:9-79
  MultiscaleDiscriminator.forward /root/reference/models/networks.py:1682-1693
  NLayerDiscriminator             /root/reference/models/networks.py:1696-1750
(their backward is PyTorch autograd in the reference).
"""
import ctypes as C
import os
from . import tune

import torch

from . import lib as L
from . import ops
from .ops import Act

LRELU, RELU, TANH = L.ACT_LRELU, L.ACT_RELU, L.ACT_TANH


def _empty(n, c, h, w, dev):
    return torch.empty(n, c, h, w, dtype=torch.float32, device=dev)


def _as_act(x):
    return x if isinstance(x, Act) else Act(x)


# =====================================================================================
# generator
# =====================================================================================

class UnetCtx:
    __slots__ = ("x", "feats", "ups", "g_out", "style", "style_ctx", "adain_in", "dstyle")


def dropout_layers(G):
    """the Up blocks that end in Dropout(0.5) when the generator was built with use_dropout (reference networks.py:1508-1519), in the
    order the forward visits them"""
    if not getattr(G, "use_dropout", False):
        return []
    return list(range(G.num_downs - 2, G.num_downs // 2 - 1, -1))


def unet_forward(G, x, style_code=None, keep=True, style_tiles=None, dropout_masks=None):
    """x: [N, input_nc, H, W] tensor / Act, or a pair (x0, x1) that is concatenated on load
    (sketch ++ positional grid).  Returns (g_out [N,5,H,W] post-tanh, ctx).
    style_tiles: {layer index: [N, style_dim, h, w]} the tiled style code when the caller holds it (it depends on the batch only: the
    model tiles it once per set_input instead of once per step)
    dropout_masks: {layer index: keep mask [N, C, h, w] of 0 / 1} for the Up blocks of dropout_layers(G) in train() mode (drawn here with
    torch.bernoulli when absent; tests pass the reference's draws)"""
    if isinstance(x, (tuple, list)):
        x, x_extra = _as_act(x[0]), _as_act(x[1])
    else:
        x, x_extra = _as_act(x), None
    xd = x.data
    n, _, h, w = xd.shape
    dev = xd.device
    nd, ch = G.num_downs, G.channels
    if h % (1 << nd) or w % (1 << nd):
        raise ValueError("unet256_custom needs H, W divisible by %d, got %dx%d" % (1 << nd, h, w))
    feats = []
    a = x
    for i in range(nd):
        blk = getattr(G, "down%d" % i).conv
        cin = blk.weight.shape[1]
        hh, ww = h >> (i + 1), w >> (i + 1)
        out = _empty(n, ch[i], hh, ww, dev)
        normed = 0 < i < nd - 1
        r = ops.conv4x4(a, blk.weight, cin * 16, 16, ch[i], out, in1=x_extra if i == 0 else None, bias=blk.bias, stride=2, pad=1,
                        act_in=LRELU if i else 0, instance_norm=normed)
        a = r if normed else Act(out)
        feats.append(a)

    style = None
    if style_code is not None:
        if not G.use_style:
            raise ValueError("generator was built without style code support")
        style = style_code.to(torch.float32)
    g_out = _empty(n, 5, h, w, dev)
    ups = {}
    extras = {}
    style_ctx = {}
    for i in range(nd):
        if style is not None and i >= nd - G.num_layer_style_code:
            if i not in (0, nd - 1):
                raise NotImplementedError("style code on a skip-connected layer needs a third concat source")
            hh, ww = h >> (i + 1), w >> (i + 1)
            if G.style_mapping == "tile":
                t = style_tiles.get(i) if style_tiles else None
                extras[i] = Act(t if t is not None else style[:, :, None, None].expand(-1, -1, hh, ww).contiguous())
            else:
                smap, style_ctx[i] = _style_map_forward(G, nd - 1 - i, style, hh, ww)
                if G.style_mode == "concat":
                    extras[i] = Act(smap)
                else:
                    style_ctx[i] += (smap,)

    drop_layers = dropout_layers(G) if G.training else []

    def up(i, name, inp):
        hh, ww = h >> (i + 1), w >> (i + 1)
        skip = None if i in (0, nd - 1) else feats[i]
        extra = extras.get(i)
        blk = getattr(G, name).conv
        outer = blk.weight.shape[1]
        if i == 0:
            c0 = 0 if name == "up0" else 3
            out = g_out[:, c0:c0 + outer]
        else:
            out = _empty(n, outer, hh * 2, ww * 2, dev)
        r = ops.conv4x4(inp, blk.weight, 16, outer * 16, outer, out, in1=skip if skip is not None else extra, bias=blk.bias,
                        stride=2, pad=1, transposed=True, act_in=RELU, act_out=TANH if i == 0 else 0, instance_norm=i != 0)
        res = Act(out) if i == 0 else r
        drop = None
        if i in drop_layers:
            # Dropout(0.5) behind the InstanceNorm: the normalised map is materialised, multiplied by keep * 2 (tiny maps: 80 channels at
            # <= 1/16 of the resolution), and the next block reads it as a plain tensor; the backward multiplies by the same map
            keep2 = dropout_masks.get(i) if dropout_masks is not None else None
            if keep2 is None:
                keep2 = torch.bernoulli(torch.full(out.shape, 0.5, device=dev))
            keep2 = (keep2.to(device=dev, dtype=torch.float32) * 2.0).contiguous()
            y = ops.pad_affine(r, (0, 0, 0, 0), 0)
            yd = torch.empty_like(y)
            ops.mask_mul(y.view(-1, 1, y.shape[2], y.shape[3]), keep2.view(-1, 1, y.shape[2], y.shape[3]), out=yd.view(-1, 1, y.shape[2], y.shape[3]))
            drop = (r, keep2)
            res = Act(yd)
        ups[name] = (inp, res, extra, drop)
        return res

    nls = G.num_layer_separate
    xm = feats[nd - 1]
    adain_in = {}
    for i in range(nd - 1, nls - 1, -1):      # shared trunk
        if i in style_ctx and G.style_mode == "adain":     # x = adaptive_instance_normalization(x, style map) (networks.py:1624-1630)
            if i != nd - 1:
                raise NotImplementedError("adain style conditioning is built for the innermost layer (num_layer_style_code 1)")
            x_raw = ops.pad_affine(xm, (0, 0, 0, 0), 0) if (xm.scale is not None) else xm.data
            adain_in[i] = x_raw
            xm = Act(ops.adain(x_raw, style_ctx[i][-1]))
        xm = up(i, "up%d" % i, xm)

    def chain(lane):                          # the visual (lane 0) and tactile (lane 1) decoders are independent chains
        x = xm
        for i in range(nls - 1, -1, -1):
            x = up(i, "up%d%s" % (i, "_T" if lane else ""), x)

    if nls > 0:
        _run_lanes(2, chain)
    if not keep:
        return g_out, None
    ctx = UnetCtx()
    ctx.x, ctx.feats, ctx.ups, ctx.g_out, ctx.style = (x, x_extra), feats, ups, g_out, style
    ctx.style_ctx, ctx.adain_in, ctx.dstyle = style_ctx, adain_in, None
    return g_out, ctx


def unet_desc(G, x, g_out, style_tile=None, side_stream=None):
    """vts_unet_desc of generator G on the input x (a tensor / Act or a pair that is concatenated on load) writing g_out [N, 5, H, W]:
    the network-level C entry vts_unet_forward (include/vts.h) runs the inference forward from it.  style_tile: [N, style_dim, h, w]
    tiled style code of the innermost block (style_code_mode concat + mapping tile), or None."""
    import ctypes as C

    if isinstance(x, (tuple, list)):
        x0, x1 = _as_act(x[0]), _as_act(x[1])
    else:
        x0, x1 = _as_act(x), None
    n, _, h, w = x0.data.shape
    nd = G.num_downs
    if nd > L.UNET_MAX_DOWNS:
        raise ValueError("vts_unet_forward takes at most %d down blocks" % L.UNET_MAX_DOWNS)
    d = L.UnetDesc()
    d.N, d.H, d.W, d.num_downs, d.num_layer_separate = n, h, w, nd, G.num_layer_separate
    d.in0 = x0.operand()
    d.in1 = x1.operand() if x1 is not None else L.Operand(None, None, None, 0, 0)
    for i in range(nd):
        dn, up = getattr(G, "down%d" % i).conv, getattr(G, "up%d" % i).conv
        d.channels[i] = dn.weight.shape[0]
        d.down_w[i], d.down_b[i] = dn.weight.data_ptr(), L.ptr(dn.bias)
        d.up_w[i], d.up_b[i], d.up_cout[i] = up.weight.data_ptr(), L.ptr(up.bias), up.weight.shape[1]
        if i < G.num_layer_separate:
            ut = getattr(G, "up%d_T" % i).conv
            d.upT_w[i], d.upT_b[i], d.upT_cout[i] = ut.weight.data_ptr(), L.ptr(ut.bias), ut.weight.shape[1]
    d.style = L.operand(style_tile) if style_tile is not None else L.Operand(None, None, None, 0, 0)
    d.out = g_out.data_ptr()
    d.side_stream = side_stream.cuda_stream if side_stream is not None else None     # the tactile branch's lane
    return d


UNET_C = tune.get("VTS_UNET_C", "1") != "0"     # inference forward through the network-level C entry (0: the Python schedule)


def unet_c_ok(G, style_code):
    """can vts_unet_forward run this generator's inference forward?  (plain U-Net, or the style code tiled into the innermost block)"""
    if not UNET_C or G.num_downs > L.UNET_MAX_DOWNS:
        return False
    if G.num_layer_separate >= G.num_downs:     # no shared decoder trunk (every up block duplicated): the C entry's check() refuses it
        return False
    if G.training and dropout_layers(G):     # a forward outside eval() keeps the Dropout of the Up blocks active
        return False
    if style_code is None:
        return True
    return bool(G.use_style) and G.style_mapping == "tile" and G.style_mode == "concat" and G.num_layer_style_code == 1


def unet_forward_infer(G, x, style_code=None, style_tiles=None):
    """inference forward (nothing kept for a backward) -> g_out: ONE call of the network-level C entry where it covers the configuration
    (the tactile decoder branch on the first side lane, as unet_forward runs it), else the Python schedule"""
    if not unet_c_ok(G, style_code):
        return unet_forward(G, x, style_code=style_code, keep=False, style_tiles=style_tiles)[0]
    tile = None
    if style_code is not None:
        i = G.num_downs - 1
        tile = style_tiles.get(i) if style_tiles else None
        if tile is None:
            x0 = _as_act(x[0] if isinstance(x, (tuple, list)) else x).data
            hh, ww = x0.shape[2] >> G.num_downs, x0.shape[3] >> G.num_downs
            tile = style_code.to(torch.float32)[:, :, None, None].expand(-1, -1, hh, ww).contiguous()
    side = None
    if PARALLEL_SCALES and G.num_layer_separate > 0:
        pool = _SIDE_STREAMS.setdefault(torch.cuda.current_device(), [])
        if not pool:
            pool.append(torch.cuda.Stream())
        side = pool[0]
    return unet_forward_c(G, x, style_tile=tile, side_stream=side)


def unet_forward_c(G, x, style_tile=None, g_out=None, side_stream=None):
    """the generator's inference forward through ONE C call (vts_unet_forward); bit-identical to unet_forward(keep=False)"""
    import ctypes as C

    lib = L.load()
    x0 = _as_act(x[0] if isinstance(x, (tuple, list)) else x)
    n, _, h, w = x0.data.shape
    if g_out is None:
        nls = G.num_layer_separate
        oc = G.up0.conv.weight.shape[1] + (G.up0_T.conv.weight.shape[1] if nls > 0 else 0)
        g_out = _empty(n, oc, h, w, x0.data.device)
    d = unet_desc(G, x, g_out, style_tile, side_stream)
    need = lib.vts_unet_forward_ws_floats(C.byref(d))
    if need < 0:
        raise RuntimeError("vts_unet_forward: %s" % lib.vts_last_error().decode())
    ws = torch.empty(int(need), dtype=torch.float32, device=x0.data.device)
    L.check(lib.vts_unet_forward(C.byref(d), ws.data_ptr(), ws.numel(), L.stream()), "vts_unet_forward")
    return g_out


def _style_map_forward(G, j, style, hh, ww):
    """style_code_mapping<j> (networks.py:1459-1465, 1611-1615): Linear(style_dim, P, bias=False) -> BatchNorm1d (batch_size > 1) |
    InstanceNorm1d -> ReLU, reshaped to [N, P / (h w), h, w].  The Linear is a 1 x 1 convolution on a 1 x 1 map; BatchNorm1d over
    [N, P] is BatchNorm2d on [N, P, 1, 1]; InstanceNorm1d on the 2-D tensor normalises every row over its P features (torch treats
    [N, P] as an unbatched (C = N, L = P) input), i.e. InstanceNorm2d on [N, 1, P, 1]."""
    m = getattr(G, "style_code_mapping%d" % j)
    lin = getattr(m, "0")
    bn = getattr(m, "1", None)
    n, k = style.shape
    P = lin.weight.shape[0]
    if P % (hh * ww):
        raise ValueError("style_code_mapping%d emits %d features, not a multiple of the %d x %d map: the reference builds it for a %d-pixel "
                         "input (networks.py:1432, 1458)" % (j, P, hh, ww, 1536))
    z = _empty(n, P, 1, 1, style.device)
    sv = style.reshape(n, k, 1, 1).contiguous()
    ops.convk(sv, lin.weight.view(P, k, 1, 1), z, pad=0)
    if bn is not None:
        if G.training:
            a = ops.norm_stats(z, 1, gamma=bn.weight, beta=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var, nbt=bn.num_batches_tracked)
        else:
            sc = bn.weight / torch.sqrt(bn.running_var + 1e-5)
            a = Act(z, sc.repeat(n).contiguous(), (bn.bias - bn.running_mean * sc).repeat(n).contiguous())
        y = ops.pad_affine(a, (0, 0, 0, 0), 0, act=RELU)
    else:
        a = ops.norm_stats(z.view(n, 1, P, 1), 0)
        y = ops.pad_affine(a, (0, 0, 0, 0), 0, act=RELU)
    return y.view(n, P // (hh * ww), hh, ww), (j, sv, a, bn)


def _style_map_backward(G, sctx, d_map, want_dstyle=False):
    """parameter gradients of style_code_mapping<j> from the gradient w.r.t. its output map; returns d/d style_code when asked"""
    j, sv, a, bn = sctx[:4]
    m = getattr(G, "style_code_mapping%d" % j)
    lin = getattr(m, "0")
    n, k = sv.shape[0], sv.shape[1]
    P = lin.weight.shape[0]
    g = d_map.reshape(a.data.shape).contiguous()
    dz = torch.empty_like(g)
    ops.act_bwd(g, a, RELU, dz)
    if bn is not None:
        if G.training:
            ops.norm_bwd(dz, a, 1, gamma=bn.weight, dgamma=bn.weight.grad, dbeta=bn.bias.grad)
        else:
            raise NotImplementedError("backward through an eval-mode BatchNorm1d")
    else:
        ops.norm_bwd(dz, a, 0)
    dz = dz.view(n, P, 1, 1)
    ops.wgradk(dz, sv, lin.weight.grad.view(P, k, 1, 1), pad=0)
    if not want_dstyle:
        return None
    ds = torch.empty_like(sv)
    ops.convk_bwd_data(dz, lin.weight.view(P, k, 1, 1), ds, pad=0)
    return ds.view(n, k)


def unet_backward(G, ctx, d_raw):
    """d_raw: [N,5,H,W] gradient wrt the pre-tanh outputs of up0 / up0_T.
    Writes every parameter's .grad (overwrite) -- the G step has a single backward.
    The deterministic reduction of all weight-gradient partials is ONE launch at the end (ops.deferred_wgrad)."""
    with ops.deferred_wgrad():
        _unet_backward(G, ctx, d_raw)


def unet_backward_decoder(G, ctx, d_raw):
    """first half of unet_backward: every decoder (up*) gradient is complete when it returns (data-parallel runs all-reduce that
    bucket while unet_backward_encoder runs).  Returns the state the second half needs."""
    with ops.deferred_wgrad():
        return _unet_backward(G, ctx, d_raw, part="decoder")


def unet_backward_encoder(G, ctx, state):
    with ops.deferred_wgrad():
        _unet_backward(G, ctx, None, part="encoder", state=state)


def _unet_backward(G, ctx, d_raw, part="all", state=None):
    if part == "encoder":
        dfeat = state
        feats = ctx.feats
        nd = G.num_downs
        dev = dfeat[nd - 1].device
        sq = SideQueue()
        return _unet_backward_encoder(G, ctx, dfeat, feats, nd, dev, sq)
    nd, ch = G.num_downs, G.channels
    n = d_raw.shape[0]
    dev = d_raw.device
    feats = ctx.feats
    sq = SideQueue()         # weight / bias gradients: off the critical path
    dfeat = [None] * nd      # grad wrt the normalised feats[i] (pre-activation), accumulated over consumers
    dx = {}                  # id(Act) of an up-output -> grad wrt its normalised value

    def add_grad(store, key, shape):
        """returns (tensor, accumulate?)"""
        if key in store and store[key] is not None:
            return store[key], True
        t = torch.empty(shape, dtype=torch.float32, device=dev)
        store[key] = t
        return t, False

    dropped = {id(v[1]) for v in ctx.ups.values() if v[3] is not None}

    def dropped_input(inp):
        """the block's primary input is a Dropout output: its gradient is not yet the gradient of a normalised map (no fused sums)"""
        return id(inp) in dropped

    def up_bwd(i, name, dx, dfeat, wq, lane_mode=False):
        """backward of one up block: weight gradient (through wq: side queue, or inline inside a lane), gradient w.r.t.
        the block input into dx / dfeat[nd-1], gradient w.r.t. the skip feature into dfeat[i]"""
        skip = None if i in (0, nd - 1) else feats[i]
        blk = getattr(G, name).conv
        inp, outp, extra, drop = ctx.ups[name]
        outer = blk.weight.shape[1]
        if i == 0:
            c0 = 0 if name == "up0" else 3
            g = d_raw[:, c0:c0 + outer]  # channel-slice view: batch stride stays 5*H*W
        else:
            g = dx.pop(id(outp))
            if drop is not None:     # Dropout backward: the same keep * 2 map, then the InstanceNorm backward on the block's own statistics
                gd = torch.empty_like(g)
                ops.mask_mul(g.view(-1, 1, g.shape[2], g.shape[3]), drop[1].view(-1, 1, g.shape[2], g.shape[3]), out=gd.view(-1, 1, g.shape[2], g.shape[3]))
                g = gd
                ops.norm_bwd(g, drop[0], 0)
            else:
                ops.norm_bwd(g, outp, 0)
        gop = Act(g)
        second = skip if skip is not None else extra
        wq(lambda: ops.wgrad4x4(inp, gop, blk.weight.grad, lo1=second, act_lo=RELU, stride=2, pad=1), g)
        if i == 0:
            wq(lambda: ops.channel_sum(g, blk.bias.grad), g)
        # else: the bias feeds an InstanceNorm, so its gradient is identically zero (the norm removes any
        # per-channel constant).  The reference computes ~1e-9 rounding noise there; the flat gradient
        # buffer is zero-initialised and this slot is never written, i.e. exactly 0.
        c_in0 = inp.data.shape[1]
        # grad wrt the primary input (normalised output of the previous up block, or feats[nd-1])
        if i == nd - 1:
            tgt, acc = add_grad_list(dfeat, nd - 1, inp.data.shape, dev)
        else:
            tgt, acc = add_grad(dx, id(inp), inp.data.shape)
        # (the tensor goes straight into the InstanceNorm backward of the block below unless it is the un-normalised innermost feature
        #  or the split point i == nls - 1, where up{i} and up{i}_T BOTH contribute -- as two lanes or, with VTS_PARALLEL_SCALES=0, as
        #  two accumulating calls: the fused sums (and the k-split epilogue's fused backward) would see only one part of the gradient)
        ops.conv4x4(gop, blk.weight, outer * 16, 16, c_in0, tgt, stride=2, pad=1, dmask=inp, dmask_act=RELU, accumulate=acc,
                    bwd_sums="in" if (i != nd - 1) and not (nls > 0 and i == nls - 1) and not dropped_input(inp) else False)
        if skip is not None:
            tgt, acc = add_grad_list(dfeat, i, skip.data.shape, dev)
            wv = blk.weight.view(-1)[c_in0 * outer * 16:]
            ops.conv4x4(gop, wv, outer * 16, 16, skip.data.shape[1], tgt, stride=2, pad=1, dmask=skip, dmask_act=RELU,
                        accumulate=acc)
        elif extra is not None and i in ctx.style_ctx and G.style_mode == "concat":
            # projected style map as the second concat source: its gradient feeds style_code_mapping<j>'s parameters
            d_map = torch.empty_like(extra.data)
            wv = blk.weight.view(-1)[c_in0 * outer * 16:]
            ops.conv4x4(gop, wv, outer * 16, 16, extra.data.shape[1], d_map, stride=2, pad=1, dmask=extra, dmask_act=RELU)
            ctx.dstyle = _style_map_backward(G, ctx.style_ctx[i], d_map, want_dstyle=True)

    nls = G.num_layer_separate
    if nls > 0 and PARALLEL_SCALES:
        # the visual and the tactile decoder are independent chains: two lanes with their own gradient buffers,
        # summed where the chains share a tensor (skip features, the split point)
        lane_dx, lane_df = [{}, {}], [[None] * nd, [None] * nd]

        def inline(fn, *keep):
            fn()

        def chain(lane):
            # (weight gradients inline.  Round 4 re-measured them on a queue of their own per lane -- four streams, both queues forked
            # from the launch stream in front of the lanes: + 0.22 ms on the step, as with one shared queue in round 3.  Round 6: the
            # lanes' SKIP-feature gradients -- first read when the encoder's backward reaches their level -- enqueued on the side queue
            # behind the lanes instead of inside them: 5.31 - 5.34 against 5.25 - 5.28 ms (profiles/r06a_experiments.md section 2b).  The backward
            # is throughput-bound: work moved beside its chain slows the chain by as much.)
            for i in range(nls):
                up_bwd(i, "up%d%s" % (i, "_T" if lane else ""), lane_dx[lane], lane_df[lane], inline, lane_mode=True)

        _run_lanes(2, chain)

        def total(a, b):
            return a if b is None else (b if a is None else ops.pad_affine(a, (0, 0, 0, 0), 0, res=b))

        for i in range(nd):
            dfeat[i] = total(lane_df[0][i], lane_df[1][i])
        for key in set(lane_dx[0]) | set(lane_dx[1]):
            dx[key] = total(lane_dx[0].get(key), lane_dx[1].get(key))
        first_shared = nls
    else:
        first_shared = 0
    for i in range(first_shared, nd):
        names = ["up%d" % i] + (["up%d_T" % i] if nls >= i + 1 else [])
        for name in names:
            up_bwd(i, name, dx, dfeat, sq.run)

    if part == "decoder":
        sq.join()
        return dfeat
    return _unet_backward_encoder(G, ctx, dfeat, feats, nd, dev, sq)


def _unet_backward_encoder(G, ctx, dfeat, feats, nd, dev, sq):
    if (nd - 1) in ctx.adain_in:      # up7 consumed adain(feats[7], style map): split its input gradient into content and style parts
        x_raw = ctx.adain_in[nd - 1]
        smap = ctx.style_ctx[nd - 1][-1]
        dxr, dsm = ops.adain_bwd(dfeat[nd - 1].contiguous(), x_raw, smap)
        dfeat[nd - 1] = dxr
        ctx.dstyle = _style_map_backward(G, ctx.style_ctx[nd - 1], dsm, want_dstyle=True)
    for i in range(nd - 1, -1, -1):
        blk = getattr(G, "down%d" % i).conv
        g = dfeat[i]
        if 0 < i < nd - 1:
            ops.norm_bwd(g, feats[i], 0)
        src, src1 = ctx.x if i == 0 else (feats[i - 1], None)
        sq.run(lambda: ops.wgrad4x4(Act(g), src, blk.weight.grad, hi1=src1, act_hi=LRELU if i else 0, stride=2, pad=1), g)
        if not (0 < i < nd - 1):
            sq.run(lambda: ops.channel_sum(g, blk.bias.grad), g)   # bias gradients of normalised layers are identically zero (see above)
        if i > 0:
            cin = blk.weight.shape[1]
            tgt, acc = add_grad_list(dfeat, i - 1, feats[i - 1].data.shape, dev)
            ops.conv4x4(Act(g), blk.weight, 16, cin * 16, cin, tgt, stride=2, pad=1, transposed=True, dmask=feats[i - 1],
                        dmask_act=LRELU, accumulate=acc, bwd_sums="in" if i - 1 > 0 else False)     # the last contribution to dfeat[i-1]: its InstanceNorm backward is next
        dfeat[i] = None
    sq.join()


# -------------------------------------------------------------------------------------
# ResNet generator (--netG resnet_{4,6,9}blocks): /root/reference/models/networks.py:1051-1154
# -------------------------------------------------------------------------------------

class ResnetCtx:
    __slots__ = ("steps", "g_out")


def _g_norm(G, r, bn):
    """normalisation of a raw conv output as an Act: InstanceNorm (bn None) or BatchNorm2d (train: batch
    statistics + running-stat update; eval: the running statistics)."""
    if bn is None:
        return ops.norm_stats(r, 0)
    if G.training:
        return ops.norm_stats(r, 1, gamma=bn.weight, beta=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var,
                              nbt=bn.num_batches_tracked)
    n = r.shape[0]
    sc = bn.weight / torch.sqrt(bn.running_var + 1e-5)     # [C] vectors: plumbing, not arithmetic on activations
    sh = bn.bias - bn.running_mean * sc
    return Act(r, sc.repeat(n).contiguous(), sh.repeat(n).contiguous())


def _g_norm_bwd(buf, a, bn):
    if bn is None:
        ops.norm_bwd(buf, a, 0)
    else:
        ops.norm_bwd(buf, a, 1, gamma=bn.weight, dgamma=bn.weight.grad, dbeta=bn.bias.grad)


def _is_wide(conv):
    """layers big enough for the GEMM-class 3x3 kernel (vts_conv3x3_wide) to pay: >= 64 x 64 channels"""
    co, ci = conv.weight.shape[:2]
    return co >= 64 and ci >= 64 and co % 4 == 0 and ci % 4 == 0 and conv.weight.shape[2] == 3


def _conv_valid3(p, conv, out):
    """valid 3x3 conv of a pre-padded identity tensor"""
    if _is_wide(conv):
        n, ci, ph, pw = p.shape
        if ops.conv3x3_wino_ok(n, ci, out.shape[1], ph - 2, pw - 2):      # Winograd F(2x2, 3x3): 1.5 - 1.9x the direct kernel (round 4)
            return ops.conv3x3_wino(p, ops.w3x3_wino_pack(conv.weight, "conv_fwd"), conv.bias, out)
        return ops.conv3x3_wide(p, ops.w3x3_pack(conv.weight, "conv_fwd"), conv.bias, out)
    return ops.convk(p, conv.weight, out, bias=conv.bias, pad=0)


def _conv_valid3_bwd_data(g, conv, dp):
    """dp (padded size) <- adjoint of _conv_valid3 w.r.t. its input"""
    if _is_wide(conv):
        q = ops.pad_affine(g, (2, 2, 2, 2), 0)
        n, co, qh, qw = q.shape
        if ops.conv3x3_wino_ok(n, co, dp.shape[1], qh - 2, qw - 2):
            return ops.conv3x3_wino(q, ops.w3x3_wino_pack(conv.weight, "conv_adj"), None, dp)
        return ops.conv3x3_wide(q, ops.w3x3_pack(conv.weight, "conv_adj"), None, dp)
    return ops.convk_bwd_data(g, conv.weight, dp, pad=0)


def _wgrad_valid3(g, p, conv):
    """conv.weight.grad <- weight gradient of _conv_valid3"""
    if _is_wide(conv):
        return ops.wgrad3x3_wide(g, p, conv.weight.grad)
    return ops.wgradk(g, p, conv.weight.grad, pad=0)


def _conv3(x, conv, out, stride, act_in):
    if stride == 1:
        return ops.convk(x, conv.weight, out, bias=conv.bias, pad=1, act_in=act_in)
    w4 = ops.tap_embed(conv.weight, 3, 0, 0, ops._w4_scratch(conv.weight, 3, "fwd")[0])
    co, ci = conv.weight.shape[:2]
    return ops.conv4x4(x, w4, ci * 16, 16, co, out, bias=conv.bias, stride=2, pad=1, act_in=act_in)


def _seq_forward(G, seq, srcs, cur, pending, dropout_masks=None):
    """Run one nn.Sequential-like segment (`seq.layout` over `seq.mod` / `seq.block_mods`).  srcs: the network input(s)
    (Acts, concatenated on store into the first padded tensor) when the segment starts at the input (cur None);
    otherwise cur is the incoming activation (Act with `pending` activation, or an identity tensor).
    Returns (cur, pending, steps); steps hold what the backward needs.  G supplies the train / eval mode."""
    first = srcs[0].data if cur is None else (cur.data if isinstance(cur, Act) else cur)
    n, _, h, w = first.shape
    dev = first.device
    steps = []
    lay = seq.layout
    i = 0

    def shape_of(t):
        return (t.data if isinstance(t, Act) else t).shape

    while i < len(lay):
        e = lay[i]
        kind = e["kind"]
        if kind == "pad":      # ReflectionPad2d(3) + conv7 (+ norm / relu | tanh)
            conv = seq.mod(lay[i + 1]["idx"])
            if cur is None:    # network input: concat the sources while padding
                cin = sum(s_.data.shape[1] for s_ in srcs)
                p = _empty(n, cin, h + 6, w + 6, dev)
                c0 = 0
                for s_ in srcs:
                    c = s_.data.shape[1]
                    ops.pad_affine(s_, (3, 3, 3, 3), 1, out=p[:, c0:c0 + c], out_nstride=p.stride(0))
                    c0 += c
                src_act = None
            else:
                p = ops.pad_affine(cur, (3, 3, 3, 3), 1, act=pending)
                src_act = cur if isinstance(cur, Act) else None
            r = _empty(n, conv.weight.shape[0], p.shape[2] - 6, p.shape[3] - 6, dev)
            ops.convk(p, conv.weight, r, bias=conv.bias, pad=0)
            if lay[i + 2]["kind"] == "norm":
                bn = seq.mod(lay[i + 2]["idx"])
                cur, pending = _g_norm(G, r, bn), RELU
                steps.append(("conv7", conv, p, src_act, cur, bn))
                i += 4
            else:              # final conv + tanh
                g_out = ops.pad_affine(r, (0, 0, 0, 0), 0, act=TANH)
                steps.append(("conv7_out", conv, p, src_act, None, None))
                i += 3
                cur = g_out
        elif kind == "conv3":  # Conv2d(3, pad 1, stride 1 | 2) + norm + relu
            conv, bn = seq.mod(e["idx"]), seq.mod(lay[i + 1]["idx"])
            inp, inp_act, stride = cur, pending, e["stride"]
            _, _, ih, iw = shape_of(cur)
            r = _empty(n, conv.weight.shape[0], (ih - 1) // stride + 1, (iw - 1) // stride + 1, dev)
            if stride == 2 and (ih % 2 or iw % 2):
                raise ValueError("stride-2 3x3 convolutions need even sizes, got %dx%d" % (ih, iw))
            xp = None
            if _is_wide(conv):     # GEMM-class kernels: materialise relu(norm(.)) with the zero padding once
                xp = ops.pad_affine(inp, (1, 1, 1, 1), 0, act=inp_act)
                wt = ops.w3x3_pack(conv.weight, "conv_fwd")
                (ops.conv3x3_wide if stride == 1 else ops.conv3x3s2_wide)(xp, wt, conv.bias, r)
            else:
                _conv3(inp, conv, r, stride, inp_act)
            cur, pending = _g_norm(G, r, bn), RELU
            steps.append(("conv3", conv, inp, inp_act, cur, bn, stride, xp))
            i += 3
        elif kind == "convT3":  # ConvTranspose2d(3, stride 2, pad 1, output_padding 1) + norm + relu
            conv, bn = seq.mod(e["idx"]), seq.mod(lay[i + 1]["idx"])
            inp, inp_act = cur, pending
            _, _, ih, iw = shape_of(cur)
            ci, co = conv.weight.shape[:2]
            r = _empty(n, co, 2 * ih, 2 * iw, dev)
            z = None
            if _is_wide(conv):
                z = ops.pad_affine(inp, (0, 0, 0, 0), 0, act=inp_act) if (inp_act or isinstance(inp, Act)) else inp
                ops.tconv3x3s2_wide(ops.pad_affine(z, (0, 1, 0, 1), 0), ops.w3x3_pack(conv.weight, "convT_fwd"), conv.bias, r)
            else:
                w4 = ops.tap_embed(conv.weight, 3, 0, 0, ops._w4_scratch(conv.weight, 3, "fwd")[0])
                ops.conv4x4(inp, w4, 16, co * 16, co, r, bias=conv.bias, stride=2, pad=1, transposed=True, act_in=inp_act)
            cur, pending = _g_norm(G, r, bn), RELU
            steps.append(("convT3", conv, inp, inp_act, cur, bn, z))
            i += 3
        elif kind == "down":
            inp = cur
            cur, pending = ops.blur_down(cur, act=pending), 0
            steps.append(("down", inp))
            i += 1
        elif kind == "up":
            inp, inp_act = cur, pending
            cur, pending = ops.blur_up(cur, act=pending), 0
            steps.append(("up", inp, inp_act))
            i += 1
        elif kind == "block":  # x + norm(conv(reflpad(relu(norm(conv(reflpad(x)))))))
            ca, na, cb, nb = seq.block_mods(e["idx"])
            if pending or isinstance(cur, Act):   # first block after a strided conv: materialise relu(norm(r))
                blk_src = (cur, pending)
                cur, pending = ops.pad_affine(cur, (0, 0, 0, 0), 0, act=pending), 0
            else:
                blk_src = None
            xb = cur           # identity tensor
            p1 = ops.pad_affine(xb, (1, 1, 1, 1), 1)
            r1 = _empty(n, ca.weight.shape[0], xb.shape[2], xb.shape[3], dev)
            _conv_valid3(p1, ca, r1)
            a1 = _g_norm(G, r1, na)
            keep2 = None
            if getattr(G, "use_dropout", False) and G.training:
                # Dropout(0.5) between the ReLU and the second reflection pad (networks.py:1305-1306): relu(norm(r1)) is materialised,
                # multiplied by keep * 2, and padded as a plain tensor; the backward multiplies by the same map
                k = dropout_masks.pop(0) if dropout_masks else torch.bernoulli(torch.full(r1.shape, 0.5, device=dev))
                keep2 = (k.to(device=dev, dtype=torch.float32) * 2.0).contiguous()
                y1 = ops.pad_affine(a1, (0, 0, 0, 0), 0, act=RELU)
                y1d = torch.empty_like(y1)
                v = lambda t: t.view(-1, 1, t.shape[2], t.shape[3])
                ops.mask_mul(v(y1), v(keep2), out=v(y1d))
                p2 = ops.pad_affine(y1d, (1, 1, 1, 1), 1)
            else:
                p2 = ops.pad_affine(a1, (1, 1, 1, 1), 1, act=RELU)
            r2 = _empty(n, cb.weight.shape[0], xb.shape[2], xb.shape[3], dev)
            _conv_valid3(p2, cb, r2)
            a2 = _g_norm(G, r2, nb)
            cur, pending = ops.pad_affine(a2, (0, 0, 0, 0), 0, res=xb), 0
            steps.append(("block", ca, cb, p1, a1, p2, a2, na, nb, blk_src, keep2))
            i += 1
        else:
            raise RuntimeError("unexpected layout entry %r" % (e,))
    return cur, pending, steps


def resnet_forward(G, x, keep=True, dropout_masks=None):
    """x: tensor / Act, or a pair (x0, x1) concatenated on store into the first padded tensor.
    Returns (g_out [N,output_nc,H,W] post-tanh, ctx).  Every 3x3 / 7x7 conv runs as 4x4 tap blocks
    (ops.convk) or, for wide layers, on the GEMM-class kernels; normalisation is a statistics pass only and is
    applied on load by the consumer."""
    if getattr(G, "is_local_enhancer", False):
        return local_enhancer_forward(G, x, keep)
    srcs = [_as_act(t) for t in (x if isinstance(x, (tuple, list)) else (x,))]
    cur, _, steps = _seq_forward(G, G, srcs, None, 0, dropout_masks=list(dropout_masks) if dropout_masks else None)
    ctx = None
    if keep:
        ctx = ResnetCtx()
        ctx.steps, ctx.g_out = steps, cur
    return cur, ctx


def local_enhancer_forward(G, x, keep=True):
    """pix2pixHD LocalEnhancer.forward (networks.py:1933-1949): the global trunk on the input average-pooled L = n_local_enhancers times,
    then enhancer n = 1 .. L: its downsampling branch on pyramid level L - n, plus the output below, through its upsampling branch"""
    x = _as_act(x)
    L = G.n_local_enhancers
    pyr = [x]
    for _ in range(L):
        pyr.append(Act(ops.avgpool(pyr[-1].data)))
    a, pa, steps_g = _seq_forward(G, G.seq_global, [pyr[-1]], None, 0)
    levels = []
    for n in range(1, L + 1):
        b, pb, steps_1 = _seq_forward(G, G.seq_down[n - 1], [pyr[L - n]], None, 0)
        bm = ops.pad_affine(b, (0, 0, 0, 0), 0, act=pb)
        s = ops.pad_affine(a, (0, 0, 0, 0), 0, act=pa, res=bm)
        out, pout, steps_2 = _seq_forward(G, G.seq_up[n - 1], None, s, 0)
        levels.append((steps_1, steps_2, a, b))
        a, pa = out, pout
    ctx = None
    if keep:
        ctx = ResnetCtx()
        ctx.steps, ctx.g_out = (steps_g, levels), a
    return a, ctx


def _through_norm_relu(g_act, a, bn):
    """gradient w.r.t. relu(norm(r)) -> gradient w.r.t. the raw conv output r (in a fresh buffer)"""
    buf = torch.empty_like(a.data)
    ops.act_bwd(g_act, a, RELU, buf)
    _g_norm_bwd(buf, a, bn)
    return buf


def _bn_map(*step_lists):
    """Act -> the BatchNorm module that produced its affine (None: InstanceNorm)"""
    bn_of = {}
    for steps in step_lists:
        for st in steps:
            if st[0] in ("conv7", "conv3", "convT3"):
                bn_of[id(st[4])] = st[5]
    return bn_of


def _seq_backward(steps, g, sq, bn_of, start=0, stop=None):
    """backward of one segment: g = gradient w.r.t. its output (the pre-tanh output for a final conv7, else w.r.t. the
    raw output of its last conv / the identity output of its last block).  Writes the parameters' .grad (overwrite;
    through the side queue sq) and returns the gradient w.r.t. the segment's input (None for a network input).
    start / stop: only steps[start:stop] (the stages of resnet_backward_stage)."""
    dev = g.device
    through_norm_relu = _through_norm_relu

    def producer_bn(a):
        return bn_of.get(id(a))

    for st in reversed(steps[start:stop]):
        kind = st[0]
        if kind == "conv7_out":
            _, conv, p, src_act, _, _ = st
            sq.run(lambda: (ops.wgradk(g, p, conv.weight.grad, pad=0), ops.channel_sum(g, conv.bias.grad)), g)
            dp = torch.empty_like(p)
            ops.convk_bwd_data(g, conv.weight, dp, pad=0)
            da = _empty(p.shape[0], p.shape[1], p.shape[2] - 6, p.shape[3] - 6, dev)
            ops.pad_bwd(dp, (3, 3, 3, 3), 1, da)
            g = through_norm_relu(da, src_act, producer_bn(src_act))   # the padded tensor was relu(norm(r_prev))
        elif kind == "conv7":
            _, conv, p, src_act, a, bn = st
            sq.run(lambda: ops.wgradk(g, p, conv.weight.grad, pad=0), g)  # network input: no gradient needed below
        elif kind == "conv3":
            _, conv, inp, inp_act, a, bn, stride, xp = st
            hi_act = inp if isinstance(inp, Act) else Act(inp)
            t = inp.data if isinstance(inp, Act) else inp
            din = torch.empty_like(t)
            if xp is not None:     # wide layer
                sq.run(lambda: ops.wgrad3x3_wide(g, xp, conv.weight.grad, stride=stride), g)
                if stride == 1:
                    dxp = torch.empty_like(xp)
                    ops.conv3x3_wide(ops.pad_affine(g, (2, 2, 2, 2), 0), ops.w3x3_pack(conv.weight, "conv_adj"), None, dxp)
                    ops.pad_bwd(dxp, (1, 1, 1, 1), 0, din)
                else:
                    ops.tconv3x3s2_wide(ops.pad_affine(g, (0, 1, 0, 1), 0), ops.w3x3_pack(conv.weight, "conv_s2_adj"), None, din)
            elif stride == 1:
                sq.run(lambda: ops.wgradk(g, hi_act, conv.weight.grad, pad=1, act_hi=inp_act), g)
                ops.convk_bwd_data(g, conv.weight, din, pad=1)
            else:
                co, ci = conv.weight.shape[:2]
                dw4 = ops._w4_scratch(conv.weight.grad, 3, "grad")[0]
                sq.run(lambda: (ops.wgrad4x4(g, hi_act, dw4, stride=2, pad=1, act_hi=inp_act, defer=False), ops.tap_extract(dw4, 3, 0, 0, conv.weight.grad)), g)
                w4 = ops.tap_embed(conv.weight, 3, 0, 0, ops._w4_scratch(conv.weight, 3, "fwd")[0])
                ops.conv4x4(g, w4, 16, ci * 16, ci, din, stride=2, pad=1, transposed=True)
            g = through_norm_relu(din, inp, producer_bn(inp)) if inp_act == RELU else din
        elif kind == "convT3":
            _, conv, inp, inp_act, a, bn, z = st
            ci, co = conv.weight.shape[:2]
            lo_act = inp if isinstance(inp, Act) else Act(inp)
            t = inp.data if isinstance(inp, Act) else inp
            din = torch.empty_like(t)
            if z is not None:      # wide layer
                gp = ops.pad_affine(g, (1, 1, 1, 1), 0)
                sq.run(lambda: ops.wgrad3x3_wide(z, gp, conv.weight.grad, stride=2), gp)
                ops.conv3x3s2_wide(gp, ops.w3x3_pack(conv.weight, "convT_adj"), None, din)
            else:
                dw4 = ops._w4_scratch(conv.weight.grad, 3, "grad")[0]
                sq.run(lambda: (ops.wgrad4x4(lo_act, g, dw4, stride=2, pad=1, act_lo=inp_act, defer=False), ops.tap_extract(dw4, 3, 0, 0, conv.weight.grad)), g)
                w4 = ops.tap_embed(conv.weight, 3, 0, 0, ops._w4_scratch(conv.weight, 3, "fwd")[0])
                ops.conv4x4(g, w4, co * 16, 16, ci, din, stride=2, pad=1)
            g = through_norm_relu(din, inp, producer_bn(inp)) if inp_act == RELU else din
        elif kind == "down":
            inp = st[1]       # Act: relu(norm(r)) on load
            da = torch.empty_like(inp.data)
            ops.blur_down_bwd(g, da)
            g = through_norm_relu(da, inp, producer_bn(inp))
        elif kind == "up":
            _, inp, inp_act = st
            t = inp.data if isinstance(inp, Act) else inp
            da = torch.empty_like(t)
            ops.blur_up_bwd(g, da)
            g = through_norm_relu(da, inp, producer_bn(inp)) if inp_act == RELU else da
        elif kind == "block":
            _, ca, cb, p1, a1, p2, a2, na, nb, blk_src, keep2 = st
            dy = g
            g2 = dy.clone()
            _g_norm_bwd(g2, a2, nb)
            sq.run(lambda: _wgrad_valid3(g2, p2, cb), g2)
            dp2 = torch.empty_like(p2)
            _conv_valid3_bwd_data(g2, cb, dp2)
            da1 = torch.empty_like(a1.data)
            ops.pad_bwd(dp2, (1, 1, 1, 1), 1, da1)
            if keep2 is not None:      # Dropout backward
                dd = torch.empty_like(da1)
                v = lambda t: t.view(-1, 1, t.shape[2], t.shape[3])
                ops.mask_mul(v(da1), v(keep2), out=v(dd))
                da1 = dd
            g1 = through_norm_relu(da1, a1, na)
            sq.run(lambda: _wgrad_valid3(g1, p1, ca), g1)
            dp1 = torch.empty_like(p1)
            _conv_valid3_bwd_data(g1, ca, dp1)
            ops.pad_bwd(dp1, (1, 1, 1, 1), 1, dy, accumulate=True)   # + the skip path
            g = dy
            if blk_src is not None:    # the block input was materialised from relu(norm(r)) of a strided conv
                src, src_act = blk_src
                g = through_norm_relu(g, src, producer_bn(src)) if src_act == RELU else g
        else:
            raise RuntimeError(kind)
    return g


def resnet_backward(G, ctx, d_raw):
    """d_raw: gradient w.r.t. the pre-tanh output.  Writes every parameter's .grad (overwrite).
    Biases that feed a normalisation have identically zero gradient and are never written."""
    sq = SideQueue()     # weight / bias gradients: off the critical path of the backward-data chain
    if getattr(G, "is_local_enhancer", False):
        steps_g, levels = ctx.steps
        bn_of = _bn_map(steps_g, *[lv[0] for lv in levels], *[lv[1] for lv in levels])
        g = d_raw
        for steps_1, steps_2, a, b in reversed(levels):
            g_s = _seq_backward(steps_2, g, sq, bn_of)         # gradient w.r.t. relu(norm(a)) + relu(norm(b))
            _seq_backward(steps_1, _through_norm_relu(g_s, b, bn_of.get(id(b))), sq, bn_of)
            g = _through_norm_relu(g_s, a, bn_of.get(id(a)))   # ... w.r.t. the raw output of the level below
        _seq_backward(steps_g, g, sq, bn_of)
    else:
        _seq_backward(ctx.steps, d_raw, sq, _bn_map(ctx.steps))
    sq.join()


def _step_weights(st):
    """the convolution weights a step of a ResNet-family generator owns"""
    if st[0] in ("conv7_out", "conv7", "conv3", "convT3"):
        return [st[1].weight]
    if st[0] == "block":
        return [st[1].weight, st[2].weight]
    return []


def resnet_stage_bounds(ctx, cut_params):
    """Step indices at which a backward pass is cut so that, once stage j (steps[b[j]:b[j+1]], run from the last stage to the first) has
    finished, every parameter of gradient bucket j has its final gradient (vts/optim.py:FlatParams.chunk: every bucket but the first
    starts at a convolution weight).  A bucket that starts inside a two-convolution block is complete after the stage that holds the
    block (the block's first convolution then belongs to the next bucket down and is simply early).  Returns [0, s_1, .., s_K-1, len(steps)]."""
    steps = ctx.steps
    where = {}
    for i, st in enumerate(steps):
        for w in _step_weights(st):
            where[w.data_ptr()] = i
    bounds = [0]
    for p in cut_params:
        i = where.get(p.data_ptr())
        if i is None:
            raise RuntimeError("a gradient bucket does not start at a layer of this generator's backward")
        bounds.append(max(i, bounds[-1]))
    bounds.append(len(steps))
    return bounds


def resnet_backward_stage(G, ctx, g, lo, hi):
    """steps[lo:hi] of resnet_backward (single generator), weight gradients joined: on return every parameter of these steps has its
    final .grad, and the returned tensor is the gradient flowing into steps[:lo].  Chaining the stages from the last to the first equals
    resnet_backward; a data-parallel step starts the all-reduce of a stage's gradient bucket behind it (models/pix2pixHD_model.py)."""
    if getattr(G, "is_local_enhancer", False):
        raise NotImplementedError("staged backward of the local enhancer")
    sq = SideQueue()
    g = _seq_backward(ctx.steps, g, sq, _bn_map(ctx.steps), lo, hi)
    sq.join()
    return g


def add_grad_list(lst, idx, shape, dev):
    if lst[idx] is not None:
        return lst[idx], True
    lst[idx] = torch.empty(shape, dtype=torch.float32, device=dev)
    return lst[idx], False


# =====================================================================================
# multiscale discriminator
# =====================================================================================

class MsdCtx:
    __slots__ = ("scales",)


def _pool_act(a):
    return None if a is None else Act(ops.avgpool(a.data))


# The scales of a multiscale discriminator are independent chains of small kernels (the D2 passes over 32x32
# tactile patches never fill 256 CUs): they run concurrently on side HIP streams, forked from and joined back
# into the launch stream (so the schedule is still a DAG that torch.cuda.CUDAGraph captures as such).
PARALLEL_SCALES = tune.get("VTS_PARALLEL_SCALES", "1") != "0"
_SIDE_STREAMS = {}


SIDE_QUEUES = int(tune.get("VTS_SIDE_QUEUES", "2"))     # round 4: 5.65 -> 5.59 ms (three: 5.61)


class SideQueue:
    """Work that nothing on the launch stream waits for until the end of a phase -- the weight / bias gradients of a
    backward pass (28 % of the step's kernel time, all of it off the critical path of the backward-data chain) --
    is enqueued on side streams: `run` forks at the current point of the launch stream (the operands are ready
    there) and keeps the operand tensors alive; `join` makes the launch stream wait once, at the end.  Round 4: the items alternate
    between TWO streams (the generator backward's trunk / encoder phase has the launch chain and nothing else on the other queues, and
    its weight gradients take longer than its backward-data chain: the join used to wait for them)."""
    LANE = 7

    def __init__(self):
        self.main = torch.cuda.current_stream()
        side = _SIDE_STREAMS.setdefault(torch.cuda.current_device(), [])
        while len(side) < SideQueue.LANE:
            side.append(torch.cuda.Stream())
        # VTS_SIDE_QUEUES=2: items alternate between two streams (scratch / partial-arena index = the stream's lane number)
        self.lanes = [SideQueue.LANE - k for k in range(max(1, SIDE_QUEUES))]
        self.streams = [side[ln - 1] for ln in self.lanes]
        self.turn = 0
        self.keep = []
        self.on = PARALLEL_SCALES

    def run(self, fn, *tensors):
        if not self.on:
            return fn()
        k = self.turn % len(self.lanes)
        self.turn += 1
        stream, ln = self.streams[k], self.lanes[k]
        stream.wait_stream(self.main)
        lane, ops.WS_LANE = ops.WS_LANE, ln
        try:
            with torch.cuda.stream(stream):
                fn()
        except BaseException:
            ops.WS_LANE = lane
            ops.wgrad_discard()
            for st in self.streams:
                self.main.wait_stream(st)
            raise
        ops.WS_LANE = lane
        self.keep.extend(tensors)

    def join(self):
        if self.on:
            lane = ops.WS_LANE
            for stream, ln in zip(self.streams, self.lanes):
                ops.WS_LANE = ln
                with torch.cuda.stream(stream):
                    ops.wgrad_flush(ln)     # this queue's deferred weight-gradient reduction, on its own stream
            ops.WS_LANE = lane
            for stream in self.streams:
                self.main.wait_stream(stream)
        self.keep = []


def _run_lanes(n_lanes, body):
    """body(lane) for lane in range(n_lanes): lane 0 on the current stream, the others on side streams forked
    from it and joined back into it (a flat fork: nested forks made hipStreamEndCapture crash), each with its own
    scratch buffer.  The discriminator scales -- and the D1 / D2 updates of one step -- are independent chains
    of mostly small kernels; run side by side they fill the CUs that one chain leaves idle, and the schedule
    is still a DAG that torch.cuda.CUDAGraph captures as such."""
    if n_lanes == 1 or not PARALLEL_SCALES:
        for lane in range(n_lanes):
            body(lane)
        return
    # A call made from the launch-stream lane of an enclosing call (the generator forward's two decoder lanes inside the lane set of
    # _msd_multi(extra_main=True)) takes the side streams / scratch indices BEHIND the enclosing call's: `_LANE_BASE`.  (From a SIDE lane
    # a fork is still forbidden: hipStreamEndCapture does not survive it.)
    global _LANE_BASE
    base = _LANE_BASE
    main = torch.cuda.current_stream()
    dev = torch.cuda.current_device()
    side = _SIDE_STREAMS.setdefault(dev, [])
    while len(side) < base + n_lanes - 1:
        side.append(torch.cuda.Stream())
    mine = side[base:base + n_lanes - 1]
    for st in mine:
        st.wait_stream(main)
    ws0 = ops.WS_LANE
    try:
        for lane in range(1, n_lanes):
            ops.WS_LANE = base + lane
            with torch.cuda.stream(mine[lane - 1]):
                body(lane)
                ops.wgrad_flush(base + lane)       # deferred weight-gradient reductions of this lane, on its own stream
        ops.WS_LANE = ws0
        _LANE_BASE = base + n_lanes - 1
        try:
            body(0)
        finally:
            _LANE_BASE = base
        ops.wgrad_flush(ws0)
    except BaseException:
        ops.WS_LANE = ws0
        ops.wgrad_discard()                 # a lane body raised: no stale partial jobs for the next backward
        raise
    finally:
        for st in mine:                     # the side streams are joined back in every case (a capture must not end with a dangling fork)
            main.wait_stream(st)


_LANE_BASE = 0


def fork_lane(fn):
    """fn() on a side stream forked from the current stream, with its own scratch lane; returns a handle for join_lane.  Until the
    join, lane calls made on the launch stream (_run_lanes, msd_chain) take the side streams BEHIND this one -- the lane may stay open
    across several of them (the D2 chain of the training step runs beside the D1 chain AND the generator's backward).  fn must not
    fork lanes itself (no fork from a side stream: hipStreamEndCapture does not survive one).  Serial schedule: fn() runs inline."""
    global _LANE_BASE
    if not PARALLEL_SCALES:
        fn()
        return None
    base = _LANE_BASE
    main = torch.cuda.current_stream()
    side = _SIDE_STREAMS.setdefault(torch.cuda.current_device(), [])
    while len(side) < base + 1:
        side.append(torch.cuda.Stream())
    st = side[base]
    st.wait_stream(main)
    ws0, ops.WS_LANE = ops.WS_LANE, base + 1
    _LANE_BASE = base + 1
    try:
        with torch.cuda.stream(st):
            fn()
    except BaseException:
        _LANE_BASE = base
        main.wait_stream(st)
        raise
    finally:
        ops.WS_LANE = ws0
    return (st, base)


def join_lane(handle):
    """the current stream waits for the lane fork_lane opened; its side stream is free for other lanes again"""
    global _LANE_BASE
    if handle is None:
        return
    st, base = handle
    torch.cuda.current_stream().wait_stream(st)
    _LANE_BASE = base


def _pyramid(D, in0, in1):
    in0 = _as_act(in0)
    in1 = _as_act(in1) if in1 is not None else None
    pyr = [(in0, in1)]
    for s in range(1, D.num_D):
        pyr.append((_pool_act(pyr[-1][0]), _pool_act(pyr[-1][1])))
    return pyr


FLAT_D = tune.get("VTS_FLAT_D", "1") != "0"
FLAT_MIN_C = int(tune.get("VTS_FLAT_MIN_C", "64"))   # small maps: channels from which the flattened GEMM-class kernel takes over


WIDE_MIN_CI = int(tune.get("VTS_WIDE_MIN_CI", "64"))   # round 3 A/B: 32 / 64 (D1 layer 3 on the GEMM-class kernels) costs + 1.0 - 1.4 ms per step
WIDE_MIN_CO = int(tune.get("VTS_WIDE_MIN_CO", "128"))


def _flat4(conv, j, h, w, oh, ow, st):
    """whether this PatchGAN layer takes the GEMM-class route with 16-tap packed weights (vts_conv4x4_wide: flattened-batch kernel
    for maps of <= 128 pixels, tiled kernel above): wide layers only; Cout = 1 heads only on small maps"""
    co, ci = conv.weight.shape[0], conv.weight.shape[1]
    if not FLAT_D or j == 0 or ci < min(FLAT_MIN_C, WIDE_MIN_CI):
        return False
    small = ops.conv4x4_flat_ok(oh, ow, st * (oh - 1) + 4, st * (ow - 1) + 4)
    if small:
        return co >= FLAT_MIN_C or ci >= 256
    if ci < WIDE_MIN_CI:                     # full-size maps of the reference's own ndf = 8 discriminators stay on the 4x4 kernels
        return False
    return co >= WIDE_MIN_CO and co % 4 == 0 and ci % 4 == 0


def _packed4(conv, mode, cache):
    """16-tap packing of a PatchGAN layer's weight, once per discriminator invocation: the passes of one scale run back to back in
    one lane and the weights only change between invocations (Adam), so `cache` (created per msd_multi / msd_forward call) is safe"""
    if cache is None:
        return ops.w4x4_pack(conv.weight, mode)
    key = (id(conv), mode)
    buf = cache.get(key)
    if buf is None:
        buf = cache[key] = ops.w4x4_pack(conv.weight, mode)
    return buf


def _msd_scale_forward(D, s, a0, a1, update_stats, cache=None, groups=None, stat_rec=None, ext=None, skip_head=False):
    """one PatchGAN of the pyramid: returns the list of layer outputs (Act), the last one is the prediction.
    groups: sample indices where the passes batched into this call start (BatchNorm statistics per pass: ops.norm_stats);
    stat_rec: dict filled with {bn layer: (mean, unbiased var)} of this call; ext: (such a dict, after) whose running-statistics
    updates are spliced in after pass `after`."""
    n, dev = a0.data.shape[0], a0.data.device
    layer = getattr(D, "layer%d" % (D.num_D - 1 - s))
    acts = []
    cur0, cur1 = a0, a1
    h, w = cur0.data.shape[2], cur0.data.shape[3]
    for j, ci in enumerate(D.CONV_IDX):
        if skip_head and j == len(D.CONV_IDX) - 1:
            break      # a pass that only advances BatchNorm statistics at this scale: the prediction head has no normalisation behind it
        conv = getattr(layer, str(ci))
        st = D.STRIDE[ci]
        cout, cin = conv.weight.shape[0], conv.weight.shape[1]
        oh, ow = (h + 4 - 4) // st + 1, (w + 4 - 4) // st + 1
        out = _empty(n, cout, oh, ow, dev)
        if _flat4(conv, j, h, w, oh, ow, st):
            ph, pw = st * (oh - 1) + 4, st * (ow - 1) + 4
            p = ops.pad_affine(cur0, (2, ph - h - 2, 2, pw - w - 2), 0, act=LRELU)
            cur0.padded = p
            ops.conv4x4_wide(p, _packed4(conv, "conv_fwd", cache), conv.bias, out, stride=st)
            flat = True
        else:
            flat = False
        bnkw = None
        if ci in D.BN_IDX:
            bn = getattr(layer, str(D.BN_IDX[ci]))
            stat_out = None
            if stat_rec is not None:
                stat_out = stat_rec[ci] = (torch.empty(cout, dtype=torch.float32, device=dev), torch.empty(cout, dtype=torch.float32, device=dev))
            bnkw = dict(gamma=bn.weight, beta=bn.bias, running_mean=bn.running_mean if update_stats else None,
                        running_var=bn.running_var if update_stats else None,
                        nbt=bn.num_batches_tracked if update_stats else None, groups=groups, stat_out=stat_out,
                        ext=(ext[0][ci][0], ext[0][ci][1], ext[1]) if ext is not None else None)
        if not flat:     # Conv2d -> BatchNorm2d: the statistics come out of the convolution's epilogue where the tiled kernel runs
            a = ops.conv4x4(cur0, conv.weight, cin * 16, 16, cout, out, in1=cur1, bias=conv.bias, stride=st, pad=2,
                            act_in=LRELU if j else 0, batch_norm=bnkw)
            if bnkw is None:
                a = Act(out)
        elif bnkw is not None:
            a = ops.norm_stats(out, 1, **bnkw)
        else:
            a = Act(out)
        acts.append(a)
        cur0, cur1 = a, None
        h, w = oh, ow
    return acts


MSD_C = tune.get("VTS_MSD_C", "1") != "0"     # forward-only discriminator passes through the network-level C entry (0: the Python schedule)


def patchgan_desc(D, s, a0, a1, update_stats=True, stat_rec=None, run_head=True, pred=None):
    """vts_patchgan_desc of scale s of a multiscale discriminator (include/vts.h): weights from the module in the reference's state-dict
    layout, BatchNorm running buffers when update_stats, `stat_rec` {conv index: (mean, uvar)} filled with recording buffers"""
    layer = getattr(D, "layer%d" % (D.num_D - 1 - s))
    d = L.PatchganDesc()
    d.in0, d.in1 = ops._op(a0), ops._op(a1)
    x = a0.data if isinstance(a0, Act) else a0
    d.N, d.H, d.W = x.shape[0], x.shape[2], x.shape[3]
    d.n_convs = len(D.CONV_IDX)
    d.eps, d.momentum = 1e-5, 0.1
    keep = []
    for j, ci in enumerate(D.CONV_IDX):
        conv = getattr(layer, str(ci))
        d.cout[j], d.stride[j] = conv.weight.shape[0], D.STRIDE[ci]
        d.w[j], d.b[j] = conv.weight.data_ptr(), L.ptr(conv.bias)
        if ci in D.BN_IDX:
            bn = getattr(layer, str(D.BN_IDX[ci]))
            d.gamma[j], d.beta[j] = bn.weight.data_ptr(), bn.bias.data_ptr()
            if update_stats:
                d.running_mean[j], d.running_var[j] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                d.num_batches_tracked[j] = bn.num_batches_tracked.data_ptr()
            if stat_rec is not None:
                so = stat_rec[ci] = (torch.empty(conv.weight.shape[0], dtype=torch.float32, device=x.device),
                                     torch.empty(conv.weight.shape[0], dtype=torch.float32, device=x.device))
                d.stat_mean_out[j], d.stat_uvar_out[j] = so[0].data_ptr(), so[1].data_ptr()
    d.run_head = int(bool(run_head))
    if run_head:
        h, w = d.H, d.W
        for ci in D.CONV_IDX:
            h, w = h // D.STRIDE[ci] + 1, w // D.STRIDE[ci] + 1
        if pred is None:
            pred = _empty(d.N, 1, h, w, x.device)
        d.pred = pred.data_ptr()
    return d, pred, keep


def patchgan_c_ok(D, h, w):
    """the C entry runs every layer on the 4x4 convolution family: not where the Python schedule sends a wide layer to the GEMM-class
    kernels (_flat4: pix2pixHD's ndf = 64, depth >= 4 at ndf = 8), and PatchGAN depths within the descriptor; h, w: the input map"""
    if not MSD_C or len(D.CONV_IDX) > L.PATCHGAN_MAX_CONVS:
        return False
    layer = getattr(D, "layer0")
    for j, ci in enumerate(D.CONV_IDX):
        conv = getattr(layer, str(ci))
        st = D.STRIDE[ci]
        oh, ow = h // st + 1, w // st + 1
        if _flat4(conv, j, h, w, oh, ow, st) or conv.weight.shape[0] > 80:
            return False
        h, w = oh, ow
    return True


def patchgan_forward_c(D, s, a0, a1, update_stats=True, stat_rec=None, run_head=True):
    """scale s of a multiscale discriminator, training-mode forward, nothing kept: ONE call of vts_patchgan_forward (the path of the
    product's forward-only passes; bit-identical to _msd_scale_forward: tests/test_network_abi_gpu.py).  Returns the prediction map or None."""
    d, pred, _ = patchgan_desc(D, s, a0, a1, update_stats, stat_rec, run_head)
    lib = L.load()
    need = int(lib.vts_patchgan_forward_ws_floats(C.byref(d)))
    if need < 0:
        raise RuntimeError("vts_patchgan_forward: %s" % lib.vts_last_error().decode())
    ws = torch.empty(max(need, 1), dtype=torch.float32, device=pred.device if pred is not None else (a0.data if isinstance(a0, Act) else a0).device)
    L.check(lib.vts_patchgan_forward(C.byref(d), ws.data_ptr(), ws.numel(), L.stream()), "vts_patchgan_forward")
    return pred


def _msd_scale_backward(D, s, a0, a1, acts, g, param_grads, accumulate, want_input_grad, cache=None, groups=None, into=None):
    """backward of one PatchGAN; returns the gradient w.r.t. the second concat source (or None).
    into = (tensor, accumulate?): write / add that gradient straight into the caller's buffer (the full-resolution scale: saves the
    separate add of the merged pyramid gradient)"""
    layer = getattr(D, "layer%d" % (D.num_D - 1 - s))
    for j in range(len(D.CONV_IDX) - 1, -1, -1):
        ci = D.CONV_IDX[j]
        conv = getattr(layer, str(ci))
        st = D.STRIDE[ci]
        cin = conv.weight.shape[1]
        if ci in D.BN_IDX:
            bn = getattr(layer, str(D.BN_IDX[ci]))
            ops.norm_bwd(g, acts[j], 1, gamma=bn.weight, dgamma=bn.weight.grad if param_grads else None,
                         dbeta=bn.bias.grad if param_grads else None, accumulate=accumulate, groups=groups, beta=bn.bias)
        src0, src1 = (a0, a1) if j == 0 else (acts[j - 1], None)
        if param_grads:
            pp = src0.padded if j else None
            if pp is not None and not ops.conv4x4_flat_ok(g.shape[2], g.shape[3], pp.shape[2], pp.shape[3]):
                ops.wgrad4x4_wide(g, pp, conv.weight.grad, stride=st, accumulate=accumulate)     # full-size map: GEMM-class
            else:
                ops.wgrad4x4(Act(g), src0, conv.weight.grad, hi1=src1, act_hi=LRELU if j else 0, stride=st, pad=2,
                             accumulate=accumulate)
            if ci not in D.BN_IDX:   # a conv bias in front of a BatchNorm has an identically zero gradient
                ops.channel_sum(g, conv.bias.grad, accumulate=accumulate)
        if j > 0:
            prev = acts[j - 1]
            tgt = torch.empty_like(prev.data)
            h, w, oh, ow = prev.data.shape[2], prev.data.shape[3], g.shape[2], g.shape[3]
            qh, qw = (oh + 2, ow + 2) if st == 1 else (oh + 1, ow + 1)
            if _flat4(conv, j, h, w, oh, ow, st) and (cin % 4 == 0 or ops.conv4x4_flat_ok(h, w, qh, qw, st == 2)):
                raw = torch.empty_like(prev.data)
                if st == 1:   # adjoint of the stride-1 conv: the same operator on the padded gradient, flipped taps
                    ops.conv4x4_wide(ops.pad_affine(g, (1, 1, 1, 1), 0), _packed4(conv, "conv_adj", cache), None, raw)
                else:
                    ops.conv4x4_wide(ops.pad_affine(g, (0, 1, 0, 1), 0), _packed4(conv, "conv_s2_adj", cache), None, raw,
                                     stride=2, transposed=True)
                ops.act_bwd(raw, prev, LRELU, tgt)
            else:
                ops.conv4x4(Act(g), conv.weight, 16, cin * 16, cin, tgt, stride=st, pad=2, transposed=True, dmask=prev,
                            dmask_act=LRELU, bwd_sums=D.CONV_IDX[j - 1] in D.BN_IDX)
            g = tgt
        elif want_input_grad:      # w.r.t. the second concat source, or the only one (D1 without the sketch: --use_cGAN False)
            src = a1 if a1 is not None else a0
            c0 = a0.data.shape[1] if a1 is not None else 0
            c1 = src.data.shape[1]
            tgt, acc_in = (torch.empty_like(src.data), False) if into is None else into
            ops.conv4x4(Act(g), conv.weight.view(-1)[c0 * 16:], 16, cin * 16, c1, tgt, stride=st, pad=2, transposed=True, accumulate=acc_in)
            return tgt
    return None


def _merge_input_grads(din_scales, input_grad, defer_last=False):
    """d in1 = d0 + pool^T(d1 + pool^T(d2 ...)) into input_grad = (tensor, accumulate?).
    defer_last (only where the full-resolution scale wrote into the caller's buffer): the last, full-resolution pool^T is left to the
    caller -- returns d1 + pool^T(d2 ...), which ops.g_out_grad(coarse=...) adds on the fly."""
    dst, acc = input_grad
    last = 1 if (defer_last and len(din_scales) > 1 and din_scales[0] is dst and din_scales[1].is_contiguous()) else 0
    for s in range(len(din_scales) - 1, last, -1):
        ops.avgpool_bwd(din_scales[s], din_scales[s - 1], accumulate=True)
    if last:
        return din_scales[1]
    if din_scales[0] is dst:      # the full-resolution scale already wrote / accumulated into the caller's buffer
        return
    if acc:
        dst.add_(din_scales[0])
    else:
        dst.copy_(din_scales[0])


def msd_forward(D, in0, in1=None, keep=True, update_stats=True):
    """in0 (++ in1): channel-concatenated input, each a tensor/Act [N,C,H,W].
    Returns (preds: list over scales (full resolution first) of [N,1,h,w], ctx).
    BatchNorm runs in training mode (batch statistics); running buffers are updated when
    update_stats (every reference D call in a train step does: sinskitG_model.py:1361,1374,1490,...)."""
    pyr = _pyramid(D, in0, in1)
    scales = [None] * D.num_D

    def lane(s):
        scales[s] = (pyr[s][0], pyr[s][1], _msd_scale_forward(D, s, pyr[s][0], pyr[s][1], update_stats))

    _run_lanes(D.num_D, lane)
    preds = [sc[2][-1].data for sc in scales]
    if not keep:
        return preds, None
    ctx = MsdCtx()
    ctx.scales = scales
    return preds, ctx


def msd_backward(D, ctx, dpreds, param_grads=True, accumulate=False, input_grad=None):
    """dpreds: list over scales of d loss / d pred.
    param_grads: write (accumulate=False) or add (accumulate=True) into every parameter's .grad.
    input_grad: None, or (tensor [N,C1,H,W], accumulate_flag) receiving the gradient wrt `in1`
    (the second concat source: fake_I in the G step), summed over the pyramid."""
    din_scales = [None] * D.num_D

    def lane(s):
        a0, a1, acts = ctx.scales[s]
        din_scales[s] = _msd_scale_backward(D, s, a0, a1, acts, dpreds[s], param_grads, accumulate, input_grad is not None)

    with ops.deferred_wgrad():
        _run_lanes(D.num_D, lane)
    if input_grad is not None:
        _merge_input_grads(din_scales, input_grad)
    return None


def msd_multi(jobs, criterion, extra=None, extra_cost=0.1, extra_main=False, streams=None):
    """Several discriminators' passes of one training phase, all scales of all of them side by side.
    extra: a callable that runs as ONE MORE lane beside them (work of the phase that needs no discriminator: the generator's L1 /
    perceptual terms, which only read the forward's outputs); extra_cost: its rough time in ms (for the packing of lanes into streams).

    jobs: [(D, [pass, ...]), ...]; the passes of one D run in order (BatchNorm running statistics and gradient
    accumulation are order dependent), different D's and different scales are independent lanes.
    pass: dict(in0, in1=None, real: bool, coeff, slot, grad_coeff=None (None: no gradient / no backward),
               param_grads=True, accumulate=False, input_grad=None, loss=True, pyr=None); `preds` is filled in.
    pyr: [(Act, Act | None)] * num_D, the average-pooled input pyramid when the caller already holds it (the sketch / real-image
    levels do not change within a step, the fake-image levels are shared by the D update and the G step).
    A pass may BATCH several of the reference's discriminator calls (same weights, same input shape) along the sample axis:
      groups=[dict(n0, n1, real, coeff, slot, grad_coeff), ...]  -- samples [n0, n1) are one reference call: own BatchNorm batch
      statistics and running-statistics update (in list order), own loss term; one backward for all of them.
    stat_only=True: a forward-only pass that only RECORDS its BatchNorm statistics; ext_from=<that pass>, ext_after=k splices
    its running-statistics update after group k of this pass (keeps the reference's call order: sinskitG_model.py:1490-1584).
    The first job's first scale (put the full-resolution discriminator first) runs on the launch stream.
    A StyleGAN2 discriminator (`--netD stylegan2`) in `jobs` runs its passes on the launch stream after the lanes have joined."""
    sg_jobs = [j for j in jobs if getattr(j[0], "is_stylegan2_d", False)]
    jobs = [j for j in jobs if not getattr(j[0], "is_stylegan2_d", False)]
    if jobs:
        _msd_multi(jobs, criterion, extra, extra_cost, extra_main, streams)
    elif extra is not None:
        extra()
    for D, passes in sg_jobs:
        _sg2d_passes(D, passes, criterion)


def _sg2d_passes(D, passes, criterion):
    """the pass dictionaries of msd_multi for a single-output StyleGAN2 discriminator"""
    for p in passes:
        in0, in1 = p["in0"], p.get("in1")
        n, c0, h, w = in0.shape
        if in1 is not None:   # torch.cat([in0, in1], 1) as concat-on-store
            x = _empty(n, c0 + in1.shape[1], h, w, in0.device)
            ops.pad_affine(in0, (0, 0, 0, 0), 0, out=x[:, :c0], out_nstride=x.stride(0))
            ops.pad_affine(in1, (0, 0, 0, 0), 0, out=x[:, c0:], out_nstride=x.stride(0))
        else:
            x = in0
        gc = p.get("grad_coeff")
        y, ctx = sg2d_forward(D, x, keep=gc is not None and p.get("loss", True))
        pred = y.view(n, 1, 1, 1)
        p["preds"] = [pred]
        if not p.get("loss", True):
            continue
        g = criterion.accumulate([pred], p["real"], p["coeff"], p["slot"], grad_coeff=gc, want_grad=gc is not None)[0]
        if gc is None:
            continue
        want_in = p.get("input_grad") is not None
        dx = sg2d_backward(D, ctx, g, accumulate=p.get("accumulate", False), input_grad=want_in, param_grads=p.get("param_grads", True))
        if want_in:
            dst, acc = p["input_grad"]
            src = dx[:, c0:] if in1 is not None else dx
            dst.add_(src) if acc else dst.copy_(src)


KO_LANES = tuple(int(k) for k in tune.get("VTS_KO_LANES", "").split(",") if k)
if KO_LANES:      # timing experiment (tools/probes/r02_ko.sh): the named discriminator lanes are skipped, losses and gradients are WRONG
    import sys
    print("WARNING: VTS_KO_LANES=%s -- discriminator lanes are knocked out, this run's results are wrong (timing experiment only)"
          % tune.get("VTS_KO_LANES", ""), file=sys.stderr, flush=True)


LANE_STREAMS = int(tune.get("VTS_LANE_STREAMS", "4"))


def _lane_groups(costs, env="VTS_LANE_GROUPS", streams=None):
    """Which lanes share a stream.  The part runs at most FOUR hardware queues side by side (GPU_MAX_HW_QUEUES, default 4; with 5 - 8 the
    step takes 10 - 11 ms instead of 5.9: the queues beyond four are time-sliced), and a replayed graph maps its parallel branches onto
    them round-robin in capture order -- with one stream per lane (six or seven) WHICH lanes end up sharing a queue was an accident of
    the enqueue order (measured 5.87 .. 6.26 ms over permutations of the same lanes; round 4).  So the lanes are packed into
    VTS_LANE_STREAMS (4) streams here, longest-processing-time first on `costs` (ms estimates), the heavier lane of a stream first.
    VTS_LANE_GROUPS="0|1,5|3|2,4" (discriminator updates) / VTS_LANE_GROUPS_G (the generator step's passes, insensitive: 5.67 - 5.72 ms
    over eleven groupings) override the packing (measurement); VTS_LANE_STREAMS=0: one stream per lane.
    Measured and dropped on top of the packing (round 4): the launch-stream lane's weight gradients appended to the lightest other
    stream (5.95 vs 5.67 ms); side-queue items issued one item late, so that the launch chain's own edge leaves a fork node of the
    captured graph first (5.87 vs 5.67 ms); 2 / 3 / 5 / all side-queue items of the generator backward behind ONE wait on the launch
    stream (5.63 - 5.70 ms: no effect); the decoder lanes' weight gradients on queues of their own (+ 0.22 ms)."""
    n = len(costs)
    spec = tune.get(env, "")
    if spec:
        groups = [[int(t) for t in g.split(",") if t.strip() != "" and int(t) < n] for g in spec.split("|")]
        groups = [g for g in groups if g]
        seen = sorted(i for g in groups for i in g)
        if len(seen) != len(set(seen)):
            raise ValueError("VTS_LANE_GROUPS names a lane twice: %s" % spec)
        return groups + [[i] for i in range(n) if i not in seen]      # lanes the spec does not mention keep their own stream
    limit = LANE_STREAMS if streams is None else min(LANE_STREAMS, streams) if LANE_STREAMS > 0 else 0
    if limit <= 0 or n <= limit:
        return [[i] for i in range(n)]
    order = sorted(range(1, n), key=lambda i: -costs[i])
    bins, load = [[0]], [costs[0]]            # lane 0 (the caller puts its heaviest lane first) stays on the launch stream
    for i in order:
        if len(bins) < limit:
            bins.append([i])
            load.append(costs[i])
            continue
        k = min(range(len(bins)), key=lambda b: load[b])
        bins[k].append(i)
        load[k] += costs[i]
    return bins


def _lane_cost(passes, s):
    """rough time of one discriminator scale over its passes, ms: a latency floor (its ~25 dependent launches) + a term per input pixel"""
    px = 0
    for p in passes:
        a0 = p["_pyr"][0][0]          # (level 0: asking a lazily pooled pyramid for level s here would pool on the caller's stream)
        t = a0.data if isinstance(a0, Act) else a0
        h, w = t.shape[2], t.shape[3]
        for _ in range(s):
            h, w = (h + 1) // 2, (w + 1) // 2
        px += t.shape[0] * h * w
    return 0.35 + 1e-7 * px


LAZY_PYRAMID = tune.get("VTS_LAZY_PYRAMID", "1") != "0"


class _LanePyramid:
    """Input pyramid whose level s is pooled INSIDE the lane that asks for it (s average pools from the full-resolution level, on that
    lane's stream).  Pooling all levels up front (`_pyramid`) put 2 x (num_D - 1) launches on the serial stretch in front of every fork of
    the lanes -- the D2 full-resolution stack and the patch stacks: 67 us per step with nothing beside them; here scale 2 repeats scale 1's
    first pool on its own stream instead (same values bit for bit).  Serial schedule (PARALLEL_SCALES off): levels are shared."""

    def __init__(self, in0, in1):
        self.base = (_as_act(in0), _as_act(in1) if in1 is not None else None)
        self.levels = {0: self.base}

    def __getitem__(self, s):
        if s in self.levels:
            return self.levels[s]
        cur = self.base
        for t in range(1, s + 1):
            if t in self.levels:
                cur = self.levels[t]
                continue
            cur = (_pool_act(cur[0]), _pool_act(cur[1]))
            if not PARALLEL_SCALES:
                self.levels[t] = cur
        return cur


def _prepare_passes(jobs):
    """per-pass bookkeeping of msd_multi / msd_chain: the input pyramid (the caller's, or pooled inside the lanes), result slots"""
    for D, passes in jobs:
        for p in passes:
            if p.get("pyr"):
                p["_pyr"] = p["pyr"]                                          # the caller's precomputed input pyramid
            elif LAZY_PYRAMID:
                p["_pyr"] = _LanePyramid(p["in0"], p.get("in1"))
            else:
                p["_pyr"] = _pyramid(D, p["in0"], p.get("in1"))
            p["preds"] = [None] * D.num_D
            p["_din"] = [None] * D.num_D


def _finish_passes(jobs):
    for D, passes in jobs:
        for p in passes:
            if p.get("input_grad") is not None:
                p["coarse_grad"] = _merge_input_grads(p["_din"], p["input_grad"], defer_last=bool(p.get("defer_merge")))
            p.pop("_pyr"), p.pop("_din")
            if not p.get("keep_stats"):
                p.pop("_stats", None)


def _scale_lane(D, s, passes, criterion, knocked_out=False):
    """the passes of ONE scale of one multiscale discriminator, in order (the body of a lane of msd_multi / msd_chain)"""
    cache = {}      # packed weights of this scale: shared by its passes (they run in order in this lane)
    for p in passes:
        prep = p.get("prep")
        if prep and s in prep:
            prep[s]()       # per-scale preparation inside the lane (pooling of this scale's input level)
        a0, a1 = p["_pyr"][s]
        if knocked_out:   # timing experiment only (VTS_KO_LANES; results are wrong): what a lane costs on the step's critical path
            p["preds"][s] = torch.zeros(1, 1, 1, 1, device=a0.data.device)
            if p.get("input_grad") is not None:
                p["_din"][s] = torch.zeros_like((a1 if a1 is not None else a0).data)
            continue
        ig = p.get("input_grad")
        gsrc = a1 if a1 is not None else a0
        into = (ig[0], ig[1]) if (ig is not None and s == 0 and ig[0].shape == gsrc.data.shape and ig[0].is_contiguous()) else None
        groups = p.get("groups")
        gstarts = [gr["n0"] for gr in groups] if groups else None
        stat_rec = p.setdefault("_stats", {}).setdefault(s, {}) if p.get("stat_only") else None
        src = p.get("ext_from")
        if KO_LANES and src is not None:
            src.setdefault("_stats", {}).setdefault(s, {})
        ext = (src["_stats"][s], p.get("ext_after", 0)) if src is not None else None
        if KO_LANES and src is not None and not ext[0]:      # (timing experiment: the early pass of this lane was knocked out too)
            ext = None
        # pred_scales: the scales whose prediction map the caller reads, for a pass without loss (the full-resolution D2 visualisation pass
        # shows the coarsest scale only; at the other scales it exists for the BatchNorm running statistics: their head is not run)
        skip_head = not p.get("loss", True) and p.get("pred_scales") is not None and s not in p["pred_scales"]
        forward_only = not groups and ext is None and (not p.get("loss", True) or p.get("grad_coeff") is None)
        # (per-launch timing and label knock-outs need the single launches: they take the Python schedule)
        if forward_only and ops.TIMER is None and not ops.KNOCKOUT and patchgan_c_ok(D, a0.data.shape[2], a0.data.shape[3]):
            # nothing of this pass is needed again: the whole scale as ONE call of the network-level C entry (vts_patchgan_forward)
            acts = None
            pred = patchgan_forward_c(D, s, a0, a1, not p.get("stat_only", False), stat_rec, run_head=not skip_head)
        else:
            acts = _msd_scale_forward(D, s, a0, a1, not p.get("stat_only", False), cache, gstarts, stat_rec, ext, skip_head=skip_head)
            pred = None if skip_head else acts[-1].data
        if skip_head:
            continue
        p["preds"][s] = pred
        if groups:
            want = any(gr.get("grad_coeff") is not None for gr in groups)
            g = torch.empty_like(pred) if want else None
            for gr in groups:
                gc = gr.get("grad_coeff")
                if want and gc is None:
                    g[gr["n0"]:gr["n1"]].zero_()
                criterion.accumulate([pred[gr["n0"]:gr["n1"]]], gr["real"], gr["coeff"], gr["slot"], grad_coeff=gc,
                                     want_grad=gc is not None, out_grads=[g[gr["n0"]:gr["n1"]]] if gc is not None else None,
                                     pre_sigmoid=getattr(D, "use_sigmoid", False))
            if want:
                p["_din"][s] = _msd_scale_backward(D, s, a0, a1, acts, g, p.get("param_grads", True), p.get("accumulate", False),
                                                   p.get("input_grad") is not None, cache, gstarts, into=into)
            continue
        if not p.get("loss", True):
            continue
        gc = p.get("grad_coeff")
        g = criterion.accumulate([pred], p["real"], p["coeff"], p["slot"], grad_coeff=gc, want_grad=gc is not None,
                                 pre_sigmoid=getattr(D, "use_sigmoid", False))[0]
        if gc is not None:
            p["_din"][s] = _msd_scale_backward(D, s, a0, a1, acts, g, p.get("param_grads", True), p.get("accumulate", False),
                                               p.get("input_grad") is not None, cache, into=into)


def _msd_multi(jobs, criterion, extra=None, extra_cost=0.1, extra_main=False, streams=None):
    lanes = []
    _prepare_passes(jobs)
    for D, passes in jobs:
        for s in range(D.num_D):
            lanes.append((D, s, passes))
    costs = [_lane_cost(passes, s) for _, s, passes in lanes]
    if extra is not None and extra_main:      # the extra work IS the launch-stream lane (it may fork lanes of its own: the generator forward)
        lanes.insert(0, (None, -1, extra))
        costs.insert(0, float(extra_cost))
    elif extra is not None:
        lanes.append((None, -1, extra))
        costs.append(float(extra_cost))

    def lane(i):
        D, s, passes = lanes[i]
        if D is None:
            passes()
            return
        _scale_lane(D, s, passes, criterion, knocked_out=i in KO_LANES)

    nograd = all(not p.get("param_grads", True) for _, passes in jobs for p in passes)     # the generator step's passes
    groups = _lane_groups(costs, "VTS_LANE_GROUPS_G" if nograd else "VTS_LANE_GROUPS", streams)
    with ops.deferred_wgrad():    # one reduction launch for the weight-gradient partials of all lanes, after they have joined
        _run_lanes(len(groups), lambda gi: [lane(i) for i in groups[gi]])
    _finish_passes(jobs)


D_CHAINS = tune.get("VTS_D_CHAINS", "1") != "0"


def msd_chain(chain, criterion, side=None, side_cost=0.1, serial=False):
    """One discriminator's part of a training step as ONE dependency chain:
        update passes (its scales side by side)  ->  `mid` (its optimiser step)  ->  its passes of the generator step.
    msd_multi joins the lanes of ALL discriminators between those stages (and the stages were separate graphs): the step waited for the
    slowest lane of the update phase, ran the weight-gradient reductions and both Adam launches alone on the chip, and forked again --
    although D1's optimiser step needs only D1's lanes, and the generator's backward only D1's chain (the D2 term of the generator is a
    logged value).  The chain's scales share TWO streams: the launch stream (the chain's join points, `mid`, the merge of the input
    gradients) and one side stream.
    serial=True: everything on the current stream -- for a chain that the caller runs inside a lane of its own (fork_lane).
    MEASURED (round 6): a chain led by a SIDE stream needs side <-> side waits (its own join in front of its optimiser step), and
    hipStreamEndCapture crashes on those exactly as on a fork from a side stream: only the chain on the launch stream can have two streams.

    chain: dict(D=, index0=<lane number of scale 0, for VTS_KO_LANES>, update=[passes], mid=callable, gstep=callable -> [passes])
    (pass dictionaries as for msd_multi).  side: callable that runs as one more lane inside the chain's stream pair and is complete before
    `mid` (the generator's L1 terms: the D1 pass of the generator step accumulates onto their gradient)."""
    global _LANE_BASE
    D = chain["D"]
    _prepare_passes([(D, chain["update"])])

    def stage_lanes(passes):
        """[(cost, body)] per scale"""
        return [(_lane_cost(passes, s), (lambda s=s: _scale_lane(D, s, passes, criterion, knocked_out=(chain.get("index0", 0) + s) in KO_LANES)))
                for s in range(D.num_D)]

    def split2(items):
        """two bins, longest first, the heaviest item first on bin 0 (the launch stream)"""
        order = sorted(range(len(items)), key=lambda i: -items[i][0])
        bins, load = ([], []), [0.0, 0.0]
        for i in order:
            k = 0 if load[0] <= load[1] else 1
            bins[k].append(items[i][1])
            load[k] += items[i][0]
        return bins

    if serial or not PARALLEL_SCALES:
        with ops.deferred_wgrad():
            for _, body in stage_lanes(chain["update"]):
                body()
            if side is not None:
                side()
            ops.wgrad_flush(ops.WS_LANE)
            _finish_passes([(D, chain["update"])])
            chain["mid"]()
            g = chain["gstep"]()
            _prepare_passes([(D, g)])
            for _, body in stage_lanes(g):
                body()
            _finish_passes([(D, g)])
        return

    base = _LANE_BASE
    main = torch.cuda.current_stream()
    side_streams = _SIDE_STREAMS.setdefault(torch.cuda.current_device(), [])
    while len(side_streams) < base + 1:
        side_streams.append(torch.cuda.Stream())
    second = side_streams[base]
    ws0 = ops.WS_LANE

    def on_second(fn):
        ops.WS_LANE = base + 1
        try:
            with torch.cuda.stream(second):
                fn()
        finally:
            ops.WS_LANE = ws0

    _LANE_BASE = base + 1
    try:
        with ops.deferred_wgrad():
            second.wait_stream(main)
            items = stage_lanes(chain["update"])
            if side is not None:
                items.append((float(side_cost), side))
            b_main, b_second = split2(items)
            on_second(lambda: ([b() for b in b_second], ops.wgrad_flush(base + 1)))
            for b in b_main:
                b()
            ops.wgrad_flush(ws0)
            main.wait_stream(second)                  # the chain's own join, its optimiser step
            _finish_passes([(D, chain["update"])])
            chain["mid"]()
            g = chain["gstep"]()
            _prepare_passes([(D, g)])
            second.wait_stream(main)
            b_main, b_second = split2(stage_lanes(g))
            on_second(lambda: [b() for b in b_second])
            for b in b_main:
                b()
            main.wait_stream(second)
            _finish_passes([(D, g)])
    except BaseException:
        ops.WS_LANE = ws0
        ops.wgrad_discard()
        raise
    finally:
        _LANE_BASE = base
        main.wait_stream(second)                # joined back in every case (a capture must not end with a dangling fork)


# ======================================================================================================================
# StyleGAN2 discriminator (`--netD stylegan2`; reference models/stylegan_networks.py:696-786)
# The equalised-learning-rate scale 1/sqrt(fan_in) (:166, :214) is an operand affine of the conv kernels (normalise-on-load),
# ResBlock's 1/sqrt 2 (:691) is folded into conv2's activation gain and into the skip convolution's operand scale, Blur is
# vts_upfirdn2d, the stride-2 3x3 / 1x1 convolutions run on the stride-2 4x4 kernels (ops.convk_s2), EqualLinear(C*16, C) on the
# flattened 4x4 map is a valid 4x4 convolution.
# ======================================================================================================================
_CONST = {}
SQRT2 = 2.0 ** 0.5


def _const(n, v, dev):
    key = (n, float(v), str(dev))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((n,), float(v), dtype=torch.float32, device=dev)
    return t


def _scaled(t, s):
    """operand t * s (per-(n, c) affine with a constant scale)"""
    nc = t.shape[0] * t.shape[1]
    return Act(t, _const(nc, s, t.device), _const(nc, 0.0, t.device))


def _sg_layer_forward(m, x, gain=SQRT2, res=None, extra_scale=1.0):
    """ConvLayer (:622-668).  Returns (output, saved) with saved = (blurred input or x, pre-activation z, operand scale)"""
    from models.stylegan2_blocks import BLUR_KERNEL
    conv, act = m.conv[0], m.act[0]
    n = x.shape[0]
    t = ops.upfirdn2d(x, BLUR_KERNEL, pad=m.blur_pad) if m.downsample else x
    s = extra_scale / (m.cin * m.k * m.k) ** 0.5
    if m.downsample:
        oh, ow = (t.shape[2] - m.k) // 2 + 1, (t.shape[3] - m.k) // 2 + 1
        z = _empty(n, m.cout, oh, ow, x.device)
        ops.convk_s2(_scaled(t, s), conv.weight, z, bias=getattr(conv, "bias", None))
    else:
        z = _empty(n, m.cout, t.shape[2], t.shape[3], x.device)
        ops.convk(_scaled(t, s), conv.weight, z, bias=getattr(conv, "bias", None), pad=m.k // 2)
    if m.activate:
        y = ops.bias_act(z, act.bias if act is not None else None, 0.2, gain, res=res)
    else:
        assert res is None
        y = z
    return y, (t, z, s)


def _sg_layer_backward(m, x_shape, saved, g, gain, accumulate, want_dx=True, dx=None, dx_accumulate=False, param_grads=True):
    """backward of _sg_layer_forward: parameter gradients into .grad, returns the gradient w.r.t. the layer input"""
    from models.stylegan2_blocks import BLUR_KERNEL
    conv, act = m.conv[0], m.act[0]
    t, z, s = saved
    if m.activate:
        dz = ops.bias_act_bwd(g, z, act.bias if act is not None else None, 0.2, gain)
        if act is not None and param_grads:
            ops.channel_sum(dz, act.bias.grad.view(-1), accumulate=accumulate)
    else:
        dz = g
        if getattr(conv, "bias", None) is not None and param_grads:
            ops.channel_sum(dz, conv.bias.grad, accumulate=accumulate)
    if not param_grads:
        pass
    elif m.downsample:
        ops.wgradk_s2(dz, _scaled(t, s), conv.weight.grad, accumulate=accumulate)
    else:
        ops.wgradk(dz, _scaled(t, s), conv.weight.grad, pad=m.k // 2, accumulate=accumulate)
    if not want_dx:
        return None
    if m.downsample:
        dt = torch.empty_like(t)
        ops.convk_s2_bwd_data(_scaled(dz, s), conv.weight, dt)
        if dx is None:
            dx = torch.empty(x_shape, dtype=torch.float32, device=g.device)
        ops.upfirdn2d_bwd(dt, dx, BLUR_KERNEL, pad=m.blur_pad, accumulate=dx_accumulate)
    else:
        if dx is None:
            dx = torch.empty(x_shape, dtype=torch.float32, device=g.device)
        ops.convk_bwd_data(_scaled(dz, s), conv.weight, dx, pad=m.k // 2, accumulate=dx_accumulate)
    return dx


class Sg2dCtx:
    __slots__ = ("x_shape", "first", "blocks", "final", "lin")


def sg2d_forward(D, x, keep=True):
    """StyleGAN2Discriminator.forward (:755-786) for netD = 'stylegan2'.  x [N, C, size, size] -> ([N, 1], ctx)"""
    n, dev = x.shape[0], x.device
    ctx = Sg2dCtx()
    ctx.x_shape = tuple(x.shape)
    y, ctx.first = _sg_layer_forward(D.convs[0], x)
    ctx.blocks = []
    for blk in list(D.convs)[1:]:
        y1, s1 = _sg_layer_forward(blk.conv1, y)
        sk, ss = _sg_layer_forward(blk.skip, y, extra_scale=1.0 / SQRT2)
        out, s2 = _sg_layer_forward(blk.conv2, y1, gain=1.0, res=sk)     # sqrt 2 (activation gain) / sqrt 2 (residual merge)
        ctx.blocks.append((tuple(y.shape), s1, tuple(y1.shape), s2, ss))
        y = out
    yf, ctx.final = _sg_layer_forward(D.final_conv, y)
    l0, l1 = D.final_linear[0], D.final_linear[1]
    c4 = l0.weight.shape[0]
    assert yf.shape[2:] == (4, 4), "the discriminator is built for inputs of its `size`"
    z0 = _empty(n, c4, 1, 1, dev)
    s0 = 1.0 / (c4 * 16) ** 0.5
    ops.conv4x4(_scaled(yf, s0), l0.weight, c4 * 16, 16, c4, z0, stride=1, pad=0)         # EqualLinear(C*16 -> C) :199-227
    a0 = ops.bias_act(z0, l0.bias, 0.2, SQRT2)
    s1l = 1.0 / c4 ** 0.5
    out = _empty(n, 1, 1, 1, dev)
    ops.convk(_scaled(a0, s1l), l1.weight.view(1, c4, 1, 1), out, bias=l1.bias, pad=0)
    ctx.lin = (tuple(y.shape), yf, z0, a0, s0, s1l)
    if not keep:
        ctx = None
    return out.view(n, 1), ctx


def sg2d_backward(D, ctx, dout, accumulate=False, input_grad=False, param_grads=True):
    """gradients of all parameters into .grad (param_grads); returns d/dx when input_grad"""
    l0, l1 = D.final_linear[0], D.final_linear[1]
    c4 = l0.weight.shape[0]
    y_shape, yf, z0, a0, s0, s1l = ctx.lin
    n = dout.shape[0]
    g = dout.reshape(n, 1, 1, 1).contiguous()
    pg = param_grads
    if pg:
        ops.channel_sum(g, l1.bias.grad, accumulate=accumulate)
        ops.wgradk(g, _scaled(a0, s1l), l1.weight.grad.view(1, c4, 1, 1), pad=0, accumulate=accumulate)
    da0 = torch.empty_like(a0)
    ops.convk_bwd_data(_scaled(g, s1l), l1.weight.view(1, c4, 1, 1), da0, pad=0)
    dz0 = ops.bias_act_bwd(da0, z0, l0.bias, 0.2, SQRT2)
    if pg:
        ops.channel_sum(dz0, l0.bias.grad, accumulate=accumulate)
        ops.wgrad4x4(dz0, _scaled(yf, s0), l0.weight.grad, stride=1, pad=0, accumulate=accumulate)
    dyf = torch.empty_like(yf)
    ops.conv4x4(_scaled(dz0, s0), l0.weight, 16, c4 * 16, c4, dyf, stride=1, pad=0, transposed=True)
    g = _sg_layer_backward(D.final_conv, y_shape, ctx.final, dyf, SQRT2, accumulate, param_grads=pg)
    blocks = list(D.convs)[1:]
    for blk, (x_shape, s1, y1_shape, s2, ss) in zip(reversed(blocks), reversed(ctx.blocks)):
        dy1 = _sg_layer_backward(blk.conv2, y1_shape, s2, g, 1.0, accumulate, param_grads=pg)
        dx = _sg_layer_backward(blk.skip, x_shape, ss, g, 1.0, accumulate, param_grads=pg)
        g = _sg_layer_backward(blk.conv1, x_shape, s1, dy1, SQRT2, accumulate, dx=dx, dx_accumulate=True, param_grads=pg)
    return _sg_layer_backward(D.convs[0], ctx.x_shape, ctx.first, g, SQRT2, accumulate, want_dx=input_grad, param_grads=pg)


def modulated_conv2d(x, style, weight, mod_weight, mod_bias, demodulate=True, downsample=False):
    """ModulatedConv2d.forward (:304-348), forward only, for the plain and the downsampling form (the upsampling form -- a
    transposed convolution followed by Blur -- is not built).  Instead of the reference's per-sample weights in a grouped
    convolution: y = conv(x * s[n,ci] / sqrt(fan_in), W) * demod[n,co] on the shared weight W = weight[0]; both factors ride on the
    operand affines.  style [N, style_dim]; the modulation is EqualLinear(style_dim, Ci, bias_init=1) (:199-227)."""
    from models.stylegan2_blocks import BLUR_KERNEL
    n, ci, h, w_ = x.shape
    wt = weight.view(weight.shape[-4], ci, weight.shape[-2], weight.shape[-1])
    co, k = wt.shape[0], wt.shape[2]
    sd = style.shape[1]
    s = _empty(n, ci, 1, 1, x.device)
    ops.convk(_scaled(style.reshape(n, sd, 1, 1).contiguous(), 1.0 / sd ** 0.5), mod_weight.view(ci, sd, 1, 1), s, bias=mod_bias, pad=0)
    scale = 1.0 / (ci * k * k) ** 0.5
    xin = Act(x, (s.view(-1) * scale).contiguous(), _const(n * ci, 0.0, x.device))
    if downsample:
        p = 2 + (k - 1)
        t = ops.pad_affine(xin, (0, 0, 0, 0), 0)                      # materialise x * s, then Blur (:323-327), then stride 2
        t = ops.upfirdn2d(t, BLUR_KERNEL, pad=((p + 1) // 2, p // 2))
        out = _empty(n, co, (t.shape[2] - k) // 2 + 1, (t.shape[3] - k) // 2 + 1, x.device)
        ops.convk_s2(t, wt, out)
    else:
        out = _empty(n, co, h, w_, x.device)
        ops.convk(xin, wt, out, pad=k // 2)
    if not demodulate:
        return out
    demod = ops.modconv_demod(wt, s.view(n, ci), scale)
    return ops.pad_affine(Act(out, demod.view(-1), _const(n * co, 0.0, x.device)), (0, 0, 0, 0), 0)


# ---- generator side (`--netG stylegan2 | smallstylegan2`; reference stylegan_networks.py:800-930) ------------------------------------
def _sg_resblock_forward(blk, y):
    """ResBlock :671-693 with skip_gain 1: (conv2(conv1(x)) + skip(x)) / sqrt 2; the 1 / sqrt 2 rides on conv2's activation gain and on
    the skip operand (identity skip: one scaled copy)"""
    y1, s1 = _sg_layer_forward(blk.conv1, y)
    if blk.skip is not None:
        sk, ss = _sg_layer_forward(blk.skip, y, extra_scale=1.0 / SQRT2)
    else:
        sk, ss = ops.pad_affine(_scaled(y, 1.0 / SQRT2), (0, 0, 0, 0), 0), None
    out, s2 = _sg_layer_forward(blk.conv2, y1, gain=1.0, res=sk)
    return out, (tuple(y.shape), s1, tuple(y1.shape), s2, ss)


def _sg_resblock_backward(blk, saved, g, accumulate, pg=True):
    x_shape, s1, y1_shape, s2, ss = saved
    dy1 = _sg_layer_backward(blk.conv2, y1_shape, s2, g, 1.0, accumulate, param_grads=pg)
    if blk.skip is not None:
        dx = _sg_layer_backward(blk.skip, x_shape, ss, g, 1.0, accumulate, param_grads=pg)
    else:
        dx = ops.pad_affine(_scaled(g, 1.0 / SQRT2), (0, 0, 0, 0), 0)
    return _sg_layer_backward(blk.conv1, x_shape, s1, dy1, SQRT2, accumulate, dx=dx, dx_accumulate=True, param_grads=pg)


def _k4():
    from models.stylegan2_blocks import BLUR_KERNEL
    return [[4.0 * v for v in row] for row in BLUR_KERNEL]


def _styled_up_forward(m, x, noise=None):
    """StyledConv(upsample=True) with style None (:399-407 -> ModulatedConv2d :304-330 -> Blur -> NoiseInjection -> FusedLeakyReLU):
    the demodulated weight depends on the parameters only (vts_modconv_weight); the transposed stride-2 convolution is the input
    adjoint of the stride-2 K x K convolution the library already has (ops.convk_s2_bwd_data), then Blur with the 4x kernel, pad (1, 1)."""
    n, ci, h, w = x.shape
    wt = ops.modconv_weight(m.conv.weight, transpose=True)            # [Ci, Co, 3, 3]
    pre = _empty(n, m.cout, 2 * h + 1, 2 * w + 1, x.device)
    ops.convk_s2_bwd_data(x, wt, pre)
    z = ops.upfirdn2d(pre, _k4(), pad=(1, 1))
    if m.inject_noise:
        # image + weight * noise (:358-363); the reference draws fresh N(0, 1) noise per call -- the draw is an input here (tests pass it)
        if noise is None:
            noise = torch.randn(n, 1, z.shape[2], z.shape[3], device=x.device)
        z = ops.pad_affine(z, (0, 0, 0, 0), 0, res=(noise * m.noise.weight).expand(-1, m.cout, -1, -1).contiguous())
    y = ops.bias_act(z, m.activate.bias, 0.2, SQRT2)
    return y, (x, wt, z, noise)


def _styled_up_backward(m, saved, g, accumulate, pg=True, want_dx=True):
    x, wt, z, noise = saved
    dz = ops.bias_act_bwd(g, z, m.activate.bias, 0.2, SQRT2)
    n, co = dz.shape[0], dz.shape[1]
    if pg:
        ops.channel_sum(dz, m.activate.bias.grad.view(-1), accumulate=accumulate)
        if m.inject_noise:
            gw = (dz.sum(1, keepdim=True) * noise).sum().view(1)       # one scalar parameter: d weight = <dz summed over channels, noise>
            m.noise.weight.grad.copy_(m.noise.weight.grad + gw if accumulate else gw)
    dpre = _empty(n, co, 2 * x.shape[2] + 1, 2 * x.shape[3] + 1, x.device)
    ops.upfirdn2d_bwd(dz, dpre, _k4(), pad=(1, 1))
    if pg:
        dwt = torch.empty_like(wt)
        ops.wgradk_s2(x, dpre, dwt)                                    # weight gradient of the stride-2 convolution pairing (dpre -> x)
        ops.modconv_weight_bwd(m.conv.weight, dwt, m.conv.weight.grad, transpose=True, accumulate=accumulate)
    if not want_dx:
        return None
    dx = torch.empty_like(x)
    ops.convk_s2(dpre, wt, dx)
    return dx


class Sg2gCtx:
    __slots__ = ("x_shape", "first", "enc", "dec", "ups", "last")


def sg2g_forward(G, x, keep=True, noises=None):
    """StyleGAN2Generator.forward (:922-930): encoder -> decoder.  x [N, C, size, size] -> ([N, 3, size, size], ctx)"""
    ctx = Sg2gCtx()
    ctx.x_shape = tuple(x.shape)
    enc, dec = list(G.encoder.convs), list(G.decoder.convs)
    y, ctx.first = _sg_layer_forward(enc[1], x)
    ctx.enc, ctx.dec, ctx.ups = [], [], []
    for blk in enc[2:]:
        y, sv = _sg_resblock_forward(blk, y)
        ctx.enc.append(sv)
    nb = G.n_blocks // 2
    for blk in dec[:nb]:
        y, sv = _sg_resblock_forward(blk, y)
        ctx.dec.append(sv)
    for i, m in enumerate(dec[nb:-1]):
        y, sv = _styled_up_forward(m, y, None if noises is None else noises[i])
        ctx.ups.append(sv)
    shape = tuple(y.shape)
    out, sv = _sg_layer_forward(dec[-1], y)
    ctx.last = (shape, sv)
    return out, (ctx if keep else None)


def sg2g_backward(G, ctx, dout, accumulate=False, input_grad=False):
    """gradients of all parameters into .grad; returns d/dx when input_grad"""
    enc, dec = list(G.encoder.convs), list(G.decoder.convs)
    nb = G.n_blocks // 2
    shape, sv = ctx.last
    g = _sg_layer_backward(dec[-1], shape, sv, dout, SQRT2, accumulate)
    for m, sv in zip(reversed(dec[nb:-1]), reversed(ctx.ups)):
        g = _styled_up_backward(m, sv, g, accumulate)
    for blk, sv in zip(reversed(dec[:nb]), reversed(ctx.dec)):
        g = _sg_resblock_backward(blk, sv, g, accumulate)
    for blk, sv in zip(reversed(enc[2:]), reversed(ctx.enc)):
        g = _sg_resblock_backward(blk, sv, g, accumulate)
    return _sg_layer_backward(enc[1], ctx.x_shape, ctx.first, g, SQRT2, accumulate, want_dx=input_grad)


# ======================================================================================================================
# SIFID (reference models/sifid.py:205-233, models/inception.py:57-67, models/model_utils.py:481-488, 541-555): Inception-v3 block 0
# on the convolution kernels of this library + the Frechet distance of the activation statistics, per image
# ======================================================================================================================
def inception_block0(net, x):
    """x [N,3,H,W] (network input, i.e. after InceptionV3.forward's 2x - 1) -> [N,64,h,w] features.  BatchNorm (eval, eps 0.001) and
    ReLU of a layer are applied by the next convolution on load; the last layer's are materialised."""
    n, dev = x.shape[0], x.device
    cur, act = x, 0
    for m, (ci, co, stride, pad) in zip(net.blocks[0], _INCEPTION_SPEC):
        h, w = (cur.data if isinstance(cur, Act) else cur).shape[2:]
        oh, ow = (h + 2 * pad - 3) // stride + 1, (w + 2 * pad - 3) // stride + 1
        out = _empty(n, co, oh, ow, dev)
        if stride == 2:
            ops.convk_s2(cur, m.conv.weight, out, act_in=act)
        else:
            ops.convk(cur, m.conv.weight, out, pad=pad, act_in=act)
        sc = m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)      # [C] vectors: plumbing, not arithmetic on activations
        sh = m.bn.bias - m.bn.running_mean * sc
        cur, act = Act(out, sc.repeat(n).contiguous(), sh.repeat(n).contiguous()), RELU
    return ops.pad_affine(cur, (0, 0, 0, 0), 0, act=RELU)


_INCEPTION_SPEC = ((3, 32, 2, 0), (32, 32, 1, 0), (32, 64, 1, 1))   # (cin, cout, stride, padding): torchvision's Conv2d_1a / 2a / 2b


def sifid_pairs(net, a, b, batch=16):
    """per-image SIFID of two stacks of network inputs [N,3,H,W] (calculate_sifid_given_arrays: the statistics are taken over the
    positions of ONE image's feature map): device tensor [N]"""
    n = a.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    for i0 in range(0, n, batch):
        fa, fb = inception_block0(net, a[i0:i0 + batch]), inception_block0(net, b[i0:i0 + batch])
        for j in range(fa.shape[0]):
            out[i0 + j:i0 + j + 1].copy_(ops.frechet_distance(fa[j].reshape(fa.shape[1], -1), fb[j].reshape(fb.shape[1], -1)))
    return out


def sifid_images(net, real_I, fake_I):
    """I_SIFID (model_utils.py:481-488): both images min-max normalised by the REAL image's range, the fake one clamped; mean over images"""
    lohi = ops.minmax(real_I)
    a = ops.sifid_input(real_I, 0, 3, lohi=lohi)
    b = ops.sifid_input(fake_I, 0, 3, lohi=lohi, clamp01=True)
    return sifid_pairs(net, a, b, batch=1).mean()


def sifid_tactile(net, real_T, fake_T, size=299):
    """T_SIFID (model_utils.py:541-555): fake patches clamped to (0, 1) (:519), nearest resize to 299 x 299, gx and gy each tiled to
    three channels, per-patch SIFID, mean of (gx + gy) / 2"""
    vals = []
    for c in (0, 1):
        a = ops.sifid_input(real_T, c, 1, size=(size, size))
        b = ops.sifid_input(fake_T, c, 1, size=(size, size), clamp01=True)
        vals.append(sifid_pairs(net, a, b))
    return ((vals[0] + vals[1]) * 0.5).mean()
