"""Thin functional wrappers over the C ABI (one Python function per vts_* entry point).

`Act` is the lazily-normalised activation the kernels consume: raw data plus the
per-(n,c) scale/shift produced by vts_norm_stats (see include/vts.h, vts_operand).
"""
import ctypes as C
import os
from . import tune

import torch

from . import lib as L


class Act:
    """Raw NCHW tensor + optional per-(n,c) affine (normalise-on-load) + saved mean/rstd."""

    __slots__ = ("data", "scale", "shift", "mean", "rstd", "padded")

    def __init__(self, data, scale=None, shift=None, mean=None, rstd=None):
        self.data, self.scale, self.shift, self.mean, self.rstd = data, scale, shift, mean, rstd
        self.padded = None   # the materialised activate(normalise(data)) with zero padding, when a GEMM-class consumer made one

    @property
    def shape(self):
        return self.data.shape

    def operand(self):
        return L.operand(self.data, self.scale, self.shift)


# When TIMER is a list, every wrapper brackets its launch with HIP events recorded on the launch
# stream and appends (label, algorithmic_bytes, algorithmic_flops, start, end).  bench.py uses this
# for its live per-kernel roofline; it is off (None) on the product path.
TIMER = None


DETAIL = None  # shape string of the launch being issued (only filled while TIMER is on)


# timing experiments only (results are wrong): VTS_KNOCKOUT=norm_stats,wgrad4x4 skips every launch whose label starts with one of
# the prefixes -- what a category of kernels contributes to the critical path of the laned step
KNOCKOUT = tuple(k for k in tune.get("VTS_KNOCKOUT", "").split(",") if k)


def _run(label, nbytes, flops, fn, *args):
    global DETAIL
    if KNOCKOUT and label.startswith(KNOCKOUT):
        return
    if TIMER is None:
        L.check(fn(*args), label)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    L.check(rc, label)
    if label.startswith(("conv4x4", "wgrad4x4", "patch_conv4x4", "patch_wgrad4x4", "norm_", "conv3x3_wide", "wgrad3x3_wide")):
        label = L.load().vts_last_kernel().decode()   # the exact kernel instance, as rocprofv3 names it
    TIMER.append((label, nbytes, flops, e0, e1, DETAIL))
    DETAIL = None


def _numel(op):
    return 0


_ws = {}


def workspace(nfloats, device):
    """Grow-only scratch; safe to share because every kernel runs in stream order."""
    key = (str(device), WS_LANE)   # one scratch per concurrent lane (engine._run_lanes / SideQueue)
    t = _ws.get(key)
    if t is None or t.numel() < nfloats:
        if t is not None and frozen_ws():
            _retired.append(t)   # captured HIP graphs hold the old pointer: keep that buffer alive, never reuse it
        t = torch.empty(max(int(nfloats), 2 * (t.numel() if t is not None else 0), 1 << 20), dtype=torch.float32, device=device)
        _ws[key] = t
    return t


_retired = []

_counters = {}
# Off by default: measured 3x SLOWER on MI355X (10.5 -> 30 ms per step).  The release / acquire fences the pattern needs are
# device-scope, and with one L2 per XCD a device-scope fence writes back / invalidates that whole L2 -- thousands of workgroups doing
# so while six other lanes keep the L2s dirty is far more expensive than the ~140 tiny finalize launches it saves.  A kernel boundary
# is the cheap cross-XCD synchronisation on this part.  (VTS_FUSE_FINALIZE=1 turns it on; tests/test_kernels_gpu.py covers both.)
FUSE_FINALIZE = tune.get("VTS_FUSE_FINALIZE", "0") == "1"


def counters(device):
    """zeroed int32 scratch of the current lane for the 'last workgroup finalises' kernels (they leave it zero); None switches
    the library back to its separate finalize launches"""
    if not FUSE_FINALIZE:
        return None
    key = (str(device), WS_LANE)
    t = _counters.get(key)
    if t is None:
        t = _counters[key] = torch.zeros(1 << 16, dtype=torch.int32, device=device)
    return t


# The workspace is frozen (a grown buffer is retired, never freed) while any captured HIP graph that holds its pointer is alive.
# Ownership is tracked per holder (a model's training graphs, its inference graph, another model's graphs ...): a bare boolean was
# cleared by whichever model dropped its graphs first while another still replayed graphs with the old pointer.
_WS_HOLDERS = set()


def freeze_ws(owner):
    _WS_HOLDERS.add(owner)


def release_ws(owner):
    _WS_HOLDERS.discard(owner)


def frozen_ws():
    return bool(_WS_HOLDERS)


WS_LANE = 0        # which scratch buffer the wrappers use: 0 = the launch stream, i = side lane i (engine._run_lanes)


def _op(a):
    if a is None:
        return L.Operand(None, None, None, 0, 0)
    if isinstance(a, Act):
        return a.operand()
    if isinstance(a, L.Operand):
        return a
    return L.operand(a)


BWD_SUMS = tune.get("VTS_BWD_SUMS", "1") != "0"
BSUMS = {}       # data_ptr of a gradient tensor -> (partials, slots, the tensor): sums its producing convolution left for norm_bwd

_stat_ws = {}


def stat_workspace(nfloats, device):
    """scratch for the epilogue statistics partials of vts_conv4x4_norm: one per lane, separate from `workspace` (the k-split
    partials of the same call live there)"""
    key = (str(device), WS_LANE)
    t = _stat_ws.get(key)
    if t is None or t.numel() < nfloats:
        if t is not None and frozen_ws():
            _retired.append(t)
        t = _stat_ws[key] = torch.empty(max(int(nfloats), 2 * (t.numel() if t is not None else 0), 1 << 18), dtype=torch.float32, device=device)
    return t


def conv4x4(in0, w, ws_co, ws_ci, cout, out, *, in1=None, bias=None, stride=2, pad=1, transposed=False, act_in=0,
            act_out=0, dmask=None, dmask_act=0, accumulate=False, out_nstride=None, in_hw=None, pad_dx=0, instance_norm=False,
            batch_norm=None, bwd_sums=False):
    """out <- conv-family(in0 ++ in1).  `w` may be an offset view into a weight tensor.
    instance_norm=True / batch_norm=dict(keyword arguments of norm_stats: gamma, beta, running_mean, ..., groups, stat_out, ext):
    returns Act(out, scale, shift, mean, rstd) of InstanceNorm2d(out) / training-mode BatchNorm2d(out) -- through vts_conv4x4_norm, which
    takes the statistics from the convolution's epilogue (tiled kernel) or its k-split epilogue (inner U-Net layers) where it can,
    and otherwise runs vts_norm_stats afterwards."""
    lib = L.load()
    d = L.ConvDesc()
    d.in0, d.in1 = _op(in0), _op(in1)
    x = in0.data if isinstance(in0, Act) else in0
    d.N = x.shape[0]
    d.IH, d.IW = in_hw if in_hw else (x.shape[2], x.shape[3])
    d.OH, d.OW = out.shape[2], out.shape[3]
    d.Cout = cout
    d.stride, d.pad, d.transposed = stride, pad, int(transposed)
    d.pad_dx = pad_dx
    d.w, d.ws_co, d.ws_ci = w.data_ptr(), ws_co, ws_ci
    d.bias = L.ptr(bias)
    d.out = out.data_ptr()
    d.out_nstride = out.stride(0) if out_nstride is None else out_nstride
    d.act_in, d.act_out = act_in, act_out
    d.dmask = _op(dmask)
    d.dmask_act = dmask_act
    d.accumulate = int(accumulate)
    if d.OH * d.OW <= 64 * 64:  # only small maps can take the k-split path; scratch is the shared workspace
        need = lib.vts_conv4x4_ws_floats(C.byref(d))
        ws = workspace(need, x.device)
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    cin = d.in0.C + d.in1.C
    taps = 4 if (transposed and stride == 2) else 16
    flops = 2.0 * d.N * d.OH * d.OW * cout * cin * taps
    nbytes = 4.0 * (d.N * cin * d.IH * d.IW + d.N * cout * d.OH * d.OW * (1 + (dmask is not None) + bool(accumulate))
                    + cout * cin * 16)
    label = "conv4x4<%s,s%d,nr%d>" % ("convT" if transposed else "conv", stride, (cout + 15) // 16)
    if d.N >= 128 and max(d.IH, d.IW, d.OH, d.OW) <= 34:
        label = "patch_" + label       # the D2 patch stacks (small-map kernels): their own label class for VTS_KNOCKOUT (round 5's
        #                                knock-out of "conv_small,wgrad_small,conv_head_small" matched NO label and measured nothing)
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d p%d%s%s" % (d.N, cin, d.IH, d.IW, cout, d.OH, d.OW, pad, " dmask" if dmask is not None else "",
                                                      " acc" if accumulate else "")
    if bwd_sums and dmask is not None and BWD_SUMS:
        # backward-data convolution in front of a normalisation backward: the epilogue also emits that backward's sums; `out` carries
        # them to norm_bwd (BSUMS: out tensor -> (partials, slots)), which then runs its apply pass only
        part = torch.empty(int(lib.vts_conv4x4_norm_ws_floats(C.byref(d))), dtype=torch.float32, device=x.device)
        slots = C.c_int(-1 if bwd_sums == "in" else 0)     # "in": the normalisation behind `out` is InstanceNorm2d(affine=False)
        _run(label, nbytes, flops, lib.vts_conv4x4_bsums, C.byref(d), part.data_ptr(), part.numel(), C.byref(slots), L.stream())
        if slots.value == -1:        # the k-split epilogue applied that backward itself: norm_bwd(out, ...) is a no-op
            BSUMS[out.data_ptr()] = (None, -1, out)
        elif slots.value:
            BSUMS[out.data_ptr()] = (part, slots.value, out)
        else:
            BSUMS.pop(out.data_ptr(), None)
        return out
    if not instance_norm and batch_norm is None:
        BSUMS.pop(out.data_ptr(), None)      # a plain write into this buffer invalidates sums an earlier convolution left for it
        _run(label, nbytes, flops, lib.vts_conv4x4, C.byref(d), L.stream())
        return out
    kw = dict(batch_norm) if batch_norm is not None else {}
    mode = 1 if batch_norm is not None else 0
    nd, st = _norm_desc(out, mode, **kw)
    sws = stat_workspace(lib.vts_conv4x4_norm_ws_floats(C.byref(d)), x.device)
    fused = C.c_int(0)
    _run(label, nbytes, flops, lib.vts_conv4x4_norm, C.byref(d), C.byref(nd), sws.data_ptr(), sws.numel(), C.byref(fused), L.stream())
    if fused.value >= 2:      # the epilogue wrote statistics partials: merge them (the second stage of norm_stats)
        if TIMER is not None:
            DETAIL = "%s N%d %dx%dx%d from %d epilogue slots" % ("BN" if mode else "IN", out.shape[0], cout, d.OH, d.OW, fused.value - 2)
        _run("norm_from_partials", 12.0 * out.shape[0] * cout * (fused.value - 2), 0.0, lib.vts_norm_stats_from_partials, C.byref(nd), sws.data_ptr(), fused.value - 2, L.stream())
        return Act(out, st[0], st[1], st[2], st[3])
    if fused.value:
        return Act(out, st[0], st[1], st[2], st[3])
    return norm_stats(out, mode, **kw)


# ---- deferred weight-gradient reduction -------------------------------------------------------------------------------
# Inside `with deferred_wgrad():` every wgrad4x4 leaves its per-workgroup partials in a per-lane arena and the deterministic
# reduction of ALL of them is one vts_wgrad_reduce_batch launch when the outermost context exits (the engine exits it after
# its side streams have joined the launch stream).  Outside the context wgrad4x4 reduces immediately.
_DEFER_DEPTH = 0
_pending = []        # ReduceJob-like dicts, in enqueue order
_pending_by_dw = {}
_arenas = {}


class _Arena:
    """bump allocator over grow-only blocks; the allocation sequence of a training step is the same every step, so a reset
    at every flush hands out the same addresses again (captured HIP graphs stay valid; blocks are never freed)"""

    def __init__(self, device):
        self.device, self.blocks, self.cur, self.off = device, [], 0, 0

    def alloc(self, nfloats):
        nfloats = (int(nfloats) + 63) // 64 * 64
        while True:
            if self.cur < len(self.blocks):
                b = self.blocks[self.cur]
                if self.off + nfloats <= b.numel():
                    t = b[self.off:self.off + nfloats]
                    self.off += nfloats
                    return t
                if self.off == 0:      # an empty block that is too small: replace it by a larger one (the old one stays alive)
                    _retired.append(b)
                    self.blocks[self.cur] = torch.empty(max(nfloats, 2 * b.numel()), dtype=torch.float32, device=self.device)
                    continue
                self.cur, self.off = self.cur + 1, 0
            else:
                self.blocks.append(torch.empty(max(nfloats, 16 << 20), dtype=torch.float32, device=self.device))

    def reset(self):
        self.cur, self.off = 0, 0


def _arena(device):
    key = (str(device), WS_LANE)
    a = _arenas.get(key)
    if a is None:
        a = _arenas[key] = _Arena(device)
    return a


DEFER_WGRAD = tune.get("VTS_WGRAD_DEFER", "1") != "0"
FLUSH_BYTES = int(tune.get("VTS_WGRAD_FLUSH_MB", "96")) << 20   # a lane reduces its pending partials once they exceed this (keeps the reduction spread over the backward)


def wgrad_flush(lane=None):
    """Reduce pending weight-gradient partials on the CURRENT stream: those of one lane (the caller is on that lane's stream, or
    has joined it), or of every lane (lane None: call on the launch stream after all lanes have joined)."""
    global _pending
    todo = [q for q in _pending if lane is None or q["lane"] == lane]
    if not todo:
        return
    jobs = (L.ReduceJob * len(todo))()
    nbytes = 0.0
    for j, q in zip(jobs, todo):
        j.dw, j.nel, j.accumulate, j.nseg = q["dw"].data_ptr(), q["nel"], int(q["accumulate"]), len(q["segs"])
        for i, (part, pw) in enumerate(q["segs"]):
            j.part[i], j.pw[i] = part.data_ptr(), pw
            nbytes += 4.0 * pw * q["nel"]
        _pending_by_dw.pop(q["dw"].data_ptr(), None)
    if TIMER is not None:
        global DETAIL
        DETAIL = "%d jobs %.1f MB partials, copies per job %d..%d" % (len(todo), nbytes / 1e6, min(pw for q in todo for _, pw in q["segs"]),
                                                                     max(pw for q in todo for _, pw in q["segs"]))
    _run("wgrad_reduce_batch", nbytes, 0.0, L.load().vts_wgrad_reduce_batch, jobs, len(todo), L.stream())
    _pending = [q for q in _pending if not (lane is None or q["lane"] == lane)]
    for (dev, ln), a in _arenas.items():
        if lane is None or ln == lane:
            a.reset()


def wgrad_discard():
    """drop every pending partial job and rewind the arenas (exceptional exit of a backward: nothing is reduced)"""
    global _pending
    _pending = []
    _pending_by_dw.clear()
    BSUMS.clear()
    for a in _arenas.values():
        a.reset()


def wgrad_pending_bytes(lane):
    return sum(4.0 * pw * q["nel"] for q in _pending if q["lane"] == lane for _, pw in q["segs"])


class deferred_wgrad:
    def __enter__(self):
        global _DEFER_DEPTH
        _DEFER_DEPTH += 1
        return self

    def __exit__(self, exc_type, exc, tb):
        global _DEFER_DEPTH
        _DEFER_DEPTH -= 1
        if _DEFER_DEPTH == 0:
            if exc_type is None:
                wgrad_flush()
            else:
                wgrad_discard()     # a failed backward must not leave stale partial jobs for the next one to reduce
        return False


def wgrad4x4(lo0, hi0, dw, *, lo1=None, hi1=None, act_lo=0, act_hi=0, stride=2, pad=1, accumulate=False, pad_dx=0, defer=None):
    """defer: None = follow the enclosing deferred_wgrad() context; False = reduce now (the caller reads dw right away)"""
    lib = L.load()
    d = L.WgradDesc()
    d.lo0, d.lo1, d.hi0, d.hi1 = _op(lo0), _op(lo1), _op(hi0), _op(hi1)
    d.act_lo, d.act_hi = act_lo, act_hi
    lo = lo0.data if isinstance(lo0, Act) else lo0
    hi = hi0.data if isinstance(hi0, Act) else hi0
    d.N, d.LH, d.LW, d.HH, d.HW = lo.shape[0], lo.shape[2], lo.shape[3], hi.shape[2], hi.shape[3]
    d.stride, d.pad = stride, pad
    d.pad_dx = pad_dx
    d.dw = dw.data_ptr()
    d.accumulate = int(accumulate)
    n = lib.vts_wgrad4x4_ws_floats(C.byref(d))
    cl, chn = d.lo0.C + d.lo1.C, d.hi0.C + d.hi1.C
    nel = cl * chn * 16
    if defer is None:
        defer = _DEFER_DEPTH > 0 and DEFER_WGRAD
    prev = _pending_by_dw.get(dw.data_ptr()) if defer else None
    if prev is not None and (len(prev["segs"]) >= 4 or prev["nel"] != nel):
        raise RuntimeError("wgrad4x4: more than 4 deferred contributions to one weight gradient")
    ws = _arena(lo.device).alloc(n) if defer else workspace(n, lo.device)
    d.defer = int(bool(defer))
    flops = 2.0 * d.N * d.LH * d.LW * cl * chn * 16
    nbytes = 4.0 * (d.N * cl * d.LH * d.LW + d.N * chn * d.HH * d.HW + cl * chn * 16)
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d lo %dx%dx%d hi %dx%dx%d p%d" % (d.N, cl, d.LH, d.LW, chn, d.HH, d.HW, pad)
    _run(("patch_wgrad4x4<s%d>" if (d.N >= 128 and max(d.HH, d.HW) <= 34) else "wgrad4x4<s%d>") % stride, nbytes, flops, lib.vts_wgrad4x4,
         C.byref(d), ws.data_ptr(), L.stream())
    if defer:
        seg = (ws, int(n // nel))
        if prev is not None:
            assert accumulate, "a second deferred contribution to a weight gradient must accumulate"
            prev["segs"].append(seg)
        else:
            q = dict(dw=dw, nel=nel, accumulate=accumulate, segs=[seg], lane=WS_LANE)
            _pending.append(q)
            _pending_by_dw[dw.data_ptr()] = q
        if wgrad_pending_bytes(WS_LANE) > FLUSH_BYTES:
            wgrad_flush(WS_LANE)      # on this lane's stream, in stream order behind the launches that wrote the partials
    return dw


# ---- K x K convolutions (K <= 8, stride 1) on the 4 x 4 kernels: one launch per 4 x 4 block of the tap grid ----
def _tap_blocks(K):
    nb = (K + 3) // 4
    return [(a, b) for a in range(nb) for b in range(nb)]


def tap_embed(w, K, a, b, w4):
    _run("tap_embed", 0.0, 0.0, L.load().vts_tap_embed, w.data_ptr(), w.numel() // (K * K), K, a, b, w4.data_ptr(), L.stream())
    return w4


def tap_extract(dw4, K, a, b, dw, accumulate=False):
    _run("tap_extract", 0.0, 0.0, L.load().vts_tap_extract, dw4.data_ptr(), dw.numel() // (K * K), K, a, b, dw.data_ptr(),
         int(accumulate), L.stream())
    return dw


def _w4_scratch(w, K, tag):
    """4x4 staging buffers for the tap blocks of `w` ([Co, Ci, K, K]); persistent (graph-capture safe)."""
    key = ("w4", w.data_ptr(), tuple(w.shape), tag)
    buf = _ws.get(key)
    if buf is None:
        buf = _ws[key] = torch.zeros(len(_tap_blocks(K)), w.shape[0], w.shape[1], 4, 4, dtype=torch.float32, device=w.device)
    return buf


def convk(x, w, out, *, bias=None, pad=0, act_in=0):
    """out <- Conv2d(K x K, stride 1, zero padding `pad`)(x); w is [Co, Ci, K, K].  Replaces the 3x3 / 7x7
    nn.Conv2d of ResnetGenerator / ResnetBlock (networks.py:1076,1084,1144,1306,1318)."""
    co, ci, K, _ = w.shape
    w4 = _w4_scratch(w, K, "fwd")
    for i, (a, b) in enumerate(_tap_blocks(K)):
        tap_embed(w, K, a, b, w4[i])
        conv4x4(x, w4[i], ci * 16, 16, co, out, bias=bias if i == 0 else None, stride=1, pad=pad - 4 * a, pad_dx=4 * (a - b),
                act_in=act_in, accumulate=i > 0)
    return out


def convk_bwd_data(dout, w, din, *, pad=0, accumulate=False):
    """din (+)= adjoint of convk w.r.t. its input: din[y] = sum_k dout[y + pad - k] w[k]."""
    co, ci, K, _ = w.shape
    w4 = _w4_scratch(w, K, "fwd")   # same blocks as the forward (already embedded in this step)
    for i, (a, b) in enumerate(_tap_blocks(K)):
        tap_embed(w, K, a, b, w4[i])
        conv4x4(dout, w4[i], 16, ci * 16, ci, din, stride=1, pad=pad - 4 * a, pad_dx=4 * (a - b), transposed=True,
                accumulate=accumulate or i > 0)
    return din


def wgradk(dout, x, dw, *, pad=0, act_hi=0, accumulate=False):
    """dw (+)= weight gradient of convk: dw[co, ci, ky, kx] = sum dout[n, co, y, x] * x[n, ci, y + ky - pad, x + kx - pad]."""
    co, ci, K, _ = dw.shape
    dw4 = _w4_scratch(dw, K, "grad")
    for i, (a, b) in enumerate(_tap_blocks(K)):
        wgrad4x4(dout, x, dw4[i], stride=1, pad=pad - 4 * a, pad_dx=4 * (a - b), act_hi=act_hi, defer=False)
        tap_extract(dw4[i], K, a, b, dw, accumulate=accumulate)
    return dw


# ---- stride-2 K x K convolutions (K <= 4, zero padding 0) on the stride-2 4 x 4 kernels: the K x K taps sit at offset o inside the
# 4 x 4 block and the operator runs with pad = o; o is chosen so that the transposed (input-adjoint) geometry check holds ----
def _s2_origin(K, h, w):
    oh, ow = (h - K) // 2 + 1, (w - K) // 2 + 1
    for o in range(0, 5 - K):
        if (h + 2 * o - 4) // 2 + 1 == oh and (w + 2 * o - 4) // 2 + 1 == ow:
            return o
    raise ValueError("no 4x4 embedding for a stride-2 %dx%d convolution on a %dx%d input" % (K, K, h, w))


def _embed_at(w, K, o, tag):
    w4 = _w4_scratch(w, K, tag)[0]
    _run("tap_embed", 0.0, 0.0, L.load().vts_tap_embed_at, w.data_ptr(), w.numel() // (K * K), K, -o, -o, w4.data_ptr(), L.stream())
    return w4


def convk_s2(x, w, out, *, bias=None, act_in=0):
    """out <- Conv2d(K x K, stride 2, padding 0)(x), K <= 4 (the EqualConv2d behind a Blur: stylegan_networks.py:639-657)"""
    co, ci, K, _ = w.shape
    xs = x.data.shape if isinstance(x, Act) else x.shape
    o = _s2_origin(K, xs[2], xs[3])
    conv4x4(x, _embed_at(w, K, o, "s2"), ci * 16, 16, co, out, bias=bias, stride=2, pad=o, act_in=act_in)
    return out


def convk_s2_bwd_data(dout, w, din, *, accumulate=False):
    co, ci, K, _ = w.shape
    o = _s2_origin(K, din.shape[2], din.shape[3])
    conv4x4(dout, _embed_at(w, K, o, "s2"), 16, ci * 16, ci, din, stride=2, pad=o, transposed=True, accumulate=accumulate)
    return din


def wgradk_s2(dout, x, dw, *, act_hi=0, accumulate=False):
    co, ci, K, _ = dw.shape
    xs = x.data.shape if isinstance(x, Act) else x.shape
    o = _s2_origin(K, xs[2], xs[3])
    dw4 = _w4_scratch(dw, K, "grad_s2")[0]
    wgrad4x4(dout, x, dw4, stride=2, pad=o, act_hi=act_hi, defer=False)
    _run("tap_extract", 0.0, 0.0, L.load().vts_tap_extract_at, dw4.data_ptr(), dw.numel() // (K * K), K, -o, -o, dw.data_ptr(),
         int(accumulate), L.stream())
    return dw


# ---- StyleGAN2 blocks (include/vts.h; reference models/stylegan_networks.py) ----
_HOST_KERNELS = {}


def _host_kernel(kernel):
    key = tuple(tuple(float(v) for v in row) for row in kernel)
    arr = _HOST_KERNELS.get(key)
    if arr is None:
        flat = [v for row in key for v in row]
        arr = _HOST_KERNELS[key] = ((C.c_float * len(flat))(*flat), len(key), len(key[0]))
    return arr


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0), out=None, accumulate=False):
    """upfirdn2d (stylegan_networks.py:38-76) of x [N,C,H,W]; kernel: 2-D host array (make_kernel); pad = (before, after)"""
    n, c, h, w = x.shape
    karr, kh, kw = _host_kernel(kernel)
    lib = L.load()
    oh, ow = lib.vts_upfirdn2d_out_size(h, kh, up, down, pad[0], pad[1]), lib.vts_upfirdn2d_out_size(w, kw, up, down, pad[0], pad[1])
    if out is None:
        out = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
    assert out.shape == (n, c, oh, ow) and x.is_contiguous() and out.is_contiguous()
    _run("upfirdn2d", 4.0 * (x.numel() + out.numel()), 2.0 * out.numel() * kh * kw / (up * up), lib.vts_upfirdn2d, x.data_ptr(), n * c, h, w,
         karr, kh, kw, up, down, pad[0], pad[1], pad[0], pad[1], out.data_ptr(), int(accumulate), L.stream())
    return out


def upfirdn2d_bwd(dout, din, kernel, up=1, down=1, pad=(0, 0), accumulate=False):
    """din [N,C,H,W] (+)= adjoint of upfirdn2d(., kernel, up, down, pad) applied to dout"""
    n, c, h, w = din.shape
    karr, kh, kw = _host_kernel(kernel)
    assert dout.is_contiguous() and din.is_contiguous()
    _run("upfirdn2d_bwd", 4.0 * (dout.numel() + din.numel()), 2.0 * dout.numel() * kh * kw / (up * up), L.load().vts_upfirdn2d_bwd,
         dout.data_ptr(), n * c, h, w, karr, kh, kw, up, down, pad[0], pad[1], pad[0], pad[1], din.data_ptr(), int(accumulate), L.stream())
    return din


def bias_act(x, bias, slope=0.2, gain=2.0 ** 0.5, res=None, out=None):
    """out = leaky_relu(x + bias[c], slope) * gain (+ res)  (fused_leaky_relu, stylegan_networks.py:18-19)"""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    if out is None:
        out = torch.empty_like(x)
    _run("bias_act", 4.0 * x.numel() * (3 if res is not None else 2), 0.0, L.load().vts_bias_act, x.data_ptr(), L.ptr(bias), L.ptr(res), n, c, hw,
         slope, gain, out.data_ptr(), L.stream())
    return out


def bias_act_bwd(g, x, bias, slope=0.2, gain=2.0 ** 0.5, dx=None):
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    if dx is None:
        dx = torch.empty_like(x)
    _run("bias_act_bwd", 12.0 * x.numel(), 0.0, L.load().vts_bias_act_bwd, g.data_ptr(), x.data_ptr(), L.ptr(bias), n, c, hw, slope, gain,
         dx.data_ptr(), L.stream())
    return dx


# ---- anti-aliased bicubic resampling (F.interpolate(mode="bicubic", align_corners=False, antialias=True)) ----------------------------
_AA_TABLES = {}


def _aa_axis(n_in, n_out):
    """index / weight table of one axis as PyTorch builds it (aten/native/cpu/UpSampleKernel.cpp, _compute_indices_min_size_weights_aa
    with the cubic filter a = -0.5 of _upsample_bicubic2d_aa): float32 arithmetic; returns (min [n_out], size [n_out], w [n_out, K])"""
    import numpy as np
    f32 = np.float32
    scale = f32(n_in) / f32(n_out)
    support = f32(2.0) * scale if scale >= 1.0 else f32(2.0)
    invscale = f32(1.0) / scale if scale >= 1.0 else f32(1.0)
    K = int(np.ceil(support)) * 2 + 1
    mins, sizes, w = np.zeros(n_out, np.int32), np.zeros(n_out, np.int32), np.zeros((n_out, K), np.float32)
    a = f32(-0.5)
    for i in range(n_out):
        center = scale * (f32(i) + f32(0.5))
        lo = max(int(center - support + f32(0.5)), 0)
        size = min(max(min(int(center + support + f32(0.5)), n_in) - lo, 0), K)
        t = np.abs((np.arange(size, dtype=np.float32) + f32(lo) - center + f32(0.5)) * invscale).astype(np.float32)
        wt = np.where(t < 1.0, ((a + f32(2.0)) * t - (a + f32(3.0))) * t * t + f32(1.0),
                      np.where(t < 2.0, (((t - f32(5.0)) * t + f32(8.0)) * t - f32(4.0)) * a, f32(0.0))).astype(np.float32)
        tot = wt.sum(dtype=np.float32)
        if tot != 0:
            wt = (wt / tot).astype(np.float32)
        mins[i], sizes[i] = lo, size
        w[i, :size] = wt
    return mins, sizes, w


def _aa_transpose(mins, sizes, w, n_in):
    """tables of the adjoint: for every input index the interval of outputs whose window holds it, with those weights"""
    import numpy as np
    n_out = len(mins)
    first = np.full(n_in, n_out, np.int64)
    last = np.full(n_in, -1, np.int64)
    for o in range(n_out):
        lo, hi = mins[o], mins[o] + sizes[o]
        first[lo:hi] = np.minimum(first[lo:hi], o)
        last[lo:hi] = np.maximum(last[lo:hi], o)
    cnt = np.maximum(last - first + 1, 0)
    K = max(int(cnt.max()), 1)
    tm, ts, tw = np.zeros(n_in, np.int32), cnt.astype(np.int32), np.zeros((n_in, K), np.float32)
    for i in range(n_in):
        if cnt[i] <= 0:
            continue
        tm[i] = first[i]
        for k in range(int(cnt[i])):
            o = first[i] + k
            j = i - mins[o]
            tw[i, k] = w[o, j] if 0 <= j < sizes[o] else 0.0
    return tm, ts, tw


def bicubic_aa_tables(ih, iw, oh, ow, device):
    """device tables ((ymin, ysize, wy), (xmin, xsize, wx)) forward and adjoint, cached per geometry"""
    key = (ih, iw, oh, ow, str(device))
    if key not in _AA_TABLES:
        def dev(t):
            return tuple(torch.from_numpy(a).to(device) for a in t)
        ay, ax = _aa_axis(ih, oh), _aa_axis(iw, ow)
        _AA_TABLES[key] = ((dev(ay), dev(ax)), (dev(_aa_transpose(*ay, ih)), dev(_aa_transpose(*ax, iw))))
    return _AA_TABLES[key]


def _resample(x, tabs, oh, ow, out, accumulate):
    (ymin, ysize, wy), (xmin, xsize, wx) = tabs
    n, c, h, w = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
    L.check(L.load().vts_resample_table(x.data_ptr(), n * c, h, w, ymin.data_ptr(), ysize.data_ptr(), wy.data_ptr(), wy.shape[1], xmin.data_ptr(),
                                        xsize.data_ptr(), wx.data_ptr(), wx.shape[1], out.data_ptr(), oh, ow, int(accumulate), L.stream()),
            "vts_resample_table")
    return out


def bicubic_aa(x, size, out=None):
    """F.interpolate(x, size, mode="bicubic", align_corners=False, antialias=True) (sinskitG_model.py:1440-1476, 1531-1557)"""
    oh, ow = size
    return _resample(x, bicubic_aa_tables(x.shape[2], x.shape[3], oh, ow, x.device)[0], oh, ow, out, False)


def bicubic_aa_bwd(dout, in_size, din=None, accumulate=False):
    """din (+)= adjoint of bicubic_aa (input size in_size) applied to dout"""
    ih, iw = in_size
    return _resample(dout, bicubic_aa_tables(ih, iw, dout.shape[2], dout.shape[3], dout.device)[1], ih, iw, din, accumulate)


def adain(x, s, eps=1e-5):
    """adaptive_instance_normalization(content x, style s) (thirdparty/AdaIN/function.py:15-23), [N,C,H,W] both"""
    n, c, h, w = x.shape
    assert s.shape == x.shape and x.is_contiguous() and s.is_contiguous()
    out = torch.empty_like(x)
    L.check(L.load().vts_adain(x.data_ptr(), s.data_ptr(), n * c, h * w, eps, out.data_ptr(), L.stream()), "vts_adain")
    return out


def adain_bwd(g, x, s, eps=1e-5):
    """(dx, ds) of adain for the output gradient g"""
    n, c, h, w = x.shape
    assert g.is_contiguous() and x.is_contiguous() and s.is_contiguous()
    dx, ds = torch.empty_like(x), torch.empty_like(s)
    L.check(L.load().vts_adain_bwd(g.data_ptr(), x.data_ptr(), s.data_ptr(), n * c, h * w, eps, dx.data_ptr(), ds.data_ptr(), L.stream()), "vts_adain_bwd")
    return dx, ds


def modconv_weight(w, transpose=False, eps=1e-8):
    """style-free demodulated weight of ModulatedConv2d (stylegan_networks.py:307-317 with style None): w [1,Co,Ci,K,K] -> [Co,Ci,K,K],
    or [Ci,Co,K,K] with transpose (include/vts.h)"""
    co, ci, k = w.shape[-4], w.shape[-3], w.shape[-1]
    out = torch.empty((ci, co, k, k) if transpose else (co, ci, k, k), dtype=torch.float32, device=w.device)
    L.check(L.load().vts_modconv_weight(w.data_ptr(), co, ci, k * k, 1.0 / (ci * k * k) ** 0.5, eps, int(transpose), out.data_ptr(), L.stream()),
            "vts_modconv_weight")
    return out


def modconv_weight_bwd(w, g, dw, transpose=False, accumulate=False, eps=1e-8):
    """dw (+)= gradient w.r.t. the raw weight given g = dL/d(modconv_weight(w, transpose))"""
    co, ci, k = w.shape[-4], w.shape[-3], w.shape[-1]
    assert g.is_contiguous() and dw.is_contiguous() and g.numel() == w.numel() == dw.numel()
    L.check(L.load().vts_modconv_weight_bwd(w.data_ptr(), g.data_ptr(), co, ci, k * k, 1.0 / (ci * k * k) ** 0.5, eps, int(transpose), dw.data_ptr(),
                                            int(accumulate), L.stream()), "vts_modconv_weight_bwd")
    return dw


def modconv_demod(w, s, scale, eps=1e-8):
    """demod [N, Co] of ModulatedConv2d (stylegan_networks.py:311-317); w [Co,Ci,K,K] (or [1,Co,Ci,K,K]), s [N,Ci]"""
    co, ci, kk = w.shape[-4], w.shape[-3], w.shape[-1] * w.shape[-2]
    n = s.shape[0]
    out = torch.empty(n, co, dtype=torch.float32, device=w.device)
    _run("modconv_demod", 4.0 * (w.numel() + s.numel() + out.numel()), 3.0 * n * w.numel(), L.load().vts_modconv_demod, w.data_ptr(), s.data_ptr(),
         n, co, ci, kk, scale, eps, out.data_ptr(), L.stream())
    return out


def w3x3_pack(w, mode, tag=None):
    """tap-major packing of a 3x3 weight for the *_wide kernels into a persistent buffer (see include/vts.h).
    w is an nn.Conv2d weight [Co,Ci,3,3] for the conv_* modes, an nn.ConvTranspose2d weight [Ci,Co,3,3] for convT_*."""
    d0, d1 = w.shape[0], w.shape[1]
    A, B, sa, sb, flip = {
        "conv_fwd": (d1, d0, 9, 9 * d1, 0), "conv_adj": (d0, d1, 9 * d1, 9, 1), "conv_s2_adj": (d0, d1, 9 * d1, 9, 0),
        "convT_fwd": (d0, d1, 9 * d1, 9, 0), "convT_adj": (d1, d0, 9, 9 * d1, 0)}[mode]
    key = ("wt", w.data_ptr(), tuple(w.shape), mode, tag)
    buf = _ws.get(key)
    if buf is None:
        buf = _ws[key] = torch.empty(A * 9 * ((B + 3) // 4 * 4), dtype=torch.float32, device=w.device)
    _run("w3x3_pack", 8.0 * w.numel(), 0.0, L.load().vts_w3x3_pack, w.data_ptr(), A, B, sa, sb, flip, buf.data_ptr(), L.stream())
    return buf


WINO = tune.get("VTS_WINO", "1") != "0"     # Winograd F(2x2, 3x3) for the frozen VGG stacks' 3x3 layers (0: direct GEMM-class kernel)


def w3x3_wino_pack(w, mode, tag=None):
    """transform-domain weights U of a 3x3 nn.Conv2d weight [Co,Ci,3,3] for conv3x3_wino, in a persistent buffer; mode conv_fwd | conv_adj"""
    d0, d1 = w.shape[0], w.shape[1]
    A, B, sa, sb, flip = {"conv_fwd": (d1, d0, 9, 9 * d1, 0), "conv_adj": (d0, d1, 9 * d1, 9, 1)}[mode]
    lib = L.load()
    key = ("wino", w.data_ptr(), tuple(w.shape), mode, tag)
    buf = _ws.get(key)
    if buf is None:
        buf = _ws[key] = torch.empty(int(lib.vts_w3x3_wino_floats(A, B)), dtype=torch.float32, device=w.device)
    _run("w3x3_wino_pack", 4.0 * (w.numel() + buf.numel()), 0.0, lib.vts_w3x3_wino_pack, w.data_ptr(), A, B, sa, sb, flip, buf.data_ptr(), L.stream())
    return buf


def conv3x3_wino_ok(n, ci, co, h, w):
    return WINO and bool(L.load().vts_conv3x3_wino_ok(n, ci, co, h, w))


def conv3x3_wino(p, U, bias, out, ep_mode=0, add=None, mask=None):
    """out <- 3x3 convolution of the pre-padded p [N,Ci,H+2,W+2] in Winograd F(2x2,3x3) form.  out [N,Co,H,W] (ep_mode 0), or the padded
    layout [N,Co,H+2,W+2] with ep_mode 0 (plain) | 1 (ReLU) | 2 ((conv + add) where mask > 0); add / mask have out's layout.
    The caller checks conv3x3_wino_ok first."""
    n, ci, ph, pw = p.shape
    co, h, w = out.shape[1], ph - 2, pw - 2
    out_pad = 1 if out.shape[2] == ph else 0
    assert out.shape == (n, co, h + 2 * out_pad, w + 2 * out_pad) and p.is_contiguous() and out.is_contiguous()
    assert ep_mode == 0 or out_pad == 1
    for t in (add, mask):
        assert t is None or (t.shape == out.shape and t.is_contiguous())
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d winograd%s" % (n, ci, h, w, co, h, w, ("", " relu+pad", " mask+pad")[ep_mode])
    # (flops: the multiplications the transform-domain GEMMs execute -- 16 per 2 x 2 outputs and channel pair, 4 / 9 of the direct form's --
    #  so that a roofline fraction computed from them stays a utilisation of the matrix pipe; the direct-equivalent rate is 2.25x that)
    _run("conv3x3_wide", 4.0 * (p.numel() + out.numel() * (1 + (add is not None) + (mask is not None)) + U.numel()),
         2.0 * n * ((h + 1) // 2) * ((w + 1) // 2) * 16 * co * ci, L.load().vts_conv3x3_wino, p.data_ptr(), U.data_ptr(), L.ptr(bias), out.data_ptr(), n, ci, co, h, w, out_pad, ep_mode, L.ptr(add), L.ptr(mask),
         L.stream())
    return out


def conv3x3_wide(p, wt, bias, out):
    """out [N,Co,H,W] <- valid 3x3 conv of the pre-padded p [N,Ci,H+2,W+2] with packed weights wt (GEMM-class kernel)"""
    n, ci, ph, pw = p.shape
    co, h, w = out.shape[1], ph - 2, pw - 2
    assert out.shape == (n, co, h, w) and p.is_contiguous() and out.is_contiguous()
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d" % (n, ci, h, w, co, h, w)
    lib = L.load()
    need = lib.vts_conv3x3_wide_ws_floats(n, ci, co, h, w)
    ws = workspace(need, p.device) if need else None
    _run("conv3x3_wide", 4.0 * (p.numel() + out.numel() + wt.numel()), 2.0 * n * h * w * co * ci * 9, lib.vts_conv3x3_wide,
         p.data_ptr(), wt.data_ptr(), L.ptr(bias), out.data_ptr(), n, ci, co, h, w, L.ptr(ws), ws.numel() if ws is not None else 0,
         L.stream())
    return out


def conv3x3s2_wide(p, wt, bias, out):
    """out [N,Co,OH,OW] <- 3x3 stride-2 conv of p [N,Ci,2OH+2,2OW+2] (the input zero-padded by 1)"""
    n, ci, ph, pw = p.shape
    co, oh, ow = out.shape[1:]
    assert (ph, pw) == (2 * oh + 2, 2 * ow + 2) and p.is_contiguous() and out.is_contiguous()
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d s2" % (n, ci, ph - 2, pw - 2, co, oh, ow)
    lib = L.load()
    need = lib.vts_conv3x3_wide_ws_floats(n, ci, co, oh, ow)
    ws = workspace(need, p.device) if need else None
    _run("conv3x3_wide", 4.0 * (p.numel() + out.numel() + wt.numel()), 2.0 * n * oh * ow * co * ci * 9, lib.vts_conv3x3s2_wide,
         p.data_ptr(), wt.data_ptr(), L.ptr(bias), out.data_ptr(), n, ci, co, oh, ow, L.ptr(ws), ws.numel() if ws is not None else 0,
         L.stream())
    return out


def tconv3x3s2_wide(p, wt, bias, out):
    """out [N,Co,2IH,2IW] <- ConvTranspose2d(3, stride 2, pad 1, output_padding 1) of p [N,Ci,IH+1,IW+1] (zero row/column appended)"""
    n, ci, ph, pw = p.shape
    co = out.shape[1]
    ih, iw = ph - 1, pw - 1
    assert out.shape == (n, co, 2 * ih, 2 * iw) and p.is_contiguous() and out.is_contiguous()
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d transposed s2" % (n, ci, ih, iw, co, 2 * ih, 2 * iw)
    lib = L.load()
    need = lib.vts_conv3x3_wide_ws_floats(n, ci, co, ih, iw)
    ws = workspace(need, p.device) if need else None
    _run("conv3x3_wide", 4.0 * (4 * p.numel() + out.numel() + wt.numel()), 2.0 * n * ih * iw * co * ci * 9, lib.vts_tconv3x3s2_wide,
         p.data_ptr(), wt.data_ptr(), L.ptr(bias), out.data_ptr(), n, ci, co, ih, iw, L.ptr(ws), ws.numel() if ws is not None else 0,
         L.stream())
    return out


def wgrad3x3_wide(dout, p, dw, accumulate=False, stride=1):
    """dw[a][b][3][3] (+)= sum dout[n,a,y,x] * p[n,b,stride*y+ky,stride*x+kx]; dout [N,A,H,W], p [N,B,stride*H+2,stride*W+2]"""
    n, co, h, w = dout.shape
    ci = p.shape[1]
    assert p.shape == (n, ci, stride * h + 2, stride * w + 2) and dw.shape == (co, ci, 3, 3) and dout.is_contiguous() and p.is_contiguous()
    lib = L.load()
    need = lib.vts_wgrad3x3_wide_ws_floats(n, ci, co, h, w, stride)
    ws = workspace(need, p.device) if need else None
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d dout %dx%dx%d in %dx%dx%d s%d" % (n, co, h, w, ci, p.shape[2], p.shape[3], stride)
    _run("wgrad3x3_wide", 4.0 * (dout.numel() + p.numel() + dw.numel()), 2.0 * n * h * w * co * ci * 9, lib.vts_wgrad3x3_wide,
         dout.data_ptr(), p.data_ptr(), dw.data_ptr(), n, ci, co, h, w, stride, int(accumulate), L.ptr(ws),
         ws.numel() if ws is not None else 0, L.stream())
    return dw


def w4x4_pack(w, mode, tag=None):
    """tap-major packing of an nn.Conv2d weight [Co,Ci,4,4] for conv4x4_flat (see include/vts.h) into a persistent buffer"""
    d0, d1 = w.shape[0], w.shape[1]
    A, B, sa, sb, flip = {"conv_fwd": (d1, d0, 16, 16 * d1, 0), "conv_adj": (d0, d1, 16 * d1, 16, 1),
                          "conv_s2_adj": (d0, d1, 16 * d1, 16, 0)}[mode]
    key = ("wt4", w.data_ptr(), tuple(w.shape), mode, tag)
    buf = _ws.get(key)
    if buf is None:
        buf = _ws[key] = torch.empty(A * 16 * ((B + 3) // 4 * 4), dtype=torch.float32, device=w.device)   # row pitch: B rounded up to 4
    _run("w4x4_pack", 8.0 * w.numel(), 0.0, L.load().vts_w4x4_pack, w.data_ptr(), A, B, sa, sb, flip, buf.data_ptr(), L.stream())
    return buf


def conv4x4_flat_ok(oh, ow, ph, pw, transposed=False):
    return bool(L.load().vts_conv4x4_flat_ok(oh, ow, ph, pw, int(transposed)))


def conv4x4_wide(p, wt, bias, out, stride=1, transposed=False):
    """4x4 conv of the pre-padded p [N,Ci,PH,PW] with 16-tap packed weights on the flattened small-map kernel; transposed:
    p is the output gradient of a Conv2d(4, s2, p2) with a zero row / column appended, out its input gradient"""
    n, ci, ph, pw = p.shape
    co, oh, ow = out.shape[1:]
    assert out.shape[0] == n and p.is_contiguous() and out.is_contiguous()
    lib = L.load()
    need = lib.vts_conv4x4_wide_ws_floats(n, ci, co, oh, ow, ph, pw, int(transposed))
    ws = workspace(need, p.device) if need else None
    taps = 4 if transposed else 16
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d s%d%s" % (n, ci, ph, pw, co, oh, ow, stride, " transposed" if transposed else "")
    _run("conv3x3_wide", 4.0 * (p.numel() + out.numel() + wt.numel()), 2.0 * n * oh * ow * co * ci * taps, lib.vts_conv4x4_wide,
         p.data_ptr(), wt.data_ptr(), L.ptr(bias), out.data_ptr(), n, ci, co, ph, pw, oh, ow, stride, int(transposed), L.ptr(ws),
         ws.numel() if ws is not None else 0, L.stream())
    return out


def wgrad4x4_wide(dout, p, dw, stride=1, accumulate=False):
    """dw[a][b][4][4] (+)= sum dout[n,a,y,x] * p[n,b,stride*y+ky,stride*x+kx]; p pre-padded (GEMM-class kernel, full-size maps)"""
    n, co, h, w = dout.shape
    ci, ph, pw = p.shape[1:]
    assert dw.shape == (co, ci, 4, 4) and dout.is_contiguous() and p.is_contiguous()
    lib = L.load()
    ws = workspace(lib.vts_wgrad4x4_wide_ws_floats(n, ci, co, h, w, stride), p.device)
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d dout %dx%dx%d in %dx%dx%d s%d" % (n, co, h, w, ci, ph, pw, stride)
    _run("wgrad3x3_wide", 4.0 * (dout.numel() + p.numel() + dw.numel()), 2.0 * n * h * w * co * ci * 16, lib.vts_wgrad4x4_wide,
         dout.data_ptr(), p.data_ptr(), dw.data_ptr(), n, ci, co, h, w, ph, pw, stride, int(accumulate), ws.data_ptr(), ws.numel(), L.stream())
    return dw


def pad_affine(x, pads, mode, out=None, act=0, res=None, out_nstride=0):
    """out <- act(pad(x)) (+ res).  pads = (top, bottom, left, right); mode 0 zero / 1 reflect / 2 replicate.
    `out` may be a channel-slice view of a wider tensor (pass its batch stride as out_nstride)."""
    op = _op(x)
    t = x.data if isinstance(x, Act) else x
    n, c, h, w = t.shape
    pt, pb, pl, pr = pads
    if out is None:
        out = torch.empty(n, c, h + pt + pb, w + pl + pr, dtype=torch.float32, device=t.device)
    if n * c > 65535:      # the kernel's grid carries (n, c) in one 16-bit dimension: sample chunks (VGG features of 256 patches x 512 channels)
        step = max(1, 65535 // c)
        ons = out_nstride if out_nstride else out.stride(0)
        for n0 in range(0, n, step):
            n1 = min(n, n0 + step)
            sub = Act(t[n0:n1], op_slice(x, "scale", n0, n1, c), op_slice(x, "shift", n0, n1, c)) if isinstance(x, Act) else t[n0:n1]
            pad_affine(sub, pads, mode, out=out[n0:n1], act=act, res=None if res is None else res[n0:n1], out_nstride=ons)
        return out
    _run("pad_affine", 4.0 * (t.numel() + out.numel() * (2 if res is not None else 1)), 0.0, L.load().vts_pad_affine, C.byref(op), n, h, w,
         pt, pb, pl, pr, mode, act, L.ptr(res), out.data_ptr(), out_nstride, L.stream())
    return out


def op_slice(a, name, n0, n1, c):
    v = getattr(a, name, None)
    return None if v is None else v[n0 * c:n1 * c]


def pad_bwd(dpad, pads, mode, din, accumulate=False):
    n, c, h, w = din.shape
    pt, pb, pl, pr = pads
    _run("pad_bwd", 4.0 * (dpad.numel() + din.numel()), 0.0, L.load().vts_pad_bwd, dpad.data_ptr(), n, c, h, w, pt, pb, pl, pr, mode,
         din.data_ptr(), int(accumulate), L.stream())
    return din


def blur_down(x, act=0, out=None):
    op = _op(x)
    t = x.data if isinstance(x, Act) else x
    n, c, h, w = t.shape
    if out is None:
        out = torch.empty(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, dtype=torch.float32, device=t.device)
    _run("blur_down", 4.0 * (t.numel() + out.numel()), 0.0, L.load().vts_blur_down, C.byref(op), act, n, h, w, out.data_ptr(), L.stream())
    return out


def blur_down_bwd(dout, din, accumulate=False):
    n, c, h, w = din.shape
    _run("blur_down_bwd", 4.0 * (dout.numel() + din.numel()), 0.0, L.load().vts_blur_down_bwd, dout.data_ptr(), n, c, h, w, din.data_ptr(),
         int(accumulate), L.stream())
    return din


def blur_up(x, act=0, out=None):
    op = _op(x)
    t = x.data if isinstance(x, Act) else x
    n, c, h, w = t.shape
    if out is None:
        out = torch.empty(n, c, 2 * h, 2 * w, dtype=torch.float32, device=t.device)
    _run("blur_up", 4.0 * (t.numel() + out.numel()), 0.0, L.load().vts_blur_up, C.byref(op), act, n, h, w, out.data_ptr(), L.stream())
    return out


def blur_up_bwd(dout, din, accumulate=False):
    n, c, h, w = din.shape
    _run("blur_up_bwd", 4.0 * (dout.numel() + din.numel()), 0.0, L.load().vts_blur_up_bwd, dout.data_ptr(), n, c, h, w, din.data_ptr(),
         int(accumulate), L.stream())
    return din


def channel_sum(x, out, accumulate=False):
    lib = L.load()
    n, c, h, w = x.shape
    ws = workspace(lib.vts_channel_sum_ws_floats(n, c, h * w), x.device)
    _run("channel_sum", 4.0 * n * c * h * w, 0.0, lib.vts_channel_sum, x.data_ptr(), x.stride(0), n, c, h * w, out.data_ptr(),
         int(accumulate), ws.data_ptr(), L.ptr(counters(x.device)) if c <= (1 << 16) else None, L.stream())
    return out


def _set_groups(d, groups, n):
    """groups: sample indices where the passes of a batched BatchNorm launch start, e.g. [0, 4] for fake | real"""
    if not groups or len(groups) <= 1:
        d.ngroups = 1
        return
    assert len(groups) <= 8 and groups[0] == 0 and all(b > a for a, b in zip(groups, list(groups[1:]) + [n]))
    d.ngroups = len(groups)
    for i, g in enumerate(list(groups) + [n]):
        d.gstart[i] = int(g)


def _norm_desc(x, mode, *, gamma=None, beta=None, running_mean=None, running_var=None, nbt=None, eps=1e-5,
               momentum=0.1, groups=None, stat_out=None, ext=None):
    """(vts_norm_desc of the statistics of x, its [4, N*C] output tensor) -- arguments as norm_stats"""
    n, c, h, w = x.shape
    st = torch.empty(4, n * c, dtype=torch.float32, device=x.device)
    d = L.NormDesc()
    d.x, d.nstride, d.N, d.C, d.HW, d.mode = x.data_ptr(), x.stride(0), n, c, h * w, mode
    d.eps, d.momentum = eps, momentum
    d.gamma, d.beta = L.ptr(gamma), L.ptr(beta)
    d.running_mean, d.running_var, d.num_batches_tracked = L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt)
    d.scale, d.shift, d.mean_out, d.rstd_out = st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr()
    _set_groups(d, groups, n)
    if stat_out is not None:
        d.stat_mean_out, d.stat_uvar_out = stat_out[0].data_ptr(), stat_out[1].data_ptr()
    if ext is not None:
        d.ext_mean, d.ext_uvar, d.ext_after = ext[0].data_ptr(), ext[1].data_ptr(), int(ext[2])
    return d, st


def norm_stats(x, mode, *, gamma=None, beta=None, running_mean=None, running_var=None, nbt=None, eps=1e-5,
               momentum=0.1, groups=None, stat_out=None, ext=None):
    """Returns Act(x, scale, shift, mean, rstd) -- x itself is not rewritten.
    groups: batched passes (BatchNorm); stat_out = (mean [C], uvar [C]) tensors to record this launch's batch statistics;
    ext = (mean [C], uvar [C], after) statistics of a separately launched pass to splice into the running-statistics sequence."""
    lib = L.load()
    n, c, h, w = x.shape
    st = torch.empty(4, n * c, dtype=torch.float32, device=x.device)
    d = L.NormDesc()
    d.x, d.nstride, d.N, d.C, d.HW, d.mode = x.data_ptr(), x.stride(0), n, c, h * w, mode
    d.eps, d.momentum = eps, momentum
    d.gamma, d.beta = L.ptr(gamma), L.ptr(beta)
    d.running_mean, d.running_var, d.num_batches_tracked = L.ptr(running_mean), L.ptr(running_var), L.ptr(nbt)
    d.scale, d.shift, d.mean_out, d.rstd_out = st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr()
    _set_groups(d, groups, n)
    if stat_out is not None:
        d.stat_mean_out, d.stat_uvar_out = stat_out[0].data_ptr(), stat_out[1].data_ptr()
    if ext is not None:
        d.ext_mean, d.ext_uvar, d.ext_after = ext[0].data_ptr(), ext[1].data_ptr(), int(ext[2])
    ws = workspace(lib.vts_norm_ws_floats(n, c, h * w), x.device)
    d.counters = L.ptr(counters(x.device)) if n * c <= (1 << 16) else None
    if TIMER is not None:
        global DETAIL
        DETAIL = "%s N%d %dx%dx%d%s" % ("BN" if mode else "IN", n, c, h, w, " groups %s" % list(groups) if groups else "")
    # (label: the many-small-maps launches of the D2 patch stacks can be knocked out separately in timing experiments)
    _run("norm_stats" if n < 128 else "norm_patch_stats", 4.0 * n * c * h * w, 0.0, lib.vts_norm_stats, C.byref(d), ws.data_ptr(), L.stream())
    return Act(x, st[0], st[1], st[2], st[3])


def norm_bwd(dy, act, mode, *, gamma=None, dgamma=None, dbeta=None, accumulate=False, groups=None, beta=None):
    """In place: dy (grad wrt normalised output) -> grad wrt the raw tensor act.data.
    beta: the BatchNorm shift (only needed when the producer of dy left epilogue sums for it: conv4x4(bwd_sums=True))"""
    lib = L.load()
    n, c, h, w = dy.shape
    d = L.NormBwdDesc()
    d.dy, d.x, d.nstride, d.N, d.C, d.HW, d.mode = dy.data_ptr(), act.data.data_ptr(), act.data.stride(0), n, c, h * w, mode
    d.mean, d.rstd = act.mean.data_ptr(), act.rstd.data_ptr()
    d.gamma, d.dgamma, d.dbeta = L.ptr(gamma), L.ptr(dgamma), L.ptr(dbeta)
    d.accumulate_param_grads = int(accumulate)
    _set_groups(d, groups, n)
    if TIMER is not None:
        global DETAIL
        DETAIL = "%s N%d %dx%dx%d%s" % ("BN" if mode else "IN", n, c, h, w, " groups %s" % list(groups) if groups else "")
    pre = BSUMS.pop(dy.data_ptr(), None)
    if pre is not None and pre[2] is dy and pre[1] == -1:
        if mode != 0 or dgamma is not None:
            raise RuntimeError("conv4x4(bwd_sums='in') applied an InstanceNorm backward, but norm_bwd is asked for mode %d" % mode)
        return dy
    if pre is not None and pre[2] is dy:
        if TIMER is not None:
            DETAIL += " from %d epilogue slots" % pre[1]
        _run("norm_bwd_from_partials", 4.0 * n * c * h * w * 3, 0.0, lib.vts_norm_bwd_from_partials, C.byref(d), pre[0].data_ptr(), pre[1], L.ptr(beta) if mode else None,
             L.stream())
        return dy
    ws = workspace(lib.vts_norm_ws_floats(n, c, h * w), dy.device)
    d.counters = L.ptr(counters(dy.device)) if n * c <= (1 << 16) else None
    _run("norm_bwd" if n < 128 else "norm_patch_bwd", 4.0 * n * c * h * w * 3, 0.0, lib.vts_norm_bwd, C.byref(d), ws.data_ptr(), L.stream())
    return dy


def act_bwd(g, act, kind, dy, accumulate=False):
    lib = L.load()
    n, c, h, w = g.shape
    o = act.operand() if isinstance(act, Act) else L.operand(act)
    L.check(lib.vts_act_bwd(g.data_ptr(), C.byref(o), n, h * w, kind, dy.data_ptr(), int(accumulate), L.stream()), "vts_act_bwd")
    return dy


def avgpool(x, y=None):
    lib = L.load()
    n, c, h, w = x.shape
    if y is None:
        y = torch.empty(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, dtype=torch.float32, device=x.device)
    L.check(lib.vts_avgpool3s2(x.data_ptr(), x.stride(0), n, c, h, w, y.data_ptr(), L.stream()), "vts_avgpool3s2")
    return y


def avgpool_bwd(dy, dx, accumulate=False, channels=None, dx_nstride=None):
    lib = L.load()
    n, c = dy.shape[0], dy.shape[1]
    h, w = dx.shape[2], dx.shape[3]
    L.check(lib.vts_avgpool3s2_bwd(dy.data_ptr(), n, c, h, w, dx.data_ptr(), dx.stride(0) if dx_nstride is None else dx_nstride,
                                   int(accumulate), L.stream()), "vts_avgpool3s2_bwd")
    return dx


LOSS_SCALE = float(2 ** 40)     # include/vts.h VTS_LOSS_SCALE: loss slots are int64 fixed point


def loss_slots(n, device):
    return torch.zeros(n, dtype=torch.int64, device=device)


def capture_node_count():
    """(nodes, kernel nodes) of the graph the current stream is capturing into, (0, 0) outside a capture"""
    n, k = C.c_int(0), C.c_int(0)
    L.check(L.load().vts_capture_node_count(L.stream(), C.byref(n), C.byref(k)), "vts_capture_node_count")
    return n.value, k.value


def step_begin(slots, counters):
    """zero the loss slots and advance the optimisers' device step counters (one launch)"""
    BSUMS.clear()      # epilogue sums nobody consumed (a backward that raised) must not outlive their step
    L.check(L.load().vts_step_begin(slots.data_ptr(), slots.numel(), L.ptr(counters), 0 if counters is None else counters.numel(), L.stream()), "vts_step_begin")


def loss_values(slots):
    """host floats of a slot tensor (one device -> host copy)"""
    return [v / LOSS_SCALE for v in slots.cpu().tolist()]


def ganloss(pred, mode, target_is_real, coeff, loss_slot, dpred=None, label=None, grad_coeff=None):
    lib = L.load()
    assert loss_slot is None or loss_slot.dtype == torch.int64
    n = pred.shape[0]
    m = pred.numel() // n
    if label is None:
        label = 1.0 if target_is_real else 0.0
    L.check(lib.vts_ganloss(pred.data_ptr(), n, m, L.GAN_MODES[mode], int(target_is_real), label, coeff,
                            coeff if grad_coeff is None else grad_coeff, L.ptr(loss_slot),
                            L.ptr(dpred), L.stream()), "vts_ganloss")


def l1(a, b, coeff, loss_slot, grad=None, accumulate=False):
    lib = L.load()
    assert loss_slot is None or loss_slot.dtype == torch.int64
    L.check(lib.vts_l1(a.data_ptr(), b.data_ptr(), a.numel(), coeff, L.ptr(loss_slot), L.ptr(grad), int(accumulate), L.stream()),
            "vts_l1")


def patch_gather(src, img, offx, offy, size, out, c0=0, channels=None):
    lib = L.load()
    c = src.shape[1] if channels is None else channels
    L.check(lib.vts_patch_gather(src.data_ptr(), src.stride(0), c, src.shape[2], src.shape[3], img.data_ptr(), offx.data_ptr(),
                                 offy.data_ptr(), img.numel(), size, out.data_ptr(), out.shape[1], c0, L.stream()), "vts_patch_gather")
    return out


def patch_scatter_bwd(dpatch, c0, channels, offx, offy, ppi, size, dsrc, accumulate=False):
    lib = L.load()
    n, _, h, w = dsrc.shape
    L.check(lib.vts_patch_scatter_bwd(dpatch.data_ptr(), dpatch.shape[1], c0, channels, None, offx.data_ptr(), offy.data_ptr(),
                                      dpatch.shape[0], ppi, size, dsrc.data_ptr(), dsrc.stride(0), n, h, w, int(accumulate), L.stream()),
            "vts_patch_scatter_bwd")
    return dsrc


def g_post(g_out, M, scale_nz, rb=None, rs=None, fake_I=None, fake_T=None, fake_N=None, aug_fake_I=None, S=None, stack_S=None, stack_M=None):
    """stack_S / stack_M: 1-channel slices of the full-resolution D2 stack that also receive the sketch S and the mask"""
    lib = L.load()
    n, _, h, w = g_out.shape
    # both slices belong to the same stack tensor: its batch stride comes from whichever is present (with --use_cGAN_G2_S False
    # only the mask slice exists; taking the stride from stack_S alone wrote every sample's mask into sample 0)
    present = [t for t in (stack_S, stack_M) if t is not None]
    if len(present) == 2 and stack_S.stride(0) != stack_M.stride(0):
        raise ValueError("g_post: stack_S and stack_M must be slices of one stack tensor (batch strides %d / %d)"
                         % (stack_S.stride(0), stack_M.stride(0)))
    stack_ns = present[0].stride(0) if present else 0
    L.check(lib.vts_g_post_stack(g_out.data_ptr(), M.data_ptr(), n, h, w, scale_nz, L.ptr(rb), L.ptr(rs), L.ptr(fake_I), L.ptr(fake_T),
                                 0 if fake_T is None else fake_T.stride(0), L.ptr(fake_N), L.ptr(aug_fake_I),
                                 0 if aug_fake_I is None else aug_fake_I.stride(0), L.ptr(S), L.ptr(stack_S), L.ptr(stack_M),
                                 stack_ns, L.stream()), "vts_g_post")


def patch_jobs(jobs, size=32):
    """ONE launch for a list of patch jobs: dict(dst, c0, src=None, channels=None, img=None, offx=None, offy=None, fill=0.0) --
    gather (img / offx / offy given) from an image tensor, copy from a [P, C, size, size] patch tensor, or fill (src None)"""
    arr = (L.PatchJob * len(jobs))()
    for a, q in zip(arr, jobs):
        dst, src = q["dst"], q.get("src")
        a.dst, a.dst_C, a.dst_c0, a.P = dst.data_ptr(), dst.shape[1], q["c0"], dst.shape[0]
        a.fill = float(q.get("fill", 0.0))
        if src is None:
            a.src, a.C = None, q["channels"]
            continue
        a.src, a.src_nstride = src.data_ptr(), src.stride(0)
        a.C = src.shape[1] if q.get("channels") is None else q["channels"]
        if q.get("img") is not None:
            a.H, a.W = src.shape[2], src.shape[3]
            a.img, a.offx, a.offy = q["img"].data_ptr(), q["offx"].data_ptr(), q["offy"].data_ptr()
            assert q["img"].numel() == dst.shape[0]
        else:
            assert src.shape[0] == dst.shape[0] and src.shape[2:] == dst.shape[2:] and src.stride(1) == size * size
    L.check(L.load().vts_patch_jobs(arr, len(jobs), size, L.stream()), "vts_patch_jobs")


def diffaug_bs_mask(x, M, rb, rs, out):
    lib = L.load()
    n, _, h, w = x.shape
    L.check(lib.vts_diffaug_bs_mask(x.data_ptr(), L.ptr(M), n, h, w, rb.data_ptr(), rs.data_ptr(), out.data_ptr(), L.stream()),
            "vts_diffaug_bs_mask")
    return out


def diffaug_draws(policy, shape, device):
    """the random numbers DiffAugment(x, policy) consumes (thirdparty/DiffAugment.py:25-80), drawn on the device: one dict per letter
    (b / s / c: r; t: tx, ty; o: ox, oy; n: sigma, noise) -- the layout of oracle/nets.py:diffaug_draws, which tests pass in instead"""
    n, c, h, w = shape
    out = []
    for letter in policy:
        if letter in "bsc":
            out.append({"r": torch.rand(n, device=device)})
        elif letter == "t":
            sx, sy = int(h * 0.125 + 0.5), int(w * 0.125 + 0.5)
            out.append({"tx": torch.randint(-sx, sx + 1, (n,), device=device), "ty": torch.randint(-sy, sy + 1, (n,), device=device)})
        elif letter == "o":
            ch, cw = int(h * 0.5 + 0.5), int(w * 0.5 + 0.5)
            out.append({"ox": torch.randint(0, h + (1 - ch % 2), (n,), device=device), "oy": torch.randint(0, w + (1 - cw % 2), (n,), device=device)})
        elif letter == "n":
            sigma = torch.rand(n, device=device) * 0.1
            sigma = torch.where(torch.rand(n, device=device) < 0.5, sigma, torch.zeros_like(sigma))
            out.append({"sigma": sigma, "noise": torch.randn(n, c, h, w, device=device)})
        else:
            raise KeyError("DiffAugment policy letter '%s' (the reference knows b s c t o n)" % letter)
    return out


def diffaug_policy(x, policy, draws, M, out):
    """out <- DiffAugment(x, policy) * M as a chain of vts_diffaug_op launches, one per letter (the mask rides on the last one).
    x: contiguous [N, C, H, W]; out: [N, C, H, W] with any sample stride (e.g. a channel slice of a stack); draws: diffaug_draws layout"""
    lib = L.load()
    n, c, h, w = x.shape
    dev = x.device
    assert x.is_contiguous() and out.stride(1) == h * w and out.stride(3) == 1 and len(draws) == len(policy)
    assert policy, "empty policy: the caller skips the augmentation"
    cur = x
    for k, (letter, d) in enumerate(zip(policy, draws)):
        last = k == len(policy) - 1
        dst = out if last else torch.empty_like(x)
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
        pf = pi0 = pi1 = noise = ws = None
        if letter in "bsc":
            pf = f32(d["r"])
        elif letter == "t":
            pi0, pi1 = i32(d["tx"]), i32(d["ty"])
        elif letter == "o":
            pi0, pi1 = i32(d["ox"]), i32(d["oy"])
        elif letter == "n":
            pf, noise = f32(d["sigma"]), f32(d["noise"])
        else:
            raise KeyError("DiffAugment policy letter '%s' (the reference knows b s c t o n)" % letter)
        if letter == "c":
            ws = torch.empty(int(lib.vts_diffaug_op_ws_floats(n)), dtype=torch.float32, device=dev)
        L.check(lib.vts_diffaug_op(cur.data_ptr(), cur.stride(0), dst.data_ptr(), dst.stride(0), n, c, h, w, ord(letter), L.ptr(pf), L.ptr(pi0),
                                   L.ptr(pi1), L.ptr(noise), L.ptr(M) if last else None, L.ptr(ws), L.stream()), "vts_diffaug_op")
        cur = dst
    return out


def g_out_grad(d_fake_I, d_fake_T, M, g_out, d_raw, coarse=None):
    """coarse: the image gradient's next pyramid level [N, 3, ceil(H/2), ceil(W/2)], whose average-pool adjoint is added to d_fake_I on
    the fly (the last avgpool_bwd of engine._merge_input_grads, fused)"""
    lib = L.load()
    n, _, h, w = g_out.shape
    if coarse is not None:
        assert d_fake_I is not None and coarse.is_contiguous() and tuple(coarse.shape) == (n, 3, (h + 1) // 2, (w + 1) // 2), coarse.shape
    L.check(lib.vts_g_out_grad_pool(L.ptr(d_fake_I), L.ptr(coarse), L.ptr(d_fake_T), M.data_ptr(), g_out.data_ptr(), n, h, w, d_raw.data_ptr(),
                                    L.stream()), "vts_g_out_grad")
    return d_raw


def u8_expand(src, normalize, out=None):
    """uint8 image data -> float32 as ToTensor (/ 255) [+ Normalize(0.5, 0.5)] would make it, bit for bit"""
    assert src.dtype == torch.uint8 and src.is_contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    L.check(L.load().vts_u8_expand(src.data_ptr(), src.numel(), int(bool(normalize)), out.data_ptr(), L.stream()), "vts_u8_expand")
    return out


def input_images_u8(S, I, M, M_out, S_out, S_out2, I_out):
    """set_input's image part in one launch: bytes -> M / 255, Normalize(ToTensor(S)) * M (twice), Normalize(ToTensor(I)) * M; bit-identical
    to u8_expand + mask_mul"""
    n, hw = S.shape[0], S.shape[2] * S.shape[3]
    for t in (S, I, M, M_out, S_out, S_out2, I_out):
        assert t is None or t.is_contiguous()
    assert S.dtype == torch.uint8 and (I is None or (I.dtype == torch.uint8 and I.shape[1] == 3)) and S.shape[1] == 1
    L.check(L.load().vts_input_images_u8(S.data_ptr(), L.ptr(I), L.ptr(M), n, hw, L.ptr(M_out), S_out.data_ptr(), L.ptr(S_out2), L.ptr(I_out),
                                         L.stream()), "vts_input_images_u8")


def pool_query(images, store, ret_slot, put_slot, out):
    """ImagePool.query on the device (vts_pool_query): out[n] = store[ret_slot[n]] or images[n]; store[put_slot[n]] = images[n]; n in order"""
    n = images.shape[0]
    elems = images.numel() // max(n, 1)
    assert images.is_contiguous() and store.is_contiguous() and out.is_contiguous() and store.numel() % max(elems, 1) == 0
    assert ret_slot.dtype == torch.int32 and put_slot.dtype == torch.int32 and ret_slot.numel() == n and put_slot.numel() == n
    L.check(L.load().vts_pool_query(images.data_ptr(), store.data_ptr(), ret_slot.data_ptr(), put_slot.data_ptr(), n, elems, out.data_ptr(),
                                    L.stream()), "vts_pool_query")
    return out


def mask_mul(x, M, out=None):
    lib = L.load()
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty_like(x)
    L.check(lib.vts_mask_mul(x.data_ptr(), M.data_ptr(), n, c, h * w, out.data_ptr(), L.stream()), "vts_mask_mul")
    return out


def spe_grid(out, dim, c0=0):
    """Fill out[:, c0:c0+2*dim] with the sinusoidal grid."""
    lib = L.load()
    n, _, h, w = out.shape
    view = out[:, c0:]
    L.check(lib.vts_spe_grid(view.data_ptr(), out.stride(0), n, h, w, dim, L.stream()), "vts_spe_grid")
    return out


def mask_candidates(M, cand=None, prefix=None):
    lib = L.load()
    n, _, h, w = M.shape
    if cand is None:
        cand = torch.empty(n, h - 14, w - 14, dtype=torch.uint8, device=M.device)
    if prefix is None:
        prefix = torch.empty(n, h - 14 + 1, dtype=torch.int32, device=M.device)
    L.check(lib.vts_mask_candidates(M.data_ptr(), n, h, w, cand.data_ptr(), prefix.data_ptr(), L.stream()), "vts_mask_candidates")
    return cand, prefix


def mask_sample_ranks(prefix, h, k, seed, ranks):
    """ranks [n, k] int64 <- k distinct uniform ranks below every image's candidate count (prefix [n, h - 14 + 1] of mask_candidates)"""
    lib = L.load()
    n = prefix.shape[0]
    L.check(lib.vts_mask_sample_ranks(prefix.data_ptr(), n, h, k, seed & 0xFFFFFFFFFFFFFFFF, ranks.data_ptr(), L.stream()), "vts_mask_sample_ranks")
    return ranks


def mask_select(cand, prefix, ranks, h, w):
    lib = L.load()
    n, k = ranks.shape
    offx = torch.empty(n * k, dtype=torch.int32, device=cand.device)
    offy = torch.empty(n * k, dtype=torch.int32, device=cand.device)
    L.check(lib.vts_mask_select(cand.data_ptr(), prefix.data_ptr(), n, h, w, ranks.data_ptr(), k, offx.data_ptr(), offy.data_ptr(),
                                L.stream()), "vts_mask_select")
    return offx, offy


def eval_metrics(real_I, fake_I, real_T, fake_T):
    """device tensor [I_PSNR, T_AE, T_MSE, I_SSIM] (model_utils.py:431-561; the metrics that need no pretrained network; I_SSIM is
    NaN for images smaller than its 11 x 11 window)"""
    lib = L.load()
    ws = workspace(lib.vts_metric_ws_floats(), real_I.device)
    out = torch.full((6,), float("nan"), dtype=torch.float32, device=real_I.device)   # [lo, hi, psnr, ae, mse, ssim]
    st = L.stream()
    L.check(lib.vts_minmax(real_I.data_ptr(), real_I.numel(), out.data_ptr(), ws.data_ptr(), st), "vts_minmax")
    L.check(lib.vts_metric_psnr(real_I.data_ptr(), fake_I.data_ptr(), real_I.numel(), out.data_ptr(), out[2:].data_ptr(), ws.data_ptr(), st),
            "vts_metric_psnr")
    p, _, h, w = real_T.shape
    L.check(lib.vts_metric_tactile(real_T.data_ptr(), fake_T.data_ptr(), p, h * w, out[3:].data_ptr(), out[4:].data_ptr(), ws.data_ptr(), st),
            "vts_metric_tactile")
    n, c, ih, iw = real_I.shape
    if ih >= 11 and iw >= 11:
        assert real_I.is_contiguous() and fake_I.is_contiguous()
        L.check(lib.vts_metric_ssim(real_I.data_ptr(), fake_I.data_ptr(), n * c, ih, iw, out.data_ptr(), out[5:].data_ptr(), ws.data_ptr(), st),
                "vts_metric_ssim")
    return out[2:]


def frechet_distance(f1, f2):
    """Frechet distance (the arithmetic of SIFID, models/sifid.py:102-176) between two channel-major feature sets [D, P1] / [D, P2]
    (D <= 64); returns a 1-element device tensor"""
    lib = L.load()
    d, p1 = f1.shape
    assert f2.shape[0] == d and f1.is_contiguous() and f2.is_contiguous()
    ws = workspace(lib.vts_frechet_ws_floats(), f1.device)
    out = torch.empty(1, dtype=torch.float32, device=f1.device)
    L.check(lib.vts_frechet_distance(f1.data_ptr(), f2.data_ptr(), d, p1, f2.shape[1], out.data_ptr(), ws.data_ptr(), L.stream()), "vts_frechet_distance")
    return out


def minmax(x):
    """device tensor {min x, max x}"""
    lib = L.load()
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    ws = workspace(lib.vts_metric_ws_floats(), x.device)
    L.check(lib.vts_minmax(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), L.stream()), "vts_minmax")
    return out


def sifid_input(src, c0, channels, size=None, lohi=None, clamp01=False):
    """[N,3,OH,OW] network input of the SIFID metrics from channels c0.. of src (3 channels, or 1 tiled three times): optional min-max
    normalisation by the device pair lohi (+ clamp), optional clamp of raw values, nearest resize to `size` (include/vts.h)"""
    lib = L.load()
    n, _, h, w = src.shape
    assert src.stride(1) == h * w and src.stride(3) == 1
    oh, ow = size if size else (h, w)
    out = torch.empty(n, 3, oh, ow, dtype=torch.float32, device=src.device)
    L.check(lib.vts_sifid_input(src.data_ptr(), src.stride(0), n, c0, channels, h, w, L.ptr(lohi), int(clamp01), out.data_ptr(), oh, ow,
                                L.stream()), "vts_sifid_input")
    return out


def adam_flat(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    lib = L.load()
    L.check(lib.vts_adam_flat(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps, step, grad_scale,
                              L.stream()), "vts_adam_flat")


def adam_flat_dev(p, g, m, v, lr_dev, beta1, beta2, eps, step_dev, grad_scale=1.0):
    lib = L.load()
    L.check(lib.vts_adam_flat_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr_dev.data_ptr(), beta1, beta2,
                                  eps, step_dev.data_ptr(), grad_scale, L.stream()), "vts_adam_flat_dev")


def l2norm_rows(x, out=None):
    lib = L.load()
    if out is None:
        out = torch.empty_like(x)
    L.check(lib.vts_l2norm_rows(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), L.stream()), "vts_l2norm_rows")
    return out


def patchnce(q, k, groups, T, gscale=1.0, want_grad=True):
    lib = L.load()
    rows, d = q.shape
    p = rows // groups
    loss = torch.empty(rows, dtype=torch.float32, device=q.device)
    dq = torch.empty_like(q) if want_grad else None
    L.check(lib.vts_patchnce(q.data_ptr(), k.data_ptr(), groups, p, d, T, gscale, loss.data_ptr(), L.ptr(dq), L.stream()), "vts_patchnce")
    return loss, dq


def patch_sample(feat, ids):
    """PatchSampleF's gather (networks.py:689-701): feat [B, C, H, W], ids int64 [P] -> [B * P, C]"""
    b, c, h, w = feat.shape
    assert feat.is_contiguous() and ids.dtype == torch.int64
    out = torch.empty(b * ids.numel(), c, dtype=torch.float32, device=feat.device)
    L.check(L.load().vts_patch_sample(feat.data_ptr(), ids.data_ptr(), b, c, h * w, ids.numel(), out.data_ptr(), L.stream()), "vts_patch_sample")
    return out


def linear_rows(x, weight, bias=None, relu=False):
    """act(x @ weight.T + bias) for row-major x [R, I], nn.Linear weight [O, I]"""
    r, i = x.shape
    o = weight.shape[0]
    assert weight.shape[1] == i and x.is_contiguous() and weight.is_contiguous()
    y = torch.empty(r, o, dtype=torch.float32, device=x.device)
    L.check(L.load().vts_linear_rows(x.data_ptr(), weight.data_ptr(), L.ptr(bias), r, i, o, int(relu), y.data_ptr(), L.stream()), "vts_linear_rows")
    return y


# ---- perceptual terms: glue around the frozen VGG stacks (include/vts.h) ----
def maxpool2_relu_pad(z, pad=1, zpad=0):
    """zpad: z is a padded tensor [N, C, H + 2 zpad, W + 2 zpad] read at its interior (the *_relu_pad convolution's output)"""
    n, c = z.shape[:2]
    h, w = z.shape[2] - 2 * zpad, z.shape[3] - 2 * zpad
    assert z.is_contiguous() and h >= 2 and w >= 2      # (odd sizes: the last row / column belongs to no window, as in MaxPool2d's floor mode)
    out = torch.empty(n, c, h // 2 + 2 * pad, w // 2 + 2 * pad, dtype=torch.float32, device=z.device)
    _run("maxpool2_relu_pad", 4.0 * (z.numel() + out.numel()), 0.0, L.load().vts_maxpool2_relu_pad, z.data_ptr(), n * c, h, w, pad, out.data_ptr(), zpad, L.stream())
    return out


def conv3x3_wide_relu_pad(p, wt, bias, out):
    """out [N, Co, H + 2, W + 2] (zero border, kept by the caller) <- interior = relu(valid 3x3 conv of the pre-padded p [N, Ci, H + 2, W + 2]);
    returns False when the shape does not take a tiled direct launch (the caller then uses conv3x3_wide + a padding pass)"""
    n, ci, ph, pw = p.shape
    co, h, w = out.shape[1], ph - 2, pw - 2
    assert out.shape == (n, co, ph, pw) and p.is_contiguous() and out.is_contiguous()
    lib = L.load()
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d relu+pad" % (n, ci, h, w, co, h, w)
    rc = [0]

    def call(*a):
        rc[0] = lib.vts_conv3x3_wide_relu_pad(*a)
        return 0 if rc[0] == L.ERR_UNSUPPORTED else rc[0]
    _run("conv3x3_wide", 4.0 * (p.numel() + out.numel() + wt.numel()), 2.0 * n * h * w * co * ci * 9, call,
         p.data_ptr(), wt.data_ptr(), L.ptr(bias), out.data_ptr(), n, ci, co, h, w, L.stream())
    return rc[0] == 0


def conv3x3_wide_mask_pad(p, wt, out, mask, add=None):
    """out [N, Co, H + 2, W + 2] (zero border) <- interior = (valid 3x3 conv of the pre-padded p + add) where mask > 0, else 0; mask / add
    have out's layout.  The input adjoint of a frozen VGG layer with the ReLU mask of the layer in front (and that layer's tap gradient)
    in its epilogue.  False: shape not taken (see conv3x3_wide_relu_pad)."""
    n, ci, ph, pw = p.shape
    co, h, w = out.shape[1], ph - 2, pw - 2
    assert out.shape == (n, co, ph, pw) and mask.shape == out.shape and (add is None or add.shape == out.shape)
    assert p.is_contiguous() and out.is_contiguous() and mask.is_contiguous() and (add is None or add.is_contiguous())
    lib = L.load()
    if TIMER is not None:
        global DETAIL
        DETAIL = "N%d %dx%dx%d -> %dx%dx%d mask+pad" % (n, ci, h, w, co, h, w)
    rc = [0]

    def call(*a):
        rc[0] = lib.vts_conv3x3_wide_mask_pad(*a)
        return 0 if rc[0] == L.ERR_UNSUPPORTED else rc[0]
    _run("conv3x3_wide", 4.0 * (p.numel() + out.numel() * (2 + (add is not None)) + wt.numel()), 2.0 * n * h * w * co * ci * 9, call,
         p.data_ptr(), wt.data_ptr(), out.data_ptr(), n, ci, co, h, w, L.ptr(add), mask.data_ptr(), L.stream())
    return rc[0] == 0


def zero_border(buf, pad=1):
    n, c, ph, pw = buf.shape
    _run("zero_border", 0.0, 0.0, L.load().vts_zero_border, buf.data_ptr(), n * c, ph - 2 * pad, pw - 2 * pad, pad, L.stream())
    return buf


def maxpool3s2_relu_pad(z, pad=0):
    """relu -> MaxPool2d(3, 2) (+ zero border): AlexNet's pooling stages (LPIPS-Alex metric)"""
    n, c, h, w = z.shape
    assert z.is_contiguous() and h >= 3 and w >= 3
    out = torch.empty(n, c, (h - 3) // 2 + 1 + 2 * pad, (w - 3) // 2 + 1 + 2 * pad, dtype=torch.float32, device=z.device)
    _run("maxpool3s2_relu_pad", 4.0 * (z.numel() + out.numel()), 0.0, L.load().vts_maxpool3s2_relu_pad, z.data_ptr(), n * c, h, w, pad, out.data_ptr(), L.stream())
    return out


def s2d4_pad(x, pad, oh, ow):
    """space-to-depth by 4 of the zero-padded x: [N, C, H, W] -> [N, 16 C, oh, ow] (AlexNet's stride-4 stem as a 3 x 3 convolution)"""
    n, c, h, w = x.shape
    assert x.is_contiguous()
    out = torch.empty(n, c * 16, oh, ow, dtype=torch.float32, device=x.device)
    _run("s2d4_pad", 4.0 * (x.numel() + out.numel()), 0.0, L.load().vts_s2d4_pad, x.data_ptr(), n, c, h, w, pad, oh, ow, out.data_ptr(), L.stream())
    return out


def maxpool2_relu_bwd(g, z, zpad=0, g2=None, pad=0):
    """g2 (this layer's tap gradient, z's layout) is added where z > 0; pad: zero-bordered output (the next adjoint's operand)"""
    n, c = z.shape[:2]
    h, w = z.shape[2] - 2 * zpad, z.shape[3] - 2 * zpad
    assert g.shape == (n, c, h // 2, w // 2) and g.is_contiguous() and z.is_contiguous() and (g2 is None or (g2.shape == z.shape and g2.is_contiguous()))
    gz = torch.empty(n, c, h + 2 * pad, w + 2 * pad, dtype=torch.float32, device=z.device)
    _run("maxpool2_relu_bwd", 4.0 * ((2 + (g2 is not None)) * gz.numel() + g.numel()), 0.0, L.load().vts_maxpool2_relu_bwd, g.data_ptr(), z.data_ptr(), n * c, h, w,
         gz.data_ptr(), zpad, L.ptr(g2), pad, L.stream())
    return gz


def relu_mask_pad(g, g2, z, pad=1, zpad=0):
    """g: dense [N, C, H, W]; g2 (a tap gradient): z's layout"""
    n, c = z.shape[:2]
    h, w = z.shape[2] - 2 * zpad, z.shape[3] - 2 * zpad
    assert g is None or (g.shape == (n, c, h, w) and g.is_contiguous())
    assert g2 is None or (g2.shape == z.shape and g2.is_contiguous())
    out = torch.empty(n, c, h + 2 * pad, w + 2 * pad, dtype=torch.float32, device=z.device)
    _run("relu_mask_pad", 4.0 * (n * c * h * w * (2 + (g is not None and g2 is not None)) + out.numel()), 0.0, L.load().vts_relu_mask_pad, L.ptr(g), L.ptr(g2),
         z.data_ptr(), n * c, h, w, pad, out.data_ptr(), zpad, L.stream())
    return out


def lpips_layer(z0, z1, w, coeff, loss_slot, dz0=None, grad_coeff=0.0, zpad=0):
    """zpad: z0 / z1 are padded [N, C, H + 2 zpad, W + 2 zpad] tensors read at their interior; dz0 (the gradient w.r.t. z0, ReLU mask
    applied) has z0's layout -- its border is the caller's"""
    n, c = z0.shape[:2]
    h, wd = z0.shape[2] - 2 * zpad, z0.shape[3] - 2 * zpad
    assert z1.shape == z0.shape and z0.is_contiguous() and z1.is_contiguous() and w.numel() == c
    assert dz0 is None or dz0.shape == z0.shape
    _run("lpips_layer", 4.0 * n * c * h * wd * (4 + 3 * (dz0 is not None)), 6.0 * n * c * h * wd, L.load().vts_lpips_layer, z0.data_ptr(), z1.data_ptr(), n, c, h * wd,
         w.data_ptr(), coeff, L.ptr(loss_slot), L.ptr(dz0), grad_coeff, wd, zpad, L.stream())


def l1_relu(za, zb, coeff, loss_slot, grad=None):
    assert za.shape == zb.shape and za.is_contiguous() and zb.is_contiguous()
    _run("l1_relu", 4.0 * za.numel() * (2 + (grad is not None)), 0.0, L.load().vts_l1_relu, za.data_ptr(), zb.data_ptr(), za.numel(), coeff, L.ptr(loss_slot),
         L.ptr(grad), L.stream())


def _triple(v):
    return (C.c_float * 3)(*[float(t) for t in v])


def lpips_input(x, shift, scale, nstride=None, channels=None, out=None):
    """LPIPS ScalingLayer output [N, 3, H, W] of x [N, 3 | 1, H, W] (a 1-channel VIEW of a wider tensor: pass its batch stride);
    out: a contiguous [N, 3, H, W] destination (a batch slice of a larger tensor)"""
    n, cx, h, w = x.shape if channels is None else (x.shape[0], channels, x.shape[2], x.shape[3])
    y = torch.empty(n, 3, h, w, dtype=torch.float32, device=x.device) if out is None else out
    assert y.shape == (n, 3, h, w) and y.is_contiguous()
    L.check(L.load().vts_lpips_input(x.data_ptr(), x.stride(0) if nstride is None else nstride, n, cx, h * w, _triple(shift), _triple(scale), y.data_ptr(), L.stream()),
            "vts_lpips_input")
    return y


def lpips_input_bwd(g, scale, dx, cx, accumulate=False, nstride=None):
    n, _, h, w = g.shape
    L.check(L.load().vts_lpips_input_bwd(g.data_ptr(), n, cx, h * w, _triple(scale), dx.data_ptr(), dx.stride(0) if nstride is None else nstride, int(accumulate),
                                         L.stream()), "vts_lpips_input_bwd")
    return dx
