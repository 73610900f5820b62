// Implicit-GEMM 4x4 convolution family for gfx950 on the fp32 MFMA path
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chains, 157 TF peak).
//
// One kernel template covers nn.Conv2d forward, nn.ConvTranspose2d forward and both
// backward-data passes (vts.h: vts_conv4x4).  GEMM view per MFMA:
//   M = 16 consecutive output pixels of one row (of one output parity phase when the
//       operator is a stride-2 transposed conv),
//   N = 16 output channels,
//   K = 4 taps of one input channel: the 4 kx taps of one ky (conv, transposed s1) or the
//       2x2 taps that reach the phase (transposed s2).
// A workgroup (4 waves) owns a TY x TX tile of the phase grid and ALL output channels, so each
// input element is fetched from HBM once.  Per input-channel chunk the (haloed) input patch is
// staged in LDS with the producer's normalisation + activation applied on the fly and the two
// concat sources resolved (normalise-on-load: no separate InstanceNorm/BatchNorm/ReLU/cat pass
// ever touches HBM), the matching weight slice is staged tap-major, and every wave accumulates
// RW x MT x phases x NR 16x16 tiles in registers.
//
// LDS reads are conflict-free by construction: A fragments read stride-S words of one patch
// row (32 lanes -> 32 distinct banks or broadcast), B fragments read [k][cout] with the cout
// pitch = 16 (mod 32).
#include "vts_conv_kernel.h"

thread_local int t_stat_spl = 0;   // statistics slots per (n, channel) of the last launch (conv4x4_impl -> vts_norm_finalize_partials)

namespace {

// sums the k-split partials in slice order and applies the epilogue of the main kernel
__global__ __launch_bounds__(256) void conv_split_epilogue_kernel(const ConvK p, int KS) {
  const int co = blockIdx.y, n = blockIdx.z;
  const int64_t oplane = (int64_t)p.OH * p.OW;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= oplane) return;
  float v = 0.f;
  for (int ks = 0; ks < KS; ++ks) v += p.part[(((int64_t)ks * p.N + n) * p.Cout + co) * oplane + o];
  v += p.bias ? p.bias[co] : 0.f;
  if (p.act_out == VTS_ACT_TANH) v = tanhf(v);
  if (p.dm) {
    const float dsc = p.dmsc ? p.dmsc[n * p.dmC + co] : 1.f, dsh = p.dmsh ? p.dmsh[n * p.dmC + co] : 0.f;
    v *= vts_act_grad(p.dm[n * p.dmns + co * oplane + o] * dsc + dsh, p.dm_act);
  }
  float* ob = p.out + n * p.ons + co * oplane + o;
  *ob = p.accumulate ? *ob + v : v;
}

// Split epilogue + InstanceNorm statistics in one launch (round 2): the six innermost U-Net layers run k-split on maps of <= 64 x 64,
// where "sum the partials" and "statistics of the result" are two latency-bound launches on the forward's critical chain.  One
// workgroup owns a (n, channel) plane of <= 4096 elements: it sums the slices in slice order, adds the bias, stores the raw output
// and computes mean / biased variance from its registers -- same thread <-> element mapping and reduction order as
// norm_stats_fused_kernel<16, 256> (vts_norm.hip), so the statistics are bit-identical to the two-launch form.
struct InStatsOut {
  float *scale, *shift, *mean, *rstd;
  float eps;
};
__global__ __launch_bounds__(256) void conv_split_epilogue_in_kernel(const ConvK p, int KS, const InStatsOut q) {
  __shared__ float red[16];
  const int g = blockIdx.x, n = g / p.Cout, co = g - n * p.Cout;
  const int HW = p.OH * p.OW;
  const float bias = p.bias ? p.bias[co] : 0.f;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int o = e * 256 + threadIdx.x;
    float a = 0.f;
    if (o < HW) {
      for (int ks = 0; ks < KS; ++ks) a += p.part[(((int64_t)ks * p.N + n) * p.Cout + co) * HW + o];
      a += bias;
      p.out[n * p.ons + (int64_t)co * HW + o] = a;
    }
    v[e] = a;
  }
  const float cnt = (float)HW;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += v[e];
  const float mean = block_sum(s, red) / cnt;
  float m2 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float d = v[e] - mean;
    if (e * 256 + (int)threadIdx.x < HW) m2 += d * d;
  }
  m2 = block_sum(m2, red);
  if (threadIdx.x == 0) {
    const float rstd = 1.f / sqrtf(m2 / cnt + q.eps);
    q.scale[g] = rstd;
    q.shift[g] = -mean * rstd;
    if (q.mean) q.mean[g] = mean;
    if (q.rstd) q.rstd[g] = rstd;
  }
}

// Split epilogue + InstanceNorm BACKWARD in one launch (round 3): the backward-data convolutions of the inner U-Net layers run k-split and
// are followed by the normalisation backward of the layer below (norm_bwd_fused_kernel: one workgroup per (n, channel) plane, two block
// sums, one apply) -- two latency-bound launches on the generator update's critical chain.  One workgroup owns a plane of <= 4096
// elements: it sums the slices in slice order, applies the derivative mask (t = dmask * scale + shift IS the normalised value xhat of the
// layer below, scale = rstd) and the accumulation, reduces S1 = sum g, S2 = sum g * xhat and stores
//     dx = rstd * (g - S1 / HW - xhat * S2 / HW)           (InstanceNorm2d(affine=False) backward; reference models/networks.py:139)
// -- the gradient w.r.t. the RAW tensor, i.e. what vts_norm_bwd would have left in place.
__global__ __launch_bounds__(256) void conv_split_epilogue_inbwd_kernel(const ConvK p, int KS) {
  __shared__ float red[16];
  const int g = blockIdx.x, n = g / p.Cout, co = g - n * p.Cout;
  const int HW = p.OH * p.OW;
  const float bias = p.bias ? p.bias[co] : 0.f;
  const float dsc = p.dmsc[n * p.dmC + co], dsh = p.dmsh[n * p.dmC + co];
  const float* dmb = p.dm + n * p.dmns + (int64_t)co * HW;
  float* ob = p.out + n * p.ons + (int64_t)co * HW;
  float v[16], tn[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int o = e * 256 + threadIdx.x;
    float a = 0.f, t = 0.f;
    if (o < HW) {
      for (int ks = 0; ks < KS; ++ks) a += p.part[(((int64_t)ks * p.N + n) * p.Cout + co) * HW + o];
      a += bias;
      t = dmb[o] * dsc + dsh;
      a *= vts_act_grad(t, p.dm_act);
      if (p.accumulate) a += ob[o];
    }
    v[e] = a;
    tn[e] = t;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    s1 += v[e];
    s2 = fmaf(v[e], tn[e], s2);
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  const float m1 = s1 / (float)HW, m2 = s2 / (float)HW;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int o = e * 256 + threadIdx.x;
    if (o < HW) ob[o] = dsc * (v[e] - m1 - tn[e] * m2);
  }
}

// tile shape (RW, MT) of the full-width variants, as instantiated by VTS_DISPATCH below
inline void full_tile(int transposed, int stride, int nr, int& rw, int& mt) {
  static const int T[2][2][5][2] = {
      {{{2, 4}, {1, 4}, {1, 4}, {1, 2}, {1, 2}}, {{2, 4}, {1, 4}, {1, 4}, {1, 2}, {1, 2}}},   // conv   s1, s2
      {{{2, 4}, {1, 4}, {1, 4}, {1, 2}, {1, 2}}, {{1, 4}, {1, 2}, {1, 2}, {1, 1}, {1, 1}}}};  // convT  s1, s2
  rw = T[transposed][stride - 1][nr - 1][0];
  mt = T[transposed][stride - 1][nr - 1][1];
}

}  // namespace

extern "C" int64_t vts_conv4x4_ws_floats(const vts_conv_desc* d) {
  if (!d) return 0;
  const int nchunks = (d->in0.C + (d->in1.data ? d->in1.C : 0) + 3) / 4;
  const int ks_max = nchunks / 2 < 1 ? 1 : (nchunks / 2 > 320 ? 320 : nchunks / 2);
  return (int64_t)ks_max * d->N * d->Cout * d->OH * d->OW;
}

struct StatWs {
  float* p;
  int64_t floats;
};
static int conv4x4_impl(const vts_conv_desc* d, void* stream, const vts_norm_desc* nd, int* fused, StatWs sw, bool bsums = false, bool in_bwd_ok = false);
int vts_norm_finalize_partials(const vts_norm_desc* d, const float* part, int spl, hipStream_t st);   // vts_norm.hip

extern "C" int vts_conv4x4(const vts_conv_desc* d, void* stream) { return conv4x4_impl(d, stream, nullptr, nullptr, StatWs{nullptr, 0}); }

extern "C" int vts_conv4x4_in(const vts_conv_desc* d, const vts_norm_desc* nd, int* fused, void* stream) {
  VTS_CHECK_ARG(nd && fused && nd->scale && nd->shift && nd->mode == 0, "vts_conv4x4_in: InstanceNorm descriptor with scale / shift outputs required");
  *fused = 0;
  return conv4x4_impl(d, stream, nd, fused, StatWs{nullptr, 0});
}

// worst case over the tile shapes of the dispatch table (smallest tile: 4 x 16 positions of the phase grid), 4 wave slots per tile
extern "C" int64_t vts_conv4x4_norm_ws_floats(const vts_conv_desc* d) {
  if (!d) return 0;
  const bool ph4 = d->transposed && d->stride == 2;
  const int GH = ph4 ? (d->OH + 1) / 2 : d->OH, GW = ph4 ? (d->OW + 1) / 2 : d->OW;
  return (int64_t)d->N * d->Cout * 3 * 4 * cdiv(GH, 4) * cdiv(GW, 16);
}

extern "C" int vts_conv4x4_norm(const vts_conv_desc* d, const vts_norm_desc* nd, float* stat_ws, int64_t stat_ws_floats, int* fused, void* stream) {
  VTS_CHECK_ARG(nd && fused && nd->scale && nd->shift && (nd->mode == 0 || nd->mode == 1), "vts_conv4x4_norm: norm descriptor with scale / shift outputs required");
  *fused = 0;
  return conv4x4_impl(d, stream, nd, fused, StatWs{stat_ws, stat_ws_floats});
}

// backward-data convolution whose epilogue also emits the sums of the normalisation backward of the layer below (see ConvK::bsum_part);
// *slots = 0: plain convolution (the caller runs vts_norm_bwd), else the (S1, S2') pairs per (n, channel) in `part`
extern "C" int vts_conv4x4_bsums(const vts_conv_desc* d, float* part, int64_t part_floats, int* slots, void* stream) {
  VTS_CHECK_ARG(d && slots && part, "vts_conv4x4_bsums: null pointer");
  const bool in_bwd_ok = *slots == -1;     // the caller's normalisation is InstanceNorm2d(affine=False): its backward may be applied right here
  *slots = 0;
  int fused = 0;
  const int rc = conv4x4_impl(d, stream, nullptr, &fused, StatWs{part, part_floats}, true, in_bwd_ok);
  if (rc == VTS_OK && fused >= 2) *slots = fused - 2;
  if (rc == VTS_OK && fused == -1) *slots = -1;
  return rc;
}

extern "C" int vts_norm_stats_from_partials(const vts_norm_desc* nd, const float* part, int slots, void* stream) {
  return vts_norm_finalize_partials(nd, part, slots, (hipStream_t)stream);
}

static int dispatch_full(const vts_conv_desc* d, const ConvK& k, int nr, int N, hipStream_t st);

static int conv4x4_impl(const vts_conv_desc* d, void* stream, const vts_norm_desc* nd, int* fused, StatWs sw, bool bsums, bool in_bwd_ok) {
  VTS_CHECK_ARG(d && d->in0.data && d->w && d->out, "vts_conv4x4: null pointer");
  VTS_CHECK_ARG(d->stride == 1 || d->stride == 2, "vts_conv4x4: stride %d unsupported", d->stride);
  VTS_CHECK_ARG(d->Cout >= 1, "vts_conv4x4: Cout %d", d->Cout);
  if (d->Cout > 80) {
    // wide layers (pix2pixHD: up to 1024 channels): one launch per group of 80 output channels.  Every group
    // re-reads the input (from L2 / MALL at these map sizes); a GEMM-class kernel that tiles Cout is the
    // planned replacement for this loop.
    for (int c0 = 0; c0 < d->Cout; c0 += 80) {
      vts_conv_desc g = *d;
      const int64_t oplane = (int64_t)d->OH * d->OW;
      g.Cout = d->Cout - c0 < 80 ? d->Cout - c0 : 80;
      g.w = d->w + (int64_t)c0 * d->ws_co;
      g.bias = d->bias ? d->bias + c0 : nullptr;
      g.out = d->out + c0 * oplane;
      if (d->dmask.data) {
        g.dmask.data = d->dmask.data + c0 * oplane;
        g.dmask.scale = d->dmask.scale ? d->dmask.scale + c0 : nullptr;
        g.dmask.shift = d->dmask.shift ? d->dmask.shift + c0 : nullptr;
      }
      const int rc = conv4x4_impl(&g, stream, nullptr, nullptr, StatWs{nullptr, 0});
      if (rc != VTS_OK) return rc;
    }
    return VTS_OK;
  }
  VTS_CHECK_ARG(d->N >= 1 && d->IH >= 1 && d->IW >= 1 && d->OH >= 1 && d->OW >= 1, "vts_conv4x4: bad shape");
  VTS_CHECK_ARG(d->in0.C >= 1 && d->in1.C >= 0, "vts_conv4x4: bad channel counts");
  // The output window is the caller's: out[y, x] for y < OH, x < OW sums the taps that fall inside the input
  // (zero outside, on every side; pad may be negative).  That is what lets a K x K kernel (K <= 8) run as 4 x 4
  // blocks of its tap grid (vts_tap_embed): block (a, b) is this operator with pad - 4a / pad - 4b.  Only the
  // stride-2 transposed form is tied to its forward convolution (the phase decomposition assumes it).
  VTS_CHECK_ARG(d->OH <= d->IH * d->stride + 16 && d->OW <= d->IW * d->stride + 16 && d->pad >= -8 && d->pad <= 8,
                "vts_conv4x4: output %dx%d / pad %d implausible for input %dx%d s%d", d->OH, d->OW, d->pad, d->IH, d->IW, d->stride);
  VTS_CHECK_ARG(d->pad_dx >= -8 && d->pad_dx <= 8 && !(d->transposed && d->stride == 2 && d->pad_dx != 0), "vts_conv4x4: bad pad_dx %d", d->pad_dx);
  if (d->transposed && d->stride == 2) {
    // output size of a transposed conv is ambiguous (output_padding); require that the forward conv maps it back
    VTS_CHECK_ARG((d->OH + 2 * d->pad - 4) / d->stride + 1 == d->IH && (d->OW + 2 * d->pad - 4) / d->stride + 1 == d->IW && d->pad >= 0,
                  "vts_conv4x4: transposed output %dx%d inconsistent with input %dx%d s%d p%d", d->OH, d->OW, d->IH,
                  d->IW, d->stride, d->pad);
  }
  {
    const int rh0 = vts_conv_head_try(d, (hipStream_t)stream);      // Cout = 1 prediction heads: full-size and small-map members
    if (rh0 != VTS_ERR_UNSUPPORTED) return rh0;
    static const int use_small = vts_tune_set("VTS_NO_SMALL") ? 0 : 1;
    if (use_small) {
      const int rc = vts_conv_small_try(d, (hipStream_t)stream);
      if (rc != VTS_ERR_UNSUPPORTED) return rc;
    }
  }
  ConvK k;
  k.s0 = d->in0.data; k.sc0 = d->in0.scale; k.sh0 = d->in0.shift; k.ns0 = d->in0.nstride; k.C0 = d->in0.C;
  k.s1 = d->in1.data; k.sc1 = d->in1.scale; k.sh1 = d->in1.shift; k.ns1 = d->in1.nstride;
  k.C1 = d->in1.data ? d->in1.C : 0;
  k.Cin = k.C0 + k.C1;
  k.IH = d->IH; k.IW = d->IW; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout; k.pad = d->pad; k.padx = d->pad + d->pad_dx;
  k.w = d->w; k.ws_co = d->ws_co; k.ws_ci = d->ws_ci; k.bias = d->bias;
  k.out = d->out; k.ons = d->out_nstride;
  k.act_in = d->act_in; k.act_out = d->act_out;
  k.dm = d->dmask.data; k.dmsc = d->dmask.scale; k.dmsh = d->dmask.shift; k.dmns = d->dmask.nstride;
  k.dm_act = d->dmask_act; k.dmC = d->dmask.C;
  k.accumulate = d->accumulate;
  k.N = d->N; k.CG = 1; k.cps = 1 << 30; k.part = nullptr; k.tiles_x = 0; k.stat_part = nullptr; k.stat_spl = 0; k.bsum_part = nullptr;
#ifdef VTS_PROFILING
  static const int ablate = vts_tune("VTS_ABLATE", 0);
  k.ablate = ablate;
  k.trace = nullptr;
#endif
  static const int xcd_swizzle = vts_tune("VTS_XCD_SWIZZLE", 1);
  k.xcd_swizzle = xcd_swizzle;
  static const int direct_epi = vts_tune("VTS_DIRECT_EPI", 1);
  k.direct_epi = direct_epi && (int64_t)d->Cout * d->OH * d->OW * 4 < (int64_t)OOB_OFF && (!d->dmask.data || (int64_t)d->dmask.C * d->OH * d->OW * 4 < (int64_t)OOB_OFF);
  k.ident = vts_ident();
  VTS_CHECK_ARG(k.ident, "vts_conv4x4: could not allocate the identity constants");
  {
    const int64_t wfl = (int64_t)(d->Cout - 1) * d->ws_co + (int64_t)(k.Cin - 1) * d->ws_ci + 16;
    VTS_CHECK_ARG(wfl * 4 < (int64_t)OOB_OFF && (int64_t)d->IH * d->IW * 4 < (int64_t)OOB_OFF, "vts_conv4x4: tensor too large for 30-bit buffer offsets");
    k.wbytes = (int)(wfl * 4);
  }
  k.slope_in = vts_slope(d->act_in);
  k.identity_in = (d->act_in == VTS_ACT_NONE && !d->in0.scale && !d->in0.shift && !(d->in1.data && (d->in1.scale || d->in1.shift))) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const int nr = (d->Cout + 15) / 16;
  const int N = d->N;
  // statistics of the output in the epilogue (round 3): only the plain "store acc + bias" form through the direct epilogue
  static const int fuse_stats = vts_tune("VTS_FUSE_STATS", 1);
  const bool stats_ok = nd && fused && sw.p && fuse_stats && d->act_out == VTS_ACT_NONE && !d->dmask.data && !d->accumulate &&
                        nd->x == d->out && nd->N == N && nd->C == d->Cout && nd->HW == d->OH * d->OW && nd->nstride == d->out_nstride;
  const bool want_stats = stats_ok && k.direct_epi && sw.floats >= vts_conv4x4_norm_ws_floats(d);
  static const int fuse_bsums = vts_tune("VTS_FUSE_BSUMS", 1);
  const bool bsums_ok = bsums && fused && sw.p && fuse_bsums && d->act_out == VTS_ACT_NONE && d->dmask.data;
  const bool want_bsums = bsums_ok && k.direct_epi && sw.floats >= vts_conv4x4_norm_ws_floats(d);
  {
    // thin stride-2 convolutions / transposed convolutions on full-size maps: the lane = pixel members (vts_conv_px.hip), with the same
    // epilogue partials
    int spl = 0;
    const int rc = vts_conv_px_try(d, st, stats_ok ? sw.p : nullptr, (!stats_ok && bsums_ok) ? sw.p : nullptr, sw.floats, &spl);
    if (rc != VTS_ERR_UNSUPPORTED) {
      if (rc == VTS_OK && spl > 0) *fused = 2 + spl;
      return rc;
    }
    const int rt = vts_conv_thin_try(d, st);   // (what the lane = pixel members do not take: VTS_NO_PXT=1, Cout 13 .. 16)
    if (rt != VTS_ERR_UNSUPPORTED) return rt;
  }
  {
    // Small grids (inner U-Net layers: <= 32x32 maps, 80..592 channels) cannot fill 256 CUs with one
    // workgroup per spatial tile: split the output channels over workgroups (no reduction needed) and,
    // if that is still too few, the input-channel loop (k-split, deterministic two-kernel reduction).
    int rw, mt;
    full_tile(d->transposed, d->stride, nr, rw, mt);
    const bool ph4 = d->transposed && d->stride == 2;
    const int GH = ph4 ? (d->OH + 1) / 2 : d->OH, GW = ph4 ? (d->OW + 1) / 2 : d->OW;
    const int full_wgs = cdiv(GW, 16 * mt) * cdiv(GH, 4 * rw) * N;
    // input channels per pipeline step of the split instance: 8 where the slice is long enough (the per-chunk cost -- two barriers, the
    // staging stores, the descriptor arithmetic: 0.25 - 0.5 us measured -- is paid half as often); VTS_SPLIT_CK=4: always 4
    static const int split_ck = vts_tune("VTS_SPLIT_CK", 8);
    int ck = 4;
    const int nchunks = (k.Cin + 3) / 4;
    static const int small_thr = vts_tune("VTS_SMALL_WGS", 300);   // < ~1.2 workgroups per CU: split (measured: 128 -> 300 = step 7.88 -> 7.51 ms)
    static const int target_wgs = vts_tune("VTS_TARGET_WGS", 256);   // round 5 sweep on the step (same box, tools/ab_env.sh): 128 5.56, 192 - 256 5.51 - 5.53, 288 5.57, 320 (the round-2 value) 5.59, 512 5.59, 768 5.61 ms
    if (full_wgs < small_thr && (nr > 1 || nchunks >= 8)) {
      const int base = cdiv(GW, 32) * cdiv(GH, 4) * N * nr;
      int KS = target_wgs / base;
      if (KS > nchunks / 2) KS = nchunks / 2;
      if (KS < 1) KS = 1;
      int cps = cdiv(nchunks, KS);
      KS = cdiv(nchunks, cps);
      const int64_t need = (int64_t)KS * N * d->Cout * d->OH * d->OW;
      if (KS > 1 && (!d->ws || d->ws_floats < need)) { KS = 1; cps = nchunks; }
      // (the slices are cut in 4-channel units as before; an even slice of >= 16 channels runs as 8-channel steps: same partition, same
      //  accumulation order, bit-identical results)
      if (split_ck == 8 && k.Cin >= 32 && cps >= 4 && (cps % 2 == 0 || KS == 1)) { ck = 8; cps = (cps + 1) / 2; }   // (20 -> 40 at 256^2 measured 15 % slower in 8-channel steps)
      k.CG = nr; k.cps = cps; k.part = KS > 1 ? d->ws : nullptr;
      const bool cg_stats = want_stats && KS == 1;    // output-channel split only: every workgroup still stores final values
      if (cg_stats) k.stat_part = sw.p;
      const bool cg_bsums = want_bsums && KS == 1;
      if (cg_bsums) k.bsum_part = sw.p;
      int rc;
      if (!d->transposed) rc = d->stride == 2 ? vts_conv_split_m0s2(k, N, st, nr, KS, ck) : vts_conv_split_m0s1(k, N, st, nr, KS, ck);
      else rc = d->stride == 2 ? vts_conv_split_m1s2(k, N, st, nr, KS, ck) : vts_conv_split_m1s1(k, N, st, nr, KS, ck);
      if (rc == VTS_OK && (cg_stats || cg_bsums)) *fused = 2 + t_stat_spl;    // partials written: the caller merges them
      if (rc != VTS_OK || KS == 1) return rc;
      static const int fuse_in = vts_tune("VTS_FUSE_SPLIT_IN", 1);
      if (nd && nd->mode == 0 && fuse_in && (int64_t)d->OH * d->OW <= 4096 && d->act_out == VTS_ACT_NONE && !d->dmask.data && !d->accumulate &&
          nd->x == d->out && nd->N == N && nd->C == d->Cout && nd->HW == d->OH * d->OW && nd->nstride == d->out_nstride) {
        const InStatsOut q{nd->scale, nd->shift, nd->mean_out, nd->rstd_out, nd->eps};
        hipLaunchKernelGGL(conv_split_epilogue_in_kernel, dim3(N * d->Cout), dim3(256), 0, st, k, KS, q);
        VTS_CHECK_LAUNCH("vts_conv4x4 split epilogue + instance norm");
        vts_set_kernel("conv4x4_kernel<%d, %d, 1, 1, 2, %d, false, 0>+ksplit+in", d->transposed ? 1 : 0, d->stride, ck);
        *fused = 1;
        return VTS_OK;
      }
      static const int fuse_inbwd = vts_tune("VTS_FUSE_SPLIT_INBWD", 1);
      if (bsums && in_bwd_ok && fused && fuse_inbwd && (int64_t)d->OH * d->OW <= 4096 && d->act_out == VTS_ACT_NONE && d->dmask.data && d->dmask.scale &&
          d->dmask.shift && d->dmask.C == d->Cout) {
        hipLaunchKernelGGL(conv_split_epilogue_inbwd_kernel, dim3(N * d->Cout), dim3(256), 0, st, k, KS);
        VTS_CHECK_LAUNCH("vts_conv4x4 split epilogue + instance norm backward");
        vts_set_kernel("conv4x4_kernel<%d, %d, 1, 1, 2, %d, false, 0>+ksplit+inbwd", d->transposed ? 1 : 0, d->stride, ck);
        *fused = -1;
        return VTS_OK;
      }
      hipLaunchKernelGGL(conv_split_epilogue_kernel, dim3((unsigned)cdiv64((int64_t)d->OH * d->OW, 256), d->Cout, N), dim3(256), 0, st, k, KS);
      VTS_CHECK_LAUNCH("vts_conv4x4 split epilogue");
      return VTS_OK;
    }
  }
  if (want_stats) k.stat_part = sw.p;
  if (want_bsums) k.bsum_part = sw.p;
  const int rc = dispatch_full(d, k, nr, N, st);
  if (rc != VTS_OK || !(want_stats || want_bsums)) return rc;
  *fused = 2 + t_stat_spl;      // partials written: the caller merges them (vts_norm_stats_from_partials)
  return VTS_OK;
}

static int dispatch_full(const vts_conv_desc* d, const ConvK& k, int nr, int N, hipStream_t st) {
  if (!d->transposed) return d->stride == 2 ? vts_conv_full_m0s2(k, nr, N, st) : vts_conv_full_m0s1(k, nr, N, st);
  return d->stride == 2 ? vts_conv_full_m1s2(k, nr, N, st) : vts_conv_full_m1s1(k, nr, N, st);
}
