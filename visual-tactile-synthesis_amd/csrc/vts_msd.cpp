// Network-level entry points (`vts_msd_fwd`), round 6: the forward of the reference's discriminators
//   NLayerDiscriminator.forward        models/networks.py:1696-1750   -> vts_patchgan_forward (one PatchGAN = one scale)
//   MultiscaleDiscriminator.forward    models/networks.py:1649-1691   -> vts_msd_forward (num_D PatchGANs over an average-pooled pyramid)
// in TRAINING mode -- BatchNorm2d normalises with the statistics of the batch and advances its running statistics, as every discriminator
// call of a reference training step does (models/sinskitG_model.py:1361, 1374, 1490, 1567, 1584, 1781) -- as ONE C call over the library's
// own operators, forward only (nothing is kept for a backward).  The Python product takes this path for its forward-only discriminator
// passes (the full-resolution D2 visualisation pass, the D2 term of the generator step: vts/engine.py:_scale_lane) and drives the same
// operators layer by layer where a backward follows; the two are bit-identical (tests/test_network_abi_gpu.py).  The schedule of one PatchGAN:
//   conv 0 (stride 2)                                   raw output; LeakyReLU(0.2) is applied by the next convolution on load
//   conv j (stride 2, the last but one stride 1) -> BatchNorm2d: statistics from the convolution's epilogue (vts_conv4x4_norm) or a
//                                                      statistics pass; scale / shift are applied by the next convolution on load
//   conv n-1 (stride 1, 1 channel)                      the prediction map (skipped with run_head = 0: a pass that exists for the
//                                                      BatchNorm running statistics at this scale)
// No allocation: the caller passes vts_patchgan_forward_ws_floats(d) / vts_msd_forward_ws_floats(d) floats.
#include <stdint.h>

#include <algorithm>

#include "vts_internal.h"

namespace {

struct PgPlan {
  int n;                                          // convolutions that run
  int oh[VTS_PATCHGAN_MAX_CONVS], ow[VTS_PATCHGAN_MAX_CONVS];
  int64_t act_off[VTS_PATCHGAN_MAX_CONVS];        // raw convolution outputs (the head writes d->pred)
  int64_t stat_off[VTS_PATCHGAN_MAX_CONVS];       // [4][N * C]: scale, shift, mean, rstd of a normalised layer
  int64_t conv_ws, conv_ws_floats, stat_ws, stat_ws_floats;
  int64_t total;
};

int pg_check(const vts_patchgan_desc* d, const char* who) {
  VTS_CHECK_ARG(d, "%s: null descriptor", who);
  VTS_CHECK_ARG(d->n_convs >= 2 && d->n_convs <= VTS_PATCHGAN_MAX_CONVS, "%s: n_convs %d (2 .. %d)", who, d->n_convs, VTS_PATCHGAN_MAX_CONVS);
  VTS_CHECK_ARG(d->N >= 1 && d->H >= 1 && d->W >= 1, "%s: bad shape N %d, %d x %d", who, d->N, d->H, d->W);
  VTS_CHECK_ARG(d->in0.data && d->in0.C >= 1 && (d->in1.C == 0 || d->in1.data), "%s: null input", who);
  for (int j = 0; j < d->n_convs; ++j) {
    VTS_CHECK_ARG(d->w[j] && d->cout[j] >= 1 && (d->stride[j] == 1 || d->stride[j] == 2), "%s: convolution %d incomplete (cout %d, stride %d)", who, j,
                  d->cout[j], d->stride[j]);
    VTS_CHECK_ARG(!(d->gamma[j] || d->beta[j]) || (d->gamma[j] && d->beta[j]), "%s: BatchNorm of convolution %d needs both weight and bias", who, j);
    VTS_CHECK_ARG(!d->running_mean[j] == !d->running_var[j], "%s: running_mean / running_var of convolution %d come as a pair", who, j);
    VTS_CHECK_ARG(!d->stat_mean_out[j] == !d->stat_uvar_out[j], "%s: stat_mean_out / stat_uvar_out of convolution %d come as a pair", who, j);
  }
  VTS_CHECK_ARG(!d->gamma[0] && !d->gamma[d->n_convs - 1], "%s: the first and the last convolution carry no BatchNorm (networks.py:1703-1741)", who);
  VTS_CHECK_ARG(!d->run_head || d->pred, "%s: run_head without a prediction buffer", who);
  return VTS_OK;
}

// the convolution descriptors of the schedule, in launch order: shared by the workspace planner and the launcher
void pg_layers(const vts_patchgan_desc* d, float* ws, const PgPlan& P, vts_conv_desc* L) {
  vts_operand cur0 = d->in0, cur1 = d->in1;
  int h = d->H, w = d->W;
  for (int j = 0; j < P.n; ++j) {
    vts_conv_desc& c = L[j];
    c = vts_conv_desc{};
    c.in0 = cur0; c.in1 = cur1;
    const int cin = cur0.C + cur1.C;
    c.N = d->N; c.IH = h; c.IW = w; c.OH = P.oh[j]; c.OW = P.ow[j]; c.Cout = d->cout[j];
    c.stride = d->stride[j]; c.pad = 2; c.transposed = 0;
    c.w = d->w[j]; c.ws_co = cin * 16; c.ws_ci = 16; c.bias = d->b[j];
    const bool head = j == d->n_convs - 1;
    c.out = head ? d->pred : ws + P.act_off[j];
    c.out_nstride = (int64_t)c.Cout * c.OH * c.OW;
    c.act_in = j ? VTS_ACT_LRELU : VTS_ACT_NONE; c.act_out = VTS_ACT_NONE;
    cur0 = vts_operand{};
    cur0.data = c.out; cur0.C = c.Cout; cur0.nstride = c.out_nstride;
    if (d->gamma[j]) { cur0.scale = ws + P.stat_off[j]; cur0.shift = ws + P.stat_off[j] + (int64_t)d->N * c.Cout; }
    cur1 = vts_operand{};
    h = c.OH; w = c.OW;
  }
}

void pg_plan(const vts_patchgan_desc* d, PgPlan& P) {
  int64_t off = 0;
  auto take = [&](int64_t n) { const int64_t o = off; off += (n + 63) / 64 * 64; return o; };
  P.n = d->run_head ? d->n_convs : d->n_convs - 1;
  int h = d->H, w = d->W;
  for (int j = 0; j < P.n; ++j) {
    // Conv2d(kernel 4, stride s, padding 2): networks.py:1703, 1716, 1728, 1739
    P.oh[j] = (h + 4 - 4) / d->stride[j] + 1; P.ow[j] = (w + 4 - 4) / d->stride[j] + 1;
    h = P.oh[j]; w = P.ow[j];
    P.act_off[j] = P.stat_off[j] = 0;
    if (j != d->n_convs - 1) P.act_off[j] = take((int64_t)d->N * d->cout[j] * h * w);
    if (d->gamma[j]) P.stat_off[j] = take(4 * (int64_t)d->N * d->cout[j]);
  }
  vts_conv_desc L[VTS_PATCHGAN_MAX_CONVS];
  pg_layers(d, reinterpret_cast<float*>(uintptr_t(4096)), P, L);      // (a placeholder base: only the shapes are read)
  int64_t cw = 0, sw = 0;
  for (int j = 0; j < P.n; ++j) {
    const vts_conv_desc& c = L[j];
    if ((int64_t)c.OH * c.OW <= 64 * 64) cw = std::max(cw, vts_conv4x4_ws_floats(&c));
    if (d->gamma[j]) {
      sw = std::max(sw, vts_conv4x4_norm_ws_floats(&c));
      cw = std::max(cw, vts_norm_ws_floats(c.N, c.Cout, c.OH * c.OW));
    }
  }
  P.conv_ws_floats = cw; P.stat_ws_floats = sw;
  P.conv_ws = take(cw); P.stat_ws = take(sw);
  P.total = off;
}

int pg_run(const vts_patchgan_desc* d, float* ws, const PgPlan& P, void* stream) {
  vts_conv_desc L[VTS_PATCHGAN_MAX_CONVS];
  pg_layers(d, ws, P, L);
  for (int j = 0; j < P.n; ++j) {
    vts_conv_desc& c = L[j];
    if ((int64_t)c.OH * c.OW <= 64 * 64) { c.ws = ws + P.conv_ws; c.ws_floats = P.conv_ws_floats; }
    if (!d->gamma[j]) {
      const int rc = vts_conv4x4(&c, stream);
      if (rc != VTS_OK) return rc;
      continue;
    }
    vts_norm_desc nd{};
    const int64_t NC = (int64_t)c.N * c.Cout;
    float* stt = ws + P.stat_off[j];
    nd.x = c.out; nd.nstride = c.out_nstride; nd.N = c.N; nd.C = c.Cout; nd.HW = c.OH * c.OW; nd.mode = 1;
    nd.eps = d->eps; nd.momentum = d->momentum;
    nd.gamma = d->gamma[j]; nd.beta = d->beta[j];
    nd.running_mean = d->running_mean[j]; nd.running_var = d->running_var[j]; nd.num_batches_tracked = d->num_batches_tracked[j];
    nd.stat_mean_out = d->stat_mean_out[j]; nd.stat_uvar_out = d->stat_uvar_out[j];
    nd.scale = stt; nd.shift = stt + NC; nd.mean_out = stt + 2 * NC; nd.rstd_out = stt + 3 * NC;
    int fused = 0;
    int rc = vts_conv4x4_norm(&c, &nd, ws + P.stat_ws, P.stat_ws_floats, &fused, stream);
    if (rc != VTS_OK) return rc;
    if (fused >= 2) rc = vts_norm_stats_from_partials(&nd, ws + P.stat_ws, fused - 2, stream);
    else if (fused == 0) rc = vts_norm_stats(&nd, ws + P.conv_ws, stream);
    if (rc != VTS_OK) return rc;
  }
  return VTS_OK;
}

struct MsdPlan {
  vts_patchgan_desc s[VTS_MSD_MAX_SCALES];      // the scales with their inputs resolved (pooled levels in the workspace)
  PgPlan p[VTS_MSD_MAX_SCALES];
  int64_t pool_off[VTS_MSD_MAX_SCALES][2];      // pooled level of in0 / in1 for scale >= 1
  int64_t scale_ws[VTS_MSD_MAX_SCALES];
  int64_t total;
};

int msd_check(const vts_msd_desc* d) {
  VTS_CHECK_ARG(d, "vts_msd_forward: null descriptor");
  VTS_CHECK_ARG(d->num_D >= 1 && d->num_D <= VTS_MSD_MAX_SCALES, "vts_msd_forward: num_D %d (1 .. %d)", d->num_D, VTS_MSD_MAX_SCALES);
  VTS_CHECK_ARG(d->scale[0].in0.data && d->scale[0].in0.C >= 1, "vts_msd_forward: scale 0 carries the input");
  return VTS_OK;
}

int msd_plan(const vts_msd_desc* d, float* ws, MsdPlan& M) {
  int64_t off = 0;
  auto take = [&](int64_t n) { const int64_t o = off; off += (n + 63) / 64 * 64; return o; };
  int h = d->scale[0].H, w = d->scale[0].W;
  const int N = d->scale[0].N, c0 = d->scale[0].in0.C, c1 = d->scale[0].in1.C;
  for (int s = 0; s < d->num_D; ++s) {
    M.s[s] = d->scale[s];
    M.s[s].N = N; M.s[s].H = h; M.s[s].W = w;
    if (s > 0) {
      // AvgPool2d(3, stride 2, padding 1, count_include_pad False) of the level above (networks.py:1670, 1688)
      M.pool_off[s][0] = take((int64_t)N * c0 * h * w);
      M.pool_off[s][1] = c1 ? take((int64_t)N * c1 * h * w) : 0;
      vts_operand a{}, b{};
      a.data = ws + M.pool_off[s][0]; a.C = c0; a.nstride = (int64_t)c0 * h * w;
      if (c1) { b.data = ws + M.pool_off[s][1]; b.C = c1; b.nstride = (int64_t)c1 * h * w; }
      M.s[s].in0 = a; M.s[s].in1 = b;
    }
    const int rc = pg_check(&M.s[s], "vts_msd_forward");
    if (rc != VTS_OK) return rc;
    pg_plan(&M.s[s], M.p[s]);
    M.scale_ws[s] = take(M.p[s].total);
    h = (h + 1) / 2; w = (w + 1) / 2;
  }
  M.total = off;
  return VTS_OK;
}

}  // namespace

extern "C" int64_t vts_patchgan_forward_ws_floats(const vts_patchgan_desc* d) {
  if (pg_check(d, "vts_patchgan_forward") != VTS_OK) return -1;
  PgPlan P{};
  pg_plan(d, P);
  return P.total;
}

extern "C" int vts_patchgan_forward(const vts_patchgan_desc* d, float* ws, int64_t ws_floats, void* stream) {
  const int rc = pg_check(d, "vts_patchgan_forward");
  if (rc != VTS_OK) return rc;
  PgPlan P{};
  pg_plan(d, P);
  VTS_CHECK_ARG(ws && ws_floats >= P.total, "vts_patchgan_forward: workspace of %lld floats, need %lld", (long long)ws_floats, (long long)P.total);
  return pg_run(d, ws, P, stream);
}

extern "C" int64_t vts_msd_forward_ws_floats(const vts_msd_desc* d) {
  if (msd_check(d) != VTS_OK) return -1;
  MsdPlan M{};
  if (msd_plan(d, reinterpret_cast<float*>(uintptr_t(4096)), M) != VTS_OK) return -1;
  return M.total;
}

extern "C" int vts_msd_forward(const vts_msd_desc* d, float* ws, int64_t ws_floats, void* stream) {
  int rc = msd_check(d);
  if (rc != VTS_OK) return rc;
  MsdPlan M{};
  rc = msd_plan(d, ws, M);
  if (rc != VTS_OK) return rc;
  VTS_CHECK_ARG(ws && ws_floats >= M.total, "vts_msd_forward: workspace of %lld floats, need %lld", (long long)ws_floats, (long long)M.total);
  for (int s = 0; s < d->num_D; ++s) {
    if (s > 0) {
      const vts_patchgan_desc& up = M.s[s - 1];
      // a lazily normalised operand cannot be pooled as stored: the pyramid is built from the RAW inputs, as the reference pools them
      VTS_CHECK_ARG(!up.in0.scale && !up.in0.shift && !up.in1.scale && !up.in1.shift, "vts_msd_forward: the input operands must be plain tensors (no scale / shift)");
      rc = vts_avgpool3s2(up.in0.data, up.in0.nstride, up.N, up.in0.C, up.H, up.W, const_cast<float*>(M.s[s].in0.data), stream);
      if (rc == VTS_OK && up.in1.C) rc = vts_avgpool3s2(up.in1.data, up.in1.nstride, up.N, up.in1.C, up.H, up.W, const_cast<float*>(M.s[s].in1.data), stream);
      if (rc != VTS_OK) return rc;
    }
    rc = pg_run(&M.s[s], ws + M.scale_ws[s], M.p[s], stream);
    if (rc != VTS_OK) return rc;
  }
  return VTS_OK;
}
