// Adaptive instance normalisation of the style-code conditioning (reference thirdparty/AdaIN/function.py:4-23, used by
// CustomUnetGenerator.forward with --style_code_mode adain, models/networks.py:1624-1630):
//   out = (x - mean_x) / std_x * std_s + mean_s   per (n, c) over the H*W positions,
//   std = sqrt(UNBIASED variance + 1e-5) for both the content x and the style map s.
// The maps are the innermost U-Net features (6 x 6 at the reference's 1536 input): one wavefront owns a (n, c) group, reductions are
// wave shuffles in a fixed order (deterministic), the backward produces the gradients w.r.t. both operands in the same launch:
//   dx_i = std_s / std_x * (g_i - mean(g) - xhat_i * sum_j(g_j xhat_j) / (m - 1))
//   ds_i = mean(g) + sum_j(g_j xhat_j) * (s_i - mean_s) / ((m - 1) * std_s)
#include "vts_internal.h"

namespace {

__device__ __forceinline__ void group_stats(const float* __restrict__ p, int m, int lane, float eps, float& mean, float& stdv) {
  float s = 0.f;
  for (int i = lane; i < m; i += 64) s += p[i];
  mean = wave_sum(s) / (float)m;
  float q = 0.f;
  for (int i = lane; i < m; i += 64) {
    const float d = p[i] - mean;
    q += d * d;
  }
  stdv = sqrtf(wave_sum(q) / (float)(m - 1) + eps);
}

__global__ __launch_bounds__(256) void adain_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s, int groups, int m, float eps,
                                                         float* __restrict__ out) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= groups) return;
  const float *px = x + (int64_t)g * m, *ps = s + (int64_t)g * m;
  float mx, sx, ms, ss;
  group_stats(px, m, lane, eps, mx, sx);
  group_stats(ps, m, lane, eps, ms, ss);
  for (int i = lane; i < m; i += 64) out[(int64_t)g * m + i] = (px[i] - mx) / sx * ss + ms;
}

__global__ __launch_bounds__(256) void adain_bwd_kernel(const float* __restrict__ go, const float* __restrict__ x, const float* __restrict__ s, int groups,
                                                         int m, float eps, float* __restrict__ dx, float* __restrict__ ds) {
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (g >= groups) return;
  const float *px = x + (int64_t)g * m, *ps = s + (int64_t)g * m, *pg = go + (int64_t)g * m;
  float mx, sx, ms, ss;
  group_stats(px, m, lane, eps, mx, sx);
  group_stats(ps, m, lane, eps, ms, ss);
  float a = 0.f, b = 0.f;
  for (int i = lane; i < m; i += 64) {
    a += pg[i];
    b += pg[i] * ((px[i] - mx) / sx);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const float gm = a / (float)m, r = b / (float)(m - 1);
  for (int i = lane; i < m; i += 64) {
    const float xh = (px[i] - mx) / sx;
    dx[(int64_t)g * m + i] = ss / sx * (pg[i] - gm - xh * r);
    ds[(int64_t)g * m + i] = gm + r * (ps[i] - ms) / ss;
  }
}

}  // namespace

extern "C" int vts_adain(const float* x, const float* s, int NC, int HW, float eps, float* out, void* stream) {
  VTS_CHECK_ARG(x && s && out && NC >= 1 && HW >= 2, "vts_adain: bad args (unbiased variance needs HW >= 2)");
  hipLaunchKernelGGL(adain_fwd_kernel, dim3(cdiv(NC, 4)), dim3(256), 0, (hipStream_t)stream, x, s, NC, HW, eps, out);
  VTS_CHECK_LAUNCH("vts_adain");
  return VTS_OK;
}

extern "C" int vts_adain_bwd(const float* g, const float* x, const float* s, int NC, int HW, float eps, float* dx, float* ds, void* stream) {
  VTS_CHECK_ARG(g && x && s && dx && ds && NC >= 1 && HW >= 2, "vts_adain_bwd: bad args");
  hipLaunchKernelGGL(adain_bwd_kernel, dim3(cdiv(NC, 4)), dim3(256), 0, (hipStream_t)stream, g, x, s, NC, HW, eps, dx, ds);
  VTS_CHECK_LAUNCH("vts_adain_bwd");
  return VTS_OK;
}
