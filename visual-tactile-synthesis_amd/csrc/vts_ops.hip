// Loss, data-movement and optimiser kernels of the training step (all HBM-bound or tiny):
// AvgPool pyramid, GAN loss, L1, patch gather / deterministic scatter, generator output
// post-processing (mask, normals, DiffAugment), positional encoding, "more fake T" sampler
// support, fused Adam, PatchNCE.
#include <stdlib.h>

#include "vts_internal.h"

namespace {

// ------------------------------------------------------------------ AvgPool2d(3,2,1,count_include_pad=False)
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, int64_t xns, int C, int H, int W, int OH, int OW,
                                                      float* __restrict__ y) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z % C, n = blockIdx.z / C;
  if (ox >= OW || oy >= OH) return;
  const float* p = x + n * xns + (int64_t)c * H * W;
  float s = 0.f;
  int cnt = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int iy = 2 * oy + dy;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int ix = 2 * ox + dx;
      if (ix < 0 || ix >= W) continue;
      s += p[(int64_t)iy * W + ix];
      ++cnt;
    }
  }
  y[(((int64_t)n * C + c) * OH + oy) * OW + ox] = s / (float)cnt;
}

// Even-width planes with 8-byte aligned rows (every level of the discriminators' input pyramids): a lane owns output column ox
// and FOUR output rows -- nine input rows, each one coalesced 8-byte load of columns (2 ox, 2 ox + 1); column 2 ox - 1 comes from the
// lane on the left (one extra 4-byte load in lane 0 of the wave).  Every input element is fetched ~1.1 times in full lines; the
// one-output-per-thread kernel above reads each line three times at half efficiency (57 us for 134 MB; round 3).
__global__ __launch_bounds__(256) void avgpool_rows4_kernel(const float* __restrict__ x, int64_t xns, int C, int H, int W, int OH, int OW,
                                                            float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ox = blockIdx.x * 64 + lane, oy0 = (blockIdx.y * 4 + wave) * 4;
  const int c = blockIdx.z % C, n = blockIdx.z / C;
  if (oy0 >= OH) return;                                  // wave-uniform
  const float* p = x + n * xns + (int64_t)c * H * W;
  const int ixc = min(2 * ox, W - 2);                     // lanes beyond the row re-read its last pair (their outputs are not stored)
  const bool has_left = 2 * ox - 1 >= 0;
  float rs[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int iy = 2 * oy0 - 1 + r;
    const bool rok = iy >= 0 && iy < H;                   // wave-uniform
    const int iyc = min(max(iy, 0), H - 1);
    const float2 v = *reinterpret_cast<const float2*>(p + (int64_t)iyc * W + ixc);
    float l = __shfl_up(v.y, 1);
    if (lane == 0) l = has_left ? p[(int64_t)iyc * W + ixc - 1] : 0.f;
    rs[r] = rok ? (has_left ? l : 0.f) + v.x + v.y : 0.f;
  }
  if (ox >= OW) return;
  const int cx = (has_left ? 1 : 0) + 1 + (2 * ox + 1 < W ? 1 : 0);
  float* q = y + (((int64_t)n * C + c) * OH) * OW + ox;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = oy0 + j;
    if (oy < OH) {
      const int cy = (2 * oy - 1 >= 0 ? 1 : 0) + 1 + (2 * oy + 1 < H ? 1 : 0);
      q[(int64_t)oy * OW] = (rs[2 * j] + rs[2 * j + 1] + rs[2 * j + 2]) / (float)(cx * cy);
    }
  }
}

__device__ __forceinline__ int pool_cnt(int o, int L) {  // valid taps of window o along one axis
  int c = 0;
  for (int d = -1; d <= 1; ++d) {
    const int i = 2 * o + d;
    c += (i >= 0 && i < L);
  }
  return c;
}

// adjoint of AvgPool2d(3, 2, padding 1, count_include_pad False) at fine position (y, x): the sum over the windows that cover it.
// One function for avgpool_bwd_kernel and g_out_grad_kernel's fused merge, so that both evaluate the same expression (bit-identical).
__device__ __forceinline__ float avgpool_adjoint_at(const float* __restrict__ g, int y, int x, int H, int W, int OH, int OW) {
  // windows covering y: oy with |y - 2 oy| <= 1 (one for even y, two for odd y); a window has 3 taps per axis minus the ones outside
  // the map, so its weight is 1 / (cy * cx) with cy, cx in {1, 2, 3}
  const int oy_lo = y >> 1, oy_hi = min((y + 1) >> 1, OH - 1), ox_lo = x >> 1, ox_hi = min((x + 1) >> 1, OW - 1);
  auto rcnt = [](int o, int L) { return 1.f / (float)(3 - (o == 0) - (2 * o + 1 >= L)); };
  const float wy0 = rcnt(oy_lo, H), wy1 = oy_hi > oy_lo ? rcnt(oy_hi, H) : 0.f;
  const float wx0 = rcnt(ox_lo, W), wx1 = ox_hi > ox_lo ? rcnt(ox_hi, W) : 0.f;
  const float* r0 = g + (int64_t)oy_lo * OW;
  const float* r1 = g + (int64_t)oy_hi * OW;
  return wy0 * (r0[ox_lo] * wx0 + r0[ox_hi] * wx1) + wy1 * (r1[ox_lo] * wx0 + r1[ox_hi] * wx1);
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, int C, int H, int W, int OH, int OW,
                                                          float* __restrict__ dx, int64_t dxns, int accumulate) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z % C, n = blockIdx.z / C;
  if (x >= W || y >= H) return;
  const float s = avgpool_adjoint_at(dy + ((int64_t)n * C + c) * OH * OW, y, x, H, W, OH, OW);
  float* o = dx + n * dxns + ((int64_t)c * H + y) * W + x;
  *o = accumulate ? *o + s : s;
}

// ------------------------------------------------------------------ GAN loss
__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // F.softplus, threshold 20
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// Loss slots are 64-bit fixed point (VTS_LOSS_SCALE = 2^40 units per 1.0): integer atomic adds commute, so a logged loss is bitwise
// reproducible whatever order the workgroups -- and the concurrent lanes that share a slot -- arrive in (float atomics were not).
__device__ __forceinline__ void loss_add(long long* slot, double v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)llrint(v * VTS_LOSS_SCALE));
}

__global__ __launch_bounds__(256) void ganloss_kernel(const float* __restrict__ pred, int64_t total, int mode, int real, float label,
                                                      float inv_count, float coeff, float gcoeff, long long* __restrict__ loss, float* __restrict__ dpred) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float p = pred[i];
    float l, g;
    switch (mode) {
      case 0: l = real ? softplus_t(-p) : softplus_t(p); g = real ? -sigmoid_f(-p) : sigmoid_f(p); break;
      case 1: l = (p - label) * (p - label); g = 2.f * (p - label); break;
      case 2: l = fmaxf(p, 0.f) - p * label + log1pf(expf(-fabsf(p))); g = sigmoid_f(p) - label; break;
      case 3: l = real ? -p : p; g = real ? -1.f : 1.f; break;
      case 5: {   // 'vanilla' behind the PatchGAN's trailing Sigmoid (reference networks.py:1659, 1731-1732): BCE-with-logits OF sigmoid(p)
        const float q = sigmoid_f(p);
        l = q - q * label + log1pf(expf(-q));      // q > 0
        g = (sigmoid_f(q) - label) * q * (1.f - q);
        break;
      }
      default: {
        const float t = real ? 1.f - p : 1.f + p;
        l = fmaxf(t, 0.f);
        g = t > 0.f ? (real ? -1.f : 1.f) : 0.f;
      }
    }
    acc += l;
    if (dpred) dpred[i] = g * inv_count * gcoeff;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) loss_add(loss, (double)acc * (double)inv_count * (double)coeff);
}

// ------------------------------------------------------------------ L1
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float coeff,
                                                 long long* __restrict__ loss, float* __restrict__ grad, int accumulate, int vec) {
  __shared__ float red[16];
  float acc = 0.f;
  if (vec) {   // n % 4 == 0 and 16-byte aligned operands: one 16-byte load per operand and lane (round 3: 3.6 -> 5+ TB/s)
    const int64_t n4 = n >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    f32x4* g4 = reinterpret_cast<f32x4*>(grad);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const f32x4 d = a4[i] - b4[i];
      f32x4 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += fabsf(d[e]);
        g[e] = coeff * (d[e] > 0.f ? 1.f : (d[e] < 0.f ? -1.f : 0.f));
      }
      if (grad) g4[i] = accumulate ? g4[i] + g : g;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const float d = a[i] - b[i];
      acc += fabsf(d);
      if (grad) {
        const float g = coeff * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        grad[i] = accumulate ? grad[i] + g : g;
      }
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) loss_add(loss, (double)acc * (double)coeff);
}

// ------------------------------------------------------------------ patches
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ src, int64_t sns, int C, int H, int W,
                                                           const int* __restrict__ img, const int* __restrict__ offx,
                                                           const int* __restrict__ offy, int size, float* __restrict__ out, int outC,
                                                           int c0) {
  const int p = blockIdx.x, c = blockIdx.y;
  const float* s = src + img[p] * sns + (int64_t)c * H * W;
  const int ox = offx[p], oy = offy[p];
  float* o = out + (((int64_t)p * outC + c0 + c) * size) * size;
  for (int i = threadIdx.x; i < size * size; i += 256) {
    const int y = i / size, x = i - y * size;
    const int sy = min(max(oy + y, 0), H - 1), sx = min(max(ox + x, 0), W - 1);
    o[i] = s[(int64_t)sy * W + sx];
  }
}

// Batched form (round 3): every channel run of the D2 patch stacks in ONE launch.  A job copies C channels of P patches into channel
// slot c0 of a [P, dstC, size, size] tensor: gathered with clamp-to-border from an image tensor (img / offx / offy given), copied from
// a [P, C, size, size] patch tensor (img NULL), or filled with a constant (src NULL).  blockIdx.y walks the (job, channel) pairs.
constexpr int PJ_MAX = 16;
struct PatchJobs {
  int njobs;
  int ch_start[PJ_MAX + 1];
  vts_patch_job job[PJ_MAX];
};
__global__ __launch_bounds__(256) void patch_jobs_kernel(const PatchJobs t, int size) {
  int j = 0;
  while (j + 1 < t.njobs && (int)blockIdx.y >= t.ch_start[j + 1]) ++j;   // uniform
  const vts_patch_job& q = t.job[j];
  const int p = blockIdx.x, c = blockIdx.y - t.ch_start[j];
  if (p >= q.P) return;
  const int S2 = size * size;
  float* o = q.dst + ((int64_t)p * q.dst_C + q.dst_c0 + c) * S2;
  if (!q.src) {
    for (int i = threadIdx.x; i < S2; i += 256) o[i] = q.fill;
  } else if (!q.img) {
    const float* s = q.src + (int64_t)p * q.src_nstride + (int64_t)c * S2;
    for (int i = threadIdx.x; i < S2; i += 256) o[i] = s[i];
  } else {
    const float* s = q.src + q.img[p] * q.src_nstride + (int64_t)c * q.H * q.W;
    const int ox = q.offx[p], oy = q.offy[p];
    for (int i = threadIdx.x; i < S2; i += 256) {
      const int y = i / size, x = i - y * size;
      const int sy = min(max(oy + y, 0), q.H - 1), sx = min(max(ox + x, 0), q.W - 1);
      o[i] = s[(int64_t)sy * q.W + sx];
    }
  }
}

// range of patch-local indices j in [0,size) with clamp(off + j, 0, L-1) == v
__device__ __forceinline__ void inv_clamp(int v, int off, int size, int L, int& lo, int& hi) {
  lo = hi = v - off;
  if (v == 0) lo = 0;                 // everything that clamps up to 0
  if (v == L - 1) hi = size - 1;      // everything that clamps down to L-1
  lo = max(lo, 0);
  hi = min(hi, size - 1);
}

// deterministic scatter-add: each 32x32 tile of the destination lists the patches that touch it
// (in patch order) and every pixel sums its contributions in that order.
__global__ __launch_bounds__(256) void patch_scatter_kernel(const float* __restrict__ dp, int dpC, int c0, int C,
                                                            const int* __restrict__ offx, const int* __restrict__ offy, int ppi,
                                                            int size, float* __restrict__ dsrc, int64_t dns, int H, int W,
                                                            int accumulate) {
  extern __shared__ int list[];  // ppi ints
  __shared__ int count;
  const int n = blockIdx.z, tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    int base = 0;
    for (int p0 = 0; p0 < ppi; p0 += 64) {
      const int p = p0 + lane;
      bool hit = false;
      if (p < ppi) {
        const int ox = offx[n * ppi + p], oy = offy[n * ppi + p];
        const int x_lo = min(max(ox, 0), W - 1), x_hi = min(max(ox + size - 1, 0), W - 1);
        const int y_lo = min(max(oy, 0), H - 1), y_hi = min(max(oy + size - 1, 0), H - 1);
        hit = x_lo < tx0 + 32 && x_hi >= tx0 && y_lo < ty0 + 32 && y_hi >= ty0;
      }
      const unsigned long long m = __ballot(hit);
      if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = p;
      base += __popcll(m);
    }
    if (lane == 0) count = base;
  }
  __syncthreads();
  const int cnt = count;
  if (cnt == 0 && accumulate) return;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int y = ty0 + (i >> 5), x = tx0 + (i & 31);
    if (y >= H || x >= W) continue;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int q = 0; q < cnt; ++q) {
        const int p = n * ppi + list[q];
        int jl, jh, il, ih;
        inv_clamp(y, offy[p], size, H, jl, jh);
        inv_clamp(x, offx[p], size, W, il, ih);
        const float* g = dp + (((int64_t)p * dpC + c0 + c) * size) * size;
        for (int j = jl; j <= jh; ++j)
          for (int ii = il; ii <= ih; ++ii) s += g[j * size + ii];
      }
      float* o = dsrc + n * dns + ((int64_t)c * H + y) * W + x;
      *o = accumulate ? *o + s : s;
    }
  }
}

// ------------------------------------------------------------------ generator output post-processing
__global__ __launch_bounds__(256) void g_post_kernel(const float* __restrict__ g, const float* __restrict__ M, int64_t HW,
                                                     float scale_nz, const float* __restrict__ rb, const float* __restrict__ rs,
                                                     float* __restrict__ fI, float* __restrict__ fT, int64_t fTns,
                                                     float* __restrict__ fN, float* __restrict__ aI, int64_t aIns,
                                                     const float* __restrict__ S, float* __restrict__ stS, float* __restrict__ stM, int64_t stns) {
  const int n = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float m = M[n * HW + i];
  if (stS) stS[n * stns + i] = S[n * HW + i];     // sketch / mask channels of the 7-channel full-resolution D2 stack
  if (stM) stM[n * stns + i] = m;
  const float* gp = g + n * 5 * HW + i;
  const float r = gp[0] * m, gg = gp[HW] * m, b = gp[2 * HW] * m, tx = gp[3 * HW] * m, ty = gp[4 * HW] * m;
  if (fI) {
    float* o = fI + n * 3 * HW + i;
    o[0] = r; o[HW] = gg; o[2 * HW] = b;
  }
  if (fT) {
    float* o = fT + n * fTns + i;
    o[0] = tx; o[HW] = ty;
  }
  if (fN) {
    const float nz = scale_nz;
    const float inv = 1.f / fmaxf(sqrtf(tx * tx + ty * ty + nz * nz), 1e-12f);  // F.normalize eps
    float* o = fN + n * 3 * HW + i;
    o[0] = tx * inv; o[HW] = ty * inv; o[2 * HW] = nz * inv;
  }
  if (aI) {
    const float db = rb[n] - 0.5f, k = rs[n] * 2.f;
    const float r1 = r + db, g1 = gg + db, b1 = b + db;
    const float mean = (r1 + g1 + b1) / 3.f;
    float* o = aI + n * aIns + i;
    o[0] = ((r1 - mean) * k + mean) * m;
    o[HW] = ((g1 - mean) * k + mean) * m;
    o[2 * HW] = ((b1 - mean) * k + mean) * m;
  }
}

__global__ __launch_bounds__(256) void diffaug_kernel(const float* __restrict__ x, const float* __restrict__ M, int64_t HW,
                                                      const float* __restrict__ rb, const float* __restrict__ rs, float* __restrict__ a) {
  const int n = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float m = M ? M[n * HW + i] : 1.f;
  const float* p = x + n * 3 * HW + i;
  const float db = rb[n] - 0.5f, k = rs[n] * 2.f;
  const float r1 = p[0] + db, g1 = p[HW] + db, b1 = p[2 * HW] + db;
  const float mean = (r1 + g1 + b1) / 3.f;
  float* o = a + n * 3 * HW + i;
  o[0] = ((r1 - mean) * k + mean) * m;
  o[HW] = ((g1 - mean) * k + mean) * m;
  o[2 * HW] = ((b1 - mean) * k + mean) * m;
}

// One DiffAugment operation beyond the fused 'bs' pair (thirdparty/DiffAugment.py:36-80): contrast, translation, cutout, noise -- and b / s
// on their own, so that any policy string runs as a chain of these.  One thread per pixel, all channels; grid (HW / 256, N).
//   'c' reads the sample's mean from `part` (DIFFAUG_PARTS partial sums per sample, summed here in a fixed order)
constexpr int DIFFAUG_PARTS = 256;

__global__ __launch_bounds__(256) void diffaug_mean_part_kernel(const float* __restrict__ x, int64_t xns, int64_t CHW, float* __restrict__ part) {
  __shared__ float red[16];
  const int n = blockIdx.y;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < CHW; i += (int64_t)DIFFAUG_PARTS * 256) acc += x[n * xns + i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[n * DIFFAUG_PARTS + blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void diffaug_op_kernel(const float* __restrict__ x, int64_t xns, float* __restrict__ out, int64_t ons, int C, int H,
                                                         int W, int op, const float* __restrict__ pf, const int* __restrict__ pi0,
                                                         const int* __restrict__ pi1, const float* __restrict__ noise,
                                                         const float* __restrict__ part, const float* __restrict__ M) {
  __shared__ float red[16];
  const int n = blockIdx.y;
  const int64_t HW = (int64_t)H * W;
  float mean_all = 0.f;
  if (op == 'c') {      // every block re-derives the sample mean from the partials (block-uniform, so the barrier inside is safe)
    mean_all = block_sum(part[n * DIFFAUG_PARTS + threadIdx.x], red) / (float)(C * HW);
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const int y = (int)(i / W), xx = (int)(i % W);
  const float m = M ? M[n * HW + i] : 1.f;
  const float* p = x + n * xns + i;
  float* o = out + n * ons + i;
  switch (op) {
    case 'b': {
      const float d = pf[n] - 0.5f;
      for (int c = 0; c < C; ++c) o[c * HW] = (p[c * HW] + d) * m;
      break;
    }
    case 's': {
      float mean = 0.f;
      for (int c = 0; c < C; ++c) mean += p[c * HW];
      mean /= (float)C;
      const float k = pf[n] * 2.f;
      for (int c = 0; c < C; ++c) o[c * HW] = ((p[c * HW] - mean) * k + mean) * m;
      break;
    }
    case 'c': {
      const float k = pf[n] + 0.5f;
      for (int c = 0; c < C; ++c) o[c * HW] = ((p[c * HW] - mean_all) * k + mean_all) * m;
      break;
    }
    case 't': {     // out[y, x] = in[y + tx, x + ty], zero outside (the reference pads by one zero row / column and clamps into it)
      const int sy = y + pi0[n], sx = xx + pi1[n];
      const bool ok = sy >= 0 && sy < H && sx >= 0 && sx < W;
      const float* q = x + n * xns + (int64_t)(ok ? sy : 0) * W + (ok ? sx : 0);
      for (int c = 0; c < C; ++c) o[c * HW] = ok ? q[c * HW] * m : 0.f;
      break;
    }
    case 'o': {     // rows / columns clamp(offset - size / 2 + [0, size)) are zeroed, size = int(0.5 * extent + 0.5)
      const int ch = (int)(H * 0.5 + 0.5), cw = (int)(W * 0.5 + 0.5);
      const int r0 = min(max(pi0[n] - ch / 2, 0), H - 1), r1 = min(max(pi0[n] - ch / 2 + ch - 1, 0), H - 1);
      const int c0 = min(max(pi1[n] - cw / 2, 0), W - 1), c1 = min(max(pi1[n] - cw / 2 + cw - 1, 0), W - 1);
      const bool cut = y >= r0 && y <= r1 && xx >= c0 && xx <= c1;
      for (int c = 0; c < C; ++c) o[c * HW] = cut ? 0.f : p[c * HW] * m;
      break;
    }
    default: {      // 'n': x + sigma * noise
      const float sg = pf[n];
      const float* z = noise + (int64_t)n * C * HW + i;
      for (int c = 0; c < C; ++c) o[c * HW] = (p[c * HW] + sg * z[c * HW]) * m;
    }
  }
}

// dIc (optional): the image gradient's NEXT pyramid level [N, 3, OH, OW] (the D1 pass of the generator step leaves one input gradient per
// scale): its average-pool adjoint is added here, i.e. the last avgpool_bwd launch of the merge (50 MB read-modify-write of dI on the
// serial stretch between the discriminator chain and the generator's backward) is this kernel's one more read of a 4x smaller tensor.
__global__ __launch_bounds__(256) void g_out_grad_kernel(const float* __restrict__ dI, const float* __restrict__ dT,
                                                         const float* __restrict__ M, const float* __restrict__ g, int64_t HW,
                                                         float* __restrict__ d, const float* __restrict__ dIc, int H, int W, int OH, int OW) {
  const int n = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float m = M[n * HW + i];
  const int y = dIc ? (int)(i / W) : 0, x = dIc ? (int)(i - (int64_t)y * W) : 0;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const float o = g[(n * 5 + c) * HW + i];
    float up = 0.f;
    if (c < 3) {
      up = dI ? dI[(n * 3 + c) * HW + i] : 0.f;
      if (dIc) up = up + avgpool_adjoint_at(dIc + ((int64_t)n * 3 + c) * OH * OW, y, x, H, W, OH, OW);
    } else {
      up = dT ? dT[(n * 2 + c - 3) * HW + i] : 0.f;
    }
    d[(n * 5 + c) * HW + i] = up * m * (1.f - o * o);
  }
}

__global__ __launch_bounds__(256) void mask_mul_kernel(const float* __restrict__ x, const float* __restrict__ M, int C, int64_t HW,
                                                       float* __restrict__ y) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  y[((int64_t)n * C + c) * HW + i] = x[((int64_t)n * C + c) * HW + i] * M[n * HW + i];
}

// ImagePool.query (util/image_pool.py:29-61): one thread per element walks the batch in order -- "return the slot's current content, then
// overwrite it" is sequential per slot, and a later image may draw the slot an earlier one has just filled
__global__ __launch_bounds__(256) void pool_query_kernel(const float* __restrict__ images, float* __restrict__ store, const int* __restrict__ ret_slot,
                                                         const int* __restrict__ put_slot, int N, int64_t elems, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= elems) return;
  for (int n = 0; n < N; ++n) {
    const float cur = images[(int64_t)n * elems + e];
    const int r = ret_slot[n], w = put_slot[n];
    const float ret = r >= 0 ? store[(int64_t)r * elems + e] : cur;
    if (w >= 0) store[(int64_t)w * elems + e] = cur;
    out[(int64_t)n * elems + e] = ret;
  }
}

// 8-bit image data -> the float tensor the dataset's transform makes of it (torchvision ToTensor: v / 255 in fp32; Normalize(0.5, 0.5):
// (t - 0.5) / 0.5), same operations in the same order and precision: bit-identical to the host tensor, a quarter of the PCIe bytes
__global__ __launch_bounds__(256) void u8_expand_kernel(const uint8_t* __restrict__ src, int64_t n, int normalize, float* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (i + j < n) {
      float t = __fdiv_rn((float)src[i + j], 255.f);
      if (normalize) t = __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
      out[i + j] = t;
    }
  }
}

// set_input's image part in one pass over 8-bit sources: M = bytes / 255, S = Normalize(ToTensor(bytes)) * M (written twice: the fake and
// the real rows of the discriminator's pair buffer), I likewise (real rows).  The same operations in the same order as u8_expand_kernel +
// mask_mul_kernel (bit-identical), 5 bytes read and 24 written per pixel instead of seven launches.
template <bool VEC>
__global__ __launch_bounds__(256) void input_images_u8_kernel(const uint8_t* __restrict__ S, const uint8_t* __restrict__ I, const uint8_t* __restrict__ Mb,
                                                              int64_t HW, float* __restrict__ Mo, float* __restrict__ So, float* __restrict__ So2,
                                                              float* __restrict__ Io) {
  const int n = blockIdx.y;
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= HW) return;
  const int cnt = VEC ? 4 : (int)min((int64_t)4, HW - i);
  auto bytes = [&](const uint8_t* p, uint8_t (&b)[4]) {      // VEC: HW % 4 == 0 and 4-byte aligned planes -> one dword load
    if (VEC) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
      b[0] = w & 255u; b[1] = (w >> 8) & 255u; b[2] = (w >> 16) & 255u; b[3] = w >> 24;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = j < cnt ? p[j] : 0;
    }
  };
  auto store = [&](float* p, const float (&v)[4]) {
    if (VEC) {
      *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < cnt) p[j] = v[j];
    }
  };
  uint8_t b[4];
  float m[4], v[4];
  if (Mb) {
    bytes(Mb + n * HW + i, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = __fdiv_rn((float)b[j], 255.f);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = 1.f;
  }
  if (Mo) store(Mo + n * HW + i, m);
  bytes(S + n * HW + i, b);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)b[j], 255.f), 0.5f), 0.5f) * m[j];
  store(So + n * HW + i, v);
  if (So2) store(So2 + n * HW + i, v);
  if (I) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bytes(I + (n * 3 + c) * HW + i, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)b[j], 255.f), 0.5f), 0.5f) * m[j];
      store(Io + (n * 3 + c) * HW + i, v);
    }
  }
}

__global__ __launch_bounds__(256) void spe_kernel(float* __restrict__ out, int64_t ons, int H, int W, int dim) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
  if (x >= W) return;
  const int half = dim / 2;
  const float step = -(logf(10000.f) / (float)(half - 1));
  for (int d = 0; d < 2 * dim; ++d) {
    const int e = d % dim;
    const float pos = (float)((d < dim ? x : y) + 1);
    const float f = expf((float)(e % half) * step);
    const float a = pos * f;
    out[n * ons + ((int64_t)d * H + y) * W + x] = e < half ? sinf(a) : cosf(a);
  }
}

// ------------------------------------------------------------------ "more fake T" candidate map
__global__ __launch_bounds__(256) void mask_cand_kernel(const float* __restrict__ M, int H, int W, int Hc, int Wc,
                                                        uint8_t* __restrict__ cand) {
  __shared__ uint8_t in[48][49];
  __shared__ uint8_t hz[48][33];
  const int n = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  const float* m = M + (int64_t)n * H * W;
  for (int i = threadIdx.x; i < 48 * 48; i += 256) {
    const int r = i / 48, c = i - r * 48;
    const int y = y0 - 1 + r, x = x0 - 1 + c;
    in[r][c] = (y >= 0 && y < H && x >= 0 && x < W && m[(int64_t)y * W + x] > 0.f) ? 1 : 0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 48 * 32; i += 256) {
    const int r = i >> 5, c = i & 31;
    uint8_t a = 0;
    for (int d = 0; d < 17; ++d) a |= in[r][c + d];
    hz[r][c] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {
    const int r = i >> 5, c = i & 31;
    const int y = y0 + r, x = x0 + c;
    if (y >= Hc || x >= Wc) continue;
    uint8_t a = 0;
    for (int d = 0; d < 17; ++d) a |= hz[r + d][c];
    cand[((int64_t)n * Hc + y) * Wc + x] = a;
  }
}

__global__ __launch_bounds__(64) void mask_rowcount_kernel(const uint8_t* __restrict__ cand, int Hc, int Wc, int* __restrict__ rc) {
  const int y = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  int s = 0;
  for (int x = lane; x < Wc; x += 64) s += cand[((int64_t)n * Hc + y) * Wc + x];
  s = (int)wave_sum((float)s);
  if (lane == 0) rc[n * (Hc + 1) + y + 1] = s;
}

// in-place inclusive scan -> rc[n][0..Hc] exclusive prefix.  One wavefront per image: every lane scans a contiguous run of rows, the run
// totals are scanned across the wave with shuffles (integers: exact and order independent).  (The single-thread loop this replaces
// took 186 us for 1010 rows: a dependent global read-modify-write per row.)
__global__ __launch_bounds__(64) void mask_prefix_kernel(int* __restrict__ rc, int Hc) {
  const int n = blockIdx.x, lane = threadIdx.x;
  int* p = rc + n * (Hc + 1);
  const int per = (Hc + 63) / 64;
  const int y0 = 1 + lane * per, y1 = min(y0 + per, Hc + 1);
  int run = 0;
  for (int y = y0; y < y1; ++y) run += p[y];
  int incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  int acc = incl - run;      // total of the lanes before this one
  if (lane == 0) p[0] = 0;
  for (int y = y0; y < y1; ++y) {
    acc += p[y];
    p[y] = acc;
  }
}

__global__ __launch_bounds__(64) void mask_select_kernel(const uint8_t* __restrict__ cand, const int* __restrict__ prefix, int Hc,
                                                         int Wc, const int64_t* __restrict__ ranks, int K, int* __restrict__ offx,
                                                         int* __restrict__ offy) {
  const int k = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const int* p = prefix + n * (Hc + 1);
  const int rank = (int)ranks[n * K + k];
  int lo = 0, hi = Hc - 1;  // find row with p[row] <= rank < p[row+1]
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (p[mid + 1] <= rank) lo = mid + 1; else hi = mid;
  }
  const int row = lo;
  int need = rank - p[row];
  int found = -1;
  for (int x0 = 0; x0 < Wc && found < 0; x0 += 64) {
    const int x = x0 + lane;
    const bool bit = x < Wc && cand[((int64_t)n * Hc + row) * Wc + x];
    const unsigned long long m = __ballot(bit);
    const int c = __popcll(m);
    if (need < c) {
      // the (need)-th set bit of m
      const int before = __popcll(m & ((1ull << lane) - 1ull));
      const unsigned long long sel = __ballot(bit && before == need);
      found = x0 + __ffsll((long long)sel) - 1;
    } else {
      need -= c;
    }
  }
  if (lane == 0) {
    offx[n * K + k] = found;
    offy[n * K + k] = row;
  }
}

// ------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float step_size, float b1, float b2, float eps,
                                                   float inv_bc2_sqrt, float gs) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gr = g[i] * gs;
    const float mi = m[i] * b1 + (1.f - b1) * gr;
    const float vi = v[i] * b2 + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
  }
}

// same update with the step counter and learning rate read from device memory, so that a captured
// HIP graph of the training step stays valid across steps and LR-schedule changes
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, const float* __restrict__ lr_dev, float b1,
                                                       float b2, float eps, const int* __restrict__ step_dev, float gs) {
  const float step = (float)step_dev[0];
  const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
  const float step_size = lr_dev[0] / bc1, inv_bc2_sqrt = 1.f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gr = g[i] * gs;
    const float mi = m[i] * b1 + (1.f - b1) * gr;
    const float vi = v[i] * b2 + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
  }
}

// ------------------------------------------------------------------ PatchNCE
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int D, float* __restrict__ y) {
  __shared__ float red[16];
  const float* p = x + (int64_t)blockIdx.x * D;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) s += p[i] * p[i];
  s = block_sum(s, red);
  const float inv = 1.f / (sqrtf(s) + 1e-7f);
  for (int i = threadIdx.x; i < D; i += 256) y[(int64_t)blockIdx.x * D + i] = p[i] * inv;
}

// one workgroup per (group b, query row i): logits against the P keys of the group, CE with the
// positive at index 0, gradient wrt q_i.  Logits live in LDS only.
__global__ __launch_bounds__(256) void patchnce_kernel(const float* __restrict__ q, const float* __restrict__ k, int P, int D, float invT,
                                                       float gscale, float* __restrict__ loss, float* __restrict__ dq) {
  extern __shared__ float sm[];  // q row [D] | weights [P+1]
  __shared__ float red[16];
  float* qrow = sm;
  float* wgt = sm + D;
  const int i = blockIdx.x, b = blockIdx.y;
  const float* qb = q + ((int64_t)b * P + i) * D;
  const float* kb = k + (int64_t)b * P * D;
  for (int d = threadIdx.x; d < D; d += 256) qrow[d] = qb[d];
  __syncthreads();
  // logits: entry 0 = positive, entry 1+j = negative j (diagonal replaced by -10)
  for (int j = threadIdx.x; j < P; j += 256) {
    const float* kr = kb + (int64_t)j * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += qrow[d] * kr[d];
    if (j == i) {
      wgt[0] = s * invT;
      wgt[1 + j] = -10.f * invT;
    } else {
      wgt[1 + j] = s * invT;
    }
  }
  __syncthreads();
  float mx = -3.4e38f;
  for (int j = threadIdx.x; j <= P; j += 256) mx = fmaxf(mx, wgt[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float pos = wgt[0];
  __syncthreads();
  float se = 0.f;
  for (int j = threadIdx.x; j <= P; j += 256) {
    const float e = expf(wgt[j] - mx);
    wgt[j] = e;
    se += e;
  }
  se = block_sum(se, red);
  if (threadIdx.x == 0 && loss) loss[(int64_t)b * P + i] = logf(se) + mx - pos;
  if (!dq) return;
  __syncthreads();
  const float inv = 1.f / se;
  // d loss / d q_i = invT * [ (p0 - 1) k_i + sum_{j != i} p_{1+j} k_j ]
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = (wgt[0] * inv - 1.f) * kb[(int64_t)i * D + d];
    for (int j = 0; j < P; ++j)
      if (j != i) s += wgt[1 + j] * inv * kb[(int64_t)j * D + d];
    dq[((int64_t)b * P + i) * D + d] = s * invT * gscale;
  }
}

inline unsigned blocks_for(int64_t n, int cap = 2048) {
  int64_t b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int vts_avgpool3s2(const float* x, int64_t xns, int N, int C, int H, int W, float* y, void* stream) {
  VTS_CHECK_ARG(x && y && N * C <= 65535, "vts_avgpool3s2: bad args");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  static const bool rows4 = !(vts_tune("VTS_AVGPOOL_ROWS4", 1) == 0);
  if (rows4 && W % 2 == 0 && W >= 4 && (reinterpret_cast<uintptr_t>(x) & 7) == 0 && xns % 2 == 0 && ((int64_t)H * W) % 2 == 0)
    hipLaunchKernelGGL(avgpool_rows4_kernel, dim3(cdiv(OW, 64), cdiv(OH, 16), N * C), dim3(256), 0, (hipStream_t)stream, x, xns, C, H, W,
                       OH, OW, y);
  else
    hipLaunchKernelGGL(avgpool_kernel, dim3(cdiv(OW, 64), cdiv(OH, 4), N * C), dim3(256), 0, (hipStream_t)stream, x, xns, C, H, W, OH,
                       OW, y);
  VTS_CHECK_LAUNCH("vts_avgpool3s2");
  return VTS_OK;
}

extern "C" int vts_avgpool3s2_bwd(const float* dy, int N, int C, int H, int W, float* dx, int64_t dxns, int accumulate, void* stream) {
  VTS_CHECK_ARG(dy && dx && N * C <= 65535, "vts_avgpool3s2_bwd: bad args");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(cdiv(W, 64), cdiv(H, 4), N * C), dim3(256), 0, (hipStream_t)stream, dy, C, H, W, OH, OW,
                     dx, dxns, accumulate);
  VTS_CHECK_LAUNCH("vts_avgpool3s2_bwd");
  return VTS_OK;
}

extern "C" int vts_ganloss(const float* pred, int N, int M, int mode, int target_is_real, float target_label, float coeff,
                           float grad_coeff, int64_t* loss_out, float* dpred, void* stream) {
  VTS_CHECK_ARG(pred && N >= 1 && M >= 1 && mode >= 0 && mode <= 5, "vts_ganloss: bad args");
  const int64_t total = (int64_t)N * M;
  hipLaunchKernelGGL(ganloss_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, pred, total, mode, target_is_real,
                     target_label, 1.f / (float)total, coeff, grad_coeff, reinterpret_cast<long long*>(loss_out), dpred);
  VTS_CHECK_LAUNCH("vts_ganloss");
  return VTS_OK;
}

extern "C" int vts_l1(const float* a, const float* b, int64_t n, float coeff, int64_t* loss_out, float* grad, int accumulate,
                      void* stream) {
  VTS_CHECK_ARG(a && b && n >= 1, "vts_l1: bad args");
  const int vec = (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0) ? 1 : 0;
  hipLaunchKernelGGL(l1_kernel, dim3(blocks_for(vec ? n / 4 : n, 1024)), dim3(256), 0, (hipStream_t)stream, a, b, n, coeff,
                     reinterpret_cast<long long*>(loss_out), grad, accumulate, vec);
  VTS_CHECK_LAUNCH("vts_l1");
  return VTS_OK;
}

extern "C" int vts_patch_gather(const float* src, int64_t sns, int C, int H, int W, const int* img, const int* offx, const int* offy,
                                int P, int size, float* out, int out_C, int out_c0, void* stream) {
  VTS_CHECK_ARG(src && img && offx && offy && out && P >= 1 && C >= 1, "vts_patch_gather: bad args");
  hipLaunchKernelGGL(patch_gather_kernel, dim3(P, C), dim3(256), 0, (hipStream_t)stream, src, sns, C, H, W, img, offx, offy, size, out,
                     out_C, out_c0);
  VTS_CHECK_LAUNCH("vts_patch_gather");
  return VTS_OK;
}

// start of a training step: zero the loss slots and advance the optimisers' device step counters -- one launch instead of a memset
// and one increment per optimiser
__global__ void step_begin_kernel(long long* slots, int nslots, int* counters, int ncounters) {
  const int i = threadIdx.x;
  if (i < nslots) slots[i] = 0;
  if (i < ncounters) counters[i] += 1;
}

extern "C" int vts_step_begin(int64_t* slots, int nslots, int* counters, int ncounters, void* stream) {
  VTS_CHECK_ARG(nslots >= 0 && ncounters >= 0 && nslots <= 256 && ncounters <= 256 && (slots || !nslots) && (counters || !ncounters), "vts_step_begin: bad args");
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<long long*>(slots), nslots, counters, ncounters);
  VTS_CHECK_LAUNCH("vts_step_begin");
  return VTS_OK;
}

extern "C" int vts_patch_jobs(const vts_patch_job* jobs, int njobs, int size, void* stream) {
  VTS_CHECK_ARG(jobs && njobs >= 1 && njobs <= PJ_MAX && size >= 1, "vts_patch_jobs: 1 .. %d jobs", PJ_MAX);
  PatchJobs t;
  t.njobs = njobs;
  int ch = 0, pmax = 0;
  for (int j = 0; j < njobs; ++j) {
    const vts_patch_job& q = jobs[j];
    VTS_CHECK_ARG(q.dst && q.C >= 1 && q.P >= 1 && q.dst_c0 >= 0 && q.dst_c0 + q.C <= q.dst_C && (!q.img || (q.src && q.offx && q.offy)), "vts_patch_jobs: job %d malformed", j);
    t.ch_start[j] = ch;
    t.job[j] = q;
    ch += q.C;
    pmax = q.P > pmax ? q.P : pmax;
  }
  t.ch_start[njobs] = ch;
  hipLaunchKernelGGL(patch_jobs_kernel, dim3(pmax, ch), dim3(256), 0, (hipStream_t)stream, t, size);
  VTS_CHECK_LAUNCH("vts_patch_jobs");
  return VTS_OK;
}

extern "C" int vts_patch_scatter_bwd(const float* dpatch, int dp_C, int dp_c0, int C, const int* img, const int* offx, const int* offy,
                                     int P, int P_per_img, int size, float* dsrc, int64_t dns, int N, int H, int W, int accumulate,
                                     void* stream) {
  (void)img;  // patches of image n are [n*P_per_img, (n+1)*P_per_img): the layout the gather produces
  VTS_CHECK_ARG(dpatch && offx && offy && dsrc && P == N * P_per_img, "vts_patch_scatter_bwd: P must equal N*P_per_img");
  hipLaunchKernelGGL(patch_scatter_kernel, dim3(cdiv(W, 32), cdiv(H, 32), N), dim3(256), P_per_img * sizeof(int), (hipStream_t)stream,
                     dpatch, dp_C, dp_c0, C, offx, offy, P_per_img, size, dsrc, dns, H, W, accumulate);
  VTS_CHECK_LAUNCH("vts_patch_scatter_bwd");
  return VTS_OK;
}

static int g_post_impl(const float* g_out, const float* M, int N, int H, int W, float scale_nz, const float* rb, const float* rs,
                       float* fake_I, float* fake_T, int64_t fake_T_nstride, float* fake_N, float* aug_fake_I, int64_t aug_nstride,
                       const float* S, float* stack_S, float* stack_M, int64_t stack_nstride, void* stream) {
  VTS_CHECK_ARG(g_out && M && (!aug_fake_I || (rb && rs)) && (!stack_S || S), "vts_g_post: bad args");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(g_post_kernel, dim3((unsigned)cdiv64(HW, 256), N), dim3(256), 0, (hipStream_t)stream, g_out, M, HW, scale_nz, rb, rs,
                     fake_I, fake_T, fake_T_nstride ? fake_T_nstride : 2 * HW, fake_N, aug_fake_I, aug_nstride ? aug_nstride : 3 * HW,
                     S, stack_S, stack_M, stack_nstride);
  VTS_CHECK_LAUNCH("vts_g_post");
  return VTS_OK;
}

extern "C" int vts_g_post(const float* g_out, const float* M, int N, int H, int W, float scale_nz, const float* rb, const float* rs,
                          float* fake_I, float* fake_T, int64_t fake_T_nstride, float* fake_N, float* aug_fake_I,
                          int64_t aug_nstride, void* stream) {
  return g_post_impl(g_out, M, N, H, W, scale_nz, rb, rs, fake_I, fake_T, fake_T_nstride, fake_N, aug_fake_I, aug_nstride, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int vts_g_post_stack(const float* g_out, const float* M, int N, int H, int W, float scale_nz, const float* rb, const float* rs,
                                float* fake_I, float* fake_T, int64_t fake_T_nstride, float* fake_N, float* aug_fake_I, int64_t aug_nstride,
                                const float* S, float* stack_S, float* stack_M, int64_t stack_nstride, void* stream) {
  return g_post_impl(g_out, M, N, H, W, scale_nz, rb, rs, fake_I, fake_T, fake_T_nstride, fake_N, aug_fake_I, aug_nstride, S, stack_S, stack_M, stack_nstride, stream);
}

extern "C" int vts_diffaug_bs_mask(const float* x, const float* M, int N, int H, int W, const float* rb, const float* rs, float* aug,
                                   void* stream) {
  VTS_CHECK_ARG(x && rb && rs && aug, "vts_diffaug_bs_mask: bad args");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(diffaug_kernel, dim3((unsigned)cdiv64(HW, 256), N), dim3(256), 0, (hipStream_t)stream, x, M, HW, rb, rs, aug);
  VTS_CHECK_LAUNCH("vts_diffaug_bs_mask");
  return VTS_OK;
}

extern "C" int vts_g_out_grad_pool(const float* d_fake_I, const float* d_fake_I_coarse, const float* d_fake_T, const float* M, const float* g_out,
                                   int N, int H, int W, float* d_raw, void* stream) {
  VTS_CHECK_ARG(M && g_out && d_raw, "vts_g_out_grad: bad args");
  VTS_CHECK_ARG(!d_fake_I_coarse || (d_fake_I && H >= 2 && W >= 2), "vts_g_out_grad_pool: a coarse gradient needs the fine one and a map of at least 2 x 2");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(g_out_grad_kernel, dim3((unsigned)cdiv64(HW, 256), N), dim3(256), 0, (hipStream_t)stream, d_fake_I, d_fake_T, M,
                     g_out, HW, d_raw, d_fake_I_coarse, H, W, (H + 1) / 2, (W + 1) / 2);
  VTS_CHECK_LAUNCH("vts_g_out_grad");
  return VTS_OK;
}

extern "C" int vts_g_out_grad(const float* d_fake_I, const float* d_fake_T, const float* M, const float* g_out, int N, int H, int W,
                              float* d_raw, void* stream) {
  return vts_g_out_grad_pool(d_fake_I, nullptr, d_fake_T, M, g_out, N, H, W, d_raw, stream);
}

extern "C" int vts_diffaug_op_ws_floats(int N) { return N * DIFFAUG_PARTS; }

extern "C" int vts_diffaug_op(const float* x, int64_t x_ns, float* out, int64_t out_ns, int N, int C, int H, int W, int op, const float* pf,
                              const int* pi0, const int* pi1, const float* noise, const float* M, float* ws, void* stream) {
  VTS_CHECK_ARG(x && out && x != out && N >= 1 && C >= 1 && H >= 1 && W >= 1, "vts_diffaug_op: bad args");
  const bool f = op == 'b' || op == 's' || op == 'c' || op == 'n', g = op == 't' || op == 'o';
  VTS_CHECK_ARG(f || g, "vts_diffaug_op: op must be one of b s c t o n");
  VTS_CHECK_ARG(!f || pf, "vts_diffaug_op: b / s / c / n need the per-sample float draw");
  VTS_CHECK_ARG(!g || (pi0 && pi1), "vts_diffaug_op: t / o need the two per-sample integer draws");
  VTS_CHECK_ARG(op != 'n' || noise, "vts_diffaug_op: n needs the noise tensor");
  VTS_CHECK_ARG(op != 'c' || ws, "vts_diffaug_op: c needs vts_diffaug_op_ws_floats(N) floats of workspace");
  const int64_t HW = (int64_t)H * W;
  if (op == 'c') {
    VTS_CHECK_ARG(x_ns == (int64_t)C * HW, "vts_diffaug_op: c needs a contiguous sample");
    hipLaunchKernelGGL(diffaug_mean_part_kernel, dim3(DIFFAUG_PARTS, N), dim3(256), 0, (hipStream_t)stream, x, x_ns, (int64_t)C * HW, ws);
  }
  hipLaunchKernelGGL(diffaug_op_kernel, dim3((unsigned)cdiv64(HW, 256), N), dim3(256), 0, (hipStream_t)stream, x, x_ns, out, out_ns, C, H, W, op,
                     pf, pi0, pi1, noise, ws, M);
  VTS_CHECK_LAUNCH("vts_diffaug_op");
  return VTS_OK;
}

extern "C" int vts_pool_query(const float* images, float* store, const int* ret_slot, const int* put_slot, int N, int64_t elems, float* out,
                              void* stream) {
  VTS_CHECK_ARG(images && store && ret_slot && put_slot && out && N >= 0 && elems >= 0, "vts_pool_query: bad args");
  if (N == 0 || elems == 0) return VTS_OK;
  hipLaunchKernelGGL(pool_query_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images, store, ret_slot, put_slot,
                     N, elems, out);
  VTS_CHECK_LAUNCH("vts_pool_query");
  return VTS_OK;
}

extern "C" int vts_mask_mul(const float* x, const float* M, int N, int C, int HW, float* y, void* stream) {
  VTS_CHECK_ARG(x && M && y, "vts_mask_mul: bad args");
  hipLaunchKernelGGL(mask_mul_kernel, dim3(cdiv(HW, 256), C, N), dim3(256), 0, (hipStream_t)stream, x, M, C, (int64_t)HW, y);
  VTS_CHECK_LAUNCH("vts_mask_mul");
  return VTS_OK;
}

extern "C" int vts_u8_expand(const uint8_t* src, int64_t n, int normalize, float* out, void* stream) {
  VTS_CHECK_ARG(src && out && n >= 1, "vts_u8_expand: bad args");
  hipLaunchKernelGGL(u8_expand_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, n, normalize, out);
  VTS_CHECK_LAUNCH("vts_u8_expand");
  return VTS_OK;
}

extern "C" int vts_input_images_u8(const uint8_t* S, const uint8_t* I, const uint8_t* M, int N, int64_t HW, float* M_out, float* S_out,
                                   float* S_out2, float* I_out, void* stream) {
  VTS_CHECK_ARG(S && S_out && N >= 1 && HW >= 1 && (!I || I_out) && (!M_out || M), "vts_input_images_u8: bad args");
  const uintptr_t al = reinterpret_cast<uintptr_t>(S) | reinterpret_cast<uintptr_t>(I) | reinterpret_cast<uintptr_t>(M);
  const uintptr_t al16 = reinterpret_cast<uintptr_t>(M_out) | reinterpret_cast<uintptr_t>(S_out) | reinterpret_cast<uintptr_t>(S_out2) |
                         reinterpret_cast<uintptr_t>(I_out);
  if (HW % 4 == 0 && (al & 3) == 0 && (al16 & 15) == 0)
    hipLaunchKernelGGL(input_images_u8_kernel<true>, dim3((unsigned)cdiv64(HW, 1024), N), dim3(256), 0, (hipStream_t)stream, S, I, M, HW, M_out,
                       S_out, S_out2, I_out);
  else
    hipLaunchKernelGGL(input_images_u8_kernel<false>, dim3((unsigned)cdiv64(HW, 1024), N), dim3(256), 0, (hipStream_t)stream, S, I, M, HW, M_out,
                       S_out, S_out2, I_out);
  VTS_CHECK_LAUNCH("vts_input_images_u8");
  return VTS_OK;
}

extern "C" int vts_spe_grid(float* out, int64_t ons, int N, int H, int W, int dim, void* stream) {
  VTS_CHECK_ARG(out && dim >= 4 && dim % 2 == 0, "vts_spe_grid: dim must be even and >= 4");
  hipLaunchKernelGGL(spe_kernel, dim3(cdiv(W, 256), H, N), dim3(256), 0, (hipStream_t)stream, out, ons, H, W, dim);
  VTS_CHECK_LAUNCH("vts_spe_grid");
  return VTS_OK;
}

extern "C" int vts_mask_candidates(const float* M, int N, int H, int W, uint8_t* cand, int* row_count, void* stream) {
  VTS_CHECK_ARG(M && cand && row_count && H > 14 && W > 14, "vts_mask_candidates: image must exceed 14 px");
  const int Hc = H - 14, Wc = W - 14;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mask_cand_kernel, dim3(cdiv(Wc, 32), cdiv(Hc, 32), N), dim3(256), 0, st, M, H, W, Hc, Wc, cand);
  VTS_CHECK_LAUNCH("vts_mask_candidates");
  hipLaunchKernelGGL(mask_rowcount_kernel, dim3(Hc, N), dim3(64), 0, st, cand, Hc, Wc, row_count);
  hipLaunchKernelGGL(mask_prefix_kernel, dim3(N), dim3(64), 0, st, row_count, Hc);
  VTS_CHECK_LAUNCH("vts_mask_candidates prefix");
  return VTS_OK;
}

// K DISTINCT uniform ranks in [0, c) per image, c = the image's candidate count (row_prefix[n][Hc]): random.sample(range(c), K) of the
// reference (models/model_utils.py:217) without the host in the loop -- the count lives on the device, and fetching it made the host wait
// for the whole upload queue before it could enqueue the step.  Floyd's algorithm (every K-subset equally likely), one wave per image:
// for j = c - K .. c - 1: t = uniform[0, j]; insert t, or j if t is already in the set (membership test across the lanes).
// Counter-based generator: splitmix64 of (seed, image, draw); t = mulhi64(hash, j + 1).  c < K (the reference raises): ranks wrap.
__device__ __forceinline__ unsigned long long vts_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(64) void mask_sample_ranks_kernel(const int* __restrict__ row_prefix, int Hc, int K, unsigned long long seed,
                                                               long long* __restrict__ ranks) {
  __shared__ long long S[1024];
  const int n = blockIdx.x, lane = threadIdx.x;
  const long long c = row_prefix[(long long)n * (Hc + 1) + Hc];
  if (c < K) {
    for (int q = lane; q < K; q += 64) ranks[(long long)n * K + q] = c > 0 ? q % c : 0;
    return;
  }
  for (int i = 0; i < K; ++i) {
    const long long j = c - K + i;
    const unsigned long long h = vts_splitmix64(vts_splitmix64(seed ^ ((unsigned long long)n * 0xD1B54A32D192ED03ull)) + (unsigned long long)i);
    const long long t = (long long)__umul64hi(h, (unsigned long long)(j + 1));
    bool hit = false;
    for (int q = lane; q < i; q += 64) hit |= S[q] == t;
    const bool any = __any(hit);
    __syncthreads();
    if (lane == 0) S[i] = any ? j : t;
    __syncthreads();
  }
  for (int q = lane; q < K; q += 64) ranks[(long long)n * K + q] = S[q];
}

extern "C" int vts_mask_sample_ranks(const int* row_prefix, int N, int H, int K, uint64_t seed, int64_t* ranks, void* stream) {
  VTS_CHECK_ARG(row_prefix && ranks && N >= 1 && H > 14 && K >= 1 && K <= 1024, "vts_mask_sample_ranks: bad args (K <= 1024)");
  hipLaunchKernelGGL(mask_sample_ranks_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, row_prefix, H - 14, K, (unsigned long long)seed,
                     (long long*)ranks);
  VTS_CHECK_LAUNCH("vts_mask_sample_ranks");
  return VTS_OK;
}

extern "C" int vts_mask_select(const uint8_t* cand, const int* row_prefix, int N, int H, int W, const int64_t* ranks, int K, int* offx,
                               int* offy, void* stream) {
  VTS_CHECK_ARG(cand && row_prefix && ranks && offx && offy && K >= 1, "vts_mask_select: bad args");
  hipLaunchKernelGGL(mask_select_kernel, dim3(K, N), dim3(64), 0, (hipStream_t)stream, cand, row_prefix, H - 14, W - 14, ranks, K, offx,
                     offy);
  VTS_CHECK_LAUNCH("vts_mask_select");
  return VTS_OK;
}

extern "C" int vts_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             int step_count, float grad_scale, void* stream) {
  VTS_CHECK_ARG(p && g && m && v && n >= 1 && step_count >= 1, "vts_adam_flat: bad args");
  const double bc1 = 1.0 - pow((double)beta1, step_count), bc2 = 1.0 - pow((double)beta2, step_count);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(lr / bc1), beta1,
                     beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  VTS_CHECK_LAUNCH("vts_adam_flat");
  return VTS_OK;
}

extern "C" int vts_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1,
                                 float beta2, float eps, const int* step_dev, float grad_scale, void* stream) {
  VTS_CHECK_ARG(p && g && m && v && lr_dev && step_dev && n >= 1, "vts_adam_flat_dev: bad args");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_dev, beta1, beta2,
                     eps, step_dev, grad_scale);
  VTS_CHECK_LAUNCH("vts_adam_flat_dev");
  return VTS_OK;
}

extern "C" int vts_l2norm_rows(const float* x, int rows, int D, float* y, void* stream) {
  VTS_CHECK_ARG(x && y && rows >= 1 && D >= 1, "vts_l2norm_rows: bad args");
  hipLaunchKernelGGL(l2norm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, D, y);
  VTS_CHECK_LAUNCH("vts_l2norm_rows");
  return VTS_OK;
}

extern "C" int vts_patchnce(const float* q, const float* k, int B, int P, int D, float T, float gscale, float* loss, float* dq,
                            void* stream) {
  VTS_CHECK_ARG(q && k && B >= 1 && P >= 1 && D >= 1 && T > 0.f, "vts_patchnce: bad args");
  static const int use_mfma = vts_tune("VTS_PATCHNCE_MFMA", 1);
  if (use_mfma && vts_patchnce_mfma_ok(P, D)) return vts_patchnce_mfma(q, k, B, P, D, T, gscale, loss, dq, (hipStream_t)stream);
  const size_t sm = (size_t)(D + P + 1) * sizeof(float);
  VTS_CHECK_ARG(sm <= 64 * 1024, "vts_patchnce: D + P too large for one LDS tile");
  hipLaunchKernelGGL(patchnce_kernel, dim3(P, B), dim3(256), sm, (hipStream_t)stream, q, k, P, D, 1.f / T, gscale, loss, dq);
  vts_set_kernel("patchnce_kernel");
  VTS_CHECK_LAUNCH("vts_patchnce");
  return VTS_OK;
}

// plain word copy on the compute queue.  The source may be PINNED HOST memory (device-accessible under unified addressing): uploads of
// the per-batch patch bookkeeping go through this kernel instead of an SDMA copy, which keeps them in stream order with the HIP-graph
// replays without a cross-engine dependency (models/sinskitG_model.py:_patch_set).
__global__ __launch_bounds__(256) void copy_words_kernel(const int* __restrict__ src, int* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int vts_copy_words(const void* src, void* dst, int64_t nwords, void* stream) {
  VTS_CHECK_ARG(src && dst && nwords >= 0, "vts_copy_words: bad args");
  if (nwords == 0) return VTS_OK;
  hipLaunchKernelGGL(copy_words_kernel, dim3(blocks_for(nwords, 1024)), dim3(256), 0, (hipStream_t)stream, (const int*)src, (int*)dst, nwords);
  VTS_CHECK_LAUNCH("vts_copy_words");
  return VTS_OK;
}
