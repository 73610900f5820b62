// InstanceNorm / BatchNorm statistics and backward, channel sums, activation backward.
// All of these are HBM-bound streaming reductions: one coalesced pass, wave64 shuffle
// reductions, fixed-order second stage (deterministic, no float atomics).
#include <stdlib.h>

#include "vts_internal.h"

namespace {

constexpr int CHUNK = 2048;  // elements per workgroup: 256 threads x 8 registers
constexpr int EPT = CHUNK / 256;

__host__ __device__ inline int splits_for(int HW) { return (HW + CHUNK - 1) / CHUNK; }

// ---- forward statistics: per (n, c, split) -> (mean, M2, count) via a register-resident two-pass ----
__device__ __forceinline__ void stats_partial_body(const float* __restrict__ x, int64_t nstride, int C, int HW, int spl,
                                                    float* __restrict__ part) {
  __shared__ float red[16];
  const int s = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const float* px = x + n * nstride + (int64_t)c * HW;
  const int base = s * CHUNK;
  const int cnt = min(CHUNK, HW - base);
  float v[EPT];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = e * 256 + threadIdx.x;
    v[e] = (i < cnt) ? px[base + i] : 0.f;
    sum += v[e];
  }
  const float mean = block_sum(sum, red) / (float)cnt;
  float m2 = 0.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = e * 256 + threadIdx.x;
    const float d = v[e] - mean;
    if (i < cnt) m2 += d * d;
  }
  m2 = block_sum(m2, red);
  if (threadIdx.x == 0) {
    float* o = part + (((int64_t)n * C + c) * spl + s) * 3;
    o[0] = mean;
    o[1] = m2;
    o[2] = (float)cnt;
  }
}

// "The last workgroup of a group finalises": every workgroup publishes its partial, takes a ticket from the group's counter and the
// one that draws the last ticket reduces all partials in the fixed order of the stand-alone finalize kernels (bit-identical results,
// deterministic) -- one launch and one dependency hop fewer per normalisation / bias gradient.  `counters` is caller-owned device
// memory, zero on entry; the last workgroup resets its counter, so it stays zero between launches (stream order).  The partials
// of the other workgroups are read through volatile pointers (L2, not this CU's vector cache).
__device__ __forceinline__ bool last_block_of(int* counters, int g, int nblk, int* sflag) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = atomicAdd(&counters[g], 1);
    const int last = t == nblk - 1;
    if (last) counters[g] = 0;
    *sflag = last;
  }
  __syncthreads();
  const bool last = *sflag != 0;
  if (last) __threadfence();
  return last;
}

// Chan merge of `np` partials strided by `stride` (one wave); returns (mean, M2, count) in every lane
__device__ __forceinline__ void chan_merge_wave(const volatile float* part, int np, int64_t stride, float& mean, float& m2, float& cnt) {
  const int lane = threadIdx.x & 63;
  float sn = 0.f, sm = 0.f;
  for (int i = lane; i < np; i += 64) {
    const volatile float* q = part + i * stride;
    sn += q[2];
    sm += q[2] * q[0];
  }
  sn = wave_sum(sn);
  sm = wave_sum(sm);
  mean = sm / sn;
  float acc = 0.f;
  for (int i = lane; i < np; i += 64) {
    const volatile float* q = part + i * stride;
    const float d = q[0] - mean;
    acc += q[1] + q[2] * d * d;
  }
  m2 = wave_sum(acc);
  cnt = sn;
}

struct NormK {
  int N, C, HW, spl, mode;
  float eps, momentum;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  int64_t* nbt;
  float *scale, *shift, *mean_out, *rstd_out;
  int ngroups;          // BatchNorm: passes batched into this launch (>= 1)
  int gstart[9];
  float *stat_mean, *stat_uvar;          // optional [C]: record batch mean / unbiased variance of group 0
  const float *ext_mean, *ext_uvar;      // optional [C]: statistics of a separately launched pass ...
  int ext_after;                         // ... whose running-statistics update follows pass `ext_after`
};

// running-statistics recurrence of one pass (PyTorch BatchNorm2d, momentum form, unbiased variance)
__device__ __forceinline__ void running_update(const NormK& k, int c, float mean, float uvar) {
  if (k.running_mean) k.running_mean[c] = (1.f - k.momentum) * k.running_mean[c] + k.momentum * mean;
  if (k.running_var) k.running_var[c] = (1.f - k.momentum) * k.running_var[c] + k.momentum * uvar;
}
__device__ __forceinline__ void after_group(const NormK& k, int c, int g, float mean, float uvar) {
  if (g == 0 && k.stat_mean) k.stat_mean[c] = mean;
  if (g == 0 && k.stat_uvar) k.stat_uvar[c] = uvar;
  running_update(k, c, mean, uvar);
  if (k.ext_mean && g == k.ext_after) running_update(k, c, k.ext_mean[c], k.ext_uvar[c]);
  if (k.nbt && c == 0) k.nbt[0] += 1 + ((k.ext_mean && g == k.ext_after) ? 1 : 0);
}

// IN: one wave per (n,c).  BN: one wave per c, merging N*spl partials, writing all n.
__device__ __forceinline__ void norm_finalize_group(const volatile float* part, const NormK& k, int group) {
  const int lane = threadIdx.x & 63;
  if (k.mode == 0) {
    const int g = group;  // n*C + c
    float mean, m2, cnt;
    chan_merge_wave(part + (int64_t)g * k.spl * 3, k.spl, 3, mean, m2, cnt);
    if (lane == 0) {
      const float rstd = 1.f / sqrtf(m2 / cnt + k.eps);
      k.scale[g] = rstd;
      k.shift[g] = -mean * rstd;
      if (k.mean_out) k.mean_out[g] = mean;
      if (k.rstd_out) k.rstd_out[g] = rstd;
    }
  } else {
    const int c = group;
    // partials of channel c: for n in a pass, s in spl -> index ((n*C + c)*spl + s); passes in order (running statistics)
    for (int gi = 0; gi < k.ngroups; ++gi) {
      const int n0 = k.gstart[gi], n1 = k.gstart[gi + 1];
      float sn = 0.f, sm = 0.f;
      const int np = (n1 - n0) * k.spl;
      for (int i = lane; i < np; i += 64) {
        const int n = n0 + i / k.spl, s = i - (i / k.spl) * k.spl;
        const volatile float* q = part + (((int64_t)n * k.C + c) * k.spl + s) * 3;
        sn += q[2];
        sm += q[2] * q[0];
      }
      sn = wave_sum(sn);
      sm = wave_sum(sm);
      const float mean = sm / sn;
      float acc = 0.f;
      for (int i = lane; i < np; i += 64) {
        const int n = n0 + i / k.spl, s = i - (i / k.spl) * k.spl;
        const volatile float* q = part + (((int64_t)n * k.C + c) * k.spl + s) * 3;
        const float d = q[0] - mean;
        acc += q[1] + q[2] * d * d;
      }
      const float m2 = wave_sum(acc);
      const float var = m2 / sn;
      const float rstd = 1.f / sqrtf(var + k.eps);
      const float g = k.gamma ? k.gamma[c] : 1.f, b = k.beta ? k.beta[c] : 0.f;
      for (int n = n0 + lane; n < n1; n += 64) {
        const int idx = n * k.C + c;
        k.scale[idx] = g * rstd;
        k.shift[idx] = b - mean * g * rstd;
        if (k.mean_out) k.mean_out[idx] = mean;
        if (k.rstd_out) k.rstd_out[idx] = rstd;
      }
      if (lane == 0) after_group(k, c, gi, mean, m2 / (sn - 1.f));
    }
  }
}

__global__ __launch_bounds__(64) void norm_finalize_kernel(const float* __restrict__ part, const NormK k) { norm_finalize_group(part, k, blockIdx.x); }

// The same second stage for MANY partials per group (round 3: the convolution's epilogue emits one partial per wave and tile, i.e.
// 1 000 .. 10 000 per group instead of HW / 2048): a 256-thread workgroup per group, plain (cached) loads, two block reductions per
// pass group.  Fixed order -> deterministic.
__global__ __launch_bounds__(256) void norm_finalize_wide_kernel(const float* __restrict__ part, const NormK k) {
  // The loops are written for memory-level parallelism: this kernel is a chain of L2 round trips (16 .. 64 workgroups on the whole
  // chip), so every thread keeps four independent record loads in flight (InstanceNorm: four slots; BatchNorm: four samples of one
  // slot) instead of one load per ~40-instruction iteration with an integer division (30 us for 8 x 1188 slots; round 3).
  __shared__ float red[16];
  const int tid = threadIdx.x;
  if (k.mode == 0) {
    const int g = blockIdx.x;
    const float* q0 = part + (int64_t)g * k.spl * 3;
    float sn = 0.f, sm = 0.f;
    int i = tid;
    for (; i + 768 < k.spl; i += 1024) {
      float c4[4], m4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c4[u] = q0[(i + 256 * u) * 3 + 2];
        m4[u] = q0[(i + 256 * u) * 3];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sn += c4[u];
        sm += c4[u] * m4[u];
      }
    }
    for (; i < k.spl; i += 256) {
      sn += q0[i * 3 + 2];
      sm += q0[i * 3 + 2] * q0[i * 3];
    }
    sn = block_sum(sn, red);
    sm = block_sum(sm, red);
    const float mean = sm / sn;
    float acc = 0.f;
    i = tid;
    for (; i + 768 < k.spl; i += 1024) {
      float c4[4], m4[4], v4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        m4[u] = q0[(i + 256 * u) * 3];
        v4[u] = q0[(i + 256 * u) * 3 + 1];
        c4[u] = q0[(i + 256 * u) * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = m4[u] - mean;
        acc += v4[u] + c4[u] * d * d;
      }
    }
    for (; i < k.spl; i += 256) {
      const float d = q0[i * 3] - mean;
      acc += q0[i * 3 + 1] + q0[i * 3 + 2] * d * d;
    }
    const float m2 = block_sum(acc, red);
    if (tid == 0) {
      const float rstd = 1.f / sqrtf(m2 / sn + k.eps);
      k.scale[g] = rstd;
      k.shift[g] = -mean * rstd;
      if (k.mean_out) k.mean_out[g] = mean;
      if (k.rstd_out) k.rstd_out[g] = rstd;
    }
    return;
  }
  const int c = blockIdx.x;
  const int64_t nstep = (int64_t)k.C * k.spl * 3;    // floats between the records of one slot in consecutive samples
  for (int gi = 0; gi < k.ngroups; ++gi) {
    const int n0 = k.gstart[gi], n1 = k.gstart[gi + 1];
    const float* base = part + ((int64_t)n0 * k.C + c) * k.spl * 3;
    float sn = 0.f, sm = 0.f;
    for (int s = tid; s < k.spl; s += 256) {
      const float* q = base + s * 3;
      int n = n0;
      for (; n + 3 < n1; n += 4, q += 4 * nstep) {
        float c4[4], m4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          c4[u] = q[u * nstep + 2];
          m4[u] = q[u * nstep];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          sn += c4[u];
          sm += c4[u] * m4[u];
        }
      }
      for (; n < n1; ++n, q += nstep) {
        sn += q[2];
        sm += q[2] * q[0];
      }
    }
    sn = block_sum(sn, red);
    sm = block_sum(sm, red);
    const float mean = sm / sn;
    float acc = 0.f;
    for (int s = tid; s < k.spl; s += 256) {
      const float* q = base + s * 3;
      int n = n0;
      for (; n + 3 < n1; n += 4, q += 4 * nstep) {
        float c4[4], m4[4], v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          m4[u] = q[u * nstep];
          v4[u] = q[u * nstep + 1];
          c4[u] = q[u * nstep + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = m4[u] - mean;
          acc += v4[u] + c4[u] * d * d;
        }
      }
      for (; n < n1; ++n, q += nstep) {
        const float d = q[0] - mean;
        acc += q[1] + q[2] * d * d;
      }
    }
    const float m2 = block_sum(acc, red);
    const float rstd = 1.f / sqrtf(m2 / sn + k.eps);
    const float g = k.gamma ? k.gamma[c] : 1.f, b = k.beta ? k.beta[c] : 0.f;
    for (int n = n0 + tid; n < n1; n += 256) {
      const int idx = n * k.C + c;
      k.scale[idx] = g * rstd;
      k.shift[idx] = b - mean * g * rstd;
      if (k.mean_out) k.mean_out[idx] = mean;
      if (k.rstd_out) k.rstd_out[idx] = rstd;
    }
    if (tid == 0) after_group(k, c, gi, mean, m2 / (sn - 1.f));
  }
}

__global__ __launch_bounds__(256) void stats_partial_kernel(const float* __restrict__ x, int64_t nstride, int C, int HW, int spl,
                                                            float* __restrict__ part) {
  stats_partial_body(x, nstride, C, HW, spl, part);
}

// partial statistics + finalize by the last workgroup of the group ((n, c) for InstanceNorm, c for BatchNorm)
__global__ __launch_bounds__(256) void stats_fin_kernel(const float* __restrict__ x, int64_t nstride, float* __restrict__ part, const NormK k,
                                                        int* __restrict__ counters) {
  __shared__ int flag;
  stats_partial_body(x, nstride, k.C, k.HW, k.spl, part);
  const int c = blockIdx.y, n = blockIdx.z;
  const int g = k.mode == 0 ? n * k.C + c : c, nblk = k.mode == 0 ? k.spl : k.N * k.spl;
  if (last_block_of(counters, g, nblk, &flag) && threadIdx.x < 64) norm_finalize_group(part, k, g);
}

// ---- backward: partial S1 = sum dy, S2 = sum dy * xhat per (n, c, split) ----
__device__ __forceinline__ void norm_bwd_partial_body(const float* __restrict__ dy, const float* __restrict__ x, int64_t nstride, int C,
                                                       int HW, int spl, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       float* __restrict__ part) {
  __shared__ float red[16];
  const int s = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const int64_t off = n * nstride + (int64_t)c * HW;
  const int base = s * CHUNK;
  const int cnt = min(CHUNK, HW - base);
  const float mu = mean[n * C + c], rs = rstd[n * C + c];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = e * 256 + threadIdx.x;
    if (i < cnt) {
      const float g = dy[off + base + i];
      const float xh = (x[off + base + i] - mu) * rs;
      s1 += g;
      s2 += g * xh;
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    float* o = part + (((int64_t)n * C + c) * spl + s) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

struct NormBwdK {
  int N, C, HW, spl, mode;
  const float *mean, *rstd, *gamma;
  float *dgamma, *dbeta;
  int acc;
  float* coef;  // [N*C][4]: A, B, C, mean
  int ngroups;  // BatchNorm: passes batched into this launch (>= 1)
  int gstart[9];
  // partials from a convolution epilogue (vts_conv4x4_bsums): `pspl` (S1, S2') pairs per (n, channel) with S2' = sum dy * (gamma xhat + beta);
  // S2 = (S2' - beta S1) / gamma.  pspl = 0: partials of norm_bwd_partial_kernel (spl per (n, channel), S2 as is).
  int pspl;
  const float* beta;
};

__device__ __forceinline__ void norm_bwd_finalize_group(const volatile float* part, const NormBwdK& k, int group) {
  const int lane = threadIdx.x & 63;
  if (k.mode == 0) {
    const int g = group;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < k.spl; i += 64) {
      s1 += part[((int64_t)g * k.spl + i) * 2];
      s2 += part[((int64_t)g * k.spl + i) * 2 + 1];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      const float m = (float)k.HW, rs = k.rstd[g], mu = k.mean[g];
      const float B = -rs * rs * s2 / m;
      k.coef[g * 4 + 0] = rs;
      k.coef[g * 4 + 1] = B;
      k.coef[g * 4 + 2] = -rs * s1 / m;
      k.coef[g * 4 + 3] = mu;
    }
  } else {
    const int c = group;
    float dg = 0.f, db = 0.f;
    for (int gi = 0; gi < k.ngroups; ++gi) {
      const int n0 = k.gstart[gi], n1 = k.gstart[gi + 1];
      float s1 = 0.f, s2 = 0.f;
      const int np = (n1 - n0) * k.spl;
      for (int i = lane; i < np; i += 64) {
        const int n = n0 + i / k.spl, s = i - (i / k.spl) * k.spl;
        const volatile float* q = part + (((int64_t)n * k.C + c) * k.spl + s) * 2;
        s1 += q[0];
        s2 += q[1];
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      const float m = (float)(n1 - n0) * (float)k.HW;
      const float rs = k.rstd[n0 * k.C + c], mu = k.mean[n0 * k.C + c];  // identical for every n of a pass
      const float g = k.gamma ? k.gamma[c] : 1.f;
      const float B = -g * rs * rs * s2 / m;
      for (int n = n0 + lane; n < n1; n += 64) {
        const int idx = n * k.C + c;
        k.coef[idx * 4 + 0] = g * rs;
        k.coef[idx * 4 + 1] = B;
        k.coef[idx * 4 + 2] = -g * rs * s1 / m;
        k.coef[idx * 4 + 3] = mu;
      }
      dg += s2;
      db += s1;
    }
    if (lane == 0) {
      if (k.dgamma) k.dgamma[c] = (k.acc ? k.dgamma[c] : 0.f) + dg;
      if (k.dbeta) k.dbeta[c] = (k.acc ? k.dbeta[c] : 0.f) + db;
    }
  }
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(float* __restrict__ dy, const float* __restrict__ x, int64_t nstride,
                                                             int C, int HW, const float* __restrict__ coef) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* q = coef + (n * C + c) * 4;
  const float A = q[0], B = q[1], Cc = q[2], mu = q[3];   // dx = A dy + B (x - mean) + C: centred, no cancellation against B * mean
  const int64_t off = n * nstride + (int64_t)c * HW;
  const int base = blockIdx.x * CHUNK;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    if (i < HW) dy[off + i] = A * dy[off + i] + B * (x[off + i] - mu) + Cc;
  }
}

// apply with the finalize folded in (round 2: one launch and one dependency hop fewer per normalisation backward): every workgroup
// reduces the partial sums of ITS group itself, in the order of norm_bwd_finalize_group (bit-identical coefficients in every
// workgroup), the BatchNorm parameter gradients are written by the first workgroup of each channel.
__device__ __forceinline__ void bwd_group_sums(const float* __restrict__ part, const NormBwdK& k, int c, int n0, int n1, float& s1, float& s2) {
  if (k.pspl > 0) {     // epilogue partials: thousands per group -> the whole workgroup sums them (fixed order), then the affine is undone
    __shared__ float red[16];
    const int np = (n1 - n0) * k.pspl;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) {
      const int n = n0 + i / k.pspl, s = i - (i / k.pspl) * k.pspl;
      const float* q = part + (((int64_t)n * k.C + c) * k.pspl + s) * 2;
      a += q[0];
      b += q[1];
    }
    s1 = block_sum(a, red);
    const float sp = block_sum(b, red);
    const float ga = (k.mode == 1 && k.gamma) ? k.gamma[c] : 1.f, be = (k.mode == 1 && k.beta) ? k.beta[c] : 0.f;
    s2 = (sp - be * s1) / ga;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int np = (n1 - n0) * k.spl;
  s1 = 0.f;
  s2 = 0.f;
  for (int i = lane; i < np; i += 64) {
    const int n = n0 + i / k.spl, s = i - (i / k.spl) * k.spl;
    const float* q = part + (((int64_t)n * k.C + c) * k.spl + s) * 2;
    s1 += q[0];
    s2 += q[1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
}

__global__ __launch_bounds__(256) void norm_bwd_apply_fin_kernel(float* __restrict__ dy, const float* __restrict__ x, int64_t nstride,
                                                                 const float* __restrict__ part, const NormBwdK k) {
  const int c = blockIdx.y, n = blockIdx.z;
  int n0 = n, n1 = n + 1;
  if (k.mode == 1) {
    int gi = 0;
    while (gi + 1 < k.ngroups && n >= k.gstart[gi + 1]) ++gi;
    n0 = k.gstart[gi];
    n1 = k.gstart[gi + 1];
  }
  float s1, s2;
  bwd_group_sums(part, k, c, n0, n1, s1, s2);
  const float m = (float)(n1 - n0) * (float)k.HW;
  const float rs = k.rstd[n0 * k.C + c], mu = k.mean[n0 * k.C + c];
  const float ga = (k.mode == 1 && k.gamma) ? k.gamma[c] : 1.f;
  const float A = ga * rs, B = -ga * rs * rs * s2 / m, Cc = -ga * rs * s1 / m;
  if (k.mode == 1 && blockIdx.x == 0 && n == 0 && (threadIdx.x < 64 || k.pspl > 0) && (k.dgamma || k.dbeta)) {    // (uniform per workgroup)
    float dg = 0.f, db = 0.f;
    for (int gi = 0; gi < k.ngroups; ++gi) {
      float t1, t2;
      bwd_group_sums(part, k, c, k.gstart[gi], k.gstart[gi + 1], t1, t2);
      dg += t2;
      db += t1;
    }
    if (threadIdx.x == 0) {
      if (k.dgamma) k.dgamma[c] = (k.acc ? k.dgamma[c] : 0.f) + dg;
      if (k.dbeta) k.dbeta[c] = (k.acc ? k.dbeta[c] : 0.f) + db;
    }
  }
  const int64_t off = n * nstride + (int64_t)c * k.HW;
  const int base = blockIdx.x * CHUNK;
  float g[EPT], xv[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    g[e] = i < k.HW ? dy[off + i] : 0.f;
    xv[e] = i < k.HW ? x[off + i] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    if (i < k.HW) dy[off + i] = A * g[e] + B * (xv[e] - mu) + Cc;
  }
}

// ---- apply pass on EPILOGUE sums (round 3).  The backward-data convolution leaves thousands of (S1, S2') pairs per group (one per
// wave and tile); norm_bwd_apply_fin_kernel above makes every 2048-element workgroup re-sum all of them -- 128 workgroups per
// 512 x 512 plane each reading 32 KB of pairs = more L2 traffic than the tensor itself, behind a chain of dependent loads (45 us for
// a 126 MB pass).  Here a workgroup owns `chunk` elements (8 - 32 K: 4 - 16x fewer re-summations), sums the pairs with four
// independent loads in flight per thread and no per-item division, and streams its elements with 16-byte accesses where the
// planes allow it.  Fixed summation order -> deterministic, identical coefficients in every workgroup of a group.
__device__ __forceinline__ void bwd_epilogue_sums(const float* __restrict__ part, const NormBwdK& k, int c, int n0, int n1, float* red, float& s1,
                                                  float& s2) {
  const int tid = threadIdx.x;
  float a = 0.f, b = 0.f;
  if (n1 - n0 == 1) {
    const float2* q = reinterpret_cast<const float2*>(part) + ((int64_t)n0 * k.C + c) * k.pspl;
    int i = tid;
    for (; i + 768 < k.pspl; i += 1024) {
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = q[i + 256 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a += v[u].x;
        b += v[u].y;
      }
    }
    for (; i < k.pspl; i += 256) {
      a += q[i].x;
      b += q[i].y;
    }
  } else {
    const int64_t nstep = (int64_t)k.C * k.pspl;     // float2 records between consecutive samples of one slot
    const float2* base = reinterpret_cast<const float2*>(part) + ((int64_t)n0 * k.C + c) * k.pspl;
    for (int sl = tid; sl < k.pspl; sl += 256) {
      const float2* q = base + sl;
      int n = n0;
      for (; n + 3 < n1; n += 4, q += 4 * nstep) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = q[u * nstep];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a += v[u].x;
          b += v[u].y;
        }
      }
      for (; n < n1; ++n, q += nstep) {
        a += q->x;
        b += q->y;
      }
    }
  }
  s1 = block_sum(a, red);
  const float sp = block_sum(b, red);
  const float ga = (k.mode == 1 && k.gamma) ? k.gamma[c] : 1.f, be = (k.mode == 1 && k.beta) ? k.beta[c] : 0.f;
  s2 = (sp - be * s1) / ga;
}

template <bool VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply_sums_kernel(float* __restrict__ dy, const float* __restrict__ x, int64_t nstride,
                                                                  const float* __restrict__ part, const NormBwdK k, int chunk) {
  __shared__ float red[16];
  const int c = blockIdx.y, n = blockIdx.z;
  int n0 = n, n1 = n + 1;
  if (k.mode == 1) {
    int gi = 0;
    while (gi + 1 < k.ngroups && n >= k.gstart[gi + 1]) ++gi;
    n0 = k.gstart[gi];
    n1 = k.gstart[gi + 1];
  }
  float s1, s2;
  bwd_epilogue_sums(part, k, c, n0, n1, red, s1, s2);
  const float m = (float)(n1 - n0) * (float)k.HW;
  const float rs = k.rstd[n0 * k.C + c], mu = k.mean[n0 * k.C + c];
  const float ga = (k.mode == 1 && k.gamma) ? k.gamma[c] : 1.f;
  const float A = ga * rs, B = -ga * rs * rs * s2 / m, Cc = -ga * rs * s1 / m;
  if (k.mode == 1 && blockIdx.x == 0 && n == 0 && (k.dgamma || k.dbeta)) {    // (uniform per workgroup)
    float dg = 0.f, db = 0.f;
    for (int gi = 0; gi < k.ngroups; ++gi) {
      float t1, t2;
      bwd_epilogue_sums(part, k, c, k.gstart[gi], k.gstart[gi + 1], red, t1, t2);
      dg += t2;
      db += t1;
    }
    if (threadIdx.x == 0) {
      if (k.dgamma) k.dgamma[c] = (k.acc ? k.dgamma[c] : 0.f) + dg;
      if (k.dbeta) k.dbeta[c] = (k.acc ? k.dbeta[c] : 0.f) + db;
    }
  }
  const int64_t off = n * nstride + (int64_t)c * k.HW;
  const int base = blockIdx.x * chunk, end = min(base + chunk, k.HW);
  if (VEC) {
    f32x4* d4 = reinterpret_cast<f32x4*>(dy + off);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + off);
    const int b4 = base >> 2, e4 = end >> 2;
    int i = b4 + threadIdx.x;
    for (; i + 256 < e4; i += 512) {
      const f32x4 g0 = d4[i], g1 = d4[i + 256], x0 = x4[i], x1 = x4[i + 256];
      d4[i] = A * g0 + B * (x0 - mu) + Cc;
      d4[i + 256] = A * g1 + B * (x1 - mu) + Cc;
    }
    for (; i < e4; i += 256) d4[i] = A * d4[i] + B * (x4[i] - mu) + Cc;
  } else {
    int i = base + threadIdx.x;
    for (; i + 768 < end; i += 1024) {
      float g[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = dy[off + i + 256 * u];
        xv[u] = x[off + i + 256 * u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) dy[off + i + 256 * u] = A * g[u] + B * (xv[u] - mu) + Cc;
    }
    for (; i < end; i += 256) dy[off + i] = A * dy[off + i] + B * (x[off + i] - mu) + Cc;
  }
}

// ---- single-launch variants for small groups (<= FUSED_MAX_GROUP elements per normalisation group):
// one workgroup owns a whole group (IN: one (n,c) plane; BN: channel c of every image), reads it twice
// (the second pass hits L2) and finishes in place.  Most normalisation calls of the step are this small
// (inner U-Net layers, every D2 patch pass), where three launches cost more than the arithmetic.
constexpr int64_t FUSED_MAX_GROUP = 8192;
constexpr int FUSED_BN_MAX_HW = 400;            // BatchNorm over many small maps (>= 32 channels): one workgroup per channel up to this map size ...
constexpr int64_t FUSED_BN_MAX_GROUP = 1 << 17;  // ... and this many elements per pass

// element j of group (c; n0..) -> offset into the NCHW tensor; BN groups flatten (n, i) so that every thread has work
__device__ __forceinline__ int64_t fused_off(int j, int HW, int n0, int c, int64_t nstride, bool bn) {
  int n = n0, i = j;
  if (bn) {
    const int q = j / HW;
    n = n0 + q;
    i = j - q * HW;
  }
  return n * nstride + (int64_t)c * HW + i;
}

// (n, i) of flattened element j of a BatchNorm group without an integer division: j < 2^24, so the float quotient is off by at most one
__device__ __forceinline__ int64_t fused_off_fast(int j, int HW, float inv_hw, int n0, int c, int64_t nstride, bool bn) {
  int n = n0, i = j;
  if (bn) {
    int q = (int)((float)j * inv_hw);
    i = j - q * HW;
    if (i < 0) { --q; i += HW; }
    if (i >= HW) { ++q; i -= HW; }
    n = n0 + q;
  }
  return n * nstride + (int64_t)c * HW + i;
}

// Register-resident form (round 2): a group of <= EPT * BLOCK elements is read ONCE, all loads of a pass group in flight together
// (the loop form below issued one dependent load per iteration and read every element twice: 15 us average, 50 us on the D2 patch
// layers).  The loads of the next BatchNorm pass group are issued before the reductions of the current one.  Same thread <-> element
// mapping and summation order as the loop form: bit-identical statistics.
template <int EPT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void norm_stats_fused_kernel(const float* __restrict__ x, int64_t nstride, const NormK k) {
  __shared__ float red[16];
  const int g = blockIdx.x;
  const bool bn = k.mode == 1;
  const int c = bn ? g : g % k.C;
  const int ngr = bn ? k.ngroups : 1;
  const float inv_hw = 1.f / (float)k.HW;
  float v[EPT], vn[EPT] = {};
  auto load = [&](int gi, float (&dst)[EPT]) {
    const int n0 = bn ? k.gstart[gi] : g / k.C, n1 = bn ? k.gstart[gi + 1] : n0 + 1;
    const int total = (n1 - n0) * k.HW;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int j = e * BLOCK + threadIdx.x;
      dst[e] = j < total ? x[fused_off_fast(j, k.HW, inv_hw, n0, c, nstride, bn)] : 0.f;
    }
  };
  load(0, v);
  for (int gi = 0; gi < ngr; ++gi) {   // BatchNorm: the passes of this channel in order (running statistics)
    const int n0 = bn ? k.gstart[gi] : g / k.C, n1 = bn ? k.gstart[gi + 1] : n0 + 1;
    const int total = (n1 - n0) * k.HW;
    const float cnt = (float)total;
    if (gi + 1 < ngr) load(gi + 1, vn);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) s += v[e];
    const float mean = block_sum(s, red) / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const float d = v[e] - mean;
      if (e * BLOCK + (int)threadIdx.x < total) m2 += d * d;
    }
    m2 = block_sum(m2, red);
    const float rstd = 1.f / sqrtf(m2 / cnt + k.eps);
    const float ga = (bn && k.gamma) ? k.gamma[c] : 1.f, be = (bn && k.beta) ? k.beta[c] : 0.f;
    for (int n = n0 + threadIdx.x; n < n1; n += BLOCK) {
      const int idx = n * k.C + c;
      k.scale[idx] = ga * rstd;
      k.shift[idx] = be - mean * ga * rstd;
      if (k.mean_out) k.mean_out[idx] = mean;
      if (k.rstd_out) k.rstd_out[idx] = rstd;
    }
    if (bn && threadIdx.x == 0) after_group(k, c, gi, mean, m2 / (cnt - 1.f));
#pragma unroll
    for (int e = 0; e < EPT; ++e) v[e] = vn[e];
  }
}

// loop form: groups beyond the register budget
__global__ __launch_bounds__(1024) void norm_stats_fused_loop_kernel(const float* __restrict__ x, int64_t nstride, const NormK k) {
  __shared__ float red[16];
  const int g = blockIdx.x;
  const bool bn = k.mode == 1;
  const int c = bn ? g : g % k.C;
  const int ngr = bn ? k.ngroups : 1;
  for (int gi = 0; gi < ngr; ++gi) {
    const int n0 = bn ? k.gstart[gi] : g / k.C, n1 = bn ? k.gstart[gi + 1] : n0 + 1;
    const int total = (n1 - n0) * k.HW;
    const float cnt = (float)total;
    float s = 0.f;
    for (int j = threadIdx.x; j < total; j += blockDim.x) s += x[fused_off(j, k.HW, n0, c, nstride, bn)];
    const float mean = block_sum(s, red) / cnt;
    float m2 = 0.f;
    for (int j = threadIdx.x; j < total; j += blockDim.x) {
      const float d = x[fused_off(j, k.HW, n0, c, nstride, bn)] - mean;
      m2 += d * d;
    }
    m2 = block_sum(m2, red);
    const float rstd = 1.f / sqrtf(m2 / cnt + k.eps);
    const float ga = (bn && k.gamma) ? k.gamma[c] : 1.f, be = (bn && k.beta) ? k.beta[c] : 0.f;
    for (int n = n0 + threadIdx.x; n < n1; n += blockDim.x) {
      const int idx = n * k.C + c;
      k.scale[idx] = ga * rstd;
      k.shift[idx] = be - mean * ga * rstd;
      if (k.mean_out) k.mean_out[idx] = mean;
      if (k.rstd_out) k.rstd_out[idx] = rstd;
    }
    if (bn && threadIdx.x == 0) after_group(k, c, gi, mean, m2 / (cnt - 1.f));
  }
}

template <int EPT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void norm_bwd_fused_kernel(float* __restrict__ dy, const float* __restrict__ x, int64_t nstride,
                                                                const NormBwdK k) {
  __shared__ float red[16];
  const int g = blockIdx.x;
  const bool bn = k.mode == 1;
  const int c = bn ? g : g % k.C;
  const int ngr = bn ? k.ngroups : 1;
  const float inv_hw = 1.f / (float)k.HW;
  float dg = 0.f, db = 0.f;
  float gv[EPT], xv[EPT];
  auto load = [&](int gi, float (&gd)[EPT], float (&xd)[EPT]) {
    const int n0 = bn ? k.gstart[gi] : g / k.C, n1 = bn ? k.gstart[gi + 1] : n0 + 1;
    const int total = (n1 - n0) * k.HW;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int j = e * BLOCK + threadIdx.x;
      const int64_t off = fused_off_fast(j < total ? j : 0, k.HW, inv_hw, n0, c, nstride, bn);
      gd[e] = j < total ? dy[off] : 0.f;
      xd[e] = j < total ? x[off] : 0.f;
    }
  };
  for (int gi = 0; gi < ngr; ++gi) {
    load(gi, gv, xv);   // (no cross-group prefetch here: two operands x two buffers spill at 1024 threads)
    const int n0 = bn ? k.gstart[gi] : g / k.C, n1 = bn ? k.gstart[gi + 1] : n0 + 1;
    const int total = (n1 - n0) * k.HW;
    const float m = (float)total;
    const float mu = k.mean[n0 * k.C + c], rs = k.rstd[n0 * k.C + c];  // identical for every n of a BN pass
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      if (e * BLOCK + (int)threadIdx.x < total) {
        s1 += gv[e];
        s2 += gv[e] * ((xv[e] - mu) * rs);
      }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float ga = (bn && k.gamma) ? k.gamma[c] : 1.f;
    const float A = ga * rs, B = -ga * rs * rs * s2 / m, Cc = -ga * rs * s1 / m;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int j = e * BLOCK + threadIdx.x;
      if (j < total) dy[fused_off_fast(j, k.HW, inv_hw, n0, c, nstride, bn)] = A * gv[e] + B * (xv[e] - mu) + Cc;
    }
    dg += s2;
    db += s1;
  }
  if (bn && threadIdx.x == 0) {
    if (k.dgamma) k.dgamma[c] = (k.acc ? k.dgamma[c] : 0.f) + dg;
    if (k.dbeta) k.dbeta[c] = (k.acc ? k.dbeta[c] : 0.f) + db;
  }
}

__global__ __launch_bounds__(1024) void norm_bwd_fused_loop_kernel(float* __restrict__ dy, const float* __restrict__ x, int64_t nstride,
                                                                   const NormBwdK k) {
  __shared__ float red[16];
  const int g = blockIdx.x;
  const bool bn = k.mode == 1;
  const int c = bn ? g : g % k.C;
  const int ngr = bn ? k.ngroups : 1;
  float dg = 0.f, db = 0.f;
  for (int gi = 0; gi < ngr; ++gi) {
    const int n0 = bn ? k.gstart[gi] : g / k.C, n1 = bn ? k.gstart[gi + 1] : n0 + 1;
    const int total = (n1 - n0) * k.HW;
    const float m = (float)total;
    const float mu = k.mean[n0 * k.C + c], rs = k.rstd[n0 * k.C + c];  // identical for every n of a BN pass
    float s1 = 0.f, s2 = 0.f;
    for (int j = threadIdx.x; j < total; j += blockDim.x) {
      const int64_t off = fused_off(j, k.HW, n0, c, nstride, bn);
      const float gdy = dy[off];
      s1 += gdy;
      s2 += gdy * ((x[off] - mu) * rs);
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float ga = (bn && k.gamma) ? k.gamma[c] : 1.f;
    const float A = ga * rs, B = -ga * rs * rs * s2 / m, Cc = -ga * rs * s1 / m;
    for (int j = threadIdx.x; j < total; j += blockDim.x) {
      const int64_t off = fused_off(j, k.HW, n0, c, nstride, bn);
      dy[off] = A * dy[off] + B * (x[off] - mu) + Cc;
    }
    dg += s2;
    db += s1;
  }
  if (bn && threadIdx.x == 0) {
    if (k.dgamma) k.dgamma[c] = (k.acc ? k.dgamma[c] : 0.f) + dg;
    if (k.dbeta) k.dbeta[c] = (k.acc ? k.dbeta[c] : 0.f) + db;
  }
}

__global__ __launch_bounds__(64) void norm_bwd_finalize_kernel(const float* __restrict__ part, const NormBwdK k) { norm_bwd_finalize_group(part, k, blockIdx.x); }

__global__ __launch_bounds__(256) void norm_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t nstride, int C,
                                                               int HW, int spl, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ part) {
  norm_bwd_partial_body(dy, x, nstride, C, HW, spl, mean, rstd, part);
}

__global__ __launch_bounds__(256) void norm_bwd_fin_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t nstride,
                                                           float* __restrict__ part, const NormBwdK k, int* __restrict__ counters) {
  __shared__ int flag;
  norm_bwd_partial_body(dy, x, nstride, k.C, k.HW, k.spl, k.mean, k.rstd, part);
  const int c = blockIdx.y, n = blockIdx.z;
  const int g = k.mode == 0 ? n * k.C + c : c, nblk = k.mode == 0 ? k.spl : k.N * k.spl;
  if (last_block_of(counters, g, nblk, &flag) && threadIdx.x < 64) norm_bwd_finalize_group(part, k, g);
}

// ---- channel sum ----
__device__ __forceinline__ void chsum_partial_body(const float* __restrict__ x, int64_t nstride, int C, int HW, int spl,
                                                    float* __restrict__ part) {
  __shared__ float red[16];
  const int s = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
  const float* px = x + n * nstride + (int64_t)c * HW;
  const int base = s * CHUNK;
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    if (i < HW) sum += px[i];
  }
  sum = block_sum(sum, red);
  if (threadIdx.x == 0) part[((int64_t)n * C + c) * spl + s] = sum;
}

__device__ __forceinline__ void chsum_finalize_group(const volatile float* part, int N, int C, int spl, float* __restrict__ out, int accumulate,
                                                      int c) {
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < N * spl; i += 64) {
    const int n = i / spl, j = i - n * spl;
    s += part[((int64_t)n * C + c) * spl + j];
  }
  s = wave_sum(s);
  if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

__global__ __launch_bounds__(64) void chsum_finalize_kernel(const float* __restrict__ part, int N, int C, int spl, float* __restrict__ out,
                                                            int accumulate) {
  chsum_finalize_group(part, N, C, spl, out, accumulate, blockIdx.x);
}

// many partials per channel (full-resolution maps: N * HW / 2048 = 1000 .. 2000): a 256-thread workgroup, four independent loads in
// flight per thread, no per-item division (the one-wave form above: 18 us for 3 x 2048 partials).  Fixed order -> deterministic.
__global__ __launch_bounds__(256) void chsum_finalize_wide_kernel(const float* __restrict__ part, int N, int C, int spl, float* __restrict__ out,
                                                                  int accumulate) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s = 0.f;
  if (spl >= 64) {                      // few samples, many splits: the threads walk the splits of one sample
    for (int n = 0; n < N; ++n) {
      const float* q = part + ((int64_t)n * C + c) * spl;
      int j = threadIdx.x;
      for (; j + 768 < spl; j += 1024) s += (q[j] + q[j + 256]) + (q[j + 512] + q[j + 768]);
      for (; j < spl; j += 256) s += q[j];
    }
  } else {                              // many samples, few splits (the D2 patch passes: 640 x 1): a thread per sample
    for (int n = threadIdx.x; n < N; n += 256) {
      const float* q = part + ((int64_t)n * C + c) * spl;
      for (int j = 0; j < spl; ++j) s += q[j];
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + s : s;
}

__global__ __launch_bounds__(256) void chsum_partial_kernel(const float* __restrict__ x, int64_t nstride, int C, int HW, int spl,
                                                            float* __restrict__ part) {
  chsum_partial_body(x, nstride, C, HW, spl, part);
}

__global__ __launch_bounds__(256) void chsum_fin_kernel(const float* __restrict__ x, int64_t nstride, int N, int C, int HW, int spl,
                                                        float* __restrict__ part, float* __restrict__ out, int accumulate,
                                                        int* __restrict__ counters) {
  __shared__ int flag;
  chsum_partial_body(x, nstride, C, HW, spl, part);
  const int c = blockIdx.y;
  if (last_block_of(counters, c, N * spl, &flag) && threadIdx.x < 64) chsum_finalize_group(part, N, C, spl, out, accumulate, c);
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, int64_t xns,
                                                      const float* __restrict__ sc, const float* __restrict__ sh, int C, int HW,
                                                      int act, float* __restrict__ dy, int accumulate) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float a = sc ? sc[n * C + c] : 1.f, b = sh ? sh[n * C + c] : 0.f;
  const int64_t offx = n * xns + (int64_t)c * HW, off = ((int64_t)n * C + c) * HW;
  const int base = blockIdx.x * CHUNK;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = base + e * 256 + threadIdx.x;
    if (i < HW) {
      const float v = g[off + i] * vts_act_grad(x[offx + i] * a + b, act);
      dy[off + i] = accumulate ? dy[off + i] + v : v;
    }
  }
}

// pass groups of a batched BatchNorm launch: validated copy; maxg = samples of the largest pass
bool fill_groups(int mode, int N, int ngroups, const int* gstart, int& out_n, int* out_start, int& maxg) {
  out_n = 1;
  out_start[0] = 0;
  out_start[1] = N;
  maxg = N;
  if (mode != 1 || ngroups <= 1) return true;
  if (ngroups > 8 || gstart[0] != 0 || gstart[ngroups] != N) return false;
  maxg = 0;
  for (int g = 0; g < ngroups; ++g) {
    if (gstart[g + 1] <= gstart[g]) return false;
    if (gstart[g + 1] - gstart[g] > maxg) maxg = gstart[g + 1] - gstart[g];
  }
  out_n = ngroups;
  for (int g = 0; g <= ngroups; ++g) out_start[g] = gstart[g];
  return true;
}

}  // namespace

extern "C" int64_t vts_norm_ws_floats(int N, int C, int HW) { return (int64_t)N * C * splits_for(HW) * 3 + (int64_t)N * C * 4; }

extern "C" int vts_norm_stats(const vts_norm_desc* d, float* ws, void* stream) {
  VTS_CHECK_ARG(d && d->x && d->scale && d->shift && ws, "vts_norm_stats: null pointer");
  VTS_CHECK_ARG(d->mode == 0 || d->mode == 1, "vts_norm_stats: mode %d", d->mode);
  VTS_CHECK_ARG(d->N >= 1 && d->C >= 1 && d->HW >= 1 && d->N <= 65535 && d->C <= 65535, "vts_norm_stats: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int spl = splits_for(d->HW);
  NormK k{d->N, d->C, d->HW, spl, d->mode, d->eps, d->momentum, d->gamma, d->beta, d->running_mean, d->running_var,
          d->num_batches_tracked, d->scale, d->shift, d->mean_out, d->rstd_out};
  int maxg = d->N;
  if (!fill_groups(d->mode, d->N, d->ngroups, d->gstart, k.ngroups, k.gstart, maxg)) {
    vts_set_error("vts_norm_stats: bad pass groups (ngroups %d)", d->ngroups);
    return VTS_ERR_ARG;
  }
  k.stat_mean = d->stat_mean_out; k.stat_uvar = d->stat_uvar_out; k.ext_mean = d->ext_mean; k.ext_uvar = d->ext_uvar; k.ext_after = d->ext_after;
  VTS_CHECK_ARG(!(k.ext_mean && !k.ext_uvar) && !(k.stat_mean && !k.stat_uvar), "vts_norm_stats: ext / stat outputs come in pairs");
  const bool grouped = k.ngroups > 1 || k.stat_mean || k.ext_mean;
  const int64_t group = (int64_t)(d->mode == 0 ? 1 : maxg) * d->HW;
  // small groups: one launch, one workgroup per group (two passes, L2-resident).  BatchNorm over many tiny maps (the D2 passes:
  // 640 patches of 6x6 .. 9x9) also goes here: the partial kernel would launch N*C workgroups of a few dozen elements each.
  if (group <= FUSED_MAX_GROUP || (d->mode == 1 && d->C >= 32 && d->HW <= FUSED_BN_MAX_HW && group <= FUSED_BN_MAX_GROUP)) {
    const dim3 fg(d->mode == 0 ? d->N * d->C : d->C);
    if (group <= 4096) hipLaunchKernelGGL((norm_stats_fused_kernel<16, 256>), fg, dim3(256), 0, st, d->x, d->nstride, k);
    else if (group <= 8192) hipLaunchKernelGGL((norm_stats_fused_kernel<8, 1024>), fg, dim3(1024), 0, st, d->x, d->nstride, k);
    else if (group <= 16384) hipLaunchKernelGGL((norm_stats_fused_kernel<16, 1024>), fg, dim3(1024), 0, st, d->x, d->nstride, k);
    else hipLaunchKernelGGL(norm_stats_fused_loop_kernel, fg, dim3(1024), 0, st, d->x, d->nstride, k);
    VTS_CHECK_LAUNCH("vts_norm_stats fused");
    vts_set_kernel(group <= 16384 ? "norm_stats_fused_kernel" : "norm_stats_fused_loop_kernel");
    return VTS_OK;
  }
  if (d->counters && !grouped) {
    hipLaunchKernelGGL(stats_fin_kernel, dim3(spl, d->C, d->N), dim3(256), 0, st, d->x, d->nstride, ws, k, d->counters);
    VTS_CHECK_LAUNCH("vts_norm_stats");
    vts_set_kernel("stats_fin_kernel");
    return VTS_OK;
  }
  hipLaunchKernelGGL(stats_partial_kernel, dim3(spl, d->C, d->N), dim3(256), 0, st, d->x, d->nstride, d->C, d->HW, spl, ws);
  VTS_CHECK_LAUNCH("vts_norm_stats partial");
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(d->mode == 0 ? d->N * d->C : d->C), dim3(64), 0, st, ws, k);
  VTS_CHECK_LAUNCH("vts_norm_stats finalize");
  vts_set_kernel("stats_partial_kernel+norm_finalize_kernel");
  return VTS_OK;
}

// second stage of vts_norm_stats on partials produced elsewhere (the convolution's epilogue, vts_conv.hip): `spl` (mean, M2, count)
// slots per (n, channel) in the layout of stats_partial_kernel
int vts_norm_finalize_partials(const vts_norm_desc* d, const float* part, int spl, hipStream_t st) {
  VTS_CHECK_ARG(d && d->scale && d->shift && part && spl >= 1, "vts_norm_finalize_partials: null pointer");
  NormK k{d->N, d->C, d->HW, spl, d->mode, d->eps, d->momentum, d->gamma, d->beta, d->running_mean, d->running_var,
          d->num_batches_tracked, d->scale, d->shift, d->mean_out, d->rstd_out};
  int maxg = d->N;
  if (!fill_groups(d->mode, d->N, d->ngroups, d->gstart, k.ngroups, k.gstart, maxg)) {
    vts_set_error("vts_norm_finalize_partials: bad pass groups (ngroups %d)", d->ngroups);
    return VTS_ERR_ARG;
  }
  k.stat_mean = d->stat_mean_out; k.stat_uvar = d->stat_uvar_out; k.ext_mean = d->ext_mean; k.ext_uvar = d->ext_uvar; k.ext_after = d->ext_after;
  VTS_CHECK_ARG(!(k.ext_mean && !k.ext_uvar) && !(k.stat_mean && !k.stat_uvar), "vts_norm_finalize_partials: ext / stat outputs come in pairs");
  hipLaunchKernelGGL(norm_finalize_wide_kernel, dim3(d->mode == 0 ? d->N * d->C : d->C), dim3(256), 0, st, part, k);
  VTS_CHECK_LAUNCH("vts_norm_finalize_partials");
  vts_set_kernel("norm_finalize_wide_kernel");
  return VTS_OK;
}

extern "C" int vts_norm_bwd(const vts_norm_bwd_desc* d, float* ws, void* stream) {
  VTS_CHECK_ARG(d && d->dy && d->x && d->mean && d->rstd && ws, "vts_norm_bwd: null pointer");
  VTS_CHECK_ARG(d->mode == 0 || d->mode == 1, "vts_norm_bwd: mode %d", d->mode);
  hipStream_t st = (hipStream_t)stream;
  const int spl = splits_for(d->HW);
  float* part = ws;
  float* coef = ws + (int64_t)d->N * d->C * spl * 3;  // same split as vts_norm_ws_floats
  NormBwdK k{d->N, d->C, d->HW, spl, d->mode, d->mean, d->rstd, d->gamma, d->dgamma, d->dbeta, d->accumulate_param_grads, coef};
  int maxg = d->N;
  if (!fill_groups(d->mode, d->N, d->ngroups, d->gstart, k.ngroups, k.gstart, maxg)) {
    vts_set_error("vts_norm_bwd: bad pass groups (ngroups %d)", d->ngroups);
    return VTS_ERR_ARG;
  }
  const bool grouped = k.ngroups > 1;
  const int64_t group = (int64_t)(d->mode == 0 ? 1 : maxg) * d->HW;
  if (group <= FUSED_MAX_GROUP || (d->mode == 1 && d->C >= 32 && d->HW <= FUSED_BN_MAX_HW && group <= FUSED_BN_MAX_GROUP)) {
    const NormBwdK& kf = k;
    const dim3 fg(d->mode == 0 ? d->N * d->C : d->C);
    if (group <= 4096) hipLaunchKernelGGL((norm_bwd_fused_kernel<16, 256>), fg, dim3(256), 0, st, d->dy, d->x, d->nstride, kf);
    else if (group <= 8192) hipLaunchKernelGGL((norm_bwd_fused_kernel<8, 1024>), fg, dim3(1024), 0, st, d->dy, d->x, d->nstride, kf);
    else if (group <= 16384) hipLaunchKernelGGL((norm_bwd_fused_kernel<16, 1024>), fg, dim3(1024), 0, st, d->dy, d->x, d->nstride, kf);
    else hipLaunchKernelGGL(norm_bwd_fused_loop_kernel, fg, dim3(1024), 0, st, d->dy, d->x, d->nstride, kf);
    VTS_CHECK_LAUNCH("vts_norm_bwd fused");
    vts_set_kernel(group <= 16384 ? "norm_bwd_fused_kernel" : "norm_bwd_fused_loop_kernel");
    return VTS_OK;
  }
  if (d->counters && !grouped) {
    hipLaunchKernelGGL(norm_bwd_fin_kernel, dim3(spl, d->C, d->N), dim3(256), 0, st, d->dy, d->x, d->nstride, part, k, d->counters);
    VTS_CHECK_LAUNCH("vts_norm_bwd partial+finalize");
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(spl, d->C, d->N), dim3(256), 0, st, d->dy, d->x, d->nstride, d->C, d->HW, coef);
    VTS_CHECK_LAUNCH("vts_norm_bwd apply");
    vts_set_kernel("norm_bwd_fin_kernel+norm_bwd_apply_kernel");
    return VTS_OK;
  }
  hipLaunchKernelGGL(norm_bwd_partial_kernel, dim3(spl, d->C, d->N), dim3(256), 0, st, d->dy, d->x, d->nstride, d->C, d->HW, spl,
                     d->mean, d->rstd, part);
  VTS_CHECK_LAUNCH("vts_norm_bwd partial");
  static const int three = vts_tune("VTS_NORM_BWD_3K", 0);   // 1: the separate finalize launch (A/B timing)
  if (!three) {
    hipLaunchKernelGGL(norm_bwd_apply_fin_kernel, dim3(spl, d->C, d->N), dim3(256), 0, st, d->dy, d->x, d->nstride, part, k);
    VTS_CHECK_LAUNCH("vts_norm_bwd apply");
    vts_set_kernel("norm_bwd_partial_kernel+norm_bwd_apply_fin_kernel");
    return VTS_OK;
  }
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(d->mode == 0 ? d->N * d->C : d->C), dim3(64), 0, st, part, k);
  VTS_CHECK_LAUNCH("vts_norm_bwd finalize");
  hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(spl, d->C, d->N), dim3(256), 0, st, d->dy, d->x, d->nstride, d->C, d->HW, coef);
  VTS_CHECK_LAUNCH("vts_norm_bwd apply");
  vts_set_kernel("norm_bwd_partial_kernel+norm_bwd_finalize_kernel+norm_bwd_apply_kernel");
  return VTS_OK;
}

// vts_norm_bwd on sums a convolution epilogue produced (vts_conv4x4_bsums): only the apply pass runs
extern "C" int vts_norm_bwd_from_partials(const vts_norm_bwd_desc* d, const float* part, int slots, const float* beta, void* stream) {
  VTS_CHECK_ARG(d && d->dy && d->x && d->mean && d->rstd && part && slots >= 1, "vts_norm_bwd_from_partials: null pointer");
  VTS_CHECK_ARG(d->mode == 0 || d->mode == 1, "vts_norm_bwd_from_partials: mode %d", d->mode);
  const int spl = splits_for(d->HW);
  NormBwdK k{d->N, d->C, d->HW, spl, d->mode, d->mean, d->rstd, d->gamma, d->dgamma, d->dbeta, d->accumulate_param_grads, nullptr};
  int maxg = d->N;
  if (!fill_groups(d->mode, d->N, d->ngroups, d->gstart, k.ngroups, k.gstart, maxg)) {
    vts_set_error("vts_norm_bwd_from_partials: bad pass groups (ngroups %d)", d->ngroups);
    return VTS_ERR_ARG;
  }
  k.pspl = slots;
  k.beta = beta;
  static const int wide = vts_tune("VTS_NORM_BWD_SUMS", 1);   // 0: the 2048-element apply kernel (A/B)
  if (wide) {
    // elements per workgroup: multiples of 2048, as many as leave >= ~768 workgroups, at most 32 K
    int chunk = CHUNK;
    while (chunk < 32768 && (int64_t)cdiv(d->HW, 2 * chunk) * d->C * d->N >= 768) chunk *= 2;
    const bool vec = d->HW % 4 == 0 && d->nstride % 4 == 0 && ((reinterpret_cast<uintptr_t>(d->dy) | reinterpret_cast<uintptr_t>(d->x)) & 15) == 0;
    const dim3 grid(cdiv(d->HW, chunk), d->C, d->N);
    if (vec) hipLaunchKernelGGL(norm_bwd_apply_sums_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, d->dy, d->x, d->nstride, part, k, chunk);
    else hipLaunchKernelGGL(norm_bwd_apply_sums_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, d->dy, d->x, d->nstride, part, k, chunk);
    VTS_CHECK_LAUNCH("vts_norm_bwd_from_partials");
    vts_set_kernel("norm_bwd_apply_sums_kernel");
    return VTS_OK;
  }
  hipLaunchKernelGGL(norm_bwd_apply_fin_kernel, dim3(spl, d->C, d->N), dim3(256), 0, (hipStream_t)stream, d->dy, d->x, d->nstride, part, k);
  VTS_CHECK_LAUNCH("vts_norm_bwd_from_partials");
  vts_set_kernel("norm_bwd_apply_fin_kernel");
  return VTS_OK;
}

extern "C" int64_t vts_channel_sum_ws_floats(int N, int C, int HW) { return (int64_t)N * C * splits_for(HW); }

extern "C" int vts_channel_sum(const float* x, int64_t nstride, int N, int C, int HW, float* out, int accumulate, float* ws,
                               int* counters, void* stream) {
  VTS_CHECK_ARG(x && out && ws, "vts_channel_sum: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int spl = splits_for(HW);
  if (counters) {
    hipLaunchKernelGGL(chsum_fin_kernel, dim3(spl, C, N), dim3(256), 0, st, x, nstride, N, C, HW, spl, ws, out, accumulate, counters);
    VTS_CHECK_LAUNCH("vts_channel_sum");
    return VTS_OK;
  }
  hipLaunchKernelGGL(chsum_partial_kernel, dim3(spl, C, N), dim3(256), 0, st, x, nstride, C, HW, spl, ws);
  VTS_CHECK_LAUNCH("vts_channel_sum partial");
  if ((int64_t)N * spl >= 256) hipLaunchKernelGGL(chsum_finalize_wide_kernel, dim3(C), dim3(256), 0, st, ws, N, C, spl, out, accumulate);
  else hipLaunchKernelGGL(chsum_finalize_kernel, dim3(C), dim3(64), 0, st, ws, N, C, spl, out, accumulate);
  VTS_CHECK_LAUNCH("vts_channel_sum finalize");
  return VTS_OK;
}

extern "C" int vts_act_bwd(const float* g, const vts_operand* x, int N, int HW, int act, float* dy, int accumulate, void* stream) {
  VTS_CHECK_ARG(g && x && x->data && dy, "vts_act_bwd: null pointer");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(splits_for(HW), x->C, N), dim3(256), 0, (hipStream_t)stream, g, x->data, x->nstride,
                     x->scale, x->shift, x->C, HW, act, dy, accumulate);
  VTS_CHECK_LAUNCH("vts_act_bwd");
  return VTS_OK;
}
