// PatchNCE on the fp32 MFMA path (gfx950), PatchSampleF's gather and MLP layers.
//
// Reference: models/patchnce.py:13-55 (PatchNCELoss), models/networks.py:667-719 (PatchSampleF), :585-594 (Normalize).
//
// patchnce_mfma_kernel: one workgroup per (image, block of 64 query patches).  The 64 x P logits never leave LDS:
//   1. S = Q_blk K^T on v_mfma_f32_16x16x4_f32 (wave w: query rows 16w..16w+15 against all P keys, P/16 accumulator tiles);
//      Q / K are staged through LDS in slices of 32 feature dimensions (coalesced 16-byte loads, pitch 36: 2-way worst case);
//   2. logits / T with the diagonal replaced by -10 / T, the positive logit q_i.k_i kept aside; row-wise softmax cross-entropy
//      with wave shuffles (a wave owns its 16 rows, a row is spread over the 64 lanes);
//   3. dQ_blk = W K with W = softmax weights (p0 - 1 on the diagonal: the positive; the -10 entry has no gradient), again on MFMA
//      (wave w: its 16 rows against D/16 accumulator tiles), K staged in slices of 32 keys (pitch = 16 mod 32: conflict-free).
// P, D <= 256 (the reference: 256 patches x 256 dims); larger problems (all negatives of a minibatch: P = B * 256) use the
// one-row-per-workgroup kernel in vts_ops.hip.
#include "vts_internal.h"

namespace {

constexpr int RB = 64;          // query rows per workgroup
constexpr int MAXT = 16;        // accumulator tiles per wave: P <= 256 keys, D <= 256 dims
constexpr int DS = 32, DSP = 36;   // feature-dimension slice of phase 1 and its LDS pitch
constexpr int KS = 32;          // key slice of phase 3
constexpr int LP = 260;         // logits pitch (>= 256 + pad, 16-byte multiple)

// C[64 x N] (+)= A[64 x K] B[N x K]^T for this workgroup: rows r0.. of A (row pitch lda), all N <= 256 rows of B (pitch ldb),
// K arbitrary.  Wave w accumulates rows 16w..16w+15; acc[t] is the 16 x 16 tile of columns 16t.. (C layout: col = lane & 15,
// row = (lane >> 4) * 4 + reg).  Rows >= rows_valid / columns >= N / k >= K are zero-filled while staging.
__device__ __forceinline__ void gemm_nt_tile(const float* __restrict__ A, int64_t lda, int rows_valid, const float* __restrict__ B, int64_t ldb,
                                             int N, int K, float* As, float* Bs, f32x4 (&acc)[MAXT]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
  const int nt = (N + 15) >> 4;
  for (int k0 = 0; k0 < K; k0 += DS) {
    // stage A slice [64][32] and B slice [N16][32]: thread -> (row, 4 consecutive k)
    for (int idx = tid; idx < RB * (DS / 4); idx += 256) {
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows_valid) {
        const float* p = A + r * lda + k0 + c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k0 + c4 + e < K) ? p[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(As + r * DSP + c4) = v;
    }
    for (int idx = tid; idx < nt * 16 * (DS / 4); idx += 256) {
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < N) {
        const float* p = B + r * ldb + k0 + c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k0 + c4 + e < K) ? p[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(Bs + r * DSP + c4) = v;
    }
    __syncthreads();
    const float* ar = As + (wave * 16 + m16) * DSP + kq;
    const float* br = Bs + m16 * DSP + kq;
#pragma unroll
    for (int s = 0; s < DS / 4; ++s) {
      const float a = ar[s * 4];
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
        if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, br[t * 16 * DSP + s * 4], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void patchnce_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k, int P, int D, float invT,
                                                            float gscale, float* __restrict__ loss, float* __restrict__ dq) {
  extern __shared__ float sm[];
  float* logit = sm;                         // [64][LP]
  float* stage = sm + RB * LP;               // phase 1: As [64][36] | Bs [256][36]; phase 3: Ks [32][D16 + 16]
  float* pos = stage + (RB + 256) * DSP;     // [64] positive logits / T
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, i0 = blockIdx.x * RB;
  const int rows = min(RB, P - i0);
  const float* qb = q + ((int64_t)b * P + i0) * D;
  const float* kb = k + (int64_t)b * P * D;

  f32x4 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  gemm_nt_tile(qb, D, rows, kb, D, P, D, stage, stage + RB * DSP, acc);

  // logits / T into LDS; diagonal -> -10 / T, positive kept aside
  const int nt = (P + 15) >> 4;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + kq * 4 + r, col = t * 16 + m16;
        float v = acc[t][r] * invT;
        if (col == i0 + row) {
          pos[row] = v;
          v = -10.f * invT;
        }
        logit[row * LP + col] = col < P ? v : -3.0e38f;
      }
    }
  __syncthreads();

  // row-wise softmax cross-entropy over [pos, negatives]: wave w owns rows 16w.., a row is spread over the 64 lanes
  for (int rr = 0; rr < 16; ++rr) {
    const int row = wave * 16 + rr;
    if (row >= rows) break;     // uniform per wave
    float* lr = logit + row * LP;
    const float p0 = pos[row];
    float mx = p0;
    for (int j = lane; j < P; j += 64) mx = fmaxf(mx, lr[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float se = 0.f;
    for (int j = lane; j < P; j += 64) {
      const float e = expf(lr[j] - mx);
      lr[j] = e;
      se += e;
    }
    se = wave_sum(se);
    const float e0 = expf(p0 - mx);
    se += e0;
    if (lane == 0 && loss) loss[(int64_t)b * P + i0 + row] = logf(se) + mx - p0;
    const float inv = 1.f / se;
    // weights of the gradient: p_j for the negatives, (p_0 - 1) on the diagonal (the -10 entry carries no gradient)
    for (int j = lane; j < P; j += 64) lr[j] = (j == i0 + row) ? (e0 * inv - 1.f) : lr[j] * inv;
  }
  if (!dq) return;
  for (int idx = tid; idx < RB * LP; idx += 256) {     // rows beyond the block / columns beyond P: zero weights
    const int row = idx / LP, col = idx - row * LP;
    if (row >= rows || col >= P) logit[idx] = 0.f;
  }
  __syncthreads();

  // dQ_blk = W K: wave w rows 16w.., D/16 tiles; K staged in slices of 32 keys with pitch = 16 (mod 32)
  const int dt = (D + 15) >> 4;
  const int KP = dt * 16 + 16;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < P; j0 += KS) {
    for (int idx = tid; idx < KS * dt * 4; idx += 256) {
      const int r = idx / (dt * 4), c4 = (idx - r * (dt * 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j0 + r < P) {
        const float* p = kb + (int64_t)(j0 + r) * D + c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (c4 + e < D) ? p[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(stage + r * KP + c4) = v;
    }
    __syncthreads();
    const float* wr = logit + (wave * 16 + m16) * LP + j0 + kq;
    const float* kr = stage + kq * KP + m16;
#pragma unroll
    for (int s = 0; s < KS / 4; ++s) {
      const float a = wr[s * 4];
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
        if (t < dt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, kr[s * 4 * KP + t * 16], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
  const float sc = invT * gscale;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
    if (t < dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + kq * 4 + r, col = t * 16 + m16;
        if (row < rows && col < D) dq[((int64_t)b * P + i0 + row) * D + col] = acc[t][r] * sc;
      }
    }
}

// y[R x O] = act(x[R x I] W[O x I]^T + bias): the Linear layers of PatchSampleF's MLP (networks.py:681-686) on the same tile routine
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          int R, int I, int O, int relu, float* __restrict__ y) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m16 = lane & 15, kq = lane >> 4;
  const int r0 = blockIdx.x * RB, o0 = blockIdx.y * 256;
  const int rows = min(RB, R - r0), cols = min(256, O - o0);
  f32x4 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  gemm_nt_tile(x + (int64_t)r0 * I, I, rows, w + (int64_t)o0 * I, I, cols, I, sm, sm + RB * DSP, acc);
  const int nt = (cols + 15) >> 4;
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + kq * 4 + r, col = t * 16 + m16;
        if (row < rows && col < cols) {
          float v = acc[t][r] + (bias ? bias[o0 + col] : 0.f);
          if (relu) v = fmaxf(v, 0.f);
          y[(int64_t)(r0 + row) * O + o0 + col] = v;
        }
      }
    }
}

// PatchSampleF's gather: out[(b * P + p) * C + c] = feat[b][c][ids[p]]  (feat.permute(0,2,3,1).flatten(1,2)[:, ids, :].flatten(0,1))
__global__ __launch_bounds__(256) void patch_sample_kernel(const float* __restrict__ feat, const int64_t* __restrict__ ids, int B, int C, int HW, int P,
                                                           float* __restrict__ out) {
  const int64_t n = (int64_t)B * P * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t bp = i / C;
    const int p = (int)(bp % P), b = (int)(bp / P);
    out[i] = feat[((int64_t)b * C + c) * HW + ids[p]];
  }
}

}  // namespace

bool vts_patchnce_mfma_ok(int P, int D) { return P <= 256 && D <= 256; }

int vts_patchnce_mfma(const float* q, const float* k, int B, int P, int D, float T, float gscale, float* loss, float* dq, hipStream_t st) {
  const int dt = (D + 15) / 16;
  const size_t stage = (size_t)((RB + 256) * DSP > KS * (dt * 16 + 16) ? (RB + 256) * DSP : KS * (dt * 16 + 16));
  const size_t sm = (RB * LP + stage + RB) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    // 160 KB of dynamic LDS is the whole CU: a refusal here must not pass silently (the launch below would then fail or be clipped)
    const hipError_t ea = hipFuncSetAttribute((const void*)patchnce_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (ea != hipSuccess) {
      vts_set_error("vts_patchnce (mfma): hipFuncSetAttribute(max dynamic LDS 160 KB) failed: %s", hipGetErrorString(ea));
      return VTS_ERR_LAUNCH;
    }
    attr = true;
  }
  hipLaunchKernelGGL(patchnce_mfma_kernel, dim3(cdiv(P, RB), B), dim3(256), sm, st, q, k, P, D, 1.f / T, gscale, loss, dq);
  vts_set_kernel("patchnce_mfma_kernel");
  VTS_CHECK_LAUNCH("vts_patchnce (mfma)");
  return VTS_OK;
}

extern "C" int vts_linear_rows(const float* x, const float* w, const float* bias, int R, int I, int O, int relu, float* y, void* stream) {
  VTS_CHECK_ARG(x && w && y && R >= 1 && I >= 1 && O >= 1, "vts_linear_rows: bad args");
  const size_t sm = (size_t)(RB + 256) * DSP * sizeof(float);
  hipLaunchKernelGGL(linear_rows_kernel, dim3(cdiv(R, RB), cdiv(O, 256)), dim3(256), sm, (hipStream_t)stream, x, w, bias, R, I, O, relu, y);
  VTS_CHECK_LAUNCH("vts_linear_rows");
  return VTS_OK;
}

extern "C" int vts_patch_sample(const float* feat, const int64_t* ids, int B, int C, int HW, int P, float* out, void* stream) {
  VTS_CHECK_ARG(feat && ids && out && B >= 1 && C >= 1 && HW >= 1 && P >= 1, "vts_patch_sample: bad args");
  const int64_t n = (int64_t)B * P * C;
  hipLaunchKernelGGL(patch_sample_kernel, dim3((unsigned)(cdiv64(n, 256) > 4096 ? 4096 : cdiv64(n, 256))), dim3(256), 0, (hipStream_t)stream, feat, ids, B, C,
                     HW, P, out);
  VTS_CHECK_LAUNCH("vts_patch_sample");
  return VTS_OK;
}
