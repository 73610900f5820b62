// GEMM-class 3x3 stride-1 convolution for wide layers (pix2pixHD GlobalGenerator: 9 ResnetBlocks at 1024
// channels, reference models/networks.py:1952-1980 / ResnetBlock :1267-1324; AI ~ 200 flop/B: the MFMA-bound
// workload of SURVEY.md §8d) on the fp32 MFMA path (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   out[n, co, y, x] = bias[co] + sum_{ci, ky, kx} in[n, ci, y + ky, x + kx] * wt[(ci * 9 + ky * 3 + kx) * Cout + co]
//
// `in` is the PRE-PADDED input [N, Cin, H + 2, W + 2] (vts_pad_affine materialises reflection / zero padding
// together with the pending normalise + activate, so this kernel reads an identity operand and never handles a
// border); `wt` is the tap-major packed weight of vts_w3x3_pack.  The adjoint w.r.t. the input is the same
// operator on the zero-padded output gradient with the flipped / transposed packing.
//
// GEMM view: M = output channels, N = pixels, K = Cin x 9 taps.  A workgroup (4 waves) owns 128 channels x
// (4 rows x 32 columns); a wave owns 64 x 64 as 2 x 2 MFMA tiles of 32 x 32 (64 accumulator registers).
// Per chunk of 8 input channels the 6 x 34 patch and the 8 x 9 x 128 weight slice are staged in LDS (44 KB);
// the next chunk's global loads (16-byte buffer loads, hardware bounds check) are in flight during the 144
// MFMAs of the current one.  LDS reads are conflict-free: an A fragment reads 32 consecutive output channels,
// a B fragment 32 consecutive pixels of one row, the two k-halves of a fragment are separate lane groups.
#include "vts_internal.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TCO = 128, TY = 4, TX = 32, CK = 8;
constexpr unsigned RSRC_FLAGS = 0x00020000;

// One kernel serves the three 3x3 operators of the wide layers through a tap table:
//   out[n, co, oy0 + os*y, ox0 + os*x] = bias[co] + sum_t sum_ci in[n, ci, S*y + dy_t, S*x + dx_t] * wt[(ci*9 + w_t)*Cout + co]
//   stride-1 conv        S = 1, os = 1, 9 taps (dy, dx) = (ky, kx)        input padded by 1 (any padding mode)
//   stride-2 conv        S = 2, os = 1, 9 taps                            input zero-padded by 1
//   ConvTranspose s2     S = 1, os = 2, one launch per output parity phase with its 1 / 2 / 2 / 4 taps, input
//                        zero-padded by 1 at the bottom / right (also the input adjoint of the stride-2 conv)
struct WideK {
  const float *in, *wt, *bias;
  float* out;
  int N, Cin, Cout, H, W;        // H x W: the (phase) grid of output pixels this launch computes
  int IPH, IPW;                  // padded input extent
  int OH, OW, os, oy0, ox0;      // full output extent, output stride and phase offset
  int ntaps;
  signed char dy[16], dx[16], wt_tap[16];   // (the 3x3 kernels use <= 9; the 4x4 flat entry up to 16)
  int KS, cps;   // k-split for small grids: blockIdx.z = n + N * slice, cps input-channel chunks per slice
  float* part;   // [KS][N][Cout][OH][OW] raw partial sums (KS > 1), reduced in slice order by wide_reduce_kernel
  // epilogue of the frozen VGG stacks (tiled direct kernels only).  1: store max(acc + bias, 0) -- the ReLU'd output straight into the next
  // layer's padded input (vts_conv3x3_wide_relu_pad).  2: store (acc + ep_add) where ep_mask > 0, else 0 -- the input adjoint's output with
  // the tap gradient added and the ReLU mask of the layer in front applied, straight into the next adjoint's padded input
  // (vts_conv3x3_wide_mask_pad); ep_add (optional) and ep_mask have the output's layout
  int ep_mode;
  const float* ep_add;
  const float* ep_mask;
};

// K: kernel extent (3 | 4: taps of the packed weight = K * K), CKT: input channels per chunk, LIVE: most taps one launch uses
// (K * K, or 4 for the parity phases of the transposed stride-2 layers, which then stage 16 channels per chunk: with 1-4 taps
// a chunk of 4-8 channels is too few MFMAs per barrier)
// MS: how the four waves split the workgroup's tile -- 2: two 64-channel halves x two row pairs (128 channels x 4 rows, the default);
// 1 (round 3, layers of <= 64 output channels: the VGG stacks' first block and its input adjoint): ONE 64-channel group x four row
// pairs (64 channels x 8 rows) -- with the 128-channel tile half of every MFMA of such a layer multiplied zero rows (59 TFLOP/s on
// 64 -> 64 at 1024 x 1024 against 124-131 TFLOP/s on the 128..512-channel layers)
template <int S, int K, int CKT, int LIVE = K * K, int MS = 2>
__device__ __forceinline__ void wide_body(const WideK& p) {
  constexpr int CK = CKT, TWT = K * K;
  constexpr int TCO = 64 * MS, TY = 8 / MS;            // (shadow the file-level tile constants)
  constexpr int QPR = TCO / 4, QSH = MS == 2 ? 5 : 4;  // weight quads per (channel, tap) row and its log2
  constexpr int PR = S * (TY - 1) + K, PC = S * (TX - 1) + K, PCP = (PC + 3) / 4 * 4;
  constexpr int PATCH_FLOATS = CK * PR * PCP;
  constexpr int W_FLOATS = CK * LIVE * TCO;
  constexpr int PQ_ROW = PCP / 4;
  constexpr int NPQ = (CK * PR * PQ_ROW + 255) / 256;
  constexpr int NWQ = (W_FLOATS / 4 + 255) / 256;       // 9 (8 for K = 4) weight quads per thread (fewer taps: fewer are live)
  __shared__ __attribute__((aligned(16))) float lds[PATCH_FLOATS + W_FLOATS];
  float* lds_p = lds;
  float* lds_w = lds + PATCH_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = MS == 2 ? (wave & 1) : 0, wpx = MS == 2 ? (wave >> 1) : wave;
  const int tiles_x = (p.W + TX - 1) / TX;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * TX, y0 = ty * TY, co0 = blockIdx.y * TCO, n = blockIdx.z % p.N, ks = blockIdx.z / p.N;
  const int PW = p.IPW, plane = p.IPH * p.IPW;
  const int ntaps = p.ntaps;

  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (int64_t)n * p.Cin * plane, 0, p.Cin * plane * 4, RSRC_FLAGS);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.Cin * TWT * p.Cout * 4, RSRC_FLAGS);

  // staging items of this thread (chunk independent): patch quads (ci, row, quad), weight quads ((ci, t), co quad)
  int pvoff[NPQ], ploff[NPQ];
#pragma unroll
  for (int e = 0; e < NPQ; ++e) {
    const int q = min(tid + e * 256, CK * PR * PQ_ROW - 1);
    const int row = q / PQ_ROW, cq = q - row * PQ_ROW;      // row = ci * PR + r
    const int ci = row / PR, r = row - ci * PR;
    pvoff[e] = (ci * plane + (S * y0 + r) * PW + S * x0 + 4 * cq) * 4;
    ploff[e] = row * PCP + 4 * cq;
  }
  int wvoff[NWQ], wloff[NWQ];
#pragma unroll
  for (int e = 0; e < NWQ; ++e) {
    const int q = tid + e * 256;
    const int row = q >> QSH, cq = q & (QPR - 1);           // row = ci * ntaps + t
    const int ci = row / ntaps, t = row - ci * ntaps;
    const bool live = row < CK * ntaps;
    wvoff[e] = live ? ((ci * TWT + p.wt_tap[live ? t : 0]) * p.Cout + co0 + 4 * cq) * 4 : 0x7ffffff0;   // dead rows: out of range -> 0
    wloff[e] = (live ? row : 0) * TCO + 4 * cq;
  }
  // Pixels past the grid and channels past Cout only ever feed outputs that are never stored; channels past Cin read
  // 0 through the bounds check.

  u32x4 pq[NPQ], wq[NWQ];
  auto load_chunk = [&](int c0) {
    const int pbase = c0 * plane * 4, wbase = c0 * TWT * p.Cout * 4;
#pragma unroll
    for (int e = 0; e < NPQ; ++e) pq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, pvoff[e] + pbase, 0, 0);
#pragma unroll
    for (int e = 0; e < NWQ; ++e)
      if (e * (256 / QPR) < CK * ntaps) wq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[e] + wbase, 0, 0);   // uniform
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < NPQ; ++e) *reinterpret_cast<u32x4*>(lds_p + ploff[e]) = pq[e];
#pragma unroll
    for (int e = 0; e < NWQ; ++e)
      if (e * (256 / QPR) < CK * ntaps && ((tid + e * 256) >> QSH) < CK * ntaps) *reinterpret_cast<u32x4*>(lds_w + wloff[e]) = wq[e];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* a_base = lds_w + kh * ntaps * TCO + wco * 64 + l32;
  const float* b_base = lds_p + kh * PR * PCP + (S * wpx * 2) * PCP + S * l32;

  const int nchunks_all = (p.Cin + CK - 1) / CK;
  const int cbeg = ks * p.cps, nchunks = min(nchunks_all, cbeg + p.cps);
  load_chunk(cbeg * CK);
  store_chunk();
  __syncthreads();
  for (int c = cbeg; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) load_chunk((c + 1) * CK);
    for (int t = 0; t < ntaps; ++t) {
      const int boff = p.dy[t] * PCP + p.dx[t];
#pragma unroll
      for (int kc = 0; kc < CK / 2; ++kc) {
        const float a0 = a_base[(kc * 2 * ntaps + t) * TCO], a1 = a_base[(kc * 2 * ntaps + t) * TCO + 32];
        const float b0 = b_base[kc * 2 * PR * PCP + boff], b1 = b_base[kc * 2 * PR * PCP + S * PCP + boff];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  // epilogue: C layout of a 32x32 tile: column (pixel) = lane % 32, row (channel) = (r / 4) * 8 + (lane / 32) * 4 + r % 4
  const int x = x0 + l32;
  const int64_t oplane = (int64_t)p.OH * p.OW;
  float* ob = p.part ? p.part + ((int64_t)ks * p.N + n) * p.Cout * oplane : p.out + (int64_t)n * p.Cout * oplane;
  const bool add_bias = p.bias && !p.part;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = y0 + wpx * 2 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 64 + i * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
        if (co < p.Cout && y < p.H && x < p.W) {
          float v = acc[i][j][r] + (add_bias ? p.bias[co] : 0.f);
          const int64_t o = co * oplane + (int64_t)(p.oy0 + p.os * y) * p.OW + p.ox0 + p.os * x;
          if (p.ep_mode == 1) v = fmaxf(v, 0.f);
          if (p.ep_mode == 2) {
            const int64_t e = (int64_t)n * p.Cout * oplane + o;
            v = p.ep_mask[e] > 0.f ? v + (p.ep_add ? p.ep_add[e] : 0.f) : 0.f;
          }
          ob[o] = v;
          if (p.ep_mode) {      // padded output: the tiles on the rim of the map also store the zero border next to them
            float* q = ob + o;
            const bool xl = x == 0, xr = x == p.W - 1;
            if (xl) q[-1] = 0.f;
            if (xr) q[1] = 0.f;
            if (y == 0) {
              q[-p.OW] = 0.f;
              if (xl) q[-p.OW - 1] = 0.f;
              if (xr) q[-p.OW + 1] = 0.f;
            }
            if (y == p.H - 1) {
              q[p.OW] = 0.f;
              if (xl) q[p.OW - 1] = 0.f;
              if (xr) q[p.OW + 1] = 0.f;
            }
          }
        }
      }
    }
}

template <int S>
__global__ __launch_bounds__(256) void conv3x3_wide_kernel(const WideK p) { wide_body<S, 3, 8>(p); }
// <= 64 output channels: 64 channels x (8 rows x 32 columns) per workgroup (MS = 1)
__global__ __launch_bounds__(256) void conv3x3_wide64_kernel(const WideK p) { wide_body<1, 3, 8, 9, 1>(p); }

// the same tiling for 4 x 4 kernels (the ndf = 64 PatchGAN discriminators of pix2pixHD on full-size images): 16-tap packed
// weights, 4 input channels per chunk (32 KB of weights + a 7 x 36 / 10 x 68 patch per channel in LDS)
template <int S>
__global__ __launch_bounds__(256) void conv4x4_wide_kernel(const WideK p) { wide_body<S, 4, 4>(p); }

// parity-phase launches (<= 4 taps, stride 1 on the input) of the transposed stride-2 layers
template <int K>
__global__ __launch_bounds__(256) void conv_wide_phase_kernel(const WideK p) { wide_body<1, K, 16, 4>(p); }

// ---- row-run variant of the 3 x 3 stride-1 kernel ---------------------------------------------------------------------
// A 4 x 32 pixel tile wastes a third of its columns on maps whose width is just above a multiple of 32 -- exactly the shape of the
// input adjoint of a reflection-padded ResnetBlock (reference models/networks.py:1267-1324: 64 x 64 features, 66 x 66 padded
// gradient).  Here the pixel dimension of the GEMM is a RUN of 128 consecutive pixels of the flattened image (35 instead of 51 tiles
// per 66 x 66 image); the staged operand is the <= 5 full-width rows the run touches, a lane finds its two pixels through plane
// offsets.  The staged rows (worst case 127 / W + 4 per channel) must fit the patch buffer: widths 64 .. 94.  (A 4352-float patch
// would admit 128 .. 134 wide maps -- the 66 x 130 gradient of a 2048 x 1024 image -- but 54,272 B of LDS no longer lets three
// workgroups share a CU: measured 2090 us against 1922 us for the tiled kernel.)
constexpr int RR_PATCH_FLOATS = 3840;
static inline int rr_rows(int W, int K) { return 127 / W + 1 + K; }

// K x K taps (3: 8 input channels per chunk, 4: 4 -- the PatchGAN layers at 130 x 130 / 66 x 66), stride 1
template <int K, int CKT>
__device__ __forceinline__ void rowrun_body(const WideK& p) {
  constexpr int CK = CKT, TWT = K * K;
  constexpr int PATCH_FLOATS = RR_PATCH_FLOATS;
  constexpr int W_FLOATS = CK * TWT * TCO;
  constexpr int NPQ = (PATCH_FLOATS / 4 + 255) / 256;
  constexpr int NWQ = W_FLOATS / 4 / 256;
  __shared__ __attribute__((aligned(16))) float lds[PATCH_FLOATS + W_FLOATS];
  float* lds_p = lds;
  float* lds_w = lds + PATCH_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wpx = wave >> 1;
  const int HW = p.H * p.W;
  const int g0 = blockIdx.x * 128, co0 = blockIdx.y * TCO, n = blockIdx.z % p.N, ks = blockIdx.z / p.N;
  const int PW = p.IPW, plane = p.IPH * p.IPW;
  const int PCP = (PW + 3) & ~3, PQ = PCP >> 2;
  const int ymin = g0 / p.W, ymax = min(g0 + 127, HW - 1) / p.W;
  const int nr = ymax - ymin + K;                        // rows of the padded input this run touches
  const int nrm = 127 / p.W + 1 + K;                     // rows allocated per channel (worst case)
  const int TQ = CK * nr * PQ;
  const int ntaps = p.ntaps;

  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (int64_t)n * p.Cin * plane, 0, p.Cin * plane * 4, RSRC_FLAGS);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.Cin * TWT * p.Cout * 4, RSRC_FLAGS);

  int pvoff[NPQ], ploff[NPQ];
#pragma unroll
  for (int e = 0; e < NPQ; ++e) {
    const int q = min(tid + e * 256, TQ - 1);
    const int row = q / PQ, cq = q - row * PQ;            // row = ci * nr + r
    const int ci = row / nr, r = row - ci * nr;
    pvoff[e] = (ci * plane + (ymin + r) * PW + 4 * cq) * 4;
    ploff[e] = (ci * nrm + r) * PCP + 4 * cq;
  }
  int wvoff[NWQ], wloff[NWQ];
#pragma unroll
  for (int e = 0; e < NWQ; ++e) {
    const int q = tid + e * 256;
    const int row = q >> 5, cq = q & 31;                    // row = ci * ntaps + t
    const int ci = row / ntaps, t = row - ci * ntaps;
    const bool live = row < CK * ntaps;
    wvoff[e] = live ? ((ci * TWT + p.wt_tap[live ? t : 0]) * p.Cout + co0 + 4 * cq) * 4 : 0x7ffffff0;
    wloff[e] = (live ? row : 0) * TCO + 4 * cq;
  }
  u32x4 pq[NPQ], wq[NWQ];
  auto load_chunk = [&](int c0) {
    const int pbase = c0 * plane * 4, wbase = c0 * TWT * p.Cout * 4;
#pragma unroll
    for (int e = 0; e < NPQ; ++e)
      if (e * 256 < TQ) pq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, pvoff[e] + pbase, 0, 0);
#pragma unroll
    for (int e = 0; e < NWQ; ++e)
      if (e * 8 < CK * ntaps) wq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[e] + wbase, 0, 0);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < NPQ; ++e)
      if (e * 256 < TQ) *reinterpret_cast<u32x4*>(lds_p + ploff[e]) = pq[e];
#pragma unroll
    for (int e = 0; e < NWQ; ++e)
      if (e * 8 < CK * ntaps && ((tid + e * 256) >> 5) < CK * ntaps) *reinterpret_cast<u32x4*>(lds_w + wloff[e]) = wq[e];
  };

  int poff[2], gpx[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int g = g0 + wpx * 64 + j * 32 + l32;
    const int y = g / p.W, x = g - y * p.W;
    gpx[j] = g < HW ? g : -1;
    poff[j] = g < HW ? (y - ymin) * PCP + x : 0;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int chs = nrm * PCP;
  const float* a_base = lds_w + kh * ntaps * TCO + wco * 64 + l32;
  const float* b0_base = lds_p + kh * chs + poff[0];
  const float* b1_base = lds_p + kh * chs + poff[1];

  const int nchunks_all = (p.Cin + CK - 1) / CK;
  const int cbeg = ks * p.cps, nchunks = min(nchunks_all, cbeg + p.cps);
  load_chunk(cbeg * CK);
  store_chunk();
  __syncthreads();
  for (int c = cbeg; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) load_chunk((c + 1) * CK);
    for (int t = 0; t < ntaps; ++t) {
      const int boff = p.dy[t] * PCP + p.dx[t];
#pragma unroll
      for (int kc = 0; kc < CK / 2; ++kc) {
        const float a0 = a_base[(kc * 2 * ntaps + t) * TCO], a1 = a_base[(kc * 2 * ntaps + t) * TCO + 32];
        const float b0 = b0_base[kc * 2 * chs + boff], b1 = b1_base[kc * 2 * chs + boff];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  float* ob = p.part ? p.part + ((int64_t)ks * p.N + n) * p.Cout * HW : p.out + (int64_t)n * p.Cout * HW;
  const bool add_bias = p.bias && !p.part;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (gpx[j] < 0) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 64 + i * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
        if (co < p.Cout) ob[(int64_t)co * HW + gpx[j]] = acc[i][j][r] + (add_bias ? p.bias[co] : 0.f);
      }
    }
}

__global__ __launch_bounds__(256) void conv3x3_rowrun_kernel(const WideK p) { rowrun_body<3, 8>(p); }
__global__ __launch_bounds__(256) void conv4x4_rowrun_kernel(const WideK p) { rowrun_body<4, 4>(p); }

__global__ __launch_bounds__(256) void wide_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, int KS,
                                                           int64_t per_slice, int HW, int Cout, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_slice) return;
  float v = bias ? bias[(i / HW) % Cout] : 0.f;
  for (int k = 0; k < KS; ++k) v += part[k * per_slice + i];
  out[i] = v;
}

// ---- flattened small-map variant -------------------------------------------------------------------------------
// Maps of <= 128 output pixels (pix2pixHD trained patch-wise: the 1024-channel ResnetBlocks then see 2 x 2 maps,
// reference models/pix2pixHD_model.py:587-722 with data/patchskit_dataset.py:277-333) would leave the 4 x 32 pixel
// tile of conv3x3_wide_kernel almost empty.  Here the N dimension of the GEMM is the FLATTENED (image, y, x) index
// of IPT whole images (IPT * H * W <= 128), the staged operand is the IPT x CK complete padded planes, and a lane
// finds its pixel through a precomputed plane offset; everything else (tap table, packed weights, MFMA tiling,
// k-split) is the same.  The work per layer is then dominated by reading the weights once, so the channel loop
// is split until the grid fills the GPU.  TW = taps of the packed weight (9, or 16 for 4 x 4 kernels).
struct FlatK {
  const float *in, *wt, *bias;
  float* out;
  int N, Cin, Cout, H, W;        // H x W: the (phase) grid of output pixels
  int IPH, IPW, S;               // padded input extent, input stride
  int OH, OW, os, oy0, ox0;      // full output extent, output stride and phase offset
  int TW, ntaps, Cw;             // taps of the packed weight, taps of this launch, weight row pitch (>= Cout, multiple of 4)
  signed char dy[16], dx[16], wt_tap[16];
  int IPT, slot;                 // images per pixel tile, LDS floats per (channel, image) plane (multiple of 4)
  int KS, cps;
  float* part;                   // [KS][N][Cout][H][W] raw partial sums of the phase grid (KS > 1)
};

constexpr int FLAT_PATCH_CAP = 8192;

template <int FCK, int MAXT>
__global__ __launch_bounds__(256) void conv_flat_kernel(const FlatK p) {
  constexpr int W_FLOATS = FCK * MAXT * TCO;
  constexpr int NPQ = FLAT_PATCH_CAP / 4 / 256;
  constexpr int NWQ = W_FLOATS / 4 / 256;
  __shared__ __attribute__((aligned(16))) float lds[FLAT_PATCH_CAP + W_FLOATS];
  float* lds_p = lds;
  float* lds_w = lds + FLAT_PATCH_CAP;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wpx = wave >> 1;
  const int n0 = blockIdx.x * p.IPT, co0 = blockIdx.y * TCO, ks = blockIdx.z;
  const int PW = p.IPW, plane = p.IPH * p.IPW, HW = p.H * p.W;
  const int ntaps = p.ntaps, slot = p.slot, sq = slot >> 2, chs = p.IPT * slot;
  const int TQ = FCK * p.IPT * sq;                     // patch quads per chunk

  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.N * p.Cin * plane * 4, RSRC_FLAGS);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.Cin * p.TW * p.Cw * 4, RSRC_FLAGS);

  int pvoff[NPQ], ploff[NPQ], pci[NPQ];
#pragma unroll
  for (int e = 0; e < NPQ; ++e) {
    const int q = min(tid + e * 256, TQ - 1);
    const int ci = q / (p.IPT * sq), r = q - ci * (p.IPT * sq);
    const int img = r / sq, pqi = r - img * sq;
    pci[e] = ci;
    pvoff[e] = n0 + img < p.N ? (((n0 + img) * p.Cin + ci) * plane + 4 * pqi) * 4 : 0x7ffffff0;
    ploff[e] = (ci * p.IPT + img) * slot + 4 * pqi;
  }
  int wvoff[NWQ], wloff[NWQ];
#pragma unroll
  for (int e = 0; e < NWQ; ++e) {
    const int q = tid + e * 256;
    const int row = q >> 5, cq = q & 31;                    // row = ci * ntaps + t
    const int ci = row / ntaps, t = row - ci * ntaps;
    const bool live = row < FCK * ntaps;
    wvoff[e] = live ? ((ci * p.TW + p.wt_tap[live ? t : 0]) * p.Cw + co0 + 4 * cq) * 4 : 0x7ffffff0;
    wloff[e] = (live ? row : 0) * TCO + 4 * cq;
  }

  u32x4 pq[NPQ], wq[NWQ];
  auto load_chunk = [&](int c0) {
    const int pbase = c0 * plane * 4, wbase = c0 * p.TW * p.Cw * 4;
#pragma unroll
    for (int e = 0; e < NPQ; ++e)
      if (e * 256 < TQ) pq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, c0 + pci[e] < p.Cin ? pvoff[e] + pbase : 0x7ffffff0, 0, 0);
#pragma unroll
    for (int e = 0; e < NWQ; ++e)
      if (e * 8 < FCK * ntaps) wq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[e] + wbase, 0, 0);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < NPQ; ++e)
      if (e * 256 < TQ) *reinterpret_cast<u32x4*>(lds_p + ploff[e]) = pq[e];
#pragma unroll
    for (int e = 0; e < NWQ; ++e)
      if (e * 8 < FCK * ntaps && ((tid + e * 256) >> 5) < FCK * ntaps) *reinterpret_cast<u32x4*>(lds_w + wloff[e]) = wq[e];
  };

  // this lane's two pixels (B fragments j = 0, 1): plane offset of tap (0, 0) and the output position
  int pixoff[2], oimg[2], opos[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int px = wpx * 64 + j * 32 + l32;
    const int img = px / HW, r = px - img * HW;
    const int y = r / p.W, x = r - y * p.W;
    const bool ok = img < p.IPT && n0 + img < p.N;
    pixoff[j] = ok ? img * slot + p.S * y * PW + p.S * x : 0;
    oimg[j] = ok ? n0 + img : -1;
    opos[j] = p.part ? r : (p.oy0 + p.os * y) * p.OW + p.ox0 + p.os * x;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* a_base = lds_w + kh * ntaps * TCO + wco * 64 + l32;
  const float* b0_base = lds_p + kh * chs + pixoff[0];
  const float* b1_base = lds_p + kh * chs + pixoff[1];

  const int nchunks_all = (p.Cin + FCK - 1) / FCK;
  const int cbeg = ks * p.cps, nchunks = min(nchunks_all, cbeg + p.cps);
  load_chunk(cbeg * FCK);
  store_chunk();
  __syncthreads();
  for (int c = cbeg; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) load_chunk((c + 1) * FCK);
    for (int t = 0; t < ntaps; ++t) {
      const int boff = p.dy[t] * PW + p.dx[t];
#pragma unroll
      for (int kc = 0; kc < FCK / 2; ++kc) {
        const float a0 = a_base[(kc * 2 * ntaps + t) * TCO], a1 = a_base[(kc * 2 * ntaps + t) * TCO + 32];
        const float b0 = b0_base[kc * 2 * chs + boff], b1 = b1_base[kc * 2 * chs + boff];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
  }

  const int64_t oplane = p.part ? HW : (int64_t)p.OH * p.OW;
  float* ob = p.part ? p.part + (int64_t)ks * p.N * p.Cout * oplane : p.out;
  const bool add_bias = p.bias && !p.part;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (oimg[j] < 0) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 64 + i * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
        if (co < p.Cout) ob[((int64_t)oimg[j] * p.Cout + co) * oplane + opos[j]] = acc[i][j][r] + (add_bias ? p.bias[co] : 0.f);
      }
    }
}

// out[n, co, oy0 + os*y, ox0 + os*x] = bias[co] + sum_k part[k][n][co][y][x]
__global__ __launch_bounds__(256) void flat_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, int KS,
                                                           int64_t per_slice, int H, int W, int Cout, int OH, int OW, int os, int oy0,
                                                           int ox0, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_slice) return;
  const int HW = H * W;
  const int64_t nc = i / HW;
  const int r = (int)(i - nc * HW), y = r / W, x = r - y * W;
  float v = bias ? bias[nc % Cout] : 0.f;
  for (int k = 0; k < KS; ++k) v += part[k * per_slice + i];
  out[nc * OH * OW + (int64_t)(oy0 + os * y) * OW + ox0 + os * x] = v;
}

// wt[(a * T + t) * Bp + b] = b < B ? w[a * sa + b * sb + (flip ? T - 1 - t : t)] : 0   (a < A: the operator's input channel, b < B: its
// output channel, Bp = B rounded up to 4).  A workgroup transposes a tile of 32 output channels x 64 (a, t) rows through LDS so that
// both the gather from the parameter tensor and the packed store run along their contiguous index where there is one.
__global__ __launch_bounds__(256) void wtap_pack_kernel(const float* __restrict__ w, int A, int B, int Bp, int64_t sa, int64_t sb, int T,
                                                         int flip, int rows_contig, float* __restrict__ wt) {
  __shared__ float tile[64][33];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int b0 = blockIdx.y * 32;
  const int64_t R = (int64_t)A * T;
  // gather: rows_contig: consecutive lanes walk the (a, t) rows (contiguous in w when sa == T), else the b index
  for (int e = threadIdx.x; e < 64 * 32; e += 256) {
    const int rr = rows_contig ? e & 63 : e >> 5, bb = rows_contig ? e >> 6 : e & 31;
    const int64_t r = r0 + rr;
    const int b = b0 + bb;
    float v = 0.f;
    if (r < R && b < B) {
      const int t = (int)(r % T);
      const int64_t a = r / T;
      v = w[a * sa + b * sb + (flip ? T - 1 - t : t)];
    }
    tile[rr][bb] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 32; e += 256) {
    const int rr = e >> 5, bb = e & 31;
    if (r0 + rr < R && b0 + bb < Bp) wt[(r0 + rr) * Bp + b0 + bb] = tile[rr][bb];
  }
}

}  // namespace

static int wtap_pack(const char* who, const float* w, int A, int B, int64_t sa, int64_t sb, int T, int flip, float* wt, void* stream) {
  VTS_CHECK_ARG(w && wt && A >= 1 && B >= 1, "%s: bad args", who);
  const int Bp = (B + 3) & ~3;
  const dim3 grid((unsigned)cdiv64((int64_t)A * T, 64), (unsigned)cdiv(Bp, 32));
  hipLaunchKernelGGL(wtap_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, A, B, Bp, sa, sb, T, flip, sa == T ? 1 : 0, wt);
  VTS_CHECK_LAUNCH(who);
  return VTS_OK;
}

extern "C" int vts_w3x3_pack(const float* w, int A, int B, int64_t sa, int64_t sb, int flip, float* wt, void* stream) {
  return wtap_pack("vts_w3x3_pack", w, A, B, sa, sb, 9, flip, wt, stream);
}

extern "C" int vts_w4x4_pack(const float* w, int A, int B, int64_t sa, int64_t sb, int flip, float* wt, void* stream) {
  return wtap_pack("vts_w4x4_pack", w, A, B, sa, sb, 16, flip, wt, stream);
}

static int wide_plan(int N, int Cin, int Cout, int H, int W, int* cps, int ck = CK) {
  const int wgs = cdiv(W, TX) * cdiv(H, TY) * cdiv(Cout, TCO) * N;
  const int nchunks = cdiv(Cin, ck);
  int KS = 1;
  if (wgs < 256) {           // too few tiles for 256 CUs: split the input-channel loop
    KS = 768 / wgs;
    if (KS > nchunks / 4) KS = nchunks / 4;
    if (KS < 1) KS = 1;
  }
  static const int force = vts_tune("VTS_WIDE_KS", 0);   // measurement override
  if (force > 0) KS = force < nchunks ? force : nchunks;
  *cps = cdiv(nchunks, KS);
  return cdiv(nchunks, *cps);
}

// flattened small-map variant: images per pixel tile, k-split
constexpr int FLAT_MAX_PIXELS = 128;
static bool flat_ok(int H, int W, int plane, int fck) { return H * W <= FLAT_MAX_PIXELS && fck * ((plane + 3) / 4 * 4) <= FLAT_PATCH_CAP; }
static int flat_plan(int N, int Cin, int Cout, int H, int W, int plane, int fck, int* ipt, int* cps) {
  const int slot = (plane + 3) / 4 * 4;
  int IPT = FLAT_MAX_PIXELS / (H * W);
  if (IPT > FLAT_PATCH_CAP / (fck * slot)) IPT = FLAT_PATCH_CAP / (fck * slot);
  if (IPT > N) IPT = N;
  *ipt = IPT;
  const int wgs = cdiv(N, IPT) * cdiv(Cout, TCO);
  const int nchunks = cdiv(Cin, fck);
  int KS = 1;
  if (wgs < 384) {
    KS = 512 / wgs;
    if (KS > nchunks / 2) KS = nchunks / 2;
    if (KS < 1) KS = 1;
  }
  *cps = cdiv(nchunks, KS);
  return cdiv(nchunks, *cps);
}

extern "C" int64_t vts_conv3x3_wide_ws_floats(int N, int Cin, int Cout, int H, int W) {
  // H x W is the grid of output pixels of ONE launch (stride-1 / stride-2 convolution: the output extent; transposed
  // stride 2: the input extent = one parity phase); the bound covers the three operators' padded planes
  int cps, ipt;
  int64_t need = 0;
  const int planes[3] = {(H + 2) * (W + 2), (2 * H + 2) * (2 * W + 2), (H + 1) * (W + 1)};
  for (int v = 0; v < 3; ++v)
    if (flat_ok(H, W, planes[v], CK)) {
      const int KS = flat_plan(N, Cin, Cout, H, W, planes[v], CK, &ipt, &cps);
      if (KS > 1 && (int64_t)KS * N * Cout * H * W > need) need = (int64_t)KS * N * Cout * H * W;
    }
  const int KS = wide_plan(N, Cin, Cout, H, W, &cps);
  if (KS > 1 && (int64_t)KS * N * Cout * H * W > need) need = (int64_t)KS * N * Cout * H * W;
  return need;
}

static int flat_launch(const WideK& k, int S, int TW, float* ws, int64_t ws_floats, hipStream_t st) {
  const int fck = TW == 16 ? 4 : CK;
  FlatK f{};
  f.in = k.in; f.wt = k.wt; f.bias = k.bias; f.out = k.out; f.N = k.N; f.Cin = k.Cin; f.Cout = k.Cout; f.H = k.H; f.W = k.W;
  f.IPH = k.IPH; f.IPW = k.IPW; f.S = S; f.OH = k.OH; f.OW = k.OW; f.os = k.os; f.oy0 = k.oy0; f.ox0 = k.ox0;
  f.TW = TW; f.ntaps = k.ntaps; f.Cw = (k.Cout + 3) & ~3;
  for (int t = 0; t < k.ntaps; ++t) { f.dy[t] = k.dy[t]; f.dx[t] = k.dx[t]; f.wt_tap[t] = k.wt_tap[t]; }
  const int plane = k.IPH * k.IPW;
  VTS_CHECK_ARG((int64_t)k.N * k.Cin * plane * 4 < (1ll << 31) && (int64_t)k.Cin * TW * k.Cout * 4 < (1ll << 31),
                "flat conv: operand exceeds the 2 GiB buffer range");
  f.slot = (plane + 3) / 4 * 4;
  int KS = flat_plan(k.N, k.Cin, k.Cout, k.H, k.W, plane, fck, &f.IPT, &f.cps);
  const int64_t per_slice = (int64_t)k.N * k.Cout * k.H * k.W;
  if (KS == 1 || !ws || ws_floats < KS * per_slice) { KS = 1; f.cps = cdiv(k.Cin, fck); }
  f.KS = KS; f.part = KS > 1 ? ws : nullptr;
  const dim3 grid(cdiv(k.N, f.IPT), cdiv(k.Cout, TCO), KS);
  if (TW == 16) hipLaunchKernelGGL((conv_flat_kernel<4, 16>), grid, dim3(256), 0, st, f);
  else hipLaunchKernelGGL((conv_flat_kernel<CK, 9>), grid, dim3(256), 0, st, f);
  vts_set_kernel(KS > 1 ? "conv_flat_kernel<%d, %d>+ksplit" : "conv_flat_kernel<%d, %d>", fck, TW);
  VTS_CHECK_LAUNCH("flat conv");
  if (KS > 1) {
    hipLaunchKernelGGL(flat_reduce_kernel, dim3((unsigned)cdiv64(per_slice, 256)), dim3(256), 0, st, ws, k.bias, KS, per_slice, k.H, k.W,
                       k.Cout, k.OH, k.OW, k.os, k.oy0, k.ox0, k.out);
    VTS_CHECK_LAUNCH("flat conv reduce");
  }
  return VTS_OK;
}

static int wide_launch(WideK& k, int S, float* ws, int64_t ws_floats, hipStream_t st, int K = 3) {
  const int ck = K == 4 ? 4 : CK, TW = K * K;
  VTS_CHECK_ARG((k.Cout & 3) == 0 || (K == 4 && flat_ok(k.H, k.W, k.IPH * k.IPW, ck)), "wide conv: Cout %d must be a multiple of 4 (16-byte weight rows)", k.Cout);
  VTS_CHECK_ARG((int64_t)k.Cin * k.IPH * k.IPW * 4 < (1ll << 31) && (int64_t)k.Cin * TW * k.Cout * 4 < (1ll << 31) && k.N <= 1024,
                "wide conv: operand exceeds the 2 GiB buffer range");
  if (flat_ok(k.H, k.W, k.IPH * k.IPW, ck)) return flat_launch(k, S, TW, ws, ws_floats, st);
  int cps;
  int KS = k.os == 1 ? wide_plan(k.N, k.Cin, k.Cout, k.H, k.W, &cps, ck) : 1;
  const int64_t per_slice = (int64_t)k.N * k.Cout * k.OH * k.OW;
  if (KS == 1 || !ws || ws_floats < KS * per_slice) { KS = 1; cps = cdiv(k.Cin, ck); }
  k.KS = KS; k.cps = cps; k.part = KS > 1 ? ws : nullptr;
  dim3 grid(cdiv(k.W, TX) * cdiv(k.H, TY), cdiv(k.Cout, TCO), k.N * KS);
  // maps whose 4 x 32 tiling wastes >= 10 %% more than runs of 128 flattened pixels (and whose rows fit the patch): the row-run kernel
  static const int no_rowrun = vts_tune_set("VTS_NO_ROWRUN") ? 1 : 0;
  const double eff_tile = (double)k.W * k.H / ((double)cdiv(k.W, TX) * TX * cdiv(k.H, TY) * TY);
  const double eff_run = (double)k.W * k.H / (128.0 * cdiv(k.W * k.H, 128));
  if (!no_rowrun && S == 1 && k.os == 1 && k.ntaps == TW && k.W >= 64 && ck * rr_rows(k.W, K) * ((k.IPW + 3) & ~3) <= RR_PATCH_FLOATS &&
      k.OH == k.H && k.OW == k.W && eff_run > 1.1 * eff_tile) {
    grid.x = cdiv(k.W * k.H, 128);
    if (K == 4) hipLaunchKernelGGL(conv4x4_rowrun_kernel, grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(conv3x3_rowrun_kernel, grid, dim3(256), 0, st, k);
    vts_set_kernel(KS > 1 ? "conv%dx%d_rowrun_kernel+ksplit" : "conv%dx%d_rowrun_kernel", K, K);
  } else if (k.os == 2 && k.ntaps <= 4 && S == 1 && KS == 1) {
    k.cps = cdiv(k.Cin, 16);
    if (K == 4) hipLaunchKernelGGL(conv_wide_phase_kernel<4>, grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(conv_wide_phase_kernel<3>, grid, dim3(256), 0, st, k);
    vts_set_kernel("conv_wide_phase_kernel<%d>", K);
  } else if (K == 4) {
    if (S == 1) hipLaunchKernelGGL(conv4x4_wide_kernel<1>, grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(conv4x4_wide_kernel<2>, grid, dim3(256), 0, st, k);
    vts_set_kernel(KS > 1 ? "conv4x4_wide_kernel<%d>+ksplit" : "conv4x4_wide_kernel<%d>", S);
  } else if (S == 1 && KS == 1 && k.os == 1 && k.Cout <= 64 && cdiv(k.W, TX) * cdiv(k.H, 8) * k.N >= 256 && !vts_tune_set("VTS_NO_WIDE64")) {
    grid = dim3(cdiv(k.W, TX) * cdiv(k.H, 8), 1, k.N);
    hipLaunchKernelGGL(conv3x3_wide64_kernel, grid, dim3(256), 0, st, k);
    vts_set_kernel("conv3x3_wide64_kernel");
  } else {
    if (S == 1) hipLaunchKernelGGL(conv3x3_wide_kernel<1>, grid, dim3(256), 0, st, k);
    else hipLaunchKernelGGL(conv3x3_wide_kernel<2>, grid, dim3(256), 0, st, k);
    vts_set_kernel(KS > 1 ? "conv3x3_wide_kernel<%d>+ksplit" : "conv3x3_wide_kernel<%d>", S);
  }
  VTS_CHECK_LAUNCH("wide conv");
  if (KS > 1) {
    hipLaunchKernelGGL(wide_reduce_kernel, dim3((unsigned)cdiv64(per_slice, 256)), dim3(256), 0, st, ws, k.bias, KS, per_slice,
                       k.OH * k.OW, k.Cout, k.out);
    VTS_CHECK_LAUNCH("wide conv reduce");
  }
  return VTS_OK;
}

static void full_taps(WideK& k) {
  k.ntaps = 9;
  for (int t = 0; t < 9; ++t) { k.dy[t] = (signed char)(t / 3); k.dx[t] = (signed char)(t % 3); k.wt_tap[t] = (signed char)t; }
}

extern "C" int vts_conv3x3_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int H, int W,
                                float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(in && wt && out && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, "vts_conv3x3_wide: bad args");
  WideK k{};
  k.in = in; k.wt = wt; k.bias = bias; k.out = out; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = H; k.W = W;
  k.IPH = H + 2; k.IPW = W + 2; k.OH = H; k.OW = W; k.os = 1; k.oy0 = 0; k.ox0 = 0;
  full_taps(k);
  return wide_launch(k, 1, ws, ws_floats, (hipStream_t)stream);
}

// The frozen VGG stacks of the perceptual terms (round 4): activations and their gradients live in the NEXT convolution's pre-padded layout
// [N, C, H + 2, W + 2] with a zero one-pixel border (stored by the tiles on the rim of the map, together with their outputs).
//   vts_conv3x3_wide_relu_pad   forward: out interior = max(conv + bias, 0).  The separate ReLU + padding pass between two convolutions
//                               (vts_pad_affine: a read and a write of every feature map) disappears; taps, pooling and the ReLU mask of
//                               the backward read the padded tensor (relu(z) > 0 <=> z > 0)
//   vts_conv3x3_wide_mask_pad   input adjoint: out interior = (conv + add) where mask > 0, else 0 -- `mask` the padded ReLU'd activation
//                               of the layer in front, `add` (optional, same layout) that layer's tap gradient: the ReLU-mask + padding
//                               pass between two adjoints (vts_relu_mask_pad) disappears
// Tiled direct launches only: VTS_ERR_UNSUPPORTED for shapes that take the flattened / k-split paths (the caller keeps the dense form there).
__global__ __launch_bounds__(256) void zero_border_kernel(float* __restrict__ buf, int PH, int PW, int pad) {
  float* o = buf + (int64_t)blockIdx.y * PH * PW;
  const int rows = 2 * pad * PW, side = 2 * pad * (PH - 2 * pad);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows + side; i += gridDim.x * 256) {
    int y, x;
    if (i < rows) {
      const int r = i / PW;
      y = r < pad ? r : PH - 2 * pad + r; x = i - r * PW;
    } else {
      const int j = i - rows, r = j / (2 * pad), c = j - r * 2 * pad;
      y = pad + r; x = c < pad ? c : PW - 2 * pad + c;
    }
    o[(int64_t)y * PW + x] = 0.f;
  }
}

extern "C" int vts_zero_border(float* buf, int64_t NC, int H, int W, int pad, void* stream) {
  VTS_CHECK_ARG(buf && NC >= 1 && NC <= 0x7fffffff && H >= 1 && W >= 1 && pad >= 1 && pad <= 8, "vts_zero_border: bad args");
  const int PH = H + 2 * pad, PW = W + 2 * pad, per = 2 * pad * (PW + H);
  for (int64_t c0 = 0; c0 < NC; c0 += 65535) {
    const int nc = (int)std::min<int64_t>(65535, NC - c0);
    hipLaunchKernelGGL(zero_border_kernel, dim3(std::min(cdiv(per, 256), 64), nc), dim3(256), 0, (hipStream_t)stream, buf + c0 * PH * PW, PH, PW, pad);
  }
  VTS_CHECK_LAUNCH("vts_zero_border");
  return VTS_OK;
}

static int wide_pad_launch(WideK& k, int N, int Cin, int Cout, int H, int W, hipStream_t st) {
  k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = H; k.W = W;
  k.IPH = H + 2; k.IPW = W + 2; k.OH = H + 2; k.OW = W + 2; k.os = 1; k.oy0 = 1; k.ox0 = 1;
  full_taps(k);
  int cps;
  if (flat_ok(H, W, k.IPH * k.IPW, CK) || wide_plan(N, Cin, Cout, H, W, &cps) > 1 || (Cout & 3)) return VTS_ERR_UNSUPPORTED;
  return wide_launch(k, 1, nullptr, 0, st);
}

extern "C" int vts_conv3x3_wide_relu_pad(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int H, int W,
                                         void* stream) {
  VTS_CHECK_ARG(in && wt && out && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, "vts_conv3x3_wide_relu_pad: bad args");
  WideK k{};
  k.in = in; k.wt = wt; k.bias = bias; k.out = out; k.ep_mode = 1;
  return wide_pad_launch(k, N, Cin, Cout, H, W, (hipStream_t)stream);
}

extern "C" int vts_conv3x3_wide_mask_pad(const float* in, const float* wt, float* out, int N, int Cin, int Cout, int H, int W, const float* add,
                                         const float* mask, void* stream) {
  VTS_CHECK_ARG(in && wt && out && mask && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, "vts_conv3x3_wide_mask_pad: bad args");
  WideK k{};
  k.in = in; k.wt = wt; k.bias = nullptr; k.out = out; k.ep_mode = 2; k.ep_add = add; k.ep_mask = mask;
  return wide_pad_launch(k, N, Cin, Cout, H, W, (hipStream_t)stream);
}

extern "C" int vts_conv3x3s2_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int OH, int OW,
                                  float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(in && wt && out && N >= 1 && Cin >= 1 && Cout >= 1 && OH >= 1 && OW >= 1, "vts_conv3x3s2_wide: bad args");
  WideK k{};
  k.in = in; k.wt = wt; k.bias = bias; k.out = out; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = OH; k.W = OW;
  k.IPH = 2 * OH + 2; k.IPW = 2 * OW + 2; k.OH = OH; k.OW = OW; k.os = 1; k.oy0 = 0; k.ox0 = 0;
  full_taps(k);
  return wide_launch(k, 2, flat_ok(OH, OW, k.IPH * k.IPW, CK) ? ws : nullptr, ws_floats, (hipStream_t)stream);
}

extern "C" int vts_tconv3x3s2_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int IH, int IW,
                                   float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(in && wt && out && N >= 1 && Cin >= 1 && Cout >= 1 && IH >= 1 && IW >= 1, "vts_tconv3x3s2_wide: bad args");
  // out[2i + py] = sum over (d, k) with k = py + 1 - 2d:  py = 0: (d 0, k 1);  py = 1: (d 0, k 2), (d 1, k 0); input index i + d
  static const int ND[2] = {1, 2}, D[2][2] = {{0, 0}, {0, 1}}, KK[2][2] = {{1, 0}, {2, 0}};
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      WideK k{};
      k.in = in; k.wt = wt; k.bias = bias; k.out = out; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = IH; k.W = IW;
      k.IPH = IH + 1; k.IPW = IW + 1; k.OH = 2 * IH; k.OW = 2 * IW; k.os = 2; k.oy0 = py; k.ox0 = px;
      k.ntaps = 0;
      for (int a = 0; a < ND[py]; ++a)
        for (int b = 0; b < ND[px]; ++b) {
          k.dy[k.ntaps] = (signed char)D[py][a]; k.dx[k.ntaps] = (signed char)D[px][b];
          k.wt_tap[k.ntaps] = (signed char)(KK[py][a] * 3 + KK[px][b]);
          ++k.ntaps;
        }
      const int rc = wide_launch(k, 1, flat_ok(IH, IW, k.IPH * k.IPW, CK) ? ws : nullptr, ws_floats, (hipStream_t)stream);
      if (rc != VTS_OK) return rc;
    }
  return VTS_OK;
}

// 4 x 4 convolutions of wide layers (the ndf = 64 PatchGAN discriminators of pix2pixHD, reference models/networks.py
// NLayerDiscriminator via MultiscaleDiscriminator, kw = 4, padw = 2): 16-tap packed weights on the GEMM-class kernels --
// conv4x4_wide_kernel for full-size maps, the flattened kernel for maps of <= 128 pixels (32 x 32 training patches).  transposed = 0: out[n,co,y,x] = bias + sum in[n,ci,S*y+ky,S*x+kx] * wt[(ci*16 + ky*4+kx)*Cout + co]
// on the pre-padded input.  transposed = 1 (stride 2, the input adjoint of Conv2d(4, s2, p2)): `in` is the output gradient
// with one zero row / column appended; out[2m+py] = sum_{d in {0,1}} in[m+d] * w[k = py + 2(1-d)], one launch per parity phase.
static bool flat4_shapes_ok(int OH, int OW, int PH, int PW, int transposed) {
  const int gh = transposed ? (OH + 1) / 2 : OH, gw = transposed ? (OW + 1) / 2 : OW;
  return flat_ok(gh, gw, PH * PW, 4);
}

extern "C" int vts_conv4x4_flat_ok(int OH, int OW, int PH, int PW, int transposed) { return flat4_shapes_ok(OH, OW, PH, PW, transposed) ? 1 : 0; }

extern "C" int64_t vts_conv4x4_wide_ws_floats(int N, int Cin, int Cout, int OH, int OW, int PH, int PW, int transposed) {
  const int gh = transposed ? (OH + 1) / 2 : OH, gw = transposed ? (OW + 1) / 2 : OW;
  int ipt, cps;
  if (flat4_shapes_ok(OH, OW, PH, PW, transposed)) {
    const int KS = flat_plan(N, Cin, Cout, gh, gw, PH * PW, 4, &ipt, &cps);
    return KS > 1 ? (int64_t)KS * N * Cout * gh * gw : 0;
  }
  if (transposed) return 0;
  const int KS = wide_plan(N, Cin, Cout, OH, OW, &cps, 4);
  return KS > 1 ? (int64_t)KS * N * Cout * OH * OW : 0;
}

extern "C" int vts_conv4x4_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int PH, int PW,
                                int OH, int OW, int stride, int transposed, float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(in && wt && out && N >= 1 && Cin >= 1 && Cout >= 1 && OH >= 1 && OW >= 1 && (stride == 1 || stride == 2),
                "vts_conv4x4_wide: bad args");
  WideK k{};
  k.in = in; k.wt = wt; k.bias = bias; k.out = out; k.N = N; k.Cin = Cin; k.Cout = Cout; k.IPH = PH; k.IPW = PW; k.OH = OH; k.OW = OW;
  if (!transposed) {
    VTS_CHECK_ARG(PH >= stride * (OH - 1) + 4 && PW >= stride * (OW - 1) + 4, "vts_conv4x4_wide: padded input %d x %d too small for output %d x %d", PH, PW, OH, OW);
    k.H = OH; k.W = OW; k.os = 1; k.oy0 = 0; k.ox0 = 0; k.ntaps = 16;
    for (int t = 0; t < 16; ++t) { k.dy[t] = (signed char)(t / 4); k.dx[t] = (signed char)(t % 4); k.wt_tap[t] = (signed char)t; }
    return wide_launch(k, stride, ws, ws_floats, (hipStream_t)stream, 4);
  }
  VTS_CHECK_ARG(stride == 2, "vts_conv4x4_wide: the transposed form is the stride-2 one (stride 1: flipped packing on the padded gradient)");
  VTS_CHECK_ARG(2 * (PH - 1) >= OH + 1 && 2 * (PW - 1) >= OW + 1, "vts_conv4x4_wide: gradient extent %d x %d too small for %d x %d", PH, PW, OH, OW);
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      k.H = (OH - py + 1) / 2; k.W = (OW - px + 1) / 2;
      if (k.H < 1 || k.W < 1) continue;
      k.os = 2; k.oy0 = py; k.ox0 = px; k.ntaps = 0;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
          k.dy[k.ntaps] = (signed char)a; k.dx[k.ntaps] = (signed char)b;
          k.wt_tap[k.ntaps] = (signed char)((py + 2 * (1 - a)) * 4 + px + 2 * (1 - b));
          ++k.ntaps;
        }
      const int rc = wide_launch(k, 1, ws, ws_floats, (hipStream_t)stream, 4);
      if (rc != VTS_OK) return rc;
    }
  return VTS_OK;
}

// =====================================================================================================
// Weight gradient of the same operator:
//   dw[co][ci][ky][kx] (+)= sum_{n,y,x} dout[n,co,y,x] * in[n,ci,y+ky,x+kx]          (in pre-padded)
// GEMM view: M = output channels, N = input channels (x 9 taps, one accumulator tile per tap), K = pixels.
// A workgroup owns 64 co x 64 ci x 9 taps (a wave 32 x 32 x 9 = 144 accumulator registers) and walks 2 x 32
// pixel tiles of its K slice; both operands vary their LANE index over channels, so their LDS planes have an
// odd pitch (conflict-free fragment reads) and are written with 4-byte stores.  Partial sums of the K slices
// are reduced in slice order by wg_wide_reduce_kernel (deterministic, no float atomics).
// =====================================================================================================
namespace {

constexpr int GCO = 64, GCI = 64, GTY = 2;

struct WgWideK {
  const float *dout, *in;
  float* part;     // [KS][Cout][Cin][9]
  int N, Cin, Cout, H, W;   // H x W: extent of dout
  int IPH, IPW;             // padded extent of `in`
  int tiles_x, tiles_per_img, ntiles, tps;   // pixel tiles; tps = tiles per K slice
  int TW, tap0, in_skip;    // taps of dw (9 | 16), first tap of this launch, floats by which `in` was advanced (bounds)
};

// S: stride of the convolution whose weight gradient this is (in is sampled at S*y + ky, S*x + kx)
// KH x KW: the taps this launch accumulates (3 x 3; a 4 x 4 kernel takes two launches of 2 x 4 taps: 128 accumulator registers each)
template <int S, int KH, int KW>
__device__ __forceinline__ void wg_wide_body(const WgWideK& p) {
  constexpr int NT = KH * KW;
  constexpr int GTX = S == 1 ? 32 : 16;
  constexpr int GPX = GTY * GTX;                           // pixels per tile
  constexpr int DO_PITCH = GPX + 1;                        // dout plane [co][px], odd pitch
  constexpr int GPR = S * (GTY - 1) + KH, GPC = (S * (GTX - 1) + KW + 3) / 4 * 4;   // patch rows, staged columns
  constexpr int P_PITCH = GPR * GPC + 1;                   // in plane [ci][r][c], odd pitch
  constexpr int GDO_FLOATS = GCO * DO_PITCH, GP_FLOATS = GCI * P_PITCH;
  constexpr int DQ_TOTAL = GCO * GPX / 4, IQ_TOTAL = GCI * GPR * (GPC / 4);
  constexpr int NDQ = (DQ_TOTAL + 255) / 256, NIQ = (IQ_TOTAL + 255) / 256;
  __shared__ float lds[GDO_FLOATS + GP_FLOATS];
  float* lds_d = lds;
  float* lds_i = lds + GDO_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wci = wave >> 1;
  const int co0 = blockIdx.x * GCO, ci0 = blockIdx.y * GCI, ks = blockIdx.z;
  const int PW = p.IPW, plane = p.IPH * p.IPW, oplane = p.H * p.W;

  // staging items: dout quads (co, row, xquad) and patch quads (ci, r, cquad); tile offsets are added per tile
  int dvoff[NDQ], dloff[NDQ];
#pragma unroll
  for (int e = 0; e < NDQ; ++e) {
    const int q = min(tid + e * 256, DQ_TOTAL - 1);
    const int co = q / (GPX / 4), r4 = q - co * (GPX / 4);
    const int row = r4 / (GTX / 4), xq = r4 - row * (GTX / 4);
    dvoff[e] = ((co0 + co) * oplane + row * p.W + 4 * xq) * 4;
    dloff[e] = co * DO_PITCH + row * GTX + 4 * xq;
  }
  int ivoff[NIQ], iloff[NIQ];
#pragma unroll
  for (int e = 0; e < NIQ; ++e) {
    const int q = min(tid + e * 256, IQ_TOTAL - 1);
    const int ci = q / (GPR * (GPC / 4)), r9 = q - ci * (GPR * (GPC / 4));
    const int r = r9 / (GPC / 4), cq = r9 - r * (GPC / 4);
    ivoff[e] = ((ci0 + ci) * plane + r * PW + 4 * cq) * 4;
    iloff[e] = ci * P_PITCH + r * GPC + 4 * cq;
  }

  u32x4 dq[NDQ], iq[NIQ];
  int cur_x0 = 0, cur_y0 = 0;   // origin of the tile held in the prefetch registers
  auto load_tile = [&](int t) {
    const int n = t / p.tiles_per_img, r = t - n * p.tiles_per_img;
    const int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    cur_x0 = tx * GTX;
    cur_y0 = ty * GTY;
    const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout) + (int64_t)n * p.Cout * oplane, 0, p.Cout * oplane * 4, RSRC_FLAGS);
    const auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (int64_t)n * p.Cin * plane, 0, p.Cin * plane * 4 - p.in_skip * 4, RSRC_FLAGS);
    const int doff = (cur_y0 * p.W + cur_x0) * 4, ioff = (S * cur_y0 * PW + S * cur_x0) * 4;
#pragma unroll
    for (int e = 0; e < NDQ; ++e) dq[e] = __builtin_amdgcn_raw_buffer_load_b128(rd, dvoff[e] + doff, 0, 0);
#pragma unroll
    for (int e = 0; e < NIQ; ++e) iq[e] = __builtin_amdgcn_raw_buffer_load_b128(ri, ivoff[e] + ioff, 0, 0);
  };
  // dout pixels outside the image must contribute 0 (the patch side may hold anything finite there); channels past
  // Cout / Cin are masked by index (the bounds check only catches them when nothing follows in memory).
  // (whole-quad casts: extracting quad[j] through __builtin_bit_cast per element mis-compiled to element 0)
  auto store_tile = [&]() {
#pragma unroll
    for (int e = 0; e < NDQ; ++e) {
      const int q = min(tid + e * 256, DQ_TOTAL - 1);
      const int co = q / (GPX / 4), r4 = q - co * (GPX / 4);
      const int row = r4 / (GTX / 4), xq = r4 - row * (GTX / 4);
      const bool rok = co0 + co < p.Cout && cur_y0 + row < p.H;
      const f32x4 v = __builtin_bit_cast(f32x4, dq[e]);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_d[dloff[e] + j] = (rok && cur_x0 + 4 * xq + j < p.W) ? v[j] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NIQ; ++e) {
      const int q = min(tid + e * 256, IQ_TOTAL - 1);
      const int ci = q / (GPR * (GPC / 4));
      const bool cok = ci0 + ci < p.Cin;
      const f32x4 v = __builtin_bit_cast(f32x4, iq[e]);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_i[iloff[e] + j] = cok ? v[j] : 0.f;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* a_base = lds_d + (wco * 32 + l32) * DO_PITCH + kh;            // A[i = co][k = pixel]
  const float* b_base = lds_i + (wci * 32 + l32) * P_PITCH + S * kh;         // B[k = pixel][j = ci]

  const int t_beg = ks * p.tps, t_end = min(p.ntiles, t_beg + p.tps);
  if (t_beg < t_end) {
    load_tile(t_beg);
    store_tile();
  }
  __syncthreads();
  for (int t = t_beg; t < t_end; ++t) {
    const bool more = t + 1 < t_end;
    if (more) load_tile(t + 1);
#pragma unroll
    for (int row = 0; row < GTY; ++row)
#pragma unroll 4
      for (int xs = 0; xs < GTX / 2; ++xs) {
        const float a = a_base[row * GTX + xs * 2];
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
          const int ky = tap / KW, kx = tap - ky * KW;
          const float b = b_base[(S * row + ky) * GPC + S * xs * 2 + kx];
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
        }
      }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }

  // C layout: column (ci) = lane % 32, row (co) = (r / 4) * 8 + (lane / 32) * 4 + r % 4
  float* ob = p.part + (int64_t)ks * p.Cout * p.Cin * p.TW + p.tap0;
  const int ci = ci0 + wci * 32 + l32;
#pragma unroll
  for (int tap = 0; tap < NT; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
      if (co < p.Cout && ci < p.Cin) ob[((int64_t)co * p.Cin + ci) * p.TW + tap] = acc[tap][r];
    }
}

template <int S>
__global__ __launch_bounds__(256) void wgrad3x3_wide_kernel(const WgWideK p) { wg_wide_body<S, 3, 3>(p); }

// two rows of the taps of a 4 x 4 kernel (p.tap0 = 4 * first row; p.in starts at that row of the padded input)
template <int S>
__global__ __launch_bounds__(256) void wgrad4x4_wide_kernel(const WgWideK p) { wg_wide_body<S, 2, 4>(p); }


__global__ __launch_bounds__(256) void wg_wide_reduce_kernel(const float* __restrict__ part, int KS, int64_t n, float* __restrict__ dw, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = accumulate ? dw[i] : 0.f;
  for (int k = 0; k < KS; ++k) v += part[k * n + i];
  dw[i] = v;
}

// ---- flattened small-map variant of the weight gradient: K = the flattened (image, y, x) index of IPT whole images
// (<= 64 pixels per tile), operands staged as complete planes, per-pixel plane offsets from an LDS table.
constexpr int WGF_PX = 64, WGF_DO_PITCH = WGF_PX + 1, WGF_IN_CAP = 256;   // WGF_IN_CAP: floats per input channel (IPT planes)

struct WgFlatK {
  const float *dout, *in;
  float* out;               // dw (direct) or part [KS][Cout][Cin][9]
  int N, Cin, Cout, H, W;   // H x W: extent of dout
  int IPH, IPW, S;
  int IPT, ntiles, tps;     // images per K tile, tiles, tiles per K slice
  int direct, accumulate;   // direct: one K slice, write (or accumulate into) dw from the epilogue
};

__global__ __launch_bounds__(256) void wgrad3x3_flat_kernel(const WgFlatK p) {
  constexpr int P_PITCH = WGF_IN_CAP + 1;
  constexpr int NIQ = GCI * (WGF_IN_CAP / 4) / 256;       // 16 quads per thread at most
  __shared__ float lds[GCO * WGF_DO_PITCH + GCI * P_PITCH];
  __shared__ int lds_pix[WGF_PX];
  float* lds_d = lds;
  float* lds_i = lds + GCO * WGF_DO_PITCH;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wci = wave >> 1;
  const int co0 = blockIdx.x * GCO, ci0 = blockIdx.y * GCI, ks = blockIdx.z;
  const int PW = p.IPW, plane = p.IPH * p.IPW, HW = p.H * p.W;
  const int pq = (plane + 3) >> 2, TQ = GCI * p.IPT * pq;
  const int KPX = p.IPT * HW;                              // pixels per full tile (<= 64)

  const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout), 0, p.N * p.Cout * HW * 4, RSRC_FLAGS);
  const auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.N * p.Cin * plane * 4, RSRC_FLAGS);

  // pixel table (tile independent): px -> offset of tap (0, 0) inside this channel's IPT planes
  if (tid < WGF_PX) {
    const int img = tid / HW, r = tid - img * HW, y = r / p.W, x = r - y * p.W;
    lds_pix[tid] = tid < KPX ? img * plane + p.S * y * PW + p.S * x : 0;
  }
  // dout staging: thread -> pixel tid % 64, channels tid / 64 + 4 e
  const int dpx = tid & 63, dimg = dpx / HW, dr = dpx - dimg * HW;
  // input staging: quads (ci, img, quad of the plane)
  int ivoff[NIQ], iloff[NIQ], irem[NIQ];
#pragma unroll
  for (int e = 0; e < NIQ; ++e) {
    const int q = min(tid + e * 256, TQ - 1);
    const int ci = q / (p.IPT * pq), r = q - ci * (p.IPT * pq);
    const int img = r / pq, qi = r - img * pq;
    ivoff[e] = ci0 + ci < p.Cin ? ((img * p.Cin + ci0 + ci) * plane + 4 * qi) * 4 : 0x7ffffff0;
    iloff[e] = ci * P_PITCH + img * plane + 4 * qi;
    irem[e] = (plane - 4 * qi) | (img << 16);             // valid floats of the quad, image index
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* a_base = lds_d + (wco * 32 + l32) * WGF_DO_PITCH + kh;
  const float* b_base = lds_i + (wci * 32 + l32) * P_PITCH;

  const int t_beg = ks * p.tps, t_end = min(p.ntiles, t_beg + p.tps);
  for (int t = t_beg; t < t_end; ++t) {
    const int n0 = t * p.IPT;
    float dv[16];
    {
      const bool pok = dpx < KPX && n0 + dimg < p.N;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + (tid >> 6) + 4 * e;
        dv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, pok && co < p.Cout ? (((n0 + dimg) * p.Cout + co) * HW + dr) * 4 : 0x7ffffff0, 0, 0));
      }
    }
    u32x4 iq[NIQ];
    const int ibase = n0 * p.Cin * plane * 4;
#pragma unroll
    for (int e = 0; e < NIQ; ++e)
      if (e * 256 < TQ) iq[e] = __builtin_amdgcn_raw_buffer_load_b128(ri, n0 + (irem[e] >> 16) < p.N ? ivoff[e] + ibase : 0x7ffffff0, 0, 0);
    if (t > t_beg) __syncthreads();                        // the previous tile's fragments have been read
#pragma unroll
    for (int e = 0; e < 16; ++e) lds_d[((tid >> 6) + 4 * e) * WGF_DO_PITCH + dpx] = dv[e];
#pragma unroll
    for (int e = 0; e < NIQ; ++e)
      if (e * 256 < TQ) {
        const f32x4 v = __builtin_bit_cast(f32x4, iq[e]);
        const int rem = irem[e] & 0xffff;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < rem) lds_i[iloff[e] + j] = v[j];
      }
    __syncthreads();
    const int kend = (min(KPX, (p.N - n0) * HW) + 1) >> 1;
    for (int kk = 0; kk < kend; ++kk) {
      const float a = a_base[2 * kk];
      const float* b = b_base + lds_pix[2 * kk + kh];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[ky * PW + kx], acc[tap], 0, 0, 0);
      }
    }
  }

  float* ob = p.direct ? p.out : p.out + (int64_t)ks * p.Cout * p.Cin * 9;
  const int ci = ci0 + wci * 32 + l32;
  const bool rmw = p.direct && p.accumulate;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
      if (co < p.Cout && ci < p.Cin) {
        float* o = ob + ((int64_t)co * p.Cin + ci) * 9 + tap;
        *o = acc[tap][r] + (rmw ? *o : 0.f);
      }
    }
}

bool wg_flat_ok(int H, int W, int plane) { return H * W <= WGF_PX && (plane + 3) / 4 * 4 <= WGF_IN_CAP; }
int wg_flat_plan(int N, int Cin, int Cout, int H, int W, int plane, int* ipt, int* tps) {
  int IPT = WGF_PX / (H * W);
  if (IPT > WGF_IN_CAP / 4 / ((plane + 3) / 4)) IPT = WGF_IN_CAP / 4 / ((plane + 3) / 4);   // staged as whole quads
  if (IPT > N) IPT = N;
  *ipt = IPT;
  const int ntiles = cdiv(N, IPT);
  const int groups = cdiv(Cout, GCO) * cdiv(Cin, GCI);
  int KS = groups >= 192 ? 1 : 384 / groups;
  if (KS > ntiles) KS = ntiles;
  if (KS < 1) KS = 1;
  *tps = cdiv(ntiles, KS);
  return cdiv(ntiles, *tps);
}

int wg_wide_plan(int N, int Cin, int Cout, int H, int W, int S, int* tps) {
  const int ntiles = N * cdiv(H, GTY) * cdiv(W, S == 1 ? 32 : 16);
  const int groups = cdiv(Cout, GCO) * cdiv(Cin, GCI);
  int KS = 512 / groups;                      // aim at two workgroups per CU
  if (KS > ntiles / 4) KS = ntiles / 4;
  if (KS < 1) KS = 1;
  *tps = cdiv(ntiles, KS);
  return cdiv(ntiles, *tps);
}

}  // namespace

extern "C" int64_t vts_wgrad3x3_wide_ws_floats(int N, int Cin, int Cout, int H, int W, int stride) {
  int tps, ipt;
  const int plane = (stride * H + 2) * (stride * W + 2);
  if (wg_flat_ok(H, W, plane)) {
    const int KS = wg_flat_plan(N, Cin, Cout, H, W, plane, &ipt, &tps);
    return KS > 1 ? (int64_t)KS * Cout * Cin * 9 : 0;
  }
  return (int64_t)wg_wide_plan(N, Cin, Cout, H, W, stride, &tps) * Cout * Cin * 9;
}

extern "C" int vts_wgrad3x3_wide(const float* dout, const float* in, float* dw, int N, int Cin, int Cout, int H, int W, int stride,
                                 int accumulate, float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(dout && in && dw && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && (stride == 1 || stride == 2),
                "vts_wgrad3x3_wide: bad args");
  const int IPH = stride * H + 2, IPW = stride * W + 2;
  const int64_t nel = (int64_t)Cout * Cin * 9;
  if (wg_flat_ok(H, W, IPH * IPW)) {
    VTS_CHECK_ARG((int64_t)N * Cin * IPH * IPW * 4 < (1ll << 31) && (int64_t)N * Cout * H * W * 4 < (1ll << 31),
                  "vts_wgrad3x3_wide: operand exceeds the 2 GiB buffer range");
    WgFlatK f;
    f.dout = dout; f.in = in; f.N = N; f.Cin = Cin; f.Cout = Cout; f.H = H; f.W = W; f.IPH = IPH; f.IPW = IPW; f.S = stride;
    const int KS = wg_flat_plan(N, Cin, Cout, H, W, IPH * IPW, &f.IPT, &f.tps);
    f.ntiles = cdiv(N, f.IPT);
    f.direct = KS == 1; f.accumulate = accumulate; f.out = KS == 1 ? dw : ws;
    VTS_CHECK_ARG(KS == 1 || (ws && ws_floats >= KS * nel), "vts_wgrad3x3_wide: workspace too small (%lld < %lld floats)",
                  (long long)ws_floats, (long long)(KS * nel));
    hipLaunchKernelGGL(wgrad3x3_flat_kernel, dim3(cdiv(Cout, GCO), cdiv(Cin, GCI), KS), dim3(256), 0, (hipStream_t)stream, f);
    vts_set_kernel("wgrad3x3_flat_kernel<%d>", stride);
    VTS_CHECK_LAUNCH("vts_wgrad3x3_wide (flat)");
    if (KS > 1) {
      hipLaunchKernelGGL(wg_wide_reduce_kernel, dim3((unsigned)cdiv64(nel, 256)), dim3(256), 0, (hipStream_t)stream, ws, KS, nel, dw, accumulate);
      VTS_CHECK_LAUNCH("vts_wgrad3x3_wide (flat) reduce");
    }
    return VTS_OK;
  }
  VTS_CHECK_ARG(ws != nullptr, "vts_wgrad3x3_wide: workspace required");
  WgWideK k;
  k.dout = dout; k.in = in; k.part = ws; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = H; k.W = W;
  k.IPH = IPH; k.IPW = IPW; k.TW = 9; k.tap0 = 0; k.in_skip = 0;
  VTS_CHECK_ARG((int64_t)Cin * k.IPH * k.IPW * 4 < (1ll << 31) && (int64_t)Cout * H * W * 4 < (1ll << 31), "vts_wgrad3x3_wide: operand exceeds the 2 GiB buffer range");
  k.tiles_x = cdiv(W, stride == 1 ? 32 : 16);
  k.tiles_per_img = k.tiles_x * cdiv(H, GTY);
  k.ntiles = N * k.tiles_per_img;
  int KS = wg_wide_plan(N, Cin, Cout, H, W, stride, &k.tps);
  VTS_CHECK_ARG(ws_floats >= KS * nel, "vts_wgrad3x3_wide: workspace too small (%lld < %lld floats)", (long long)ws_floats, (long long)(KS * nel));
  if (stride == 1) {      // Winograd F(3x3, 2x2) form (vts_conv3x3_wino.hip) where it takes the shape: the same partial layout, fewer slices
    const int wks = vts_wgrad3x3_wino_try(dout, in, ws, N, Cin, Cout, H, W, KS, (hipStream_t)stream);
    if (wks > 0) {
      VTS_CHECK_LAUNCH("vts_wgrad3x3_wide (winograd)");
      hipLaunchKernelGGL(wg_wide_reduce_kernel, dim3((unsigned)cdiv64(nel, 256)), dim3(256), 0, (hipStream_t)stream, ws, wks, nel, dw, accumulate);
      VTS_CHECK_LAUNCH("vts_wgrad3x3_wide reduce");
      return VTS_OK;
    }
  }
  const dim3 grid(cdiv(Cout, GCO), cdiv(Cin, GCI), KS);
  if (stride == 1) hipLaunchKernelGGL(wgrad3x3_wide_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, k);
  else hipLaunchKernelGGL(wgrad3x3_wide_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, k);
  vts_set_kernel("wgrad3x3_wide_kernel<%d>", stride);
  VTS_CHECK_LAUNCH("vts_wgrad3x3_wide");
  hipLaunchKernelGGL(wg_wide_reduce_kernel, dim3((unsigned)cdiv64(nel, 256)), dim3(256), 0, (hipStream_t)stream, ws, KS, nel, dw, accumulate);
  VTS_CHECK_LAUNCH("vts_wgrad3x3_wide reduce");
  return VTS_OK;
}

// Weight gradient of Conv2d(4, stride, padding) of a wide layer on full-size maps (the ndf = 64 PatchGAN layers of pix2pixHD):
//   dw[co][ci][ky][kx] (+)= sum_{n,y,x} dout[n,co,y,x] * in[n,ci,stride*y+ky,stride*x+kx]      in [N,Cin,PH,PW] pre-padded
// two launches of wgrad4x4_wide_kernel (tap rows 0-1 and 2-3) into one partial buffer, one slice reduction.
extern "C" int64_t vts_wgrad4x4_wide_ws_floats(int N, int Cin, int Cout, int H, int W, int stride) {
  int tps;
  return (int64_t)wg_wide_plan(N, Cin, Cout, H, W, stride, &tps) * Cout * Cin * 16;
}

extern "C" int vts_wgrad4x4_wide(const float* dout, const float* in, float* dw, int N, int Cin, int Cout, int H, int W, int PH, int PW,
                                 int stride, int accumulate, float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(dout && in && dw && ws && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && (stride == 1 || stride == 2),
                "vts_wgrad4x4_wide: bad args");
  VTS_CHECK_ARG(PH >= stride * (H - 1) + 4 && PW >= stride * (W - 1) + 4, "vts_wgrad4x4_wide: padded input %d x %d too small for a %d x %d gradient", PH, PW, H, W);
  VTS_CHECK_ARG((int64_t)Cin * PH * PW * 4 < (1ll << 31) && (int64_t)Cout * H * W * 4 < (1ll << 31), "vts_wgrad4x4_wide: operand exceeds the 2 GiB buffer range");
  WgWideK k;
  k.dout = dout; k.part = ws; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = H; k.W = W; k.IPH = PH; k.IPW = PW; k.TW = 16;
  k.tiles_x = cdiv(W, stride == 1 ? 32 : 16);
  k.tiles_per_img = k.tiles_x * cdiv(H, GTY);
  k.ntiles = N * k.tiles_per_img;
  const int KS = wg_wide_plan(N, Cin, Cout, H, W, stride, &k.tps);
  const int64_t nel = (int64_t)Cout * Cin * 16;
  VTS_CHECK_ARG(ws_floats >= KS * nel, "vts_wgrad4x4_wide: workspace too small (%lld < %lld floats)", (long long)ws_floats, (long long)(KS * nel));
  const dim3 grid(cdiv(Cout, GCO), cdiv(Cin, GCI), KS);
  for (int half = 0; half < 2; ++half) {
    k.in = in + (int64_t)2 * half * PW; k.in_skip = 2 * half * PW; k.tap0 = 8 * half;
    if (stride == 1) hipLaunchKernelGGL(wgrad4x4_wide_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, k);
    else hipLaunchKernelGGL(wgrad4x4_wide_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, k);
    VTS_CHECK_LAUNCH("vts_wgrad4x4_wide");
  }
  vts_set_kernel("wgrad4x4_wide_kernel<%d>", stride);
  hipLaunchKernelGGL(wg_wide_reduce_kernel, dim3((unsigned)cdiv64(nel, 256)), dim3(256), 0, (hipStream_t)stream, ws, KS, nel, dw, accumulate);
  VTS_CHECK_LAUNCH("vts_wgrad4x4_wide reduce");
  return VTS_OK;
}
