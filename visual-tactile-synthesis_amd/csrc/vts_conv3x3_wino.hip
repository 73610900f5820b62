// Winograd F(2 x 2, 3 x 3) form of the GEMM-class 3 x 3 convolution (round 4), for the frozen VGG stacks of the perceptual terms
// (lpips.LPIPS(net="vgg") as the reference calls it, models/sinskitG_model.py:495, 1639-1646, 1711; VGGLoss, models/networks.py:2021-2067):
// those stacks are 75 of the 94 ms of the reference-default step and already run at 0.76 of the fp32 MFMA peak as direct convolutions, so
// the multiplications themselves have to go: 16 instead of 36 per 2 x 2 outputs and channel pair (Lavin & Gray 2016; cuDNN / MIOpen offer
// the same algorithm for fp32 3 x 3 layers).  Arithmetic stays fp32; the result differs from the direct form by rounding only (observed
// ~1e-6 relative), inside the tolerance the parity tests state.
//
//   Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A       d: 4 x 4 input patch, g: 3 x 3 taps, Y: 2 x 2 outputs, '.' elementwise
//
// Weights are transformed once (the stacks are frozen): U[ci][p][co], p = 4 i + j the position in the 4 x 4 transform domain.
// Kernel: a workgroup (4 waves, ONE per SIMD: the 16 accumulator tiles of a wave fill the 256 accumulation registers) owns 64 output
// channels x 64 tiles (8 x 8 tiles = 16 x 16 output pixels) of one image.  Per chunk of 8 input channels it transforms the 512
// (channel, tile) patches to V[p][ci][tile] in LDS (two per thread, straight from global memory: rows of four floats at even columns),
// stages U[p][ci][co] beside it, and runs 16 positions x 4 k-steps of v_mfma_f32_32x32x2_f32 (64 per wave and chunk: a 64 x 64 x 8 GEMM per
// position).  The next chunk's global loads are issued into registers before the MFMA phase.  The output transform runs on the
// accumulators in registers (every position of one (channel, tile) pair lives in the same lane), with the epilogues of the direct kernel:
// bias, ReLU into the padded layout, mask + tap gradient (vts_conv3x3_wide_relu_pad / _mask_pad).
#include "vts_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));     // an 8-byte access at dword alignment
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int RSRC_FLAGS = 0x00020000;     // raw buffer, dword range check (as vts_conv3x3_wide.hip)

struct WinoK {
  const float* in;    // pre-padded [N][Cin][IPH][IPW]
  const float* U;     // [Cin8][16][Cout]  (Cin8 = Cin rounded up to 8, zero rows beyond Cin)
  const float* bias;
  float* out;         // [N][Cout][OH][OW], the H x W result at (oy0, ox0)
  int N, Cin, Cout, H, W, IPH, IPW, OH, OW, oy0, ox0;
  int ep_mode;        // 0 plain, 1 relu (+ zero border of the padded layout), 2 (acc + ep_add) where ep_mask > 0 (+ zero border)
  const float* ep_add;
  const float* ep_mask;
  // flat-tile mode (maps smaller than a 16 x 16 block, large batches: the tactile patches' deep layers): a workgroup takes 64 consecutive
  // 2 x 2 tiles of the FLATTENED (image, tile row, tile column) index instead of an 8 x 8 block of tiles of one image
  int flat, tiles_x, tpi, ntiles;
};

constexpr int CKW = 8, TCO = 64, TT = 64;     // channels per chunk, output channels and tiles per workgroup

__device__ __forceinline__ f32x4 ld4(const rsrc_t& rs, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}

__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(const WinoK p) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 16 * CKW * 64];
  float* lds_u = lds;                       // [p][ci][64 co]
  float* lds_v = lds + 16 * CKW * 64;       // [p][ci][64 tiles]
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mi = wave & 1, ni = wave >> 1;
  const int bx_n = (p.W + 15) >> 4;
  const int bx = blockIdx.x % bx_n, by = blockIdx.x / bx_n;
  const int x0 = bx * 16, y0 = by * 16, co0 = blockIdx.y * TCO, n = blockIdx.z;
  const int plane = p.IPH * p.IPW;
  const int nchunks = (p.Cin + CKW - 1) / CKW;

  // buffer resources: the whole input from this image on (a patch row beyond the map reads the next plane -- finite garbage that only
  // reaches outputs beyond the map, which are not stored; beyond the tensor: zeros), the transformed weights
  const int64_t in_floats = (int64_t)(p.N - n) * p.Cin * plane;
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (int64_t)n * p.Cin * plane, 0,
                                                       (int)(in_floats * 4 > 0x7fffffff ? 0x7fffffff : in_floats * 4), RSRC_FLAGS);
  const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, nchunks * CKW * 16 * p.Cout * 4, RSRC_FLAGS);

  // this thread's two (channel, tile) patches of a chunk and its eight weight quads
  int doff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int j = tid + e * 256, ci = j >> 6, t = j & 63, ty = t >> 3, tx = t & 7;
    doff[e] = (ci * plane + (y0 + 2 * ty) * p.IPW + x0 + 2 * tx) * 4;
  }
  int uoff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int idx = tid + e * 256, row = idx >> 4, q = idx & 15;     // row = ci * 16 + p of the chunk, 16 quads of output channels
    uoff[e] = (row * p.Cout + co0 + 4 * q) * 4;
  }
  f32x4 dreg[2][4], ureg[8];
  auto load_chunk = [&](int c) {
    const int cb = c * CKW * plane * 4, ub = c * CKW * 16 * p.Cout * 4;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) dreg[e][r] = ld4(rs_in, cb + doff[e] + r * p.IPW * 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) ureg[e] = ld4(rs_u, ub + uoff[e]);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = tid + e * 256, row = idx >> 4, q = idx & 15, ci = row >> 4, pp = row & 15;
      *reinterpret_cast<f32x4*>(lds_u + (pp * CKW + ci) * 64 + 4 * q) = ureg[e];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = tid + e * 256, ci = j >> 6, t = j & 63;
      // V = B^T d B
      float tt[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d0 = dreg[e][0][c], d1 = dreg[e][1][c], d2 = dreg[e][2][c], d3 = dreg[e][3][c];
        tt[0][c] = d0 - d2; tt[1][c] = d1 + d2; tt[2][c] = d2 - d1; tt[3][c] = d1 - d3;
      }
      float* v = lds_v + ci * 64 + t;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[((4 * i + 0) * CKW) * 64] = tt[i][0] - tt[i][2];
        v[((4 * i + 1) * CKW) * 64] = tt[i][1] + tt[i][2];
        v[((4 * i + 2) * CKW) * 64] = tt[i][2] - tt[i][1];
        v[((4 * i + 3) * CKW) * 64] = tt[i][1] - tt[i][3];
      }
    }
  };

  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    store_chunk();
    __syncthreads();
    if (c + 1 < nchunks) load_chunk(c + 1);
    const float* ua = lds_u + kh * 64 + mi * 32 + l32;
    const float* vb = lds_v + kh * 64 + ni * 32 + l32;
#pragma unroll
    for (int pp = 0; pp < 16; ++pp)
#pragma unroll
      for (int ks = 0; ks < CKW / 2; ++ks) {
        const float a = ua[(pp * CKW + 2 * ks) * 64], b = vb[(pp * CKW + 2 * ks) * 64];
        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[pp], 0, 0, 0);
      }
    __syncthreads();
  }

  // output transform Y = A^T M A per (channel, tile); C layout of a 32 x 32 tile: column (tile) = lane % 32, row (channel) =
  // (r / 4) * 8 + (lane / 32) * 4 + r % 4
  const int t = ni * 32 + l32, ty = t >> 3, tx = t & 7;
  const int oy = y0 + 2 * ty, ox = x0 + 2 * tx;
  const int64_t oplane = (int64_t)p.OH * p.OW;
  float* ob = p.out + (int64_t)n * p.Cout * oplane;
  if (oy < p.H && ox < p.W) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mi * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
      float s[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[0][j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
        s[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
      }
      const float bsv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float y[2] = {s[a][0] + s[a][1] + s[a][2] + bsv, s[a][1] - s[a][2] - s[a][3] + bsv};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int yy = oy + a, xx = ox + b;
          if (yy < p.H && xx < p.W) {
            const int64_t o = co * oplane + (int64_t)(p.oy0 + yy) * p.OW + p.ox0 + xx;
            float v = y[b];
            if (p.ep_mode == 1) v = fmaxf(v, 0.f);
            if (p.ep_mode == 2) {
              const int64_t e = (int64_t)n * p.Cout * oplane + o;
              v = p.ep_mask[e] > 0.f ? v + (p.ep_add ? p.ep_add[e] : 0.f) : 0.f;
            }
            ob[o] = v;
            if (p.ep_mode) {      // padded output: the pixels on the rim of the map also store the zero border next to them
              float* q = ob + o;
              const bool xl = xx == 0, xr = xx == p.W - 1;
              if (xl) q[-1] = 0.f;
              if (xr) q[1] = 0.f;
              if (yy == 0) {
                q[-p.OW] = 0.f;
                if (xl) q[-p.OW - 1] = 0.f;
                if (xr) q[-p.OW + 1] = 0.f;
              }
              if (yy == p.H - 1) {
                q[p.OW] = 0.f;
                if (xl) q[p.OW - 1] = 0.f;
                if (xr) q[p.OW + 1] = 0.f;
              }
            }
          }
        }
      }
    }
  }
}

// The same tile on EIGHT waves (two per SIMD, round 4).  With one wave per SIMD the phases of a chunk run one after the other -- ablation:
// issuing the chunk's sixteen 16-byte loads stalls the wave ~1300 cycles in the memory pipeline, the transform + LDS writes take ~750, the
// 64 MFMAs 4096 -- because an in-order wave has nothing to overlap them with, and every attempt to interleave them inside one wave lost to
// conservative waits.  Here the sixteen transform positions are split between two waves (rows 0-1 / rows 2-3 of the 4 x 4 transform
// domain: 128 accumulation registers each), so that each SIMD holds two instruction streams and issues one wave's loads / transform
// between the other's MFMAs.  The price is the output transform: Y = A^T M A needs all four rows, so the two waves exchange their
// partial row sums through LDS once per workgroup (each then finishes one of the two output rows of every tile).
// ABL: profiling instances (env VTS_WINO_ABLATE; compile-time -- a run-time test inside the unrolled loops wrecks the code): 1 no global loads after
// the first chunk, 8 no transform / LDS staging after the first chunk (results are wrong, timing only)
template <int ABL>
__global__ __launch_bounds__(512, 2) void conv3x3_wino8_kernel(const WinoK p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // two buffers of (U [p][ci][64 co], V [p][ci][64 tiles]): 2 x 64 KB
  constexpr int BUF = 2 * 16 * CKW * 64;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = wave & 3, half = wave >> 2, mi = sub & 1, ni = sub >> 1;
  const int co0 = blockIdx.y * TCO;
  const int plane = p.IPH * p.IPW;
  const int nchunks = (p.Cin + CKW - 1) / CKW;
  // tile t (0 .. 63) of this workgroup -> (image, top-left output pixel): an 8 x 8 block of tiles of image blockIdx.z, or 64 consecutive
  // tiles of the flattened (image, tile row, tile column) index
  auto tile_of = [&](int t, int& img, int& py, int& px) {
    if (p.flat) {
      const int T = blockIdx.x * 64 + t;
      img = T / p.tpi;
      const int r = T - img * p.tpi, ty = r / p.tiles_x;
      py = 2 * ty; px = 2 * (r - ty * p.tiles_x);
      if (T >= p.ntiles) { img = p.N; py = 0; px = 0; }      // beyond the batch: loads return zeros, nothing is stored
    } else {
      const int bx_n = (p.W + 15) >> 4;
      img = blockIdx.z;
      py = (blockIdx.x / bx_n) * 16 + 2 * (t >> 3);
      px = (blockIdx.x % bx_n) * 16 + 2 * (t & 7);
    }
  };
  const int n0 = p.flat ? 0 : blockIdx.z;                      // first image the input descriptor covers
  const int64_t in_floats = (int64_t)(p.N - n0) * p.Cin * plane;
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (int64_t)n0 * p.Cin * plane, 0,
                                                       (int)(in_floats * 4 > 0x7fffffff ? 0x7fffffff : in_floats * 4), RSRC_FLAGS);
  const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U), 0, nchunks * CKW * 16 * p.Cout * 4, RSRC_FLAGS);

  // this thread's (channel, tile) patch of a chunk and its four weight quads
  const int pci = tid >> 6, pt = tid & 63;
  int l_img, l_py, l_px;
  tile_of(pt, l_img, l_py, l_px);
  const int doff = (((l_img - n0) * p.Cin + pci) * plane + l_py * p.IPW + l_px) * 4;
  int uoff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + e * 512, row = idx >> 4, q = idx & 15;
    uoff[e] = (row * p.Cout + co0 + 4 * q) * 4;
  }
  f32x4 dreg[4], ureg[4];
  auto load_chunk = [&](int c) {
    const int cb = c * CKW * plane * 4, ub = c * CKW * 16 * p.Cout * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) dreg[r] = ld4(rs_in, cb + doff + r * p.IPW * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) ureg[e] = ld4(rs_u, ub + uoff[e]);
  };
  auto store_chunk = [&](float* buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 512, row = idx >> 4, q = idx & 15, ci = row >> 4, pp = row & 15;
      *reinterpret_cast<f32x4*>(buf + (pp * CKW + ci) * 64 + 4 * q) = ureg[e];
    }
    float tt[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d0 = dreg[0][c], d1 = dreg[1][c], d2 = dreg[2][c], d3 = dreg[3][c];
      tt[0][c] = d0 - d2; tt[1][c] = d1 + d2; tt[2][c] = d2 - d1; tt[3][c] = d1 - d3;
    }
    float* v = buf + 16 * CKW * 64 + pci * 64 + pt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[((4 * i + 0) * CKW) * 64] = tt[i][0] - tt[i][2];
      v[((4 * i + 1) * CKW) * 64] = tt[i][1] + tt[i][2];
      v[((4 * i + 2) * CKW) * 64] = tt[i][2] - tt[i][1];
      v[((4 * i + 3) * CKW) * 64] = tt[i][1] - tt[i][3];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // The two waves of a SIMD run the chunk's phases in OPPOSITE order, so that one's transform / LDS writes issue under the other's MFMAs
  // (with the same order both stage at the same time, in front of the barrier: ablation -- 11 % of the kernel in the staging, 11 % in the
  // load issue):   half 0:  load(c + 1)   -> MFMA(c) -> stage(c + 1) -> barrier
  //                half 1:  stage(c + 1)  -> load(c + 2) -> MFMA(c)  -> barrier      (its registers run one chunk further ahead)
  load_chunk(0);
  store_chunk(lds);
  if (half && nchunks > 1 && !(ABL & 1)) load_chunk(1);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    float* cur = lds + (c & 1) * BUF;
    float* nxt = lds + ((c + 1) & 1) * BUF;
    const bool more = c + 1 < nchunks;
    const float* ua = cur + (half * 8 * CKW + kh) * 64 + mi * 32 + l32;
    const float* vb = cur + 16 * CKW * 64 + (half * 8 * CKW + kh) * 64 + ni * 32 + l32;
    if (half == 1 && more && !(ABL & 8)) store_chunk(nxt);
    const int lc = c + 1 + half;
    if (lc < nchunks && !(ABL & 1)) load_chunk(lc);
    // (ONE copy of the MFMA block: with a copy in each branch of `half` the accumulators spilled -- 400 bytes of scratch, 5x slower)
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
      for (int ks = 0; ks < CKW / 2; ++ks) {
        const float a = ua[(pp * CKW + 2 * ks) * 64], b = vb[(pp * CKW + 2 * ks) * 64];
        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[pp], 0, 0, 0);
      }
    if (half == 0 && more && !(ABL & 8)) store_chunk(nxt);
    __syncthreads();
  }

  // Output transform.  Rows of M this wave holds: half 0: M0, M1; half 1: M2, M3 (acc[4 * i' + j]).  s0 = M0 + M1 + M2, s1 = M1 - M2 - M3:
  // half 0 finishes output row 0 of every tile (it needs M2 from its partner), half 1 row 1 (it needs M1).  Exchange through LDS:
  // x[r][j][thread of the partner pair], 4 floats per accumulator register.
  float* xch = lds + half * (16 * 4 * 256) ;      // 64 KB per half: [r][j][256 threads of this half]
  const int th = tid & 255;
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) xch[(r * 4 + j) * 256 + th] = half ? acc[j][r] : acc[4 + j][r];      // half 1 sends M2, half 0 sends M1
  __syncthreads();
  const float* got = lds + (half ^ 1) * (16 * 4 * 256);
  int n, oy, ox;
  tile_of(ni * 32 + l32, n, oy, ox);
  oy += half;                                                  // this wave's output row of the tile
  const int64_t oplane = (int64_t)p.OH * p.OW;
  float* ob = p.out + (int64_t)n * p.Cout * oplane;
  if (n < p.N && oy < p.H && ox < p.W) {
    const bool pair = ox + 1 < p.W;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mi * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
      float sj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float o = got[(r * 4 + j) * 256 + th];
        sj[j] = half ? o - acc[j][r] - acc[4 + j][r] : acc[j][r] + acc[4 + j][r] + o;      // s1 = M1 - M2 - M3 | s0 = M0 + M1 + M2
      }
      const float bsv = p.bias ? p.bias[co] : 0.f;
      float v0 = sj[0] + sj[1] + sj[2] + bsv, v1 = sj[1] - sj[2] - sj[3] + bsv;
      const int64_t o = co * oplane + (int64_t)(p.oy0 + oy) * p.OW + p.ox0 + ox;
      if (p.ep_mode == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      if (p.ep_mode == 2) {
        const int64_t e = (int64_t)n * p.Cout * oplane + o;
        float m0, m1 = 0.f, a0 = 0.f, a1 = 0.f;
        if (pair) {       // one 8-byte load per operand and row (dword-aligned is enough)
          const f32x2u m = *reinterpret_cast<const f32x2u*>(p.ep_mask + e);
          m0 = m[0]; m1 = m[1];
          if (p.ep_add) {
            const f32x2u ad = *reinterpret_cast<const f32x2u*>(p.ep_add + e);
            a0 = ad[0]; a1 = ad[1];
          }
        } else {
          m0 = p.ep_mask[e];
          if (p.ep_add) a0 = p.ep_add[e];
        }
        v0 = m0 > 0.f ? v0 + a0 : 0.f;
        v1 = m1 > 0.f ? v1 + a1 : 0.f;
      }
      if (pair) {
        const f32x2u v = {v0, v1};
        *reinterpret_cast<f32x2u*>(ob + o) = v;
      } else {
        ob[o] = v0;
      }
      if (p.ep_mode) {      // padded output: the rows on the rim of the map also store the zero border next to them
        float* q = ob + o;
        const int xlast = pair ? 1 : 0;                        // offset of this tile's last column inside the map
        const bool left = ox == 0, right = ox + xlast == p.W - 1;
        if (left) q[-1] = 0.f;
        if (right) q[xlast + 1] = 0.f;
        const int xa = left ? -1 : 0, xb = right ? xlast + 1 : xlast;
        if (oy == 0)
          for (int xx = xa; xx <= xb; ++xx) q[-p.OW + xx] = 0.f;
        if (oy == p.H - 1)
          for (int xx = xa; xx <= xb; ++xx) q[p.OW + xx] = 0.f;
      }
    }
  }
}

// Weight gradient of the same convolution in Winograd F(3 x 3, 2 x 2) form (round 4): per 2 x 2 tile of the output gradient dY and its
// 4 x 4 input patch X,   dW = sum_tiles A'^T [ (G' dY G'^T) . (B^T X B) ] A',   G' = [[1,0],[.5,.5],[.5,-.5],[0,1]],
// A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]], B^T as in the forward form -- 16 instead of 36 multiplications per tile and channel pair.
// The kernel is the forward kernel with the roles turned: a workgroup owns 64 output x 64 input channels, the GEMM's K dimension is
// the TILE index (8 tiles per chunk), both operands are transformed on the fly (every thread one (output channel, tile) and one (input
// channel, tile) pair per chunk) into Z[p][tile][co] / V[p][tile][ci] in LDS (row pitch 72: the eight tiles of a wave's store land on
// distinct banks), the 16 positions are split between two waves per SIMD running the chunk's phases in opposite order, and the wave
// pairs exchange two rows of the 4 x 4 accumulator block through LDS before the 3 x 3 inverse transform.  Partials [ks][co][ci][9] like
// wgrad3x3_wide_kernel's (reduced by the same wg_wide_reduce_kernel).
struct WinoWgK {
  const float* dout;    // [N][Cout][H][W]
  const float* in;      // pre-padded [N][Cin][H + 2][W + 2]
  float* part;          // [KS][Cout][Cin][9]
  int N, Cin, Cout, H, W, IPW, iplane, oplane;
  int tiles_x, tpi, ntiles, cps, nchunks;   // 2 x 2 tiles per row / image / in all; chunks (of 8 tiles) per K slice / in all
};

constexpr int WG_PITCH = 72;

__global__ __launch_bounds__(512, 2) void wgrad3x3_wino_kernel(const WinoWgK p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // two buffers of (Z [16][8][72], V [16][8][72]); the exchange reuses them
  constexpr int HALF = 16 * 8 * WG_PITCH, BUF = 2 * HALF;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = wave & 3, half = wave >> 2, mi = sub & 1, ni = sub >> 1;
  const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64, ks = blockIdx.z;
  const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dout), 0, (int)((int64_t)p.N * p.Cout * p.oplane * 4), RSRC_FLAGS);
  const auto rs_i = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)((int64_t)p.N * p.Cin * p.iplane * 4), RSRC_FLAGS);

  // this thread's (channel, tile-of-the-chunk) pair on both sides: eight consecutive channels x eight tiles per wave
  const int ch = wave * 8 + (lane & 7), tl = lane >> 3;
  const bool co_ok = co0 + ch < p.Cout, ci_ok = ci0 + ch < p.Cin;
  float dy[2][2];
  f32x4 xr[4];
  auto load_chunk = [&](int c) {
    const int t = c * 8 + tl;
    const int n = t / p.tpi, r = t - n * p.tpi, ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const bool tok = t < p.ntiles;
    const int dbase = ((n * p.Cout + co0 + ch) * p.oplane + 2 * ty * p.W + 2 * tx) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool ok = tok && co_ok && 2 * ty + a < p.H && 2 * tx + b < p.W;
        dy[a][b] = ok ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, dbase + (a * p.W + b) * 4, 0, 0)) : 0.f;
      }
    const int xbase = (tok && ci_ok) ? ((n * p.Cin + ci0 + ch) * p.iplane + 2 * ty * p.IPW + 2 * tx) * 4 : 0x7ffffff0;   // (beyond the buffer: zeros)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) xr[rr] = ld4(rs_i, xbase + rr * p.IPW * 4);
  };
  auto store_chunk = [&](float* buf) {
    // Z = G' dY G'^T
    float t0[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      t0[0][b] = dy[0][b]; t0[1][b] = 0.5f * (dy[0][b] + dy[1][b]); t0[2][b] = 0.5f * (dy[0][b] - dy[1][b]); t0[3][b] = dy[1][b];
    }
    float* z = buf + tl * WG_PITCH + ch;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[((4 * i + 0) * 8) * WG_PITCH] = t0[i][0];
      z[((4 * i + 1) * 8) * WG_PITCH] = 0.5f * (t0[i][0] + t0[i][1]);
      z[((4 * i + 2) * 8) * WG_PITCH] = 0.5f * (t0[i][0] - t0[i][1]);
      z[((4 * i + 3) * 8) * WG_PITCH] = t0[i][1];
    }
    // V = B^T X B
    float tt[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d0 = xr[0][c], d1 = xr[1][c], d2 = xr[2][c], d3 = xr[3][c];
      tt[0][c] = d0 - d2; tt[1][c] = d1 + d2; tt[2][c] = d2 - d1; tt[3][c] = d1 - d3;
    }
    float* v = buf + HALF + tl * WG_PITCH + ch;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[((4 * i + 0) * 8) * WG_PITCH] = tt[i][0] - tt[i][2];
      v[((4 * i + 1) * 8) * WG_PITCH] = tt[i][1] + tt[i][2];
      v[((4 * i + 2) * 8) * WG_PITCH] = tt[i][2] - tt[i][1];
      v[((4 * i + 3) * 8) * WG_PITCH] = tt[i][1] - tt[i][3];
    }
  };

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int c_beg = ks * p.cps, c_end = min(p.nchunks, c_beg + p.cps);
  if (c_beg < c_end) {
    load_chunk(c_beg);
    store_chunk(lds);
    if (half && c_beg + 1 < c_end) load_chunk(c_beg + 1);
  }
  __syncthreads();
  for (int c = c_beg; c < c_end; ++c) {
    float* cur = lds + ((c - c_beg) & 1) * BUF;
    float* nxt = lds + ((c - c_beg + 1) & 1) * BUF;
    const bool more = c + 1 < c_end;
    const float* za = cur + (half * 64 + kh) * WG_PITCH + mi * 32 + l32;             // A[i = co][k = tile]
    const float* vb = cur + HALF + (half * 64 + kh) * WG_PITCH + ni * 32 + l32;      // B[k = tile][j = ci]
    if (half == 1 && more) store_chunk(nxt);
    const int lc = c + 1 + half;
    if (lc < c_end) load_chunk(lc);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float a = za[(pp * 8 + 2 * kk) * WG_PITCH], b = vb[(pp * 8 + 2 * kk) * WG_PITCH];
        acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[pp], 0, 0, 0);
      }
    if (half == 0 && more) store_chunk(nxt);
    __syncthreads();
  }

  // Inverse transform dW = A'^T Q A'.  Rows of Q this wave holds: half 0: Q0, Q1; half 1: Q2, Q3.  s0 = Q0 + Q1 + Q2 (half 0 finishes it: tap
  // row 0), s1 = Q1 - Q2, s2 = Q1 + Q2 - Q3 (half 1: tap rows 1, 2): half 0 sends Q1, half 1 sends Q2.
  float* xch = lds + half * (16 * 4 * 256);
  const int th = tid & 255;
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) xch[(r * 4 + j) * 256 + th] = half ? acc[j][r] : acc[4 + j][r];
  __syncthreads();
  const float* got = lds + (half ^ 1) * (16 * 4 * 256);
  const int ci = ci0 + ni * 32 + l32;
  float* ob = p.part + (int64_t)ks * p.Cout * p.Cin * 9;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + mi * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = got[(r * 4 + j) * 256 + th];
    if (co < p.Cout && ci < p.Cin) {
      float* w = ob + ((int64_t)co * p.Cin + ci) * 9;
      if (half == 0) {
        float s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = acc[j][r] + acc[4 + j][r] + o[j];
        w[0] = s[0] + s[1] + s[2]; w[1] = s[1] - s[2]; w[2] = s[1] + s[2] - s[3];
      } else {
        float s1[4], s2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] = o[j] - acc[j][r];
          s2[j] = o[j] + acc[j][r] - acc[4 + j][r];
        }
        w[3] = s1[0] + s1[1] + s1[2]; w[4] = s1[1] - s1[2]; w[5] = s1[1] + s1[2] - s1[3];
        w[6] = s2[0] + s2[1] + s2[2]; w[7] = s2[1] - s2[2]; w[8] = s2[1] + s2[2] - s2[3];
      }
    }
  }
}

// U[(a * 16 + p) * B + b] = (G g G^T)[p] of the taps g[t] = w[a * sa + b * sb + (flip ? 8 - t : t)], a < A8 (zero rows for a >= A)
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, int A, int A8, int B, int64_t sa, int64_t sb, int flip,
                                                          float* __restrict__ U) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)A8 * B) return;
  const int a = (int)(i / B), b = (int)(i % B);
  float g[3][3];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = a < A ? w[a * sa + b * sb + (flip ? 8 - t : t)] : 0.f;
  float tg[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    tg[0][c] = g[0][c];
    tg[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
    tg[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
    tg[3][c] = g[2][c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u0 = tg[r][0], u1 = 0.5f * (tg[r][0] + tg[r][1] + tg[r][2]), u2 = 0.5f * (tg[r][0] - tg[r][1] + tg[r][2]), u3 = tg[r][2];
    float* o = U + ((int64_t)a * 16 + 4 * r) * B + b;
    o[0] = u0; o[B] = u1; o[2 * (int64_t)B] = u2; o[3 * (int64_t)B] = u3;
  }
}

}  // namespace

extern "C" int64_t vts_w3x3_wino_floats(int A, int B) { return (int64_t)((A + 7) / 8 * 8) * 16 * B; }

extern "C" int vts_w3x3_wino_pack(const float* w, int A, int B, int64_t sa, int64_t sb, int flip, float* U, void* stream) {
  VTS_CHECK_ARG(w && U && A >= 1 && B >= 1, "vts_w3x3_wino_pack: bad args");
  const int A8 = (A + 7) / 8 * 8;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)cdiv64((int64_t)A8 * B, 256)), dim3(256), 0, (hipStream_t)stream, w, A, A8, B, sa, sb, flip, U);
  VTS_CHECK_LAUNCH("vts_w3x3_wino_pack");
  return VTS_OK;
}

// shapes the kernel takes: 64-channel output groups, maps that give the chip enough 16 x 16 blocks, operands inside 31-bit byte offsets
// flat-tile mode: maps of at most 16 x 16 pixels whose batch gives the chip >= 128 workgroups of 64 tiles x 64 channels, the whole input
// inside 31-bit byte offsets (VTS_WINO_FLAT=0: off)
static bool wino_flat(int N, int Cin, int Cout, int H, int W) {
  static const int off = vts_tune("VTS_WINO_FLAT", 1) == 0;
  if (off || H > 16 || W > 16 || (H == 16 && W == 16)) return false;
  if ((int64_t)N * ((Cin + 7) / 8 * 8) * (H + 2) * (W + 2) * 4 > 0x7fffffffll) return false;
  const int64_t tiles = (int64_t)N * cdiv(H, 2) * cdiv(W, 2);
  return cdiv64(tiles, 64) * (Cout / TCO) >= 128;
}

extern "C" int vts_conv3x3_wino_ok(int N, int Cin, int Cout, int H, int W) {
  if (Cout % TCO || Cin < 32) return 0;
  if ((int64_t)((Cin + 7) / 8 * 8) * 16 * Cout * 4 > 0x7fffffff) return 0;
  if (wino_flat(N, Cin, Cout, H, W)) return 1;
  if (H < 8 || W < 8) return 0;
  if ((int64_t)((Cin + 7) / 8 * 8) * (H + 2) * (W + 2) * 4 > 0x7fffffff) return 0;         // byte offsets inside one image
  // a workgroup multiplies a full 16 x 16 block whatever part of it lies inside the map: maps that leave more than ~40 % of their blocks
  // empty (8 x 8 and smaller) are faster on the flattened direct kernel
  if ((int64_t)H * W * 10 < (int64_t)cdiv(W, 16) * 16 * cdiv(H, 16) * 16 * 6) return 0;
  const int64_t wgs = (int64_t)cdiv(W, 16) * cdiv(H, 16) * (Cout / TCO) * N;
  return wgs >= 256;
}

extern "C" int vts_conv3x3_wino(const float* in, const float* U, const float* bias, float* out, int N, int Cin, int Cout, int H, int W, int out_pad,
                                int ep_mode, const float* ep_add, const float* ep_mask, void* stream) {
  VTS_CHECK_ARG(in && U && out && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && (out_pad == 0 || out_pad == 1), "vts_conv3x3_wino: bad args");
  VTS_CHECK_ARG(ep_mode >= 0 && ep_mode <= 2 && (ep_mode == 0 || out_pad == 1) && (ep_mode != 2 || ep_mask), "vts_conv3x3_wino: epilogue mode %d", ep_mode);
  if (!vts_conv3x3_wino_ok(N, Cin, Cout, H, W)) return VTS_ERR_UNSUPPORTED;
  VTS_CHECK_ARG((int64_t)Cin * (H + 2) * (W + 2) * 4 <= 0x7fffffff && N <= 65535, "vts_conv3x3_wino: operand exceeds the 2 GiB buffer range");
  WinoK k{};
  k.in = in; k.U = U; k.bias = bias; k.out = out; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = H; k.W = W;
  k.IPH = H + 2; k.IPW = W + 2; k.OH = H + 2 * out_pad; k.OW = W + 2 * out_pad; k.oy0 = out_pad; k.ox0 = out_pad;
  k.ep_mode = ep_mode; k.ep_add = ep_add; k.ep_mask = ep_mask;
  k.flat = wino_flat(N, Cin, Cout, H, W) ? 1 : 0;
  k.tiles_x = cdiv(W, 2); k.tpi = k.tiles_x * cdiv(H, 2); k.ntiles = N * k.tpi;
  const dim3 grid = k.flat ? dim3(cdiv(k.ntiles, 64), Cout / TCO, 1) : dim3(cdiv(W, 16) * cdiv(H, 16), Cout / TCO, N);
  static const int v1 = vts_tune_set("VTS_WINO_V1") ? 1 : 0;      // the one-wave-per-SIMD kernel (A/B timing)
  if (v1 && !k.flat) {
    hipLaunchKernelGGL(conv3x3_wino_kernel, grid, dim3(256), 0, (hipStream_t)stream, k);
    vts_set_kernel("conv3x3_wino_kernel");
  } else {
    constexpr int LDS_BYTES = 2 * 2 * 16 * CKW * 64 * 4;       // 128 KB: two buffers of the weight and the patch tile
    static const int ablate = vts_tune("VTS_WINO_ABLATE", 0);
    void (*kern)(const WinoK) = ablate == 1 ? conv3x3_wino8_kernel<1> : ablate == 8 ? conv3x3_wino8_kernel<8> : ablate == 9 ? conv3x3_wino8_kernel<9> : conv3x3_wino8_kernel<0>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (attr != hipSuccess) {
      vts_set_error("vts_conv3x3_wino: %d bytes of LDS per workgroup refused: %s", LDS_BYTES, hipGetErrorString(attr));
      return VTS_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), LDS_BYTES, (hipStream_t)stream, k);
    vts_set_kernel("conv3x3_wino8_kernel");
  }
  VTS_CHECK_LAUNCH("vts_conv3x3_wino");
  return VTS_OK;
}

// Internal (vts_conv3x3_wide.hip: vts_wgrad3x3_wide): partials of the stride-1 weight gradient in Winograd form into part [KS][Cout][Cin][9]
// with KS <= max_ks slices; returns the number of slices written, 0 if the shape is not taken.
int vts_wgrad3x3_wino_try(const float* dout, const float* in, float* part, int N, int Cin, int Cout, int H, int W, int max_ks, hipStream_t st) {
  static const int off = vts_tune("VTS_WINO", 1) == 0 ? 1 : (vts_tune("VTS_WINO_WGRAD", 1) == 0 ? 1 : 0);
  if (off || Cin < 64 || Cout < 64 || H < 8 || W < 8 || max_ks < 1) return 0;
  if ((int64_t)N * Cin * (H + 2) * (W + 2) * 4 > 0x7fffffe0ll || (int64_t)N * Cout * H * W * 4 > 0x7fffffe0ll) return 0;
  WinoWgK k{};
  k.dout = dout; k.in = in; k.part = part; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = H; k.W = W;
  k.IPW = W + 2; k.iplane = (H + 2) * (W + 2); k.oplane = H * W;
  k.tiles_x = cdiv(W, 2); k.tpi = k.tiles_x * cdiv(H, 2); k.ntiles = N * k.tpi; k.nchunks = cdiv(k.ntiles, 8);
  const int groups = cdiv(Cout, 64) * cdiv(Cin, 64);
  int KS = cdiv(256, groups);                  // one workgroup per CU (128 KB of LDS each)
  if (KS > max_ks) KS = max_ks;
  if (KS > k.nchunks / 8) KS = k.nchunks / 8;  // >= 8 chunks per slice
  if (KS < 1) KS = 1;
  k.cps = cdiv(k.nchunks, KS);
  KS = cdiv(k.nchunks, k.cps);
  constexpr int LDS_BYTES = 2 * 2 * 16 * 8 * WG_PITCH * 4;      // 147 KB
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (attr != hipSuccess) return 0;
  hipLaunchKernelGGL(wgrad3x3_wino_kernel, dim3(cdiv(Cout, 64), cdiv(Cin, 64), KS), dim3(512), LDS_BYTES, st, k);
  vts_set_kernel("wgrad3x3_wino_kernel");
  return KS;
}

