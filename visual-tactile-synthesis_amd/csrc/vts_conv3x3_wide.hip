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
constexpr int PR = TY + 2, PC = TX + 2, PCP = 36;
constexpr int PATCH_FLOATS = CK * PR * PCP;           // 1728
constexpr int W_FLOATS = CK * 9 * TCO;                // 9216
constexpr int PQ_ROW = PCP / 4;                       // 9 quads per patch row
constexpr int NPQ = (CK * PR * PQ_ROW + 255) / 256;   // 2 patch quads per thread
constexpr int NWQ = W_FLOATS / 4 / 256;               // 9 weight quads per thread
constexpr unsigned RSRC_FLAGS = 0x00020000;

struct WideK {
  const float *in, *wt, *bias;
  float* out;
  int N, Cin, Cout, H, W;
  int KS, cps;   // k-split for small grids: blockIdx.z = n + N * slice, cps input-channel chunks per slice
  float* part;   // [KS][N][Cout][H][W] raw partial sums (KS > 1), reduced in slice order by wide_reduce_kernel
};

__global__ __launch_bounds__(256) void conv3x3_wide_kernel(const WideK p) {
  __shared__ __attribute__((aligned(16))) float lds[PATCH_FLOATS + W_FLOATS];
  float* lds_p = lds;
  float* lds_w = lds + PATCH_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wpx = wave >> 1;
  const int tiles_x = (p.W + TX - 1) / TX;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * TX, y0 = ty * TY, co0 = blockIdx.y * TCO, n = blockIdx.z % p.N, ks = blockIdx.z / p.N;
  const int PH = p.H + 2, PW = p.W + 2;
  const int plane = PH * PW;

  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (int64_t)n * p.Cin * plane, 0, p.Cin * plane * 4, RSRC_FLAGS);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.Cin * 9 * p.Cout * 4, RSRC_FLAGS);

  // staging items of this thread (chunk independent): patch quads (ci, row, quad), weight quads (ci*9+tap, co quad)
  int pvoff[NPQ], ploff[NPQ];
#pragma unroll
  for (int e = 0; e < NPQ; ++e) {
    const int q = min(tid + e * 256, CK * PR * PQ_ROW - 1);
    const int row = q / PQ_ROW, cq = q - row * PQ_ROW;      // row = ci * PR + r
    const int ci = row / PR, r = row - ci * PR;
    pvoff[e] = (ci * plane + (y0 + r) * PW + x0 + 4 * cq) * 4;
    ploff[e] = row * PCP + 4 * cq;
  }
  int wvoff[NWQ], wloff[NWQ];
#pragma unroll
  for (int e = 0; e < NWQ; ++e) {
    const int q = tid + e * 256;
    const int row = q >> 5, cq = q & 31;                    // row = ci * 9 + tap
    wvoff[e] = (row * p.Cout + co0 + 4 * cq) * 4;
    wloff[e] = row * TCO + 4 * cq;
  }
  // rows past the image (bottom tiles) lie beyond the descriptor only for the last channel: clamp by masking the
  // store instead -- values computed from them are never written.  Channels past Cin read 0 (bounds check).

  u32x4 pq[NPQ], wq[NWQ];
  auto load_chunk = [&](int c0) {
    const int pbase = c0 * plane * 4, wbase = c0 * 9 * p.Cout * 4;
#pragma unroll
    for (int e = 0; e < NPQ; ++e) pq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, pvoff[e] + pbase, 0, 0);
#pragma unroll
    for (int e = 0; e < NWQ; ++e) wq[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[e] + wbase, 0, 0);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < NPQ; ++e) *reinterpret_cast<u32x4*>(lds_p + ploff[e]) = pq[e];
#pragma unroll
    for (int e = 0; e < NWQ; ++e) *reinterpret_cast<u32x4*>(lds_w + wloff[e]) = wq[e];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* a_base = lds_w + kh * 9 * TCO + wco * 64 + l32;
  const float* b_base = lds_p + kh * PR * PCP + (wpx * 2) * PCP + l32;

  const int nchunks_all = (p.Cin + CK - 1) / CK;
  const int cbeg = ks * p.cps, nchunks = min(nchunks_all, cbeg + p.cps);
  load_chunk(cbeg * CK);
  store_chunk();
  __syncthreads();
  for (int c = cbeg; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) load_chunk((c + 1) * CK);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int kc = 0; kc < CK / 2; ++kc) {
        const float a0 = a_base[(kc * 2 * 9 + tap) * TCO], a1 = a_base[(kc * 2 * 9 + tap) * TCO + 32];
        const float b0 = b_base[kc * 2 * PR * PCP + ky * PCP + kx], b1 = b_base[kc * 2 * PR * PCP + (ky + 1) * PCP + kx];
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
  float* ob = p.part ? p.part + ((int64_t)ks * p.N + n) * p.Cout * p.H * p.W : p.out + (int64_t)n * p.Cout * p.H * p.W;
  const bool add_bias = p.bias && !p.part;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int y = y0 + wpx * 2 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 64 + i * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
        if (co < p.Cout && y < p.H && x < p.W) ob[((int64_t)co * p.H + y) * p.W + x] = acc[i][j][r] + (add_bias ? p.bias[co] : 0.f);
      }
    }
}

__global__ __launch_bounds__(256) void wide_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, int KS,
                                                           int64_t per_slice, int HW, int Cout, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_slice) return;
  float v = bias ? bias[(i / HW) % Cout] : 0.f;
  for (int k = 0; k < KS; ++k) v += part[k * per_slice + i];
  out[i] = v;
}

// w [Cout, Cin, 3, 3]  ->  mode 0: wt[(ci*9 + t) * Cout + co] = w[co][ci][t]                (forward)
//                          mode 1: wt[(co*9 + t) * Cin  + ci] = w[co][ci][8 - t]            (adjoint w.r.t. the input)
__global__ __launch_bounds__(256) void w3x3_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int mode, float* __restrict__ wt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)Cout * Cin * 9;
  if (i >= total) return;
  // i indexes the OUTPUT so that writes are coalesced
  if (mode == 0) {
    const int co = (int)(i % Cout);
    const int64_t r = i / Cout;
    const int t = (int)(r % 9), ci = (int)(r / 9);
    wt[i] = w[((int64_t)co * Cin + ci) * 9 + t];
  } else {
    const int ci = (int)(i % Cin);
    const int64_t r = i / Cin;
    const int t = (int)(r % 9), co = (int)(r / 9);
    wt[i] = w[((int64_t)co * Cin + ci) * 9 + (8 - t)];
  }
}

}  // namespace

extern "C" int vts_w3x3_pack(const float* w, int Cout, int Cin, int mode, float* wt, void* stream) {
  VTS_CHECK_ARG(w && wt && Cout >= 1 && Cin >= 1 && (mode == 0 || mode == 1), "vts_w3x3_pack: bad args");
  const int64_t total = (int64_t)Cout * Cin * 9;
  hipLaunchKernelGGL(w3x3_pack_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, mode, wt);
  VTS_CHECK_LAUNCH("vts_w3x3_pack");
  return VTS_OK;
}

static int wide_plan(int N, int Cin, int Cout, int H, int W, int* cps) {
  const int wgs = cdiv(W, TX) * cdiv(H, TY) * cdiv(Cout, TCO) * N;
  const int nchunks = cdiv(Cin, CK);
  int KS = 1;
  if (wgs < 256) {           // too few tiles for 256 CUs: split the input-channel loop
    KS = 768 / wgs;
    if (KS > nchunks / 4) KS = nchunks / 4;
    if (KS < 1) KS = 1;
  }
  *cps = cdiv(nchunks, KS);
  return cdiv(nchunks, *cps);
}

extern "C" int64_t vts_conv3x3_wide_ws_floats(int N, int Cin, int Cout, int H, int W) {
  int cps;
  const int KS = wide_plan(N, Cin, Cout, H, W, &cps);
  return KS > 1 ? (int64_t)KS * N * Cout * H * W : 0;
}

extern "C" int vts_conv3x3_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int H, int W,
                                float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(in && wt && out && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, "vts_conv3x3_wide: bad args");
  VTS_CHECK_ARG((Cout & 3) == 0, "vts_conv3x3_wide: Cout %d must be a multiple of 4 (16-byte weight rows)", Cout);
  VTS_CHECK_ARG((int64_t)Cin * (H + 2) * (W + 2) * 4 < (1ll << 31) && (int64_t)Cin * 9 * Cout * 4 < (1ll << 31) && N <= 1024,
                "vts_conv3x3_wide: operand exceeds the 2 GiB buffer range");
  int cps;
  int KS = wide_plan(N, Cin, Cout, H, W, &cps);
  const int64_t per_slice = (int64_t)N * Cout * H * W;
  if (KS > 1 && (!ws || ws_floats < KS * per_slice)) { KS = 1; cps = cdiv(Cin, CK); }
  WideK k{in, wt, bias, out, N, Cin, Cout, H, W, KS, cps, KS > 1 ? ws : nullptr};
  const int tiles = cdiv(W, TX) * cdiv(H, TY);
  hipLaunchKernelGGL(conv3x3_wide_kernel, dim3(tiles, cdiv(Cout, TCO), N * KS), dim3(256), 0, (hipStream_t)stream, k);
  vts_set_kernel(KS > 1 ? "conv3x3_wide_kernel+ksplit" : "conv3x3_wide_kernel");
  VTS_CHECK_LAUNCH("vts_conv3x3_wide");
  if (KS > 1) {
    hipLaunchKernelGGL(wide_reduce_kernel, dim3((unsigned)cdiv64(per_slice, 256)), dim3(256), 0, (hipStream_t)stream, ws, bias, KS, per_slice,
                       H * W, Cout, out);
    VTS_CHECK_LAUNCH("vts_conv3x3_wide reduce");
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

constexpr int GCO = 64, GCI = 64, GTY = 2, GTX = 32;
constexpr int GPX = GTY * GTX;                      // 64 pixels per tile
constexpr int DO_PITCH = GPX + 1;                   // dout plane [co][px], odd pitch
constexpr int GPR = GTY + 2, GPC = 36;              // patch rows, staged columns (9 quads)
constexpr int P_PITCH = GPR * GPC + 1;              // in plane [ci][r][c], odd pitch (145)
constexpr int GDO_FLOATS = GCO * DO_PITCH, GP_FLOATS = GCI * P_PITCH;
constexpr int NDQ = GCO * GPX / 4 / 256;            // 4 dout quads per thread
constexpr int NIQ = GCI * GPR * (GPC / 4) / 256;    // 9 patch quads per thread

struct WgWideK {
  const float *dout, *in;
  float* part;     // [KS][Cout][Cin][9]
  int N, Cin, Cout, H, W;
  int tiles_x, tiles_per_img, ntiles, tps;   // pixel tiles; tps = tiles per K slice
};

__global__ __launch_bounds__(256) void wgrad3x3_wide_kernel(const WgWideK p) {
  __shared__ float lds[GDO_FLOATS + GP_FLOATS];
  float* lds_d = lds;
  float* lds_i = lds + GDO_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wci = wave >> 1;
  const int co0 = blockIdx.x * GCO, ci0 = blockIdx.y * GCI, ks = blockIdx.z;
  const int PW = p.W + 2, plane = (p.H + 2) * PW, oplane = p.H * p.W;

  // staging items: dout quads (co, row, xquad) and patch quads (ci, r, cquad); tile offsets are added per tile
  int dvoff[NDQ], dloff[NDQ];
#pragma unroll
  for (int e = 0; e < NDQ; ++e) {
    const int q = tid + e * 256;
    const int co = q / (GPX / 4), r4 = q - co * (GPX / 4);
    const int row = r4 / (GTX / 4), xq = r4 - row * (GTX / 4);
    dvoff[e] = ((co0 + co) * oplane + row * p.W + 4 * xq) * 4;
    dloff[e] = co * DO_PITCH + row * GTX + 4 * xq;
  }
  int ivoff[NIQ], iloff[NIQ];
#pragma unroll
  for (int e = 0; e < NIQ; ++e) {
    const int q = tid + e * 256;
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
    const auto ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (int64_t)n * p.Cin * plane, 0, p.Cin * plane * 4, RSRC_FLAGS);
    const int doff = (cur_y0 * p.W + cur_x0) * 4, ioff = (cur_y0 * PW + cur_x0) * 4;
#pragma unroll
    for (int e = 0; e < NDQ; ++e) dq[e] = __builtin_amdgcn_raw_buffer_load_b128(rd, dvoff[e] + doff, 0, 0);
#pragma unroll
    for (int e = 0; e < NIQ; ++e) iq[e] = __builtin_amdgcn_raw_buffer_load_b128(ri, ivoff[e] + ioff, 0, 0);
  };
  // dout pixels outside the image must contribute 0 (the patch side may hold anything there); channels past
  // Cout / Cin read 0 through the bounds check except where the next image / channel follows: mask by index.
  auto store_tile = [&]() {
#pragma unroll
    for (int e = 0; e < NDQ; ++e) {
      const int q = tid + e * 256;
      const int co = q / (GPX / 4), r4 = q - co * (GPX / 4);
      const int row = r4 / (GTX / 4), xq = r4 - row * (GTX / 4);
      const bool rok = co0 + co < p.Cout && cur_y0 + row < p.H;
      const f32x4 v = __builtin_bit_cast(f32x4, dq[e]);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_d[dloff[e] + j] = (rok && cur_x0 + 4 * xq + j < p.W) ? v[j] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NIQ; ++e) {
      const int q = tid + e * 256;
      const int ci = q / (GPR * (GPC / 4));
      const bool cok = ci0 + ci < p.Cin;
      const f32x4 v = __builtin_bit_cast(f32x4, iq[e]);
#pragma unroll
      for (int j = 0; j < 4; ++j) lds_i[iloff[e] + j] = cok ? v[j] : 0.f;
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* a_base = lds_d + (wco * 32 + l32) * DO_PITCH + kh;            // A[i = co][k = pixel]
  const float* b_base = lds_i + (wci * 32 + l32) * P_PITCH + kh;             // B[k = pixel][j = ci]

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
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap - ky * 3;
          const float b = b_base[(row + ky) * GPC + xs * 2 + kx];
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
  float* ob = p.part + (int64_t)ks * p.Cout * p.Cin * 9;
  const int ci = ci0 + wci * 32 + l32;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + (r >> 2) * 8 + kh * 4 + (r & 3);
      if (co < p.Cout && ci < p.Cin) ob[((int64_t)co * p.Cin + ci) * 9 + tap] = acc[tap][r];
    }
}

__global__ __launch_bounds__(256) void wg_wide_reduce_kernel(const float* __restrict__ part, int KS, int64_t n, float* __restrict__ dw, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = accumulate ? dw[i] : 0.f;
  for (int k = 0; k < KS; ++k) v += part[k * n + i];
  dw[i] = v;
}

int wg_wide_plan(int N, int Cin, int Cout, int H, int W, int* tps) {
  const int ntiles = N * cdiv(H, GTY) * cdiv(W, GTX);
  const int groups = cdiv(Cout, GCO) * cdiv(Cin, GCI);
  int KS = 512 / groups;                      // aim at two workgroups per CU
  if (KS > ntiles / 4) KS = ntiles / 4;
  if (KS < 1) KS = 1;
  *tps = cdiv(ntiles, KS);
  return cdiv(ntiles, *tps);
}

}  // namespace

extern "C" int64_t vts_wgrad3x3_wide_ws_floats(int N, int Cin, int Cout, int H, int W) {
  int tps;
  return (int64_t)wg_wide_plan(N, Cin, Cout, H, W, &tps) * Cout * Cin * 9;
}

extern "C" int vts_wgrad3x3_wide(const float* dout, const float* in, float* dw, int N, int Cin, int Cout, int H, int W, int accumulate,
                                 float* ws, int64_t ws_floats, void* stream) {
  VTS_CHECK_ARG(dout && in && dw && ws && N >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, "vts_wgrad3x3_wide: bad args");
  VTS_CHECK_ARG((int64_t)Cin * (H + 2) * (W + 2) * 4 < (1ll << 31) && (int64_t)Cout * H * W * 4 < (1ll << 31), "vts_wgrad3x3_wide: operand exceeds the 2 GiB buffer range");
  WgWideK k;
  k.dout = dout; k.in = in; k.part = ws; k.N = N; k.Cin = Cin; k.Cout = Cout; k.H = H; k.W = W;
  k.tiles_x = cdiv(W, GTX);
  k.tiles_per_img = k.tiles_x * cdiv(H, GTY);
  k.ntiles = N * k.tiles_per_img;
  const int KS = wg_wide_plan(N, Cin, Cout, H, W, &k.tps);
  const int64_t nel = (int64_t)Cout * Cin * 9;
  VTS_CHECK_ARG(ws_floats >= KS * nel, "vts_wgrad3x3_wide: workspace too small (%lld < %lld floats)", (long long)ws_floats, (long long)(KS * nel));
  hipLaunchKernelGGL(wgrad3x3_wide_kernel, dim3(cdiv(Cout, GCO), cdiv(Cin, GCI), KS), dim3(256), 0, (hipStream_t)stream, k);
  vts_set_kernel("wgrad3x3_wide_kernel");
  VTS_CHECK_LAUNCH("vts_wgrad3x3_wide");
  hipLaunchKernelGGL(wg_wide_reduce_kernel, dim3((unsigned)cdiv64(nel, 256)), dim3(256), 0, (hipStream_t)stream, ws, KS, nel, dw, accumulate);
  VTS_CHECK_LAUNCH("vts_wgrad3x3_wide reduce");
  return VTS_OK;
}
