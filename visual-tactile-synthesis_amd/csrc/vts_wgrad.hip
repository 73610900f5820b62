// Weight gradient of the 4x4 convolution family on the fp32 MFMA path (gfx950).
//
//   dw[cl][ch][ky][kx] = sum_{n,y,x} lo[n,cl,y,x] * hi[n,ch, y*S+ky-pad, x*S+kx-pad]
//
// GEMM view per v_mfma_f32_16x16x4_f32: M = 16 channels of the low-resolution operand,
// N = the 16 taps of ONE high-resolution channel, K = 4 consecutive pixels of a row.
// The reduction dimension (all pixels of the batch) is split over `PW` persistent
// workgroups that each walk a strided list of 4x32-pixel tiles and keep CLT x CHT 16x16
// accumulators per wave in registers; waves of a workgroup own one tile row each and are
// combined through LDS in a fixed order; the PW partials are summed by a second kernel in
// a fixed order, so the result is deterministic (no float atomics).
// Both operands are (dual-source, normalise-on-load) like in vts_conv.hip.
#include "vts_internal.h"

namespace {

struct Src {
  const float *d0, *d1, *sc0, *sh0, *sc1, *sh1;
  int64_t ns0, ns1;
  int C0, C1, C;
  int act;
  float slope;         // activation as t > 0 ? t : slope * t
  const float* ident;  // {1, 0}
};

struct WgK {
  Src lo, hi;
  int N, LH, LW, HH, HW, pad, padx;
  int cl_groups, ch_groups;
  int tiles_y, tiles_x, ntiles;
  float* part;  // [PW][CL][CH][16]
};

// Staging discipline as in vts_conv.hip: every global load is unconditional on a clamped (always valid)
// address and is issued before anything consumes it; normalisation + activation + zero padding are
// applied branch-free afterwards.
__device__ __forceinline__ const float* src_ptr(const Src& s, int n, int c, int y, int H, int W) {
  const int cc = min(c, s.C - 1), yc = min(max(y, 0), H - 1);
  const bool first = cc < s.C0;
  const int cl = first ? cc : cc - s.C0;
  const float* base = first ? s.d0 + n * s.ns0 : s.d1 + n * s.ns1;
  return base + cl * ((int64_t)H * W) + (int64_t)yc * W;
}

__device__ __forceinline__ void src_affine(const Src& s, int n, int c, float& sc, float& sh) {
  const int cc = min(c, s.C - 1);
  const bool first = cc < s.C0;
  const int cl = first ? cc : cc - s.C0;
  const float* scp = first ? s.sc0 : s.sc1;
  const float* shp = first ? s.sh0 : s.sh1;
  const int aidx = n * (first ? s.C0 : s.C1) + cl;
  const bool hsc = scp != nullptr, hsh = shp != nullptr;
  sc = (hsc ? scp : s.ident)[hsc ? aidx : 0];
  sh = (hsh ? shp : s.ident)[hsh ? aidx : 1];
}

__device__ __forceinline__ float finish(float x, float sc, float sh, float slope, bool inside) {
  const float t = fmaf(x, sc, sh);
  const float a = fmaxf(t, 0.f) + slope * fminf(t, 0.f);
  return inside ? a : 0.f;
}

constexpr int TYL = 4;  // tile rows; the tile width TXL is 32, or 8 for maps at most 8 wide (D2 patch passes)

template <int S, int CLT, int CHT, int TXL>
__global__ __launch_bounds__(256) void wgrad4x4_kernel(const WgK p) {
  constexpr int TXLP = TXL + 2;  // 34 = 2 (mod 32) / 10: A reads (16 channels x 2 pixels) hit distinct banks
  constexpr int CLP = CLT * 16;
  constexpr int PRH = (TYL - 1) * S + 4;
  constexpr int PCH = (TXL - 1) * S + 4;
  constexpr int PCHP = (S == 2 && TXL == 32) ? 72 : 40;  // = 8 (mod 32): B reads hit banks 8*ky + kx + S*k
  static_assert(PCHP >= PCH, "pitch");
  constexpr int LO_FLOATS = TYL * CLP * TXLP;
  constexpr int HI_FLOATS = CHT * PRH * PCHP;
  constexpr int RED_FLOATS = CLT * CHT * 256;
  constexpr int AFF_FLOATS = 2 * (CLP + CHT);
  constexpr int STAGE_FLOATS = LO_FLOATS + HI_FLOATS + AFF_FLOATS;
  constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
  __shared__ float lds[LDS_FLOATS];
  float* lo = lds;
  float* hi = lds + LO_FLOATS;
  float* aff_sc = lds + LO_FLOATS + HI_FLOATS;
  float* aff_sh = aff_sc + CLP + CHT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, kq = lane >> 4;
  const int clg = blockIdx.y / p.ch_groups, chg = blockIdx.y - clg * p.ch_groups;
  const int cl0 = clg * CLP, ch0 = chg * CHT;

  f32x4 acc[CLT][CHT];
#pragma unroll
  for (int t = 0; t < CLT; ++t)
#pragma unroll
    for (int h = 0; h < CHT; ++h) acc[t][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Software pipeline over this workgroup's tiles: the raw loads (and per-channel scale/shift) of tile
  // t+1 are issued into registers before the MFMA phase of tile t and consumed after it.
  // All address arithmetic is wave-uniform (SALU) except per-lane column constants computed once per
  // tile: a (row, channel-pair) line of the low-res operand per wave instruction (one channel per
  // half-wave), one patch row of one high-res channel per wave instruction.
  constexpr int NLO = TYL * CLP * TXL / 256;   // low-res elements per thread
  constexpr int REM = PCH % 64;
  constexpr int NCMH = PCH / 64 + ((REM > 16) ? 1 : 0);
  constexpr int PCM = (REM > 16) ? PCH : (PCH / 64) * 64, TW = PCH - PCM;
  constexpr int RPWH = (PRH + 3) / 4;          // high-res patch rows per wave per channel
  constexpr int NHM = CHT * RPWH * NCMH > 0 ? CHT * RPWH * NCMH : 1;
  constexpr int NHT = (CHT * PRH * TW + 255) / 256;
  const int half = lane >> 5, xl = lane & 31;
  float lv[NLO], hv[NHM], ht[NHT > 0 ? NHT : 1];
  float asc = 1.f, ash = 0.f;

  auto decode = [&](int tile, int& n, int& y0, int& x0) {
    n = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - n * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    y0 = ty * TYL;
    x0 = tx * TXL;
  };

  auto load_tile = [&](int tile) {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const int hy0 = y0 * S - p.pad, hx0 = x0 * S - p.padx;
    if (tid < CLP) src_affine(p.lo, n, cl0 + tid, asc, ash);
    else if (tid < CLP + CHT) src_affine(p.hi, n, ch0 + tid - CLP, asc, ash);
    if (TXL == 32) {
      const int lcol = min(x0 + xl, p.LW - 1);
#pragma unroll
      for (int i = 0; i < NLO; ++i) {
        const int ip = wave + 4 * i;                       // uniform
        const int row = (2 * ip) / CLP, cl = 2 * ip - row * CLP;
        const float* pa = src_ptr(p.lo, n, cl0 + cl, y0 + row, p.LH, p.LW);
        const float* pb = src_ptr(p.lo, n, cl0 + cl + 1, y0 + row, p.LH, p.LW);
        lv[i] = (half ? pb : pa)[lcol];
      }
    } else {   // narrow tile: per-lane (row, channel, x) decode, few elements
#pragma unroll
      for (int i = 0; i < NLO; ++i) {
        const int idx = tid + 256 * i;
        const int x = idx % TXL, line = idx / TXL;
        const int row = line / CLP, cl = line - row * CLP;
        lv[i] = src_ptr(p.lo, n, cl0 + cl, y0 + row, p.LH, p.LW)[min(x0 + x, p.LW - 1)];
      }
    }
    if (NCMH > 0) {
#pragma unroll
      for (int h = 0; h < CHT; ++h)
#pragma unroll
        for (int j = 0; j < RPWH; ++j) {
          const float* src = src_ptr(p.hi, n, ch0 + h, hy0 + min(wave + 4 * j, PRH - 1), p.HH, p.HW);
#pragma unroll
          for (int cm = 0; cm < NCMH; ++cm) hv[(h * RPWH + j) * NCMH + cm] = src[min(max(hx0 + cm * 64 + lane, 0), p.HW - 1)];
        }
    }
#pragma unroll
    for (int e = 0; e < NHT; ++e) {
      const int idx = min(tid + e * 256, CHT * PRH * TW - 1);
      const int rr = idx / TW, col = PCM + (idx - rr * TW);
      const int h = rr / PRH, r = rr - h * PRH;
      ht[e] = src_ptr(p.hi, n, ch0 + h, hy0 + r, p.HH, p.HW)[min(max(hx0 + col, 0), p.HW - 1)];
    }
  };

  auto store_tile = [&](int tile) {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const int hy0 = y0 * S - p.pad, hx0 = x0 * S - p.padx;
    if (tid < CLP + CHT) {
      aff_sc[tid] = asc;
      aff_sh[tid] = ash;
    }
    __syncthreads();
    // normalise + activate + pad, write the LDS tiles: lo[row][cl][x], hi[ch][r][col]
    if (TXL == 32) {
      const bool xok = x0 + xl < p.LW;
      float* ldst = lo + half * TXLP + xl;
#pragma unroll
      for (int i = 0; i < NLO; ++i) {
        const int ip = wave + 4 * i;
        const int row = (2 * ip) / CLP, cl = 2 * ip - row * CLP;
        const bool ok = cl0 + cl + half < p.lo.C && y0 + row < p.LH && xok;
        ldst[2 * ip * TXLP] = finish(lv[i], aff_sc[cl + half], aff_sh[cl + half], p.lo.slope, ok);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NLO; ++i) {
        const int idx = tid + 256 * i;
        const int x = idx % TXL, line = idx / TXL;
        const int row = line / CLP, cl = line - row * CLP;
        const bool ok = cl0 + cl < p.lo.C && y0 + row < p.LH && x0 + x < p.LW;
        lo[line * TXLP + x] = finish(lv[i], aff_sc[cl], aff_sh[cl], p.lo.slope, ok);
      }
    }
    if (NCMH > 0) {
#pragma unroll
      for (int h = 0; h < CHT; ++h) {
        const float hsc = aff_sc[CLP + h], hsh = aff_sh[CLP + h];
        const bool cok = ch0 + h < p.hi.C;
#pragma unroll
        for (int j = 0; j < RPWH; ++j) {
          const int r = wave + 4 * j;
          const int iy = hy0 + r;
          const bool rok = cok && iy >= 0 && iy < p.HH;
#pragma unroll
          for (int cm = 0; cm < NCMH; ++cm) {
            const int col = cm * 64 + lane, ix = hx0 + col;
            const float v = finish(hv[(h * RPWH + j) * NCMH + cm], hsc, hsh, p.hi.slope, rok && ix >= 0 && ix < p.HW);
            if (r < PRH && col < PCM) hi[(h * PRH + r) * PCHP + col] = v;
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < NHT; ++e) {
      const int idx = tid + e * 256;
      const int rr = min(idx / TW, CHT * PRH - 1), col = PCM + (idx - (idx / TW) * TW);
      const int h = rr / PRH, r = rr - h * PRH;
      const int iy = hy0 + r, ix = hx0 + col;
      const bool ok = ch0 + h < p.hi.C && iy >= 0 && iy < p.HH && ix >= 0 && ix < p.HW;
      if (idx < CHT * PRH * TW) hi[rr * PCHP + col] = finish(ht[e], aff_sc[CLP + h], aff_sh[CLP + h], p.hi.slope, ok);
    }
    __syncthreads();
  };

  if ((int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    store_tile(tile);
    const int next = tile + gridDim.x;
    if (next < p.ntiles) load_tile(next);
    const float* lrow = lo + (wave * CLP + m16) * TXLP + kq;
    const float* hrow = hi + (wave * S + (m16 >> 2)) * PCHP + kq * S + (m16 & 3);
#pragma unroll
    for (int xs = 0; xs < TXL / 4; ++xs) {
      float a[CLT], b[CHT];
#pragma unroll
      for (int t = 0; t < CLT; ++t) a[t] = lrow[t * 16 * TXLP + xs * 4];
#pragma unroll
      for (int h = 0; h < CHT; ++h) b[h] = hrow[h * PRH * PCHP + xs * 4 * S];
#pragma unroll
      for (int t = 0; t < CLT; ++t)
#pragma unroll
        for (int h = 0; h < CHT; ++h) acc[t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[h], acc[t][h], 0, 0, 0);
    }
    __syncthreads();
  }

  // fixed-order cross-wave reduction through LDS: red[t][h][lane][4]
  float* red = lds;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < CLT; ++t)
#pragma unroll
        for (int h = 0; h < CHT; ++h) {
          f32x4* slot = reinterpret_cast<f32x4*>(red + ((t * CHT + h) * 64 + lane) * 4);
          if (w == 0)
            *slot = acc[t][h];
          else
            *slot = *slot + acc[t][h];
        }
    }
    __syncthreads();
  }
  // D layout: row (cl) = (lane>>4)*4 + reg, col (tap) = lane&15
  const int CL = p.lo.C, CH = p.hi.C;
  float* part = p.part + (int64_t)blockIdx.x * CL * CH * 16;
  for (int idx = tid; idx < CLT * CHT * 256; idx += 256) {
    const int e = idx & 255, th = idx >> 8;
    const int t = th / CHT, h = th - t * CHT;
    const int tap = e & 15, clr = e >> 4;  // clr in 0..15 -> lane>>4 = clr>>2, reg = clr&3
    const int ln = ((clr >> 2) << 4) | tap;
    const float v = red[((t * CHT + h) * 64 + ln) * 4 + (clr & 3)];
    const int cl = cl0 + t * 16 + clr, ch = ch0 + h;
    if (cl < CL && ch < CH) part[((int64_t)cl * CH + ch) * 16 + tap] = v;
  }
}

// Fixed-order sum of the PW partials.  64 consecutive elements per workgroup (one coalesced 256-B
// row per wave-load), 16 waves each summing every 16th partial, combined through LDS in wave order.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, int64_t n, int pw, float* __restrict__ dw,
                                                            int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n)
    for (int k = w; k < pw; k += 16) s += part[(int64_t)k * n + i];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    dw[i] = accumulate ? dw[i] + t : t;
  }
}

struct Plan {
  int clt, cht, cl_groups, ch_groups, pw, tiles_y, tiles_x, ntiles, txl;
};

Plan make_plan(const vts_wgrad_desc* d) {
  Plan pl;
  const int CL = d->lo0.C + (d->lo1.data ? d->lo1.C : 0), CH = d->hi0.C + (d->hi1.data ? d->hi1.C : 0);
  pl.clt = CL <= 16 ? 1 : CL <= 32 ? 2 : CL <= 48 ? 3 : 5;
  pl.cht = (CH <= 4 || CH > 10 || pl.clt == 5) ? 4 : 10;
  pl.cl_groups = cdiv(CL, pl.clt * 16);
  pl.ch_groups = cdiv(CH, pl.cht);
  pl.tiles_y = cdiv(d->LH, TYL);
  pl.txl = d->LW <= 8 ? 8 : 32;
  pl.tiles_x = cdiv(d->LW, pl.txl);
  pl.ntiles = d->N * pl.tiles_y * pl.tiles_x;
  const int groups = pl.cl_groups * pl.ch_groups;
  int pw = 1024 / groups;
  const int64_t nel = (int64_t)CL * CH * 16;
  const int64_t cap = (4 << 20) / nel;  // keep the partial buffer around <= 16 MB
  if (pw > cap) pw = (int)cap;
  if (pw < 1) pw = 1;
  if (pw > pl.ntiles) pw = pl.ntiles;
  pl.pw = pw;
  return pl;
}

void fill_src(Src& s, const vts_operand& a, const vts_operand& b, int act) {
  s.d0 = a.data; s.sc0 = a.scale; s.sh0 = a.shift; s.ns0 = a.nstride; s.C0 = a.C;
  s.d1 = b.data; s.sc1 = b.scale; s.sh1 = b.shift; s.ns1 = b.nstride; s.C1 = b.data ? b.C : 0;
  s.C = s.C0 + s.C1;
  s.act = act;
  s.slope = vts_slope(act);
  s.ident = vts_ident();
}

template <int S, int CLT, int CHT>
void launch_wg(const WgK& k, const Plan& pl, hipStream_t st) {
  dim3 grid(pl.pw, pl.cl_groups * pl.ch_groups);
  if (pl.txl == 8) hipLaunchKernelGGL((wgrad4x4_kernel<S, CLT, CHT, 8>), grid, dim3(256), 0, st, k);
  else hipLaunchKernelGGL((wgrad4x4_kernel<S, CLT, CHT, 32>), grid, dim3(256), 0, st, k);
  vts_set_kernel("wgrad4x4_kernel<%d, %d, %d, %d>", S, CLT, CHT, pl.txl);
}

}  // namespace

extern "C" int64_t vts_wgrad4x4_ws_floats(const vts_wgrad_desc* d) {
  if (!d) return 0;
  const Plan pl = make_plan(d);
  const int64_t CL = d->lo0.C + (d->lo1.data ? d->lo1.C : 0), CH = d->hi0.C + (d->hi1.data ? d->hi1.C : 0);
  return (int64_t)pl.pw * CL * CH * 16;
}

extern "C" int vts_wgrad4x4(const vts_wgrad_desc* d, float* ws, void* stream) {
  VTS_CHECK_ARG(d && d->lo0.data && d->hi0.data && d->dw && ws, "vts_wgrad4x4: null pointer");
  VTS_CHECK_ARG(d->stride == 1 || d->stride == 2, "vts_wgrad4x4: stride %d unsupported", d->stride);
  // lo x hi window semantics as in vts_conv4x4: hi is zero outside its HH x HW extent, pad may be negative
  VTS_CHECK_ARG(d->LH <= d->HH + 16 && d->LW <= d->HW + 16 && d->pad >= -8 && d->pad <= 8,
                "vts_wgrad4x4: lo %dx%d / pad %d implausible for hi %dx%d s%d", d->LH, d->LW, d->pad, d->HH, d->HW, d->stride);
  const Plan pl = make_plan(d);
  WgK k;
  fill_src(k.lo, d->lo0, d->lo1, d->act_lo);
  fill_src(k.hi, d->hi0, d->hi1, d->act_hi);
  k.N = d->N; k.LH = d->LH; k.LW = d->LW; k.HH = d->HH; k.HW = d->HW; k.pad = d->pad; k.padx = d->pad + d->pad_dx;
  k.cl_groups = pl.cl_groups; k.ch_groups = pl.ch_groups;
  k.tiles_y = pl.tiles_y; k.tiles_x = pl.tiles_x; k.ntiles = pl.ntiles;
  k.part = ws;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nel = (int64_t)k.lo.C * k.hi.C * 16;
  // groups that end beyond CL/CH never write their out-of-range rows, and every in-range element is
  // written by exactly one (cl-group, ch-group) workgroup of every pixel worker.
#define WG_CASE(S, CLT, CHT) \
  if (d->stride == S && pl.clt == CLT && pl.cht == CHT) launch_wg<S, CLT, CHT>(k, pl, st);
  WG_CASE(2, 1, 4) WG_CASE(2, 1, 10) WG_CASE(2, 2, 4) WG_CASE(2, 2, 10) WG_CASE(2, 3, 4) WG_CASE(2, 3, 10) WG_CASE(2, 5, 4)
  WG_CASE(1, 1, 4) WG_CASE(1, 1, 10) WG_CASE(1, 2, 4) WG_CASE(1, 2, 10) WG_CASE(1, 3, 4) WG_CASE(1, 3, 10) WG_CASE(1, 5, 4)
#undef WG_CASE
  VTS_CHECK_LAUNCH("vts_wgrad4x4");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(nel, 64)), dim3(1024), 0, st, ws, nel, pl.pw, d->dw,
                     d->accumulate);
  VTS_CHECK_LAUNCH("vts_wgrad4x4 reduce");
  return VTS_OK;
}
