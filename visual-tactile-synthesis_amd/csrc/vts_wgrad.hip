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
#include <stdio.h>
#include <stdlib.h>

#include "vts_internal.h"

namespace {

struct Src {
  const float *d0, *d1, *sc0, *sh0, *sc1, *sh1;
  int64_t ns0, ns1;
  int C0, C1, C;
  int act;
  float slope;         // activation as t > 0 ? t : slope * t
  const float* ident;  // {1, 0}
  int plain;           // neither source has an affine and there is no activation: values are used as loaded
};

struct WgK {
  Src lo, hi;
  int N, LH, LW, HH, HW, pad, padx;
  int cl_groups, ch_groups;
  int tiles_y, tiles_x, ntiles;
  float* part;  // [PW][CL][CH][16]
  int ablate;   // profiling only (env VTS_ABLATE): 1 skip the global loads, 2 skip the MFMA phase, 4 skip the LDS staging stores
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
      const int rr = idx / (TW > 0 ? TW : 1), col = PCM + (idx - rr * (TW > 0 ? TW : 1));
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
      const int rr = min(idx / (TW > 0 ? TW : 1), CHT * PRH - 1), col = PCM + (idx - (idx / (TW > 0 ? TW : 1)) * (TW > 0 ? TW : 1));
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

// ---- N-split member -------------------------------------------------------------------------------------------------
// The four waves of a workgroup share ONE pixel tile (the GEMM's K) and split the high-resolution channels (the GEMM's N):
// wave w owns channels [w*CHT, (w+1)*CHT) of the workgroup's 4*CHT and all CLT low-resolution channel tiles, so a staged
// low-resolution tile feeds 4x the MFMA work of the K-split kernel above, the low-resolution operand is re-read by 4x fewer
// channel groups, and there is no cross-wave reduction at the end (every wave writes its accumulators to its own slice of
// the partial).  MFMA work of tile rows / column groups beyond the map edge is skipped (wave-uniform loop bounds), so
// ragged maps (129, 130, 257, 513 wide) do not pay for the padding of their last tile.
//
// Staging goes through buffer loads whose descriptor is ONE CHANNEL PLANE (base = plane start, num_records = H*W*4):
//   byte offset = [lane part: (half-wave row * W + x) * 4, or OOB for a column outside the row] + [uniform part: row * W * 4]
// A row above / below the map makes the sum negative (= huge unsigned) / >= num_records, a column outside the row carries the
// OOB sentinel, a channel beyond C gets num_records = 0: the hardware range check returns 0 for all of them (measured on
// gfx950: per dword, soffset included; tools/probes/buffer_oob.hip).  So zero padding costs no instruction, the per-load
// address arithmetic is one VALU add plus a handful of SALU instructions, and an operand without affine / activation (every
// gradient) goes from the load straight to LDS.  The K-split kernel spends ~40 instructions per loaded dword on clamped 64-bit
// addresses -- measured (ablation, profiles/r02_wgrad_ablation.txt): its loads, LDS stores and MFMAs do not overlap and the
// first two cost as much as the MFMAs.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
struct TagT { static constexpr bool value = true; };
struct TagF { static constexpr bool value = false; };
constexpr unsigned OOB_OFF = 0x40000000u;   // byte offset beyond every channel plane (planes are < 2^26 bytes: checked by the host side)

struct PlaneRef {
  rsrc_t rs;
  unsigned nrec;
};

// uniform: descriptor of channel plane c of one sample of a (dual-source) operand; b0 / b1 are the sample bases of the two sources
__device__ __forceinline__ PlaneRef plane_ref(const Src& s, const float* b0, const float* b1, int c, int plane) {
  const bool cok = c < s.C;
  const int cc = cok ? c : 0;
  const bool first = cc < s.C0;
  const float* base = (first ? b0 : b1) + (int64_t)(first ? cc : cc - s.C0) * plane;
  PlaneRef r;
  r.nrec = cok ? (unsigned)plane * 4u : 0u;
  r.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)r.nrec, 0x00020000);
  return r;
}

__device__ __forceinline__ float ld_buf(const rsrc_t& rs, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
}

template <int S, int CLT, int CHT, int TYT = (S == 2 ? 2 : 4)>
__global__ __launch_bounds__(256) void wgrad4x4_ns_kernel(const WgK p) {
  // tile rows (even: a low-resolution wave instruction covers two rows, one per half-wave).  The 2 - 4 channel layers (one
  // high-resolution channel per wave) take 8 rows: a 2-row tile gives them 14 MFMAs per wave between two barriers -- the per-tile
  // decode / descriptor / barrier overhead and the unoverlapped load issue were 2/3 of the kernel (ablation, round 3) -- and a taller
  // tile also re-reads less halo (18 patch rows per 8 output rows instead of 6 per 2).
  constexpr int TY = TYT;
  constexpr int TX = 28, KSTEPS = TX / 4;     // 28 pixels: the high-resolution patch row (58 / 31 columns) fits one wave / half-wave
  constexpr int TXLP = 30;                    // A reads: bank = (-2 * channel + k) mod 32 -> conflict-free
  constexpr int CLP = CLT * 16, CHW = 4 * CHT;
  constexpr int PRH = (TY - 1) * S + 4;       // 6 / 7 patch rows
  constexpr int PCH = (TX - 1) * S + 4;       // 58 / 31 patch columns
  constexpr int PCHP = S == 2 ? 72 : 40;      // = 8 (mod 32): B reads hit banks 8*ky + kx + S*k
  constexpr int HRP = S == 2 ? 1 : 2;         // patch rows per wave instruction
  constexpr int HLPC = (PRH + HRP - 1) / HRP; // wave instructions per high-resolution channel: 6 / 4
  constexpr int NLO = (TY / 2) * CLP / 4;     // low-resolution (row pair, channel) lines per wave
  constexpr int NHL = (CHW * HLPC + 3) / 4;   // high-resolution lines per wave
  constexpr int LO_FLOATS = TY * CLP * TXLP;
  constexpr int HI_FLOATS = CHW * PRH * PCHP;
  __shared__ float lds[LO_FLOATS + HI_FLOATS + 2 * (CLP + CHW)];
  float* lo = lds;
  float* hi = lds + LO_FLOATS;
  float* aff_sc = hi + HI_FLOATS;
  float* aff_sh = aff_sc + CLP + CHW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, kq = lane >> 4;
  const int half = lane >> 5, xl = lane & 31;
  const int clg = blockIdx.y / p.ch_groups, chg = blockIdx.y - clg * p.ch_groups;
  const int cl0 = clg * CLP, ch0 = chg * CHW;
  const int lplane = p.LH * p.LW, hplane = p.HH * p.HW;
  const bool lo_plain = p.lo.plain != 0, hi_plain = p.hi.plain != 0;   // uniform: no affine, no activation

  f32x4 acc[CLT][CHT];
#pragma unroll
  for (int t = 0; t < CLT; ++t)
#pragma unroll
    for (int h = 0; h < CHT; ++h) acc[t][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float lv[NLO], hv[NHL];
#pragma unroll
  for (int i = 0; i < NLO; ++i) lv[i] = 1.f;
#pragma unroll
  for (int i = 0; i < NHL; ++i) hv[i] = 1.f;
  float asc = 1.f, ash = 0.f;

  struct Tile {
    int n, y0, x0;
    unsigned lo_vo, hi_vo;   // lane parts of the byte offsets
  };
  auto decode = [&](int tile) {
    Tile t;
    t.n = tile / (p.tiles_y * p.tiles_x);
    const int rem = tile - t.n * (p.tiles_y * p.tiles_x);
    const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
    t.y0 = ty * TY;
    t.x0 = tx * TX;
    const int x = t.x0 + xl;
    t.lo_vo = (xl < TX && x < p.LW) ? (unsigned)(half * p.LW + x) * 4u : OOB_OFF;
    const int col = HRP == 2 ? xl : lane, ix = t.x0 * S - p.padx + col;
    t.hi_vo = (col < PCH && ix >= 0 && ix < p.HW) ? (unsigned)((HRP == 2 ? half * p.HW : 0) + ix) * 4u : OOB_OFF;
    return t;
  };
  // Each wave stages whole channels (CLP/4 low-resolution, CHT high-resolution ones), so one plane descriptor serves all
  // lines of a channel.  The channel index carries an opaque zero that is re-read per tile: without it LLVM hoists the
  // (tile-invariant) per-line descriptors out of the tile loop and spills ~200 SGPRs through v_writelane / v_readlane.
  auto opaque_zero = []() {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
  };
  constexpr int LCW = CLP / 4;       // low-resolution channels per wave
  constexpr int LRP = TY / 2;        // row pairs per tile
  static_assert(NLO == LCW * LRP && NHL == CHT * HLPC, "line counts");

  auto load_tile = [&](const Tile& t) {
    if (tid < CLP) src_affine(p.lo, t.n, cl0 + tid, asc, ash);
    else if (tid < CLP + CHW) src_affine(p.hi, t.n, ch0 + tid - CLP, asc, ash);
    const int z = opaque_zero();
    const float* lb0 = p.lo.d0 + t.n * p.lo.ns0;
    const float* lb1 = p.lo.d1 + t.n * p.lo.ns1;
    const float* hb0 = p.hi.d0 + t.n * p.hi.ns0;
    const float* hb1 = p.hi.d1 + t.n * p.hi.ns1;
    const unsigned lrow0 = (unsigned)(t.y0 * p.LW * 4), lstep = (unsigned)(2 * p.LW * 4);
#pragma unroll
    for (int j = 0; j < LCW; ++j) {
      const PlaneRef r = plane_ref(p.lo, lb0, lb1, cl0 + wave + 4 * j + z, lplane);
#pragma unroll
      for (int rp = 0; rp < LRP; ++rp) lv[j * LRP + rp] = ld_buf(r.rs, t.lo_vo + lrow0 + rp * lstep);
    }
    const unsigned hrow0 = (unsigned)((t.y0 * S - p.pad) * p.HW * 4), hstep = (unsigned)(HRP * p.HW * 4);
#pragma unroll
    for (int j = 0; j < CHT; ++j) {
      const PlaneRef r = plane_ref(p.hi, hb0, hb1, ch0 + wave * CHT + j + z, hplane);
#pragma unroll
      for (int q = 0; q < HLPC; ++q) hv[j * HLPC + q] = ld_buf(r.rs, t.hi_vo + hrow0 + q * hstep);
    }
  };

  // LDS stores are unconditional: lanes beyond the tile / patch width write the pad column of their line, and the half-wave
  // whose patch row does not exist (odd PRH, last row pair) writes the pad column of row 0 (no exec-mask branch per store).
  auto store_lo = [&](const Tile& t, int z, auto plain_tag) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    float* dst = lo + half * CLP * TXLP + min(xl, TX);
    const unsigned lrow0 = (unsigned)(t.y0 * p.LW * 4), lstep = (unsigned)(2 * p.LW * 4);
#pragma unroll
    for (int j = 0; j < LCW; ++j) {
      const int cl = wave + 4 * j + z;
      const unsigned nrec = cl0 + cl < p.lo.C ? (unsigned)lplane * 4u : 0u;
      float sc = 1.f, sh = 0.f;
      if (!PLAIN) {
        sc = aff_sc[cl];
        sh = aff_sh[cl];
      }
#pragma unroll
      for (int rp = 0; rp < LRP; ++rp) {
        float v = lv[j * LRP + rp];
        if (!PLAIN) v = finish(v, sc, sh, p.lo.slope, t.lo_vo + lrow0 + rp * lstep < nrec);
        dst[(2 * rp * CLP + cl) * TXLP] = v;
      }
    }
  };
  auto store_hi = [&](const Tile& t, int z, auto plain_tag) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    const int col = min(HRP == 2 ? xl : lane, PCH);
    float* dst = hi + (HRP == 2 ? half * PCHP : 0) + col;
    float* dst_last = (HRP == 2 && (PRH & 1) && half) ? hi + PCH - (HLPC - 1) * HRP * PCHP : dst;
    const unsigned hrow0 = (unsigned)((t.y0 * S - p.pad) * p.HW * 4), hstep = (unsigned)(HRP * p.HW * 4);
#pragma unroll
    for (int j = 0; j < CHT; ++j) {
      const int h = wave * CHT + j + z;
      const unsigned nrec = ch0 + h < p.hi.C ? (unsigned)hplane * 4u : 0u;
      float sc = 1.f, sh = 0.f;
      if (!PLAIN) {
        sc = aff_sc[CLP + h];
        sh = aff_sh[CLP + h];
      }
#pragma unroll
      for (int q = 0; q < HLPC; ++q) {
        float v = hv[j * HLPC + q];
        if (!PLAIN) v = finish(v, sc, sh, p.hi.slope, t.hi_vo + hrow0 + q * hstep < nrec);
        (q == HLPC - 1 ? dst_last : dst)[(h * PRH + q * HRP) * PCHP] = v;
      }
    }
  };

  auto store_tile = [&](const Tile& t) {
    if (tid < CLP + CHW) {
      aff_sc[tid] = asc;
      aff_sh[tid] = ash;
    }
    __syncthreads();
    const int z = opaque_zero();
    if (lo_plain) store_lo(t, z, TagT());
    else store_lo(t, z, TagF());
    if (hi_plain) store_hi(t, z, TagT());
    else store_hi(t, z, TagF());
    __syncthreads();
  };

  Tile cur = decode(min((int)blockIdx.x, p.ntiles - 1));
  if (!(p.ablate & 1)) load_tile(cur);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    if (!(p.ablate & 4)) store_tile(cur);
    const int y0 = cur.y0, x0 = cur.x0;
    const int next = tile + gridDim.x;
    if (next < p.ntiles) {
      cur = decode(next);
      if (!(p.ablate & 1)) load_tile(cur);
    }
    const int nrow = (p.ablate & 2) ? 0 : min(TY, p.LH - y0), nxs = min(KSTEPS, (p.LW - x0 + 3) >> 2);   // uniform: skip the padding of edge tiles
    for (int row = 0; row < nrow; ++row) {
      const float* lrow = lo + (row * CLP + m16) * TXLP + kq;
      const float* hrow = hi + (wave * CHT * PRH + row * S + (m16 >> 2)) * PCHP + kq * S + (m16 & 3);
      for (int xs = 0; xs < nxs; ++xs) {
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
    }
    // (round 4, measured and dropped: a two-set register pipeline of the fragment reads -- reads of k-step j + 1 issued before the MFMAs
    //  of step j, unconditional and clamped, scheduling barriers -- was 3 - 8 % SLOWER on every shape of tools/mb_wgrad.py and + 0.1 ms on
    //  the step: the co-resident workgroups already cover the LDS latency, the extra registers and scalar bookkeeping only cost.)
    __syncthreads();
  }

  // D layout of a 16x16 tile: row (cl) = (lane>>4)*4 + reg, col (tap) = lane&15 -> 64-byte runs per (cl, ch)
  const int CL = p.lo.C, CH = p.hi.C;
  float* part = p.part + (int64_t)blockIdx.x * CL * CH * 16;
#pragma unroll
  for (int t = 0; t < CLT; ++t)
#pragma unroll
    for (int h = 0; h < CHT; ++h) {
      const int ch = ch0 + wave * CHT + h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = cl0 + t * 16 + kq * 4 + r;
        if (cl < CL && ch < CH) part[((int64_t)cl * CH + ch) * 16 + m16] = acc[t][h][r];
      }
    }
}

// ---- small-map member (round 2): the D2 discriminator's weight gradients over hundreds of 32x32 patches whose maps shrink to
// 2x2 .. 17x17.  Tiling single images (the kernels above) leaves most of every 4x8 pixel tile empty and pays the per-tile staging
// overhead per image (80 us for 64 x 32 channels on 640 maps of 6x6: 19 TFLOP/s); here the reduction dimension of the GEMM is the
// FLATTENED (image, y, x) position index of IPB whole images per workgroup step: both operands of those images are staged once
// (normalise + activate on load; the high-res planes zero-haloed so that taps outside the map read zeros), a position table gives
// the plane offset of every position's window, and the four waves split the high-res channels while each holds all low-res
// channel tiles: per group of four positions CLT + CHW LDS reads feed CLT x CHW MFMAs.  A workgroup keeps the whole CL x CH x 16
// result in registers across its image blocks and writes ONE partial copy (reduced by wgrad_reduce_batch_kernel as for the others).
struct SmallWK {
  const float *lo, *losc, *losh, *hi, *hisc, *hish;
  int64_t lons, hins;
  int CL, CH, N, LH, LW, HH, HW, S, pad, padx;
  float lo_slope, hi_slope;
  int IPB, nblocks, POS, POSP, PW_, PLANE;   // positions per block (multiple of 4), lo row pitch, hi row pitch / plane (floats)
  float* part;
};
constexpr int SW_HALO = 2;

// e / d for 0 <= e < 2^23 through the float reciprocal (off by at most one: fixed up); ~8 instructions instead of the ~40 of a 32-bit division
__device__ __forceinline__ int fdiv(int e, int d, float inv, int& rem) {
  int q = (int)((float)e * inv);
  int r = e - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
  rem = r;
  return q;
}

template <int CLT, int CHW>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const SmallWK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* lo_t = smem;                                   // [CLT*16][POSP]
  float* hi_t = lo_t + CLT * 16 * p.POSP;               // [IPB][CH][PLANE]
  int* postab = reinterpret_cast<int*>(hi_t + p.IPB * p.CH * p.PLANE);   // [POS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, kq = lane >> 4;
  const int lhw = p.LH * p.LW, hhw = p.HH * p.HW;
  const int imgstride = p.CH * p.PLANE;
  const float inv_blk = 1.f / (float)(p.IPB * lhw), inv_lhw = 1.f / (float)lhw, inv_hhw = 1.f / (float)hhw, inv_ch = 1.f / (float)p.CH,
              inv_hw = 1.f / (float)p.HW;
  // zero everything once: halos, padded positions and absent channel rows stay zero
  for (int i = tid; i < CLT * 16 * p.POSP + p.IPB * imgstride; i += 256) smem[i] = 0.f;
  for (int q = tid; q < p.POS; q += 256) {
    const int img = q / lhw, rem = q - img * lhw;
    const int y = rem / p.LW, x = rem - y * p.LW;
    postab[q] = img < p.IPB ? img * imgstride + (y * p.S - p.pad + SW_HALO) * p.PW_ + (x * p.S - p.padx + SW_HALO) : 0;
  }
  const int tapoff = (m16 >> 2) * p.PW_ + (m16 & 3);     // B fragment: column n = tap (ky, kx)
  f32x4 acc[CLT][CHW];
#pragma unroll
  for (int t = 0; t < CLT; ++t)
#pragma unroll
    for (int c = 0; c < CHW; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  for (int blk = blockIdx.x; blk < p.nblocks; blk += gridDim.x) {
    const int n0 = blk * p.IPB;
    const int nimg = min(p.IPB, p.N - n0);
    // ---- stage lo [cl][pos] (positions of absent images: zero) and the interior of the hi planes
    // (batches of 8 elements per thread: the global loads of a batch are all in flight before the first LDS store)
    const int lo_total = p.CL * p.IPB * lhw;
    for (int e0 = tid; e0 < lo_total; e0 += 256 * 8) {
      float raw[8], fsc[8], fsh[8];
      int dst[8];
      bool live[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = min(e0 + j * 256, lo_total - 1);
        int q, r;
        const int c = fdiv(e, p.IPB * lhw, inv_blk, q);
        const int img = fdiv(q, lhw, inv_lhw, r);
        const int n = n0 + min(img, nimg - 1);
        raw[j] = p.lo[n * p.lons + (int64_t)c * lhw + r];
        fsc[j] = p.losc ? p.losc[n * p.CL + c] : 1.f;
        fsh[j] = p.losh ? p.losh[n * p.CL + c] : 0.f;
        dst[j] = e0 + j * 256 < lo_total ? c * p.POSP + q : -1;
        live[j] = img < nimg;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = fmaf(raw[j], fsc[j], fsh[j]);
        if (dst[j] >= 0) lo_t[dst[j]] = live[j] ? fmaxf(t, 0.f) + p.lo_slope * fminf(t, 0.f) : 0.f;
      }
    }
    const int hi_total = nimg * p.CH * hhw;
    for (int e0 = tid; e0 < hi_total; e0 += 256 * 8) {
      float raw[8], fsc[8], fsh[8];
      int dst[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = min(e0 + j * 256, hi_total - 1);
        int r, c, x;
        const int ic = fdiv(e, hhw, inv_hhw, r);
        const int img = fdiv(ic, p.CH, inv_ch, c);
        const int y = fdiv(r, p.HW, inv_hw, x);
        const int n = n0 + img;
        raw[j] = p.hi[n * p.hins + (int64_t)c * hhw + r];
        fsc[j] = p.hisc ? p.hisc[n * p.CH + c] : 1.f;
        fsh[j] = p.hish ? p.hish[n * p.CH + c] : 0.f;
        dst[j] = e0 + j * 256 < hi_total ? img * imgstride + c * p.PLANE + (y + SW_HALO) * p.PW_ + x + SW_HALO : -1;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = fmaf(raw[j], fsc[j], fsh[j]);
        if (dst[j] >= 0) hi_t[dst[j]] = fmaxf(t, 0.f) + p.hi_slope * fminf(t, 0.f);
      }
    }
    __syncthreads();
    // ---- MFMA over groups of four positions, UG groups per iteration: all LDS reads of an iteration are issued before its MFMAs (one
    // wave per SIMD at this LDS footprint: nothing else hides the read latency)
    const int ngroups = (nimg * lhw + 3) >> 2;
    constexpr int UG = (CLT * CHW >= 16) ? 2 : 4;
    for (int g0 = 0; g0 < ngroups; g0 += UG) {
      float a[UG][CLT], b[UG][CHW];
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const int pos = min(g0 + u, ngroups - 1) * 4 + kq;
#pragma unroll
        for (int t = 0; t < CLT; ++t) a[u][t] = lo_t[(t * 16 + m16) * p.POSP + pos];
        const int base = postab[pos] + tapoff;
#pragma unroll
        for (int c = 0; c < CHW; ++c) b[u][c] = hi_t[base + min(wave * CHW + c, p.CH - 1) * p.PLANE];
      }
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        if (g0 + u < ngroups) {
#pragma unroll
          for (int t = 0; t < CLT; ++t)
#pragma unroll
            for (int c = 0; c < CHW; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], b[u][c], acc[t][c], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- this workgroup's partial copy [CL][CH][16]: C/D layout col = lane & 15 (tap), row = (lane >> 4) * 4 + r (low-res channel)
  float* part = p.part + (int64_t)blockIdx.x * p.CL * p.CH * 16;
#pragma unroll
  for (int t = 0; t < CLT; ++t)
#pragma unroll
    for (int c = 0; c < CHW; ++c) {
      const int ch = wave * CHW + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = t * 16 + kq * 4 + r;
        if (cl < p.CL && ch < p.CH) part[((int64_t)cl * p.CH + ch) * 16 + m16] = acc[t][c][r];
      }
    }
}

// ---- single-channel member (round 3): the weight gradient of the PatchGAN prediction heads (Conv2d(ndf*8, 1, 4, 1, 2), reference
// models/networks.py:1739-1741): lo is the ONE-channel output gradient, hi the 64-channel input.  On the MFMA tiles fifteen of
// the sixteen low-resolution channel rows are zeros (57 us for 8 x 64 x 130^2, against 4.4 us of HBM time); this is a reduction
// dw[ch][ky][kx] = sum_p lo[p] * hi[ch][p + (ky, kx)], so it runs on the vector ALUs: a workgroup stages a 16 x 32 tile of lo and
// the 19 x 35 patch of 16 hi channels (normalise + activate + zero padding on the way), a thread owns (channel, ky) and every
// fourth tile row: per row nine + eight 16-byte LDS reads (conflict-free: lanes of one (channel) differ in row = bank group, equal
// rows broadcast) feed 128 FMAs for its four kx.  One partial copy [CH][16] per tile, reduced by the batched reduction.
struct HeadWK {
  const float* lo;
  int64_t lons;
  const float *hi, *hisc, *hish;
  int64_t hins;
  int CH, N, LH, LW, HH, HW, pad, padx;
  float hi_slope;
  int tiles_y, tiles_x;
  float* part;
};
constexpr int HWG_TY = 16, HWG_TX = 32, HWG_PR = HWG_TY + 3, HWG_PC = HWG_TX + 3, HWG_PCP = 36, HWG_CK = 16, HWG_PLANE = HWG_PR * HWG_PCP;

__global__ __launch_bounds__(256) void wgrad_head_kernel(const HeadWK p) {
  __shared__ __attribute__((aligned(16))) float patch[HWG_CK * HWG_PLANE];
  __shared__ __attribute__((aligned(16))) float dyt[HWG_TY * HWG_PCP];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int n = tile / (p.tiles_y * p.tiles_x), rem = tile - n * (p.tiles_y * p.tiles_x);
  const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
  const int y0 = ty * HWG_TY, x0 = tx * HWG_TX;
  const int hy0 = y0 - p.pad, hx0 = x0 - p.padx;
  // the lo tile (zeros beyond the map: those positions then contribute nothing)
  for (int e = tid; e < HWG_TY * HWG_TX; e += 256) {
    const int r = e >> 5, c = e & 31;
    const int y = y0 + r, x = x0 + c;
    dyt[r * HWG_PCP + c] = (y < p.LH && x < p.LW) ? p.lo[n * p.lons + (int64_t)y * p.LW + x] : 0.f;
  }
  const int rg = tid & 3, ky = (tid >> 2) & 3, chl = tid >> 4;
  const int64_t hplane = (int64_t)p.HH * p.HW;
  float* part = p.part + (int64_t)tile * p.CH * 16;
  // Software pipeline over the 16-channel chunks: the raw loads of chunk k + 1 are issued before the multiply phase of chunk k and
  // finished (affine, activation, padding -> LDS) after it.  Row-wise staging: wave w takes channels w, w + 4, ... of the chunk, one
  // patch row per wave instruction (lanes = columns), so channel, row, bounds and the channel's scale / shift are wave-uniform
  // (the element-wise form spent ~45 instructions per loaded value); all 76 loads of a chunk are in flight together.
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ix = hx0 + lane;
  const bool colok = lane < HWG_PC && ix >= 0 && ix < p.HW;
  const int ixc = min(max(ix, 0), p.HW - 1);
  float raw[HWG_CK / 4][HWG_PR];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int cc = 0; cc < HWG_CK / 4; ++cc) {
      const int chc = min(c0 + wv + 4 * cc, p.CH - 1);
      const float* src = p.hi + n * p.hins + chc * hplane + ixc;
#pragma unroll
      for (int r = 0; r < HWG_PR; ++r) raw[cc][r] = src[(int64_t)min(max(hy0 + r, 0), p.HH - 1) * p.HW];
    }
  };
  auto store_chunk = [&](int c0) {
#pragma unroll
    for (int cc = 0; cc < HWG_CK / 4; ++cc) {
      const int c = wv + 4 * cc, ch = c0 + c;
      const bool chok = ch < p.CH;
      const int chc = chok ? ch : p.CH - 1;
      const float sc = p.hisc ? p.hisc[n * p.CH + chc] : 1.f, sh = p.hish ? p.hish[n * p.CH + chc] : 0.f;
      float* dstp = patch + c * HWG_PLANE + min(lane, HWG_PCP - 1);
#pragma unroll
      for (int r = 0; r < HWG_PR; ++r) {
        const int iy = hy0 + r;
        const bool ok = chok && colok && iy >= 0 && iy < p.HH;
        const float t = fmaf(raw[cc][r], sc, sh);
        if (lane < HWG_PCP) dstp[r * HWG_PCP] = ok ? fmaxf(t, 0.f) + p.hi_slope * fminf(t, 0.f) : 0.f;
      }
    }
  };
  load_chunk(0);
  for (int c0 = 0; c0 < p.CH; c0 += HWG_CK) {
    __syncthreads();                      // the previous chunk's readers are done (first pass: the lo tile is complete)
    store_chunk(c0);
    if (c0 + HWG_CK < p.CH) load_chunk(c0 + HWG_CK);
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HWG_TY / 4; ++j) {
      const int r = rg + 4 * j;
      const f32x4* vp = reinterpret_cast<const f32x4*>(patch + chl * HWG_PLANE + (r + ky) * HWG_PCP);
      const f32x4* dp = reinterpret_cast<const f32x4*>(dyt + r * HWG_PCP);
      float v[HWG_PCP], d[HWG_TX];
#pragma unroll
      for (int i = 0; i < HWG_PCP / 4; ++i) {
        const f32x4 t = vp[i];
        v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
      }
#pragma unroll
      for (int i = 0; i < HWG_TX / 4; ++i) {
        const f32x4 t = dp[i];
        d[4 * i] = t[0]; d[4 * i + 1] = t[1]; d[4 * i + 2] = t[2]; d[4 * i + 3] = t[3];
      }
#pragma unroll
      for (int x = 0; x < HWG_TX; ++x)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) acc[kx] = fmaf(d[x], v[x + kx], acc[kx]);
    }
    // the four row groups of a (channel, ky) sit in adjacent lanes: fixed-order combination (0 + 1) + (2 + 3)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      const float a = acc[kx] + __shfl_xor(acc[kx], 1);
      acc[kx] = a + __shfl_xor(a, 2);
    }
    if (rg == 0 && c0 + chl < p.CH) *reinterpret_cast<f32x4*>(part + (c0 + chl) * 16 + ky * 4) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
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

// Batched form: blockIdx.x -> (job, 64-element block) through a table in the kernel arguments.
constexpr int RB_JOBS = 40;
struct ReduceTable {
  int njobs;
  int blk_start[RB_JOBS + 1];
  unsigned char narrow[RB_JOBS];   // 1: 64 elements per workgroup, the lanes of a wave also split the copies (many copies of a small dw)
  vts_reduce_job job[RB_JOBS];
};

// 256 elements per workgroup (round 2; was 64 elements by 1024 threads with one 4-byte load in flight per thread: 43 us per launch):
// a lane owns four elements -- one 16-byte load per partial copy when the copies are 16-byte aligned (they are whenever nel % 4 == 0:
// every 4x4 weight tensor), four 4-byte loads otherwise -- the sixteen waves take every sixteenth copy with eight independent
// accumulators (eight loads in flight per lane: thin layers have ~1000 copies of a few hundred elements, the chain of dependent
// loads is what bounds them), and everything is combined in a fixed order: pairwise over a0..a7 per wave, then waves 0..15.
// Deterministic, no float atomics.
constexpr int RB_ELEMS = 256;
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(const ReduceTable t) {
  __shared__ f32x4 red[16][64];
  int lo = 0, hi = t.njobs - 1;   // uniform binary search: last job with blk_start <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.blk_start[mid] <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const vts_reduce_job& j = t.job[lo];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (t.narrow[lo]) {
    // Thin layers have 500 .. 1500 copies of a few thousand elements: with 256 elements per workgroup six workgroups would each walk
    // ~100 copies per wave (a chain of a dozen dependent load rounds).  Here a workgroup owns 64 elements, 16 lanes x 16 bytes, and
    // the four 16-lane groups of each of its 16 waves take every 64th copy: 4x the workgroups, a quarter of the chain.  Combination
    // order is fixed: accumulators pairwise, lane groups (0+1)+(2+3), then waves 0..15.
    const int el = lane & 15, cg = lane >> 4;
    const int64_t nb = (int64_t)(blockIdx.x - t.blk_start[lo]) * 64;
    f32x4 sn = {0.f, 0.f, 0.f, 0.f};
    for (int sg = 0; sg < j.nseg; ++sg) {
      const float* part = j.part[sg];
      const int pw = j.pw[sg];
      const bool vec = nb + 64 <= j.nel && (j.nel & 3) == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
      f32x4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int k = w * 4 + cg; k < pw; k += 512) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int kk = k + 64 * u;
          if (kk < pw) {
            const float* row = part + (int64_t)kk * j.nel + nb;
            f32x4 v;
            if (vec) {
              v = *reinterpret_cast<const f32x4*>(row + el * 4);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (nb + el * 4 + e < j.nel) ? row[el * 4 + e] : 0.f;
            }
            a[u] += v;
          }
        }
      }
      sn += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p1 = sn[e] + __shfl_xor(sn[e], 16);       // groups (0 + 1) and (2 + 3): the same value in both lanes of a pair
      o[e] = p1 + __shfl_xor(p1, 32);                       // (0 + 1) + (2 + 3) == (2 + 3) + (0 + 1): fp addition commutes
    }
    if (cg == 0) red[w][el] = o;
    __syncthreads();
    if (w == 0 && lane < 16) {
      f32x4 v = red[0][lane];
#pragma unroll
      for (int k = 1; k < 16; ++k) v += red[k][lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t i = nb + lane * 4 + e;
        if (i < j.nel) j.dw[i] = j.accumulate ? j.dw[i] + v[e] : v[e];
      }
    }
    return;
  }
  const int64_t base = (int64_t)(blockIdx.x - t.blk_start[lo]) * RB_ELEMS;
  const bool tail = base + RB_ELEMS > j.nel;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int sg = 0; sg < j.nseg; ++sg) {
    const float* part = j.part[sg];
    const int pw = j.pw[sg];
    const bool vec = !tail && (j.nel & 3) == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
    f32x4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = w; k < pw; k += 128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = k + 16 * u;
        if (kk < pw) {
          const float* row = part + (int64_t)kk * j.nel + base;
          f32x4 v;
          if (vec) {
            v = *reinterpret_cast<const f32x4*>(row + lane * 4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (base + lane * 4 + e < j.nel) ? row[lane * 4 + e] : 0.f;
          }
          a[u] += v;
        }
      }
    }
    s += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0) {
    f32x4 v = red[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += red[k][lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = base + lane * 4 + e;
      if (i < j.nel) j.dw[i] = j.accumulate ? j.dw[i] + v[e] : v[e];
    }
  }
}

struct Plan {
  int clt, cht, cl_groups, ch_groups, pw, tiles_y, tiles_x, ntiles, txl;
  int ns;   // 1: N-split kernel (wgrad4x4_ns_kernel), cht = high-res channels per WAVE
  int ty;   // ... its tile rows
  int head; // 1: single-channel vector-ALU member (wgrad_head_kernel): one partial copy per 16 x 32 tile
  int small;   // 1: small-map kernel (wgrad_small_kernel<clt, cht>): flattened positions of ipb whole images per workgroup step
  int ipb, nblocks, pos, posp, pwf, plane, lds_bytes;
};

int ns_tile_rows(int clt, int cht, int LH);

Plan make_plan(const vts_wgrad_desc* d) {
  Plan pl;
  pl.ty = 0;
  pl.head = 0;
  const int CL = d->lo0.C + (d->lo1.data ? d->lo1.C : 0), CH = d->hi0.C + (d->hi1.data ? d->hi1.C : 0);
  pl.ns = 0;
  pl.small = 0;
  static const int use_small = vts_tune("VTS_WGRAD_SMALL", 1);
  if (use_small && !d->lo1.data && !d->hi1.data && d->N >= 32 && d->LH * d->LW <= 324 && d->HH <= 34 && d->HW <= 34 && CL <= 64 && CH <= 64 &&
      d->pad >= 0 && d->pad <= SW_HALO && d->pad + d->pad_dx >= 0 && d->pad + d->pad_dx <= SW_HALO) {
    const int clt = CL <= 16 ? 1 : (CL <= 32 ? 2 : 4);
    const int need = cdiv(CH, 4);
    const int chw = need <= 2 ? 2 : (need <= 4 ? 4 : (need <= 8 ? 8 : 16));
    if (clt * chw <= 32) {
      const int ph = d->HH + 2 * SW_HALO > (d->LH - 1) * d->stride - d->pad + SW_HALO + 4 ? d->HH + 2 * SW_HALO : (d->LH - 1) * d->stride - d->pad + SW_HALO + 4;
      const int padx = d->pad + d->pad_dx;
      const int pwf = d->HW + 2 * SW_HALO > (d->LW - 1) * d->stride - padx + SW_HALO + 4 ? d->HW + 2 * SW_HALO : (d->LW - 1) * d->stride - padx + SW_HALO + 4;
      const int plane = ph * pwf;
      int ipb = cdiv(d->N, 256);
      for (; ipb >= 1; --ipb) {
        const int pos = (ipb * d->LH * d->LW + 3) & ~3;
        const int posp = pos + ((4 - pos) & 31);       // row pitch = 4 (mod 32): the 16 channel rows of an A fragment land on distinct banks
        const int64_t floats = (int64_t)clt * 16 * posp + (int64_t)ipb * CH * plane + pos;
        static const int lds_cap_kb = vts_tune("VTS_WGRAD_SMALL_LDS_KB", 150);
        if (floats * 4 <= (int64_t)lds_cap_kb * 1024 || (ipb == 1 && floats * 4 <= 150 * 1024)) {
          pl.small = 1; pl.clt = clt; pl.cht = chw; pl.ipb = ipb; pl.pos = pos; pl.posp = posp; pl.pwf = pwf; pl.plane = plane;
          pl.lds_bytes = (int)(floats * 4);
          pl.nblocks = cdiv(d->N, ipb);
          pl.pw = pl.nblocks < 256 ? pl.nblocks : 256;
          pl.cl_groups = pl.ch_groups = 1; pl.tiles_y = pl.tiles_x = pl.ntiles = 0; pl.txl = 0;
          return pl;
        }
      }
    }
  }
  static const int use_head = vts_tune("VTS_WGRAD_HEAD", 1);
  if (use_head && CL == 1 && !d->lo1.data && !d->hi1.data && d->stride == 1 && d->act_lo == VTS_ACT_NONE && !d->lo0.scale && !d->lo0.shift &&
      d->LH >= 8 && d->LW >= 16 && d->act_hi != VTS_ACT_TANH) {
    pl.head = 1;
    pl.clt = pl.cht = pl.cl_groups = pl.ch_groups = 1;
    pl.tiles_y = cdiv(d->LH, HWG_TY);
    pl.tiles_x = cdiv(d->LW, HWG_TX);
    pl.ntiles = d->N * pl.tiles_y * pl.tiles_x;
    pl.txl = HWG_TX;
    pl.pw = pl.ntiles;
    return pl;
  }
  static const int use_ns = vts_tune("VTS_WGRAD_NS", 1);
  static const int ns_min_ch = vts_tune("VTS_WGRAD_NS_MINCH", 2);   // (5 until round 3: 2 - 4 channel layers on the K-split kernel)
  static const int ns_min_w = vts_tune("VTS_WGRAD_NS_MINW", 8);
  // (the buffer-load addressing of the N-split kernel needs channel planes below 2^26 bytes)
  if (use_ns && CH >= ns_min_ch && d->LW > ns_min_w && (int64_t)d->HH * d->HW < (1 << 24) && (int64_t)d->LH * d->LW < (1 << 24)) {
    // N-split kernel: the waves of a workgroup split 4*cht high-res channels; groups are balanced so that the last one is not mostly empty
    pl.ns = 1;
    pl.cl_groups = cdiv(CL, 80);
    pl.clt = cdiv(CL, 16 * pl.cl_groups);
    pl.ch_groups = cdiv(CH, 20);
    pl.cht = cdiv(CH, 4 * pl.ch_groups);
    const int ty = d->stride == 2 ? ns_tile_rows(pl.clt, pl.cht, d->LH) : 4;
    pl.ty = ty;
    pl.tiles_y = cdiv(d->LH, ty);
    pl.txl = 28;
    pl.tiles_x = cdiv(d->LW, 28);
    pl.ntiles = d->N * pl.tiles_y * pl.tiles_x;
    const int64_t nel = (int64_t)CL * CH * 16;
    static const int old_plan = vts_tune("VTS_WGRAD_OLDPLAN", 0);
    if (old_plan) {   // round-2 rule (A/B switch): 512 workgroups, partial copies capped at 2x the operand bytes
      const int groups = pl.cl_groups * pl.ch_groups;
      int pw = 512 / groups;
      const int64_t operand = (int64_t)d->N * ((int64_t)CL * d->LH * d->LW + (int64_t)CH * d->HH * d->HW);
      int64_t cap = (2 * operand > (1 << 18) ? 2 * operand : (1 << 18)) / nel;
      if (cap > (16 << 20) / nel) cap = (16 << 20) / nel;
      if (pw > cap) pw = (int)cap;
      if (pw < 1) pw = 1;
      if (pw > pl.ntiles) pw = pl.ntiles;
      pl.pw = pw;
    } else {
      // Round 3 (tools/probes/wgrad_sweep.py, 21 shapes x ~130 plans on the MI355X):
      //  * a workgroup wants ~5 pixel tiles (fewer: its prologue, partial copy and the copy's reduction dominate);
      //  * waves with >= 12 accumulator tiles are MFMA-bound: ONE workgroup per CU (256), more only queue on the MFMA pipe and skew
      //    the finish times; lighter waves are bound by the load -> LDS -> MFMA latency chain and want 3-6 workgroups per CU;
      //  * when 5 tiles per workgroup leave CUs empty (inner layers: 32-400 tiles), split the CHANNELS over more workgroups rather
      //    than the pixels: a channel group re-reads an operand from L2, a pixel group writes (and the reduction re-reads) a whole
      //    copy of dw (0.2-0.8 MB for the 80-160 channel layers).
      //  * workgroup counts just above a multiple of 256 leave a tail (520 workgroups: 44 us, 512: 37 us), and channel groups
      //    that pad the channel count (40 channels as 3 x 16) run MFMAs on zeros.
      int copies = pl.ntiles / 5 > 1 ? pl.ntiles / 5 : 1;
      auto next_split = [](int C, int unit, int tiles_now, int& groups, int& tiles) {   // fewer tiles per wave without > 12 % padding
        for (int t = tiles_now - 1; t >= 2; --t) {
          const int g = cdiv(C, unit * t), tt = cdiv(C, unit * g);
          if (tt < tiles_now && (int64_t)g * tt * unit * 100 <= (int64_t)C * 112) {
            groups = g;
            tiles = tt;
            return true;
          }
        }
        return false;
      };
      while (pl.ty <= 4 && copies * pl.cl_groups * pl.ch_groups < 224) {      // (tall-tile thin layers keep their tile: their copies are cheap)
        if (next_split(CH, 4, pl.cht, pl.ch_groups, pl.cht)) continue;
        if (next_split(CL, 16, pl.clt, pl.cl_groups, pl.clt)) continue;
        break;
      }
      const int groups = pl.cl_groups * pl.ch_groups, w = pl.clt * pl.cht;
      // pixel split against copy cost: a tile costs ~1 us of latency chain + 13.3 ns per MFMA of one wave, a copy 2.6 ns per element
      // (written here, re-read by the reduction); the optimum of tiles/copies * t_tile + copies * t_copy, at least 2 tiles per workgroup
      const double t_tile = 1.0 + 0.0133 * w * (d->stride == 2 ? 14 : 28), t_copy = 2.6e-6 * (double)nel;
      int best = (int)__builtin_sqrt((double)pl.ntiles * t_tile / t_copy);
      if (best > pl.ntiles / 2) best = pl.ntiles / 2;
      if (best > copies) copies = best;
      const int wgs_max = w >= 12 ? 256 : (w >= 5 ? 768 : 1536);
      if (copies * groups > wgs_max) copies = wgs_max / groups;
      if (copies * groups > 256) {   // whole waves of 256 workgroups
        const int waves = (copies * groups + 128) / 256;
        copies = waves * 256 / groups;
      }
      if ((int64_t)copies * nel > (16 << 20)) copies = (int)((16 << 20) / nel);   // never more than 64 MB of partial copies
      if (copies < 1) copies = 1;
      if (copies > pl.ntiles) copies = pl.ntiles;
      pl.pw = copies;
    }
    static const bool tune = vts_tune_set("VTS_WGRAD_TUNE");   // tools/probes/wgrad_sweep.py: "cl_groups,ch_groups,copies" re-read per call
    if (tune) {
      const char* e = vts_tune_str("VTS_WGRAD_PLAN");
      int a = 0, b = 0, c = 0;
      if (e && sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a >= 1 && b >= 1 && c >= 1) {
        pl.cl_groups = a;
        pl.clt = cdiv(CL, 16 * a);
        pl.ch_groups = b;
        pl.cht = cdiv(CH, 4 * b);
        pl.pw = c > pl.ntiles ? pl.ntiles : c;
      }
    }
    return pl;
  }
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
  s.plain = (act == VTS_ACT_NONE && !a.scale && !a.shift && !(b.data && (b.scale || b.shift))) ? 1 : 0;
}

template <int S, int CLT, int CHT>
void launch_wg(const WgK& k, const Plan& pl, hipStream_t st) {
  dim3 grid(pl.pw, pl.cl_groups * pl.ch_groups);
  if (pl.txl == 8) hipLaunchKernelGGL((wgrad4x4_kernel<S, CLT, CHT, 8>), grid, dim3(256), 0, st, k);
  else hipLaunchKernelGGL((wgrad4x4_kernel<S, CLT, CHT, 32>), grid, dim3(256), 0, st, k);
  vts_set_kernel("wgrad4x4_kernel<%d, %d, %d, %d>", S, CLT, CHT, pl.txl);
}

template <int S, int CLT, int CHT>
void launch_ns(const WgK& k, const Plan& pl, hipStream_t st) {
  hipLaunchKernelGGL((wgrad4x4_ns_kernel<S, CLT, CHT>), dim3(pl.pw, pl.cl_groups * pl.ch_groups), dim3(256), 0, st, k);
  vts_set_kernel("wgrad4x4_ns_kernel<%d, %d, %d>", S, CLT, CHT);
}

template <int CLT, int CHT, int TY>
void launch_ns_tall(const WgK& k, const Plan& pl, hipStream_t st) {
  hipLaunchKernelGGL((wgrad4x4_ns_kernel<2, CLT, CHT, TY>), dim3(pl.pw, pl.cl_groups * pl.ch_groups), dim3(256), 0, st, k);
  vts_set_kernel("wgrad4x4_ns_kernel<2, %d, %d, %d>", CLT, CHT, TY);
}

// tile rows of the stride-2 N-split kernel for (clt, cht) accumulator tiles per wave on an LH-row map
int ns_tile_rows(int clt, int cht, int LH) {
  static const int tall = vts_tune("VTS_WGRAD_TALL", 1);
  // measured (tools/probes/wgrad_sweep.py): 8 rows help only with ONE high-resolution channel per wave (2 - 4 channel layers: up0
  // 52 -> 42 us, D layer 0 99 -> 77 us); with 2 - 3 channels per wave the 36 - 42 prefetch registers and 30 - 48 KB of patch rows
  // cost more co-resident workgroups than the longer tile saves (16 -> 8 channel layer: 43 -> 59 us)
  if (!tall || cht != 1 || clt > 2 || LH < 64) return 2;
  return 8;
}

template <int S>
bool dispatch_ns(const WgK& k, const Plan& pl, hipStream_t st) {
  if (S == 2 && pl.ty > 2) {
#define TALL_CASE(CLT, CHT, TY) \
  if (pl.clt == CLT && pl.cht == CHT && pl.ty == TY) { launch_ns_tall<CLT, CHT, TY>(k, pl, st); return true; }
    TALL_CASE(1, 1, 8) TALL_CASE(2, 1, 8)
#undef TALL_CASE
    return false;
  }
#define NS_CASE(CLT, CHT) \
  if (pl.clt == CLT && pl.cht == CHT) { launch_ns<S, CLT, CHT>(k, pl, st); return true; }
#define NS_ROW(CLT) NS_CASE(CLT, 1) NS_CASE(CLT, 2) NS_CASE(CLT, 3) NS_CASE(CLT, 4) NS_CASE(CLT, 5)
  NS_ROW(1) NS_ROW(2) NS_ROW(3) NS_ROW(4) NS_ROW(5)
#undef NS_ROW
#undef NS_CASE
  return false;
}

}  // namespace

// (round 4 built a producer / consumer member -- persistent workgroups of 4 MFMA + 2..8 loader waves, one read per operand; faster alone on
//  the thin full-size layers, 0.16 ms SLOWER in the step because its 512 / 768-thread workgroups stop sharing CUs with the other lanes.  It was
//  never the default; round 5 removed it from the library: tools/probes/wgrad_run_r04.hip.txt, numbers in profiles/r04b_wgrad_microbench_*.txt)

extern "C" int64_t vts_wgrad4x4_ws_floats(const vts_wgrad_desc* d) {
  if (!d) return 0;
  const int64_t CL = d->lo0.C + (d->lo1.data ? d->lo1.C : 0), CH = d->hi0.C + (d->hi1.data ? d->hi1.C : 0);
  const Plan pl = make_plan(d);
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
  static const int ablate = vts_tune("VTS_ABLATE", 0);
  k.ablate = ablate;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nel = (int64_t)k.lo.C * k.hi.C * 16;
  if (pl.small) {
    SmallWK q;
    q.lo = d->lo0.data; q.losc = d->lo0.scale; q.losh = d->lo0.shift; q.lons = d->lo0.nstride;
    q.hi = d->hi0.data; q.hisc = d->hi0.scale; q.hish = d->hi0.shift; q.hins = d->hi0.nstride;
    q.CL = k.lo.C; q.CH = k.hi.C; q.N = d->N; q.LH = d->LH; q.LW = d->LW; q.HH = d->HH; q.HW = d->HW; q.S = d->stride; q.pad = d->pad;
    q.padx = d->pad + d->pad_dx;
    q.lo_slope = vts_slope(d->act_lo); q.hi_slope = vts_slope(d->act_hi);
    q.IPB = pl.ipb; q.nblocks = pl.nblocks; q.POS = pl.pos; q.POSP = pl.posp; q.PW_ = pl.pwf; q.PLANE = pl.plane;
    q.part = ws;
    bool ok = false;
#define SW_CASE(CLT, CHW)                                                                                                      \
  if (pl.clt == CLT && pl.cht == CHW) {                                                                                        \
    static bool attr = false;                                                                                                  \
    if (!attr) {                                                                                                               \
      (void)hipFuncSetAttribute((const void*)wgrad_small_kernel<CLT, CHW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                             \
    }                                                                                                                          \
    hipLaunchKernelGGL((wgrad_small_kernel<CLT, CHW>), dim3(pl.pw), dim3(256), pl.lds_bytes, st, q);                           \
    vts_set_kernel("wgrad_small_kernel<%d, %d>", CLT, CHW);                                                                    \
    ok = true;                                                                                                                 \
  }
    SW_CASE(1, 2) SW_CASE(1, 4) SW_CASE(1, 8) SW_CASE(1, 16) SW_CASE(2, 2) SW_CASE(2, 4) SW_CASE(2, 8) SW_CASE(2, 16) SW_CASE(4, 2) SW_CASE(4, 4)
    SW_CASE(4, 8)
#undef SW_CASE
    VTS_CHECK_ARG(ok, "vts_wgrad4x4: no small-map instance for clt %d chw %d", pl.clt, pl.cht);
    VTS_CHECK_LAUNCH("vts_wgrad4x4 (small maps)");
    if (d->defer) return VTS_OK;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(nel, 64)), dim3(1024), 0, st, ws, nel, pl.pw, d->dw, d->accumulate);
    VTS_CHECK_LAUNCH("vts_wgrad4x4 reduce");
    return VTS_OK;
  }
  if (pl.head) {
    HeadWK q;
    q.lo = d->lo0.data; q.lons = d->lo0.nstride;
    q.hi = d->hi0.data; q.hisc = d->hi0.scale; q.hish = d->hi0.shift; q.hins = d->hi0.nstride;
    q.CH = k.hi.C; q.N = d->N; q.LH = d->LH; q.LW = d->LW; q.HH = d->HH; q.HW = d->HW; q.pad = d->pad; q.padx = d->pad + d->pad_dx;
    q.hi_slope = vts_slope(d->act_hi);
    q.tiles_y = pl.tiles_y; q.tiles_x = pl.tiles_x;
    q.part = ws;
    hipLaunchKernelGGL(wgrad_head_kernel, dim3(pl.ntiles), dim3(256), 0, st, q);
    vts_set_kernel("wgrad_head_kernel");
    VTS_CHECK_LAUNCH("vts_wgrad4x4 (single channel)");
    if (d->defer) return VTS_OK;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(nel, 64)), dim3(1024), 0, st, ws, nel, pl.pw, d->dw, d->accumulate);
    VTS_CHECK_LAUNCH("vts_wgrad4x4 reduce");
    return VTS_OK;
  }
  // groups that end beyond CL/CH never write their out-of-range rows, and every in-range element is
  // written by exactly one (cl-group, ch-group) workgroup of every pixel worker.
  if (pl.ns) {
    const bool ok = d->stride == 2 ? dispatch_ns<2>(k, pl, st) : dispatch_ns<1>(k, pl, st);
    VTS_CHECK_ARG(ok, "vts_wgrad4x4: no N-split instance for clt %d cht %d", pl.clt, pl.cht);
    VTS_CHECK_LAUNCH("vts_wgrad4x4 (N-split)");
    if (d->defer) return VTS_OK;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(nel, 64)), dim3(1024), 0, st, ws, nel, pl.pw, d->dw, d->accumulate);
    VTS_CHECK_LAUNCH("vts_wgrad4x4 reduce");
    return VTS_OK;
  }
#define WG_CASE(S, CLT, CHT) \
  if (d->stride == S && pl.clt == CLT && pl.cht == CHT) launch_wg<S, CLT, CHT>(k, pl, st);
  WG_CASE(2, 1, 4) WG_CASE(2, 1, 10) WG_CASE(2, 2, 4) WG_CASE(2, 2, 10) WG_CASE(2, 3, 4) WG_CASE(2, 3, 10) WG_CASE(2, 5, 4)
  WG_CASE(1, 1, 4) WG_CASE(1, 1, 10) WG_CASE(1, 2, 4) WG_CASE(1, 2, 10) WG_CASE(1, 3, 4) WG_CASE(1, 3, 10) WG_CASE(1, 5, 4)
#undef WG_CASE
  VTS_CHECK_LAUNCH("vts_wgrad4x4");
  if (d->defer) return VTS_OK;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(nel, 64)), dim3(1024), 0, st, ws, nel, pl.pw, d->dw,
                     d->accumulate);
  VTS_CHECK_LAUNCH("vts_wgrad4x4 reduce");
  return VTS_OK;
}

extern "C" int vts_wgrad_reduce_batch(const vts_reduce_job* jobs, int njobs, void* stream) {
  VTS_CHECK_ARG(jobs && njobs >= 0, "vts_wgrad_reduce_batch: null pointer");
  hipStream_t st = (hipStream_t)stream;
  for (int j0 = 0; j0 < njobs; j0 += RB_JOBS) {
    ReduceTable t;
    t.njobs = njobs - j0 < RB_JOBS ? njobs - j0 : RB_JOBS;
    int blocks = 0;
    for (int j = 0; j < t.njobs; ++j) {
      const vts_reduce_job& q = jobs[j0 + j];
      VTS_CHECK_ARG(q.dw && q.nel > 0 && q.nseg >= 1 && q.nseg <= VTS_REDUCE_MAX_SEG, "vts_wgrad_reduce_batch: bad job %d", j0 + j);
      for (int sg = 0; sg < q.nseg; ++sg) VTS_CHECK_ARG(q.part[sg] && q.pw[sg] >= 1, "vts_wgrad_reduce_batch: bad segment %d of job %d", sg, j0 + j);
      t.job[j] = q;
      t.blk_start[j] = blocks;
      int maxpw = 0;
      for (int sg = 0; sg < q.nseg; ++sg) maxpw = q.pw[sg] > maxpw ? q.pw[sg] : maxpw;
      static const int narrow_min = vts_tune("VTS_REDUCE_NARROW_MIN", 256);
      t.narrow[j] = maxpw > narrow_min ? 1 : 0;      // more than 16 copies per wave of the 256-element form
      blocks += (int)cdiv64(q.nel, t.narrow[j] ? 64 : RB_ELEMS);
    }
    t.blk_start[t.njobs] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, t);
    VTS_CHECK_LAUNCH("vts_wgrad_reduce_batch");
  }
  return VTS_OK;
}
