// Weight gradient of the 4x4 convolution family, round-4 member: producer / consumer workgroups on double-buffered LDS.
//
//   dw[cl][ch][ky][kx] = sum_{n,y,x} lo[n,cl,y,x] * hi[n,ch, y*S+ky-pad, x*S+kx-padx]          (vts.h: vts_wgrad4x4)
//
// What the earlier members (vts_wgrad.hip) left on the table, measured in rounds 2-3: one wave per SIMD whose loads, LDS stores
// and MFMAs run strictly one after the other (the MFMA pipe is busy 30-47 % of a wave's life), 4-byte staging loads, and up to
// 1500 partial copies of dw per layer (116 + 102 MB of partials per step for the batched reduction to re-read).
//
// This member runs ONE persistent 512-thread workgroup per CU with two roles:
//   waves 4-7 (LOADERS, one per SIMD): global -> registers -> normalise + activate + zero padding -> LDS tile buffer (i+1) & 1.
//       16-byte raw buffer loads at dword alignment (gfx950 range-checks them per dword: tools/probes/buffer_oob.hip) through ONE
//       descriptor per operand source; a lane owns (channel, row, quad) units whose decode is tile-invariant (registers), rows and
//       columns outside the map become the out-of-range offset / a 4-bit column mask, interior tiles skip both.
//   waves 0-3 (CONSUMERS, one per SIMD): LDS tile buffer i & 1 -> v_mfma_f32_16x16x4_f32.  GEMM view as before: M = 16 low-resolution
//       channels, N = the 16 taps of one high-resolution channel, K = 4 consecutive pixels.  N-split form: wave w owns CHT high-res
//       channels and all CLT low-res channel tiles (no reduction inside the workgroup); K-split form (thin layers, <= 12 high-res
//       channels): wave w owns every fourth tile row and ALL channels, combined once at the end through LDS in wave order.
// One barrier per tile; the loaders' global loads for tile i+2 are in flight while the consumers multiply tile i.  The MFMA pipe and
// the VALU / memory pipes of a SIMD then work on different waves at the same time (MI355X_MICROARCH.md: "a MFMA-only wave and a
// VALU-only wave on the same CU run concurrently").
//
// A workgroup walks a fixed, strided list of 32-pixel-wide tiles and writes ONE partial copy [CL][CH][16]; the copies (<= 256 per
// channel group instead of up to 1536) are summed in a fixed order by the existing reduction kernels: deterministic, no atomics.
// Matches autograd's weight gradient of reference thirdparty/unet/unet_parts_custom.py:9-79 / models/networks.py:1696-1750.
#include <stdio.h>
#include <stdlib.h>

#include "vts_internal.h"

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;   // byte offset beyond every operand sample (the host side checks: sample < 2^31 bytes)

struct RSrc {
  const float *d0, *d1, *sc0, *sh0, *sc1, *sh1;
  int64_t ns0, ns1;
  int C0, C1, C;
  float slope;
  int plain;
};

struct RunK {
  RSrc lo, hi;
  int N, LH, LW, HH, HW, pad, padx;
  int cl_groups, ch_groups;
  int tiles_y, tiles_x, ntiles;
  float* part;
  int ablate;    // profiling only (VTS_ABLATE): 1 no global loads, 2 no LDS staging stores, 4 no fragment reads / MFMAs
  unsigned long long* trace;   // profiling only (VTS_WGRAD_TRACE=1): per workgroup and wave 4 x 64-bit cycle sums (s_memtime): phase a, phase b, barrier wait, total
};

__device__ __forceinline__ f32x4 ld_q(const rsrc_t& rs, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

// branch-free  pad( act( x * scale + shift ) ), activation in slope form
__device__ __forceinline__ float fin(float x, float sc, float sh, float slope, bool inside) {
  const float t = fmaf(x, sc, sh);
  const float a = fmaxf(t, 0.f) + slope * fminf(t, 0.f);
  return inside ? a : 0.f;
}

// One operand of one channel group as the loaders see it: NU unit slots per lane; slot i of lane l is unit (round i, lane l) of the
// 256 loader lanes.  Rounds [0, nr0) address source 0, the rest source 1 (a wave instruction never mixes descriptors).
template <int NU>
struct Units {
  unsigned boff[NU];   // byte offset of the unit inside its source's sample at tile origin (0, 0), OOB: absent channel / padding lane
  int rq[NU];          // row | quad << 8 | channel-in-group << 16, -1: padding lane (its store goes to the dump quad)
};
template <int NU>
struct Quads {
  f32x4 v[NU];
  float sc[NU], sh[NU];   // scale / shift of the unit's channel in the image of the tile this set holds (absent channel: 0, 0)
  int n;                  // ... that image
};

// NLW loader waves (4 or 8) behind the 4 consumer waves: a loader wave is bound by the latency of its own dependent instructions
// (measured with s_memtime, round 4: ~9 cycles per instruction with ONE loader wave per SIMD, the staging stores of a tile took twice
// the consumers' MFMA time), two per SIMD interleave.
template <int S, int CLT, int CHT, int TY, bool KSPLIT, int DEPTH, int NLW>
__global__ __launch_bounds__(256 + 64 * NLW) void wgrad_run_kernel(const RunK p) {
  constexpr int NL = 64 * NLW;                                  // loader lanes
  constexpr int TX = 32, TXP = 36, QL = TX / 4;
  constexpr int CLP = CLT * 16, CHW = KSPLIT ? CHT : 4 * CHT;
  constexpr int PRH = (TY - 1) * S + 4;
  constexpr int QH = S == 2 ? 18 : 10, PCHP = QH * 4;      // 72 / 40 floats = 8 (mod 32): B reads hit banks 8*ky + kx + S*k
  constexpr int LO_FLOATS = TY * CLP * TXP, HI_FLOATS = CHW * PRH * PCHP, BUF = LO_FLOATS + HI_FLOATS;
  constexpr int ULO = CLP * TY * QL, UHI = CHW * PRH * QH;
  constexpr int NULO = (ULO + NL - 1) / NL + 1, NUHI = (UHI + NL - 1) / NL + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][BUF] + 4 (dump quad)
  constexpr int DUMP = 2 * BUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int clg = blockIdx.y / p.ch_groups, chg = blockIdx.y - clg * p.ch_groups;
  const int cl0 = clg * CLP, ch0 = chg * CHW;
  const int offx = (int)((unsigned)(-p.padx) & 3u);      // column of the patch's first needed element inside its aligned-down quad
  const int my_tiles = ((int)blockIdx.x < p.ntiles) ? (p.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int iters = (my_tiles + DEPTH) / DEPTH * DEPTH;      // barrier-separated iterations of both roles (>= my_tiles + 1, multiple of DEPTH)

  // Tile walk without divisions in the loop: a workgroup's tiles are blockIdx.x + k * gridDim.x in (n, ty, tx) raster order; the
  // stride is decomposed once and every step is an add with two carries.
  struct Tile {
    int n, y0, x0;
  };
  struct Walk {
    int n, ty, tx;
  };
  const int tpi = p.tiles_y * p.tiles_x;
  const int sdn = (int)gridDim.x / tpi, sdy = ((int)gridDim.x - sdn * tpi) / p.tiles_x, sdx = (int)gridDim.x - sdn * tpi - sdy * p.tiles_x;
  auto walk_start = [&](int k) {    // position of tile blockIdx.x + k * gridDim.x (prologue only: divisions)
    const int tile = blockIdx.x + k * gridDim.x;
    Walk w;
    w.n = tile / tpi;
    const int rem = tile - w.n * tpi;
    w.ty = rem / p.tiles_x;
    w.tx = rem - w.ty * p.tiles_x;
    return w;
  };
  auto walk_step = [&](Walk& w, int times) {
    for (int i = 0; i < times; ++i) {
      w.tx += sdx;
      w.ty += sdy;
      w.n += sdn;
      if (w.tx >= p.tiles_x) { w.tx -= p.tiles_x; ++w.ty; }
      if (w.ty >= p.tiles_y) { w.ty -= p.tiles_y; ++w.n; }
    }
  };
  auto tile_of = [&](const Walk& w) {
    Tile t;
    t.n = w.n;
    t.y0 = w.ty * TY;
    t.x0 = w.tx * TX;
    return t;
  };

  if (wave >= 4) {
    // =========================================== LOADERS ===========================================
    const int ltid = tid - 256;
    Units<NULO> ul;
    Units<NUHI> uh;
    const int lplane = p.LH * p.LW, hplane = p.HH * p.HW;
    // channels of this group that live in source 0
    const int lc0 = min(max(p.lo.C0 - cl0, 0), CLP), hc0 = min(max(p.hi.C0 - ch0, 0), CHW);
    const int lnr0 = (lc0 * TY * QL + NL - 1) / NL, hnr0 = (hc0 * PRH * QH + NL - 1) / NL;
#pragma unroll
    for (int i = 0; i < NULO; ++i) {
      const bool s0 = i < lnr0;
      const int u = s0 ? i * NL + ltid : (i - lnr0) * NL + ltid + lc0 * TY * QL;
      const bool slot = s0 ? u < lc0 * TY * QL : u < ULO;
      const int q = u % QL, rc = u / QL, r = rc % TY, c = rc / TY;
      const int cg = cl0 + c;
      const bool ok = slot && cg < p.lo.C;
      ul.boff[i] = ok ? (unsigned)(((s0 ? cg : cg - p.lo.C0) * lplane + r * p.LW + 4 * q) * 4) : OOB;
      ul.rq[i] = slot ? (r | (q << 8) | (c << 16)) : -1;
    }
#pragma unroll
    for (int i = 0; i < NUHI; ++i) {
      const bool s0 = i < hnr0;
      const int u = s0 ? i * NL + ltid : (i - hnr0) * NL + ltid + hc0 * PRH * QH;
      const bool slot = s0 ? u < hc0 * PRH * QH : u < UHI;
      const int q = u % QH, rc = u / QH, r = rc % PRH, c = rc / PRH;
      const int cg = ch0 + c;
      const bool ok = slot && cg < p.hi.C;
      uh.boff[i] = ok ? (unsigned)(((s0 ? cg : cg - p.hi.C0) * hplane + r * p.HW + 4 * q) * 4) : OOB;
      uh.rq[i] = slot ? (r | (q << 8) | (c << 16)) : -1;
    }
    const bool lo_plain = p.lo.plain != 0, hi_plain = p.hi.plain != 0;

    // scale / shift of a unit's channel in image n; absent channels get (0, 0): act(0 * 0 + 0) = 0 without a select
    auto affine_of = [&](const RSrc& s, int c0g, int rq, unsigned boff, int n, float& sc, float& sh) {
      const int cg = min(c0g + ((rq >> 16) & 255), s.C - 1);
      const bool s0 = cg < s.C0;
      const float* scp = s0 ? s.sc0 : s.sc1;
      const float* shp = s0 ? s.sh0 : s.sh1;
      const int idx = n * (s0 ? s.C0 : s.C1) + (s0 ? cg : cg - s.C0);
      const bool present = boff != OOB;
      sc = present ? (scp ? scp[idx] : 1.f) : 0.f;
      sh = present ? (shp ? shp[idx] : 0.f) : 0.f;
    };

    auto issue = [&](const Tile& t0, bool live, Quads<NULO>& dl, Quads<NUHI>& dh) {
      // (always executed: a conditional issue would leave the number of younger loads unknown and force vmcnt(0) before every store;
      //  beyond the workgroup's last tile every offset is out of range, which costs no memory traffic)
      Tile t = t0;
      t.n = live ? t.n : 0;
      // per-sample descriptors of the two sources of both operands
      const int lb0 = (int)min((int64_t)p.lo.C0 * lplane * 4, (int64_t)0x7fffffff), lb1 = (int)min((int64_t)p.lo.C1 * lplane * 4, (int64_t)0x7fffffff);
      const int hb0 = (int)min((int64_t)p.hi.C0 * hplane * 4, (int64_t)0x7fffffff), hb1 = (int)min((int64_t)p.hi.C1 * hplane * 4, (int64_t)0x7fffffff);
      const rsrc_t rl0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lo.d0 + t.n * p.lo.ns0), 0, lb0, 0x00020000);
      const rsrc_t rl1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lo.d1 ? p.lo.d1 + t.n * p.lo.ns1 : p.lo.d0), 0, p.lo.d1 ? lb1 : 0, 0x00020000);
      const rsrc_t rh0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.hi.d0 + t.n * p.hi.ns0), 0, hb0, 0x00020000);
      const rsrc_t rh1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.hi.d1 ? p.hi.d1 + t.n * p.hi.ns1 : p.hi.d0), 0, p.hi.d1 ? hb1 : 0, 0x00020000);
      const unsigned lt = (unsigned)((t.y0 * p.LW + t.x0) * 4);
      const bool ledge = t.y0 + TY > p.LH;   // uniform: rows beyond the map would address the next channel's first rows
#pragma unroll
      for (int i = 0; i < NULO; ++i) {
        unsigned off = ul.boff[i] + lt;
        if (ledge) off = (t.y0 + (ul.rq[i] & 255) < p.LH) ? off : OOB;
        dl.v[i] = ld_q(i < lnr0 ? rl0 : rl1, live ? off : OOB);
      }
      const int ys = t.y0 * S - p.pad, xs = t.x0 * S - p.padx - offx;     // xs: aligned-down first column (may be -4)
      const unsigned ht = (unsigned)((ys * p.HW + xs) * 4);
      const bool hedge = ys < 0 || ys + PRH > p.HH;
#pragma unroll
      for (int i = 0; i < NUHI; ++i) {
        unsigned off = uh.boff[i] + ht;
        if (hedge) off = ((unsigned)(ys + (uh.rq[i] & 255)) < (unsigned)p.HH) ? off : OOB;
        dh.v[i] = ld_q(i < hnr0 ? rh0 : rh1, live ? off : OOB);
      }
      if (live && t.n != dl.n) {      // uniform branch, once per image and register set: the units' scale / shift
        dl.n = t.n;
        if (!lo_plain) {
#pragma unroll
          for (int i = 0; i < NULO; ++i) affine_of(p.lo, cl0, ul.rq[i], ul.boff[i], t.n, dl.sc[i], dl.sh[i]);
        }
        if (!hi_plain) {
#pragma unroll
          for (int i = 0; i < NUHI; ++i) affine_of(p.hi, ch0, uh.rq[i], uh.boff[i], t.n, dh.sc[i], dh.sh[i]);
        }
      }
    };

    // pad( act( x * scale + shift ) ): activation as max(t, slope * t) (slope in [0, 1]: identity, LeakyReLU, ReLU); the zero padding
    // needs a select only in tiles that touch the border of the map (uniform branch)
    auto store = [&](const Tile& t, int bufoff, const Quads<NULO>& dl, const Quads<NUHI>& dh) {
      const bool ledge = t.y0 + TY > p.LH || t.x0 + TX > p.LW;
#pragma unroll
      for (int i = 0; i < NULO; ++i) {
        f32x4 v = dl.v[i];
        const int r = ul.rq[i] & 255, q = (ul.rq[i] >> 8) & 255, c = (ul.rq[i] >> 16) & 255;
        if (!lo_plain) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float tt = fmaf(v[j], dl.sc[i], dl.sh[i]);
            v[j] = fmaxf(tt, p.lo.slope * tt);
          }
        }
        if (ledge) {
          const int x = t.x0 + 4 * q;
          const bool rok = t.y0 + r < p.LH;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (rok && x + j < p.LW) ? v[j] : 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + (ul.rq[i] < 0 ? DUMP : bufoff + (r * CLP + c) * TXP + 4 * q)) = v;
      }
      const int ys = t.y0 * S - p.pad, xs = t.x0 * S - p.padx - offx;
      const bool hedge = ys < 0 || ys + PRH > p.HH || xs < 0 || xs + PCHP > p.HW;
#pragma unroll
      for (int i = 0; i < NUHI; ++i) {
        f32x4 v = dh.v[i];
        const int r = uh.rq[i] & 255, q = (uh.rq[i] >> 8) & 255, c = (uh.rq[i] >> 16) & 255;
        if (!hi_plain) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float tt = fmaf(v[j], dh.sc[i], dh.sh[i]);
            v[j] = fmaxf(tt, p.hi.slope * tt);
          }
        }
        if (hedge) {
          const int x = xs + 4 * q;
          const bool rok = (unsigned)(ys + r) < (unsigned)p.HH;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (rok && (unsigned)(x + j) < (unsigned)p.HW) ? v[j] : 0.f;
        }
        *reinterpret_cast<f32x4*>(lds + (uh.rq[i] < 0 ? DUMP : bufoff + LO_FLOATS + (c * PRH + r) * PCHP + 4 * q)) = v;
      }
    };

    // DEPTH register sets: the loads of tile i + DEPTH are issued right after tile i went to LDS, so DEPTH tiles are in flight per CU
    // (thin layers multiply a tile faster than a memory round trip: with one set they were latency-bound at ~3 us per tile)
    Quads<NULO> dl[DEPTH];
    Quads<NUHI> dh[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      dl[k].n = -1;
#pragma unroll
      for (int i = 0; i < NULO; ++i) {
        dl[k].v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dl[k].sc[i] = dl[k].sh[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < NUHI; ++i) {
        dh[k].v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dh[k].sc[i] = dh[k].sh[i] = 0.f;
      }
    }
    Tile tl[DEPTH];
    const bool loads_on = !(p.ablate & 1);
    Walk wk = walk_start(0);
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      tl[k] = tile_of(wk);
      issue(tl[k], k < my_tiles && loads_on, dl[k], dh[k]);
      walk_step(wk, 1);
    }
    __syncthreads();
    // Branch-free steady state: every iteration stores a tile, issues the loads of the tile DEPTH ahead and meets the consumers at the
    // barrier -- also beyond the last tile (zeros from out-of-range loads go to a buffer nobody reads): with the stores / issues under a
    // condition the compiler cannot count the younger loads and waits for all of them before every store.
    unsigned long long ta = 0, tb = 0, tw = 0;
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        const int i = it + k;
        const unsigned long long c0 = p.trace ? __builtin_amdgcn_s_memtime() : 0;
        if (!(p.ablate & 2)) store(tl[k], (i & 1) * BUF, dl[k], dh[k]);
        const unsigned long long c1 = p.trace ? __builtin_amdgcn_s_memtime() : 0;
        tl[k] = tile_of(wk);
        issue(tl[k], i + DEPTH < my_tiles && loads_on, dl[k], dh[k]);
        walk_step(wk, 1);
        const unsigned long long c2 = p.trace ? __builtin_amdgcn_s_memtime() : 0;
        __syncthreads();
        if (p.trace) {
          const unsigned long long c3 = __builtin_amdgcn_s_memtime();
          ta += c1 - c0;
          tb += c2 - c1;
          tw += c3 - c2;
        }
      }
    }
    if (p.trace && lane == 0) {
      unsigned long long* o = p.trace + ((int64_t)(blockIdx.x + gridDim.x * blockIdx.y) * 12 + wave) * 4;
      o[0] = ta; o[1] = tb; o[2] = tw; o[3] = __builtin_amdgcn_s_memtime() - tstart;
    }
    if (KSPLIT) {   // the consumers' cross-wave reduction uses four more barriers
      __syncthreads();
      __syncthreads();
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  // =========================================== CONSUMERS ===========================================
  const int m16 = lane & 15, kq = lane >> 4;
  f32x4 acc[CLT][CHT];
#pragma unroll
  for (int t = 0; t < CLT; ++t)
#pragma unroll
    for (int h = 0; h < CHT; ++h) acc[t][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

  __syncthreads();
  Walk wk = walk_start(0);
  unsigned long long ta = 0, tw = 0;
  const unsigned long long tstart = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned long long c0 = p.trace ? __builtin_amdgcn_s_memtime() : 0;
    if (it >= 1 && it <= my_tiles && !(p.ablate & 4)) {
      const Tile t = tile_of(wk);
      walk_step(wk, 1);
      const float* buf = lds + ((it - 1) & 1) * BUF;
      const int nrow = min(TY, p.LH - t.y0);   // uniform: skip the padding of edge tiles
      // Fragment pipeline over (row, group of four k-steps = 16 pixels): lane group kq owns pixels 16 g + 4 kq + {0..3}, so the A
      // fragments of the four steps are ONE 16-byte LDS read per channel tile and the B fragments two ds_read2_b32 per channel
      // (measured round 4: a single wave per SIMD issuing ds_read_b32 gets a fraction of the LDS rate, the reads -- not the MFMAs --
      // bounded the consumers).  The reads of group j + 1 are issued between the MFMAs of group j (sched_group_barrier pattern);
      // they are unconditional (clamped at the last group): a conditional read leaves the number of younger LDS operations unknown
      // and the waits in front of the MFMAs then cover the reads just issued.  Steps beyond the map multiply zero columns of lo.
      const int ngr = (min(TX, p.LW - t.x0) + 15) >> 4;                    // groups per row: 1 or 2
      const int rows_mine = KSPLIT ? (nrow > wave ? (nrow - wave + 3) >> 2 : 0) : nrow;
      const int total = rows_mine * ngr;
      const float* lbase = buf + m16 * TXP + 4 * kq;
      const float* hbase = buf + LO_FLOATS + ((KSPLIT ? 0 : wave * CHT) * PRH + (m16 >> 2)) * PCHP + offx + 4 * kq * S + (m16 & 3);
      int rrow = KSPLIT ? wave : 0, rg = 0;       // next group to read
      f32x4 fa[2][CLT];
      float fb[2][CHT][4];
      auto rdgrp = [&](int set) {
        const float* lp = lbase + rrow * CLP * TXP + rg * 16;
        const float* hp = hbase + rrow * S * PCHP + rg * 16 * S;
#pragma unroll
        for (int tt = 0; tt < CLT; ++tt) fa[set][tt] = *reinterpret_cast<const f32x4*>(lp + tt * 16 * TXP);
#pragma unroll
        for (int h = 0; h < CHT; ++h)
#pragma unroll
          for (int st = 0; st < 4; ++st) fb[set][h][st] = hp[h * PRH * PCHP + st * S];
        const bool last = rg + 1 == ngr;
        const int nrow_next = rrow + (KSPLIT ? 4 : 1);
        const bool end = last && nrow_next >= nrow;
        rg = end ? rg : (last ? 0 : rg + 1);
        rrow = (last && !end) ? nrow_next : rrow;
      };
      auto mmgrp = [&](int set) {
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int tt = 0; tt < CLT; ++tt)
#pragma unroll
            for (int h = 0; h < CHT; ++h) acc[tt][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][tt][st], fb[set][h][st], acc[tt][h], 0, 0, 0);
      };
      auto interleave = [&]() {      // one LDS read behind every MFMA until the reads run out, the remaining MFMAs back to back
#pragma unroll
        for (int i = 0; i < 4 * CLT * CHT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      };
      if (total > 0 && (p.ablate & 24)) {       // profiling only: 8 = MFMAs without fragment reads, 16 = fragment reads without MFMAs
        rdgrp(0);
        rdgrp(1);
        for (int j = 0; j < total; j += 2) {
          if (p.ablate & 16) {
            rdgrp(1);
            rdgrp(0);
          } else {
            mmgrp(0);
            if (j + 1 < total) mmgrp(1);
          }
        }
        if (p.ablate & 16) mmgrp(0), mmgrp(1);
      } else if (total > 0) {
        rdgrp(0);
        for (int j = 0; j < total; j += 2) {
          __builtin_amdgcn_sched_barrier(0);
          rdgrp(1);
          mmgrp(0);
          interleave();
          __builtin_amdgcn_sched_barrier(0);
          if (j + 1 < total) {       // (the same reads on both paths: their count stays known)
            rdgrp(0);
            mmgrp(1);
            interleave();
          } else {
            rdgrp(0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (it >= 1) {
      walk_step(wk, 1);
    }
    const unsigned long long c1 = p.trace ? __builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    if (p.trace) {
      ta += c1 - c0;
      tw += __builtin_amdgcn_s_memtime() - c1;
    }
  }
  if (p.trace && lane == 0) {
    unsigned long long* o = p.trace + ((int64_t)(blockIdx.x + gridDim.x * blockIdx.y) * 12 + wave) * 4;
    o[0] = ta; o[1] = 0; o[2] = tw; o[3] = __builtin_amdgcn_s_memtime() - tstart;
  }

  const int CL = p.lo.C, CH = p.hi.C;
  float* part = p.part + (int64_t)blockIdx.x * CL * CH * 16;
  if (KSPLIT) {
    // fixed-order combination of the four consumer waves through LDS: red[t][h][lane] (the tile buffers are free now)
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int t = 0; t < CLT; ++t)
#pragma unroll
          for (int h = 0; h < CHT; ++h) {
            f32x4* slot = red + (t * CHT + h) * 64 + lane;
            if (w == 0) *slot = acc[t][h];
            else if (w < 3) *slot = *slot + acc[t][h];
            else acc[t][h] = *slot + acc[t][h];
          }
      }
      __syncthreads();
    }
    if (wave != 3) return;
  }
  // D layout of a 16x16 tile: row (cl) = (lane>>4)*4 + reg, col (tap) = lane&15 -> 64-byte runs per (cl, ch)
#pragma unroll
  for (int t = 0; t < CLT; ++t)
#pragma unroll
    for (int h = 0; h < CHT; ++h) {
      const int ch = ch0 + (KSPLIT ? 0 : wave * CHT) + h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cl = cl0 + t * 16 + kq * 4 + r;
        if (cl < CL && ch < CH) part[((int64_t)cl * CH + ch) * 16 + m16] = acc[t][h][r];
      }
    }
}

struct RunPlan {
  int ok, ksplit, clt, cht, ty, cl_groups, ch_groups, tiles_y, tiles_x, ntiles, copies, lds_bytes;
};

int lds_bytes_of(int S, int clt, int chw, int ty) {
  const int prh = (ty - 1) * S + 4, pchp = S == 2 ? 72 : 40;
  return (2 * (ty * clt * 16 * 36 + chw * prh * pchp) + 4) * 4;
}

// N-split instances: two register sets (two tiles in flight) where a tile is multiplied faster than a memory round trip
constexpr int ns_depth(int clt, int cht) { return clt * cht <= 6 ? 2 : 1; }
// loader waves: eight (three waves per SIMD: <= 168 registers) unless the consumers' accumulators need the two-waves-per-SIMD budget
constexpr int ns_loaders(int clt, int cht, bool ksplit) { return (ksplit ? clt * cht <= 12 : clt * cht <= 12) ? 8 : 4; }

RunPlan make_run_plan(const vts_wgrad_desc* d) {
  RunPlan pl;
  pl.ok = 0;
  static const int on = getenv("VTS_WGRAD_RUN") ? atoi(getenv("VTS_WGRAD_RUN")) : 0;   // off by default: see the measurements in DESIGN.md (round 4)
  if (!on) return pl;
  const int CL = d->lo0.C + (d->lo1.data ? d->lo1.C : 0), CH = d->hi0.C + (d->hi1.data ? d->hi1.C : 0);
  const int S = d->stride;
  // full-size maps only: the small-map / single-channel members keep their shapes
  if (d->LW < 48 || d->LH < 16 || CL < 2 || CH < 2) return pl;
  const int64_t lsample = (int64_t)(d->lo0.C > (d->lo1.data ? d->lo1.C : 0) ? d->lo0.C : d->lo1.C) * d->LH * d->LW * 4;
  const int64_t hsample = (int64_t)(d->hi0.C > (d->hi1.data ? d->hi1.C : 0) ? d->hi0.C : d->hi1.C) * d->HH * d->HW * 4;
  if (lsample >= (int64_t)1 << 30 || hsample >= (int64_t)1 << 30) return pl;
  if (d->pad < -4 || d->pad + d->pad_dx < -4) return pl;
  static const int force_ty = getenv("VTS_WGRAD_RUN_TY") ? atoi(getenv("VTS_WGRAD_RUN_TY")) : 0;
  static const int wgs = getenv("VTS_WGRAD_RUN_WGS") ? atoi(getenv("VTS_WGRAD_RUN_WGS")) : 256;   // one 512-thread workgroup per CU
  pl.tiles_x = cdiv(d->LW, 32);
  const bool plain_lo = d->act_lo == VTS_ACT_NONE && !d->lo0.scale && !d->lo0.shift && !(d->lo1.data && (d->lo1.scale || d->lo1.shift));
  const bool plain_hi = d->act_hi == VTS_ACT_NONE && !d->hi0.scale && !d->hi0.shift && !(d->hi1.data && (d->hi1.scale || d->hi1.shift));
  static const int ns_on = getenv("VTS_WGRAD_RUN_NS") ? atoi(getenv("VTS_WGRAD_RUN_NS")) : 0;   // N-split form: measured at parity or behind wgrad4x4_ns_kernel, off by default
  static const int ks_maxcl = getenv("VTS_WGRAD_RUN_KS_MAXCL") ? atoi(getenv("VTS_WGRAD_RUN_KS_MAXCL")) : 16;
  if (!(S == 2 && CH <= 12 && CL <= ks_maxcl) && !ns_on) return pl;
  if (S == 2 && CH <= 12 && CL <= 32) {
    // thin layers: every consumer wave multiplies all channels of every fourth tile row
    pl.ksplit = 1;
    pl.clt = cdiv(CL, 16);
    static const int kcht[] = {2, 3, 4, 5, 8, 9, 10, 12};
    pl.cht = 12;
    for (int c : kcht)
      if (c >= CH) { pl.cht = c; break; }
    pl.cl_groups = pl.ch_groups = 1;
    pl.ty = force_ty ? force_ty : 4;
    if (pl.ty != 4 && pl.ty != 8) pl.ty = 4;
    if (pl.ty == 8 && (pl.clt > 1 || pl.cht > 5)) pl.ty = 4;
    if (lds_bytes_of(S, pl.clt, pl.cht, pl.ty) > 160 * 1024) pl.ty = 4;
  } else {
    // N-split: the (CLT x 4 CHT) register tile of dw per workgroup by a cost model -- a workgroup's time is its tiles times
    // max(MFMA time, loader time) plus the pipeline fill, plus what its partial copy costs to write and to reduce.  Small tiles
    // mean many channel groups, i.e. FEW copies (copies = workgroups / groups), and the loaders have the slack to re-stage.
    pl.ksplit = 0;
    pl.ty = S == 2 ? 2 : 4;
    const int prh = (pl.ty - 1) * S + 4, qh = S == 2 ? 18 : 10;
    const int tiles_y = cdiv(d->LH, pl.ty);
    const int64_t ntiles = (int64_t)d->N * tiles_y * pl.tiles_x;
    double best = 1e30;
    int bclt = 0, bcht = 0;
    static const char* force = getenv("VTS_WGRAD_RUN_TILE");     // "clt,cht" (sweeps)
    int fclt = 0, fcht = 0;
    if (force) (void)sscanf(force, "%d,%d", &fclt, &fcht);
    for (int clt = 1; clt <= 5; ++clt)
      for (int cht = 1; cht <= 5; ++cht) {
        const int clg = cdiv(CL, 16 * clt), chg = cdiv(CH, 4 * cht);
        if (cdiv(CL, 16 * clg) != clt || cdiv(CH, 4 * chg) != cht) continue;       // a smaller tile covers the same groups
        if (lds_bytes_of(S, clt, 4 * cht, pl.ty) > 160 * 1024) continue;
        if (fclt && (clt != fclt || cht != fcht)) continue;
        const int groups = clg * chg;
        int copies = wgs / groups;
        if (copies < 1) copies = 1;
        if (copies > ntiles) copies = (int)ntiles;
        const double tpw = (double)cdiv64(ntiles, copies);
        const double rounds = (double)cdiv(groups * copies, wgs);                     // > 1 when there are more groups than workgroups
        const double mfma = clt * cht * pl.ty * 8 * 32.0;
        const double quads_lo = clt * 16.0 * pl.ty * 8 / 256, quads_hi = 4.0 * cht * prh * qh / 256;
        const double load = quads_lo * (plain_lo ? 110 : 210) + quads_hi * (plain_hi ? 110 : 210) + 300;
        const double lds_rd = (4.0 * clt + 2.0 * cht) * 4 * pl.ty * 8;               // LDS cycles of the four consumers' fragment reads
        const double lat = ns_depth(clt, cht) == 2 ? 2500 : 5000;                     // memory round trip per tile that one register set exposes
        double iter = mfma > load ? mfma : load;
        if (lds_rd > iter) iter = lds_rd;
        if (lat > iter) iter = lat;
        const double part_bytes = (double)copies * groups * (clt * 16.0) * (4.0 * cht) * 16 * 4;
        const double cost = rounds * (tpw * iter + 8000) + part_bytes * 2.0 / 1500.0;   // ~1.5 KB per cycle for the copy written + re-read
        if (cost < best) { best = cost; bclt = clt; bcht = cht; }
      }
    if (!bclt) return pl;
    pl.clt = bclt;
    pl.cht = bcht;
    pl.cl_groups = cdiv(CL, 16 * bclt);
    pl.ch_groups = cdiv(CH, 4 * bcht);
  }
  pl.tiles_y = cdiv(d->LH, pl.ty);
  pl.ntiles = d->N * pl.tiles_y * pl.tiles_x;
  const int groups = pl.cl_groups * pl.ch_groups;
  int copies = wgs / groups;
  if (copies < 1) copies = 1;
  if (copies > pl.ntiles) copies = pl.ntiles;
  if (pl.ntiles / copies < 2) return pl;      // too few tiles per workgroup for the two-stage pipeline to pay
  pl.copies = copies;
  pl.lds_bytes = lds_bytes_of(S, pl.clt, pl.ksplit ? pl.cht : 4 * pl.cht, pl.ty);
  pl.ok = 1;
  return pl;
}

void fill(RSrc& s, const vts_operand& a, const vts_operand& b, int act) {
  s.d0 = a.data; s.sc0 = a.scale; s.sh0 = a.shift; s.ns0 = a.nstride; s.C0 = a.C;
  s.d1 = b.data; s.sc1 = b.scale; s.sh1 = b.shift; s.ns1 = b.nstride; s.C1 = b.data ? b.C : 0;
  s.C = s.C0 + s.C1;
  s.slope = vts_slope(act);
  s.plain = (act == VTS_ACT_NONE && !a.scale && !a.shift && !(b.data && (b.scale || b.shift))) ? 1 : 0;
}

template <int S, int CLT, int CHT, int TY, bool KSPLIT>
bool launch(const RunK& k, const RunPlan& pl, hipStream_t st) {
  constexpr int DEPTH = KSPLIT ? 2 : ns_depth(CLT, CHT);     // thin layers / small register tiles: two tiles in flight per CU
  constexpr int NLW = ns_loaders(CLT, CHT, KSPLIT);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wgrad_run_kernel<S, CLT, CHT, TY, KSPLIT, DEPTH, NLW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((wgrad_run_kernel<S, CLT, CHT, TY, KSPLIT, DEPTH, NLW>), dim3(pl.copies, pl.cl_groups * pl.ch_groups), dim3(256 + 64 * NLW), pl.lds_bytes, st, k);
  vts_set_kernel("wgrad_run_kernel<%d, %d, %d, %d, %s, %d, %d>", S, CLT, CHT, TY, KSPLIT ? "true" : "false", DEPTH, NLW);
  return true;
}

bool dispatch(const RunK& k, const RunPlan& pl, int S, hipStream_t st) {
  if (pl.ksplit) {
#define KS_CASE(CLT, CHT, TY) \
  if (S == 2 && pl.clt == CLT && pl.cht == CHT && pl.ty == TY) return launch<2, CLT, CHT, TY, true>(k, pl, st);
#define KS_ROW(CHT) KS_CASE(1, CHT, 4) KS_CASE(2, CHT, 4)
    KS_ROW(2) KS_ROW(3) KS_ROW(4) KS_ROW(5) KS_ROW(8) KS_ROW(9) KS_ROW(10) KS_ROW(12)
    KS_CASE(1, 2, 8) KS_CASE(1, 3, 8) KS_CASE(1, 4, 8) KS_CASE(1, 5, 8)
#undef KS_ROW
#undef KS_CASE
    return false;
  }
#define NS_CASE(SS, CLT, CHT, TY) \
  if (S == SS && pl.clt == CLT && pl.cht == CHT && pl.ty == TY) return launch<SS, CLT, CHT, TY, false>(k, pl, st);
#define NS_ROW(SS, CLT, TY) NS_CASE(SS, CLT, 1, TY) NS_CASE(SS, CLT, 2, TY) NS_CASE(SS, CLT, 3, TY) NS_CASE(SS, CLT, 4, TY) NS_CASE(SS, CLT, 5, TY)
  NS_ROW(2, 1, 2) NS_ROW(2, 2, 2) NS_ROW(2, 3, 2) NS_ROW(2, 4, 2) NS_ROW(2, 5, 2)
  NS_ROW(1, 1, 4) NS_ROW(1, 2, 4) NS_ROW(1, 3, 4) NS_ROW(1, 4, 4) NS_ROW(1, 5, 4)
#undef NS_ROW
#undef NS_CASE
  return false;
}

}  // namespace

// partial copies the round-4 member would write (0: the shape is not taken)
int vts_wgrad_run_copies(const vts_wgrad_desc* d) {
  const RunPlan pl = make_run_plan(d);
  return pl.ok ? pl.copies : 0;
}

// VTS_ERR_UNSUPPORTED: not taken (the caller falls through to the earlier members)
int vts_wgrad_run_try(const vts_wgrad_desc* d, float* ws, hipStream_t st) {
  const RunPlan pl = make_run_plan(d);
  if (!pl.ok) return VTS_ERR_UNSUPPORTED;
  RunK k;
  fill(k.lo, d->lo0, d->lo1, d->act_lo);
  fill(k.hi, d->hi0, d->hi1, d->act_hi);
  k.N = d->N; k.LH = d->LH; k.LW = d->LW; k.HH = d->HH; k.HW = d->HW; k.pad = d->pad; k.padx = d->pad + d->pad_dx;
  k.cl_groups = pl.cl_groups; k.ch_groups = pl.ch_groups;
  k.tiles_y = pl.tiles_y; k.tiles_x = pl.tiles_x; k.ntiles = pl.ntiles;
  k.part = ws;
  static const int ablate = getenv("VTS_ABLATE") ? atoi(getenv("VTS_ABLATE")) : 0;
  k.ablate = ablate;
  static const bool want_trace = getenv("VTS_WGRAD_TRACE") != nullptr;
  const int nwg = pl.copies * pl.cl_groups * pl.ch_groups;
  k.trace = nullptr;
  if (want_trace && hipMalloc(&k.trace, (size_t)nwg * 12 * 4 * 8) != hipSuccess) k.trace = nullptr;
  if (k.trace) (void)hipMemsetAsync(k.trace, 0, (size_t)nwg * 12 * 4 * 8, st);
  if (!dispatch(k, pl, d->stride, st)) return VTS_ERR_UNSUPPORTED;
  VTS_CHECK_LAUNCH("vts_wgrad4x4 (producer / consumer)");
  if (k.trace) {      // profiling only: mean / max phase cycles per role (s_memtime counts at 100 MHz)
    (void)hipStreamSynchronize(st);
    unsigned long long* h = (unsigned long long*)malloc((size_t)nwg * 12 * 4 * 8);
    (void)hipMemcpy(h, k.trace, (size_t)nwg * 12 * 4 * 8, hipMemcpyDeviceToHost);
    double sum[2][4] = {{0}}, mx[2][4] = {{0}}, cnt[2] = {0, 0};
    for (int w = 0; w < nwg * 12; ++w) {
      const int role = (w % 12) >= 4;
      if (h[w * 4 + 3] == 0) continue;      // (wave slot not used by this instance)
      cnt[role] += 1;
      for (int i = 0; i < 4; ++i) {
        sum[role][i] += (double)h[w * 4 + i];
        if ((double)h[w * 4 + i] > mx[role][i]) mx[role][i] = (double)h[w * 4 + i];
      }
    }
    const double n = cnt[0] > 0 ? cnt[0] : 1, m = cnt[1] > 0 ? cnt[1] : 1, tk = 1e-3;   // s_memtime ticks (shader clock) -> kilocycles
    fprintf(stderr, "[wgrad_run trace] %s | %d wgs, tiles/wg %.1f | kcycles consumer: compute %.1f (max %.1f) barrier %.1f total %.1f | loader: store %.1f (max %.1f) issue %.1f (max %.1f) barrier %.1f total %.1f\n",
            vts_last_kernel(), nwg, (double)pl.ntiles / pl.copies, sum[0][0] / n * tk, mx[0][0] * tk, sum[0][2] / n * tk, sum[0][3] / n * tk, sum[1][0] / m * tk, mx[1][0] * tk,
            sum[1][1] / m * tk, mx[1][1] * tk, sum[1][2] / m * tk, sum[1][3] / m * tk);
    free(h);
    (void)hipFree(k.trace);
  }
  return VTS_OK;
}
