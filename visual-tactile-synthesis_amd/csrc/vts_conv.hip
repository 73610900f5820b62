// Implicit-GEMM 4x4 convolution family for gfx950 on the fp32 MFMA path
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chains, 157 TF peak).
//
// One kernel template covers nn.Conv2d forward, nn.ConvTranspose2d forward and both
// backward-data passes (vts.h: vts_conv4x4).  GEMM view per MFMA:
//   M = 16 consecutive output pixels of one row (of one output parity phase when the
//       operator is a stride-2 transposed conv),
//   N = 16 output channels,
//   K = 4 taps of one input channel: the 4 kx taps of one ky (conv, transposed s1) or the
//       2x2 taps that reach the phase (transposed s2).
// A workgroup (4 waves) owns a TY x TX tile of the phase grid and ALL output channels, so each
// input element is fetched from HBM once.  Per input-channel chunk the (haloed) input patch is
// staged in LDS with the producer's normalisation + activation applied on the fly and the two
// concat sources resolved (normalise-on-load: no separate InstanceNorm/BatchNorm/ReLU/cat pass
// ever touches HBM), the matching weight slice is staged tap-major, and every wave accumulates
// RW x MT x phases x NR 16x16 tiles in registers.
//
// LDS reads are conflict-free by construction: A fragments read stride-S words of one patch
// row (32 lanes -> 32 distinct banks or broadcast), B fragments read [k][cout] with the cout
// pitch = 16 (mod 32).
#include "vts_internal.h"

namespace {

struct ConvK {
  const float *s0, *s1, *sc0, *sh0, *sc1, *sh1;
  int64_t ns0, ns1;
  int C0, C1, Cin;
  int IH, IW, OH, OW, Cout, pad;
  const float* w;
  int ws_co, ws_ci;
  const float* bias;
  float* out;
  int64_t ons;
  int act_in, act_out;
  const float *dm, *dmsc, *dmsh;
  int64_t dmns;
  int dm_act, dmC;
  int accumulate;
  // small-grid decomposition: blockIdx.z = n + N * (cout_group + CG * k_slice)
  int N, CG, cps;   // cps = input-channel chunks per k-slice
  float* part;      // k-split partial sums [KS][N][Cout][OH][OW] (raw accumulators), or nullptr
};

template <int MODE, int S, int NR, int RW, int MT, int CK>
__global__ __launch_bounds__(256) void conv4x4_kernel(const ConvK p) {
  constexpr int P = (MODE == 1 && S == 2) ? 4 : 1;
  constexpr int TY = 4 * RW, TX = 16 * MT;
  constexpr int PR = MODE == 0 ? (TY - 1) * S + 4 : (S == 2 ? TY + 2 : TY + 3);
  constexpr int PC = MODE == 0 ? (TX - 1) * S + 4 : (S == 2 ? TX + 2 : TX + 3);
  constexpr int PCP = PC + 1;
  constexpr int COP = (NR % 2 == 1) ? NR * 16 : NR * 16 + 16;
  constexpr int PCM = (PC / 64) * 64;  // columns handled row-wise; the tail goes element-wise
  constexpr int TW = PC - PCM;

  __shared__ float lds_patch[CK * PR * PCP];
  __shared__ float lds_w[CK * 16 * COP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m16 = lane & 15, kq = lane >> 4;
  const int n = blockIdx.z % p.N;
  const int cg = (blockIdx.z / p.N) % p.CG, ks = blockIdx.z / (p.N * p.CG);
  const int co0 = cg * NR * 16;
  const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY;
  const int podd = p.pad & 1;

  int iy0, ix0;
  if (MODE == 0) {
    iy0 = ty0 * S - p.pad;
    ix0 = tx0 * S - p.pad;
  } else if (S == 2) {
    iy0 = ty0 + (p.pad >> 1) - 1;
    ix0 = tx0 + (p.pad >> 1) - 1;
  } else {
    iy0 = ty0 + p.pad - 3;
    ix0 = tx0 + p.pad - 3;
  }

  // per-lane A-fragment base offsets inside one channel plane of the patch
  int aoff[P];
  if (MODE == 0) {
    aoff[0] = (wave * RW * S) * PCP + m16 * S + kq;
  } else if (S == 2) {
#pragma unroll
    for (int ph = 0; ph < P; ++ph) {
      const int dpy = (ph >> 1) & podd, dpx = (ph & 1) & podd;
      aoff[ph] = (wave * RW + dpy + 1 - (kq >> 1)) * PCP + m16 + dpx + 1 - (kq & 1);
    }
  } else {
    aoff[0] = (wave * RW + 3) * PCP + m16 + 3 - kq;
  }

  f32x4 acc[RW][MT][P][NR];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ph = 0; ph < P; ++ph)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) acc[r][mt][ph][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t plane = (int64_t)p.IH * p.IW;
  const int nchunks = (p.Cin + CK - 1) / CK;
  const int chunk_end = min(nchunks, (ks + 1) * p.cps);
  for (int chunk = ks * p.cps; chunk < chunk_end; ++chunk) {
    const int cbase = chunk * CK;
    // ---- stage the input patch (normalise + activate + concat on load) ----
    auto row_setup = [&](int rr, const float*& src, float& sc, float& sh) -> bool {
      const int c = rr / PR, r = rr - c * PR;
      const int ci = cbase + c, iy = iy0 + r;
      if (ci >= p.Cin || iy < 0 || iy >= p.IH) return false;
      if (ci < p.C0) {
        src = p.s0 + n * p.ns0 + ci * plane + (int64_t)iy * p.IW;
        sc = p.sc0 ? p.sc0[n * p.C0 + ci] : 1.f;
        sh = p.sh0 ? p.sh0[n * p.C0 + ci] : 0.f;
      } else {
        const int c1 = ci - p.C0;
        src = p.s1 + n * p.ns1 + c1 * plane + (int64_t)iy * p.IW;
        sc = p.sc1 ? p.sc1[n * p.C1 + c1] : 1.f;
        sh = p.sh1 ? p.sh1[n * p.C1 + c1] : 0.f;
      }
      return true;
    };
    if (PCM > 0) {
      for (int rr = wave; rr < CK * PR; rr += 4) {
        const float* src = nullptr;
        float sc = 1.f, sh = 0.f;
        const bool ok = row_setup(rr, src, sc, sh);
#pragma unroll
        for (int col = lane; col < PCM; col += 64) {
          const int ix = ix0 + col;
          float v = 0.f;
          if (ok && ix >= 0 && ix < p.IW) v = vts_act(src[ix] * sc + sh, p.act_in);
          lds_patch[rr * PCP + col] = v;
        }
      }
    }
    if (TW > 0) {
      for (int idx = tid; idx < CK * PR * TW; idx += 256) {
        const int rr = idx / TW, col = PCM + (idx - rr * TW);
        const float* src = nullptr;
        float sc = 1.f, sh = 0.f;
        const bool ok = row_setup(rr, src, sc, sh);
        const int ix = ix0 + col;
        float v = 0.f;
        if (ok && ix >= 0 && ix < p.IW) v = vts_act(src[ix] * sc + sh, p.act_in);
        lds_patch[rr * PCP + col] = v;
      }
    }
    // ---- stage the weight slice: lds_w[c][slot][co], slot = K-group * 4 + k ----
    {
      constexpr int NCO = NR * 16;
      const bool co_major = p.ws_co >= p.ws_ci;  // Conv2d layout: taps of (co, ci..ci+CK) are contiguous
      for (int idx = tid; idx < CK * 16 * NCO; idx += 256) {
        int co, c, slot;
        if (co_major) {
          co = idx / (CK * 16);
          const int rem = idx - co * (CK * 16);
          c = rem >> 4;
          slot = rem & 15;
        } else {
          c = idx / (NCO * 16);
          const int rem = idx - c * (NCO * 16);
          co = rem >> 4;
          slot = rem & 15;
        }
        int tap = slot;
        if (MODE == 1 && S == 2) {
          const int ph = slot >> 2, a = (slot >> 1) & 1, b = slot & 1;
          const int ky = (((ph >> 1) + p.pad) & 1) + 2 * a, kx = (((ph & 1) + p.pad) & 1) + 2 * b;
          tap = ky * 4 + kx;
        }
        const int ci = cbase + c;
        float v = 0.f;
        if (co0 + co < p.Cout && ci < p.Cin) v = p.w[(int64_t)(co0 + co) * p.ws_co + (int64_t)ci * p.ws_ci + tap];
        lds_w[(c * 16 + slot) * COP + co] = v;
      }
    }
    __syncthreads();
    // ---- MFMA accumulate ----
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const float* pp = lds_patch + c * PR * PCP;
      const float* ww = lds_w + c * 16 * COP + kq * COP + m16;
      if (MODE == 1 && S == 2) {
#pragma unroll
        for (int ph = 0; ph < P; ++ph) {
          float b[NR];
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) b[nr] = ww[ph * 4 * COP + nr * 16];
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const float a = pp[aoff[ph] + r * PCP + mt * 16];
#pragma unroll
              for (int nr = 0; nr < NR; ++nr)
                acc[r][mt][ph][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nr], acc[r][mt][ph][nr], 0, 0, 0);
            }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float b[NR];
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) b[nr] = ww[g * 4 * COP + nr * 16];
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              float a;
              if (MODE == 0)
                a = pp[aoff[0] + (r * S + g) * PCP + mt * 16 * S];
              else
                a = pp[aoff[0] + (r - g) * PCP + mt * 16];
#pragma unroll
              for (int nr = 0; nr < NR; ++nr)
                acc[r][mt][0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nr], acc[r][mt][0][nr], 0, 0, 0);
            }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 16x16 tiles: col (cout) = lane&15, row (pixel) = (lane>>4)*4 + reg ----
  const int64_t oplane = (int64_t)p.OH * p.OW;
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int co = co0 + nr * 16 + m16;
    if (co >= p.Cout) continue;
    if (p.part) {  // k-split: raw accumulators, the epilogue runs in conv_split_epilogue_kernel
      float* pb = p.part + (((int64_t)ks * p.N + n) * p.Cout + co) * oplane;
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int ph = 0; ph < P; ++ph) {
            const int gy = ty0 + wave * RW + r;
            const int y = (P == 4) ? gy * 2 + (ph >> 1) : gy;
            if (y >= p.OH) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int gx = tx0 + mt * 16 + kq * 4 + j;
              const int x = (P == 4) ? gx * 2 + (ph & 1) : gx;
              if (x < p.OW) pb[(int64_t)y * p.OW + x] = acc[r][mt][ph][nr][j];
            }
          }
      continue;
    }
    const float bias = p.bias ? p.bias[co] : 0.f;
    float dsc = 1.f, dsh = 0.f;
    if (p.dm) {
      dsc = p.dmsc ? p.dmsc[n * p.dmC + co] : 1.f;
      dsh = p.dmsh ? p.dmsh[n * p.dmC + co] : 0.f;
    }
    float* obase = p.out + n * p.ons + co * oplane;
    const float* dbase = p.dm ? p.dm + n * p.dmns + co * oplane : nullptr;
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ph = 0; ph < P; ++ph) {
          const int gy = ty0 + wave * RW + r;
          const int y = (P == 4) ? gy * 2 + (ph >> 1) : gy;
          if (y >= p.OH) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int gx = tx0 + mt * 16 + kq * 4 + j;
            const int x = (P == 4) ? gx * 2 + (ph & 1) : gx;
            if (x >= p.OW) continue;
            float v = acc[r][mt][ph][nr][j] + bias;
            if (p.act_out == VTS_ACT_TANH) v = tanhf(v);
            const int64_t o = (int64_t)y * p.OW + x;
            if (dbase) v *= vts_act_grad(dbase[o] * dsc + dsh, p.dm_act);
            if (p.accumulate) v += obase[o];
            obase[o] = v;
          }
        }
  }
}

// sums the k-split partials in slice order and applies the epilogue of the main kernel
__global__ __launch_bounds__(256) void conv_split_epilogue_kernel(const ConvK p, int KS) {
  const int co = blockIdx.y, n = blockIdx.z;
  const int64_t oplane = (int64_t)p.OH * p.OW;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= oplane) return;
  float v = 0.f;
  for (int ks = 0; ks < KS; ++ks) v += p.part[(((int64_t)ks * p.N + n) * p.Cout + co) * oplane + o];
  v += p.bias ? p.bias[co] : 0.f;
  if (p.act_out == VTS_ACT_TANH) v = tanhf(v);
  if (p.dm) {
    const float dsc = p.dmsc ? p.dmsc[n * p.dmC + co] : 1.f, dsh = p.dmsh ? p.dmsh[n * p.dmC + co] : 0.f;
    v *= vts_act_grad(p.dm[n * p.dmns + co * oplane + o] * dsc + dsh, p.dm_act);
  }
  float* ob = p.out + n * p.ons + co * oplane + o;
  *ob = p.accumulate ? *ob + v : v;
}

template <int MODE, int S, int NR, int RW, int MT, int CK>
int launch(const ConvK& k, int N, hipStream_t st, int CG = 1, int KS = 1) {
  constexpr int P = (MODE == 1 && S == 2) ? 4 : 1;
  const int GH = P == 4 ? (k.OH + 1) / 2 : k.OH, GW = P == 4 ? (k.OW + 1) / 2 : k.OW;
  dim3 grid(cdiv(GW, 16 * MT), cdiv(GH, 4 * RW), N * CG * KS);
  hipLaunchKernelGGL((conv4x4_kernel<MODE, S, NR, RW, MT, CK>), grid, dim3(256), 0, st, k);
  VTS_CHECK_LAUNCH("vts_conv4x4");
  return VTS_OK;
}

// tile shape (RW, MT) of the full-width variants, as instantiated by VTS_DISPATCH below
inline void full_tile(int transposed, int stride, int nr, int& rw, int& mt) {
  static const int T[2][2][5][2] = {
      {{{2, 4}, {1, 4}, {1, 4}, {1, 2}, {1, 2}}, {{2, 4}, {1, 4}, {1, 4}, {1, 2}, {1, 2}}},   // conv   s1, s2
      {{{2, 4}, {1, 4}, {1, 4}, {1, 2}, {1, 2}}, {{1, 4}, {1, 2}, {1, 2}, {1, 1}, {1, 1}}}};  // convT  s1, s2
  rw = T[transposed][stride - 1][nr - 1][0];
  mt = T[transposed][stride - 1][nr - 1][1];
}

}  // namespace

extern "C" int64_t vts_conv4x4_ws_floats(const vts_conv_desc* d) {
  if (!d) return 0;
  const int nchunks = (d->in0.C + (d->in1.data ? d->in1.C : 0) + 3) / 4;
  const int ks_max = nchunks / 2 < 1 ? 1 : (nchunks / 2 > 320 ? 320 : nchunks / 2);
  return (int64_t)ks_max * d->N * d->Cout * d->OH * d->OW;
}

extern "C" int vts_conv4x4(const vts_conv_desc* d, void* stream) {
  VTS_CHECK_ARG(d && d->in0.data && d->w && d->out, "vts_conv4x4: null pointer");
  VTS_CHECK_ARG(d->stride == 1 || d->stride == 2, "vts_conv4x4: stride %d unsupported", d->stride);
  VTS_CHECK_ARG(d->Cout >= 1 && d->Cout <= 80, "vts_conv4x4: Cout %d outside 1..80", d->Cout);
  VTS_CHECK_ARG(d->N >= 1 && d->IH >= 1 && d->IW >= 1 && d->OH >= 1 && d->OW >= 1, "vts_conv4x4: bad shape");
  VTS_CHECK_ARG(d->in0.C >= 1 && d->in1.C >= 0, "vts_conv4x4: bad channel counts");
  if (!d->transposed) {
    VTS_CHECK_ARG(d->OH == (d->IH + 2 * d->pad - 4) / d->stride + 1 && d->OW == (d->IW + 2 * d->pad - 4) / d->stride + 1,
                  "vts_conv4x4: conv output %dx%d inconsistent with input %dx%d s%d p%d", d->OH, d->OW, d->IH, d->IW,
                  d->stride, d->pad);
  } else {
    // output size of a transposed conv is ambiguous (output_padding); require that the forward conv maps it back
    VTS_CHECK_ARG((d->OH + 2 * d->pad - 4) / d->stride + 1 == d->IH && (d->OW + 2 * d->pad - 4) / d->stride + 1 == d->IW,
                  "vts_conv4x4: transposed output %dx%d inconsistent with input %dx%d s%d p%d", d->OH, d->OW, d->IH,
                  d->IW, d->stride, d->pad);
  }
  ConvK k;
  k.s0 = d->in0.data; k.sc0 = d->in0.scale; k.sh0 = d->in0.shift; k.ns0 = d->in0.nstride; k.C0 = d->in0.C;
  k.s1 = d->in1.data; k.sc1 = d->in1.scale; k.sh1 = d->in1.shift; k.ns1 = d->in1.nstride;
  k.C1 = d->in1.data ? d->in1.C : 0;
  k.Cin = k.C0 + k.C1;
  k.IH = d->IH; k.IW = d->IW; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout; k.pad = d->pad;
  k.w = d->w; k.ws_co = d->ws_co; k.ws_ci = d->ws_ci; k.bias = d->bias;
  k.out = d->out; k.ons = d->out_nstride;
  k.act_in = d->act_in; k.act_out = d->act_out;
  k.dm = d->dmask.data; k.dmsc = d->dmask.scale; k.dmsh = d->dmask.shift; k.dmns = d->dmask.nstride;
  k.dm_act = d->dmask_act; k.dmC = d->dmask.C;
  k.accumulate = d->accumulate;
  k.N = d->N; k.CG = 1; k.cps = 1 << 30; k.part = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int nr = (d->Cout + 15) / 16;
  const int N = d->N;
  {
    // Small grids (inner U-Net layers: <= 32x32 maps, 80..592 channels) cannot fill 256 CUs with one
    // workgroup per spatial tile: split the output channels over workgroups (no reduction needed) and,
    // if that is still too few, the input-channel loop (k-split, deterministic two-kernel reduction).
    int rw, mt;
    full_tile(d->transposed, d->stride, nr, rw, mt);
    const bool ph4 = d->transposed && d->stride == 2;
    const int GH = ph4 ? (d->OH + 1) / 2 : d->OH, GW = ph4 ? (d->OW + 1) / 2 : d->OW;
    const int full_wgs = cdiv(GW, 16 * mt) * cdiv(GH, 4 * rw) * N;
    const int nchunks = (k.Cin + 3) / 4;
    if (full_wgs < 128 && (nr > 1 || nchunks >= 8)) {
      const int base = cdiv(GW, 32) * cdiv(GH, 4) * N * nr;
      int KS = 320 / base;
      if (KS > nchunks / 2) KS = nchunks / 2;
      if (KS < 1) KS = 1;
      int cps = cdiv(nchunks, KS);
      KS = cdiv(nchunks, cps);
      const int64_t need = (int64_t)KS * N * d->Cout * d->OH * d->OW;
      if (KS > 1 && (!d->ws || d->ws_floats < need)) { KS = 1; cps = nchunks; }
      k.CG = nr; k.cps = cps; k.part = KS > 1 ? d->ws : nullptr;
      int rc;
      if (!d->transposed) rc = d->stride == 2 ? launch<0, 2, 1, 1, 2, 4>(k, N, st, nr, KS) : launch<0, 1, 1, 1, 2, 4>(k, N, st, nr, KS);
      else rc = d->stride == 2 ? launch<1, 2, 1, 1, 2, 4>(k, N, st, nr, KS) : launch<1, 1, 1, 1, 2, 4>(k, N, st, nr, KS);
      if (rc != VTS_OK || KS == 1) return rc;
      hipLaunchKernelGGL(conv_split_epilogue_kernel, dim3((unsigned)cdiv64((int64_t)d->OH * d->OW, 256), d->Cout, N), dim3(256), 0, st, k, KS);
      VTS_CHECK_LAUNCH("vts_conv4x4 split epilogue");
      return VTS_OK;
    }
  }
#define VTS_DISPATCH(MODE, S, RW1, MT1, RW2, MT2, RW3, MT3, RW4, MT4, RW5, MT5) \
  switch (nr) {                                                                 \
    case 1: return launch<MODE, S, 1, RW1, MT1, 4>(k, N, st);                   \
    case 2: return launch<MODE, S, 2, RW2, MT2, 4>(k, N, st);                   \
    case 3: return launch<MODE, S, 3, RW3, MT3, 4>(k, N, st);                   \
    case 4: return launch<MODE, S, 4, RW4, MT4, 4>(k, N, st);                   \
    default: return launch<MODE, S, 5, RW5, MT5, 4>(k, N, st);                  \
  }
  if (!d->transposed) {
    if (d->stride == 2) { VTS_DISPATCH(0, 2, 2, 4, 1, 4, 1, 4, 1, 2, 1, 2) }
    VTS_DISPATCH(0, 1, 2, 4, 1, 4, 1, 4, 1, 2, 1, 2)
  } else {
    if (d->stride == 2) { VTS_DISPATCH(1, 2, 1, 4, 1, 2, 1, 2, 1, 1, 1, 1) }
    VTS_DISPATCH(1, 1, 2, 4, 1, 4, 1, 4, 1, 2, 1, 2)
  }
#undef VTS_DISPATCH
  return VTS_ERR_UNSUPPORTED;
}
