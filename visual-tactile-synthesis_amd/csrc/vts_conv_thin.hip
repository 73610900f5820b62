// Thin-layer variant of the stride-2 transposed member of the vts_conv4x4 family (ConvTranspose2d(4, s2) forward of the outer
// U-Net decoder layers, reference thirdparty/unet/unet_parts_custom.py:49-79, and the backward-data pass of the stride-2
// Conv2d(4) layers of the encoder / the PatchGAN discriminators, models/networks.py:1696-1750) for Cout <= 16 on full-size maps.
//
// These layers have an arithmetic intensity of 9-40 flop/B: they are HBM-bound, and on the 16 x 16 x 4 MFMA tiles of
// conv4x4_kernel the 16-wide output-channel dimension is mostly empty (Cout = 2, 3, 8, 10), so the matrix pipe does 2-8x the useful
// work.  Here a thread owns one 2 x 2 output quad (the four parity phases of one low-resolution position) for ALL output channels:
// per input channel it reads the 3 x 3 (pad odd) or 2 x 2 (pad even) neighbourhood it needs straight from global memory (lanes run
// along x: coalesced; the overlap between neighbouring lanes is served by L1 / TA), applies normalise + activate, and feeds
// 16 taps x Cout packed FMAs (v_pk_fma_f32 on channel pairs) whose weights come from LDS as wave-uniform 16-byte broadcast reads.
// No LDS tile, no barrier in the main loop, exact fp32.  Algorithmic bytes = 4 (in + out (1 + mask + acc) + w).
#include "vts_internal.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ThinK {
  const float *s0, *s1, *sc0, *sh0, *sc1, *sh1;
  int64_t ns0, ns1;
  int C0, Cin, IH, IW, OH, OW, Cout, ps;   // ps = (pad - (pad & 1)) / 2
  const float* w;
  int ws_co, ws_ci;
  const float* bias;
  float* out;
  int64_t ons;
  float slope_in;
  int act_out;
  const float *dm, *dmsc, *dmsh;
  int64_t dmns;
  int dm_act, dmC;
  int accumulate;
  int QH, QW;
};

// PP: parity of the padding; COQ: output channels / 4 (rounded up)
template <int PP, int COQ>
__global__ __launch_bounds__(256, 4) void convt2_thin_kernel(const ThinK p) {   // <= 128 VGPRs: 4 waves per SIMD
  constexpr int CO = 4 * COQ, ND = PP ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [ci][tap][CO], zero for co >= Cout
  const int tid = threadIdx.x, n = blockIdx.y;
  for (int e = tid; e < p.Cin * 16 * CO; e += 256) {
    const int co = e % CO, r = e / CO, tap = r & 15, ci = r >> 4;
    wl[e] = co < p.Cout ? p.w[(int64_t)co * p.ws_co + (int64_t)ci * p.ws_ci + tap] : 0.f;
  }
  __syncthreads();
  const int g = blockIdx.x * 256 + tid;
  if (g >= p.QH * p.QW) return;
  const int qy = g / p.QW, qx = g - qy * p.QW;
  const int by = qy + p.ps - 1, bx = qx + p.ps - 1;     // top-left of the neighbourhood
  int off[ND][ND];
  bool ok[ND][ND];
#pragma unroll
  for (int d = 0; d < ND; ++d)
#pragma unroll
    for (int e = 0; e < ND; ++e) {
      const int iy = by + d, ix = bx + e;
      ok[d][e] = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
      off[d][e] = min(max(iy, 0), p.IH - 1) * p.IW + min(max(ix, 0), p.IW - 1);
    }
  f32x2 acc[2][2][CO / 2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < CO / 2; ++c) acc[a][b][c] = f32x2{0.f, 0.f};

  const int64_t plane = (int64_t)p.IH * p.IW;
  const float slope = p.slope_in;
#pragma unroll 1
  for (int ci = 0; ci < p.Cin; ++ci) {
    const bool first = ci < p.C0;                        // wave-uniform
    const int cl = first ? ci : ci - p.C0;
    const float* src = first ? p.s0 + n * p.ns0 + cl * plane : p.s1 + n * p.ns1 + cl * plane;
    const float* scp = first ? p.sc0 : p.sc1;
    const float* shp = first ? p.sh0 : p.sh1;
    const int cn = first ? p.C0 : p.Cin - p.C0;
    const float sc = scp ? scp[n * cn + cl] : 1.f, sh = shp ? shp[n * cn + cl] : 0.f;
    float v[ND][ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int e = 0; e < ND; ++e) {
        float t = src[off[d][e]] * sc + sh;
        t = fmaxf(t, 0.f) + slope * fminf(t, 0.f);
        v[d][e] = ok[d][e] ? t : 0.f;
      }
    const float* wrow = wl + ci * 16 * CO;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ky = ((a + PP) & 1) + 2 * j;
        const int d = (a + PP - ky) / 2 + 1;             // (a + PP - ky) is even: exact
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int kx = ((b + PP) & 1) + 2 * i;
            const int e = (b + PP - kx) / 2 + 1;
            const float x = v[d][e];
#pragma unroll
            for (int cq = 0; cq < COQ; ++cq) {
              const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + (ky * 4 + kx) * CO + 4 * cq);
              acc[a][b][2 * cq] = f32x2{x, x} * f32x2{w4[0], w4[1]} + acc[a][b][2 * cq];
              acc[a][b][2 * cq + 1] = f32x2{x, x} * f32x2{w4[2], w4[3]} + acc[a][b][2 * cq + 1];
            }
          }
      }
  }

  const int64_t oplane = (int64_t)p.OH * p.OW;
  float* ob = p.out + n * p.ons;
  const float* db = p.dm ? p.dm + n * p.dmns : nullptr;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    if (c >= p.Cout) continue;
    const float bias = p.bias ? p.bias[c] : 0.f;
    const float msc = (p.dm && p.dmsc) ? p.dmsc[n * p.dmC + c] : 1.f, msh = (p.dm && p.dmsh) ? p.dmsh[n * p.dmC + c] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int y = 2 * qy + a, x = 2 * qx + b;
        if (y >= p.OH || x >= p.OW) continue;
        const int64_t o = c * oplane + (int64_t)y * p.OW + x;
        float val = acc[a][b][c >> 1][c & 1] + bias;
        if (p.act_out == VTS_ACT_TANH) val = tanhf(val);
        if (db) val *= vts_act_grad(db[o] * msc + msh, p.dm_act);
        if (p.accumulate) val += ob[o];
        ob[o] = val;
      }
  }
}

template <int PP>
int thin_launch(const ThinK& k, int N, hipStream_t st) {
  const int coq = (k.Cout + 3) / 4;
  const dim3 grid(cdiv(k.QH * k.QW, 256), N), block(256);
  const size_t lds = (size_t)k.Cin * 16 * 4 * coq * sizeof(float);
  switch (coq) {
    case 1: hipLaunchKernelGGL((convt2_thin_kernel<PP, 1>), grid, block, lds, st, k); break;
    case 2: hipLaunchKernelGGL((convt2_thin_kernel<PP, 2>), grid, block, lds, st, k); break;
    case 3: hipLaunchKernelGGL((convt2_thin_kernel<PP, 3>), grid, block, lds, st, k); break;
    default: hipLaunchKernelGGL((convt2_thin_kernel<PP, 4>), grid, block, lds, st, k); break;
  }
  vts_set_kernel("convt2_thin_kernel<%d, %d>", PP, coq);
  VTS_CHECK_LAUNCH("vts_conv4x4 (thin transposed)");
  return VTS_OK;
}

}  // namespace

// VTS_ERR_UNSUPPORTED: not a thin full-size stride-2 transposed case, use the MFMA kernels.
// (A forward stride-2 member of the same design -- one output pixel, or a 2 x 2 block, per thread with a 4 x 4 / 6 x 6 window -- was
// measured and dropped: 54 us vs 58 us for 4 -> 8 @513^2 at best, 2x slower with the blocked window at 2-3 waves per SIMD; its 16
// stride-2 loads and activations per output leave it instruction-bound like the MFMA kernel.)
int vts_conv_thin_try(const vts_conv_desc* d, hipStream_t st) {
  static const int enabled = vts_tune_set("VTS_NO_THIN") ? 0 : 1;
  const int Cin = d->in0.C + (d->in1.data ? d->in1.C : 0);
  if (!enabled || !d->transposed || d->stride != 2 || d->Cout > 16 || d->pad_dx != 0 || d->pad < 0) return VTS_ERR_UNSUPPORTED;
  // measured (profiles/r01k): wins 1.5-2.3x where Cin x Cout(padded) <= 128 (10 -> 3 @1024^2: 104 -> 51 us, 16 -> 8 @513^2: 70 -> 48 us);
  // beyond that the wave-uniform 16-byte LDS weight reads (one per two packed FMAs, shared by the four SIMDs of a CU) bound it
  // and the MFMA kernel is faster (40 -> 10 @512^2: 84 vs 172 us).  Small maps stay on conv_small.
  if ((int64_t)d->OH * d->OW < 128 * 128 || Cin * 4 * ((d->Cout + 3) / 4) > 128) return VTS_ERR_UNSUPPORTED;
  if (d->act_in == VTS_ACT_TANH || (d->act_out != VTS_ACT_NONE && d->act_out != VTS_ACT_TANH)) return VTS_ERR_UNSUPPORTED;
  ThinK k;
  k.s0 = d->in0.data; k.sc0 = d->in0.scale; k.sh0 = d->in0.shift; k.ns0 = d->in0.nstride; k.C0 = d->in0.C;
  k.s1 = d->in1.data; k.sc1 = d->in1.scale; k.sh1 = d->in1.shift; k.ns1 = d->in1.nstride;
  k.Cin = Cin; k.IH = d->IH; k.IW = d->IW; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout;
  const int pp = d->pad & 1;
  k.ps = (d->pad - pp) / 2;
  k.w = d->w; k.ws_co = d->ws_co; k.ws_ci = d->ws_ci; k.bias = d->bias; k.out = d->out; k.ons = d->out_nstride;
  k.slope_in = vts_slope(d->act_in); k.act_out = d->act_out;
  k.dm = d->dmask.data; k.dmsc = d->dmask.scale; k.dmsh = d->dmask.shift; k.dmns = d->dmask.nstride; k.dm_act = d->dmask_act; k.dmC = d->dmask.C;
  k.accumulate = d->accumulate;
  k.QH = (d->OH + 1) / 2; k.QW = (d->OW + 1) / 2;
  return pp ? thin_launch<1>(k, d->N, st) : thin_launch<0>(k, d->N, st);
}
