// StyleGAN2 building blocks (reference models/stylegan_networks.py): upfirdn2d with its adjoint
// (:38-76, used by Blur :140-156 in front of every stride-2 EqualConv2d and behind the transposed ModulatedConv2d),
// the fused bias + LeakyReLU + gain (:18-35) with an optional residual add (ResBlock's (out + skip) / sqrt 2, :686-693) and
// its derivative, and the demodulation coefficients of ModulatedConv2d (:311-317).  All of these are HBM-bound elementwise /
// short-FIR passes (algorithmic bytes = 4 (in + out) per element).  Blur -- upfirdn2d with up = down = 1, every use in the networks of
// this path -- and its adjoint run on an LDS-tiled kernel (ufd_tile_kernel: a 16 x 64 output tile per workgroup, the haloed input
// tile staged once with the zero padding resolved, four outputs per thread from 16-byte LDS reads; the role the reference's CUDA
// upfirdn2d plays, thirdparty/stylegan2_ada/torch_utils/ops/upfirdn2d.cu).  The general up / down form keeps the one-thread-per-output
// gather (Upsample / Downsample modules: not used by the networks built here).
// The convolutions of the blocks run on the conv kernels of this library with the equalised-learning-rate scale folded into the
// operand affine (normalise-on-load), see vts/engine.py:sg2d_forward.
#include "vts_internal.h"

namespace {

struct UfdK {
  const float* in;
  float* out;
  int64_t NC;
  int IH, IW, OH, OW;
  int KH, KW, up, down, px0, py0;
  int accumulate;
  float k[64];   // the FLIPPED kernel (correlation taps), row major KH x KW
};

// out[nc, oy, ox] = sum_{ky,kx} k[ky][kx] * U[oy*down + ky - py0, ox*down + kx - px0],  U = `in` with up-1 zeros inserted
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const UfdK p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = p.NC * p.OH * p.OW;
  if (i >= total) return;
  const int ox = (int)(i % p.OW);
  const int64_t r = i / p.OW;
  const int oy = (int)(r % p.OH);
  const int64_t nc = r / p.OH;
  const float* src = p.in + nc * p.IH * p.IW;
  float acc = 0.f;
  for (int ky = 0; ky < p.KH; ++ky) {
    const int Y = oy * p.down + ky - p.py0;
    if (Y < 0 || Y % p.up) continue;
    const int iy = Y / p.up;
    if (iy >= p.IH) continue;
    for (int kx = 0; kx < p.KW; ++kx) {
      const int X = ox * p.down + kx - p.px0;
      if (X < 0 || X % p.up) continue;
      const int ix = X / p.up;
      if (ix >= p.IW) continue;
      acc += p.k[ky * p.KW + kx] * src[iy * p.IW + ix];
    }
  }
  p.out[i] = p.accumulate ? p.out[i] + acc : acc;
}

// up = down = 1 (Blur) and its adjoint: out[oy, ox] = sum k[ky][kx] * in[oy + ky - py0, ox + kx - px0], zero outside the input.
// Workgroup = 16 x 64 outputs of one plane; LDS tile (16 + KH - 1) x (64 + KW - 1), row pitch 72 (16-byte aligned rows).
constexpr int UT_Y = 16, UT_X = 64, UT_PITCH = 72, UT_MAXK = 8;
__global__ __launch_bounds__(256) void ufd_tile_kernel(const UfdK p) {
  __shared__ __attribute__((aligned(16))) float tile[(UT_Y + UT_MAXK - 1) * UT_PITCH];
  const int tid = threadIdx.x;
  const int64_t nc = blockIdx.z;
  const int oy0 = blockIdx.y * UT_Y, ox0 = blockIdx.x * UT_X;
  const int rows = UT_Y + p.KH - 1, cols = UT_X + p.KW - 1;
  const float* src = p.in + nc * p.IH * p.IW;
  for (int e = tid; e < rows * UT_PITCH; e += 256) {
    const int r = e / UT_PITCH, c = e - r * UT_PITCH;
    const int iy = oy0 + r - p.py0, ix = ox0 + c - p.px0;
    tile[e] = (c < cols && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) ? src[(int64_t)iy * p.IW + ix] : 0.f;
  }
  __syncthreads();
  const int tx = (tid & 15) * 4, ty = tid >> 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < p.KH; ++ky) {
    const float* row = tile + (ty + ky) * UT_PITCH + tx;
    const f32x4 a = *reinterpret_cast<const f32x4*>(row), b = *reinterpret_cast<const f32x4*>(row + 4), c = *reinterpret_cast<const f32x4*>(row + 8);
    const float v[12] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]};
#pragma unroll
    for (int kx = 0; kx < UT_MAXK; ++kx) {
      if (kx < p.KW) {
        const float w = p.k[ky * p.KW + kx];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, v[j + kx], acc[j]);
      }
    }
  }
  const int oy = oy0 + ty;
  if (oy >= p.OH) return;
  float* o = p.out + (nc * p.OH + oy) * p.OW;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ox = ox0 + tx + j;
    if (ox < p.OW) o[ox] = p.accumulate ? o[ox] + acc[j] : acc[j];
  }
}

// adjoint: din[nc, iy, ix] = sum_{ky,kx} k[ky][kx] * dout[nc, oy, ox]  with  oy*down + ky - py0 = iy*up  (gather: deterministic)
// here in = dout [OH x OW of the forward], out = din [IH x IW of the forward]
__global__ __launch_bounds__(256) void upfirdn2d_adj_kernel(const UfdK p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = p.NC * p.IH * p.IW;
  if (i >= total) return;
  const int ix = (int)(i % p.IW);
  const int64_t r = i / p.IW;
  const int iy = (int)(r % p.IH);
  const int64_t nc = r / p.IH;
  const float* src = p.in + nc * p.OH * p.OW;
  float acc = 0.f;
  for (int ky = 0; ky < p.KH; ++ky) {
    const int t = iy * p.up + p.py0 - ky;
    if (t < 0 || t % p.down) continue;
    const int oy = t / p.down;
    if (oy >= p.OH) continue;
    for (int kx = 0; kx < p.KW; ++kx) {
      const int s = ix * p.up + p.px0 - kx;
      if (s < 0 || s % p.down) continue;
      const int ox = s / p.down;
      if (ox >= p.OW) continue;
      acc += p.k[ky * p.KW + kx] * src[oy * p.OW + ox];
    }
  }
  p.out[i] = p.accumulate ? p.out[i] + acc : acc;
}

__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ res,
                                                        int64_t total, int C, int HW, float slope, float gain, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float v = x[i] + (bias ? bias[(i / HW) % C] : 0.f);
  out[i] = (v > 0.f ? v : slope * v) * gain + (res ? res[i] : 0.f);
}

__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ bias,
                                                            int64_t total, int C, int HW, float slope, float gain, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float v = x[i] + (bias ? bias[(i / HW) % C] : 0.f);
  dx[i] = g[i] * gain * (v > 0.f ? 1.f : slope);
}

// demod[n, co] = rsqrt(scale^2 * sum_ci s[n, ci]^2 * w2[co, ci] + eps),  w2[co, ci] = sum_k w[co, ci, k]^2 (computed here)
// one workgroup per (n, co)
__global__ __launch_bounds__(256) void demod_kernel(const float* __restrict__ w, const float* __restrict__ s, int Cout, int Cin, int KK,
                                                     float scale, float eps, float* __restrict__ demod) {
  __shared__ float red[16];
  const int co = blockIdx.x, n = blockIdx.y;
  float acc = 0.f;
  for (int e = threadIdx.x; e < Cin * KK; e += 256) {
    const float wv = w[(int64_t)co * Cin * KK + e], sv = s[n * Cin + e / KK];
    acc += wv * wv * sv * sv;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) demod[n * Cout + co] = rsqrtf(scale * scale * acc + eps);
}

// Style-free ModulatedConv2d weight (StyleGAN2Decoder's StyledConv layers call the convolution with style = None, i.e. s = 1:
// stylegan_networks.py:307-317): wout = v * d[co], v = scale * w, d[co] = rsqrt(sum_{ci,k} v^2 + eps); `transpose` writes
// [Ci, Co, KK] (the conv-layout weight whose input adjoint is the upsampling transposed convolution, :320-330).  One workgroup per co.
__global__ __launch_bounds__(256) void modw_kernel(const float* __restrict__ w, int Cout, int Cin, int KK, float scale, float eps, int transpose,
                                                    float* __restrict__ wout) {
  __shared__ float red[16];
  const int co = blockIdx.x;
  float acc = 0.f;
  for (int e = threadIdx.x; e < Cin * KK; e += 256) {
    const float v = scale * w[(int64_t)co * Cin * KK + e];
    acc += v * v;
  }
  const float d = rsqrtf(block_sum(acc, red) + eps);
  for (int e = threadIdx.x; e < Cin * KK; e += 256) {
    const int ci = e / KK, k = e - ci * KK;
    const float v = scale * w[(int64_t)co * Cin * KK + e] * d;
    wout[transpose ? ((int64_t)ci * Cout + co) * KK + k : (int64_t)co * Cin * KK + e] = v;
  }
}

// gradient through it: with G = dL/dwout, S[co] = sum_{ci,k} G v:  dL/dw = scale * (d G - d^3 v S)
__global__ __launch_bounds__(256) void modw_bwd_kernel(const float* __restrict__ w, const float* __restrict__ g, int Cout, int Cin, int KK, float scale,
                                                        float eps, int transpose, float* __restrict__ dw, int accumulate) {
  __shared__ float red[16];
  const int co = blockIdx.x;
  float q = 0.f, sgv = 0.f;
  for (int e = threadIdx.x; e < Cin * KK; e += 256) {
    const int ci = e / KK, k = e - ci * KK;
    const float v = scale * w[(int64_t)co * Cin * KK + e];
    const float gv = g[transpose ? ((int64_t)ci * Cout + co) * KK + k : (int64_t)co * Cin * KK + e];
    q += v * v;
    sgv += gv * v;
  }
  q = block_sum(q, red);
  sgv = block_sum(sgv, red);
  const float d = rsqrtf(q + eps), d3 = d * d * d;
  for (int e = threadIdx.x; e < Cin * KK; e += 256) {
    const int ci = e / KK, k = e - ci * KK;
    const int64_t i = (int64_t)co * Cin * KK + e;
    const float v = scale * w[i];
    const float gv = g[transpose ? ((int64_t)ci * Cout + co) * KK + k : i];
    const float r = scale * (d * gv - d3 * v * sgv);
    dw[i] = accumulate ? dw[i] + r : r;
  }
}

int ufd_fill(UfdK& p, const char* who, const float* in, int64_t NC, int IH, int IW, const float* kernel, int KH, int KW, int up, int down,
             int px0, int px1, int py0, int py1, float* out, int accumulate) {
  VTS_CHECK_ARG(in && out && kernel && NC >= 1 && IH >= 1 && IW >= 1 && KH >= 1 && KW >= 1 && KH * KW <= 64 && up >= 1 && down >= 1,
                "%s: bad args", who);
  const int OH = (IH * up + py0 + py1 - KH) / down + 1, OW = (IW * up + px0 + px1 - KW) / down + 1;
  VTS_CHECK_ARG(IH * up + py0 + py1 >= KH && IW * up + px0 + px1 >= KW, "%s: padded extent smaller than the kernel", who);
  p.NC = NC; p.IH = IH; p.IW = IW; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.up = up; p.down = down; p.px0 = px0; p.py0 = py0;
  p.accumulate = accumulate;
  for (int a = 0; a < KH; ++a)
    for (int b = 0; b < KW; ++b) p.k[a * KW + b] = kernel[(KH - 1 - a) * KW + (KW - 1 - b)];   // upfirdn2d correlates with the flipped kernel
  return VTS_OK;
}

}  // namespace

extern "C" int vts_upfirdn2d_out_size(int in, int k, int up, int down, int pad0, int pad1) { return (in * up + pad0 + pad1 - k) / down + 1; }

extern "C" int vts_upfirdn2d(const float* in, int64_t NC, int IH, int IW, const float* kernel, int KH, int KW, int up, int down, int px0,
                             int px1, int py0, int py1, float* out, int accumulate, void* stream) {
  UfdK p;
  const int rc = ufd_fill(p, "vts_upfirdn2d", in, NC, IH, IW, kernel, KH, KW, up, down, px0, px1, py0, py1, out, accumulate);
  if (rc != VTS_OK) return rc;
  p.in = in; p.out = out;
  if (up == 1 && down == 1 && KH <= UT_MAXK && KW <= UT_MAXK && NC <= 65535)   // Blur: LDS-tiled
    hipLaunchKernelGGL(ufd_tile_kernel, dim3(cdiv(p.OW, UT_X), cdiv(p.OH, UT_Y), (unsigned)NC), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3((unsigned)cdiv64(NC * p.OH * p.OW, 256)), dim3(256), 0, (hipStream_t)stream, p);
  VTS_CHECK_LAUNCH("vts_upfirdn2d");
  return VTS_OK;
}

extern "C" int vts_upfirdn2d_bwd(const float* dout, int64_t NC, int IH, int IW, const float* kernel, int KH, int KW, int up, int down, int px0,
                                 int px1, int py0, int py1, float* din, int accumulate, void* stream) {
  UfdK p;
  const int rc = ufd_fill(p, "vts_upfirdn2d_bwd", dout, NC, IH, IW, kernel, KH, KW, up, down, px0, px1, py0, py1, din, accumulate);
  if (rc != VTS_OK) return rc;
  p.in = dout; p.out = din;
  if (up == 1 && down == 1 && KH <= UT_MAXK && KW <= UT_MAXK && NC <= 65535) {
    // the adjoint of a correlation is the correlation with the point-reflected taps and the complementary padding: same tiled kernel
    UfdK q = p;
    q.IH = p.OH; q.IW = p.OW; q.OH = IH; q.OW = IW;
    q.py0 = KH - 1 - p.py0; q.px0 = KW - 1 - p.px0;
    for (int a = 0; a < KH; ++a)
      for (int b = 0; b < KW; ++b) q.k[a * KW + b] = p.k[(KH - 1 - a) * KW + (KW - 1 - b)];
    hipLaunchKernelGGL(ufd_tile_kernel, dim3(cdiv(IW, UT_X), cdiv(IH, UT_Y), (unsigned)NC), dim3(256), 0, (hipStream_t)stream, q);
    VTS_CHECK_LAUNCH("vts_upfirdn2d_bwd");
    return VTS_OK;
  }
  hipLaunchKernelGGL(upfirdn2d_adj_kernel, dim3((unsigned)cdiv64(NC * IH * IW, 256)), dim3(256), 0, (hipStream_t)stream, p);
  VTS_CHECK_LAUNCH("vts_upfirdn2d_bwd");
  return VTS_OK;
}

extern "C" int vts_bias_act(const float* x, const float* bias, const float* res, int N, int C, int64_t HW, float slope, float gain, float* out,
                            void* stream) {
  VTS_CHECK_ARG(x && out && N >= 1 && C >= 1 && HW >= 1 && HW < (1ll << 31), "vts_bias_act: bad args");
  const int64_t total = (int64_t)N * C * HW;
  hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, res, total, C, (int)HW, slope,
                     gain, out);
  VTS_CHECK_LAUNCH("vts_bias_act");
  return VTS_OK;
}

extern "C" int vts_bias_act_bwd(const float* g, const float* x, const float* bias, int N, int C, int64_t HW, float slope, float gain, float* dx,
                                void* stream) {
  VTS_CHECK_ARG(g && x && dx && N >= 1 && C >= 1 && HW >= 1 && HW < (1ll << 31), "vts_bias_act_bwd: bad args");
  const int64_t total = (int64_t)N * C * HW;
  hipLaunchKernelGGL(bias_act_bwd_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, g, x, bias, total, C, (int)HW,
                     slope, gain, dx);
  VTS_CHECK_LAUNCH("vts_bias_act_bwd");
  return VTS_OK;
}

extern "C" int vts_modconv_weight(const float* w, int Cout, int Cin, int KK, float scale, float eps, int transpose, float* wout, void* stream) {
  VTS_CHECK_ARG(w && wout && Cout >= 1 && Cin >= 1 && KK >= 1, "vts_modconv_weight: bad args");
  hipLaunchKernelGGL(modw_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, KK, scale, eps, transpose, wout);
  VTS_CHECK_LAUNCH("vts_modconv_weight");
  return VTS_OK;
}

extern "C" int vts_modconv_weight_bwd(const float* w, const float* g, int Cout, int Cin, int KK, float scale, float eps, int transpose, float* dw,
                                      int accumulate, void* stream) {
  VTS_CHECK_ARG(w && g && dw && Cout >= 1 && Cin >= 1 && KK >= 1, "vts_modconv_weight_bwd: bad args");
  hipLaunchKernelGGL(modw_bwd_kernel, dim3(Cout), dim3(256), 0, (hipStream_t)stream, w, g, Cout, Cin, KK, scale, eps, transpose, dw, accumulate);
  VTS_CHECK_LAUNCH("vts_modconv_weight_bwd");
  return VTS_OK;
}

extern "C" int vts_modconv_demod(const float* w, const float* s, int N, int Cout, int Cin, int KK, float scale, float eps, float* demod,
                                 void* stream) {
  VTS_CHECK_ARG(w && s && demod && N >= 1 && Cout >= 1 && Cin >= 1 && KK >= 1, "vts_modconv_demod: bad args");
  hipLaunchKernelGGL(demod_kernel, dim3(Cout, N), dim3(256), 0, (hipStream_t)stream, w, s, Cout, Cin, KK, scale, eps, demod);
  VTS_CHECK_LAUNCH("vts_modconv_demod");
  return VTS_OK;
}
