// Padding and anti-aliased resampling operators of the ResNet generator (reference models/networks.py:
// ReflectionPad2d / ReplicationPad2d in ResnetGenerator :1075 and ResnetBlock :1300, Downsample :51-74,
// Upsample :87-107), each with its adjoint, plus the K x K <-> 4 x 4 weight-tap embedding that lets the
// 3x3 / 7x7 convolutions run on the 4x4 implicit-GEMM kernels.  All of these are HBM-bound streaming
// kernels: one output element per thread, x fastest (coalesced), normalise-on-load on the input side.
#include "vts_internal.h"

namespace {

// index map of the padding modes: position t in [-pad, H + pad) -> source row, or -1 (zero)
__device__ __forceinline__ int pad_src(int t, int H, int mode) {
  if (t >= 0 && t < H) return t;
  if (mode == 1) return t < 0 ? -t : 2 * H - 2 - t;   // reflect (no edge repeat)
  if (mode == 2) return t < 0 ? 0 : H - 1;            // replicate
  return -1;
}

__device__ __forceinline__ float act_fwd(float t, int act) {
  if (act == VTS_ACT_TANH) return tanhf(t);
  return vts_act(t, act);
}

struct PadK {
  const float *x, *sc, *sh, *res;
  int64_t xns, ons;
  int C, H, W, pt, pl, PH, PW, mode, act;
  float* out;
};

__global__ __launch_bounds__(256) void pad_affine_kernel(const PadK k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z % k.C, n = blockIdx.z / k.C;
  if (x >= k.PW || y >= k.PH) return;
  const int sy = pad_src(y - k.pt, k.H, k.mode), sx = pad_src(x - k.pl, k.W, k.mode);
  const float a = k.sc ? k.sc[n * k.C + c] : 1.f, b = k.sh ? k.sh[n * k.C + c] : 0.f;
  float v = 0.f;
  if (sy >= 0 && sx >= 0) v = act_fwd(fmaf(k.x[n * k.xns + ((int64_t)c * k.H + sy) * k.W + sx], a, b), k.act);
  const int64_t o = n * k.ons + ((int64_t)c * k.PH + y) * k.PW + x;
  if (k.res) v += k.res[(((int64_t)n * k.C + c) * k.PH + y) * k.PW + x];
  k.out[o] = v;
}

// adjoint of the index map: every source pixel gathers the (<= 2 per axis) padded positions that read it
__device__ __forceinline__ int pad_preimages(int i, int H, int pad_lo, int pad_hi, int mode, int* t) {
  int cnt = 0;
  t[cnt++] = i;
  if (mode == 1) {
    if (i >= 1 && i <= pad_lo) t[cnt++] = -i;
    if (i <= H - 2 && H - 1 - i <= pad_hi) t[cnt++] = 2 * H - 2 - i;
  }
  return cnt;
}

struct PadBK {
  const float* dpad;
  int C, H, W, pt, pb, pl, pr, PH, PW, mode, accumulate;
  float* din;
};

__global__ __launch_bounds__(256) void pad_bwd_kernel(const PadBK k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= k.W || y >= k.H) return;
  const float* dp = k.dpad + (int64_t)blockIdx.z * k.PH * k.PW;
  float s = 0.f;
  if (k.mode == 2) {  // replicate: edge pixels collect the whole pad strip
    const int y0 = y == 0 ? -k.pt : y, y1 = y == k.H - 1 ? k.H - 1 + k.pb : y;
    const int x0 = x == 0 ? -k.pl : x, x1 = x == k.W - 1 ? k.W - 1 + k.pr : x;
    for (int ty = y0; ty <= y1; ++ty)
      for (int tx = x0; tx <= x1; ++tx) s += dp[(int64_t)(ty + k.pt) * k.PW + tx + k.pl];
  } else {
    int ty[3], tx[3];
    const int ny = pad_preimages(y, k.H, k.pt, k.pb, k.mode, ty), nx = pad_preimages(x, k.W, k.pl, k.pr, k.mode, tx);
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += dp[(int64_t)(ty[a] + k.pt) * k.PW + tx[b] + k.pl];
  }
  float* o = k.din + ((int64_t)blockIdx.z * k.H + y) * k.W + x;
  *o = k.accumulate ? *o + s : s;
}

// ---- Downsample: reflect pad 1, depthwise [1 2 1] x [1 2 1] / 16, stride 2 -------------------------------
struct BlurK {
  const float *x, *sc, *sh;
  int64_t xns;
  int C, H, W, OH, OW, act, accumulate;
  float* out;
};

__global__ __launch_bounds__(256) void blur_down_kernel(const BlurK k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z % k.C, n = blockIdx.z / k.C;
  if (x >= k.OW || y >= k.OH) return;
  const float a = k.sc ? k.sc[n * k.C + c] : 1.f, b = k.sh ? k.sh[n * k.C + c] : 0.f;
  const float* px = k.x + n * k.xns + (int64_t)c * k.H * k.W;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int sy = pad_src(2 * y + i - 1, k.H, 1);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int sx = pad_src(2 * x + j - 1, k.W, 1);
      const float f = (float)((i == 1 ? 2 : 1) * (j == 1 ? 2 : 1)) * (1.f / 16.f);
      s += f * vts_act(fmaf(px[(int64_t)sy * k.W + sx], a, b), k.act);
    }
  }
  k.out[(((int64_t)n * k.C + c) * k.OH + y) * k.OW + x] = s;
}

// adjoint: g(ty, tx) = sum_{i,j} f_i f_j dout[(ty+1-i)/2, (tx+1-j)/2] over the padded grid, folded by the reflection
__global__ __launch_bounds__(256) void blur_down_bwd_kernel(const BlurK k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= k.W || y >= k.H) return;
  const float* pd = k.x + (int64_t)blockIdx.z * k.OH * k.OW;   // dout
  int ty[3], tx[3];
  const int ny = pad_preimages(y, k.H, 1, 1, 1, ty), nx = pad_preimages(x, k.W, 1, 1, 1, tx);
  float s = 0.f;
  for (int a = 0; a < ny; ++a)
    for (int b = 0; b < nx; ++b) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int u = ty[a] + 1 - i;
        if (u < 0 || (u & 1) || (u >> 1) >= k.OH) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int v = tx[b] + 1 - j;
          if (v < 0 || (v & 1) || (v >> 1) >= k.OW) continue;
          const float f = (float)((i == 1 ? 2 : 1) * (j == 1 ? 2 : 1)) * (1.f / 16.f);
          s += f * pd[(int64_t)(u >> 1) * k.OW + (v >> 1)];
        }
      }
    }
  float* o = k.out + ((int64_t)blockIdx.z * k.H + y) * k.W + x;
  *o = k.accumulate ? *o + s : s;
}

// ---- Upsample: replicate pad 1, depthwise transposed [1 3 3 1] x [1 3 3 1] * 4 / 64, stride 2, cropped to 2H x 2W.
// out[y] = sum_u P[u] f[y + 3 - 2u],  P[u] = in[clamp(u - 1)],  u in [0, H + 2)
__device__ __forceinline__ float up_tap(int k) { return (k == 0 || k == 3) ? 0.25f : 0.75f; }   // [1 3 3 1] * 2 / 8 per axis

__global__ __launch_bounds__(256) void blur_up_kernel(const BlurK k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z % k.C, n = blockIdx.z / k.C;
  if (x >= k.OW || y >= k.OH) return;
  const float a = k.sc ? k.sc[n * k.C + c] : 1.f, b = k.sh ? k.sh[n * k.C + c] : 0.f;
  const float* px = k.x + n * k.xns + (int64_t)c * k.H * k.W;
  float s = 0.f;
#pragma unroll
  for (int du = 0; du < 2; ++du) {
    const int u = (y >> 1) + du + (y & 1);   // the two u with 0 <= y + 3 - 2u <= 3
    const int ky = y + 3 - 2 * u;
    const int sy = min(max(u - 1, 0), k.H - 1);
#pragma unroll
    for (int dv = 0; dv < 2; ++dv) {
      const int v = (x >> 1) + dv + (x & 1);
      const int kx = x + 3 - 2 * v;
      const int sx = min(max(v - 1, 0), k.W - 1);
      s += up_tap(ky) * up_tap(kx) * vts_act(fmaf(px[(int64_t)sy * k.W + sx], a, b), k.act);
    }
  }
  k.out[(((int64_t)n * k.C + c) * k.OH + y) * k.OW + x] = s;
}

// adjoint: dP[u] = sum_y dout[y] f[y + 3 - 2u];  din[i] = dP[i + 1] + (i == 0) dP[0] + (i == H - 1) dP[H + 1]
__global__ __launch_bounds__(256) void blur_up_bwd_kernel(const BlurK k) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= k.W || y >= k.H) return;
  const float* pd = k.x + (int64_t)blockIdx.z * k.OH * k.OW;   // dout [2H][2W]
  int uy[3], ux[3];
  int ny = 0, nx = 0;
  uy[ny++] = y + 1;
  if (y == 0) uy[ny++] = 0;
  if (y == k.H - 1) uy[ny++] = k.H + 1;
  ux[nx++] = x + 1;
  if (x == 0) ux[nx++] = 0;
  if (x == k.W - 1) ux[nx++] = k.W + 1;
  float s = 0.f;
  for (int a = 0; a < ny; ++a)
    for (int b = 0; b < nx; ++b) {
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        const int oy = ky - 3 + 2 * uy[a];
        if (oy < 0 || oy >= k.OH) continue;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const int ox = kx - 3 + 2 * ux[b];
          if (ox < 0 || ox >= k.OW) continue;
          s += up_tap(ky) * up_tap(kx) * pd[(int64_t)oy * k.OW + ox];
        }
      }
    }
  float* o = k.out + ((int64_t)blockIdx.z * k.H + y) * k.W + x;
  *o = k.accumulate ? *o + s : s;
}

// ---- K x K weights <-> the (a, b) 4 x 4 block of their zero-extended 8 x 8 tap grid -----------------------
// (oy, ox): position of the block's tap (0, 0) in the K x K grid (4a, 4b for the blocks of a large kernel; negative to place a
// small kernel inside the 4 x 4 block: the stride-2 3x3 / 1x1 convolutions of the StyleGAN2 ConvLayers)
__global__ __launch_bounds__(256) void tap_embed_kernel(const float* __restrict__ w, int64_t rows, int K, int oy, int ox,
                                                         float* __restrict__ w4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 16) return;
  const int64_t r = i >> 4;
  const int ky = oy + (((int)i >> 2) & 3), kx = ox + ((int)i & 3);
  w4[i] = (ky >= 0 && kx >= 0 && ky < K && kx < K) ? w[(r * K + ky) * K + kx] : 0.f;
}

__global__ __launch_bounds__(256) void tap_extract_kernel(const float* __restrict__ dw4, int64_t rows, int K, int oy, int ox,
                                                           float* __restrict__ dw, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 16) return;
  const int64_t r = i >> 4;
  const int ky = oy + (((int)i >> 2) & 3), kx = ox + ((int)i & 3);
  if (ky >= 0 && kx >= 0 && ky < K && kx < K) {
    float* o = dw + (r * K + ky) * K + kx;
    *o = accumulate ? *o + dw4[i] : dw4[i];
  }
}

}  // namespace

extern "C" int vts_pad_affine(const vts_operand* in, int N, int H, int W, int pt, int pb, int pl, int pr, int mode, int act,
                              const float* res, float* out, int64_t out_nstride, void* stream) {
  VTS_CHECK_ARG(in && in->data && out, "vts_pad_affine: null pointer");
  VTS_CHECK_ARG(mode >= 0 && mode <= 2 && pt >= 0 && pb >= 0 && pl >= 0 && pr >= 0, "vts_pad_affine: bad mode / pads");
  VTS_CHECK_ARG(mode != 1 || (pt < H && pb < H && pl < W && pr < W), "vts_pad_affine: reflect pad must be smaller than the image");
  VTS_CHECK_ARG((int64_t)N * in->C <= 65535, "vts_pad_affine: N*C too large");
  const int64_t ons = out_nstride ? out_nstride : (int64_t)in->C * (H + pt + pb) * (W + pl + pr);
  PadK k{in->data, in->scale, in->shift, res, in->nstride, ons, in->C, H, W, pt, pl, H + pt + pb, W + pl + pr, mode, act, out};
  hipLaunchKernelGGL(pad_affine_kernel, dim3(cdiv(k.PW, 64), cdiv(k.PH, 4), N * in->C), dim3(256), 0, (hipStream_t)stream, k);
  VTS_CHECK_LAUNCH("vts_pad_affine");
  return VTS_OK;
}

extern "C" int vts_pad_bwd(const float* dpad, int N, int C, int H, int W, int pt, int pb, int pl, int pr, int mode, float* din,
                           int accumulate, void* stream) {
  VTS_CHECK_ARG(dpad && din && mode >= 0 && mode <= 2 && (int64_t)N * C <= 65535, "vts_pad_bwd: bad args");
  PadBK k{dpad, C, H, W, pt, pb, pl, pr, H + pt + pb, W + pl + pr, mode, accumulate, din};
  hipLaunchKernelGGL(pad_bwd_kernel, dim3(cdiv(W, 64), cdiv(H, 4), N * C), dim3(256), 0, (hipStream_t)stream, k);
  VTS_CHECK_LAUNCH("vts_pad_bwd");
  return VTS_OK;
}

extern "C" int vts_blur_down(const vts_operand* in, int act, int N, int H, int W, float* out, void* stream) {
  VTS_CHECK_ARG(in && in->data && out && H >= 2 && W >= 2 && (int64_t)N * in->C <= 65535, "vts_blur_down: bad args");
  BlurK k{in->data, in->scale, in->shift, in->nstride, in->C, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, act, 0, out};
  hipLaunchKernelGGL(blur_down_kernel, dim3(cdiv(k.OW, 64), cdiv(k.OH, 4), N * in->C), dim3(256), 0, (hipStream_t)stream, k);
  VTS_CHECK_LAUNCH("vts_blur_down");
  return VTS_OK;
}

extern "C" int vts_blur_down_bwd(const float* dout, int N, int C, int H, int W, float* din, int accumulate, void* stream) {
  VTS_CHECK_ARG(dout && din && H >= 2 && W >= 2 && (int64_t)N * C <= 65535, "vts_blur_down_bwd: bad args");
  BlurK k{dout, nullptr, nullptr, 0, C, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 0, accumulate, din};
  hipLaunchKernelGGL(blur_down_bwd_kernel, dim3(cdiv(W, 64), cdiv(H, 4), N * C), dim3(256), 0, (hipStream_t)stream, k);
  VTS_CHECK_LAUNCH("vts_blur_down_bwd");
  return VTS_OK;
}

extern "C" int vts_blur_up(const vts_operand* in, int act, int N, int H, int W, float* out, void* stream) {
  VTS_CHECK_ARG(in && in->data && out && H >= 1 && W >= 1 && (int64_t)N * in->C <= 65535, "vts_blur_up: bad args");
  BlurK k{in->data, in->scale, in->shift, in->nstride, in->C, H, W, 2 * H, 2 * W, act, 0, out};
  hipLaunchKernelGGL(blur_up_kernel, dim3(cdiv(k.OW, 64), cdiv(k.OH, 4), N * in->C), dim3(256), 0, (hipStream_t)stream, k);
  VTS_CHECK_LAUNCH("vts_blur_up");
  return VTS_OK;
}

extern "C" int vts_blur_up_bwd(const float* dout, int N, int C, int H, int W, float* din, int accumulate, void* stream) {
  VTS_CHECK_ARG(dout && din && H >= 1 && W >= 1 && (int64_t)N * C <= 65535, "vts_blur_up_bwd: bad args");
  BlurK k{dout, nullptr, nullptr, 0, C, H, W, 2 * H, 2 * W, 0, accumulate, din};
  hipLaunchKernelGGL(blur_up_bwd_kernel, dim3(cdiv(W, 64), cdiv(H, 4), N * C), dim3(256), 0, (hipStream_t)stream, k);
  VTS_CHECK_LAUNCH("vts_blur_up_bwd");
  return VTS_OK;
}

extern "C" int vts_tap_embed(const float* w, int64_t rows, int K, int a, int b, float* w4, void* stream) {
  VTS_CHECK_ARG(w && w4 && rows >= 1 && K >= 1 && K <= 8 && a >= 0 && a <= 1 && b >= 0 && b <= 1, "vts_tap_embed: bad args");
  hipLaunchKernelGGL(tap_embed_kernel, dim3((unsigned)cdiv64(rows * 16, 256)), dim3(256), 0, (hipStream_t)stream, w, rows, K, 4 * a, 4 * b, w4);
  VTS_CHECK_LAUNCH("vts_tap_embed");
  return VTS_OK;
}

extern "C" int vts_tap_extract(const float* dw4, int64_t rows, int K, int a, int b, float* dw, int accumulate, void* stream) {
  VTS_CHECK_ARG(dw4 && dw && rows >= 1 && K >= 1 && K <= 8 && a >= 0 && a <= 1 && b >= 0 && b <= 1, "vts_tap_extract: bad args");
  hipLaunchKernelGGL(tap_extract_kernel, dim3((unsigned)cdiv64(rows * 16, 256)), dim3(256), 0, (hipStream_t)stream, dw4, rows, K, 4 * a, 4 * b, dw,
                     accumulate);
  VTS_CHECK_LAUNCH("vts_tap_extract");
  return VTS_OK;
}

extern "C" int vts_tap_embed_at(const float* w, int64_t rows, int K, int oy, int ox, float* w4, void* stream) {
  VTS_CHECK_ARG(w && w4 && rows >= 1 && K >= 1 && K <= 8 && oy >= -3 && oy <= 7 && ox >= -3 && ox <= 7, "vts_tap_embed_at: bad args");
  hipLaunchKernelGGL(tap_embed_kernel, dim3((unsigned)cdiv64(rows * 16, 256)), dim3(256), 0, (hipStream_t)stream, w, rows, K, oy, ox, w4);
  VTS_CHECK_LAUNCH("vts_tap_embed_at");
  return VTS_OK;
}

extern "C" int vts_tap_extract_at(const float* dw4, int64_t rows, int K, int oy, int ox, float* dw, int accumulate, void* stream) {
  VTS_CHECK_ARG(dw4 && dw && rows >= 1 && K >= 1 && K <= 8 && oy >= -3 && oy <= 7 && ox >= -3 && ox <= 7, "vts_tap_extract_at: bad args");
  hipLaunchKernelGGL(tap_extract_kernel, dim3((unsigned)cdiv64(rows * 16, 256)), dim3(256), 0, (hipStream_t)stream, dw4, rows, K, oy, ox, dw,
                     accumulate);
  VTS_CHECK_LAUNCH("vts_tap_extract_at");
  return VTS_OK;
}


// ---- separable windowed resampling from host-built tables (F.interpolate(mode="bicubic", antialias=True) of the patch / image
// resampling in compute_D2_loss and get_patch_in_input: reference sinskitG_model.py:1440-1476, 1531-1557, model_utils.py:300-340).
// Output (y, x) = sum_j wy[y][j] * sum_i wx[x][i] * in[ymin[y] + j][xmin[x] + i]; tables = (first index, window length, K weights per
// output index).  The adjoint is the same operator on the transposed tables (windows of a monotone resampling are intervals, so the
// outputs that read an input sample also form an interval): a gather, deterministic.
namespace {
__global__ __launch_bounds__(256) void resample_table_kernel(const float* __restrict__ in, int64_t NC, int IH, int IW, const int* __restrict__ ymin,
                                                              const int* __restrict__ ysize, const float* __restrict__ wy, int KY,
                                                              const int* __restrict__ xmin, const int* __restrict__ xsize,
                                                              const float* __restrict__ wx, int KX, float* __restrict__ out, int OH, int OW,
                                                              int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= NC * OH * OW) return;
  const int x = (int)(i % OW);
  const int64_t r = i / OW;
  const int y = (int)(r % OH);
  const int64_t nc = r / OH;
  const float* src = in + nc * IH * IW;
  const int y0 = ymin[y], ny = ysize[y], x0 = xmin[x], nx = xsize[x];
  float acc = 0.f;
  for (int j = 0; j < ny; ++j) {
    float row = 0.f;
    for (int k = 0; k < nx; ++k) row = fmaf(wx[x * KX + k], src[(int64_t)(y0 + j) * IW + x0 + k], row);
    acc = fmaf(wy[y * KY + j], row, acc);
  }
  out[i] = accumulate ? out[i] + acc : acc;
}
}  // namespace

extern "C" int vts_resample_table(const float* in, int64_t NC, int IH, int IW, const int* ymin, const int* ysize, const float* wy, int KY,
                                  const int* xmin, const int* xsize, const float* wx, int KX, float* out, int OH, int OW, int accumulate,
                                  void* stream) {
  VTS_CHECK_ARG(in && out && ymin && ysize && wy && xmin && xsize && wx && NC >= 1 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1 && KY >= 1 && KX >= 1,
                "vts_resample_table: bad args");
  hipLaunchKernelGGL(resample_table_kernel, dim3((unsigned)cdiv64(NC * OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, in, NC, IH, IW, ymin, ysize,
                     wy, KY, xmin, xsize, wx, KX, out, OH, OW, accumulate);
  VTS_CHECK_LAUNCH("vts_resample_table");
  return VTS_OK;
}
