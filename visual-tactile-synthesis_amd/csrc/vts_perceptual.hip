// Kernels around the frozen VGG feature stacks of the perceptual terms (round 3):
//   LPIPS-VGG16      lpips.LPIPS(net="vgg") as the reference calls it (models/sinskitG_model.py:495, 1639-1646, 1711;
//                    models/model_utils.py:477, 523-527) -- third-party package, algorithm restated in oracle/perceptual.py
//   VGG19 features   VGGLoss / Vgg19 of the pix2pixHD baseline (models/networks.py:2021-2067)
// The 3x3 convolutions run on the GEMM-class kernels (vts_conv3x3_wide.hip, MFMA-bound); everything here is the HBM-bound glue between
// them, each a single pass: ReLU + 2x2 max-pool + zero padding in one kernel, its adjoint, ReLU mask + padding of a gradient, the
// LPIPS head (channel unit-normalisation, weighted squared difference, spatial mean) with its gradient, and the ReLU-on-load L1.
// Activations are kept as RAW convolution outputs z; relu(z) is applied by whoever reads them (normalise-on-load, as everywhere
// in this library).
#include "vts_internal.h"

namespace {

__device__ __forceinline__ void loss_add64(long long* slot, double v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)llrint(v * VTS_LOSS_SCALE));
}

// out[nc][1 + y][1 + x] = max over the 2x2 window of relu(z); the one-pixel border (pad = 1) is written as zeros: the next
// convolution's pre-padded input in one pass.  pad = 0: plain pooled map.
__global__ __launch_bounds__(256) void maxpool2_relu_pad_kernel(const float* __restrict__ z, int H, int W, int pad, float* __restrict__ out, int zpad) {
  const int OH = H >> 1, OW = W >> 1, PH = OH + 2 * pad, PW = OW + 2 * pad;
  const int64_t nc = blockIdx.y;
  const int ZW = W + 2 * zpad;                    // (zpad: z itself is a padded tensor [H + 2 zpad][W + 2 zpad], read at its interior)
  const float* zi = z + nc * (int64_t)(H + 2 * zpad) * ZW + (int64_t)zpad * ZW + zpad;
  float* o = out + nc * (int64_t)PH * PW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < PH * PW; i += gridDim.x * 256) {
    const int py = i / PW, px = i - py * PW;
    const int y = py - pad, x = px - pad;
    float v = 0.f;
    if (y >= 0 && y < OH && x >= 0 && x < OW) {
      const float* q = zi + (int64_t)(2 * y) * ZW + 2 * x;
      v = fmaxf(fmaxf(fmaxf(q[0], q[1]), fmaxf(q[ZW], q[ZW + 1])), 0.f);
    }
    o[i] = v;
  }
}

// AlexNet's MaxPool2d(kernel 3, stride 2) behind a ReLU (torchvision alexnet.features[1:3], [4:6]): out[nc][pad + y][pad + x] = max over the
// 3 x 3 window at (2y, 2x) of relu(z), OH = (H - 3) / 2 + 1; the `pad`-pixel border is written as zeros (the next convolution's padding).
__global__ __launch_bounds__(256) void maxpool3s2_relu_pad_kernel(const float* __restrict__ z, int H, int W, int pad, float* __restrict__ out) {
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1, PH = OH + 2 * pad, PW = OW + 2 * pad;
  const int64_t nc = blockIdx.y;
  const float* zi = z + nc * (int64_t)H * W;
  float* o = out + nc * (int64_t)PH * PW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < PH * PW; i += gridDim.x * 256) {
    const int py = i / PW, px = i - py * PW;
    const int y = py - pad, x = px - pad;
    float v = 0.f;
    if (y >= 0 && y < OH && x >= 0 && x < OW) {
      const float* q = zi + (int64_t)(2 * y) * W + 2 * x;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) v = fmaxf(v, q[a * W + b]);
    }
    o[i] = v;
  }
}

// Space-to-depth by 4 of the zero-padded input of AlexNet's 11 x 11 stride-4 stem: out[n][(c * 4 + i) * 4 + j][Y][X] = xp[n][c][4Y + i][4X + j],
// xp = x zero-padded by `pad` (zero beyond, too).  The stem then is a VALID 3 x 3 stride-1 convolution over 16 C channels (weights
// w'[o][(c, i, j)][a][b] = w[o][c][4a + i][4b + j], zero where 4a + i > 10) and runs on the GEMM-class 3 x 3 kernel.
__global__ __launch_bounds__(256) void s2d4_pad_kernel(const float* __restrict__ x, int C, int H, int W, int pad, int OH, int OW, float* __restrict__ out) {
  const int64_t n = blockIdx.y;
  const int64_t total = (int64_t)C * 16 * OH * OW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int X = (int)(e % OW);
    const int64_t r = e / OW;
    const int Y = (int)(r % OH), ch = (int)(r / OH);
    const int c = ch >> 4, i = (ch >> 2) & 3, j = ch & 3;
    const int y = 4 * Y + i - pad, xx = 4 * X + j - pad;
    out[n * total + e] = (y >= 0 && y < H && xx >= 0 && xx < W) ? x[(n * C + c) * (int64_t)H * W + (int64_t)y * W + xx] : 0.f;
  }
}

// adjoint of (relu -> MaxPool2d(2, 2)) w.r.t. relu(z): the pooled gradient goes to the FIRST element (row-major scan of the
// window, PyTorch's tie rule) that holds the window maximum of relu(z).  Windows whose maximum is <= 0 route to their first element
// in PyTorch and the ReLU mask then removes it: zero here.  Elements outside any window (odd H / W) get zero.
// g2 (optional): the tap gradient of this layer, in z's layout, added where z > 0.  pad: the result is written into the interior of a
// [H + 2 pad][W + 2 pad] map with a zero border (the next input adjoint's pre-padded operand), i.e. the routed gradient is the gradient
// w.r.t. z itself (routing only reaches elements with z > 0).
__global__ __launch_bounds__(256) void maxpool2_relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ z, int H, int W,
                                                                float* __restrict__ gz, int zpad, const float* __restrict__ g2, int pad) {
  const int OH = H >> 1, OW = W >> 1, PH = H + 2 * pad, PW = W + 2 * pad;
  const int64_t nc = blockIdx.y;
  const int ZW = W + 2 * zpad;
  const int64_t zo = nc * (int64_t)(H + 2 * zpad) * ZW + (int64_t)zpad * ZW + zpad;
  const float* zi = z + zo;
  const float* ti = g2 ? g2 + zo : nullptr;
  const float* gi = g + nc * (int64_t)OH * OW;
  float* o = gz + nc * (int64_t)PH * PW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < PH * PW; i += gridDim.x * 256) {
    const int py = i / PW, px = i - py * PW;
    const int y = py - pad, x = px - pad;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const int wy = y >> 1, wx = x >> 1;
      if (wy < OH && wx < OW) {
        const float* q = zi + (int64_t)(2 * wy) * ZW + 2 * wx;
        const float e[4] = {q[0], q[1], q[ZW], q[ZW + 1]};
        const float m = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
        const int k = (y & 1) * 2 + (x & 1);
        bool first = e[k] == m && m > 0.f;
        for (int j = 0; j < k; ++j) first = first && e[j] != m;
        if (first) v = gi[(int64_t)wy * OW + wx];
      }
      if (ti && zi[(int64_t)y * ZW + x] > 0.f) v += ti[(int64_t)y * ZW + x];
    }
    o[i] = v;
  }
}

// The same for even H, W, one thread per 2 x 2 WINDOW (round 4, late): the window's four z values, its pooled gradient and its four tap
// gradients are read once (8-byte accesses; the per-element form reads every window four times and divides per element), the four routed
// gradients go out as two 8-byte stores; the zero border of the padded result is written by the threads of the first / last window of a row
// and by the first / last window rows.  Same routing rule (the FIRST maximum in row-major order, only where it is positive).
__global__ __launch_bounds__(256) void maxpool2_relu_bwd_win_kernel(const float* __restrict__ g, const float* __restrict__ z, int H, int W,
                                                                    float* __restrict__ gz, int zpad, const float* __restrict__ g2, int pad) {
  typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
  const int OH = H >> 1, OW = W >> 1, PW = W + 2 * pad;
  const int64_t nc = blockIdx.y;
  const int ZW = W + 2 * zpad;
  const int64_t zo = nc * (int64_t)(H + 2 * zpad) * ZW + (int64_t)zpad * ZW + zpad;
  const float* zi = z + zo;
  const float* ti = g2 ? g2 + zo : nullptr;
  const float* gi = g + nc * (int64_t)OH * OW;
  float* o = gz + nc * (int64_t)(H + 2 * pad) * PW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < OH * OW; i += gridDim.x * 256) {
    const int wy = i / OW, wx = i - wy * OW;
    const float* q = zi + (int64_t)(2 * wy) * ZW + 2 * wx;
    const f2u r0 = *reinterpret_cast<const f2u*>(q), r1 = *reinterpret_cast<const f2u*>(q + ZW);
    const float e[4] = {r0.x, r0.y, r1.x, r1.y};
    const float m = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
    const float gv = gi[i];
    int k = 4;                                   // index of the first maximum, if it is positive
    if (m > 0.f) k = e[0] == m ? 0 : e[1] == m ? 1 : e[2] == m ? 2 : 3;
    float v[4] = {k == 0 ? gv : 0.f, k == 1 ? gv : 0.f, k == 2 ? gv : 0.f, k == 3 ? gv : 0.f};
    if (ti) {
      const float* t = ti + (int64_t)(2 * wy) * ZW + 2 * wx;
      const f2u t0 = *reinterpret_cast<const f2u*>(t), t1 = *reinterpret_cast<const f2u*>(t + ZW);
      v[0] += e[0] > 0.f ? t0.x : 0.f;
      v[1] += e[1] > 0.f ? t0.y : 0.f;
      v[2] += e[2] > 0.f ? t1.x : 0.f;
      v[3] += e[3] > 0.f ? t1.y : 0.f;
    }
    float* d = o + (int64_t)(2 * wy + pad) * PW + 2 * wx + pad;
    *reinterpret_cast<f2u*>(d) = f2u{v[0], v[1]};
    *reinterpret_cast<f2u*>(d + PW) = f2u{v[2], v[3]};
    if (pad) {
      if (wx == 0)
        for (int r = 0; r < 2; ++r)
          for (int c = 0; c < pad; ++c) d[r * PW - pad + c] = 0.f;
      if (wx == OW - 1)
        for (int r = 0; r < 2; ++r)
          for (int c = 0; c < pad; ++c) d[r * PW + 2 + c] = 0.f;
      if (wy == 0 || wy == OH - 1) {             // the border rows above / below this window's two columns (+ the corners at the row ends)
        const int c_lo = wx == 0 ? -pad : 0, c_hi = wx == OW - 1 ? 2 + pad : 2;
        for (int r = 1; r <= pad; ++r) {
          float* b = wy == 0 ? d - (int64_t)r * PW : d + (int64_t)(1 + r) * PW;
          for (int c = c_lo; c < c_hi; ++c) b[c] = 0.f;
        }
        if (OH == 1) {                           // one window row: both the top and the bottom border belong to it
          for (int r = 1; r <= pad; ++r) {
            float* b = d + (int64_t)(1 + r) * PW;
            for (int c = c_lo; c < c_hi; ++c) b[c] = 0.f;
          }
        }
      }
    }
  }
}

// out[nc][pad + y][pad + x] = (g + g2) * (z > 0), zero border: ReLU backward fused with the zero padding the adjoint convolution wants.
// g (optional) is dense [H][W]; g2 (optional, a tap gradient) has z's layout.
__global__ __launch_bounds__(256) void relu_mask_pad_kernel(const float* __restrict__ g, const float* __restrict__ g2, const float* __restrict__ z,
                                                            int H, int W, int pad, float* __restrict__ out, int zpad) {
  const int PH = H + 2 * pad, PW = W + 2 * pad;
  const int64_t nc = blockIdx.y;
  const int64_t off = nc * (int64_t)H * W;
  const int ZW = W + 2 * zpad;
  const int64_t zo = nc * (int64_t)(H + 2 * zpad) * ZW + (int64_t)zpad * ZW + zpad;
  const float* zi = z + zo;
  const float* ti = g2 ? g2 + zo : nullptr;
  float* o = out + nc * (int64_t)PH * PW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < PH * PW; i += gridDim.x * 256) {
    const int py = i / PW, px = i - py * PW;
    const int y = py - pad, x = px - pad;
    float v = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const float t = (g ? g[off + (int64_t)y * W + x] : 0.f) + (ti ? ti[(int64_t)y * ZW + x] : 0.f);
      v = zi[(int64_t)y * ZW + x] > 0.f ? t : 0.f;
    }
    o[i] = v;
  }
}

// LPIPS head of one tap.  Thread = pixel, loop over channels (coalesced across the wave: channel stride HW).
//   a = f0 / (|f0| + eps), b = f1 / (|f1| + eps), d = sum_c w_c (a_c - b_c)^2, value = coeff * sum_{n, pixels} d / HW
//   dz0 (optional) = grad_coeff / HW * d(d)/d(f0) where z0 > 0, else 0: the gradient w.r.t. z0 itself (ReLU mask applied), in z0's layout
__global__ __launch_bounds__(256) void lpips_layer_kernel(const float* __restrict__ z0, const float* __restrict__ z1, int C, int HW,
                                                          const float* __restrict__ w, float coeff, long long* __restrict__ loss,
                                                          float* __restrict__ dz0, float grad_coeff, int W, int zpad) {
  __shared__ float red[16];
  const int n = blockIdx.y;
  // zpad: both feature tensors are padded [C][H + 2 zpad][W + 2 zpad] (the ReLU'd, pre-padded outputs of vts_conv3x3_wide_relu_pad), read
  // at their interior; the gradient dz0 has the same layout (its border is the caller's: vts_zero_border)
  const int ZW = W + 2 * zpad;
  const int64_t ZP = zpad ? (int64_t)(HW / W + 2 * zpad) * ZW : HW;
  const float* p0 = z0 + (int64_t)n * C * ZP;
  const float* p1 = z1 + (int64_t)n * C * ZP;
  float* pg = dz0 ? dz0 + (int64_t)n * C * ZP : nullptr;
  const float eps = 1e-10f;
  float acc = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const int64_t zi = zpad ? (int64_t)(i / W + zpad) * ZW + (i % W) + zpad : i;
    float s0 = 0.f, s1 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float f0 = fmaxf(p0[(int64_t)c * ZP + zi], 0.f), f1 = fmaxf(p1[(int64_t)c * ZP + zi], 0.f);
      s0 = fmaf(f0, f0, s0);
      s1 = fmaf(f1, f1, s1);
    }
    const float r0 = sqrtf(s0), r1 = sqrtf(s1);
    const float i0 = 1.f / (r0 + eps), i1 = 1.f / (r1 + eps);
    float d = 0.f, S = 0.f;
    for (int c = 0; c < C; ++c) {
      const float f0 = fmaxf(p0[(int64_t)c * ZP + zi], 0.f), f1 = fmaxf(p1[(int64_t)c * ZP + zi], 0.f);
      const float t = f0 * i0 - f1 * i1;
      const float wt = w[c] * t;
      d = fmaf(wt, t, d);
      S = fmaf(wt, f0, S);
    }
    acc += d;
    if (pg) {
      // d a_k / d f0_c = delta_kc / (r + eps) - f0_k f0_c / ((r + eps)^2 r);  sum_k 2 w_k t_k (.) = 2 w_c t_c i0 - 2 S f0_c i0^2 / r
      const float k1 = 2.f * grad_coeff / (float)HW * i0;
      const float k2 = r0 > 0.f ? 2.f * grad_coeff / (float)HW * S * i0 * i0 / r0 : 0.f;
      for (int c = 0; c < C; ++c) {
        const float f0 = fmaxf(p0[(int64_t)c * ZP + zi], 0.f), f1 = fmaxf(p1[(int64_t)c * ZP + zi], 0.f);
        const float t = f0 * i0 - f1 * i1;
        pg[(int64_t)c * ZP + zi] = f0 > 0.f ? k1 * w[c] * t - k2 * f0 : 0.f;
      }
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) loss_add64(loss, (double)acc * (double)coeff / (double)HW);
}

// The same head for the VGG channel counts (64 | 128 | 256 | 512), ONE pass over the features (round 4): a workgroup takes 64 pixels
// (one per lane), its NW waves split the channels (CPW each) and keep their slice of f0 / f1 in registers; the two channel reductions
// (|f|^2, then d and S) cross the waves through LDS in a fixed order.  The generic kernel above reads every feature three times.
template <int CPW, int NW>
__global__ __launch_bounds__(64 * NW) void lpips_layer_regs_kernel(const float* __restrict__ z0, const float* __restrict__ z1, int HW,
                                                                    const float* __restrict__ w, float coeff, long long* __restrict__ loss,
                                                                    float* __restrict__ dz0, float grad_coeff, int W, int zpad) {
  constexpr int C = CPW * NW;
  __shared__ float part[4][NW][64];
  __shared__ float red[16];
  const int n = blockIdx.y, lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ZW = W + 2 * zpad;
  const int64_t ZP = zpad ? (int64_t)(HW / W + 2 * zpad) * ZW : HW;
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < HW;
  const int ic = live ? i : HW - 1;
  const int64_t zi = zpad ? (int64_t)(ic / W + zpad) * ZW + (ic % W) + zpad : ic;
  const int64_t base = ((int64_t)n * C + wv * CPW) * ZP + zi;
  const float* p0 = z0 + base;
  const float* p1 = z1 + base;
  const float eps = 1e-10f;
  float f0[CPW], f1[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    f0[c] = fmaxf(p0[(int64_t)c * ZP], 0.f);
    f1[c] = fmaxf(p1[(int64_t)c * ZP], 0.f);
  }
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    s0 = fmaf(f0[c], f0[c], s0);
    s1 = fmaf(f1[c], f1[c], s1);
  }
  part[0][wv][lane] = s0;
  part[1][wv][lane] = s1;
  __syncthreads();
  s0 = 0.f; s1 = 0.f;
#pragma unroll
  for (int k = 0; k < NW; ++k) { s0 += part[0][k][lane]; s1 += part[1][k][lane]; }
  const float r0 = sqrtf(s0), r1 = sqrtf(s1);
  const float i0 = 1.f / (r0 + eps), i1 = 1.f / (r1 + eps);
  float d = 0.f, S = 0.f;
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const float t = f0[c] * i0 - f1[c] * i1;
    const float wt = w[wv * CPW + c] * t;
    d = fmaf(wt, t, d);
    S = fmaf(wt, f0[c], S);
  }
  part[2][wv][lane] = d;
  part[3][wv][lane] = S;
  __syncthreads();
  d = 0.f; S = 0.f;
#pragma unroll
  for (int k = 0; k < NW; ++k) { d += part[2][k][lane]; S += part[3][k][lane]; }
  if (dz0 && live) {
    float* pg = dz0 + base;
    const float k1 = 2.f * grad_coeff / (float)HW * i0;
    const float k2 = r0 > 0.f ? 2.f * grad_coeff / (float)HW * S * i0 * i0 / r0 : 0.f;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const float t = f0[c] * i0 - f1[c] * i1;
      pg[(int64_t)c * ZP] = f0[c] > 0.f ? k1 * w[wv * CPW + c] * t - k2 * f0[c] : 0.f;
    }
  }
  const float acc = block_sum(live && wv == 0 ? d : 0.f, red);
  if (threadIdx.x == 0 && loss) loss_add64(loss, (double)acc * (double)coeff / (double)HW);
}

// L1 between relu(za) and relu(zb) (VGGLoss: nn.L1Loss on ReLU outputs): value into the fixed-point slot, gradient w.r.t. relu(za)
__global__ __launch_bounds__(256) void l1_relu_kernel(const float* __restrict__ za, const float* __restrict__ zb, int64_t n, float coeff,
                                                      long long* __restrict__ loss, float* __restrict__ grad) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = fmaxf(za[i], 0.f) - fmaxf(zb[i], 0.f);
    acc += fabsf(d);
    if (grad) grad[i] = coeff * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0 && loss) loss_add64(loss, (double)acc * (double)coeff);
}

// dx[n][c'] = sum over the 3 network channels that image channel c' feeds of g[n][c] * s[c]:  Cx = 3: one to one;
// Cx = 1 (a tactile channel broadcast to three network channels by the scaling layer): the sum of the three
__global__ __launch_bounds__(256) void lpips_input_bwd_kernel(const float* __restrict__ g, int HW, int Cx, float s0, float s1, float s2,
                                                              float* __restrict__ dx, int64_t dx_nstride, int accumulate) {
  const int n = blockIdx.y;
  const float* gi = g + (int64_t)n * 3 * HW;
  float* o = dx + n * dx_nstride;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const float a = gi[i] * s0, b = gi[HW + i] * s1, c = gi[2 * (int64_t)HW + i] * s2;
    if (Cx == 1) {
      o[i] = (accumulate ? o[i] : 0.f) + (a + b + c);
    } else {
      o[i] = (accumulate ? o[i] : 0.f) + a;
      o[HW + i] = (accumulate ? o[HW + i] : 0.f) + b;
      o[2 * (int64_t)HW + i] = (accumulate ? o[2 * (int64_t)HW + i] : 0.f) + c;
    }
  }
}

// y[n][c] = (x[n][Cx == 1 ? 0 : c] - shift_c) / scale_c: the LPIPS scaling layer, materialised (3 channels)
__global__ __launch_bounds__(256) void lpips_input_kernel(const float* __restrict__ x, int64_t x_nstride, int HW, int Cx, float a0, float a1, float a2,
                                                          float b0, float b1, float b2, float* __restrict__ y) {
  const int n = blockIdx.y;
  const float* xi = x + n * x_nstride;
  float* o = y + (int64_t)n * 3 * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const float v0 = xi[i], v1 = Cx == 1 ? v0 : xi[HW + i], v2 = Cx == 1 ? v0 : xi[2 * (int64_t)HW + i];
    o[i] = fmaf(v0, a0, b0);
    o[HW + i] = fmaf(v1, a1, b1);
    o[2 * (int64_t)HW + i] = fmaf(v2, a2, b2);
  }
}

inline unsigned blocks_1d(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int vts_maxpool2_relu_pad(const float* z, int NC, int H, int W, int pad, float* out, int zpad, void* stream) {
  VTS_CHECK_ARG(z && out && NC >= 1 && H >= 2 && W >= 2 && (pad == 0 || pad == 1) && NC <= 65535 * 16 && zpad >= 0 && zpad <= 2, "vts_maxpool2_relu_pad: bad args");
  const int PH = H / 2 + 2 * pad, PW = W / 2 + 2 * pad;
  for (int c0 = 0; c0 < NC; c0 += 65535) {
    const int nc = NC - c0 < 65535 ? NC - c0 : 65535;
    hipLaunchKernelGGL(maxpool2_relu_pad_kernel, dim3(blocks_1d((int64_t)PH * PW), nc), dim3(256), 0, (hipStream_t)stream, z + (int64_t)c0 * (H + 2 * zpad) * (W + 2 * zpad), H, W, pad,
                       out + (int64_t)c0 * PH * PW, zpad);
  }
  VTS_CHECK_LAUNCH("vts_maxpool2_relu_pad");
  return VTS_OK;
}

extern "C" int vts_maxpool3s2_relu_pad(const float* z, int NC, int H, int W, int pad, float* out, void* stream) {
  VTS_CHECK_ARG(z && out && NC >= 1 && H >= 3 && W >= 3 && pad >= 0 && pad <= 2, "vts_maxpool3s2_relu_pad: bad args");
  const int PH = (H - 3) / 2 + 1 + 2 * pad, PW = (W - 3) / 2 + 1 + 2 * pad;
  for (int c0 = 0; c0 < NC; c0 += 65535) {
    const int nc = NC - c0 < 65535 ? NC - c0 : 65535;
    hipLaunchKernelGGL(maxpool3s2_relu_pad_kernel, dim3(blocks_1d((int64_t)PH * PW), nc), dim3(256), 0, (hipStream_t)stream, z + (int64_t)c0 * H * W, H, W, pad,
                       out + (int64_t)c0 * PH * PW);
  }
  VTS_CHECK_LAUNCH("vts_maxpool3s2_relu_pad");
  return VTS_OK;
}

extern "C" int vts_s2d4_pad(const float* x, int N, int C, int H, int W, int pad, int OH, int OW, float* out, void* stream) {
  VTS_CHECK_ARG(x && out && N >= 1 && N <= 65535 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && OH >= 1 && OW >= 1, "vts_s2d4_pad: bad args");
  hipLaunchKernelGGL(s2d4_pad_kernel, dim3(blocks_1d((int64_t)C * 16 * OH * OW), N), dim3(256), 0, (hipStream_t)stream, x, C, H, W, pad, OH, OW, out);
  VTS_CHECK_LAUNCH("vts_s2d4_pad");
  return VTS_OK;
}

extern "C" int vts_maxpool2_relu_bwd(const float* g, const float* z, int NC, int H, int W, float* gz, int zpad, const float* g2, int pad,
                                     void* stream) {
  VTS_CHECK_ARG(g && z && gz && NC >= 1 && H >= 2 && W >= 2 && zpad >= 0 && zpad <= 2 && pad >= 0 && pad <= 2, "vts_maxpool2_relu_bwd: bad args");
  const int PH = H + 2 * pad, PW = W + 2 * pad;
  for (int c0 = 0; c0 < NC; c0 += 65535) {
    const int nc = NC - c0 < 65535 ? NC - c0 : 65535;
    const int64_t zo = (int64_t)c0 * (H + 2 * zpad) * (W + 2 * zpad);
    static const int per_element = vts_tune_set("VTS_MAXPOOL_BWD_ELEM") ? 1 : 0;
    if (!per_element && H % 2 == 0 && W % 2 == 0)
      hipLaunchKernelGGL(maxpool2_relu_bwd_win_kernel, dim3(blocks_1d((int64_t)(H / 2) * (W / 2)), nc), dim3(256), 0, (hipStream_t)stream,
                         g + (int64_t)c0 * (H / 2) * (W / 2), z + zo, H, W, gz + (int64_t)c0 * PH * PW, zpad, g2 ? g2 + zo : nullptr, pad);
    else
      hipLaunchKernelGGL(maxpool2_relu_bwd_kernel, dim3(blocks_1d((int64_t)PH * PW), nc), dim3(256), 0, (hipStream_t)stream,
                         g + (int64_t)c0 * (H / 2) * (W / 2), z + zo, H, W, gz + (int64_t)c0 * PH * PW, zpad, g2 ? g2 + zo : nullptr, pad);
  }
  VTS_CHECK_LAUNCH("vts_maxpool2_relu_bwd");
  return VTS_OK;
}

extern "C" int vts_relu_mask_pad(const float* g, const float* g2, const float* z, int NC, int H, int W, int pad, float* out, int zpad, void* stream) {
  VTS_CHECK_ARG((g || g2) && z && out && NC >= 1 && H >= 1 && W >= 1 && pad >= 0 && pad <= 2 && zpad >= 0 && zpad <= 2, "vts_relu_mask_pad: bad args");
  const int PH = H + 2 * pad, PW = W + 2 * pad;
  for (int c0 = 0; c0 < NC; c0 += 65535) {
    const int nc = NC - c0 < 65535 ? NC - c0 : 65535;
    const int64_t o = (int64_t)c0 * H * W, zo = (int64_t)c0 * (H + 2 * zpad) * (W + 2 * zpad);
    hipLaunchKernelGGL(relu_mask_pad_kernel, dim3(blocks_1d((int64_t)PH * PW), nc), dim3(256), 0, (hipStream_t)stream, g ? g + o : nullptr,
                       g2 ? g2 + zo : nullptr, z + zo, H, W, pad, out + (int64_t)c0 * PH * PW, zpad);
  }
  VTS_CHECK_LAUNCH("vts_relu_mask_pad");
  return VTS_OK;
}

extern "C" int vts_lpips_layer(const float* z0, const float* z1, int N, int C, int HW, const float* w, float coeff, int64_t* loss_slot,
                               float* dz0, float grad_coeff, int W, int zpad, void* stream) {
  VTS_CHECK_ARG(z0 && z1 && w && N >= 1 && N <= 65535 && C >= 1 && HW >= 1 && zpad >= 0 && zpad <= 2 && (zpad == 0 || (W >= 1 && HW % W == 0)),
                "vts_lpips_layer: bad args");
  static const int generic = vts_tune_set("VTS_LPIPS_GENERIC") ? 1 : 0;
  const int Wm = W > 0 ? W : HW;
  auto* slot = reinterpret_cast<long long*>(loss_slot);
  const dim3 grid((HW + 63) / 64, N);
  hipStream_t st = (hipStream_t)stream;
  if (!generic && C == 64) hipLaunchKernelGGL((lpips_layer_regs_kernel<16, 4>), grid, dim3(256), 0, st, z0, z1, HW, w, coeff, slot, dz0, grad_coeff, Wm, zpad);
  else if (!generic && C == 128) hipLaunchKernelGGL((lpips_layer_regs_kernel<32, 4>), grid, dim3(256), 0, st, z0, z1, HW, w, coeff, slot, dz0, grad_coeff, Wm, zpad);
  else if (!generic && C == 256) hipLaunchKernelGGL((lpips_layer_regs_kernel<64, 4>), grid, dim3(256), 0, st, z0, z1, HW, w, coeff, slot, dz0, grad_coeff, Wm, zpad);
  else if (!generic && C == 512) hipLaunchKernelGGL((lpips_layer_regs_kernel<64, 8>), grid, dim3(512), 0, st, z0, z1, HW, w, coeff, slot, dz0, grad_coeff, Wm, zpad);
  else
    hipLaunchKernelGGL(lpips_layer_kernel, dim3(blocks_1d(HW), N), dim3(256), 0, st, z0, z1, C, HW, w, coeff, slot, dz0, grad_coeff, Wm, zpad);
  VTS_CHECK_LAUNCH("vts_lpips_layer");
  return VTS_OK;
}

extern "C" int vts_l1_relu(const float* za, const float* zb, int64_t n, float coeff, int64_t* loss_slot, float* grad, void* stream) {
  VTS_CHECK_ARG(za && zb && n >= 1, "vts_l1_relu: bad args");
  hipLaunchKernelGGL(l1_relu_kernel, dim3(blocks_1d(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, za, zb, n, coeff,
                     reinterpret_cast<long long*>(loss_slot), grad);
  VTS_CHECK_LAUNCH("vts_l1_relu");
  return VTS_OK;
}

extern "C" int vts_lpips_input(const float* x, int64_t x_nstride, int N, int Cx, int HW, const float* shift3, const float* scale3, float* y, void* stream) {
  VTS_CHECK_ARG(x && y && shift3 && scale3 && N >= 1 && N <= 65535 && (Cx == 1 || Cx == 3) && HW >= 1, "vts_lpips_input: bad args (host shift / scale triples)");
  hipLaunchKernelGGL(lpips_input_kernel, dim3(blocks_1d(HW), N), dim3(256), 0, (hipStream_t)stream, x, x_nstride, HW, Cx, 1.f / scale3[0], 1.f / scale3[1],
                     1.f / scale3[2], -shift3[0] / scale3[0], -shift3[1] / scale3[1], -shift3[2] / scale3[2], y);
  VTS_CHECK_LAUNCH("vts_lpips_input");
  return VTS_OK;
}

extern "C" int vts_lpips_input_bwd(const float* g, int N, int Cx, int HW, const float* scale3, float* dx, int64_t dx_nstride, int accumulate, void* stream) {
  VTS_CHECK_ARG(g && dx && scale3 && N >= 1 && N <= 65535 && (Cx == 1 || Cx == 3) && HW >= 1, "vts_lpips_input_bwd: bad args");
  hipLaunchKernelGGL(lpips_input_bwd_kernel, dim3(blocks_1d(HW), N), dim3(256), 0, (hipStream_t)stream, g, HW, Cx, 1.f / scale3[0], 1.f / scale3[1],
                     1.f / scale3[2], dx, dx_nstride, accumulate);
  VTS_CHECK_LAUNCH("vts_lpips_input_bwd");
  return VTS_OK;
}
