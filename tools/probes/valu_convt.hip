// Feasibility probe (round 4, NEGATIVE): ConvTranspose2d 4x4 stride 2 pad 1 with FEW output channels on the packed-fp32 VALU instead of MFMA.
// The fp32 vector peak of the part equals its fp32 matrix peak (v_pk_fma_f32: 2 FMA / lane / issue), and a VALU kernel pays no padding
// of Cout = 10 / 20 to the MFMA tile's 16 / 32 -- the 37.5 % the convolution template loses on the decoder's outer layers.
// One lane = one input-aligned position -> its 2 x 2 output pixels x Cout accumulators; inputs through an LDS tile; weights wave-uniform,
// so the compiler feeds them as SGPR pairs straight into v_pk_fma_f32 (checked in the ISA: `v_pk_fma_f32 v[a:b], v[c:d], s[e:f], v[a:b] op_sel_hi:[0,1,1]`).
// Measured on the MI355X (plain input, ReLU on load, bias; no statistics epilogue):
//     40 -> 10, N4 256^2 -> 512^2 :  81 -  97 us = 35 - 42 TFLOP/s   (the MFMA template with its full epilogue: 88 us, 38 TFLOP/s)
//     80 -> 20, N4 128^2 -> 256^2 : 403 - 416 us =  8 TFLOP/s        (100 KB of weights stream through the 16 KB scalar cache; 179 us with cache-resident weights)
//     20 ->  6, N4 512^2 -> 1024^2:  96 - 103 us = 39 - 42 TFLOP/s
// i.e. 0.22 - 0.27 of the vector peak: ~ 18 cycles per v_pk_fma_f32 instead of 4.  Neither the LDS reads (ABL 2) nor scalar-cache misses
// (ABL 1) are the bound for the 10-channel case, and two rows per lane (P = 2: half the weight traffic per FMA) is SLOWER; the schedule is
// `s_load_dwordx16 ... s_waitcnt lgkmcnt(0)` every ~ 12 FMAs -- the scalar operand delivery (lgkmcnt is shared with the LDS reads and returns
// out of order, so every wait drains everything) paces the wave.  The MFMA pipe exists to avoid exactly this operand-delivery problem: dropped.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_convt.hip -o /tmp/valu_convt && /tmp/valu_convt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// one lane = one input-aligned position (y, x) -> the 2 x 2 output pixels (2y + py, 2x + px), all COUT channels
// weights repacked [ci][ph = py * 2 + px][t = dy * 2 + dx][co]; neighbour of (ph, t): row y + py - 1 + dy... see host packing
template <int CIN, int COUT, int P, int CK, int ABL>
__global__ __launch_bounds__(256) void convt_valu(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                  float* __restrict__ out, int H, int W) {
  // workgroup tile: 64 columns x (4 * P) rows of input-aligned positions; wave w owns rows w * P .. w * P + P - 1, lane = column
  constexpr int TW = 64, TH = 4 * P, LW = TW + 2, LH = TH + 2;
  __shared__ float tile[CK][LH][LW + 2];
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x2 acc[P][4][COUT / 2];
#pragma unroll
  for (int r = 0; r < P; ++r)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < COUT / 2; ++c) acc[r][p][c] = f32x2{bias[2 * c], bias[2 * c + 1]};
  const float* inn = in + (size_t)n * CIN * H * W;
  const int xa = x0 + lane - 1;                 // this lane's staging column (tile column `lane`), and tile columns 64 / 65 for lanes 0 / 1
  const int xb = x0 + 63 + lane;
  const bool xa_ok = xa >= 0 && xa < W, xb_ok = lane < 2 && xb < W;
  for (int c0 = 0; c0 < CIN; c0 += CK) {
    __syncthreads();
    for (int rr = wv; rr < CK * LH; rr += 4) {  // one tile row per wave and iteration
      const int c = rr / LH, r = rr % LH;       // (wave-uniform: scalar arithmetic)
      const int yy = y0 + r - 1;
      const bool y_ok = yy >= 0 && yy < H;
      const float* src = inn + ((size_t)(c0 + c) * H + (y_ok ? yy : 0)) * W;
      const float va = (y_ok && xa_ok) ? src[xa] : 0.f;
      tile[c][r][lane] = fmaxf(va, 0.f);
      if (lane < 2) tile[c][r][64 + lane] = fmaxf((y_ok && xb_ok) ? src[xb] : 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const float* wc = wp + (size_t)((ABL & 1) ? (c & 1) : (c0 + c)) * 16 * COUT;
      float nb[P + 2][3];
#pragma unroll
      for (int r = 0; r < P + 2; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) nb[r][q] = (ABL & 2) ? tile[0][r][lane] * (float)(q + c) : tile[c][wv * P + r][lane + q];
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int dy = t >> 1, dx = t & 1;
          const float* w = wc + (ph * 4 + t) * COUT;
#pragma unroll
          for (int co = 0; co < COUT / 2; ++co) {
            const f32x2 ww = f32x2{w[2 * co], w[2 * co + 1]};
#pragma unroll
            for (int r = 0; r < P; ++r) {
              const float v = nb[r + py + dy][px + dx];
              acc[r][ph][co] = __builtin_elementwise_fma(f32x2{v, v}, ww, acc[r][ph][co]);
            }
          }
        }
      }
    }
  }
  const int x = x0 + lane;
#pragma unroll
  for (int r = 0; r < P; ++r) {
    const int y = y0 + wv * P + r;
    if (y < H && x < W) {
      float* o = out + (size_t)n * COUT * 4 * H * W;
#pragma unroll
      for (int co = 0; co < COUT / 2; ++co)
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          f32x2 a = f32x2{acc[r][py * 2][co].x, acc[r][py * 2 + 1][co].x};
          f32x2 b = f32x2{acc[r][py * 2][co].y, acc[r][py * 2 + 1][co].y};
          *reinterpret_cast<f32x2*>(o + ((size_t)(2 * co) * 2 * H + 2 * y + py) * 2 * W + 2 * x) = a;
          *reinterpret_cast<f32x2*>(o + ((size_t)(2 * co + 1) * 2 * H + 2 * y + py) * 2 * W + 2 * x) = b;
        }
    }
  }
}

template <int CIN, int COUT, int P, int ABL = 0>
static void run(int N, int H, int W) {
  constexpr int TW = 64, TH = 4 * P, CK = (CIN % 8 == 0) ? 8 : 4;
  std::vector<float> hin((size_t)N * CIN * H * W), hw((size_t)CIN * COUT * 16), hb(COUT), hwp(hw.size());
  srand(1);
  for (auto& v : hin) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) / 4000.f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 1000.f;
  // ConvTranspose2d weight [ci][co][ky][kx]; out[2y+py] takes ky = 1 (iy = y), 3 (iy = y-1) for py = 0; ky = 2 (iy = y), 0 (iy = y+1) for py = 1.
  // neighbour rows are nb[py + dy] = row y - 1 + py + dy: py=0: dy=0 -> y-1 (ky 3), dy=1 -> y (ky 1); py=1: dy=0 -> y (ky 2), dy=1 -> y+1 (ky 0)
  auto kof = [](int p, int d) { return p == 0 ? (d == 0 ? 3 : 1) : (d == 0 ? 2 : 0); };
  for (int ci = 0; ci < CIN; ++ci)
    for (int ph = 0; ph < 4; ++ph)
      for (int t = 0; t < 4; ++t)
        for (int co = 0; co < COUT; ++co)
          hwp[((size_t)ci * 16 + ph * 4 + t) * COUT + co] = hw[(((size_t)ci * COUT + co) * 4 + kof(ph >> 1, t >> 1)) * 4 + kof(ph & 1, t & 1)];
  float *din, *dw, *db, *dout;
  const size_t on = (size_t)N * COUT * 4 * H * W;
  hipMalloc(&din, hin.size() * 4); hipMalloc(&dw, hwp.size() * 4); hipMalloc(&db, COUT * 4); hipMalloc(&dout, on * 4);
  hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, hwp.data(), hwp.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), COUT * 4, hipMemcpyHostToDevice);
  dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, N);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((convt_valu<CIN, COUT, P, CK, ABL>), grid, dim3(256), 0, 0, din, dw, db, dout, H, W);
  hipEventRecord(e0);
  const int R = 50;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL((convt_valu<CIN, COUT, P, CK, ABL>), grid, dim3(256), 0, 0, din, dw, db, dout, H, W);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / R, fl = 2.0 * N * 4.0 * H * W * COUT * CIN * 4;
  // spot check
  std::vector<float> hout(on);
  hipMemcpy(hout.data(), dout, on * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 2000; ++s) {
    const int n = rand() % N, co = rand() % COUT, oy = rand() % (2 * H), ox = rand() % (2 * W);
    double r = hb[co];
    for (int ci = 0; ci < CIN; ++ci)
      for (int ky = 0; ky < 4; ++ky)
        for (int kx = 0; kx < 4; ++kx) {
          const int ty = oy + 1 - ky, tx = ox + 1 - kx;
          if (ty < 0 || tx < 0 || (ty & 1) || (tx & 1)) continue;
          const int iy = ty / 2, ix = tx / 2;
          if (iy >= H || ix >= W) continue;
          r += std::fmax(hin[(((size_t)n * CIN + ci) * H + iy) * W + ix], 0.f) * (double)hw[(((size_t)ci * COUT + co) * 4 + ky) * 4 + kx];
        }
    worst = std::fmax(worst, std::fabs(r - hout[(((size_t)n * COUT + co) * 2 * H + oy) * 2 * W + ox]));
  }
  printf("ABL=%d P=%d convT %d -> %d, N%d %dx%d: %.1f us  %.1f TFLOP/s  (max err %.2e)\n", ABL, P, CIN, COUT, N, H, W, us, fl / us * 1e-6, worst);
  hipFree(din); hipFree(dw); hipFree(db); hipFree(dout);
}

int main() {
  run<40, 10, 1>(4, 256, 256);
  run<40, 10, 1, 1>(4, 256, 256);
  run<40, 10, 1, 2>(4, 256, 256);
  run<40, 10, 1, 3>(4, 256, 256);
  run<80, 20, 1>(4, 128, 128);
  run<80, 20, 1, 1>(4, 128, 128);
  run<80, 20, 1, 3>(4, 128, 128);
  run<20, 6, 1>(4, 512, 512);
  run<20, 6, 1, 3>(4, 512, 512);
  return 0;
}
