// Network-level entry point (`vts_unet_fwd`): the inference forward of the reference's generator
//   CustomUnetGenerator.forward   models/networks.py:1430-1645
//   Down / Up                     thirdparty/unet/unet_parts_custom.py:9-37, 40-79
// as ONE C call over the library's own operators, for hosts that are not Python (the Python product drives the same operators from
// vts/engine.py:unet_forward, which also keeps what a backward needs; the two produce bit-identical outputs:
// tests/test_network_abi_gpu.py).  The schedule:
//   down_i (i = 0 .. nd-1)   [LeakyReLU(0.2) ->] Conv2d(4, 2, 1) [-> InstanceNorm2d]     (down0: convolution only; the innermost: no norm)
//   up_i   (i = nd-1 .. 0)   ReLU -> ConvTranspose2d(4, 2, 1) on cat(x, skip_i) [-> InstanceNorm2d]; up0: Tanh
//   layers nls-1 .. 0 exist twice (visual branch / tactile branch `_T`), both fed by up_nls's output
// Nothing is materialised between the layers but the RAW convolution outputs: LeakyReLU / ReLU, the InstanceNorm scale / shift and the
// skip concatenation are applied by the consuming convolution on load (vts_conv4x4's operand pairs), the statistics come out of the
// producing convolution's epilogue (vts_conv4x4_norm).  No allocation: the caller passes vts_unet_forward_ws_floats(d) floats.
#include <stdint.h>

#include <algorithm>

#include "vts_internal.h"

#define VTS_CHECK_HIP(x)                                                        \
  do {                                                                         \
    hipError_t e__ = (x);                                                      \
    if (e__ != hipSuccess) {                                                   \
      vts_set_error("vts_unet_forward: %s: %s", #x, hipGetErrorString(e__));   \
      return VTS_ERR_LAUNCH;                                                   \
    }                                                                          \
  } while (0)

namespace {

struct Plan {
  int64_t act_off[VTS_UNET_MAX_DOWNS];     // raw encoder outputs
  int64_t upact_off[2][VTS_UNET_MAX_DOWNS];  // raw decoder outputs per branch (layer 0 writes d->out)
  int64_t stat_off[3][VTS_UNET_MAX_DOWNS];   // [4][N*C] scale, shift, mean, rstd: encoder, decoder branch 0 / 1
  int64_t conv_ws[2], conv_ws_floats;      // k-split partials / stand-alone statistics scratch (never live together), one per lane
  int64_t stat_ws[2], stat_ws_floats;      // epilogue statistics partials, one per lane
  int64_t total;
};

int up_cout(const vts_unet_desc* d, int branch, int i) { return branch ? d->upT_cout[i] : d->up_cout[i]; }

// the convolution descriptors of the schedule, in launch order: shared by the workspace planner and the launcher
struct Layer {
  vts_conv_desc c;
  bool normed;
  int C;                  // output channels
  int64_t stat;           // offset of its statistics block (normed)
  int lane;               // 1: a layer of the tactile branch (runs on d->side_stream when given)
};

int build(const vts_unet_desc* d, float* ws, const Plan& P, Layer* L, int* count) {
  const int nd = d->num_downs, nls = d->num_layer_separate;
  const int N = d->N;
  int n = 0;
  auto stats_operand = [&](float* data, int C, int64_t hw, int64_t stat_off, bool normed) {
    vts_operand o{};
    o.data = data; o.C = C; o.nstride = (int64_t)C * hw;
    if (normed) { o.scale = ws + stat_off; o.shift = ws + stat_off + (int64_t)N * C; }
    return o;
  };
  vts_operand feat[VTS_UNET_MAX_DOWNS];
  for (int i = 0; i < nd; ++i) {
    Layer& l = L[n++];
    l = Layer{};
    const int oh = d->H >> (i + 1), ow = d->W >> (i + 1);
    vts_conv_desc& c = l.c;
    if (i == 0) { c.in0 = d->in0; c.in1 = d->in1; } else c.in0 = feat[i - 1];
    const int cin = c.in0.C + c.in1.C;
    c.N = N; c.IH = d->H >> i; c.IW = d->W >> i; c.OH = oh; c.OW = ow; c.Cout = d->channels[i];
    c.stride = 2; c.pad = 1; c.transposed = 0;
    c.w = d->down_w[i]; c.ws_co = cin * 16; c.ws_ci = 16; c.bias = d->down_b[i];
    c.out = ws + P.act_off[i]; c.out_nstride = (int64_t)c.Cout * oh * ow;
    c.act_in = i ? VTS_ACT_LRELU : VTS_ACT_NONE; c.act_out = VTS_ACT_NONE;
    l.normed = i > 0 && i < nd - 1; l.C = c.Cout; l.stat = P.stat_off[0][i];
    feat[i] = stats_operand(c.out, c.Cout, (int64_t)oh * ow, l.stat, l.normed);
  }
  const int out_c = d->up_cout[0] + (nls > 0 ? d->upT_cout[0] : 0);
  vts_operand trunk = feat[nd - 1];
  for (int branch = 0; branch < (nls > 0 ? 2 : 1); ++branch) {
    vts_operand x = trunk;
    // branch 0 walks the shared trunk (nd-1 .. nls) and then its own layers; branch 1 only its own layers, from the trunk's end
    for (int i = (branch ? nls - 1 : nd - 1); i >= 0; --i) {
      Layer& l = L[n++];
      l = Layer{};
      const int ih = d->H >> (i + 1), iw = d->W >> (i + 1);
      const bool own = i < nls;
      const int b = own ? branch : 0;
      vts_conv_desc& c = l.c;
      c.in0 = x;
      if (i != 0 && i != nd - 1) c.in1 = feat[i];                 // skip connection: torch.cat([x, skip], 1)
      else if (i == nd - 1 && d->style.C > 0) c.in1 = d->style;   // the tiled style code enters at the innermost block
      const int cout = up_cout(d, b, i);
      c.N = N; c.IH = ih; c.IW = iw; c.OH = 2 * ih; c.OW = 2 * iw; c.Cout = cout;
      c.stride = 2; c.pad = 1; c.transposed = 1;
      c.w = b ? d->upT_w[i] : d->up_w[i]; c.ws_co = 16; c.ws_ci = cout * 16; c.bias = b ? d->upT_b[i] : d->up_b[i];
      if (i == 0) {
        c.out = d->out + (b ? (int64_t)d->up_cout[0] * d->H * d->W : 0); c.out_nstride = (int64_t)out_c * d->H * d->W;
      } else {
        c.out = ws + P.upact_off[b][i]; c.out_nstride = (int64_t)cout * c.OH * c.OW;
      }
      c.act_in = VTS_ACT_RELU; c.act_out = i == 0 ? VTS_ACT_TANH : VTS_ACT_NONE;
      l.normed = i != 0; l.C = cout; l.stat = P.stat_off[1 + b][i]; l.lane = branch;
      x = stats_operand(c.out, cout, (int64_t)c.OH * c.OW, l.stat, l.normed);
      if (!branch && i == nls) trunk = x;
    }
  }
  *count = n;
  return VTS_OK;
}

int check(const vts_unet_desc* d) {
  VTS_CHECK_ARG(d, "vts_unet_forward: null descriptor");
  const int nd = d->num_downs, nls = d->num_layer_separate;
  VTS_CHECK_ARG(nd >= 2 && nd <= VTS_UNET_MAX_DOWNS && nls >= 0 && nls < nd, "vts_unet_forward: num_downs %d / num_layer_separate %d", nd, nls);
  VTS_CHECK_ARG(d->N >= 1 && d->H >= 1 && d->W >= 1 && d->H % (1 << nd) == 0 && d->W % (1 << nd) == 0,
                "vts_unet_forward: H, W must be divisible by %d, got %dx%d", 1 << nd, d->H, d->W);
  VTS_CHECK_ARG(d->in0.data && d->in0.C >= 1 && (d->in1.C == 0 || d->in1.data) && d->out, "vts_unet_forward: null input / output");
  for (int i = 0; i < nd; ++i) {
    VTS_CHECK_ARG(d->channels[i] >= 1 && d->down_w[i] && d->up_w[i] && d->up_cout[i] >= 1, "vts_unet_forward: layer %d incomplete", i);
    VTS_CHECK_ARG(i >= nls || (d->upT_w[i] && d->upT_cout[i] >= 1), "vts_unet_forward: tactile branch layer %d incomplete", i);
    // the decoder mirrors the encoder: up_i's input is cat(up_{i+1} output, skip_i)
    VTS_CHECK_ARG(i == 0 || d->up_cout[i] == d->channels[i - 1], "vts_unet_forward: up%d emits %d channels, down%d has %d", i, d->up_cout[i], i - 1,
                  d->channels[i - 1]);
    VTS_CHECK_ARG(i == 0 || i >= nls || d->upT_cout[i] == d->channels[i - 1], "vts_unet_forward: up%d_T emits %d channels, down%d has %d", i,
                  d->upT_cout[i], i - 1, d->channels[i - 1]);
  }
  VTS_CHECK_ARG(d->style.C == 0 || d->style.data, "vts_unet_forward: style operand without data");
  return VTS_OK;
}

int plan(const vts_unet_desc* d, Plan& P) {
  const int nd = d->num_downs, nls = d->num_layer_separate;
  const int64_t N = d->N;
  int64_t off = 0;
  auto take = [&](int64_t n) { const int64_t o = off; off += (n + 63) / 64 * 64; return o; };
  for (int i = 0; i < nd; ++i) {
    const int64_t hw = (int64_t)(d->H >> (i + 1)) * (d->W >> (i + 1));
    P.act_off[i] = take(N * d->channels[i] * hw);
    P.stat_off[0][i] = take(4 * N * d->channels[i]);
  }
  for (int b = 0; b < 2; ++b)
    for (int i = 1; i < nd; ++i) {
      P.upact_off[b][i] = P.stat_off[1 + b][i] = 0;
      if (b && i >= nls) continue;
      const int64_t hw = (int64_t)(d->H >> i) * (d->W >> i);
      const int c = up_cout(d, b, i);
      P.upact_off[b][i] = take(N * c * hw);
      P.stat_off[1 + b][i] = take(4 * N * c);
    }
  P.upact_off[0][0] = P.upact_off[1][0] = P.stat_off[1][0] = P.stat_off[2][0] = 0;
  // scratch: sized over the schedule's descriptors (built against a null workspace: only shapes matter)
  Layer L[3 * VTS_UNET_MAX_DOWNS];
  int n = 0;
  build(d, reinterpret_cast<float*>(uintptr_t(4096)), P, L, &n);    // (a placeholder base: only the shapes are read)
  int64_t cw = 0, sw = 0;
  for (int k = 0; k < n; ++k) {
    const vts_conv_desc& c = L[k].c;
    if ((int64_t)c.OH * c.OW <= 64 * 64) cw = std::max(cw, vts_conv4x4_ws_floats(&c));
    if (L[k].normed) {
      sw = std::max(sw, vts_conv4x4_norm_ws_floats(&c));
      cw = std::max(cw, vts_norm_ws_floats(c.N, c.Cout, c.OH * c.OW));
    }
  }
  P.conv_ws_floats = cw; P.stat_ws_floats = sw;
  for (int lane = 0; lane < 2; ++lane) {
    P.conv_ws[lane] = take(cw);
    P.stat_ws[lane] = take(sw);
  }
  P.total = off;
  return VTS_OK;
}

}  // namespace

extern "C" int64_t vts_unet_forward_ws_floats(const vts_unet_desc* d) {
  if (check(d) != VTS_OK) return -1;
  Plan P{};
  plan(d, P);
  return P.total;
}

// fork / join events of the two-lane form (created once per host thread; recorded and waited on inside the caller's stream order, so the
// call stays capturable into a HIP graph)
static int lane_events(hipEvent_t* fork, hipEvent_t* join) {
  static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  if (!ev[0]) {
    VTS_CHECK_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    VTS_CHECK_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  }
  *fork = ev[0]; *join = ev[1];
  return VTS_OK;
}

extern "C" int vts_unet_forward(const vts_unet_desc* d, float* ws, int64_t ws_floats, void* stream) {
  const int rc0 = check(d);
  if (rc0 != VTS_OK) return rc0;
  Plan P{};
  plan(d, P);
  VTS_CHECK_ARG(ws && ws_floats >= P.total, "vts_unet_forward: workspace of %lld floats, need %lld", (long long)ws_floats, (long long)P.total);
  Layer L[3 * VTS_UNET_MAX_DOWNS];
  int n = 0;
  build(d, ws, P, L, &n);
  const bool lanes = d->side_stream && d->side_stream != stream && d->num_layer_separate > 0;
  hipEvent_t fork = nullptr, join = nullptr;
  if (lanes) {
    const int rc = lane_events(&fork, &join);
    if (rc != VTS_OK) return rc;
  }
  bool forked = false;
  // an error after the fork still joins the side stream (inside a graph capture a dangling fork invalidates the capture; outside, the
  // side stream would be left unordered with the caller's stream)
  auto finish = [&](int rc) {
    if (forked && (hipEventRecord(join, (hipStream_t)d->side_stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)stream, join, 0) != hipSuccess) && rc == VTS_OK) {
      vts_set_error("vts_unet_forward: joining the side stream failed");
      return (int)VTS_ERR_LAUNCH;
    }
    return rc;
  };
  for (int k = 0; k < n; ++k) {
    vts_conv_desc& c = L[k].c;
    // the visual branch's own layers follow the trunk on `stream`; the tactile branch (all at the end of the list) runs on the side
    // stream from the trunk's last layer on.  The fork is recorded when the first own layer of the visual branch is reached.
    const bool own0 = lanes && !forked && k >= d->num_downs + (d->num_downs - d->num_layer_separate);
    if (own0) {
      VTS_CHECK_HIP(hipEventRecord(fork, (hipStream_t)stream));
      forked = true;
      if (hipStreamWaitEvent((hipStream_t)d->side_stream, fork, 0) != hipSuccess) {
        forked = false;
        vts_set_error("vts_unet_forward: forking the side stream failed");
        return VTS_ERR_LAUNCH;
      }
    }
    const int lane = lanes ? L[k].lane : 0;
    void* st = lane ? d->side_stream : stream;
    if ((int64_t)c.OH * c.OW <= 64 * 64) { c.ws = ws + P.conv_ws[lane]; c.ws_floats = P.conv_ws_floats; }
    if (!L[k].normed) {
      const int rc = vts_conv4x4(&c, st);
      if (rc != VTS_OK) return finish(rc);
      continue;
    }
    vts_norm_desc nd{};
    const int64_t NC = (int64_t)c.N * L[k].C;
    float* stt = ws + L[k].stat;
    nd.x = c.out; nd.nstride = c.out_nstride; nd.N = c.N; nd.C = L[k].C; nd.HW = c.OH * c.OW; nd.mode = 0;
    nd.eps = 1e-5f; nd.momentum = 0.1f;          // nn.InstanceNorm2d defaults (models/networks.py:139)
    nd.scale = stt; nd.shift = stt + NC; nd.mean_out = stt + 2 * NC; nd.rstd_out = stt + 3 * NC;
    int fused = 0;
    int rc = vts_conv4x4_norm(&c, &nd, ws + P.stat_ws[lane], P.stat_ws_floats, &fused, st);
    if (rc != VTS_OK) return finish(rc);
    if (fused >= 2) rc = vts_norm_stats_from_partials(&nd, ws + P.stat_ws[lane], fused - 2, st);
    else if (fused == 0) rc = vts_norm_stats(&nd, ws + P.conv_ws[lane], st);
    if (rc != VTS_OK) return finish(rc);
  }
  return finish(VTS_OK);
}
