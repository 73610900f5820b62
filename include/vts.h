/*
 * vts.h -- C ABI of libvts_hip.so: the MI355X (gfx950) kernels behind the
 * sketch -> (RGB, tactile) conditional-GAN training step.
 *
 * The reference has NO native/FFI layer on this path: its
 * boundary is the Python class contract models.create_model()/BaseModel, and every
 * op below replaces a PyTorch call made inside that class.  Each entry point cites
 * the reference lines (relative to /root/reference) whose arithmetic it performs.
 *
 * Conventions
 *   - all tensors are device pointers to fp32, NCHW, contiguous unless a batch
 *     stride ("nstride", in floats) is given, which allows channel-sliced views;
 *   - no torch types, no allocation inside any call: the caller owns every buffer
 *     (scratch sizes are returned by the *_ws_floats helpers);
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it;
 *   - return value: 0 = ok, otherwise a negative vts error code; vts_last_error()
 *     returns a thread-local message for the last failing call.
 */
#ifndef VTS_H
#define VTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTS_OK 0
#define VTS_ERR_ARG (-1)
#define VTS_ERR_UNSUPPORTED (-2)
#define VTS_ERR_LAUNCH (-3)

/* activation codes (applied to the gathered input after its affine, or as derivative masks) */
#define VTS_ACT_NONE 0
#define VTS_ACT_LRELU 1 /* LeakyReLU(0.2) */
#define VTS_ACT_RELU 2
#define VTS_ACT_TANH 3 /* epilogue only */

const char* vts_last_error(void);
/* Kernel instance chosen by the last vts_conv4x4 / vts_wgrad4x4 call of this thread (profiling aid). */
const char* vts_last_kernel(void);
int vts_version(void);
/* Number of nodes (and of kernel nodes among them) in the graph `stream` is capturing into, at this point of the capture; both 0
 * when the stream is not capturing.  Measurement aid: the launch count of the REPLAYED step (bench.py `launches_per_step_replayed`),
 * which differs from the eager single-stream schedule's.  No reference counterpart (the reference launches eagerly, train.py:39-66). */
int vts_capture_node_count(void* stream, int* nodes, int* kernel_nodes);

/* A (possibly channel-concatenated, lazily normalised) activation operand:
 *   value(n, c, y, x) = act( data[n*nstride + c*H*W + y*W + x] * scale[n*C + c] + shift[n*C + c] )
 * scale/shift may be NULL (identity).  This is how InstanceNorm / BatchNorm outputs are
 * consumed: the producer stores the raw conv result plus per-(n,c) scale/shift, and every
 * consumer normalises on load (no separate normalisation pass over HBM). */
typedef struct vts_operand {
  const float* data;
  const float* scale;
  const float* shift;
  int C;
  int64_t nstride;
} vts_operand;

/* Convolution-like operator (4x4 kernels).  transposed = 0:
 *   out[n,co,oy,ox] = bias[co] + sum_{ci,ky,kx} in[n,ci,oy*stride+ky-pad,ox*stride+kx-pad] * W(co,ci,ky,kx)
 * transposed = 1 (ConvTranspose2d / backward-data of a stride-`stride` conv):
 *   out[n,co,y,x]   = bias[co] + sum_{ci,ky,kx : (y+pad-ky) % stride == 0} in[n,ci,(y+pad-ky)/stride,(x+pad-kx)/stride] * W(co,ci,ky,kx)
 * with W(co,ci,ky,kx) = w[co*ws_co + ci*ws_ci + ky*4 + kx], so the same weight tensor serves
 * nn.Conv2d ([Cout,Cin,4,4]: ws_co=Cin*16, ws_ci=16), nn.ConvTranspose2d ([Cin,Cout,4,4]:
 * ws_co=16, ws_ci=Cout*16), their backward-data passes and channel sub-ranges (offset w).
 * `in` is the channel concatenation of in0 and in1 (in1.C = 0 if unused), each normalised and
 * activated on load (act_in).  Epilogue: + bias, act_out (NONE|TANH), optional derivative mask
 *   out *= act'( dmask value )      (LRELU: v>0 ? 1 : 0.2; RELU: v>0 ? 1 : 0)
 * where the dmask operand has the shape of `out`, then optional accumulation into `out`.
 *
 * Replaces: F.conv2d / F.conv_transpose2d and their autograd backward-data inside
 * Down/Up (thirdparty/unet/unet_parts_custom.py:9-79), NLayerDiscriminator
 * (models/networks.py:1696-1750), incl. the LeakyReLU/ReLU, torch.cat skip concat,
 * InstanceNorm/BatchNorm application and Tanh that surround them. */
typedef struct vts_conv_desc {
  vts_operand in0, in1;
  int N, IH, IW, OH, OW, Cout;
  int stride, pad, transposed;
  const float* w;
  int ws_co, ws_ci;
  const float* bias; /* [Cout] or NULL */
  float* out;
  int64_t out_nstride;
  int act_in, act_out;
  vts_operand dmask; /* data == NULL: no mask */
  int dmask_act;
  int accumulate;
  float* ws;          /* optional scratch for the small-grid k-split path (may be NULL: path not taken) */
  int64_t ws_floats;  /* vts_conv4x4_ws_floats(d) is always enough */
  int pad_dx;         /* horizontal padding = pad + pad_dx (0: square padding) */
} vts_conv_desc;

int64_t vts_conv4x4_ws_floats(const vts_conv_desc* d);

int vts_conv4x4(const vts_conv_desc* d, void* stream);

/* Weight gradient of the same operator family:
 *   dw[cl*dw_s_cl + ch*dw_s_ch + ky*4+kx] = sum_{n,y,x} lo[n,cl,y,x] * hi[n,ch,y*stride+ky-pad,x*stride+kx-pad]
 * `lo` is the low-resolution side (LH x LW), `hi` the high-resolution side (HH x HW); each is a
 * (dual, normalise-on-load) operand pair.  nn.Conv2d: lo = grad_out, hi = input, dw layout
 * [Cout,Cin,4,4]; nn.ConvTranspose2d: lo = input, hi = grad_out, dw layout [Cin,Cout,4,4].
 * Deterministic: per-workgroup partials in `ws` (vts_wgrad4x4_ws_floats floats) reduced in fixed order.
 * Replaces the autograd weight-gradient of the convolutions listed above. */
typedef struct vts_wgrad_desc {
  vts_operand lo0, lo1, hi0, hi1;
  int act_lo, act_hi;
  int N, LH, LW, HH, HW;
  int stride, pad;
  float* dw;
  int accumulate;
  int pad_dx; /* horizontal padding = pad + pad_dx */
  int defer;  /* 1: leave the per-workgroup partials in `ws` ([pw][CL][CH][16], pw = vts_wgrad4x4_ws_floats / (CL*CH*16)) and do not
                 reduce them: the caller reduces the partials of many weight gradients at once with vts_wgrad_reduce_batch */
} vts_wgrad_desc;

int64_t vts_wgrad4x4_ws_floats(const vts_wgrad_desc* d);
int vts_wgrad4x4(const vts_wgrad_desc* d, float* ws, void* stream);

/* Deferred, batched form of the deterministic partial reduction: dw[i] (+)= sum over the segments, in order, of
 * sum_{k < pw} part[k*nel + i] (fixed summation order, no float atomics).  One launch reduces the partials of every weight
 * gradient of a backward pass (the headline step used to launch ~117 single reductions).  A weight gradient that receives
 * several contributions (accumulation over passes) lists them as segments of ONE job, so jobs never share a dw. */
#define VTS_REDUCE_MAX_SEG 4
typedef struct vts_reduce_job {
  float* dw;
  int64_t nel;
  int accumulate; /* add to the existing dw instead of overwriting it */
  int nseg;
  const float* part[VTS_REDUCE_MAX_SEG];
  int pw[VTS_REDUCE_MAX_SEG];
} vts_reduce_job;
int vts_wgrad_reduce_batch(const vts_reduce_job* jobs /* host array */, int njobs, void* stream);

/* Per-channel sum over (N, H, W): out[c] (+)= sum x[n,c,:,:]  (bias gradients). */
int vts_channel_sum(const float* x, int64_t nstride, int N, int C, int HW, float* out, int accumulate,
                    float* ws, int* counters /* optional, >= C zeroed ints: see vts_norm_desc */, void* stream);
int64_t vts_channel_sum_ws_floats(int N, int C, int HW);

/* Normalisation statistics -> per-(n,c) scale/shift for normalise-on-load.
 * mode 0: InstanceNorm2d(affine=False, eps) (models/networks.py:139): group = (n,c)
 * mode 1: BatchNorm2d training mode (networks.py:137): group = c over (N,H,W); gamma/beta
 *         applied; running_mean/var updated with `momentum` using the unbiased variance and
 *         num_batches_tracked += 1 when the pointers are non-NULL.
 * Robust two-level (Chan) variance.  mean_out / rstd_out [N*C] are kept for the backward. */
typedef struct vts_norm_desc {
  const float* x;
  int64_t nstride;
  int N, C, HW;
  int mode;
  float eps, momentum;
  const float* gamma; /* BN only */
  const float* beta;
  float* running_mean;
  float* running_var;
  int64_t* num_batches_tracked;
  float* scale;    /* [N*C] out */
  float* shift;    /* [N*C] out */
  float* mean_out; /* [N*C] out */
  float* rstd_out; /* [N*C] out */
  int* counters;   /* optional: >= N*C ints of caller-owned device memory, ZERO on entry and left zero; when given, the last
                      workgroup of each group finalises (one launch instead of two: same arithmetic, same order).  One buffer
                      per concurrently running stream.  Measured SLOWER on MI355X inside a busy step (device-scope fences
                      flush the per-XCD L2): leave NULL unless the launch count matters more than the time. */
  /* BatchNorm over SEVERAL forward passes batched into one launch (mode 1 only; ngroups <= 1: one group = the whole batch).
   * The reference runs the discriminator on fake and real batches one after the other (sinskitG_model.py:1361,1374,1490,
   * 1567,1584): each call normalises with its own batch statistics and advances the running statistics once.  Batched,
   * samples [gstart[g], gstart[g+1]) form pass g: statistics, scale/shift and the running-statistics recurrence are per
   * pass, applied in pass order, so the result equals the sequential calls. */
  int ngroups;
  int gstart[9];
  /* running-statistics bookkeeping of a pass that ran as its OWN launch but belongs into the sequence of this one
   * (the full-resolution visualisation pass between the fake and the "more fake" patch passes, sinskitG_model.py:1495):
   *   stat_mean_out / stat_uvar_out [C]: this launch only records its batch mean and unbiased variance (no running update
   *                                      when running_mean is NULL);
   *   ext_mean / ext_uvar [C], ext_after: statistics recorded that way, applied after pass `ext_after` of this launch. */
  float* stat_mean_out;
  float* stat_uvar_out;
  const float* ext_mean;
  const float* ext_uvar;
  int ext_after;
} vts_norm_desc;

int64_t vts_norm_ws_floats(int N, int C, int HW);
int vts_norm_stats(const vts_norm_desc* d, float* ws, void* stream);

/* vts_conv4x4 followed by the InstanceNorm statistics of its output (Down / Up blocks: conv -> InstanceNorm2d,
 * thirdparty/unet/unet_parts_custom.py:24-37, 66-79) with the chance to fuse them: when the convolution takes its k-split path on a
 * map of <= 4096 pixels, the slice-summing epilogue also computes the statistics (one launch instead of two, bit-identical) and
 * *fused = 1; otherwise the convolution runs as vts_conv4x4, *fused = 0 and the caller calls vts_norm_stats.  nd: mode 0, x = d->out. */
int vts_conv4x4_in(const vts_conv_desc* d, const vts_norm_desc* nd, int* fused, void* stream);

/* The general form (round 3): convolution + InstanceNorm (mode 0) or training-mode BatchNorm (mode 1: pass groups, running statistics,
 * recorded / spliced statistics exactly as vts_norm_stats) of its output.  On top of the k-split fusion of vts_conv4x4_in, the tiled
 * kernel's epilogue emits per-wave (mean, M2, count) partials of the values it stores into stat_ws (vts_conv4x4_norm_ws_floats floats)
 * and only the merging second stage of vts_norm_stats runs afterwards: the layer's output is not read again for its statistics
 * (Down / Up blocks as above; Conv2d -> BatchNorm2d of NLayerDiscriminator, models/networks.py:1716-1738).  *fused = 0: the
 * convolution ran as vts_conv4x4 (small-map / thin / head members, output activation, mask or accumulation) and the caller calls
 * vts_norm_stats. */
int64_t vts_conv4x4_norm_ws_floats(const vts_conv_desc* d);
int vts_conv4x4_norm(const vts_conv_desc* d, const vts_norm_desc* nd, float* stat_ws, int64_t stat_ws_floats, int* fused, void* stream);
/* *fused after vts_conv4x4_norm: 0 = plain convolution (call vts_norm_stats); 1 = statistics complete (k-split epilogue); 2 + s = the
 * epilogue wrote s (mean, M2, count) slots per (n, channel) into stat_ws and this second stage merges them into nd's outputs (two
 * entry points so that a caller timing launches sees the convolution and the merge as what they are: two kernels). */
int vts_norm_stats_from_partials(const vts_norm_desc* nd, const float* part, int slots, void* stream);

/* Backward of the same normalisation (in place on dy):
 *   dx = A*dy + B*x + C  with the per-group coefficients of InstanceNorm / BatchNorm backward;
 * BN additionally writes dgamma/dbeta (accumulate flag).  */
typedef struct vts_norm_bwd_desc {
  float* dy; /* in: grad wrt normalised(+affine) output; out: grad wrt raw x */
  const float* x;
  int64_t nstride;
  int N, C, HW;
  int mode;
  const float* mean;
  const float* rstd;
  const float* gamma;
  float* dgamma;
  float* dbeta;
  int accumulate_param_grads;
  int* counters; /* optional, as in vts_norm_desc */
  int ngroups;   /* batched passes, as in vts_norm_desc: coefficients per pass, dgamma / dbeta summed over the passes */
  int gstart[9];
} vts_norm_bwd_desc;

int vts_norm_bwd(const vts_norm_bwd_desc* d, float* ws, void* stream);

/* Normalisation backward fused with the backward-data convolution that produces its input gradient (round 3).  vts_conv4x4_bsums runs
 * vts_conv4x4 (d->dmask = the normalised tensor of the layer below, as in every backward-data call of the U-Net / PatchGAN schedules) and,
 * where the tiled kernel's direct epilogue runs, also emits per wave and tile S1 = sum out, S2' = sum out * (dmask * scale + shift) of the
 * values it stores (after mask and accumulation) into `part` (vts_conv4x4_norm_ws_floats(d) floats are enough); *slots = pairs per
 * (n, channel), 0 = not fused (call vts_norm_bwd).  vts_norm_bwd_from_partials then runs ONLY the apply pass of vts_norm_bwd on them
 * (`beta` = the BatchNorm shift that S2' contains, NULL for InstanceNorm): the partial-sum pass -- one read of dy and of x per
 * normalised layer, and its launch -- disappears from the backward chains.
 * *slots is also an INPUT: -1 tells the convolution that the normalisation is InstanceNorm2d(affine=False) (its scale IS rstd), and where
 * the convolution takes the small-grid k-split path on a map of <= 4096 pixels its slice-summing epilogue then applies that backward
 * itself (one launch instead of two); *slots = -1 on return says so: `out` already holds the gradient w.r.t. the RAW tensor, do not call
 * vts_norm_bwd.  Pass 0 for BatchNorm or to keep the two-call form. */
int vts_conv4x4_bsums(const vts_conv_desc* d, float* part, int64_t part_floats, int* slots, void* stream);
int vts_norm_bwd_from_partials(const vts_norm_bwd_desc* d, const float* part, int slots, const float* beta, void* stream);

/* dy (+)= g * act'(x*scale+shift): derivative mask as a stand-alone op (used where the
 * producer of g is not one of the conv kernels). */
int vts_act_bwd(const float* g, const vts_operand* x, int N, int HW, int act, float* dy, int accumulate, void* stream);

/* ---- ResNet generator building blocks (--netG resnet_{4,6,9}blocks, models/networks.py:1051-1154) ----
 *
 * Padding fused with normalise-on-load and an optional residual:
 *   out[n,c,y,x] = act(in[n,c,map(y-pt),map(x-pl)] * scale[n,c] + shift[n,c]) (+ res[n,c,y,x])
 * `out` is [N, C, H+pt+pb, W+pl+pr] with batch stride out_nstride (0: contiguous; a larger stride writes a
 * channel slice of a wider tensor, i.e. torch.cat on store; `res` is always contiguous); mode 0 zero, 1 reflect (nn.ReflectionPad2d,
 * networks.py:1075,1146,1300), 2 replicate; act may be VTS_ACT_TANH.  With all pads 0 it is the fused
 * `x + norm(conv_block(x))` of ResnetBlock.forward (networks.py:1322) or a plain activation. */
int vts_pad_affine(const vts_operand* in, int N, int H, int W, int pt, int pb, int pl, int pr, int mode, int act,
                   const float* res, float* out, int64_t out_nstride, void* stream);
/* Adjoint of the padding index map: din[n,c,i,j] (+)= sum of dpad over the padded positions that read (i,j). */
int vts_pad_bwd(const float* dpad, int N, int C, int H, int W, int pt, int pb, int pl, int pr, int mode, float* din,
                int accumulate, void* stream);

/* Separable windowed resampling from host-built tables: out[nc, y, x] (+)= sum_j wy[y*KY+j] * sum_i wx[x*KX+i] * in[nc, ymin[y]+j, xmin[x]+i]
 * with j < ysize[y], i < xsize[x] (device int / float arrays).  With the index / weight tables of PyTorch's anti-aliased bicubic filter
 * (cubic a = -0.5, support scaled by the downsampling factor, windows truncated at the border and renormalised) this is
 * F.interpolate(mode="bicubic", align_corners=False, antialias=True) -- the patch / image resampling of compute_D2_loss and
 * get_patch_in_input for T_resolution_multiplier 2 / 4 or patch cut-outs that are not 32 px (sinskitG_model.py:1440-1476, 1531-1557,
 * model_utils.py:300-340); the adjoint is the same call on the transposed tables (vts/ops.py:bicubic_aa_tables builds both). */
int vts_resample_table(const float* in, int64_t NC, int IH, int IW, const int* ymin, const int* ysize, const float* wy, int KY,
                       const int* xmin, const int* xsize, const float* wx, int KX, float* out, int OH, int OW, int accumulate,
                       void* stream);

/* Anti-aliased resampling (networks.py:51-74 Downsample filt 3 / stride 2 / reflect; :87-107 Upsample filt 4 /
 * stride 2 / replicate), depthwise, with normalise-on-load of the input.  Down: [H,W] -> [(H-1)/2+1, (W-1)/2+1];
 * up: [H,W] -> [2H,2W].  The *_bwd entry points are the adjoints (gradient w.r.t. the activated input). */
int vts_blur_down(const vts_operand* in, int act, int N, int H, int W, float* out, void* stream);
int vts_blur_down_bwd(const float* dout, int N, int C, int H, int W, float* din, int accumulate, void* stream);
int vts_blur_up(const vts_operand* in, int act, int N, int H, int W, float* out, void* stream);
int vts_blur_up_bwd(const float* dout, int N, int C, int H, int W, float* din, int accumulate, void* stream);

/* K x K (K <= 8) weights <-> block (a, b) of their zero-extended 8 x 8 tap grid, as a 4 x 4 kernel:
 *   w4[r][i][j] = w[r][4a+i][4b+j] (0 outside K x K), r over rows = Cout*Cin.
 * The 3x3 / 7x7 convolutions of the ResNet generator run as 1 / 4 launches of vts_conv4x4 / vts_wgrad4x4
 * on these blocks (pad -> pad - 4a / 4b; outputs accumulate). */
int vts_tap_embed(const float* w, int64_t rows, int K, int a, int b, float* w4, void* stream);
int vts_tap_extract(const float* dw4, int64_t rows, int K, int a, int b, float* dw, int accumulate, void* stream);
/* the same with an explicit origin: (oy, ox) in [-3, 7] is the position of the block's tap (0, 0) in the K x K grid; a negative origin
 * places a small kernel inside the block, which (with pad = -origin) runs the stride-2 3x3 / 1x1 EqualConv2d of the StyleGAN2
 * ConvLayers (models/stylegan_networks.py:622-668) on the stride-2 4x4 kernels */
int vts_tap_embed_at(const float* w, int64_t rows, int K, int oy, int ox, float* w4, void* stream);
int vts_tap_extract_at(const float* dw4, int64_t rows, int K, int oy, int ox, float* dw, int accumulate, void* stream);

/* ---- GEMM-class 3x3 kernels for wide layers (pix2pixHD GlobalGenerator, models/networks.py:1952-1980: stride-2
 * 3x3 downsampling convs, ResnetBlocks :1267-1324 at up to 1024 channels, ConvTranspose2d(3, s2, p1, op1) upsampling).
 * All of them read PRE-PADDED identity inputs (vts_pad_affine) and tap-major packed weights:
 *
 *   vts_w3x3_pack:  wt[(a*9 + t)*B + b] = w[a*sa + b*sb + (flip ? 8-t : t)],  a < A = operator input channels,
 *                   b < B = operator output channels (B % 4 == 0), element strides sa / sb into the parameter tensor.
 *       nn.Conv2d weight [Co,Ci,3,3]:           forward A=Ci,B=Co,sa=9,sb=9Ci;   input adjoint (s1) A=Co,B=Ci,sa=9Ci,sb=9,flip;
 *                                               input adjoint of the stride-2 conv (a transposed conv): A=Co,B=Ci,sa=9Ci,sb=9
 *       nn.ConvTranspose2d weight [Ci,Co,3,3]:  forward A=Ci,B=Co,sa=9Co,sb=9;   input adjoint (a stride-2 conv) A=Co,B=Ci,sa=9,sb=9Co
 *
 *   vts_conv3x3_wide     out[n,co,y,x]   = bias + sum in[n,ci,y+ky,x+kx]   * wt[..]   in [N,Cin,H+2,W+2]   -> out [N,Cout,H,W]
 *   vts_conv3x3s2_wide   out[n,co,y,x]   = bias + sum in[n,ci,2y+ky,2x+kx] * wt[..]   in [N,Cin,2OH+2,2OW+2] (zero pad 1) -> [N,Cout,OH,OW]
 *   vts_tconv3x3s2_wide  out[n,co,y,x]   = bias + sum_{i: k=y+1-2i in 0..2} in[n,ci,i,j] * wt[..]   in [N,Cin,IH+1,IW+1] (zero row /
 *                        column appended) -> out [N,Cout,2IH,2IW]; one launch per output parity phase
 *   vts_wgrad3x3_wide    dw[co][ci][ky][kx] (+)= sum_{n,y,x} dout[n,co,y,x] * in[n,ci,stride*y+ky,stride*x+kx]
 *                        (dout [N,Cout,H,W], in [N,Cin,stride*H+2,stride*W+2]; for a ConvTranspose2d weight pass the layer input
 *                        as `dout` and the padded output gradient as `in`); deterministic slice reduction through `ws`.
 * Grids too small to fill the GPU split the channel loop (scratch: vts_conv3x3_wide_ws_floats, called with the grid of output
 * pixels of ONE launch: the output extent for the convolutions, the input extent for the transposed one).
 * Maps of <= 128 output pixels per image (pix2pixHD trained patch-wise, models/pix2pixHD_model.py:587-722 on the 32 x 32
 * patches of data/patchskit_dataset.py:277-333: the 1024-channel blocks see 2 x 2 maps) run on flattened-batch variants of the
 * same kernels (GEMM N / K dimension = the (image, y, x) index of whole images); results and interface are unchanged.
 * `ws` of vts_wgrad3x3_wide may be NULL when vts_wgrad3x3_wide_ws_floats returns 0. */
int vts_w3x3_pack(const float* w, int A, int B, int64_t sa, int64_t sb, int flip, float* wt, void* stream);
int vts_conv3x3_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int H, int W,
                     float* ws, int64_t ws_floats, void* stream);
/* The same convolution with the frozen VGG stacks' padded layout as its OUTPUT (perceptual terms, reference models/sinskitG_model.py:1711,
 * models/networks.py:2021-2067): out is [N][Cout][H + 2][W + 2], the next 3x3 convolution's pre-padded operand; the kernel stores the
 * interior and the zero one-pixel border:
 *   vts_conv3x3_wide_relu_pad   max(conv + bias, 0): no ReLU + padding pass between two convolutions
 *   vts_conv3x3_wide_mask_pad   (conv + add) where mask > 0, else 0; `mask` (the padded ReLU'd activation of the layer in front) and `add`
 *                               (optional: that layer's tap gradient) have out's layout: no ReLU-mask + padding pass between two input adjoints
 * VTS_ERR_UNSUPPORTED for shapes that do not take a tiled direct launch (small maps): the caller keeps the dense form there.
 * vts_zero_border: buf [NC][H + 2 pad][W + 2 pad], the border of `pad` pixels <- 0 (interior untouched). */
int vts_conv3x3_wide_relu_pad(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int H, int W, void* stream);
int vts_conv3x3_wide_mask_pad(const float* in, const float* wt, float* out, int N, int Cin, int Cout, int H, int W, const float* add,
                              const float* mask, void* stream);
int vts_zero_border(float* buf, int64_t NC, int H, int W, int pad, void* stream);
/* Winograd F(2x2, 3x3) form of the same stride-1 convolution (csrc/vts_conv3x3_wino.hip; round 4) for the frozen VGG stacks: 16 instead
 * of 36 multiplications per 2 x 2 outputs and channel pair, fp32 throughout (the result differs from the direct form by rounding only).
 *   vts_w3x3_wino_pack   U[(a * 16 + p) * B + b] = (G g G^T)[p] of the taps g[t] = w[a * sa + b * sb + (flip ? 8 - t : t)] -- arguments as
 *                        vts_w3x3_pack; a runs to A rounded up to 8 (zero rows); vts_w3x3_wino_floats(A, B) floats
 *   vts_conv3x3_wino     out <- conv of the pre-padded in [N][Cin][H + 2][W + 2]; out_pad 1: out is [N][Cout][H + 2][W + 2] and the kernel stores
 *                        the interior and the zero border (the padded layout of vts_conv3x3_wide_relu_pad / _mask_pad: ep_mode 1 / 2 with
 *                        ep_add / ep_mask as there; ep_mode 0: plain); VTS_ERR_UNSUPPORTED unless vts_conv3x3_wino_ok (Cout a multiple of
 *                        64, Cin >= 32, >= 256 workgroups of 16 x 16 pixels x 64 channels) */
int64_t vts_w3x3_wino_floats(int A, int B);
int vts_w3x3_wino_pack(const float* w, int A, int B, int64_t sa, int64_t sb, int flip, float* U, void* stream);
int vts_conv3x3_wino_ok(int N, int Cin, int Cout, int H, int W);
int vts_conv3x3_wino(const float* in, const float* U, const float* bias, float* out, int N, int Cin, int Cout, int H, int W, int out_pad,
                     int ep_mode, const float* ep_add, const float* ep_mask, void* stream);
int64_t vts_conv3x3_wide_ws_floats(int N, int Cin, int Cout, int H, int W);
int vts_conv3x3s2_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int OH, int OW,
                       float* ws, int64_t ws_floats, void* stream);
int vts_tconv3x3s2_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int IH, int IW,
                        float* ws, int64_t ws_floats, void* stream);
int64_t vts_wgrad3x3_wide_ws_floats(int N, int Cin, int Cout, int H, int W, int stride);
int vts_wgrad3x3_wide(const float* dout, const float* in, float* dw, int N, int Cin, int Cout, int H, int W, int stride,
                      int accumulate, float* ws, int64_t ws_floats, void* stream);

/* ---- 4 x 4 convolutions of wide layers (the ndf = 64 PatchGAN discriminators of pix2pixHD, full-size images or 32 x 32 patches:
 * models/networks.py MultiscaleDiscriminator / NLayerDiscriminator, kw = 4, padw = 2, called from models/pix2pixHD_model.py:587-722).
 * The GEMM-class kernels (tiled for full-size maps, flattened-batch for maps of <= 128 pixels) with 16-tap packed weights (vts_w4x4_pack: wt[(a*16 + t)*B + b] = w[a*sa + b*sb + (flip ? 15-t : t)];
 * nn.Conv2d weight [Co,Ci,4,4]: forward A=Ci,B=Co,sa=16,sb=16Ci; input adjoint of the stride-1 conv A=Co,B=Ci,sa=16Ci,sb=16,flip on the
 * gradient zero-padded by 3 - pad; input adjoint of the stride-2 conv: the same A/B/sa/sb without flip and transposed = 1).
 *   transposed = 0:  out[n,co,y,x] = bias + sum in[n,ci,stride*y+ky,stride*x+kx] * wt[..]   in [N,Cin,PH,PW] pre-padded (vts_pad_affine)
 *   transposed = 1:  (stride 2, pad 2) `in` = output gradient with one zero row / column appended, out = input gradient [N,Cout,OH,OW],
 *                    one launch per output parity phase.
 * vts_conv4x4_flat_ok tells whether a shape is a small-map case (<= 128 output pixels per image and launch); the tiled kernel
 * for larger maps needs Cout % 4 == 0.  Packed rows have a pitch of B rounded up to 4 floats (zero filled), so Cout = 1 prediction heads qualify as well
 * (vts_w3x3_pack: the same; its consumers require B % 4 == 0). */
int vts_w4x4_pack(const float* w, int A, int B, int64_t sa, int64_t sb, int flip, float* wt, void* stream);
int vts_conv4x4_flat_ok(int OH, int OW, int PH, int PW, int transposed);
int64_t vts_conv4x4_wide_ws_floats(int N, int Cin, int Cout, int OH, int OW, int PH, int PW, int transposed);
int vts_conv4x4_wide(const float* in, const float* wt, const float* bias, float* out, int N, int Cin, int Cout, int PH, int PW,
                     int OH, int OW, int stride, int transposed, float* ws, int64_t ws_floats, void* stream);
/* weight gradient of the same layers on full-size maps: dw[co][ci][ky][kx] (+)= sum dout[n,co,y,x] * in[n,ci,stride*y+ky,stride*x+kx],
 * dout [N,Cout,H,W], in [N,Cin,PH,PW] pre-padded; GEMM-class (K = pixels), deterministic slice reduction through `ws`
 * (vts_wgrad4x4_wide_ws_floats).  Small maps stay on vts_wgrad4x4. */
int64_t vts_wgrad4x4_wide_ws_floats(int N, int Cin, int Cout, int H, int W, int stride);
int vts_wgrad4x4_wide(const float* dout, const float* in, float* dw, int N, int Cin, int Cout, int H, int W, int PH, int PW,
                      int stride, int accumulate, float* ws, int64_t ws_floats, void* stream);

/* ---- StyleGAN2 building blocks (models/stylegan_networks.py) ----
 * vts_upfirdn2d      upfirdn2d_native :38-76 (Blur :140-156, Upsample :98-116, Downsample :119-137): insert up-1 zeros, pad
 *                    (negative pad crops) by (px0, px1, py0, py1), correlate with the FLIPPED `kernel` (HOST pointer, KH x KW <= 64
 *                    floats, copied into the launch), keep every down-th sample; in [NC, IH, IW] -> out [NC, OH, OW],
 *                    OH = vts_upfirdn2d_out_size(IH, KH, up, down, py0, py1).  vts_upfirdn2d_bwd is its adjoint (same arguments
 *                    as the forward; dout [NC, OH, OW] -> din [NC, IH, IW]), a gather: deterministic.
 * vts_bias_act       fused_leaky_relu :18-19 / FusedLeakyReLU :22-35 / ScaledLeakyReLU :236-245:
 *                    out = leaky_relu(x + bias[c], slope) * gain (+ res); bias / res may be NULL.  The residual input serves
 *                    ResBlock :686-693, (out * skip_gain + skip) / sqrt(skip_gain^2 + 1), with the constants folded into `gain`
 *                    and into the skip convolution's operand scale.  vts_bias_act_bwd: dx = g * gain * (x + bias > 0 ? 1 : slope)
 *                    (the bias gradient is vts_channel_sum of dx).
 * vts_modconv_demod  ModulatedConv2d :311-317: demod[n,co] = rsqrt(scale^2 * sum_{ci,k} (w[co,ci,k] * s[n,ci])^2 + eps); with it the
 *                    modulated convolution is conv(x * s[n,ci] * scale, w) * demod[n,co] on the shared weight (both factors are
 *                    operand affines of the conv kernels) instead of the reference's per-sample grouped convolution. */
int vts_upfirdn2d_out_size(int in, int k, int up, int down, int pad0, int pad1);
int vts_upfirdn2d(const float* in, int64_t NC, int IH, int IW, const float* kernel, int KH, int KW, int up, int down, int px0,
                  int px1, int py0, int py1, float* out, int accumulate, void* stream);
int vts_upfirdn2d_bwd(const float* dout, int64_t NC, int IH, int IW, const float* kernel, int KH, int KW, int up, int down, int px0,
                      int px1, int py0, int py1, float* din, int accumulate, void* stream);
int vts_bias_act(const float* x, const float* bias, const float* res, int N, int C, int64_t HW, float slope, float gain, float* out,
                 void* stream);
int vts_bias_act_bwd(const float* g, const float* x, const float* bias, int N, int C, int64_t HW, float slope, float gain, float* dx,
                     void* stream);
int vts_modconv_demod(const float* w, const float* s, int N, int Cout, int Cin, int KK, float scale, float eps, float* demod,
                      void* stream);
/* Style-free ModulatedConv2d weight of the StyleGAN2 generator's StyledConv layers (style = None -> s = 1, stylegan_networks.py:307-317,
 * 399-407): wout[co,ci,k] = v d[co], v = scale w, d[co] = rsqrt(sum_{ci,k} v^2 + eps); transpose != 0 stores [Ci,Co,KK] (the weight of
 * the stride-2 convolution whose input adjoint is the upsampling transposed convolution :320-330).  _bwd: dw (+)= scale (d g - d^3 v sum(g v))
 * for g = dL/dwout in the same layout. */
int vts_modconv_weight(const float* w, int Cout, int Cin, int KK, float scale, float eps, int transpose, float* wout, void* stream);
int vts_modconv_weight_bwd(const float* w, const float* g, int Cout, int Cin, int KK, float scale, float eps, int transpose, float* dw,
                           int accumulate, void* stream);

/* Adaptive instance normalisation of the style-code conditioning (thirdparty/AdaIN/function.py:4-23; CustomUnetGenerator.forward with
 * --style_code_mode adain, models/networks.py:1624-1630): per (n, c) group of HW positions
 *   out = (x - mean x) / std x * std s + mean s,  std = sqrt(unbiased variance + eps), eps = 1e-5 upstream.
 * x, s, out (and g, dx, ds of the backward) are [NC, HW] contiguous; the backward writes both operand gradients. */
int vts_adain(const float* x, const float* s, int NC, int HW, float eps, float* out, void* stream);
int vts_adain_bwd(const float* g, const float* x, const float* s, int NC, int HW, float eps, float* dx, float* ds, void* stream);

/* ---- Evaluation metrics that need no pretrained network (models/model_utils.py:431-561 compute_evaluation_metric) ----
 * vts_minmax:          out2 = {min x, max x}
 * vts_metric_psnr:     I_PSNR (:481-496): both images mapped with the REAL image's range {lo, hi} (range2, device memory) to
 *                      [0,1], the fake one clamped, PSNR with data_range 1 = 10 log10(1 / mse)
 * vts_metric_tactile:  T_AE (:531-536: mean angle in degrees between normalize(gx, gy, 1) of the real and of the fake patches,
 *                      normal_losses.py:10-33 mode 'evaluate') and T_MSE (:557); fake patches clamped to [0,1] first (:521).
 * real_T / fake_T are [P, 2, HW]; `ws` holds vts_metric_ws_floats() floats; fixed-order (deterministic) reductions. */
int64_t vts_metric_ws_floats(void);
int vts_minmax(const float* x, int64_t n, float* out2, float* ws, void* stream);
int vts_metric_psnr(const float* real, const float* fake, int64_t n, const float* range2, float* out, float* ws, void* stream);
int vts_metric_tactile(const float* real_T, const float* fake_T, int64_t P, int HW, float* out_ae, float* out_mse, float* ws,
                       void* stream);
/* I_SSIM (:498-499: torchmetrics.functional.structural_similarity_index_measure(real_I, fake_I, data_range=1), an un-pinned pip
 * dependency absent from this image: its published algorithm is restated -- 11 x 11 Gaussian window (sigma 1.5), k1 0.01, k2 0.03,
 * reflect padding by 5 cropped again = mean of the SSIM map over the (H-10) x (W-10) positions whose window lies inside the image,
 * variances clamped at 0) on the same {lo, hi}-normalised images as vts_metric_psnr; real / fake are [NC, H, W]. */
int vts_metric_ssim(const float* real, const float* fake, int NC, int H, int W, const float* range2, float* out, float* ws, void* stream);

/* Fréchet distance between two feature sets -- the arithmetic of SIFID behind the Inception features (models/sifid.py:102-176:
 * np.mean / np.cov(rowvar=False) per set, then |mu1-mu2|^2 + tr(S1) + tr(S2) - 2 tr(sqrtm(S1 S2))).  feat1 / feat2 are channel-major
 * [D, P] float32 (one image's NCHW feature map: D <= 64 channels, P1 / P2 positions); moments and the matrix square root (coupled
 * Newton-Schulz) are computed in float64; out[0] is the distance.  ws: vts_frechet_ws_floats() floats, 8-byte aligned.  (The
 * reference's fallback for a singular product -- adding 1e-6 to the diagonals -- is not reproduced.) */
int64_t vts_frechet_ws_floats(void);
int vts_frechet_distance(const float* feat1, const float* feat2, int D, int64_t P1, int64_t P2, float* out, float* ws, void* stream);

/* Network input of the SIFID metrics (models/model_utils.py:481-488 I_SIFID, :541-555 T_SIFID; InceptionV3.forward's 2x - 1,
 * models/inception.py:135): out [N,3,OH,OW] from channels c0.. of src [N,*,IH,IW] (C = 3, or C = 1 tiled three times), nearest
 * resize as F.interpolate's default.  lohi != NULL (device {min, max} of the real image): v -> 2 * clamp?((v - lo) / (hi - lo)) - 1
 * (clamp01: the fake image's clamp); lohi == NULL: v -> clamp01 ? clamp(v, 0, 1) : v (tactile patches; their normalisation and the
 * network's 2x - 1 cancel). */
int vts_sifid_input(const float* src, int64_t nstride, int N, int c0, int C, int IH, int IW, const float* lohi, int clamp01,
                    float* out, int OH, int OW, void* stream);

/* AvgPool2d(3, stride 2, padding 1, count_include_pad=False) forward / backward
 * (models/networks.py:1670).  Backward accumulates into dx when accumulate != 0. */
int vts_avgpool3s2(const float* x, int64_t x_nstride, int N, int C, int H, int W, float* y, void* stream);
int vts_avgpool3s2_bwd(const float* dy, int N, int C, int H, int W, float* dx, int64_t dx_nstride, int accumulate,
                       void* stream);

/* ---- perceptual terms (round 3): the glue around the frozen VGG feature stacks, whose 3x3 convolutions run on vts_conv3x3_wide.
 * LPIPS-VGG16 = lpips.LPIPS(net="vgg") as called at models/sinskitG_model.py:495, 1639-1646, 1711 and models/model_utils.py:477,
 * 523-527 (third-party package: algorithm restated in oracle/perceptual.py); VGG19 features = Vgg19 / VGGLoss, models/networks.py:
 * 2021-2067.  Activations are RAW convolution outputs z; every consumer applies relu on load. ---- */
/* out [NC][H/2 + 2 pad][W/2 + 2 pad] = zero-padded MaxPool2d(2, 2)(relu(z)), z [NC][H][W]: replaces ReLU + MaxPool2d + the next
 * convolution's padding (torchvision vgg features) */
/* (zpad, here and in the three entries below: z is itself a padded tensor [NC][H + 2 zpad][W + 2 zpad] read at its interior -- the ReLU'd,
 * pre-padded output of vts_conv3x3_wide_relu_pad; 0: dense [NC][H][W]) */
int vts_maxpool2_relu_pad(const float* z, int NC, int H, int W, int pad, float* out, int zpad, void* stream);
/* AlexNet variant of LPIPS (the reference's test-phase eval_LPIPS = lpips.LPIPS(net="alex"), models/sinskitG_model.py:501; pip package
 * `lpips`, lpips/pretrained_networks.py:alexnet over torchvision alexnet.features):
 *   vts_maxpool3s2_relu_pad  out[nc][pad + y][pad + x] = max over the 3 x 3 window at (2y, 2x) of relu(z), OH = (H - 3) / 2 + 1, zero border of
 *                            `pad` pixels (MaxPool2d(3, 2) behind a ReLU + the next convolution's zero padding in one pass)
 *   vts_s2d4_pad             space-to-depth by 4 of x zero-padded by `pad`: out[n][(c * 4 + i) * 4 + j][Y][X] = xp[n][c][4Y + i][4X + j], Y < OH,
 *                            X < OW -- the 11 x 11 stride-4 stem becomes a valid 3 x 3 convolution over 16 C channels (vts_conv3x3_wide) */
int vts_maxpool3s2_relu_pad(const float* z, int NC, int H, int W, int pad, float* out, void* stream);
int vts_s2d4_pad(const float* x, int N, int C, int H, int W, int pad, int OH, int OW, float* out, void* stream);
/* gz [NC][H + 2 pad][W + 2 pad] = zero-padded (adjoint of relu -> MaxPool2d(2, 2) applied to g [NC][H/2][W/2] (first-maximum tie rule of
 * PyTorch) + g2 where z > 0); g2 (optional, this layer's tap gradient) has z's layout */
int vts_maxpool2_relu_bwd(const float* g, const float* z, int NC, int H, int W, float* gz, int zpad, const float* g2, int pad, void* stream);
/* out [NC][H + 2 pad][W + 2 pad] = zero-padded (g + g2) * (z > 0); g (dense [NC][H][W]) or g2 (z's layout) may be NULL: ReLU backward + the
 * padding of the adjoint conv */
int vts_relu_mask_pad(const float* g, const float* g2, const float* z, int NC, int H, int W, int pad, float* out, int zpad, void* stream);
/* one LPIPS tap: loss_slot += coeff * sum_n mean_pixels sum_c w[c] (f0n - f1n)^2 with f = relu(z), fn = f / (|f|_channels + 1e-10);
 * dz0 (optional, z0's layout; with zpad its border is the caller's) = grad_coeff * d(that sum)/d z0, i.e. the gradient w.r.t. relu(z0) where
 * z0 > 0 and 0 elsewhere  (lpips.LPIPS.forward: normalize_tensor, lin layers, spatial_average) */
int vts_lpips_layer(const float* z0, const float* z1, int N, int C, int HW, const float* w, float coeff, int64_t* loss_slot, float* dz0,
                    float grad_coeff, int W, int zpad, void* stream);   /* W: map width (needed when zpad > 0) */
/* loss_slot += coeff * sum |relu(za) - relu(zb)|;  grad (optional) = coeff * sign(.)  (VGGLoss: nn.L1Loss on ReLU features) */
int vts_l1_relu(const float* za, const float* zb, int64_t n, float coeff, int64_t* loss_slot, float* grad, void* stream);
/* LPIPS ScalingLayer: y [N][3][HW] = (x - shift_c) / scale_c; Cx = 1 broadcasts the single channel (tactile gx / gy); shift3 / scale3
 * are HOST triples.  vts_lpips_input_bwd: dx (+)= adjoint applied to g [N][3][HW] */
int vts_lpips_input(const float* x, int64_t x_nstride, int N, int Cx, int HW, const float* shift3, const float* scale3, float* y, void* stream);
int vts_lpips_input_bwd(const float* g, int N, int Cx, int HW, const float* scale3, float* dx, int64_t dx_nstride, int accumulate, void* stream);

/* Start of a training step: loss slots <- 0, every optimiser's device step counter += 1 (vts_adam_flat_dev reads them): one launch */
int vts_step_begin(int64_t* loss_slots, int nslots, int* step_counters, int ncounters, void* stream);

/* Loss slots: 64-bit fixed point, VTS_LOSS_SCALE units per 1.0 (value = slot / VTS_LOSS_SCALE).  Integer atomic adds commute, so a
 * logged loss is bitwise reproducible whatever order workgroups and concurrent streams add to a slot in. */
#define VTS_LOSS_SCALE 1099511627776.0 /* 2^40 */

/* GANLoss on one scale (models/networks.py:497-521), forward value and gradient in one pass.
 *   mode: 0 nonsaturating, 1 lsgan, 2 vanilla(BCE logits), 3 wgan, 4 hinge, 5 vanilla on sigmoid(pred) (the Sigmoid the reference's
 *         MultiscaleDiscriminator appends for gan_mode 'vanilla', networks.py:1659, 1731-1732, still followed by BCEWithLogits :507-509)
 *   loss_out[0] += coeff * mean_over_batch( per-sample loss )   (lsgan/vanilla/wgan: global mean)
 *   dpred (if non-NULL) = grad_coeff * d(mean_over_batch(per-sample loss)) / dpred   */
int vts_ganloss(const float* pred, int N, int M, int mode, int target_is_real, float target_label, float coeff,
                float grad_coeff, int64_t* loss_out, float* dpred, void* stream);

/* loss_out[0] += coeff * sum|a-b| ;  grad (+)= coeff * sign(a-b)   (nn.L1Loss pieces,
 * sinskitG_model.py:1702, 1812-1814; the caller folds 1/numel into coeff). */
int vts_l1(const float* a, const float* b, int64_t n, float coeff, int64_t* loss_out, float* grad, int accumulate,
           void* stream);

/* Patch gather with clamp-to-border (models/model_utils.py:252-333), for P patches of
 * size x size taken from image index img[p] at offsets (offx[p], offy[p]):
 *   out[p, out_c0 + c, y, x] = src[img[p], c, clamp(offy+y), clamp(offx+x)]
 * and its deterministic backward (each source pixel sums the patches covering it, in patch order). */
int vts_patch_gather(const float* src, int64_t src_nstride, int C, int H, int W, const int* img, const int* offx,
                     const int* offy, int P, int size, float* out, int out_C, int out_c0, void* stream);
int vts_patch_scatter_bwd(const float* dpatch, int dp_C, int dp_c0, int C, const int* img, const int* offx,
                          const int* offy, int P, int P_per_img, int size, float* dsrc, int64_t dsrc_nstride, int N, int H,
                          int W, int accumulate, void* stream);

/* Batched patch gather / copy / fill (round 3): all channel runs of the D2 patch stacks -- [fake_T | S | aug_fake_I | mask] and its real /
 * "more fake" counterparts, sinskitG_model.py:1268-1291, 1477-1484, 1506-1560 -- in ONE launch (the step used nine gathers and five
 * copies).  Job: C channels of P patches into channel slot dst_c0 of dst [P, dst_C, size, size]; img != NULL: gather with
 * clamp-to-border from src [*, C.., H, W] (batch stride src_nstride) as vts_patch_gather; img == NULL: copy from the patch tensor src
 * [P, C, size, size] (patch stride src_nstride); src == NULL: fill with `fill`.  At most 16 jobs per call. */
typedef struct vts_patch_job {
  const float* src;
  int64_t src_nstride;
  int C, H, W;
  const int *img, *offx, *offy;
  int P;
  float* dst;
  int dst_C, dst_c0;
  float fill;
} vts_patch_job;
int vts_patch_jobs(const vts_patch_job* jobs /* host array */, int njobs, int size, void* stream);

/* Generator output post-processing (sinskitG_model.py:1309-1340), one pass over g_out [N,5,H,W]:
 *   fake_I = g_out[:, :3]*M ; fake_T = g_out[:, 3:]*M ; fake_N = normalize(gx, gy, scale_nz)
 *   aug_fake_I = DiffAugment_bs(fake_I; rb, rs) * M            (thirdparty/DiffAugment.py:25-33)
 * Any output pointer may be NULL.  fake_T / aug_fake_I take a batch stride (floats; 0 = contiguous) so they
 * can be written straight into channel slices of the 7-channel D2 input stack. */
int vts_g_post(const float* g_out, const float* M, int N, int H, int W, float scale_nz, const float* rb, const float* rs,
               float* fake_I, float* fake_T, int64_t fake_T_nstride, float* fake_N, float* aug_fake_I,
               int64_t aug_nstride, void* stream);
/* the same pass also writing the sketch and the mask into their channels of the 7-channel full-resolution D2 stack
 * (stack_S / stack_M: channel-slice pointers with batch stride stack_nstride; sinskitG_model.py:1495-1501 builds that stack with torch.cat) */
int vts_g_post_stack(const float* g_out, const float* M, int N, int H, int W, float scale_nz, const float* rb, const float* rs,
                     float* fake_I, float* fake_T, int64_t fake_T_nstride, float* fake_N, float* aug_fake_I, int64_t aug_nstride,
                     const float* S, float* stack_S, float* stack_M, int64_t stack_nstride, void* stream);

/* One DiffAugment operation on x [N, C, H, W] (sample stride x_ns floats) -> out (sample stride out_ns; out != x), times the mask M
 * [N, 1, H, W] when M is given: any policy string over the reference's letters runs as a chain of these
 * (thirdparty/DiffAugment.py:25-80, AUGMENT_FNS :89-96; the default 'bs' has the fused entry below).  The random numbers are the
 * caller's (DiffAugment draws them from torch's generator), one set per sample:
 *   op 'b'  out = x + (pf - 0.5)                                  pf = the uniform draw
 *      's'  out = (x - mean_c x) * 2 pf + mean_c x
 *      'c'  out = (x - mean_chw x) * (pf + 0.5) + mean_chw x       ws: vts_diffaug_op_ws_floats(N) floats; x contiguous per sample
 *      't'  out[i, j] = x[i + pi0, j + pi1], zero outside          pi0 / pi1 = row / column translation (randint(-s, s + 1), s = int(extent / 8 + 0.5))
 *      'o'  rows clamp(pi0 - ch / 2 + [0, ch)), columns clamp(pi1 - cw / 2 + [0, cw)) zeroed, ch = int(H / 2 + 0.5), cw likewise
 *      'n'  out = x + pf * noise                                   pf = sigma (rand * 0.1, zeroed where the gate draw >= 0.5), noise [N, C, H, W] */
int vts_diffaug_op_ws_floats(int N);
int vts_diffaug_op(const float* x, int64_t x_ns, float* out, int64_t out_ns, int N, int C, int H, int W, int op, const float* pf,
                   const int* pi0, const int* pi1, const float* noise, const float* M, float* ws, void* stream);

/* aug = DiffAugment_bs(x; rb, rs) * M for a 3-channel image. */
int vts_diffaug_bs_mask(const float* x, const float* M, int N, int H, int W, const float* rb, const float* rs, float* aug,
                        void* stream);
/* d g_raw = cat(d fake_I, d fake_T) * M * (1 - g_out^2)   (mask multiply + Tanh backward). */
int vts_g_out_grad(const float* d_fake_I, const float* d_fake_T, const float* M, const float* g_out, int N, int H, int W,
                   float* d_raw, void* stream);
/* the same with the image gradient's next pyramid level d_fake_I_coarse [N, 3, ceil(H / 2), ceil(W / 2)] (or NULL): its adjoint of
 * AvgPool2d(3, 2, padding 1, count_include_pad False) -- the pyramid of MultiscaleDiscriminator, models/networks.py:1682-1691 -- is added to
 * d_fake_I on the fly, bit-identical to vts_avgpool3s2_bwd(accumulate) followed by vts_g_out_grad. */
int vts_g_out_grad_pool(const float* d_fake_I, const float* d_fake_I_coarse, const float* d_fake_T, const float* M, const float* g_out,
                        int N, int H, int W, float* d_raw, void* stream);
/* ImagePool.query (reference util/image_pool.py:29-61; pix2pixHD's fake_pool, pix2pixHD_model.py:334, 582) on the device.  The host
 * makes the reference's draws (Python's `random`) and hands over, per image n of the batch in order: ret_slot[n] = the pool slot whose
 * CURRENT content is returned in place of image n (-1: image n itself) and put_slot[n] = the slot image n is stored into (-1: none).
 * out[n] = ret_slot[n] < 0 ? images[n] : store[ret_slot[n]], then store[put_slot[n]] = images[n], for n = 0 .. N-1 IN ORDER (a later
 * image of the batch may draw a slot an earlier one has just written); every thread walks the N images of its own elements, so the
 * order holds without a barrier.  images, out [N, elems]; store [pool_size, elems]; ret_slot, put_slot: device int32 [N]. */
int vts_pool_query(const float* images, float* store, const int* ret_slot, const int* put_slot, int N, int64_t elems, float* out,
                   void* stream);
/* y = x * M (M broadcast over channels). */
int vts_mask_mul(const float* x, const float* M, int N, int C, int HW, float* y, void* stream);
/* Sinusoidal positional grid, SPE(dim,0) (thirdparty/mmgeneration/positional_encoding.py:54-160): out [N,2*dim,H,W]. */
int vts_spe_grid(float* out, int64_t out_nstride, int N, int H, int W, int dim, void* stream);
/* Dilated-mask candidate map of the "more fake T" sampler (models/model_utils.py:212-216):
 * cand[n, y, x] = any(M[n, y-1 .. y+15, x-1 .. x+15]) on the (H-14)x(W-14) grid; row_count[n, y] = number of
 * candidates in row y.  vts_mask_select resolves row-major ranks into (offx, offy). */
int vts_mask_candidates(const float* M, int N, int H, int W, uint8_t* cand, int* row_count, void* stream);
int vts_mask_select(const uint8_t* cand, const int* row_prefix, int N, int H, int W, const int64_t* ranks, int K, int* offx,
                    int* offy, void* stream);
/* ranks[n, 0..K) <- K distinct uniform ranks in [0, row_prefix[n][H-14]) -- random.sample(range(candidates), K) of the same sampler
 * (models/model_utils.py:217), drawn on the device from (seed, image, draw) so that the host never waits for the candidate count
 * (Floyd's algorithm: every K-subset equally likely; K <= 1024). */
int vts_mask_sample_ranks(const int* row_prefix, int N, int H, int K, uint64_t seed, int64_t* ranks, void* stream);

/* Fused Adam over a flat fp32 buffer (torch.optim.Adam defaults; sinskitG_model.py:589-599).
 * step_count: 1-based step index; grad_scale multiplies the gradient first (1/world for DDP mean). */
int vts_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  int step_count, float grad_scale, void* stream);

/* Same update with the 1-based step counter and the learning rate read from device memory
 * (*step_dev, *lr_dev), so a captured HIP graph of the step stays valid across iterations. */
int vts_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                      float eps, const int* step_dev, float grad_scale, void* stream);

/* PatchNCE loss forward+backward (models/patchnce.py:13-55): B groups of P patches, dim D.
 * loss[b*P+i] = CE([q_i.k_i, q_i.k_j (j != i; diagonal -> -10)] / T, 0); dq = d sum(loss*gscale) / dq.
 * P, D <= 256 (the reference's 256 patches x 256 dims): one workgroup per (image, 64 queries), q k^T and the gradient W k on
 * v_mfma_f32_16x16x4_f32, logits in LDS only, wave-shuffle softmax; larger P (all negatives of a minibatch) take a
 * one-query-per-workgroup kernel. */
int vts_patchnce(const float* q, const float* k, int B, int P, int D, float T, float gscale, float* loss, float* dq,
                 void* stream);
/* Row L2 normalisation x / (||x|| + 1e-7) (models/networks.py:585-594). */
int vts_l2norm_rows(const float* x, int rows, int D, float* y, void* stream);

/* PatchSampleF (models/networks.py:667-719):
 *   vts_patch_sample: out[(b*P + p)*C + c] = feat[b, c, ids[p]]   (feat.permute(0,2,3,1).flatten(1,2)[:, ids, :].flatten(0,1); the same
 *                     ids for every image of the batch, as in the reference), feat [B, C, HW], ids int64 [P];
 *   vts_linear_rows:  y[R x O] = act(x[R x I] W[O x I]^T + bias)  -- the nn.Linear layers of its optional 2-layer MLP (relu: 0 / 1),
 *                     on the MFMA tile routine of the PatchNCE kernel. */
int vts_patch_sample(const float* feat, const int64_t* ids, int B, int C, int HW, int P, float* out, void* stream);
int vts_linear_rows(const float* x, const float* w, const float* bias, int R, int I, int O, int relu, float* y, void* stream);

/* dst[i] = src[i] for i < nwords (4-byte words) as a kernel on `stream`; src may be pinned host memory. */
int vts_copy_words(const void* src, void* dst, int64_t nwords, void* stream);

/* set_input (models/sinskitG_model.py:702-793) with 8-bit sources: out[i] = src[i] / 255 (normalize 0: ToTensor) or (src[i] / 255 - 0.5) / 0.5
 * (normalize 1: + Normalize(0.5, 0.5)), evaluated in fp32 in the transform's own order -- bit-identical to the float tensor the reference's
 * dataset caches from the same PNG pixels (data/singleskit_dataset.py:317-329), so a batch may travel as uint8 (a quarter of the bytes). */
int vts_u8_expand(const uint8_t* src, int64_t n, int normalize, float* out, void* stream);

/* set_input's image part (reference models/sinskitG_model.py:702-760: real_S = S * M, real_I = I * M on the dataset's tensors) in one pass
 * over 8-bit sources: S [N,1,HW], I [N,3,HW] (or NULL), M [N,1,HW] (or NULL: no mask) bytes ->
 *   M_out [N,1,HW] = M / 255 (or NULL), S_out = Normalize(ToTensor(S)) * M, S_out2 = the same again (or NULL; the real rows of the
 *   discriminator's pair buffer), I_out [N,3,HW] likewise.  Bit-identical to vts_u8_expand followed by vts_mask_mul. */
int vts_input_images_u8(const uint8_t* S, const uint8_t* I, const uint8_t* M, int N, int64_t HW, float* M_out, float* S_out, float* S_out2,
                        float* I_out, void* stream);

/* ---- network-level entry (csrc/vts_unet.cpp; SURVEY.md 8b: `vts_unet_fwd`) ------------------------------------------------------------
 * The inference forward of the reference's generator as ONE call, for hosts that are not Python: CustomUnetGenerator.forward
 * (models/networks.py:1430-1645) over Down / Up (thirdparty/unet/unet_parts_custom.py:9-79) -- num_downs x [LeakyReLU(0.2) -> Conv2d(4, 2, 1)
 * -> InstanceNorm2d] (down0: convolution only; the innermost block: no norm), num_downs x [ReLU -> ConvTranspose2d(4, 2, 1) on cat(x, skip)
 * -> InstanceNorm2d] with the layers num_layer_separate-1 .. 0 duplicated for the tactile branch (`_T`), Tanh on both outermost blocks.
 * This is what test.py's loop spends its time in (test.py:62-74 -> SinSKITGModel.forward, models/sinskitG_model.py:1309-1319); the
 * Python product runs the same operators from vts/engine.py:unet_forward (bit-identical output: tests/test_network_abi_gpu.py).
 *   in0 / in1      the network input as a channel concatenation of two sources (sketch ++ positional grid, sinskitG_model.py:1309-1313);
 *                  in1.C = 0: one source.  Plain tensors: scale = shift = NULL
 *   channels[i]    output channels of down_i; down_w[i] [channels[i]][Cin_i][4][4], down_b[i] [channels[i]] (NULL: no bias)
 *   up_w[i]        nn.ConvTranspose2d layout [Cin_i][up_cout[i]][4][4] of up_i (Cin_i = channels of cat(x, skip_i), + style.C at the
 *                  innermost block), up_b[i] [up_cout[i]]; upT_* the same for up_i_T, i < num_layer_separate
 *   style          optional: the tiled style code [N][style.C][H >> num_downs][W >> num_downs] concatenated to the innermost block's
 *                  input (skitG, style_code_mode concat + mapping tile: models/networks.py:1600-1630); style.C = 0: none
 *   out            [N][up_cout[0] + upT_cout[0]][H][W]: visual channels first, then the tactile ones (the reference's torch.cat)
 * H and W must be divisible by 2^num_downs.  No allocation, no host synchronisation: every launch goes to `stream` (and side_stream),
 * scratch is vts_unet_forward_ws_floats(d) floats (-1 on a bad descriptor). */
#define VTS_UNET_MAX_DOWNS 10
typedef struct vts_unet_desc {
  int N, H, W;
  int num_downs, num_layer_separate;
  vts_operand in0, in1;
  int channels[VTS_UNET_MAX_DOWNS];
  const float* down_w[VTS_UNET_MAX_DOWNS];
  const float* down_b[VTS_UNET_MAX_DOWNS];
  const float* up_w[VTS_UNET_MAX_DOWNS];
  const float* up_b[VTS_UNET_MAX_DOWNS];
  int up_cout[VTS_UNET_MAX_DOWNS];
  const float* upT_w[VTS_UNET_MAX_DOWNS];
  const float* upT_b[VTS_UNET_MAX_DOWNS];
  int upT_cout[VTS_UNET_MAX_DOWNS];
  vts_operand style;
  float* out;
  void* side_stream; /* optional second hipStream_t: the tactile branch (up_i_T, i < num_layer_separate) runs on it beside the visual
                        branch, forked from and joined back into `stream` by events (capturable); NULL: everything on `stream` */
} vts_unet_desc;
int64_t vts_unet_forward_ws_floats(const vts_unet_desc* d);
int vts_unet_forward(const vts_unet_desc* d, float* ws, int64_t ws_floats, void* stream);

/* ---- network-level entries: the discriminators' forward (round 6; SURVEY.md 8(b): `vts_msd_fwd`; csrc/vts_msd.cpp) ---------------------
 * vts_patchgan_forward = NLayerDiscriminator.forward (models/networks.py:1696-1750), vts_msd_forward = MultiscaleDiscriminator.forward
 * (models/networks.py:1649-1691: num_D PatchGANs over an AvgPool2d(3, 2, 1, count_include_pad False) pyramid, full resolution first), in
 * TRAINING mode -- BatchNorm2d normalises with the batch statistics and advances its running statistics (momentum, unbiased variance,
 * num_batches_tracked += 1), as every discriminator call of a reference training step does (models/sinskitG_model.py:1361, 1374, 1490,
 * 1567, 1584, 1781) -- forward only: nothing is kept for a backward.  Weights in the reference's state-dict layout
 * (layer<k>.<i>.weight [Cout, Cin, 4, 4]).  Convolution j: Conv2d(4, stride[j], padding 2); LeakyReLU(0.2) in front of every convolution
 * but the first; BatchNorm2d behind convolution j where gamma[j] / beta[j] are given (never the first or the last).
 *   running_mean / running_var / num_batches_tracked [j]   NULL: this call does not advance the running statistics
 *   stat_mean_out / stat_uvar_out [j]                      optional [cout[j]]: record the batch mean / unbiased variance (a pass whose
 *                                                          running-statistics update is spliced into another launch: vts_norm_desc.ext_*)
 *   run_head = 0                                           stop in front of the last convolution (a pass that exists for the statistics)
 * No allocation, no host synchronisation (capturable); the workspace holds the raw layer outputs. */
#define VTS_PATCHGAN_MAX_CONVS 8
typedef struct vts_patchgan_desc {
  vts_operand in0, in1; /* channel-concatenated input (in1.C = 0: one source), e.g. torch.cat((real_S, image), 1) */
  int N, H, W;
  int n_convs; /* n_layers + 2 */
  int cout[VTS_PATCHGAN_MAX_CONVS];
  int stride[VTS_PATCHGAN_MAX_CONVS];
  const float* w[VTS_PATCHGAN_MAX_CONVS];
  const float* b[VTS_PATCHGAN_MAX_CONVS];
  const float* gamma[VTS_PATCHGAN_MAX_CONVS];
  const float* beta[VTS_PATCHGAN_MAX_CONVS];
  float* running_mean[VTS_PATCHGAN_MAX_CONVS];
  float* running_var[VTS_PATCHGAN_MAX_CONVS];
  int64_t* num_batches_tracked[VTS_PATCHGAN_MAX_CONVS];
  float* stat_mean_out[VTS_PATCHGAN_MAX_CONVS];
  float* stat_uvar_out[VTS_PATCHGAN_MAX_CONVS];
  float eps, momentum; /* nn.BatchNorm2d defaults: 1e-5, 0.1 */
  int run_head;
  float* pred; /* [N, 1, h, w] of the last convolution (run_head) */
} vts_patchgan_desc;
int64_t vts_patchgan_forward_ws_floats(const vts_patchgan_desc* d);
int vts_patchgan_forward(const vts_patchgan_desc* d, float* ws, int64_t ws_floats, void* stream);

#define VTS_MSD_MAX_SCALES 4
typedef struct vts_msd_desc {
  int num_D;
  /* scale[0] = the full-resolution PatchGAN (the reference's layer<num_D - 1>) and carries the input (in0, in1: plain tensors, N, H, W);
   * the inputs / sizes of the other scales are filled in by the call (their pooled levels live in the workspace). */
  vts_patchgan_desc scale[VTS_MSD_MAX_SCALES];
} vts_msd_desc;
int64_t vts_msd_forward_ws_floats(const vts_msd_desc* d);
int vts_msd_forward(const vts_msd_desc* d, float* ws, int64_t ws_floats, void* stream);

/* ---- optional collective of the data-parallel path (csrc/vts_comm.cpp; off by default, vts/ddp.py: VTS_DDP_DIRECT=1) ----------------
 * Sum-all-reduce of one flat fp32 gradient bucket as reduce-scatter + all-gather on the library's OWN RCCL communicator and side stream
 * (each rank reduces 1 / world of the bucket, all xGMI links carry a slice).  Replaces nn.DataParallel's gradient
 * reduction of the reference (models/base_model.py:104-108).  RCCL is dlopen'ed: single-GPU users never load it.
 *   vts_comm_unique_id        128-byte ncclUniqueId (rank 0 creates it; the caller distributes it, e.g. with torch.distributed.broadcast)
 *   vts_comm_init             ncclCommInitRank on the CURRENT device; creates the side stream
 *   vts_allreduce_flat_async  in place, ordered behind everything enqueued on `producer_stream` so far
 *   vts_allreduce_flat_wait   `consumer_stream` waits for the last async call of this communicator
 *   vts_comm_destroy
 *   vts_allreduce_slice_plan  host arithmetic only: the slice [offset, offset + chunk) rank `rank` reduces and the remainder
 *                             [tail_offset, tail_offset + tail) all ranks all-reduce; what vts_allreduce_flat_async itself uses */
int vts_allreduce_slice_plan(int64_t n, int world, int rank, int64_t* offset, int64_t* chunk, int64_t* tail_offset, int64_t* tail);
int vts_comm_unique_id(void* id128);
int vts_comm_init(const void* id128, int rank, int world, void** comm);
int vts_allreduce_flat_async(void* comm, float* buf, int64_t n, void* producer_stream);
int vts_allreduce_flat_wait(void* comm, void* consumer_stream);
int vts_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
