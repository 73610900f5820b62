"""ctypes binding of libvts_hip.so (include/vts.h) -- the only way the product path computes.

There is deliberately NO fallback: if the shared library is missing or a call fails, this
module raises.  Tensors are passed as raw device pointers (`tensor.data_ptr()`), shapes as
ints, and every launch goes to torch's current HIP stream, so torch provides memory and
stream plumbing only.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VTS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libvts_hip.so")   # (VTS_LIB_PATH: A/B of builds, tools/)
if os.environ.get("VTS_LIB_PATH"):
    import sys
    print("NOTE: VTS_LIB_PATH overrides the in-tree library: loading %s" % LIB_PATH, file=sys.stderr, flush=True)

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3
ERR_UNSUPPORTED = -2      # VTS_ERR_UNSUPPORTED (include/vts.h): the shape does not take this entry's kernel; the caller uses the general form
GAN_MODES = {"nonsaturating": 0, "lsgan": 1, "vanilla": 2, "wgan": 3, "wgangp": 3, "hinge": 4, "vanilla_sigmoid": 5}

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)


class Operand(C.Structure):
    _fields_ = [("data", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("C", C.c_int), ("nstride", C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("in0", Operand), ("in1", Operand),
        ("N", C.c_int), ("IH", C.c_int), ("IW", C.c_int), ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("transposed", C.c_int),
        ("w", C.c_void_p), ("ws_co", C.c_int), ("ws_ci", C.c_int),
        ("bias", C.c_void_p), ("out", C.c_void_p), ("out_nstride", C.c_int64),
        ("act_in", C.c_int), ("act_out", C.c_int),
        ("dmask", Operand), ("dmask_act", C.c_int), ("accumulate", C.c_int),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64), ("pad_dx", C.c_int),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("lo0", Operand), ("lo1", Operand), ("hi0", Operand), ("hi1", Operand),
        ("act_lo", C.c_int), ("act_hi", C.c_int),
        ("N", C.c_int), ("LH", C.c_int), ("LW", C.c_int), ("HH", C.c_int), ("HW", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("dw", C.c_void_p), ("accumulate", C.c_int), ("pad_dx", C.c_int), ("defer", C.c_int),
    ]


class ReduceJob(C.Structure):
    _fields_ = [("dw", C.c_void_p), ("nel", C.c_int64), ("accumulate", C.c_int), ("nseg", C.c_int), ("part", C.c_void_p * 4), ("pw", C.c_int * 4)]


class NormDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("nstride", C.c_int64), ("N", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("mode", C.c_int),
        ("eps", C.c_float), ("momentum", C.c_float), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean_out", C.c_void_p), ("rstd_out", C.c_void_p), ("counters", C.c_void_p),
        ("ngroups", C.c_int), ("gstart", C.c_int * 9),
        ("stat_mean_out", C.c_void_p), ("stat_uvar_out", C.c_void_p), ("ext_mean", C.c_void_p), ("ext_uvar", C.c_void_p), ("ext_after", C.c_int),
    ]


UNET_MAX_DOWNS = 10


class UnetDesc(C.Structure):
    """vts_unet_desc (include/vts.h): the generator's inference forward as one C call"""
    _fields_ = [
        ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("num_downs", C.c_int), ("num_layer_separate", C.c_int),
        ("in0", Operand), ("in1", Operand), ("channels", C.c_int * UNET_MAX_DOWNS),
        ("down_w", C.c_void_p * UNET_MAX_DOWNS), ("down_b", C.c_void_p * UNET_MAX_DOWNS),
        ("up_w", C.c_void_p * UNET_MAX_DOWNS), ("up_b", C.c_void_p * UNET_MAX_DOWNS), ("up_cout", C.c_int * UNET_MAX_DOWNS),
        ("upT_w", C.c_void_p * UNET_MAX_DOWNS), ("upT_b", C.c_void_p * UNET_MAX_DOWNS), ("upT_cout", C.c_int * UNET_MAX_DOWNS),
        ("style", Operand), ("out", C.c_void_p), ("side_stream", C.c_void_p),
    ]


PATCHGAN_MAX_CONVS, MSD_MAX_SCALES = 8, 4


class PatchganDesc(C.Structure):
    """vts_patchgan_desc (include/vts.h): one PatchGAN's training-mode forward as one C call"""
    _fields_ = [
        ("in0", Operand), ("in1", Operand), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("n_convs", C.c_int),
        ("cout", C.c_int * PATCHGAN_MAX_CONVS), ("stride", C.c_int * PATCHGAN_MAX_CONVS),
        ("w", C.c_void_p * PATCHGAN_MAX_CONVS), ("b", C.c_void_p * PATCHGAN_MAX_CONVS),
        ("gamma", C.c_void_p * PATCHGAN_MAX_CONVS), ("beta", C.c_void_p * PATCHGAN_MAX_CONVS),
        ("running_mean", C.c_void_p * PATCHGAN_MAX_CONVS), ("running_var", C.c_void_p * PATCHGAN_MAX_CONVS),
        ("num_batches_tracked", C.c_void_p * PATCHGAN_MAX_CONVS),
        ("stat_mean_out", C.c_void_p * PATCHGAN_MAX_CONVS), ("stat_uvar_out", C.c_void_p * PATCHGAN_MAX_CONVS),
        ("eps", C.c_float), ("momentum", C.c_float), ("run_head", C.c_int), ("pred", C.c_void_p),
    ]


class MsdDesc(C.Structure):
    """vts_msd_desc: the multiscale discriminator's training-mode forward as one C call"""
    _fields_ = [("num_D", C.c_int), ("scale", PatchganDesc * MSD_MAX_SCALES)]


class PatchJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_nstride", C.c_int64), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("img", C.c_void_p), ("offx", C.c_void_p), ("offy", C.c_void_p), ("P", C.c_int), ("dst", C.c_void_p),
                ("dst_C", C.c_int), ("dst_c0", C.c_int), ("fill", C.c_float)]


class NormBwdDesc(C.Structure):
    _fields_ = [
        ("dy", C.c_void_p), ("x", C.c_void_p), ("nstride", C.c_int64), ("N", C.c_int), ("C", C.c_int), ("HW", C.c_int),
        ("mode", C.c_int), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p),
        ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("accumulate_param_grads", C.c_int), ("counters", C.c_void_p),
        ("ngroups", C.c_int), ("gstart", C.c_int * 9),
    ]


_lib = None

# every symbol include/vts.h declares (tests/test_abi.py checks the export list against the header)
SYMBOLS = [
    "vts_last_error", "vts_last_kernel", "vts_version", "vts_capture_node_count", "vts_conv4x4", "vts_conv4x4_in", "vts_conv4x4_norm", "vts_conv4x4_norm_ws_floats", "vts_norm_stats_from_partials", "vts_conv4x4_ws_floats", "vts_wgrad4x4_ws_floats", "vts_wgrad4x4", "vts_wgrad_reduce_batch", "vts_channel_sum",
    "vts_channel_sum_ws_floats", "vts_norm_ws_floats", "vts_norm_stats", "vts_norm_bwd", "vts_act_bwd",
    "vts_avgpool3s2", "vts_avgpool3s2_bwd", "vts_ganloss", "vts_l1", "vts_patch_gather", "vts_patch_scatter_bwd",
    "vts_g_post", "vts_diffaug_bs_mask", "vts_diffaug_op", "vts_diffaug_op_ws_floats", "vts_g_out_grad", "vts_g_out_grad_pool", "vts_pool_query", "vts_mask_mul", "vts_input_images_u8", "vts_spe_grid", "vts_mask_candidates",
    "vts_pad_affine", "vts_pad_bwd", "vts_blur_down", "vts_blur_down_bwd", "vts_blur_up", "vts_blur_up_bwd", "vts_tap_embed", "vts_tap_extract", "vts_tap_embed_at", "vts_tap_extract_at", "vts_w3x3_pack", "vts_conv3x3_wide", "vts_w3x3_wino_floats", "vts_w3x3_wino_pack", "vts_conv3x3_wino_ok", "vts_conv3x3_wino", "vts_conv3x3_wide_relu_pad", "vts_conv3x3_wide_mask_pad", "vts_zero_border", "vts_conv3x3_wide_ws_floats", "vts_conv3x3s2_wide", "vts_tconv3x3s2_wide", "vts_wgrad3x3_wide", "vts_wgrad3x3_wide_ws_floats", "vts_upfirdn2d_out_size", "vts_upfirdn2d", "vts_upfirdn2d_bwd", "vts_bias_act", "vts_bias_act_bwd", "vts_modconv_demod", "vts_w4x4_pack", "vts_conv4x4_flat_ok", "vts_conv4x4_wide_ws_floats", "vts_conv4x4_wide", "vts_wgrad4x4_wide_ws_floats", "vts_wgrad4x4_wide",
    "vts_metric_ws_floats", "vts_minmax", "vts_metric_psnr", "vts_metric_tactile", "vts_metric_ssim", "vts_frechet_ws_floats", "vts_frechet_distance", "vts_sifid_input", "vts_modconv_weight", "vts_modconv_weight_bwd", "vts_adain", "vts_adain_bwd", "vts_resample_table",
    "vts_mask_select", "vts_mask_sample_ranks", "vts_adam_flat", "vts_adam_flat_dev", "vts_patchnce", "vts_l2norm_rows", "vts_patch_sample", "vts_linear_rows", "vts_copy_words",
    "vts_maxpool2_relu_pad", "vts_maxpool3s2_relu_pad", "vts_s2d4_pad", "vts_maxpool2_relu_bwd", "vts_relu_mask_pad", "vts_lpips_layer", "vts_l1_relu", "vts_lpips_input", "vts_lpips_input_bwd",
    "vts_patch_jobs", "vts_g_post_stack", "vts_step_begin", "vts_conv4x4_bsums", "vts_norm_bwd_from_partials",
    "vts_u8_expand", "vts_unet_forward", "vts_unet_forward_ws_floats", "vts_patchgan_forward", "vts_patchgan_forward_ws_floats", "vts_msd_forward", "vts_msd_forward_ws_floats", "vts_allreduce_slice_plan", "vts_comm_unique_id", "vts_comm_init", "vts_allreduce_flat_async", "vts_allreduce_flat_wait", "vts_comm_destroy",
]


def load():
    """Load libvts_hip.so (once).  Raises if it is not built: the product has no CPU/eager path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvts_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C visual-tactile-synthesis_amd/csrc`.  There is no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.vts_last_error.restype = C.c_char_p
    lib.vts_last_kernel.restype = C.c_char_p
    lib.vts_metric_ws_floats.argtypes = []
    lib.vts_frechet_ws_floats.argtypes = []
    lib.vts_frechet_ws_floats.restype = C.c_int64
    lib.vts_metric_ws_floats.restype = C.c_int64
    lib.vts_conv3x3_wide_ws_floats.argtypes = [C.c_int] * 5
    lib.vts_wgrad3x3_wide_ws_floats.argtypes = [C.c_int] * 6
    lib.vts_conv4x4_wide_ws_floats.argtypes = [C.c_int] * 8
    lib.vts_wgrad4x4_wide_ws_floats.argtypes = [C.c_int] * 6
    lib.vts_wgrad4x4_wide_ws_floats.restype = C.c_int64
    lib.vts_conv4x4_wide_ws_floats.restype = C.c_int64
    lib.vts_conv4x4_flat_ok.argtypes = [C.c_int] * 5
    lib.vts_upfirdn2d_out_size.argtypes = [C.c_int] * 6
    lib.vts_upfirdn2d_out_size.restype = C.c_int
    lib.vts_conv4x4_flat_ok.restype = C.c_int
    for name in ("vts_wgrad4x4_ws_floats", "vts_norm_ws_floats", "vts_channel_sum_ws_floats", "vts_conv4x4_ws_floats", "vts_conv3x3_wide_ws_floats",
                 "vts_wgrad3x3_wide_ws_floats"):
        getattr(lib, name).restype = C.c_int64
    lib.vts_conv4x4_ws_floats.argtypes = [C.POINTER(ConvDesc)]
    lib.vts_unet_forward_ws_floats.argtypes = [C.POINTER(UnetDesc)]
    lib.vts_w3x3_wino_floats.argtypes = [C.c_int, C.c_int]
    lib.vts_w3x3_wino_floats.restype = C.c_int64
    lib.vts_unet_forward_ws_floats.restype = C.c_int64
    lib.vts_patchgan_forward_ws_floats.argtypes = [C.POINTER(PatchganDesc)]
    lib.vts_patchgan_forward_ws_floats.restype = C.c_int64
    lib.vts_msd_forward_ws_floats.argtypes = [C.POINTER(MsdDesc)]
    lib.vts_msd_forward_ws_floats.restype = C.c_int64
    lib.vts_conv4x4_norm_ws_floats.argtypes = [C.POINTER(ConvDesc)]
    lib.vts_conv4x4_norm_ws_floats.restype = C.c_int64
    lib.vts_norm_ws_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.vts_channel_sum_ws_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    vp, i, i64, f = C.c_void_p, C.c_int, C.c_int64, C.c_float
    sig = {
        "vts_conv4x4": [C.POINTER(ConvDesc), vp],
        "vts_conv4x4_in": [C.POINTER(ConvDesc), C.POINTER(NormDesc), C.POINTER(C.c_int), vp],
        "vts_conv4x4_norm": [C.POINTER(ConvDesc), C.POINTER(NormDesc), vp, i64, C.POINTER(C.c_int), vp],
        "vts_norm_stats_from_partials": [C.POINTER(NormDesc), vp, i, vp],
        "vts_wgrad4x4_ws_floats": [C.POINTER(WgradDesc)],
        "vts_wgrad4x4": [C.POINTER(WgradDesc), vp, vp],
        "vts_wgrad_reduce_batch": [C.POINTER(ReduceJob), i, vp],
        "vts_channel_sum": [vp, i64, i, i, i, vp, i, vp, vp, vp],
        "vts_norm_stats": [C.POINTER(NormDesc), vp, vp],
        "vts_norm_bwd": [C.POINTER(NormBwdDesc), vp, vp],
        "vts_act_bwd": [vp, C.POINTER(Operand), i, i, i, vp, i, vp],
        "vts_avgpool3s2": [vp, i64, i, i, i, i, vp, vp],
        "vts_avgpool3s2_bwd": [vp, i, i, i, i, vp, i64, i, vp],
        "vts_ganloss": [vp, i, i, i, i, f, f, f, vp, vp, vp],
        "vts_l1": [vp, vp, i64, f, vp, vp, i, vp],
        "vts_patch_gather": [vp, i64, i, i, i, vp, vp, vp, i, i, vp, i, i, vp],
        "vts_patch_scatter_bwd": [vp, i, i, i, vp, vp, vp, i, i, i, vp, i64, i, i, i, i, vp],
        "vts_g_post": [vp, vp, i, i, i, f, vp, vp, vp, vp, i64, vp, vp, i64, vp],
        "vts_g_post_stack": [vp, vp, i, i, i, f, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, i64, vp],
        "vts_patch_jobs": [C.POINTER(PatchJob), i, i, vp],
        "vts_step_begin": [vp, i, vp, i, vp],
        "vts_conv4x4_bsums": [C.POINTER(ConvDesc), vp, i64, C.POINTER(C.c_int), vp],
        "vts_norm_bwd_from_partials": [C.POINTER(NormBwdDesc), vp, i, vp, vp],
        "vts_diffaug_bs_mask": [vp, vp, i, i, i, vp, vp, vp, vp],
        "vts_diffaug_op_ws_floats": [i],
        "vts_diffaug_op": [vp, i64, vp, i64, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp],
        "vts_g_out_grad": [vp, vp, vp, vp, i, i, i, vp, vp], "vts_g_out_grad_pool": [vp, vp, vp, vp, vp, i, i, i, vp, vp],
        "vts_mask_mul": [vp, vp, i, i, i, vp, vp],
        "vts_pool_query": [vp, vp, vp, vp, i, i64, vp, vp],
        "vts_spe_grid": [vp, i64, i, i, i, i, vp],
        "vts_input_images_u8": [vp, vp, vp, i, i64, vp, vp, vp, vp, vp],
        "vts_mask_candidates": [vp, i, i, i, vp, vp, vp],
        "vts_mask_select": [vp, vp, i, i, i, vp, i, vp, vp, vp],
        "vts_mask_sample_ranks": [vp, i, i, i, C.c_uint64, vp, vp],
        "vts_adam_flat": [vp, vp, vp, vp, i64, f, f, f, f, i, f, vp],
        "vts_adam_flat_dev": [vp, vp, vp, vp, i64, vp, f, f, f, vp, f, vp],
        "vts_patchnce": [vp, vp, i, i, i, f, f, vp, vp, vp],
        "vts_l2norm_rows": [vp, i, i, vp, vp],
        "vts_copy_words": [vp, vp, i64, vp],
        "vts_patch_sample": [vp, vp, i, i, i, i, vp, vp],
        "vts_linear_rows": [vp, vp, vp, i, i, i, i, vp, vp],
        "vts_minmax": [vp, i64, vp, vp, vp],
        "vts_metric_psnr": [vp, vp, i64, vp, vp, vp, vp],
        "vts_metric_ssim": [vp, vp, i, i, i, vp, vp, vp, vp],
        "vts_frechet_distance": [vp, vp, i, i64, i64, vp, vp, vp],
        "vts_sifid_input": [vp, i64, i, i, i, i, i, vp, i, vp, i, i, vp],
        "vts_adain": [vp, vp, i, i, f, vp, vp],
        "vts_resample_table": [vp, i64, i, i, vp, vp, vp, i, vp, vp, vp, i, vp, i, i, i, vp],
        "vts_adain_bwd": [vp, vp, vp, i, i, f, vp, vp, vp],
        "vts_modconv_weight": [vp, i, i, i, f, f, i, vp, vp],
        "vts_modconv_weight_bwd": [vp, vp, i, i, i, f, f, i, vp, i, vp],
        "vts_metric_tactile": [vp, vp, i64, i, vp, vp, vp, vp],
        "vts_pad_affine": [C.POINTER(Operand), i, i, i, i, i, i, i, i, i, vp, vp, i64, vp],
        "vts_pad_bwd": [vp, i, i, i, i, i, i, i, i, i, vp, i, vp],
        "vts_blur_down": [C.POINTER(Operand), i, i, i, i, vp, vp],
        "vts_blur_down_bwd": [vp, i, i, i, i, vp, i, vp],
        "vts_blur_up": [C.POINTER(Operand), i, i, i, i, vp, vp],
        "vts_blur_up_bwd": [vp, i, i, i, i, vp, i, vp],
        "vts_tap_embed": [vp, i64, i, i, i, vp, vp],
        "vts_tap_extract": [vp, i64, i, i, i, vp, i, vp],
        "vts_tap_embed_at": [vp, i64, i, i, i, vp, vp],
        "vts_tap_extract_at": [vp, i64, i, i, i, vp, i, vp],
        "vts_w3x3_pack": [vp, i, i, i64, i64, i, vp, vp],
        "vts_w4x4_pack": [vp, i, i, i64, i64, i, vp, vp],
        "vts_wgrad4x4_wide": [vp, vp, vp, i, i, i, i, i, i, i, i, i, vp, i64, vp],
        "vts_upfirdn2d": [vp, i64, i, i, vp, i, i, i, i, i, i, i, i, vp, i, vp],
        "vts_upfirdn2d_bwd": [vp, i64, i, i, vp, i, i, i, i, i, i, i, i, vp, i, vp],
        "vts_bias_act": [vp, vp, vp, i, i, i64, f, f, vp, vp],
        "vts_bias_act_bwd": [vp, vp, vp, i, i, i64, f, f, vp, vp],
        "vts_modconv_demod": [vp, vp, i, i, i, i, f, f, vp, vp],
        "vts_conv4x4_wide": [vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp, i64, vp],
        "vts_conv3x3s2_wide": [vp, vp, vp, vp, i, i, i, i, i, vp, i64, vp],
        "vts_tconv3x3s2_wide": [vp, vp, vp, vp, i, i, i, i, i, vp, i64, vp],
        "vts_conv3x3_wide": [vp, vp, vp, vp, i, i, i, i, i, vp, i64, vp],
        "vts_wgrad3x3_wide": [vp, vp, vp, i, i, i, i, i, i, i, vp, i64, vp],
        "vts_maxpool2_relu_pad": [vp, i, i, i, i, vp, i, vp],
        "vts_conv3x3_wide_relu_pad": [vp, vp, vp, vp, i, i, i, i, i, vp],
        "vts_conv3x3_wide_mask_pad": [vp, vp, vp, i, i, i, i, i, vp, vp, vp],
        "vts_w3x3_wino_pack": [vp, i, i, i64, i64, i, vp, vp], "vts_conv3x3_wino_ok": [i, i, i, i, i],
        "vts_conv3x3_wino": [vp, vp, vp, vp, i, i, i, i, i, i, i, vp, vp, vp],
        "vts_zero_border": [vp, i64, i, i, i, vp],
        "vts_maxpool3s2_relu_pad": [vp, i, i, i, i, vp, vp],
        "vts_u8_expand": [vp, i64, i, vp, vp], "vts_unet_forward": [C.POINTER(UnetDesc), vp, i64, vp], "vts_patchgan_forward": [C.POINTER(PatchganDesc), vp, i64, vp], "vts_msd_forward": [C.POINTER(MsdDesc), vp, i64, vp], "vts_comm_unique_id": [vp], "vts_comm_init": [vp, i, i, vp], "vts_allreduce_flat_async": [vp, vp, i64, vp],
        "vts_allreduce_flat_wait": [vp, vp], "vts_comm_destroy": [vp], "vts_allreduce_slice_plan": [i64, i, i, vp, vp, vp, vp],
        "vts_s2d4_pad": [vp, i, i, i, i, i, i, i, vp, vp],
        "vts_maxpool2_relu_bwd": [vp, vp, i, i, i, vp, i, vp, i, vp],
        "vts_relu_mask_pad": [vp, vp, vp, i, i, i, i, vp, i, vp],
        "vts_lpips_layer": [vp, vp, i, i, i, vp, f, vp, vp, f, i, i, vp],
        "vts_l1_relu": [vp, vp, i64, f, vp, vp, vp],
        "vts_lpips_input": [vp, i64, i, i, i, c_f32p, c_f32p, vp, vp],
        "vts_lpips_input_bwd": [vp, i, i, i, c_f32p, vp, i64, i, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        if name not in ("vts_wgrad4x4_ws_floats",):
            fn.restype = C.c_int
    _lib = lib
    return lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().vts_last_error().decode()))


def ptr(t):
    return None if t is None else t.data_ptr()


def operand(t, scale=None, shift=None, C_=None, nstride=None):
    """Operand view of a contiguous NCHW tensor (or a channel slice given C_/nstride)."""
    if t is None:
        return Operand(None, None, None, 0, 0)
    assert t.dtype == torch.float32 and t.is_cuda
    c = t.shape[1] if C_ is None else C_
    ns = t.stride(0) if nstride is None else nstride
    return Operand(t.data_ptr(), ptr(scale), ptr(shift), c, ns)
