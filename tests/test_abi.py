"""The C-ABI library loads and exports every symbol include/vts.h declares (no compute calls: CPU-safe)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "visual-tactile-synthesis_amd", "libvts_hip.so")
HEADER = os.path.join(ROOT, "include", "vts.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vts_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("vts_conv4x4", "vts_wgrad4x4", "vts_norm_stats", "vts_norm_bwd", "vts_ganloss", "vts_patch_gather",
                 "vts_patch_scatter_bwd", "vts_adam_flat", "vts_patchnce", "vts_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "visual-tactile-synthesis_amd", "csrc"), "-j8"])
    lib = ctypes.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.vts_version.restype = ctypes.c_int
    assert lib.vts_version() >= 1
    lib.vts_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.vts_last_error(), bytes)


def test_binding_lists_the_same_symbols():
    from vts import lib as L

    assert sorted(L.SYMBOLS) == declared_symbols()


def test_argument_errors_are_reported_not_crashed():
    """Null-pointer / bad-shape descriptors are rejected on the host before any launch."""
    from vts import lib as L

    lib = L.load()
    d = L.ConvDesc()
    rc = lib.vts_conv4x4(ctypes.byref(d), None)
    assert rc == -1 and b"null pointer" in lib.vts_last_error()
    n = L.NormDesc()
    assert lib.vts_norm_stats(ctypes.byref(n), None, None) == -1


def test_product_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from models import create_model
    from options.train_options import TrainOptions

    opt = TrainOptions(cmd_line="--model sinskitG --gpu_ids -1 --lambda_G1_lpips 0 --lambda_G2_lpips 0 "
                                "--use_vision_aided_loss False --checkpoints_dir /tmp/vts_t --name a").parse()
    with pytest.raises(RuntimeError, match="HIP path only"):
        create_model(opt)
