"""Experiment switches of the Python layer (schedules, lane packing, kernel-family thresholds, "run the round-3 path" knobs).

They exist for same-box A/B measurements (tools/ab_env.sh, tools/probes/*) and the bit-compatibility tests between two schedules; two of
them (VTS_KO_LANES, VTS_KNOCKOUT) knowingly break the arithmetic for timing experiments.  None is a user option, so they are read from
the environment ONLY while VTS_TUNING=1 is set: without it every switch has its measured default and a stray VTS_* variable in a user's
shell changes nothing.  (The C library has the same rule at compile time: csrc/vts_internal.h:vts_tune, make PROFILING=1.)

What a user may set, always honoured: VTS_LIB_PATH (another build of the library), VTS_DDP_BACKEND / VTS_DDP_DIRECT / VTS_DDP_FORCE /
VTS_G_BUCKETS / VTS_G_BUCKET_MIN_MB (data parallel), VTS_{LPIPS,LPIPS_ALEX,VGG,INCEPTION}_WEIGHTS (weight files), VTS_SIFID,
VTS_LPIPS_METRICS (optional metrics), VTS_RCCL_LIB (C library)."""
import os


def on():
    return os.environ.get("VTS_TUNING", "0") == "1"


def get(name, default):
    """the environment's value of an experiment switch while VTS_TUNING=1, else `default`"""
    return os.environ.get(name, default) if on() else default


def is_set(name):
    return on() and bool(os.environ.get(name))
