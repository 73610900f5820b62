"""Small helpers the options/model layer needs.

Mirrors the behaviour of the reference helpers it replaces:
  str2bool  -> /root/reference/util/util.py:22-30
  mkdirs    -> /root/reference/util/util.py (mkdirs/mkdir)
"""
import argparse
import os


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def mkdirs(paths):
    if isinstance(paths, (list, tuple)):
        for p in paths:
            os.makedirs(p, exist_ok=True)
    else:
        os.makedirs(paths, exist_ok=True)
