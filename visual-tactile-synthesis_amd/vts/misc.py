"""Helpers shared by the options / model layers that must not depend on the top-level `util` package: when a maintainer keeps the
reference's own util / data / options packages and swaps only the model (INTEGRATION.md recipe 2), `util` is the reference's."""
import argparse


def str2bool(v):
    """/root/reference/util/util.py:22-30"""
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")
