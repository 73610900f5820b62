"""Small helpers the options/model layer needs.

Mirrors the behaviour of the reference helpers it replaces:
  str2bool  -> /root/reference/util/util.py:22-30
  mkdirs    -> /root/reference/util/util.py (mkdirs/mkdir)
"""
import argparse
import os


from vts.misc import str2bool  # noqa: E402,F401  (one definition; the models package imports it from vts.misc)


def mkdirs(paths):
    if isinstance(paths, (list, tuple)):
        for p in paths:
            os.makedirs(p, exist_ok=True)
    else:
        os.makedirs(paths, exist_ok=True)
