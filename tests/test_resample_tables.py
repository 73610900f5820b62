"""Host-side tables of the anti-aliased bicubic resampler (vts/ops.py:_aa_axis / _aa_transpose) against PyTorch's own CPU kernel:
applying the tables as dense matrices must reproduce F.interpolate(mode="bicubic", align_corners=False, antialias=True), and the
transposed tables must be the transposed matrices (no GPU involved)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-tactile-synthesis_amd"))


def dense(mins, sizes, w, n_in):
    m = np.zeros((len(mins), n_in), np.float32)
    for o in range(len(mins)):
        m[o, mins[o]:mins[o] + sizes[o]] = w[o, :sizes[o]]
    return m


@pytest.mark.parametrize("geom", [(32, 32, 64, 64), (40, 40, 32, 32), (64, 48, 32, 32), (37, 53, 32, 32), (32, 32, 128, 128), (100, 100, 25, 25), (32, 32, 32, 32)])
def test_tables_reproduce_torch_antialiased_bicubic(geom):
    from vts import ops
    ih, iw, oh, ow = geom
    x = torch.randn(2, 3, ih, iw, generator=torch.Generator().manual_seed(5))
    ref = F.interpolate(x, (oh, ow), mode="bicubic", align_corners=False, antialias=True)
    ay, ax = ops._aa_axis(ih, oh), ops._aa_axis(iw, ow)
    wy, wx = dense(*ay, ih), dense(*ax, iw)
    got = torch.einsum("oh,nchw,pw->ncop", torch.from_numpy(wy), x, torch.from_numpy(wx))
    assert float((got - ref).abs().max()) < 5e-6
    assert np.array_equal(dense(*ops._aa_transpose(*ay, ih), oh), wy.T) and np.array_equal(dense(*ops._aa_transpose(*ax, iw), ow), wx.T)
    if (ih, iw) == (oh, ow):
        assert np.array_equal(wy, np.eye(ih, dtype=np.float32))     # the equal-size resamples of the default configuration are identities
