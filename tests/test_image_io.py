"""Output writers (util/image_io.py) against the reference's util.tensor2im / tensor2arr semantics (util/util.py:58-122, restated:
clamp to (-1, 1), map to 0..255 with truncation, gray -> 3 channels) and the file set of save_images (util/visualizer.py:30-148)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "visual-tactile-synthesis_amd"))


def test_tensor2im_tensor2arr_and_save_images(tmp_path):
    from util.image_io import save_images, tensor2arr, tensor2im
    x = torch.linspace(-1.5, 1.5, 2 * 3 * 4 * 5).reshape(2, 3, 4, 5)
    im = tensor2im(x)
    ref = ((np.clip(x[0].numpy(), -1, 1) + 1) / 2.0).transpose(1, 2, 0) * 255.0
    assert im.dtype == np.uint8 and im.shape == (4, 5, 3) and np.array_equal(im, ref.astype(np.uint8))
    g = tensor2im(x[:, :1])
    assert g.shape == (4, 5, 3) and np.array_equal(g[..., 0], g[..., 2])
    assert tensor2im(x[0, 0]).shape == (4, 5, 3)
    a = tensor2arr(x[:1, :1], imtype=np.float32)
    assert a.shape == (1, 4, 5) and a.dtype == np.float32 and np.allclose(a[0], np.clip(x[0, 0].numpy(), -1, 1))
    visuals = {"fake_I": x[:1], "fake_gx": x[:1, :1], "fake_gy": -x[:1, :1], "val_patch_coords": np.array([[10, 20], [30, 40]])}
    out = str(tmp_path)
    written = save_images(out, visuals, ["/some/dir/sample7.png"], save_raw_gxgy=True, save_raw_arr_vis=True)
    assert sorted(os.path.relpath(p, out) for p in written) == ["fake_I/sample7.png", "fake_gx/sample7.png", "fake_gy/sample7.png"]
    raw = np.load(os.path.join(out, "fake_gxgy_raw", "fake_gxgy_raw.npz"))
    assert set(raw.files) == {"fake_gx", "fake_gy"} and raw["fake_gx"].dtype == np.float64 and raw["fake_gx"].shape == (1, 4, 5)
    assert np.load(os.path.join(out, "fake_gx", "sample7.npy")).shape == (4, 5)
    import json
    c = json.load(open(os.path.join(out, "val_patch_coords.json")))["coords"]
    assert c == {"x": [10, 30], "y": [1516, 1496], "len": 2}
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(os.path.join(out, "fake_I", "sample7.png"))), tensor2im(x[:1]))


def test_tensor2im_matches_reference(golden_dir):
    """against the arrays the REFERENCE's util.tensor2im / tensor2arr produced for the same ramp tensor (tests/golden/image_io.npz)"""
    from util.image_io import tensor2arr, tensor2im
    g = np.load(os.path.join(golden_dir, "image_io.npz"))
    x = torch.linspace(-1.5, 1.5, 2 * 3 * 4 * 5).reshape(2, 3, 4, 5)
    assert np.array_equal(tensor2im(x), g["im_rgb"]) and np.array_equal(tensor2im(x[:, :1]), g["im_gray"])
    assert np.array_equal(tensor2im(x[0, 0]), g["im_2d"])
    assert np.array_equal(tensor2arr(x[:1, :1], imtype=np.float32), g["arr_gray"]) and np.array_equal(tensor2arr(x[:1]), g["arr_rgb"])
