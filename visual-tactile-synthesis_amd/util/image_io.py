"""Output writers of the inference flow (reference util/visualizer.py:30-148 save_images, util/util.py:58-122 tensor2im / tensor2arr)
without the HTML page / wandb: per visual a PNG under <image_dir>/<label>/<name>.png, the raw tactile gradients of every 'gx' / 'gy'
visual in <image_dir>/fake_gxgy_raw/fake_gxgy_raw.npz (what the 3-D reconstruction / rendering post-processing reads), and with
save_raw_arr_vis the float arrays next to the PNGs as .npy.  The reference's .exr copies need skimage / OpenEXR, which this image does
not have: they are skipped (the .npy holds the same array)."""
import json
import ntpath
import os

import numpy as np
import torch


def tensor2im(input_image, imtype=np.uint8):
    """(-1, 1) tensor [1|3, H, W] (or [N, ...]: first item; [H, W]: one channel) -> uint8 HWC image (util/util.py:58-92)"""
    if isinstance(input_image, np.ndarray):
        return input_image.astype(imtype)
    if not isinstance(input_image, torch.Tensor):
        return input_image
    t = input_image.detach()
    if t.dim() == 2:
        t = t[None, None]
    if t.dim() == 3:
        t = t[None]
    a = t[0].clamp(-1.0, 1.0).cpu().float().numpy()
    a = (a + 1) / 2.0
    a = np.transpose(np.tile(a, (3, 1, 1)), (1, 2, 0)) if a.shape[0] == 1 else np.transpose(a, (1, 2, 0))
    return (a * 255.0).astype(imtype)


def tensor2arr(input_image, imtype=np.float64, convert2RGB=False):
    """raw clamped (-1, 1) array [C, H, W] of the first item (util/util.py:95-122)"""
    if isinstance(input_image, np.ndarray):
        a = input_image
    else:
        t = torch.squeeze(input_image.detach())
        if t.dim() == 2:
            t = t[None, None]
        if t.dim() == 3:
            t = t[None]
        a = t[0].clamp(-1.0, 1.0).cpu().float().numpy()
    if convert2RGB and a.shape[0] == 1:
        a = np.tile(a, (3, 1, 1))
    return a.astype(imtype)


def save_image(image_numpy, image_path):
    from PIL import Image
    a = image_numpy
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    Image.fromarray(a).save(image_path)


def save_images(image_dir, visuals, image_path, save_raw_gxgy=False, save_raw_arr_vis=False, style_image_name=None):
    """writes the files the reference's save_images writes (minus HTML / wandb / .exr); returns the list of PNG paths"""
    name = os.path.splitext(ntpath.basename(image_path[0] if isinstance(image_path, (list, tuple)) else image_path))[0]
    if style_image_name is not None:
        name += "_style_%s" % style_image_name
    if save_raw_gxgy:
        raw = {label: tensor2arr(v) for label, v in visuals.items() if "gx" in label or "gy" in label}
        os.makedirs(os.path.join(image_dir, "fake_gxgy_raw"), exist_ok=True)
        np.savez(os.path.join(image_dir, "fake_gxgy_raw", "fake_gxgy_raw.npz"), **raw)
    written = []
    for label, v in visuals.items():
        if "patch_coords" in label:
            c = np.asarray(v).astype(int)
            with open(os.path.join(image_dir, label + ".json"), "w") as f:
                json.dump({"coords": {"x": c[:, 0].tolist(), "y": (1536 - c[:, 1]).tolist(), "len": int(c.shape[0])}}, f)
            continue
        os.makedirs(os.path.join(image_dir, label), exist_ok=True)
        path = os.path.join(image_dir, label, name + ".png")
        save_image(tensor2im(v), path)
        written.append(path)
        if save_raw_arr_vis and ("gx" in label or "gy" in label) and "bb" not in label and "coord" not in label:
            a = tensor2arr(v, imtype=np.float32)
            a = a[0] if a.shape[0] == 1 else (a.transpose(1, 2, 0) if a.shape[0] == 3 else a)
            np.save(path.replace(".png", ".npy"), a)
    return written
