"""Output writers of the inference flow (reference util/visualizer.py:30-148 save_images, util/util.py:58-122 tensor2im / tensor2arr)
without the HTML page / wandb: per visual a PNG under <image_dir>/<label>/<name>.png, the raw tactile gradients of every 'gx' / 'gy'
visual in <image_dir>/fake_gxgy_raw/fake_gxgy_raw.npz (what the 3-D reconstruction / rendering post-processing reads), and with
save_raw_arr_vis the float arrays next to the PNGs as .npy and .exr.  The reference writes the .exr through skimage / imageio
(visualizer.py:146); neither is in this image, so write_exr below emits the OpenEXR 2 scanline layout itself (uncompressed 32-bit
float channels: Y for one channel, B/G/R for three) -- the file any OpenEXR reader opens; read_exr reads that subset back (tests).
postprocess_gz is the friction map of the rendering post-processing (Step2_Postprocessing_for_Rendering.py:18-140)."""
import json
import ntpath
import os
import struct

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
            write_exr(path.replace(".png", ".exr"), a)
    return written


# ---- OpenEXR 2.0 single-part scanline files, NO_COMPRESSION, FLOAT channels (the subset skimage.io.imsave(..., '.exr') of a float32
# array needs: visualizer.py:146).  Layout (openexr.com "OpenEXR File Layout"): magic, version, attributes (name\0 type\0 size value)
# closed by \0, one 64-bit offset per scanline, then per scanline: y, byte count, the row of every channel in alphabetical order.
_EXR_MAGIC = 20000630


def _exr_attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(path, arr):
    """arr: float array [H, W] (channel Y) or [H, W, 3] (R, G, B)"""
    a = np.asarray(arr, dtype=np.float32)
    if a.ndim == 2:
        planes = {"Y": a}
    elif a.ndim == 3 and a.shape[2] == 3:
        planes = {"R": a[:, :, 0], "G": a[:, :, 1], "B": a[:, :, 2]}
    else:
        raise ValueError("write_exr: expected [H, W] or [H, W, 3], got %s" % (a.shape,))
    h, w = a.shape[:2]
    names = sorted(planes)       # channels are stored in alphabetical order
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for n in names) + b"\0"   # 2 = FLOAT, linear 0, sampling 1 x 1
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    head = struct.pack("<ii", _EXR_MAGIC, 2)
    head += _exr_attr("channels", "chlist", chlist)
    head += _exr_attr("compression", "compression", b"\0")
    head += _exr_attr("dataWindow", "box2i", box)
    head += _exr_attr("displayWindow", "box2i", box)
    head += _exr_attr("lineOrder", "lineOrder", b"\0")
    head += _exr_attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += _exr_attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    head += _exr_attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    head += b"\0"
    row_bytes = 4 * w * len(names)
    first = len(head) + 8 * h
    offsets = struct.pack("<%dQ" % h, *[first + y * (8 + row_bytes) for y in range(h)])
    rows = np.stack([np.ascontiguousarray(planes[n]) for n in names], axis=1).astype("<f4")   # [H, channel, W]
    with open(path, "wb") as f:
        f.write(head)
        f.write(offsets)
        for y in range(h):
            f.write(struct.pack("<ii", y, row_bytes))
            f.write(rows[y].tobytes())


def read_exr(path):
    """reads the files write_exr writes (uncompressed FLOAT scanlines); returns [H, W] (Y) or [H, W, 3] (R, G, B)"""
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", b, 0)
    if magic != _EXR_MAGIC or (version & 0xFF) != 2 or version & 0x200:
        raise ValueError("read_exr: not a single-part scanline OpenEXR 2 file")
    o, attrs = 8, {}
    while b[o] != 0:
        e = b.index(b"\0", o)
        name = b[o:e].decode()
        o = e + 1
        e = b.index(b"\0", o)
        typ = b[o:e].decode()
        o = e + 1
        (size,) = struct.unpack_from("<i", b, o)
        o += 4
        attrs[name] = (typ, b[o:o + size])
        o += size
    o += 1
    if attrs["compression"][1] != b"\0":
        raise ValueError("read_exr: only uncompressed files")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, c, q = [], attrs["channels"][1], 0
    while c[q] != 0:
        e = c.index(b"\0", q)
        names.append(c[q:e].decode())
        (ptype,) = struct.unpack_from("<i", c, e + 1)
        if ptype != 2:
            raise ValueError("read_exr: only FLOAT channels")
        q = e + 1 + 16
    offs = struct.unpack_from("<%dQ" % h, b, o)
    planes = {n: np.empty((h, w), np.float32) for n in names}
    for y in range(h):
        yy, nbytes = struct.unpack_from("<ii", b, offs[y])
        row = np.frombuffer(b, "<f4", w * len(names), offs[y] + 8).reshape(len(names), w)
        for i, n in enumerate(names):
            planes[n][yy - y0] = row[i]
    if names == ["Y"]:
        return planes["Y"]
    return np.stack([planes["R"], planes["G"], planes["B"]], axis=2)


def clahe_u8(img, clip_limit=4.0, tiles=(4, 4)):
    """Contrast-limited adaptive histogram equalisation of a uint8 image [H, W], restated from the published algorithm of OpenCV's
    CLAHE (`cv2.createCLAHE(clipLimit, tileGridSize).apply`, modules/imgproc/src/clahe.cpp, 4.x) -- the reference's default friction-map
    mapping runs it through myutils.equalize_this (myutils.py:103-118).  PARITY UNPINNED: OpenCV is not in this image (requirements.txt
    of the reference does not pin it either), so the restatement is checked against the algorithm's properties only (tests/test_image_io.py).
      * the image is extended to a multiple of the tile grid (BORDER_REFLECT_101 on the bottom / right);
      * per tile: 256-bin histogram; bins clipped at max(1, int(clip_limit * tile_area / 256)); the clipped mass is redistributed
        uniformly (integer batch to every bin, the residual to every (256 / residual)-th bin from bin 0); LUT = round(cdf * 255 / tile_area);
      * per pixel: bilinear interpolation between the LUTs of the four nearest tile CENTRES (float32), rounded half to even."""
    a = np.ascontiguousarray(img, dtype=np.uint8)
    assert a.ndim == 2
    h, w = a.shape
    ty, tx = int(tiles[1]), int(tiles[0])
    ext = a
    if w % tx or h % ty:
        ext = np.pad(a, ((0, ty - h % ty if h % ty else 0), (0, tx - w % tx if w % tx else 0)), mode="reflect")
    th, tw = ext.shape[0] // ty, ext.shape[1] // tx
    area = th * tw
    scale = np.float32(255.0) / np.float32(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    luts = np.empty((ty, tx, 256), np.float32)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(ext[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch, residual = clipped // 256, clipped % 256
                hist += batch
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    hist[idx] += 1
            cdf = np.cumsum(hist).astype(np.float32)
            luts[j, i] = np.clip(np.rint(cdf * scale), 0, 255)
    yf = np.arange(h, dtype=np.float32) * np.float32(1.0 / th) - np.float32(0.5)
    xf = np.arange(w, dtype=np.float32) * np.float32(1.0 / tw) - np.float32(0.5)
    y1, x1 = np.floor(yf).astype(np.int64), np.floor(xf).astype(np.int64)
    ya, xa = (yf - y1).astype(np.float32), (xf - x1).astype(np.float32)
    y2, x2 = np.minimum(y1 + 1, ty - 1), np.minimum(x1 + 1, tx - 1)
    y1, x1 = np.maximum(y1, 0), np.maximum(x1, 0)
    v = a.astype(np.int64)
    l11, l12 = luts[y1[:, None], x1[None, :], v], luts[y1[:, None], x2[None, :], v]
    l21, l22 = luts[y2[:, None], x1[None, :], v], luts[y2[:, None], x2[None, :], v]
    xa_, ya_ = xa[None, :], ya[:, None]
    res = (l11 * (1 - xa_) + l12 * xa_) * (1 - ya_) + (l21 * (1 - xa_) + l22 * xa_) * ya_
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def postprocess_gz(fake_I, M, gx, gy, Tanvas_width=1280, Tanvas_height=800, use_raw_arr=False, thresholding=False, threshold_quantile=0.9,
                   method="log10", compute_gz=True, gz=None, change_bg_color=False, bg_color=(255, 255, 255)):
    """Friction map for haptic rendering from the tactile output (Step2_Postprocessing_for_Rendering.py:18-140): gz = gx^2 + gy^2,
    optional quantile clipping, min-max normalisation, a non-linear mapping ('equalize': CLAHE, restated from OpenCV's algorithm and
    parity-unpinned, see clahe_u8; 'log10' / 'exp2': pinned to the reference), uint8 images and their resized copies for the TanvasTouch
    screen.  'dilation' (skimage Sobel + OpenCV morphology) is not built: upstream it raises NameError (it reads gz_equalize unassigned).
    Returns (gz_im, fake_I_im, gz_postprocess_im, gz_im_Tanvas, fake_I_im_Tanvas, gz_postprocess_im_Tanvas) like the reference."""
    from PIL import Image
    if compute_gz:
        gx, gy = np.asarray(gx, dtype=np.float64), np.asarray(gy, dtype=np.float64)
        if not use_raw_arr:        # PNG inputs, range (0, 255)
            gx = gx / 255.0 * 2.0 - 1
            gy = gy / 255.0 * 2.0 - 1
        gz = gx ** 2 + gy ** 2
    elif gz is None:
        raise ValueError("postprocess_gz: pass gz, or gx and gy with compute_gz=True")
    else:
        gz = np.array(gz, dtype=np.float64)
    if thresholding:
        t = np.quantile(gz, threshold_quantile)
        gz[gz > t] = t
    gz = (gz - np.min(gz)) / (np.max(gz) - np.min(gz))
    if gz.ndim == 2:
        gz = np.tile(gz[:, :, None], (1, 1, 3))
    if method == "equalize":
        # the reference's default (Step2_Postprocessing_for_Rendering.py:88-92): the three identical channels -> bytes -> gray (the same
        # byte: OpenCV's RGB2GRAY weights sum to one) -> CLAHE(clipLimit 4, 4 x 4 tiles) -> min-max; a 2-D map from here on
        eq = clahe_u8((gz[:, :, 0] * 255 if np.max(gz) <= 1 else gz[:, :, 0]).astype(np.uint8), 4.0, (4, 4)).astype(np.float64)
        post = (eq - np.min(eq)) / (np.max(eq) - np.min(eq))
    elif method == "log10":
        post = np.log10(gz * 9.0 + 1.0)      # [0, 1] -> [1, 10] -> log in [0, 1]
    elif method == "exp2":
        post = np.exp2(gz * 3.0 - 3.0)       # [0, 1] -> [-3, 0]
    else:
        # ('dilation' cannot run upstream either: it reads gz_equalize before any assignment, Step2_Postprocessing_for_Rendering.py:95)
        raise NotImplementedError("friction-map mapping '%s' is not built; use 'equalize', 'log10' or 'exp2'" % method)
    post = (post - post.min()) / (np.max(post) - post.min())
    gz_im = np.uint8(gz * 255)
    fake_I_im = np.uint8(fake_I)
    if change_bg_color:
        fake_I_im[np.asarray(M) < 255] = bg_color
    post_im = np.uint8(post * 255)
    size = (Tanvas_width, Tanvas_height)
    return (gz_im, fake_I_im, post_im, np.array(Image.fromarray(gz_im).resize(size)), np.array(Image.fromarray(fake_I_im).resize(size)),
            np.array(Image.fromarray(post_im).resize(size)))
