"""SingleSkitDataset: the CPU front-end that turns one TouchClothing material folder into the batch dict the training step consumes
(SURVEY.md §8b / §8f-3).  PIL + numpy only (the reference needs cv2 and torchvision, which this image does not have).

On-disk format (reference data/singleskit_dataset.py:86-180, data/image_folder.py:27-60, data/dataset_util.py:5-60):
    <dataroot>/{trainS,trainI,trainM}/<one image>          sketch (grayscale), visual image (RGB), object mask (grayscale)
    <dataroot>/{trainT,valT}/**/<name>_tactile.npz         keys gx_raw, gy_raw [h, w] float surface gradients, vision_mask_{x,y,h,w}
                                                           (the GelSight rectangle in the visual image), touch_thresh /
                                                           touch_center_thresh [h, w] contact masks (0/1 or 0/255)
    test phase: testS / testI / testM / testT, no valT.

What it computes, in the reference's order (so that the same `random` / `numpy.random` seeds give the same batch):
    __init__                 np.random.uniform zoom levels (singleskit_dataset.py:178-186)
    per augmentation index   [zoom ->] crop (random position in train, centre in test: dataset_util.py:163-198) -> sides rounded to a
                             multiple of 256 (:216-227) -> S / I in (-1, 1), M in [0, 1] tensors (:318-328)
    per tactile file         GelSight rectangle through padding / zoom / crop / power-2 (:506-566), valid = inside the crop
    per valid rectangle      32 x 32 squares centred on touch_center_thresh pixels whose contact mask x object mask reaches 1
                             (:769-808); `sample_bbox_per_patch` of them (random in train, the middle ones in test, :811-819)
    selection                batch_size_G2 squares: random.choices weighted by the clipped Laplacian variance of the sketch patch
                             (w_resampling, :1078-1082, util/util.py:261-265) or random.sample; validation batch_size_G2_val; test: all
Reference quirks kept on purpose: H, W = S_img.size[:2] is (width, height); `name` is derived from the FIRST CHARACTER of the sketch
path (ntpath.basename(S_path[0]), :392-393); the valid-rectangle lists are indexed by the tactile FILE index (:742-754), which is the
list index only while every rectangle is valid (otherwise the reference raises IndexError or pairs the wrong rectangle -- here: the
same behaviour).  Not carried over: the debugging plots / cv2.imwrite dumps and the `logs/<date>` directory the reference creates."""
import ntpath
import os
from vts import tune
import random
import time

import numpy as np
import torch
import torch.utils.data
from PIL import Image, ImageOps

from vts.misc import str2bool

IMG_EXTENSIONS = [".jpg", ".JPG", ".jpeg", ".JPEG", ".png", ".PNG", ".ppm", ".PPM", ".bmp", ".BMP", ".tif", ".TIF", ".tiff", ".TIFF"]


# ---------------------------------------------------------------------------------------------- file lists (data/image_folder.py)
def make_dataset(directory, max_dataset_size=float("inf")):
    assert os.path.isdir(directory) or os.path.islink(directory), "%s is not a valid directory" % directory
    images = []
    for root, _, fnames in sorted(os.walk(directory, followlinks=True)):
        for fname in fnames:
            if any(fname.endswith(e) for e in IMG_EXTENSIONS):
                images.append(os.path.join(root, fname))
    return images[:min(max_dataset_size, len(images))]


def make_touch_image_dataset(directory, max_dataset_size=float("inf")):
    assert os.path.isdir(directory) or os.path.islink(directory), "%s is not a valid directory for tactile image dataset" % directory
    if len(os.listdir(directory)) == 0:
        print("Empty directory for %s, return empty list for touch data" % directory)
        return []
    paths = []
    for root, _, fnames in sorted(os.walk(directory, followlinks=True)):
        for fname in fnames:
            if fname.endswith("_tactile.npz"):
                paths.append(os.path.join(root, fname))
    return paths[:min(max_dataset_size, len(paths))]


# --------------------------------------------------------------------------------------- tactile npz (data/dataset_util.py:5-60)
def touch_data_loader(path, convert2im=True, verbose=False, return_mask=True):
    """-> gx, gy, ROI_x, ROI_y, ROI_h, ROI_w, touch_mask, touch_center_mask (masks normalised to [0, 1])"""
    z = np.load(path)
    ROI_x, ROI_y, ROI_h, ROI_w = z["vision_mask_x"], z["vision_mask_y"], z["vision_mask_h"], z["vision_mask_w"]
    gx, gy = z["gx_raw"], z["gy_raw"]
    if convert2im:   # [-1, 1] -> 8-bit grayscale
        gx = Image.fromarray(np.uint8((gx + 1) / 2 * 255), "L")
        gy = Image.fromarray(np.uint8((gy + 1) / 2 * 255), "L")
    touch_mask = touch_center_mask = None
    if return_mask:
        assert "touch_thresh" in z.files, "touch_thresh not found in npz_data"
        assert "touch_center_thresh" in z.files, "touch_center_thresh not found in npz_data"
        touch_mask, touch_center_mask = z["touch_thresh"], z["touch_center_thresh"]
        if np.max(touch_mask) > 1:
            touch_mask = touch_mask / 255
        if np.max(touch_center_mask) > 1:
            touch_center_mask = touch_center_mask / 255
    return gx, gy, ROI_x, ROI_y, ROI_h, ROI_w, touch_mask, touch_center_mask


# ----------------------------------------------------------------- image transforms and their rectangle maps (dataset_util.py:150-240)
def zoom_find_coords(x, y, h, w, scale_factor_h=1, scale_factor_w=1):
    return x * scale_factor_w, y * scale_factor_h, h * scale_factor_h, w * scale_factor_w


def zoom_img(img, scale_factor_h=1, scale_factor_w=1, method=Image.BICUBIC):
    ow, oh = img.size
    return img.resize((int(round(ow * scale_factor_w)), int(round(oh * scale_factor_h))), method)


def get_params(size, crop_size_h=512, crop_size_w=512, center_w=0, center_h=0, center_crop=False):
    w, h = size
    assert w >= crop_size_w and h >= crop_size_h, "The image is smaller than crop_size. Cannot perform get_params for cropping"
    assert crop_size_h >= center_h and crop_size_w >= center_w, "crop_size h {} w {} cannot cover the center region h {} w {}".format(
        crop_size_h, crop_size_w, center_h, center_w)
    if center_crop:
        return (w - crop_size_w) // 2, (h - crop_size_h) // 2
    if center_w > 0 or center_h > 0:
        buffer = min(np.maximum(0, (w - center_w) // 2), np.maximum(0, (h - center_h) // 2), h - crop_size_h, w - crop_size_w)
        x = random.randint(0, buffer)
        y = random.randint(0, buffer)
    else:
        x = random.randint(0, np.maximum(0, w - crop_size_w))
        y = random.randint(0, np.maximum(0, h - crop_size_h))
    return x, y


def crop_img(img, crop_size_h, crop_size_w, method=Image.BICUBIC, resize_ratio=None, crop_pos_x=None, crop_pos_y=None, center_w=0, center_h=0,
             center_crop=False):
    w, h = img.size
    if resize_ratio is None:
        resize_ratio = 1 if (w >= crop_size_w and h >= crop_size_h) else max(crop_size_w / w, crop_size_h / h)
    img = img.resize((int(round(w * resize_ratio)), int(round(h * resize_ratio))), method)
    if crop_pos_x is None and crop_pos_y is None:
        crop_pos_x, crop_pos_y = get_params(img.size, crop_size_h=crop_size_h, crop_size_w=crop_size_w, center_w=center_w, center_h=center_h,
                                            center_crop=center_crop)
    return img.crop((crop_pos_x, crop_pos_y, crop_pos_x + crop_size_w, crop_pos_y + crop_size_h)), resize_ratio, crop_pos_x, crop_pos_y


def crop_find_coords(x, y, h, w, crop_size_h, crop_size_w, resize_ratio, crop_pos_x, crop_pos_y):
    x, y, h, w = x * resize_ratio, y * resize_ratio, h * resize_ratio, w * resize_ratio
    nx, ny = x - crop_pos_x, y - crop_pos_y
    valid = not (nx < 0 or nx + w > crop_size_w or ny < 0 or ny + h > crop_size_h)
    return valid, nx, ny, h, w


def make_power_2_img(img, base, method=Image.BICUBIC):
    ow, oh = img.size
    h, w = int(round(oh / base) * base), int(round(ow / base) * base)
    if h == oh and w == ow:
        return img, 1, 1
    return img.resize((w, h), method), w / ow, h / oh


def make_power_2_find_coords(x, y, h, w, resize_ratio_w, resize_ratio_h):
    return x * resize_ratio_w, y * resize_ratio_h, h * resize_ratio_h, w * resize_ratio_w


def global_padding_find_coords(x, y, h, w, org_w=1280, org_h=960, padded_size=1600):
    return x + (padded_size - org_w) // 2, y + (padded_size - org_h) // 2, h, w


# ---------------------------------------------------------------------------------------- tensors (torchvision ToTensor / Normalize)
def to_tensor(pic):
    """torchvision.transforms.ToTensor: uint8 PIL image / HWC uint8 array -> CHW float in [0, 1]; a float [H, W] array keeps its
    dtype and values and gains a leading channel axis"""
    if isinstance(pic, Image.Image):
        a = np.array(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t
    a = np.array(pic)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


def to_u8(pic):
    """the uint8 CHW tensor to_tensor() starts from, or None when the picture is not 8-bit (optional S_u8 / I_u8 / M_u8 batch keys: the
    model uploads those instead of the float tensors and expands them on the device, bit for bit -- a quarter of the PCIe bytes)"""
    if pic is None:
        return None
    a = np.array(pic)
    if a.dtype != np.uint8:
        return None
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


def normalize_half(t):
    """Normalize(mean 0.5, std 0.5): [0, 1] -> [-1, 1]"""
    return (t - 0.5) / 0.5


def variance_of_laplacian(image, ref=None):
    """util/util.py:261-265: cv2.Laplacian(image - ref, CV_64F).var() -- aperture 1 = the 4-neighbour kernel [[0,1,0],[1,-4,1],[0,1,0]],
    border BORDER_REFLECT_101; `image - ref` is evaluated in the image's own dtype (uint8 patches wrap around, as in the reference).
    cv2 is absent from this image: this restates OpenCV's documented operator (parity unpinned for this one function)."""
    if ref is None:
        ref = np.ones_like(image) * 127
    d = (image - ref).astype(np.float64)
    p = np.pad(d, 1, mode="reflect") if d.ndim == 2 else np.pad(d, ((1, 1), (1, 1), (0, 0)), mode="reflect")
    lap = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] - 4.0 * p[1:-1, 1:-1]
    return lap.var()


class SingleSkitDataset(torch.utils.data.Dataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        """flags and phase defaults of the reference (singleskit_dataset.py:43-84)"""
        parser.add_argument("--subdir_S", type=str, default="trainS", help="subdirectory for S input")
        parser.add_argument("--subdir_I", type=str, default="trainI", help="subdirectory for I input")
        parser.add_argument("--subdir_T", type=str, default="trainT", help="subdirectory for T input")
        parser.add_argument("--subdir_M", type=str, default="trainM", help="subdirectory for mask input")
        parser.add_argument("--subdir_valT", type=str, default="valT", help="subdirectory for T input for validation")
        parser.add_argument("--is_train", type=str2bool, default=True, help="whether the model is in training mode")
        if is_train:
            parser.set_defaults(subdir_S="trainS", subdir_I="trainI", subdir_T="trainT", subdir_M="trainM", subdir_valT="valT", is_train=True)
        else:
            parser.set_defaults(subdir_S="testS", subdir_I="testI", subdir_T="testT", subdir_M="testM", subdir_valT=None, is_train=False)
        return parser

    def __init__(self, opt, verbose=False, default_len=1000):
        self.opt = opt
        self.root = opt.dataroot
        self.current_epoch = 0
        self.verbose = verbose
        self.data_dict = {}
        self.data_len = opt.data_len if hasattr(opt, "data_len") else default_len
        self.dir_S, self.dir_I = os.path.join(opt.dataroot, opt.subdir_S), os.path.join(opt.dataroot, opt.subdir_I)
        self.dir_T, self.dir_M = os.path.join(opt.dataroot, opt.subdir_T), os.path.join(opt.dataroot, opt.subdir_M)
        self.is_train = opt.is_train
        if opt.subdir_valT is not None:
            self.dir_valT = os.path.join(opt.dataroot, opt.subdir_valT)
            assert os.path.exists(self.dir_valT), "missing val T data for train datasets {}".format(self.dir_valT)
        assert os.path.exists(self.dir_S), "missing S data for datasets {}".format(self.dir_S)
        self.S_paths = sorted(make_dataset(self.dir_S, opt.max_dataset_size))
        assert len(self.S_paths) == 1, "SingleSkitDataset class should be used with one image in sketch S_paths {}".format(self.S_paths)
        if opt.sketch_nc == 1:
            self.S_img = ImageOps.grayscale(Image.open(self.S_paths[0]))
        else:
            assert opt.sketch_nc == 3, "Load sketch either in grayscale or RGB"
            self.S_img = Image.open(self.S_paths[0]).convert("RGB")
        self.M_paths, self.M_img = [], None
        if opt.use_bg_mask is True:
            assert os.path.exists(self.dir_M), "Cannot find valid path for binary mask, %s" % self.dir_M
            self.M_paths = sorted(make_dataset(self.dir_M, opt.max_dataset_size))
            assert len(self.M_paths) == 1, "SingleSkitDataset class should be used with one image for mask"
            self.M_img = ImageOps.grayscale(Image.open(self.M_paths[0]))
        if not os.path.exists(self.dir_I):   # an edited sketch: no ground-truth image / tactile data
            print("Warning: missing I data opt dataroot {}, opt subdir_I {}".format(opt.dataroot, opt.subdir_I))
            assert "edit" in opt.dataroot, "I and T data are required for original sketches"
            self.I_paths, self.I_img, self.T_paths, self.T_size = [], None, [], 0
        else:
            assert os.path.exists(self.dir_T), "datasets directories are invalid, \n dir_I {} \n dir_T {}".format(self.dir_I, self.dir_T)
            self.I_paths = sorted(make_dataset(self.dir_I, opt.max_dataset_size))
            assert len(self.I_paths) == 1, "SingleSkitDataset class should be used with one image in sketch and visual image"
            assert opt.image_nc == 3, "Visual image should have RGB 3 channels"
            self.I_img = Image.open(self.I_paths[0]).convert("RGB")
            self.T_paths = make_touch_image_dataset(self.dir_T, opt.max_dataset_size)
            self.T_size = len(self.T_paths)
        if opt.subdir_valT is not None:
            self.val_T_paths = make_touch_image_dataset(self.dir_valT, opt.max_dataset_size)
            self.val_T_size = len(self.val_T_paths)
        else:
            self.val_T_paths, self.val_T_size = None, 0
        A_zoom = 1 / opt.random_scale_max if opt.is_train else 1
        zoom = np.random.uniform(A_zoom, 1.0, size=(len(self) // opt.batch_size + 1, 1, 2))
        self.zoom_levels_A = np.reshape(np.tile(zoom, (1, opt.batch_size, 1)), [-1, 2])
        self.preprocess_data()

    # ------------------------------------------------------------------------------------------------------------------
    def preprocess_data(self):
        """the cache build of the reference (singleskit_dataset.py:194-432): one entry per augmentation index"""
        opt = self.opt
        print("Preprocess data for singleskit_dataset and save them in cache, len %d..." % len(self))
        t0 = time.time()
        if "padded" in opt.dataroot:
            self.padded_size = int(opt.dataroot.split("padded_")[1].split("/")[0].split("_")[0])
        method = Image.LANCZOS
        for index in range(len(self)):
            S_img, I_img, M_img = self.S_img, self.I_img, self.M_img
            if "zoom" in opt.preprocess:
                sfh, sfw = self.zoom_levels_A[0]
                S1 = zoom_img(S_img, sfh, sfw, method)
                I1 = zoom_img(I_img, sfh, sfw, method) if I_img is not None else None
                M1 = zoom_img(M_img, sfh, sfw, method) if M_img is not None else None
            else:
                S1, I1, M1, sfh, sfw = S_img, I_img, M_img, 1, 1
            H, W = S_img.size[:2]
            ch = cw = opt.crop_size
            S2, resize_ratio, cpx, cpy = crop_img(S1, ch, cw, method, None, None, None, opt.center_w, opt.center_h,
                                                  center_crop="crop" not in opt.preprocess)
            I2 = crop_img(I1, ch, cw, method, resize_ratio, cpx, cpy)[0] if I_img is not None else None
            M2 = crop_img(M1, ch, cw, method, resize_ratio, cpx, cpy)[0] if M_img is not None else None
            S3, rrw, rrh = make_power_2_img(S2, 256, method)
            I3 = M3 = None
            if I_img is not None:
                I3, rrw, rrh = make_power_2_img(I2, 256, method)
            if M_img is not None:
                M3, rrw, rrh = make_power_2_img(M2, 256, method)
            S_tensor = normalize_half(to_tensor(S3))
            I_tensor = normalize_half(to_tensor(I3)) if I_img is not None else None
            M_tensor = to_tensor(M3) if M_img is not None else None
            aug = {"H": H, "W": W, "scale_factor_h": sfh, "scale_factor_w": sfw, "crop_size_h": ch, "crop_size_w": cw, "resize_ratio": resize_ratio,
                   "crop_pos_x": cpx, "crop_pos_y": cpy, "resize_ratio_w": rrw, "resize_ratio_h": rrh, "patch_crop_size": 32}
            T_images, T_coords, full_T_coords, I_masks = [], [], [], []
            if self.T_size > 0:
                T_images, T_coords, full_T_coords, I_masks = self.find_validate_touch_patches_and_coords(
                    self.T_size, self.T_paths, aug, S3, M3, is_train=opt.is_train, is_val=False)
            val_T_images, val_T_coords, val_full_T_coords, val_I_masks = [], [], [], []
            if self.val_T_size > 0:
                val_T_images, val_T_coords, val_full_T_coords, val_I_masks = self.find_validate_touch_patches_and_coords(
                    self.val_T_size, self.val_T_paths, aug, S3, M3, is_train=opt.is_train, is_val=True)
            name = os.path.splitext(ntpath.basename(self.S_paths[0][0]))[0]   # sic: the first CHARACTER of the path
            if I_img is not None:
                d = {"S": S_tensor, "I": I_tensor, "name": name, "I_masks": I_masks, "val_I_masks": val_I_masks, "T_images": T_images,
                     "T_coords": T_coords, "S_paths": self.S_paths[0], "augmentation_params": aug, "full_T_coords": full_T_coords,
                     "val_T_images": val_T_images, "val_T_coords": val_T_coords, "val_full_T_coords": val_full_T_coords}
            else:
                d = {"S": S_tensor, "name": name, "S_paths": self.S_paths[0], "T_images": [], "augmentation_params": aug}
            if M_img is not None:
                d.update({"M": M_tensor, "M_paths": self.M_paths[0]})
            if tune.get("VTS_U8_BATCH", "1") != "0":      # (not a reference key: see to_u8)
                for key, pic in (("S", S3), ("I", I3), ("M", M3)):
                    raw = to_u8(pic) if key in d else None
                    if raw is not None and tuple(raw.shape) == tuple(d[key].shape):
                        d[key + "_u8"] = raw
            self.data_dict[index] = d
        print("Finish preprocessing %d data, takes " % len(self), time.time() - t0)

    # ------------------------------------------------------------------------------------------------------------------
    def find_validate_touch_patches_and_coords(self, T_size, T_paths, aug, S3, M3, is_train=False, is_val=False, I3=None, compute_SIM_patches=False):
        """GelSight rectangles through the augmentation (singleskit_dataset.py:434-658) -> (T_images, T_coords, full_T_coords, I_masks);
        compute_SIM_patches (PatchSkitDataset, :1024-1036): also the sketch / image / mask patches under every tactile square -> 7 values"""
        opt = self.opt
        valid_idx, roi1, roi2, roi3 = [], [], [], []
        for i in range(int(T_size)):
            _, _, x, y, h, w, _, _ = touch_data_loader(T_paths[i], convert2im=False, return_mask=True)
            if "padded" in opt.dataroot:
                x, y, h, w = global_padding_find_coords(x, y, h, w, padded_size=self.padded_size, org_h=opt.center_h, org_w=opt.center_w)
            x1, y1, h1, w1 = zoom_find_coords(x, y, h, w, aug["scale_factor_h"], aug["scale_factor_w"])
            valid, x2, y2, h2, w2 = crop_find_coords(x1, y1, h1, w1, aug["crop_size_h"], aug["crop_size_w"], aug["resize_ratio"],
                                                     aug["crop_pos_x"], aug["crop_pos_y"])
            x3, y3, h3, w3 = make_power_2_find_coords(x2, y2, h2, w2, aug["resize_ratio_w"], aug["resize_ratio_h"])
            if valid:
                valid_idx.append(i)
                roi3.append([int(round(x3)), int(round(y3)), int(round(h3)), int(round(w3))])
                roi1.append([int(round(x1)), int(round(y1)), int(round(h1)), int(round(w1))])
                roi2.append([int(round(x2)), int(round(y2)), int(round(h2)), int(round(w2))])
        calc_weight = bool(getattr(opt, "w_resampling", False))
        all_T, all_C, all_K, weights, roi3_update, all_S, all_I, all_M = self.process_all_valid_patches(
            valid_idx, roi3, T_paths, aug, S3, M3, calc_weight, is_train, I3=I3, compute_SIM_patches=compute_SIM_patches)
        total = len(all_T)
        bs = min(opt.batch_size_G2, total) if getattr(opt, "batch_size_G2", 0) > 0 else total
        bs_val = min(opt.batch_size_G2_val, total) if getattr(opt, "batch_size_G2_val", 0) > 0 else total
        if is_train:
            if not is_val:
                if getattr(opt, "w_resampling", False):
                    sel = random.choices(range(len(all_C)), weights=weights, k=bs)
                else:
                    sel = random.sample(range(len(all_C)), bs)
            else:
                sel = random.sample(range(len(all_C)), bs_val)
        else:
            print("test set, select all patches")
            sel = range(len(all_C))
        if compute_SIM_patches:
            return all_T[sel], all_C[sel], roi3_update, all_K[sel], all_S[sel], all_I[sel], all_M[sel]
        return all_T[sel], all_C[sel], roi3_update, all_K[sel]

    def process_all_valid_patches(self, valid_idx, roi3, T_paths, aug, S3, M3, calc_weight, is_train, I3=None, compute_SIM_patches=False):
        """32 x 32 squares of every valid GelSight rectangle (singleskit_dataset.py:660-1128, contact-mask method)"""
        opt = self.opt
        mult = opt.T_resolution_multiplier
        size_t = aug["patch_crop_size"] * mult
        half = size_t // 2
        T_images, T_coords, I_masks, weights, roi3_update = [], [], [], [], []
        M3_arr = np.array(M3)
        for i in range(len(valid_idx)):
            patch_index = valid_idx[i]
            nx, ny, nh, nw = roi3[patch_index]                          # sic: indexed by the tactile file index
            if np.sum(M3_arr[ny:ny + nh, nx:nx + nw]) == 0:             # no pixel of the rectangle inside the object mask
                continue
            roi3_update.append(roi3[patch_index])
            path = T_paths[valid_idx[patch_index]]                      # sic
            gx, gy, _, _, _, _, touch_mask, center_mask = touch_data_loader(path, convert2im=False, return_mask=True)
            ys, xs = np.where(center_mask > 0)
            cxs, cys, masks = [], [], []
            for cx, cy in zip(xs, ys):
                sq = touch_mask[cy - half:cy + half, cx - half:cx + half]
                px, py = int((cx - half) / mult), int((cy - half) / mult)
                ox, oy = np.round((nx + px) * mult), np.round((ny + py) * mult)
                cut = np.round(aug["patch_crop_size"] * mult)
                M_patch = np.array(M3.crop((ox, oy, ox + cut, oy + cut)))   # 0 .. 255
                sq = sq * M_patch / 255
                if np.max(sq) >= 1:
                    cxs.append(cx), cys.append(cy), masks.append(sq)
            num = min(len(cxs), opt.sample_bbox_per_patch)
            chosen = random.sample(range(len(cxs)), num) if is_train else np.arange(len(cxs) // 2, len(cxs) // 2 + num)
            for k in chosen:
                cx, cy = cxs[k], cys[k]
                gxs, gys = gx[cy - half:cy + half, cx - half:cx + half], gy[cy - half:cy + half, cx - half:cx + half]
                gxy = torch.cat((to_tensor(gxs), to_tensor(gys)), 0)
                assert gxy.shape == (2, size_t, size_t), "gxy shape %s, center_x %d, center_y %d" % (str(gxy.shape), cx, cy)
                T_images.append(gxy)
                T_coords.append([nx, ny, nh, nw, aug["patch_crop_size"], 1, int((cx - half) / mult), int((cy - half) / mult)])
                I_masks.append(masks[k])
        S_images, I_images, M_images = [], [], []
        if calc_weight or compute_SIM_patches:
            for nx, ny, nh, nw, pcs, rr, px, py in T_coords:
                ox, oy = np.round((nx + px / rr) * mult), np.round((ny + py / rr) * mult)
                cut = np.round(pcs / rr * mult)
                S_patch = np.array(S3.crop((ox, oy, ox + cut, oy + cut)))
                if compute_SIM_patches:      # S_tf / I_tf / M_tf of the caller: ToTensor (+ Normalize 0.5 / 0.5 for S and I), :1024-1033
                    S_images.append(normalize_half(to_tensor(S_patch)))
                    I_images.append(normalize_half(to_tensor(np.array(I3.crop((ox, oy, ox + cut, oy + cut))))))
                    M_images.append(to_tensor(np.array(M3.crop((ox, oy, ox + cut, oy + cut)))))
                if calc_weight:
                    v = variance_of_laplacian(S_patch, ref=np.ones_like(S_patch) * 255)   # the sketch's reference level is white
                    weights.append(min(max(opt.resampling_w_min, v), opt.resampling_w_max))
        if len(T_images) > 1:
            T_images, T_coords = torch.stack(T_images, dim=0), np.stack(T_coords, axis=0)
            I_masks = torch.from_numpy(np.array(I_masks))
        elif len(T_images) == 1:
            T_images, T_coords = torch.unsqueeze(T_images[0], 0), np.array(T_coords)
            I_masks = torch.unsqueeze(torch.from_numpy(I_masks[0]), 0)
        if compute_SIM_patches:
            assert len(S_images) == len(I_images) == len(M_images) and len(S_images) > 1, "S_images, I_images, M_images should have the same length and > 1"
            S_images, I_images, M_images = torch.stack(S_images, dim=0), torch.stack(I_images, dim=0), torch.stack(M_images, dim=0)
        else:
            S_images = I_images = M_images = None
        weights = np.array(weights) if calc_weight else None
        if calc_weight:
            assert len(weights) == len(T_coords), "weights and T_coords should have the same length"
        return T_images, T_coords, I_masks, weights, roi3_update, S_images, I_images, M_images

    def __getitem__(self, index):
        assert index in self.data_dict.keys(), "Cannot find index %d in dataset" % index
        return self.data_dict[index]

    def __len__(self):
        return self.data_len
