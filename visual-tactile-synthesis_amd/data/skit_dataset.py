"""SkitDataset: the multi-material front-end of the skitG model (`--dataset_mode skit --material_list A B ...`).

What the reference class does (data/skit_dataset.py:86-500), kept:
    * one material folder per entry of `material_list`, found at the reference's fixed place relative to the working directory:
      ./datasets/singleskit_<material>_padded_<padded_size>_x<T_resolution_multiplier>/{trainS,trainI,trainM,trainT,valT} (:147-170);
      one sketch / image / mask per material, the tactile file lists kept per material;
    * cache entry `index` belongs to material `index % len(material_list)` (:243) and, unlike singleskit (which zooms every entry by
      zoom level 0), takes zoom level `index` (:287);
    * `"padded" in opt.dataroot` -- the OPTION, not the material's folder -- still decides whether the GelSight rectangles are shifted
      into the padded frame (:230-231, and inside find_validate_touch_patches_and_coords);
    * entry keys as singleskit; `S_paths` is ALWAYS the first material's sketch path (:473), `name` the entry's own sketch file stem.
Not kept:
    * `opt.load_contact_mask` (:423): no parser of the reference defines it, so the published class raises AttributeError for any
      material with tactile data; the parent's default (True: contact masks are loaded) is what the golden vectors were generated with
      (oracle/make_dataset_golden.py sets the attribute) and what this class does;
Added: a precomputed style code per material (<material folder>/style_code.npy) travels as the batch key `style_code` (the reference
encodes the visual image with CLIP inside the model, skitG_model.py:484-489; CLIP's weights cannot exist offline).
Not kept (continued):
    * `use_external_test_input` (:113-141: a sketch of one material with the style IMAGE of another, for the CLIP style encoder): the style
      code is an input of this package (no CLIP weights offline), so the style image has no consumer -- raises.
"""
import ntpath
import os
from vts import tune
import time

import numpy as np
import torch
from PIL import Image, ImageOps

from .singleskit_dataset import (SingleSkitDataset, crop_img, make_dataset, make_power_2_img, make_touch_image_dataset, normalize_half, to_tensor,
                                 to_u8, zoom_img)


class SkitDataset(SingleSkitDataset):
    def __init__(self, opt, verbose=False, default_len=1000):
        self.opt = opt
        self.root = opt.dataroot
        self.current_epoch = 0
        self.verbose = verbose
        self.data_dict = {}
        self.data_len = opt.data_len if hasattr(opt, "data_len") else default_len
        self.is_train = opt.is_train
        if getattr(opt, "use_external_test_input", False):
            raise NotImplementedError("--use_external_test_input: the style IMAGE of a second material feeds the CLIP style encoder, which is "
                                      "not built (the style code is a batch input: key `style_code`)")
        materials = list(getattr(opt, "material_list", []))
        if not materials:
            raise ValueError("--dataset_mode skit needs --material_list")
        print("material_list is {}".format(materials))
        self.S_paths, self.I_paths, self.M_paths = [], [], []
        self.T_paths, self.T_sizes, self.val_T_paths, self.val_T_sizes, self.style_codes = [], [], [], [], []
        for material in materials:
            dataroot = "./datasets/singleskit_%s_padded_%s_x%s/" % (material, opt.padded_size, opt.T_resolution_multiplier)
            dir_S, dir_I = os.path.join(dataroot, opt.subdir_S), os.path.join(dataroot, opt.subdir_I)
            dir_T, dir_M = os.path.join(dataroot, opt.subdir_T), os.path.join(dataroot, opt.subdir_M)
            dir_valT = os.path.join(dataroot, opt.subdir_valT) if opt.subdir_valT is not None else None
            assert os.path.exists(dir_S) and os.path.exists(dir_I) and os.path.exists(dir_T) and os.path.exists(dir_M), \
                "datasets directories are invalid, \n dir_S {} \n dir_I {} \n dir_T {} \n dir_M {}".format(dir_S, dir_I, dir_T, dir_M)
            self.S_paths.extend(sorted(make_dataset(dir_S, opt.max_dataset_size)))
            self.I_paths.extend(sorted(make_dataset(dir_I, opt.max_dataset_size)))
            self.M_paths.extend(sorted(make_dataset(dir_M, opt.max_dataset_size)))
            t = make_touch_image_dataset(dir_T, opt.max_dataset_size)
            self.T_paths.append(t)
            self.T_sizes.append(len(t))
            v = make_touch_image_dataset(dir_valT, opt.max_dataset_size) if dir_valT is not None else []
            self.val_T_paths.append(v)
            self.val_T_sizes.append(len(v))
            # (not in the reference, which encodes the visual image with CLIP inside the model: a PRECOMPUTED style code of the material,
            #  <material folder>/style_code.npy [style_code_dim], becomes the batch key `style_code` the skitG model of this package reads)
            sc = os.path.join(dataroot, "style_code.npy")
            self.style_codes.append(np.load(sc).astype(np.float32).reshape(-1) if os.path.exists(sc) else None)
        # all or none: a batch collated from several materials needs the same keys in every entry, and skitG reads `style_code` of each
        have = [c is not None for c in self.style_codes]
        if any(have) and not all(have):
            raise FileNotFoundError("style_code.npy is present for some materials but missing for: %s (expected <dataset folder>/style_code.npy "
                                    "for every material of --material_list, or for none)" % [m for m, h in zip(materials, have) if not h])
        dim = getattr(opt, "style_code_dim", None)
        for m, c in zip(materials, self.style_codes):
            if c is not None and dim is not None and c.shape[0] != int(dim):
                raise ValueError("style_code.npy of material %s holds %d values, --style_code_dim is %d" % (m, c.shape[0], int(dim)))
        if opt.sketch_nc == 1:
            self.S_imgs = [ImageOps.grayscale(Image.open(p)) for p in self.S_paths]
        else:
            assert opt.sketch_nc == 3, "Load sketch either in grayscale or RGB"
            self.S_imgs = [Image.open(p).convert("RGB") for p in self.S_paths]
        assert opt.image_nc == 3, "Visual image should have RGB 3 channels"
        self.I_imgs = [Image.open(p).convert("RGB") for p in self.I_paths] if len(self.I_paths) > 0 else None
        self.M_imgs = [ImageOps.grayscale(Image.open(p)) for p in self.M_paths] if opt.use_bg_mask is True else None
        A_zoom = 1 / opt.random_scale_max if opt.is_train else 1
        zoom = np.random.uniform(A_zoom, 1.0, size=(len(self) // opt.batch_size + 1, 1, 2))
        self.zoom_levels_A = np.reshape(np.tile(zoom, (1, opt.batch_size, 1)), [-1, 2])
        self.preprocess_data()

    def preprocess_data(self):
        """the cache build (data/skit_dataset.py:211-500): singleskit's pipeline per entry, on the entry's material"""
        opt = self.opt
        nm = len(opt.material_list)
        print("Preprocess data for skit_dataset and save them in cache, len %d..." % len(self))
        t0 = time.time()
        if "padded" in opt.dataroot:
            self.padded_size = int(opt.dataroot.split("padded_")[1].split("/")[0].split("_")[0])
        method = Image.LANCZOS
        for index in range(len(self)):
            mi = index % nm
            S_path, S_img = self.S_paths[mi], self.S_imgs[mi]
            I_img = self.I_imgs[mi] if self.I_imgs is not None else None
            M_img = self.M_imgs[mi] if opt.use_bg_mask else None
            if "zoom" in opt.preprocess:
                sfh, sfw = self.zoom_levels_A[index]
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
            val_T_images, val_T_coords, val_full_T_coords, val_I_masks = [], [], [], []
            if I_img is not None:
                if self.T_sizes[mi] > 0:
                    T_images, T_coords, full_T_coords, I_masks = self.find_validate_touch_patches_and_coords(
                        self.T_sizes[mi], self.T_paths[mi], aug, S3, M3, is_train=opt.is_train, is_val=False)
                if self.val_T_sizes[mi] > 0:
                    val_T_images, val_T_coords, val_full_T_coords, val_I_masks = self.find_validate_touch_patches_and_coords(
                        self.val_T_sizes[mi], self.val_T_paths[mi], aug, S3, M3, is_train=opt.is_train, is_val=True)
            name = os.path.splitext(ntpath.basename(S_path))[0]
            if I_img is not None:
                d = {"S": S_tensor, "I": I_tensor, "name": name, "I_masks": I_masks, "val_I_masks": val_I_masks, "T_images": T_images,
                     "T_coords": T_coords, "S_paths": self.S_paths[0], "augmentation_params": aug, "full_T_coords": full_T_coords,
                     "val_T_images": val_T_images, "val_T_coords": val_T_coords, "val_full_T_coords": val_full_T_coords}
            else:
                d = {"S": S_tensor, "name": name, "S_paths": self.S_paths[0], "T_images": [], "augmentation_params": aug}
            if M_img is not None:
                d.update({"M": M_tensor, "M_paths": self.M_paths[mi]})
            if self.style_codes[mi] is not None:
                d["style_code"] = torch.from_numpy(self.style_codes[mi])
            if tune.get("VTS_U8_BATCH", "1") != "0":      # (not a reference key: singleskit_dataset.to_u8)
                for key, pic in (("S", S3), ("I", I3), ("M", M3)):
                    raw = to_u8(pic) if key in d else None
                    if raw is not None and tuple(raw.shape) == tuple(d[key].shape):
                        d[key + "_u8"] = raw
            self.data_dict[index] = d
        print("Finish preprocessing %d data, takes " % len(self), time.time() - t0)
