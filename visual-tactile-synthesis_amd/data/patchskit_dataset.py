"""PatchSkitDataset: the paired-patch front-end of the baselines (pix2pixHD) -- reference data/patchskit_dataset.py:20-340.

Built on SingleSkitDataset (same folders, same tactile records, same augmentation chain and random-number consumption); it differs in
what it caches (reference :57-340, one cache build instead of one per augmentation index):
    return_patch True  (training)  every selected tactile square with the 32 x 32 sketch / image / mask patches under it:
                                   S_images [P,1,32,32], I_images [P,3,32,32], M_images [P,1,32,32], T_images [P,2,32,32], I_masks [P,32,32];
                                   an item is ONE paired patch, len(dataset) = P
    return_patch False (testing)   the whole crop: S [1,1,H,W], I, M, T_images [1,P,2,32,32], T_coords, full_T_coords, I_masks [1,P,1,32,32]
Quirks kept: `name` from the first CHARACTER of the sketch path (:263-264), H, W = S_img.size[:2] = (width, height) (:110), and the
separate-validation-set branch is never taken (preprocess_data is called without separate_val_set, base __init__)."""
import ntpath
import os
import time

import torch
from PIL import Image

from .singleskit_dataset import SingleSkitDataset, crop_img, make_power_2_img, normalize_half, to_tensor, zoom_img


class PatchSkitDataset(SingleSkitDataset):
    def __init__(self, opt, verbose=False, default_len=1000, return_patch=True):
        self.return_patch = opt.return_patch if hasattr(opt, "return_patch") else return_patch
        SingleSkitDataset.__init__(self, opt)

    def preprocess_data(self):
        opt = self.opt
        print("Preprocess data for patchskit_dataset and save them in cache, len %d..." % len(self))
        t0 = time.time()
        if "padded" in opt.dataroot:
            self.padded_size = int(opt.dataroot.split("padded_")[1].split("/")[0].split("_")[0])
        S_img, I_img = self.S_img, self.I_img
        M_img = self.M_img if opt.use_bg_mask else None
        method = Image.LANCZOS
        if "zoom" in opt.preprocess:
            sfh, sfw = self.zoom_levels_A[0]
            S1 = zoom_img(S_img, sfh, sfw, method)
            I1 = zoom_img(I_img, sfh, sfw, method) if I_img is not None else None
            M1 = zoom_img(M_img, sfh, sfw, method) if M_img is not None else None
        else:
            S1, I1, M1, sfh, sfw = S_img, I_img, M_img, 1, 1
        H, W = S_img.size[:2]
        ch = cw = opt.crop_size
        S2, resize_ratio, cpx, cpy = crop_img(S1, ch, cw, method, None, None, None, opt.center_w, opt.center_h, center_crop="crop" not in opt.preprocess)
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
        if self.T_size > 0:
            T_images, T_coords, full_T_coords, I_masks, S_images, I_images, M_images = self.find_validate_touch_patches_and_coords(
                self.T_size, self.T_paths, aug, S3, M3, is_train=opt.is_train, is_val=False, I3=I3, compute_SIM_patches=True)
        name = os.path.splitext(ntpath.basename(self.S_paths[0][0]))[0]      # sic: the first CHARACTER of the path
        if self.return_patch:
            n = len(S_images)
            self.data_dict = {"S_images": S_images, "name": [name for _ in range(n)], "S_paths": [self.S_paths[0] for _ in range(n)],
                              "augmentation_params": [aug for _ in range(n)]}
            if I_img is not None:
                self.data_dict.update({"I_images": I_images, "T_images": T_images, "I_masks": I_masks})
            if M_img is not None:
                self.data_dict.update({"M_images": M_images})
            self.data_len = len(self.data_dict["S_images"])
        else:
            self.data_dict = {"S": torch.unsqueeze(S_tensor, 0), "name": [name], "S_paths": [self.S_paths[0]], "augmentation_params": [aug]}
            if I_img is not None:
                self.data_dict.update({"I": torch.unsqueeze(I_tensor, 0), "T_images": torch.unsqueeze(T_images, 0), "T_coords": [T_coords],
                                       "full_T_coords": [full_T_coords], "I_masks": torch.unsqueeze(torch.unsqueeze(I_masks, 0), -2)})
            if M_img is not None:
                self.data_dict.update({"M": torch.unsqueeze(M_tensor, 0)})
            self.data_len = len(self.data_dict["S"])
        print("Finish preprocessing %d data, takes " % len(self), time.time() - t0)

    def __getitem__(self, index):
        return {k: v[index] for k, v in self.data_dict.items()}

    def __len__(self):
        return self.data_len
