"""Seeded synthetic sketch -> (RGB, tactile) samples with the reference's batch contract.

Contract followed (post-collate shapes, N = batch size, NT = patches per image):
/root/reference/data/singleskit_dataset.py:332-345,406-430 and SURVEY.md §8(b)/(d):
  S [N,1,H,W] in [-1,1]; I [N,3,H,W] in [-1,1]; M [N,1,H,W] in {0,1};
  T_images [N,NT,2,32,32] raw surface gradients; I_masks [N,NT,32,32] float64;
  T_coords [N,NT,8] float64 = (ROI_x, ROI_y, ROI_h, ROI_w, 32, resize_ratio, crop_x, crop_y);
  val_* with NT_val patches; augmentation_params = identity crop.
"""
import numpy as np
import torch
import torch.utils.data

from util import util


def _box_blur(a, k):
    """Separable k x k mean filter with edge replication (numpy, host side)."""
    r = k // 2
    for axis in (-2, -1):
        pad = [(0, 0)] * a.ndim
        pad[axis] = (r, r)
        ap = np.pad(a, pad, mode="edge")
        acc = np.zeros_like(a)
        for d in range(k):
            sl = [slice(None)] * a.ndim
            sl[axis] = slice(d, d + a.shape[axis])
            acc += ap[tuple(sl)]
        a = acc / k
    return a


def make_sample(size, nt, nt_val, seed, patch=32, style_dim=0, quantize8=False):
    """One un-collated sample dict.  Deterministic in (size, nt, nt_val, seed).
    quantize8: sketch / image / mask as a real material delivers them -- 8-bit PNG pixels through ToTensor [+ Normalize(0.5, 0.5)] -- with
    the optional S_u8 / I_u8 / M_u8 keys of the dataset front-ends next to the float tensors."""
    g = np.random.default_rng(seed)
    H = W = int(size)
    S = np.where(g.random((1, H, W)) > 0.9, -1.0, 1.0)
    S = _box_blur(S, 3).astype(np.float32)
    I = _box_blur(g.uniform(-1.0, 1.0, (3, H, W)), 5).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    cy, cx = (H - 1) / 2.0, (W - 1) / 2.0
    ay, ax = 0.4 * H, 0.3 * W  # semi-axes: ellipse axes 0.8H x 0.6W
    M = ((((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2) <= 1.0).astype(np.float32)[None]

    def patches(n):
        T = np.clip(g.normal(0.0, 0.05, (n, 2, patch, patch)), -0.3, 0.3).astype(np.float32)
        margin = 48
        x_lo, x_hi = int(cx - ax) + margin, max(int(cx - ax) + margin + 1, int(cx + ax) - margin - 40)
        y_lo, y_hi = int(cy - ay) + margin, max(int(cy - ay) + margin + 1, int(cy + ay) - margin - 40)
        x_lo, y_lo = max(0, min(x_lo, W - 41)), max(0, min(y_lo, H - 41))
        x_hi, y_hi = max(x_lo + 1, min(x_hi, W - 40)), max(y_lo + 1, min(y_hi, H - 40))
        nbase = 40
        base = np.stack([g.integers(x_lo, x_hi, nbase), g.integers(y_lo, y_hi, nbase)], 1)
        pick = g.integers(0, nbase, n)  # with replacement: overlapping patches included
        coords = np.zeros((n, 8), np.float64)
        coords[:, 0:2] = base[pick]
        coords[:, 2:4] = 40
        coords[:, 4] = patch
        coords[:, 5] = 1.0
        coords[:, 6:8] = g.integers(0, 8, (n, 2))
        masks = np.ones((n, patch, patch), np.float64)
        return T, coords, masks

    if quantize8:
        u8 = {"S_u8": torch.from_numpy(np.round((S + 1.0) * 127.5).astype(np.uint8)), "I_u8": torch.from_numpy(np.round((I + 1.0) * 127.5).astype(np.uint8)),
              "M_u8": torch.from_numpy((M * 255).astype(np.uint8))}
        S = ((u8["S_u8"].to(torch.float32).div(255) - 0.5) / 0.5).numpy()
        I = ((u8["I_u8"].to(torch.float32).div(255) - 0.5) / 0.5).numpy()
        M = u8["M_u8"].to(torch.float32).div(255).numpy()
    T, C, K = patches(nt)
    vT, vC, vK = patches(nt_val)
    aug = {
        "H": H, "W": W, "scale_factor_h": 1.0, "scale_factor_w": 1.0,
        "crop_size_h": H, "crop_size_w": W, "resize_ratio": 1.0,
        "crop_pos_x": 0, "crop_pos_y": 0, "resize_ratio_w": 1.0, "resize_ratio_h": 1.0,
        "patch_crop_size": patch,
    }
    extra = {}
    if style_dim:
        sc = np.random.default_rng(seed + 7919).normal(0.0, 1.0, style_dim)
        extra["style_code"] = torch.from_numpy((sc / np.linalg.norm(sc)).astype(np.float32))
    if quantize8:
        extra.update(u8)
    return {
        **extra,
        "S": torch.from_numpy(S), "I": torch.from_numpy(I), "M": torch.from_numpy(M),
        "name": "synthetic_%d" % seed, "S_paths": "synthetic/%d.png" % seed, "M_paths": "synthetic/%d_mask.png" % seed,
        "T_images": T, "T_coords": C, "I_masks": K, "full_T_coords": C.copy(),
        "val_T_images": vT, "val_T_coords": vC, "val_I_masks": vK, "val_full_T_coords": vC.copy(),
        "augmentation_params": aug,
    }


def make_patch_sample(seed, patch=32, height=None, width=None):
    """One un-collated patch sample with the patchskit contract used by pix2pixHD's patch-wise training
    (/root/reference/data/patchskit_dataset.py:277-333, `return_patch=True`): aligned 32x32 sketch / mask /
    image / tactile patches (height / width: a rectangular whole image instead, BASELINE config 3's 2048 x 1024)."""
    g = np.random.default_rng(seed)
    ph, pw = int(height or patch), int(width or patch)
    S = _box_blur(np.where(g.random((1, ph, pw)) > 0.9, -1.0, 1.0), 3).astype(np.float32)
    I = _box_blur(g.uniform(-1.0, 1.0, (3, ph, pw)), 5).astype(np.float32)
    yy, xx = np.mgrid[0:ph, 0:pw]
    M = (((yy - ph / 2) / (0.45 * ph)) ** 2 + ((xx - pw / 2) / (0.4 * pw)) ** 2 <= 1.0).astype(np.float32)[None]
    T = np.clip(g.normal(0.0, 0.05, (2, ph, pw)), -0.3, 0.3).astype(np.float32)
    return {"S_images": torch.from_numpy(S), "M_images": torch.from_numpy(M), "I_images": torch.from_numpy(I),
            "T_images": torch.from_numpy(T), "I_masks": np.ones((ph, pw), np.float64), "name": "synthetic_%d" % seed,
            "S_paths": "synthetic/%d.png" % seed, "augmentation_params": {"patch_crop_size": patch}}


class SyntheticDataset(torch.utils.data.Dataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--synthetic_size", type=int, default=0,
                            help="H=W of synthetic samples (0: use crop_size)")
        parser.add_argument("--data_seed", type=int, default=1234, help="base seed of the synthetic generator")
        parser.add_argument("--cache_samples", type=util.str2bool, default=True,
                            help="keep generated samples in host RAM (reference caches its dataset too)")
        return parser

    def __init__(self, opt):
        self.opt = opt
        self.size = opt.synthetic_size if getattr(opt, "synthetic_size", 0) else opt.crop_size
        self.length = max(1, int(getattr(opt, "data_len", 1)))
        self.nt = int(getattr(opt, "batch_size_G2", 64))
        self.nt_val = int(getattr(opt, "batch_size_G2_val", self.nt)) if opt.isTrain else self.nt
        self.rank = int(getattr(opt, "rank", 0))
        self.style_dim = int(getattr(opt, "style_code_dim", 0)) if getattr(opt, "use_style_code", False) else 0
        self.current_epoch = 0
        self._cache = {}

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        seed = self.opt.data_seed + 100003 * self.rank + index
        if getattr(self.opt, "model", "") == "pix2pixHD" and getattr(self.opt, "return_patch", False):
            return make_patch_sample(seed)
        if getattr(self.opt, "cache_samples", True):
            if index not in self._cache:
                self._cache[index] = make_sample(self.size, self.nt, self.nt_val, seed, style_dim=self.style_dim)
            return self._cache[index]
        return make_sample(self.size, self.nt, self.nt_val, seed, style_dim=self.style_dim)
