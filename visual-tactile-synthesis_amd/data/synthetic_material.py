"""Write a seeded synthetic material in the TouchClothing on-disk format that data/singleskit_dataset.py reads (PNG sketch / image /
mask + `*_tactile.npz` touch records): lets the `--dataset_mode singleskit` path be exercised offline, and is the input of the
dataset parity fixture (oracle/make_dataset_golden.py runs the reference's SingleSkitDataset on the same folder).

    python -m data.synthetic_material /tmp/material [seed]"""
import os
import sys

import numpy as np
from PIL import Image


def write_material(root, seed=0, width=400, height=360, n_train=12, n_val=5, phase="train"):
    g = np.random.default_rng(seed)
    sub = {"train": ("trainS", "trainI", "trainM", "trainT", "valT"), "test": ("testS", "testI", "testM", "testT", None)}[phase]
    for d in sub:
        if d:
            os.makedirs(os.path.join(root, d), exist_ok=True)
    yy, xx = np.mgrid[0:height, 0:width]
    S = np.full((height, width), 255, np.uint8)
    for _ in range(40):   # strokes
        x0, y0 = int(g.integers(20, width - 20)), int(g.integers(20, height - 20))
        ln, horiz = int(g.integers(20, 120)), bool(g.integers(0, 2))
        if horiz:
            S[y0:y0 + 2, x0:min(width, x0 + ln)] = 0
        else:
            S[y0:min(height, y0 + ln), x0:x0 + 2] = 0
    I = np.clip(g.normal(128, 40, (height, width, 3)) + 40 * np.sin(xx / 17.0)[..., None] + 30 * np.cos(yy / 11.0)[..., None], 0, 255).astype(np.uint8)
    M = ((((yy - height / 2) / (0.46 * height)) ** 2 + ((xx - width / 2) / (0.45 * width)) ** 2) <= 1.0).astype(np.uint8) * 255
    Image.fromarray(S, "L").save(os.path.join(root, sub[0], "material.png"))
    Image.fromarray(I, "RGB").save(os.path.join(root, sub[1], "material.png"))
    Image.fromarray(M, "L").save(os.path.join(root, sub[2], "material.png"))

    def touch(dirname, count, tag):
        th, tw = 72, 88                              # one GelSight frame
        for k in range(count):
            gx = np.clip(g.normal(0.0, 0.05, (th, tw)), -0.3, 0.3).astype(np.float32)
            gy = np.clip(g.normal(0.02, 0.05, (th, tw)), -0.3, 0.3).astype(np.float32)
            contact = np.zeros((th, tw), np.uint8)
            contact[4:th - 4, 6:tw - 6] = 255          # 0 / 255 convention (the loader normalises)
            centre = np.zeros((th, tw), np.uint8)
            for _ in range(7):
                centre[int(g.integers(18, th - 18)), int(g.integers(18, tw - 18))] = 255
            np.savez(os.path.join(root, dirname, "material_%s%02d_tactile.npz" % (tag, k)), gx_raw=gx, gy_raw=gy,
                     vision_mask_x=np.int64(g.integers(105, 190)), vision_mask_y=np.int64(g.integers(105, 180)),
                     vision_mask_h=np.int64(48), vision_mask_w=np.int64(56), touch_thresh=contact, touch_center_thresh=centre)

    touch(sub[3], n_train, "t")
    if sub[4]:
        touch(sub[4], n_val, "v")
    return root


if __name__ == "__main__":
    print(write_material(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0))
