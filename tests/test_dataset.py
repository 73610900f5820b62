"""Dataset front-end against the REFERENCE's SingleSkitDataset: tests/golden/singleskit_dataset.npz holds what the
reference class cached for a seeded synthetic material (oracle/make_dataset_golden.py, build container); here the same material is
written again and data/singleskit_dataset.py must reproduce every array -- images, tactile squares, contact masks, coordinates,
augmentation parameters -- bit for bit under the same `random` / `numpy.random` seeds (train: random crop + sampled squares + sampled
batch; test: centre crop, middle squares, all patches)."""
import os
import random

import numpy as np
import pytest
import torch

from data.singleskit_dataset import SingleSkitDataset, touch_data_loader, variance_of_laplacian
from data.synthetic_material import write_material
from oracle.make_dataset_golden import dataset_opt


def _compare_with_reference_cache(ds, g, phase):
    """every array / scalar / string the reference class cached, bit for bit"""
    keys = [k for k in g.files if k.startswith(phase + "/") and k != phase + "/len"]
    assert keys
    seen = set()
    for index in range(len(ds)):
        item = ds[index]
        for k, v in item.items():
            key = "%s/%d/%s" % (phase, index, k)
            if k.endswith("_u8"):       # not a reference key: the bytes the float tensor was made of (uploaded instead of it, expanded on the device)
                f = v.to(torch.float32).div(255)
                assert v.dtype == torch.uint8 and torch.equal((f - 0.5) / 0.5 if k[0] in "SI" else f, item[k[:-3]]), key
                continue
            if torch.is_tensor(v) and k in ("S", "I", "M"):
                a = v.numpy().astype(np.float64)
                assert np.array_equal(v.numpy()[:, ::4, ::4], g[key + "/sub"]), key
                assert np.allclose(np.array(list(a.shape) + [a.sum(), (a * a).sum()]), g[key + "/shape_sum_sq"], rtol=1e-12), key
                seen.update((key + "/sub", key + "/shape_sum_sq"))
            elif torch.is_tensor(v) or isinstance(v, np.ndarray):
                ref = g[key]
                a = v.numpy() if torch.is_tensor(v) else v
                assert a.shape == ref.shape and a.dtype == ref.dtype and np.array_equal(a, ref), key
                seen.add(key)
            elif isinstance(v, dict):
                for kk, vv in v.items():
                    assert float(np.asarray(vv, dtype=np.float64)) == float(g[key + "/" + kk]), (key, kk)
                    seen.add(key + "/" + kk)
            elif isinstance(v, str):
                assert (os.path.basename(v) if "paths" in k else v) == str(g[key]), key
                seen.add(key)
            elif isinstance(v, list):
                assert np.array_equal(np.asarray(v, dtype=np.float64) if len(v) else np.zeros((0,)), g[key]), key
                seen.add(key)
    assert seen == set(keys), set(keys) ^ seen


@pytest.mark.parametrize("phase,mseed,seed", [("train", 11, 5), ("test", 12, 6)])
def test_singleskit_dataset_matches_reference_cache(phase, mseed, seed, tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "singleskit_dataset.npz"))
    root = write_material(str(tmp_path / ("mat_" + phase)), seed=mseed, phase=phase)
    random.seed(seed)
    np.random.seed(seed)
    ds = SingleSkitDataset(dataset_opt(root, phase))
    assert len(ds) == 2
    _compare_with_reference_cache(ds, g, phase)
    b = ds[0]
    if phase == "train":
        assert b["T_images"].shape == (8, 2, 32, 32) and b["T_coords"].shape == (8, 8) and b["I_masks"].shape == (8, 32, 32)
        assert b["I_masks"].dtype == torch.float64 and b["val_T_images"].shape[0] == 6 and b["name"] == ""      # the reference's name quirk


def test_collated_batch_feeds_the_training_step_contract(tmp_path):
    """default_collate of the cached items gives the post-collate batch dict of SURVEY.md 8b (what SinSKITGModel.set_input consumes)"""
    from torch.utils.data import default_collate

    root = write_material(str(tmp_path / "m"), seed=3)
    random.seed(1)
    np.random.seed(1)
    ds = SingleSkitDataset(dataset_opt(root, "train", w_resampling=True))
    b = default_collate([ds[0]])
    assert b["S"].shape == (1, 1, 256, 256) and b["I"].shape == (1, 3, 256, 256) and b["M"].shape == (1, 1, 256, 256)
    assert float(b["S"].min()) >= -1 and float(b["S"].max()) <= 1 and set(np.unique(b["M"].numpy())) <= {0.0, 1.0}
    assert b["T_images"].shape == (1, 8, 2, 32, 32) and b["T_coords"].shape == (1, 8, 8) and b["I_masks"].shape == (1, 8, 32, 32)
    c = b["T_coords"][0].numpy()
    assert (c[:, 4] == 32).all() and (c[:, 5] == 1).all() and (c[:, 0] + c[:, 6] + 32 <= 256).all() and (c[:, 1] + c[:, 7] + 32 <= 256).all()
    assert set(b["augmentation_params"]) >= {"H", "W", "crop_pos_x", "crop_pos_y", "resize_ratio_w", "patch_crop_size"}


def test_touch_npz_reader_and_laplacian_weight(tmp_path):
    root = write_material(str(tmp_path / "m"), seed=4)
    path = sorted(os.listdir(os.path.join(root, "trainT")))[0]
    gx, gy, x, y, h, w, tm, cm = touch_data_loader(os.path.join(root, "trainT", path), convert2im=False)
    assert gx.shape == gy.shape == tm.shape == cm.shape == (72, 88) and tm.max() == 1.0 and cm.max() == 1.0 and (int(h), int(w)) == (48, 56)
    im, _, *_ = touch_data_loader(os.path.join(root, "trainT", path), convert2im=True, return_mask=False)
    assert im.size == (88, 72) and im.mode == "L"
    # 4-neighbour Laplacian with reflect-101 borders on (patch - 255) evaluated in uint8 (wraps): a white patch has zero variance, a
    # single black pixel gives the variance of the kernel's impulse response
    white = np.full((32, 32), 255, np.uint8)
    assert variance_of_laplacian(white, ref=np.ones_like(white) * 255) == 0.0
    one = white.copy()
    one[10, 10] = 254                      # (254 - 255) wraps to 255
    lap = np.zeros((32, 32))
    lap[10, 10], lap[9, 10], lap[11, 10], lap[10, 9], lap[10, 11] = -4 * 255, 255, 255, 255, 255
    assert abs(variance_of_laplacian(one, ref=np.ones_like(one) * 255) - lap.var()) < 1e-9


@pytest.mark.parametrize("phase,mseed,seed", [("train", 21, 7), ("test", 22, 8)])
def test_patchskit_dataset_matches_reference_cache(phase, mseed, seed, tmp_path, golden_dir):
    """data/patchskit_dataset.py against what the REFERENCE's PatchSkitDataset cached for the same seeded material: training items are
    single paired (sketch, image, mask, tactile) patches, the test item is the whole crop with all tactile squares"""
    from data.patchskit_dataset import PatchSkitDataset

    g = np.load(os.path.join(golden_dir, "patchskit_dataset.npz"))
    root = write_material(str(tmp_path / ("mat_" + phase)), seed=mseed, phase=phase)
    random.seed(seed)
    np.random.seed(seed)
    ds = PatchSkitDataset(dataset_opt(root, phase, return_patch=phase == "train"))
    assert len(ds) == int(g[phase + "/len"])
    _compare_with_reference_cache(ds, g, phase)
    if phase == "train":
        from torch.utils.data import default_collate
        b = default_collate([ds[i] for i in range(4)])      # the pix2pixHD step's batch (models/pix2pixHD_model.py:set_input)
        assert b["S_images"].shape == (4, 1, 32, 32) and b["I_images"].shape == (4, 3, 32, 32) and b["M_images"].shape == (4, 1, 32, 32)
        assert b["T_images"].shape == (4, 2, 32, 32) and b["I_masks"].shape == (4, 32, 32)
    else:
        assert ds[0]["S"].shape == (1, 256, 256) and ds[0]["T_images"].ndim == 4 and ds[0]["I_masks"].shape[1:] == (32, 1, 32)      # sic: the reference unsqueezes axis -2 (patchskit_dataset.py:313, its own TODO)


@pytest.mark.parametrize("phase,mseeds,seed,pre", [("train", (31, 32), 9, "zoom_crop"), ("test", (33, 34), 10, "none")])
def test_skit_dataset_matches_reference_cache(phase, mseeds, seed, pre, tmp_path, golden_dir, monkeypatch):
    """data/skit_dataset.py (the multi-material front-end of skitG) against what the REFERENCE's SkitDataset cached for two seeded
    materials at its fixed relative place ./datasets/singleskit_<material>_padded_<size>_x<multiplier>/: entry i belongs to material
    i % 2 and takes zoom level i; `S_paths` is always the first material's, `name` the entry's own"""
    from data.skit_dataset import SkitDataset

    g = np.load(os.path.join(golden_dir, "skit_dataset.npz"))
    for m, ms in zip(("matA", "matB"), mseeds):
        write_material(str(tmp_path / "datasets" / ("singleskit_%s_padded_400_x1" % m)), seed=ms, phase=phase)
    monkeypatch.chdir(tmp_path)
    random.seed(seed)
    np.random.seed(seed)
    ds = SkitDataset(dataset_opt("unused_root", phase, material_list=["matA", "matB"], padded_size=400, data_len=3, preprocess=pre,
                                 random_scale_max=1.04, crop_size=320))
    assert len(ds) == int(g[phase + "/len"]) == 3
    _compare_with_reference_cache(ds, g, phase)
    assert "style_code" not in ds[0]
    if phase == "train":        # the package's addition: a precomputed style code per material becomes the batch key the skitG model reads
        np.save(str(tmp_path / "datasets" / "singleskit_matB_padded_400_x1" / "style_code.npy"), np.arange(512, dtype=np.float64))
        kw = dict(material_list=["matA", "matB"], padded_size=400, data_len=3, preprocess=pre, random_scale_max=1.04, crop_size=320)
        with pytest.raises(FileNotFoundError, match="matA"):      # a code for one material only: refused at construction, the missing one named
            SkitDataset(dataset_opt("unused_root", phase, **kw))
        np.save(str(tmp_path / "datasets" / "singleskit_matA_padded_400_x1" / "style_code.npy"), np.ones(7, dtype=np.float64))
        with pytest.raises(ValueError, match="style_code_dim"):   # wrong length
            SkitDataset(dataset_opt("unused_root", phase, style_code_dim=512, **kw))
        np.save(str(tmp_path / "datasets" / "singleskit_matA_padded_400_x1" / "style_code.npy"), -np.arange(512, dtype=np.float64))
        random.seed(seed)
        np.random.seed(seed)
        ds2 = SkitDataset(dataset_opt("unused_root", phase, style_code_dim=512, **kw))
        assert ds2[1]["style_code"].dtype == torch.float32 and ds2[1]["style_code"].shape == (512,) and ds2[1]["style_code"][3] == 3
        assert ds2[0]["style_code"][3] == -3 and ds2[2]["style_code"][3] == -3
        assert torch.equal(ds2[1]["S"], ds[1]["S"])
    assert ds[0]["name"] == ds[2]["name"] and ds[0]["S_paths"] == ds[1]["S_paths"] and ds[0]["M_paths"] != ds[1]["M_paths"]
    if phase == "train":
        assert not torch.equal(ds[0]["S"], ds[2]["S"])       # same material, another zoom level / crop position


def test_skit_is_a_creatable_dataset_mode():
    from data import find_dataset_using_name
    from data.skit_dataset import SkitDataset

    assert find_dataset_using_name("skit") is SkitDataset
    with pytest.raises(NotImplementedError):
        SkitDataset(dataset_opt("x", "test", material_list=["a"], padded_size=1, use_external_test_input=True))
    with pytest.raises(ValueError):
        SkitDataset(dataset_opt("x", "test", material_list=[], padded_size=1))


def test_patchskit_is_a_creatable_dataset_mode(tmp_path):
    """--dataset_mode patchskit resolves through the factory (it used to raise 'not built')"""
    from data import find_dataset_using_name
    from data.patchskit_dataset import PatchSkitDataset

    assert find_dataset_using_name("patchskit") is PatchSkitDataset
