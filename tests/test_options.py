"""Flags round-trip: names, types and defaults equal the reference's parser for both models and
both phases (fixture: tests/golden/ref_option_defaults.json, dumped from the reference's argparse)."""
import contextlib
import io
import json
import os

import pytest


def parse(cls, cmd):
    with contextlib.redirect_stdout(io.StringIO()):
        return cls(cmd_line=cmd).parse()


@pytest.mark.parametrize("model", ["sinskitG", "skitG", "pix2pixHD"])
@pytest.mark.parametrize("phase", ["train", "test"])
def test_defaults_match_reference(golden_dir, model, phase):
    from options.test_options import TestOptions
    from options.train_options import TrainOptions

    ref = json.load(open(os.path.join(golden_dir, "ref_option_defaults.json")))["%s_%s" % (model, phase)]
    opt = parse(TrainOptions if phase == "train" else TestOptions, "--model %s --gpu_ids -1 --checkpoints_dir /tmp/vts_opt" % model)
    for k, v in ref.items():
        if k in ("gpu_ids", "checkpoints_dir", "model"):   # given on the command line
            continue
        assert hasattr(opt, k), k
        d = float("inf") if v["default"] == "inf" else v["default"]
        assert getattr(opt, k) == d or str(getattr(opt, k)) == str(d), (k, getattr(opt, k), d)
    assert opt.gpu_ids == []


def test_quirks_kept():
    from options.train_options import TrainOptions

    # unknown flags are ignored; prefix abbreviation works (--dataset -> --dataset_mode); str2bool parsing
    opt = parse(TrainOptions, "--model sinskitG --gpu_ids -1 --no_such_flag 3 --dataset synthetic --use_diffaug false "
                              "--smooth_GAN_label --checkpoints_dir /tmp/vts_opt --suffix x{ngf}")
    assert opt.dataset_mode == "synthetic" and opt.use_diffaug is False and opt.smooth_GAN_label is True
    assert opt.name.endswith("_x10")
    assert os.path.exists("/tmp/vts_opt/%s/train_opt.txt" % opt.name)


def test_unbuilt_third_party_terms_raise():
    from models.sinskitG_model import SinSKITGModel
    from options.train_options import TrainOptions

    opt = parse(TrainOptions, "--model sinskitG --gpu_ids 0 --checkpoints_dir /tmp/vts_opt")
    # the reference's default flags: accepted at construction and before the warm-up epoch (upstream does not call the CLIP discriminator
    # there either); the term itself is not built and raises from the warm-up epoch on
    SinSKITGModel._check_unbuilt_terms(opt)
    SinSKITGModel._check_unbuilt_terms(opt, opt.vision_aided_warmup_epoch - 1)
    with pytest.raises(NotImplementedError, match="CLIP vision-aided"):
        SinSKITGModel._check_unbuilt_terms(opt, opt.vision_aided_warmup_epoch)
    # LPIPS is built since round 3: the reference's default lambdas pass the check once the CLIP term is switched off
    opt = parse(TrainOptions, "--model sinskitG --gpu_ids 0 --checkpoints_dir /tmp/vts_opt --use_vision_aided_loss False")
    assert opt.lambda_G1_lpips == 1.0 and opt.lambda_G2_lpips == 10.0
    SinSKITGModel._check_unbuilt_terms(opt, 10 ** 6)


def test_synthetic_dataset_contract():
    from data import create_dataset
    from options.train_options import TrainOptions

    opt = parse(TrainOptions, "--model skitG --gpu_ids -1 --dataset_mode synthetic --crop_size 64 --batch_size 2 --data_len 4 "
                              "--checkpoints_dir /tmp/vts_opt")
    ds = create_dataset(opt)
    assert len(ds) == 4
    b = next(iter(ds))
    assert b["S"].shape == (2, 1, 64, 64) and b["I"].shape == (2, 3, 64, 64) and b["M"].shape == (2, 1, 64, 64)
    assert b["T_images"].shape == (2, 64, 2, 32, 32) and b["T_coords"].shape == (2, 64, 8) and b["I_masks"].shape == (2, 64, 32, 32)
    assert b["T_coords"].dtype.is_floating_point and float(b["M"].max()) == 1.0
    assert b["style_code"].shape == (2, 512)
    assert abs(float(b["style_code"][0].norm()) - 1) < 1e-5
    assert set(b["augmentation_params"]) >= {"H", "W", "scale_factor_h", "crop_pos_x", "resize_ratio_w"}


def test_unbuilt_dataset_modes_raise_instead_of_aliasing():
    """a reference dataset mode whose front-end is not built must never resolve to synthetic noise"""
    from data import create_dataset
    from options.train_options import TrainOptions

    opt = parse(TrainOptions, "--model skitG --gpu_ids -1 --dataset_mode aligned --checkpoints_dir /tmp/vts_opt")
    with pytest.raises(NotImplementedError, match="dataset_mode aligned"):
        create_dataset(opt)
    # `skit` is built since round 4 (data/skit_dataset.py): it resolves through the command line, and says what it needs
    opt = parse(TrainOptions, "--model skitG --gpu_ids -1 --dataset_mode skit --checkpoints_dir /tmp/vts_opt")
    with pytest.raises(ValueError, match="material_list"):
        create_dataset(opt)


def test_rank_device_comes_from_local_rank(monkeypatch):
    """torchrun launches: parse() binds the process to cuda:LOCAL_RANK even though --gpu_ids defaults to 0"""
    import torch
    from options.train_options import TrainOptions

    chosen = []
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: chosen.append(i))
    opt = parse(TrainOptions, "--model skitG --gpu_ids 0 --dataset_mode synthetic --checkpoints_dir /tmp/vts_opt")
    assert opt.gpu_ids == [5] and chosen == [5]


def test_define_G_builds_the_stylegan2_generators_with_reference_keys():
    """networks.define_G(netG='smallstylegan2' | 'stylegan2') (reference networks.py:297-300): parameter container with the reference's
    state-dict keys (the list the reference module reported is in tests/golden/stylegan2_g_32.npz); CPU-safe (no compute)"""
    import argparse
    import os

    import numpy as np

    from models import networks
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stylegan2_g_32.npz"))
    opt = argparse.Namespace(load_size=32, crop_size=32, stylegan2_G_num_downsampling=2)
    small = networks.define_G(4, 3, 2, "smallstylegan2", opt=opt)
    assert sorted(small.state_dict().keys()) == sorted(g["ref_keys"].tolist()) and small.inject_noise is False and small.n_blocks == 2
    full = networks.define_G(4, 3, 2, "stylegan2", opt=opt)
    assert full.inject_noise is True and full.n_blocks == 6 and len(full.decoder.convs) == 3 + 2 + 1
