"""The drop-in boundary against the REFERENCE's own entry points (build container only: /root/reference does not exist on the GPU box,
these tests skip there).  No GPU here, so both recipes are driven up to `create_model`, which must refuse with the "HIP path only" error
-- everything before it (option parsing through the reference's flags, dataset creation, the imports of train.py / test.py:
util.visualizer.save_images, util.myhtml, util.util) has then resolved against the MI355X packages."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "visual-tactile-synthesis_amd")
REF = os.environ.get("VTS_REFERENCE_ROOT", "/root/reference")

needs_ref = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="reference tree not present (GPU box)")
FLAGS = ["--model", "sinskitG", "--gpu_ids", "0", "--dataset_mode", "synthetic", "--crop_size", "64", "--data_len", "2", "--lambda_G1_lpips", "0",
         "--lambda_G2_lpips", "0", "--use_vision_aided_loss", "False"]


def _run(cmd, cwd, env_extra=None):
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", **(env_extra or {}))
    env.pop("PYTHONPATH", None)
    return subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


@needs_ref
@pytest.mark.parametrize("script", ["train.py", "test.py"])
def test_reference_entry_points_run_on_the_shadow_packages(script, tmp_path):
    """INTEGRATION.md recipe 1: run_reference.py <reference>/train.py|test.py.  The run must get through the script's imports, the
    option parser and create_dataset, and stop in create_model because there is no GPU in this container."""
    r = _run([sys.executable, os.path.join(PKG, "run_reference.py"), os.path.join(REF, script)] + FLAGS +
             ["--checkpoints_dir", str(tmp_path), "--results_dir", str(tmp_path / "res")], cwd=str(tmp_path))
    err = r.stderr.decode()
    assert r.returncode != 0 and "runs on the MI355X HIP path only" in err, err[-3000:]
    assert "/reference/models" not in err and "/reference/data" not in err, err[-3000:]     # no frame inside the reference's packages
    out = r.stdout.decode()
    if script == "test.py":
        assert "The number of test images" in out


@needs_ref
def test_plain_pythonpath_invocation_is_the_documented_trap(tmp_path):
    """what INTEGRATION.md used to recommend: PYTHONPATH=<amd> python <reference>/train.py imports the REFERENCE's packages (the script
    directory precedes PYTHONPATH) -- it fails on the first pip package this image lacks instead of reaching the HIP model"""
    r = _run([sys.executable, os.path.join(REF, "train.py")] + FLAGS + ["--checkpoints_dir", str(tmp_path)], cwd=str(tmp_path),
             env_extra={"PYTHONPATH": PKG})
    assert r.returncode != 0 and "runs on the MI355X HIP path only" not in r.stderr.decode()


STUB_DRIVER = r'''
import importlib.util, os, sys
sys.path.insert(0, %(root)r)
from oracle import ref_import
ref_import.load()                                   # the reference first on sys.path, pip packages this image lacks stubbed
import types
for name in ("gspread", "oauth2client", "oauth2client.service_account"):      # imported by the reference's options (myutils.py:8-11)
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["oauth2client.service_account"].ServiceAccountCredentials = object
import models                                       # the REFERENCE's package
assert models.__file__.startswith(%(ref)r), models.__file__
spec = importlib.util.spec_from_file_location("vts_stub_gen", os.path.join(%(pkg)r, "models", "reference_stub.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
overlay = %(tmp)r                                   # stands in for <reference>/models (the reference tree is read-only here)
os.makedirs(os.path.join(overlay, "models"), exist_ok=True)
path = gen.write_stub(overlay, "sinskitG")
models.__path__.append(os.path.join(overlay, "models"))
cls = models.find_model_using_name("sinskitG_hip")  # the reference's own discovery rule (models/__init__.py:25-45)
from models.base_model import BaseModel
import vts_models.sinskitG_model as hip
assert issubclass(cls, BaseModel) and issubclass(cls, hip.SinSKITGModel) and cls.__name__ == "SinSKITGHipModel"
sys.argv = ["train.py", "--model", "sinskitG_hip", "--gpu_ids", "-1", "--checkpoints_dir", %(tmp)r, "--lambda_G1_lpips", "0", "--lambda_G2_lpips", "0",
            "--use_vision_aided_loss", "False"]
from options.train_options import TrainOptions      # the REFERENCE's parser: flags come from the stub class' modify_commandline_options
opt = TrainOptions().parse()
assert opt.model == "sinskitG_hip" and opt.netG == "unet256_custom" and opt.lambda_G1_L1 == 100.0 and opt.ngf == 10
try:
    models.create_model(opt)
except RuntimeError as e:
    assert "HIP path only" in str(e), e
    print("STUB_OK")
'''


@needs_ref
def test_recipe2_stub_class_resolves_through_the_reference_factory(tmp_path):
    """INTEGRATION.md recipe 2: the generated <reference>/models/sinskitG_hip_model.py is found by the reference's find_model_using_name,
    parses through the reference's option parser, and constructs up to the no-GPU refusal"""
    r = _run([sys.executable, "-c", STUB_DRIVER % dict(root=ROOT, ref=REF, pkg=PKG, tmp=str(tmp_path))], cwd=str(tmp_path))
    assert r.returncode == 0 and "STUB_OK" in r.stdout.decode(), (r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])


def test_myhtml_and_save_images_shims(tmp_path):
    """the two names reference/test.py:6-7 imports from util: a page writer and save_images(webpage, visuals, image_path, ...)"""
    import numpy as np
    import torch
    from util import myhtml
    from util.visualizer import save_images

    page = myhtml.HTML(str(tmp_path), "t")
    vis = {"fake_I": torch.rand(1, 3, 8, 8) * 2 - 1, "fake_gx": torch.rand(1, 1, 8, 8) * 0.2, "fake_gy": torch.rand(1, 1, 8, 8) * 0.2}
    written = save_images(page, vis, ["/x/y/sample_7.png"], width=64, save_raw_gxgy=True, save_raw_arr_vis=True, save_style_image_name=True,
                          style_image_name="Denim")
    page.save()
    assert len(written) == 3 and all(os.path.isfile(p) and p.endswith("sample_7_style_Denim.png") for p in written)
    raw = np.load(os.path.join(page.get_image_dir(), "fake_gxgy_raw", "fake_gxgy_raw.npz"))
    assert set(raw.files) == {"fake_gx", "fake_gy"}
    html = open(os.path.join(str(tmp_path), "index.html")).read()
    assert "sample_7_style_Denim" in html and "fake_I" in html
    with pytest.raises(NotImplementedError):
        save_images(page, vis, ["a.png"], use_wandb=True)
