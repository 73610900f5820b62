"""Run a few training steps of the HIP path over a matrix of sizes / batch sizes / option toggles and report finiteness.
python tools/probes/robustness_matrix.py"""
import contextlib
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402
from torch.utils.data import default_collate  # noqa: E402

from data.synthetic_dataset import make_sample  # noqa: E402
from models import create_model  # noqa: E402
from options.train_options import TrainOptions  # noqa: E402

BASE = ("--gpu_ids 0 --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False --checkpoints_dir /tmp/vts_rob --name r ")
CASES = [
    ("skitG 1536 b1", "--model skitG --crop_size 1536 --batch_size 1", 1536, 1),
    ("skitG 1536 b2 style project+concat", "--model skitG --crop_size 1536 --batch_size 2 --style_code_mapping_mode project", 1536, 2),
    ("skitG 1536 b1 style project+adain", "--model skitG --crop_size 1536 --batch_size 1 --style_code_mapping_mode project --style_code_mode adain", 1536, 1),
    ("skitG 512 b8", "--model skitG --crop_size 512 --batch_size 8", 512, 8),
    ("sinskitG 768 b2", "--model sinskitG --crop_size 768 --batch_size 2", 768, 2),
    ("sinskitG lsgan", "--model sinskitG --crop_size 256 --batch_size 2 --gan_mode lsgan", 256, 2),
    ("sinskitG hinge no-diffaug", "--model sinskitG --crop_size 256 --batch_size 1 --gan_mode hinge --use_diffaug False", 256, 1),
    ("sinskitG vanilla no-moreT", "--model sinskitG --crop_size 256 --batch_size 1 --gan_mode vanilla --use_more_fakeT False", 256, 1),
    ("sinskitG no-L1", "--model sinskitG --crop_size 256 --batch_size 1 --lambda_G1_L1 0 --lambda_G2_L1 0", 256, 1),
    ("sinskitG G-only-D1", "--model sinskitG --crop_size 256 --batch_size 1 --lambda_G2_GAN 0", 256, 1),
    ("sinskitG separate0", "--model sinskitG --crop_size 256 --batch_size 1 --num_layer_separate 0", 256, 1),
    ("sinskitG resnet6", "--model sinskitG --crop_size 256 --batch_size 2 --netG resnet_6blocks", 256, 2),
    ("sinskitG netD stylegan2", "--model sinskitG --crop_size 256 --load_size 256 --batch_size 2 --netD stylegan2", 256, 2),
    ("sinskitG D depth 2 / D2 depth 4", "--model sinskitG --crop_size 256 --batch_size 2 --n_layers_D 2 --n_layers_D2 4", 256, 2),
    ("skitG 1024 b4 D depth 4", "--model skitG --crop_size 1024 --batch_size 4 --n_layers_D 4", 1024, 4),
    ("sinskitG vanilla D depth 5", "--model sinskitG --crop_size 512 --batch_size 1 --gan_mode vanilla --n_layers_D 5", 512, 1),
    ("sinskitG diffaugment bsctno", "--model sinskitG --crop_size 256 --batch_size 2 --diffaugment bsctno", 256, 2),
    ("skitG 1024 b4 diffaugment cto", "--model skitG --crop_size 1024 --batch_size 4 --diffaugment cto", 1024, 4),
    ("sinskitG Up-block dropout", "--model sinskitG --crop_size 256 --batch_size 2 --no_dropout False", 256, 2),
    ("skitG 1024 b4 Up-block dropout", "--model skitG --crop_size 1024 --batch_size 4 --no_dropout False", 1024, 4),
    ("sinskitG resnet9 block dropout", "--model sinskitG --crop_size 256 --batch_size 2 --netG resnet_9blocks --no_dropout False", 256, 2),
    ("sinskitG 200 steps + lr decay", "--model sinskitG --crop_size 256 --batch_size 1", 256, 1),
]


def main():
    for name, flags, size, n in CASES:
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                opt = TrainOptions(cmd_line=BASE + flags).parse()
                model = create_model(opt)
                model.setup(opt)
                model.parallelize()
                model.train()
            sd = opt.style_code_dim if getattr(opt, "use_style_code", False) else 0
            batch = default_collate([make_sample(size, 64, 64, 7 + i, style_dim=sd) for i in range(n)])
            steps = 200 if "200" in name else 6
            for it in range(steps):
                model.set_input(batch, phase="train")
                model.optimize_parameters(epoch=1)
                if it % 50 == 49:
                    model.update_learning_rate()
            torch.cuda.synchronize()
            losses = model.get_current_losses()
            ok = all(v == v and abs(v) < 1e30 for v in losses.values())
            print("%-32s %s  %s" % (name, "ok " if ok else "NON-FINITE", {k: round(v, 3) for k, v in list(losses.items())[:4]}))
            del model
        except Exception as e:   # report and go on
            print("%-32s FAILED: %s: %s" % (name, type(e).__name__, str(e)[:200]))


if __name__ == "__main__":
    main()
