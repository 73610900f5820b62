"""Headless training entry point with the reference train.py's control flow
(/root/reference/train.py:20-215): per-iteration set_input -> optimize_parameters, periodic
loss logging and `latest` checkpoints, per-epoch metrics / best / epoch checkpoints, LR decay.

    python train.py --model sinskitG --gpu_ids 0 --dataset_mode synthetic --crop_size 1024 \
        --lambda_G1_lpips 0 --lambda_G2_lpips 0 --use_vision_aided_loss False
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   # data parallel

The reference's own train.py also runs unchanged against this package:
it only touches create_dataset / create_model / Visualizer and the BaseModel methods.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# Host threads: torch's CPU ops fan out over an OpenMP pool (128 workers on the MI355X hosts) whose workers SPIN after each
# parallel region; that starved the HIP runtime's completion thread and stalled graph-replayed steps by 70..170 ms
# (tools/probes/stall_bisect2.py).  Passive waiting must be chosen before libgomp starts, i.e. before `import torch`.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
import torch  # noqa: E402

from data import create_dataset  # noqa: E402
from models import create_model  # noqa: E402
from options.train_options import TrainOptions  # noqa: E402
from util.visualizer import Visualizer  # noqa: E402
from vts import ddp  # noqa: E402


def train_model(epoch, total_iters, dataset, model, opt, visualizer, dataset_size):
    model.train()
    epoch_iter = 0
    t_iter = time.time()
    for i, data in enumerate(dataset):
        t_data = time.time() - t_iter
        batch_size = data["S"].size(0)
        total_iters += batch_size
        epoch_iter += batch_size
        t0 = time.time()
        if epoch == opt.epoch_count and i == 0:
            model.setup(opt)
            model.parallelize()
        model.set_input(data, phase="train", verbose=False)
        t_input = (time.time() - t0) / batch_size
        t1 = time.time()
        model.optimize_parameters(epoch)
        if total_iters % opt.print_freq == 0:
            torch.cuda.synchronize()
            t_opt = (time.time() - t1) / batch_size
            visualizer.print_current_losses(epoch, epoch_iter, model.get_current_losses(), t_opt, t_data, t_input)
        if total_iters % opt.save_latest_freq == 0 and opt.rank == 0:
            model.save_networks("iter_%d" % total_iters if opt.save_by_iter else "latest")
        t_iter = time.time()
    return total_iters


if __name__ == "__main__":
    rank, world = ddp.init_from_env("cuda")
    opt = TrainOptions().parse()
    opt.rank, opt.world_size = rank, world
    # parse() bound this process to cuda:LOCAL_RANK; every rank draws its own DiffAugment / sampler stream
    torch.manual_seed(1234 + rank)
    dataset = create_dataset(opt)
    dataset_size = len(dataset)
    model = create_model(opt)
    visualizer = Visualizer(opt)
    print("The number of training images = %d" % dataset_size)
    total_iters = (opt.epoch_count - 1) * dataset_size
    best = None
    for epoch in range(opt.epoch_count, opt.n_epochs + opt.n_epochs_decay + 1):
        t_epoch = time.time()
        dataset.set_epoch(epoch)
        if opt.train_for_each_epoch:
            total_iters = train_model(epoch, total_iters, dataset, model, opt, visualizer, dataset_size)
        if getattr(opt, "val_for_each_epoch", False) and hasattr(model, "compute_metrics"):
            # validation pass (train.py:89-160 of the reference): forward on a sample, metrics on its validation patches
            for data in dataset:
                model.eval()
                model.set_input(data, phase="val")
                model.test()
                model.compute_metrics()
                model.train()
                break
        metrics = model.get_current_metrics()
        visualizer.print_current_metrics(epoch, metrics)
        visualizer.save_current_metrics(metrics, epoch=epoch)
        if rank == 0:
            if best is None:
                best = metrics
                model.save_networks("best")
            else:
                better = sum(1 for k, v in metrics.items() if "train" not in k and
                             (v < best[k] if any(x in k for x in ("LPIPS", "AE", "MSE", "SIFID")) else v > best[k]))
                total = sum(1 for k in metrics if "train" not in k)
                if better >= total // 2:
                    best = metrics
                    model.save_networks("best")
            if epoch % opt.save_epoch_freq == 0:
                model.save_networks("latest")
                model.save_networks(epoch)
        print("End of epoch %d / %d \t Time Taken: %d sec" % (epoch, opt.n_epochs + opt.n_epochs_decay, time.time() - t_epoch))
        if opt.train_for_each_epoch:
            model.update_learning_rate()
        # pix2pixHD: after niter_fix_global epochs on the local enhancer alone, train the whole generator (reference train.py:209-211)
        if getattr(opt, "niter_fix_global", 0) != 0 and epoch == opt.niter_fix_global:
            model.update_fixed_params()
