"""Headless inference entry point with the reference test.py's control flow
(/root/reference/test.py:31-116): load `<epoch>_net_G.pth`, forward every test item, time it.

    python test.py --model sinskitG --gpu_ids 0 --dataset_mode synthetic --crop_size 1024 --epoch latest --eval
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
from options.test_options import TestOptions  # noqa: E402
from util.image_io import save_images  # noqa: E402

if __name__ == "__main__":
    opt = TestOptions().parse()
    opt.num_threads = 0
    opt.batch_size = max(1, opt.batch_size)
    opt.serial_batches = True
    opt.no_flip = True
    opt.rank = 0
    dataset = create_dataset(opt)
    model = create_model(opt)
    out_dir = os.path.join(opt.results_dir, opt.name, "%s_%s" % (opt.phase, opt.epoch))
    os.makedirs(out_dir, exist_ok=True)
    for i, data in enumerate(dataset):
        if i == 0:
            model.setup(opt)
            model.parallelize()
            if opt.eval:
                model.eval()
        if i >= opt.num_test:
            break
        model.set_input(data, phase="test")
        torch.cuda.synchronize()
        t0 = time.time()
        model.test(timing=True)
        torch.cuda.synchronize()
        dt = time.time() - t0
        visuals = model.get_current_visuals()
        name = data["name"][0] if isinstance(data["name"], (list, tuple)) else data["name"]
        torch.save({k: v.detach().cpu() for k, v in visuals.items() if torch.is_tensor(v)}, os.path.join(out_dir, "%s.pt" % name))
        # the files the reference's save_images writes (test.py:93): PNG per visual, raw gx / gy as .npz (and .npy on request)
        save_images(os.path.join(out_dir, "images"), visuals, name, save_raw_gxgy=True,
                    save_raw_arr_vis=bool(getattr(opt, "save_raw_arr_vis", False)))
        print("processed %s in %.2f ms (%d images)" % (name, dt * 1e3, data["S"].size(0)))
        if hasattr(model, "compute_metrics"):
            model.compute_metrics()
            print("metrics:", dict(model.get_current_metrics()))
