"""Headless Visualizer: loss_log.txt + pickled metrics only.

The reference Visualizer (/root/reference/util/visualizer.py:151-483) also drives
visdom / wandb / HTML; those are observability and out of scope.
The method names and the loss_log.txt line format are kept so train.py drops in.
"""
import ntpath
import os
import pickle
import time

from . import image_io, util


def save_images(webpage, visuals, image_path, aspect_ratio=1.0, width=256, use_wandb=False, save_raw_gxgy=False, save_raw_arr_vis=False,
                save_style_image_name=False, style_image_name=None):
    """Signature of the reference's save_images (/root/reference/util/visualizer.py:30-148), as its test.py:93 calls it: writes the PNG
    / .npz / .npy / coords-json file set through util.image_io.save_images into webpage.get_image_dir() and lists the PNGs on the page.
    wandb logging is observability (out of scope): use_wandb=True raises."""
    if use_wandb:
        raise NotImplementedError("wandb logging is not built in the MI355X package (pass --use_wandb False)")
    if save_style_image_name:
        assert style_image_name is not None, "style_image_name is None"
    image_dir = webpage.get_image_dir()
    written = image_io.save_images(image_dir, visuals, image_path, save_raw_gxgy=save_raw_gxgy, save_raw_arr_vis=save_raw_arr_vis,
                                   style_image_name=style_image_name if save_style_image_name else None)
    name = os.path.splitext(ntpath.basename(image_path[0] if isinstance(image_path, (list, tuple)) else image_path))[0]
    if save_style_image_name:
        name += "_style_%s" % style_image_name
    webpage.add_header(name)
    rel = [os.path.relpath(pth, image_dir) for pth in written]
    webpage.add_images(rel, [os.path.dirname(r) for r in rel], rel, width=width)
    return written


class Visualizer:
    def __init__(self, opt):
        self.opt = opt
        self.name = opt.name
        self.saved = False
        self.log_dir = os.path.join(opt.checkpoints_dir, opt.name)
        util.mkdirs(self.log_dir)
        self.log_name = os.path.join(self.log_dir, "loss_log.txt")
        with open(self.log_name, "a") as f:
            f.write("================ Training Loss (%s) ================\n" % time.strftime("%c"))

    def reset(self):
        self.saved = False

    def display_current_results(self, visuals, epoch=None, save_result=False, step=None):
        pass

    def plot_current_losses(self, epoch, counter_ratio, losses, use_visdom=True, step=None):
        pass

    def print_current_losses(self, epoch, iters, losses, t_comp, t_data, t_input):
        # same line layout as /root/reference/util/visualizer.py:388-407
        message = "(epoch: %d, iters: %d, time: %.3f, data: %.3f, input: %.3f) " % (epoch, iters, t_comp, t_data, t_input)
        for k, v in losses.items():
            message += "%s: %.3f " % (k, v)
        print(message)
        with open(self.log_name, "a") as f:
            f.write("%s\n" % message)

    def print_current_metrics(self, epoch, eval_metrics):
        message = "(epoch: %d) " % epoch
        for k, v in eval_metrics.items():
            message += "%s: %.3f " % (k, v)
        print(message)
        with open(self.log_name, "a") as f:
            f.write("%s\n" % message)

    def plot_current_metrics(self, eval_metrics, use_visdom=False, step=None):
        pass

    def save_current_metrics(self, eval_metrics, return_web_dir=False, save_metrics=True, epoch=None, verbose=False, save_metric_index=False, i=0):
        """reference util/visualizer.py:443-471: <results_dir>/<name>/<phase>_<epoch>/eval_metrics[_<i>].pkl holds the metrics dict"""
        if epoch is None:
            epoch = getattr(self.opt, "epoch", "latest")
        web_dir = os.path.join(self.opt.results_dir, self.opt.name, "%s_%s" % (self.opt.phase, epoch))
        os.makedirs(web_dir, exist_ok=True)
        if save_metrics:
            dict_path = os.path.join(web_dir, "eval_metrics_%d.pkl" % i if save_metric_index else "eval_metrics.pkl")
            with open(dict_path, "wb") as f:
                pickle.dump(dict(eval_metrics), f)
            if verbose:
                print("save eval metrics to %s" % dict_path)
        if return_web_dir:
            return web_dir

    def plot_epoch_time(self, epoch, epoch_time):
        pass
