"""Headless Visualizer: loss_log.txt + pickled metrics only.

The reference Visualizer (/root/reference/util/visualizer.py:151-483) also drives
visdom / wandb / HTML; those are observability and out of scope.
The method names and the loss_log.txt line format are kept so train.py drops in.
"""
import os
import pickle
import time

from . import util


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

    def display_current_results(self, visuals, epoch, save_result, step=None):
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

    def save_current_metrics(self, eval_metrics, epoch=None, **kw):
        path = os.path.join(self.log_dir, "eval_metrics.pkl")
        hist = {}
        if os.path.exists(path):
            with open(path, "rb") as f:
                hist = pickle.load(f)
        hist[epoch] = dict(eval_metrics)
        with open(path, "wb") as f:
            pickle.dump(hist, f)

    def plot_epoch_time(self, epoch, epoch_time):
        pass
