"""BaseModel: the duck-typed contract train.py / test.py drive.

Mirror of /root/reference/models/base_model.py:8-338 (method names, name-list driven
losses / metrics / visuals, checkpoint file layout `<label>_net_<G|D|D2>.pth` with the
reference's state_dict keys, `module.` prefix stripping on load).  Differences, all on the
far side of the boundary: networks are parameter containers executed by vts.engine on the
HIP path; `parallelize()` attaches RCCL gradient buckets (one process per GPU) instead of
wrapping in nn.DataParallel; optimiser state is not part of the checkpoint (as upstream).
"""
import os
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch

from . import networks


class BaseModel(ABC):
    def __init__(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.device = torch.device("cuda:{}".format(self.gpu_ids[0])) if self.gpu_ids else torch.device("cpu")
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names = []
        self.model_names = []
        self.visual_names = []
        self.metric_names = []
        self.optimizers = []
        self.image_paths = []
        self.metric = 0

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    @abstractmethod
    def set_input(self, input):
        pass

    @abstractmethod
    def forward(self):
        pass

    @abstractmethod
    def optimize_parameters(self):
        pass

    def setup(self, opt):
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
        if not self.isTrain or opt.continue_train:
            self.load_networks(opt.epoch)
        self.print_networks(opt.verbose)

    def parallelize(self):
        """Reference: nn.DataParallel wrap (base_model.py:104-108).  Here: data-parallel ranks
        (one process per GPU) sync replicas and all-reduce flat gradient buckets over RCCL."""
        from vts import ddp

        self.ddp = ddp.attach(self)

    def data_dependent_initialize(self, data):
        pass

    def eval(self):
        for name in self.model_names:
            getattr(self, "net" + name).eval()

    def train(self):
        for name in self.model_names:
            getattr(self, "net" + name).train()

    def test(self):
        with torch.no_grad():
            self.forward()
            self.compute_visuals()

    def compute_visuals(self):
        pass

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            if self.opt.lr_policy == "plateau":
                scheduler.step(self.metric)
            else:
                scheduler.step()
        lr = self.optimizers[0].param_groups[0]["lr"]
        print("learning rate = %.7f" % lr)

    def get_current_visuals(self):
        visual_ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str) and hasattr(self, name):
                visual_ret[name] = getattr(self, name)
        return visual_ret

    def get_current_losses(self):
        errors_ret = OrderedDict()
        for name in self.loss_names:
            errors_ret["l_" + name] = float(getattr(self, "loss_" + name))
        return errors_ret

    def get_current_metrics(self):
        metrics_ret = OrderedDict()
        for name in self.metric_names:
            metrics_ret["m_" + name] = float(getattr(self, "metric_" + name))
        return metrics_ret

    def save_networks(self, epoch):
        os.makedirs(self.save_dir, exist_ok=True)
        for name in self.model_names:
            net = getattr(self, "net" + name)
            sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
            torch.save(sd, os.path.join(self.save_dir, "%s_net_%s.pth" % (epoch, name)))

    def load_networks(self, epoch):
        for name in self.model_names:
            load_filename = "%s_net_%s.pth" % (epoch, name)
            if self.opt.isTrain and getattr(self.opt, "pretrained_name", None) is not None:
                load_dir = os.path.join(self.opt.checkpoints_dir, self.opt.pretrained_name)
            else:
                load_dir = self.save_dir
            load_path = os.path.join(load_dir, load_filename)
            if not os.path.exists(load_path):
                print("cannot find model path", load_path, "skip")  # reference: warn and continue (:264-267)
                continue
            net = getattr(self, "net" + name)
            state_dict = torch.load(load_path, map_location="cpu", weights_only=True)
            clean = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())
            own = net.state_dict()
            extra = [k for k in clean if k not in own]
            if extra:
                # e.g. the reference's never-used style_code_mapping<i> layers
                print("ignoring %d checkpoint entries with no counterpart: %s ..." % (len(extra), extra[:3]))
                for k in extra:
                    del clean[k]
            missing = [k for k in own if k not in clean]
            if missing:
                print("Error loading model: checkpoint lacks", missing[:5])
                continue
            net.load_state_dict(clean)  # copies in place: flat-buffer views stay attached

    def print_networks(self, verbose):
        print("---------- Networks initialized -------------")
        for name in self.model_names:
            net = getattr(self, "net" + name)
            num_params = sum(p.numel() for p in net.parameters())
            if verbose:
                print(net)
            print("[Network %s] Total number of parameters : %.3f M" % (name, num_params / 1e6))
        print("-----------------------------------------------")

    def set_requires_grad(self, nets, requires_grad=False):
        """No-op: the HIP path has no autograd; which gradients are computed is decided by the step schedule."""
        pass

    def generate_visuals_for_evaluation(self, data, mode):
        return {}
