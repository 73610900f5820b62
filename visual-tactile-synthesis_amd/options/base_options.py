"""Three-pass argparse configuration: base flags -> model flags -> dataset flags.

Behavioural mirror of /root/reference/options/base_options.py:221-312
(`gather_options`, `print_options`, `parse`).  Flags are declared from compact
tables rather than one add_argument call each; names, types and defaults are
checked against the reference's parser by tests/test_options.py using the
fixture tests/golden/ref_option_defaults.json.

Quirks kept on purpose (SURVEY.md §5 "Config / flags"):
  * parse_known_args: unknown flags are ignored silently;
  * argparse prefix abbreviation stays enabled (`--dataset patchskit` works);
  * `--gpu_ids -1` selects CPU.
"""
import argparse
import os

import data
import models
from util import util

S2B = util.str2bool
FLAG = "flag"  # action="store_true"
NB = "nargs_bool"  # type=str2bool, nargs="?", const=True

# (name, type, default[, choices])
BASE_TABLE = [
    ("dataroot", str, "placeholder"),
    ("name", str, "experiment_name"),
    ("use_wandb", FLAG, False),
    ("easy_label", str, "experiment_name"),
    ("gpu_ids", str, "0"),
    ("checkpoints_dir", str, "./checkpoints"),
    ("model", str, "sinskit"),
    ("ngf", int, 64),
    ("ndf", int, 64),
    ("netD", str, "basic", ["basic", "n_layers", "pixel", "patch", "tilestylegan2", "stylegan2", "multiscale"]),
    ("netG", str, "resnet_9blocks",
     ["resnet_9blocks", "resnet_6blocks", "unet_256", "unet_128", "stylegan2", "smallstylegan2", "resnet_cat", "unet256_custom"]),
    ("n_layers_D", int, 3),
    ("normG", str, "instance", ["instance", "batch", "none"]),
    ("normD", str, "batch", ["instance", "batch", "none"]),
    ("init_type", str, "xavier", ["normal", "xavier", "kaiming", "orthogonal"]),
    ("init_gain", float, 0.02),
    ("no_dropout", NB, True),
    ("no_antialias", FLAG, False),
    ("no_antialias_up", FLAG, False),
    ("dataset_mode", str, "unaligned"),
    ("direction", str, "AtoB"),
    ("serial_batches", FLAG, False),
    ("num_threads", int, 4),
    ("batch_size", int, 1),
    ("load_size", int, 286),
    ("crop_size", int, 256),
    ("max_dataset_size", int, float("inf")),
    ("preprocess", str, "resize_and_crop"),
    ("no_flip", FLAG, False),
    ("display_winsize", int, 256),
    ("random_scale_max", float, 3.0),
    ("epoch", str, "latest"),
    ("verbose", FLAG, False),
    ("suffix", str, ""),
    ("display_freq", int, 400),
    ("display_ncols", int, 20),
    ("display_id", int, None),
    ("display_server", str, "http://localhost"),
    ("display_env", str, "main"),
    ("display_port", int, 8097),
    ("update_html_freq", int, 10000),
    ("update_html_epch_freq", int, 50),
    ("print_freq", int, 100),
    ("no_html", FLAG, False),
    ("results_dir", str, "./results/"),
    ("stylegan2_G_num_downsampling", int, 1),
]


def add_table(parser, table):
    """Declare every row of a flag table on `parser`."""
    for row in table:
        name, typ, default = row[0], row[1], row[2]
        kw = {}
        if len(row) > 3 and row[3] is not None:
            kw["choices"] = row[3]
        if len(row) > 4:
            kw["nargs"] = row[4]
        if typ == FLAG:
            parser.add_argument("--" + name, action="store_true")
            if default:
                parser.set_defaults(**{name: default})
        elif typ == NB:
            parser.add_argument("--" + name, type=S2B, nargs="?", const=True, default=default)
        else:
            parser.add_argument("--" + name, type=typ, default=default, **kw)
    return parser


class BaseOptions:
    def __init__(self, cmd_line=None):
        self.initialized = False
        self.cmd_line = cmd_line.split() if cmd_line is not None else None

    def initialize(self, parser):
        add_table(parser, BASE_TABLE)
        self.initialized = True
        return parser

    def _known(self, parser):
        if self.cmd_line is None:
            return parser.parse_known_args()[0]
        return parser.parse_known_args(self.cmd_line)[0]

    def gather_options(self):
        if not self.initialized:
            parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
            parser = self.initialize(parser)
        opt = self._known(parser)
        parser = models.get_option_setter(opt.model)(parser, self.isTrain)
        opt = self._known(parser)
        parser = data.get_option_setter(opt.dataset_mode)(parser, self.isTrain)
        self.parser = parser
        return self._known(parser)

    def print_options(self, opt):
        lines = ["----------------- Options ---------------"]
        for k, v in sorted(vars(opt).items()):
            default = self.parser.get_default(k)
            comment = "\t[default: %s]" % str(default) if v != default else ""
            lines.append("{:>25}: {:<30}{}".format(str(k), str(v), comment))
        lines.append("----------------- End -------------------")
        message = "\n".join(lines)
        if not getattr(opt, "quiet", False):
            print(message)
        expr_dir = os.path.join(opt.checkpoints_dir, opt.name)
        try:
            util.mkdirs(expr_dir)
            with open(os.path.join(expr_dir, "{}_opt.txt".format(opt.phase)), "wt") as f:
                f.write(message + "\n")
        except (PermissionError, OSError) as e:
            print("could not write options file: {}".format(e))

    def parse(self):
        import torch

        opt = self.gather_options()
        opt.isTrain = self.isTrain
        if opt.suffix:
            opt.name = opt.name + "_" + opt.suffix.format(**vars(opt))
        self.print_options(opt)
        ids = [int(s) for s in opt.gpu_ids.split(",")]
        opt.gpu_ids = [i for i in ids if i >= 0]
        # one process per GPU (torchrun): the rank's device is LOCAL_RANK, whatever --gpu_ids says (its default "0" would
        # otherwise put every rank of a data-parallel launch on GPU 0)
        if opt.gpu_ids and int(os.environ.get("WORLD_SIZE", "1")) > 1 and "LOCAL_RANK" in os.environ:
            ndev = torch.cuda.device_count() if torch.cuda.is_available() else 1
            opt.gpu_ids = [int(os.environ["LOCAL_RANK"]) % max(ndev, 1)]
        if opt.gpu_ids and torch.cuda.is_available():
            torch.cuda.set_device(opt.gpu_ids[0])
        self.opt = opt
        return opt
