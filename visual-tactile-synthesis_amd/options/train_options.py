"""Training flags (mirror of /root/reference/options/train_options.py:13-81)."""
from .base_options import FLAG, NB, BaseOptions, add_table

TRAIN_TABLE = [
    ("save_latest_freq", int, 5000),
    ("save_epoch_freq", int, 5),
    ("evaluation_freq", int, 5000),
    ("save_by_iter", FLAG, False),
    ("continue_train", FLAG, False),
    ("epoch_count", int, 1),
    ("phase", str, "train"),
    ("pretrained_name", str, None),
    ("n_epochs", int, 200),
    ("n_epochs_decay", int, 200),
    ("beta1", float, 0.5),
    ("beta2", float, 0.999),
    ("lr", float, 0.0002),
    ("gan_mode", str, "nonsaturating"),
    ("pool_size", int, 50),
    ("lr_policy", str, "linear"),
    ("lr_decay_iters", int, 50),
    ("val_for_each_epoch", NB, True),
    ("train_for_each_epoch", NB, True),
    ("validation_freq", int, 100),
]


class TrainOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        add_table(parser, TRAIN_TABLE)
        self.isTrain = True
        return parser
