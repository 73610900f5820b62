"""Dataset factory (mirror of /root/reference/data/__init__.py:17-104).

`--dataset_mode singleskit` is the TouchClothing front-end (data/singleskit_dataset.py, pinned to the reference's class),
`--dataset_mode patchskit` its paired-patch form for the pix2pixHD baseline (data/patchskit_dataset.py, pinned likewise);
`--dataset_mode synthetic` is the seeded generator bench.py / the tests use (same post-collate batch-dict contract).  A dataset mode whose front-end is not built in this package RAISES when a dataset is created -- it is never silently
replaced by synthetic noise (a maintainer pointing --dataroot at real TouchClothing data must not train on noise).
"""
import importlib

import torch.utils.data

# reference dataset modes whose CPU front-end is not built here: their option setter resolves (the reference parser asks for it
# while gathering options, options/base_options.py:238-240) but create_dataset refuses them
_UNBUILT = ("aligned",)


def find_dataset_using_name(dataset_name):
    if dataset_name in _UNBUILT:
        raise NotImplementedError(
            "--dataset_mode %s: this dataset front-end is not built in the MI355X package (SURVEY.md 8f-3). Use --dataset_mode "
            "singleskit / skit / patchskit for TouchClothing material folders or --dataset_mode synthetic for the seeded generator." % dataset_name)
    try:
        lib = importlib.import_module("data." + dataset_name + "_dataset")
    except ImportError as e:
        raise NotImplementedError("--dataset_mode %s: no module data/%s_dataset.py (%s)" % (dataset_name, dataset_name, e))
    target = dataset_name.replace("_", "") + "dataset"
    for name, cls in lib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type):
            return cls
    raise NotImplementedError("no dataset class matching %s in data/%s_dataset.py" % (target, dataset_name))


def get_option_setter(dataset_name):
    if dataset_name in _UNBUILT:
        return lambda parser, is_train: parser   # flags of an unbuilt front-end: none; creating the dataset raises
    return find_dataset_using_name(dataset_name).modify_commandline_options


def create_dataset(opt):
    return CustomDatasetDataLoader(opt).load_data()


class CustomDatasetDataLoader:
    def __init__(self, opt):
        self.opt = opt
        self.dataset = find_dataset_using_name(opt.dataset_mode)(opt)
        self.dataloader = torch.utils.data.DataLoader(
            self.dataset,
            batch_size=opt.batch_size,
            shuffle=not opt.serial_batches,
            num_workers=int(opt.num_threads),
            drop_last=bool(opt.isTrain),
            pin_memory=torch.cuda.is_available(),
        )

    def set_epoch(self, epoch):
        self.dataset.current_epoch = epoch

    def load_data(self):
        return self

    def __len__(self):
        return min(len(self.dataset), self.opt.max_dataset_size)

    def __iter__(self):
        for i, batch in enumerate(self.dataloader):
            if i * self.opt.batch_size >= self.opt.max_dataset_size:
                break
            yield batch
