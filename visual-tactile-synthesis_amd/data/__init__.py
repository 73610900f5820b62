"""Dataset factory (mirror of /root/reference/data/__init__.py:17-104).

Only a synthetic batch generator is built: the reference's CPU dataset front-end
(PNG/npz loading, crop/zoom augmentation, patch bookkeeping) is SURVEY.md §8(f)
row 3 and the TouchClothing data is not available offline.  `singleskit` and
`skit` therefore resolve to the synthetic dataset, which honours the same
post-collate batch-dict contract.
"""
import importlib

import torch.utils.data

_ALIASES = {"singleskit": "synthetic", "skit": "synthetic", "patchskit": "synthetic", "aligned": "synthetic"}


def find_dataset_using_name(dataset_name):
    resolved = _ALIASES.get(dataset_name, dataset_name)
    lib = importlib.import_module("data." + resolved + "_dataset")
    target = resolved.replace("_", "") + "dataset"
    for name, cls in lib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type):
            return cls
    raise NotImplementedError("no dataset class matching %s in data/%s_dataset.py" % (target, resolved))


def get_option_setter(dataset_name):
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
