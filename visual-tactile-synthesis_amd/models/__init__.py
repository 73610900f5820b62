"""Model factory (mirror of /root/reference/models/__init__.py:25-67).

`create_model(opt)` is the drop-in boundary: train.py / test.py only ever touch the object
it returns through the BaseModel method set.
"""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    modellib = importlib.import_module("." + model_name + "_model", __package__)
    target = model_name.replace("_", "") + "model"
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            return cls
    raise NotImplementedError(
        "In models/%s_model.py there should be a subclass of BaseModel whose lower-case name is %s" % (model_name, target))


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    instance = find_model_using_name(opt.model)(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
