"""Import harness for the upstream reference (TEST INFRASTRUCTURE ONLY).

Used exclusively by oracle/make_golden.py, in the build container, to run the
reference's own Python on CPU and dump golden vectors.  /root/reference never
travels to the GPU box, so nothing under tests/ (-m gpu), bench.py or smoke()
may import this module.

The reference imports several packages that are absent from this image
(SURVEY.md §8c).  They are replaced by empty stub modules; the only stubs with
behaviour are `lpips.LPIPS` (a zero-weight surrogate: the LPIPS terms are
disabled with lambda=0 in every golden vector) and `util.str2bool` users.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("VTS_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "tkinter", "turtle", "cv2", "lpips", "clip", "vision_aided_loss",
    "torchvision", "torchvision.models", "torchvision.models.resnet",
    "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
    "torchmetrics", "torchmetrics.functional", "dominate", "dominate.tags",
    "wandb", "visdom", "skimage", "skimage.metrics", "skimage.transform", "skimage.io",
    "GPUtil", "imageio", "OpenEXR", "Imath", "kornia",
]


def _install_stubs():
    import torch

    for name in _STUBS:
        if name in sys.modules:
            continue
        try:
            __import__(name)
            continue
        except Exception:
            pass
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
    # attributes the reference touches at import time
    sys.modules["tkinter"].N = None
    sys.modules["turtle"].forward = None
    sys.modules["torchvision.models.resnet"].model_urls = {}
    tvm = sys.modules["torchvision.models"]
    for n in ("resnet18", "resnet34", "resnet50", "vgg19", "inception_v3"):
        if not hasattr(tvm, n):
            setattr(tvm, n, lambda *a, **k: None)
    sys.modules["torchvision"].models = tvm
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

    class _LPIPS(torch.nn.Module):
        """Surrogate: returns zeros [N,1,1,1]; golden vectors use lambda_lpips=0."""

        def __init__(self, net="vgg", **kw):
            super().__init__()

        def forward(self, a, b, **kw):
            return (a - b).abs().mean(dim=(1, 2, 3), keepdim=True) * 0.0

    sys.modules["lpips"].LPIPS = _LPIPS

    sys.modules["torchmetrics"].MeanSquaredError = type("MeanSquaredError", (), {})
    tmf = sys.modules["torchmetrics.functional"]
    for n in ("peak_signal_noise_ratio", "structural_similarity_index_measure"):
        if not hasattr(tmf, n):
            setattr(tmf, n, lambda *a, **k: torch.tensor(0.0))


def load():
    """Put the reference on sys.path (front) with stubs installed."""
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return REF_ROOT


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))
