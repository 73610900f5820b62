import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "visual-tactile-synthesis_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is the checker: keep it on <= 8 threads.  With the 128+ threads of a GPU host PyTorch-CPU's
    # convolution weight gradient drifts by up to 5e-3 from its own single-thread result (measured on the
    # 1024-pixel D layers; the HIP path agrees with the single-thread oracle to 3e-6), which is not the
    # algorithm under test.
    import torch

    torch.set_num_threads(min(8, os.cpu_count() or 1))


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
