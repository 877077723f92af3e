import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "semantic-segmentation-unet_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    """The package directory name has hyphens, so it is imported by string."""
    return importlib.import_module(PKG + ("." + sub if sub else ""))


def have_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def hip():
    """C-ABI library handle; GPU tests fail (not skip) if the extension is missing on a GPU box."""
    if not have_gpu():
        pytest.skip("no GPU in this container")
    return pkg("_lib").lib()
