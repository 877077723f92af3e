import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "semantic-segmentation-unet_amd"

# The torch restatements the full-size GPU tests compare against run torch's own convolutions (MIOpen).  On a fresh box MIOpen's default
# find mode benchmarks and compiles candidates for every new shape -- minutes for the ~70 shapes of a U-Net step, none of it ours.  FAST
# picks a solver from its heuristics instead; the results are the same fp32 convolutions.  (Only the checker is affected: the product
# path never calls MIOpen.)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


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
