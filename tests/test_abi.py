"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/unet_hip.h declares (no compute is attempted without a GPU); the host class keeps the reference's surface."""
import ctypes
import inspect
import os
import re

import pytest

from conftest import ROOT, pkg


def test_library_builds_and_exports_every_declared_symbol():
    build = pkg("_build")
    path = build.build_library()
    assert os.path.exists(path)
    cdll = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "unet_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = set(re.findall(r"\b(unet_\w+)\s*\(", header))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(cdll, n), "library does not export " + n
    protos = pkg("_lib").parse_header()
    assert set(protos) == names
    assert cdll.unet_hip_abi_version() == 1
    assert cdll.unet_conv3x3_mfma_supported(64, 128) == 1 and cdll.unet_conv3x3_mfma_supported(1, 64) == 0


def test_unet_class_keeps_reference_surface():
    model = pkg("model")
    U = model.UNet
    sig = inspect.signature(U.__init__)
    assert list(sig.parameters)[:6] == ["self", "number_classes", "global_batch_size", "number_channels",
                                        "learning_rate", "label_smoothing"]
    assert sig.parameters["learning_rate"].default == 3e-4 and sig.parameters["label_smoothing"].default == 0
    assert U.SIZE_FACTOR == 16 and U.RADIUS == 96
    for m in ("load_checkpoint", "get_keras_model", "get_optimizer", "set_learning_rate", "get_learning_rate",
              "estimate_radius", "train_step", "dist_train_step", "test_step", "dist_test_step"):
        assert callable(getattr(U, m)), m


def test_product_path_refuses_cpu():
    model = pkg("model")
    with pytest.raises(RuntimeError):
        model.UNet(2, 2, 1, device="cpu")


def test_product_path_never_imports_the_oracle():
    pdir = os.path.join(ROOT, "semantic-segmentation-unet_amd")
    for fn in os.listdir(pdir):
        if fn.endswith(".py"):
            src = open(os.path.join(pdir, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), fn
