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
    lib_mod = pkg("_lib")
    assert cdll.unet_hip_abi_version() == lib_mod.header_abi_version() == 9
    assert cdll.unet_conv3x3_mfma_supported(64, 128) == 1 and cdll.unet_conv3x3_mfma_supported(1, 64) == 0


def test_loader_refuses_a_library_of_another_abi_version(tmp_path, monkeypatch):
    # functions are bound by NAME: a diagnostic (UNET_HIP_LIB) or stale library built against another header would take arguments in
    # the wrong slots.  The version check runs even where the content-stamp check is skipped.
    import subprocess
    lib_mod = pkg("_lib")
    src = tmp_path / "fake.c"
    src.write_text("int unet_hip_abi_version(void) { return 1; }\n")
    so = tmp_path / "libfake.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(lib_mod, "LIB_PATH", str(so))
    monkeypatch.setenv("UNET_HIP_LIB", str(so))
    with pytest.raises(lib_mod.UnetHipError, match="ABI version"):
        lib_mod._Lib()


def test_library_reads_no_environment():
    # include/unet_hip.h: "no global mutable state and no environment variables: every option is an argument"
    cdir = os.path.join(ROOT, "semantic-segmentation-unet_amd", "csrc")
    for fn in sorted(os.listdir(cdir)):
        if fn.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(cdir, fn)).read(), fn


def test_step_plan_states_routes_and_storage_at_any_size():
    # plan.build_plan needs the library's shape predicates only (host code): the plan of BASELINE config 4 (512x512x3, 4 classes, batch 8,
    # bf16) and of a tiny tile are the SAME contract, and equal the checker's default rounding plan field by field; config 2 / 5 (fp32) take
    # the fused Winograd kernels with BatchNorm-apply on load for 13 layers
    from oracle import unet_numpy as on
    plan = pkg("plan")
    L = pkg("_lib").lib()
    want = on.Bf16Plan.default()
    for shp in ((8, 512, 512), (2, 64, 64), (1, 16, 176), (3, 48, 80)):
        pl = plan.build_plan(plan.EngineOptions(compute_dtype="bf16"), 3, 4, *shp, True, True, L)
        got = pl.rounding_points()
        assert all(got[k] == getattr(want, k) for k in got), {k: got[k] ^ getattr(want, k) for k in got if isinstance(got[k], frozenset) and got[k] != getattr(want, k)}
        assert all(p.fwd in ("bf16", "convt_bf16") for n, p in pl.layer.items() if n not in ("conv_1a", "logits"))
        assert "dec_3a" in pl.describe()
    ev = plan.build_plan(plan.EngineOptions(compute_dtype="bf16"), 3, 4, 1, 64, 64, False, False, L)      # inference: nothing but operands is rounded
    assert not any(p.r == plan.BF16 or p.dz == plan.BF16 for p in ev.layer.values()) and ev.layer["dec_1b"].y == plan.F32
    for c, k, shp in ((1, 2, (8, 512, 512)), (3, 6, (2, 1024, 1024))):
        pl = plan.build_plan(plan.EngineOptions(), c, k, *shp, True, True, L)
        assert sum(p.defer_y for p in pl.layer.values()) == 13 and sum(p.x_on_load for p in pl.layer.values()) == 13
        assert all(p.fwd == p.dgrad == p.wgrad == "winograd" for n, p in pl.layer.items() if p.kind == "conv3" and n != "conv_1a")
        # ... and their forward / data-gradient products run on the bf16 matrix pipe at fp32 grade (BF16x6) by default, natively on request
        assert all(p.fwd_x6 and p.dgrad_x6 for n, p in pl.layer.items() if p.kind == "conv3" and n != "conv_1a") and "winograd/x6" in pl.describe()
        nat = plan.build_plan(plan.EngineOptions(fp32_matrix="native"), c, k, *shp, True, True, L)
        assert not any(p.fwd_x6 or p.dgrad_x6 for p in nat.layer.values())
        assert not any(v == plan.BF16 for p in pl.layer.values() for v in (p.r, p.y, p.dz, p.dx))
    # a bf16 operand beyond 2 GiB (1024x1024, batch 4: dec_1a's concat input) falls back to the fp32 kernels for that ONE layer
    big = plan.build_plan(plan.EngineOptions(compute_dtype="bf16"), 3, 6, 4, 1024, 1024, True, True, L)
    assert [n for n, p in big.layer.items() if p.kind == "conv3" and p.fwd == "winograd"] == ["dec_1a"]


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
