"""Golden-fixture tests.  tests/golden/unet_c1k2_32.npz (made by tests/golden/make_golden.py from two real tiles of the
reference's bundled data/ and THIS repo's fp64 oracle -- parity vs TensorFlow is unpinned) pins:
  * CPU: the oracle in fp32 (numpy) and the independent torch restatement against the frozen fp64 snapshot;
  * GPU: the HIP path behind the UNet class against the same snapshot (argmax mask bit-exact)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg
from oracle import unet_numpy as on
from oracle import unet_torch as ot

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import golden_params      # noqa: E402  (seeded weights are regenerated, not stored)

G = np.load(os.path.join(ROOT, "tests", "golden", "unet_c1k2_32.npz"))
MASKS = {"drop_4": G["drop_4"], "drop_b": G["drop_b"]}


def _check_grads(g, rtol):
    for key in G.files:
        if key.startswith("grad/"):
            ref = G[key]
            a = np.asarray(g[key[5:]], dtype=np.float64)
            assert np.linalg.norm(a - ref) <= rtol * max(np.linalg.norm(ref), 1e-6 * np.sqrt(ref.size)), key
        elif key.startswith("gradnorm/"):
            a = np.asarray(g[key[9:]], dtype=np.float64)
            assert abs(np.linalg.norm(a) - float(G[key])) <= rtol * float(G[key]), key


def test_fixture_is_real_data_with_both_classes():
    assert G["images"].shape == (2, 1, 32, 32) and G["labels"].shape == (2, 32, 32, 2)
    assert 0.3 < G["labels"][..., 1].mean() < 0.6
    assert abs(G["images"][0].mean()) < 1e-4 and abs(G["images"][0].std() - 1) < 1e-3     # z-scored per tile
    assert (G["labels"].sum(-1) == 1).all()


def test_torch_restatement_fp64_reproduces_golden():
    t = ot.TorchUNet(2, 2, 1, params=golden_params(), dtype=torch.float64)
    with torch.no_grad():
        sm = t.forward(G["images"], False)[0].numpy()
    assert np.abs(sm - G["softmax_eval"]).max() < 1e-10
    assert np.array_equal(np.argmax(sm, -1), G["mask_eval"])
    loss, smt, g, _ = t.loss_and_grads(G["images"], G["labels"], MASKS)
    assert abs(float(loss) - float(G["loss_train"])) < 1e-11
    _check_grads({k: v.numpy() for k, v in g.items()}, 1e-8)


def test_numpy_oracle_fp32_close_to_golden():
    o = on.OracleUNet(2, 2, 1, params=golden_params(), dtype=np.float32)
    sm, _ = o.forward(G["images"], training=False)
    assert np.abs(sm - G["softmax_eval"]).max() < 5e-5
    gap = np.abs(G["softmax_eval"][..., 0] - G["softmax_eval"][..., 1])
    assert (np.argmax(sm, -1) == G["mask_eval"])[gap > 1e-4].all()
    le, _ = o.test_step(G["images"], G["labels"])
    assert abs(le - float(G["loss_eval"])) < 1e-5


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    net = pkg("model").UNet(2, 2, 1)
    net.engine.load_parameters(golden_params())
    sm = net.get_keras_model()(G["images"])
    assert np.abs(sm - G["softmax_eval"]).max() < 2e-5                 # stated fp32 forward tolerance
    assert np.array_equal(np.argmax(sm, -1), G["mask_eval"])           # argmax mask bit-exact
    mask = net.engine.argmax(net.engine.forward(torch.as_tensor(G["images"]))).cpu().numpy()
    assert np.array_equal(mask, G["mask_eval"])
    lm, am = pkg("model").Mean(), pkg("model").CategoricalAccuracy()
    le = net.test_step((G["images"], G["labels"], lm, am)).numpy()
    assert abs(le - float(G["loss_eval"])) < 1e-5 * float(G["loss_eval"])
    assert float(am.result()) == pytest.approx((G["mask_eval"] == np.argmax(G["labels"], -1)).mean(), abs=1e-6)
    e = net.engine
    e.forward(torch.as_tensor(G["images"]), training=True, dropout_masks=MASKS, labels=torch.as_tensor(G["labels"]),
              global_batch_size=2, want_grad=True)
    e.backward()
    assert abs(e.loss_buf[0].item() - float(G["loss_train"])) < 1e-5 * float(G["loss_train"])
    # training-mode forward: batch statistics over as few as 8 samples (2x2 bottleneck, N=2) amplify fp32 rounding
    assert np.abs(e.bufs["softmax"].cpu().numpy() - G["softmax_train"]).max() < 1e-4
    _check_grads(e.export_gradients(), 5e-2)        # fp32 ReLU-mask flips bound this (see test_gpu_unet.grad_errors)
