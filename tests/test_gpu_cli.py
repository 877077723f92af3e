"""GPU: the callers of the hot path with the reference's flags -- train loop outputs (reference UNet/train.py:173-184)
and the inference driver: reflect-pad to x16, whole-image vs 1024-tile + halo stitching (reference UNet/inference.py:27-227)."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import unet_torch as ot

pytestmark = pytest.mark.gpu


def test_train_cli_then_inference_cli(tmp_path):
    train = pkg("train")
    out = str(tmp_path / "out")
    train.main(["--output_dir", out, "--batch_size", "2", "--number_classes", "2", "--test_every_n_steps", "3",
                "--early_stopping", "1", "--synthetic", "64x64x1x8", "--max_epochs", "2"])
    losses = [float(v) for v in open(os.path.join(out, "test_loss.csv")).read().split()]
    assert len(losses) == 2 and all(np.isfinite(losses))
    assert os.path.exists(os.path.join(out, "checkpoint", "ckpt.npz"))
    assert any(d.startswith("tensorboard-") for d in os.listdir(out))

    # inference from that checkpoint: one small image whose size is not a multiple of 16, one image > 1024 px (tiled)
    imgs = tmp_path / "imgs"; imgs.mkdir()
    rng = np.random.default_rng(0)
    small = (rng.standard_normal((70, 83)) * 50 + 300).astype(np.float32)
    big = (rng.standard_normal((1090, 1040)) * 50 + 300).astype(np.float32)
    np.save(imgs / "small.npy", small); np.save(imgs / "big.npy", big)
    inf = pkg("inference")
    inf.main(["--checkpoint_filepath", os.path.join(out, "checkpoint", "ckpt"), "--image_folder", str(imgs),
              "--output_folder", str(tmp_path / "masks"), "--number_classes", "2", "--number_channels", "1",
              "--image_format", "npy"])
    m_small = np.load(tmp_path / "masks" / "small.npy")
    m_big = np.load(tmp_path / "masks" / "big.npy")
    assert m_small.shape == small.shape and m_small.dtype == np.uint8
    assert m_big.shape == big.shape and m_big.max() <= 1

    # oracle on the small image: same z-score, reflect pad to x16, eval forward, argmax, crop
    ck = np.load(os.path.join(out, "checkpoint", "ckpt.npz"))
    prm = {k[len("model/"):]: ck[k] for k in ck.files if k.startswith("model/")}
    ref = ot.TorchUNet(2, 1, 1, params=prm, dtype=torch.float64)
    z = (small - small.mean()) / small.std()
    zp = np.pad(z, ((0, (-70) % 16), (0, (-83) % 16)), mode="reflect")
    with torch.no_grad():
        sm = ref.forward(zp[None, None], False)[0].numpy()[0]
    gap = np.abs(sm[..., 0] - sm[..., 1])[:70, :83]
    ref_mask = np.argmax(sm, -1)[:70, :83]
    assert (m_small == ref_mask)[gap > 1e-4].all()

    # tiled path == whole-image path wherever the class margin is not within fp32 noise (halo >= receptive field)
    net = pkg("model").UNet(2, 1, 1)
    net.load_checkpoint(os.path.join(out, "checkpoint", "ckpt"))
    zb = ((big - big.mean()) / big.std()).astype(np.float32)[:, :, None]
    whole = inf._inference(zb, net)
    assert whole.shape == big.shape
    assert (whole == m_big).mean() > 0.9999


def test_train_cli_bf16(tmp_path):
    # the same loop in the mixed-precision mode (bf16 contractions, fp32 master weights): runs, writes the same outputs, and the
    # checkpoint (fp32 master weights) is readable by the fp32 inference path
    train = pkg("train")
    out = str(tmp_path / "out16")
    train.main(["--output_dir", out, "--batch_size", "2", "--number_classes", "3", "--test_every_n_steps", "3",
                "--early_stopping", "1", "--synthetic", "64x64x3x8", "--max_epochs", "2", "--compute_dtype", "bf16"])
    losses = [float(v) for v in open(os.path.join(out, "test_loss.csv")).read().split()]
    assert len(losses) == 2 and all(np.isfinite(losses))
    net = pkg("model").UNet(3, 1, 3)
    net.load_checkpoint(os.path.join(out, "checkpoint", "ckpt"))
    x = torch.randn(1, 3, 64, 64)
    p = net.get_keras_model()(x, training=False)
    assert tuple(p.shape) == (1, 64, 64, 3) and torch.isfinite(p).all()
