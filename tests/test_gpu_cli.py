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
    # the reference's checkpoint files (tf.train.Checkpoint.write, UNet/train.py:184): a TensorBundle
    assert sorted(os.listdir(os.path.join(out, "checkpoint"))) == ["ckpt.data-00000-of-00001", "ckpt.index"]
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

    # the same image as a uint16 TIFF (the reference's data/ format): --image_format tif reads it and writes the mask as the reference's
    # imsave call asks (deflate BigTIFF in 1024-tiles, UNet/inference.py:221-222); same mask as the .npy run
    from PIL import Image
    timgs = tmp_path / "timgs"; timgs.mkdir()
    small16 = np.clip(small, 0, 65535).astype(np.uint16)
    Image.fromarray(small16).save(str(timgs / "small.tif"))
    np.save(imgs / "small16.npy", small16)
    inf.main(["--checkpoint_filepath", os.path.join(out, "checkpoint", "ckpt"), "--image_folder", str(timgs),
              "--output_folder", str(tmp_path / "tmasks"), "--number_classes", "2", "--number_channels", "1"])
    inf.main(["--checkpoint_filepath", os.path.join(out, "checkpoint", "ckpt"), "--image_folder", str(imgs),
              "--output_folder", str(tmp_path / "masks"), "--number_classes", "2", "--number_channels", "1", "--image_format", "npy"])
    raw = open(tmp_path / "tmasks" / "small.tif", "rb").read(4)
    assert raw == b"II+\x00"
    m_tif = np.array(Image.open(str(tmp_path / "tmasks" / "small.tif")))
    assert m_tif.dtype == np.uint8 and np.array_equal(m_tif, np.load(tmp_path / "masks" / "small16.npy"))

    # oracle on the small image: same z-score, reflect pad to x16, eval forward, argmax, crop
    tfc = pkg("tf_checkpoint")
    ck = tfc.read_bundle(os.path.join(out, "checkpoint", "ckpt"))
    prm = {eng_name: ck[stem + tfc.ATTR] for stem, eng_name, _ in tfc.variable_keys(pkg("engine").layer_table(1, 2))}
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


def test_checkpoint_roundtrip_restores_model_optimizer_and_legacy_files(tmp_path):
    # tf.train.Checkpoint(optimizer=..., model=...) semantics (reference UNet/train.py:96,184; UNet/model.py:81-83): after
    # restore the next train step is bit-identical to the one the saving model takes -- weights, BN moving statistics, both Adam
    # slots, the iteration counter (bias correction) and the learning rate all came back
    model = pkg("model")
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, 32, 32, generator=g)
    lab = torch.nn.functional.one_hot(torch.randint(0, 4, (2, 32, 32), generator=g), 4).to(torch.int32)
    a = model.UNet(4, 2, 3, learning_rate=1e-3, seed=5)
    for _ in range(3):
        a.train_step((img, lab, None, None))
    a.set_learning_rate(7e-4)
    stem = str(tmp_path / "checkpoint" / "ckpt")
    a.save_checkpoint(stem)
    keys = pkg("tf_checkpoint").list_bundle(stem)
    assert keys["model/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE"] == ("float32", (3, 3, 3, 64))
    assert keys["model/layer_with_weights-20/kernel/.ATTRIBUTES/VARIABLE_VALUE"] == ("float32", (2, 2, 512, 1024))     # Conv2DTranspose: [kh,kw,Cout,Cin]
    assert keys["model/layer_with_weights-45/moving_variance/.ATTRIBUTES/VARIABLE_VALUE"] == ("float32", (4,))
    assert keys["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] == ("int64", ())
    assert keys["model/layer_with_weights-44/bias/.OPTIMIZER_SLOT/optimizer/v/.ATTRIBUTES/VARIABLE_VALUE"] == ("float32", (4,))
    assert len(keys) == 138 + 184 + 5 + 1
    b = model.UNet(4, 2, 3, learning_rate=1e-3, seed=99)
    b.load_checkpoint(stem)
    assert b.engine.iterations == 3 and abs(b.get_learning_rate() - 7e-4) < 1e-9
    assert torch.equal(a.engine.theta, b.engine.theta) and torch.equal(a.engine.adam_m, b.engine.adam_m) and torch.equal(a.engine.adam_v, b.engine.adam_v)
    b.engine.dropout_seed = a.engine.dropout_seed
    la = float(a.train_step((img, lab, None, None)).numpy()); lb = float(b.train_step((img, lab, None, None)).numpy())
    assert la == lb and torch.equal(a.engine.theta, b.engine.theta)
    for k in a.engine.moving:
        assert torch.equal(a.engine.moving[k], b.engine.moving[k])
    # a model with other class / channel counts refuses the file with a shape message; a missing file says what was expected
    with pytest.raises(ValueError):
        model.UNet(2, 2, 3).load_checkpoint(stem)
    with pytest.raises(IOError):
        model.UNet(4, 2, 3).load_checkpoint(str(tmp_path / "nothing" / "ckpt"))
    # round-1 files (<stem>.npz) still load
    e = a.engine
    legacy = {"model/" + k: v for k, v in e.export_parameters().items()}
    legacy["optimizer/iterations"] = np.int64(e.iterations)
    np.savez(str(tmp_path / "old.npz"), **legacy)
    c = model.UNet(4, 2, 3, seed=1)
    c.load_checkpoint(str(tmp_path / "old"))
    assert torch.equal(c.engine.theta, a.engine.theta) and c.engine.iterations == e.iterations
