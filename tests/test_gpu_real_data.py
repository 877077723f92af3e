"""GPU: the whole training path on the reference's own bundled tiles (BASELINE config 1's data: 256x256, 1 channel, 2 classes).
Reader -> device feed -> device augmentation -> z-score -> one-hot -> train steps, then eval on held-out tiles."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
FIX = os.path.join(os.path.dirname(__file__), "golden", "data_tiles.npz")


def _write_tiles(folder, imgs, masks):
    os.makedirs(folder, exist_ok=True)
    for i, (im, mk) in enumerate(zip(imgs, masks)):
        np.save(os.path.join(folder, "t%02d.npy" % i), im)
        np.save(os.path.join(folder, "t%02d_mask.npy" % i), mk)


@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
def test_learns_real_tiles_with_device_augmentation(tmp_path, compute_dtype):
    # (bf16: the mixed-precision mode must train the reference's own tiles to the same held-out quality bar)
    d = np.load(FIX)
    imgs, masks = d["images"], d["masks"]
    _write_tiles(tmp_path / "train", imgs[:12], masks[:12])
    readers, feed, aug, model = pkg("readers"), pkg("feed"), pkg("augment"), pkg("model")
    dev = torch.device("cuda", 0)
    rd = readers.TileFolderReader(str(tmp_path / "train"), 2, shuffle=True, seed=0)
    pipe = aug.AugmentingFeed(
        feed.DeviceFeed(rd.batches(4, classmap=True, pin=False, raw=True), dev, classmap=True, number_classes=2, onehot=False),
        aug.DeviceAugmenter(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1, noise_augmentation_severity=0.02,
                            scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2, seed=0, device=dev), 2)
    net = model.UNet(2, 4, 1, learning_rate=1e-3, seed=0, compute_dtype=compute_dtype)
    # held-out tiles, eval mode (moving statistics), argmax on the device: beat the all-background guess by a wide margin.
    # Judged on FOUR evaluations (steps 400 .. 700), not one: at lr 1e-3 the eval-mode accuracy of a single step swings between 0.89 and 0.98 on
    # every route and seed (scripts/real_tiles_spread.py -> profiles/r06_real_tiles_spread.txt: native fp32 reads 0.887 at step 600 of one
    # seed), so a one-point threshold tests the rounding-level chaos of the trajectory, not the arithmetic -- the round-6 change of the
    # BF16x6 split moved step 500 of this seed from above to 0.895.  The median of the four is > 0.93 in all twelve measured runs.
    test_x = torch.as_tensor(np.stack([readers.zscore_normalize(im[None].astype(np.float32)) for im in imgs[12:]]))
    truth = masks[12:].astype(np.int64)
    losses, accs, ious = [], [], []
    for it in range(1, 701):       # BN moving statistics (momentum 0.99) need a few hundred steps before eval mode is meaningful
        x, y = next(pipe)
        losses.append(float(net.train_step((x, y, None, None)).numpy()))
        if it in (400, 500, 600, 700):
            pred = net.engine.argmax(net.engine.forward(test_x, training=False)).cpu().numpy()
            accs.append((pred == truth).mean())
            inter = ((pred == 1) & (truth == 1)).sum(); union = ((pred == 1) | (truth == 1)).sum()
            ious.append(inter / max(union, 1))
    pipe.close()
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < 0.6 * np.mean(losses[:5])
    print("%s: held-out pixel accuracy %s, IoU %s, background fraction %.4f" % (compute_dtype, ["%.4f" % a for a in accs], ["%.4f" % a for a in ious], (truth == 0).mean()))
    assert np.median(accs) > max(0.92, (truth == 0).mean() + 0.05) and max(accs) > 0.95, accs
    assert np.median(ious) > 0.6, ious


def test_config1_exact_form_bundled_tiles_batch2_through_the_cli(tmp_path):
    """BASELINE config 1 in its exact form: the reference's bundled tiles (256x256, 1 channel, uint16, 2 classes) through the train CLI at
    --batch_size 2 with the reference's default flags (augmentation on, one reader), then the inference CLI on the held-out tiles as uint16
    TIFFs from the checkpoint the run wrote.  (The LMDB container itself is out of scope -- `lmdb` is not installable here; the folder store
    of readers.TileFolderReader holds the same tiles.)  Plumbing, not throughput: files, shapes, dtypes, finite losses, a mask per image."""
    from PIL import Image
    d = np.load(FIX)
    imgs, masks = d["images"], d["masks"]
    _write_tiles(tmp_path / "train", imgs[:12], masks[:12])
    _write_tiles(tmp_path / "test", imgs[12:], masks[12:])
    train, inf = pkg("train"), pkg("inference")
    out = str(tmp_path / "out")
    train.main(["--train_database", str(tmp_path / "train"), "--test_database", str(tmp_path / "test"), "--output_dir", out,
                "--batch_size", "2", "--number_classes", "2", "--test_every_n_steps", "6", "--early_stopping", "1", "--max_epochs", "2"])
    losses = [float(v) for v in open(os.path.join(out, "test_loss.csv")).read().split()]
    assert len(losses) == 2 and all(np.isfinite(losses))
    assert sorted(os.listdir(os.path.join(out, "checkpoint"))) == ["ckpt.data-00000-of-00001", "ckpt.index"]
    tb = [e for e in os.listdir(out) if e.startswith("tensorboard-")]
    assert len(tb) == 1 and sorted(os.listdir(os.path.join(out, tb[0]))) == ["test", "train"]
    # 7 + 7 optimizer steps (N + 1 per epoch, UNet/train.py:137-138), each logged once per tag; 4 held-out tiles / batch 2 -> 3 test steps per epoch
    rows = [l for l in open(os.path.join(out, tb[0], "train", "scalars.jsonl"))]
    assert len(rows) == 2 * 14
    folder = tmp_path / "images"; folder.mkdir()
    for i, im in enumerate(imgs[12:]):
        Image.fromarray(im).save(str(folder / ("tile%02d.tif" % i)))
    inf.main(["--checkpoint_filepath", os.path.join(out, "checkpoint", "ckpt"), "--image_folder", str(folder),
              "--output_folder", str(tmp_path / "masks"), "--number_classes", "2", "--number_channels", "1"])
    names = sorted(os.listdir(tmp_path / "masks"))
    assert names == ["tile%02d.tif" % i for i in range(4)]
    for nm in names:
        m = np.array(Image.open(str(tmp_path / "masks" / nm)))
        assert m.shape == (256, 256) and m.dtype == np.uint8 and m.max() <= 1
