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
    losses = []
    for _ in range(500):           # BN moving statistics (momentum 0.99) need a few hundred steps before eval mode is meaningful
        x, y = next(pipe)
        losses.append(float(net.train_step((x, y, None, None)).numpy()))
    pipe.close()
    assert all(np.isfinite(losses)) and np.mean(losses[-10:]) < 0.6 * np.mean(losses[:5])

    # held-out tiles, eval mode (moving statistics), argmax on the device: beat the all-background guess by a wide margin
    test_x = np.stack([readers.zscore_normalize(im[None].astype(np.float32)) for im in imgs[12:]])
    prob = net.engine.forward(torch.as_tensor(test_x), training=False)
    pred = net.engine.argmax(prob).cpu().numpy()
    truth = masks[12:].astype(np.int64)
    acc = (pred == truth).mean()
    inter = ((pred == 1) & (truth == 1)).sum(); union = ((pred == 1) | (truth == 1)).sum()
    print("%s: held-out pixel accuracy %.4f, IoU %.4f, background fraction %.4f" % (compute_dtype, acc, inter / max(union, 1), (truth == 0).mean()))
    assert acc > max(0.9, (truth == 0).mean() + 0.05), acc
    assert inter / max(union, 1) > 0.6, inter / max(union, 1)
