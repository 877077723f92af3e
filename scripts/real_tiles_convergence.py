"""Train on the reference's bundled tiles (tests/golden/data_tiles.npz: 12 train / 4 held-out, 256x256x1, 2 classes) with the device
feed + device augmentation, in both precisions, and report held-out accuracy / IoU at a few checkpoints."""
import importlib, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = lambda m: importlib.import_module("semantic-segmentation-unet_amd." + m)
readers, feed, aug, model = pkg("readers"), pkg("feed"), pkg("augment"), pkg("model")
d = np.load(os.path.join(ROOT, "tests", "golden", "data_tiles.npz"))
imgs, masks = d["images"], d["masks"]
tmp = tempfile.mkdtemp()
for i, (im, mk) in enumerate(zip(imgs[:12], masks[:12])):
    np.save(os.path.join(tmp, "t%02d.npy" % i), im); np.save(os.path.join(tmp, "t%02d_mask.npy" % i), mk)
dev = torch.device("cuda", 0)
test_x = torch.as_tensor(np.stack([readers.zscore_normalize(im[None].astype(np.float32)) for im in imgs[12:]]))
truth = masks[12:].astype(np.int64)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for cd in ("fp32", "bf16"):
    rd = readers.TileFolderReader(tmp, 2, shuffle=True, seed=0)
    pipe = aug.AugmentingFeed(
        feed.DeviceFeed(rd.batches(4, classmap=True, pin=False, raw=True), dev, classmap=True, number_classes=2, onehot=False),
        aug.DeviceAugmenter(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1, noise_augmentation_severity=0.02,
                            scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2, seed=0, device=dev), 2)
    net = model.UNet(2, 4, 1, learning_rate=1e-3, seed=0, compute_dtype=cd)
    for s in range(1, steps + 1):
        x, y = next(pipe)
        out = net.train_step((x, y, None, None))
        if s % 250 == 0:
            loss = float(out.numpy())
            pred = net.engine.argmax(net.engine.forward(test_x, training=False)).cpu().numpy()
            inter = ((pred == 1) & (truth == 1)).sum(); union = ((pred == 1) | (truth == 1)).sum()
            print("%s step %5d  train loss %.4f  held-out accuracy %.4f  IoU %.4f" % (cd, s, loss, (pred == truth).mean(), inter / max(union, 1)), flush=True)
    pipe.close()
