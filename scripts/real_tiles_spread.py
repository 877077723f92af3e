"""Diagnostic: spread of the held-out accuracy of tests/test_gpu_real_data.py's 500-step training over network seeds and fp32 routes
(the test asserts > 0.9 on ONE trajectory; rounding-level changes of the arithmetic move that trajectory).  usage: real_tiles_spread.py [steps=500]"""
import importlib, os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = lambda m: importlib.import_module("semantic-segmentation-unet_amd." + m)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
d = np.load(os.path.join(ROOT, "tests", "golden", "data_tiles.npz"))
imgs, masks = d["images"], d["masks"]
readers, feed, aug, model, plan = pkg("readers"), pkg("feed"), pkg("augment"), pkg("model"), pkg("plan")
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
for i, (im, mk) in enumerate(zip(imgs[:12], masks[:12])):
    np.save(os.path.join(tmp, "t%02d.npy" % i), im); np.save(os.path.join(tmp, "t%02d_mask.npy" % i), mk)
test_x = torch.as_tensor(np.stack([readers.zscore_normalize(im[None].astype(np.float32)) for im in imgs[12:]]))
truth = masks[12:].astype(np.int64)
for route in ("bf16x6", "native"):
    for seed in (0, 1, 2):
        rd = readers.TileFolderReader(tmp, 2, shuffle=True, seed=0)
        pipe = aug.AugmentingFeed(
            feed.DeviceFeed(rd.batches(4, classmap=True, pin=False, raw=True), dev, classmap=True, number_classes=2, onehot=False),
            aug.DeviceAugmenter(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1, noise_augmentation_severity=0.02,
                                scale_augmentation_severity=0.1, blur_augmentation_max_sigma=2, seed=0, device=dev), 2)
        net = model.UNet(2, 4, 1, learning_rate=1e-3, seed=seed, compute_dtype="fp32")
        net.engine.opt.fp32_matrix = route
        accs = []
        for it in range(1, steps + 201):
            x, y = next(pipe)
            net.train_step((x, y, None, None))
            if it in (steps - 200, steps - 100, steps, steps + 100, steps + 200):
                pred = net.engine.argmax(net.engine.forward(test_x, training=False)).cpu().numpy()
                accs.append("%d: %.4f" % (it, (pred == truth).mean()))
        pipe.close()
        print("%-7s net seed %d  held-out accuracy at step %s" % (route, seed, "  ".join(accs)), flush=True)
