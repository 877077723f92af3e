"""Training-loop throughput with a LIVE reader against inputs resident in HBM (what bench.py times): BASELINE config 2 (fp32) and config 4's
per-GPU workload (bf16), the path train.py takes by default -- reader threads -> pinned staging ring -> copy stream (feed.DeviceFeed, raw
tiles + uint8 class maps) -> device augmentation (augment.DeviceAugmenter, the reference's default settings) -> z-score + one-hot on the
device -> train step -- with 1 / 2 / 4 / 6 reader threads.  python scripts/bench_feed.py [steps] > profiles/rNN_feed.txt"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
P = "semantic-segmentation-unet_amd."
m, readers, feed, aug = (importlib.import_module(P + n) for n in ("model", "readers", "feed", "augment"))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
print("# live-loop images/s vs resident inputs, batch 8, 512x512, %d timed steps after 5 warm-up steps; host: %d usable cores" % (steps, len(os.sched_getaffinity(0))))
for dtype, C, K in (("fp32", 1, 2), ("bf16", 3, 4)):
    B, S = 8, 512
    net = m.UNet(K, B, C, 3e-4, device=dev, compute_dtype=dtype)

    def run(nxt):
        for _ in range(5):
            net.train_step(nxt() + (None, None))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            net.train_step(nxt() + (None, None))
        torch.cuda.synchronize()
        return B * steps / (time.perf_counter() - t0)
    img0, lab0 = next(readers.SyntheticReader(64, S, S, C, K, seed=1).batches(B))
    img0, lab0 = img0.cuda(), lab0.cuda()
    res = run(lambda: (img0, lab0))
    print("%s %dx%dx%d/%d classes: resident inputs %7.1f images/s" % (dtype, S, S, C, K, res), flush=True)
    for W in (1, 2, 4, 6):
        its = [readers.SyntheticReader(64, S, S, C, K, seed=10 + w).batches(B, classmap=True, pin=False, raw=True) for w in range(W)]
        raw = feed.DeviceFeed(its, dev, classmap=True, number_classes=K, onehot=False)
        pipe = aug.AugmentingFeed(raw, aug.DeviceAugmenter(rotation_flag=True, reflection_flag=True, jitter_augmentation_severity=0.1,
                                                          noise_augmentation_severity=0.02, scale_augmentation_severity=0.1,
                                                          blur_augmentation_max_sigma=2, seed=0, device=dev), K)
        v = run(lambda: tuple(next(pipe)))
        pipe.close()
        print("%s   live, %d reader thread%s + device augmentation: %7.1f images/s = %.3f of resident" % (dtype, W, "" if W == 1 else "s", v, v / res), flush=True)
    del net
    torch.cuda.empty_cache()
