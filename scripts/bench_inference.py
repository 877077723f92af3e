"""GPU micro-benchmark: the inference forward + argmax (reference UNet/inference.py:101-107,159-166) on one 1024 x 1024 tile, the
driver's tile size, in both precisions.  Prints images (tiles) per second and the agreement of the two arg-max masks."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
model = importlib.import_module("semantic-segmentation-unet_amd.model")
C, K, HW = int(os.environ.get("C", 1)), int(os.environ.get("K", 2)), int(os.environ.get("HW", 1024))
x = torch.randn(1, C, HW, HW, generator=torch.Generator().manual_seed(0)).cuda()
masks = {}
for cd in ("fp32", "bf16"):
    net = model.UNet(K, 1, C, 1e-4, seed=0, compute_dtype=cd)
    e = net.engine
    for _ in range(3):
        m = e.argmax(e.forward(x, training=False))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        m = e.argmax(e.forward(x, training=False))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    masks[cd] = m.clone()
    print("%s: %.2f ms per %dx%dx%d tile (%d classes) = %.1f tiles/s, %.1f Mpixel/s" % (cd, dt * 1e3, HW, HW, C, K, 1 / dt, HW * HW / dt / 1e6), flush=True)
print("arg-max agreement bf16 vs fp32 (random-init weights): %.4f" % (masks["fp32"] == masks["bf16"]).float().mean().item())
