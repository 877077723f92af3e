"""GPU micro-benchmark: BatchNorm kernels (HBM-bound) at the BASELINE config-2 activation shapes; reports achieved TB/s."""
import ctypes, os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 8
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tot = [0.0, 0.0, 0.0]
for h, c in [(512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
    npx = B * h * h
    r = torch.randn(B, h, h, c, device="cuda"); dy = torch.randn_like(r); y = torch.empty_like(r); dz = torch.empty_like(r)
    g = torch.ones(c, device="cuda"); bt = torch.zeros(c, device="cuda")
    mean, invstd, scale, shift, dg, db, dbias = [torch.empty(c, device="cuda") for _ in range(7)]
    nb = L.unet_bn_workspace(npx, c); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    t0 = timeit(lambda: L.unet_bn_train_stats(P(r), c, npx, c, P(g), P(bt), 1e-3, 0.99, 1, None, None, P(mean), P(invstd), P(scale), P(shift), P(ws), nb, ST()))
    t1 = timeit(lambda: L.unet_bn_apply(P(r), c, P(scale), P(shift), P(y), c, npx, c, ST()))
    t2 = timeit(lambda: L.unet_bn_bwd(P(dy), c, P(r), c, P(g), P(mean), P(invstd), npx, c, 1, P(dz), c, P(dg), P(db), P(dbias), P(ws), nb, ST()))
    byts = npx * c * 4.0
    tot[0] += t0; tot[1] += t1; tot[2] += t2
    print("%4d^2 x %4d ch (%5.0f MB)  stats %6.3f ms %5.2f TB/s | apply %6.3f ms %5.2f TB/s | bwd (reduce+apply) %6.3f ms %5.2f TB/s"
          % (h, c, byts / 1e6, t0, byts / t0 / 1e9, t1, 2 * byts / t1 / 1e9, t2, 5 * byts / t2 / 1e9), flush=True)
print("TOTAL stats %.2f ms  apply %.2f ms  bwd %.2f ms" % tuple(tot))
# the same passes on bf16-stored tensors (r, y, dy, dz all bf16: 2 bytes per element)
tot = [0.0, 0.0]
for h, c in [(512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
    npx = B * h * h
    r = torch.randn(B, h, h, c, device="cuda").to(torch.bfloat16); dy = torch.randn(B, h, h, c, device="cuda").to(torch.bfloat16)
    y = torch.empty_like(r); dz = torch.empty_like(r)
    g = torch.ones(c, device="cuda"); bt = torch.zeros(c, device="cuda")
    mean, invstd, scale, shift, dg, db, dbias = [torch.rand(c, device="cuda") for _ in range(7)]
    nb = L.unet_bn_workspace(npx, c); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    t1 = timeit(lambda: L.unet_bn_apply_any(P(r), c, 1, P(scale), P(shift), P(y), c, 1, None, 0, None, B, h, h, c, ST()))
    t2 = timeit(lambda: L.unet_bn_bwd_any(P(dy), c, None, 0, None, B, h, h, P(r), c, P(g), P(mean), P(invstd), c, 1, P(dz), c, 1, P(dg), P(db), P(dbias),
                                          None, 0, P(ws), nb, ST(), 1, 1, 0, None))
    byts = npx * c * 2.0
    tot[0] += t1; tot[1] += t2
    print("bf16 %4d^2 x %4d ch (%5.0f MB)  apply %6.3f ms %5.2f TB/s | bwd (reduce+apply) %6.3f ms %5.2f TB/s"
          % (h, c, byts / 1e6, t1, 2 * byts / t1 / 1e9, t2, 5 * byts / t2 / 1e9), flush=True)
print("TOTAL bf16 apply %.2f ms  bwd %.2f ms" % tuple(tot))
# the form 16 of 23 layers take in the mixed-precision step: sums from a data-gradient epilogue (finalize + ONE apply pass), all bf16
tot = 0.0
for h, c in [(512, 64), (256, 128), (128, 256), (64, 512), (32, 1024)]:
    npx = B * h * h
    r = torch.randn(B, h, h, c, device="cuda").to(torch.bfloat16); dy = torch.randn(B, h, h, c, device="cuda").to(torch.bfloat16)
    dz = torch.empty_like(r)
    g = torch.ones(c, device="cuda"); mean, invstd, dg, db, dbias = [torch.rand(c, device="cuda") for _ in range(5)]
    rows = 64
    part = torch.rand((c // 64) * rows * 128, device="cuda")
    nb = L.unet_bn_workspace(npx, c); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    t2 = timeit(lambda: L.unet_bn_bwd_any(P(dy), c, None, 0, None, B, h, h, P(r), c, P(g), P(mean), P(invstd), c, 1, P(dz), c, 1, P(dg), P(db), P(dbias),
                                          P(part), rows, P(ws), nb, ST(), 1, 1, 0, None))
    byts = npx * c * 2.0
    tot += t2
    print("bf16 %4d^2 x %4d ch  bwd from partial sums (finalize + apply + colsum) %6.3f ms %5.2f TB/s" % (h, c, t2, 3 * byts / t2 / 1e9), flush=True)
print("TOTAL bf16 bwd-from-partials %.2f ms" % tot)
