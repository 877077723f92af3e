"""How many CUs does an HBM-bound BatchNorm pass need?  Runs unet_bn_apply (reads 537 MB, writes 537 MB) on HIP streams created with
hipExtStreamCreateWithCUMask for several CU counts / placements, and the fused Winograd weight gradient on the complementary mask at the
same time.  Diagnostic for the two-stream backward schedule."""
import ctypes, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
hip = ctypes.CDLL(None)
try:
    hip.hipExtStreamCreateWithCUMask
except AttributeError:
    hip = ctypes.CDLL([m.split()[-1] for m in open("/proc/self/maps") if "libamdhip64" in m][0])
P = lambda t: ctypes.c_void_p(t.data_ptr())

def masked_stream(cus):
    words = (ctypes.c_uint32 * 8)(*[0] * 8)
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)

def timeit(fn, stream, reps=10):
    with torch.cuda.stream(stream):
        fn(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

n, h, w, c = 8, 512, 512, 64
r = torch.randn(n, h, w, c, device="cuda"); y = torch.empty_like(r)
sc = torch.rand(c, device="cuda") + 0.5; sh = torch.randn(c, device="cuda")
gb = 2 * r.numel() * 4 / 1e9
def bn(stream):
    return lambda: L.unet_bn_apply(P(r), c, P(sc), P(sh), P(y), c, n * h * w, c, ctypes.c_void_p(stream.cuda_stream))
full = torch.cuda.current_stream()
t = timeit(bn(full), full)
print("all 256 CUs (default stream): %.3f ms  %.2f TB/s" % (t, gb / t), flush=True)
for name, cus in [("256 via mask", range(256)), ("128 strided (every 2nd)", range(0, 256, 2)), ("128 contiguous", range(128)),
                  ("64 strided (every 4th)", range(0, 256, 4)), ("64 contiguous", range(64)), ("32 strided (every 8th)", range(0, 256, 8)),
                  ("32 contiguous", range(32)), ("96 strided", [i for i in range(256) if i % 8 < 3])]:
    s = masked_stream(list(cus))
    t = timeit(bn(s), s)
    print("%-26s %.3f ms  %.2f TB/s" % (name, t, gb / t), flush=True)

# ---- the same BatchNorm pass while the fused Winograd weight gradient (MFMA-bound, one 512-register workgroup per CU) runs beside it
ci = co = 128; hh = 256
x = torch.randn(n, hh, hh, ci, device="cuda"); dz = torch.randn(n, hh, hh, co, device="cuda"); dw = torch.empty(3, 3, ci, co, device="cuda")
nb = L.unet_conv3x3_wgrad_winograd_fused_workspace(n, hh, hh, ci, co, 0); ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
def wg(stream):
    return lambda: L.unet_conv3x3_wgrad_winograd_fused(P(x), ci, P(dz), co, P(dw), n, hh, hh, ci, co, 0, P(ws), nb, ctypes.c_void_p(stream.cuda_stream))
tw = timeit(wg(full), full)
print("weight gradient 128->128 @256^2 alone: %.3f ms (grid per UNET_WGRAD_CUS)" % tw, flush=True)
A = [i for i in range(256) if i % 8 == 0]; B = [i for i in range(256) if i % 8 != 0]
for name, s_w, s_b in [("wgrad default stream, BN plain second stream", full, torch.cuda.Stream()),
                       ("wgrad masked to 224 CUs (all but every 8th), BN masked to the other 32", masked_stream(B), masked_stream(A)),
                       ("wgrad masked to 224 CUs, BN plain second stream", masked_stream(B), torch.cuda.Stream())]:
    torch.cuda.synchronize()
    with torch.cuda.stream(s_w):
        for _ in range(12): wg(s_w)()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s_b):
        bn(s_b)(); e0.record()
        for _ in range(10): bn(s_b)()
        e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print("%-76s BN %.3f ms  %.2f TB/s" % (name, t, gb / t), flush=True)
