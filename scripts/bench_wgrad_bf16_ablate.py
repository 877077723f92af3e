"""bf16 weight-gradient kernel per layer shape.  Ablations are COMPILE-TIME now: build a variant with
    scripts/build_variant.sh wgabl1 conv_bf16.hip "-DUNET_CB_ABLATE=1"      (1 = no MFMA stream, 2 = no staging, 4 = no LDS writes, 8 = no barrier)
and run this script with UNET_HIP_LIB=semantic-segmentation-unet_amd/csrc/libunet_hip_wgabl1.so (results are wrong, timing only)."""
import ctypes, os, sys, importlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 8
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
bf = torch.bfloat16
for name, h, ci, co in [("1b", 512, 64, 64), ("2b", 256, 128, 128), ("3b", 128, 256, 256), ("4b", 64, 512, 512), ("bott_b", 32, 1024, 1024), ("dec4a", 64, 1024, 512), ("dec1a", 512, 128, 64)]:
    x = torch.randn(B, h, h, ci, device="cuda").to(bf); dz = torch.randn(B, h, h, co, device="cuda").to(bf)
    dw = torch.empty(3, 3, ci, co, device="cuda")
    nbw = L.unet_conv3x3_wgrad_bf16_workspace(B, h, h, ci, co); wsw = torch.empty(nbw + 256, dtype=torch.uint8, device="cuda")
    tw = timeit(lambda: L.unet_conv3x3_wgrad_bf16(P(x), ci, 1, P(dz), co, 1, P(dw), B, h, h, ci, co, P(wsw), nbw, ST()))
    fl = 2.0 * 9 * B * h * h * ci * co
    print("%-7s lib=%s wgrad %6.3f ms (%5.0f TF) ws %5.1f MB" % (name, os.path.basename(os.environ.get("UNET_HIP_LIB", "default")), tw, fl / tw / 1e9, nbw / 1e6), flush=True)
