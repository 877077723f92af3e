"""Diagnostic: the phases of scripts/x6_timeline.py for the BF16x6 DATA-GRADIENT kernel with BatchNorm-backward sums (STATS 2: the producer's saved
activation is read at the end of every tile) beside the forward kernel with sums (STATS 1) on the same shapes.  Build and run as x6_timeline.py."""
import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def show(tag, shape):
    out = (ctypes.c_longlong * 16)()
    L.cdll.unet_debug_x6_timeline(out)
    t = list(out)
    ch, tiles = max(t[9], 1), max(t[12], 1)
    print("%-6s %-26s chunks %4d tiles %3d | per chunk: " % (tag, str(shape), ch, tiles) + " ".join("P%d %5.0f+%-5.0f" % (j, t[2 * j] / ch, t[2 * j + 1] / ch) for j in range(4))
          + " barrier %5.0f | per tile: loop %7.0f epilogue %6.0f (column stage + sends %5.0f, barrier %5.0f, combine %5.0f, finish %5.0f)" % (
              t[8] / ch, t[10] / tiles, t[11] / tiles, t[13] / tiles, t[14] / tiles, t[15] / tiles, (t[11] - t[13] - t[14] - t[15]) / tiles), flush=True)
for shape in [(8, 64, 64, 512, 512), (8, 512, 512, 64, 64), (8, 256, 256, 128, 128)]:
    n, h, w, ci, co = shape
    x = torch.randn(n, h, w, ci, device="cuda"); dz = torch.randn(n, h, w, co, device="cuda"); r = torch.randn(n, h, w, ci, device="cuda")
    wt = torch.randn(3, 3, ci, co, device="cuda") / float(np.sqrt(9 * ci))
    u = torch.empty(L.unet_winograd_x6_weight_bytes(ci, co), dtype=torch.uint8, device="cuda"); ud = torch.empty_like(u)
    L.unet_winograd_weight_transform_x6(P(wt), P(u), ci, co, 0, ST()); L.unet_winograd_weight_transform_x6(P(wt), P(ud), ci, co, 1, ST())
    y = torch.empty(n, h, w, co, device="cuda"); dx = torch.empty(n, h, w, ci, device="cuda")
    rows_f = L.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, ci, co); part_f = torch.zeros((co // 64) * rows_f * 128, device="cuda")
    rows_d = L.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, co, ci); part_d = torch.zeros((ci // 64) * rows_d * 128, device="cuda")
    for _ in range(3):
        L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(u), None, P(y), co, n, h, w, ci, co, 1, P(part_f), part_f.numel() * 4, ST())
    torch.cuda.synchronize(); show("fwd+s", shape)
    for _ in range(3):
        L.unet_conv3x3_dgrad_winograd_x6(P(dz), co, P(ud), P(dx), ci, n, h, w, ci, co, P(r), ci, 0, ci, P(part_d), part_d.numel() * 4, ST())
    torch.cuda.synchronize(); show("dgrad", shape)
