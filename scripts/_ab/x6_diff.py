import ctypes, importlib, os, sys
import numpy as np, torch
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import fp32_error_cases as fc
L = importlib.import_module("semantic-segmentation-unet_amd._lib").lib()
P, ST = fc.P, fc.ST
shape = (2, 16, 16, 64, 64)
n, h, w, ci, co = shape
x, _, wt, b = fc.inputs(shape, 11)
U6 = fc.x6_weights(L, wt, 0)
rows = L.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, ci, co)
part = torch.zeros((co // 64) * rows * 128, device="cuda")
r = torch.full((n, h, w, co), -7.0, device="cuda"); r2 = torch.full_like(r, -7.0)
L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(U6), P(b), P(r), co, n, h, w, ci, co, 1, P(part), part.numel() * 4, ST())
L.unet_conv3x3_fwd_winograd_x6(P(x), ci, None, P(U6), P(b), P(r2), co, n, h, w, ci, co, 1, None, 0, ST())
torch.cuda.synchronize()
d = (r != r2)
print("differ", int(d.sum()), "of", d.numel(), "untouched r:", int((r == -7).sum()), "r2:", int((r2 == -7).sum()))
idx = d.nonzero()
print(idx[:20].tolist())
print("by image", d.sum((1,2,3)).tolist()); print("by row", d.sum((0,2,3)).tolist()); print("by col", d.sum((0,1,3)).tolist()); print("by ch", d.sum((0,1,2)).tolist())
ref = fc.ref_conv(x, wt, b, True)
print("err r", float((r.double()-ref).abs().max()), "err r2", float((r2.double()-ref).abs().max()))
