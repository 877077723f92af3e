"""Diagnostic (GPU box): per-tensor gradient error table of the HIP engine vs the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import pkg
from test_gpu_unet import make_case
from oracle import unet_torch as ot
n, c, k, hw = 2, 1, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 64
img, lab, prm, masks = make_case(23, n, c, k, hw)
net = pkg("model").UNet(k, n, c); net.engine.load_parameters(prm)
ref = ot.TorchUNet(k, n, c, params=prm, dtype=torch.float64)
e = net.engine
e.forward(torch.as_tensor(img), training=True, dropout_masks=masks, labels=torch.as_tensor(lab), global_batch_size=n, want_grad=True)
e.backward()
loss_ref, _, g_ref, _ = ref.loss_and_grads(img, lab, masks)
print("loss", e.loss_buf[0].item(), float(loss_ref))
g = e.export_gradients()
for key in g_ref:
    r = g_ref[key].numpy(); a = g[key].astype(np.float64)
    print("%-16s max|ref| %.3e  max|err| %.3e  rel %.3e  rel_l2 %.3e" % (key, np.abs(r).max(), np.abs(a - r).max(), np.abs(a - r).max() / (np.abs(r).max() + 1e-30), np.linalg.norm(a - r) / (np.linalg.norm(r) + 1e-30)))
