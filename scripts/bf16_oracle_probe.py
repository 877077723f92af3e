"""Probe: bf16 HIP step vs the bf16-emulating oracle (oracle.Bf16Plan) with the device run's branch decisions imposed --
prints loss / softmax / per-tensor gradient errors for the shapes of tests/test_gpu_unet.py."""
import sys, os, importlib
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import pkg
from oracle import unet_numpy as on
import test_gpu_unet as T

for cfg in [(2, 1, 2, 32), (2, 3, 4, 64), (1, 1, 2, 128), (3, 1, 2, (48, 80)), (1, 3, 6, (16, 176)), (1, 2, 11, 32), (5, 4, 3, 32)]:
    n, c, k, hw = cfg
    img, lab, prm, masks = T.make_case(41, n, c, k, hw)
    net = pkg("model").UNet(k, n, c, compute_dtype="bf16")
    e = net.engine
    e.load_parameters(prm)
    e.forward(torch.as_tensor(img), training=True, dropout_masks=masks, labels=torch.as_tensor(lab), global_batch_size=n, want_grad=True)
    e.backward(); torch.cuda.synchronize()
    relu = {name: (e.saved[name][1].float().permute(0, 3, 1, 2) > 0).cpu().numpy() for name, kind, _, _ in e.layers if kind != "deconv"}
    pidx = {"pool_%d" % l: e.idx[l].permute(0, 3, 1, 2).cpu().numpy().astype(np.int64) for l in (1, 2, 3, 4)}
    ref = on.OracleUNet(k, n, c, params=prm, dtype=np.float64, contract=on.Contract(compute_dtype="bf16"))
    _, cache = ref.forward(img, training=True, dropout_masks=masks)
    rr = {}
    for name, m in relu.items():
        r64 = cache[name][1]
        r_dev = e.saved[name][1].float().permute(0, 3, 1, 2).cpu().numpy()
        rr[name] = (np.abs(r64[m != (r64 > 0)]).max(initial=0.0) / np.abs(r64).max(), np.linalg.norm(r_dev - r64) / np.linalg.norm(r64))
    loss_ref, sm_ref, g_ref, _, _ = ref.loss_and_grads(img, lab, masks, relu_masks=relu, pool_idx=pidx)
    errs = T.grad_errors(e.export_gradients(), g_ref)
    print(cfg, "loss rel %.2e" % (abs(e.loss_buf[0].item() - loss_ref) / abs(loss_ref)),
          "softmax max %.2e" % np.abs(e.bufs["softmax"].cpu().numpy() - sm_ref).max())
    print("   mask-flip max |r|/max, r rel L2:", {k2: "%.1e/%.1e" % v for k2, v in rr.items() if k2 in ("conv_1a", "conv_1b", "conv_3a", "bott_b", "dec_3a", "dec_1b", "logits")})
    print("   worst grads:", ["%s %.1e" % (k2, v) for k2, v in sorted(errs.items(), key=lambda t: -t[1])[:8]])
    # the same with the fp32 oracle (no plan): how far the bf16 step is from the reference arithmetic
