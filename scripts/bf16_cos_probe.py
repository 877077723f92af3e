import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
model = importlib.import_module("semantic-segmentation-unet_amd.model")
n, c, k, h, w = 1, 1, 2, 16, 176
g = torch.Generator().manual_seed(h + w)
img = torch.randn(n, c, h, w, generator=g)
cls = torch.randint(0, k, (n, h // 8, w // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
lab = torch.nn.functional.one_hot(cls, k).to(torch.int32)
batch = (img.cuda(), lab.cuda(), None, None)
ref = model.UNet(k, n, c, learning_rate=1e-3, seed=0, compute_dtype="fp32")
net = model.UNet(k, n, c, learning_rate=1e-3, seed=0, compute_dtype="bf16")
l0 = float(net.train_step(batch).numpy()); l1 = float(ref.train_step(batch).numpy())
e, er = net.engine, ref.engine
cos = lambda a, b: float((a.flatten().double() @ b.flatten().double()) / (a.norm().double() * b.norm().double()))
print(os.environ.get("TAG"), "loss", l0, l1, "cos logits %.4f dec_1b %.4f dec_1a %.4f conv_1a %.4f" % tuple(cos(e.g[nm + "/kernel"], er.g[nm + "/kernel"]) for nm in ("logits", "dec_1b", "dec_1a", "conv_1a")))
