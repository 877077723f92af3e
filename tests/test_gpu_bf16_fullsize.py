"""BASELINE config 4's per-GPU workload AT FULL SIZE (512x512x3, 4 classes, batch 8, bf16 mode), every tensor of one real training step
checked layer by layer -- the teacher-forced check of tests/test_gpu_bf16_oracle.py with the CPU oracle's arithmetic replaced by torch's OWN
GPU kernels (fp32 convolutions, fp64 reductions), because the numpy oracle cannot reach this size.  Same rounding plan (oracle.Bf16Plan),
same logic: every layer is evaluated on the DEVICE run's own stored inputs and must reproduce the device's stored output -- bf16 tensors to
one bf16 ulp on a small fraction of the elements, parameter gradients to 1e-4 (fp32 sums over 2 M pixels on both sides).

Why it exists: the kernels' multi-tile walks, 1024-block grids and >100 MB tensors never run in the small-shape tests.  In round 3 a kernel
that passed every small-shape oracle test stored 0.05 % of its dz elements wrong at this size (zeros where values belong): exactly what
the `dz` comparison below rejects."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import pkg
from oracle import unet_numpy as on

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def R(t):
    return t.to(BF).float()


def nchw(t):
    return t.float().permute(0, 3, 1, 2)


def close16(dev, ref, what, frac=0.03, l2=1e-3):
    scale = ref.abs().max().item() + 1e-30
    d = (dev - ref).abs()
    assert bool((d <= 2.0 ** -7 * ref.abs() + 2e-5 * scale).all()), (what, float((d / (ref.abs() + 2e-3 * scale)).max()))
    nz = ref.abs() > 1e-3 * scale
    differ = float((d[nz] > 0).float().mean()) if bool(nz.any()) else 0.0
    assert differ < frac, (what, differ)
    assert float((dev - ref).double().norm()) <= l2 * float(ref.double().norm()) + 1e-30, what


def close32(dev, ref, what, tol):
    assert float((dev - ref).abs().max()) <= tol * (float(ref.abs().max()) + 1e-30), (what, float((dev - ref).abs().max()) / (float(ref.abs().max()) + 1e-30))


def unpool(d, idx):
    n, c, h2, w2 = d.shape
    out = torch.zeros(n, c, 2 * h2, 2 * w2, device=d.device, dtype=d.dtype)
    for pos in range(4):
        out[:, :, pos >> 1::2, pos & 1::2] = torch.where(idx == pos, d, torch.zeros_like(d))
    return out


@pytest.mark.parametrize("cfg", [(8, 3, 4, 512), (3, 1, 2, (208, 304))])
def test_bf16_full_size_step_every_tensor_against_torch_gpu_kernels(cfg):
    n, c, k, hw = cfg
    h, w = hw if isinstance(hw, tuple) else (hw, hw)
    model = pkg("model")
    g = torch.Generator().manual_seed(11)
    img = torch.randn(n, c, h, w, generator=g)
    cls = torch.randint(0, k, (n, h // 8, w // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
    lab = torch.nn.functional.one_hot(cls, k).to(torch.int32)
    rng = np.random.default_rng(5)
    masks = {"drop_4": rng.integers(0, 2, (n, 512, h // 8, w // 8)), "drop_b": rng.integers(0, 2, (n, 1024, h // 16, w // 16))}
    net = model.UNet(k, n, c, seed=3, compute_dtype="bf16")
    e = net.engine
    # biases / gammas / betas away from their init values (0 / 1 / 0), like make_case
    prm = e.export_parameters()
    for key in prm:
        if key.endswith(("bias", "beta")):
            prm[key] = rng.normal(0, 0.1, prm[key].shape).astype(np.float32)
        if key.endswith("gamma"):
            prm[key] = rng.uniform(0.5, 1.5, prm[key].shape).astype(np.float32)
    e.load_parameters(prm)
    e.forward(img, training=True, dropout_masks=masks, labels=lab, global_batch_size=n, want_grad=True)
    e.backward()
    torch.cuda.synchronize()
    dev = e.dev
    plan = on.Bf16Plan.default()
    got = e.pl.rounding_points()
    assert all(got[kk] == getattr(plan, kk) for kk in got)
    P = {kk: torch.as_tensor(v).to(dev) for kk, v in prm.items()}
    kind = {nm: kd for nm, kd, _, _ in e.layers}
    md4 = torch.as_tensor(masks["drop_4"]).to(dev).float() * 2.0
    mdb = torch.as_tensor(masks["drop_b"]).to(dev).float() * 2.0
    B = on.BASE
    eps = on.Contract().bn_eps

    def x_of(name):
        return img.to(dev) if name == "conv_1a" else nchw(e.saved[name][0])

    def y_of(name):
        """where the layer's BatchNorm output went, as its readers see it"""
        if name == "logits":
            return nchw(e.bufs["y_logits"])
        if name == "bott_a":
            return x_of("bott_b")
        if name == "bott_b":
            return x_of("up_4")
        lvl = int(name.split("_")[1][0])
        ch = B << (lvl - 1)
        if name.startswith("conv_") and name.endswith("a"):
            return x_of("conv_%db" % lvl)
        if name.startswith("conv_"):
            return x_of("dec_%da" % lvl)[:, :ch]
        if name.startswith("up_"):
            return x_of("dec_%da" % lvl)[:, ch:]
        if name.endswith("a"):
            return x_of("dec_%db" % lvl)
        return x_of("up_%d" % (lvl - 1)) if lvl > 1 else x_of("logits")

    def wt(name):
        w_ = P[name + "/kernel"]
        return R(w_) if name in plan.contract else w_

    stats = {}
    # ---- forward
    for name, kd, _, _ in e.layers:
        x = x_of(name)
        if name in plan.contract:
            x = R(x)
        if kd == "deconv":
            r = F.conv_transpose2d(x, wt(name).permute(3, 2, 0, 1).contiguous(), P[name + "/bias"], stride=2)
        else:
            pad = (P[name + "/kernel"].shape[0] - 1) // 2
            r = torch.relu(F.conv2d(x, wt(name).permute(3, 2, 0, 1).contiguous(), P[name + "/bias"], padding=pad))
        mu = r.double().mean((0, 2, 3)); var = r.double().var((0, 2, 3), unbiased=False)
        inv = 1.0 / torch.sqrt(var + eps)
        r_dev = nchw(e.saved[name][1])
        (close16 if name in plan.r_bf16 else (lambda a, b_, wh: close32(a, b_, wh, 2e-5)))(r_dev, R(r) if name in plan.r_bf16 else r, "r " + name)
        st = e.stat[name].double()[:, :mu.numel()]
        close32(st[0], mu, "mean " + name, 1e-5)
        assert float((st[1] / inv - 1).abs().max()) < 1e-5, "invstd " + name
        y = (P[name + "/gamma"].double() * inv)[None, :, None, None] * (r_dev.double() - mu[None, :, None, None]) + P[name + "/beta"].double()[None, :, None, None]
        y = y.float()
        if name == "conv_4b":
            y = y * md4
        if name == "bott_b":
            y = y * mdb
        if name == "logits":
            close32(y_of(name), y, "y " + name, 1e-5)
        else:
            close16(y_of(name), R(y), "y " + name)
        stats[name] = (mu, inv)
        del x, r, r_dev, y
    for l in (1, 2, 3, 4):
        skip = y_of("conv_%db" % l)
        pooled = F.max_pool2d(skip, 2)
        nxt = "conv_%da" % (l + 1) if l < 4 else "bott_a"
        assert torch.equal(x_of(nxt), pooled), "pool_%d" % l
        idx = e.idx[l].permute(0, 3, 1, 2).long()
        win = torch.stack([skip[:, :, pos >> 1::2, pos & 1::2] for pos in range(4)], -1)
        assert torch.equal(torch.gather(win, -1, idx[..., None])[..., 0], pooled), "pool_%d winners" % l
        del skip, pooled, win
    logits = nchw(e.bufs["y_logits"]).permute(0, 2, 3, 1).double()
    logp = torch.log_softmax(logits, -1)
    loss_ref = float(-(lab.to(dev).double() * logp).sum(-1).sum(0).div(n).mean())
    assert abs(e.loss_buf[0].item() - loss_ref) < 1e-5 * abs(loss_ref)

    # ---- backward
    gdev = {kk: v for kk, v in e.g.items()}

    def dx_buf(name):
        return nchw(e.bufs["dy16_in_" + name] if ("dy16_in_" + name) in e.bufs else e.bufs["dy_in_" + name])

    def dz_buf(name):
        return nchw(e.bufs["dz16_" + name] if ("dz16_" + name) in e.bufs else e.bufs["dz_" + name])

    def bwd(name, dy, sums=None, sel=None, post=None, need_dx=True):
        kd = kind[name]
        mu, inv = stats[name]
        r = nchw(e.saved[name][1])
        xhat = (r.double() - mu[None, :, None, None]) * inv[None, :, None, None]
        m = dy.shape[0] * dy.shape[2] * dy.shape[3]
        ds = sums if (sums is not None and name in plan.sums_from_dgrad) else dy
        dbt = ds.double().sum((0, 2, 3))
        dg = (ds.double() * xhat).sum((0, 2, 3))
        gam = (P[name + "/gamma"].double() * inv)[None, :, None, None]
        dr = gam * (dy.double() - dbt[None, :, None, None] / m - xhat * dg[None, :, None, None] / m)
        dz = dr if kd == "deconv" else dr * (r > 0)
        db = dz.sum((0, 2, 3))
        dz = dz.float()
        dzd = dz_buf(name)
        (close16 if name in plan.dz_bf16 else (lambda a, b_, wh: close32(a, b_, wh, 2e-5)))(dzd, R(dz) if name in plan.dz_bf16 else dz, "dz " + name)
        for sfx, ref_ in (("/gamma", dg), ("/beta", dbt), ("/bias", db)):
            if kd == "deconv" and sfx == "/bias":
                continue
            a = gdev[name + sfx].double()
            assert float((a - ref_).norm()) <= 1e-4 * float(ref_.norm()) + 1e-12, (name + sfx, float((a - ref_).norm()) / float(ref_.norm()))
        del xhat, dr, dz, r
        # parameter / data gradients from the DEVICE's stored dz and input
        x = x_of(name)
        wq = wt(name).permute(3, 2, 0, 1).contiguous()
        if kd == "deconv":
            dw = torch.nn.grad.conv2d_weight(dzd.contiguous(), tuple(wq.shape), x.contiguous(), stride=2).permute(2, 3, 1, 0)
            dxu = F.conv2d(dzd, wq, stride=2) if need_dx else None
        else:
            pad = (wq.shape[2] - 1) // 2
            dw = torch.nn.grad.conv2d_weight(x.contiguous(), tuple(wq.shape), dzd.contiguous(), padding=pad).permute(2, 3, 1, 0)
            dxu = F.conv_transpose2d(dzd, wq, padding=pad) if need_dx else None
        close32(gdev[name + "/kernel"], dw, "dw " + name, 1e-4)
        if not need_dx:
            return None, None
        dxs = R(dxu) if name in plan.dx_bf16 else dxu
        got_ = dx_buf(name)
        exp = dxs if post is None else post(dxs)
        sl = (slice(None),) if sel is None else sel
        (close16 if name in plan.dx_bf16 else (lambda a, b_, wh: close32(a, b_, wh, 2e-5)))(got_[sl], exp[sl], "dx " + name)
        return got_, dxu

    d, du = bwd("logits", nchw(e.bufs["dy_logits"]))
    skips = {}
    pidx = {l: e.idx[l].permute(0, 3, 1, 2).long() for l in (1, 2, 3, 4)}
    for l, ch in ((1, B), (2, 2 * B), (3, 4 * B), (4, 8 * B)):
        d, du = bwd("dec_%db" % l, d, du)
        d, du = bwd("dec_%da" % l, d, du, sel=(slice(None), slice(ch, 2 * ch)) if l == 4 else None)
        skips[l] = (d[:, :ch], du[:, :ch])
        d, du = bwd("up_%d" % l, d[:, ch:], du[:, ch:], post=(lambda t: t * mdb) if l == 4 else None)
    d, du = bwd("bott_b", d)
    d, du = bwd("bott_a", d, du)
    # level 4: dy(conv_4b) = dropout(bf16(skip gradient + un-pooled bottleneck gradient)), formed in place.  The skip gradient the device
    # added was ITS rounding of the data gradient (overwritten since): where that rounding fell the other way the sum differs by one ulp of
    # the OPERAND, which is unbounded relative to a sum that cancels -- so the bound is relative to the operands' magnitudes here
    a_skip, a_pool = R(skips[4][1]), unpool(d, pidx[4])
    acc = R(a_skip + a_pool) * md4
    dd = (skips[4][0] - acc).abs()
    assert bool((dd <= 2.0 ** -6 * (a_skip.abs() + a_pool.abs()) * md4 + 2e-5 * acc.abs().max()).all()), "dy conv_4b"
    assert float((dd > 0).float().mean()) < 0.03 and float(dd.double().norm()) < 2e-3 * float(acc.double().norm()), "dy conv_4b"
    del a_skip, a_pool, acc, dd
    d, du = bwd("conv_4b", skips[4][0])
    d, du = bwd("conv_4a", d, du)
    for l in (3, 2, 1):
        dy = skips[l][0] + unpool(d, pidx[l])
        d, du = bwd("conv_%db" % l, dy)
        d, du = bwd("conv_%da" % l, d, du, need_dx=(l != 1))
