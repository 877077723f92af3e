"""BASELINE config 4's arithmetic (bf16 forward/backward on fp32 master weights) pinned to the oracle.

`oracle.Contract(compute_dtype="bf16")` + `oracle.Bf16Plan` restate every rounding point of the mode with fp64 accumulation.  Two
evaluations of a bf16-storing network cannot be compared end to end at 1e-3, however faithful both are: a difference delta << ulp in
front of a rounding becomes, behind it, a difference of one ulp on a fraction delta/ulp of the elements -- rms sqrt(delta * ulp) >> delta --
so ANY perturbation (here: fp32 vs fp64 accumulation, 1e-6) grows layer by layer to the bf16 quantisation level itself (measured below:
two evaluations of the ORACLE, one accumulating in float32 and one in float64, end 1e-2 .. 4e-2 apart).  So the mode is pinned twice:

 1. TEACHER-FORCED, tensor by tensor through the real training step (test_bf16_step_every_tensor_matches_the_oracle_layer_by_layer):
    every layer of the oracle is evaluated on the DEVICE run's own stored inputs, forward and backward, and must reproduce the
    device's stored output -- bf16 tensors to one bf16 ulp on a small fraction of elements (roundings that fall the other way) and
    1e-3 relative L2, fp32 tensors (all 92 parameter gradients, the BatchNorm statistics, the loss) to 1e-4 / 2e-5.  A wrong rounding
    point, a dropped channel group in one weight gradient or a sign error in a deep layer is an O(2^-9 .. 1) error of one tensor
    here, with nothing upstream to hide behind.
 2. FREE-RUNNING with a calibrated bound (test_bf16_free_running_step_stays_within_the_quantisation_floor): the device step's
    distance from the fp64 oracle, per gradient tensor, is at most twice the distance between the float32- and float64-accumulating
    oracles under the same plan -- i.e. the device is as close to the oracle as the oracle is to itself.
"""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import unet_numpy as on
from test_gpu_unet import make_case, grad_errors

pytestmark = pytest.mark.gpu

SHAPES = [(2, 1, 2, 32), (2, 3, 4, 64), (1, 1, 2, 128), (3, 1, 2, (48, 80)), (1, 3, 6, (16, 176)), (1, 2, 11, 32), (5, 4, 3, 32)]


def nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu().numpy().astype(np.float64)


def device_plan(e):
    """the storage decisions of the bf16 engine's last training step, read off the tensors it used (-> oracle.Bf16Plan fields)"""
    bf = torch.bfloat16
    names = [n for n, _, _, _ in e.layers]
    r16 = frozenset(n for n in names if e.saved[n][1].dtype == bf)
    dz16 = frozenset(n for n in names if ("dz16_" + n) in e.bufs and ("dz_" + n) not in e.bufs)
    dx16 = frozenset(n for n in names if ("dy16_in_" + n) in e.bufs and ("dy_in_" + n) not in e.bufs)
    x16 = frozenset(n for n in names if e.saved[n][0].dtype == bf)
    return r16, dz16, dx16, x16


def run_device_step(cfg, seed=41):
    n, c, k, hw = cfg
    img, lab, prm, masks = make_case(seed, n, c, k, hw)
    net = pkg("model").UNet(k, n, c, compute_dtype="bf16")
    e = net.engine
    e.load_parameters(prm)
    e.forward(torch.as_tensor(img), training=True, dropout_masks=masks, labels=torch.as_tensor(lab), global_batch_size=n, want_grad=True)
    e.backward()
    torch.cuda.synchronize()
    return e, img, lab, prm, masks


def check_bf16(dev, ref, what, frac=0.03, l2=1e-3):
    """a bf16-stored device tensor against the oracle's (already rounded) one: equal but for roundings that fell the other way --
    at most one bf16 ulp (2^-7 relative; plus accumulation noise on elements that cancel to near zero), on a small fraction of the
    elements, 1e-3 in relative L2"""
    scale = np.abs(ref).max() + 1e-30
    d = np.abs(dev - ref)
    assert (d <= 2.0 ** -7 * np.abs(ref) + 2e-5 * scale).all(), (what, float((d / (np.abs(ref) + 2e-3 * scale)).max()))
    nz = np.abs(ref) > 1e-3 * scale
    differ = float((d[nz] > 0).mean()) if nz.any() else 0.0
    assert differ < frac, (what, differ)
    assert np.linalg.norm(dev - ref) <= l2 * np.linalg.norm(ref) + 1e-30, (what, np.linalg.norm(dev - ref) / np.linalg.norm(ref))


def check_f32(dev, ref, what, tol=2e-5):
    assert np.abs(dev - ref).max() <= tol * (np.abs(ref).max() + 1e-30), (what, np.abs(dev - ref).max() / (np.abs(ref).max() + 1e-30))


def chain_check(e, cfg, img, lab, prm, masks, plan, check_storage=True):
    """every tensor of the device step `e` just ran against the oracle under `plan`, one layer at a time on the device's own inputs"""
    n, c, k, hw = cfg
    if check_storage:               # the device stored exactly the tensors the plan says as bf16
        r16, dz16, dx16, x16 = device_plan(e)
        assert r16 == plan.r_bf16 and dz16 == plan.dz_bf16 and dx16 == plan.dx_bf16, (r16 ^ plan.r_bf16, dz16 ^ plan.dz_bf16, dx16 ^ plan.dx_bf16)
        assert x16 == plan.contract | {"logits"}, x16 ^ (plan.contract | {"logits"})        # inputs of the bf16 contractions + y_bf16 = {dec_1b}
    ref = on.OracleUNet(k, n, c, params=prm, dtype=np.float64, contract=on.Contract(compute_dtype="bf16"), plan=plan)
    kind = {nm: kd for nm, kd, _, _ in ref.layers}
    B = on.BASE
    md4, mdb = masks["drop_4"].astype(np.float64), masks["drop_b"].astype(np.float64)

    # ---- forward, teacher-forced: layer(name) on the device's stored input must give the device's stored r, statistics and y
    x_dev = {nm: nchw(e.saved[nm][0]) for nm in kind}
    x_dev["conv_1a"] = img.astype(np.float64)
    r_dev = {nm: nchw(e.saved[nm][1]) for nm in kind}
    # where each layer's BatchNorm output went (as its readers see it)
    y_dev = {}
    for l, ch in ((1, B), (2, 2 * B), (3, 4 * B), (4, 8 * B)):
        y_dev["conv_%da" % l] = x_dev["conv_%db" % l]
        y_dev["conv_%db" % l] = x_dev["dec_%da" % l][:, :ch]              # skip half of the concat (level 4: after the dropout)
        y_dev["up_%d" % l] = x_dev["dec_%da" % l][:, ch:]
        y_dev["dec_%da" % l] = x_dev["dec_%db" % l]
        y_dev["dec_%db" % l] = x_dev["up_%d" % (l - 1)] if l > 1 else x_dev["logits"]
    y_dev["bott_a"] = x_dev["bott_b"]
    y_dev["bott_b"] = x_dev["up_4"]                                       # after the dropout
    y_dev["logits"] = nchw(e.bufs["y_logits"])
    cache = {}
    for name, kd, _, _ in ref.layers:
        y = ref.layer_forward(name, kd, x_dev[name], True, cache)
        _, r_ref, (xhat, inv, mu, var) = cache[name]
        (check_bf16 if name in plan.r_bf16 else check_f32)(r_dev[name], r_ref, "r " + name)
        st = e.stat[name].cpu().numpy().astype(np.float64)[:, :mu.size]
        check_f32(st[0], mu, "mean " + name, 1e-5)
        assert np.abs(st[1] / inv - 1).max() < 1e-5, "invstd " + name
        # BatchNorm apply on the DEVICE's stored r (so that a rounding of r that fell the other way does not count twice)
        y = ref.params[name + "/gamma"][None, :, None, None] * ((r_dev[name] - mu[None, :, None, None]) * inv[None, :, None, None]) \
            + ref.params[name + "/beta"][None, :, None, None]
        if name == "conv_4b":
            y = y * md4 * 2.0
        if name == "bott_b":
            y = y * mdb * 2.0
        if name == "logits":
            check_f32(y_dev[name], y, "y " + name, 1e-5)
        else:
            check_bf16(y_dev[name], on.bf16_round(y), "y " + name)
        cache[name] = (cache[name][0], r_dev[name], ((r_dev[name] - mu[None, :, None, None]) * inv[None, :, None, None], inv, mu, var))
    for l in (1, 2, 3, 4):                                                # pooled tensors and first-max indices from the stored skip halves
        skip = y_dev["conv_%db" % l]
        pooled, idx = on.maxpool2x2_fwd(skip)
        nxt = "conv_%da" % (l + 1) if l < 4 else "bott_a"
        assert np.array_equal(x_dev[nxt], pooled), "pool_%d" % l
        # winners: levels 1-3 pool inside the BatchNorm-apply kernel, which decides on its fp32 values BEFORE the storage rounding (two
        # window elements may round to the same bf16 value: the device's winner is then the true maximum, not the first of the tie);
        # level 4 pools the stored tensor (after the dropout): first maximum in row-major window order.  Either way the winner must
        # hold the window's maximum, and be the first-max wherever that is unique.
        idx_dev = e.idx[l].permute(0, 3, 1, 2).cpu().numpy().astype(np.int64)
        nn_, cc_, h2, w2 = idx.shape
        win = skip.reshape(nn_, cc_, h2, 2, w2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(nn_, cc_, h2, w2, 4)
        assert np.array_equal(np.take_along_axis(win, idx_dev[..., None], -1)[..., 0], pooled), "pool_%d winners" % l
        unique = (win == pooled[..., None]).sum(-1) == 1
        assert np.array_equal(idx_dev[unique], idx[unique]) and (l < 4 or np.array_equal(idx_dev, idx)), "pool_%d winners" % l
    logits_nhwc = np.ascontiguousarray(y_dev["logits"].transpose(0, 2, 3, 1))
    loss_ref, p_ref, yl = on.ce_loss_fwd(logits_nhwc, lab, n, 0, ref.contract)
    assert abs(e.loss_buf[0].item() - loss_ref) < 1e-5 * abs(loss_ref)
    assert np.abs(e.bufs["softmax"].cpu().numpy() - p_ref).max() < 1e-5

    # ---- backward, teacher-forced: layer_backward(name) on the device's stored dy (and the unrounded data gradient the oracle derives
    # from the consumer's stored dz for the layers whose sums come from a data-gradient epilogue)
    g_ref, g_dev = {}, e.export_gradients()
    pidx = {l: e.idx[l].permute(0, 3, 1, 2).cpu().numpy().astype(np.int64) for l in (1, 2, 3, 4)}

    def dx_buf(name):
        return nchw(e.bufs["dy16_in_" + name] if ("dy16_in_" + name) in e.bufs else e.bufs["dy_in_" + name])

    def dz_buf(name):
        return nchw(e.bufs["dz16_" + name] if ("dz16_" + name) in e.bufs else e.bufs["dz_" + name])

    def bwd(name, dy, sums=None, dx_expect=None, dx_post=None, need_dx=True):
        """one layer: dz, parameter gradients and the stored data gradient against the device's"""
        dx_s, dx_u = ref.layer_backward(name, kind[name], dy, cache, g_ref, None, sums)
        (check_bf16 if name in plan.dz_bf16 else check_f32)(dz_buf(name), cache[name + "/dz"], "dz " + name)
        # continue from the DEVICE's dz: parameter / data gradients are functions of stored tensors only
        dzd = dz_buf(name)
        w = ref.params[name + "/kernel"]
        xs = cache[name][0]
        if name in plan.contract:
            w = on.bf16_round(w)
        if kind[name] == "deconv":
            dx_u, dw, _ = on.deconv2x2_bwd(xs, w, dzd)
        else:
            dx_u, dw, _ = on.conv_same_bwd(xs, w, dzd)
        check_f32(g_dev[name + "/kernel"].astype(np.float64), dw, "dw " + name)
        for sfx, tol in (("/gamma", 2e-5), ("/beta", 2e-5), ("/bias", 1e-4)):
            if kind[name] == "deconv" and sfx == "/bias":
                continue                                # exact gradient 0 (BatchNorm removes a constant): rounding noise on both sides
            a, b = g_dev[name + sfx].astype(np.float64), g_ref[name + sfx]
            assert np.linalg.norm(a - b) <= tol * np.linalg.norm(b) + 1e-12, (name + sfx, np.linalg.norm(a - b) / np.linalg.norm(b))
        if not need_dx:
            return None, None
        dx_s = on.bf16_round(dx_u) if name in plan.dx_bf16 else dx_u
        got = dx_buf(name)
        exp = dx_s if dx_post is None else dx_post(dx_s)
        sel = (slice(None),) if dx_expect is None else dx_expect
        (check_bf16 if name in plan.dx_bf16 else check_f32)(got[sel], exp[sel], "dx " + name)
        return got, dx_u

    d, du = bwd("logits", nchw(e.bufs["dy_logits"]))
    skips = {}
    for l, ch in ((1, B), (2, 2 * B), (3, 4 * B), (4, 8 * B)):
        d, du = bwd("dec_%db" % l, d, du)
        # dec_Na writes the whole concat gradient; level 4's skip half is later overwritten in place (pool gradient added, dropout): check
        # its upper half here and the skip half where it is consumed
        d, du = bwd("dec_%da" % l, d, du, dx_expect=(slice(None), slice(ch, 2 * ch)) if l == 4 else None)
        skips[l] = (d[:, :ch], du[:, :ch])
        post = (lambda t: t * mdb * 2.0) if l == 4 else None              # up_4's data gradient: the dropout ran in place on it
        d, du = bwd("up_%d" % l, np.ascontiguousarray(d[:, ch:]), np.ascontiguousarray(du[:, ch:]), dx_post=post)
    d, du = bwd("bott_b", d)
    d, du = bwd("bott_a", d, du)
    # level 4: dy(conv_4b) = dropout(bf16(skip gradient + un-pooled bottleneck gradient)), formed in place in dec_4a's buffer
    # (the skip gradient the device added was ITS rounding of the data gradient, overwritten since: where that rounding fell the other way
    # the sum is off by one ulp of the OPERAND, unbounded relative to a sum that cancels -- the bound is relative to the operands here)
    skip4_s, pool4 = on.bf16_round(skips[4][1]), on.maxpool2x2_bwd(d, pidx[4])
    acc = on.bf16_round(skip4_s + pool4) * md4 * 2.0
    dd = np.abs(skips[4][0] - acc)
    assert (dd <= 2.0 ** -6 * (np.abs(skip4_s) + np.abs(pool4)) * md4 * 2.0 + 2e-5 * np.abs(acc).max()).all(), "dy conv_4b"
    assert (dd > 0).mean() < 0.03 and np.linalg.norm(dd) < 2e-3 * np.linalg.norm(acc), "dy conv_4b"
    d, du = bwd("conv_4b", skips[4][0])
    d, du = bwd("conv_4a", d, du)
    for l in (3, 2, 1):
        dy = skips[l][0] + on.maxpool2x2_bwd(d, pidx[l])                  # added in fp32 inside conv_Nb's BatchNorm backward
        d, du = bwd("conv_%db" % l, dy)
        d, du = bwd("conv_%da" % l, d, du, need_dx=(l != 1))


@pytest.mark.parametrize("cfg", SHAPES)
def test_bf16_step_every_tensor_matches_the_oracle_layer_by_layer(cfg):
    e, img, lab, prm, masks = run_device_step(cfg)
    chain_check(e, cfg, img, lab, prm, masks, on.Bf16Plan.default())


def test_the_layer_by_layer_check_rejects_every_wrong_plan():
    # the teeth of the test above: the same device step checked against plans that differ from the device's in ONE rounding point each
    # must fail -- an unrounded weight operand, a tensor stored in the other precision, BatchNorm sums taken on the other side of a
    # rounding
    import dataclasses
    cfg = (2, 3, 4, 64)
    e, img, lab, prm, masks = run_device_step(cfg)
    good = on.Bf16Plan.default()
    chain_check(e, cfg, img, lab, prm, masks, good)
    wrong = {
        "weights of up_3 not rounded": dataclasses.replace(good, contract=good.contract - {"up_3"}),
        "r of dec_2a fp32": dataclasses.replace(good, r_bf16=good.r_bf16 - {"dec_2a"}),
        "r of logits bf16": dataclasses.replace(good, r_bf16=good.r_bf16 | {"logits"}),
        "dz of conv_3b fp32": dataclasses.replace(good, dz_bf16=good.dz_bf16 - {"conv_3b"}),
        "dx of dec_2b fp32": dataclasses.replace(good, dx_bf16=good.dx_bf16 - {"dec_2b"}),
        "dx of up_4 bf16": dataclasses.replace(good, dx_bf16=good.dx_bf16 | {"up_4"}),
        "sums of dec_2a from the rounded dy": dataclasses.replace(good, sums_from_dgrad=good.sums_from_dgrad - {"dec_2a"}),
        "sums of dec_1b from the unrounded dy": dataclasses.replace(good, sums_from_dgrad=good.sums_from_dgrad | {"dec_1b"}),
    }
    undetected = []
    for what, plan in wrong.items():
        try:
            chain_check(e, cfg, img, lab, prm, masks, plan, check_storage=False)
            undetected.append(what)
        except AssertionError:
            pass
    assert not undetected, undetected


@pytest.mark.parametrize("cfg", [(2, 3, 4, 64), (3, 1, 2, (48, 80))])
def test_bf16_free_running_step_stays_within_the_quantisation_floor(cfg):
    n, c, k, hw = cfg
    e, img, lab, prm, masks = run_device_step(cfg)
    relu = {name: (e.saved[name][1].float().permute(0, 3, 1, 2) > 0).cpu().numpy() for name, kd, _, _ in e.layers if kd != "deconv"}
    pidx = {"pool_%d" % l: e.idx[l].permute(0, 3, 1, 2).cpu().numpy().astype(np.int64) for l in (1, 2, 3, 4)}
    con = on.Contract(compute_dtype="bf16")
    ref64 = on.OracleUNet(k, n, c, params=prm, dtype=np.float64, contract=con)
    ref32 = on.OracleUNet(k, n, c, params=prm, dtype=np.float32, contract=con)
    l64, s64, g64, _, _ = ref64.loss_and_grads(img, lab, masks, relu_masks=relu, pool_idx=pidx)
    l32, s32, g32, _, _ = ref32.loss_and_grads(img, lab, masks, relu_masks=relu, pool_idx=pidx)
    floor = grad_errors({kk: np.asarray(v, np.float64) for kk, v in g32.items()}, g64)
    errs = grad_errors(e.export_gradients(), g64)
    for l in (1, 2, 3, 4):
        errs.pop("up_%d/bias" % l); floor.pop("up_%d/bias" % l)
    med = float(np.median(list(floor.values())))
    assert med > 2e-3, med              # the floor itself is far above 1e-3: what the module docstring says about end-to-end comparisons
    bad = {kk: (v, floor[kk]) for kk, v in errs.items() if v > 2.0 * max(floor[kk], med)}
    assert not bad, sorted(bad.items(), key=lambda t: -t[1][0])[:6]
    assert abs(e.loss_buf[0].item() - l64) < max(2.0 * abs(l32 - l64), 1e-3 * abs(l64))
    assert np.abs(e.bufs["softmax"].cpu().numpy() - s64).mean() < 2.0 * max(np.abs(s32 - s64).mean(), 1e-3)
