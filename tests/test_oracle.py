"""CPU tests of the oracle itself: hand-derived known-answer tests, finite differences, numpy-vs-torch cross-check.
PARITY UNPINNED vs TensorFlow: these pin the oracle to the published Keras semantics only."""
import numpy as np
import pytest
import torch

from oracle import unet_numpy as on
from oracle import unet_torch as ot


def test_parameter_count_matches_reference_graph():
    # 92 trainable tensors / 31,044,358 parameters for C=1, K=2
    assert on.count_trainable(1, 2) == 31044358
    assert len(on.trainable_names(1, 2)) == 92


def test_conv_known_answer_ramp():
    # all-ones 3x3 kernel on a ramp: interior pixel = 9 * centre value; corner sees 4 taps (zero 'same' padding)
    x = np.arange(25, dtype=np.float64).reshape(1, 1, 5, 5)
    w = np.ones((3, 3, 1, 1))
    z = on.conv_same_fwd(x, w, np.zeros(1))
    assert z[0, 0, 2, 2] == 9 * 12
    assert z[0, 0, 0, 0] == 0 + 1 + 5 + 6
    # cross-correlation, not convolution: a kernel with a single 1 at (a=0,b=0) reads the upper-left neighbour
    w2 = np.zeros((3, 3, 1, 1)); w2[0, 0] = 1
    z2 = on.conv_same_fwd(x, w2, np.zeros(1))
    assert z2[0, 0, 2, 2] == x[0, 0, 1, 1] and z2[0, 0, 0, 0] == 0


def test_deconv_index_map():
    # out[2i+a, 2j+b] = x[i,j] * w[a,b]
    x = np.array([[1., 2.], [3., 4.]]).reshape(1, 1, 2, 2)
    w = np.array([[10., 20.], [30., 40.]]).reshape(2, 2, 1, 1)
    z = on.deconv2x2_fwd(x, w, np.zeros(1))[0, 0]
    assert z[0, 0] == 10 and z[0, 1] == 20 and z[1, 0] == 30 and z[1, 1] == 40
    assert z[2, 3] == 4 * 20 and z[3, 2] == 4 * 30


def test_bn_known_answer():
    r = np.array([1., 3., 5., 7.]).reshape(4, 1, 1, 1)
    y, (xhat, inv, mu, var) = on.bn_train_fwd(r, np.ones(1), np.zeros(1), 1e-3)
    assert mu[0] == 4 and var[0] == 5                      # biased variance
    assert np.allclose(y[:, 0, 0, 0], (r[:, 0, 0, 0] - 4) / np.sqrt(5 + 1e-3))


def test_maxpool_first_max_on_ties():
    x = np.zeros((1, 1, 2, 2)); x[0, 0, 0, 1] = 1; x[0, 0, 1, 0] = 1       # tie between positions 1 and 2
    y, idx = on.maxpool2x2_fwd(x)
    assert y[0, 0, 0, 0] == 1 and idx[0, 0, 0, 0] == 1
    d = on.maxpool2x2_bwd(np.ones((1, 1, 1, 1)), idx)
    assert d[0, 0, 0, 1] == 1 and d.sum() == 1
    y0, idx0 = on.maxpool2x2_fwd(np.zeros((1, 1, 2, 2)))
    assert idx0[0, 0, 0, 0] == 0


def test_ce_uniform_is_ln2_and_reduction():
    z = np.zeros((2, 4, 4, 2))
    lab = np.zeros((2, 4, 4, 2), np.int32); lab[..., 0] = 1
    loss, p, y = on.ce_loss_fwd(z, lab, 2, 0, on.Contract())
    assert np.isclose(loss, np.log(2))                      # sum_n / G with G == N, then mean over H, W
    loss4, _, _ = on.ce_loss_fwd(z, lab, 4, 0, on.Contract())
    assert np.isclose(loss4, np.log(2) / 2)                 # per-replica partial loss with global batch 4


def test_keras_adam_first_step_closed_form():
    c = on.Contract()
    g = np.array([1e-5, -2e-3, 0.5])
    th, m, v = on.adam_keras_step(np.zeros(3), g, np.zeros(3), np.zeros(3), 1, 3e-4, c)
    expect = -3e-4 * np.sign(g) / (1 + c.adam_eps / (np.sqrt(1 - c.adam_beta2) * np.abs(g)))
    assert np.allclose(th, expect, rtol=2e-5)       # (1-beta) is formed in float32 like TF's kernel: 1.3e-5 off the ideal
    torch_style = -3e-4 * np.sign(g) / (1 + 1e-7 / np.abs(g))
    assert abs(th[0] - torch_style[0]) > 1e-8               # differs measurably from torch.optim.Adam for tiny |g|


def _small_case(seed=3, n=2, c=1, k=2, hw=16):
    rng = np.random.default_rng(seed)
    img, lab = on.synthetic_batch(n, c, k, hw, hw, seed=seed)
    P = on.init_params(c, k, seed=seed)
    for key in P:
        if key.endswith(("bias", "beta")):
            P[key] = rng.normal(0, 0.1, P[key].shape).astype(np.float32)
        if key.endswith("gamma"):
            P[key] = rng.uniform(0.5, 1.5, P[key].shape).astype(np.float32)
    masks = {"drop_4": rng.integers(0, 2, (n, 512, hw // 8, hw // 8)), "drop_b": rng.integers(0, 2, (n, 1024, hw // 16, hw // 16))}
    return img, lab, P, masks


def test_numpy_oracle_matches_torch_autograd_fp64():
    img, lab, P, masks = _small_case()
    o = on.OracleUNet(2, 2, 1, params=P, dtype=np.float64)
    loss, sm, g, _, _ = o.loss_and_grads(img, lab, masks)
    t = ot.TorchUNet(2, 2, 1, params=P, dtype=torch.float64)
    l2, sm2, g2, _ = t.loss_and_grads(img, lab, masks)
    assert abs(loss - float(l2)) < 1e-12
    assert np.abs(sm - sm2.numpy()).max() < 1e-11
    for k in g:
        ref = g2[k].numpy()
        assert np.abs(g[k] - ref).max() <= 1e-9 * (np.abs(ref).max() + 1e-30) + 1e-15, k


def test_finite_difference_gradient_fp64():
    img, lab, P, masks = _small_case(seed=5)
    o = on.OracleUNet(2, 2, 1, params=P, dtype=np.float64)
    loss, _, g, _, _ = o.loss_and_grads(img, lab, masks)
    rng = np.random.default_rng(0)
    for key in ("logits/kernel", "dec_1b/bias", "up_1/kernel", "conv_1a/kernel", "dec_1a/gamma"):
        idx = tuple(rng.integers(0, s) for s in o.params[key].shape)
        h = 1e-7          # the net is piecewise smooth (ReLU / max-pool kinks): larger steps cross kinks
        old = o.params[key][idx]
        o.params[key][idx] = old + h
        lp = o.loss_and_grads(img, lab, masks)[0]
        o.params[key][idx] = old - h
        lm = o.loss_and_grads(img, lab, masks)[0]
        o.params[key][idx] = old
        fd = (lp - lm) / (2 * h)
        assert abs(fd - g[key][idx]) <= 1e-4 * max(abs(fd), abs(g[key][idx])) + 2e-8, (key, fd, g[key][idx])


def test_train_step_and_eval_match_between_restatements():
    img, lab, P, masks = _small_case(seed=7)
    o = on.OracleUNet(2, 2, 1, params=P, dtype=np.float64)
    t = ot.TorchUNet(2, 2, 1, params=P, dtype=torch.float64)
    for _ in range(2):
        lo, _, _ = o.train_step(img, lab, masks)
        lt, _, _ = t.train_step(img, lab, masks)
        assert abs(lo - float(lt)) < 1e-10
    po, pt = o.params, t.numpy_params()
    for k in po:
        assert np.abs(po[k] - pt[k]).max() <= 1e-8 * (np.abs(pt[k]).max() + 1e-12) + 1e-12, k
    le, se = o.test_step(img, lab)
    lt, st = t.test_step(img, lab)
    assert abs(le - float(lt)) < 1e-10 and np.abs(se - st.numpy()).max() < 1e-10
    assert (o.predict_mask(img) == t.predict_mask(img)).all()


def test_ce_clip_path_numpy_backward_matches_torch_autograd():
    """Contract.ce_from_softmax_logits=False (Keras' clipped-probability cross-entropy): the hand-written numpy backward equals
    torch autograd through the literal formula (renormalise, clamp, -sum y log q), including pixels outside the clip range."""
    import torch
    from oracle import unet_torch as ot
    rng = np.random.default_rng(3)
    n, h, w, k, G = 2, 6, 5, 4, 4
    z = rng.standard_normal((n, h, w, k)) * 8
    lab = (rng.integers(0, k, (n, h, w))[..., None] == np.arange(k)).astype(np.int32)
    for eps, ls in ((1e-7, 0.0), (1e-3, 0.1)):
        c = on.Contract(ce_from_softmax_logits=False, ce_clip_eps=eps)
        loss, p, y = on.ce_loss_fwd(z, lab, G, ls, c)
        dl = on.ce_loss_bwd(p, y, G, c)
        assert ((p < eps) | (p > 1 - eps)).any()
        net = ot.TorchUNet.__new__(ot.TorchUNet)
        net.contract, net.dtype, net.label_smoothing, net.global_batch_size = c, torch.float64, ls, G
        zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
        lt = net.loss(zt, lab)
        lt.backward()
        assert abs(float(lt) - loss) < 1e-12 * abs(loss)
        assert np.abs(zt.grad.numpy() - dl).max() < 1e-12 * np.abs(dl).max() + 1e-18


def test_bf16_round_is_round_to_nearest_even_like_torch():
    # known answers: 1 + 2^-8 is a tie between 1.0 and 1 + 2^-7 -> even mantissa (1.0); 1 + 3*2^-8 ties up to 1 + 2^-6
    x = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -8 + 2.0 ** -20, -1.0 - 2.0 ** -8, 0.0, 3.0e38, 1e-40], np.float32)
    want = np.array([1.0, 1.0, 1.0 + 2.0 ** -6, 1.0 + 2.0 ** -7, -1.0, 0.0, 3.0e38, 1e-40], np.float32)
    got = on.bf16_round(x)
    assert np.array_equal(got[:6], want[:6])
    r = np.random.default_rng(0).standard_normal(200000).astype(np.float32) * np.float32(37.0)
    assert np.array_equal(on.bf16_round(r), torch.from_numpy(r).to(torch.bfloat16).float().numpy())
    assert np.array_equal(on.bf16_round(on.bf16_round(r)), on.bf16_round(r))                 # idempotent
    assert on.bf16_round(r.astype(np.float64)).dtype == np.float64


def test_bf16_oracle_mode_reduces_to_the_reference_arithmetic_and_documents_its_noise_floor():
    n, c, k, hw = 2, 3, 4, 32
    img, lab = on.synthetic_batch(n, c, k, hw, hw, seed=3)
    prm = on.init_params(c, k, seed=3)
    rng = np.random.default_rng(1)
    masks = {"drop_4": rng.integers(0, 2, (n, 512, hw // 8, hw // 8)), "drop_b": rng.integers(0, 2, (n, 1024, hw // 16, hw // 16))}
    l32, _, g32, _, _ = on.OracleUNet(k, n, c, params=prm).loss_and_grads(img, lab, masks)
    con = on.Contract(compute_dtype="bf16")
    # a plan that rounds nothing IS the fp32 contract
    e = frozenset()
    l0, _, g0, _, _ = on.OracleUNet(k, n, c, params=prm, contract=con, plan=on.Bf16Plan(e, e, e, e, e, e, False)).loss_and_grads(img, lab, masks)
    assert l0 == l32 and all(np.array_equal(g0[kk], g32[kk]) for kk in g32)
    # the default plan: close to the reference arithmetic in the loss, gradients at the bf16 noise level
    ref64 = on.OracleUNet(k, n, c, params=prm, contract=con)
    l16, _, g16, c16, _ = ref64.loss_and_grads(img, lab, masks)
    assert abs(l16 - l32) < 1e-2 * abs(l32)
    a, b = g16["logits/kernel"].ravel(), g32["logits/kernel"].ravel()
    assert a @ b / np.linalg.norm(a) / np.linalg.norm(b) > 0.95
    # every stored tensor the plan names is bf16-valued
    for name in ref64.plan.r_bf16:
        assert np.array_equal(on.bf16_round(c16[name][1]), c16[name][1]), name
    for name in ref64.plan.dz_bf16:
        assert np.array_equal(on.bf16_round(c16[name + "/dz"]), c16[name + "/dz"]), name
    # two faithful evaluations of the SAME plan (float32 vs float64 accumulation), same branch decisions, end far more than 1e-3
    # apart: a rounding turns a difference delta << ulp into one ulp on a fraction delta / ulp of the elements (rms sqrt(delta ulp)),
    # so free-running end-to-end comparisons of this mode are bounded by the quantisation level, not by the accumulation error
    relu = {name: c16[name][1] > 0 for name, kind, _, _ in ref64.layers if kind != "deconv"}
    pidx = {"pool_%d" % l: c16["pool_%d" % l] for l in (1, 2, 3, 4)}
    _, _, ga, _, _ = ref64.loss_and_grads(img, lab, masks, relu_masks=relu, pool_idx=pidx)
    _, _, gb, _, _ = on.OracleUNet(k, n, c, params=prm, contract=con, dtype=np.float32).loss_and_grads(img, lab, masks, relu_masks=relu, pool_idx=pidx)
    rel = [np.linalg.norm(gb[kk] - ga[kk]) / np.linalg.norm(ga[kk]) for kk in ga if kk.endswith("kernel")]
    assert np.median(rel) > 2e-3, np.median(rel)
