"""CPU oracle: numpy restatement of the reference U-Net train / test / inference step.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported by the product path
(`semantic-segmentation-unet_amd/`); only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg use it, and there only as the checker / reported baseline.

PARITY UNPINNED.  The reference's arithmetic lives in TensorFlow 2.x / Keras (pinned only as
`tensorflow-gpu>=2.0.0`, UNet/requirements.txt:2), which is not installed here and cannot be
(SURVEY.md 8(c)); the reference ships no tests, golden vectors or fixtures for this path.  This
file therefore restates the *published Keras layer semantics* at the reference's own call sites
(cited per function).  Every Keras default that is not visible in the reference source is
collected in `Contract` below so it can be flipped in one place.

Layouts follow the reference: activations NCHW (`data_format='channels_first'`,
UNet/model.py:35,46,52), conv kernels HWIO `[kh,kw,Cin,Cout]`, transposed-conv kernels
`[kh,kw,Cout,Cin]` (Keras Conv2DTranspose), labels one-hot `[N,H,W,K]`, softmax output
`[N,H,W,K]` (Permute((2,3,1)), UNet/model.py:139).
"""
from dataclasses import dataclass

import numpy as np


@dataclass
class Contract:
    """Keras/TensorFlow defaults the reference relies on but does not spell out (SURVEY.md 8(a) "(K)")."""
    bn_eps: float = 1e-3             # BatchNormalization(epsilon=1e-3)          UNet/model.py:36,47
    bn_momentum: float = 0.99        # BatchNormalization(momentum=0.99)
    bn_moving_var_unbiased: bool = True   # fused BN feeds n/(n-1)*var into moving_variance
    dropout_rate: float = 0.5        # UNet/model.py:62, inverted scaling 1/(1-rate)
    adam_beta1: float = 0.9          # tf.keras.optimizers.Adam defaults          UNet/model.py:79
    adam_beta2: float = 0.999
    adam_eps: float = 1e-7
    ce_from_softmax_logits: bool = True   # graph-mode Keras CE sees the Softmax op and uses its logits
    ce_clip_eps: float = 1e-7        # used only when ce_from_softmax_logits is False
    pool_first_max: bool = True      # max-pool gradient goes to the first max in row-major window order
    # "fp32": the reference's arithmetic.  "bf16": the mixed-precision policy the reference keeps commented out
    # (UNet/train.py:52-54) as BASELINE config 4 states it -- bf16 forward/backward on fp32 master weights; the exact
    # rounding points are `Bf16Plan` below.  There is no reference implementation of this mode to be faithful to: the
    # plan IS the contract, and the device path is tested against it.
    compute_dtype: str = "fp32"


def bf16_round(a):
    """Round to bfloat16 (nearest, ties to even) and return in the array's own dtype -- what `v_cvt_pk_bf16_f32` does to an fp32
    value (a float64 input goes through float32 first, as a device value would)."""
    a = np.asarray(a)
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32)
    u = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return u.view(np.float32).astype(a.dtype)


@dataclass(frozen=True)
class Bf16Plan:
    """Where the bf16 mode (Contract.compute_dtype == "bf16") rounds to bfloat16.  Everything else -- accumulation, bias, ReLU,
    BatchNorm arithmetic and statistics, pooling, dropout, loss, Adam, master weights -- is fp32 on the device (fp64 here).

    contract        layers whose contractions (forward, data gradient, weight gradient) take BOTH operands rounded to bf16 with exact
                    products and wide accumulation: the 17 wide 3x3 layers and the 4 transposed convs (UNet/model.py:88-134 minus conv_1a;
                    the first layer, Cin = number_channels, and the 1x1 class map stay fp32)
    r_bf16          layers whose conv output r (post-ReLU, what BatchNorm reads) is STORED as bf16 in a training step.  The BatchNorm
                    batch statistics are taken from the unrounded values (the conv epilogue sums its fp32 accumulators); BatchNorm
                    apply, the ReLU mask and the BatchNorm backward read the rounded tensor
    y_bf16          layers whose BatchNorm output is stored as bf16 although its reader computes in fp32 (training step only).  (Every
                    other BatchNorm output feeds only bf16 contractions, where storage rounding and operand rounding coincide.)
    dz_bf16         layers whose BatchNorm-backward output dz (gradient w.r.t. the conv output, ReLU mask applied) is stored as bf16; the
                    bias gradient sum(dz) is taken before the rounding
    dx_bf16         layers whose data gradient (the dy of the layers below) is stored as bf16
    sums_from_dgrad layers whose BatchNorm-backward sums (sum dy, sum dy*r) are taken from the UNROUNDED data gradient of their one
                    consumer (its kernel's epilogue); all other layers reduce the stored (rounded) dy.  The element-wise part of the
                    BatchNorm backward always reads the stored dy.
    lvl4_accumulate_bf16   the level-4 skip gradient: pooled gradient of the bottleneck added into dec_4a's skip-half gradient and the
                    sum stored as bf16 (levels 1-3 add the two in fp32 inside conv_Nb's BatchNorm backward without storing the sum)
    """
    contract: frozenset
    r_bf16: frozenset
    y_bf16: frozenset
    dz_bf16: frozenset
    dx_bf16: frozenset
    sums_from_dgrad: frozenset
    lvl4_accumulate_bf16: bool = True

    @staticmethod
    def default():
        wide = [n for n, k, ci, co in layer_table(64, 64) if k in ("conv3", "deconv") and n != "conv_1a"]
        lv = (1, 2, 3, 4)
        return Bf16Plan(
            contract=frozenset(wide),
            r_bf16=frozenset(wide + ["conv_1a"]),
            y_bf16=frozenset(["dec_1b"]),
            dz_bf16=frozenset(wide + ["conv_1a"]),
            # every data gradient but the transposed conv in front of the dropout (up_4 -> bott_b) and the first layer's (not computed)
            dx_bf16=frozenset([n for n in wide if n != "up_4"] + ["logits"]),
            sums_from_dgrad=frozenset(["conv_%da" % l for l in lv] + ["dec_%da" % l for l in lv] + ["up_%d" % l for l in lv]
                                      + ["bott_a"] + ["dec_%db" % l for l in (2, 3, 4)]))


BASE = 64            # UNet._BASELINE_FEATURE_DEPTH   UNet/model.py:20
SIZE_FACTOR = 16     # UNet.SIZE_FACTOR               UNet/model.py:25
RADIUS = 96          # UNet.RADIUS                    UNet/model.py:26

# (name, kind, cin_fn, cout) in Keras layer-creation order, UNet/model.py:85-136.
def layer_table(number_channels, number_classes):
    C, K, B = number_channels, number_classes, BASE
    return [
        ("conv_1a", "conv3", C, B), ("conv_1b", "conv3", B, B),
        ("conv_2a", "conv3", B, 2 * B), ("conv_2b", "conv3", 2 * B, 2 * B),
        ("conv_3a", "conv3", 2 * B, 4 * B), ("conv_3b", "conv3", 4 * B, 4 * B),
        ("conv_4a", "conv3", 4 * B, 8 * B), ("conv_4b", "conv3", 8 * B, 8 * B),
        ("bott_a", "conv3", 8 * B, 16 * B), ("bott_b", "conv3", 16 * B, 16 * B),
        ("up_4", "deconv", 16 * B, 8 * B), ("dec_4a", "conv3", 16 * B, 8 * B), ("dec_4b", "conv3", 8 * B, 8 * B),
        ("up_3", "deconv", 8 * B, 4 * B), ("dec_3a", "conv3", 8 * B, 4 * B), ("dec_3b", "conv3", 4 * B, 4 * B),
        ("up_2", "deconv", 4 * B, 2 * B), ("dec_2a", "conv3", 4 * B, 2 * B), ("dec_2b", "conv3", 2 * B, 2 * B),
        ("up_1", "deconv", 2 * B, B), ("dec_1a", "conv3", 2 * B, B), ("dec_1b", "conv3", B, B),
        ("logits", "conv1", B, K),
    ]


def kernel_shape(kind, cin, cout):
    if kind == "conv3":
        return (3, 3, cin, cout)
    if kind == "conv1":
        return (1, 1, cin, cout)
    return (2, 2, cout, cin)          # Conv2DTranspose kernel is (kh, kw, out, in)


def init_params(number_channels, number_classes, seed=0, dtype=np.float32):
    """Glorot-uniform kernels, zero biases, gamma=1, beta=0, moving_mean=0, moving_var=1 (Keras defaults)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, kind, cin, cout in layer_table(number_channels, number_classes):
        shp = kernel_shape(kind, cin, cout)
        rf = shp[0] * shp[1]
        limit = np.sqrt(6.0 / (rf * shp[2] + rf * shp[3]))
        p[name + "/kernel"] = rng.uniform(-limit, limit, size=shp).astype(dtype)
        p[name + "/bias"] = np.zeros(cout, dtype)
        p[name + "/gamma"] = np.ones(cout, dtype)
        p[name + "/beta"] = np.zeros(cout, dtype)
        p[name + "/moving_mean"] = np.zeros(cout, dtype)
        p[name + "/moving_var"] = np.ones(cout, dtype)
    return p


def trainable_names(number_channels, number_classes):
    out = []
    for name, _, _, _ in layer_table(number_channels, number_classes):
        out += [name + "/kernel", name + "/bias", name + "/gamma", name + "/beta"]
    return out


# ------------------------------------------------------------------------------------------------
# primitive ops (each: forward + backward), NCHW
# ------------------------------------------------------------------------------------------------
def _windows(xp, k):
    return np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(2, 3))   # [N,C,H,W,k,k]


def conv_same_fwd(x, w, b):
    """Conv2D(padding='same', strides=1, channels_first): cross-correlation, zero pad. UNet/model.py:30-35."""
    k = w.shape[0]
    p = (k - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    win = _windows(xp, k)                                        # [N,Ci,H,W,a,b]
    z = np.tensordot(win, w, axes=([1, 4, 5], [2, 0, 1]))        # [N,H,W,Co]
    return np.ascontiguousarray(z.transpose(0, 3, 1, 2)) + b[None, :, None, None]


def conv_same_bwd(x, w, dz):
    k = w.shape[0]
    p = (k - 1) // 2
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    win = _windows(xp, k)                                        # [N,Ci,H,W,a,b]
    dw = np.tensordot(win, dz, axes=([0, 2, 3], [0, 2, 3]))      # [Ci,a,b,Co]
    dw = dw.transpose(1, 2, 0, 3)
    db = dz.sum(axis=(0, 2, 3))
    # dx = correlation of dz with the 180-degree rotated, in/out swapped kernel
    wr = w[::-1, ::-1].transpose(0, 1, 3, 2)                     # [a,b,Co,Ci]
    dzp = np.pad(dz, ((0, 0), (0, 0), (p, p), (p, p)))
    dwin = _windows(dzp, k)
    dx = np.tensordot(dwin, wr, axes=([1, 4, 5], [2, 0, 1]))
    return np.ascontiguousarray(dx.transpose(0, 3, 1, 2)), dw, db


def deconv2x2_fwd(x, w, b):
    """Conv2DTranspose(kernel=2, strides=2, padding='same'): z[n,co,2i+a,2j+b] = bias + sum_ci x[n,ci,i,j] w[a,b,co,ci].
    UNet/model.py:41-46."""
    n, ci, h, wd = x.shape
    co = w.shape[2]
    t = np.einsum("ncij,abdc->ndiajb", x, w, optimize=True)      # [N,Co,H,2,W,2]
    return t.reshape(n, co, 2 * h, 2 * wd) + b[None, :, None, None]


def deconv2x2_bwd(x, w, dz):
    n, ci, h, wd = x.shape
    co = w.shape[2]
    d6 = dz.reshape(n, co, h, 2, wd, 2)
    dx = np.einsum("ndiajb,abdc->ncij", d6, w, optimize=True)
    dw = np.einsum("ndiajb,ncij->abdc", d6, x, optimize=True)
    db = dz.sum(axis=(0, 2, 3))
    return dx, dw, db


def relu_fwd(z):
    return np.maximum(z, 0)


def bn_train_fwd(r, gamma, beta, eps):
    """BatchNormalization(axis=1), training=True: biased batch variance over (N,H,W). UNet/model.py:36,47."""
    mu = r.mean(axis=(0, 2, 3))
    var = r.var(axis=(0, 2, 3))
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (r - mu[None, :, None, None]) * inv[None, :, None, None]
    y = gamma[None, :, None, None] * xhat + beta[None, :, None, None]
    return y, (xhat, inv, mu, var)


def bn_train_bwd(dy, gamma, cache):
    xhat, inv, _, _ = cache
    m = dy.shape[0] * dy.shape[2] * dy.shape[3]
    dgamma = (dy * xhat).sum(axis=(0, 2, 3))
    dbeta = dy.sum(axis=(0, 2, 3))
    g = (gamma * inv)[None, :, None, None]
    dr = g * (dy - dbeta[None, :, None, None] / m - xhat * dgamma[None, :, None, None] / m)
    return dr, dgamma, dbeta


def bn_eval_fwd(r, gamma, beta, mm, mv, eps):
    inv = 1.0 / np.sqrt(mv + eps)
    return (gamma * inv)[None, :, None, None] * (r - mm[None, :, None, None]) + beta[None, :, None, None]


def maxpool2x2_fwd(x):
    """MaxPool2D(pool_size=2) stride 2, 'valid'. UNet/model.py:50-53.  Returns y and first-max index 0..3 (a*2+b)."""
    n, c, h, w = x.shape
    v = x.reshape(n, c, h // 2, 2, w // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    idx = v.argmax(axis=-1)                                       # first max in row-major window order
    return v.max(axis=-1), idx


def maxpool2x2_bwd(dy, idx):
    n, c, h2, w2 = dy.shape
    d = np.zeros((n, c, h2, w2, 4), dy.dtype)
    np.put_along_axis(d, idx[..., None], dy[..., None], axis=-1)
    return d.reshape(n, c, h2, w2, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, 2 * h2, 2 * w2)


def softmax_lastaxis(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def ce_loss_fwd(logits_nhwc, labels_onehot, global_batch_size, label_smoothing, contract):
    """loss_fn(labels, softmax) [N,H,W]; sum over N / G; mean over H,W.  UNet/model.py:77,211-215."""
    dt = logits_nhwc.dtype
    y = labels_onehot.astype(dt)
    k = y.shape[-1]
    if label_smoothing:
        y = y * (1.0 - label_smoothing) + label_smoothing / k
    p = softmax_lastaxis(logits_nhwc)
    if contract.ce_from_softmax_logits:
        zs = logits_nhwc - logits_nhwc.max(axis=-1, keepdims=True)
        logp = zs - np.log(np.exp(zs).sum(axis=-1, keepdims=True))
        ell = -(y * logp).sum(axis=-1)
    else:
        q = p / p.sum(axis=-1, keepdims=True)
        q = np.clip(q, contract.ce_clip_eps, 1.0 - contract.ce_clip_eps)
        ell = -(y * np.log(q)).sum(axis=-1)
    n, h, w = ell.shape
    loss = (ell.sum(axis=0) / global_batch_size).mean()
    return loss, p, y


def ce_loss_bwd(p, y, global_batch_size, contract=None):
    """d loss / d logits.  From-logits path: (p * sum(y) - y) / (G*H*W).  Clipped-probability path
    (keras.backend.categorical_crossentropy, from_logits=False): q = p / sum(p) (= p), clip to [eps, 1-eps] -- TF's
    clip_by_value passes the gradient only where eps <= q <= 1-eps -- so g_k = dl/dp_k = -y_k/p_k inside the range and 0
    outside; through the softmax Jacobian dl/dz_i = p_i (g_i - sum_j g_j p_j)."""
    n, h, w, _ = p.shape
    if contract is None or contract.ce_from_softmax_logits:
        return (p * y.sum(axis=-1, keepdims=True) - y) / (global_batch_size * h * w)
    eps = contract.ce_clip_eps
    inside = (p >= eps) & (p <= 1.0 - eps)
    gp = np.where(inside, -y, 0.0)                      # g_k * p_k
    return (gp - p * gp.sum(axis=-1, keepdims=True)) / (global_batch_size * h * w)


def adam_keras_step(theta, g, m, v, t, lr, contract):
    """Keras OptimizerV2 Adam (non-amsgrad), epsilon OUTSIDE the bias correction. UNet/model.py:79,223."""
    b1, b2, eps = contract.adam_beta1, contract.adam_beta2, contract.adam_eps
    alpha = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    # the reference runs in fp32: TF's ApplyAdam forms (1 - beta) as float32(1) - float32(beta), which is
    # 0.100000024 / 0.00099998713, not 0.1 / 0.001 (a 1.3e-5 relative difference in v) (K)
    omb1 = float(np.float32(1.0) - np.float32(b1))
    omb2 = float(np.float32(1.0) - np.float32(b2))
    m = m + (g - m) * omb1
    v = v + (g * g - v) * omb2
    theta = theta - alpha * m / (np.sqrt(v) + eps)
    return theta, m, v


# ------------------------------------------------------------------------------------------------
# whole network
# ------------------------------------------------------------------------------------------------
class OracleUNet:
    """Restatement of class UNet (UNet/model.py:19-256) on numpy arrays."""

    def __init__(self, number_classes, global_batch_size, number_channels, learning_rate=3e-4, label_smoothing=0,
                 params=None, seed=0, dtype=np.float64, contract=None, plan=None):
        self.number_classes = number_classes
        self.global_batch_size = global_batch_size
        self.number_channels = number_channels
        self.learning_rate = learning_rate
        self.label_smoothing = label_smoothing
        self.dtype = dtype
        self.contract = contract or Contract()
        # bf16 mode: `plan` names every rounding point (default: Bf16Plan.default(), the device path's default training plan)
        self.plan = (plan or Bf16Plan.default()) if self.contract.compute_dtype == "bf16" else None
        assert plan is None or self.plan is not None, "a Bf16Plan needs Contract(compute_dtype='bf16')"
        self.layers = layer_table(number_channels, number_classes)
        src = params if params is not None else init_params(number_channels, number_classes, seed)
        self.params = {k: np.array(v, dtype=dtype) for k, v in src.items()}
        self.trainable = trainable_names(number_channels, number_classes)
        self.adam_m = {k: np.zeros_like(self.params[k]) for k in self.trainable}
        self.adam_v = {k: np.zeros_like(self.params[k]) for k in self.trainable}
        self.iterations = 0

    # -- one "_conv_layer"/"_deconv_layer": linear -> (ReLU) -> BN    UNet/model.py:28-48
    def _block_fwd(self, name, kind, x, training, cache):
        P = self.params
        w, b = P[name + "/kernel"], P[name + "/bias"]
        pl = self.plan
        if pl is not None and name in pl.contract:
            x, w = bf16_round(x), bf16_round(w)          # both operands of the contraction; products exact, wide accumulation
        if kind == "deconv":
            r = deconv2x2_fwd(x, w, b)
        else:
            r = relu_fwd(conv_same_fwd(x, w, b))
        if training:
            gamma, beta = P[name + "/gamma"], P[name + "/beta"]
            if pl is not None and name in pl.r_bf16:
                # statistics from the unrounded conv output (epilogue sums), everything downstream reads the stored bf16 tensor
                _, (_, inv, mu, var) = bn_train_fwd(r, gamma, beta, self.contract.bn_eps)
                r = bf16_round(r)
                xhat = (r - mu[None, :, None, None]) * inv[None, :, None, None]
                y, bnc = gamma[None, :, None, None] * xhat + beta[None, :, None, None], (xhat, inv, mu, var)
            else:
                y, bnc = bn_train_fwd(r, gamma, beta, self.contract.bn_eps)
            if pl is not None and name in pl.y_bf16:
                y = bf16_round(y)
            cache[name] = (x, r, bnc)
        else:
            y = bn_eval_fwd(r, P[name + "/gamma"], P[name + "/beta"], P[name + "/moving_mean"],
                            P[name + "/moving_var"], self.contract.bn_eps)
            cache[name] = (x, r, None)
        return y

    def forward(self, images, training=False, dropout_masks=None, keep=None):
        """images [N,C,H,W] -> (softmax [N,H,W,K], cache).  dropout_masks: {"drop_4","drop_b"} of 0/1 arrays
        (NCHW shapes of conv_4b / bott_b outputs); required when training (RNG cannot match TF's)."""
        x = np.asarray(images, self.dtype)
        L = {n: k for n, k, _, _ in self.layers}
        c = {}
        acts = {}
        f = lambda name, t: self._block_fwd(name, L[name], t, training, c)
        scale = 1.0 / (1.0 - self.contract.dropout_rate)

        def drop(t, key):
            if not training:
                return t
            msk = np.asarray(dropout_masks[key], self.dtype)
            c[key] = msk
            return t * msk * scale

        s1 = f("conv_1b", f("conv_1a", x)); p1, c["pool_1"] = maxpool2x2_fwd(s1)
        s2 = f("conv_2b", f("conv_2a", p1)); p2, c["pool_2"] = maxpool2x2_fwd(s2)
        s3 = f("conv_3b", f("conv_3a", p2)); p3, c["pool_3"] = maxpool2x2_fwd(s3)
        s4 = drop(f("conv_4b", f("conv_4a", p3)), "drop_4"); p4, c["pool_4"] = maxpool2x2_fwd(s4)
        bt = drop(f("bott_b", f("bott_a", p4)), "drop_b")
        u4 = f("up_4", bt); d4 = f("dec_4b", f("dec_4a", np.concatenate([s4, u4], axis=1)))
        u3 = f("up_3", d4); d3 = f("dec_3b", f("dec_3a", np.concatenate([s3, u3], axis=1)))
        u2 = f("up_2", d3); d2 = f("dec_2b", f("dec_2a", np.concatenate([s2, u2], axis=1)))
        u1 = f("up_1", d2); d1 = f("dec_1b", f("dec_1a", np.concatenate([s1, u1], axis=1)))
        lg = f("logits", d1)                                           # 1x1 conv + ReLU + BN   UNet/model.py:136
        logits_nhwc = np.ascontiguousarray(lg.transpose(0, 2, 3, 1))   # Permute((2,3,1))       UNet/model.py:139
        softmax = softmax_lastaxis(logits_nhwc)                        # Softmax(axis=-1)       UNet/model.py:142
        c["logits_nhwc"] = logits_nhwc
        if keep is not None:
            loc = dict(s1=s1, p1=p1, s2=s2, p2=p2, s3=s3, p3=p3, s4=s4, p4=p4, bt=bt, u4=u4, d4=d4, u3=u3, d3=d3,
                       u2=u2, d2=d2, u1=u1, d1=d1, lg=lg)
            keep.update(loc)
        return softmax, c

    def _block_bwd(self, name, kind, dy, cache, grads, relu_masks=None, dy_sums=None):
        """dy: gradient w.r.t. the layer's BatchNorm output as the layer reads it.  Returns the data gradient (bf16 mode: as STORED,
        see Bf16Plan.dx_bf16) and, second, the same gradient before any storage rounding (what a fused epilogue sums).
        dy_sums = the unrounded dy (bf16 mode, Bf16Plan.sums_from_dgrad): sum(dy), sum(dy * r) come from it."""
        P = self.params
        pl = self.plan
        x, r, bnc = cache[name]
        w = P[name + "/kernel"]
        if pl is None:
            dr, dg, dbt = bn_train_bwd(dy, P[name + "/gamma"], bnc)
        else:
            xhat, inv, mu, _ = bnc
            m = dy.shape[0] * dy.shape[2] * dy.shape[3]
            ds = dy_sums if (dy_sums is not None and name in pl.sums_from_dgrad) else dy
            dbt = ds.sum(axis=(0, 2, 3))
            dg = (ds * xhat).sum(axis=(0, 2, 3))          # = invstd * (sum ds*r - mean * sum ds), r the stored tensor
            g = (P[name + "/gamma"] * inv)[None, :, None, None]
            dr = g * (dy - dbt[None, :, None, None] / m - xhat * dg[None, :, None, None] / m)
        grads[name + "/gamma"], grads[name + "/beta"] = dg, dbt
        dz = dr if kind == "deconv" else dr * (relu_masks[name] if relu_masks is not None else (r > 0))
        db = dz.sum(axis=(0, 2, 3))                       # bias gradient: before the storage rounding of dz
        if pl is not None and name in pl.dz_bf16:
            dz = bf16_round(dz)
        cache[name + "/dz"] = dz                          # as stored: what the weight / data gradient kernels read
        if pl is not None and name in pl.contract:
            w = bf16_round(w)                             # (x was saved rounded, dz is rounded: dz_bf16 >= contract)
            assert name in pl.dz_bf16
        if kind == "deconv":
            dx, dw, _ = deconv2x2_bwd(x, w, dz)
        else:
            dx, dw, _ = conv_same_bwd(x, w, dz)
        grads[name + "/kernel"], grads[name + "/bias"] = dw, db
        if pl is not None:
            return (bf16_round(dx) if name in pl.dx_bf16 else dx), dx
        return dx, dx

    # one layer at a time, on tensors supplied by the caller (tests feed the device run's own stored tensors: "teacher forcing")
    layer_forward = _block_fwd
    layer_backward = _block_bwd

    def loss_and_grads(self, images, labels, dropout_masks, relu_masks=None, pool_idx=None):
        """Forward (training=True) + loss + gradients of every trainable tensor.  UNet/model.py:208-219.
        relu_masks {layer: 0/1 [N,C,H,W]} / pool_idx {"pool_l": first-max index [N,C,H/2,W/2]} replace the oracle's own
        ReLU masks / pool winners IN THE BACKWARD PASS: the network is piecewise linear, and with the branch decisions of
        another evaluation (the HIP run's) imposed, the gradient is a smooth function of the inputs -- tests use this to
        compare gradients at 1e-4 instead of the 5e-2 that mask flips of near-zero pre-activations otherwise force."""
        softmax, c = self.forward(images, training=True, dropout_masks=dropout_masks)
        if pool_idx is not None:
            c.update(pool_idx)
        loss, p, y = ce_loss_fwd(c["logits_nhwc"], labels, self.global_batch_size, self.label_smoothing, self.contract)
        dl = ce_loss_bwd(p, y, self.global_batch_size, self.contract)
        L = {n: k for n, k, _, _ in self.layers}
        g = {}
        pl = self.plan
        R = bf16_round if pl is not None else (lambda t: t)
        b = lambda name, d, sums=None: self._block_bwd(name, L[name], d, c, g, relu_masks, sums)
        scale = 1.0 / (1.0 - self.contract.dropout_rate)
        d = np.ascontiguousarray(dl.transpose(0, 3, 1, 2))
        d, du = b("logits", d)

        def dec(d, du, a, bb, up, nskip):
            d, du = b(bb, d, du)
            d, du = b(a, d, du)
            dskip = d[:, :nskip]
            d, du = b(up, np.ascontiguousarray(d[:, nskip:]), np.ascontiguousarray(du[:, nskip:]))
            return dskip, d, du

        ds1, d, du = dec(d, None, "dec_1a", "dec_1b", "up_1", BASE)     # dec_1b reduces its own (stored) dy: the class map's kernel sums nothing
        ds2, d, du = dec(d, du, "dec_2a", "dec_2b", "up_2", 2 * BASE)
        ds3, d, du = dec(d, du, "dec_3a", "dec_3b", "up_3", 4 * BASE)
        ds4, d, du = dec(d, du, "dec_4a", "dec_4b", "up_4", 8 * BASE)
        d = d * c["drop_b"] * scale
        d, du = b("bott_b", d)
        d, du = b("bott_a", d, du)
        d = maxpool2x2_bwd(d, c["pool_4"]) + ds4
        if pl is not None and pl.lvl4_accumulate_bf16:
            d = R(d)
        d = d * c["drop_4"] * scale
        d, du = b("conv_4b", d)
        d, du = b("conv_4a", d, du)
        d = maxpool2x2_bwd(d, c["pool_3"]) + ds3
        d, du = b("conv_3b", d)
        d, du = b("conv_3a", d, du)
        d = maxpool2x2_bwd(d, c["pool_2"]) + ds2
        d, du = b("conv_2b", d)
        d, du = b("conv_2a", d, du)
        d = maxpool2x2_bwd(d, c["pool_1"]) + ds1
        d, du = b("conv_1b", d)
        dimg, _ = b("conv_1a", d, du)
        return loss, softmax, g, c, dimg

    def train_step(self, images, labels, dropout_masks):
        """UNet.train_step (UNet/model.py:204-228): forward, loss, grads, Adam apply, BN moving-stat update."""
        loss, softmax, g, c, _ = self.loss_and_grads(images, labels, dropout_masks)
        self.apply_gradients(g)
        self._update_moving(c)
        return loss, softmax, g

    def apply_gradients(self, g):
        self.iterations += 1
        for k in self.trainable:
            self.params[k], self.adam_m[k], self.adam_v[k] = adam_keras_step(
                self.params[k], g[k], self.adam_m[k], self.adam_v[k], self.iterations, self.learning_rate,
                self.contract)

    def _update_moving(self, c):
        mom = self.contract.bn_momentum
        for name, _, _, _ in self.layers:
            x, r, bnc = c[name]
            _, _, mu, var = bnc
            m = r.shape[0] * r.shape[2] * r.shape[3]
            v = var * (m / (m - 1.0)) if self.contract.bn_moving_var_unbiased else var
            P = self.params
            P[name + "/moving_mean"] = P[name + "/moving_mean"] * mom + mu * (1.0 - mom)
            P[name + "/moving_var"] = P[name + "/moving_var"] * mom + v * (1.0 - mom)

    def test_step(self, images, labels):
        """UNet.test_step (UNet/model.py:237-250): eval-mode forward + the same loss reduction."""
        softmax, c = self.forward(images, training=False)
        loss, _, _ = ce_loss_fwd(c["logits_nhwc"], labels, self.global_batch_size, self.label_smoothing, self.contract)
        return loss, softmax

    def predict_mask(self, images):
        """inference.py:_inference core (UNet/inference.py:159-166): eval forward, argmax over K (first max wins)."""
        softmax, _ = self.forward(images, training=False)
        return np.argmax(softmax, axis=-1).astype(np.int32)


def count_trainable(number_channels, number_classes):
    n = 0
    for name, kind, cin, cout in layer_table(number_channels, number_classes):
        n += int(np.prod(kernel_shape(kind, cin, cout))) + 3 * cout
    return n


def synthetic_batch(batch, number_channels, number_classes, height, width, seed=1234):
    """SURVEY.md 8(d) synthetic inputs: N(0,1) images; piecewise-constant labels on an 8x8 block grid, one-hot int32."""
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((batch, number_channels, height, width)).astype(np.float32)
    cls = rng.integers(0, number_classes, size=(batch, (height + 7) // 8, (width + 7) // 8))
    cls = np.repeat(np.repeat(cls, 8, axis=1), 8, axis=2)[:, :height, :width]
    onehot = (cls[..., None] == np.arange(number_classes)).astype(np.int32)
    return img, onehot
