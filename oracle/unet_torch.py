"""Second-opinion CPU restatement on torch-CPU ops + autograd (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED).

Independent of oracle/unet_numpy.py's hand-written backward: forward is built from torch.nn.functional
CPU ops under the layout maps of SURVEY.md 8(a) (`W_torch_conv[co,ci,a,b] = W_keras[a,b,ci,co]`,
`W_torch_convT[ci,co,a,b] = W_keras[a,b,co,ci]`) and gradients come from autograd.  Used to
(1) cross-check the numpy oracle, (2) check the HIP path at sizes numpy is too slow for, and
(3) as bench.py's `cpu_baseline` ("port") timed on the host cores.  Never imported by the product path.

Follows the reference graph UNet/model.py:85-146, loss UNet/model.py:211-215, Keras-Adam UNet/model.py:79,223.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import unet_numpy as on


class TorchUNet:
    device = torch.device("cpu")          # (class default: objects assembled without __init__ in tests stay on the CPU)

    def __init__(self, number_classes, global_batch_size, number_channels, learning_rate=3e-4, label_smoothing=0,
                 params=None, seed=0, dtype=torch.float32, contract=None, device="cpu"):
        self.number_classes = number_classes
        self.global_batch_size = global_batch_size
        self.number_channels = number_channels
        self.learning_rate = learning_rate
        self.label_smoothing = label_smoothing
        self.contract = contract or on.Contract()
        self.dtype = dtype
        # device: "cpu" (the oracle proper, and bench.py's cpu_baseline) or "cuda": the same restatement on torch's own GPU kernels, used
        # ONLY by tests as an independent implementation at sizes the CPU cannot reach (BASELINE config 2 at full size and batch)
        self.device = torch.device(device)
        self.layers = on.layer_table(number_channels, number_classes)
        src = params if params is not None else on.init_params(number_channels, number_classes, seed)
        self.params = {k: torch.tensor(np.asarray(v), dtype=dtype).to(self.device) for k, v in src.items()}
        self.trainable = on.trainable_names(number_channels, number_classes)
        for k in self.trainable:
            self.params[k].requires_grad_(True)
        self.adam_m = {k: torch.zeros_like(self.params[k]) for k in self.trainable}
        self.adam_v = {k: torch.zeros_like(self.params[k]) for k in self.trainable}
        self.iterations = 0

    def _block(self, name, kind, x, training, stats, relu_masks=None):
        P = self.params
        w, b = P[name + "/kernel"], P[name + "/bias"]
        if kind == "deconv":
            r = F.conv_transpose2d(x, w.permute(3, 2, 0, 1), b, stride=2)
        else:
            k = w.shape[0]
            z = F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=(k - 1) // 2)
            if relu_masks is not None and name in relu_masks:
                # the ReLU decision of a device run imposed (NCHW bool): the network is piecewise linear and fp32 rounding flips decisions
                # of pre-activations at rounding level; `imposed_flips[name]` = largest |z| among the overridden decisions relative to
                # max |z| of the layer, for the caller to bound
                m = torch.as_tensor(np.asarray(relu_masks[name])).to(self.device)
                flipped = (z.detach() > 0) != m
                self.imposed_flips[name] = float(z.detach().abs()[flipped].max() / z.detach().abs().max()) if bool(flipped.any()) else 0.0
                r = z * m.to(self.dtype)
            else:
                r = F.relu(z)
        g, bt = P[name + "/gamma"], P[name + "/beta"]
        eps = self.contract.bn_eps
        if training:
            mu = r.mean(dim=(0, 2, 3))
            var = r.var(dim=(0, 2, 3), unbiased=False)
            stats[name] = (mu.detach(), var.detach(), r.shape[0] * r.shape[2] * r.shape[3])
        else:
            mu, var = P[name + "/moving_mean"], P[name + "/moving_var"]
        inv = torch.rsqrt(var + eps)
        return (r - mu[None, :, None, None]) * (g * inv)[None, :, None, None] + bt[None, :, None, None]

    def forward(self, images, training=False, dropout_masks=None, relu_masks=None):
        x = torch.as_tensor(np.asarray(images) if not torch.is_tensor(images) else images).to(self.dtype).to(self.device)
        L = {n: k for n, k, _, _ in self.layers}
        st = {}
        self.imposed_flips = {}
        f = lambda name, t: self._block(name, L[name], t, training, st, relu_masks)
        scale = 1.0 / (1.0 - self.contract.dropout_rate)

        def drop(t, key):
            if not training:
                return t
            return t * torch.as_tensor(np.asarray(dropout_masks[key])).to(self.dtype).to(self.device) * scale

        s1 = f("conv_1b", f("conv_1a", x)); p1 = F.max_pool2d(s1, 2)
        s2 = f("conv_2b", f("conv_2a", p1)); p2 = F.max_pool2d(s2, 2)
        s3 = f("conv_3b", f("conv_3a", p2)); p3 = F.max_pool2d(s3, 2)
        s4 = drop(f("conv_4b", f("conv_4a", p3)), "drop_4"); p4 = F.max_pool2d(s4, 2)
        bt = drop(f("bott_b", f("bott_a", p4)), "drop_b")
        d4 = f("dec_4b", f("dec_4a", torch.cat([s4, f("up_4", bt)], 1)))
        d3 = f("dec_3b", f("dec_3a", torch.cat([s3, f("up_3", d4)], 1)))
        d2 = f("dec_2b", f("dec_2a", torch.cat([s2, f("up_2", d3)], 1)))
        d1 = f("dec_1b", f("dec_1a", torch.cat([s1, f("up_1", d2)], 1)))
        logits = f("logits", d1).permute(0, 2, 3, 1)
        return torch.softmax(logits, dim=-1), logits, st

    def loss(self, logits, labels):
        y = torch.as_tensor(np.asarray(labels) if not torch.is_tensor(labels) else labels).to(self.dtype).to(self.device)
        if self.label_smoothing:
            y = y * (1.0 - self.label_smoothing) + self.label_smoothing / y.shape[-1]
        if self.contract.ce_from_softmax_logits:
            ell = -(y * torch.log_softmax(logits, dim=-1)).sum(-1)
        else:
            # keras.backend.categorical_crossentropy(from_logits=False) as written: renormalise, clip, -sum y log q
            # (torch.clamp, like tf.clip_by_value, passes the gradient where min <= x <= max)
            q = torch.softmax(logits, dim=-1)
            q = q / q.sum(-1, keepdim=True)
            q = torch.clamp(q, self.contract.ce_clip_eps, 1.0 - self.contract.ce_clip_eps)
            ell = -(y * torch.log(q)).sum(-1)
        return (ell.sum(0) / self.global_batch_size).mean()

    def loss_and_grads(self, images, labels, dropout_masks, relu_masks=None):
        softmax, logits, st = self.forward(images, True, dropout_masks, relu_masks)
        loss = self.loss(logits, labels)
        grads = torch.autograd.grad(loss, [self.params[k] for k in self.trainable])
        return loss.detach(), softmax.detach(), dict(zip(self.trainable, grads)), st

    def train_step(self, images, labels, dropout_masks):
        loss, softmax, g, st = self.loss_and_grads(images, labels, dropout_masks)
        c = self.contract
        self.iterations += 1
        t = self.iterations
        alpha = self.learning_rate * np.sqrt(1.0 - c.adam_beta2 ** t) / (1.0 - c.adam_beta1 ** t)
        with torch.no_grad():
            for k in self.trainable:
                m, v = self.adam_m[k], self.adam_v[k]
                m.add_((g[k] - m) * float(np.float32(1.0) - np.float32(c.adam_beta1)))
                v.add_((g[k] * g[k] - v) * float(np.float32(1.0) - np.float32(c.adam_beta2)))
                self.params[k].sub_(alpha * m / (v.sqrt() + c.adam_eps))
            for name, (mu, var, n) in st.items():
                vv = var * (n / (n - 1.0)) if c.bn_moving_var_unbiased else var
                self.params[name + "/moving_mean"].mul_(c.bn_momentum).add_(mu * (1 - c.bn_momentum))
                self.params[name + "/moving_var"].mul_(c.bn_momentum).add_(vv * (1 - c.bn_momentum))
        return loss, softmax, g

    def test_step(self, images, labels):
        with torch.no_grad():
            softmax, logits, _ = self.forward(images, False)
            return self.loss(logits, labels), softmax

    def predict_mask(self, images):
        with torch.no_grad():
            softmax, _, _ = self.forward(images, False)
        return np.argmax(softmax.cpu().numpy(), axis=-1).astype(np.int32)

    def input_gradient_eval(self, images, dprob):
        """d sum(dprob * softmax) / d image with the graph in eval mode (autograd)."""
        x = torch.as_tensor(np.asarray(images)).to(self.dtype).to(self.device).requires_grad_(True)
        softmax, _, _ = self.forward(x, False)
        (softmax * torch.as_tensor(np.asarray(dprob)).to(self.dtype).to(self.device)).sum().backward()
        return x.grad.detach().cpu().numpy()

    def estimate_radius(self, img):
        """UNet.estimate_radius (reference UNet/model.py:165-202) on a given probe image [1,C,N,N]."""
        x = torch.as_tensor(np.asarray(img)).to(self.dtype).to(self.device).requires_grad_(True)
        mid = int(x.shape[2] / 2)
        softmax, _, _ = self.forward(x, False)
        msk = softmax.detach().clone()
        msk[0, mid, mid, :] = 1.0 - msk[0, mid, mid, :]
        loss = (msk - softmax).abs().mean(-1)                      # MeanAbsoluteError(reduction=NONE): [1,H,W]
        loss.sum().backward()                                      # tape.gradient of a non-scalar sums it
        g = np.abs(x.grad[0].cpu().numpy())
        g = np.average(g, axis=0) if g.shape[0] > 1 else g[0]
        vec = np.maximum(np.max(g, axis=0), np.max(g, axis=1))
        idx = np.nonzero(vec > 1e-8)[0]
        if len(idx) < 2:
            return on.RADIUS
        erf = int((np.max(idx) - np.min(idx)) / 2)
        return int(on.SIZE_FACTOR * np.ceil(float(erf) / on.SIZE_FACTOR))

    def numpy_params(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.params.items()}
