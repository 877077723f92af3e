"""Drop-in for the reference's `UNet/model.py`: same class name, constructor signature, constants and method names
(reference UNet/model.py:19-256), with the TensorFlow graph replaced by the HIP engine (engine.py).

    UNet(number_classes, global_batch_size, number_channels, learning_rate=3e-4, label_smoothing=0)

Callers written against the reference keep working:
  * `train.py` uses get_keras_model()/get_optimizer()/set_learning_rate()/dist_train_step()/dist_test_step()
    (reference UNet/train.py:94-161);
  * `inference.py` uses load_checkpoint(), estimate_radius() and `model(batch_data)` returning softmax [N,H,W,K] that
    np.squeeze/np.argmax accept (reference UNet/inference.py:54,105-107,164-166,191-192).
"""
import numpy as np
import torch

from . import engine as _engine
from .engine import Engine


class Mean:
    """Look-alike of tf.keras.metrics.Mean (reference UNet/train.py:105,107): update_state / result / reset_states."""

    def __init__(self, name="mean"):
        self.name = name
        self.total, self.count = 0.0, 0.0

    def update_state(self, value, sample_weight=None):
        self.total += float(value)
        self.count += 1.0

    def add(self, total, count):
        """(total, count) pair summed over replicas -- tf.keras metrics under MirroredStrategy aggregate their `total` and
        `count` variables with SUM on read."""
        self.total += float(total)
        self.count += float(count)

    def result(self):
        return np.float32(self.total / self.count) if self.count else np.float32(0.0)

    def reset_states(self):
        self.total, self.count = 0.0, 0.0


class CategoricalAccuracy(Mean):
    """Look-alike of tf.keras.metrics.CategoricalAccuracy (reference UNet/train.py:106,108): mean over pixels of
    argmax(labels) == argmax(softmax).  The engine counts matches on the device; update_state(correct, total) adds them."""

    def update_state(self, correct, total):
        self.total += float(correct)
        self.count += float(total)


class _Optimizer:
    """The part of tf.keras.optimizers.Adam the reference touches: `.learning_rate` (UNet/model.py:154-158)."""

    def __init__(self, learning_rate):
        self.learning_rate = learning_rate


class _Loss:
    """What dist_train_step returns: something with .numpy() (reference UNet/train.py:141,159,161)."""

    def __init__(self, tensor):
        self._t = tensor

    def numpy(self):
        return np.float32(self._t.item())

    def __float__(self):
        return float(self._t.item())


class _KerasLikeModel:
    """`unet.get_keras_model()`: callable as m(x, training=False) with x fp32 [N,C,H,W] (numpy or torch), returning the
    softmax [N,H,W,K] (reference UNet/inference.py:105,164; UNet/model.py:177,209,239)."""

    def __init__(self, owner):
        self._o = owner

    def __call__(self, x, training=False, dropout_masks=None):
        """numpy in -> numpy out (the reference's callers run np.squeeze / np.argmax(...).astype on the result,
        UNet/inference.py:105-107,164-166); torch tensor in -> device tensor out (no host copy)."""
        o = self._o
        if not torch.is_tensor(x):
            xa = np.asarray(x, dtype=np.float32)
            # host arrays are checked where they enter: the fp32 route forms fp32 values as three bf16 pieces (csrc/winograd_x6.hip), which turns an
            # Inf into NaN (Inf - Inf in the split) where TensorFlow's fp32 kernels would carry the Inf -- either way the mask is garbage, so say so
            if not np.isfinite(xa).all():
                raise ValueError("input image contains Inf / NaN")
            xt = torch.as_tensor(xa)
        else:
            xt = x.float()
        prob = o.engine.forward(xt, training=bool(training), dropout_masks=dropout_masks)
        if torch.is_tensor(x):
            return prob.clone()      # engine buffers are reused by the next call
        return prob.cpu().numpy()

    @property
    def trainable_weights(self):
        return [self._o.engine.p[k] for k in self._o.engine.trainable_names()]


class UNet:
    _BASELINE_FEATURE_DEPTH = 64
    _KERNEL_SIZE = 3
    _DECONV_KERNEL_SIZE = 2
    _POOLING_STRIDE = 2

    SIZE_FACTOR = 16
    RADIUS = 96

    def __init__(self, number_classes, global_batch_size, number_channels, learning_rate=3e-4, label_smoothing=0,
                 device="cuda", seed=0, compute_dtype=None):
        self.number_channels = number_channels
        self.learning_rate = learning_rate
        self.number_classes = number_classes
        self.global_batch_size = global_batch_size
        self.label_smoothing = label_smoothing
        self.engine = Engine(number_classes, number_channels, device=device, seed=seed)
        if compute_dtype is not None:        # "fp32" (reference arithmetic) | "bf16" (mixed precision the reference leaves commented
            if compute_dtype not in ("fp32", "bf16"):                                  # out, UNet/train.py:52-54)
                raise ValueError("compute_dtype must be 'fp32' or 'bf16'")
            self.engine.compute_dtype = compute_dtype
        self.model = _KerasLikeModel(self)
        self.optimizer = _Optimizer(learning_rate)
        self.parallel = None         # set by parallel.DataParallel

    # -- reference UNet/model.py:81-83: tf.train.Checkpoint's own files (TensorBundle ckpt.index + ckpt.data-*, object-graph keys;
    #    checkpoint.py / tf_checkpoint.py); round-1 .npz files still load
    def load_checkpoint(self, checkpoint_filepath):
        from .checkpoint import load_checkpoint
        load_checkpoint(self, checkpoint_filepath)

    def save_checkpoint(self, checkpoint_filepath):
        from .checkpoint import save_checkpoint
        save_checkpoint(self, checkpoint_filepath)

    def get_keras_model(self):
        return self.model

    def get_optimizer(self):
        return self.optimizer

    def set_learning_rate(self, learning_rate):
        self.optimizer.learning_rate = learning_rate

    def get_learning_rate(self):
        return self.optimizer.learning_rate

    @staticmethod
    def _round_radius(x):
        return int(UNet.SIZE_FACTOR * np.ceil(float(x) / UNet.SIZE_FACTOR))

    def input_gradient(self, img, dprob):
        """d sum(dprob * softmax(img)) / d img with the model in eval mode; img [N,C,H,W], dprob [N,H,W,K]."""
        x = torch.as_tensor(np.asarray(img, dtype=np.float32)) if not torch.is_tensor(img) else img.float()
        self.engine.forward(x, training=False)
        g = torch.as_tensor(np.asarray(dprob, dtype=np.float32)) if not torch.is_tensor(dprob) else dprob.float()
        return self.engine.input_gradient_eval(g)

    def estimate_radius(self, img=None):
        """Effective-receptive-field probe (reference UNet/model.py:165-202): eval-mode forward of a random-normal
        [1,C,192,192] image, mean-absolute-error loss against a copy of the softmax whose centre pixel is flipped
        (1 - p), gradient of that loss w.r.t. the image, extent of |grad| > 1e-8, rounded up to a multiple of 16;
        falls back to the theoretical radius when fewer than 2 rows/columns respond.  (The reference repeats the
        identical forward 10 times and keeps the last tape; one pass is the same result.)"""
        N = 2 * UNet.RADIUS
        if img is None:
            img = np.random.normal(size=(1, self.number_channels, N, N))
        img = np.asarray(img, dtype=np.float32)
        mid = int(img.shape[2] / 2)
        x = torch.as_tensor(img)
        prob = self.engine.forward(x, training=False)
        k = prob.shape[-1]
        # loss[h,w] = mean_k |msk - p| is non-zero only at the centre pixel, where msk = 1 - p:
        # d/dp_k of mean_k |1 - 2 p_k| = -2 sign(1 - 2 p_k) / K
        g = torch.zeros_like(prob)
        pm = prob[0, mid, mid, :]
        g[0, mid, mid, :] = -2.0 * torch.sign(1.0 - 2.0 * pm) / k
        grad_img = np.abs(self.engine.input_gradient_eval(g)[0].cpu().numpy())      # [C,H,W]
        grad_img = np.average(grad_img, axis=0) if self.number_channels > 1 else grad_img[0]
        print('Theoretical RF: {}'.format(UNet.RADIUS))
        eps = 1e-8
        vec = np.maximum(np.max(grad_img, axis=0), np.max(grad_img, axis=1))
        idx = np.nonzero(vec > eps)[0]
        if len(idx) < 2:
            radius = UNet.RADIUS
            print('ERF based radius detection failed, defaulting to theoretical radius: {}'.format(radius))
        else:
            erf = int((np.max(idx) - np.min(idx)) / 2)
            radius = UNet._round_radius(erf)
            print('computed radius : "{}"'.format(radius))
        return radius

    # -- reference UNet/model.py:204-228
    def train_step(self, inputs, dropout_masks=None):
        (images, labels, loss_metric, accuracy_metric) = inputs
        e = self.engine
        images = torch.as_tensor(np.asarray(images, dtype=np.float32)) if not torch.is_tensor(images) else images
        labels = torch.as_tensor(np.asarray(labels, dtype=np.int32)) if not torch.is_tensor(labels) else labels
        e.forward(images, training=True, dropout_masks=dropout_masks, labels=labels,
                  global_batch_size=self.global_batch_size, label_smoothing=self.label_smoothing, want_grad=True)
        if self.parallel is not None:
            self.parallel.begin_step()
        e.backward()
        if self.parallel is not None:
            self.parallel.finish_step()
        e.adam_step(float(self.optimizer.learning_rate))
        return self._update_metrics(images, loss_metric, accuracy_metric)

    # -- reference UNet/model.py:237-250
    def test_step(self, inputs):
        (images, labels, loss_metric, accuracy_metric) = inputs
        e = self.engine
        images = torch.as_tensor(np.asarray(images, dtype=np.float32)) if not torch.is_tensor(images) else images
        labels = torch.as_tensor(np.asarray(labels, dtype=np.int32)) if not torch.is_tensor(labels) else labels
        e.forward(images, training=False, labels=labels, global_batch_size=self.global_batch_size,
                  label_smoothing=self.label_smoothing)
        return self._update_metrics(images, loss_metric, accuracy_metric)

    def _update_metrics(self, images, loss_metric, accuracy_metric):
        e = self.engine
        vals = e.loss_buf.clone()                                        # [per-replica loss, correctly classified pixels]
        self._reduced = None
        if loss_metric is not None or accuracy_metric is not None:       # forces the per-step host sync the reference has
            world = 1
            if self.parallel is not None and self.parallel.world_size > 1:
                # metric variables are summed over the replicas (X3); the same 8-byte all-reduce serves the loss SUM (X2)
                self._reduced = self.parallel.reduce_sum(vals.clone())
                world = self.parallel.world_size
            lb = (self._reduced if self._reduced is not None else vals).tolist()
            n, _, h, w = images.shape
            if loss_metric is not None:
                loss_metric.add(lb[0], world) if hasattr(loss_metric, "add") else loss_metric.update_state(lb[0] / world)
            if accuracy_metric is not None:
                accuracy_metric.update_state(lb[1], n * h * w * world)
        return _Loss(vals[0:1])

    # -- reference UNet/model.py:230-235,252-256: one replica per process here; the cross-replica SUM of the per-replica
    #    losses is an RCCL all-reduce in parallel.DataParallel (reference: dist_strategy.reduce(SUM, ...)).
    def dist_train_step(self, dist_strategy, inputs, dropout_masks=None):
        loss = self.train_step(inputs, dropout_masks=dropout_masks)
        return self._reduce_loss(dist_strategy, loss)

    def dist_test_step(self, dist_strategy, inputs):
        loss = self.test_step(inputs)
        return self._reduce_loss(dist_strategy, loss)

    def _reduce_loss(self, dist_strategy, loss):
        strat = dist_strategy if dist_strategy is not None else self.parallel
        if strat is not None and getattr(strat, "world_size", 1) > 1:
            if getattr(self, "_reduced", None) is not None:              # already summed together with the metric pairs
                return _Loss(self._reduced[0:1])
            return _Loss(strat.reduce_sum(loss._t))
        return loss
