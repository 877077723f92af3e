"""Static forward / backward / Adam schedule of the U-Net hot path on one MI355X.

Replaces the part of TensorFlow the reference's `UNet` class uses (UNet/model.py:85-146 graph, :204-228 train step,
:237-250 test step): every op is a call into the C-ABI library (include/unet_hip.h) on raw device pointers.  PyTorch is
only the allocator (tensors own the HBM), the stream provider and -- in parallel.py -- the RCCL binding; there is no
autograd graph: the backward schedule below is written out by hand.

HBM layout
  * activations: fp32 NHWC; each layer keeps r (post-ReLU, pre-BN; BN backward + ReLU mask need it) and y (BN output =
    the next conv's input, needed by its weight gradient);
  * the four skip concatenations are zero-copy: `cat_l` is one [N,H,W,2C] buffer; the encoder's BN-apply writes
    channels [0,C), the decoder's transposed-conv BN-apply writes [C,2C); consumers read with channel stride 2C.
    Gradients mirror this (`dcat_l`): the decoder dgrad fills all 2C channels, max-pool backward accumulates into [0,C);
  * parameters, gradients and both Adam moments are four flat fp32 buffers in backward-completion order
    (logits first, conv_1a last) so data-parallel gradient buckets are contiguous ranges that become ready in order.
"""
import ctypes
import math
import os

import numpy as np
import torch

from ._lib import lib

BASE = 64                      # UNet._BASELINE_FEATURE_DEPTH  (reference UNet/model.py:20)
SIZE_FACTOR = 16               # UNet.SIZE_FACTOR              (reference UNet/model.py:25)

# Keras defaults the reference relies on (SURVEY.md 8(a) "(K)"): one place to flip them.
BN_EPS = 1e-3
BN_MOMENTUM = 0.99
BN_MOVING_VAR_UNBIASED = 1
DROPOUT_RATE = 0.5
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-7
CE_CLIP_EPS = 0.0              # 0: CE from the softmax's logits (graph-mode Keras); 1e-7: Keras' clipped-probability path


def layer_table(number_channels, number_classes):
    """(name, kind, Cin, Cout) in Keras layer-creation order (reference UNet/model.py:85-136)."""
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
    return {"conv3": (3, 3, cin, cout), "conv1": (1, 1, cin, cout), "deconv": (2, 2, cout, cin)}[kind]


BACKWARD_ORDER = ["logits", "dec_1b", "dec_1a", "up_1", "dec_2b", "dec_2a", "up_2", "dec_3b", "dec_3a", "up_3",
                  "dec_4b", "dec_4a", "up_4", "bott_b", "bott_a", "conv_4b", "conv_4a", "conv_3b", "conv_3a",
                  "conv_2b", "conv_2a", "conv_1b", "conv_1a"]


# layer -> (producer, first, last, parts): the layer's input channels [first/parts, last/parts) of its Cin are EXACTLY the BatchNorm
# output of `producer` and feed nothing else, so the layer's data gradient over that range is the producer's dy
PRODUCER = {"bott_b": ("bott_a", 0, 1, 1)}
for _l in (1, 2, 3, 4):
    PRODUCER["conv_%db" % _l] = ("conv_%da" % _l, 0, 1, 1)
    PRODUCER["dec_%db" % _l] = ("dec_%da" % _l, 0, 1, 1)
    PRODUCER["dec_%da" % _l] = ("up_%d" % _l, 1, 2, 2)          # concat [skip, upsampled] (UNet/model.py:55-58): upper half
for _l in (1, 2, 3):
    PRODUCER["up_%d" % _l] = ("dec_%db" % (_l + 1), 0, 1, 1)   # (up_4's input went through the dropout: no direct producer)


def flat_layout(number_channels, number_classes):
    """-> (slices {name/suffix: (offset, count, shape)}, layer_range {layer: (start, end)}, total): the flat parameter /
    gradient / Adam-moment buffers, layers in backward-completion order (logits first, conv_1a last), every tensor padded to
    4 floats.  Data-parallel buckets are contiguous ranges of this order (parallel.py)."""
    kind = {n: k for n, k, _, _ in layer_table(number_channels, number_classes)}
    cin = {n: ci for n, _, ci, _ in layer_table(number_channels, number_classes)}
    cout = {n: co for n, _, _, co in layer_table(number_channels, number_classes)}
    slices, layer_range, off = {}, {}, 0
    for name in BACKWARD_ORDER:
        start = off
        for suffix, shape in (("kernel", kernel_shape(kind[name], cin[name], cout[name])),
                              ("bias", (cout[name],)), ("gamma", (cout[name],)), ("beta", (cout[name],))):
            n = int(np.prod(shape))
            slices[name + "/" + suffix] = (off, n, shape)
            off += (n + 3) // 4 * 4
        layer_range[name] = (start, off)
    return slices, layer_range, off


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _ld(t):
    """channel stride (elements between consecutive pixels) of an NHWC view; checks the view is a plain channel slice."""
    n, h, w, c = t.shape
    ld = t.stride(2)
    assert t.stride(3) == 1 and t.stride(1) == w * ld and t.stride(0) == h * w * ld, "not an NHWC channel slice"
    return ld


class Engine:
    def __init__(self, number_classes, number_channels, device="cuda", seed=0):
        self.K, self.C = number_classes, number_channels
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("the U-Net hot path runs on an MI355X only (device must be cuda); no CPU fallback exists")
        self.L = lib()
        self.layers = layer_table(number_channels, number_classes)
        self.kind = {n: k for n, k, _, _ in self.layers}
        self.cin = {n: ci for n, _, ci, _ in self.layers}
        self.cout = {n: co for n, _, _, co in self.layers}
        # ---- flat parameter / gradient / moment buffers, backward-completion order, every tensor padded to 4 floats
        self.slices, self.layer_range, off = flat_layout(number_channels, number_classes)
        self.n_flat = off
        self.theta = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.grad = torch.zeros_like(self.theta)
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.p = {k: self.theta[o:o + n].view(shape) for k, (o, n, shape) in self.slices.items()}
        self.g = {k: self.grad[o:o + n].view(shape) for k, (o, n, shape) in self.slices.items()}
        self.moving = {}
        self.stat = {}
        for name, _, _, co in self.layers:
            self.moving[name + "/moving_mean"] = torch.zeros(co, dtype=torch.float32, device=self.dev)
            self.moving[name + "/moving_var"] = torch.ones(co, dtype=torch.float32, device=self.dev)
            cp = (co + 3) // 4 * 4
            self.stat[name] = torch.zeros(4, cp, dtype=torch.float32, device=self.dev)   # mean, invstd, scale, shift
        self.iterations = 0
        self.bufs = {}
        self._ws = None
        self.loss_buf = torch.zeros(2, dtype=torch.float32, device=self.dev)             # [loss, correct]
        self.dropout_seed = seed
        self.ce_clip_eps = CE_CLIP_EPS
        self._eval_folded = set()               # layers whose eval-mode (moving-statistics) BatchNorm-on-load fold is current
        self._eval_coefs = set()                # (layer, destination) pairs whose eval-mode scale / shift are current
        self.init_parameters(seed)
        self.on_layer_grads_ready = None        # hook(name) for data-parallel bucketing (parallel.py)
        self.profile = None                     # bench.py: {"conv3x3_fwd": [(ev0, ev1, flops)], ...} when enabled
        # Backward runs two HIP streams: the critical chain dgrad(L) -> bn_bwd(L-1) -> dgrad(L-1) ... on the caller's
        # stream, every weight gradient on `side` (it is needed only by the all-reduce / Adam).  The persistent wgrad
        # workgroups (106 KB LDS) co-reside with igemm workgroups and with the HBM-bound BN streams.
        self.overlap_wgrad = True
        # 3x3 route: "fused" = fully fused Winograd F(2x2,3x3) kernels wherever they apply (default, fastest on every layer shape
        # measured), "direct" = the implicit-GEMM MFMA kernels (also the fallback for odd tile sizes)
        self.conv_route = os.environ.get("UNET_CONV_ROUTE", "fused")
        self.wgrad_route = os.environ.get("UNET_WGRAD_ROUTE", "fused")
        # BatchNorm-apply on load (fp32 fused route): a layer whose only consumer is a fused Winograd conv does not materialise its
        # BatchNorm output; the consumer reads the conv output r through scaled weights, a folded bias and a per-channel padding
        # value (unet_winograd_weight_fold), its weight gradient is corrected by unet_conv3x3_wgrad_fold_fix.  UNET_BN_ON_LOAD=0: two passes.
        self.bn_on_load = os.environ.get("UNET_BN_ON_LOAD", "1") != "0"
        # bf16 activation storage also at the two ends of the network (the first layer's conv output, the class-map layer's input and
        # input gradient): the kernels there compute in fp32 on fp32 weights, only the 64-channel tensors they exchange with the bf16
        # layers are stored as bf16 (A/B switch)
        self.bf16_edge_activations = os.environ.get("UNET_BF16_EDGE_ACTIVATIONS", "1") != "0"
        # level 4 (whose skip goes through the dropout and the unfused pool kernels) stores its concat / pooled tensors and their gradients
        # as bf16 like levels 1-3 (A/B switch)
        self.bf16_level4 = os.environ.get("UNET_BF16_LEVEL4", "1") != "0"
        self._gamma_zero = None
        self.view = {}
        self._fused_U, self._fused_dirty = None, True
        self.fuse_bn_stats = os.environ.get("UNET_FUSE_BN_STATS", "1") != "0"      # BN sums from the conv epilogue (A/B switch)
        self.bnbwd_part = {}
        self.fuse_pool = os.environ.get("UNET_FUSE_POOL", "1") != "0"             # BN apply + max pool in one pass (A/B switch)
        # contraction precision of the wide 3x3 layers: "fp32" (the reference's arithmetic) or "bf16" (BASELINE config 4: bf16
        # forward/backward on fp32 master weights -- operands rounded to bf16, fp32 accumulation, everything else fp32)
        self.compute_dtype = os.environ.get("UNET_COMPUTE_DTYPE", "fp32")
        self._bf16_W, self._bf16_dirty = {}, True
        # bf16 mode, stage 2: tensors whose ONLY readers are bf16 contractions are stored as bf16 -- the BatchNorm outputs between
        # the two convs of a block and every dz.  The producing kernel rounds exactly as the consumers' staging would have, so
        # results are bit-identical to fp32 storage (tests assert that); only the bytes moved change.
        self.bf16_storage = os.environ.get("UNET_BF16_STORAGE", "1") != "0"
        self.bf16_storage_cat = os.environ.get("UNET_BF16_STORAGE_CAT", "1") != "0"    # ... the concat / pooled tensors too (A/B switch)
        # stage 3 (default on; UNET_BF16_ACTIVATIONS=0 restores stage 2): in training ALSO keep the wide layers' conv outputs r (what
        # BatchNorm reads) and the activation gradients dy the data-gradient kernels write as bf16 -- the Keras mixed_bfloat16
        # convention (bf16 activations and activation gradients, fp32 BatchNorm arithmetic: the fused sums are taken from the fp32
        # accumulators before the rounding).  Unlike the storage above this changes what BatchNorm sees (by one bf16 rounding).
        self.bf16_activations = os.environ.get("UNET_BF16_ACTIVATIONS", "1") != "0"
        self.bf16_convt_activations = os.environ.get("UNET_BF16_CONVT_ACTIVATIONS", "1") != "0"     # ... the transposed convs' too (A/B switch)
        self.side = torch.cuda.Stream(device=self.dev)
        self._ws_side = None

    def _timed(self, key, flops, fn, *args):
        """Call fn(*args); when profiling is on, bracket it with HIP events on the launch stream."""
        if self.profile is None:
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*args)
        e1.record()
        self.profile.setdefault(key, []).append((e0, e1, flops))

    # ------------------------------------------------------------------------------------------------ parameters
    def trainable_names(self):
        return [n + "/" + s for n, _, _, _ in self.layers for s in ("kernel", "bias", "gamma", "beta")]

    def init_parameters(self, seed=0):
        """Keras defaults: glorot-uniform kernels, zero bias, gamma 1, beta 0, moving mean 0 / var 1."""
        rng = np.random.default_rng(seed)
        vals = {}
        for name, kind, cin, cout in self.layers:
            shp = kernel_shape(kind, cin, cout)
            rf = shp[0] * shp[1]
            limit = math.sqrt(6.0 / (rf * shp[2] + rf * shp[3]))
            vals[name + "/kernel"] = rng.uniform(-limit, limit, size=shp).astype(np.float32)
            vals[name + "/bias"] = np.zeros(cout, np.float32)
            vals[name + "/gamma"] = np.ones(cout, np.float32)
            vals[name + "/beta"] = np.zeros(cout, np.float32)
            vals[name + "/moving_mean"] = np.zeros(cout, np.float32)
            vals[name + "/moving_var"] = np.ones(cout, np.float32)
        self.load_parameters(vals)

    def _use_fused(self, name, h, w, dgrad=False):
        """fully fused Winograd kernel: H, W even, reduce channels % 8 == 0, output channels % 64 == 0"""
        cin, cout = self.cin[name], self.cout[name]
        k, nn = (cout, cin) if dgrad else (cin, cout)
        return (self.conv_route == "fused" and self.kind[name] == "conv3" and h % 2 == 0 and w % 2 == 0
                and k % 8 == 0 and nn % 64 == 0)

    def _ybuf(self, name, shape, consumer):
        """BatchNorm-output buffer of `name`, read only by the 3x3 layer `consumer`: bf16 when that layer contracts in bf16."""
        n, h, w, _ = shape
        if self.compute_dtype == "bf16" and self.bf16_storage:
            ci, co = self.cin[consumer], self.cout[consumer]
            if self.kind[consumer] == "deconv":
                ok = self._use_bf16_convt(consumer, n, h, w) and self.L.unet_convT2x2_wgrad_bf16_supported(n, h, w, ci, co) == 1
            else:
                ok = self._use_bf16(consumer, n, h, w) and self.L.unet_conv3x3_wgrad_bf16_supported(n, h, w, ci, co) == 1
            if ok:
                return self._buf("y16_" + name, shape, torch.bfloat16)
        return self._buf("y_" + name, shape)

    def _use_bf16(self, name, n, h, w, dgrad=False):
        if self.compute_dtype != "bf16" or self.kind[name] != "conv3":
            return False
        cin, cout = self.cin[name], self.cout[name]
        k, nn = (cout, cin) if dgrad else (cin, cout)
        # the bf16 kernels address their tensors with 32-bit buffer offsets: every operand (leading dimension <= max(Cin, Cout):
        # a concat input IS the layer's Cin) must stay below 2 GiB, larger problems fall back to the fp32 kernels
        if n * h * w * max(cin, cout) * 4 >= 2 ** 31:
            return False
        return self.L.unet_conv3x3_bf16_supported(n, h, w, k, nn) == 1

    def _bf16_kernels(self, name):
        """(forward operand, data-gradient operand): the layer's fp32 master kernel packed to bf16, refreshed after every
        parameter change."""
        if not self._bf16_W:
            # persistent operand buffers + the job table of the batched pack (one launch per parameter change)
            rows, blk = [], 0
            for n, kind, cin, cout in self.layers:
                if kind not in ("conv3", "deconv") or cin % 64 or cout % 64:
                    continue
                taps = 4 if kind == "deconv" else 9
                nb = taps * cin * cout * 2
                w = (torch.empty(nb, dtype=torch.uint8, device=self.dev), torch.empty(nb, dtype=torch.uint8, device=self.dev))
                self._bf16_W[n] = w
                rows.append([self.p[n + "/kernel"].data_ptr(), w[0].data_ptr(), w[1].data_ptr(), cin | (cout << 32),
                             1 if kind == "deconv" else 0, blk])
                blk += (taps * cin * cout // 8 + 255) // 256
            self._bf16_jobs = torch.tensor(rows, dtype=torch.int64, device=self.dev)
            self._bf16_blocks = blk
            self._bf16_dirty = True
        if self._bf16_dirty:
            self.L.unet_bf16_pack_weights_batch(_p(self._bf16_jobs), self._bf16_jobs.shape[0], self._bf16_blocks, self._stream())
            self._bf16_dirty = False
        return self._bf16_W[name]

    def _use_bf16_convt(self, name, n, h, w):
        return (self.compute_dtype == "bf16" and self.kind[name] == "deconv"
                and n * h * w * 4 * self.cout[name] * 4 < 2 ** 31 and n * h * w * self.cin[name] * 4 < 2 ** 31
                and self.L.unet_convT2x2_bf16_supported(n, h, w, self.cin[name], self.cout[name]) == 1)

    def _fused_kernels(self, name):
        """(Uc forward, Uc dgrad) in the chunked layout of the fused kernel.  The buffers are persistent; after a parameter
        change ALL fused-route layers are re-transformed by one batched launch at the first use."""
        if self._fused_U is None:
            names = [n for n, _, _, _ in self.layers if self.kind[n] == "conv3" and self.cin[n] % 8 == 0 and self.cout[n] % 64 == 0]
            self._fused_U, rows, blk = {}, [], 0
            for n in names:
                cin, cout = self.cin[n], self.cout[n]
                u = (torch.empty(16 * cin * cout, dtype=torch.float32, device=self.dev),
                     torch.empty(16 * cin * cout, dtype=torch.float32, device=self.dev))
                self._fused_U[n] = u
                rows.append([self.p[n + "/kernel"].data_ptr(), u[0].data_ptr(), u[1].data_ptr(), cin | (cout << 32), blk, 0])
                blk += (cin * cout + 2047) // 2048
            self._fused_jobs = torch.tensor(rows, dtype=torch.int64, device=self.dev)
            self._fused_blocks = blk
            self._fused_dirty = True
        if self._fused_dirty:
            self.L.unet_winograd_weight_transform_batch(_p(self._fused_jobs), self._fused_jobs.shape[0], self._fused_blocks, self._stream())
            self._fused_dirty = False
        return self._fused_U[name]

    def parameters_changed(self):
        """theta was written from outside (broadcast, checkpoint): every cached transform of the kernels is stale."""
        self._fused_dirty = True
        self._bf16_dirty = True
        self._eval_folded.clear(); self._eval_coefs.clear()
        self._gamma_zero = None                 # re-checked lazily (BatchNorm-apply on load needs every gamma != 0)

    def load_parameters(self, values):
        """values: {keras-style name: array in the Keras layout}.  Resets nothing else."""
        self.parameters_changed()
        for k, v in values.items():
            t = torch.as_tensor(np.ascontiguousarray(np.asarray(v, dtype=np.float32)))
            if k in self.p:
                self.p[k].copy_(t.view(self.p[k].shape))
            elif k in self.moving:
                self.moving[k].copy_(t)
            else:
                raise KeyError(k)

    def export_parameters(self):
        out = {k: v.detach().cpu().numpy().copy() for k, v in self.p.items()}
        out.update({k: v.detach().cpu().numpy().copy() for k, v in self.moving.items()})
        return out

    def export_gradients(self):
        return {k: v.detach().cpu().numpy().copy() for k, v in self.g.items()}

    # ------------------------------------------------------------------------------------------------ buffers
    def _buf(self, name, shape, dtype=torch.float32):
        t = self.bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.dev)
            self.bufs[name] = t
        return t

    def _workspace(self, nbytes, side=False):
        cur = self._ws_side if side else self._ws
        if cur is None or cur.numel() < nbytes:
            cur = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.dev)
            if side:
                self._ws_side = cur
            else:
                self._ws = cur
        return cur

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------------------------------------------ forward
    def _gamma_nonzero(self):
        """BatchNorm-apply on load divides by the BatchNorm scale (pad = -shift / scale): usable while no gamma is exactly 0.
        Checked (one host sync) after parameters were written from outside, not after optimizer steps."""
        if self._gamma_zero is None:
            self._gamma_zero = any(bool((self.p[n + "/gamma"] == 0).any().item()) for n, _, _, _ in self.layers)
        return not self._gamma_zero

    def _can_defer(self, name, consumer, n, h, w):
        """`name`'s BatchNorm output feeds only the 3x3 layer `consumer` (spatial size h x w), which runs the fp32 fused Winograd
        forward and weight-gradient kernels: the output need not be materialised (BatchNorm-apply on load)."""
        return (self.bn_on_load and self.compute_dtype == "fp32" and self.wgrad_route == "fused" and self._use_fused(consumer, h, w)
                and self.L.unet_winograd_wgrad_fused_supported(n, h, w, self.cin[consumer], self.cout[consumer]) == 1
                and self._gamma_nonzero())

    def _fold_buffers(self, name):
        cin, cout = self.cin[name], self.cout[name]
        return (self._buf("Ufold_" + name, (16 * cin * cout,)), self._buf("bfold_" + name, (cout,)), self._buf("pad_" + name, (cin + 8,)))

    def _block_fwd(self, name, x, y_out, training, pool=None, in_view=None, r_out=None, stat_out=None):
        """x: NHWC view (input of the layer), y_out: NHWC view the BN output is written to, or None: the BatchNorm output is NOT
        materialised (its consumer applies it on load) and the conv output r is returned instead.
        in_view = (scale, shift) per input channel: x is a producer's conv output (BatchNorm-apply on load, fused Winograd route);
        r_out: where the conv output goes (default: the layer's own buffer); stat_out = (scale, shift) destinations of the BatchNorm
        coefficients (default: self.stat[name][2:4])."""
        L, st = self.L, self._stream()
        kind, cin, cout = self.kind[name], self.cin[name], self.cout[name]
        n, h, w, _ = x.shape
        w_, b_ = self.p[name + "/kernel"], self.p[name + "/bias"]
        fused_stats = None
        assert in_view is None or (kind == "conv3" and self._use_fused(name, h, w) and not self._use_bf16(name, n, h, w))
        if kind == "deconv":
            r = r_out if r_out is not None else self._buf("r_" + name, (n, 2 * h, 2 * w, cout))
            if self._use_bf16_convt(name, n, h, w):
                rows = L.unet_convT2x2_bf16_stats_rows(n, h, w, cin, cout, 0) if (training and self.fuse_bn_stats) else 0
                stat_part = self._buf("bnpart_" + name, ((cout // 64) * rows * 128,)) if rows > 0 else None
                if self.bf16_activations and self.bf16_convt_activations and self.bf16_storage and rows > 0 and r_out is None and self._dz16_shape(name, n, h, w):
                    r = self._buf("r16_" + name, (n, 2 * h, 2 * w, cout), torch.bfloat16)
                L.unet_convT2x2_fwd_bf16(_p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(self._bf16_kernels(name)[0]), _p(b_), _p(r), _ld(r),
                                         int(r.dtype == torch.bfloat16), n, h, w, cin, cout, _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, st)
                if rows > 0:
                    fused_stats = (stat_part, rows)
            elif L.unet_convT2x2_fwd_stream_supported(n, h, w, cin, cout) == 1 and _ld(x) <= 4096:           # persistent stream kernel
                rows = L.unet_convT2x2_fwd_stream_stats_rows(n, h, w, cin, cout) if (training and self.fuse_bn_stats) else 0
                if rows > 0:
                    stat_part = self._buf("bnpart_" + name, ((cout // 64) * rows * 128,))
                    L.unet_convT2x2_fwd_stream_stats(_p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout,
                                                     _p(stat_part), stat_part.numel() * 4, st)
                    fused_stats = (stat_part, rows)
                else:
                    L.unet_convT2x2_fwd_stream(_p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout, st)
            else:
                L.unet_convT2x2_fwd(_p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout, st)
        elif kind == "conv1":
            r = self._buf("r_" + name, (n, h, w, cout))
            L.unet_conv1x1_fwd(_p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(w_), _p(b_), _p(r), cout, n * h * w, cin, cout, 1, st)
        else:
            r = r_out if r_out is not None else self._buf("r_" + name, (n, h, w, cout))
            if self._use_bf16(name, n, h, w):
                rows = L.unet_conv3x3_bf16_stats_rows(n, h, w, cin, cout) if (training and self.fuse_bn_stats) else 0
                stat_part = self._buf("bnpart_" + name, ((cout // 64) * rows * 128,)) if rows > 0 else None
                if (self.bf16_activations and self.bf16_storage and rows > 0 and r_out is None
                        and L.unet_conv3x3_wgrad_bf16_supported(n, h, w, cin, cout) == 1 and self._use_bf16(name, n, h, w, dgrad=True)):      # = _dz16(name): the BatchNorm backward that reads r takes bf16
                    r = self._buf("r16_" + name, (n, h, w, cout), torch.bfloat16)
                self._timed("conv3x3_fwd_bf16", 2.0 * 9 * n * h * w * cin * cout, L.unet_conv3x3_fwd_bf16,
                            _p(x), _ld(x), int(x.dtype == torch.bfloat16), None, None, _p(self._bf16_kernels(name)[0]), _p(b_), _p(r), _ld(r),
                            int(r.dtype == torch.bfloat16), n, h, w, cin, cout, 1, _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, st)
                if rows > 0:
                    fused_stats = (stat_part, rows)
            elif self._use_fused(name, h, w):
                rows = L.unet_conv3x3_fwd_winograd_fused_stats_rows(n, h, w, cin, cout) if (training and self.fuse_bn_stats) else 0
                stat_part = self._buf("bnpart_" + name, ((cout // 64) * rows * 128,)) if rows > 0 else None
                if in_view is not None:
                    # BatchNorm-apply on load: scaled weight transform, folded bias, per-channel padding value (this step's coefficients)
                    uc, bias_eff, pad = self._fold_buffers(name)
                    if training or name not in self._eval_folded:
                        # (inference: the coefficients come from the moving statistics -- constants until the parameters change -- so
                        # the fold of one forward serves every later tile)
                        nbf = L.unet_winograd_weight_fold_workspace(cin, cout)
                        L.unet_winograd_weight_fold(_p(w_), _p(b_), _p(in_view[0]), _p(in_view[1]), _p(uc), _p(bias_eff), _p(pad), cin, cout,
                                                    _p(self._workspace(nbf)), nbf, st)
                        if training:
                            self._eval_folded.discard(name)
                        else:
                            self._eval_folded.add(name)
                else:
                    uc, bias_eff, pad = self._fused_kernels(name)[0], b_, None
                # (with stat_part the conv kernel also leaves the BatchNorm sums of its output: one activation read less per layer)
                self._timed("conv3x3_fwd_winograd_fused", 2.0 * 9 * n * h * w * cin * cout, L.unet_conv3x3_fwd_winograd_fused,
                            _p(x), _ld(x), _p(pad), _p(uc), _p(bias_eff), _p(r), _ld(r), n, h, w, cin, cout, 1,
                            _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, st)
                if rows > 0:
                    fused_stats = (stat_part, rows)
            elif L.unet_conv3x3_mfma_supported(cin, cout):
                self._timed("conv3x3_fwd", 2.0 * 9 * n * h * w * cin * cout, L.unet_conv3x3_fwd_mfma,
                            _p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout, 1, st)
            else:
                rows = L.unet_conv3x3_fwd_direct_stats_rows(n, h, w, cin, cout) if (training and self.fuse_bn_stats) else 0
                if rows > 0:
                    stat_part = self._buf("bnpart_" + name, ((cout // 64) * rows * 128,))
                    if (self.compute_dtype == "bf16" and self.bf16_storage and self.bf16_activations and self.bf16_edge_activations
                            and r_out is None and cout % 8 == 0):
                        r = self._buf("r16_" + name, (n, h, w, cout), torch.bfloat16)
                    L.unet_conv3x3_fwd_direct_stats(_p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), int(r.dtype == torch.bfloat16), n, h, w, cin, cout, 1,
                                                    _p(stat_part), stat_part.numel() * 4, st)
                    fused_stats = (stat_part, rows)
                else:
                    L.unet_conv3x3_fwd_direct(_p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout, 1, st)
        P = r.shape[0] * r.shape[1] * r.shape[2]
        s = self.stat[name]
        sc_out, sh_out = stat_out if stat_out is not None else (s[2], s[3])
        gm, bt = self.p[name + "/gamma"], self.p[name + "/beta"]
        mm, mv = self.moving[name + "/moving_mean"], self.moving[name + "/moving_var"]
        if training and fused_stats is not None:
            L.unet_bn_train_finalize_partials(_p(fused_stats[0]), fused_stats[1], P, cout, _p(gm), _p(bt), BN_EPS, BN_MOMENTUM,
                                              BN_MOVING_VAR_UNBIASED, _p(mm), _p(mv), _p(s[0]), _p(s[1]), _p(sc_out), _p(sh_out), st)
        elif training:
            nb = L.unet_bn_workspace(P, cout)
            ws = self._workspace(nb)
            L.unet_bn_train_stats(_p(r), _ld(r), P, cout, _p(gm), _p(bt), BN_EPS, BN_MOMENTUM, BN_MOVING_VAR_UNBIASED,
                                  _p(mm), _p(mv), _p(s[0]), _p(s[1]), _p(sc_out), _p(sh_out), _p(ws), nb, st)
        elif (name, sc_out.data_ptr()) not in self._eval_coefs:             # (inference: constants until the parameters change)
            L.unet_bn_eval_coeffs(_p(gm), _p(bt), _p(mm), _p(mv), BN_EPS, cout, _p(sc_out), _p(sh_out), st)
            self._eval_coefs.add((name, sc_out.data_ptr()))
        if training:
            self._eval_coefs.clear()            # the batch statistics' coefficients went into the same buffers
        self.saved[name] = (x, r)
        self.view[name] = in_view
        self.coef[name] = (sc_out, sh_out)
        if y_out is None:                       # deferred: the consumer applies (sc_out, sh_out) on load
            return r
        if r.dtype == torch.bfloat16 or y_out.dtype == torch.bfloat16:
            L.unet_bn_apply_any(_p(r), _ld(r), int(r.dtype == torch.bfloat16), _p(sc_out), _p(sh_out), _p(y_out), _ld(y_out), int(y_out.dtype == torch.bfloat16),
                                _p(pool[0]) if pool is not None else None, cout, _p(pool[1]) if pool is not None else None,
                                r.shape[0], r.shape[1], r.shape[2], cout, st)
        elif pool is not None:         # (pooled, idx): BN apply and the level's max pool in one pass
            L.unet_bn_apply_maxpool(_p(r), _ld(r), _p(sc_out), _p(sh_out), _p(y_out), _ld(y_out), _p(pool[0]), cout, _p(pool[1]),
                                    r.shape[0], r.shape[1], r.shape[2], cout, st)
        else:
            L.unet_bn_apply(_p(r), _ld(r), _p(sc_out), _p(sh_out), _p(y_out), _ld(y_out), P, cout, st)
        return y_out

    def _dropout(self, t, key, masks, backward=False):
        n, h, w, c = t.shape
        m = None
        if masks is not None:
            m = masks[key]
        seed = (self.dropout_seed * 1000003 + self.iterations * 2 + (0 if key == "drop_4" else 1)) & 0xFFFFFFFF
        self.L.unet_dropout(_p(t), _ld(t), _p(t), _ld(t), n * h * w, c, _p(m), seed, DROPOUT_RATE, int(t.dtype == torch.bfloat16), self._stream())

    def _prep_masks(self, dropout_masks):
        """NCHW 0/1 arrays (the oracle's convention) -> dense NHWC uint8 device tensors."""
        if dropout_masks is None:
            return None
        out = {}
        for k, v in dropout_masks.items():
            a = np.ascontiguousarray(np.asarray(v).transpose(0, 2, 3, 1)).astype(np.uint8)
            out[k] = torch.as_tensor(a).to(self.dev)
        return out

    def forward(self, images, training=False, dropout_masks=None, labels=None, global_batch_size=None,
                label_smoothing=0.0, want_grad=False):
        """images: fp32 [N,C,H,W] (reference input contract, UNet/imagereader.py:298-300).  Returns softmax [N,H,W,K]
        (a device tensor owned by the engine; valid until the next call).  With labels (int32 one-hot [N,H,W,K]) the
        loss and the count of correctly classified pixels land in self.loss_buf."""
        L, st = self.L, self._stream()
        x = images
        assert x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == self.C, "images must be fp32 [N,C,H,W]"
        x = x.to(self.dev).contiguous()
        n, c, h, w = x.shape
        if h % SIZE_FACTOR or w % SIZE_FACTOR:
            raise IOError("Input Image tile size must be a multiple of %d" % SIZE_FACTOR)   # cf. UNet/inference.py:39-40
        if c == 1:
            x0 = x.view(n, h, w, 1)
        else:
            x0 = self._buf("x_nhwc", (n, h, w, c))
            L.unet_nchw_to_nhwc(_p(x), _p(x0), n, c, h, w, st)
        self.saved = {}
        self.view = {}
        self.coef = {}
        self.masks = self._prep_masks(dropout_masks) if training else None
        B = BASE
        f = self._block_fwd
        cur, cur_view = x0, None
        self.idx = {}
        self.cat = {}
        self.catstat = {}

        def pair(a_name, b_name, xin, xin_view, hh, ww, ch, b_out, **kw):
            """the two convs of a block: a -> b.  a's BatchNorm output feeds only b, so with BatchNorm-apply on load it is never
            materialised: b reads a's conv output through (scale, shift)."""
            if self._can_defer(a_name, b_name, n, hh, ww):
                ra = f(a_name, xin, None, training, in_view=xin_view)
                sa = self.stat[a_name]
                return f(b_name, ra, b_out, training, in_view=(sa[2], sa[3]), **kw)
            ya = f(a_name, xin, self._ybuf(a_name, (n, hh, ww, ch), b_name), training, in_view=xin_view)
            return f(b_name, ya, b_out, training, **kw)

        for lvl, ch in ((1, B), (2, 2 * B), (3, 4 * B), (4, 8 * B)):
            hh, ww = cur.shape[1], cur.shape[2]
            # the concat buffer [skip, upsampled] and the pooled tensor feed 3x3 layers only (dec_Na / the next level's first conv):
            # bf16 storage applies to them as well (levels 1-3; level 4 goes through the dropout and the unfused pool kernels)
            nxt = "conv_%da" % (lvl + 1) if lvl < 4 else "bott_a"
            c16 = (self.compute_dtype == "bf16" and self.bf16_storage and self.bf16_storage_cat and self.fuse_pool
                   and (lvl < 4 or (self.bf16_activations and self.bf16_level4))     # level 4: its dropout / unfused pool kernels take bf16 too
                   and self._use_bf16("dec_%da" % lvl, n, hh, ww) and self._use_bf16(nxt, n, hh // 2, ww // 2)
                   and L.unet_conv3x3_wgrad_bf16_supported(n, hh, ww, 2 * ch, ch) == 1)
            cat = self._buf(("cat16_%d" if c16 else "cat_%d") % lvl, (n, hh, ww, 2 * ch), torch.bfloat16 if c16 else torch.float32)
            self.cat[lvl] = cat
            pooled = self._buf(("pool16_%d" if c16 else "pool_%d") % lvl, (n, hh // 2, ww // 2, ch), torch.bfloat16 if c16 else torch.float32)
            idx = self._buf("idx_%d" % lvl, (n, hh // 2, ww // 2, ch), torch.uint8)
            fuse_pool = self.fuse_pool and not (lvl == 4 and training)   # level 4 drops out between BN and pool (UNet/model.py:105-107)
            skip = pair("conv_%da" % lvl, "conv_%db" % lvl, cur, cur_view, hh, ww, ch, cat[..., :ch], pool=(pooled, idx) if fuse_pool else None)
            if not fuse_pool:
                if lvl == 4 and training:
                    self._dropout(skip, "drop_4", self.masks)
                assert skip.dtype == pooled.dtype
                L.unet_maxpool2x2_fwd(_p(skip), _ld(skip), _p(pooled), ch, _p(idx), n, hh, ww, ch, int(skip.dtype == torch.bfloat16), st)
            self.idx[lvl] = idx
            cur, cur_view = pooled, None
        hh, ww = cur.shape[1], cur.shape[2]
        yb = self._ybuf("bott_b", (n, hh, ww, 16 * B), "up_4") if self.bf16_level4 else self._buf("y_bott_b", (n, hh, ww, 16 * B))
        cur = pair("bott_a", "bott_b", cur, None, hh, ww, 16 * B, yb)
        if training:
            self._dropout(cur, "drop_b", self.masks)
        for lvl, ch in ((4, 8 * B), (3, 4 * B), (2, 2 * B), (1, B)):
            cat = self.cat[lvl]
            hh, ww = cat.shape[1], cat.shape[2]
            cat_view = None
            if self._can_defer("up_%d" % lvl, "dec_%da" % lvl, n, hh, ww):
                # the transposed conv writes its output r straight into the upper half of the concat buffer; dec_Na reads the whole
                # buffer through per-channel coefficients: (1, 0) for the materialised skip half, up_N's BatchNorm for the upper half
                cs = self.bufs.get("catstat_%d" % lvl)
                if cs is None:
                    cs = self._buf("catstat_%d" % lvl, (2, 2 * ch))
                    cs[0, :ch].fill_(1.0); cs[1, :ch].zero_()
                self.catstat[lvl] = cs
                f("up_%d" % lvl, cur, None, training, r_out=cat[..., ch:], stat_out=(cs[0, ch:], cs[1, ch:]))
                cat_view = (cs[0], cs[1])
            else:
                f("up_%d" % lvl, cur, cat[..., ch:], training)
            if lvl > 1:
                yb = self._ybuf("dec_%db" % lvl, (n, hh, ww, ch), "up_%d" % (lvl - 1))
            elif training and want_grad and self.compute_dtype == "bf16" and self.bf16_storage and self.bf16_activations and self.bf16_edge_activations:
                # the class-map conv computes in fp32: storing its input as bf16 changes the result, so only the training step does it
                # (bf16 activation storage, Keras mixed_bfloat16 semantics); inference keeps the fp32 tensor
                yb = self._buf("y16_dec_1b", (n, hh, ww, ch), torch.bfloat16)
            else:
                yb = self._buf("y_dec_1b", (n, hh, ww, ch))
            cur = pair("dec_%da" % lvl, "dec_%db" % lvl, cat, cat_view, hh, ww, ch, yb)
        yl = f("logits", cur, self._buf("y_logits", (n, h, w, self.K)), training)
        prob = self._buf("softmax", (n, h, w, self.K))
        P = n * h * w
        nb = L.unet_softmax_ce_workspace(P)
        ws = self._workspace(nb)
        if labels is None:
            L.unet_softmax_ce(_p(yl), self.K, None, _p(prob), None, 0, P, self.K, 0.0, 0.0, 0.0, 0.0, None, None, _p(ws), nb, st)
        else:
            lab = labels.to(self.dev).contiguous()
            assert lab.dtype == torch.int32 and tuple(lab.shape) == (n, h, w, self.K), "labels must be int32 one-hot [N,H,W,K]"
            G = global_batch_size if global_batch_size else n
            scale = 1.0 / (float(G) * h * w)                       # sum_n / G then mean over H,W  (UNet/model.py:213-215)
            dl = self._buf("dy_logits", (n, h, w, self.K)) if want_grad else None
            L.unet_softmax_ce(_p(yl), self.K, _p(lab), _p(prob), _p(dl), self.K, P, self.K, float(label_smoothing), scale,
                              scale, float(self.ce_clip_eps), _p(self.loss_buf[0:1]), _p(self.loss_buf[1:2]), _p(ws), nb, st)
            self._labels_keepalive = lab
        return prob

    # ------------------------------------------------------------------------------------------------ backward
    def _block_bwd(self, name, dy, need_dx=True, eval_mode=False, pool_grad=None):
        """dy: NHWC view = gradient w.r.t. the layer's BN output.  Returns gradient w.r.t. the layer input (or None).
        eval_mode: BN used its moving statistics (an affine map) and no parameter gradients are wanted."""
        L, st = self.L, self._stream()
        kind, cin, cout = self.kind[name], self.cin[name], self.cout[name]
        x, r = self.saved[name]
        n, ho, wo, _ = r.shape
        P = n * ho * wo
        s = self.stat[name]
        dz16 = self._dz16(name, need_dx, eval_mode)
        dz = self._buf("dz16_" + name, tuple(r.shape), torch.bfloat16) if dz16 else self._buf("dz_" + name, tuple(r.shape))
        pre = self.bnbwd_part.pop(name, None) if not eval_mode else None
        if dz16 or (not eval_mode and (dy.dtype == torch.bfloat16 or r.dtype == torch.bfloat16)):
            # one entry point for the three forms (plain / pooled / sums from the consumer's data gradient), any of dy / r / dz stored as bf16
            part_ptr, rows = None, 0
            if pre is not None:
                part, rows, c0 = pre
                part_ptr = ctypes.c_void_p(part.data_ptr() + (c0 // 64) * rows * 128 * 4)
            pdy, pidx = pool_grad if (pool_grad is not None and pre is None) else (None, None)
            assert not (pool_grad is not None and pre is not None)
            nb = L.unet_bn_workspace(P, cout)
            ws = self._workspace(nb)
            L.unet_bn_bwd_any(_p(dy), _ld(dy), _p(pdy), _ld(pdy) if pdy is not None else 0, _p(pidx), n, ho, wo, _p(r), _ld(r),
                              _p(self.p[name + "/gamma"]), _p(s[0]), _p(s[1]), cout, 0 if kind == "deconv" else 1, _p(dz), cout, int(dz.dtype == torch.bfloat16),
                              _p(self.g[name + "/gamma"]), _p(self.g[name + "/beta"]), _p(self.g[name + "/bias"]), part_ptr, rows,
                              _p(ws), nb, st, int(r.dtype == torch.bfloat16), int(dy.dtype == torch.bfloat16),
                              int(pdy is not None and pdy.dtype == torch.bfloat16))
        elif eval_mode:
            L.unet_bn_eval_bwd(_p(dy), _ld(dy), _p(r), _ld(r), _p(self.coef[name][0]), _p(dz), cout, P, cout, 0 if kind == "deconv" else 1, st)
        elif pre is not None:
            # sum(dy), sum(dy*r) already came out of the consumer layer's data-gradient kernel: no reduction pass
            part, rows, c0 = pre
            nb = L.unet_bn_workspace(P, cout)
            ws = self._workspace(nb)
            L.unet_bn_bwd_from_partials(_p(dy), _ld(dy), _p(r), _ld(r), _p(self.p[name + "/gamma"]), _p(s[0]), _p(s[1]), P, cout,
                                        0 if kind == "deconv" else 1, _p(dz), cout, _p(self.g[name + "/gamma"]), _p(self.g[name + "/beta"]),
                                        _p(self.g[name + "/bias"]), ctypes.c_void_p(part.data_ptr() + (c0 // 64) * rows * 128 * 4), rows,
                                        _p(ws), nb, st)
        elif pool_grad is not None:
            # dy = skip gradient + un-pooled gradient of the level below, formed inside the BatchNorm-backward kernels
            pdy, pidx = pool_grad
            nb = L.unet_bn_workspace(P, cout)
            ws = self._workspace(nb)
            L.unet_bn_bwd_pooled(_p(dy), _ld(dy), _p(pdy), _ld(pdy), _p(pidx), n, ho, wo, _p(r), _ld(r), _p(self.p[name + "/gamma"]),
                                 _p(s[0]), _p(s[1]), cout, 1, _p(dz), cout, _p(self.g[name + "/gamma"]), _p(self.g[name + "/beta"]),
                                 _p(self.g[name + "/bias"]), _p(ws), nb, st)
        else:
            nb = L.unet_bn_workspace(P, cout)
            ws = self._workspace(nb)
            L.unet_bn_bwd(_p(dy), _ld(dy), _p(r), _ld(r), _p(self.p[name + "/gamma"]), _p(s[0]), _p(s[1]), P, cout,
                          0 if kind == "deconv" else 1, _p(dz), cout, _p(self.g[name + "/gamma"]), _p(self.g[name + "/beta"]),
                          _p(self.g[name + "/bias"]), _p(ws), nb, st)
        w_, dw = self.p[name + "/kernel"], self.g[name + "/kernel"]
        hi, wi = x.shape[1], x.shape[2]
        dx = None

        def wgrad():
            sd = self.overlap_wgrad
            st2 = self._stream()
            if kind == "deconv" and self._use_bf16_convt(name, n, hi, wi) and L.unet_convT2x2_wgrad_bf16_supported(n, hi, wi, cin, cout) == 1:
                nb2 = L.unet_convT2x2_wgrad_bf16_workspace(n, hi, wi, cin, cout)
                L.unet_convT2x2_wgrad_bf16(_p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(dz), cout, int(dz.dtype == torch.bfloat16), _p(dw),
                                              n, hi, wi, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
            elif kind == "deconv":
                nb2 = L.unet_convT2x2_wgrad_workspace(n, hi, wi, cin, cout)
                L.unet_convT2x2_wgrad(_p(x), _ld(x), _p(dz), cout, _p(dw), n, hi, wi, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
            elif kind == "conv1":
                nb2 = L.unet_conv1x1_wgrad_workspace(P, cin, cout)
                L.unet_conv1x1_wgrad(_p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(dz), cout, _p(dw), P, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
            elif (self.compute_dtype == "bf16" and n * ho * wo * max(cin, cout) * 4 < 2 ** 31
                  and L.unet_conv3x3_wgrad_bf16_supported(n, ho, wo, cin, cout) == 1):
                nb2 = L.unet_conv3x3_wgrad_bf16_workspace(n, ho, wo, cin, cout)
                self._timed("conv3x3_wgrad_bf16", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_wgrad_bf16,
                            _p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(dz), cout, int(dz.dtype == torch.bfloat16), _p(dw),
                            n, ho, wo, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
            elif self.wgrad_route == "fused" and L.unet_winograd_wgrad_fused_supported(n, ho, wo, cin, cout) == 1:
                nb2 = L.unet_conv3x3_wgrad_winograd_fused_workspace(n, ho, wo, cin, cout)
                self._timed("conv3x3_wgrad_winograd_fused", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_wgrad_winograd_fused,
                            _p(x), _ld(x), _p(dz), cout, _p(dw), n, ho, wo, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
                vw = self.view.get(name)
                if vw is not None:
                    # x was read through BatchNorm-apply on load: dw (computed on the producer's raw conv output) -> scale . dw + shift (x) S
                    nb3 = L.unet_conv3x3_wgrad_fold_fix_workspace(cout)
                    L.unet_conv3x3_wgrad_fold_fix(_p(dw), _p(vw[0]), _p(vw[1]), _p(dz), cout, _p(self.g[name + "/bias"]), n, ho, wo, cin, cout,
                                                  _p(self._workspace(nb3, sd)), nb3, st2)
            elif L.unet_conv3x3_mfma_supported(cin, cout) and cin % 64 == 0:
                nb2 = L.unet_conv3x3_wgrad_mfma_workspace(n, ho, wo, cin, cout)
                self._timed("conv3x3_wgrad", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_wgrad_mfma,
                            _p(x), _ld(x), _p(dz), cout, _p(dw), n, ho, wo, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
            else:
                nb2 = L.unet_conv3x3_wgrad_direct_workspace(n, ho, wo, cin, cout)
                L.unet_conv3x3_wgrad_direct(_p(x), _ld(x), _p(dz), cout, int(dz.dtype == torch.bfloat16), _p(dw), n, ho, wo, cin, cout,
                                            _p(self._workspace(nb2, sd)), nb2, st2)
            if self.on_layer_grads_ready is not None:
                self.on_layer_grads_ready(name)          # under the stream the gradients were produced on

        if self.overlap_wgrad and not eval_mode:
            self.side.wait_stream(torch.cuda.current_stream())       # dz (and this layer's bias/gamma/beta grads) ready
            with torch.cuda.stream(self.side):
                wgrad()
        if need_dx:
            if self._dx16(name, eval_mode):
                dx = self._buf("dy16_in_" + name, (n, hi, wi, cin), torch.bfloat16)
            else:
                dx = self._buf("dy_in_" + name, (n, hi, wi, cin))
            if kind == "deconv" and self._use_bf16_convt(name, n, hi, wi):
                prod = PRODUCER.get(name) if (self.fuse_bn_stats and not eval_mode) else None
                rows = L.unet_convT2x2_bf16_stats_rows(n, hi, wi, cin, cout, 1) if prod else 0
                r_prev = self.saved[prod[0]][1] if rows > 0 else None
                part = self._buf("bnbwd_" + name, ((cin // 64) * rows * 128,)) if rows > 0 else None
                L.unet_convT2x2_dgrad_bf16(_p(dz), cout, int(dz.dtype == torch.bfloat16), _p(self._bf16_kernels(name)[1]), _p(dx), cin,
                                           int(dx.dtype == torch.bfloat16), n, hi, wi, cin, cout, _p(r_prev), _ld(r_prev) if rows > 0 else 0,
                                           int(rows > 0 and r_prev.dtype == torch.bfloat16), _p(part), part.numel() * 4 if rows > 0 else 0, st)
                if rows > 0:
                    self.bnbwd_part[prod[0]] = (part, rows, 0)
            elif kind == "deconv":
                L.unet_convT2x2_dgrad(_p(dz), cout, _p(w_), _p(dx), cin, n, hi, wi, cin, cout, st)
            elif kind == "conv1":
                L.unet_conv1x1_dgrad(_p(dz), cout, _p(w_), _p(dx), cin, int(dx.dtype == torch.bfloat16), P, cin, cout, st)
            elif self._use_bf16(name, n, ho, wo, dgrad=True):
                prod = PRODUCER.get(name) if (self.fuse_bn_stats and not eval_mode) else None
                rows = L.unet_conv3x3_bf16_stats_rows(n, ho, wo, cout, cin) if prod else 0
                z16 = int(dz.dtype == torch.bfloat16)
                if rows > 0:
                    pname, c0, c1 = prod[0], prod[1] * (cin // prod[3]), prod[2] * (cin // prod[3])
                    r_prev = self.saved[pname][1]
                    part = self._buf("bnbwd_" + name, ((cin // 64) * rows * 128,))
                    self._timed("conv3x3_dgrad_bf16", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_dgrad_bf16,
                                _p(dz), cout, z16, _p(self._bf16_kernels(name)[1]), _p(dx), cin, int(dx.dtype == torch.bfloat16), n, ho, wo, cin, cout,
                                _p(r_prev), _ld(r_prev), int(r_prev.dtype == torch.bfloat16), c0, c1, _p(part), part.numel() * 4, st)
                    self.bnbwd_part[pname] = (part, rows, c0)
                else:
                    self._timed("conv3x3_dgrad_bf16", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_dgrad_bf16,
                                _p(dz), cout, z16, _p(self._bf16_kernels(name)[1]), _p(dx), cin, int(dx.dtype == torch.bfloat16), n, ho, wo, cin, cout,
                                None, 0, 0, 0, 0, None, 0, st)
            elif self._use_fused(name, ho, wo, dgrad=True):
                prod = PRODUCER.get(name) if (self.fuse_bn_stats and not eval_mode) else None
                if prod is not None and self.saved[prod[0]][1].dtype != torch.float32:
                    prod = None          # (a size-fallback fp32 layer behind a bf16-storing producer: the producer's BatchNorm backward runs its own reduction)
                rows = L.unet_conv3x3_fwd_winograd_fused_stats_rows(n, ho, wo, cout, cin) if prod else 0
                if rows > 0:
                    # dx (or a channel range of it) is the dy of the producer layer's BatchNorm: leave its backward sums too
                    pname, c0, c1 = prod[0], prod[1] * (cin // prod[3]), prod[2] * (cin // prod[3])
                    r_prev = self.saved[pname][1]
                    part = self._buf("bnbwd_" + name, ((cin // 64) * rows * 128,))
                    self._timed("conv3x3_dgrad_winograd_fused", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_dgrad_winograd_fused,
                                _p(dz), cout, _p(self._fused_kernels(name)[1]), _p(dx), cin, n, ho, wo, cin, cout,
                                _p(r_prev), _ld(r_prev), c0, c1, _p(part), part.numel() * 4, st)
                    self.bnbwd_part[pname] = (part, rows, c0)
                else:
                    self._timed("conv3x3_dgrad_winograd_fused", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_dgrad_winograd_fused,
                                _p(dz), cout, _p(self._fused_kernels(name)[1]), _p(dx), cin, n, ho, wo, cin, cout,
                                None, 0, 0, 0, None, 0, st)
            elif L.unet_conv3x3_mfma_supported(cout, cin):
                self._timed("conv3x3_dgrad", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_dgrad_mfma,
                            _p(dz), cout, _p(w_), _p(dx), cin, n, ho, wo, cin, cout, st)
            else:                                   # first layer (Cin = number_channels): only the ERF probe needs it
                L.unet_conv3x3_dgrad_direct(_p(dz), cout, _p(w_), _p(dx), cin, n, ho, wo, cin, cout, st)
        if not self.overlap_wgrad and not eval_mode:
            wgrad()
        return dx

    def _dz16(self, name, need_dx=True, eval_mode=False):
        """the layer's BatchNorm backward takes the unified bf16-capable entry point (dz stored as bf16; dy / r may be bf16)"""
        if self.compute_dtype != "bf16" or not self.bf16_storage or eval_mode or name not in self.saved:
            return False
        L = self.L
        kind, cin, cout = self.kind[name], self.cin[name], self.cout[name]
        x, r = self.saved[name]
        n, ho, wo, _ = r.shape
        if kind == "conv3":
            if (not need_dx and self.bf16_activations and self.bf16_edge_activations and cin % 64 != 0 and cout % 8 == 0
                    and r.dtype == torch.bfloat16):
                return True              # the first layer: its weight gradient (the fp32 stencil kernel) reads dz in either storage
            return (n * ho * wo * max(cin, cout) * 4 < 2 ** 31 and L.unet_conv3x3_wgrad_bf16_supported(n, ho, wo, cin, cout) == 1
                    and (not need_dx or self._use_bf16(name, n, ho, wo, dgrad=True)))
        if kind == "deconv":
            return (self._use_bf16_convt(name, n, x.shape[1], x.shape[2])
                    and L.unet_convT2x2_wgrad_bf16_supported(n, x.shape[1], x.shape[2], cin, cout) == 1)
        return False

    def _dz16_shape(self, name, n, h, w):
        """(transposed conv, input size h x w) its BatchNorm backward will take the bf16-capable entry point"""
        return (self.compute_dtype == "bf16" and self.bf16_storage and self._use_bf16_convt(name, n, h, w)
                and self.L.unet_convT2x2_wgrad_bf16_supported(n, h, w, self.cin[name], self.cout[name]) == 1)

    def _dx16(self, name, eval_mode):
        """the 3x3 layer's data gradient may be WRITTEN as bf16: every reader of it is a BatchNorm backward that takes bf16 dy"""
        if self.compute_dtype != "bf16" or not (self.bf16_storage and self.bf16_activations) or eval_mode:
            return False
        if self.kind[name] == "conv1":                                  # class map: its input gradient is dec_1b's dy
            return self.bf16_edge_activations and self.cin[name] % 8 == 0 and self._dz16("dec_1b", True, eval_mode)
        if self.kind[name] == "deconv":
            # dx is the dy of the producer (dec_(N+1)b): bf16 when that layer's BatchNorm backward takes bf16 dy and this kernel leaves its sums
            n, hi, wi, _ = self.saved[name][0].shape
            prod = PRODUCER.get(name) if self.fuse_bn_stats else None
            return (prod is not None and self.bf16_convt_activations and self._use_bf16_convt(name, n, hi, wi)
                    and self.L.unet_convT2x2_bf16_stats_rows(n, hi, wi, self.cin[name], self.cout[name], 1) > 0
                    and self._dz16(prod[0], True, eval_mode))
        if self.kind[name] != "conv3":
            return False
        n, ho, wo, _ = self.saved[name][1].shape
        if not self._use_bf16(name, n, ho, wo, dgrad=True):            # the fp32 kernels (size fallback) write fp32
            return False
        # a reader takes bf16 dy when its BatchNorm backward goes through the unified entry point: its dz is bf16, or (the first layer, whose
        # weight gradient is the fp32 stencil kernel) its saved conv output is
        need1 = lambda nm: self._dz16(nm, nm != "conv_1a", eval_mode) or (nm in self.saved and self.saved[nm][1].dtype == torch.bfloat16)
        if name.startswith("dec_") and name.endswith("a"):             # [skip, upsampled]: conv_Nb (through the fused pool path) and up_N
            lvl = int(name[4])
            if lvl == 4:     # skip half -> pool-backward accumulate + dropout kernels (either storage), then conv_4b's BatchNorm backward
                return self._lvl4_grad16() and need1("up_4")
            return self.fuse_pool and need1("conv_%db" % lvl) and need1("up_%d" % lvl)
        if name.startswith("conv_") and name.endswith("a"):            # pooled gradient of the level above, formed inside its BatchNorm backward
            lvl = int(name[5])
            return lvl >= 2 and self.fuse_pool and need1("conv_%db" % (lvl - 1))
        if name == "bott_a":                                           # read by the separate pool-backward kernel, added into dec_4a's skip gradient
            return self._lvl4_grad16()
        prod = PRODUCER.get(name)
        return prod is not None and prod[3] == 1 and need1(prod[0])

    def _lvl4_grad16(self):
        """dec_4a's and bott_a's data gradients are both bf16 or both fp32: the pool-backward kernel adds one into the other"""
        if not (self.bf16_level4 and "dec_4a" in self.saved and "bott_a" in self.saved):
            return False
        ok = True
        for nm in ("dec_4a", "bott_a"):
            n, ho, wo, _ = self.saved[nm][1].shape
            ok = ok and self._use_bf16(nm, n, ho, wo, dgrad=True)
        return ok and self.saved["dec_4a"][0].dtype == torch.bfloat16        # the forward took the bf16 level-4 tensors

    def input_gradient_eval(self, dprob):
        """After forward(training=False): gradient of sum(dprob * softmax) w.r.t. the input image, fp32 [N,C,H,W].
        (tf.GradientTape().gradient(loss, img) with the model in eval mode, reference UNet/model.py:176-184.)"""
        prob = self.bufs["softmax"]
        n, h, w, k = prob.shape
        g = dprob.to(self.dev).contiguous()
        assert g.dtype == torch.float32 and tuple(g.shape) == (n, h, w, k)
        dl = self._buf("dy_logits", (n, h, w, k))
        self.L.unet_softmax_bwd(_p(prob), _p(g), _p(dl), k, n * h * w, k, self._stream())
        dimg = self.backward(eval_mode=True)
        return dimg.permute(0, 3, 1, 2).contiguous()

    def backward(self, eval_mode=False):
        """Gradients of the loss computed by the last forward(training=True, labels=..., want_grad=True) -> self.grad.
        eval_mode=True instead propagates `dy_logits` through the eval-mode graph down to the input image."""
        L, st = self.L, self._stream()
        b = lambda name, dy, need_dx=True: self._block_bwd(name, dy, need_dx=need_dx, eval_mode=eval_mode)
        B = BASE
        d = b("logits", self.bufs["dy_logits"])
        for lvl, ch in ((1, B), (2, 2 * B), (3, 4 * B), (4, 8 * B)):
            d = b("dec_%db" % lvl, d)
            dcat = b("dec_%da" % lvl, d)                       # [N,H,W,2ch]: [0,ch) skip grad, [ch,2ch) upsampled grad
            self.bufs["dcat_%d" % lvl] = dcat
            d = b("up_%d" % lvl, dcat[..., ch:])
        if not eval_mode:
            self._dropout(d, "drop_b", self.masks)
        d = b("bott_b", d)
        d = b("bott_a", d)
        for lvl, ch in ((4, 8 * B), (3, 4 * B), (2, 2 * B), (1, B)):
            dcat = self.bufs["dcat_%d" % lvl]
            ds = dcat[..., :ch]
            n, hh, ww, _ = ds.shape
            if self.fuse_pool and not eval_mode and lvl != 4 and ch % 4 == 0:
                d = self._block_bwd("conv_%db" % lvl, ds, pool_grad=(d, self.idx[lvl]))       # no separate pool-backward pass
            else:
                assert d.dtype == ds.dtype, (lvl, d.dtype, ds.dtype)
                L.unet_maxpool2x2_bwd(_p(d), _ld(d), _p(self.idx[lvl]), _p(ds), _ld(ds), n, hh, ww, ch, 1, int(ds.dtype == torch.bfloat16), st)
                if lvl == 4 and not eval_mode:
                    self._dropout(ds, "drop_4", self.masks)
                d = b("conv_%db" % lvl, ds)
            d = b("conv_%da" % lvl, d, need_dx=(lvl != 1 or eval_mode))
        if self.overlap_wgrad and not eval_mode:
            torch.cuda.current_stream().wait_stream(self.side)        # every weight gradient done before Adam
        return d

    def adam_step(self, learning_rate):
        """Keras Adam on the flat buffers (reference UNet/model.py:79,223)."""
        self.iterations += 1
        self._fused_dirty = True
        self._bf16_dirty = True
        self._eval_folded.clear(); self._eval_coefs.clear()
        t = self.iterations
        alpha = learning_rate * math.sqrt(1.0 - ADAM_BETA2 ** t) / (1.0 - ADAM_BETA1 ** t)
        self.L.unet_adam_keras(_p(self.theta), _p(self.grad), _p(self.adam_m), _p(self.adam_v), self.n_flat, alpha,
                               ADAM_BETA1, ADAM_BETA2, ADAM_EPS, self._stream())

    def argmax(self, prob):
        n, h, w, k = prob.shape
        out = self._buf("argmax", (n, h, w), torch.int32)
        self.L.unet_argmax(_p(prob), k, _p(out), n * h * w, k, self._stream())
        return out
