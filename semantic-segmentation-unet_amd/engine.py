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

import numpy as np
import torch

from ._lib import lib
from .plan import BF16, EngineOptions, PRODUCER, build_plan, layer_table      # noqa: F401  (layer_table / PRODUCER re-exported)

BASE = 64                      # UNet._BASELINE_FEATURE_DEPTH  (reference UNet/model.py:20)
SIZE_FACTOR = 16               # UNet.SIZE_FACTOR              (reference UNet/model.py:25)

# Keras defaults the reference relies on (SURVEY.md 8(a) "(K)"): one place to flip them.
BN_EPS = 1e-3
BN_MOMENTUM = 0.99
BN_MOVING_VAR_UNBIASED = 1
DROPOUT_RATE = 0.5
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-7
CE_CLIP_EPS = 0.0              # 0: CE from the softmax's logits (graph-mode Keras); 1e-7: Keras' clipped-probability path


def kernel_shape(kind, cin, cout):
    return {"conv3": (3, 3, cin, cout), "conv1": (1, 1, cin, cout), "deconv": (2, 2, cout, cin)}[kind]


BACKWARD_ORDER = ["logits", "dec_1b", "dec_1a", "up_1", "dec_2b", "dec_2a", "up_2", "dec_3b", "dec_3a", "up_3",
                  "dec_4b", "dec_4a", "up_4", "bott_b", "bott_a", "conv_4b", "conv_4a", "conv_3b", "conv_3a",
                  "conv_2b", "conv_2a", "conv_1b", "conv_1a"]


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
    def __init__(self, number_classes, number_channels, device="cuda", seed=0, options=None):
        self.K, self.C = number_classes, number_channels
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("the U-Net hot path runs on an MI355X only (device must be cuda); no CPU fallback exists")
        self.L = lib()
        # every switch of the schedule lives in ONE options object (plan.EngineOptions; UNET_* diagnostics variables are read once, here);
        # routes and storage precisions are decided per shape by plan.build_plan and only looked up below
        self.opt = options if options is not None else EngineOptions.from_env()
        self._plans = {}
        self.pl = None
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
        self.view = {}
        self._fused_U, self._fused_dirty = None, True
        self._x6_U, self._x6_dirty = None, True
        self._ctx6_W, self._ctx6_dirty = None, True
        self._bf16_W, self._bf16_dirty = {}, True
        self.bnbwd_part = {}
        self.init_parameters(seed)
        self.on_layer_grads_ready = None        # hook(name) for data-parallel bucketing (parallel.py)
        self.profile = None                     # bench.py: {"conv3x3_fwd": [(ev0, ev1, flops)], ...} when enabled
        # Backward runs two HIP streams (opt.overlap_wgrad): the critical chain dgrad(L) -> bn_bwd(L-1) -> dgrad(L-1) ... on the caller's
        # stream, every weight gradient on `side` (it is needed only by the all-reduce / Adam).
        self.side = torch.cuda.Stream(device=self.dev)
        self._ws_side = None

    # the arithmetic mode is the one option callers set after construction (model.UNet(compute_dtype=...))
    @property
    def compute_dtype(self):
        return self.opt.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, v):
        if v not in ("fp32", "bf16"):
            raise ValueError("compute_dtype must be 'fp32' or 'bf16'")
        self.opt.compute_dtype = v

    @property
    def overlap_wgrad(self):
        return self.opt.overlap_wgrad

    @overlap_wgrad.setter
    def overlap_wgrad(self, v):
        self.opt.overlap_wgrad = bool(v)

    def plan(self, n, h, w, training=False, want_grad=False):
        """The step plan for images [n, C, h, w] (plan.StepPlan): kernel family per layer and direction, storage precision per tensor.
        Built once per shape / mode / option set and cached."""
        key = (n, h, w, bool(training), bool(want_grad), self.opt.key())
        pl = self._plans.get(key)
        if pl is None:
            pl = self._plans[key] = build_plan(self.opt, self.C, self.K, n, h, w, training, want_grad, self.L)
        return pl

    def _timed(self, key, work, fn, *args):
        """Call fn(*args); when profiling is on, bracket it with HIP events on the launch stream.  work: the launch's algorithmic FLOPs
        (matrix-core families) or algorithmic HBM bytes (the HBM-bound families: tensors read + written once, `_nb`)."""
        if self.profile is None:
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(*args)
        e1.record()
        self.profile.setdefault(key, []).append((e0, e1, work))

    @staticmethod
    def _nb(*tensors):
        """algorithmic bytes of NHWC views read or written once (the logical elements, not the leading dimension)"""
        return float(sum(t.numel() * t.element_size() for t in tensors if t is not None))

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

    def _x6_kernels(self, name, forward=False):
        """(U6 forward, U6 data gradient): the layer's Winograd-transformed kernel as three bf16 pieces per value (csrc/winograd_x6.hip).
        Buffers exist only for the layers and directions the CURRENT plan routes through BF16x6 (96 B per weight and direction: ~600 MB for
        the whole network, so a bf16-mode plan with one fp32 fallback layer allocates and re-splits that one layer).  After a parameter
        change ONE batched launch re-transforms every data-gradient operand in use and the forward operands of the layers the plan reads
        without BatchNorm-apply on load (the other layers' forward operands come out of unet_winograd_weight_fold_x6 every step:
        transforming them here as well was 39 % of the launch's bytes).  forward=True: the caller needs [0]; a forward operand the batch
        skipped (another plan of the same engine, e.g. bn_on_load off) is transformed on demand."""
        if self._x6_U is None:
            self._x6_U, self._x6_jobs, self._x6_fwd_valid, self._x6_valid_d = {}, {}, set(), set()
        pl = self.pl
        want_f = frozenset(n for n, p in pl.layer.items() if p.fwd_x6 and not p.x_on_load) | ({name} if forward else frozenset())
        want_d = frozenset(n for n, p in pl.layer.items() if p.dgrad_x6) | (frozenset() if forward else {name})
        for n in want_f | want_d:
            u = self._x6_U.setdefault(n, [None, None])
            for mode, want in ((0, want_f), (1, want_d)):
                if n in want and u[mode] is None:
                    u[mode] = torch.empty(self.L.unet_winograd_x6_weight_bytes(self.cin[n], self.cout[n]), dtype=torch.uint8, device=self.dev)
                    self._x6_dirty = True
        if self._x6_dirty:
            key = (want_f, want_d)
            job = self._x6_jobs.get(key)
            if job is None:
                rows, blk = [], 0
                for n in sorted(want_f | want_d):
                    cin, cout = self.cin[n], self.cout[n]
                    for mode, want in ((0, want_f), (1, want_d)):
                        if n in want:
                            rows.append([self.p[n + "/kernel"].data_ptr(), self._x6_U[n][mode].data_ptr(), cin | (cout << 32), blk, mode, 0])
                            blk += (cin * cout + 2047) // 2048
                job = self._x6_jobs[key] = (torch.tensor(rows, dtype=torch.int64, device=self.dev), blk)
            self.L.unet_winograd_weight_transform_x6_batch(_p(job[0]), job[0].shape[0], job[1], self._stream())
            self._x6_fwd_valid = set(want_f)
            self._x6_valid_d = set(want_d)
            self._x6_dirty = False
        if forward and name not in self._x6_fwd_valid:
            self.L.unet_winograd_weight_transform_x6(_p(self.p[name + "/kernel"]), _p(self._x6_U[name][0]), self.cin[name], self.cout[name], 0, self._stream())
            self._x6_fwd_valid.add(name)
        if not forward and name not in self._x6_valid_d:
            self.L.unet_winograd_weight_transform_x6(_p(self.p[name + "/kernel"]), _p(self._x6_U[name][1]), self.cin[name], self.cout[name], 1, self._stream())
            self._x6_valid_d.add(name)
        return self._x6_U[name]

    def _convt_x6_kernels(self, name):
        """(forward operand, data-gradient operand) of a transposed conv on the BF16x6 route: the layer's kernel as three bf16 pieces per
        weight in the GEMM operand layouts (csrc/convt_x6.hip); all layers and both directions in one launch after a parameter change."""
        if self._ctx6_W is None:
            self._ctx6_W, rows, blk = {}, [], 0
            for n, kind, cin, cout in self.layers:
                if kind != "deconv" or cin % 16 or cout % 16:
                    continue
                nb = self.L.unet_convT2x2_x6_weight_bytes(cin, cout)
                u = (torch.empty(nb, dtype=torch.uint8, device=self.dev), torch.empty(nb, dtype=torch.uint8, device=self.dev))
                self._ctx6_W[n] = u
                for mode in (0, 1):
                    rows.append([self.p[n + "/kernel"].data_ptr(), u[mode].data_ptr(), cin | (cout << 32), blk, mode, 0])
                    blk += (4 * cin * cout // 8 + 255) // 256
            self._ctx6_jobs = torch.tensor(rows, dtype=torch.int64, device=self.dev)
            self._ctx6_blocks = blk
            self._ctx6_dirty = True
        if self._ctx6_dirty:
            self.L.unet_convT2x2_weight_transform_x6_batch(_p(self._ctx6_jobs), self._ctx6_jobs.shape[0], self._ctx6_blocks, self._stream())
            self._ctx6_dirty = False
        return self._ctx6_W[name]

    def _fin_counters(self):
        """one device word per layer for unet_bn_finalize_apply_any (it only grows; the running targets live here, not in the library)"""
        if getattr(self, "_fin_cnt", None) is None:
            self._fin_cnt = torch.zeros(len(self.layers), dtype=torch.int32, device=self.dev)
            self._fin_index = {n: i for i, (n, _, _, _) in enumerate(self.layers)}
            self._fin_target = [0] * len(self.layers)
        return self._fin_cnt

    def parameters_changed(self):
        """theta was written from outside (broadcast, checkpoint): every cached transform of the kernels is stale."""
        self._fused_dirty = True
        self._x6_dirty = True
        self._ctx6_dirty = True
        self._bf16_dirty = True
        self._eval_folded.clear(); self._eval_coefs.clear()

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
    def _fold_buffers(self, name, x6=False):
        cin, cout = self.cin[name], self.cout[name]
        if x6:
            return (self._buf("U6fold_" + name, (self.L.unet_winograd_x6_weight_bytes(cin, cout),), torch.uint8),
                    self._buf("bfold_" + name, (cout,)), self._buf("pad_" + name, (cin + 8,)))
        return (self._buf("Ufold_" + name, (16 * cin * cout,)), self._buf("bfold_" + name, (cout,)), self._buf("pad_" + name, (cin + 8,)))

    def _block_fwd(self, name, x, y_out, training, pool=None, in_view=None, r_out=None, stat_out=None):
        """x: NHWC view (input of the layer), y_out: NHWC view the BN output is written to, or None: the BatchNorm output is NOT
        materialised (its consumer applies it on load) and the conv output r is returned instead.
        in_view = (scale, shift) per input channel: x is a producer's conv output (BatchNorm-apply on load, fused Winograd route);
        r_out: where the conv output goes (default: the layer's own buffer); stat_out = (scale, shift) destinations of the BatchNorm
        coefficients (default: self.stat[name][2:4]).  Kernel family and storage precisions: self.pl.layer[name]."""
        L, st = self.L, self._stream()
        cap = self.opt.cap                    # max_workgroups of every persistent kernel (0: one workgroup per CU)
        lp = self.pl.layer[name]
        kind, cin, cout = lp.kind, lp.cin, lp.cout
        n, h, w, _ = x.shape
        assert (h, w) == (lp.hi, lp.wi) and (in_view is not None) == lp.x_on_load and (y_out is None) == lp.defer_y, name
        w_, b_ = self.p[name + "/kernel"], self.p[name + "/bias"]
        fused_stats = None
        r16 = lp.r == BF16
        assert not (r16 and r_out is not None)
        if r_out is not None:
            r = r_out
        else:
            r = self._buf(("r16_" if r16 else "r_") + name, (n, lp.ho, lp.wo, cout), torch.bfloat16 if r16 else torch.float32)
        part = lambda rows: self._buf("bnpart_" + name, ((cout // 64) * rows * 128,)) if rows > 0 else None
        if lp.fwd == "convt_bf16":
            rows = L.unet_convT2x2_bf16_stats_rows(n, h, w, cin, cout, 0) if lp.fwd_stats else 0
            stat_part = part(rows)
            self._timed("convt_fwd_bf16", 8.0 * n * h * w * cin * cout, L.unet_convT2x2_fwd_bf16,
                        _p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(self._bf16_kernels(name)[0]), _p(b_), _p(r), _ld(r),
                        int(r16), n, h, w, cin, cout, _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, st)
            fused_stats = (stat_part, rows) if rows > 0 else None
        elif lp.fwd == "convt_x6":                                              # GEMM on the bf16 matrix pipe at fp32 grade (csrc/convt_x6.hip)
            rows = L.unet_convT2x2_x6_stats_rows(n, h, w, cin, cout) if lp.fwd_stats else 0
            stat_part = part(rows)
            self._timed("convt_fwd_x6", 8.0 * n * h * w * cin * cout, L.unet_convT2x2_fwd_x6,
                        _p(x), _ld(x), _p(self._convt_x6_kernels(name)[0]), _p(b_), _p(r), _ld(r), n, h, w, cin, cout,
                        _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, st)
            fused_stats = (stat_part, rows) if rows > 0 else None
        elif lp.fwd == "convt_stream" and _ld(x) <= 4096:                       # persistent stream kernel
            rows = L.unet_convT2x2_fwd_stream_stats_rows_wg(n, h, w, cin, cout, cap) if lp.fwd_stats else 0
            stat_part = part(rows)
            self._timed("convt_fwd", 8.0 * n * h * w * cin * cout, L.unet_convT2x2_fwd_stream_wg,
                        _p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout,
                        _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, cap, st)
            fused_stats = (stat_part, rows) if rows > 0 else None
        elif kind == "deconv":
            L.unet_convT2x2_fwd(_p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout, st)
        elif kind == "conv1":
            self._timed("classmap_fwd", self._nb(x, r), L.unet_conv1x1_fwd,
                        _p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(w_), _p(b_), _p(r), cout, n * h * w, cin, cout, 1, st)
        elif lp.fwd == "bf16":
            rows = L.unet_conv3x3_bf16_stats_rows(n, h, w, cin, cout) if lp.fwd_stats else 0
            stat_part = part(rows)
            self._timed("conv3x3_fwd_bf16", 2.0 * 9 * n * h * w * cin * cout, L.unet_conv3x3_fwd_bf16_wg,
                        _p(x), _ld(x), int(x.dtype == torch.bfloat16), None, None, _p(self._bf16_kernels(name)[0]), _p(b_), _p(r), _ld(r),
                        int(r16), n, h, w, cin, cout, 1, _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, cap, st)
            fused_stats = (stat_part, rows) if rows > 0 else None
        elif lp.fwd == "winograd":
            rows = L.unet_conv3x3_fwd_winograd_fused_stats_rows_wg(n, h, w, cin, cout, cap) if lp.fwd_stats else 0
            stat_part = part(rows)
            x6 = lp.fwd_x6
            if in_view is not None:
                # BatchNorm-apply on load: scaled weight transform, folded bias, per-channel padding value (this step's coefficients)
                uc, bias_eff, pad = self._fold_buffers(name, x6)
                if training or (name, x6) not in self._eval_folded:
                    # (inference: the coefficients come from the moving statistics -- constants until the parameters change -- so
                    # the fold of one forward serves every later tile)
                    if x6:
                        L.unet_winograd_weight_fold_x6(_p(w_), _p(b_), _p(in_view[0]), _p(in_view[1]), _p(uc), _p(bias_eff), _p(pad), cin, cout, st)
                    else:
                        nbf = L.unet_winograd_weight_fold_workspace(cin, cout)
                        L.unet_winograd_weight_fold(_p(w_), _p(b_), _p(in_view[0]), _p(in_view[1]), _p(uc), _p(bias_eff), _p(pad), cin, cout,
                                                    _p(self._workspace(nbf)), nbf, st)
                    if training:
                        self._eval_folded.discard((name, x6))
                    else:
                        self._eval_folded.add((name, x6))      # (keyed by the route too: the two routes fold into different buffers)
            else:
                uc, bias_eff, pad = (self._x6_kernels(name, forward=True)[0] if x6 else self._fused_kernels(name)[0]), b_, None
            # (with stat_part the conv kernel also leaves the BatchNorm sums of its output: one activation read less per layer)
            self._timed("conv3x3_fwd_winograd_x6" if x6 else "conv3x3_fwd_winograd_fused", 2.0 * 9 * n * h * w * cin * cout,
                        L.unet_conv3x3_fwd_winograd_x6_wg if x6 else L.unet_conv3x3_fwd_winograd_fused_wg,
                        _p(x), _ld(x), _p(pad), _p(uc), _p(bias_eff), _p(r), _ld(r), n, h, w, cin, cout, 1,
                        _p(stat_part), stat_part.numel() * 4 if rows > 0 else 0, cap, st)
            fused_stats = (stat_part, rows) if rows > 0 else None
        elif lp.fwd == "mfma":
            self._timed("conv3x3_fwd", 2.0 * 9 * n * h * w * cin * cout, L.unet_conv3x3_fwd_mfma,
                        _p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout, 1, st)
        else:
            rows = L.unet_conv3x3_fwd_direct_stats_rows(n, h, w, cin, cout) if lp.fwd_stats else 0
            if rows > 0:
                stat_part = part(rows)
                self._timed("first_layer_fwd", self._nb(x, r), L.unet_conv3x3_fwd_direct_stats,
                            _p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), int(r16), n, h, w, cin, cout, 1, _p(stat_part), stat_part.numel() * 4, st)
                fused_stats = (stat_part, rows)
            else:
                assert not r16
                L.unet_conv3x3_fwd_direct(_p(x), _ld(x), _p(w_), _p(b_), _p(r), _ld(r), n, h, w, cin, cout, 1, st)
        assert fused_stats is not None or not r16, name            # a bf16-stored r has its sums taken before the rounding
        P = r.shape[0] * r.shape[1] * r.shape[2]
        s = self.stat[name]
        sc_out, sh_out = stat_out if stat_out is not None else (s[2], s[3])
        gm, bt = self.p[name + "/gamma"], self.p[name + "/beta"]
        mm, mv = self.moving[name + "/moving_mean"], self.moving[name + "/moving_var"]
        merged = training and fused_stats is not None and self.opt.merge_bn_finalize and y_out is not None and cout % 64 == 0
        if merged:
            # finalize + apply (+ pool) in ONE launch: the apply grid's first workgroups finalize, all wait on the layer's counter word
            cnt = self._fin_counters()
            i = self._fin_index[name]
            self._fin_target[i] = (self._fin_target[i] + cout) & 0xFFFFFFFF
            self._timed("bn_apply", self._nb(r, y_out, *(pool or ())), L.unet_bn_finalize_apply_any,
                        _p(fused_stats[0]), fused_stats[1], _p(gm), _p(bt), BN_EPS, BN_MOMENTUM, BN_MOVING_VAR_UNBIASED, _p(mm), _p(mv), _p(s[0]), _p(s[1]),
                        ctypes.c_void_p(cnt.data_ptr() + 4 * i), self._fin_target[i],
                        _p(r), _ld(r), int(r.dtype == torch.bfloat16), _p(sc_out), _p(sh_out), _p(y_out), _ld(y_out), int(y_out.dtype == torch.bfloat16),
                        _p(pool[0]) if pool is not None else None, cout, _p(pool[1]) if pool is not None else None,
                        r.shape[0], r.shape[1], r.shape[2], cout, st)
            self._eval_coefs.clear()
            self.saved[name] = (x, r)
            self.view[name] = in_view
            self.coef[name] = (sc_out, sh_out)
            return y_out
        if training and fused_stats is not None:
            L.unet_bn_train_finalize_partials(_p(fused_stats[0]), fused_stats[1], P, cout, _p(gm), _p(bt), BN_EPS, BN_MOMENTUM,
                                              BN_MOVING_VAR_UNBIASED, _p(mm), _p(mv), _p(s[0]), _p(s[1]), _p(sc_out), _p(sh_out), st)
        elif training:
            nb = L.unet_bn_workspace(P, cout)
            ws = self._workspace(nb)
            self._timed("bn_stats", self._nb(r), L.unet_bn_train_stats, _p(r), _ld(r), P, cout, _p(gm), _p(bt), BN_EPS, BN_MOMENTUM,
                        BN_MOVING_VAR_UNBIASED, _p(mm), _p(mv), _p(s[0]), _p(s[1]), _p(sc_out), _p(sh_out), _p(ws), nb, st)
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
            self._timed("bn_apply", self._nb(r, y_out, *(pool or ())), L.unet_bn_apply_any,
                        _p(r), _ld(r), int(r.dtype == torch.bfloat16), _p(sc_out), _p(sh_out), _p(y_out), _ld(y_out), int(y_out.dtype == torch.bfloat16),
                        _p(pool[0]) if pool is not None else None, cout, _p(pool[1]) if pool is not None else None,
                        r.shape[0], r.shape[1], r.shape[2], cout, st)
        elif pool is not None:         # (pooled, idx): BN apply and the level's max pool in one pass
            self._timed("bn_apply", self._nb(r, y_out, *pool), L.unet_bn_apply_maxpool,
                        _p(r), _ld(r), _p(sc_out), _p(sh_out), _p(y_out), _ld(y_out), _p(pool[0]), cout, _p(pool[1]),
                        r.shape[0], r.shape[1], r.shape[2], cout, st)
        else:
            self._timed("bn_apply", self._nb(r, y_out), L.unet_bn_apply, _p(r), _ld(r), _p(sc_out), _p(sh_out), _p(y_out), _ld(y_out), P, cout, st)
        return y_out

    def _dropout(self, t, key, masks, backward=False):
        n, h, w, c = t.shape
        m = None
        if masks is not None:
            m = masks[key]
        seed = (self.dropout_seed * 1000003 + self.iterations * 2 + (0 if key == "drop_4" else 1)) & 0xFFFFFFFF
        self._timed("dropout", 2 * self._nb(t), self.L.unet_dropout,
                    _p(t), _ld(t), _p(t), _ld(t), n * h * w, c, _p(m), seed, DROPOUT_RATE, int(t.dtype == torch.bfloat16), self._stream())

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
        pl = self.pl = self.plan(n, h, w, training, want_grad)
        if training:
            self._eval_folded.clear()           # the moving statistics move: no eval-mode fold survives a training forward
        B = BASE
        f = self._block_fwd
        tdt = lambda d: torch.bfloat16 if d == BF16 else torch.float32
        cur, cur_view = x0, None
        self.idx = {}
        self.cat = {}
        self.catstat = {}

        def ybuf(name, shape):
            """buffer for the layer's materialised BatchNorm output, in the precision the plan stores it"""
            d = pl.layer[name].y
            return self._buf(("y16_" if d == BF16 else "y_") + name, shape, tdt(d))

        def pair(a_name, b_name, xin, xin_view, hh, ww, ch, b_out, **kw):
            """the two convs of a block: a -> b.  a's BatchNorm output feeds only b, so with BatchNorm-apply on load it is never
            materialised: b reads a's conv output through (scale, shift)."""
            if pl.layer[a_name].defer_y:
                ra = f(a_name, xin, None, training, in_view=xin_view)
                sa = self.stat[a_name]
                return f(b_name, ra, b_out, training, in_view=(sa[2], sa[3]), **kw)
            ya = f(a_name, xin, ybuf(a_name, (n, hh, ww, ch)), training, in_view=xin_view)
            return f(b_name, ya, b_out, training, **kw)

        for lvl, ch in ((1, B), (2, 2 * B), (3, 4 * B), (4, 8 * B)):
            hh, ww = cur.shape[1], cur.shape[2]
            # the concat buffer [skip, upsampled] and the pooled tensor feed 3x3 layers only (dec_Na / the next level's first conv)
            c16 = pl.cat[lvl] == BF16
            cat = self._buf(("cat16_%d" if c16 else "cat_%d") % lvl, (n, hh, ww, 2 * ch), tdt(pl.cat[lvl]))
            self.cat[lvl] = cat
            pooled = self._buf(("pool16_%d" if c16 else "pool_%d") % lvl, (n, hh // 2, ww // 2, ch), tdt(pl.cat[lvl]))
            idx = self._buf("idx_%d" % lvl, (n, hh // 2, ww // 2, ch), torch.uint8)
            fuse_pool = pl.fuse_pool[lvl]
            skip = pair("conv_%da" % lvl, "conv_%db" % lvl, cur, cur_view, hh, ww, ch, cat[..., :ch], pool=(pooled, idx) if fuse_pool else None)
            if not fuse_pool:
                if lvl == 4 and training:
                    self._dropout(skip, "drop_4", self.masks)
                assert skip.dtype == pooled.dtype
                self._timed("pool", self._nb(skip, pooled, idx), L.unet_maxpool2x2_fwd,
                            _p(skip), _ld(skip), _p(pooled), ch, _p(idx), n, hh, ww, ch, int(skip.dtype == torch.bfloat16), st)
            self.idx[lvl] = idx
            cur, cur_view = pooled, None
        hh, ww = cur.shape[1], cur.shape[2]
        cur = pair("bott_a", "bott_b", cur, None, hh, ww, 16 * B, ybuf("bott_b", (n, hh, ww, 16 * B)))
        if training:
            self._dropout(cur, "drop_b", self.masks)
        for lvl, ch in ((4, 8 * B), (3, 4 * B), (2, 2 * B), (1, B)):
            cat = self.cat[lvl]
            hh, ww = cat.shape[1], cat.shape[2]
            cat_view = None
            if pl.layer["up_%d" % lvl].defer_y:
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
            # (dec_1b: the class-map conv computes in fp32, so storing its input as bf16 changes the result -- only the training step of
            # the mixed-precision mode does it (Keras mixed_bfloat16 semantics); inference keeps the fp32 tensor: plan.py)
            cur = pair("dec_%da" % lvl, "dec_%db" % lvl, cat, cat_view, hh, ww, ch, ybuf("dec_%db" % lvl, (n, hh, ww, ch)))
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
            self._timed("softmax_ce", self._nb(yl, lab, prob, dl), L.unet_softmax_ce,
                        _p(yl), self.K, _p(lab), _p(prob), _p(dl), self.K, P, self.K, float(label_smoothing), scale,
                        scale, float(self.ce_clip_eps), _p(self.loss_buf[0:1]), _p(self.loss_buf[1:2]), _p(ws), nb, st)
            self._labels_keepalive = lab
        return prob

    # ------------------------------------------------------------------------------------------------ backward
    def _block_bwd(self, name, dy, eval_mode=False, pool_grad=None):
        """dy: NHWC view = gradient w.r.t. the layer's BN output.  Returns gradient w.r.t. the layer input (or None).
        eval_mode: BN used its moving statistics (an affine map) and no parameter gradients are wanted.  Kernel families and the
        storage of dz / dx: self.pl.layer[name] (the plan of the forward pass this backward belongs to)."""
        L, st = self.L, self._stream()
        lp = self.pl.layer[name]
        kind, cin, cout = lp.kind, lp.cin, lp.cout
        x, r = self.saved[name]
        n, ho, wo, _ = r.shape
        P = n * ho * wo
        s = self.stat[name]
        dz16 = lp.dz == BF16
        need_dx = lp.dx is not None
        assert not (eval_mode and (dz16 or self.pl.training))
        dz = self._buf("dz16_" + name, tuple(r.shape), torch.bfloat16) if dz16 else self._buf("dz_" + name, tuple(r.shape))
        pre = self.bnbwd_part.pop(name, None) if not eval_mode else None
        assert (pre is not None) == (lp.sums_from_dgrad and not eval_mode), name
        bias_rows = None
        if eval_mode:
            L.unet_bn_eval_bwd(_p(dy), _ld(dy), _p(r), _ld(r), _p(self.coef[name][0]), _p(dz), cout, P, cout, 0 if kind == "deconv" else 1, st)
        else:
            # one entry point for the three forms (plain / pooled / sums from the consumer's data gradient: no reduction pass), any of dy /
            # r / dz stored as bf16.  The bias gradient sum(dz) is left as per-block partials in the layer's own workspace and finished beside
            # the weight gradient (unet_bn_bwd_bias below): nothing on the chain BatchNorm backward -> data gradient needs it.
            part_ptr, rows = None, 0
            if pre is not None:
                part, rows, c0 = pre
                part_ptr = ctypes.c_void_p(part.data_ptr() + (c0 // 64) * rows * 128 * 4)
            pdy, pidx = pool_grad if pool_grad is not None else (None, None)
            assert not (pool_grad is not None and pre is not None)
            nb = L.unet_bn_workspace(P, cout)
            bnws = self._buf("bnws_" + name, (int(nb) + 256,), torch.uint8)
            bias_rows = ctypes.c_int(0)
            self._timed("bn_bwd", (1 if pre is not None else 2) * self._nb(dy, r, pdy, pidx if pdy is not None else None) + self._nb(dz), L.unet_bn_bwd_any,
                        _p(dy), _ld(dy), _p(pdy), _ld(pdy) if pdy is not None else 0, _p(pidx), n, ho, wo, _p(r), _ld(r),
                        _p(self.p[name + "/gamma"]), _p(s[0]), _p(s[1]), cout, 0 if kind == "deconv" else 1, _p(dz), cout, int(dz.dtype == torch.bfloat16),
                        _p(self.g[name + "/gamma"]), _p(self.g[name + "/beta"]), None, part_ptr, rows,
                        _p(bnws), nb, st, int(r.dtype == torch.bfloat16), int(dy.dtype == torch.bfloat16),
                        int(pdy is not None and pdy.dtype == torch.bfloat16), ctypes.byref(bias_rows))
        w_, dw = self.p[name + "/kernel"], self.g[name + "/kernel"]
        hi, wi = x.shape[1], x.shape[2]
        dx = None
        overlap = self.opt.overlap_wgrad
        cap = self.opt.cap
        z16 = int(dz.dtype == torch.bfloat16)

        def wgrad():
            sd = overlap
            st2 = self._stream()
            L.unet_bn_bwd_bias(_p(bnws), bias_rows.value, cout, _p(self.g[name + "/bias"]), st2)
            if lp.wgrad == "convt_bf16":
                nb2 = L.unet_convT2x2_wgrad_bf16_workspace_wg(n, hi, wi, cin, cout, cap)
                self._timed("convt_wgrad_bf16", 8.0 * n * hi * wi * cin * cout, L.unet_convT2x2_wgrad_bf16_wg,
                            _p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(dz), cout, z16, _p(dw),
                            n, hi, wi, cin, cout, cap, _p(self._workspace(nb2, sd)), nb2, st2)
            elif lp.wgrad == "convt_x6":
                nb2 = L.unet_convT2x2_wgrad_x6_workspace_wg(n, hi, wi, cin, cout, cap)
                self._timed("convt_wgrad_x6", 8.0 * n * hi * wi * cin * cout, L.unet_convT2x2_wgrad_x6_wg,
                            _p(x), _ld(x), _p(dz), cout, _p(dw), n, hi, wi, cin, cout, cap, _p(self._workspace(nb2, sd)), nb2, st2)
            elif kind == "deconv":
                nb2 = L.unet_convT2x2_wgrad_workspace_wg(n, hi, wi, cin, cout, cap)
                self._timed("convt_wgrad", 8.0 * n * hi * wi * cin * cout, L.unet_convT2x2_wgrad_wg,
                            _p(x), _ld(x), _p(dz), cout, _p(dw), n, hi, wi, cin, cout, cap, _p(self._workspace(nb2, sd)), nb2, st2)
            elif kind == "conv1":
                nb2 = L.unet_conv1x1_wgrad_workspace(P, cin, cout)
                self._timed("classmap_wgrad", self._nb(x, dz), L.unet_conv1x1_wgrad,
                            _p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(dz), cout, _p(dw), P, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
            elif lp.wgrad == "bf16":
                nb2 = L.unet_conv3x3_wgrad_bf16_workspace_wg(n, ho, wo, cin, cout, cap)
                self._timed("conv3x3_wgrad_bf16", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_wgrad_bf16_wg,
                            _p(x), _ld(x), int(x.dtype == torch.bfloat16), _p(dz), cout, z16, _p(dw),
                            n, ho, wo, cin, cout, cap, _p(self._workspace(nb2, sd)), nb2, st2)
            elif lp.wgrad == "winograd":
                nb2 = L.unet_conv3x3_wgrad_winograd_fused_workspace(n, ho, wo, cin, cout, cap)
                self._timed("conv3x3_wgrad_winograd_fused", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_wgrad_winograd_fused,
                            _p(x), _ld(x), _p(dz), cout, _p(dw), n, ho, wo, cin, cout, cap, _p(self._workspace(nb2, sd)), nb2, st2)
                vw = self.view.get(name)
                if vw is not None:
                    # x was read through BatchNorm-apply on load: dw (computed on the producer's raw conv output) -> scale . dw + shift (x) S
                    nb3 = L.unet_conv3x3_wgrad_fold_fix_workspace(cout)
                    L.unet_conv3x3_wgrad_fold_fix(_p(dw), _p(vw[0]), _p(vw[1]), _p(dz), cout, _p(self.g[name + "/bias"]), n, ho, wo, cin, cout,
                                                  _p(self._workspace(nb3, sd)), nb3, st2)
            elif lp.wgrad == "mfma":
                nb2 = L.unet_conv3x3_wgrad_mfma_workspace_wg(n, ho, wo, cin, cout, cap)
                self._timed("conv3x3_wgrad", 2.0 * 9 * n * ho * wo * cin * cout, L.unet_conv3x3_wgrad_mfma_wg,
                            _p(x), _ld(x), _p(dz), cout, _p(dw), n, ho, wo, cin, cout, cap, _p(self._workspace(nb2, sd)), nb2, st2)
            else:
                nb2 = L.unet_conv3x3_wgrad_direct_workspace(n, ho, wo, cin, cout)
                self._timed("first_layer_wgrad", self._nb(x, dz), L.unet_conv3x3_wgrad_direct,
                            _p(x), _ld(x), _p(dz), cout, z16, _p(dw), n, ho, wo, cin, cout, _p(self._workspace(nb2, sd)), nb2, st2)
            if self.on_layer_grads_ready is not None:
                self.on_layer_grads_ready(name)          # under the stream the gradients were produced on

        if overlap and not eval_mode:
            self.side.wait_stream(torch.cuda.current_stream())       # dz (and this layer's bias/gamma/beta grads) ready
            with torch.cuda.stream(self.side):
                wgrad()
        if need_dx:
            dx16 = lp.dx == BF16
            dx = self._buf(("dy16_in_" if dx16 else "dy_in_") + name, (n, hi, wi, cin), torch.bfloat16 if dx16 else torch.float32)
            # a data gradient whose output (or a channel range of it) is exactly the dy of the producer layer's BatchNorm also leaves
            # that layer's backward sums (sum dy, sum dy * r): no reduction pass there
            prod = lp.leaves_sums_for if not eval_mode else None
            r_prev = part = None
            rows, c0, c1 = 0, 0, 0
            if prod is not None:
                pname, c0, c1 = prod
                r_prev = self.saved[pname][1]
                rows = (L.unet_convT2x2_bf16_stats_rows(n, hi, wi, cin, cout, 1) if lp.dgrad == "convt_bf16"
                        else L.unet_convT2x2_x6_bnbwd_rows(n, hi, wi, cin, cout) if lp.dgrad == "convt_x6"
                        else L.unet_conv3x3_bf16_stats_rows(n, ho, wo, cout, cin) if lp.dgrad == "bf16"
                        else L.unet_conv3x3_fwd_winograd_fused_stats_rows_wg(n, ho, wo, cout, cin, cap))
                assert rows > 0
                part = self._buf("bnbwd_" + name, ((cin // 64) * rows * 128,))
                self.bnbwd_part[pname] = (part, rows, c0)
            nbp = part.numel() * 4 if part is not None else 0
            ldr_prev = _ld(r_prev) if r_prev is not None else 0
            r16_prev = int(r_prev is not None and r_prev.dtype == torch.bfloat16)
            fl = 2.0 * 9 * n * ho * wo * cin * cout
            if lp.dgrad == "convt_bf16":
                self._timed("convt_dgrad_bf16", 8.0 * n * hi * wi * cin * cout, L.unet_convT2x2_dgrad_bf16,
                            _p(dz), cout, z16, _p(self._bf16_kernels(name)[1]), _p(dx), cin, int(dx16), n, hi, wi, cin, cout,
                            _p(r_prev), ldr_prev, r16_prev, _p(part), nbp, st)
            elif lp.dgrad == "convt_x6":
                self._timed("convt_dgrad_x6", 8.0 * n * hi * wi * cin * cout, L.unet_convT2x2_dgrad_x6_sums,
                            _p(dz), cout, _p(self._convt_x6_kernels(name)[1]), _p(dx), cin, n, hi, wi, cin, cout,
                            _p(r_prev), ldr_prev, _p(part), nbp, st)
            elif kind == "deconv":
                self._timed("convt_dgrad", 8.0 * n * hi * wi * cin * cout, L.unet_convT2x2_dgrad, _p(dz), cout, _p(w_), _p(dx), cin, n, hi, wi, cin, cout, st)
            elif kind == "conv1":
                self._timed("classmap_dgrad", self._nb(dz, dx), L.unet_conv1x1_dgrad, _p(dz), cout, _p(w_), _p(dx), cin, int(dx16), P, cin, cout, st)
            elif lp.dgrad == "bf16":
                self._timed("conv3x3_dgrad_bf16", fl, L.unet_conv3x3_dgrad_bf16_wg,
                            _p(dz), cout, z16, _p(self._bf16_kernels(name)[1]), _p(dx), cin, int(dx16), n, ho, wo, cin, cout,
                            _p(r_prev), ldr_prev, r16_prev, c0, c1, _p(part), nbp, cap, st)
            elif lp.dgrad == "winograd" and lp.dgrad_x6:
                self._timed("conv3x3_dgrad_winograd_x6", fl, L.unet_conv3x3_dgrad_winograd_x6_wg,
                            _p(dz), cout, _p(self._x6_kernels(name)[1]), _p(dx), cin, n, ho, wo, cin, cout,
                            _p(r_prev), ldr_prev, c0, c1, _p(part), nbp, cap, st)
            elif lp.dgrad == "winograd":
                self._timed("conv3x3_dgrad_winograd_fused", fl, L.unet_conv3x3_dgrad_winograd_fused_wg,
                            _p(dz), cout, _p(self._fused_kernels(name)[1]), _p(dx), cin, n, ho, wo, cin, cout,
                            _p(r_prev), ldr_prev, c0, c1, _p(part), nbp, cap, st)
            elif lp.dgrad == "mfma":
                self._timed("conv3x3_dgrad", fl, L.unet_conv3x3_dgrad_mfma, _p(dz), cout, _p(w_), _p(dx), cin, n, ho, wo, cin, cout, st)
            else:                                   # first layer (Cin = number_channels): only the ERF probe needs it
                L.unet_conv3x3_dgrad_direct(_p(dz), cout, _p(w_), _p(dx), cin, n, ho, wo, cin, cout, st)
        if not overlap and not eval_mode:
            wgrad()
        return dx

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
        b = lambda name, dy: self._block_bwd(name, dy, eval_mode=eval_mode)
        assert self.pl is not None and self.pl.training == (not eval_mode), "backward() follows the forward pass of the same mode"
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
            if self.opt.fuse_pool and not eval_mode and lvl != 4 and ch % 4 == 0:
                d = self._block_bwd("conv_%db" % lvl, ds, pool_grad=(d, self.idx[lvl]))       # no separate pool-backward pass
            else:
                assert d.dtype == ds.dtype, (lvl, d.dtype, ds.dtype)
                self._timed("pool", self._nb(d, self.idx[lvl]) + 2 * self._nb(ds), L.unet_maxpool2x2_bwd,
                            _p(d), _ld(d), _p(self.idx[lvl]), _p(ds), _ld(ds), n, hh, ww, ch, 1, int(ds.dtype == torch.bfloat16), st)
                if lvl == 4 and not eval_mode:
                    self._dropout(ds, "drop_4", self.masks)
                d = b("conv_%db" % lvl, ds)
            d = b("conv_%da" % lvl, d)          # (training: the first layer's data gradient is not computed -- plan.py)
        if self.opt.overlap_wgrad and not eval_mode:
            torch.cuda.current_stream().wait_stream(self.side)        # every weight gradient done before Adam
        return d

    def adam_step(self, learning_rate):
        """Keras Adam on the flat buffers (reference UNet/model.py:79,223)."""
        self.iterations += 1
        self._fused_dirty = True
        self._x6_dirty = True
        self._ctx6_dirty = True
        self._bf16_dirty = True
        self._eval_folded.clear(); self._eval_coefs.clear()
        t = self.iterations
        alpha = learning_rate * math.sqrt(1.0 - ADAM_BETA2 ** t) / (1.0 - ADAM_BETA1 ** t)
        self._timed("adam", 7.0 * 4 * self.n_flat, self.L.unet_adam_keras,          # reads theta, g, m, v; writes theta, m, v
                    _p(self.theta), _p(self.grad), _p(self.adam_m), _p(self.adam_v), self.n_flat, alpha,
                    ADAM_BETA1, ADAM_BETA2, ADAM_EPS, self._stream())

    def argmax(self, prob):
        n, h, w, k = prob.shape
        out = self._buf("argmax", (n, h, w), torch.int32)
        self.L.unet_argmax(_p(prob), k, _p(out), n * h * w, k, self._stream())
        return out
