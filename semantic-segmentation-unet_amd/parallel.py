"""Single-node data parallelism, one process per GPU, replacing tf.distribute.MirroredStrategy (reference
UNet/train.py:57-58; UNet/model.py:230-235,252-256).

Collective call sites of the reference and what replaces them:
  X1  optimizer.apply_gradients all-reduces gradients with SUM (the loss is already divided by the GLOBAL batch,
      UNet/model.py:213)            -> bucketed all_reduce(SUM) over contiguous ranges of the flat gradient buffer,
      launched from the backward schedule as soon as a bucket's last layer has its gradients (engine hook), so RCCL
      runs on its own stream underneath the remaining backward conv kernels; Adam waits for the last bucket;
  X2  strategy.reduce(SUM, per-replica loss)  -> one 4-byte all_reduce (reduce_sum);
  X4  BN moving statistics are per replica; they are averaged only when a checkpoint is written (average_moving_stats);
  X5  variables mirrored from replica 0 at creation -> broadcast from rank 0 (broadcast_state).
BatchNorm batch statistics stay per replica, as in the reference (plain BatchNormalization, UNet/model.py:36,47).

The class touches the engine only through: grad, theta, adam_m, adam_v, moving, layer_range, on_layer_grads_ready --
so the bucket logic is testable on CPU with the gloo backend and a stub engine.
"""
import torch
import torch.distributed as dist

from .engine import BACKWARD_ORDER


def make_buckets(layer_range, bucket_bytes):
    """contiguous buckets of the flat gradient buffer in gradient-readiness order (logits ... conv_1a): [(start, end, last layer)];
    a bucket closes behind the layer that takes it past `bucket_bytes`"""
    buckets, start = [], None
    for name in BACKWARD_ORDER:
        a, b = layer_range[name]
        if start is None:
            start = a
        if (b - start) * 4 >= bucket_bytes or name == BACKWARD_ORDER[-1]:
            buckets.append((start, b, name))
            start = None
    return buckets


class DataParallel:
    OVERLAP_WORKGROUPS = 224

    def __init__(self, engine, bucket_bytes=25 * 1024 * 1024, process_group=None, broadcast=True, force=False):
        assert dist.is_initialized(), "torch.distributed must be initialised (backend nccl == RCCL on ROCm)"
        self.engine = engine
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.buckets = make_buckets(engine.layer_range, bucket_bytes)            # (start, end, last_layer_name)
        self._trigger = {last: i for i, (_, _, last) in enumerate(self.buckets)}
        self._pending = []
        self.force = force            # tests: issue the collectives even with a single rank
        self.trace = None             # tests / profiling: list of (bucket index, event recorded on the issuing stream right before its all-reduce)
        self.done_trace = None        # ... and (bucket index, event on the compute stream right behind the wait for that all-reduce)
        engine.on_layer_grads_ready = self._on_layer
        # every replica draws its OWN dropout masks (MirroredStrategy replicas do); the init seed stays common -- parameters are
        # broadcast from rank 0 anyway
        if hasattr(engine, "dropout_seed"):
            engine.dropout_seed = engine.dropout_seed * self.world_size + self.rank
        # Every persistent kernel of the step takes `EngineOptions.max_workgroups` (the `_wg` entry points of include/unet_hip.h); 224
        # (OVERLAP_WORKGROUPS) leaves ~4 CUs per XCD to a collective's kernels.  It is NOT applied here: on one GPU with a stand-in
        # collective resident (tests/test_gpu_overlap.py, profiles/r04_overlap_standin.txt) the capped step is slower than the uncapped one
        # in 11 of 12 cases of the committed table and faster in the test's case on another box -- the cap costs every CU-bound kernel
        # 1/8 of the chip, the collective costs an uncapped grid a second wave some of the time; no stable sign.  A caller with an
        # N > 1 RCCL measurement passes EngineOptions(max_workgroups=224) (bench.py --max-workgroups 224).
        if broadcast and (self.world_size > 1 or force):
            self.broadcast_state()

    def broadcast_state(self):
        e = self.engine
        for t in [e.theta, e.adam_m, e.adam_v] + list(e.moving.values()):
            dist.broadcast(t, src=0, group=self.group)
        if hasattr(e, "parameters_changed"):
            e.parameters_changed()

    def begin_step(self):
        self._pending = []

    def _on_layer(self, name):
        i = self._trigger.get(name)
        if i is None or (self.world_size == 1 and not self.force):
            return
        a, b, _ = self.buckets[i]
        if self.trace is not None:
            import torch
            ev = torch.cuda.Event(enable_timing=True); ev.record()
            self.trace.append((i, ev))
        self._pending.append((i, dist.all_reduce(self.engine.grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def finish_step(self):
        for i, w in self._pending:
            w.wait()                 # makes the compute stream wait for the collective; no host sync on nccl
            if self.done_trace is not None:
                ev = torch.cuda.Event(enable_timing=True); ev.record()
                self.done_trace.append((i, ev))
        self._pending = []

    def reduce_sum(self, t):
        if self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def average_moving_stats(self):
        """sync-on-read MEAN of BN moving statistics at checkpoint time."""
        if self.world_size > 1:
            for t in self.engine.moving.values():
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                t.div_(self.world_size)
            if hasattr(self.engine, "parameters_changed"):
                self.engine.parameters_changed()         # cached eval-mode folds of the moving statistics are stale
