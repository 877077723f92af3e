"""Data-parallel path on CPU: world_size-2 gloo processes drive parallel.DataParallel with a stub engine (the class only
touches grad/theta/adam/moving/layer_range/on_layer_grads_ready), checking bucket construction, that every bucket is
all-reduced with SUM exactly once per step in gradient-readiness order, the rank-0 broadcast, the loss reduce and the
moving-stat average; and an oracle-backed engine (fp64 train step behind the same flat layout and hook
protocol) showing that an R = 2 run through DataParallel equals the reference's global-batch semantics exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class StubEngine:
    """Same flat-buffer layout rules as engine.Engine, without touching the GPU library."""

    def __init__(self, rank):
        eng = pkg("engine")
        self.layer_range, off = {}, 0
        for name in eng.BACKWARD_ORDER:
            n = {"logits": 136, "bott_b": 9000, "dec_4a": 5000}.get(name, 1000)
            self.layer_range[name] = (off, off + n); off += n
        self.n_flat = off
        g = torch.Generator().manual_seed(100 + rank)
        self.theta = torch.randn(off, generator=g)
        self.adam_m = torch.randn(off, generator=g)
        self.adam_v = torch.rand(off, generator=g)
        self.grad = torch.zeros(off)
        self.moving = {"x/moving_mean": torch.full((4,), float(rank)), "x/moving_var": torch.full((4,), 1.0 + rank)}
        self.on_layer_grads_ready = None


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng_mod, par = pkg("engine"), pkg("parallel")
        e = StubEngine(rank)
        theta0 = e.theta.clone()
        dp = par.DataParallel(e, bucket_bytes=4 * 6000)
        # X5: state mirrored from rank 0
        ref = [torch.zeros_like(e.theta) for _ in range(world)]
        dist.all_gather(ref, e.theta)
        assert torch.equal(ref[0], ref[1])
        if rank == 0:
            assert torch.equal(e.theta, theta0)
        # buckets: contiguous, ordered, cover the whole flat buffer, closed on layer boundaries
        assert dp.buckets[0][0] == 0 and dp.buckets[-1][1] == e.n_flat
        for (a0, b0, _), (a1, b1, _) in zip(dp.buckets, dp.buckets[1:]):
            assert b0 == a1 and b0 > a0
        assert len(dp.buckets) > 2
        # one step: each rank's gradient is a different constant; after finish_step every element holds the SUM
        launched = []
        orig = dp._on_layer
        def spy(name):
            before = len(dp._pending); orig(name); launched.extend([name] * (len(dp._pending) - before))
        e.on_layer_grads_ready = spy
        dp.begin_step()
        for name in eng_mod.BACKWARD_ORDER:            # the backward schedule calls the hook in this order
            a, b = e.layer_range[name]
            e.grad[a:b] = float(rank + 1)
            e.on_layer_grads_ready(name)
        dp.finish_step()
        assert torch.all(e.grad == float(sum(r + 1 for r in range(world))))
        assert launched == [last for _, _, last in dp.buckets]          # every bucket once, in readiness order
        # X2: loss reduce; X4: moving-stat mean
        assert float(dp.reduce_sum(torch.tensor([0.25 * (rank + 1)]))) == pytest.approx(0.75)
        e.moving["x/moving_mean"].fill_(float(rank)); e.moving["x/moving_var"].fill_(1.0 + rank)   # replicas diverge
        dp.average_moving_stats()
        assert torch.allclose(e.moving["x/moving_mean"], torch.full((4,), 0.5))
        assert torch.allclose(e.moving["x/moving_var"], torch.full((4,), 1.5))
        out.put((rank, "ok"))
    except Exception as ex:            # surface the failure in the parent
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_data_parallel_buckets_gloo_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(out.get(timeout=120) for _ in procs)
    [p.join(30) for p in procs]
    assert res == {0: "ok", 1: "ok"}, res


class OracleEngine:
    """CPU stand-in for engine.Engine behind parallel.DataParallel: the SAME flat layout (engine.flat_layout) and hook
    protocol, with the oracle's fp64 train step as the arithmetic -- per-replica BatchNorm, loss divided by the GLOBAL batch
    (reference UNet/model.py:213), gradients written layer by layer in backward-completion order with
    on_layer_grads_ready(name) after each, then one Keras-Adam step on the flat buffers."""

    def __init__(self, k, G, c, prm, lr):
        from oracle import unet_numpy as on
        eng = pkg("engine")
        self.on, self.eng, self.lr, self.G = on, eng, lr, G
        self.slices, self.layer_range, self.n_flat = eng.flat_layout(c, k)
        self.net = on.OracleUNet(k, G, c, learning_rate=lr, params=prm, dtype=np.float64)
        self.theta = torch.zeros(self.n_flat, dtype=torch.float64)
        self.grad = torch.full((self.n_flat,), float("nan"), dtype=torch.float64)
        self.adam_m = torch.zeros(self.n_flat, dtype=torch.float64)
        self.adam_v = torch.zeros(self.n_flat, dtype=torch.float64)
        for key, (o, n, shape) in self.slices.items():
            self.theta[o:o + n] = torch.as_tensor(np.asarray(prm[key], np.float64).reshape(-1))
        self.moving = {key: torch.as_tensor(np.asarray(v, np.float64)) for key, v in prm.items() if "moving" in key}
        self.on_layer_grads_ready = None
        self.dropout_seed = 0
        self.iterations = 0

    def parameters_changed(self):
        pass

    def backward(self, img, lab, masks):
        for key, (o, n, shape) in self.slices.items():               # theta may have been broadcast: the oracle reads it back
            self.net.params[key] = self.theta[o:o + n].numpy().reshape(shape).copy()
        loss, _, g, cache, _ = self.net.loss_and_grads(img, lab, masks)
        self.grad.zero_()
        for name in self.eng.BACKWARD_ORDER:
            for sfx in ("kernel", "bias", "gamma", "beta"):
                o, n, _ = self.slices[name + "/" + sfx]
                self.grad[o:o + n] = torch.as_tensor(np.ascontiguousarray(g[name + "/" + sfx]).reshape(-1))
            self.on_layer_grads_ready(name)
        return loss

    def adam_step(self):
        self.iterations += 1
        th, m, v = self.on.adam_keras_step(self.theta.numpy(), self.grad.numpy(), self.adam_m.numpy(), self.adam_v.numpy(),
                                           self.iterations, self.lr, self.net.contract)
        self.theta, self.adam_m, self.adam_v = torch.as_tensor(th), torch.as_tensor(m), torch.as_tensor(v)


def _case(n=2, c=1, k=2, hw=16, seed=2):
    from oracle import unet_numpy as on
    img, lab = on.synthetic_batch(n, c, k, hw, hw, seed=seed)
    prm = on.init_params(c, k, seed=seed)
    rng = np.random.default_rng(seed)
    for key in prm:                                                    # non-trivial bias / gamma / beta so their gradients matter
        if key.endswith(("bias", "beta")):
            prm[key] = rng.normal(0, 0.1, prm[key].shape).astype(np.float32)
        if key.endswith("gamma"):
            prm[key] = rng.uniform(0.5, 1.5, prm[key].shape).astype(np.float32)
    masks = {"drop_4": rng.integers(0, 2, (n, 512, hw // 8, hw // 8)), "drop_b": rng.integers(0, 2, (n, 1024, hw // 16, hw // 16))}
    return img, lab, prm, masks


def _dp_worker(rank, world, port, bucket_bytes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = pkg("parallel")
        img, lab, prm, masks = _case()
        if rank != 0:                                    # replicas start from DIFFERENT weights: the rank-0 broadcast must fix that
            prm = {key: v + 1.0 for key, v in prm.items()}
        e = OracleEngine(2, world, 1, prm, 3e-4)
        dp = par.DataParallel(e, bucket_bytes=bucket_bytes)
        assert e.dropout_seed == rank                                  # every replica draws its own dropout stream
        sl = slice(rank, rank + 1)                                     # this replica's image of the global batch of 2
        dp.begin_step()
        loss = e.backward(img[sl], lab[sl], {key: v[sl] for key, v in masks.items()})
        dp.finish_step()
        e.adam_step()
        total = dp.reduce_sum(torch.tensor([loss], dtype=torch.float64))
        out.put((rank, dict(grad=e.grad.numpy().copy(), theta=e.theta.numpy().copy(), loss=float(total), nb=len(dp.buckets))))
    except Exception:
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [25 * 1024 * 1024, 512])
def test_two_rank_step_equals_the_reference_global_batch_semantics(bucket_bytes):
    """R = 2 replicas of one image each through parallel.DataParallel over gloo == what the reference's MirroredStrategy step
    computes for the global batch of 2 (UNet/model.py:204-235): per-replica BatchNorm statistics, per-replica loss = sum of the
    replica's pixel losses / GLOBAL batch, gradients combined with SUM, ONE Keras-Adam step on the summed gradient; the
    returned loss is the SUM of the per-replica losses.  With 512-byte buckets every layer closes its own bucket, so the bucket
    edges sit directly behind each layer's bias / gamma / beta (and the 16-byte padding of the flat layout): every element,
    padding included, must be reduced exactly once.  fp64 and a commutative two-term sum: the comparison is exact."""
    from oracle import unet_numpy as on
    eng = pkg("engine")
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, bucket_bytes, out)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(out.get(timeout=600) for _ in procs)
    [p.join(60) for p in procs]
    assert all(isinstance(v, dict) for v in res.values()), res
    # single-process statement of the same semantics
    img, lab, prm, masks = _case()
    slices, _, n_flat = eng.flat_layout(1, 2)
    gsum, lsum = None, 0.0
    for r in range(2):
        o = on.OracleUNet(2, 2, 1, params=prm, dtype=np.float64)               # global batch 2, this replica holds image r
        loss, _, g, _, _ = o.loss_and_grads(img[r:r + 1], lab[r:r + 1], {key: v[r:r + 1] for key, v in masks.items()})
        lsum += loss
        gsum = g if gsum is None else {key: gsum[key] + g[key] for key in g}
    ref = on.OracleUNet(2, 2, 1, learning_rate=3e-4, params=prm, dtype=np.float64)
    ref.apply_gradients(gsum)
    for r in range(2):
        got = res[r]
        assert got["nb"] == (23 if bucket_bytes == 512 else got["nb"]) and got["nb"] >= 3
        assert got["loss"] == pytest.approx(lsum, rel=1e-14)
        covered = np.zeros(n_flat, bool)
        for key, (o_, n_, shape) in slices.items():
            assert np.array_equal(got["grad"][o_:o_ + n_].reshape(shape), gsum[key]), key
            assert np.allclose(got["theta"][o_:o_ + n_].reshape(shape), ref.params[key], rtol=0, atol=1e-15), key
            covered[o_:o_ + n_] = True
        assert np.all(got["grad"][~covered] == 0.0)                       # padding: reduced (0 + 0), never left as garbage
    assert np.array_equal(res[0]["theta"], res[1]["theta"])                # replicas stay in lock-step
    # and it is NOT the single-replica batch-of-2 step (BatchNorm couples the images there): the distinction is real
    o = on.OracleUNet(2, 2, 1, params=prm, dtype=np.float64)
    _, _, gj, _, _ = o.loss_and_grads(img, lab, masks)
    assert not np.allclose(gj["conv_1a/kernel"], gsum["conv_1a/kernel"], rtol=1e-3)
