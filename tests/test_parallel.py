"""Data-parallel path on CPU: world_size-2 gloo processes drive parallel.DataParallel with a stub engine (the class only
touches grad/theta/adam/moving/layer_range/on_layer_grads_ready), checking bucket construction, that every bucket is
all-reduced with SUM exactly once per step in gradient-readiness order, the rank-0 broadcast, the loss reduce and the
moving-stat average."""
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


def test_split_batch_gradient_equals_sum_of_replica_gradients():
    """Why SUM is the right collective: with the loss divided by the GLOBAL batch (reference UNet/model.py:213) and
    per-replica BatchNorm, the R-replica gradient is the sum of the per-chunk gradients (oracle, fp64)."""
    from oracle import unet_numpy as on
    n, c, k, hw = 2, 1, 2, 16
    img, lab = on.synthetic_batch(n, c, k, hw, hw, seed=2)
    prm = on.init_params(c, k, seed=2)
    rng = np.random.default_rng(2)
    masks = {"drop_4": rng.integers(0, 2, (n, 512, 2, 2)), "drop_b": rng.integers(0, 2, (n, 1024, 1, 1))}
    total, loss_total = None, 0.0
    for r in range(2):
        o = on.OracleUNet(k, n, c, params=prm, dtype=np.float64)            # global batch n, this replica holds 1 image
        loss, _, g, _, _ = o.loss_and_grads(img[r:r + 1], lab[r:r + 1], {kk: v[r:r + 1] for kk, v in masks.items()})
        loss_total += loss
        total = g if total is None else {kk: total[kk] + g[kk] for kk in g}
    # a single replica holding both images but normalising each image separately is the same function:
    o = on.OracleUNet(k, n, c, params=prm, dtype=np.float64)
    la, _, ga, _, _ = o.loss_and_grads(img[0:1], lab[0:1], {kk: v[0:1] for kk, v in masks.items()})
    lb, _, gb, _, _ = o.loss_and_grads(img[1:2], lab[1:2], {kk: v[1:2] for kk, v in masks.items()})
    assert loss_total == pytest.approx(la + lb, rel=1e-12)
    for kk in total:
        assert np.allclose(total[kk], ga[kk] + gb[kk], rtol=1e-12, atol=1e-15)
    assert loss_total > 0
