"""Overlap-readiness of the data-parallel backward on ONE GPU (reference UNet/train.py:57-61, UNet/model.py:223: the gradient all-reduce
inside apply_gradients).  No N > 1 hardware is available to run RCCL itself, so a STAND-IN collective plays its part: for
every gradient bucket (parallel.make_buckets, 25 MB) one launch of unet_standin_collective on a third stream, released -- like an async
all-reduce on RCCL's stream -- by an event behind the bucket's last weight gradient.  The stand-in is what a collective's kernel is to the
rest of the chip: 16-32 workgroups of 512 threads that hold their CU slots and some LDS for as long as a 25 MB bucket takes over xGMI
(0.3 / 0.8 / 1.4 ms), moving no data.

Two things are read off a run (scripts/overlap_probe.py prints the whole table -> profiles/rNN_overlap_standin.txt):
  * wait = completion - max(release, completion of the previous stand-in) - hold: how long the stand-in's workgroups waited for CU slots.  Every persistent kernel of the step
    (`_wg` entry points of include/unet_hip.h) sizes its grid by EngineOptions.max_workgroups; uncapped, such a grid owns every CU until
    its first workgroup has drained its tiles;
  * the step time with the stand-ins resident against the same step without them: what the resident workgroups cost the backward pass
    (a persistent grid launched while 32 CUs are held runs its last workgroups as a second wave unless it is capped).
Measured (MI355X, profiles/r04_overlap_standin.txt): capped at 224, every stand-in gets its CUs within 0.01-0.02 ms; uncapped, it waits
up to one or two persistent kernels (0.1-1.0 ms in the fp32 step).  Which STEP is faster is not stable: uncapped in 11 of the table's 12
cells (by 0.3-1.2 ms), capped by 1.1-1.5 ms in this test's cell on another box the same day.  The cap costs every CU-bound kernel 1/8 of the
chip all the time; the resident stand-in costs an uncapped grid a second wave some of the time.  So `DataParallel` does NOT cap by default
(EngineOptions.max_workgroups stays an explicit option for an N > 1 RCCL measurement to decide); the test pins what is stable."""
import ctypes
import dataclasses

import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def run_step_with_standin(cap, hold_us=800, wgs=32, lds=32768, steps=4, n=8, hw=512, dtype="fp32", standin=True):
    """One training step (BASELINE config 2 by default) with a stand-in collective per gradient bucket -> dict of times [ms]"""
    model, par = pkg("model"), pkg("parallel")
    net = model.UNet(2, n, 1, seed=1, compute_dtype=dtype)
    e = net.engine
    e.opt = dataclasses.replace(e.opt, max_workgroups=cap)
    L = e.L
    buckets = par.make_buckets(e.layer_range, 25 * 1024 * 1024)
    trigger = {last: i for i, (_, _, last) in enumerate(buckets)}
    third = torch.cuda.Stream()
    rec = []

    def hook(name):
        i = trigger.get(name)
        if i is None or not standin:
            return
        ready = torch.cuda.Event(enable_timing=True); ready.record()           # behind the bucket's last weight gradient, on its stream
        third.wait_event(ready)
        with torch.cuda.stream(third):
            L.unet_standin_collective(wgs, lds, hold_us, ctypes.c_void_p(third.cuda_stream))
            done = torch.cuda.Event(enable_timing=True); done.record()
        rec.append((i, ready, done))

    e.on_layer_grads_ready = hook
    g = torch.Generator().manual_seed(0)
    img = torch.randn(n, 1, hw, hw, generator=g).cuda()
    lab = torch.nn.functional.one_hot(torch.randint(0, 2, (n, hw, hw), generator=g), 2).to(torch.int32).cuda()
    outs = []
    for s in range(steps):
        rec.clear()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t0.record()
        e.forward(img, training=True, labels=lab, global_batch_size=n, want_grad=True)
        e.backward()                                                            # ends with the main stream waiting for the side stream
        last_wgrad = torch.cuda.Event(enable_timing=True); last_wgrad.record()
        torch.cuda.current_stream().wait_stream(third)
        e.adam_step(1e-4)
        t1 = torch.cuda.Event(enable_timing=True); t1.record()
        torch.cuda.synchronize()
        ready = [t0.elapsed_time(r) for _, r, _ in rec]; done = [t0.elapsed_time(d) for _, _, d in rec]
        # a stand-in can start once it is released AND the previous one (same stream, like consecutive all-reduces) has finished
        start = [max(r, done[i - 1]) if i else r for i, r in enumerate(ready)]
        outs.append(dict(buckets=len(buckets), last_wgrad_ms=t0.elapsed_time(last_wgrad), step_ms=t0.elapsed_time(t1), ready_ms=ready, done_ms=done,
                   wait_ms=[d - s0 - hold_us / 1000.0 for s0, d in zip(start, done)]))
    out = sorted(outs[1:], key=lambda o: o["step_ms"])[(len(outs) - 1) // 2]        # the median step after the first (which builds the plan)
    del net, e
    torch.cuda.empty_cache()
    return out


def test_standin_kernel_holds_its_workgroups_for_the_requested_time():
    L = pkg("_lib").lib()
    st = torch.cuda.current_stream()
    for us in (300, 1400):
        L.unet_standin_collective(32, 32768, us, ctypes.c_void_p(st.cuda_stream)); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); L.unet_standin_collective(32, 32768, us, ctypes.c_void_p(st.cuda_stream)); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        assert us / 1000.0 <= ms < us / 1000.0 + 0.5, (us, ms)
    for bad in ((0, 0, 10), (32, 1 << 20, 10), (32, 0, -1)):
        with pytest.raises(pkg("_lib").UnetHipError, match="bad argument"):
            L.unet_standin_collective(*bad, None)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_capped_grids_leave_cus_to_a_resident_collective(dtype):
    par = pkg("parallel")
    cap = par.DataParallel.OVERLAP_WORKGROUPS
    alone = run_step_with_standin(0, dtype=dtype, standin=False)
    capped_alone = run_step_with_standin(cap, dtype=dtype, standin=False)
    capped = run_step_with_standin(cap, dtype=dtype)
    free = run_step_with_standin(0, dtype=dtype)
    fmt = lambda r: " ".join("%.2f" % w for w in r["wait_ms"])
    print("\n%s step alone %.2f ms, capped alone %.2f ms" % (dtype, alone["step_ms"], capped_alone["step_ms"]))
    print("cap %3d + stand-ins (32 wg x 0.8 ms): step %.2f ms, wait for CUs per bucket [ms] %s" % (cap, capped["step_ms"], fmt(capped)))
    print("no cap  + stand-ins                 : step %.2f ms, wait for CUs per bucket [ms] %s" % (free["step_ms"], fmt(free)))
    for run in (capped, free):
        assert run["buckets"] >= 4 and len(run["done_ms"]) == run["buckets"]
        # every stand-in but the last bucket's (which closes behind the LAST weight gradient by construction) completes under the backward pass
        for i in range(run["buckets"] - 1):
            assert run["done_ms"][i] < run["last_wgrad_ms"] + 0.9, (i, run)
    # the buckets close spread over the last part of the step (the large weights sit deep in the network), not bunched at its end
    assert capped["ready_ms"][0] < 0.85 * capped["last_wgrad_ms"]
    # with the cap a collective's workgroups find their CUs at once; uncapped they wait for at most a persistent kernel or two to drain
    assert max(capped["wait_ms"]) < 0.15, capped
    assert max(free["wait_ms"]) < 1.5, free
    # Step times are printed, not compared: which of the two is faster changes from box to box and with what ran before (seen in this
    # suite: capped 4 ms slower; alone: capped 1.3 ms faster; the committed table: uncapped faster in 11 of 12 cells) -- the reason the
    # cap is an option and not the data-parallel default.  Only gross breakage is caught here.
    for run in (capped_alone, capped, free):
        assert 0.95 * alone["step_ms"] < run["step_ms"] < 1.6 * alone["step_ms"] + 1.0, (run["step_ms"], alone["step_ms"])
