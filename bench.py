#!/usr/bin/env python3
"""Headline benchmark: U-Net training images/sec on synthetic 512x512x1 tiles, 2 classes, batch 8 per GPU, fp32
(BASELINE.json configs[1]; configs[2] when launched on 8 GPUs).  One "step" = one full optimizer step of the hot path:
forward + per-pixel softmax-CE + backward + (gradient all-reduce) + Keras-Adam, dropout active, inputs resident in HBM.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.

`roofline` is for the dominant kernel = the 3x3-conv kernel family with the most ms per step (forward, data gradient or
weight gradient).  Durations are HIP events recorded on the launch stream, live inside the timed region, on every 8th
timed step; those sampled steps run the backward on ONE stream (the `--no-overlap` schedule) so that every family's
events are exclusive durations -- on the other steps the weight gradients overlap the data-gradient chain on a side
stream and a bracketed launch would include shared time.
  * `achieved`/`frac`: EXECUTED matrix-core FLOP/s and its fraction of the dense MFMA peak for the dtype.  The fp32
    kernels are Winograd F(2x2,3x3): they execute 16 multiplies per 2x2 output tile and channel pair where the direct
    algorithm needs 36, so executed = algorithmic / 2.25; the bf16 kernels are implicit GEMMs (executed = algorithmic);
  * `effective`: the algorithmic (direct-convolution, 2*9*N*H*W*Cin*Cout per launch) rate -- it may exceed the
    peak for a Winograd kernel and is NOT a roofline fraction;
  * `traffic`: HBM bytes per launch from separate rocprofv3 --pmc passes committed under profiles/ (`traffic_source`:
    "offline PMC" -- it is read from that file, not measured by this run), null when no pass exists for the family.
`kernels` lists every instrumented family of the sampled steps: the matrix-core ones as above, and the HBM-bound passes (BatchNorm
apply / backward / statistics, pool, dropout, softmax-CE, Adam, first layer, class map: `"bound": "hbm"`) as ALGORITHMIC bytes (every
tensor the pass reads or writes, once) / exclusive HIP-event time, against the guide's measured copy bandwidth (6.29 TB/s) and the 8 TB/s
spec.  With N > 1, `distributed` carries world size, backend and -- from one traced, untimed step -- when each gradient bucket's
all-reduce was issued and passed.  The sampled single-stream steps (one in eight) are INSIDE the timed region (they cost ~0.2 % of `value`).
`roofline.by_family` = [{family, ms, frac, pipe}] for every instrumented family (largest first) and `matrix_time_frac` = per matrix pipe
the time-weighted fraction of that pipe's peak (sum of family ms x frac / sum of family ms): the whole step, not only the dominant kernel.
The dominant kernel counts the forward and data-gradient launches of one kernel body as ONE family.
`cpu_baseline` times the oracle's torch-CPU fp32 restatement of the same train step on the host cores (rank 0, N=1 only):
config 2 at batch 8, 1 warm-up + 3 timed steps -- a reported baseline, not the target.
`extra_configs` (N=1 only): short driver-visible runs of BASELINE configs 4 and 5's per-GPU workloads (bf16 512x512x3 / 4
classes / batch 8, and fp32 1024x1024x3 / 6 classes / batch 2), each with its own roofline block.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "semantic-segmentation-unet_amd"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before anything initialises HIP (RCCL needs dmabuf IPC here)

PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, chip table (dense, spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same table: ~2.5 PF dense bf16 (spec)
TRAFFIC_TAGS = ("r06", "r05")        # profiles/<tag>_*_pmc_traffic.json: this round's passes, else the last round's (the file name is reported)
WINOGRAD_MULT_RATIO = 2.25           # F(2x2,3x3): 36 direct multiplies per tile and channel pair -> 16

X6_PRODUCTS = 6                      # BF16x6: six bf16 piece products per fp32-grade product (hh hm mh hl lh mm)
ARITHMETIC = {"bf16x6": "fp32 operands as 3 bf16 pieces, 6 products, fp32 accumulate (3x3 forward / data gradient: csrc/winograd_x6.hip; transposed "
                        "convs, all three directions: csrc/convt_x6.hip; 3x3 weight gradient and everything else: native fp32)",
              "native": "fp32 (v_mfma_f32_32x32x2_f32 for every contraction)"}

FAMILY = {   # engine profile key -> (kernel description, winograd?, bf16 matrix pipe?, offline PMC traffic file)
    "conv3x3_fwd_winograd_x6": ("wino_x6_stream_stats_kernel (3x3 conv forward + BatchNorm sums, Winograd F(2x2,3x3), fp32-grade products as 6 x v_mfma_f32_32x32x16_bf16 on three-piece operands; one point row per wave, both operands from registers)", True, True, "x6_fwd_pmc_traffic.json"),
    "conv3x3_dgrad_winograd_x6": ("wino_x6_stream_kernel / _bnbwd (3x3 conv data gradient + producer BatchNorm-backward sums, Winograd F(2x2,3x3), fp32-grade products as 6 x v_mfma_f32_32x32x16_bf16)", True, True, "x6_dgrad_pmc_traffic.json"),
    "conv3x3_fwd_winograd_fused": ("wino_fused_stream_stats_kernel (3x3 conv forward + BatchNorm sums, Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32)", True, False, "wino_fwd_pmc_traffic.json"),
    "conv3x3_dgrad_winograd_fused": ("wino_fused_stream_kernel / _bnstats (3x3 conv data gradient + producer BatchNorm-backward sums, Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32)", True, False, "wino_dgrad_pmc_traffic.json"),
    "conv3x3_wgrad_winograd_fused": ("wino_wgrad_fused_kernel (3x3 conv weight gradient, Winograd F(2x2,3x3), reduce over tiles on v_mfma_f32_32x32x2_f32)", True, False, "wino_wgrad_pmc_traffic.json"),
    "conv3x3_fwd_bf16": ("conv_bf16_stream[_in]_stats_kernel_{128,64} (3x3 conv forward + BatchNorm sums, persistent implicit GEMM on v_mfma_f32_32x32x16_bf16)", False, True, "bf16_fwd_pmc_traffic.json"),
    "conv3x3_dgrad_bf16": ("conv_bf16_stream[_in]_bnbwd_kernel_{128,64} (3x3 conv data gradient + producer BatchNorm-backward sums, persistent implicit GEMM on v_mfma_f32_32x32x16_bf16)", False, True, "bf16_dgrad_pmc_traffic.json"),
    "conv3x3_wgrad_bf16": ("wgrad_bf16_dma_kernel (3x3 conv weight gradient, pixel contraction on v_mfma_f32_32x32x16_bf16, LDS-DMA staging)", False, True, "bf16_wgrad_pmc_traffic.json"),
}


GEMM_FAMILY = {   # the transposed convs (plain GEMMs, not candidates for the `roofline` block): engine profile key -> (kernel description, bf16 matrix pipe?)
    "convt_fwd_x6": ("convt_x6_fwd_stats_kernel (2x2/stride-2 transposed conv forward + BatchNorm sums, one GEMM over the input pixels, fp32-grade products as 6 x v_mfma_f32_32x32x16_bf16)", True),
    "convt_dgrad_x6": ("convt_x6_dgrad_kernel_{256,128} (transposed conv data gradient, GEMM over the input pixels, BF16x6)", True),
    "convt_wgrad_x6": ("convt_x6_wgrad_kernel + reduce (transposed conv weight gradient, GEMM over pixels with transposing LDS reads, BF16x6, split-K)", True),
    "convt_fwd": ("convt_fwd_stream_stats_kernel (transposed conv forward + BatchNorm sums on v_mfma_f32_32x32x2_f32)", False),
    "convt_dgrad": ("igemm_kernel (transposed conv data gradient on v_mfma_f32_32x32x2_f32)", False),
    "convt_wgrad": ("wgrad_kernel<1> + reduce (transposed conv weight gradient on v_mfma_f32_32x32x2_f32)", False),
    "convt_fwd_bf16": ("convt_bf16_fwd_stats_dma_kernel (transposed conv forward + BatchNorm sums on v_mfma_f32_32x32x16_bf16)", True),
    "convt_dgrad_bf16": ("convt_bf16_dgrad[_bnbwd]_dma_kernel (transposed conv data gradient + producer BatchNorm-backward sums on v_mfma_f32_32x32x16_bf16)", True),
    "convt_wgrad_bf16": ("convt_wgrad_bf16_kernel_64 + reduce (transposed conv weight gradient on v_mfma_f32_32x32x16_bf16)", True),
}

PEAK_HBM_SPEC_GBS = 8000.0           # same guide: HBM3E ~8 TB/s spec
PEAK_HBM_COPY_GBS = 6290.0           # ... and its measured copy bandwidth (the figure HBM-bound kernels are priced against)

HBM_FAMILY = {   # engine profile key -> description; `work` of these launches is ALGORITHMIC BYTES: every tensor the pass reads / writes, once
    "bn_apply": "bn_apply_kernel / bn_apply_pool_kernel (BatchNorm apply [+ 2x2 max pool]: r -> y [, pooled, winners])",
    "bn_bwd": "bn_bwd_reduce_*/bn_bwd_apply_* (BatchNorm backward: [reduce pass over dy, r] + apply pass dy, r -> dz; pooled layers add the un-pooled gradient on the fly)",
    "bn_stats": "bn_stats_kernel (BatchNorm statistics pass; only the class-map layer has no conv epilogue to take them from)",
    "pool": "maxpool_fwd/bwd_kernel (level 4: the dropout sits between BatchNorm and pool)",
    "dropout": "dropout_kernel (in place, mask from a counter hash)",
    "softmax_ce": "softmax_ce_kernel (softmax + cross-entropy + d logits, K classes)",
    "adam": "adam_keras_kernel (flat buffers: theta, g, m, v -> theta, m, v)",
    "first_layer_fwd": "conv3x3_first_mfma_fwd_kernel (first layer, Cin = image channels -> 64: window gathered from global memory, v_mfma_f32_32x32x2_f32, + BatchNorm sums)",
    "first_layer_wgrad": "first layer weight gradient (conv3x3_first_mfma_wgrad_kernel with a bf16 dz, the stencil kernel with an fp32 one)",
    "classmap_fwd": "classmap64_fwd_kernel (class map 64 -> K, a channel octet per lane)",
    "classmap_dgrad": "classmap64_dgrad_kernel (bf16 input gradient) / conv1x1_narrow_dgrad_kernel (fp32)",
    "classmap_wgrad": "classmap64_wgrad_kernel",
}


def train_flops_per_image(hw, channels, classes):
    """(total, part executed by Winograd kernels in fp32 mode, part executed by bf16 MFMA kernels in bf16 mode): algorithmic
    FLOPs of one training image, 2 flops/MAC over conv + transposed conv; backward = dgrad + wgrad, no dgrad for conv_1a
    (512x512x1, 2 classes -> 1154.00 G)."""
    eng = importlib.import_module(PKG + ".engine")
    res = {"conv_1": hw, "conv_2": hw // 2, "conv_3": hw // 4, "conv_4": hw // 8, "bott_": hw // 16,
           "up_4": hw // 16, "dec_4": hw // 8, "up_3": hw // 8, "dec_3": hw // 4, "up_2": hw // 4, "dec_2": hw // 2,
           "up_1": hw // 2, "dec_1": hw, "logits": hw}
    total = wide = 0.0
    for name, kind, cin, cout in eng.layer_table(channels, classes):
        s = next(v for k, v in res.items() if name.startswith(k))
        taps = {"conv3": 9, "deconv": 4, "conv1": 1}[kind]
        f = 2.0 * taps * s * s * cin * cout * (2 if name == "conv_1a" else 3)
        total += f
        if kind == "conv3" and cin % 64 == 0 and cout % 64 == 0:
            wide += f
    return total, wide


def synthetic(batch, channels, classes, hw, seed, device):
    import torch
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    img = torch.randn(batch, channels, hw, hw, generator=g)
    cls = torch.randint(0, classes, (batch, hw // 8, hw // 8), generator=g)
    cls = cls.repeat_interleave(8, 1).repeat_interleave(8, 2)
    lab = torch.nn.functional.one_hot(cls, classes).to(torch.int32)
    return img.to(device), lab.to(device)


def usable_cores():
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (os.cpu_count() reports the
    host's cores, and oversubscribing torch's thread pool on a 16-core share makes the baseline meaninglessly slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("UNET_CPU_BASELINE_THREADS", "32"))))


def cpu_baseline(hw, channels, classes, batch=8, steps=3):
    """config 2 at B=8, 1 warm-up + 3 timed steps of the oracle's torch-CPU fp32 restatement.  Prints a
    partial result after every timed step so that the parent still has a number if its time budget runs out."""
    import numpy as np
    import torch
    from oracle import unet_numpy as on
    from oracle import unet_torch as ot
    cores = usable_cores()
    torch.set_num_threads(cores)
    img, lab = on.synthetic_batch(batch, channels, classes, hw, hw, seed=1234)
    rng = np.random.default_rng(0)
    masks = {"drop_4": rng.integers(0, 2, (batch, 512, hw // 8, hw // 8)),
             "drop_b": rng.integers(0, 2, (batch, 1024, hw // 16, hw // 16))}
    net = ot.TorchUNet(classes, batch, channels, dtype=torch.float32)
    net.train_step(img, lab, masks)                                               # the 1 warm-up step (threads, allocator)
    t0 = time.time()
    for i in range(steps):
        net.train_step(img, lab, masks)
        dt = time.time() - t0
        print(json.dumps({"value": round(batch * (i + 1) / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
                          "sample": "1 warm-up + %d timed train steps of batch %d (%dx%dx%d, %d classes), oracle torch-CPU fp32 "
                                    "restatement, %.1f s timed" % (i + 1, batch, hw, hw, channels, classes, dt)}), flush=True)


def cpu_baseline_bounded(args, budget_s=420):
    """Run the CPU leg in a child process (never touches the GPU) with a hard time budget, so a slow or oversubscribed
    host cannot hold back the benchmark's JSON line; the child's last per-step line is the result."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(args.size),
           "--channels", str(args.channels), "--classes", str(args.classes), "--batch", str(args.batch)]
    out, note = "", ""
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            out, err = p.communicate(timeout=budget_s)
            if p.returncode != 0:
                note = "child failed: " + (err.strip().splitlines() or ["?"])[-1][:160]
        except subprocess.TimeoutExpired:
            p.kill()                                   # exactly the child this function started
            out, err = p.communicate()
            note = "stopped at the %d s budget" % budget_s
    except Exception as e:                              # noqa: BLE001
        note = "could not start: %r" % (e,)
    for line in reversed((out or "").strip().splitlines()):
        if line.startswith("{"):
            r = json.loads(line)
            if note:
                r["sample"] += " (" + note + ")"
            return r
    return {"value": None, "unit": "images/sec", "cores": usable_cores(), "kind": "port", "sample": "cpu baseline: " + (note or "no output")}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N child processes of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, rendezvous on 127.0.0.1), started from a parent that never initialises HIP and never exec()s.  Rank 0's
    stdout (the ONE JSON line) is relayed verbatim; the other ranks' stdout goes to stderr.  Returns the exit code: 0 only if every rank
    exited 0.  If a rank dies, exactly the children started here are terminated."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    chunks = []
    drain = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)     # rank 0's line may exceed a pipe buffer
    drain.start()
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    for q in pending:                       # a rank is gone: the others would wait in a collective forever
                        procs[q].terminate()
            time.sleep(0.05)
        drain.join(10)
        sys.stdout.write("".join(chunks))
        sys.stdout.flush()
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()
    return rc


def run_workload(model, dev, world, rank, size, channels, classes, batch, dtype, steps, warmup, no_overlap, kernel_events,
                 sample_every, barrier, fp32_matrix="bf16x6", max_workgroups=None, bucket_mb=25.0):
    """W warm-up + K timed optimizer steps of one workload -> dict with dt, per-family kernel figures, final loss."""
    import torch
    G = batch * world
    net = model.UNet(classes, G, channels, learning_rate=3e-4, device=dev, seed=0,
                     compute_dtype={"f32": "fp32", "bf16": "bf16"}[dtype])
    net.engine.overlap_wgrad = not no_overlap
    net.engine.opt.fp32_matrix = fp32_matrix
    net.engine.opt.max_workgroups = max_workgroups
    if world > 1:
        par = importlib.import_module(PKG + ".parallel")
        net.parallel = par.DataParallel(net.engine, bucket_bytes=int(bucket_mb * 1024 * 1024))
    img, lab = synthetic(batch, channels, classes, size, 1234 + rank, dev)
    inputs = (img, lab, None, None)          # no metric objects -> no per-step host sync inside the timed loop
    step = net.train_step if world == 1 else (lambda inp: net.dist_train_step(net.parallel, inp))   # N>1: + the loss SUM (X2)
    for _ in range(warmup):
        step(inputs)
    buckets = None
    if world > 1:
        # one extra untimed step with the collectives traced: when each gradient bucket's all-reduce was ISSUED (event on the weight-gradient
        # stream right in front of it) and by when the compute stream had passed the wait for it, from the start of the step
        net.parallel.trace, net.parallel.done_trace = [], []
        t0e = torch.cuda.Event(enable_timing=True); t0e.record()
        step(inputs)
        t1e = torch.cuda.Event(enable_timing=True); t1e.record(); torch.cuda.synchronize()
        buckets = {"bucket_mb": [round((b - a) * 4 / 1e6, 2) for a, b, _ in net.parallel.buckets],
                   "closes_behind": [last for _, _, last in net.parallel.buckets],
                   "issued_ms": [round(t0e.elapsed_time(ev), 3) for _, ev in net.parallel.trace],
                   "passed_ms": [round(t0e.elapsed_time(ev), 3) for _, ev in net.parallel.done_trace],
                   "step_ms": round(t0e.elapsed_time(t1e), 3),
                   "max_workgroups": net.engine.opt.max_workgroups}
        net.parallel.trace = net.parallel.done_trace = None
    prof = {} if kernel_events else None
    sampled = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        s = prof is not None and i % sample_every == sample_every // 2
        net.engine.profile = prof if s else None
        net.engine.overlap_wgrad = (not no_overlap) and not s      # sampled steps: one stream -> exclusive event durations
        sampled += int(s)
        step(inputs)
    barrier()
    dt = time.perf_counter() - t0
    net.engine.profile = None
    final_loss = float(net.engine.loss_buf[0].item())
    kernels = {}
    if prof:
        for key, evs in prof.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            fl = sum(f for _, _, f in evs)
            if key in HBM_FAMILY:           # HBM-bound pass: algorithmic bytes / exclusive time against the measured-copy and the spec bandwidth
                gbs = fl / (ms * 1e-3) / 1e9
                kernels[key] = {"bound": "hbm", "kernel": HBM_FAMILY[key], "launches_per_step": len(evs) // sampled,
                                "ms_per_step": round(ms / sampled, 3), "algorithmic_mb_per_step": round(fl / sampled / 1e6, 1),
                                "achieved_gbs": round(gbs, 1), "frac_of_copy_bw": round(gbs / PEAK_HBM_COPY_GBS, 4),
                                "frac_of_spec_bw": round(gbs / PEAK_HBM_SPEC_GBS, 4), "exclusive": True}
                continue
            if key in GEMM_FAMILY:
                desc, wino, bf16 = GEMM_FAMILY[key][0], False, GEMM_FAMILY[key][1]
            else:
                desc, wino, bf16, _ = FAMILY.get(key, (key, False, False, None))
            eff = fl / (ms * 1e-3) / 1e12
            x6 = key.endswith("_x6")
            grade = eff / WINOGRAD_MULT_RATIO if wino else eff           # fp32-grade multiply-adds actually performed (as FLOP/s)
            ex = grade * X6_PRODUCTS if x6 else grade                    # matrix-pipe FLOP/s executed (BF16x6: six bf16 products each)
            peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
            kernels[key] = {"kernel": desc, "launches_per_step": len(evs) // sampled, "ms_per_step": round(ms / sampled, 3),
                            "avg_launch_ms": round(ms / len(evs), 4), "effective_tflops": round(eff, 2),
                            "executed_tflops": round(ex, 2), "executed_frac": round(ex / peak, 4), "exclusive": True,
                            "pipe": "bf16 mfma" if bf16 else "fp32 mfma"}
            if x6:
                kernels[key]["fp32_grade_tflops"] = round(grade, 2)
                kernels[key]["fp32_grade_vs_fp32_mfma_peak"] = round(grade / PEAK_FP32_MFMA_TFLOPS, 4)
    del net
    torch.cuda.empty_cache()
    return {"dt": dt, "kernels": kernels, "final_loss": final_loss, "sampled_steps": sampled, "G": G, "buckets": buckets}


def baseline_config_id(args, world):
    """which BASELINE.json config the command-line workload is ("2", "3" = config 2's per-GPU work on N > 1 GPUs, "4", "5", else "custom")"""
    key = (args.size, args.channels, args.classes, args.batch, args.dtype)
    if key == (512, 1, 2, 8, "f32"):
        return "2" if world == 1 else "3"
    return {(512, 3, 4, 8, "bf16"): "4", (1024, 3, 6, 2, "f32"): "5"}.get(key, "custom")


# forward and data gradient of a route are ONE kernel body (x6_stream_body<0|1|2>, wino_fused_stream_body, conv_bf16_stream_body) launched on
# different operands: they count as one family when the dominant kernel is chosen, so the `roofline` block cannot skip the larger half of the step
STREAM_PAIRS = {
    "conv3x3_stream_winograd_x6": ("conv3x3_fwd_winograd_x6", "conv3x3_dgrad_winograd_x6"),
    "conv3x3_stream_winograd_fused": ("conv3x3_fwd_winograd_fused", "conv3x3_dgrad_winograd_fused"),
    "conv3x3_stream_bf16": ("conv3x3_fwd_bf16", "conv3x3_dgrad_bf16"),
}


def _traffic_of(tfile, workload_key, launches):
    """offline PMC passes of this round (scripts/collect_profiles.sh): config 2 / 4 files and the config-5 ones; the file must be FOR this workload"""
    for tname in [tag + mid + tfile for tag in TRAFFIC_TAGS for mid in ("_", "_config5_")]:
        tf = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tf):
            continue
        t = json.load(open(tf))
        if str(t.get("workload", "")).startswith(workload_key) and t.get("launches_per_step") == launches:
            return tname, t
    return None, None


def family_table(kernels):
    """[{family, ms, frac, pipe}] over every instrumented family, largest first: matrix-core families as executed FLOP/s over the dense peak
    of THEIR pipe, HBM-bound passes as algorithmic bytes over the measured copy bandwidth."""
    rows = []
    for k, v in kernels.items():
        if v.get("bound") == "hbm":
            rows.append({"family": k, "ms": v["ms_per_step"], "frac": v["frac_of_copy_bw"], "pipe": "hbm (6.29 TB/s copy)"})
        else:
            rows.append({"family": k, "ms": v["ms_per_step"], "frac": round(v["executed_frac"], 4), "pipe": v.get("pipe")})
    rows.sort(key=lambda r: -r["ms"])
    return rows


def matrix_time_frac(kernels):
    """per pipe: sum(family ms x family fraction of that pipe's peak) / sum(family ms) -- the time-weighted fraction of peak over the launches
    that run on the pipe (replaces round 5's mixed-pipe step_executed_frac, which priced bf16-pipe work against the fp32 peak)."""
    acc = {}
    for v in kernels.values():
        if v.get("bound") == "hbm":
            continue
        a = acc.setdefault(v.get("pipe"), [0.0, 0.0])
        a[0] += v["ms_per_step"] * v["executed_frac"]; a[1] += v["ms_per_step"]
    return {p: {"frac": round(a[0] / a[1], 4), "ms_per_step": round(a[1], 3)} for p, a in acc.items() if a[1] > 0}


def roofline_of(kernels, workload_key):
    """The 3x3 KERNEL with the most exclusive ms per step; the forward and data-gradient launches of one kernel body are one candidate."""
    cands = {}
    for k, v in kernels.items():
        if k not in FAMILY:
            continue
        pair = next((p for p, members in STREAM_PAIRS.items() if k in members), None)
        cands.setdefault(pair or k, []).append(k)
    if not cands:
        return None
    key = max(cands, key=lambda c: sum(kernels[k]["ms_per_step"] for k in cands[c]))
    members = sorted(cands[key])
    vs = [kernels[k] for k in members]
    desc, wino, bf16, _ = FAMILY[members[0]]
    if len(members) > 1:
        desc = " + ".join(FAMILY[k][0] for k in members)
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    x6 = members[0].endswith("_x6")
    ms = sum(v["ms_per_step"] for v in vs)
    launches = sum(v["launches_per_step"] for v in vs)
    wsum = lambda f: sum(v[f] * v["ms_per_step"] for v in vs) / ms          # time-weighted rate = total work / total time
    r = {"bound": "mfma", "kernel": desc, "family": key if len(members) > 1 else members[0], "members": members,
         "achieved": round(wsum("executed_tflops"), 2), "peak": peak, "unit": "TFLOP/s",
         "frac": round(wsum("executed_tflops") / peak, 4), "effective": round(wsum("effective_tflops"), 2),
         "flop_accounting": ("executed = 6 x algorithmic / 2.25 bf16-MFMA FLOP (Winograd F(2x2,3x3), six piece products per fp32-grade product) against the bf16 peak"
                             if x6 else "executed = algorithmic / 2.25 (Winograd F(2x2,3x3))" if wino else "executed = algorithmic (implicit GEMM)"),
         "launches_per_step": launches, "avg_launch_ms": round(ms / launches, 4), "ms_per_step": round(ms, 3),
         "timing": "HIP events on the launch stream, sampled timed steps, single-stream backward (exclusive)",
         "traffic": None, "traffic_source": None}
    if x6:
        r["fp32_grade_tflops"] = round(wsum("fp32_grade_tflops"), 2)
        r["fp32_grade_vs_fp32_mfma_peak"] = round(wsum("fp32_grade_tflops") / PEAK_FP32_MFMA_TFLOPS, 4)
    # HBM bytes per launch: the launch-weighted mean over the members' offline PMC files (every member needs one)
    found = [(_traffic_of(FAMILY[k][3], workload_key, kernels[k]["launches_per_step"]), kernels[k]["launches_per_step"]) for k in members]
    if all(t is not None for (_, t), _ in found):
        mean = lambda f: sum(t[f] * n for (_, t), n in found) / launches
        r["traffic"] = round(mean("hbm_bytes_per_launch"))
        r["algorithmic_bytes_per_launch"] = round(mean("algorithmic_bytes_per_launch"))
        ff = sorted({t.get("fetch_size_factor", 2.0) for (_, t), _ in found})
        r["traffic_source"] = "offline PMC (%s; FETCH_SIZE x%s + WRITE_SIZE, separate passes%s)" % (
            ", ".join("profiles/" + n for (n, _), _ in found), "/".join("%g" % f for f in ff),
            "; gfx950 correction" if ff == [2.0] else "; x1 = calibrated for these kernels' 16-byte gathers, the x2 reading is traffic_fetch_x2")
        if all("hbm_bytes_per_launch_fetch_x2" in t for (_, t), _ in found):
            r["traffic_fetch_x2"] = round(mean("hbm_bytes_per_launch_fetch_x2"))
        r["traffic_over_algorithmic"] = round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)
    r["by_family"] = family_table(kernels)
    r["matrix_time_frac"] = matrix_time_frac(kernels)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="contraction precision of the 3x3 layers; bf16 = BASELINE config 4 (fp32 master weights, fp32 accumulation)")
    ap.add_argument("--fp32-matrix", choices=["bf16x6", "native"], default="bf16x6",
                    help="fp32 mode: how the fused Winograd forward / data gradient multiply -- fp32-grade on the bf16 matrix pipe (three-piece "
                         "operands, six products; default) or the native fp32 matrix instruction")
    ap.add_argument("--max-workgroups", type=int, default=None,
                    help="cap on every persistent kernel's grid (e.g. 224 leaves ~4 CUs per XCD to RCCL's kernels; default: one workgroup per CU -- "
                         "profiles/r04_overlap_standin.txt is why)")
    ap.add_argument("--bucket-mb", type=float, default=25.0,
                    help="N > 1: size of the gradient all-reduce buckets (default 25; 1000 = one all-reduce behind the whole backward pass, no overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run weight gradients on the main stream in every step (A/B switch)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short config-4 / config-5 runs")
    ap.add_argument("--sample-every", type=int, default=8, help="kernel events on one timed step in this many")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        cpu_baseline(args.size, args.channels, args.classes, batch=args.batch)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU), BEFORE anything in this process has touched
        # the GPU (torch is not even imported yet), and relay rank 0's line.  Under torch.distributed.run WORLD_SIZE is set and this is skipped.
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    # UNET_BENCH_REHEARSAL=1: every rank on GPU 0, collectives over gloo -- runs the N>1 code path (rank-sharded synthetic data,
    # DataParallel buckets, barriers, MAX over ranks) on a ONE-GPU box.  The line is marked "rehearsal": it is not a scaling figure.
    rehearsal = world > 1 and os.environ.get("UNET_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model = importlib.import_module(PKG + ".model")
    sample_every = max(2, min(args.sample_every, args.steps)) if args.steps > 1 else 1
    res = run_workload(model, dev, world, rank, args.size, args.channels, args.classes, args.batch, args.dtype, args.steps,
                       args.warmup, args.no_overlap, not args.no_kernel_events, sample_every, barrier, args.fp32_matrix, args.max_workgroups, args.bucket_mb)
    dt = res["dt"]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def summarize(r, dt_, size, channels, classes, batch, dtype, steps, warmup, nworld):
        ips = r["G"] * steps / dt_
        total, wide = train_flops_per_image(size, channels, classes)
        peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_FP32_MFMA_TFLOPS
        executed = (total - wide) + (wide / WINOGRAD_MULT_RATIO if dtype == "f32" else wide)
        wk = "%dx%dx%d/%d classes/batch %d/%s" % (size, size, channels, classes, batch, dtype)
        return ips, {
            "ms_per_step": round(dt_ / steps * 1e3, 3),
            "roofline": roofline_of(r["kernels"], wk), "kernels": r["kernels"],
            "step_effective_tflops_per_gpu": round(ips / nworld * total / 1e12, 2),
            "step_executed_tflops_per_gpu": round(ips / nworld * executed / 1e12, 2),
            "matrix_time_frac": matrix_time_frac(r["kernels"]),
            "train_gflop_per_image": round(total / 1e9, 2), "final_loss": round(r["final_loss"], 6),
            "sampled_steps": r["sampled_steps"],
        }

    extras = []
    if world == 1 and not args.no_extra and (args.size, args.channels, args.classes, args.batch, args.dtype) == (512, 1, 2, 8, "f32"):
        other = "native" if args.fp32_matrix == "bf16x6" else "bf16x6"
        for (size, ch, kc, b, dty, label, fm, cid) in (
                (512, 3, 4, 8, "bf16", "BASELINE config 4 per-GPU workload", args.fp32_matrix, "4"),
                (1024, 3, 6, 2, "f32", "BASELINE config 5 per-GPU workload", args.fp32_matrix, "5/" + args.fp32_matrix),
                (512, 1, 2, 8, "f32", "BASELINE config 2 on the OTHER fp32 route (fp32_matrix = %s)" % other, other, "2/" + other)):
            st, wu = 16, 4
            try:
                r2 = run_workload(model, dev, 1, 0, size, ch, kc, b, dty, st, wu, False, True, 8, barrier, fm)
                ips2, s2 = summarize(r2, r2["dt"], size, ch, kc, b, dty, st, wu, 1)
                s2.pop("kernels")
                extras.append(dict({"id": cid, "workload": "%s: synthetic %dx%dx%d, %d classes, batch %d, %s" % (label, size, size, ch, kc, b, dty),
                                    "value": round(ips2, 3), "unit": "images/sec", "steps": st, "warmup": wu, "dtype": dty,
                                    "arithmetic": ARITHMETIC[fm] if dty == "f32" else "bf16 contractions, fp32 accumulation and master weights"}, **s2))
            except Exception as e:                      # noqa: BLE001 -- never lose the headline line to an extra run
                extras.append({"id": cid, "workload": label, "error": repr(e)[:300]})

    if rank == 0:
        ips, s = summarize(res, dt, args.size, args.channels, args.classes, args.batch, args.dtype, args.steps, args.warmup, world)
        out = {
            "metric": "training images/sec", "value": round(ips, 3), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": s.pop("ms_per_step"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "arithmetic": ARITHMETIC[args.fp32_matrix] if args.dtype == "f32" else "bf16 contractions, fp32 accumulation and master weights",
            "config": {"workload": "U-Net train step%s, synthetic %dx%dx%d, %d classes, batch %d/GPU, %s (BASELINE config %s)"
                                   % (" + RCCL all-reduce" if world > 1 else "", args.size, args.size, args.channels, args.classes, args.batch,
                                      args.dtype, baseline_config_id(args, world)),
                       "global_batch": res["G"], "parallelism": "dp%d" % world},
        }
        # every workload this run measured, compact and FIRST (a record that keeps only the head or only known keys still shows all of them);
        # the same list closes the line as `summary` (a record that keeps only the tail shows it too).  Full blocks: `extra_configs`.
        rf = s.get("roofline") or {}
        compact = [{"id": baseline_config_id(args, world) + ("" if args.dtype == "bf16" else "/" + args.fp32_matrix), "img_s": round(ips, 1),
                    "ms": out["ms_per_step"], "roofline_frac": rf.get("frac"), "roofline_family": rf.get("family")}]
        for e in extras:
            if "value" in e:
                r2 = e.get("roofline") or {}
                compact.append({"id": e["id"], "img_s": round(e["value"], 1), "ms": e["ms_per_step"], "roofline_frac": r2.get("frac"),
                                "roofline_family": r2.get("family")})
        compact[0]["roofline_by_family"] = [f for f in (rf.get("by_family") or []) if "mfma" in (f.get("pipe") or "")]
        compact[0]["matrix_time_frac"] = rf.get("matrix_time_frac")
        out["config"]["measured"] = "; ".join("%s: %.1f img/s" % (c["id"], c["img_s"]) for c in compact)
        out["configs"] = compact
        out.update(s)
        if world > 1:
            out["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                  "gradient_allreduce": res["buckets"]}
        if rehearsal:
            out["rehearsal"] = "all %d ranks share GPU 0, collectives over gloo: exercises the N>1 path, not a scaling measurement" % world
        out["cpu_baseline"] = cpu_baseline_bounded(args) if (world == 1 and not args.no_cpu_baseline) else None
        if extras:
            out["extra_configs"] = extras
        out["summary"] = compact
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
