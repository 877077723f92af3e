"""GPU: the data-parallel path with TWO ranks and the real HIP engine.  The box has one GPU, and RCCL refuses two ranks on one
device, so both ranks use cuda:0 and the collectives go over gloo (same torch.distributed calls, same bucket schedule, same
parallel.DataParallel object; the RCCL transport itself is covered with one rank in test_gpu_rccl.py).  What is checked is what
the reference's MirroredStrategy step computes for a global batch split over two replicas (UNet/model.py:204-235,
UNet/train.py:57-61,85-90): rank-0 weights broadcast, per-replica BatchNorm, loss / GLOBAL batch, SUM of gradients, one
Keras-Adam step, replicas bit-identical afterwards."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import unet_numpy as on
from test_gpu_unet import grad_errors, make_case

pytestmark = pytest.mark.gpu

CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
model = importlib.import_module("semantic-segmentation-unet_amd.model")
par = importlib.import_module("semantic-segmentation-unet_amd.parallel")
from test_gpu_unet import make_case
n, c, k, hw = 2, 1, 2, 32
img, lab, prm, masks = make_case(61, n, c, k, hw)
if rank != 0:                                   # replicas start from DIFFERENT weights: the rank-0 broadcast must fix that
    prm = {key: v + np.float32(0.5) for key, v in prm.items()}
net = model.UNet(k, n, c, learning_rate=3e-4)   # global batch 2, this replica holds image `rank`
net.engine.load_parameters(prm)
net.parallel = par.DataParallel(net.engine, bucket_bytes=%(bucket)d)
e = net.engine
sl = slice(rank, rank + 1)
mr = {kk: v[sl] for kk, v in masks.items()}
theta0 = e.theta.clone()
loss = net.dist_train_step(net.parallel, (img[sl], lab[sl], None, None), dropout_masks=mr)
torch.cuda.synchronize()
relu = {name: (e.saved[name][1].float().permute(0, 3, 1, 2) > 0).cpu().numpy() for name, kind, _, _ in e.layers if kind != "deconv"}
pidx = {"pool_%%d" %% l: e.idx[l].permute(0, 3, 1, 2).cpu().numpy().astype(np.int64) for l in (1, 2, 3, 4)}
g = e.export_gradients()
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), theta0=theta0.cpu().numpy(), theta=e.theta.cpu().numpy(), loss=float(loss.numpy()),
         own_loss=float(e.loss_buf[0].item()), nb=len(net.parallel.buckets),
         **{"g:" + kk: v for kk, v in g.items()}, **{"relu:" + kk: v for kk, v in relu.items()}, **{"pool:" + kk: v for kk, v in pidx.items()})
dist.barrier()
dist.destroy_process_group()
print("TWO_RANK_OK", rank)
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run_two(script, timeout=420):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                                  # exactly the children started above
            raise
    return outs


@pytest.mark.parametrize("bucket", [25 * 1024 * 1024, 1 << 20])
def test_two_rank_hip_step_equals_the_global_batch_semantics(tmp_path, bucket):
    outs = _run_two(CHILD % dict(root=ROOT, out=str(tmp_path), bucket=bucket))
    for r, (o, e) in enumerate(outs):
        assert "TWO_RANK_OK %d" % r in o, o[-1500:] + e[-3000:]
    res = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    n, c, k, hw = 2, 1, 2, 32
    img, lab, prm, masks = make_case(61, n, c, k, hw)
    assert np.array_equal(res[0]["theta0"], res[1]["theta0"])             # X5: rank 0's variables everywhere before the step
    g_sum, loss_sum = None, 0.0
    for r in range(2):
        sl = slice(r, r + 1)
        ref = on.OracleUNet(k, n, c, params=prm, dtype=np.float64)
        relu = {kk[5:]: res[r][kk] for kk in res[r].files if kk.startswith("relu:")}
        pidx = {kk[5:]: res[r][kk] for kk in res[r].files if kk.startswith("pool:")}
        loss_r, _, g_r, _, _ = ref.loss_and_grads(img[sl], lab[sl], {kk: v[sl] for kk, v in masks.items()}, relu_masks=relu, pool_idx=pidx)
        assert abs(float(res[r]["own_loss"]) - loss_r) < 1e-5 * abs(loss_r)
        loss_sum += loss_r
        g_sum = g_r if g_sum is None else {kk: g_sum[kk] + g_r[kk] for kk in g_r}
    for r in range(2):
        assert abs(float(res[r]["loss"]) - loss_sum) < 1e-5 * abs(loss_sum)          # X2: SUM of the per-replica losses, on every rank
        errs = grad_errors({kk[2:]: res[r][kk] for kk in res[r].files if kk.startswith("g:")}, g_sum)
        for l in (1, 2, 3, 4):
            errs.pop("up_%d/bias" % l)                                     # exactly zero gradient (see the branch-decision test)
        worst = max((v, key) for key, v in errs.items())
        assert worst[0] < 5e-4, sorted(errs.items(), key=lambda t: -t[1])[:5]         # (tolerance: see test_two_replicas_compose...)
        assert int(res[r]["nb"]) >= (5 if bucket == 1 << 20 else 2)
    assert np.array_equal(res[0]["theta"], res[1]["theta"])               # replicas in lock-step after the Adam step
    assert not np.array_equal(res[0]["theta"], res[0]["theta0"])


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_two_rank_rehearsal(launcher):
    """bench.py's N > 1 path on the one-GPU box (UNET_BENCH_REHEARSAL=1 puts both ranks on GPU 0 over gloo), started both ways a driver may
    start it: through torch.distributed.run (one process per rank), and as plain `python bench.py --gpus 2` (bench.py then starts the two
    ranks itself from a parent that never touches the GPU).  The ONE line must carry the whole-job aggregate."""
    env = dict(os.environ, UNET_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    head = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(_free_port())] if launcher == "torchrun" else [sys.executable])
    cmd = head + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--size", "256", "--no-extra"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
    assert d["cpu_baseline"] is None and "rehearsal" in d and np.isfinite(d["final_loss"])
    assert d["configs"] == d["summary"] and d["configs"][0]["img_s"] == pytest.approx(d["value"], abs=0.06)
