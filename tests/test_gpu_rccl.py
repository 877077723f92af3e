"""GPU: the RCCL code path of parallel.DataParallel with a single-rank `nccl` process group (the box has one GPU): the
bucketed all-reduces are issued from the backward schedule on the side stream, Adam waits for them, and the result is
identical to the non-distributed step.  Runs in a child process so the process group does not leak into other tests."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
model = importlib.import_module("semantic-segmentation-unet_amd.model")
par = importlib.import_module("semantic-segmentation-unet_amd.parallel")
from oracle import unet_numpy as on
img, lab = on.synthetic_batch(2, 1, 2, 64, 64, seed=3)
rng = np.random.default_rng(0)
masks = {"drop_4": rng.integers(0, 2, (2, 512, 8, 8)), "drop_b": rng.integers(0, 2, (2, 1024, 4, 4))}
for cd in ("bf16", "fp32"):                                 # the mixed-precision mode goes through the same buckets / hooks
    a = model.UNet(2, 2, 1, seed=5, compute_dtype=cd)
    b = model.UNet(2, 2, 1, seed=5, compute_dtype=cd)
    b.parallel = par.DataParallel(b.engine, bucket_bytes=8 << 20, force=True)
    assert len(b.parallel.buckets) >= 5
    for _ in range(2):
        la = a.train_step((img, lab, None, None), dropout_masks=masks).numpy()
        lb = b.train_step((img, lab, None, None), dropout_masks=masks).numpy()
        assert la == lb, (cd, la, lb)
    assert torch.equal(a.engine.theta, b.engine.theta)      # bit-identical: SUM over one rank is the identity
red = b._reduce_loss(b.parallel, b.train_step((img, lab, None, None), dropout_masks=masks))
assert np.isfinite(red.numpy())
b.parallel.average_moving_stats()
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_PATH_OK")
'''


def test_rccl_bucket_allreduce_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert "RCCL_PATH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
