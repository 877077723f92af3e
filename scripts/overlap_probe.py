"""Stand-in collective under the backward pass (tests/test_gpu_overlap.py): step time and the wait of the stand-in's workgroups for CU slots,
for every workgroup cap x stand-in size x hold time, fp32 (BF16x6 route) and bf16 steps of BASELINE config 2 / 4.
usage: python scripts/overlap_probe.py > profiles/rNN_overlap_standin.txt"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import pkg
import test_gpu_overlap as T

CAPS = (0, 240, 224, 192)

print("one MI355X; step = forward + backward + Adam of the 8 x 1 x 512 x 512 batch; stand-in = unet_standin_collective(workgroups, 32 KB LDS, hold) per 25 MB bucket,")
print("released behind the bucket's last weight gradient on a third stream; wait = completion - max(release, previous completion) - hold per bucket; median of 5 steps")
for dtype in ("fp32", "bf16"):
    base = {}
    for cap in CAPS:
        r = T.run_step_with_standin(cap, dtype=dtype, standin=False, steps=6)
        base[cap] = r["step_ms"]
        print("%s max_workgroups=%-3d no stand-in           : step %6.2f ms" % (dtype, cap, r["step_ms"]), flush=True)
    for wgs in (16, 32):
        for hold in (300, 800, 1400):
            for cap in CAPS:
                r = T.run_step_with_standin(cap, hold_us=hold, wgs=wgs, dtype=dtype, steps=6)
                print("%s max_workgroups=%-3d stand-in %2d wg x %4d us: step %6.2f ms (%+5.2f vs the same cap alone, %+5.2f vs uncapped alone) | wait per bucket [ms]: %s" % (
                    dtype, cap, wgs, hold, r["step_ms"], r["step_ms"] - base[cap], r["step_ms"] - base[0], " ".join("%.2f" % w for w in r["wait_ms"])), flush=True)
