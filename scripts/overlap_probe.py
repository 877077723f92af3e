"""Stand-in collective under the backward pass, capped vs uncapped fused Winograd weight gradient (tests/test_gpu_overlap.py):
prints the per-bucket release -> completion times for both; redirect into profiles/rNN_overlap_standin.txt."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import pkg
import test_gpu_overlap as T

for cap in (224, 0, 192):
    r = T.run_step_with_standin(cap, steps=4)
    print("wgrad_workgroups=%-3d last_wgrad %.2f ms | bucket release->done [ms]: %s | latency [ms]: %s" % (
        cap, r["last_wgrad_ms"], " ".join("%.1f->%.1f" % (a, b) for a, b in zip(r["ready_ms"], r["done_ms"])),
        " ".join("%.2f" % (b - a) for a, b in zip(r["ready_ms"], r["done_ms"]))))
