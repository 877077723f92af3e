"""Per-layer table of the 3x3 kernel families INSIDE the training step, from ONE rocprofv3 pass of a single-stream step:
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d DIR -- \\
      python3 bench.py [--dtype bf16 --channels 3 --classes 4] --steps 2 --warmup 1 --no-extra --no-cpu-baseline --no-kernel-events --no-overlap
usage: layer_table.py <counter_collection.csv> <f32|f32native|bf16> <channels> <classes> [size=512] [batch=8]
(f32: the default BF16x6 route -- forward / data gradient are wino_x6_* kernels whose executed flops are 6 bf16 products per fp32-grade
product, priced against the dense bf16 peak; the weight gradient is the native fp32-MFMA kernel.  f32native: --fp32-matrix native.)
The last 17 dispatches of each family (= the last step) are matched to the layers by launch order: forward in Keras layer order, data and
weight gradient in backward order.  ms = the dispatch's duration under the counter pass (kernels are serialised there; a few % above the
--stats run), executed TFLOP/s = algorithmic 2*9*N*H*W*Cin*Cout / ms (/ 2.25 for the Winograd kernels), mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
plan = importlib.import_module("semantic-segmentation-unet_amd.plan")
path, dtype, C, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
size = int(sys.argv[5]) if len(sys.argv) > 5 else 512
batch = int(sys.argv[6]) if len(sys.argv) > 6 else 8
fams = {"f32": {"fwd": ["wino_x6_stream_stats_kernel"], "dgrad": ["wino_x6_stream_bnbwd_kernel", "wino_x6_stream_kernel"], "wgrad": ["wino_wgrad_fused_kernel"]},
        "f32native": {"fwd": ["wino_fused_stream_stats_kernel", "wino_fused_stats_kernel"], "dgrad": ["wino_fused_stream_bnbwd_kernel", "wino_fused_stream_kernel", "wino_fused_bnbwd_kernel", "wino_fused_kernel"],
                "wgrad": ["wino_wgrad_fused_kernel"]},
        "bf16": {"fwd": ["conv_bf16_stream_stats_kernel", "conv_bf16_stream_in_stats_kernel", "conv_bf16_stats_kernel"],
                 "dgrad": ["conv_bf16_stream_bnbwd_kernel", "conv_bf16_stream_kernel_", "conv_bf16_stream_in_bnbwd_kernel", "conv_bf16_stream_in_kernel_", "conv_bf16_bnbwd_kernel", "conv_bf16_kernel_"],
                 "wgrad": ["::wgrad_bf16_dma_kernel", "::wgrad_bf16_kernel"]}}[dtype]
rows = {}
for r in csv.DictReader(open(path)):
    d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
layers = [(n, ci, co) for n, kind, ci, co in plan.layer_table(C, K) if kind == "conv3" and n != "conv_1a"]
lvl = lambda n: 5 if n.startswith("bott") else int(n.split("_")[1][0])
fwd_order = layers
bwd_names = ["dec_1b", "dec_1a", "dec_2b", "dec_2a", "dec_3b", "dec_3a", "dec_4b", "dec_4a", "bott_b", "bott_a", "conv_4b", "conv_4a", "conv_3b", "conv_3a", "conv_2b", "conv_2a", "conv_1b"]
bwd_order = [next(l for l in layers if l[0] == n) for n in bwd_names]
wino = dtype in ("f32", "f32native")
x6 = lambda fam: dtype == "f32" and fam != "wgrad"            # six bf16 products per fp32-grade product, on the bf16 pipe
peak_of = lambda fam: 2500.0 if (x6(fam) or not wino) else 157.3
print("# %s, %dx%dx%d, %d classes, batch %d: the 3x3 families inside one single-stream training step (counter pass: %s)" % (dtype, size, size, C, K, batch, os.path.basename(os.path.dirname(path)) or path))
print("%-6s %-8s %5s %5s %5s %9s %11s %8s %9s %6s" % ("family", "layer", "Cin", "Cout", "HxW", "ms", "exec TF/s", "of peak", "mfma_busy", "GHz"))
for fam, order in (("fwd", fwd_order), ("dgrad", bwd_order), ("wgrad", bwd_order)):
    ds = [rows[k] for k in sorted(rows) if any(s in rows[k]["name"] for s in fams[fam])][-17:]
    assert len(ds) == 17, (fam, len(ds))
    tot_ms = tot_fl = 0.0
    for d, (n, ci, co) in zip(ds, order):
        hw = size >> (lvl(n) - 1)
        fl = 2.0 * 9 * batch * hw * hw * ci * co / (2.25 if wino else 1.0) * (6.0 if x6(fam) else 1.0)
        peak = peak_of(fam)
        ms = d["t"] / 1e6
        cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else float("nan")
        tot_ms += ms; tot_fl += fl
        print("%-6s %-8s %5d %5d %5d %9.4f %11.1f %8.3f %9.3f %6.2f" % (fam, n, ci, co, hw, ms, fl / ms / 1e9, fl / ms / 1e9 / peak, busy, cyc / d["t"] if cyc else 0.0))
    print("%-6s %-8s %27s %9.4f %11.1f %8.3f" % (fam, "TOTAL", "", tot_ms, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / peak_of(fam)))
