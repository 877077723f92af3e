"""MFMA-pipe occupancy of kernels from one rocprofv3 PMC pass.
usage: pmc_mfma_busy.py <counter_collection.csv> [kernel substring]
Pass: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -- python3 <bench>
GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles = value / 8); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles);
clock = cycles / duration.  Prints one line per dispatch of the matching kernels, in dispatch order."""
import csv, sys
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
        continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "t": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(rows):
    d = rows[k]
    cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    print("%6d %-60s %8.1f us  clock %.2f GHz  mfma_busy %.3f  wait_inst_any %.3f" % (
        k, d["name"][:60], d["t"] / 1e3, cyc / d["t"], d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc),
        d.get("SQ_WAIT_INST_ANY", 0.0) / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
