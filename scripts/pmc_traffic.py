"""HBM traffic of one kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE are KiB per dispatch; they do not fit
one pass).  usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substrings, comma separated>
<launches per step> [workload key] [algorithmic bytes per step] [read factor, default 2]
Takes the LAST `launches per step` dispatches of the matching kernels (one whole step) and prints the JSON that bench.py reads as
`roofline.traffic`.  FETCH_SIZE is doubled: the gfx950 correction of MI355X_MICROARCH.md (wide coalesced reads are tallied at
half their bytes).  That correction is calibrated for wide coalesced streaming reads; a kernel with another access shape passes its own
factor, calibrated on a launch whose unique input bytes are known and exceed the Infinity Cache (the bf16 persistent conv kernels,
which gather 16-byte pieces at a 128-byte pixel stride: raw FETCH_SIZE = 1.06 x the input of 64->64 @512^2 x 8 -> factor 1)."""
import csv, json, sys
def per_dispatch(path, counter, subs):
    rows = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and any(s in r.get("Kernel_Name", "") for s in subs):
            rows[int(r["Dispatch_Id"])] = rows.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]
subs = sys.argv[3].split(",")
fetch = per_dispatch(sys.argv[1], "FETCH_SIZE", subs); write = per_dispatch(sys.argv[2], "WRITE_SIZE", subs)
n = int(sys.argv[4])
assert len(fetch) >= n and len(write) >= n, (len(fetch), len(write))
rf = float(sys.argv[7]) if len(sys.argv) > 7 else 2.0
rd = rf * 1024.0 * sum(fetch[-n:]); wr = 1024.0 * sum(write[-n:])
raw = 1024.0 * sum(fetch[-n:])
out = {"kernels": subs, "launches_per_step": n, "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr,
       "hbm_bytes_per_launch": (rd + wr) / n,
       # both readings of FETCH_SIZE side by side, whichever factor this file's headline figure uses
       "hbm_bytes_per_launch_fetch_x1": (raw + wr) / n, "hbm_bytes_per_launch_fetch_x2": (2.0 * raw + wr) / n}
if len(sys.argv) > 5:
    out["workload"] = sys.argv[5]
if len(sys.argv) > 6:
    alg = float(sys.argv[6])
    out.update({"algorithmic_bytes_per_step": alg, "algorithmic_bytes_per_launch": alg / n, "traffic_over_algorithmic": (rd + wr) / alg,
                "traffic_over_algorithmic_fetch_x1": (raw + wr) / alg, "traffic_over_algorithmic_fetch_x2": (2.0 * raw + wr) / alg})
out["fetch_size_factor"] = rf
out["note"] = (("FETCH_SIZE (KiB) doubled per the gfx950 correction in MI355X_MICROARCH.md (wide coalesced reads are tallied at half their bytes)"
                if rf == 2.0 else
                "FETCH_SIZE (KiB) x %.2f: these kernels gather 16-byte pieces at a 128-byte pixel stride, outside the guide's calibrated wide coalesced "
                "case; calibrated on 64->64 @512^2 x 8 (268 MB input + 268 MB output, beyond the Infinity Cache): raw FETCH_SIZE = 1.06 x the "
                "input's bytes (profiles/r02c_fetch_size_calibration.txt), so the raw value is used.  That is a ONE-layer calibration applied to all 17 "
                "layers of the family, whose deep members also re-read weights through L2 / Infinity Cache: a lower-bound argument (a 268 MB input cannot "
                "be read less than once), not a measurement per layer -- the guide's x2 reading is kept beside it in the *_fetch_x2 fields" % rf)
               + "; WRITE_SIZE (KiB) as read; two separate --pmc passes of `bench.py --steps 2 --warmup 1 --no-overlap`, last step")
print(json.dumps(out, indent=1))
