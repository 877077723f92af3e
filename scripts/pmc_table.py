"""Per-kernel means of every counter in one or more rocprofv3 counter_collection.csv files: pmc_table.py <kernel substring> <csv>..."""
import csv, sys
acc = {}
for path in sys.argv[2:]:
    per = {}
    for r in csv.DictReader(open(path)):
        if sys.argv[1] not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40], int(r["Dispatch_Id"]))
        d = per.setdefault(k, {"dur_us": (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (name, _), d in per.items():
        for c, v in d.items():
            acc.setdefault(name, {}).setdefault(c, []).append(v)
for name, d in acc.items():
    print(name)
    for c in sorted(d):
        v = d[c]
        print("   %-32s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
