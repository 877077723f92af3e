"""Raw per-dispatch values of one PMC counter: pmc_raw.py <counter_collection.csv> <counter> [kernel substring]"""
import csv, sys
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and (len(sys.argv) < 4 or sys.argv[3] in r["Kernel_Name"]):
        d = rows.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48], 0.0])
        d[1] += float(r["Counter_Value"])
for k in sorted(rows):
    print("%6d %-48s %14.1f" % (k, rows[k][0], rows[k][1]))
