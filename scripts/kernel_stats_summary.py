"""rocprofv3 --kernel-trace --stats summary (<name>_kernel_stats.csv) -> per-step table.  usage: kernel_stats_summary.py <csv> <steps in the run> [title]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
print(sys.argv[3] if len(sys.argv) > 3 else sys.argv[1])
print("sum of kernel durations: %.2f ms per step (%d steps incl. warm-up)" % (sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6, steps))
print("%-52s %10s %10s %10s %10s %7s" % ("kernel", "calls/step", "ms/step", "avg us", "min us", "%"))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n); n = re.sub(r"\((?!.*\().*", "", n)[:52]
    if float(r["TotalDurationNs"]) / steps < 5e3:
        continue
    print("%-52s %10.1f %10.3f %10.1f %10.1f %7.2f" % (n, int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e6,
                                                    float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["Percentage"])))
