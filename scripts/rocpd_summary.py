"""Summarise a rocprofv3 rocpd database (kernel trace) as a per-kernel table: calls, total / average duration, share.

usage: python scripts/rocpd_summary.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_bench_kernel_stats.txt

The 3x3 forward/dgrad kernel is launched 34 times per step (17 forward, then 17 dgrad); the header splits its launches by that
position, because only the forward launches run alone on the GPU (bench.py's roofline uses them).
"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace, %d kernels, %.3f ms of kernel time in total (%.3f ms per step over %d steps incl. warm-up)" % (sum(r[1] for r in rows), tot / 1e6, tot / 1e6 / steps, steps))
print("%-100s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for n, c, t, a, mn, mx in rows:
    print("%-100s %8d %12.3f %10.1f %10.1f %10.1f %6.2f" % (n[:100], c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))

# forward launches run the statistics variant (training), dgrad launches the plain persistent kernel
for kname, what in (("wino_fused_stream_stats_kernel", "FORWARD launches (exclusive on the GPU; bench.py's roofline uses them)"),
                    ("wino_fused_stream_bnbwd_kernel", "dgrad launches that also leave the producer's BatchNorm-backward sums (share the GPU with the side-stream weight gradients)"),
                    ("wino_fused_stream_kernel", "other dgrad launches (same sharing)")):
    ev = [e - s0 for s0, e in db.execute("select start, end from kernels where name like ?", ("%" + kname + "(%",))]
    if ev:
        print("# %s %s: %d, average %.1f us" % (kname, what, len(ev), sum(ev) / len(ev) / 1e3))
