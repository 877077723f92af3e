"""Diagnostic: per-kernel timeline of the LAST train step in a rocprofv3 rocpd database (kernel trace of bench.py).

usage: python scripts/step_timeline.py x_results.db [substring]      prints start / end (us from the step's first kernel),
duration and queue of every kernel between two consecutive adam kernels; with a substring, only a +-3 kernel window around
the matching kernels.  Shows what actually overlaps what across the main and the weight-gradient stream.
"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else None
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else "0"
rows = list(db.execute("select name, start, end, %s from kernels order by start" % q))
adam = [i for i, r in enumerate(rows) if "adam" in r[0]]
assert len(adam) >= 2, "need two steps"
lo, hi = adam[-2] + 1, adam[-1] + 1
step = rows[lo:hi]
t0 = step[0][1]
sel = range(len(step))
if sub:
    hit = [i for i, r in enumerate(step) if sub in r[0]]
    sel = sorted({j for i in hit for j in range(max(0, i - 3), min(len(step), i + 4))})
prev_end = {}
for i in sel:
    n, s, e, qq = step[i]
    print("%9.1f %9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qq, re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:70]))
print("# step span %.3f ms, kernel time sum %.3f ms" % ((step[-1][2] - t0) / 1e6, sum(r[2] - r[1] for r in step) / 1e6))
# union of busy intervals: time in the step during which NO kernel runs, and the largest idle gaps with the kernel that ends them
iv = sorted((r[1], r[2], r[0]) for r in step)
busy_end, idle, gaps = iv[0][0], 0.0, []
for s, e, n in iv:
    if s > busy_end:
        idle += s - busy_end
        gaps.append(((s - busy_end) / 1e3, (s - t0) / 1e3, re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:60]))
    busy_end = max(busy_end, e)
print("# GPU idle inside the step (no kernel on any queue): %.3f ms in %d gaps" % (idle / 1e6, len(gaps)))
for g in sorted(gaps, reverse=True)[:12]:
    print("#   gap %7.1f us before t=%9.1f us  %s" % g)
