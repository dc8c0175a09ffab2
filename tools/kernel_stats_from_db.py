"""profiles/*_kernel_stats.csv from the database a `rocprofv3 --kernel-trace --stats -d DIR -o NAME` run leaves
(DIR/NAME_results.db): per-kernel calls, total, average, share, min, max, stddev (ns), like rocprofv3's own stats CSV.

  python tools/kernel_stats_from_db.py gpurun_out/prof_final/final_results.db profiles/r1_final_kernel_stats.csv
"""
import csv
import sqlite3
import statistics
import sys
from collections import defaultdict

rows = sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels").fetchall()
d = defaultdict(list)
for n, s, e in rows:
    d[n].append(e - s)
tot = sum(sum(v) for v in d.values())
out = sorted(((n, len(v), sum(v), sum(v) / len(v), 100 * sum(v) / tot, min(v), max(v), statistics.pstdev(v))
              for n, v in d.items()), key=lambda r: -r[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in out:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(r[4], 2), r[5], r[6], round(r[7], 3)])
