"""Per-kernel means of rocprofv3 --pmc counters: python tools/pmc_by_kernel.py <counter_collection.csv> [name filter]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
