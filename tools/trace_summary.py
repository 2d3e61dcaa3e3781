"""Median kernel duration per (kernel, grid) from a rocprofv3 kernel trace csv: tools/trace_summary.py <csv> [substr...]"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:] or [""]
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in keys):
        d[(n.replace("void (anonymous namespace)::", "")[:48], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(f"{k[0]:48s} grid {k[1]:>8s} x {k[2]:>4s}  n={len(v):4d}  median {statistics.median(v):8.2f} us  min {min(v):8.2f}")
