"""Summarise rocprofv3 PMC csvs written by tools/pmc_gemm.sh: median per-dispatch counter values and kernel duration."""
import csv
import statistics
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
for sub in sorted(p for p in root.iterdir() if p.is_dir()):
    cc = next(sub.glob("*_counter_collection.csv"), None)
    kt = next(sub.glob("*_kernel_trace.csv"), None)
    if not cc:
        continue
    vals = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(cc)):
        vals[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = defaultdict(list)
    if kt:
        for r in csv.DictReader(open(kt)):
            dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, cs in vals.items():
        if "gemm" not in k and "quant" not in k and "absmax" not in k:
            continue
        d = statistics.median(dur[k]) if dur.get(k) else float("nan")
        print(f"[{sub.name}] {k}  n={len(next(iter(cs.values())))}  median duration {d:.1f} us")
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} median {statistics.median(v):16.0f}")
