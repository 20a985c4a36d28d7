"""Summarise rocprofv3 csv output (kernel stats + PMC counters) into a small text report."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print("== kernel-trace stats ==")
for f in find("trace/**/*kernel_stats.csv"):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 8:
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== PMC (mean per dispatch, per kernel) ==")
for d in ("pmc1", "pmc2", "pmc3", "pmc4", "pmc5"):
    for f in find(f"{d}/**/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")[:60]
            acc[name][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
        for name, ctrs in acc.items():
            if "stft2048" not in name and "logmel" not in name:
                continue
            if d == "pmc2" and "SQ_INSTS_MFMA" in ctrs:
                n = sum(ctrs["SQ_INSTS_MFMA"]) / len(ctrs["SQ_INSTS_MFMA"])
                print(f"    (MFMA instructions per launch {n:.4g}: 4x4x1_16b blocks = 512 flop each -> {n * 512 / 1e9:.3f} GFLOP per launch)")
            print(d, name)
            for c, v in ctrs.items():
                print(f"    {c:28s} mean {sum(v)/len(v):.4g}  (n={len(v)})")
