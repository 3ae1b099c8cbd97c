"""Aggregates a rocprofv3 --pmc counter_collection.csv per kernel: mean of each counter over the kernel's dispatches.
usage: python tools/pmc_ubench.py <counter_collection.csv> [...]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("kernel".ljust(60), *[n[:22].rjust(23) for n in names])
for k, cs in acc.items():
    print(k[:60].ljust(60), *[(f"{sum(cs[n]) / len(cs[n]):.0f}" if n in cs else "-").rjust(23) for n in names])
