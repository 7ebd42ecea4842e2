#!/usr/bin/env python
"""Sums rocprofv3 --pmc counter_collection.csv per (kernel, counter)."""
import csv, re, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^(]{0,20}>)?)", k).group(1)
    a = agg[(k, r["Counter_Name"])]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    if "at::" in k or "torch" in k:
        continue
    print(f"{k:40s} {c:28s} n={n:4d} avg={v / n:.4g}")
