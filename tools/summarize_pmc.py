#!/usr/bin/env python3
"""Condenses rocprofv3 csv output (kernel_stats.csv + pmc*.csv) into per-kernel averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    for row in csv.DictReader(open(ks)):
        name = row.get("Name", "")[:90]
        print(f"{name:90s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} total_ns={row.get('TotalDurationNs')} pct={row.get('Percentage')}")
print("== counters: average per dispatch, per kernel ==")
for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")[:60]
        acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    for k, cs in acc.items():
        if "st_" not in k:
            continue
        print(k, {c: f"{sum(v) / len(v):.4g}" for c, v in cs.items()}, f"n={len(next(iter(cs.values())))}")
