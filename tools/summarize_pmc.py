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
        if "st_" not in k and "stokes" not in k:
            continue
        print(k, {c: f"{sum(v) / len(v):.4g}" for c, v in cs.items()}, f"n={len(next(iter(cs.values())))}")

# traffic.json for bench.py: FETCH_SIZE / WRITE_SIZE (KB) summed over the kernels of one vmult
try:
    import json
    per = {}
    for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "stfem" in k and ("st_sweep" in k or "fixup" in k or "st_general" in k):  # the kernels of a vmult, not the set-up ones
                acc[k][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
        for k, cs in acc.items():
            for c, v in cs.items():
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    per.setdefault(c, {})[k] = (sum(v), len(v))
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        # every vmult launches each kernel a fixed number of times; normalise by the fix-up count
        calls = min(n for (_, n) in per["FETCH_SIZE"].values())
        fetch = sum(s for (s, _) in per["FETCH_SIZE"].values()) / calls
        write = sum(s for (s, _) in per["WRITE_SIZE"].values()) / calls
        kern = ("st_sweep_pencil" if any("sweep_pencil" in k for k in per["FETCH_SIZE"]) else
                "st_sweep_cart_tile" if any("cart_tile" in k for k in per["FETCH_SIZE"]) else "st_sweep_cart_atomic")
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import sweep_source_hash
        json.dump({"kernel": kern, "fetch_kb_per_vmult": fetch, "write_kb_per_vmult": write, "sources_sha256": sweep_source_hash(),
                   "note": "sum over the kernels of one stfem_st_vmult (sweep launch(es) + fix-up)"},
                  open(os.path.join(d, "traffic.json"), "w"), indent=1)
        print("traffic.json:", fetch, write)
except Exception as e:  # noqa
    print("traffic.json not written:", e)
