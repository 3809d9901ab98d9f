#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection / kernel stats CSVs per kernel (sum over dispatches, and per launch)."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "")[:60]
out = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = short(row.get("Kernel_Name", "?")); c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
        agg[k][c] += v; cnt[k].add(row.get("Dispatch_Id"))
    for k in agg:
        out.setdefault(k, {}).update({c: v for c, v in agg[k].items()}); out[k]["_dispatches_" + os.path.basename(os.path.dirname(f))] = len(cnt[k])
for f in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row.get("Name", "?")); out.setdefault(k, {}).update({"calls": int(row["Calls"]), "total_ns": float(row["TotalDurationNs"]), "avg_ns": float(row["AverageNs"]), "pct": float(row["Percentage"])})
keys = sorted(out, key=lambda k: -out[k].get("total_ns", 0))
for k in keys[:24]:
    print(k); print("   ", json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in sorted(out[k].items())}))
