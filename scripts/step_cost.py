#!/usr/bin/env python3
"""Reads the counter runs of scripts/step_cost.sh: per-launch instruction counts of k_play for each build, and each
build's difference from the one tagged `base` (for a study build: the cost of what it executes once more)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for tag in sorted(d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d))):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_play" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[tag] = {k: sum(v) / len(v) for k, v in acc.items()}
    try:
        t = json.loads(open(os.path.join(out, tag + ".time.json")).read().strip().splitlines()[-1])
        res[tag]["ms"] = t["roofline"]["avg_launch_ms"]
    except Exception:   # noqa: BLE001
        res[tag]["ms"] = float("nan")
keys = ["ms", "SQ_INSTS_SALU", "SQ_INSTS_VALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES"]


def row(r):
    return " ".join(f"{k[3:] if k.startswith('SQ_') else k}={r.get(k, float('nan')) / (1e6 if k.startswith('SQ') else 1):9.1f}" for k in keys)


for tag, r in res.items():
    print(f"{tag:10s} {row(r)}")
for tag, r in res.items():
    b = res.get("r2base" if tag.startswith("r2") else "base")
    if b and b is not r:
        print(f"{tag:10s} - base: {row({k: r.get(k, 0) - b.get(k, 0) for k in keys})}")
json.dump(res, open(os.path.join(out, "step_cost.json"), "w"), indent=1)
