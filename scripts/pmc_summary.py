#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, launches and the per-launch mean/sum of
each counter.  usage: pmc_summary.py <dir> > summary.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    name = name.split("(")[0]
    return name[-90:]


def main(root):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(dict)     # kernel -> dispatch id -> ns (when the CSV carries timestamps)
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", "?"))
                c = row.get("Counter_Name", "?")
                a = acc[k][c]
                a[0] += 1
                a[1] += float(row.get("Counter_Value", 0) or 0)
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    dur[k][row.get("Dispatch_Id", len(dur[k]))] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    out = {}
    for k, cs in acc.items():
        out[k] = {c: {"launches": n, "sum": s, "per_launch": s / max(n, 1)} for c, (n, s) in cs.items()}
        if dur[k]:
            out[k]["_duration_ns"] = {"launches": len(dur[k]), "sum": sum(dur[k].values()),
                                      "per_launch": sum(dur[k].values()) / len(dur[k])}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
