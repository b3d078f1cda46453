#!/usr/bin/env python3
"""Per-launch durations of one kernel in the last training step of a rocprofv3 kernel trace:
    python scripts/wgrad_launches.py gpurun_out/prof_<tag>/stats_train/train_kernel_trace.csv [name=wgrad] [per_step=22]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
per = int(sys.argv[3]) if len(sys.argv) > 3 else 22
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
last = d[-per:]
print(len(d), "launches;", "last step (us):", [round(x) for x in last], "sum", round(sum(last)))
