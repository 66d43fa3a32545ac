#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel: python profiles/summarize_pmc.py <dir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add((path, r["Dispatch_Id"]))
res = {k: dict(v, dispatches=len(calls[k]) // max(1, len({p for p, _ in calls[k]}))) for k, v in out.items()}
print(json.dumps(res, indent=1, sort_keys=True))
