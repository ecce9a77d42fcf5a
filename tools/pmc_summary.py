#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: average counter value per kernel name.
usage: pmc_summary.py <dir> [<dir> ...]   (prints JSON: {kernel: {counter: avg, n: dispatches}})"""
import csv
import glob
import json
import os
import sys

out = {}
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            c = row["Counter_Name"]
            e = out.setdefault(k, {}).setdefault(c, [0.0, 0])
            e[0] += float(row["Counter_Value"])
            e[1] += 1
print(json.dumps({k: {c: {"avg": v[0] / v[1], "n": v[1]} for c, v in cs.items()} for k, cs in out.items()}, indent=1))
