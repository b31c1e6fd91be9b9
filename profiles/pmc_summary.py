#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one directory per counter group) per kernel: sum over dispatches of each counter,
plus dispatch counts.   python profiles/pmc_summary.py gpurun_out/pmc_<tag>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(name):
    if "tutu::" not in name:
        return None
    n = name.split("(")[0].replace("void ", "").replace("tutu::", "")
    return n


def main(root):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(set))
    files = glob.glob(f"{root}/g*/**/*counter_collection.csv", recursive=True)
    groups = [f[len(root):].lstrip("/").split("/")[0] for f in files]
    if len(groups) != len(set(groups)):
        raise SystemExit("refusing: more than one counter_collection.csv per counter group -- collect into an empty directory")
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if not k:
                continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
    out = {}
    for k in agg:
        out[k] = {c: v for c, v in sorted(agg[k].items())}
        out[k]["_dispatches"] = max(len(s) for s in calls[k].values())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
