#!/usr/bin/env python3
"""HBM traffic per launch of the three pipeline stages, from rocprofv3 --pmc passes (profiles/run_pmc.sh).

    python profiles/make_traffic.py gpurun_out/pmc_<tag>  ->  profiles/traffic.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is
exact for 16 B/lane stores.  The record arrays of this pipeline are read and written as dwordx4 per lane."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def stage(name):
    if "k_trace<" in name:
        params = name.split("k_trace<")[1].split(">")[0].replace(" ", "").split(",")  # <LDS_SCENE, ANY, SPH>
        return "k_trace_any" if params[1] == "true" else "k_trace_closest"
    if "k_shade<" in name:  # the per-material-class launches at depth >= 1 (MODE 1..4); depth 0 and connect-only are separate
        mode = name.split("k_shade<")[1].split(",")[0].strip()
        return "k_shade" if mode in ("1", "2", "3", "4") else None
    return None


def main(root):
    val = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    for f in glob.glob(f"{root}/g*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            st = stage(row["Kernel_Name"])
            if st and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                val[st][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[st][row["Counter_Name"]].add(row["Dispatch_Id"])
    out = {}
    for st in val:
        n = max(len(v) for v in disp[st].values())
        fetch = 2.0 * val[st]["FETCH_SIZE"] * 1024.0
        write = val[st]["WRITE_SIZE"] * 1024.0
        out[st] = {"bytes_per_launch": (fetch + write) / n, "launches_profiled": n, "fetch_bytes_corrected": fetch, "write_bytes": write,
                   "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes, averaged over the profiled launches"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
