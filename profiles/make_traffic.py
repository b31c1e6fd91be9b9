#!/usr/bin/env python3
"""HBM traffic of the pipeline's kernels from rocprofv3 --pmc passes (profiles/run_pmc.sh), stored PER UNIT (per shaded
vertex / per ray) together with the config and pass size they were collected at -- bench.py refuses them for any other.

    python profiles/make_traffic.py gpurun_out/pmc_<tag>_<config> <tag> <config>   ->  updates profiles/traffic.json

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact for
16 B/lane stores.  The record arrays of this pipeline are read and written as dwordx4 per lane.  Units per kernel come
from the bench line of the same run (g2.log): extension rays for k_trace_closest and k_shade, shadow rays for k_trace_any.

Also per kernel, from the SQ counters of the same collection (totals over the same dispatches, one counter group per pass):
  valu_per_simd_cycle = SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs): vector wave-instructions issued per SIMD
                      and shader cycle (GRBM_GUI_ACTIVE is summed over the 8 XCDs);
  valu_issue_frac   = valu_per_simd_cycle / ISSUE_PEAK, where ISSUE_PEAK is MEASURED, not assumed: the rate of independent
                      `v_add_f32` at 8 waves per SIMD from profiles/issuebench (profiles/issue_peak.json, written by
                      issuebench_summary.py) -- the roofline of a kernel that does not touch memory.  Round 2 assumed 1/4
                      (4 cycles per wave64 instruction); the guide states 1/2 for >= 2 waves per SIMD;
  valu_lanes_active = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): lanes switched on in an average vector instruction;
  l2_hit_rate       = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum);  wave_cycles_waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES;
  salu_per_valu     = SQ_INSTS_SALU / SQ_INSTS_VALU."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def stage(name):
    if "k_trace<" in name:
        params = name.split("k_trace<")[1].split(">")[0].replace(" ", "").split(",")  # <LDS_SCENE, ANY, SPH, DEEP, WIDE>
        return "k_trace_any" if params[1] == "true" else "k_trace_closest"
    if "k_trace_flat<" in name:
        params = name.split("k_trace_flat<")[1].split(">")[0].replace(" ", "").split(",")  # <ANY, SPH>
        return "k_trace_any" if params[0] == "true" else "k_trace_closest"
    if "k_trace_wide8<" in name:
        params = name.split("k_trace_wide8<")[1].split(">")[0].replace(" ", "").split(",")  # <ANY, SPH, EARLY>
        return "k_trace_any" if params[0] == "true" else "k_trace_closest"
    if "k_trace_wide<" in name:
        params = name.split("k_trace_wide<")[1].split(">")[0].replace(" ", "").split(",")  # <ANY, SPH, EARLY>
        return "k_trace_any" if params[0] == "true" else "k_trace_closest"
    if "k_shade<" in name:  # the stage launches at depth >= 1 (MODE 1..4, 6); depth 0 (MODE 0) and connect-only (MODE 5) are separate
        mode = name.split("k_shade<")[1].split(",")[0].strip()
        return {"0": "k_shade_depth0", "5": "k_shade_connect_only"}.get(mode, "k_shade")
    for k in ("k_list_count", "k_list_scan", "k_list_scatter", "k_resolve", "k_finalize", "k_primary"):
        if k in name:
            return "other"
    return None


def main(root, tag, cfg):
    val = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    files = glob.glob(f"{root}/g*/**/*counter_collection.csv", recursive=True)
    per_group = defaultdict(list)
    for f in files:
        per_group[os.path.relpath(f, root).split(os.sep)[0]].append(f)
    dup = {g: fs for g, fs in per_group.items() if len(fs) > 1}
    if dup:  # round 2: two collections in one directory were summed and every byte count doubled
        raise SystemExit(f"refusing: more than one counter_collection.csv in {sorted(dup)} -- collect into an empty directory (run_pmc.sh does)")
    for f in files:
        for row in csv.DictReader(open(f)):
            st = stage(row["Kernel_Name"])
            if st and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_THREAD_CYCLES_VALU",
                                              "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_WAIT_ANY",
                                              "SQ_WAVE_CYCLES"):
                val[st][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[st][row["Counter_Name"]].add(row["Dispatch_Id"])
    line = json.loads([x for x in open(f"{root}/g2.log") if x.startswith("{")][0])
    # the counters cover every dispatch of the run: the line's launch counts must too (an extra, uncounted frame doubled every
    # byte count once more while this tool was being written -- and 5.9 TB/s slipped under the 8 TB/s check)
    n_disp = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] in ("FETCH_SIZE",) and stage(row["Kernel_Name"]) == "k_shade":
                n_disp[row["Dispatch_Id"]] = 1
    want = line["roofline"]["per_kernel"]["k_shade"]["launches"]
    if n_disp and len(n_disp) != want:
        raise SystemExit(f"refusing: {len(n_disp)} k_shade dispatches were profiled, the bench line accounts for {want}")
    pk = line["roofline"]["per_kernel"]
    units = {k: pk[k]["units_per_launch"] * pk[k]["launches"] for k in pk}
    samples = line["value"] * 1e6 * line["ms_per_step"] * 1e-3 * line["steps"]
    here = os.path.dirname(os.path.abspath(__file__))
    issue_peak = None
    try:
        issue_peak = float(json.load(open(os.path.join(here, "issue_peak.json")))["valu_per_simd_cycle_peak"])
    except (OSError, ValueError, KeyError):
        pass
    knobs = line["config"].get("knobs", {})
    kernels, total = {}, 0.0
    for st in val:
        fetch = 2.0 * val[st]["FETCH_SIZE"] * 1024.0
        write = val[st]["WRITE_SIZE"] * 1024.0
        total += fetch + write
        n = max(len(disp[st][c]) for c in ("FETCH_SIZE", "WRITE_SIZE") if c in disp[st])
        e = {"launches_profiled": n, "fetch_bytes_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": (fetch + write) / n}
        if st in units and units[st] > 0:
            e["units"] = units[st]
            e["hbm_bytes_per_unit"] = (fetch + write) / units[st]
        v = val[st]
        if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_INSTS_VALU"):
            e["valu_per_simd_cycle"] = v["SQ_INSTS_VALU"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
            if issue_peak:
                e["valu_issue_frac"] = e["valu_per_simd_cycle"] / issue_peak
            e["salu_per_valu"] = v.get("SQ_INSTS_SALU", 0.0) / v["SQ_INSTS_VALU"]
        if v.get("TCC_HIT_sum") or v.get("TCC_MISS_sum"):
            e["l2_hit_rate"] = v.get("TCC_HIT_sum", 0.0) / (v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0))
        if v.get("SQ_WAVE_CYCLES"):
            e["wave_cycles_waiting"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]  # s_waitcnt / barrier
        if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_THREAD_CYCLES_VALU"):
            e["valu_lanes_active"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"])
        kernels[st] = e
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
    try:
        out = json.load(open(path))
        if not all(isinstance(v, dict) and "kernels" in v for v in out.values()):
            out = {}
    except (OSError, ValueError):
        out = {}
    # sanity: no kernel can move more than the HBM peak (8 TB/s) -- the check that would have caught the doubled collection
    for st, e in kernels.items():
        ms = pk.get(st, {}).get("avg_launch_ms")
        if ms and e.get("hbm_bytes_per_unit") and pk[st].get("units_per_launch"):
            tbs = e["hbm_bytes_per_unit"] * pk[st]["units_per_launch"] / (ms * 1e-3) / 1e12
            e["implied_TB_per_s"] = tbs
            if tbs > 8.0:
                raise SystemExit(f"refusing: {st} would move {tbs:.1f} TB/s, above the HBM peak -- the collection is inconsistent")
    out[cfg] = {"tag": tag, "config": cfg, "sets": knobs.get("sets_default"), "shade_bpc": knobs.get("shade_bpc"),
                "issue_peak_valu_per_simd_cycle": issue_peak, "spp": line["config"]["workload"], "spp_per_pass": line["config"]["spp_per_pass"],
                "samples": samples, "hbm_bytes_per_sample": total / samples, "kernels": kernels,
                "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes; per unit = / the run's own ray counts"}
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out[cfg], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
