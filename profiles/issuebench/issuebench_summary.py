#!/usr/bin/env python3
"""profiles/issuebench/issuebench -> profiles/issue_peak.json: the MEASURED vector issue peak of one SIMD, the unit of
`valu_issue_frac` (profiles/make_traffic.py, bench.py).   python profiles/issuebench/issuebench_summary.py <issue.json> <tag>
Peak = independent v_add_f32 at 8 waves per SIMD (the occupancy the traversal kernels run at); the whole table is kept
next to it so that a reader can see what one wave alone, packed instructions, the divide sequence etc. sustain."""
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
d = json.load(open(src))
tab = {}
for r in d["results"]:
    tab.setdefault(r["op"], {})[str(r["waves_per_simd"])] = round(r["per_simd_cycle"], 4)
peak = tab["v_add_f32"]["8"]
out = {"tag": tag, "device": d["device"], "valu_per_simd_cycle_peak": peak,
       "definition": "wave64 v_add_f32 instructions (independent) issued per SIMD and shader cycle at 8 waves per SIMD, s_memtime-bracketed",
       "cycles_per_wave64_valu": 1.0 / peak, "implied_clock_mhz": sorted(r["implied_clock_mhz"] for r in d["results"])[len(d["results"]) // 2],
       "table_per_simd_cycle_by_waves_per_simd": tab}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "issue_peak.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
