#!/usr/bin/env python3
"""one line per bench log:  python profiles/summarize_bench.py gpurun_out/<tag>/bench_*.log"""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads([x for x in open(f) if x.startswith("{")][0])
    except (IndexError, OSError):
        print(f, "NO LINE")
        continue
    r = d["roofline"]
    ex = r.get("exclusive_kernel_ms_per_step", {})
    print(f"{f}: {d['config'].get('id', '?')} {d['value']:.1f} Ms/s {d['ms_per_step']:.1f} ms | dominant {r['kernel']} frac {r['frac']:.3f} "
          f"launch {r['avg_launch_ms']:.3f} ms | exclusive step {r.get('exclusive_step_ms', 0):.1f} ms: "
          + " ".join(f"{k.replace('k_', '')}={v:.1f}" for k, v in ex.items())
          + " | N/T " + " ".join(f"{v:.1f}" for v in r["measured_per_ray"].values())
          + (f" | k_shade frac {r['per_kernel']['k_shade']['frac']:.3f}" if "per_kernel" in r else "")
          + (f" | cpu {d['cpu_baseline']['value']:.2f} on {d['cpu_baseline']['cores']}" if d.get("cpu_baseline") else ""))
