#!/bin/bash
R=$(pwd)
for cfg in c2 c5; do
for gw in "0 4" "1 4" "1 6" "1 8" "1 16" "0 4"; do
  set -- $gw
  TUTU_GREEDY=$1 TUTU_INNER_STEPS=$2 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > /tmp/line.json 2>/tmp/err.log || tail -5 /tmp/err.log
  python3 - <<PY
import json
d = json.loads([l for l in open("/tmp/line.json") if l.startswith("{")][-1])
r = d["roofline"]; t = r["traversal"]; m = r["measured_per_ray"]
print("$cfg greedy=$1 cap=$2 value %.1f excl closest %.1f any %.1f lanes node %.3f/%.3f leaf %.3f/%.3f N %.2f/%.2f T %.2f/%.2f" % (d["value"],
      r["exclusive_kernel_ms_per_step"]["k_trace_closest"], r["exclusive_kernel_ms_per_step"]["k_trace_any"], t["lanes_active_node_step_closest"], t["lanes_active_node_step_any"],
      t["lanes_active_leaf_step_closest"], t["lanes_active_leaf_step_any"], m["N_closest"], m["N_shadow"], m["T_closest"], m["T_shadow"]), flush=True)
PY
done
done
