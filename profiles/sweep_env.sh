#!/bin/bash
# usage: profiles/sweep_env.sh VAR v1 v2 ...   -- one-set (no overlap) bench per value, prints value / kernel ms
VAR=$1; shift
for v in "$@"; do
  env_line=$(env $VAR=$v TUTU_ONE_SET=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$VAR=$v $(echo "$env_line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(round(d['value'],1), {a:round(b,1) for a,b in k.items()})")"
done
