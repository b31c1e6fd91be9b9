#!/bin/bash
# usage: bash profiles/sweep.sh "<bench args A>" "<bench args B>" ...   -> one summary line per variant
for a in "$@"; do
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline $a 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a', '| Ms/s', round(d['value'],1), '| ms/step', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
done
