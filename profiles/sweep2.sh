#!/bin/bash
# overlapped (two work sets) bench over TUTU_SHADE_BPC x TUTU_TRACE_BPC
for sb in 3 4 6 8; do for tb in 2 3 4 5; do
  v=$(TUTU_SHADE_BPC=$sb TUTU_TRACE_BPC=$tb python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "shade_bpc=$sb trace_bpc=$tb $v"
done; done
