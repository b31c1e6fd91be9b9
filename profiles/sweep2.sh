#!/bin/bash
# overlapped bench over TUTU_SHADE_BPC x TUTU_TRACE_BPC (4 steps each)
for sb in ${SB:-4 8}; do for tb in ${TB:-3 4 5 8}; do
  v=$(TUTU_SHADE_BPC=$sb TUTU_TRACE_BPC=$tb python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "shade_bpc=$sb trace_bpc=$tb $v"
done; done
