#!/bin/bash
# A/B of two builds on the same box: profiles/ab.sh <base.so> <new.so> [reps]   (one-set kernel times + overlapped value)
B=$1; N=$2; REPS=${3:-2}
for rep in $(seq $REPS); do for lib in $B $N; do
  one=$(TUTU_HIP_LIB=$lib TUTU_ONE_SET=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']; print(round(d['value'],1), {a:round(b,1) for a,b in k.items()})")
  ovl=$(TUTU_HIP_LIB=$lib python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "$(basename $lib) overlapped=$ovl one_set=$one"
done; done
