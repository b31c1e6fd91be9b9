#!/bin/bash
# A/B of two builds on the SAME box (boxes differ by up to 12 %, runs on one box by ~1 %):
#   profiles/ab.sh <base.so> <new.so> [reps] [configs...]      prints bench summaries, alternating the libraries
B=$1; N=$2; REPS=${3:-2}; shift 3
CFGS=${@:-c2}
O=gpurun_out/ab; mkdir -p $O
for rep in $(seq $REPS); do for c in $CFGS; do for lib in $B $N; do
  tag=$(basename $lib .so)_${c}_r$rep
  TUTU_HIP_LIB=$lib timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/$tag.log 2>&1
  python profiles/summarize_bench.py $O/$tag.log | sed "s#^$O/##" | cut -c1-250
done; done; done
