set -x
mkdir -p gpurun_out/s47
export TMPDIR=/tmp
O=gpurun_out/s47
for rep in 1 2; do for st in 0 1 2 3; do
  for c in c2 c5; do
  TUTU_STAGGER=$st timeout -k 10 300 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/st${st}_${c}_r$rep.log 2>&1
  python profiles/summarize_bench.py $O/st${st}_${c}_r$rep.log | sed "s#^$O/##" | cut -c1-60
  done
done; done
