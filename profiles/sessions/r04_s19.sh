set -x
mkdir -p gpurun_out/s19
export TMPDIR=/tmp
for c in c5 c3 c4; do
for bpc in 0 6 5 4; do
st=4; [ $c = c4 ] && st=1
TUTU_TRACE_BPC=$bpc timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > gpurun_out/s19/bench_${c}_bpc$bpc.log 2>&1 && python profiles/summarize_bench.py gpurun_out/s19/bench_${c}_bpc$bpc.log
done
done
