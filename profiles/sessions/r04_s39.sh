set -x
mkdir -p gpurun_out/s39
export TMPDIR=/tmp
O=gpurun_out/s39
for rep in 1 2; do
for sp in 0 128 32 86; do
timeout -k 10 300 python bench.py --config c2 --steps 4 --warmup 2 --no-cpu-baseline --no-extras $( [ $sp != 0 ] && echo --spp-per-pass $sp ) > $O/bench_c2_sp${sp}_$rep.log 2> $O/bench_c2_sp${sp}_$rep.err; python profiles/summarize_bench.py $O/bench_c2_sp${sp}_$rep.log | cut -c1-70
done; done
