# round 5, session 5: the whole GPU suite on the tree with the eight-wide walk (default: small trees), the RCCL leg of the C library's gather, the advisor's fixes
set -x
O=gpurun_out/s5; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log
[ $rc -ne 0 ] && exit 1
for c in c2 c3 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/$c.log 2>&1
  python profiles/summarize_bench.py $O/$c.log | cut -c1-250
done
