# round 5, session 1: the round-4 tree on this round's first box -- GPU suite + one line per config (baseline of the round)
set -x
O=gpurun_out/s1; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for c in c2 c3 c5 c4; do
  st=3; [ $c = c4 ] && st=1
  timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/$c.log 2>&1
  python profiles/summarize_bench.py $O/$c.log | cut -c1-250
done
