# round 5, session 17: the default of 336 Mi path slots in flight -- the whole GPU suite, then one line per config
set -x
O=gpurun_out/s17; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/gpu_tests.log; tail -4 $O/gpu_tests.log
for c in c2 c1 c3 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/$c.log 2>&1
  python profiles/summarize_bench.py $O/$c.log | cut -c1-230
done
timeout -k 10 300 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline > $O/c4.log 2>&1; python profiles/summarize_bench.py $O/c4.log | cut -c1-230
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/s17/c2.log') if l.startswith('{')][-1]); print('drop_in', {k:(round(v,3) if isinstance(v,float) else v) for k,v in d['drop_in'].items() if k!='note'})
PY
