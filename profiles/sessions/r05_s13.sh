# round 5, session 13: the suite on the final tree (-> profiles/r05_a_gpu_tests.log) and two lines with the gather figures
set -x
O=gpurun_out/s13; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/gpu_tests.log; tail -4 $O/gpu_tests.log
for c in c3 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/$c.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open('$O/$c.log') if l.startswith('{')][-1]); print('$c', round(d['value'],1), json.dumps(d['roofline']['per_kernel']['k_trace_closest'].get('gather'))[:400])
PY
done
