set -x
mkdir -p gpurun_out/s2
timeout -k 10 200 profiles/allocbench/allocbench 64 > gpurun_out/s2/allocbench.jsonl 2>&1
cat gpurun_out/s2/allocbench.jsonl
for a in "0" "168" "48" "12" "0 0"; do
  timeout -k 10 200 python tests/tools/cold_probe.py $a >> gpurun_out/s2/cold_probe.jsonl 2>&1
done
cat gpurun_out/s2/cold_probe.jsonl
timeout -k 10 500 python tests/tools/needle_probe.py exact > gpurun_out/s2/needle_probe.jsonl 2>&1
cat gpurun_out/s2/needle_probe.jsonl
timeout -k 10 300 python bench.py --config c2 --steps 10 --warmup 2 > gpurun_out/s2/bench_c2.json 2> gpurun_out/s2/bench_c2.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/s2/bench_c2.json') if x.startswith('{')]
d=json.loads(l[-1]); print('c2', round(d['value'],1), d['drop_in'], d['frame'])
PY
timeout -k 10 600 python -m pytest tests/test_hip_wide.py tests/test_hip_frames.py -m gpu -x -q -s -k "needles or far_origin or c2_ or c5_ or native" > gpurun_out/s2/tests.log 2>&1
tail -40 gpurun_out/s2/tests.log
