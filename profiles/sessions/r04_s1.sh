set -x
mkdir -p gpurun_out/s1
timeout -k 10 200 profiles/allocbench/allocbench 64 > gpurun_out/s1/allocbench.jsonl 2>&1
cat gpurun_out/s1/allocbench.jsonl
timeout -k 10 400 python tests/tools/needle_probe.py > gpurun_out/s1/needle_probe.jsonl 2>&1
cat gpurun_out/s1/needle_probe.jsonl
for mp in 8 16 32 48 96 168; do
  timeout -k 10 120 python bench.py --config c2 --steps 4 --warmup 2 --no-extras --no-cpu-baseline --max-paths $((mp*1048576)) > gpurun_out/s1/c2_mp$mp.json 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/s1/c2_mp$mp.json') if x.startswith('{')]
d=json.loads(l[-1]); print('max_paths Mi', $mp, 'Msamples/s', round(d['value'],1), 'spp/pass', d['config']['spp_per_pass'], 'passes', d['config']['passes_per_step'])
PY
done
