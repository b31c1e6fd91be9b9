#!/bin/bash
run() { c=$1; spass=$2; steps=$3
  timeout -k 10 300 python bench.py --config $c --spp-per-pass $spass --steps $steps --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/sw.log 2>&1
  python - <<PY
import json
ok=False
for l in open('gpurun_out/sw.log'):
    if l.startswith('{'):
        ok=True; d=json.loads(l); print('$c spp_per_pass $spass ->', d['config'].get('spp_per_pass'), 'passes', d['config'].get('passes_per_step'), 'ms', round(d['ms_per_step'],2), round(d['value'],1), d['frame']['crc32'], flush=True)
if not ok: print('$c $spass FAILED', open('gpurun_out/sw.log').read()[-300:])
PY
}
for s in 0 64 74 86 103 128; do run c2 $s 3; done
for s in 0 16 22 32 43; do run c3 $s 2; done
for s in 0 43 64 86; do run c5 $s 2; done
for s in 0 16 22 32; do run c4 $s 1; done
