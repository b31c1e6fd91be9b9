# round 5, session 43: c1 (16 spp: one pass, nothing overlaps) split into smaller passes on the four streams
O=gpurun_out/s43; mkdir -p $O
export TMPDIR=/tmp
for spp in 0 8 4 2; do
timeout -k 10 120 python bench.py --config c1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --spp-per-pass $spp > $O/c1_$spp.log 2>&1
python - <<PY
import json
d=json.loads([l for l in open('$O/c1_$spp.log') if l.startswith('{')][-1])
print(f"c1 spp-per-pass $spp: {d['value']:.0f} Ms/s {d['ms_per_step']:.2f} ms crc {d['frame']['crc32']} passes {d['config'].get('passes')} sets {d['config'].get('n_sets')}")
PY
done
