set -x
mkdir -p gpurun_out/ab
for rep in 1 2; do for c in c4 c3; do for lib in build/r03/tuturenderer_amd/libtutu_hip.so build/dev/libtutu_hip_dev.so; do
  tag=$(basename $(dirname $(dirname $lib)))_$(basename $lib .so)_${c}_r$rep
  TUTU_HIP_LIB=$(pwd)/$lib timeout -k 10 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/ab/$tag.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/ab/$tag.log') if x.startswith('{')]
print('$tag', round(json.loads(l[-1])['value'],1) if l else open('gpurun_out/ab/$tag.log').read()[-400:])
PY
done; done; done
