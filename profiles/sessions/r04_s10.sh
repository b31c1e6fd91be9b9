set -x
mkdir -p gpurun_out/s10
timeout -k 10 300 python -m pytest tests/test_hip_wide.py -m gpu -q -s -k "flat_scan" > gpurun_out/s10/tests.log 2>&1; tail -5 gpurun_out/s10/tests.log
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --config c2 --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/s10/$tag.log 2>&1; python - <<PY
import json
l=[x for x in open('gpurun_out/s10/$tag.log') if x.startswith('{')]
d=json.loads(l[-1]); print('$tag', round(d['value'],1))
PY
}
run base A=1
run bpc7 TUTU_TRACE_BPC=7
run bpc6 TUTU_TRACE_BPC=6
run bpc5 TUTU_TRACE_BPC=5
run sbpc2 TUTU_SHADE_BPC=2
run sbpc3 TUTU_SHADE_BPC=3
run bpc6_sbpc2 TUTU_TRACE_BPC=6 TUTU_SHADE_BPC=2
run base2 A=1
run sets3 TUTU_SETS=3
run tree TUTU_FLAT=0
