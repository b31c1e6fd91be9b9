set -x
mkdir -p gpurun_out/s25
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_wide.py tests/test_hip_parity.py tests/test_hip_integrators.py -m gpu -x -q > gpurun_out/s25/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s25/tests.log
tail -5 gpurun_out/s25/tests.log
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > gpurun_out/s25/bench_${c}_$tag.log 2>gpurun_out/s25/bench_${c}_$tag.err && python profiles/summarize_bench.py gpurun_out/s25/bench_${c}_$tag.log
}
run ch4 c5 4 TUTU_TRACE_CHUNK=4
run ch1 c5 4 TUTU_TRACE_CHUNK=1
run ch16 c5 4 TUTU_TRACE_CHUNK=16
run ch4 c3 4 TUTU_TRACE_CHUNK=4
run ch16 c3 4 TUTU_TRACE_CHUNK=16
run ch4 c4 1 TUTU_TRACE_CHUNK=4
run ch16 c4 1 TUTU_TRACE_CHUNK=16
run ch4 c1 6 TUTU_TRACE_CHUNK=4
run ch4 c2 4 TUTU_TRACE_CHUNK=4
python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room 2>&1 | grep "^{" | cut -c1-140
TUTU_TRACE_CHUNK=2 python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room 2>&1 | grep "^{" | cut -c1-140
python - <<'PY'
import json
for c in ['c5','c2']:
    d=json.loads([l for l in open(f'gpurun_out/s25/bench_{c}_ch4.log') if l.startswith('{')][-1])
    print(c,'drop_in',{k:v for k,v in d.get('drop_in',{}).items() if not isinstance(v,(dict,str))})
PY
