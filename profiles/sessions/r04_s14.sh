set -x
mkdir -p gpurun_out/s14
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_integrators.py -m gpu -q -x -k "stages or larger_frame" > gpurun_out/s14/tests.log 2>&1; tail -4 gpurun_out/s14/tests.log
R=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s14/rocprof -- python3 $R/profiles/bench_integrators.py --steps 2 --no-cpu > $R/gpurun_out/s14/bench.log 2>&1)
grep "^{" gpurun_out/s14/bench.log | cut -c1-120
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/s14/rocprof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:10]:
    print(f"{float(r['TotalDurationNs'])/1e6:10.2f} ms {int(r['Calls']):6d} calls  {r['Name'][:100]}")
PY
