set -x
mkdir -p gpurun_out/s24
export TMPDIR=/tmp
R=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s24/rocprof -- python3 $R/profiles/bench_integrators.py --steps 2 --no-cpu --scene veach_room > $R/gpurun_out/s24/bench.log 2>&1)
grep "^{" gpurun_out/s24/bench.log | cut -c1-160
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/s24/rocprof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs'])/1e6:10.2f} ms {int(r['Calls']):6d} calls  {r['Name'][:110]}")
PY
