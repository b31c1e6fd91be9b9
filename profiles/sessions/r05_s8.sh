# round 5, session 8: where a BDPT frame of the veach room goes, kernel by kernel (rocprofv3 --kernel-trace --stats)
set -x
O=gpurun_out/s8; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rp -- python3 $R/profiles/bench_integrators.py --scene veach_room --width 800 --height 600 --steps 3 --no-cpu > $R/$O/bench.log 2>&1); echo "rc=$?"
cat $O/bench.log | cut -c1-200
f=$(find $O/rp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
cp $f $O/kernel_stats.csv; rm -rf $O/rp
