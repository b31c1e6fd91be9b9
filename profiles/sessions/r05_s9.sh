# round 5, session 9: the whole GPU suite on the tree with the shade kernel's window tables; then one kernel trace of the c2 frame for
# the question "where does the frame go beyond the kernels' own times" (profiles/frame_attribution.py)
set -x
O=gpurun_out/s9; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/tests.log
[ $rc -ne 0 ] && exit 1
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/$O/rp -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/bench_traced.log 2>&1); echo "rocprof rc=$?"
f=$(find $O/rp -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$O/c2_timeline.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w")
out.write("name,start_ns,end_ns,queue,stream\n")
for r in rows:
    name = r.get("Kernel_Name", "").split("(")[0][-70:].replace(",", ";")
    out.write(f'{name},{r.get("Start_Timestamp")},{r.get("End_Timestamp")},{r.get("Queue_Id")},{r.get("Stream_Id","")}\n')
out.close()
print(len(rows), "dispatches")
PY
rm -rf $O/rp
timeout -k 10 300 python bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline > $O/c2.log 2>&1; python profiles/summarize_bench.py $O/c2.log | cut -c1-250
