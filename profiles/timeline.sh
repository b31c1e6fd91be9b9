#!/bin/bash
# kernel timeline of one bench step (rocprofv3 --kernel-trace), reduced to a small csv: name,start_ns,end_ns,stream/queue
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/timeline
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/rp -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1); echo "rocprof rc=$?"
f=$(find $O/rp -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$O/timeline_small.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
out = open(sys.argv[2], "w")
for r in rows:
    name = r.get("Kernel_Name", "")
    short = name.split("(")[0][-60:]
    out.write(f'{short},{r.get("Start_Timestamp")},{r.get("End_Timestamp")},{r.get("Queue_Id")},{r.get("Stream_Id","")}\n')
out.close()
print(len(rows), "dispatches")
PY
rm -rf $O/rp
