# round 5, session 46: L1 accesses and L1 -> L2 requests of the eight-wide walk with and without the top of the tree in LDS (c5):
# does staging remove accesses that were cheap anyway (lanes of a wave on one node)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s46; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
export TUTU_SETS=1
cd /tmp
for top in 0 40; do
export TUTU_WIDE8_TOP=$top
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/top$top -- python3 $R/bench.py --config c5 --spp 256 --spp-per-pass 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/top$top.log 2>&1 || echo "pmc top$top failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/busy$top -- python3 $R/bench.py --config c5 --spp 256 --spp-per-pass 64 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/busy$top.log 2>&1 || echo "pmc busy$top failed"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for top in (0, 40):
    for kind in ("top", "busy"):
        f = glob.glob(f"gpurun_out/s46/{kind}{top}/**/*counter_collection.csv", recursive=True)
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for fn in f:
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"]
                name = "closest" if "k_trace_wide8<false" in k else ("any" if "k_trace_wide8<true" in k else None)
                if name: acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        for name, d in acc.items():
            print(f"wide8_top {top} {name}: " + ", ".join(f"{c} {v / 1e9:.3f} G" for c, v in sorted(d.items())))
PY
