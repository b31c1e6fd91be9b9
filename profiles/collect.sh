#!/bin/bash
# Refreshes everything under profiles/ for a tag, in TWO gpurun calls (the bench lines quote profiles/traffic.json, which is
# made from the first call's PMC passes):
#   gpurun -- 'bash profiles/collect.sh <tag> pmc'     GPU tests; PMC passes for c2..c5
#   bash profiles/publish.sh <tag> pmc                 (here: traffic.json, PMC summaries, kernel stats -> profiles/)
#   gpurun -- 'bash profiles/collect.sh <tag> lines'   rocprofv3 kernel stats of the default bench command (four passes overlapped
#                                                      and with one pass in flight); bench lines of all five configs with the
#                                                      reference CPU baseline; shard balance of an 8-way split (c2, c4); the
#                                                      other integrators
#   bash profiles/publish.sh <tag> lines
TAG=${1:-r03}
STAGE=${2:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "$STAGE" = "pmc" ]; then
  # PMC_CONFIGS="c4 c5" restricts the stage to those configs and skips the tests (a gpurun call is 20 minutes at most: two calls)
  if [ -z "$PMC_CONFIGS" ]; then
    timeout -k 10 900 python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/gpu_tests.log
  fi
  # <config> <spp> <spp per pass, as bench.py sizes it at the config's full spp: bench.py refuses counters of another pass size>
  for a in "c2 256 64" "c3 128 32" "c4 116 29" "c5 256 64"; do
    set -- $a
    if [ -n "$PMC_CONFIGS" ] && ! echo " $PMC_CONFIGS " | grep -q " $1 "; then continue; fi
    bash profiles/run_pmc.sh $TAG $1 $2 $3 > $O/pmc_$1.log 2>&1; echo "pmc $1 rc=$?"
  done
else
  # (the rocprofv3 runs sit in THIS stage so that their per-kernel averages and the bench lines' avg_launch_ms come from one box)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.log 2>&1); echo "rocprof rc=$?"
  grep "^{" $O/bench_under_rocprof.log > $O/bench_line_under_rocprof.json
  # the same command with ONE pass in flight (kernels run one at a time: exclusive per-launch durations, cf. the bench line's roofline)
  (cd /tmp && export TUTU_SETS=1 && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_exclusive -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_exclusive_under_rocprof.log 2>&1); echo "rocprof exclusive rc=$?"
  for c in c2 c1 c3 c5; do
    timeout -k 10 400 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.log 2>&1; echo "bench $c rc=$?"
  done
  timeout -k 10 400 python bench.py --config c4 --steps 1 --warmup 1 > $O/bench_c4.log 2>&1; echo "bench c4 rc=$?"
  for c in c1 c2 c3 c4 c5; do grep "^{" $O/bench_$c.log > $O/bench_line_$c.json; done
  timeout -k 10 300 python bench.py --config c2 --emulate-shard 8 --shard -1 --steps 2 --warmup 1 2> /dev/null | grep "^{" > $O/shard_balance_c2.json; echo "shards c2 rc=$?"
  timeout -k 10 600 python bench.py --config c4 --emulate-shard 8 --shard -1 --steps 1 --warmup 1 2> /dev/null | grep "^{" > $O/shard_balance_c4.json; echo "shards c4 rc=$?"
  timeout -k 10 300 python profiles/bench_integrators.py --steps 3 2> /dev/null | grep "^{" > $O/integrators_bench.jsonl; echo "integrators rc=$?"
  # (round 4) the same on the veach room, and BDPT through the one-lane-per-unit kernel the stages replaced
  timeout -k 10 300 python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room --width 800 --height 600 2> /dev/null | grep "^{" | sed 's/^{/{"scene": "veach_room", /' >> $O/integrators_bench.jsonl
  TUTU_BDPT_UNIT_KERNEL=1 timeout -k 10 300 python profiles/bench_integrators.py --steps 3 --no-cpu 2> /dev/null | grep '"bdpt"' | sed 's/^{/{"bdpt_form": "unit kernel (TUTU_BDPT_UNIT_KERNEL=1)", /' >> $O/integrators_bench.jsonl
  TUTU_BDPT_UNIT_KERNEL=1 timeout -k 10 300 python profiles/bench_integrators.py --steps 3 --no-cpu --scene veach_room --width 800 --height 600 2> /dev/null | grep '"bdpt"' | sed 's/^{/{"scene": "veach_room", "bdpt_form": "unit kernel (TUTU_BDPT_UNIT_KERNEL=1)", /' >> $O/integrators_bench.jsonl
  python profiles/summarize_bench.py $O/bench_c*.log
fi
