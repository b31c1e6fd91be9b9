#!/bin/bash
# PMC passes on one of the stand-in scenes:  bash profiles/run_pmc_scene.sh <tag> <scene>
TAG=$1; SCENE=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  mkdir -p $R/gpurun_out/pmc_${TAG}
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}/g$i -- python3 $R/tests/tools/bench_scenes.py $SCENE > $R/gpurun_out/pmc_${TAG}/g$i.log 2>&1 || echo "group $i failed"
done
echo done
