#!/bin/bash
# PMC passes for the bench workload (run on the GPU box through gpurun):  bash profiles/run_pmc.sh <tag> [spp]
# Counters are collected in their own runs (one rocprofv3 invocation per counter group, --kernel-trace only).
set -e
TAG=${1:-r01}
SPP=${2:-64}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}/g$i -- python3 $R/bench.py --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}/g$i.log 2>&1 || echo "group $i failed"
  echo "group $i done"
done
