#!/bin/bash
# PMC passes for one bench config (run on the GPU box through gpurun):  bash profiles/run_pmc.sh <tag> <config> <spp> <spp_per_pass>
# Counters are collected in their own runs (one rocprofv3 invocation per counter group, --kernel-trace only), at the
# BENCHMARKED pass size (spp_per_pass) with fewer passes, so that bytes per unit compare like for like with bench.py.
set -e
TAG=${1:-r02}
CFG=${2:-c2}
SPP=${3:-76}
SPASS=${4:-19}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${TAG}_${CFG}
# start from an EMPTY directory: a second collection into the same directory used to leave two runs' CSVs side by side
# and make_traffic.py summed them (round 2: every byte count doubled)
rm -rf $OUT
mkdir -p $OUT
# Launch geometry of the collection = the EXCLUSIVE step of bench.py (one pass in flight, every kernel with all the
# blocks that fit): that is the step whose per-kernel times the SQ counters are multiplied with (pipeline.valu).
export TUTU_SETS=${TUTU_SETS:-1}
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/bench.py --config $CFG --spp $SPP --spp-per-pass $SPASS --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/g$i.log 2>&1 || echo "group $i failed"
  echo "group $i done"
done
