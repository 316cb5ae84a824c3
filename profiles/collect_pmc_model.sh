#!/bin/bash
# Counter passes behind the memory-pipeline model of the state gather
# ("where k_state_dd's time goes"):
#   gpurun --timeout 1100 -- 'bash profiles/collect_pmc_model.sh r02'
# One rocprofv3 --pmc pass per group (a block's slots are few: SQ 8, TCC 4,
# TA/TCP/TD a handful each), all on the same bench.py command (2 windows of
# 12 steps; no --stats / trace domains next to --pmc).  The per-kernel
# averages of every pass are appended to gpurun_out/profiles_<tag>/<tag>_pmc_kstate.txt
set -e
tag=${1:-r02}
extra=${2:-}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmc_$tag
R=gpurun_out/profiles_$tag
mkdir -p $O $R
OUT=$R/${tag}_pmc_kstate${extra:+_$extra}.txt
: > $OUT
CMD="python3 bench.py --no-cpu-baseline --no-whole-episode --windows 2"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rm -rf $O/p$i
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/p$i -- $CMD > /dev/null 2> $O/p$i.log || echo "pass $i failed: $group" >> $OUT
  echo "## pass $i: $group" >> $OUT
  python3 profiles/pmc_any.py $O/p$i k_state >> $OUT 2>&1 || true
  rm -rf $O/p$i
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
GRBM_GUI_ACTIVE GRBM_TA_BUSY
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TD_BUSY_sum
TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TD_SPI_STALL_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_sum
TCC_READ_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_NORMAL_EVICT_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum
GROUPS
wc -l $OUT
