#!/bin/bash
# Counter passes over the step's small kernels (k_advance, k_prefix,
# k_proc_scatter, scripted policy): where their 48 us per step go.
#   gpurun --timeout 600 -- 'bash profiles/collect_pmc_small.sh r03'
# One rocprofv3 --pmc pass per group on the headline leg (2 windows of 12
# steps); per-kernel averages -> gpurun_out/profiles_<tag>/<tag>_pmc_small.txt
tag=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/pmcs_$tag
R=gpurun_out/profiles_$tag
mkdir -p $O $R
OUT=$R/${tag}_pmc_small.txt
: > $OUT
CMD="python3 bench.py --no-cpu-baseline --no-whole-episode --legs weak --windows 2"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rm -rf $O/p$i
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/p$i -- $CMD > /dev/null 2> $O/p$i.log || echo "pass $i failed: $group" >> $OUT
  echo "## pass $i: $group" >> $OUT
  for k in k_advance k_prefix k_proc_scatter k_scripted; do
    python3 profiles/pmc_any.py $O/p$i $k >> $OUT 2>&1 || true
  done
  rm -rf $O/p$i
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH
SQ_WAIT_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ
GRBM_GUI_ACTIVE GRBM_SPI_BUSY GRBM_TA_BUSY GRBM_TC_BUSY
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
GROUPS
wc -l $OUT
