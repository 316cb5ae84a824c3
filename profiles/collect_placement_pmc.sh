#!/bin/bash
# Counters of the state gather in its fast and slow placement modes (same box, same process):
#   gpurun --timeout 1100 -- 'bash profiles/collect_placement_pmc.sh r02'
set -e
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/placement_pmc_$tag
mkdir -p $O
OUT=$O/${tag}_placement_pmc.txt
: > $OUT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rm -rf $O/p$i
  export PLACEMENT_PMC_OUT=$PWD/$O/marker$i.json
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/p$i -- python3 benchmarks/placement_pmc.py > /dev/null 2> $O/p$i.log || echo "pass $i failed: $group" >> $OUT
  echo "## pass $i: $group" >> $OUT
  python3 benchmarks/placement_pmc_summary.py $O/p$i $O/marker$i.json >> $OUT 2>&1 || true
  rm -rf $O/p$i
done <<'GROUPS'
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum
GROUPS
cat $OUT
