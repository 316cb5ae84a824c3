set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/placement_pmc2
mkdir -p $O
OUT=$O/placement_pmc2.txt
: > $OUT
i=0
for group in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum"; do
  i=$((i+1))
  rm -rf $O/p$i
  export PLACEMENT_PMC_OUT=$PWD/$O/marker$i.json
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $O/p$i -- python3 benchmarks/placement_pmc.py > /dev/null 2> $O/p$i.log || echo "pass $i failed" >> $OUT
  echo "## pass $i: $group" >> $OUT
  python3 benchmarks/placement_pmc_summary.py $O/p$i $O/marker$i.json >> $OUT 2>&1 || true
  rm -rf $O/p$i
done
cat $OUT
