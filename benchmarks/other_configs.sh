#!/bin/bash
# The other BASELINE shapes (profiles/<tag>_other_configs.jsonl) + the record-order A/B on
# config 4's shard (the volume that does not fit the Infinity Cache).
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
R=gpurun_out/profiles_$tag
mkdir -p $R
: > $R/${tag}_other_configs.jsonl
for c in c2-K100 c3-env c4-shard c1 c2-host; do
  timeout -k 10 200 python3 benchmarks/bench_configs.py $c 2>/dev/null | grep '^{' >> $R/${tag}_other_configs.jsonl
done
timeout -k 10 200 python3 benchmarks/profile_small.py 2>/dev/null | grep '^{' >> $R/${tag}_other_configs.jsonl
: > $R/${tag}_c4shard_layout_ab.jsonl
for rep in 1 2 3; do
  for lay in linear brick4; do
    TTL_SH_LAYOUT=$lay timeout -k 10 200 python3 benchmarks/bench_configs.py c4-shard 2>/dev/null | grep '^{' | sed "s/^{/{\"layout\": \"$lay\", /" >> $R/${tag}_c4shard_layout_ab.jsonl
  done
done
cut -c1-260 $R/${tag}_other_configs.jsonl
cut -c1-60,150-330 $R/${tag}_c4shard_layout_ab.jsonl
