#!/bin/bash
# HBM traffic of the state gather in the regime where HBM binds: one GPU's
# shard of BASELINE config 4 (145^3 x 45 volume = 585 MB packed, larger than
# the 256 MB Infinity Cache; 131072 streamlines, K = 100, float64 directions).
#   gpurun --timeout 900 -- 'bash profiles/collect_c4.sh r02'
# kernel-trace --stats pass + separate FETCH_SIZE / WRITE_SIZE passes of
# `benchmarks/bench_configs.py c4-shard`; summary -> <tag>_c4shard_*.
set -e
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/c4_$tag
R=gpurun_out/profiles_$tag
mkdir -p $O $R
T="timeout -k 10 300"
CMD="python3 benchmarks/bench_configs.py c4-shard"
$T $CMD > $R/${tag}_c4shard_bench.json 2> $O/plain.log
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > /dev/null 2> $O/stats.log
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > /dev/null 2> $O/fetch.log
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > /dev/null 2> $O/write.log
python3 profiles/pmc_summary.py ${tag}_c4shard $O/stats $O/fetch $O/write ${tag}_c4shard_pmc_traffic.json
cp profiles/${tag}_c4shard_kernel_stats.csv profiles/${tag}_c4shard_pmc_traffic.txt profiles/${tag}_c4shard_pmc_traffic.json $R/
rm -rf $O/stats $O/fetch $O/write
cat $R/${tag}_c4shard_bench.json
