#!/bin/bash
# The judged profile set, from one box and one invocation:
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r01'
# (profiled passes run `bench.py --no-cpu-baseline --no-whole-episode`: the same timed
#  windows, without the forked CPU workers and without the episode-to-exhaustion tail whose
#  small launches would dilute the per-kernel averages)
# 1. bench.py alone            -> profiles/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command
#                               -> profiles/<tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; TCC has 4 slots)
#                               -> profiles/<tag>_pmc_traffic.txt, pmc_traffic.json
# 4. bench.py again with the fresh pmc_traffic.json (roofline.achieved / frac filled;
#    the line carries the whole-episode figure too)
# Results are copied under gpurun_out/profiles_<tag>/ (the only path that
# travels back); copy them into profiles/ afterwards.
set -e
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof_$tag
mkdir -p $O
T="timeout -k 10 400"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline --no-whole-episode > $O/stats.json 2> $O/stats.log
cp $O/stats.json $O/bench_under_rocprof.json
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --no-cpu-baseline --no-whole-episode > $O/fetch.json 2> $O/fetch.log
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --no-cpu-baseline --no-whole-episode > $O/write.json 2> $O/write.log
python3 profiles/pmc_summary.py $tag $O/stats $O/fetch $O/write
$T python3 bench.py > $O/bench.json
R=gpurun_out/profiles_$tag
mkdir -p $R
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc_traffic.txt profiles/pmc_traffic.json $R/
cp $O/bench.json $R/${tag}_bench.json
cp $O/bench_under_rocprof.json $R/${tag}_bench_under_rocprof.json
rm -rf $O/stats $O/fetch $O/write
cut -c1-600 $R/${tag}_bench.json
