#!/bin/bash
# The judged profile set, from one box and one invocation:
#   gpurun --timeout 1150 -- 'bash profiles/collect.sh r04'
# Profiled passes run ONE leg of bench.py at a time (`--legs`, the same timed
# windows; without the forked CPU workers and without the episode-to-exhaustion
# tail whose small launches would dilute the per-kernel averages):
#   headline leg  bench.py --no-cpu-baseline --no-whole-episode --legs weak
#   HBM regime    bench.py --no-cpu-baseline --no-whole-episode --legs hbm   (145^3 volume, 131072 rows)
# per leg:
# 1. rocprofv3 --kernel-trace --stats        -> profiles/<tag>[_c4shard]_kernel_stats.csv,
#                                               <tag>[_c4shard]_bench_under_rocprof.json
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; TCC has 4 slots)
#                                             -> profiles/<tag>[_c4shard]_pmc_traffic.txt,
#                                                pmc_traffic.json / pmc_traffic_c4shard.json
# then 3. python3 bench.py (every leg, fresh pmc json files: roofline.achieved / frac filled)
#                                             -> profiles/<tag>_bench.json
# Results are copied under gpurun_out/profiles_<tag>/ (the only path that
# travels back); copy them into profiles/ afterwards.
set -e
tag=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof_$tag
R=gpurun_out/profiles_$tag
mkdir -p $O $R
T="timeout -k 10 400"
for leg in weak hbm; do
  sfx=""; js=pmc_traffic.json
  if [ $leg = hbm ]; then sfx="_c4shard"; js=pmc_traffic_c4shard.json; fi
  B="python3 bench.py --no-cpu-baseline --no-whole-episode --legs $leg"
  rm -f $O/stats$sfx.rows
  export TTL_GATHER_ROWS_LOG=$O/stats$sfx.rows
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$sfx -- $B > $O/stats$sfx.json 2> $O/stats$sfx.log
  # (the library logs the rows of every gather launch: with the fused step tail the
  # grid of the gather covers more slots than rows; exported, not passed through
  # `env`: nothing may sit between rocprofv3 and the program)
  rm -f $O/fetch$sfx.rows $O/write$sfx.rows
  export TTL_GATHER_ROWS_LOG=$O/fetch$sfx.rows
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch$sfx -- $B > $O/fetch$sfx.json 2> $O/fetch$sfx.log
  export TTL_GATHER_ROWS_LOG=$O/write$sfx.rows
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write$sfx -- $B > $O/write$sfx.json 2> $O/write$sfx.log
  unset TTL_GATHER_ROWS_LOG
  python3 profiles/pmc_summary.py ${tag}$sfx $O/stats$sfx $O/fetch$sfx $O/write$sfx $js
  cp $O/stats$sfx.json $R/${tag}${sfx}_bench_under_rocprof.json
  cp profiles/${tag}${sfx}_kernel_stats.csv profiles/${tag}${sfx}_pmc_traffic.txt profiles/$js $R/
  rm -rf $O/stats$sfx $O/fetch$sfx $O/write$sfx $O/fetch$sfx.rows $O/write$sfx.rows $O/stats$sfx.rows
done
# round 4: the learner leg (config 3's training step; SACAuto.update alone over 30 updates)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/learner -- python3 benchmarks/bench_training.py --config c3 > $O/learner.json 2> $O/learner.log
cp $(find $O/learner -name "*kernel_stats.csv" | head -1) $R/${tag}_learner_kernel_stats.csv
python3 profiles/learner_trace.py $O/learner > $R/${tag}_learner_update_trace.txt
tail -1 $O/learner.json > $R/${tag}_learner_under_rocprof.json
rm -rf $O/learner
# ... config 5's training step on one GPU's shard (oracle bonus on: ttl_oracle_segments,
# k_oracle_net_wg, k_oracle_bonus, k_replay_add next to the update's kernels)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 benchmarks/bench_training.py --config c5 > $O/c5.json 2> $O/c5.log
cp $(find $O/c5 -name "*kernel_stats.csv" | head -1) $R/${tag}_config5_kernel_stats.csv
tail -1 $O/c5.json > $R/${tag}_config5_under_rocprof.json
rm -rf $O/c5
# ... and the oracle network (config 5): fused kernel against the autocast module, counters
$T python3 benchmarks/bench_oracle_net.py > $R/${tag}_oracle_net_bench.jsonl 2> /dev/null
bash profiles/collect_pmc_oracle_net.sh $tag > /dev/null 2>&1 || true
cp profiles/${tag}_pmc_oracle_net.txt $R/ 2> /dev/null || true
timeout -k 10 900 python3 bench.py > $O/bench.json
cp $O/bench.json $R/${tag}_bench.json
python3 benchmarks/show_bench.py $R/${tag}_bench.json
