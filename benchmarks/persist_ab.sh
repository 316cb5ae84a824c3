#!/bin/bash
# A/B of TTL_GATHER_PERSIST_ROWS (gathers of <= 98304 rows as ONE resident round of
# workgroups walking the 20-row blocks with a stride) on the strong-scaling
# shard sizes: three interleaved rounds on one box.
#   bash benchmarks/persist_ab.sh > gpurun_out/persist_ab.log
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for knob in 0 98304; do
    echo "== round $round TTL_GATHER_PERSIST_ROWS=$knob"
    TTL_GATHER_PERSIST_ROWS=$knob python benchmarks/rows_sweep.py 262144 65536 32768 2>/dev/null
  done
done
