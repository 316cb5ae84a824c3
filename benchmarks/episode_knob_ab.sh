#!/bin/bash
# whole-episode rate of bench.py's headline leg against TTL_FUSE_MAX_ROWS (the
# largest batch of the one-launch step tail), interleaved on one box
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do
  for rows in 16384 32768 65536; do
    TTL_FUSE_MAX_ROWS=$rows timeout -k 10 200 python bench.py --no-cpu-baseline --legs weak --windows 3 2>/dev/null > gpurun_out/knob_$rows.json
    python3 -c "
import json; d=json.load(open('gpurun_out/knob_$rows.json')); e=d['whole_episode']
print('TTL_FUSE_MAX_ROWS=$rows', 'round $r', 'episode %.1f M  %.3f ms  steps %d free %d' % (e['streamline_steps_per_s_rank0']/1e6, e['ms'], e['steps'], e['free_running_tail_steps']), 'value %.1f M' % (d['value']/1e6), flush=True)"
  done
done
