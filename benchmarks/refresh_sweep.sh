#!/bin/bash
# whole-episode rate against the period of the global order refresh
cd "${GRAFT_REPO_ROOT:-.}"
for r in ${REFRESH_PERIODS:-0 8 16 32}; do
  TTL_ORDER_REFRESH=$r timeout -k 10 120 python3 bench.py --no-cpu-baseline --windows 3 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('refresh', $r, 'value %.1f M' % (j['value']/1e6), 'whole episode %.1f M' % (j['whole_episode']['streamline_steps_per_s_rank0']/1e6), 'ms %.2f' % j['whole_episode']['ms'])"
done
