#!/bin/bash
# Same-box A/B of the step tail on bench.py's headline leg (262 144 rows), interleaved:
#   A = k_prefix + k_proc_scatter (TTL_TAIL_FUSED=0)
#   B = k_tail at every size (TTL_TAIL_FUSED_MAX_ROWS=1048576)
rounds=${1:-4}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then export TTL_TAIL_FUSED=0; unset TTL_TAIL_FUSED_MAX_ROWS; else export TTL_TAIL_FUSED=1 TTL_TAIL_FUSED_MAX_ROWS=1048576; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --legs weak 2>/dev/null > gpurun_out/tail_ab_$v.json
    python3 - "$v" "$r" <<'PY'
import json, sys
v, r = sys.argv[1:3]
d = json.load(open(f'gpurun_out/tail_ab_{v}.json'))
roof = d['roofline']
print(v, r, f"value {d['value']/1e6:.1f} M  ms/step {d['ms_per_step']:.4f}  k_state {roof['avg_launch_ms']:.4f}  "
      f"other {json.dumps({k: round(x, 4) for k, x in roof['other_kernels_ms_per_step'].items() if not isinstance(x, str)})}  "
      f"episode {d['whole_episode']['streamline_steps_per_s_rank0']/1e6:.1f} M", flush=True)
PY
  done
done
