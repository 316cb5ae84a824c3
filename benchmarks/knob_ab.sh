#!/bin/bash
# Same-box A/B of one environment knob on bench.py's headline leg, interleaved:
#   gpurun -- 'bash benchmarks/knob_ab.sh TTL_MASK_CUBES 0 1 [rounds]'
knob=$1; a=$2; b=$3; rounds=${4:-3}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    env $knob=$v timeout -k 10 200 python bench.py --no-cpu-baseline --legs weak 2>/dev/null > gpurun_out/knob_ab_$v.json
    python3 - "$knob" "$v" "$r" <<'PY'
import json, sys
knob, v, r = sys.argv[1:4]
d = json.load(open(f'gpurun_out/knob_ab_{v}.json'))
roof = d['roofline']
print(f'{knob}={v}', r, f"value {d['value']/1e6:.1f} M  ms/step {d['ms_per_step']:.4f}  k_state {roof['avg_launch_ms']:.4f}  "
      f"other {json.dumps({k: round(x, 4) for k, x in roof['other_kernels_ms_per_step'].items() if not isinstance(x, str)})}  "
      f"episode {d['whole_episode']['streamline_steps_per_s_rank0']/1e6:.1f} M", flush=True)
PY
  done
done
