#!/bin/bash
# Same-box A/B of two source trees on bench.py's headline leg: alternates
#   (A) an older checkout under _ab/<name> (git worktree, built in this container)
#   (B) this tree
# ROUNDS times and prints value / k_state / the other kernels of each run.
#   gpurun -- 'bash benchmarks/ab_trees.sh r02 3'
name=${1:-r02}
rounds=${2:-3}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  (cd _ab/$name && timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null) > gpurun_out/ab_${name}_A$r.json
  timeout -k 10 200 python bench.py --no-cpu-baseline --legs weak 2>/dev/null > gpurun_out/ab_${name}_B$r.json
  python3 - "$name" "$r" <<'PY'
import json, sys
name, r = sys.argv[1:3]
for side in 'AB':
    d = json.load(open(f'gpurun_out/ab_{name}_{side}{r}.json'))
    roof = d['roofline']
    print(side, r, f"value {d['value']/1e6:.1f} M  ms/step {d['ms_per_step']:.4f}  k_state {roof['avg_launch_ms']:.4f}  "
          f"other {json.dumps({k: round(v, 4) for k, v in roof['other_kernels_ms_per_step'].items() if not isinstance(v, str)})}  "
          f"episode {d['whole_episode']['streamline_steps_per_s_rank0']/1e6:.1f} M", flush=True)
PY
done
