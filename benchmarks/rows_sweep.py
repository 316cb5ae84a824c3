#!/usr/bin/env python3
"""Step rate of bench.py's headline workload against the batch size: what one
GPU of N does in the strong-scaling leg (262144 streamlines in total ->
262144 / N per GPU).  First 12 steps from a reset, median of 7 windows.

    python benchmarks/rows_sweep.py [rows ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracktolearn_amd.utils.synthetic import synthetic_seeds  # noqa: E402


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [262144, 131072, 65536, 32768]
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 'c2')
    grp = bench.Dist(1, 'cpu')
    base = None
    for rows in rows_list:
        env.seeds = synthetic_seeds(subject[1].data, rows, seed=100)
        w = bench.timed_windows(env, rows, 12, 3, 7, 1, grp)
        if base is None:
            base = w['ms_per_step']
        print(json.dumps({'rows': rows, 'fuse_max_rows': os.environ.get('TTL_FUSE_MAX_ROWS', '16384'),
                          'ms_per_step': round(w['ms_per_step'], 5),
                          'streamline_steps_per_s_M': round(w['value'] / 1e6, 1),
                          'k_state_ms': round(w['state_ms'] / max(w['state_n'], 1), 5),
                          'step_time_ratio_vs_first': round(base / w['ms_per_step'], 2)}), flush=True)


if __name__ == '__main__':
    main()
