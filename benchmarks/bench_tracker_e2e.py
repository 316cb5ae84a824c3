#!/usr/bin/env python3
"""End-to-end Tracker.track on a synthetic subject (what ttl_track.py runs
between loading the volumes and writing the file): seeds -> batches ->
validation_episode -> length filter -> file space -> TractogramItems.  Wall
time split into the tracking episodes and everything after them.

    python benchmarks/bench_tracker_e2e.py [n_seeds] [n_actor] [hidden] [D] [ext]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.bench_tracking_loop import make  # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
    n_actor = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    hidden = sys.argv[3] if len(sys.argv) > 3 else '1024-1024-1024'
    D = int(sys.argv[4]) if len(sys.argv) > 4 else 48
    ext = sys.argv[5] if len(sys.argv) > 5 else '.trk'
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    from tracktolearn_amd.tracking.tracker import Tracker, detect_format
    torch.manual_seed(0)
    env = make(D, n_seeds)
    alg = SACAuto(env.get_state_size(), 3, hidden, n_actors=n_actor, rng=None,
                  device=torch.device('cuda:0'))
    tracker = Tracker(alg, n_actor, prob=0.0, compress=0.0, min_length=5.0, max_length=200.0,
                      save_seeds=False)
    episode_s = [0.0]
    orig = alg.validation_episode

    def timed_episode(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = orig(*a, **k)
        torch.cuda.synchronize()
        episode_s[0] += time.perf_counter() - t0
        return out

    alg.validation_episode = timed_episode
    out = {}
    for rep in range(2):            # the second pass is reported
        episode_s[0] = 0.0
        np.random.seed(0)
        t0 = time.perf_counter()
        tractogram = tracker.track(env, detect_format('x' + ext))
        n_items, n_points = 0, 0
        for item in tractogram:
            n_items += 1
            n_points += len(item.streamline)
        total = time.perf_counter() - t0
        out = dict(workload=f'{D}^3, {n_seeds} seeds, n_actor {n_actor}, SAC {hidden}, {ext}',
                   streamlines_kept=n_items, points=n_points, total_ms=round(total * 1e3, 1),
                   episodes_ms=round(episode_s[0] * 1e3, 1),
                   after_episodes_ms=round((total - episode_s[0]) * 1e3, 1))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
