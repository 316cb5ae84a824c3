#!/usr/bin/env python3
"""Experiment: the 262 144-streamline batch as TWO half-batches software-pipelined
on two HIP streams (the latency-bound small kernels of one half under the state
gather of the other) against the one-batch loop.  Same streamlines, same
results; only the schedule differs.

    python benchmarks/two_halves.py [parts]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tracktolearn_amd.utils.synthetic import synthetic_seeds  # noqa: E402


def window(envs, streams, rows, steps=12):
    states = []
    for env, s in zip(envs, streams):
        with torch.cuda.stream(s):
            states.append(env.reset(0, rows))
    torch.cuda.synchronize()
    total, t0 = 0, time.perf_counter()
    for step in range(steps):
        for k, (env, s) in enumerate(zip(envs, streams)):
            with torch.cuda.stream(s):
                total += env._n_active
                env.step_device(env.scripted_actions(states[k], step, 1, bench.WOBBLE))
        for k, (env, s) in enumerate(zip(envs, streams)):
            with torch.cuda.stream(s):
                states[k], _ = env.harvest()
    torch.cuda.synchronize()
    return total, time.perf_counter() - t0


def main():
    parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    subject = bench.make_subject()
    seeds = synthetic_seeds(subject[1].data, bench.N_ACTOR, seed=100)
    out = {}
    for p in (1, parts):
        rows = bench.N_ACTOR // p
        envs, streams = [], []
        for k in range(p):
            env = bench.make_env(subject, 'cuda:0', 'c2')
            env.seeds = seeds[k * rows:(k + 1) * rows]
            envs.append(env)
            streams.append(torch.cuda.Stream() if p > 1 else torch.cuda.current_stream())
        for _ in range(3):
            window(envs, streams, rows)
        rates = []
        for _ in range(7):
            n, dt = window(envs, streams, rows)
            rates.append(n / dt)
        out[p] = float(np.median(rates))
        print(json.dumps({'parts': p, 'rows_each': rows, 'Msteps_per_s_median': out[p] / 1e6,
                          'ms_per_step': bench.N_ACTOR * 0.94 / out[p] * 1e3}), flush=True)
        del envs
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
