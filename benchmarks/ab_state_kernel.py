#!/usr/bin/env python3
"""Same-box A/B of the state-gather kernel variants (TTL_STATE_KERNEL) on
bench.py's workload: per variant, the average launch time of the gather over
the first 12 steps (HIP events on the launch stream) and the step rate, in
interleaved rounds; plus the largest state difference against variant 4 (the
round-1 kernel) on the first steps.

    python benchmarks/ab_state_kernel.py [variant ...]        (default: a set)
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


def parse(variant):
    """'k=4,r=16,fine=1,ls=1' -> dict (k: TTL_STATE_KERNEL, r: refresh period,
    fine: TTL_ORDER_KEY, ls: TTL_LOCAL_SORT, lay: TTL_SH_LAYOUT)."""
    cfg = {'k': '4', 'r': '16', 'fine': '2', 'ls': '1', 'lay': 'brick4', 'st': '0'}
    for part in str(variant).split(','):
        if part:
            key, val = part.split('=')
            cfg[key] = val
    return cfg


def make(variant, subject):
    cfg = parse(variant)
    os.environ['TTL_STATE_KERNEL'] = cfg['k']
    os.environ['TTL_LOCAL_SORT'] = cfg['ls']
    os.environ['TTL_ORDER_KEY'] = cfg['fine']
    os.environ['TTL_SH_LAYOUT'] = cfg['lay']
    os.environ['TTL_STORE_FLAVOUR'] = cfg['st']
    env = bench.make_env(subject, 'cuda:0', 0)
    env.SPATIAL_ORDER_REFRESH = int(cfg['r'])
    env._fine = cfg['fine']
    env.reset(0, 64)                     # creates the handle (reads the variables)
    env._destroy_handle()
    env._n_max = 0
    state = env.reset(0, bench.N_ACTOR)
    return env, state


def window(env, steps=12):
    os.environ['TTL_ORDER_KEY'] = getattr(env, '_fine', '0')
    counter = {'state': env.reset(0, bench.N_ACTOR), 'step': 0, 'resets': 0}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = bench.run_steps(env, steps, 1, counter)
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


def main():
    variants = sys.argv[1:] or ['fine=0', 'fine=2', 'fine=3', 'fine=0,lay=linear', 'fine=2,lay=linear', 'fine=2,r=8']
    subject = bench.make_subject()
    envs = {}
    ref_states = None
    for v in variants:
        env, state = make(v, subject)
        # correctness: first 3 steps' rows against variant 4
        rows = [state.clone()]
        for step in range(3):
            a = env.scripted_actions(state, step, 1, bench.WOBBLE)
            ns, _, _, info = env.step_device(a)
            full = torch.empty_like(ns)
            full[:] = ns
            rows.append(full[info['row_dest'].long()].clone())
            state, _ = env.harvest()
        if ref_states is None:
            ref_states = rows
            diff = 0.0
        else:
            diff = max(float((a - b).abs().max()) for a, b in zip(rows, ref_states))
        envs[v] = (env, diff)
        for _ in range(2):
            window(env)
    results = {v: {'ms': [], 'rate': []} for v in variants}
    for rnd in range(6):
        for v in variants:
            env, _ = envs[v]
            env.profile_begin(64, classes=('state',))
            n, dt = window(env)
            ms, cnt = env.profile_end()['state']
            results[v]['ms'].append(ms / max(cnt, 1))
            results[v]['rate'].append(n / dt)
    for v in variants:
        r = results[v]
        print(json.dumps({'variant': v, 'k_state_ms_median': float(np.median(r['ms'])),
                          'k_state_ms_min': float(np.min(r['ms'])),
                          'Msteps_per_s_median': float(np.median(r['rate'])) / 1e6,
                          'max_abs_diff_vs_first': envs[v][1]}), flush=True)


if __name__ == '__main__':
    main()
