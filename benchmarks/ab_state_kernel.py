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
    fine: TTL_ORDER_KEY, ls: TTL_LOCAL_SORT, lay: TTL_SH_LAYOUT, st: TTL_STORE_FLAVOUR,
    cv: TTL_CONTIGUOUS_VOLUME, vc: TTL_VOLUME_CANDIDATES)."""
    cfg = {'k': '4', 'r': '16', 'fine': '0', 'ls': '1', 'lay': 'brick4', 'st': '0', 'd': '96', 'cv': '1', 'vc': '4'}
    for part in str(variant).split(','):
        if part:
            key, val = part.split('=')
            cfg[key] = val
    return cfg


_SUBJECTS = {}


def make(variant, subject):
    cfg = parse(variant)
    d = int(cfg['d'])
    if d != 96:          # another volume size (cache-residency experiments)
        if d not in _SUBJECTS:
            from tracktolearn_amd.utils.synthetic import synthetic_subject
            _SUBJECTS[d] = synthetic_subject(d, bench.C, seed=1234, peaks=False,
                                             affine_dtype=np.float32)
        subject = _SUBJECTS[d]
    os.environ['TTL_STATE_KERNEL'] = cfg['k']
    os.environ['TTL_LOCAL_SORT'] = cfg['ls']
    os.environ['TTL_ORDER_KEY'] = cfg['fine']
    os.environ['TTL_SH_LAYOUT'] = cfg['lay']
    os.environ['TTL_STORE_FLAVOUR'] = cfg['st']
    os.environ['TTL_CONTIGUOUS_VOLUME'] = cfg['cv']
    os.environ['TTL_VOLUME_CANDIDATES'] = cfg['vc']
    from tracktolearn_amd.utils.synthetic import synthetic_seeds
    env = bench.make_env(subject, 'cuda:0', 'c2')
    env.seeds = synthetic_seeds(subject[1].data, bench.N_ACTOR, seed=100)
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
    n = bench.run_steps(env, steps, 1, counter, bench.N_ACTOR)
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


def main():
    variants = sys.argv[1:] or ['lay=brick4', 'lay=linear']
    copies = int(os.environ.get('AB_COPIES', '2'))     # env instances per variant
    subject = bench.make_subject()
    envs = {}
    ref_states = {}
    for v in variants:
        envs[v] = []
        for c in range(copies):
            env, state = make(v, subject)
            # correctness: first 3 steps' rows against the first variant of the
            # same volume size
            rows = [state.contiguous().clone()]
            for step in range(3):
                a = env.scripted_actions(state, step, 1, bench.WOBBLE)
                ns, _, _, info = env.step_device(a)
                rows.append(ns[info['row_dest'].long()].contiguous().clone())
                state, _ = env.harvest()
            d = parse(v)['d']
            if d not in ref_states:
                ref_states[d] = rows
                diff = 0.0
            else:
                diff = max(float((a - b).abs().max()) for a, b in zip(rows, ref_states[d]))
            envs[v].append((env, diff))
            for _ in range(2):
                window(env)
    results = {v: {'ms': [], 'rate': []} for v in variants}
    for rnd in range(5):
        for v in variants:
            for env, _ in envs[v]:
                env.profile_begin(64, classes=('state', 'advance'))
                n, dt = window(env)
                prof = env.profile_end()
                ms, cnt = prof['state']
                results[v].setdefault('adv', []).append(prof['advance'][0] / max(prof['advance'][1], 1))
                results[v]['ms'].append(ms / max(cnt, 1))
                results[v]['rate'].append(n / dt)
    for v in variants:
        r = results[v]
        per_copy = [float(np.median(r['ms'][c::copies])) for c in range(copies)]
        print(json.dumps({'variant': v, 'k_state_ms_median': float(np.median(r['ms'])),
                          'k_state_ms_per_instance': per_copy,
                          'k_advance_ms_median': float(np.median(r['adv'])),
                          'Msteps_per_s_median': float(np.median(r['rate'])) / 1e6,
                          'volume_candidates_ms': [e._sh_tuned for e, _ in envs[v]],
                          'max_abs_diff_vs_first': max(d for _, d in envs[v])}), flush=True)


if __name__ == '__main__':
    main()
