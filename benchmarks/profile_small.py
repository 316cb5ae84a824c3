#!/usr/bin/env python3
"""Host-side breakdown of one small-batch step (BASELINE config 1's shape:
32^3, n_actor 4096, K = 100, noisy env): wall time per step of the
scripted-actions -> step_device -> harvest loop, per call, and a cProfile
listing of where the Python time goes."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(D=32, N=4096, K=100):
    from tracktolearn_amd.environments import NoisyTrackingEnvironment
    from tracktolearn_amd.utils.synthetic import (synthetic_seeds,
                                                  synthetic_subject)
    subject = synthetic_subject(D, 45, seed=1234, peaks=False, affine_dtype=np.float64)
    dto = dict(n_dirs=K, theta=30.0, npv=1, binary_stopping_threshold=0.1,
               step_size=0.75, min_length=20.0, max_length=300.0,
               compute_reward=False, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device('cuda:0'),
               target_sh_order=8, noise=0.0, fa_map=None)
    env = NoisyTrackingEnvironment(subject, 'testing', dto)
    env.seeds = synthetic_seeds(subject[1].data, N, seed=100)
    return env


def episode(env, N, timers=None):
    state = env.reset(0, N)
    step, units = 0, 0
    while env._n_active:
        units += env._n_active
        t0 = time.perf_counter()
        a = env.scripted_actions(state, step, 1, 0.05)
        t1 = time.perf_counter()
        env.step_device(a)
        t2 = time.perf_counter()
        state, _ = env.harvest()
        t3 = time.perf_counter()
        if timers is not None:
            timers[0] += t1 - t0
            timers[1] += t2 - t1
            timers[2] += t3 - t2
        step += 1
    return step, units


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    env = make(N=N)
    episode(env, N)
    torch.cuda.synchronize()
    timers = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    steps, units = 0, 0
    for _ in range(5):
        s, u = episode(env, N, timers)
        steps += s
        units += u
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'n_actor': N, 'steps': steps, 'us_per_step': dt / steps * 1e6,
                      'Msteps_per_s': units / dt / 1e6,
                      'us_scripted': timers[0] / steps * 1e6,
                      'us_step_device': timers[1] / steps * 1e6,
                      'us_harvest': timers[2] / steps * 1e6}), flush=True)
    # the same episodes on free-running steps: scripted policy + step launched for
    # the newest reported survivor count, the host never waiting for a step
    for graph in (False, True):
        policy = (lambda st: env.scripted_actions_free(st, 1, 0.05))
        for _ in range(2):
            state = env.reset(0, N)
            (env.run_free(policy, state, key='scripted') if graph
             else env.run_free_eager(policy, state))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fsteps = 0
        for _ in range(5):
            state = env.reset(0, N)
            out = env.run_free(policy, state, key='scripted') if graph \
                else env.run_free_eager(policy, state)
            fsteps += out[1]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({'n_actor': N, 'loop': 'free-running, graphed' if graph
                          else 'free-running, eager', 'steps': fsteps,
                          'us_per_step': dt / fsteps * 1e6}), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    episode(env, N)
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(18)
    print(out.getvalue())


if __name__ == '__main__':
    main()
