#!/usr/bin/env python3
"""Tracking loop of BASELINE config 1's shape (32^3 volume, K = 100, noisy env
without noise, SAC policy `hidden`) at a given n_actor: wall time per step and
streamline-steps/s of RLAlgorithm.validation_episode, step by step
(TTL_GRAPH_EPISODE=0: policy -> step_device -> harvest, one survivor count
fetched per step), free-running (policy + free-running step launched for the
newest reported survivor count, the host never waiting for a step) and graphed
(policy + free-running step captured in one HIP graph over the whole batch,
replayed until the pinned survivor count reads zero).

    python benchmarks/bench_tracking_loop.py [n_actor] [hidden] [D]
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(D, N, K=100):
    from tracktolearn_amd.environments import NoisyTrackingEnvironment
    from tracktolearn_amd.utils.synthetic import (synthetic_seeds,
                                                  synthetic_subject)
    subject = synthetic_subject(D, 45, seed=1234, peaks=False, affine_dtype=np.float64)
    dto = dict(n_dirs=K, theta=30.0, npv=1, binary_stopping_threshold=0.1,
               step_size=0.75, min_length=20.0, max_length=200.0,
               compute_reward=False, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device('cuda:0'),
               target_sh_order=8, noise=0.0, fa_map=None)
    env = NoisyTrackingEnvironment(subject, 'testing', dto)
    env.seeds = synthetic_seeds(subject[1].data, N, seed=100)
    return env


def run(env, alg, N, reps):
    best = None
    for _ in range(reps):
        state = env.reset(0, N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        alg.validation_episode(state, env, 0.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = env.length - 1
        units = int(env._buf_lengths[:N].sum().item()) - N   # points added = streamline-steps
        if best is None or dt < best[0]:
            best = (dt, steps, units)
    return best


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    hidden = sys.argv[2] if len(sys.argv) > 2 else '1024-1024-1024'
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    torch.manual_seed(0)
    env = make(D, N)
    alg = SACAuto(env.get_state_size(), 3, hidden, n_actors=N, rng=None,
                  device=torch.device('cuda:0'))
    alg.agent.eval()
    out = dict(workload=f'{D}^3 x 45 SH, n_actor {N}, K 100, SAC {hidden}, prob 0')
    for name, flag, limit in (('step_by_step', '0', 0.0), ('free_running', '1', 0.0),
                              ('graphed', '1', 1e9)):
        os.environ['TTL_GRAPH_EPISODE'] = flag
        os.environ['TTL_FREE_RUNNING_EAGER'] = '1'
        type(alg).graph_policy_us = limit     # 0: never the graph, 1e9: always
        run(env, alg, N, 2)                      # warm-up (and graph capture)
        dt, steps, units = run(env, alg, N, 5)
        out[name] = dict(ms=round(dt * 1e3, 3), steps=steps,
                         us_per_step=round(dt / steps * 1e6, 2),
                         streamline_steps=units,
                         M_streamline_steps_per_s=round(units / dt / 1e6, 2))
    out['policy_us_full_batch'] = round(max(fr.policy_us for fr in env._free_runs.values()), 1)
    out['speedup_free_running'] = round(out['step_by_step']['ms'] / out['free_running']['ms'], 2)
    out['speedup_graphed'] = round(out['step_by_step']['ms'] / out['graphed']['ms'], 2)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
