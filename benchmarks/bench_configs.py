#!/usr/bin/env python3
"""Env-step throughput of the other BASELINE.json shapes (auxiliary to
bench.py, same loop: scripted actions -> step_device -> harvest, first 12
steps after 3 warm-up steps):
  c2-K100  96^3, 262144 streamlines, K = 100 (the shipped model's state width)
  c3-env   96^3, 65536, K = 4, reward on (the env side of config 3)
  c4-shard 145^3 (> Infinity Cache), 131072, K = 100, noisy env (float64
           directions): one GPU's shard of config 4
  c1       32^3, 4096, K = 100, noisy env (config 1's shape)
  c2-host  config 2 through the reference's own calling contract: the action
           batch crosses PCIe as a host numpy array, `step()` returns host
           reward / dones, `harvest()` copies the survivors' rows
           (the PCIe-inclusive rate)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    'c2-K100': dict(D=96, N=262144, K=100, noisy=False, reward=False, max_length=200.0),
    'c3-env': dict(D=96, N=65536, K=4, noisy=False, reward=True, max_length=200.0),
    'c4-shard': dict(D=145, N=131072, K=100, noisy=True, reward=False, max_length=300.0),
    'c1': dict(D=32, N=4096, K=100, noisy=True, reward=False, max_length=300.0),
    'c2-host': dict(D=96, N=262144, K=4, noisy=False, reward=False,
                    max_length=200.0, host_contract=True),
}


def run(name, D, N, K, noisy, reward, max_length, steps=12, warmup=3,
        host_contract=False):
    from tracktolearn_amd.environments import (NoisyTrackingEnvironment,
                                               TrackingEnvironment)
    from tracktolearn_amd.utils.synthetic import (synthetic_seeds,
                                                  synthetic_subject)
    subject = synthetic_subject(D, 45, seed=1234, peaks=reward,
                                affine_dtype=np.float64 if noisy else np.float32)
    dto = dict(n_dirs=K, theta=30.0, npv=1, binary_stopping_threshold=0.1,
               step_size=0.75, min_length=20.0, max_length=max_length,
               compute_reward=reward, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device('cuda:0'),
               target_sh_order=8, noise=0.0, fa_map=None)
    env = (NoisyTrackingEnvironment if noisy else TrackingEnvironment)(
        subject, 'testing', dto)
    env.seeds = synthetic_seeds(subject[1].data, N, seed=100)

    def loop(n_steps):
        state = env.reset(0, N)
        total = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(n_steps):
            if env._n_active == 0:
                break
            total += env._n_active
            a = env.scripted_actions(state, step, 1, 0.05)
            if host_contract:
                # rl.py:93-94: action.to('cpu').numpy() -> env.step(numpy)
                env.step(a.to(device='cpu', copy=True).numpy())
            else:
                env.step_device(a)
            state, _ = env.harvest()
        torch.cuda.synchronize()
        return total, time.perf_counter() - t0
    loop(warmup)
    env.profile_begin(64, classes=('state',))
    total, dt = loop(steps)
    prof = env.profile_end()
    W = 7 * 45 + 3 * K
    kern_b = 4 * 56 * 45 + 12 * (K + 1) + 4 * W
    ms, n = prof['state']
    print(json.dumps({
        'config': name, 'volume': [D, D, D, 45], 'n_actor': N, 'n_dirs': K,
        'mode': 'f64dir' if noisy else 'f32', 'reward': reward,
        'loop': 'step(numpy)+harvest (host contract)' if host_contract else 'step_device+harvest',
        'streamline_steps_per_s': total / dt, 'ms_per_step': dt / steps * 1e3,
        'k_state_ms': ms / max(n, 1),
        'k_state_algorithmic_GBs': kern_b * (total / max(n, 1)) / (ms / max(n, 1) * 1e-3) / 1e9,
    }), flush=True)


if __name__ == '__main__':
    import os
    extra = {}
    if os.environ.get('BENCH_STEPS'):        # e.g. a window deeper into the episode
        extra['steps'] = int(os.environ['BENCH_STEPS'])
    for name in (sys.argv[1:] or CONFIGS):
        run(name, **CONFIGS[name], **extra)
