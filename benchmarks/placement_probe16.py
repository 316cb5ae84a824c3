#!/usr/bin/env python3
"""K = 100: does the placement of the streamline history (1.2 KB read per
streamline and step for the direction block) matter as the volume's and the
rows' do?  Six fresh allocations for the history, the gather's and the advance
kernel's time on each (volume and ring as tuned).

    python benchmarks/placement_probe16.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchmarks.bench_configs import CONFIGS  # noqa: E402
from tracktolearn_amd import _lib  # noqa: E402


def main():
    from tracktolearn_amd.environments import TrackingEnvironment
    from tracktolearn_amd.utils.synthetic import synthetic_seeds, synthetic_subject
    cfg = CONFIGS['c2-K100']
    D, N, K = cfg['D'], cfg['N'], cfg['K']
    subject = synthetic_subject(D, 45, seed=1234, peaks=False, affine_dtype=np.float32)
    dto = dict(n_dirs=K, theta=30.0, npv=1, binary_stopping_threshold=0.1,
               step_size=0.75, min_length=20.0, max_length=cfg['max_length'],
               compute_reward=False, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device('cuda:0'),
               target_sh_order=8, noise=0.0, fa_map=None)
    env = TrackingEnvironment(subject, 'testing', dto)
    env.seeds = synthetic_seeds(subject[1].data, N, seed=100)

    def measure():
        state = env.reset(0, N)
        env.profile_begin(64, classes=('state', 'advance'))
        for step in range(12):
            env.step_device(env.scripted_actions(state, step, 1, 0.05))
            state, _ = env.harvest()
        prof = env.profile_end()
        return {k: round(prof[k][0] / max(prof[k][1], 1), 4) for k in ('state', 'advance')}

    measure()
    print(json.dumps(dict(tuned=measure(), placement_candidates=env._sh_tuned)), flush=True)
    hist_shape = tuple(env._buf_streamlines.shape)
    real_empty = torch.empty
    keep = []
    for k in range(6):
        mem = _lib.DeviceVolume(0, int(np.prod(hist_shape)) * 4, 0)
        keep.append(mem)
        hist = torch.as_tensor(mem, device='cuda:0').view(torch.float32).view(hist_shape)

        def fake_empty(*args, **kw):
            shape = args[0] if len(args) == 1 and isinstance(args[0], (tuple, list)) else args
            if tuple(shape) == hist_shape and kw.get('dtype') is torch.float32:
                return hist
            return real_empty(*args, **kw)

        env._destroy_handle()
        env._n_max = 0
        torch.empty = fake_empty
        try:
            env._ensure_capacity(N)
        finally:
            torch.empty = real_empty
        measure()
        print(json.dumps(dict(history_allocation=k, **measure())), flush=True)


if __name__ == '__main__':
    main()
