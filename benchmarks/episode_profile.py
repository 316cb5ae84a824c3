#!/usr/bin/env python3
"""Per-step profile of one whole episode of bench.py's workload: active rows
and GPU time per step (HIP events around scripted actions + step + harvest; the
host is not synchronised inside the episode), so the decay of the rate over the
episode can be attributed (order decay, launch-bound tail, refresh cost)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchmarks.placement_probe import make_bench_env  # noqa: E402


def main():
    subject = bench.make_subject()
    env = make_bench_env(subject)
    for rep in range(2):
        state = env.reset(0, bench.N_ACTOR)
        torch.cuda.synchronize()
        evs, ns = [torch.cuda.Event(enable_timing=True)], []
        evs[0].record()
        step = 0
        while env._n_active:
            ns.append(env._n_active)
            a = env.scripted_actions(state, step, 1, bench.WOBBLE)
            env.step_device(a)
            state, _ = env.harvest()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
            step += 1
        torch.cuda.synchronize()
    ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(len(ns))])
    ns = np.array(ns)
    print(json.dumps({'steps': len(ns), 'total_ms': float(ms.sum()),
                      'Msteps_per_s': float(ns.sum() / ms.sum() / 1e3)}))
    print('step n_active ms ns_per_row')
    for i in range(len(ns)):
        if i < 20 or i % 8 == 0 or ns[i] < 20000:
            print(i, int(ns[i]), round(float(ms[i]), 4), round(float(ms[i] * 1e6 / ns[i]), 3))
    # buckets
    for lo, hi in ((131072, 1 << 30), (65536, 131072), (16384, 65536), (4096, 16384), (0, 4096)):
        sel = (ns >= lo) & (ns < hi)
        if sel.any():
            print(f'rows in [{lo}, {hi}): {int(sel.sum())} steps, {ms[sel].sum():.3f} ms, '
                  f'{ns[sel].sum() / ms[sel].sum() / 1e3:.1f} M steps/s')


if __name__ == '__main__':
    main()
