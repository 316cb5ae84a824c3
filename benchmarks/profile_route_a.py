#!/usr/bin/env python3
"""Where a step of the reference's own calling contract goes (route A of
INTEGRATION.md): `env.step(numpy actions)` -> host reward / dones ->
`env.harvest()`, BASELINE config 2's shape.  Prints host time per phase and the
overall rate next to the device-resident loop's.

    python benchmarks/profile_route_a.py [steps]
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
from benchmarks.placement_probe import make_bench_env  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    subject = bench.make_subject()
    env = make_bench_env(subject)
    N = bench.N_ACTOR
    for rep in range(3):
        state = env.reset(0, N)
        torch.cuda.synchronize()
        t = {'policy+d2h': 0.0, 'step': 0.0, 'harvest': 0.0}
        total, t0 = 0, time.perf_counter()
        for step in range(steps):
            total += env._n_active
            ta = time.perf_counter()
            a = env.scripted_actions(state, step, 1, bench.WOBBLE)
            a_host = a.to(device='cpu', copy=True).numpy()      # rl.py:93-94
            tb = time.perf_counter()
            env.step(a_host)
            tc = time.perf_counter()
            state, _ = env.harvest()
            td = time.perf_counter()
            t['policy+d2h'] += tb - ta
            t['step'] += tc - tb
            t['harvest'] += td - tc
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({'route': 'A: step(numpy)+harvest', 'steps': steps,
                      'streamline_steps_per_s': total / dt, 'ms_per_step': dt / steps * 1e3,
                      'host_ms_per_step': {k: v / steps * 1e3 for k, v in t.items()}}))


if __name__ == '__main__':
    main()
