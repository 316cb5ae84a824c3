#!/usr/bin/env python3
"""Does the placement of the state rows (the gather's output) change its time?
One env; env._new_state is patched to hand out views of one big arena at chosen
byte offsets (two alternating buffers, as the caching allocator gives a step
loop); prints the gather's launch time per placement.

    python benchmarks/placement_probe2.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchmarks.ab_state_kernel import window  # noqa: E402


def timed(env, rounds=3):
    ms = []
    for _ in range(rounds):
        env.profile_begin(64, classes=('state',))
        window(env)
        t, c = env.profile_end()['state']
        ms.append(t / max(c, 1))
    return float(np.median(ms))


def main():
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    env.reset(0, bench.N_ACTOR)
    env._state_ring = None          # this script places the state rows itself
    window(env)
    print(json.dumps(dict(default_ms=round(timed(env), 4))), flush=True)
    W = env._state_width
    rows_bytes = bench.N_ACTOR * W * 4
    arena = torch.empty(4 * rows_bytes + (64 << 20), dtype=torch.uint8, device='cuda:0')
    base = arena.data_ptr()
    align = (-base) % (2 << 20)            # start of a 2 MiB-aligned region inside the arena
    flip = [0]

    def patched(offsets):
        def new_state(n):
            off = offsets[flip[0] & 1]
            flip[0] += 1
            v = arena[align + off: align + off + n * W * 4].view(torch.float32).view(n, W)
            return v
        return new_state

    orig = env._new_state
    gap = rows_bytes + ((-rows_bytes) % (2 << 20))      # second buffer: next 2 MiB boundary
    cases = [('2MiB aligned, 2MiB-aligned gap', (0, gap)),
             ('+4 B', (4, gap + 4)), ('+64 B', (64, gap + 64)), ('+256 B', (256, gap + 256)),
             ('+4 KiB', (4096, gap + 4096)), ('+64 KiB', (65536, gap + 65536)),
             ('+1 MiB', (1 << 20, gap + (1 << 20))),
             ('second buffer +1 MiB only', (0, gap + (1 << 20))),
             ('second buffer directly behind the first', (0, rows_bytes)),
             ('same buffer for both', (0, 0))]
    for name, offs in cases:
        env._new_state = patched(offs)
        window(env)
        print(json.dumps(dict(case=name, ms=round(timed(env), 4))), flush=True)
    env._new_state = orig
    print(json.dumps(dict(default_again_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
