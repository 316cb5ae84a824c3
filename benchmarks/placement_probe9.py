#!/usr/bin/env python3
"""Does it matter WHICH allocation the state rows live in (as it does for the
SH volume)?  One env (volume placement as tuned); env._new_state alternates
between the two halves of an arena; eight arenas from separate hipMalloc calls,
the gather's time on each.

    python benchmarks/placement_probe9.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchmarks.ab_state_kernel import window  # noqa: E402
from benchmarks.placement_probe import timed  # noqa: E402
from tracktolearn_amd import _lib  # noqa: E402


def main():
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    env.reset(0, bench.N_ACTOR)
    env._state_ring = None          # this script places the state rows itself
    window(env)
    print(json.dumps(dict(allocator_ms=round(timed(env), 4), volume_candidates_ms=env._sh_tuned)),
          flush=True)
    W = env._state_width
    rows = bench.N_ACTOR * W * 4
    orig = env._new_state
    keep = []
    for k in range(8):
        mem = _lib.DeviceVolume(0, 2 * rows, False)
        keep.append(mem)
        arena = torch.as_tensor(mem, device='cuda:0').view(torch.float32)
        halves = [arena[:rows // 4].view(bench.N_ACTOR, W), arena[rows // 4:].view(bench.N_ACTOR, W)]
        flip = [0]

        def new_state(n, halves=halves, flip=flip):
            flip[0] += 1
            return halves[flip[0] & 1][:n]

        env._new_state = new_state
        window(env)
        print(json.dumps(dict(arena=k, ptr=hex(mem.ptr), gather_ms=round(timed(env), 4))), flush=True)
    env._new_state = orig
    print(json.dumps(dict(allocator_again_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
