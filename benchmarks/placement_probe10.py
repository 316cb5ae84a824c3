#!/usr/bin/env python3
"""Map of the placement effect inside ONE large allocation: the packed SH volume
copied to offsets k * 256 MiB of an arena (default 8 GiB), the step's gather
timed at each.  Are fast and slow placements regions of physical memory?

    python benchmarks/placement_probe10.py [arena GiB]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['TTL_VOLUME_CANDIDATES'] = '1'
import bench  # noqa: E402
from benchmarks.ab_state_kernel import window  # noqa: E402
from benchmarks.placement_probe import rehandle, timed  # noqa: E402
from tracktolearn_amd import _lib  # noqa: E402


def main():
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    env.reset(0, bench.N_ACTOR)
    window(env)
    own = env._sh_packed
    print(json.dumps(dict(own_allocation_ms=round(timed(env), 4))), flush=True)
    mem = _lib.DeviceVolume(0, gib << 30, False)
    arena = torch.as_tensor(mem, device='cuda:0')
    nbytes = own.numel() * 4
    step = 256 << 20
    row = []
    for k in range((gib << 30) // step):
        if k * step + nbytes > (gib << 30):
            break
        vol = arena[k * step:k * step + nbytes].view(torch.float32).view(own.shape)
        vol.copy_(own)
        env._sh_packed = vol
        rehandle(env)
        row.append(round(timed(env, rounds=2), 4))
    print(json.dumps(dict(arena=hex(mem.ptr), step_MiB=256, gather_ms_by_offset=row)), flush=True)
    env._sh_packed = own
    rehandle(env)
    print(json.dumps(dict(own_allocation_again_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
