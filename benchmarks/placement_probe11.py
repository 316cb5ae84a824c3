#!/usr/bin/env python3
"""Follow-up of placement_probe10.py (every offset of an 8 GiB allocation is a
fast placement of the SH volume): how large does the allocation have to be?
For arena sizes from 256 MiB to 16 GiB, three fresh allocations each: the
gather's time with the volume at the start of the arena.

    python benchmarks/placement_probe11.py
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
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    env.reset(0, bench.N_ACTOR)
    window(env)
    own = env._sh_packed
    nbytes = own.numel() * 4
    print(json.dumps(dict(volume_MiB=nbytes >> 20, own_allocation_ms=round(timed(env), 4))), flush=True)
    for mib in (256, 512, 1024, 2048, 4096, 16384):
        row = []
        for rep in range(3):
            mem = _lib.DeviceVolume(0, mib << 20, False)
            arena = torch.as_tensor(mem, device='cuda:0')
            vol = arena[:nbytes].view(torch.float32).view(own.shape)
            vol.copy_(own)
            env._sh_packed = vol
            rehandle(env)
            row.append(round(timed(env, rounds=2), 4))
            env._sh_packed = own
            del vol, arena, mem
        print(json.dumps(dict(arena_MiB=mib, gather_ms=row)), flush=True)
    rehandle(env)
    print(json.dumps(dict(own_allocation_again_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
