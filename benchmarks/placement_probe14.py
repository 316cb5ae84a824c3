#!/usr/bin/env python3
"""Along a sequence of allocations, which ones are fast?  Two volume copies
first, then 28 ring allocations (1.37 GB each, held) timed with both volumes;
then 24 volume allocations (170 MB each, held) timed with the first and the
last ring.  Looks for a period in allocation position.

    python benchmarks/placement_probe14.py
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
    W, P, N = env._state_width, env._state_pitch, bench.N_ACTOR
    keep = []

    def new_volume():
        mem = _lib.DeviceVolume(0, nbytes, 0)
        vol = torch.as_tensor(mem, device='cuda:0').view(torch.float32).view(own.shape)
        vol.copy_(own)
        keep.append(mem)
        return vol

    def new_ring():
        mem = _lib.DeviceVolume(0, 4 * N * P * 4, 0)
        flat = torch.as_tensor(mem, device='cuda:0').view(torch.float32)
        keep.append(mem)
        return [flat[i * N * P:(i + 1) * N * P].view(N, P)[:, :W] for i in range(4)]

    v = [new_volume(), new_volume()]
    rings = []
    rows = [[], []]
    for k in range(28):
        rings.append(new_ring())
    for vi in range(2):
        env._sh_packed = v[vi]
        rehandle(env)
        for ring in rings:
            env._state_ring, env._state_ring_pos = ring, 0
            env.reset(0, N)
            window(env)
            rows[vi].append(round(timed(env, rounds=1), 4))
        print(json.dumps(dict(volume=vi, gather_ms_by_ring_in_allocation_order=rows[vi])), flush=True)
    vols = [new_volume() for _ in range(24)]
    for ri in (0, 27):
        env._state_ring = rings[ri]
        row = []
        for vol in vols:
            env._sh_packed = vol
            rehandle(env)
            row.append(round(timed(env, rounds=1), 4))
        print(json.dumps(dict(ring=ri, gather_ms_by_volume_in_allocation_order=row)), flush=True)


if __name__ == '__main__':
    main()
