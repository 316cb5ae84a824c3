#!/usr/bin/env python3
"""Is the placement effect of the SH volume independent of that of the state
rows?  Five volume allocations x five ring allocations (fresh hipMalloc each),
the gather's time for every pair.

    python benchmarks/placement_probe13.py [0 | 1 | 2]      (hipMalloc, contiguous, virtual-memory API)
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
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0     # 0 hipMalloc, 1 contiguous, 2 virtual-memory API
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    env.reset(0, bench.N_ACTOR)
    window(env)
    own = env._sh_packed
    nbytes = own.numel() * 4
    W, P, N = env._state_width, env._state_pitch, bench.N_ACTOR
    vols, rings, keep = [], [], []
    for k in range(5):
        mem = _lib.DeviceVolume(0, nbytes, kind)
        vol = torch.as_tensor(mem, device='cuda:0').view(torch.float32).view(own.shape)
        vol.copy_(own)
        vols.append(vol)
        keep.append(mem)
        mem = _lib.DeviceVolume(0, 4 * N * P * 4, kind)
        flat = torch.as_tensor(mem, device='cuda:0').view(torch.float32)
        rings.append([flat[i * N * P:(i + 1) * N * P].view(N, P)[:, :W] for i in range(4)])
        keep.append(mem)
    for vi, vol in enumerate(vols):
        row = []
        env._sh_packed = vol
        for ring in rings + [None]:
            env._state_ring, env._state_ring_pos = ring, 0
            rehandle(env)
            row.append(round(timed(env, rounds=2), 4))
        print(json.dumps(dict(volume=vi, ptr=hex(vol.data_ptr()), gather_ms_by_ring=row[:-1],
                              allocator_rows_ms=row[-1])), flush=True)
    print(json.dumps(dict(kind=kind, granted=[m.contiguous for m in keep[:2]],
                          ring_ptrs=[hex(r[0].data_ptr()) for r in rings])), flush=True)


if __name__ == '__main__':
    main()
