#!/usr/bin/env python3
"""Is the first device allocation of a process a fast home for the state ring
(and the volume)?  Grabs an arena before anything else touches the GPU, builds
the env, then times the gather with the ring / the volume / both carved from
that arena, against fresh late allocations.

    python benchmarks/placement_probe15.py [arena GiB]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['TTL_VOLUME_CANDIDATES'] = '1'
from tracktolearn_amd import _lib  # noqa: E402


def main():
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.cuda.init()
    early = _lib.DeviceVolume(0, gib << 30, 0)          # before any other allocation
    import bench
    from benchmarks.ab_state_kernel import window
    from benchmarks.placement_probe import rehandle, timed
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    env.reset(0, bench.N_ACTOR)
    window(env)
    own = env._sh_packed
    nbytes = own.numel() * 4
    W, P, N = env._state_width, env._state_pitch, bench.N_ACTOR
    print(json.dumps(dict(own_volume_allocator_rows_ms=round(timed(env), 4))), flush=True)

    def carve(mem, offset):
        arena = torch.as_tensor(mem, device='cuda:0')
        flat = arena[offset:offset + 4 * N * P * 4].view(torch.float32)
        return [flat[i * N * P:(i + 1) * N * P].view(N, P)[:, :W] for i in range(4)]

    ring_early = carve(early, 0)
    vol_early = torch.as_tensor(early, device='cuda:0')[2 << 30:(2 << 30) + nbytes] \
        .view(torch.float32).view(own.shape)
    vol_early.copy_(own)
    late = [_lib.DeviceVolume(0, 4 * N * P * 4, 0) for _ in range(3)]
    cases = [('ring in the early arena, own volume', own, ring_early),
             ('ring and volume in the early arena', vol_early, ring_early),
             ('volume in the early arena, late ring 0', vol_early, carve(late[0], 0)),
             ('own volume, late ring 0', own, carve(late[0], 0)),
             ('own volume, late ring 1', own, carve(late[1], 0)),
             ('own volume, late ring 2', own, carve(late[2], 0))]
    for name, vol, ring in cases:
        env._sh_packed, env._state_ring, env._state_ring_pos = vol, ring, 0
        rehandle(env)
        print(json.dumps(dict(case=name, gather_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
