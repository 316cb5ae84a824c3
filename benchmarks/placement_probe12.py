#!/usr/bin/env python3
"""Volume AND state rows carved from one large allocation (default 8 GiB),
against the env's own tuned placements, same process: the gather's time, three
fresh arenas.

    python benchmarks/placement_probe12.py [arena GiB]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
    print(json.dumps(dict(tuned_ms=round(timed(env), 4), placement_candidates=env._sh_tuned)), flush=True)
    own, own_ring = env._sh_packed, env._state_ring
    nbytes = own.numel() * 4
    W, P, N = env._state_width, env._state_pitch, bench.N_ACTOR
    for rep in range(3):
        mem = _lib.DeviceVolume(0, gib << 30, False)
        arena = torch.as_tensor(mem, device='cuda:0')
        vol = arena[:nbytes].view(torch.float32).view(own.shape)
        vol.copy_(own)
        base = (nbytes + (1 << 21) - 1) >> 21 << 21
        flat = arena[base:base + 4 * N * P * 4].view(torch.float32)
        ring = [flat[i * N * P:(i + 1) * N * P].view(N, P)[:, :W] for i in range(4)]
        env._sh_packed, env._state_ring, env._state_ring_pos = vol, ring, 0
        rehandle(env)
        both = timed(env)
        env._state_ring = own_ring
        rehandle(env)
        vol_only = timed(env)
        env._sh_packed, env._state_ring = own, ring
        rehandle(env)
        ring_only = timed(env)
        print(json.dumps(dict(arena_GiB=gib, volume_and_ring_in_arena_ms=round(both, 4),
                              volume_only_ms=round(vol_only, 4),
                              ring_only_ms=round(ring_only, 4))), flush=True)
        env._sh_packed, env._state_ring = own, own_ring
        del vol, ring, flat, arena, mem
    rehandle(env)
    print(json.dumps(dict(tuned_again_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
