#!/usr/bin/env python3
"""State rows (the gather's output, 343 MB per step, read once by the policy) in
memory of other kinds: does keeping them out of the caches leave more of the
caches to the SH volume?  hipExtMallocWithFlags: default, fine-grained (0x1),
uncached (0x3), contiguous (0x4); the gather's time and the scripted policy's
(the reader of the rows) per kind.

    python benchmarks/placement_probe8.py
"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchmarks.ab_state_kernel import window  # noqa: E402
from benchmarks.placement_probe import timed  # noqa: E402

hip = C.CDLL('libamdhip64.so')
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipExtMallocWithFlags.restype = C.c_int


class _Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1',
                                         'data': (ptr, False), 'version': 2}


def alloc(nbytes, flags):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flags)
    if rc != 0:
        raise RuntimeError(f'hipExtMallocWithFlags({flags:#x}) -> {rc}')
    return torch.as_tensor(_Raw(p.value, nbytes), device='cuda:0')


def main():
    subject = bench.make_subject()
    env = bench.make_env(subject, 'cuda:0', 0)
    env.reset(0, bench.N_ACTOR)
    env._state_ring = None          # this script places the state rows itself
    window(env)
    print(json.dumps(dict(allocator_ms=round(timed(env), 4))), flush=True)
    W = env._state_width
    orig = env._new_state
    for name, flags in (('default', 0x0), ('fine-grained', 0x1), ('uncached', 0x3),
                        ('contiguous', 0x4)):
        try:
            arenas = [alloc(bench.N_ACTOR * W * 4, flags).view(torch.float32).view(bench.N_ACTOR, W)
                      for _ in range(2)]
        except RuntimeError as exc:
            print(json.dumps(dict(kind=name, error=str(exc))), flush=True)
            continue
        flip = [0]

        def new_state(n, arenas=arenas, flip=flip):
            flip[0] += 1
            return arenas[flip[0] & 1][:n]

        env._new_state = new_state
        window(env)
        ms = timed(env)
        # whole windows: the policy kernel reads the rows
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(5):
            k, _dt = window(env)
            n += k
        torch.cuda.synchronize()
        rate = n / (time.perf_counter() - t0)
        print(json.dumps(dict(kind=name, gather_ms=round(ms, 4),
                              window_Msteps_per_s=round(rate / 1e6, 1))), flush=True)
        env._new_state = orig
    print(json.dumps(dict(allocator_again_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
