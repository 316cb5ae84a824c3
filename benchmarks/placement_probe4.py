#!/usr/bin/env python3
"""Is the slow mode of the gather (0.20 against 0.178 ms, per box and per env
instance) a matter of physical contiguity (page fragments, TLB reach)?  Puts
the packed SH volume and / or the state rows of an instance into memory from
hipExtMallocWithFlags(hipDeviceMallocContiguous) and times the gather again.

    python benchmarks/placement_probe4.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchmarks.ab_state_kernel import window  # noqa: E402
from benchmarks.placement_probe import rehandle, timed  # noqa: E402

hip = C.CDLL('libamdhip64.so')
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipExtMallocWithFlags.restype = C.c_int


class _Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1',
                                         'data': (ptr, False), 'version': 2}


def contiguous_bytes(nbytes):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, 0x4)      # hipDeviceMallocContiguous
    if rc != 0:
        raise RuntimeError(f'hipExtMallocWithFlags(contiguous, {nbytes}) -> {rc}')
    return torch.as_tensor(_Raw(p.value, nbytes), device='cuda:0')


def main():
    subject = bench.make_subject()
    envs = []
    for i in range(3):
        env = bench.make_env(subject, 'cuda:0', 0)
        env.reset(0, bench.N_ACTOR)
        window(env)
        envs.append(env)
    base = [timed(env) for env in envs]
    print(json.dumps(dict(instances_ms=[round(b, 4) for b in base])), flush=True)
    env = envs[int(np.argmax(base))]
    W = env._state_width
    sh = env._sh_packed
    vol = contiguous_bytes(sh.numel() * 4).view(torch.float32).view(sh.shape)
    vol.copy_(sh)
    arenas = [contiguous_bytes(bench.N_ACTOR * W * 4).view(torch.float32).view(bench.N_ACTOR, W)
              for _ in range(2)]
    flip = [0]

    def new_state(n):
        flip[0] += 1
        return arenas[flip[0] & 1][:n]

    orig_new, orig_sh = env._new_state, env._sh_packed
    env._new_state = new_state
    window(env)
    print(json.dumps(dict(case='state rows in contiguous memory', ms=round(timed(env), 4))), flush=True)
    env._new_state = orig_new
    env._sh_packed = vol
    rehandle(env)
    print(json.dumps(dict(case='SH volume in contiguous memory', ms=round(timed(env), 4))), flush=True)
    env._new_state = new_state
    window(env)
    print(json.dumps(dict(case='both', ms=round(timed(env), 4))), flush=True)
    env._new_state = orig_new
    env._sh_packed = orig_sh
    rehandle(env)
    print(json.dumps(dict(case='back to the allocator', ms=round(timed(env), 4))), flush=True)
    # the streamline history (840 MB, one 3.2-KB row per streamline: the buffer
    # with the least locality per page) in contiguous memory
    hist_shape = tuple(env._buf_streamlines.shape)
    hist = contiguous_bytes(int(np.prod(hist_shape)) * 4).view(torch.float32).view(hist_shape)
    real_empty = torch.empty

    def fake_empty(*args, **kw):
        shape = args[0] if len(args) == 1 and isinstance(args[0], (tuple, list)) else args
        if tuple(shape) == hist_shape and kw.get('dtype') is torch.float32:
            return hist
        return real_empty(*args, **kw)

    torch.empty = fake_empty
    try:
        rehandle(env)
    finally:
        torch.empty = real_empty
    assert env._buf_streamlines.data_ptr() == hist.data_ptr()
    print(json.dumps(dict(case='history in contiguous memory', ms=round(timed(env), 4))), flush=True)
    rehandle(env)
    print(json.dumps(dict(case='back to the allocator', ms=round(timed(env), 4))), flush=True)
    for i, e in enumerate(envs):
        print(json.dumps(dict(i=i, final_ms=round(timed(e), 4))), flush=True)


if __name__ == '__main__':
    main()
