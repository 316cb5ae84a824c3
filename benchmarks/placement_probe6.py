#!/usr/bin/env python3
"""Does ttl_volume_probe rank placements of the packed SH volume reproducibly,
and does its ranking carry over to the step's gather?  Packs the bench volume
into 8 allocations, probes each three times (interleaved), then runs the bench
env on the best and on the worst candidate.

    python benchmarks/placement_probe6.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np
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
    print(json.dumps(dict(own_allocation_ms=round(timed(env), 4))), flush=True)
    src = env._sh_packed
    n_rec, pitch = src.shape
    dims = (C.c_int32 * 3)(*env._sh_dim)
    mask = subject[1].data
    vox = np.argwhere(mask)
    rng = np.random.RandomState(12345)
    n = 131072
    pts = vox[rng.randint(0, len(vox), n)] + rng.uniform(-0.5, 0.5, (n, 3))
    key = ((pts[:, 0].astype(np.int64) >> 3) << 40) | ((pts[:, 1].astype(np.int64) >> 3) << 20) | \
        (pts[:, 2].astype(np.int64) >> 3)
    pts = np.ascontiguousarray(pts[np.argsort(key, kind='stable')], dtype=np.float32)
    pts_dev = torch.from_numpy(pts).cuda()
    order = torch.arange(n, dtype=torch.int32, device='cuda:0')
    scratch = torch.empty((n, 7 * env._n_coef), dtype=torch.float32, device='cuda:0')
    cands = []
    for c in range(8):
        mem = _lib.DeviceVolume(0, n_rec * pitch * 4, c % 2 == 1)
        vol = torch.as_tensor(mem, device='cuda:0').view(torch.float32).view(n_rec, pitch)
        vol.copy_(src)
        cands.append((vol, mem))

    def probe(vol, reps=20):
        ms = C.c_double()
        _lib.check(env._lib.ttl_volume_probe(
            vol.data_ptr(), dims, env._n_coef, pitch, env._sh_layout,
            float(np.float32(env.add_neighborhood_vox)), pts_dev.data_ptr(), order.data_ptr(), n,
            scratch.data_ptr(), reps, env._stream(), C.byref(ms)))
        return round(ms.value, 5)

    table = [[probe(v) for v, _ in cands] for _ in range(3)]
    for row in table:
        print(json.dumps(dict(probe_ms=row)), flush=True)
    mean = np.mean(table, axis=0)
    print(json.dumps(dict(contiguous=[m.contiguous for _, m in cands],
                          mean_probe_ms=[round(float(m), 5) for m in mean])), flush=True)
    for name, c in (('best', int(np.argmin(mean))), ('worst', int(np.argmax(mean))),
                    ('best again', int(np.argmin(mean)))):
        env._sh_packed = cands[c][0]
        rehandle(env)
        print(json.dumps(dict(candidate=name, index=c, probe_ms=round(float(mean[c]), 5),
                              step_gather_ms=round(timed(env), 4))), flush=True)


if __name__ == '__main__':
    main()
