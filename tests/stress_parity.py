#!/usr/bin/env python3
"""Randomised differential stress test (not part of the pytest suite): many
random configurations of the HIP env against the CPU oracle, run to
exhaustion, looking for rare decision mismatches (curvature threshold band,
mask-class margin, border folding, processing order).

    python tests/stress_parity.py [n_configs] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import env_oracle as orc  # noqa: E402
from tracktolearn_amd.datasets.utils import MRIDataVolume as Vol  # noqa: E402
from tracktolearn_amd.environments import (NoisyTrackingEnvironment,  # noqa: E402
                                           TrackingEnvironment)


def one(rng, k):
    shape = tuple(int(v) for v in rng.randint(10, 40, 3))
    C = int(rng.choice([6, 15, 28, 45]))
    K = int(rng.choice([1, 4, 7, 100]))
    theta = float(rng.choice([15, 20, 30, 45, 60, 90]))
    thr = float(rng.choice([0.05, 0.1, 0.3, 0.5, 0.8]))
    step_mm = float(rng.choice([0.3, 0.5, 0.75, 0.9, 1.0, 1.3]))
    noisy = bool(rng.randint(2))
    aff_dt = np.float64 if rng.randint(2) else np.float32
    reward = bool(rng.randint(2))
    N = int(rng.choice([1, 63, 257, 5000, 20000]))
    TrackingEnvironment.SPATIAL_ORDER_MIN = int(rng.choice([1, 1 << 30]))
    # (own generator: keeps the configurations of earlier logged runs)
    knobs = np.random.RandomState(1000 + k)
    TrackingEnvironment.SPATIAL_ORDER_REFRESH = int(knobs.choice([0, 1, 2, 5, 16]))
    # round-2 scheduling / layout knobs (none may change a result): fused small-
    # batch tail, record order of the packed SH volume, per-block re-sort of the
    # processing order, Morton / voxel-level order keys
    os.environ['TTL_FUSE_SMALL'] = str(knobs.choice([0, 1]))
    os.environ['TTL_SH_LAYOUT'] = str(knobs.choice(['brick4', 'linear']))
    os.environ['TTL_LOCAL_SORT'] = str(knobs.choice([0, 1]))
    os.environ['TTL_ORDER_KEY'] = str(knobs.choice([0, 1, 2, 3]))
    os.environ['TTL_ORDER_SORT'] = str(knobs.choice([0, 1]))
    os.environ['TTL_STORE_FLAVOUR'] = str(knobs.choice([0, 8]))      # tail-merge variants
    os.environ['TTL_XCD_ROTATE'] = str(knobs.randint(8))
    # round-3 knobs (again a generator of their own): largest batch of the
    # one-launch step tail, step() handing out lazily gathered state rows
    knobs3 = np.random.RandomState(3000 + k)
    fuse_max = int(knobs3.choice([4096, 16384, 65536]))
    os.environ['TTL_FUSE_MAX_ROWS'] = str(fuse_max)        # read when the handle is created ...
    TrackingEnvironment.FREERUN_MAX = fuse_max              # ... and by the host class at import
    TrackingEnvironment.lazy_step_state = bool(knobs3.randint(2))
    # the step tail of batches with a processing order: rows and slots in one
    # launch (the order keeps holes between refreshes) or the two-kernel tail;
    # how many holes the host tolerates before it refreshes early
    tail_max = int(knobs3.choice([0, 98304, 1048576]))
    os.environ['TTL_TAIL_FUSED'] = '1' if tail_max else '0'
    os.environ['TTL_TAIL_FUSED_MAX_ROWS'] = str(max(tail_max, 1))
    TrackingEnvironment.TAIL_FUSED_MAX_ROWS = tail_max
    TrackingEnvironment.ORDER_MIN_FILL = float(knobs3.choice([0.0, 0.5, 0.8, 0.99]))
    if os.environ.get('TTL_STRESS_VERBOSE'):
        print('config', k, dict(shape=shape, C=C, K=K, theta=theta, thr=thr, step=step_mm,
                                noisy=noisy, reward=reward, N=N,
                                order_min=TrackingEnvironment.SPATIAL_ORDER_MIN,
                                refresh=TrackingEnvironment.SPATIAL_ORDER_REFRESH,
                                knobs={v: os.environ[v] for v in sorted(os.environ)
                                       if v.startswith('TTL_')}), flush=True)
    X, Y, Z = shape
    sh = (0.1 * rng.standard_normal((X, Y, Z, C))).astype(np.float32)
    g = np.stack(np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing='ij'))
    ctr = np.array([(X - 1) / 2, (Y - 1) / 2, (Z - 1) / 2])[:, None, None, None]
    rad = rng.uniform(0.3, 0.55) * np.array([X, Y, Z])[:, None, None, None]
    mask = ((((g - ctr) / rad) ** 2).sum(0) < 1).astype(np.uint8)
    if mask.sum() == 0:
        mask[X // 2, Y // 2, Z // 2] = 1
    pk = rng.standard_normal((X, Y, Z, 15)).astype(np.float32)
    aff = np.eye(4, dtype=aff_dt)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    max_length = float(rng.choice([6.0, 20.0, 45.0]))
    dto = dict(n_dirs=K, theta=theta, npv=1, binary_stopping_threshold=thr,
               step_size=step_mm, min_length=1.0, max_length=max_length,
               compute_reward=reward, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device('cuda:0'),
               target_sh_order=None, noise=0.0, fa_map=None)
    cls = NoisyTrackingEnvironment if noisy else TrackingEnvironment
    env = cls((Vol(sh, aff), Vol(mask, aff), Vol(mask, aff), Vol(pk, aff), None),
              'testing', dto)
    env.seeds = seeds
    kw = dict(n_dirs=K, theta=theta, step_size=env.step_size,
              max_nb_steps=env.max_nb_steps, mask_threshold=thr, peaks=pk,
              compute_reward=reward, alignment_weighting=1.0, spline_eval='scipy')
    ref = (orc.OracleNoisyTrackingEnv(sh, mask, seeds, noise=0.0, **kw) if noisy
           else orc.OracleTrackingEnv(sh, mask, seeds, **kw))
    s_hip, s_ref = env.reset(0, N), ref.reset(0, N)
    worst = float(np.abs(s_hip.cpu().numpy() - s_ref).max())
    step = 0
    # round 2: some episodes run on free-running steps (ttl_env_freerun_step
    # launched eagerly for all N rows, the row count kept on the device)
    free = bool(knobs.randint(3) == 0) and env.freerun_supported()
    if free:
        import ctypes
        from tracktolearn_amd import _lib
        fr_state = env._new_state(N)
        fr_done = torch.empty(N, dtype=torch.uint8, device='cuda:0')
        fr_rew = torch.empty(N, dtype=torch.float64, device='cuda:0') if reward else None
        _lib.check(env._lib.ttl_env_freerun_begin(env._handle, env._host_counts.data_ptr(),
                                                  env._stream()))
    wob = float(rng.choice([0.05, 0.2, 0.5]))
    with np.errstate(all='ignore'):
        while len(ref.continue_idx):
            n = len(ref.continue_idx)
            if step == 0:
                a = rng.standard_normal((n, 3)).astype(np.float32)
            else:
                prev = s_ref[:, 7 * C:7 * C + 3].astype(np.float64)
                nrm = np.linalg.norm(prev, axis=1, keepdims=True)
                nrm[nrm == 0] = 1
                a = (prev / nrm + wob * rng.standard_normal((n, 3))).astype(np.float32)
            if free:
                a_all = np.zeros((N, 3), np.float32)
                a_all[:n] = a
                a_dev = torch.from_numpy(a_all).cuda()
                _lib.check(env._lib.ttl_env_freerun_step(
                    env._handle, a_dev.data_ptr(), N, fr_state.data_ptr(), env._state_pitch,
                    fr_rew.data_ptr() if reward else None, fr_done.data_ptr(), env._stream()))
                ns = fr_state.cpu().numpy()[env._row_dest_view(n).cpu().numpy()]
                d_all = fr_done.cpu().numpy().astype(bool)
                assert d_all[n:].all(), (k, step, 'rows that left earlier report done')
                d_hip = d_all[:n]
                r_hip = fr_rew.cpu().numpy()[:n] if reward else np.zeros(N)
                if reward:
                    assert not fr_rew.cpu().numpy()[n:].any(), (k, step, 'stale reward')
            elif rng.randint(2):
                ns_hip, r_hip, d_hip, _ = env.step(a.copy())
                ns = ns_hip.cpu().numpy()
            else:
                ns_hip, r_dev, d_dev, info = env.step_device(torch.from_numpy(a).cuda())
                ns = ns_hip.cpu().numpy()[info['row_dest'].cpu().numpy()]
                d_hip = d_dev.cpu().numpy().astype(bool)
                r_hip = r_dev.cpu().numpy() if r_dev is not None else np.zeros(N)
            ns_ref, r_ref, d_ref, _ = ref.step(a.copy())
            assert np.array_equal(d_hip, d_ref), (k, step, 'dones')
            both = np.isnan(ns) & np.isnan(ns_ref)
            err = np.abs(np.where(both, 0, ns - ns_ref))
            worst = max(worst, float(np.nanmax(err)))
            assert np.nanmax(err) <= 1e-5, (k, step, 'state', float(np.nanmax(err)))
            assert np.abs(r_hip - r_ref).max() <= 1e-5, (k, step, 'reward')
            s_ref, _ = ref.harvest()
            if free:        # the host view follows by hand; the device needs no harvest
                env._cur ^= 1
                env._n_active = len(ref.continue_idx)
                env.length += 1
                assert int(env._host_counts_np[0]) == env._n_active, (k, step, 'count')
            else:
                s_hip, _ = env.harvest()
            assert np.array_equal(env.continue_idx, ref.continue_idx), (k, step, 'idx')
            step += 1
    if free:
        n_left, length, steps_done = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(env._lib.ttl_env_freerun_end(env._handle, ctypes.byref(n_left), ctypes.byref(length),
                                                ctypes.byref(steps_done), env._stream()))
        assert (n_left.value, length.value, steps_done.value) == (0, step + 1, step), (k, 'freerun_end')
    assert np.array_equal(env.flags, ref.flags), (k, 'flags')
    assert np.array_equal(env.lengths, ref.lengths), (k, 'lengths')
    assert np.array_equal(env.streamlines, ref.streamlines), (k, 'positions')
    return dict(shape=shape, C=C, K=K, theta=theta, thr=thr, step=step_mm,
                noisy=noisy, aff=aff_dt.__name__, reward=reward, N=N, steps=step, free=free,
                worst_state_err=worst,
                stops=(int((ref.flags & 1).astype(bool).sum()),
                       int((ref.flags & 2).astype(bool).sum()),
                       int((ref.flags & 4).astype(bool).sum())))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.RandomState(seed)
    t0 = time.time()
    totals = np.zeros(3, np.int64)
    streamline_steps = 0
    worst = 0.0
    for k in range(n):
        r = one(rng, k)
        totals += np.array(r['stops'])
        streamline_steps += r['N'] * r['steps']
        worst = max(worst, r['worst_state_err'])
        print(k, r, flush=True)
    print(f'OK: {n} configurations, mask/length/curvature stops {totals.tolist()}, '
          f'worst |state - oracle| {worst:.3g}, {time.time() - t0:.0f} s')


if __name__ == '__main__':
    main()
