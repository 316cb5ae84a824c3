"""GPU parity tests: the HIP environment (through the C ABI) against
 (a) the golden traces captured from the reference, and
 (b) the CPU oracle on seeded random inputs.

Bars (BASELINE.json north_star): stopping masks / indices / flags / lengths
bit-exact; positions, states and rewards within 1e-5.  Positions are in fact
required to be bit-identical here (IEEE arithmetic in the reference's order).
"""
import numpy as np
import pytest
import torch

from helpers import (TRACES, load_trace, synthetic_subject, trace_noise,
                     trace_step_size)

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _dto(*, n_dirs, theta=30.0, max_length, reward, noise=0.0, thr=0.1,
         step_size=0.75):
    return dict(n_dirs=n_dirs, theta=theta, npv=1,
                binary_stopping_threshold=thr, step_size=step_size,
                min_length=2.0,
                max_length=max_length, compute_reward=reward,
                alignment_weighting=1.0, oracle_bonus=0.0,
                oracle_checkpoint=None, oracle_stopping_criterion=False,
                rng=np.random.RandomState(0), device=torch.device('cuda:0'),
                target_sh_order=8, noise=noise, fa_map=None)


def _hip_env(D, *, noisy, affine_dtype, seeds, affine=None, **kw):
    from tracktolearn_amd.datasets.utils import MRIDataVolume as Vol
    from tracktolearn_amd.environments import (NoisyTrackingEnvironment,
                                               TrackingEnvironment)
    sh, mask, pk = synthetic_subject(D)
    aff = np.eye(4, dtype=affine_dtype)
    if affine is not None:
        aff = np.asarray(affine).astype(affine_dtype)
    subject = (Vol(sh, aff), Vol(mask.astype(np.float32), aff),
               Vol(mask.astype(np.float32), aff), Vol(pk, aff), None)
    cls = NoisyTrackingEnvironment if noisy else TrackingEnvironment
    env = cls(subject, 'testing', _dto(**kw))
    env.seeds = seeds
    return env


def _env_from_trace(z):
    max_length = float(z['max_nb_steps']) * 0.75 + 0.01
    aff = np.float32 if str(z['step_size_dtype']) == 'float32' else np.float64
    env = _hip_env(int(z['D']), noisy=bool(z['noisy']), affine_dtype=aff,
                   affine=z['affine'] if 'affine' in z.files else None,
                   seeds=z['seeds'], n_dirs=int(z['n_dirs']),
                   theta=float(z['theta']), max_length=max_length,
                   reward=bool(z['reward']), noise=trace_noise(z)[0],
                   thr=float(z['mask_threshold']))
    if trace_noise(z)[1] is not None:
        env.rng.set_state(trace_noise(z)[1].get_state())
    assert env.max_nb_steps == int(z['max_nb_steps'])
    assert env.step_size == trace_step_size(z)
    if 'mask_coef' in z.files:
        assert np.array_equal(env._mask_coef.cpu().numpy(), z['mask_coef'])
    return env


def _close(a, b, tol=TOL):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    both_nan = np.isnan(a) & np.isnan(b)
    return np.all((np.abs(a - b) <= tol) | both_nan)


#: small batches (<= 16384 rows, no processing order) run prefix + compaction
#: + gather as ONE kernel with a polled count word (TTL_FUSE_SMALL, default
#: on); '0' keeps them on the three-launch path of the large batches
FUSED = pytest.mark.parametrize('fused', ['1', '0'])


#: step() hands its state rows out as a lazily gathered tensor over rows written
#: survivors first (harvest() is then a view); False: rows written in the
#: reference's order, harvest() copies the survivors' rows (k_finish, k_copy_rows)
LAZY = pytest.mark.parametrize('lazy', [True, False])


@LAZY
@FUSED
@pytest.mark.parametrize('name', TRACES)
def test_reference_trace_step_harvest(name, fused, lazy, monkeypatch):
    """step()/harvest() -- the reference's calling contract."""
    monkeypatch.setenv('TTL_FUSE_SMALL', fused)
    z = load_trace(name)
    env = _env_from_trace(z)
    env.lazy_step_state = lazy
    N = z['seeds'].shape[0]
    state = env.reset(0, N)
    assert state.dtype == torch.float32 and state.is_cuda
    assert _close(state.cpu().numpy(), z['state_reset'])
    for s in range(int(z['n_steps'])):
        assert np.array_equal(env.continue_idx, z[f'continue_idx_{s}'])
        nstate, rew, done, info = env.step(z[f'actions_{s}'].copy())
        assert isinstance(nstate, torch.Tensor) and nstate.is_cuda
        assert nstate.dtype == torch.float32 and nstate.shape[0] == len(z[f'dones_{s}'])
        assert done.dtype == bool and np.array_equal(done, z[f'dones_{s}'])
        assert np.array_equal(info['continue_idx'], z[f'continue_idx_{s}'])
        assert np.array_equal(env.flags, z[f'flags_{s}'])
        assert rew.dtype == np.float64 and rew.shape == z[f'reward_{s}'].shape
        assert _close(rew, z[f'reward_{s}'])
        if bool(z['reward']):
            assert abs(info['reward_info']['peaks_reward'] -
                       float(z[f'reward_info_peaks_{s}'])) <= TOL
        idx = z[f'continue_idx_{s}']
        head = env.streamlines[idx, env.length - 1]
        assert np.array_equal(head, z[f'head_{s}'])         # bit-identical
        ns = nstate.cpu().numpy()
        if f'state_{s}' in z.files:
            assert _close(ns, z[f'state_{s}'])
        assert np.abs(ns.astype(np.float64).sum(axis=1) -
                      z[f'state_rowsum_{s}']).max() <= 327 * TOL
        hstate, not_stopping = env.harvest()
        assert np.array_equal(not_stopping, ~z[f'dones_{s}'])
        assert hstate.shape[0] == int(z[f'harvest_rows_{s}'])
        assert torch.equal(hstate, nstate[torch.from_numpy(~z[f'dones_{s}'])])
        assert np.array_equal(env.lengths, z[f'lengths_{s}'])
        assert np.array_equal(env.continue_idx, z[f'new_continue_idx_{s}'])
    assert np.array_equal(env.streamlines, z['streamlines'])
    tg = env.get_streamlines()
    assert np.array_equal([len(s) for s in tg.streamlines], z['tract_lengths'])
    assert np.array_equal(np.concatenate(tg.streamlines), z['tract_points'])
    assert np.array_equal(tg.data_per_streamline['flags'], z['tract_flags'])
    assert np.array_equal(tg.data_per_streamline['seeds'], z['tract_seeds'])


@FUSED
@pytest.mark.parametrize('name', TRACES)
def test_reference_trace_device_loop(name, fused, monkeypatch):
    """step_device()/harvest(): survivors-first rows, nothing leaves the GPU
    except the 8-byte counters."""
    monkeypatch.setenv('TTL_FUSE_SMALL', fused)
    z = load_trace(name)
    env = _env_from_trace(z)
    N = z['seeds'].shape[0]
    env.reset(0, N)
    for s in range(int(z['n_steps'])):
        acts = torch.from_numpy(z[f'actions_{s}']).cuda()
        nstate, rew, done, info = env.step_device(acts)
        dest = info['row_dest'].cpu().numpy()
        done_np = done.cpu().numpy().astype(bool)
        assert np.array_equal(done_np, z[f'dones_{s}'])
        # row_dest is the stable partition: survivors, then stopped
        n_keep = int((~done_np).sum())
        want = np.empty(len(done_np), np.int64)
        want[~done_np] = np.arange(n_keep)
        want[done_np] = n_keep + np.arange(len(done_np) - n_keep)
        assert np.array_equal(dest, want)
        ns = nstate.cpu().numpy()[dest]          # back to active-row order
        if f'state_{s}' in z.files:
            assert _close(ns, z[f'state_{s}'])
        if bool(z['reward']):
            assert _close(rew.cpu().numpy(), z[f'reward_{s}'])
        hstate, not_stopping = env.harvest()
        if hstate.shape[0]:
            assert hstate.data_ptr() == nstate.data_ptr()  # a view, no copy
        assert hstate.shape[0] == int(z[f'harvest_rows_{s}'])
        assert not_stopping is None
        assert np.array_equal(env.continue_idx, z[f'new_continue_idx_{s}'])
        assert np.array_equal(env.lengths, z[f'lengths_{s}'])
        assert np.array_equal(env.flags, z[f'flags_{s}'])
    assert np.array_equal(env.streamlines, z['streamlines'])


def test_isolated_stopping_vectors():
    """Adversarial curvature triples (angles within 1e-6 rad of theta, zero
    length and reversed segments) and mask points on / next to every border,
    against the reference's recorded decisions."""
    z = load_trace('isolated_functions')
    D = z['mask_in'].shape[0]
    env = _hip_env(D, noisy=False, affine_dtype=np.float32,
                   seeds=np.zeros((1, 3)), n_dirs=4,
                   theta=float(z['curvy_theta']), max_length=1000.0,
                   reward=False)
    assert np.array_equal(env._mask_coef.cpu().numpy(), z['mask_coef'])
    # mask: single-point streamlines
    pts = z['mask_pts']
    stop, flags = env._compute_stopping_flags(pts[:, None, :])
    assert np.array_equal(stop, z['mask_stop'])
    assert set(np.unique(flags)) <= {0, 1}
    # curvature: 3-point streamlines deep inside the mask region are also
    # tested by the mask, so compare only the curvature bit
    tri = z['curvy_in']
    _, flags = env._compute_stopping_flags(tri)
    assert np.array_equal((flags & 4) != 0, z['curvy_out'])


def _scripted(rng, state_np, n_sh, step, wobble):
    n = state_np.shape[0]
    if step == 0:
        return rng.standard_normal((n, 3)).astype(np.float32)
    prev = state_np[:, n_sh:n_sh + 3].astype(np.float64)
    nrm = np.linalg.norm(prev, axis=1, keepdims=True)
    nrm[nrm == 0] = 1.0
    return (prev / nrm + wobble * rng.standard_normal((n, 3))).astype(np.float32)


@pytest.mark.parametrize('noisy,affine,K,reward,step_mm,kernel', [
    (False, np.float32, 4, True, 0.75, None),
    (True, np.float64, 100, False, 0.75, None),
    (False, np.float64, 4, False, 0.75, None),  # plain env, f64 affine: numpy-2 promotion
    (False, np.float32, 4, False, 0.75, '0'),   # the 56-fetch state kernel
    (False, np.float32, 4, False, 1.25, None),  # radius >= 1 voxel -> 56-fetch kernel
    (False, np.float32, 4, False, 0.30, None),  # small radius: few outer slices
    (False, np.float32, 4, True, 0.75, 'sorted'),   # brick-sorted processing order
    (True, np.float64, 100, False, 0.75, 'sorted'),
])
def test_random_episode_against_oracle(noisy, affine, K, reward, step_mm, kernel,
                                       monkeypatch):
    """4096 streamlines on a 24^3 volume, scripted actions, run to exhaustion:
    HIP env vs CPU oracle step by step."""
    from oracle import env_oracle as orc
    if kernel == 'sorted':
        from tracktolearn_amd.environments import TrackingEnvironment
        monkeypatch.setattr(TrackingEnvironment, 'SPATIAL_ORDER_MIN', 1)
        # ... and re-sorted from the current positions every other step
        monkeypatch.setattr(TrackingEnvironment, 'SPATIAL_ORDER_REFRESH', 2)
    elif kernel is not None:
        monkeypatch.setenv('TTL_STATE_KERNEL', kernel)
    D, N = 24, 4096
    sh, mask, pk = synthetic_subject(D)
    rng = np.random.RandomState(11)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    max_length = 30.0
    env = _hip_env(D, noisy=noisy, affine_dtype=affine, seeds=seeds, n_dirs=K,
                   max_length=max_length, reward=reward, step_size=step_mm)
    kw = dict(n_dirs=K, theta=30.0, step_size=env.step_size,
              max_nb_steps=env.max_nb_steps, mask_threshold=0.1, peaks=pk,
              compute_reward=reward, alignment_weighting=1.0)
    ref = (orc.OracleNoisyTrackingEnv(sh, mask, seeds, noise=0.0, **kw) if noisy
           else orc.OracleTrackingEnv(sh, mask, seeds, **kw))
    s_hip = env.reset(0, N)
    s_ref = ref.reset(0, N)
    assert _close(s_hip.cpu().numpy(), s_ref)
    step = 0
    seen = 0
    while len(ref.continue_idx):
        a = _scripted(rng, s_ref, 7 * 45, step, 0.2)
        ns_hip, r_hip, d_hip, _ = env.step(a.copy())
        ns_ref, r_ref, d_ref, _ = ref.step(a.copy())
        assert np.array_equal(d_hip, d_ref)
        assert _close(ns_hip.cpu().numpy(), ns_ref)
        assert _close(r_hip, r_ref)
        s_hip, _ = env.harvest()
        s_ref, _ = ref.harvest()
        assert np.array_equal(env.continue_idx, ref.continue_idx)
        seen |= int(np.bitwise_or.reduce(ref.flags))
        step += 1
    assert np.array_equal(env.flags, ref.flags)
    assert np.array_equal(env.lengths, ref.lengths)
    assert np.array_equal(env.streamlines, ref.streamlines)   # bit-identical
    if step_mm == 0.75:
        assert seen == 7      # MASK, LENGTH and CURVATURE all exercised


@pytest.mark.parametrize('C,K,sorted_order', [
    (1, 4, False),     # < 4 coefficients: scalar tail stores
    (6, 4, True),      # 2 columns, lane groups of 4, 2-float tail
    (15, 1, False),    # 3-float tail
    (28, 7, True),     # lane groups of 8, no tail
    (64, 4, True),     # lane groups of 16, no tail
    (63, 13, False),   # lane groups of 16, 3-float tail, K > lane group
    (66, 4, True),     # 17 columns: the looping 32-lane variant
    (91, 100, False),  # SH order 12
])
def test_other_sh_orders(C, K, sorted_order, monkeypatch):
    """State rows for every lane-group size / tail shape of the gather kernel
    (the traces and the other tests all use 45 coefficients)."""
    from oracle import env_oracle as orc
    from tracktolearn_amd.datasets.utils import MRIDataVolume as Vol
    from tracktolearn_amd.environments import TrackingEnvironment
    monkeypatch.setattr(TrackingEnvironment, 'SPATIAL_ORDER_MIN',
                        1 if sorted_order else 1 << 30)
    D, N = 14, 700
    sh, mask, pk = synthetic_subject(D, C=C)
    rng = np.random.RandomState(C)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    aff = np.eye(4, dtype=np.float32)
    env = TrackingEnvironment(
        (Vol(sh, aff), Vol(mask.astype(np.float32), aff),
         Vol(mask.astype(np.float32), aff), Vol(pk, aff), None), 'testing',
        _dto(n_dirs=K, max_length=12.0, reward=False))
    env.seeds = seeds
    ref = orc.OracleTrackingEnv(sh, mask, seeds, n_dirs=K, theta=30.0,
                                step_size=env.step_size, max_nb_steps=env.max_nb_steps,
                                mask_threshold=0.1, peaks=pk, compute_reward=False,
                                alignment_weighting=1.0)
    s_hip, s_ref = env.reset(0, N), ref.reset(0, N)
    assert s_hip.shape == (N, 7 * C + 3 * K)
    assert _close(s_hip.cpu().numpy(), s_ref)
    step = 0
    while len(ref.continue_idx):
        a = _scripted(rng, s_ref, 7 * C, step, 0.2)
        if step % 2:
            import torch
            ns_hip, _, d_dev, info = env.step_device(torch.from_numpy(a).cuda())
            ns = ns_hip.cpu().numpy()[info['row_dest'].cpu().numpy()]
            d_hip = d_dev.cpu().numpy().astype(bool)
        else:
            ns_hip, _, d_hip, _ = env.step(a.copy())
            ns = ns_hip.cpu().numpy()
        ns_ref, _, d_ref, _ = ref.step(a.copy())
        assert np.array_equal(d_hip, d_ref)
        assert _close(ns, ns_ref)
        s_hip, _ = env.harvest()
        s_ref, _ = ref.harvest()
        assert _close(s_hip.cpu().numpy(), s_ref)
        step += 1
    assert np.array_equal(env.streamlines, ref.streamlines)
    assert step > 3


def test_edge_cases():
    """Single streamline, zero action (NaN direction -> mask stop), seeds
    outside the volume, ragged re-use of a larger handle."""
    from oracle import env_oracle as orc
    D = 12
    sh, mask, pk = synthetic_subject(D)
    seeds = np.array([[5.5, 5.5, 5.5], [-3.0, 2.0, 2.0], [5.0, 6.0, 40.0],
                      [0.2, 0.3, 11.2], [6.0, 6.0, 6.0]])
    env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds,
                   n_dirs=4, max_length=20.0, reward=True)
    ref = orc.OracleTrackingEnv(
        sh, mask, seeds, n_dirs=4, theta=30.0, step_size=env.step_size,
        max_nb_steps=env.max_nb_steps, mask_threshold=0.1, peaks=pk,
        compute_reward=True, alignment_weighting=1.0)
    for (a, b) in [(0, 5), (0, 1), (1, 4)]:        # shrinking batches re-use buffers
        s_hip, s_ref = env.reset(a, b), ref.reset(a, b)
        assert _close(s_hip.cpu().numpy(), s_ref)
        n = b - a
        acts = np.tile(np.array([[1.0, 0.5, -0.25]], np.float32), (n, 1))
        acts[0] = 0.0                               # zero action -> NaN step
        with np.errstate(all='ignore'):
            ns_ref, r_ref, d_ref, _ = ref.step(acts.copy())
        ns_hip, r_hip, d_hip, _ = env.step(acts.copy())
        assert np.array_equal(d_hip, d_ref)
        assert d_hip[0]                             # NaN position leaves the mask
        assert _close(ns_hip.cpu().numpy(), ns_ref)
        assert _close(r_hip, r_ref)
        env.harvest()
        ref.harvest()
        assert np.array_equal(env.flags, ref.flags)
        assert np.array_equal(env.continue_idx, ref.continue_idx)
    assert env.get_state_size() == 7 * 45 + 12


@pytest.mark.parametrize('N', [1, 255, 256, 257, 5000, 16384, 16385, 20000])
def test_fused_and_three_launch_tails_agree_at_the_batch_boundaries(N, monkeypatch):
    """The one-launch tail (batches <= 64 x 256 rows) against the three-launch
    path on the same inputs, around the workgroup, wave and path boundaries:
    identical rows, indices, flags and counts, in both row orders, with both
    record orders of the packed SH volume (dims not multiples of 4)."""
    D = 13
    sh, mask, pk = synthetic_subject(D)
    rng = np.random.RandomState(N)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    acts = [rng.standard_normal((N, 3)).astype(np.float32) for _ in range(3)]
    runs = {}
    for fused, layout in (('1', 'brick4'), ('0', 'brick4'), ('1', 'linear'), ('0', 'linear')):
        monkeypatch.setenv('TTL_FUSE_SMALL', fused)
        monkeypatch.setenv('TTL_SH_LAYOUT', layout)
        env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds, n_dirs=4,
                       max_length=20.0, reward=True)
        out = [env.reset(0, N).cpu().numpy()]
        for step in range(3):
            n = env._n_active
            if n == 0:
                break
            a = torch.from_numpy(acts[step][:n]).cuda()
            if step % 2 == 0:
                ns, rew, done, info = env.step_device(a)
                out += [ns[info['row_dest'].long()].cpu().numpy(), rew.cpu().numpy(),
                        done.cpu().numpy()]
            else:
                ns, rew, done, info = env.step(a)
                out += [ns.cpu().numpy(), rew, done.astype(np.uint8)]
            state, _ = env.harvest()
            out += [state.cpu().numpy(), env.continue_idx.copy()]
        out += [env.flags.copy(), env.lengths.copy(), env.streamlines.copy()]
        runs[(fused, layout)] = out
    first = runs[('1', 'brick4')]
    for key, other in runs.items():
        assert len(other) == len(first), key
        for x, y in zip(first, other):
            assert np.array_equal(x, y, equal_nan=True), key


@pytest.mark.parametrize('N', [3000, 20000])
def test_count_delivery_paths_agree(N, monkeypatch):
    """The survivor count written by the kernel into the pinned buffer and
    polled (default) against the side-stream copy + event (TTL_POLL_COUNTS=0),
    on the one-launch tail (N = 3000) and on the large-batch path (N = 20000)."""
    D = 16
    sh, mask, pk = synthetic_subject(D)
    rng = np.random.RandomState(N)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    acts = [rng.standard_normal((N, 3)).astype(np.float32) for _ in range(6)]
    runs = []
    for poll in ('1', '0'):
        monkeypatch.setenv('TTL_POLL_COUNTS', poll)
        env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds, n_dirs=4,
                       max_length=20.0, reward=False)
        env.reset(0, N)
        out = []
        for step in range(6):
            n = env._n_active
            if n == 0:
                break
            _, _, done, _ = env.step_device(torch.from_numpy(acts[step][:n]).cuda())
            state, _ = env.harvest()
            assert env._n_active == int((done == 0).sum()) == state.shape[0]
            assert int(env._host_counts_np[0]) + int(env._host_counts_np[1]) == n
            out.append((env._n_active, env.continue_idx.copy()))
        runs.append(out)
    assert len(runs[0]) == len(runs[1]) > 2
    for (na, ia), (nb, ib) in zip(*runs):
        assert na == nb and np.array_equal(ia, ib)


def _anisotropic_subject(shape, C=45, seed=77):
    """Non-cubic volumes: catches any x/y/z stride mix-up."""
    from tracktolearn_amd.datasets.utils import MRIDataVolume as Vol
    rng = np.random.RandomState(seed)
    X, Y, Z = shape
    sh = (0.1 * rng.standard_normal((X, Y, Z, C))).astype(np.float32)
    sh[..., 0] = 1.0
    g = np.stack(np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z),
                             indexing='ij')).astype(np.float64)
    centre = np.array([(X - 1) / 2, (Y - 1) / 2, (Z - 1) / 2])[:, None, None, None]
    radii = np.array([0.42 * X, 0.42 * Y, 0.42 * Z])[:, None, None, None]
    mask = ((((g - centre) / radii) ** 2).sum(0) < 1.0).astype(np.uint8)
    pk = rng.standard_normal((X, Y, Z, 15)).astype(np.float32)
    aff = np.eye(4, dtype=np.float32)
    return (Vol(sh, aff), Vol(mask, aff), Vol(mask, aff), Vol(pk, aff), None), sh, mask, pk


@pytest.mark.parametrize('theta,thr', [(30.0, 0.1), (20.0, 0.5), (60.0, 0.3), (90.0, 0.1)])
def test_non_cubic_volume_other_angles_and_thresholds(theta, thr, monkeypatch):
    """18 x 26 x 34 volume (all strides differ), several curvature angles and
    mask thresholds, brick-sorted processing order on: HIP env vs oracle."""
    from oracle import env_oracle as orc
    from tracktolearn_amd.environments import TrackingEnvironment
    monkeypatch.setattr(TrackingEnvironment, 'SPATIAL_ORDER_MIN', 1)
    shape = (18, 26, 34)
    subject, sh, mask, pk = _anisotropic_subject(shape)
    N = 3000
    rng = np.random.RandomState(5)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    dto = _dto(n_dirs=4, theta=theta, max_length=25.0, reward=True, thr=thr)
    env = TrackingEnvironment(subject, 'testing', dto)
    env.seeds = seeds
    ref = orc.OracleTrackingEnv(
        sh, mask, seeds, n_dirs=4, theta=theta, step_size=env.step_size,
        max_nb_steps=env.max_nb_steps, mask_threshold=thr, peaks=pk,
        compute_reward=True, alignment_weighting=1.0)
    s_hip, s_ref = env.reset(0, N), ref.reset(0, N)
    assert _close(s_hip.cpu().numpy(), s_ref)
    step = 0
    while len(ref.continue_idx):
        a = _scripted(rng, s_ref, 7 * 45, step, 0.35)
        ns_hip, r_hip, d_hip, _ = env.step(a.copy())
        ns_ref, r_ref, d_ref, _ = ref.step(a.copy())
        assert np.array_equal(d_hip, d_ref)
        assert _close(ns_hip.cpu().numpy(), ns_ref)
        assert _close(r_hip, r_ref)
        s_hip, _ = env.harvest()
        s_ref, _ = ref.harvest()
        step += 1
    assert np.array_equal(env.flags, ref.flags)
    assert np.array_equal(env.streamlines, ref.streamlines)
    assert (ref.flags & 4).any() and (ref.flags & 1).any()


def test_noisy_env_with_host_rng_noise():
    """NoisyTrackingEnvironment with sigma > 0: the noise is drawn from
    env_dto['rng'] on the host exactly as the reference does
    (noisy_tracking_env.py:73-77), so equal RandomStates give bit-identical
    tracks."""
    from oracle import env_oracle as orc
    D, N = 20, 2048
    sh, mask, pk = synthetic_subject(D)
    rng = np.random.RandomState(8)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    env = _hip_env(D, noisy=True, affine_dtype=np.float64, seeds=seeds, n_dirs=4,
                   max_length=25.0, reward=False, noise=0.08)
    env.rng = np.random.RandomState(4242)
    ref = orc.OracleNoisyTrackingEnv(
        sh, mask, seeds, noise=0.08, rng=np.random.RandomState(4242), n_dirs=4,
        theta=30.0, step_size=env.step_size, max_nb_steps=env.max_nb_steps,
        mask_threshold=0.1, peaks=pk, compute_reward=False)
    s_hip, s_ref = env.reset(0, N), ref.reset(0, N)
    step = 0
    while len(ref.continue_idx):
        a = _scripted(rng, s_ref, 7 * 45, step, 0.1)
        _, _, d_hip, _ = env.step(a.copy())
        _, _, d_ref, _ = ref.step(a.copy())
        assert np.array_equal(d_hip, d_ref)
        s_hip, _ = env.harvest()
        s_ref, _ = ref.harvest()
        step += 1
    assert step > 3
    assert np.array_equal(env.streamlines, ref.streamlines)
    assert np.array_equal(env.flags, ref.flags)


def test_mask_class_shortcut_changes_no_decision(monkeypatch):
    """The per-cell class table (ttl_mask_classes) only skips spline
    evaluations whose outcome is certain: decisions with and without it agree
    on random points, on points hugging every border and cell corner, and with
    scipy; and it really does decide most cells."""
    from scipy.ndimage import map_coordinates
    z = load_trace('isolated_functions')
    D = z['mask_in'].shape[0]
    rng = np.random.RandomState(12)
    pts = np.concatenate([
        z['mask_pts'],
        rng.uniform(-1.0, D + 1.0, (200000, 3)).astype(np.float32),
        (rng.randint(0, D + 1, (20000, 3)) + 0.5 +
         rng.choice([-1e-6, 0.0, 1e-6], (20000, 3))).astype(np.float32)])
    results = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('TTL_MASK_CLASSES', flag)
        for thr in (0.1, 0.5, 0.9):
            env = _hip_env(D, noisy=False, affine_dtype=np.float32,
                           seeds=np.zeros((1, 3)), n_dirs=4, max_length=1000.0,
                           reward=False, thr=thr)
            stop, _ = env._compute_stopping_flags(pts[:, None, :])
            results[(flag, thr)] = stop
            if flag == '1':
                cls = env._mask_cls.cpu().numpy()
                # a 12^3 volume is mostly boundary shell; the 96^3 benchmark
                # volume decides > 80 % of its cells this way
                assert (cls == 0).any() and (cls != 0).any()
                assert set(np.unique(cls)) <= {0, 1, 2}
            else:
                assert env._mask_cls is None
    coef = z['mask_coef']
    vals = map_coordinates(coef, pts.T - np.float32(0.5), prefilter=False)
    for thr in (0.1, 0.5, 0.9):
        assert np.array_equal(results[('1', thr)], results[('0', thr)])
        assert np.array_equal(results[('1', thr)], vals < thr)


def test_error_paths_return_codes_not_faults():
    """Call-order and argument errors come back as TTL_ERR_* with a message
    (C ABI) or as exceptions (host classes); nothing is launched for them and
    the handle keeps working afterwards."""
    from tracktolearn_amd import _lib
    D, N = 12, 64
    sh, mask, pk = synthetic_subject(D)
    rng = np.random.RandomState(3)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds, n_dirs=4,
                   max_length=10.0, reward=False)
    with pytest.raises(RuntimeError):
        env.harvest()          # nothing stepped (no handle yet either)
    env._ensure_capacity(N)    # fresh handle, never reset
    lib, h = env._lib, env._handle
    W = env._state_width
    state = torch.empty((N, W), dtype=torch.float32, device='cuda')
    acts = torch.randn(N, 3, device='cuda')
    done = torch.empty(N, dtype=torch.uint8, device='cuda')
    stream = env._stream()
    # step before reset
    rc = lib.ttl_env_step(h, acts.data_ptr(), None, N, 0, state.data_ptr(), W, None,
                          done.data_ptr(), None, stream)
    assert rc == -3 and lib.ttl_last_error()
    # harvest with nothing stepped
    assert lib.ttl_env_harvest(h, state.data_ptr(), state.data_ptr(), W, stream) == -3
    with pytest.raises(RuntimeError):
        env.harvest()
    s0 = env.reset(0, N)
    # wrong number of active rows, bad order, pitch narrower than a row, nulls
    assert lib.ttl_env_step(h, acts.data_ptr(), None, N - 1, 0, state.data_ptr(), W,
                            None, done.data_ptr(), None, stream) == -1
    assert lib.ttl_env_step(h, acts.data_ptr(), None, N, 7, state.data_ptr(), W,
                            None, done.data_ptr(), None, stream) == -1
    assert lib.ttl_env_step(h, acts.data_ptr(), None, N, 0, state.data_ptr(), W - 1,
                            None, done.data_ptr(), None, stream) == -1
    assert lib.ttl_env_step(h, None, None, N, 0, state.data_ptr(), W, None,
                            done.data_ptr(), None, stream) == -1
    assert lib.ttl_env_step(None, acts.data_ptr(), None, N, 0, state.data_ptr(), W,
                            None, done.data_ptr(), None, stream) == -1
    assert lib.ttl_env_reset(h, s0.data_ptr(), N + 10 ** 6, None, state.data_ptr(), W,
                             stream) == -1
    with pytest.raises(ValueError):
        env.step(np.zeros((N - 1, 3), np.float32))
    # the failed calls changed nothing: a normal step still matches the oracle
    from oracle import env_oracle as orc
    ref = orc.OracleTrackingEnv(sh, mask, seeds, n_dirs=4, theta=30.0,
                                step_size=env.step_size, max_nb_steps=env.max_nb_steps,
                                mask_threshold=0.1, peaks=pk, compute_reward=False,
                                alignment_weighting=1.0)
    ref.reset(0, N)
    a = rng.standard_normal((N, 3)).astype(np.float32)
    ns, _, d, _ = env.step(a.copy())
    ns_ref, _, d_ref, _ = ref.step(a.copy())
    assert np.array_equal(d, d_ref) and _close(ns.cpu().numpy(), ns_ref)
    with pytest.raises(RuntimeError):
        env.step(a)            # second step without harvest
    env.harvest()


def test_counts_into_pageable_host_memory_fall_back_to_the_copy():
    """`host_counts` that is NOT pinned / device-visible (a plain numpy
    buffer): the library notices (hipHostGetDevicePointer fails), keeps the
    kernel from writing through it and delivers the counts with the side-stream
    copy instead; same numbers as through the pinned, polled buffer."""
    D, N = 12, 300
    sh, mask, pk = synthetic_subject(D)
    rng = np.random.RandomState(4)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds, n_dirs=4,
                   max_length=10.0, reward=False)
    env.reset(0, N)
    lib, h, W = env._lib, env._handle, env._state_width
    stream = env._stream()
    acts = torch.randn(N, 3, device='cuda')
    state = torch.empty((N, W), dtype=torch.float32, device='cuda')
    done = torch.empty(N, dtype=torch.uint8, device='cuda')
    pageable = np.full(4, -7, dtype=np.int32)
    assert lib.ttl_env_step(h, acts.data_ptr(), None, N, 1, state.data_ptr(), W, None,
                            done.data_ptr(), pageable.ctypes.data, stream) == 0
    assert lib.ttl_env_harvest(h, None, None, W, stream) == 0
    assert lib.ttl_env_wait_counts(h) == 0
    n_stopped = int(done.sum())
    assert pageable[0] == N - n_stopped and pageable[1] == n_stopped
    assert pageable[2] == -7                     # no sequence word on this path
    # the next step must be launched for exactly the survivors
    n = int(pageable[0])
    assert lib.ttl_env_step(h, acts.data_ptr(), None, n + 1, 1, state.data_ptr(), W, None,
                            done.data_ptr(), pageable.ctypes.data, stream) == -1
    assert lib.ttl_env_step(h, acts.data_ptr(), None, n, 1, state.data_ptr(), W, None,
                            done.data_ptr(), pageable.ctypes.data, stream) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize('N,tail_fused', [(3000, '1'), (40000, '1'), (40000, '0')],
                         ids=['one_launch_tail', 'k_tail', 'k_prefix'])
def test_stopped_rows_list(N, tail_fused, monkeypatch):
    """ttl_env_stopped: between a step and its harvest, the {active row,
    streamline id} pairs of the rows that stopped, in row order -- what
    `np.arange(N)[dones]` and `continue_idx[dones]` give (oracle_reward.py:78)
    -- from each of the three kernels that compact a step."""
    import ctypes as C

    from tracktolearn_amd import _lib
    from tracktolearn_amd.environments import TrackingEnvironment
    monkeypatch.setenv('TTL_TAIL_FUSED', tail_fused)
    if tail_fused == '0':       # the Python mirror of the knob is read at import
        monkeypatch.setattr(TrackingEnvironment, 'TAIL_FUSED_MAX_ROWS', 0)
    monkeypatch.setattr(TrackingEnvironment, 'SPATIAL_ORDER_MIN', 1)
    monkeypatch.setattr(TrackingEnvironment, 'SPATIAL_ORDER_REFRESH', 3)
    D = 20
    sh, mask, pk = synthetic_subject(D)
    rng = np.random.RandomState(11)
    vox = np.argwhere(mask)
    seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
    env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds, n_dirs=4,
                   max_length=25.0, reward=False)
    lib = _lib.load()
    lst, n_stop = C.c_void_p(), C.c_int32()
    state = env.reset(0, N)
    assert lib.ttl_env_stopped(env._handle, C.byref(lst), C.byref(n_stop)) == _lib.ERR_STATE
    step, seen = 0, 0
    while env._n_active:
        n = env._n_active
        idx = env.continue_idx.copy()
        _, _, done, _ = env.step_device(env.scripted_actions(state, step, 9, 0.25))
        _lib.check(lib.ttl_env_stopped(env._handle, C.byref(lst), C.byref(n_stop)), 'stopped')
        d = done.cpu().numpy().astype(bool)
        assert n_stop.value == int(d.sum())
        if n_stop.value:
            # the list lives in the workspace the env handed to the library
            off = lst.value - env._buf_ws.data_ptr()
            got = env._buf_ws[off:off + 8 * n_stop.value].view(torch.int32).cpu().numpy() \
                .reshape(-1, 2)
            assert np.array_equal(got[:, 0], np.nonzero(d)[0])
            assert np.array_equal(got[:, 1], idx[d])
            seen += n_stop.value
        # asking twice is harmless, and the harvest still finds its counts
        _lib.check(lib.ttl_env_stopped(env._handle, C.byref(lst), C.byref(n_stop)), 'stopped')
        state, _ = env.harvest()
        assert lib.ttl_env_stopped(env._handle, C.byref(lst), C.byref(n_stop)) == _lib.ERR_STATE
        step += 1
    assert seen == N


def test_episode_on_a_non_default_stream():
    """Everything is ordered on the caller's stream (torch's current stream):
    a whole device-resident episode of 20 000 streamlines issued inside a
    `torch.cuda.stream(...)` block, with the sorted processing order on,
    tracks exactly what the oracle tracks."""
    from oracle import env_oracle as orc
    from tracktolearn_amd.environments import TrackingEnvironment
    saved = TrackingEnvironment.SPATIAL_ORDER_MIN
    saved_refresh = TrackingEnvironment.SPATIAL_ORDER_REFRESH
    TrackingEnvironment.SPATIAL_ORDER_MIN = 1
    TrackingEnvironment.SPATIAL_ORDER_REFRESH = 3
    try:
        D, N = 20, 20000
        sh, mask, pk = synthetic_subject(D)
        rng = np.random.RandomState(5)
        vox = np.argwhere(mask)
        seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds,
                           n_dirs=4, max_length=25.0, reward=False)
            state = env.reset(0, N)
            recorded = []
            step = 0
            while env._n_active:
                a = env.scripted_actions(state, step, 9, 0.2)
                recorded.append((a.clone(), env._n_active))
                env.step_device(a)
                state, _ = env.harvest()
                step += 1
            flags, lengths, lines = env.flags, env.lengths, env.streamlines
        ref = orc.OracleTrackingEnv(sh, mask, seeds, n_dirs=4, theta=30.0,
                                    step_size=env.step_size, max_nb_steps=env.max_nb_steps,
                                    mask_threshold=0.1, peaks=pk, compute_reward=False,
                                    alignment_weighting=1.0)
        ref.reset(0, N)
        for a, n in recorded:
            assert len(ref.continue_idx) == n
            ref.step(a.cpu().numpy())
            ref.harvest()
        assert len(ref.continue_idx) == 0
        assert np.array_equal(flags, ref.flags)
        assert np.array_equal(lengths, ref.lengths)
        assert np.array_equal(lines, ref.streamlines)
    finally:
        TrackingEnvironment.SPATIAL_ORDER_MIN = saved
        TrackingEnvironment.SPATIAL_ORDER_REFRESH = saved_refresh


@pytest.mark.parametrize('fused,sorter', [('0', '1'), ('1', '1'), ('0', '0')])
def test_caller_supplied_processing_orders_change_nothing(fused, sorter, monkeypatch):
    """ttl_env_set_processing_order with arbitrary permutations between steps
    (and the library's own refresh in between -- counting sort or rocPRIM):
    a scheduling hint only.  With the fused tail on, a batch this small drops
    the order again at its next step."""
    monkeypatch.setenv('TTL_FUSE_SMALL', fused)
    monkeypatch.setenv('TTL_ORDER_SORT', sorter)
    from oracle import env_oracle as orc
    from tracktolearn_amd import _lib
    from tracktolearn_amd.environments import TrackingEnvironment
    saved = (TrackingEnvironment.SPATIAL_ORDER_MIN, TrackingEnvironment.SPATIAL_ORDER_REFRESH)
    TrackingEnvironment.SPATIAL_ORDER_MIN, TrackingEnvironment.SPATIAL_ORDER_REFRESH = 1, 3
    try:
        D, N = 16, 9000
        sh, mask, pk = synthetic_subject(D)
        rng = np.random.RandomState(8)
        vox = np.argwhere(mask)
        seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
        env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds, n_dirs=4,
                       max_length=20.0, reward=False)
        ref = orc.OracleTrackingEnv(sh, mask, seeds, n_dirs=4, theta=30.0,
                                    step_size=env.step_size, max_nb_steps=env.max_nb_steps,
                                    mask_threshold=0.1, peaks=pk, compute_reward=False,
                                    alignment_weighting=1.0)
        s_hip, s_ref = env.reset(0, N), ref.reset(0, N)
        step = 0
        while len(ref.continue_idx):
            n = env._n_active
            if step % 2:
                perm = torch.randperm(n, device='cuda').to(torch.int32)
                _lib.check(env._lib.ttl_env_set_processing_order(
                    env._handle, perm.data_ptr(), n, env._stream()), 'set order')
                # wrong sizes are refused, nothing is launched for them
                assert env._lib.ttl_env_set_processing_order(
                    env._handle, perm.data_ptr(), n + 1, env._stream()) == -1
            a = _scripted(rng, s_ref, 7 * 45, step, 0.15)
            ns_hip, _, d_hip, _ = env.step(a.copy())
            ns_ref, _, d_ref, _ = ref.step(a.copy())
            assert np.array_equal(d_hip, d_ref)
            assert _close(ns_hip.cpu().numpy(), ns_ref)
            s_hip, _ = env.harvest()
            s_ref, _ = ref.harvest()
            assert _close(s_hip.cpu().numpy(), s_ref)
            step += 1
        assert step > 5
        assert np.array_equal(env.streamlines, ref.streamlines)
    finally:
        TrackingEnvironment.SPATIAL_ORDER_MIN, TrackingEnvironment.SPATIAL_ORDER_REFRESH = saved


@pytest.mark.parametrize('tail,refresh,local_sort', [('1', 0, '1'), ('1', 4, '1'), ('1', 0, '0'),
                                                     ('0', 4, '1')])
def test_uncompacted_processing_order_changes_nothing(tail, refresh, local_sort, monkeypatch):
    """Round 3: with the fused step tail (k_tail, TTL_TAIL_FUSED=1) the processing
    order of a large batch is not compacted between refreshes -- stopped
    streamlines leave holes that the per-block re-sort moves to the end of their
    256-slot block and the gather skips.  A batch that is not a multiple of 256,
    never refreshed (the holes only grow) or refreshed early once a fifth of
    the slots are holes, with and without the re-sort, against the two-kernel
    tail: every step equals the oracle's, the tractogram is bit-identical."""
    monkeypatch.setenv('TTL_TAIL_FUSED', tail)
    monkeypatch.setenv('TTL_TAIL_FUSED_MAX_ROWS', '1048576')
    monkeypatch.setenv('TTL_LOCAL_SORT', local_sort)
    from oracle import env_oracle as orc
    from tracktolearn_amd.environments import TrackingEnvironment
    monkeypatch.setattr(TrackingEnvironment, 'TAIL_FUSED_MAX_ROWS', 1048576 if tail == '1' else 0)
    saved = (TrackingEnvironment.SPATIAL_ORDER_MIN, TrackingEnvironment.SPATIAL_ORDER_REFRESH)
    TrackingEnvironment.SPATIAL_ORDER_MIN, TrackingEnvironment.SPATIAL_ORDER_REFRESH = 1, refresh
    try:
        D, N = 20, 40000 + 77
        sh, mask, pk = synthetic_subject(D)
        rng = np.random.RandomState(11)
        vox = np.argwhere(mask)
        seeds = vox[rng.randint(0, len(vox), N)] + rng.uniform(-0.5, 0.5, (N, 3))
        env = _hip_env(D, noisy=False, affine_dtype=np.float32, seeds=seeds, n_dirs=4,
                       max_length=24.0, reward=False)
        ref = orc.OracleTrackingEnv(sh, mask, seeds, n_dirs=4, theta=30.0,
                                    step_size=env.step_size, max_nb_steps=env.max_nb_steps,
                                    mask_threshold=0.1, peaks=pk, compute_reward=False,
                                    alignment_weighting=1.0)
        s_hip, s_ref = env.reset(0, N), ref.reset(0, N)
        step, large_steps = 0, 0
        while len(ref.continue_idx):
            large_steps += env._n_active >= 16384 + 1
            a = _scripted(rng, s_ref, 7 * 45, step, 0.1)
            if step % 2:
                ns_hip, _, d_hip, _ = env.step(a.copy())
                ns = ns_hip.cpu().numpy()
            else:
                ns_hip, _, d_dev, info = env.step_device(torch.from_numpy(a).cuda())
                ns = ns_hip.cpu().numpy()[info['row_dest'].cpu().numpy()]
                d_hip = d_dev.cpu().numpy().astype(bool)
            ns_ref, _, d_ref, _ = ref.step(a.copy())
            assert np.array_equal(d_hip, d_ref), step
            assert _close(ns, ns_ref), step
            s_hip, _ = env.harvest()
            s_ref, _ = ref.harvest()
            assert np.array_equal(env.continue_idx, ref.continue_idx)
            assert _close(s_hip.cpu().numpy(), s_ref), step
            step += 1
        assert large_steps >= 4          # the large-batch tail ran with holes in its order
        assert np.array_equal(env.flags, ref.flags)
        assert np.array_equal(env.lengths, ref.lengths)
        assert np.array_equal(env.streamlines, ref.streamlines)
    finally:
        TrackingEnvironment.SPATIAL_ORDER_MIN, TrackingEnvironment.SPATIAL_ORDER_REFRESH = saved
