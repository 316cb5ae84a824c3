"""The CPU oracle replays every golden trace captured from the reference
(tests/golden/make_golden.py) bit for bit, and matches the isolated-function
vectors.  CPU only."""
import numpy as np
import pytest

from oracle import env_oracle as orc
from helpers import (TRACES, load_trace, synthetic_subject, trace_noise,
                     trace_step_size)


def _make_env(z, spline_eval):
    D = int(z['D'])
    sh, mask, pk = synthetic_subject(D)
    kw = dict(n_dirs=int(z['n_dirs']), theta=float(z['theta']),
              step_size=trace_step_size(z), max_nb_steps=int(z['max_nb_steps']),
              mask_threshold=float(z['mask_threshold']), peaks=pk,
              compute_reward=bool(z['reward']), alignment_weighting=1.0,
              spline_eval=spline_eval)
    if bool(z['noisy']):
        sigma, rs = trace_noise(z)
        return orc.OracleNoisyTrackingEnv(sh, mask, z['seeds'], noise=sigma, rng=rs, **kw)
    return orc.OracleTrackingEnv(sh, mask, z['seeds'], **kw)


@pytest.mark.parametrize('spline_eval', ['restated', 'scipy'])
@pytest.mark.parametrize('name', TRACES)
def test_oracle_replays_reference_trace(name, spline_eval):
    z = load_trace(name)
    env = _make_env(z, spline_eval)
    if 'mask_coef' in z.files:
        assert np.array_equal(env.coef, z['mask_coef'])
    N = z['seeds'].shape[0]
    state = env.reset(0, N)
    assert np.array_equal(state, z['state_reset'])
    for s in range(int(z['n_steps'])):
        assert np.array_equal(env.continue_idx, z[f'continue_idx_{s}'])
        nstate, rew, done, info = env.step(z[f'actions_{s}'].copy())
        assert np.array_equal(done, z[f'dones_{s}'])
        assert np.array_equal(env.new_continue_idx, z[f'new_continue_idx_{s}'])
        assert np.array_equal(env.stopping_idx, z[f'stopping_idx_{s}'])
        assert np.array_equal(env.flags, z[f'flags_{s}'])
        assert rew.dtype == z[f'reward_{s}'].dtype
        assert np.array_equal(rew, z[f'reward_{s}'])
        if bool(z['reward']):
            assert info['reward_info']['peaks_reward'] == z[f'reward_info_peaks_{s}']
        idx = z[f'continue_idx_{s}']
        assert np.array_equal(env.streamlines[idx, env.length - 1], z[f'head_{s}'])
        if f'state_{s}' in z.files:
            assert np.array_equal(nstate, z[f'state_{s}'])
        assert np.array_equal(nstate.astype(np.float64).sum(axis=1),
                              z[f'state_rowsum_{s}'])
        state, _ = env.harvest()
        assert state.shape[0] == int(z[f'harvest_rows_{s}'])
        assert np.array_equal(env.lengths, z[f'lengths_{s}'])
    assert len(env.continue_idx) == 0
    assert np.array_equal(env.streamlines, z['streamlines'])
    lines, seeds, flags = env.get_streamlines()
    assert np.array_equal(np.array([len(s) for s in lines]), z['tract_lengths'])
    assert np.array_equal(np.concatenate(lines), z['tract_points'])
    assert np.array_equal(flags, z['tract_flags'])
    assert np.array_equal(seeds, z['tract_seeds'])


def test_isolated_vectors():
    z = load_trace('isolated_functions')
    with np.errstate(all='ignore'):
        a = z['norm_in']
        assert np.array_equal(orc.unit_rows(a), z['norm_f32'], equal_nan=True)
        assert np.array_equal(orc.unit_rows(a.astype(np.float64)), z['norm_f64'],
                              equal_nan=True)
        assert np.array_equal(orc.scale_actions(a, np.float32(0.75)),
                              z['scaled_f32'], equal_nan=True)
        assert np.array_equal(
            orc.scale_actions(a + np.zeros(a.shape), np.float64(0.75)),
            z['scaled_f64'], equal_nan=True)
        tri = z['curvy_in']
        got = orc.stop_too_curvy(tri[:, 2], tri[:, 1], tri[:, 0],
                                 float(z['curvy_theta']))
        assert np.array_equal(got, z['curvy_out'])
        assert got.any() and (~got).any()
    # mask: prefilter, restated per-point values and decisions, all bit-exact
    coef = orc.prefilter_mask(z['mask_in'])
    assert np.array_equal(coef, z['mask_coef'])
    pts = z['mask_pts']
    vals = orc.spline3_sample(coef, pts - 0.5)
    assert np.array_equal(vals, z['mask_values'])
    assert np.array_equal(orc.stop_outside_mask(coef, pts, 0.1), z['mask_stop'])
    assert np.array_equal(orc.stop_outside_mask_scipy(coef, pts, 0.1), z['mask_stop'])
    # reward
    s3 = z['reward_in']
    pk = z['reward_peaks']
    with np.errstate(all='ignore'):
        r3 = orc.peaks_alignment_reward(pk, s3[:, 2], s3[:, 1], s3[:, 0])
        r2 = orc.peaks_alignment_reward(pk, s3[:, 2], s3[:, 1], None)
    assert r3.dtype == z['reward_L3'].dtype
    assert np.array_equal(r3, z['reward_L3'])
    assert np.array_equal(r2, z['reward_L2'])
    assert z['reward_L1'].dtype == np.uint8 and (z['reward_L1'] == 1).all()


def test_trilinear_matches_scipy_linear_nearest():
    """Independent cross-check of the (unpinned) dwi_ml restatement: same
    function as scipy order-1 interpolation with edge replication."""
    from scipy.ndimage import map_coordinates
    rng = np.random.RandomState(3)
    D, C = 9, 6
    vol = rng.standard_normal((D, D, D, C)).astype(np.float32)
    pts = rng.uniform(-2.5, D + 1.5, (3000, 3)).astype(np.float32)
    neigh = orc.neighborhood_offsets(np.float32(0.75))
    got = orc.trilinear_neighborhood(vol, pts, neigh).reshape(-1, 7, C)
    for p in range(7):
        q = (pts + neigh[p]).astype(np.float64)
        for c in range(C):
            want = map_coordinates(vol[..., c].astype(np.float64), q.T, order=1,
                                   mode='nearest')
            assert np.abs(got[:, p, c] - want).max() < 5e-6
