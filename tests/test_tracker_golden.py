"""The Tracker against output recorded from the reference's own
`Tracker.track` / `track_and_validate` (tests/golden/make_golden_tracker.py;
TrackToLearn/tracking/tracker.py:62-150, 204-259): seed shuffle and batching,
`get_streamlines`, the length filter in voxel units, `.trk` `(s + 0.5) *
vox_size`, `.tck` `s @ A[:3, :3] + A[:3, 3]` with a rotated (non-symmetric)
affine, saved seeds `seed - 0.5`.

CPU: the oracle replays the recorded actions and the host-side conversion
helpers reproduce the reference's output from the reference's own voxel-space
streamlines.  GPU: `tracktolearn_amd.tracking.tracker.Tracker` end to end.
"""
import numpy as np
import pytest

from helpers import load_trace, synthetic_subject


class ReplayAgent:
    """Policy stand-in that returns the recorded action batches in order."""

    def __init__(self, z, device=None):
        counts = z['action_counts']
        offs = np.concatenate(([0], np.cumsum(counts)))
        self.batches = [z['actions'][offs[i]:offs[i + 1]] for i in range(len(counts))]
        self.device, self.i = device, 0

    def eval(self):
        pass

    def select_action(self, state, probabilistic=0.0):
        import torch
        a = self.batches[self.i]
        self.i += 1
        assert state.shape[0] == len(a), 'survivor count differs from the reference'
        t = torch.from_numpy(a)
        return t.to(self.device) if self.device is not None else t


def _split(points, lengths):
    offs = np.concatenate(([0], np.cumsum(lengths)))
    return [points[offs[i]:offs[i + 1]] for i in range(len(lengths))]


def _oracle_batches(z, noisy, reward):
    """Voxel-space streamlines + flags of every seed batch from the oracle fed
    with the recorded actions."""
    from oracle import env_oracle as orc
    D = int(z['D'])
    sh, mask, pk = synthetic_subject(D)
    seeds = z['seeds_after_shuffle'] if 'seeds_after_shuffle' in z.files else z['seeds']
    step = np.float64(z['step_size'])
    kw = dict(n_dirs=int(z['n_dirs']), theta=30.0, step_size=step,
              max_nb_steps=int(z['max_nb_steps']), mask_threshold=0.1, peaks=pk,
              compute_reward=reward, alignment_weighting=1.0)
    env = (orc.OracleNoisyTrackingEnv(sh, mask, seeds, noise=0.0, **kw) if noisy
           else orc.OracleTrackingEnv(sh, mask, seeds, **kw))
    agent = ReplayAgent(z)
    n_actor = int(z['n_actor'])
    lines, flags, total = [], [], 0.0
    for start in range(0, len(seeds), n_actor):
        state = env.reset(start, min(start + n_actor, len(seeds)))
        while len(env.continue_idx):
            a = agent.select_action(state).numpy()
            _, r, _, _ = env.step(a)
            total += sum(r)
            state, _ = env.harvest()
        ls, _, fl = env.get_streamlines()
        lines += ls
        flags.append(fl)
    assert agent.i == len(agent.batches)
    return lines, np.concatenate(flags), total


@pytest.mark.parametrize('name', ['tracker_trk', 'tracker_tck', 'tracker_trk_compress'])
def test_host_side_of_track_matches_the_reference(name):
    from tracktolearn_amd.tracking.tracker import (TckFile, TrkFile,
                                                   to_file_space)
    from tracktolearn_amd.tractogram import compress_streamline
    z = load_trace(name)
    # the shuffle: numpy's global generator, as tracker.py:94
    seeds = z['seeds_before_shuffle'].copy()
    np.random.seed(int(z['shuffle_seed']))
    np.random.shuffle(seeds)
    assert np.array_equal(seeds, z['seeds_after_shuffle'])
    # the oracle reproduces the reference's voxel-space batches bit for bit
    lines, flags, _ = _oracle_batches(z, noisy=True, reward=False)
    want_vox = _split(z['vox_points'], z['vox_lengths'])
    assert np.array_equal(flags, z['vox_flags'])
    assert len(lines) == len(want_vox)
    for got, want in zip(lines, want_vox):
        assert np.array_equal(got, want)
    # filter + conversion, from the reference's own voxel-space streamlines
    aff = z['affine']
    vox_size = np.mean(np.abs(aff)[np.diag_indices(4)][:3])
    lo, hi = float(z['min_length']) / vox_size, float(z['max_length']) / vox_size
    fmt = TckFile if name == 'tracker_tck' else TrkFile
    out, out_seeds = [], []
    for s, seed in zip(want_vox, z['seeds_after_shuffle']):
        d = (s[1:] - s[:-1]).astype(np.float64)
        arc = np.sqrt((d * d).sum(axis=1)).sum() if len(s) > 1 else 0.0
        if not lo <= arc <= hi:
            continue
        if float(z['compress']):
            s = compress_streamline(s, float(z['compress']) / vox_size)
        out.append(to_file_space(s, fmt, aff, vox_size))
        out_seeds.append(seed - 0.5)
    assert np.array_equal([len(s) for s in out], z['out_lengths'])
    got = np.concatenate(out)
    assert str(got.dtype) == str(z['out_dtype'])
    assert np.array_equal(got, z['out_points'])            # bit for bit
    assert np.array_equal(np.stack(out_seeds), z['out_seeds'])
    if name == 'tracker_tck':
        # pins `s @ A` (row vector times matrix) against the textbook `A @ s`
        other = np.concatenate([s @ aff[:3, :3].T + aff[:3, 3] for s in
                                [w for w in want_vox][:5]])
        assert not np.allclose(other, np.concatenate(
            [to_file_space(w, fmt, aff, vox_size) for w in want_vox[:5]]))


def test_oracle_replays_track_and_validate():
    z = load_trace('tracker_validate')
    lines, flags, total = _oracle_batches(z, noisy=False, reward=True)
    assert np.array_equal(flags, z['flags'])
    for got, want in zip(lines, _split(z['points'], z['lengths'])):
        assert np.array_equal(got, want)
    assert abs(total - float(z['reward'])) <= 1e-9 * abs(float(z['reward']))


# --------------------------------------------------------------------------
def _gpu_env(z, noisy, reward):
    import torch
    from tracktolearn_amd.datasets.utils import MRIDataVolume as Vol
    from tracktolearn_amd.environments import (NoisyTrackingEnvironment,
                                               TrackingEnvironment)
    sh, mask, pk = synthetic_subject(int(z['D']))
    aff = z['affine']
    dto = dict(n_dirs=int(z['n_dirs']), theta=30.0, npv=1,
               binary_stopping_threshold=0.1, step_size=0.75, min_length=2.0,
               max_length=40.0, compute_reward=reward, alignment_weighting=1.0,
               oracle_bonus=0.0, rng=np.random.RandomState(3),
               device=torch.device('cuda:0'), target_sh_order=8, noise=0.0,
               fa_map=None)
    cls = NoisyTrackingEnvironment if noisy else TrackingEnvironment
    env = cls((Vol(sh, aff), Vol(mask.astype(np.float32), aff),
               Vol(mask.astype(np.float32), aff), Vol(pk, aff), None),
              'testing', dto)
    assert float(env.step_size) == float(z['step_size'])
    assert env.max_nb_steps == int(z['max_nb_steps'])
    return env


class _Alg:
    """RLAlgorithm with the replay agent as its policy."""

    def __init__(self, agent):
        from tracktolearn_amd.algorithms.rl import RLAlgorithm
        self.agent = agent
        self.validation_episode = RLAlgorithm.validation_episode.__get__(self)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['tracker_trk', 'tracker_tck', 'tracker_trk_compress'])
def test_tracker_track_matches_the_reference_on_the_gpu(name):
    """Bit-exact voxel-space points (checked through the .trk / .tck output,
    whose conversions are exact functions of them: `.trk` compared bit for
    bit, `.tck` -- float64 -- to 1e-5), same kept streamlines, same order,
    same saved seeds."""
    import torch
    from tracktolearn_amd.tracking.tracker import TckFile, Tracker, TrkFile
    z = load_trace(name)
    env = _gpu_env(z, noisy=True, reward=False)
    env.seeds = z['seeds_before_shuffle'].copy()
    agent = ReplayAgent(z, torch.device('cuda:0'))
    tracker = Tracker(_Alg(agent), n_actor=int(z['n_actor']), prob=0.0,
                      compress=float(z['compress']),
                      min_length=float(z['min_length']),
                      max_length=float(z['max_length']), save_seeds=True)
    np.random.seed(int(z['shuffle_seed']))
    fmt = TckFile if name == 'tracker_tck' else TrkFile
    lazy = tracker.track(env, fmt)
    items = list(lazy)
    assert agent.i == len(agent.batches)
    assert np.array_equal(env.seeds, z['seeds_after_shuffle'])
    assert np.array_equal(lazy.affine_to_rasmm, z['affine'])
    assert np.array_equal([len(it.streamline) for it in items], z['out_lengths'])
    got = np.concatenate([it.streamline for it in items])
    if fmt is TrkFile:
        assert got.dtype == np.float32
        assert np.array_equal(got, z['out_points'])
    else:
        assert got.dtype == np.float64
        assert np.abs(got - z['out_points']).max() <= 1e-5
        assert np.array_equal(got, z['out_points'])       # in fact identical
    seeds = np.stack([it.data_for_streamline['seeds'] for it in items])
    assert np.array_equal(seeds, z['out_seeds'])


@pytest.mark.gpu
def test_tracker_track_and_validate_matches_the_reference_on_the_gpu():
    import torch
    from tracktolearn_amd.tracking.tracker import Tracker
    z = load_trace('tracker_validate')
    env = _gpu_env(z, noisy=False, reward=True)
    env.seeds = z['seeds'].copy()
    agent = ReplayAgent(z, torch.device('cuda:0'))
    tracker = Tracker(_Alg(agent), n_actor=int(z['n_actor']), prob=0.0)
    tg, reward = tracker.track_and_validate(env)
    assert agent.i == len(agent.batches)
    assert len(tg) == len(z['lengths'])
    assert np.array_equal(tg.data_per_streamline['flags'], z['flags'])
    assert np.array_equal(tg.data_per_streamline['seeds'], z['tg_seeds'])
    for got, want in zip(tg.streamlines, _split(z['points'], z['lengths'])):
        assert np.array_equal(got, want)                   # bit-exact points
    assert abs(reward - float(z['reward'])) <= 1e-5 * abs(float(z['reward']))
