#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own environment
classes, imported read-only from /root/reference in the build container.

Run:  python tests/golden/make_golden.py         (only where /root/reference
exists; the GPU box never sees the reference, only the .npz files travel).
Every trace seeds its own seed stream (`seed_stream`), so each fixture is
reproduced byte for byte by this script whatever traces are added or removed
around it; tests/test_golden_reproducible.py checks that for all four
generators wherever the reference tree is present.

How the reference is made importable: its third-party
imports that are absent from this image (nibabel, dipy, dwi_ml, scilpy, h5py,
comet_ml) are replaced by inert ``MagicMock`` modules.  Three callables the
step path really executes get working bodies:

  * ``dwi_ml ... interpolate_volume_in_neighborhood`` /
    ``get_neighborhood_vectors_axes`` -- third-party arithmetic whose source is
    not under /root/reference.  The body used here is this repo's own
    restatement (``oracle.env_oracle.trilinear_neighborhood``), so the SH part
    of the recorded ``state`` is NOT evidence about dwi_ml: that piece stays
    "parity unpinned" (DESIGN.md).  Everything else in the fixtures --
    action scaling, first-step flip, position update, the three stopping
    tests, flags, index compaction, lengths, previous-direction block of the
    state, alignment reward, get_streamlines truncation -- is computed by the
    reference's own, unmodified code.
  * ``dipy.tracking.utils.random_seeds_from_mask`` -- seeds are inputs.
  * ``nibabel.streamlines.Tractogram`` -- a plain holder.

The fixtures record numpy/scipy/torch versions and the dtype mode (SURVEY F7/F8,
App. D): they encode numpy >= 2 promotion semantics.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import scipy
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REFERENCE = '/root/reference'

from oracle import env_oracle as orc  # noqa: E402  (only for the dwi_ml body)

_ABSENT = """nibabel nibabel.streamlines nibabel.streamlines.tractogram
nibabel.nifti1 dipy dipy.core dipy.core.sphere dipy.core.geometry dipy.data
dipy.direction dipy.direction.peaks dipy.tracking dipy.tracking.utils
dipy.tracking.metrics dipy.tracking.streamline dipy.tracking.streamlinespeed
dipy.io dipy.io.stateful_tractogram dipy.io.streamline dipy.io.utils
dipy.reconst dipy.reconst.shm dipy.reconst.csdeconv dwi_ml dwi_ml.data
dwi_ml.data.processing dwi_ml.data.processing.volume
dwi_ml.data.processing.volume.interpolation dwi_ml.data.processing.space
dwi_ml.data.processing.space.neighborhood scilpy scilpy.reconst
scilpy.reconst.utils scilpy.reconst.sh scilpy.io scilpy.io.utils
scilpy.tracking scilpy.tracking.utils h5py comet_ml""".split()


class TractogramHolder:
    def __init__(self, streamlines=None, data_per_streamline=None, **kw):
        self.streamlines = streamlines
        self.data_per_streamline = data_per_streamline or {}


_SEED_RNG = np.random.RandomState(4321)


def _seeds_from_mask(mask, affine, seeds_count=1, **kw):
    vox = np.argwhere(mask)
    vox = np.repeat(vox, seeds_count, axis=0)
    return vox + _SEED_RNG.uniform(size=vox.shape) - 0.5


def _interp(volume, coords, neigh, *a, **kw):
    out = orc.trilinear_neighborhood(
        volume.cpu().numpy(), coords.cpu().numpy(), neigh.cpu().numpy())
    return torch.from_numpy(out), None


def _neigh_axes(radius, step):
    eye = torch.eye(3)
    return torch.cat((eye, -eye)) * float(step) * float(radius)


def import_reference():
    sys.path.insert(0, REFERENCE)
    for name in _ABSENT:
        sys.modules[name] = MagicMock()
    sys.modules['nibabel.streamlines'].Tractogram = TractogramHolder
    sys.modules['dipy.io.stateful_tractogram'].Tractogram = TractogramHolder
    sys.modules['dipy.tracking.utils'].random_seeds_from_mask = _seeds_from_mask
    sys.modules['dipy.tracking'].utils = sys.modules['dipy.tracking.utils']
    sys.modules['dwi_ml.data.processing.volume.interpolation'] \
        .interpolate_volume_in_neighborhood = _interp
    sys.modules['dwi_ml.data.processing.space.neighborhood'] \
        .get_neighborhood_vectors_axes = _neigh_axes
    from TrackToLearn.datasets.utils import MRIDataVolume
    from TrackToLearn.environments.tracking_env import TrackingEnvironment
    from TrackToLearn.environments.noisy_tracking_env import \
        NoisyTrackingEnvironment
    import TrackToLearn.environments.utils as env_utils
    import TrackToLearn.environments.stopping_criteria as sc
    import TrackToLearn.environments.local_reward as lr
    import TrackToLearn.utils.utils as uu
    return dict(MRIDataVolume=MRIDataVolume,
                TrackingEnvironment=TrackingEnvironment,
                NoisyTrackingEnvironment=NoisyTrackingEnvironment,
                env_utils=env_utils, sc=sc, lr=lr, uu=uu)


# --------------------------------------------------------------------------
# synthetic subject (SURVEY 8d recipe, small)
# --------------------------------------------------------------------------
def synthetic_subject(D, C=45, seed=1234, peaks=True):
    rng = np.random.RandomState(seed)
    sh = (0.1 * rng.standard_normal((D, D, D, C))).astype(np.float32)
    sh[..., 0] = 1.0
    g = np.indices((D, D, D)).astype(np.float64)
    r = np.sqrt(((g - (D - 1) / 2.0) ** 2).sum(0))
    mask = (r < 0.42 * D).astype(np.uint8)
    pk = rng.standard_normal((D, D, D, 15)).astype(np.float32) if peaks else None
    if pk is not None:
        pk[0, 0, 0] = 0.0            # a zero-peak voxel (NaN -> 0 path)
        pk[D // 2, D // 2, D // 2, 3:6] = 0.0
    return sh, mask, pk


def scripted_actions(rng, state, n_sh, step, wobble):
    """Policy stand-in: random first direction, then previous direction (read
    from the state's direction block) plus gaussian wobble."""
    n = state.shape[0]
    if step == 0:
        a = rng.standard_normal((n, 3))
    else:
        prev = state[:, n_sh:n_sh + 3].astype(np.float64)
        nrm = np.linalg.norm(prev, axis=1, keepdims=True)
        nrm[nrm == 0] = 1.0
        a = prev / nrm + wobble * rng.standard_normal((n, 3))
    return a.astype(np.float32)


def run_trace(ref, name, *, D, N, K, noisy, affine_dtype, reward, max_length,
              wobble, theta=30.0, npv=2, seed=7, state_every=1, aim_centre=False,
              state_steps=(), keep_history=True, voxel=1.0, origin=(0.0, 0.0, 0.0),
              seed_stream=None, noise=0.0, thr=0.1):
    if seed_stream is not None:     # a trace that can be (re)generated on its own
        _SEED_RNG.seed(seed_stream)
    sh, mask, pk = synthetic_subject(D)
    aff = np.eye(4, dtype=affine_dtype)
    aff[0, 0] = aff[1, 1] = aff[2, 2] = voxel
    aff[:3, 3] = origin
    Vol = ref['MRIDataVolume']
    subject = (Vol(sh, aff), Vol(mask.astype(np.float32), aff),
               Vol(mask.astype(np.float32), aff),
               Vol(pk, aff) if pk is not None else None, None)
    dto = dict(dataset_file=None, fa_map=None, n_dirs=K, step_size=0.75,
               theta=theta, min_length=2.0, max_length=max_length, noise=noise,
               npv=npv, rng=np.random.RandomState(seed),
               alignment_weighting=1.0, oracle_bonus=0.0,
               oracle_validator=False, oracle_stopping_criterion=False,
               oracle_checkpoint=None, scoring_data=None,
               tractometer_validator=False, binary_stopping_threshold=thr,
               compute_reward=reward, device=torch.device('cpu'),
               target_sh_order=8)
    cls = ref['NoisyTrackingEnvironment' if noisy else 'TrackingEnvironment']
    env = cls(subject, 'testing', dto)
    n_sh = 7 * sh.shape[-1]
    rng = np.random.RandomState(seed + 1)
    pick = rng.permutation(len(env.seeds))[:N]
    env.seeds = env.seeds[pick]
    out = dict(sh_seed=1234, D=D, C=sh.shape[-1], n_dirs=K, theta=theta,
               step_size=np.asarray(env.step_size),
               step_size_dtype=str(np.asarray(env.step_size).dtype),
               max_nb_steps=env.max_nb_steps, mask_threshold=thr,
               noisy=noisy, reward=reward, seeds=env.seeds.copy(),
               alignment_weighting=1.0, affine=aff.copy(),
               mask_coef=env.stopping_criteria[
                   ref['sc'].StoppingFlags.STOPPING_MASK].mask)
    if D >= 64:      # 7 MB of float64: the smaller traces already pin it
        del out['mask_coef']
    state = env.reset(0, N)
    out['state_reset'] = state.numpy().copy()
    out['noise'] = np.float64(noise)
    if noise > 0:       # the generator's state when the first step draws from it
        st = env.rng.get_state()
        out['rng_key'], out['rng_pos'] = st[1].copy(), np.int64(st[2])
        out['rng_has_gauss'], out['rng_cached'] = np.int64(st[3]), np.float64(st[4])
    step = 0
    done = False
    while not np.all(done):
        a = scripted_actions(rng, state.numpy(), n_sh, step, wobble)
        if aim_centre and step == 0:
            # head for the centre of the ball: near-diametral, long chords
            here = env.streamlines[env.continue_idx, 0].astype(np.float64)
            a = ((D - 1) / 2.0 - here + 0.3 * rng.standard_normal(here.shape)
                 ).astype(np.float32)
        idx_before = env.continue_idx.copy()
        next_state, rew, done, info = env.step(a.copy())
        out[f'actions_{step}'] = a
        out[f'continue_idx_{step}'] = idx_before
        out[f'dones_{step}'] = np.asarray(done).copy()
        out[f'reward_{step}'] = np.asarray(rew).copy()
        if reward:
            out[f'reward_info_peaks_{step}'] = np.float64(
                info['reward_info']['peaks_reward'])
        out[f'new_continue_idx_{step}'] = env.new_continue_idx.copy()
        out[f'stopping_idx_{step}'] = env.stopping_idx.copy()
        out[f'flags_{step}'] = env.flags.copy()
        out[f'head_{step}'] = env.streamlines[idx_before, env.length - 1].copy()
        ns = next_state.numpy()
        if step % state_every == 0 or step < 3 or step in state_steps:
            out[f'state_{step}'] = ns.copy()
        out[f'state_rowsum_{step}'] = ns.astype(np.float64).sum(axis=1)
        state, not_stopping = env.harvest()
        out[f'harvest_rows_{step}'] = np.int64(state.shape[0])
        out[f'lengths_{step}'] = env.lengths.copy()
        step += 1
    out['n_steps'] = step
    if keep_history:
        out['streamlines'] = env.streamlines.copy()
    else:       # long runs: a float64 checksum per streamline instead
        out['streamlines_rowsum'] = env.streamlines.astype(np.float64).sum(axis=(1, 2))
    tg = env.get_streamlines()
    out['tract_lengths'] = np.array([len(s) for s in tg.streamlines], np.int64)
    out['tract_points'] = np.concatenate(
        [np.asarray(s, np.float32).reshape(-1, 3) for s in tg.streamlines])
    out['tract_flags'] = np.asarray(tg.data_per_streamline['flags'])
    out['tract_seeds'] = np.asarray(tg.data_per_streamline['seeds'])
    _save(name, out)
    flags = env.flags
    print(f'{name}: steps={step} flags mask={np.sum(flags & 1 > 0)} '
          f'length={np.sum(flags & 2 > 0)} curv={np.sum(flags & 4 > 0)}')


def _save(name, out):
    out['versions'] = np.array(
        [f'numpy {np.__version__}', f'scipy {scipy.__version__}',
         f'torch {torch.__version__}'])
    # TTL_GOLDEN_OUT: write somewhere else (tests/test_golden_reproducible.py
    # regenerates every fixture into a scratch directory and compares)
    path = os.path.join(os.environ.get('TTL_GOLDEN_OUT') or HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


# --------------------------------------------------------------------------
# isolated function vectors
# --------------------------------------------------------------------------
def isolated(ref):
    rng = np.random.RandomState(99)
    uu, eu, sc, lr = ref['uu'], ref['env_utils'], ref['sc'], ref['lr']
    out = {}
    # 1. normalize_vectors / _format_actions, both dtypes, degenerate rows
    a32 = rng.standard_normal((512, 3)).astype(np.float32)
    a32[:8] = [[0, 0, 0], [1e-30, 0, 0], [1e-38, 1e-38, 1e-38], [1e19, 1e19, 0],
               [3e38, 0, 0], [1, -1, 1], [-0.0, 0.0, 2.0], [1e-20, 1e20, 1]]
    with np.errstate(all='ignore'):
        out['norm_in'] = a32
        out['norm_f32'] = uu.normalize_vectors(a32)
        out['norm_f64'] = uu.normalize_vectors(a32.astype(np.float64))
        out['scaled_f32'] = uu.normalize_vectors(a32) * np.float32(0.75)
        out['scaled_f64'] = uu.normalize_vectors(a32 + np.zeros((512, 3))) \
            * np.float64(0.75)
    # 2. is_too_curvy: random triples + angles hugging the threshold
    theta = 30.0
    base = rng.standard_normal((2048, 3)).astype(np.float32) * 5 + 8
    d1 = rng.standard_normal((2048, 3))
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    ang = np.deg2rad(theta) + (rng.standard_normal(2048) * 1e-6)
    ang[:256] = rng.uniform(0, np.pi, 256)
    perp = np.cross(d1, rng.standard_normal((2048, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    d2 = np.cos(ang)[:, None] * d1 + np.sin(ang)[:, None] * perp
    p0 = base
    p1 = (p0 + 0.75 * d1).astype(np.float32)
    p2 = (p1 + 0.75 * d2).astype(np.float32)
    p2[-4:] = p1[-4:]                      # zero-length last segment -> NaN
    p1[-8:-4] = p0[-8:-4]                  # zero-length previous segment
    p2[-12:-8] = p0[-12:-8] + (p0[-12:-8] - p1[-12:-8])  # reversed, colinear
    tri = np.stack([p0, p1, p2], axis=1)
    with np.errstate(all='ignore'):
        out['curvy_in'] = tri
        out['curvy_out'] = eu.is_too_curvy(tri, theta)
        out['curvy_theta'] = theta
    # 3. BinaryStoppingCriterion
    D = 12
    _, mask, _ = synthetic_subject(D, peaks=False)
    crit = sc.BinaryStoppingCriterion(mask, 0.1)
    pts = rng.uniform(-1.5, D + 0.5, (4096, 3)).astype(np.float32)
    edge = np.array([0.5, 0.5 - 1e-6, 0.5 + 1e-6, D - 0.5, D - 0.5 - 1e-6,
                     D - 0.5 + 1e-6, 1.0, 1.5, 2.5, D - 1.0, 0.0, D + 0.0],
                    np.float32)
    g = np.stack(np.meshgrid(edge, edge, edge, indexing='ij'), -1).reshape(-1, 3)
    pts = np.concatenate([pts, g.astype(np.float32)])
    from scipy.ndimage import map_coordinates
    out['mask_in'] = mask
    out['mask_coef'] = crit.mask
    out['mask_pts'] = pts
    out['mask_values'] = map_coordinates(crit.mask, pts.T - 0.5, prefilter=False)
    out['mask_stop'] = crit(pts[:, None, :])
    # 4. PeaksAlignmentReward, isolated
    _, _, pk = synthetic_subject(D)
    fn = lr.PeaksAlignmentReward(ref['MRIDataVolume'](pk, np.eye(4)))
    s3 = tri[:1024].copy()
    s3 = np.clip(s3, -1, D + 1).astype(np.float32)
    s3[:4, 1] = 0.2                        # peak lookup at voxel (0,0,0): zeros
    with np.errstate(all='ignore'):
        out['reward_in'] = s3
        out['reward_L3'] = fn(s3, np.zeros(1024, bool))
        out['reward_L2'] = fn(s3[:, 1:], np.zeros(1024, bool))
        out['reward_L1'] = fn(s3[:, 2:], np.zeros(1024, bool))
    out['reward_peaks'] = pk
    _save('isolated_functions', out)


def extra_traces(ref):
    """Traces added after the first set (`python make_golden.py extra`
    regenerates them alone)."""
    # 2 mm isotropic voxels and an origin offset (files-style float64 affine):
    # pins convert_length_mm2vox / step size / neighbourhood radius in voxels
    run_trace(ref, 'trace_f64_K4_vox2mm', D=16, N=96, K=4, noisy=True,
              affine_dtype=np.float64, reward=True, max_length=24.0,
              wobble=0.1, state_every=2, voxel=2.0, origin=(-16.0, -20.0, -12.0),
              seed_stream=977)
    # the same geometry through the training class with a float32 affine
    run_trace(ref, 'trace_f32_K4_vox2mm', D=16, N=96, K=4, noisy=False,
              affine_dtype=np.float32, reward=False, max_length=24.0,
              wobble=0.3, state_every=2, voxel=2.0, origin=(-16.0, -20.0, -12.0),
              seed_stream=978)
    # sigma > 0: pins how NoisyTrackingEnvironment draws from env_dto['rng']
    # (one rng.normal(size=(n_active, 3)) per step, noisy_tracking_env.py:74)
    run_trace(ref, 'trace_f64_K4_sigma', D=12, N=80, K=4, noisy=True,
              affine_dtype=np.float64, reward=False, max_length=40.0,
              wobble=0.05, state_every=2, seed_stream=979, noise=0.3)
    # another angle and mask threshold (every other trace: 30 degrees, 0.1)
    run_trace(ref, 'trace_f32_K4_theta60_thr05', D=14, N=96, K=4, noisy=False,
              affine_dtype=np.float32, reward=False, max_length=40.0,
              wobble=0.6, theta=60.0, thr=0.5, state_every=3, seed_stream=980)
    run_trace(ref, 'trace_f64_K7_theta20_thr03', D=14, N=96, K=7, noisy=True,
              affine_dtype=np.float64, reward=True, max_length=40.0,
              wobble=0.2, theta=20.0, thr=0.3, state_every=3, seed_stream=981)


def main():
    if not os.path.isdir(REFERENCE):
        sys.exit('reference tree not present; fixtures are committed')
    ref = import_reference()
    if sys.argv[1:] == ['extra']:
        extra_traces(ref)
        return
    isolated(ref)
    # train env, float32 affine -> float32 direction arithmetic, reward on
    run_trace(ref, 'trace_f32_K4_reward', D=12, N=96, K=4, noisy=False,
              affine_dtype=np.float32, reward=True, max_length=6.0,
              wobble=0.12, seed_stream=971)
    # track env: noisy class, sigma 0, float64 affine -> float64 directions
    run_trace(ref, 'trace_f64_K100', D=10, N=48, K=100, noisy=True,
              affine_dtype=np.float64, reward=False, max_length=300.0,
              wobble=0.10, state_every=4, seed_stream=972)
    # larger batch: states recorded as row sums after step 2
    run_trace(ref, 'trace_f32_K4_n512', D=16, N=512, K=4, noisy=False,
              affine_dtype=np.float32, reward=False, max_length=60.0,
              wobble=0.15, state_every=1000, seed_stream=973)
    # the shipped model's state: K = 100 with streamlines longer than 101
    # points, so the direction block is full and slides (96^3 volume, as
    # BASELINE config 2; near-diametral chords)
    run_trace(ref, 'trace_f64_K100_long', D=96, N=40, K=100, noisy=True,
              affine_dtype=np.float64, reward=False, max_length=300.0,
              wobble=0.02, state_every=100000, aim_centre=True,
              state_steps=(99, 100, 101, 102, 103), keep_history=True,
              seed_stream=974)
    # noisy class with float32 affine (HDF5 validation env), reward on
    run_trace(ref, 'trace_f64_K4_f32affine', D=12, N=64, K=4, noisy=True,
              affine_dtype=np.float32, reward=True, max_length=30.0,
              wobble=0.2, state_every=2, seed_stream=975)
    extra_traces(ref)


if __name__ == '__main__':
    main()
