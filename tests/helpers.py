"""Shared test helpers: synthetic subjects and golden-trace replay."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

TRACES = ['trace_f32_K4_reward', 'trace_f64_K100', 'trace_f32_K4_n512',
          'trace_f64_K4_f32affine', 'trace_f64_K100_long',
          'trace_f64_K4_vox2mm', 'trace_f32_K4_vox2mm', 'trace_f64_K4_sigma',
          'trace_f32_K4_theta60_thr05', 'trace_f64_K7_theta20_thr03']


def synthetic_subject(D, C=45, seed=1234, peaks=True):
    """SURVEY 8(d) recipe (same as tests/golden/make_golden.py)."""
    rng = np.random.RandomState(seed)
    sh = (0.1 * rng.standard_normal((D, D, D, C))).astype(np.float32)
    sh[..., 0] = 1.0
    g = np.indices((D, D, D)).astype(np.float64)
    r = np.sqrt(((g - (D - 1) / 2.0) ** 2).sum(0))
    mask = (r < 0.42 * D).astype(np.uint8)
    pk = rng.standard_normal((D, D, D, 15)).astype(np.float32) if peaks else None
    if pk is not None:
        pk[0, 0, 0] = 0.0
        pk[D // 2, D // 2, D // 2, 3:6] = 0.0
    return sh, mask, pk


def load_trace(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return z


def trace_noise(z):
    """(sigma, RandomState in the state the reference's generator had before
    its first step) of a trace, or (0.0, None)."""
    sigma = float(z['noise']) if 'noise' in z.files else 0.0
    if sigma <= 0:
        return 0.0, None
    rs = np.random.RandomState(0)
    rs.set_state(('MT19937', z['rng_key'], int(z['rng_pos']),
                  int(z['rng_has_gauss']), float(z['rng_cached'])))
    return sigma, rs


def trace_step_size(z):
    """The step size with the numpy scalar type the reference held."""
    return np.dtype(str(z['step_size_dtype'])).type(z['step_size'])
