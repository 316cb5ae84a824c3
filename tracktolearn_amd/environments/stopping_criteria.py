"""Stopping flags and the host-side constants of the device stopping tests.

Mirrors TrackToLearn/environments/stopping_criteria.py:10-35 (StoppingFlags,
is_flag_set, count_flags).  The criteria themselves (length, curvature,
cubic-spline mask) run inside the HIP kernel ``k_advance``; this module only
prepares what the kernel needs from the host: the spline coefficients
(``spline_filter`` at load time, as stopping_criteria.py:58-59 does) and the
float32 dot-product threshold equivalent to numpy's ``arccos(dot) > theta``.
"""
from enum import Enum

import numpy as np
from scipy.ndimage import spline_filter


class StoppingFlags(Enum):
    """Bit flags, stopping_criteria.py:10-20."""
    STOPPING_MASK = int('00000001', 2)
    STOPPING_LENGTH = int('00000010', 2)
    STOPPING_CURVATURE = int('00000100', 2)
    STOPPING_TARGET = int('00001000', 2)
    STOPPING_LOOP = int('00010000', 2)
    STOPPING_ANGULAR_ERROR = int('00100000', 2)
    STOPPING_ORACLE = int('01000000', 2)


def is_flag_set(flags, ref_flag):
    """Which entries of ``flags`` have ``ref_flag`` set
    (stopping_criteria.py:23-28)."""
    if type(ref_flag) is StoppingFlags:
        ref_flag = ref_flag.value
    return ((flags.astype(np.uint8) & ref_flag) >>
            np.log2(ref_flag).astype(np.uint8)).astype(bool)


def count_flags(flags, ref_flag):
    """How many entries have ``ref_flag`` set (stopping_criteria.py:31-35)."""
    if type(ref_flag) is StoppingFlags:
        ref_flag = ref_flag.value
    return is_flag_set(flags, ref_flag).sum()


def mask_spline_coefficients(mask):
    """Cubic B-spline coefficients of the tracking mask, float64
    (BinaryStoppingCriterion.__init__, stopping_criteria.py:58-59)."""
    return spline_filter(np.ascontiguousarray(mask, dtype=float), order=3)


def _ordered_to_f32(k):
    """Monotone map from int64 ``k`` to float32: k >= 0 -> the float with bit
    pattern k, k < 0 -> minus the float with bit pattern -k - 1 (so that
    -0.0 < +0.0 are adjacent and the order of k is the order of the values)."""
    k = np.asarray(k, dtype=np.int64)
    mag = np.where(k >= 0, k, -k - 1).astype(np.uint32)
    out = mag.view(np.float32).copy()
    out[k < 0] *= np.float32(-1.0)
    return out


_ONE_BITS = int(np.array([1.0], np.float32).view(np.uint32)[0])


def curvature_dot_threshold(max_theta_deg, window=2048):
    """Largest float32 ``c`` with ``np.arccos(c) > deg2rad(theta)``.

    The reference decides curvature with numpy's float32 ``arccos`` of the
    unclipped dot product (TrackToLearn/environments/utils.py:162-173).  That
    kernel is not correctly rounded and depends on numpy's SIMD dispatch, so
    instead of re-implementing acos on the GPU the decision is turned into a
    comparison of the dot product itself: arccos is decreasing, hence
    ``arccos(dot) > theta  <=>  -1 <= dot <= c``.  ``c`` is found by bisecting
    *this host's* numpy over the ordered float32 values of [-1, 1] with
    exactly the reference's comparison (including its dtype promotion), and a
    window of neighbours around the boundary is then verified to be a clean
    step so the equivalence is exact.

    Returns (c, enabled).  ``enabled`` is False when no dot product in
    [-1, 1] can exceed the angle (theta >= 180 deg).
    """
    max_theta_rad = np.deg2rad(max_theta_deg)

    def too_curvy(values):
        with np.errstate(invalid='ignore'):
            return np.arccos(np.ascontiguousarray(values, np.float32)) > max_theta_rad

    def probe(k):          # one value, evaluated inside a full SIMD vector
        return bool(too_curvy(np.repeat(_ordered_to_f32([k]), 64))[0])

    lo, hi = -_ONE_BITS - 1, _ONE_BITS      # ordered indices of -1.0 and +1.0
    if not probe(lo):
        return np.float32(-2.0), False
    if probe(hi):
        return np.float32(1.0), True        # every valid dot is too curvy
    while hi - lo > 1:                      # invariant: probe(lo), not probe(hi)
        mid = (lo + hi) // 2
        if probe(mid):
            lo = mid
        else:
            hi = mid
    ks = np.arange(max(lo - window, -_ONE_BITS - 1),
                   min(hi + window, _ONE_BITS) + 1, dtype=np.int64)
    curvy = too_curvy(_ordered_to_f32(ks))
    n_true = int(curvy.sum())
    if n_true == 0 or n_true == len(ks) or not curvy[:n_true].all():
        raise RuntimeError(
            'numpy arccos is not a clean step around cos(theta); cannot '
            'derive an exact curvature threshold on this host')
    return np.float32(_ordered_to_f32(ks[n_true - 1:n_true])[0]), True
