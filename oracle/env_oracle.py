"""TEST INFRASTRUCTURE ONLY -- numpy/scipy restatement of the reference env step.

This file is the CPU oracle for the hot path named by BASELINE.json
(TrackToLearn/environments: reset / step / harvest / get_streamlines).  It is
never imported by the product package; see ``oracle/__init__.py``.

Every function cites the reference lines it restates (paths relative to
``/root/reference``, ``TTL`` = ``TrackToLearn``).  Arithmetic that decides a
stopping mask is written with the *same numpy expressions* as the reference so
that dtype promotion, summation order and the numpy ``arccos`` kernel are the
ones the reference would run on this host.

Pinning status ("Oracle"):
  * everything except ``trilinear_neighborhood`` is pinned by
    ``tests/golden/*.npz``, captured from the reference's own classes imported
    in the build container (``tests/golden/make_golden.py``);
  * ``trilinear_neighborhood`` restates a third-party function that is absent
    from /root/reference (dwi_ml, branch ``for_beluga_scilpy2``;
    requirements.txt:1) -> PARITY UNPINNED for that piece; it is cross-checked
    against ``scipy.ndimage.map_coordinates(order=1, mode='nearest')``.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import map_coordinates, spline_filter

# TTL/environments/stopping_criteria.py:10-20
FLAG_MASK = 1
FLAG_LENGTH = 2
FLAG_CURVATURE = 4
FLAG_TARGET = 8
FLAG_LOOP = 16
FLAG_ANGULAR_ERROR = 32
FLAG_ORACLE = 64


# --------------------------------------------------------------------------
# small vector helpers
# --------------------------------------------------------------------------
def unit_rows(v):
    """TTL/utils/utils.py:117-121 (normalize_vectors): v / |v|, times 1.0.

    einsum keeps the input dtype and sums ((x0*x0 + x1*x1) + x2*x2).
    """
    return (v / np.sqrt(np.einsum('...i,...i', v, v))[..., None]) * 1.


def scale_actions(actions, step_size):
    """TTL/environments/env.py:493-502 (_format_actions)."""
    return unit_rows(actions) * step_size


# --------------------------------------------------------------------------
# stopping tests
# --------------------------------------------------------------------------
def stop_too_long(n_rows, n_points, max_nb_steps):
    """TTL/environments/utils.py:127-142 (is_too_long): batch-wide boolean."""
    return np.full(n_rows, n_points >= max_nb_steps)


def stop_too_curvy(p_last, p_prev, p_prev2, theta_deg):
    """TTL/environments/utils.py:145-173 (is_too_curvy) on the last 3 points.

    ``arccos`` is applied to the *unclipped* float32 dot product; |dot| > 1
    gives NaN and NaN > theta is False.
    """
    max_theta_rad = np.deg2rad(theta_deg)
    u = unit_rows(p_last - p_prev)
    v = unit_rows(p_prev - p_prev2)
    with np.errstate(invalid='ignore'):
        angles = np.arccos(np.einsum('ij,ij->i', u, v))
        return angles > max_theta_rad


def prefilter_mask(mask):
    """TTL/environments/stopping_criteria.py:58-59: cubic B-spline
    coefficients (float64, mirror boundary) of the tracking mask."""
    return spline_filter(np.ascontiguousarray(mask, dtype=float), order=3)


def stop_outside_mask_scipy(coef, p_last, threshold):
    """TTL/environments/stopping_criteria.py:79-82 -- scipy is the arithmetic
    owner here, so the oracle calls it exactly as the reference does."""
    coords = p_last.T - 0.5
    return map_coordinates(coef, coords, prefilter=False) < threshold


def _mirror_index(idx, n):
    """scipy/ndimage/src/ni_interpolation.c border folding used by
    NI_GeometricTransform for every mode except grid-constant (whole-sample
    mirror, period 2n-2).  ``idx`` is an int64 array."""
    if n <= 1:
        return np.zeros_like(idx)
    s2 = 2 * n - 2
    out = idx.copy()
    neg = idx < 0
    if neg.any():
        t = idx[neg]
        t = s2 * ((-t) // s2) + t
        t = np.where(t <= 1 - n, t + s2, -t)
        out[neg] = t
    big = idx >= n
    if big.any():
        t = idx[big]
        t = t - s2 * (t // s2)
        t = np.where(t >= n, s2 - t, t)
        out[big] = t
    return out


def spline3_sample(coef, pts):
    """Per-point restatement of scipy ``map_coordinates(coef, pts.T,
    order=3, mode='constant', cval=0, prefilter=False)`` (float64).

    This is the arithmetic the HIP kernel implements: a point
    with any coordinate outside [0, n-1] (or NaN) evaluates to 0.0; otherwise
    64 taps, mirror-folded at the border, accumulated sequentially as
    ((c*wx)*wy)*wz in lexicographic tap order.
    """
    pts = np.asarray(pts)
    c = pts.astype(np.float64)
    M = c.shape[0]
    dims = coef.shape
    inside = np.ones(M, dtype=bool)
    for ax in range(3):
        with np.errstate(invalid='ignore'):
            inside &= (c[:, ax] >= 0.0) & (c[:, ax] <= dims[ax] - 1)
    out = np.zeros(M, dtype=np.float64)
    if not inside.any():
        return out
    ci = c[inside]
    fl = np.floor(ci)
    start = fl.astype(np.int64) - 1
    y = ci - fl
    z = 1.0 - y
    w = np.empty((4,) + y.shape, dtype=np.float64)
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0
    w[0] = z * z * z / 6.0
    w[3] = 1.0 - w[0] - w[1] - w[2]
    taps = [[_mirror_index(start[:, ax] + j, dims[ax]) for j in range(4)]
            for ax in range(3)]
    t = np.zeros(ci.shape[0], dtype=np.float64)
    for a in range(4):
        for b in range(4):
            for d in range(4):
                v = coef[taps[0][a], taps[1][b], taps[2][d]]
                v = v * w[a][:, 0]
                v = v * w[b][:, 1]
                v = v * w[d][:, 2]
                t = t + v
    out[inside] = t
    return out


def stop_outside_mask(coef, p_last, threshold):
    """Same decision as ``stop_outside_mask_scipy`` through the restated
    per-point evaluation (the float32 ``p - 0.5`` is kept)."""
    coords = p_last - 0.5
    return spline3_sample(coef, coords) < threshold


# --------------------------------------------------------------------------
# state: 7-point trilinear SH gather + previous directions
# --------------------------------------------------------------------------
_CORNERS = np.array([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1],
                     [1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]])


def _corner_polynomial_matrix():
    """Rows: monomials [1,dx,dy,dz,dxdy,dydz,dxdz,dxdydz]; columns: the 8
    corners in ``_CORNERS`` order.  Entry = coefficient of the monomial in the
    expansion of prod_axis (d if corner bit else 1-d)."""
    monos = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1),
             (1, 1, 0), (0, 1, 1), (1, 0, 1), (1, 1, 1)]
    B = np.zeros((8, 8), dtype=np.float32)
    for k, corner in enumerate(_CORNERS):
        for j, mono in enumerate(monos):
            coeff = 1
            for ax in range(3):
                if corner[ax]:      # factor d: only the d^1 term, coeff +1
                    coeff *= 1 if mono[ax] else 0
                else:               # factor (1-d): 1 for d^0, -1 for d^1
                    coeff *= -1 if mono[ax] else 1
            B[j, k] = coeff
    return B


_B1 = _corner_polynomial_matrix()


def neighborhood_offsets(radius):
    """TTL/environments/env.py:207-213: zero row + dwi_ml
    get_neighborhood_vectors_axes(1, r) = [+x,+y,+z,-x,-y,-z] * r, float32."""
    eye = np.eye(3, dtype=np.float32)
    return np.concatenate(
        (np.zeros((1, 3), np.float32), eye * np.float32(radius),
         -eye * np.float32(radius))).astype(np.float32)


def trilinear_neighborhood(vol, coords, neigh):
    """dwi_ml interpolate_volume_in_neighborhood (third-party, absent;
    restated from SURVEY App. B -- PARITY UNPINNED).  Called at
    TTL/environments/env.py:538-541.

    vol (X,Y,Z,C) f32, coords (M,3) f32, neigh (P,3) f32 -> (M, P*C) f32,
    point-major / coefficient-minor.  Corner indices are clipped to the
    volume, weights are not (edge replication); no +-0.5 shift.
    """
    vol = np.asarray(vol, dtype=np.float32)
    coords = np.asarray(coords, dtype=np.float32)
    M = coords.shape[0]
    P = neigh.shape[0]
    pts = np.repeat(coords, P, axis=0) + np.tile(neigh, (M, 1))
    fl = np.floor(pts)
    d = pts - fl
    dx, dy, dz = d[:, 0], d[:, 1], d[:, 2]
    q = np.stack([np.ones_like(dx), dx, dy, dz, dx * dy, dy * dz, dx * dz,
                  dx * dy * dz], axis=1).astype(np.float32)
    w = q @ _B1                                       # (M*P, 8) float32
    with np.errstate(invalid='ignore'):
        base = np.nan_to_num(fl, nan=0.0, posinf=1e9, neginf=-1e9)
    base = base.astype(np.int64)
    upper = np.asarray(vol.shape[:3]) - 1
    out = np.zeros((M * P, vol.shape[3]), dtype=np.float32)
    for k in range(8):
        idx = np.clip(base + _CORNERS[k], 0, upper)
        out += vol[idx[:, 0], idx[:, 1], idx[:, 2]] * w[:, k:k + 1]
    return out.reshape(M, P * vol.shape[3])


def previous_directions(tail, n_dirs):
    """TTL/environments/env.py:549-556.  ``tail`` holds the last
    min(L, n_dirs+1) points (N, T, 3) f32; returns (N, n_dirs, 3) f32, most
    recent segment first, zero padded."""
    N = tail.shape[0]
    prev = np.zeros((N, n_dirs, 3), dtype=np.float32)
    if tail.shape[1] > 1:
        dirs = np.diff(tail, axis=1)
        prev[:, :min(dirs.shape[1], n_dirs), :] = dirs[:, :-(n_dirs + 1):-1, :]
    return prev


def format_state(vol, neigh, history, length, n_dirs):
    """TTL/environments/env.py:504-565 (_format_state) for streamlines of
    ``length`` points stored in ``history`` (N, >=length, 3) f32."""
    N = history.shape[0]
    C = vol.shape[3]
    S = neigh.shape[0] * C
    if N <= 0:
        return np.zeros((0, S + 3 * n_dirs), dtype=np.float32)
    head = history[:, length - 1, :]
    signal = trilinear_neighborhood(vol, head, neigh)
    state = np.zeros((N, S + 3 * n_dirs), dtype=np.float32)
    state[:, :S] = signal
    first = max(0, length - (n_dirs + 1))
    prev = previous_directions(history[:, first:length, :], n_dirs)
    state[:, S:] = prev.reshape(N, 3 * n_dirs)
    return state


# --------------------------------------------------------------------------
# reward
# --------------------------------------------------------------------------
def nearest_peaks(peaks, idx):
    """TTL/environments/interpolation.py:7-26."""
    unclipped = np.round(idx).astype(np.int32)
    upper = np.asarray(peaks.shape[:3]) - 1
    ijk = np.clip(unclipped, 0, upper).astype(int).T
    return peaks[tuple(ijk)]


def peaks_alignment_reward(peaks, p_last, p_prev, p_prev2):
    """TTL/environments/local_reward.py:29-107 for streamlines with >= 2
    points (``p_prev2`` is None when the streamline has exactly 2 points)."""
    N = p_last.shape[0]
    P = peaks.shape[3]
    idx = p_prev.astype(np.int32)
    v = nearest_peaks(peaks, idx)
    v = np.reshape(v, (N * 5, P // 5))
    with np.errstate(divide='ignore', invalid='ignore'):
        v = unit_rows(v)
    v = np.nan_to_num(np.reshape(v, (N, 5, P // 5)))
    u = p_last - p_prev
    with np.errstate(divide='ignore', invalid='ignore'):
        u = unit_rows(u)
    u = np.nan_to_num(u)
    dot = np.abs(np.einsum('ijk,ik->ij', v, u))
    rewards = np.amax(dot, axis=-1)
    factors = np.ones((N))
    if p_prev2 is not None:
        w = p_prev - p_prev2
        with np.errstate(divide='ignore', invalid='ignore'):
            w = unit_rows(w)
        w = np.nan_to_num(w)
        np.einsum('ik,ik->i', u, w, out=factors)
    rewards *= factors
    return rewards


def combine_rewards(weighted_factors, n_rows):
    """TTL/environments/reward.py:46-79 (RewardFunction.__call__): float64
    (F, N) table of w * factor (factors with w <= 0 stay zero), per-factor
    means, column sum.  ``weighted_factors`` = [(name, w, values-or-None)]."""
    table = np.zeros((len(weighted_factors), n_rows))
    for i, (_, w, values) in enumerate(weighted_factors):
        if w > 0 and values is not None:
            table[i] = w * values
    info = {name: np.mean(table[i])
            for i, (name, _, _) in enumerate(weighted_factors)}
    return np.sum(table, axis=0), info


# --------------------------------------------------------------------------
# the environment
# --------------------------------------------------------------------------
class OracleTrackingEnv:
    """Restatement of TrackingEnvironment (TTL/environments/tracking_env.py)
    over BaseEnv (TTL/environments/env.py) for one already-loaded subject.

    Parameters mirror what ``BaseEnv.load_subject`` (env.py:143-281) derives:
    ``step_size`` is the step in voxels *with the numpy scalar type the
    reference would hold* (np.float32 for an HDF5/float32 affine, np.float64
    for a nibabel affine) because that type decides float32 vs float64
    direction arithmetic under numpy >= 2.
    """

    def __init__(self, sh, mask, seeds, *, n_dirs, theta, step_size,
                 max_nb_steps, mask_threshold, peaks=None,
                 compute_reward=False, alignment_weighting=1.0,
                 spline_eval='restated'):
        self.vol = np.ascontiguousarray(sh, dtype=np.float32)
        self.coef = prefilter_mask(np.asarray(mask).astype(np.uint8))
        self.seeds = seeds
        self.n_dirs = int(n_dirs)
        self.theta = theta
        self.step_size = step_size
        self.max_nb_steps = int(max_nb_steps)
        self.mask_threshold = mask_threshold
        self.peaks = peaks
        self.compute_reward = bool(compute_reward)
        self.alignment_weighting = alignment_weighting
        self.neigh = neighborhood_offsets(step_size)
        self._mask_stop = (stop_outside_mask if spline_eval == 'restated'
                           else stop_outside_mask_scipy)

    # -- TTL/environments/env.py:567-603 + tracking_env.py:22-45 -----------
    def _stopping(self, idx, n_points):
        hist = self.streamlines
        n = len(idx)
        should_stop = np.zeros(n, dtype=np.bool_)
        flags = np.zeros(n, dtype=int)
        hit = stop_too_long(n, n_points, self.max_nb_steps)
        flags[hit] |= FLAG_LENGTH
        should_stop[hit] = True
        if n_points >= 3:
            hit = stop_too_curvy(hist[idx, n_points - 1], hist[idx, n_points - 2],
                                 hist[idx, n_points - 3], self.theta)
        else:
            hit = np.zeros(n, dtype=bool)
        flags[hit] |= FLAG_CURVATURE
        should_stop[hit] = True
        hit = self._mask_stop(self.coef, hist[idx, n_points - 1],
                              self.mask_threshold)
        flags[hit] |= FLAG_MASK
        should_stop[hit] = True
        return should_stop, flags

    def _start(self, initial_points):
        n = initial_points.shape[0]
        self.initial_points = initial_points
        self.streamlines = np.zeros((n, self.max_nb_steps + 1, 3),
                                    dtype=np.float32)
        self.streamlines[:, 0, :] = initial_points
        self.flags = np.zeros(n, dtype=int)
        self.lengths = np.ones(n, dtype=np.int32)
        self.length = 1
        self.dones = np.full(n, False)
        self.continue_idx = np.arange(n)
        self.state = format_state(self.vol, self.neigh, self.streamlines,
                                  self.length, self.n_dirs)
        return self.state[self.continue_idx]

    def reset(self, start, end):
        """TTL/environments/tracking_env.py:91-133."""
        return self._start(self.seeds[start:end])

    def nreset(self, n_seeds):
        """TTL/environments/tracking_env.py:47-89 (global numpy RNG)."""
        replace = n_seeds > len(self.seeds)
        pick = np.random.choice(np.arange(len(self.seeds)), size=n_seeds,
                                replace=replace)
        return self._start(self.seeds[pick])

    def _perturb(self, actions):
        return actions

    def step(self, actions):
        """TTL/environments/tracking_env.py:135-221."""
        actions = self._perturb(actions)
        idx = self.continue_idx
        L = self.length
        hist = self.streamlines
        directions = scale_actions(actions, self.step_size)
        self.last_flip = np.zeros(len(idx), dtype=bool)
        if L == 1:
            # trial step on a scratch copy, then flip what would stop
            saved = hist[idx, L, :].copy()
            hist[idx, L, :] = hist[idx, L - 1, :] + directions
            stopping, _ = self._stopping(idx, L + 1)
            hist[idx, L, :] = saved
            directions[stopping] *= -1
            self.last_flip = stopping
        hist[idx, L, :] = hist[idx, L - 1, :] + directions
        self.length = L = L + 1

        stopping, new_flags = self._stopping(idx, L)
        self.not_stopping = np.logical_not(stopping)
        self.new_continue_idx = idx[~stopping]
        self.stopping_idx = idx[stopping]
        self.flags[self.stopping_idx] = new_flags[stopping]
        self.dones[self.stopping_idx] = 1

        reward = np.zeros(hist.shape[0])
        reward_info = {}
        if self.compute_reward:
            p2 = hist[idx, L - 3] if L >= 3 else None
            align = None
            if self.alignment_weighting > 0:
                align = peaks_alignment_reward(
                    self.peaks, hist[idx, L - 1], hist[idx, L - 2], p2)
            # the sparse oracle factor (oracle_reward.py) is out of scope for
            # this oracle: weight 0 -> a zero row, as reward.py:70 skips it
            reward, reward_info = combine_rewards(
                [('peaks_reward', self.alignment_weighting, align),
                 ('oracle_reward', 0.0, None)], len(idx))

        self.state[idx] = format_state(self.vol, self.neigh, hist[idx], L,
                                       self.n_dirs)
        return (self.state[idx], reward, self.dones[idx],
                {'continue_idx': idx, 'reward_info': reward_info})

    def harvest(self):
        """TTL/environments/tracking_env.py:223-245."""
        self.lengths[self.stopping_idx] = self.length
        self.continue_idx = self.new_continue_idx
        return self.state[self.continue_idx], self.not_stopping

    def get_streamlines(self):
        """TTL/environments/tracking_env.py:247-294: ragged list, last point
        dropped when the CURVATURE or MASK bit is set."""
        drop = ((self.flags & FLAG_CURVATURE) != 0) | ((self.flags & FLAG_MASK) != 0)
        out = []
        for i in range(len(self.streamlines)):
            s = self.streamlines[i, :self.lengths[i], :]
            out.append(s[:-1] if drop[i] else s)
        return out, self.initial_points, self.flags


class OracleNoisyTrackingEnv(OracleTrackingEnv):
    """TTL/environments/noisy_tracking_env.py:38-77 without an FA map:
    float64 gaussian noise (also for noise == 0) is added to the action first,
    so the direction arithmetic runs in float64."""

    def __init__(self, *args, noise=0.0, rng=None, **kw):
        super().__init__(*args, **kw)
        self.noise = noise
        self.rng = rng if rng is not None else np.random.RandomState(0)

    def _perturb(self, actions):
        return actions + self.rng.normal(0., self.noise, size=actions.shape)
