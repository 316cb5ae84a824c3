"""Minimal in-memory tractogram containers.

nibabel is absent from this image; these classes give the few members the
env/tracker path touches (TrackToLearn/environments/tracking_env.py:289-292,
TrackToLearn/tracking/tracker.py:118-142,252-256): iteration over items with
``.streamline`` and ``.data_for_streamline``, ``len``, ``+=``, ``.streamlines``,
``.data_per_streamline`` and ``apply_affine``.
"""
import numpy as np


class TractogramItem(object):
    def __init__(self, streamline, data_for_streamline=None, data_for_points=None):
        self.streamline = streamline
        self.data_for_streamline = data_for_streamline or {}
        self.data_for_points = data_for_points or {}


class Tractogram(object):
    def __init__(self, streamlines=None, data_per_streamline=None,
                 affine_to_rasmm=None):
        self.streamlines = list(streamlines) if streamlines is not None else []
        self.data_per_streamline = {
            k: np.asarray(v) for k, v in (data_per_streamline or {}).items()}
        self.affine_to_rasmm = affine_to_rasmm

    def __len__(self):
        return len(self.streamlines)

    def __iter__(self):
        for i, s in enumerate(self.streamlines):
            yield TractogramItem(
                s, {k: v[i] for k, v in self.data_per_streamline.items()})

    def __iadd__(self, other):
        n_self = len(self.streamlines)
        self.streamlines = self.streamlines + list(other.streamlines)
        keys = set(self.data_per_streamline) | set(other.data_per_streamline)
        if n_self == 0:
            self.data_per_streamline = {
                k: np.asarray(v) for k, v in other.data_per_streamline.items()}
        else:
            for k in keys:
                self.data_per_streamline[k] = np.concatenate(
                    [self.data_per_streamline[k], other.data_per_streamline[k]])
        return self

    def apply_affine(self, affine):
        """Points p -> A[:3,:3] @ p + A[:3,3], in place."""
        A = np.asarray(affine, dtype=np.float64)
        self.streamlines = [
            (np.asarray(s, np.float64) @ A[:3, :3].T + A[:3, 3]).astype(np.float32)
            for s in self.streamlines]
        return self


class LazyTractogram(object):
    """Generator-backed tractogram (the subset of nibabel's LazyTractogram
    that TrackToLearn/tracking/tracker.py:147-148 and the file writers use):
    ``from_data_func(gen_fn)`` + iteration over TractogramItems."""

    def __init__(self, data_func=None, affine_to_rasmm=None):
        self._data_func = data_func
        self.affine_to_rasmm = affine_to_rasmm

    @classmethod
    def from_data_func(cls, data_func):
        return cls(data_func=data_func)

    def __iter__(self):
        return iter(self._data_func())


def streamline_length(streamline):
    """Arc length of a polyline (float64 accumulation), the quantity
    ``dipy.tracking.streamlinespeed.length`` returns for one streamline
    (TrackToLearn/tracking/tracker.py:120)."""
    s = np.asarray(streamline, dtype=np.float64)
    if len(s) < 2:
        return 0.0
    return float(np.sqrt((np.diff(s, axis=0) ** 2).sum(axis=1)).sum())


def compress_streamline(streamline, tol_error=0.01, max_segment_length=10.0):
    """Linearisation-based compression (Presseau et al. 2015), the algorithm
    behind ``dipy.tracking.streamlinespeed.compress_streamlines`` used at
    TrackToLearn/tracking/tracker.py:123-125 (dipy is absent here -> own
    restatement, parity unpinned): walk along the streamline and drop a point
    while every dropped point stays within ``tol_error`` of the chord and the
    chord is no longer than ``max_segment_length``."""
    s = np.asarray(streamline)
    n = len(s)
    if n <= 2:
        return s.copy()
    keep = [0]
    prev = 0
    for nxt in range(2, n):
        a, b = s[prev].astype(np.float64), s[nxt].astype(np.float64)
        ab = b - a
        seg_len = np.linalg.norm(ab)
        ok = seg_len <= max_segment_length
        if ok and seg_len > 0:
            mid = s[prev + 1:nxt].astype(np.float64) - a
            t = np.clip(mid @ ab / (seg_len ** 2), 0.0, 1.0)
            dist = np.linalg.norm(mid - t[:, None] * ab, axis=1)
            ok = bool((dist <= tol_error).all())
        elif ok:
            ok = bool((np.linalg.norm(
                s[prev + 1:nxt].astype(np.float64) - a, axis=1) <= tol_error).all())
        if not ok:
            keep.append(nxt - 1)
            prev = nxt - 1
    keep.append(n - 1)
    return s[keep].copy()
