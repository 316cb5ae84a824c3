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
