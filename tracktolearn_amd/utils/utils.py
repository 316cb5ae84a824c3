"""Host utilities used outside the hot path.

API-compatible with the few helpers of TrackToLearn/utils/utils.py that the
trainers touch: ``LossHistory`` (per-epoch running mean written to
``<path>/plots/<filename>.npy``), ``Timer`` and ``normalize_vectors``.
"""
import os
import sys
import time

import numpy as np


class LossHistory(object):
    """Running mean of a scalar within an epoch; ``end_epoch(e)`` appends
    ``(e, mean)`` to ``epochs``, persists the list and starts a new epoch.
    Infinite values are ignored (utils/utils.py:24-80)."""

    def __init__(self, name, filename, path):
        self.name, self.filename, self.path = name, filename, path
        self.history = []          # every accepted value, in order
        self.epochs = []           # (epoch, mean over that epoch)
        self._open = []            # values of the epoch in progress
        self.num_iter = 0
        self.num_epochs = 0

    def __len__(self):
        return len(self.history)

    @property
    def avg(self):
        return float(np.mean(self._open)) if self._open else 0.0

    # kept for callers that read the raw accumulators
    @property
    def sum(self):
        return float(np.sum(self._open)) if self._open else 0.0

    @property
    def count(self):
        return len(self._open)

    def update(self, value):
        if np.isinf(value):
            return
        self.history.append(value)
        self._open.append(value)
        self.num_iter += 1

    def end_epoch(self, epoch):
        self.epochs.append((epoch, self.avg))
        self._open = []
        self.num_epochs += 1
        target = os.path.join(self.path, 'plots')
        os.makedirs(target, exist_ok=True)
        np.save(os.path.join(target, self.filename + '.npy'), self.epochs)


class Timer:
    """``with Timer('loading'):`` prints the wall time of the block."""

    def __init__(self, txt, newline=False, color=None):
        self.txt, self.newline = txt, newline

    def __enter__(self):
        self._t0 = time.time()
        sys.stdout.write(self.txt + '... ' + ('\n' if self.newline else ''))
        sys.stdout.flush()
        return self

    def __exit__(self, *exc):
        prefix = (self.txt + ' done in ') if self.newline else ''
        print('%s%.2f sec.' % (prefix, time.time() - self._t0))


def normalize_vectors(v, norm=1.):
    """v / |v| * norm along the last axis (utils/utils.py:117-121).  On the
    step path this arithmetic runs inside the HIP kernel ``k_advance``; the
    host version serves callers outside the hot path."""
    length = np.sqrt(np.einsum('...i,...i', v, v))
    return (v / length[..., None]) * norm
