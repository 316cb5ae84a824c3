"""Host utilities (TrackToLearn/utils/utils.py): LossHistory, Timer and
normalize_vectors."""
import os
import sys
from os.path import join as pjoin
from time import time

import numpy as np


class LossHistory(object):
    """Running history of a scalar, saved per epoch to ``plots/<file>.npy``
    (utils/utils.py:24-80)."""

    def __init__(self, name, filename, path):
        self.name = name
        self.history = []
        self.epochs = []
        self.sum = 0.0
        self.count = 0
        self._avg = 0.0
        self.num_iter = 0
        self.num_epochs = 0
        self.filename = filename
        self.path = path

    def __len__(self):
        return len(self.history)

    def update(self, value):
        if np.isinf(value):
            return
        self.history.append(value)
        self.sum += value
        self.count += 1
        self._avg = self.sum / self.count
        self.num_iter += 1

    @property
    def avg(self):
        return self._avg

    def end_epoch(self, epoch):
        self.epochs.append((epoch, self._avg))
        self.sum = 0.0
        self.count = 0
        self._avg = 0.0
        self.num_epochs += 1
        directory = pjoin(self.path, 'plots')
        os.makedirs(directory, exist_ok=True)
        with open(pjoin(directory, '{}.npy'.format(self.filename)), 'wb') as f:
            np.save(f, self.epochs)


class Timer:
    """``with Timer('loading'):`` prints the wall time of the block."""

    def __init__(self, txt, newline=False, color=None):
        self.txt = txt
        self.newline = newline

    def __enter__(self):
        self.start = time()
        print(self.txt + '... ', end='' if not self.newline else '\n')
        sys.stdout.flush()

    def __exit__(self, type, value, tb):
        if self.newline:
            print(self.txt + ' done in ', end='')
        print('{:.2f} sec.'.format(time() - self.start))


def normalize_vectors(v, norm=1.):
    """v / |v| * norm along the last axis (utils/utils.py:117-121).  On the
    step path this arithmetic runs inside the HIP kernel ``k_advance``; the
    host version is kept for callers outside the hot path."""
    return (v / np.sqrt(np.einsum('...i,...i', v, v))[..., None]) * norm
