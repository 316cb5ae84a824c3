"""Every committed fixture is reproduced by the committed generator.

Where /root/reference exists (the build container), the four generators under
tests/golden/ are run into a scratch directory (TTL_GOLDEN_OUT) and every
.npz they write is compared with the committed file of the same name, array
for array (dtype, shape and bytes).  On the GPU box the reference tree is
absent and the test is skipped: only the fixtures travel.
"""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
GENERATORS = ('make_golden.py', 'make_golden_tracker.py', 'make_golden_learner.py',
              'make_golden_oracle.py')


@pytest.mark.skipif(not os.path.isdir('/root/reference'),
                    reason='the reference tree is only present in the build container')
def test_generators_reproduce_every_committed_fixture(tmp_path):
    env = dict(os.environ, TTL_GOLDEN_OUT=str(tmp_path), PYTHONPATH=ROOT,
               PYTHONDONTWRITEBYTECODE='1')
    for gen in GENERATORS:
        out = subprocess.run([sys.executable, os.path.join(GOLDEN, gen)],
                             capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, (gen, out.stderr[-2000:])
    made = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / '*.npz')))
    committed = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, '*.npz')))
    assert made == committed, set(made) ^ set(committed)
    for name in made:
        new = np.load(str(tmp_path / name), allow_pickle=False)
        old = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), (name, set(new.files) ^ set(old.files))
        for key in new.files:
            a, b = new[key], old[key]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, key)
            if key == 'versions':
                continue            # library versions of the generating run
            assert a.tobytes() == b.tobytes(), (name, key)
