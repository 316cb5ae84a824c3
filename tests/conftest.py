import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """Build libttl_hip.so if it is missing or stale (hipcc cross-compiles
    gfx950 without a GPU; the built library is git-ignored)."""
    try:
        from tracktolearn_amd.csrc import build as hip_build
        if not hip_build.up_to_date():
            hip_build.build(verbose=False)
    except Exception as exc:          # pragma: no cover
        print(f'WARNING: could not build libttl_hip.so: {exc}')
