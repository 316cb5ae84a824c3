"""A slice of the randomised differential run (tests/stress_parity.py) inside
the GPU suite: random volume shapes, SH orders, K, angles, thresholds, step
sizes, env classes, affine dtypes, batch sizes, processing orders and a random
mix of step()/step_device(), each run to exhaustion against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_twelve_random_configurations():
    import stress_parity
    from tracktolearn_amd.environments import TrackingEnvironment
    saved = TrackingEnvironment.SPATIAL_ORDER_MIN
    saved_refresh = TrackingEnvironment.SPATIAL_ORDER_REFRESH
    rng = np.random.RandomState(2024)
    stops = np.zeros(3, np.int64)
    try:
        for k in range(12):
            r = stress_parity.one(rng, k)
            stops += np.array(r['stops'])
            assert r['worst_state_err'] <= 1e-5
    finally:
        TrackingEnvironment.SPATIAL_ORDER_MIN = saved
        TrackingEnvironment.SPATIAL_ORDER_REFRESH = saved_refresh
    assert (stops > 0).all()      # mask, length and curvature stops all seen
