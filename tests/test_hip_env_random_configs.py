"""A slice of the randomised differential run (tests/stress_parity.py) inside
the GPU suite: random volume shapes, SH orders, K, angles, thresholds, step
sizes, env classes, affine dtypes, batch sizes, processing orders and a random
mix of step()/step_device() / free-running steps and of every scheduling knob
of rounds 2 and 3 (order refresh period, one-launch tails and their caps, SH
record order, order keys, early refresh threshold, lazy step state), each run
to exhaustion against the CPU oracle: 150 configurations, fixed seed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_hundred_and_fifty_random_configurations():
    import os
    import stress_parity
    from tracktolearn_amd.environments import TrackingEnvironment
    # the stress script randomises class attributes and TTL_* environment knobs:
    # put all of them back, the rest of the suite runs in this process
    attrs = ('SPATIAL_ORDER_MIN', 'SPATIAL_ORDER_REFRESH', 'FREERUN_MAX', 'lazy_step_state',
             'ORDER_MIN_FILL', 'TAIL_FUSED_MAX_ROWS')
    saved = {a: getattr(TrackingEnvironment, a) for a in attrs}
    saved_env = dict(os.environ)
    rng = np.random.RandomState(2024)
    stops = np.zeros(3, np.int64)
    try:
        for k in range(150):          # ~0.2 s each (profiles/r03_stress_parity_*.log)
            r = stress_parity.one(rng, k)
            stops += np.array(r['stops'])
            assert r['worst_state_err'] <= 1e-5
    finally:
        for a, v in saved.items():
            setattr(TrackingEnvironment, a, v)
        for key in [k for k in os.environ if k.startswith('TTL_') and k not in saved_env]:
            del os.environ[key]
        for key, v in saved_env.items():
            if key.startswith('TTL_'):
                os.environ[key] = v
    assert (stops > 0).all()      # mask, length and curvature stops all seen
