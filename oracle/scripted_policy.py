"""TEST INFRASTRUCTURE ONLY -- numpy twin of the scripted, policy-free action
generator (``ttl_scripted_actions`` in tracktolearn_amd/csrc/ttl_hip.hip;
SURVEY 8d).  Same counter-based hash, same float32 operation order, so the CPU
oracle and the GPU env consume bit-identical actions regardless of compaction.
"""
import numpy as np

_U = np.uint32


def _mix32(h):
    h = h ^ (h >> _U(16))
    h = h * _U(0x85EBCA6B)
    h = h ^ (h >> _U(13))
    h = h * _U(0xC2B2AE35)
    h = h ^ (h >> _U(16))
    return h


def scripted_noise(seed, step, gid, comp):
    """Centred sum of 4 hashed uniforms times sqrt(3): float32, unit variance."""
    with np.errstate(over='ignore'):
        gid = np.asarray(gid).astype(np.uint32)
        a = _mix32(np.array([seed], np.uint32) * _U(0x9E3779B1) + _U(step))
        b = _mix32(gid * _U(0x27D4EB2F) + _U(comp) * _U(0x165667B1) + _U(0x1234567))
        base = a ^ b
        acc = np.zeros(gid.shape, np.float32)
        for k in range(4):
            h = _mix32(base + _U(k) * _U(0x9E3779B9))
            acc = acc + (h >> _U(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (acc - np.float32(2.0)) * np.float32(1.7320508075688772)


def scripted_actions(state, dir_offset, continue_idx, seed, step, wobble):
    """Actions (n, 3) float32 for the active rows of ``state`` (n, W)."""
    n = len(continue_idx)
    noise = np.stack([scripted_noise(seed, step, continue_idx, c)
                      for c in range(3)], axis=1)
    if step == 0:
        return noise
    prev = np.asarray(state)[:, dir_offset:dir_offset + 3].astype(np.float32)
    s = np.sqrt((prev[:, 0] * prev[:, 0] + prev[:, 1] * prev[:, 1]) +
                prev[:, 2] * prev[:, 2])
    s = np.where(s > 0, s, np.float32(1.0)).astype(np.float32)
    w = np.float32(wobble)
    return (prev / s[:, None] + w * noise).astype(np.float32).reshape(n, 3)
