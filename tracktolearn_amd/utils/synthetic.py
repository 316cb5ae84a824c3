"""Seeded synthetic subjects for benchmarks and tests (SURVEY.md 8d recipe;
there is no network for real datasets): a descoteaux07-shaped SH volume of C
coefficients, a ball-shaped WM/tracking/seeding mask, random peaks, and seeds
drawn (with replacement) from the mask voxels."""
import numpy as np

from tracktolearn_amd.datasets.utils import MRIDataVolume


def synthetic_volumes(D, C=45, seed=1234, peaks=True):
    """sh = 0.1*N(0,1) float32 (D,D,D,C) with sh[...,0] = 1; mask = ball of
    radius 0.42*D (uint8); peaks = N(0,1) float32 (D,D,D,15)."""
    rng = np.random.RandomState(seed)
    sh = rng.standard_normal((D, D, D, C)).astype(np.float32)
    sh *= np.float32(0.1)
    sh[..., 0] = 1.0
    g = np.indices((D, D, D)).astype(np.float32)
    r2 = ((g - np.float32((D - 1) / 2.0)) ** 2).sum(0)
    mask = (r2 < np.float32((0.42 * D) ** 2)).astype(np.uint8)
    pk = rng.standard_normal((D, D, D, 15)).astype(np.float32) if peaks else None
    return sh, mask, pk


def synthetic_subject(D, C=45, seed=1234, peaks=True, affine_dtype=np.float32):
    """The 5-tuple ``BaseEnv`` accepts as ``subject_data`` (1 mm iso affine)."""
    sh, mask, pk = synthetic_volumes(D, C, seed, peaks)
    aff = np.eye(4, dtype=affine_dtype)
    return (MRIDataVolume(sh, aff), MRIDataVolume(mask, aff),
            MRIDataVolume(mask, aff),
            MRIDataVolume(pk, aff) if pk is not None else None, None)


def synthetic_seeds(mask, n, seed=0):
    """n seeds: a mask voxel (with replacement) + U[-0.5, 0.5)^3, float64."""
    rng = np.random.RandomState(seed)
    vox = np.argwhere(mask)
    pick = rng.randint(0, len(vox), n)
    return vox[pick] + rng.uniform(-0.5, 0.5, (n, 3))
