"""Volume container and length conversion (host side of the env path).

Mirrors TrackToLearn/datasets/utils.py:10-43 (MRIDataVolume) and :88-124
(convert_length_mm2vox).  The HDF5 / NIfTI loaders of that file depend on
h5py / nibabel, which are absent from this image; they are "next" rows of
SURVEY.md 8(f).
"""
import numpy as np


class MRIDataVolume(object):
    """A data volume and its vox->rasmm affine (datasets/utils.py:10-43)."""

    def __init__(self, data=None, affine_vox2rasmm=None):
        self._data = data
        self.affine_vox2rasmm = affine_vox2rasmm

    @property
    def data(self):
        return self._data

    @property
    def shape(self):
        return self.data.shape


def convert_length_mm2vox(length_mm, affine_vox2rasmm):
    """Length in mm -> isotropic voxel units (datasets/utils.py:88-124).

    The numpy scalar type of the result follows the affine's dtype exactly as
    in the reference: that type later decides float32 vs float64 direction
    arithmetic.
    """
    diag = np.diagonal(affine_vox2rasmm)[:3]
    vox2mm = np.mean(np.abs(diag))
    if not np.allclose(np.abs(diag), vox2mm, rtol=5e-2, atol=5e-2):
        raise ValueError('Voxel space is not iso,  cannot convert a scalar '
                         'length in mm to voxel space. Affine provided : '
                         '{}'.format(affine_vox2rasmm))
    return length_mm / vox2mm


def get_sh_order_and_fullness(n_coefs):
    """SH order and whether the basis is "full" (odd orders included) from
    the number of coefficients: symmetric bases have (n+1)(n+2)/2, full bases
    (n+1)^2 (what scilpy.reconst.utils.get_sh_order_and_fullness returns)."""
    n_coefs = int(n_coefs)
    sym = (-3.0 + np.sqrt(1.0 + 8.0 * n_coefs)) / 2.0
    if abs(sym - round(sym)) < 1e-9 and int(round(sym)) % 2 == 0:
        return int(round(sym)), False
    full = np.sqrt(n_coefs) - 1.0
    if abs(full - round(full)) < 1e-9:
        return int(round(full)), True
    raise ValueError(f'{n_coefs} coefficients match no SH order')


def _full_basis_degrees(order):
    """Degree l of every coefficient of a full basis up to ``order``."""
    return np.concatenate([np.full(2 * l + 1, l) for l in range(order + 1)])


def set_sh_order_basis(sh, sh_basis, target_basis='descoteaux07', target_order=6):
    """Bring SH coefficients to the target order (and basis).  Mirrors
    TrackToLearn/datasets/utils.py:127-179: a full basis keeps only its even
    degrees; a different order is truncated or zero padded.  Converting
    between the tournier07 and descoteaux07 bases needs scilpy (absent): only
    ``sh_basis == target_basis`` is accepted."""
    n_coefs = sh.shape[-1]
    sh_order, full_basis = get_sh_order_and_fullness(n_coefs)
    if full_basis:
        print('SH coefficients are in "full" basis, only even coefficients '
              'will be used.')
        sh = sh[..., _full_basis_degrees(sh_order) % 2 == 0]
    target_order = int(target_order)
    if sh_order != target_order:
        print('SH coefficients are of order {}, converting them to order {}.'
              .format(sh_order, target_order))
        target_n_coefs = (target_order + 1) * (target_order + 2) // 2
        if n_coefs > target_n_coefs:
            sh = sh[..., :target_n_coefs]
        else:
            X, Y, Z = sh.shape[:3]
            n_missing = target_n_coefs - n_coefs
            sh = np.concatenate((sh, np.zeros((X, Y, Z, n_missing))), axis=-1)
    if sh_basis != target_basis:
        raise NotImplementedError(
            'converting SH from the {} to the {} basis needs scilpy, which is '
            'not available here; provide {} coefficients'.format(
                sh_basis, target_basis, target_basis))
    return sh
