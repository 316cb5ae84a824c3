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
