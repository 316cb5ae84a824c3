"""Minimal NIfTI-1 reader/writer (single-file .nii / .nii.gz).

nibabel is absent from this image; ``ttl_track.py``'s inputs (fODF SH volume,
seeding mask, tracking mask -- TrackToLearn/environments/env.py:390,435-447)
are NIfTI-1 files, so the subset of the format they use is implemented here
from the NIfTI-1 specification: 348-byte header, sform / qform / pixdim
affines (sform preferred, as ``nibabel``'s ``get_best_affine``), scl_slope /
scl_inter scaling, Fortran-ordered data block at ``vox_offset``.
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32,
           64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32,
           1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


class NiftiImage(object):
    """Data + affine + zooms of one image (what the env loader needs from
    ``nib.load``: ``get_fdata``, ``affine``, ``header.get_zooms()``, shape)."""

    def __init__(self, dataobj, affine, zooms, slope=None, inter=None):
        self.dataobj = dataobj
        self.affine = affine
        self.zooms = tuple(float(z) for z in zooms)
        self._slope, self._inter = slope, inter
        self.shape = dataobj.shape

    def get_zooms(self):
        return self.zooms

    def get_fdata(self, dtype=np.float64):
        data = self.dataobj.astype(dtype)
        if self._slope not in (None, 0.0) and np.isfinite(self._slope):
            if self._slope != 1.0 or (self._inter or 0.0) != 0.0:
                data = data * dtype(self._slope) + dtype(self._inter or 0.0)
        return data


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith('.gz') else open(path, mode)


def _quaternion_affine(b, c, d, qfac, pixdim, offset):
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 0 else 0.0
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    scale = np.array([pixdim[1], pixdim[2], pixdim[3] * (qfac if qfac else 1.0)])
    A = np.eye(4)
    A[:3, :3] = R * scale
    A[:3, 3] = offset
    return A


def load(path):
    """Read a NIfTI-1 image."""
    with _open(path, 'rb') as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError(f'{path}: not a NIfTI-1 file')
    end = '<' if struct.unpack('<i', raw[:4])[0] == 348 else '>'
    if struct.unpack(end + 'i', raw[:4])[0] != 348:
        raise ValueError(f'{path}: bad sizeof_hdr')
    if raw[344:347] not in (b'n+1', b'ni1'):
        raise ValueError(f'{path}: bad NIfTI-1 magic')
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype = struct.unpack(end + 'h', raw[70:72])[0]
    pixdim = struct.unpack(end + '8f', raw[76:108])
    vox_offset = int(struct.unpack(end + 'f', raw[108:112])[0])
    slope, inter = struct.unpack(end + '2f', raw[112:120])
    qform_code, sform_code = struct.unpack(end + '2h', raw[252:256])
    quat = struct.unpack(end + '6f', raw[256:280])
    srow = np.array(struct.unpack(end + '12f', raw[280:328]), dtype=np.float64)
    if datatype not in _DTYPES:
        raise ValueError(f'{path}: unsupported datatype code {datatype}')
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    count = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=count, offset=max(vox_offset, 352))
    data = data.reshape(shape, order='F')
    if sform_code > 0:
        affine = np.eye(4)
        affine[:3, :] = srow.reshape(3, 4)
    elif qform_code > 0:
        affine = _quaternion_affine(quat[0], quat[1], quat[2], pixdim[0],
                                    pixdim, quat[3:6])
    else:
        affine = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0])
    return NiftiImage(data, affine, pixdim[1:4], slope, inter)


def save(path, data, affine):
    """Write ``data`` (C- or F-ordered ndarray of a supported dtype) with an
    sform affine."""
    data = np.asarray(data)
    code = _CODES.get(data.dtype.name)
    if code is None:
        raise ValueError(f'unsupported dtype {data.dtype}')
    affine = np.asarray(affine, dtype=np.float64)
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into('<8h', hdr, 40, *dim)
    struct.pack_into('<h', hdr, 70, code)
    struct.pack_into('<h', hdr, 72, data.dtype.itemsize * 8)
    zooms = np.sqrt((affine[:3, :3] ** 2).sum(axis=0))
    pixdim = [1.0] + list(zooms) + [1.0] * 4
    struct.pack_into('<8f', hdr, 76, *pixdim)
    struct.pack_into('<f', hdr, 108, 352.0)
    struct.pack_into('<2f', hdr, 112, 1.0, 0.0)
    struct.pack_into('<2h', hdr, 252, 0, 1)            # qform 0, sform 1
    struct.pack_into('<12f', hdr, 280, *affine[:3, :].reshape(-1))
    hdr[344:348] = b'n+1\0'
    with _open(path, 'wb') as f:
        f.write(bytes(hdr))
        f.write(b'\0' * 4)
        f.write(np.asfortranarray(data).tobytes(order='F'))
