"""CPU tests of the file formats either side of the path."""
import numpy as np
import pytest

from tracktolearn_amd.io import nifti, streamlines as sio
from tracktolearn_amd.tractogram import LazyTractogram, Tractogram, TractogramItem


@pytest.mark.parametrize('ext', ['.nii', '.nii.gz'])
@pytest.mark.parametrize('dtype', [np.float32, np.uint8, np.int16, np.float64])
def test_nifti_round_trip(tmp_path, ext, dtype):
    rng = np.random.RandomState(0)
    data = (rng.standard_normal((5, 6, 7, 4)) * 20).astype(dtype)
    aff = np.array([[-1.25, 0, 0, 90.0], [0, 1.25, 0, -126.0],
                    [0, 0, 1.25, -72.0], [0, 0, 0, 1]])
    p = str(tmp_path / ('vol' + ext))
    nifti.save(p, data, aff)
    img = nifti.load(p)
    assert img.shape == (5, 6, 7, 4)
    assert np.allclose(img.affine, aff)
    assert np.allclose(img.get_zooms(), (1.25, 1.25, 1.25))
    got = img.get_fdata(dtype=np.float32)
    assert got.dtype == np.float32 and np.array_equal(got, data.astype(np.float32))


def test_nifti_qform_and_scaling(tmp_path):
    """A hand-built big-endian header with a qform and scl_slope/inter."""
    import struct
    data = np.arange(24, dtype='>i2').reshape((2, 3, 4), order='F')
    hdr = bytearray(348)
    struct.pack_into('>i', hdr, 0, 348)
    struct.pack_into('>8h', hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into('>h', hdr, 70, 4)
    struct.pack_into('>8f', hdr, 76, 1.0, 2.0, 2.0, 2.0, 1, 1, 1, 1)
    struct.pack_into('>f', hdr, 108, 352.0)
    struct.pack_into('>2f', hdr, 112, 0.5, 10.0)
    struct.pack_into('>2h', hdr, 252, 1, 0)
    struct.pack_into('>6f', hdr, 256, 0.0, 0.0, 0.0, 5.0, 6.0, 7.0)
    hdr[344:348] = b'n+1\0'
    p = tmp_path / 'be.nii'
    p.write_bytes(bytes(hdr) + b'\0' * 4 + data.tobytes(order='F'))
    img = nifti.load(str(p))
    assert np.allclose(img.affine, np.array([[2, 0, 0, 5], [0, 2, 0, 6],
                                             [0, 0, 2, 7], [0, 0, 0, 1.0]]))
    assert np.allclose(img.get_fdata(), data.astype(np.float64) * 0.5 + 10.0)


def _some_lines(rng, n):
    return [rng.uniform(0, 30, (rng.randint(2, 40), 3)).astype(np.float32)
            for _ in range(n)]


def test_trk_round_trip(tmp_path):
    rng = np.random.RandomState(1)
    aff = np.array([[-2.0, 0, 0, 80.0], [0, 2.0, 0, -100.0], [0, 0, 2.0, -60.0],
                    [0, 0, 0, 1]])
    lines = _some_lines(rng, 17)                       # already in RAS+mm
    seeds = rng.uniform(size=(17, 3)).astype(np.float32)
    tg = Tractogram(lines, {'seeds': seeds})
    header = sio.create_tractogram_header(aff, (40, 50, 30), (2.0, 2.0, 2.0))
    assert header['voxel_order'] == 'LAS'
    p = str(tmp_path / 'out.trk')
    assert sio.save(tg, p, header) == 17
    back, hdr = sio.load_trk(p)
    assert hdr['nb_streamlines'] == 17 and hdr['voxel_order'] == 'LAS'
    assert np.allclose(hdr['voxel_to_rasmm'], aff)
    for a, b in zip(back.streamlines, lines):
        assert np.allclose(a, b, atol=1e-4)
    assert np.allclose(back.data_per_streamline['seeds'], seeds)


def test_lazy_tractogram_applies_affine_to_rasmm(tmp_path):
    """Items of a lazy tractogram are brought to RAS+mm with affine_to_rasmm
    before writing, as nibabel's save does for the Tracker's output."""
    rng = np.random.RandomState(2)
    aff = np.diag([1.5, 1.5, 1.5, 1.0])
    aff[:3, 3] = [10, 20, 30]
    vox_lines = _some_lines(rng, 5)

    def gen():
        for s in vox_lines:
            yield TractogramItem(s, {}, {})
    lazy = LazyTractogram.from_data_func(gen)
    lazy.affine_to_rasmm = aff
    header = sio.create_tractogram_header(aff, (40, 40, 40), (1.5, 1.5, 1.5))
    for name in ('a.trk', 'a.tck'):
        p = str(tmp_path / name)
        assert sio.save(lazy, p, header) == 5
        back = (sio.load_trk(p) if name.endswith('trk') else sio.load_tck(p))[0]
        for got, s in zip(back.streamlines, vox_lines):
            assert np.allclose(got, s.astype(np.float64) * 1.5 + [10, 20, 30], atol=1e-4)


def test_tck_empty_and_format(tmp_path):
    p = str(tmp_path / 'e.tck')
    assert sio.save(Tractogram([], {}), p) == 0
    raw = open(p, 'rb').read()
    assert raw.startswith(b'mrtrix tracks\n') and b'datatype: Float32LE' in raw
    back, fields = sio.load_tck(p)
    assert len(back) == 0 and int(fields['count']) == 0
    with pytest.raises(ValueError):
        sio.save(Tractogram([], {}), str(tmp_path / 'x.vtk'))
