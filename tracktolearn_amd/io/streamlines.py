"""TrackVis (.trk) and MRtrix (.tck) tractogram files.

nibabel is absent from this image; the two formats ``ttl_track.py`` can write
(TrackToLearn/runners/ttl_track.py:178-186) are implemented from their public
specifications.  ``save`` follows what ``nib.streamlines.save(tractogram,
path, header=header)`` does with a (lazy) tractogram: points are first taken
to RAS+mm with ``tractogram.affine_to_rasmm`` and then, for .trk, to TrackVis
"voxmm" (voxel * voxel size, origin at the voxel corner) with the header's
vox->rasmm.  [recollection of nibabel's behaviour -- parity unpinned]
"""
import struct

import numpy as np

from tracktolearn_amd.tractogram import Tractogram

TRK_HEADER_SIZE = 1000


def axcodes(affine):
    """RAS axis codes of a vox->rasmm affine (closest-axis rule)."""
    R = np.asarray(affine)[:3, :3]
    codes = []
    labels = (('L', 'R'), ('P', 'A'), ('I', 'S'))
    for col in range(3):
        v = R[:, col]
        ax = int(np.argmax(np.abs(v)))
        codes.append(labels[ax][1 if v[ax] >= 0 else 0])
    return ''.join(codes)


def create_tractogram_header(affine, dimensions, voxel_sizes, voxel_order=None):
    """The dict ``dipy.io.utils.create_tractogram_header`` builds from a
    reference image (ttl_track.py:182-183)."""
    return {'voxel_to_rasmm': np.asarray(affine, dtype=np.float64),
            'dimensions': tuple(int(d) for d in dimensions[:3]),
            'voxel_sizes': tuple(float(v) for v in voxel_sizes[:3]),
            'voxel_order': voxel_order or axcodes(affine)}


def _to_rasmm(tractogram):
    A = getattr(tractogram, 'affine_to_rasmm', None)
    identity = A is None or np.array_equal(np.asarray(A), np.eye(4))
    A = None if identity else np.asarray(A, dtype=np.float64)
    for item in tractogram:
        s = np.asarray(item.streamline, dtype=np.float64)
        if A is not None:
            s = s @ A[:3, :3].T + A[:3, 3]
        yield s, item.data_for_streamline


def save_trk(tractogram, path, header):
    vox2ras = np.asarray(header['voxel_to_rasmm'], dtype=np.float64)
    vs = np.asarray(header['voxel_sizes'], dtype=np.float64)
    ras2vox = np.linalg.inv(vox2ras)
    props = None
    count = 0
    with open(path, 'wb') as f:
        f.write(b'\0' * TRK_HEADER_SIZE)
        for s, per in _to_rasmm(tractogram):
            if props is None:
                props = [(k, int(np.asarray(v).size)) for k, v in sorted(per.items())]
            vox = s @ ras2vox[:3, :3].T + ras2vox[:3, 3]
            voxmm = ((vox + 0.5) * vs).astype('<f4')
            f.write(struct.pack('<i', len(voxmm)))
            f.write(voxmm.tobytes())
            for k, _ in props:
                f.write(np.asarray(per[k], dtype='<f4').reshape(-1).tobytes())
            count += 1
        props = props or []
        names = []
        for k, n in props:          # nibabel names multi-valued properties k, then pads
            names += [k] + [k] * (n - 1)
        if len(names) > 10:
            raise ValueError('TRK holds at most 10 property values per streamline')
        hdr = bytearray(TRK_HEADER_SIZE)
        hdr[0:6] = b'TRACK\0'
        struct.pack_into('<3h', hdr, 6, *header['dimensions'])
        struct.pack_into('<3f', hdr, 12, *vs)
        struct.pack_into('<h', hdr, 238, len(names))
        for i, nm in enumerate(names):
            raw = nm.encode('latin1')[:19]
            hdr[240 + 20 * i:240 + 20 * i + len(raw)] = raw
        struct.pack_into('<16f', hdr, 440, *vox2ras.reshape(-1))
        order = header['voxel_order'].encode('latin1')[:3]
        hdr[948:948 + len(order)] = order
        struct.pack_into('<i', hdr, 988, count)
        struct.pack_into('<i', hdr, 992, 2)
        struct.pack_into('<i', hdr, 996, TRK_HEADER_SIZE)
        f.seek(0)
        f.write(bytes(hdr))
    return count


def load_trk(path):
    """Read back a .trk: streamlines in RAS+mm, properties, header dict."""
    raw = open(path, 'rb').read()
    if raw[:5] != b'TRACK':
        raise ValueError(f'{path}: not a TRK file')
    dims = struct.unpack('<3h', raw[6:12])
    vs = np.array(struct.unpack('<3f', raw[12:24]), dtype=np.float64)
    n_scalars = struct.unpack('<h', raw[36:38])[0]
    n_props = struct.unpack('<h', raw[238:240])[0]
    names = [raw[240 + 20 * i:260 + 20 * i].split(b'\0')[0].decode('latin1')
             for i in range(n_props)]
    vox2ras = np.array(struct.unpack('<16f', raw[440:504]), dtype=np.float64).reshape(4, 4)
    n_count = struct.unpack('<i', raw[988:992])[0]
    pos = TRK_HEADER_SIZE
    lines, props = [], []
    while pos < len(raw):
        n = struct.unpack('<i', raw[pos:pos + 4])[0]
        pos += 4
        pts = np.frombuffer(raw, '<f4', n * (3 + n_scalars), pos).reshape(n, 3 + n_scalars)
        pos += 4 * n * (3 + n_scalars)
        props.append(np.frombuffer(raw, '<f4', n_props, pos).copy())
        pos += 4 * n_props
        vox = pts[:, :3].astype(np.float64) / vs - 0.5
        lines.append((vox @ vox2ras[:3, :3].T + vox2ras[:3, 3]).astype(np.float32))
    assert n_count in (0, len(lines))
    per = {}
    if n_props:
        P = np.stack(props) if props else np.zeros((0, n_props), np.float32)
        for nm in dict.fromkeys(names):
            cols = [i for i, x in enumerate(names) if x == nm]
            per[nm] = P[:, cols]
    header = {'voxel_to_rasmm': vox2ras, 'dimensions': dims, 'voxel_sizes': tuple(vs),
              'voxel_order': raw[948:951].decode('latin1'), 'nb_streamlines': n_count}
    return Tractogram(lines, per), header


def save_tck(tractogram, path, header=None):
    count = 0
    chunks = []
    for s, _ in _to_rasmm(tractogram):
        chunks.append(s.astype('<f4').tobytes())
        chunks.append(np.full(3, np.nan, '<f4').tobytes())
        count += 1
    chunks.append(np.full(3, np.inf, '<f4').tobytes())
    lines = ['mrtrix tracks', f'count: {count:010d}', 'datatype: Float32LE']
    # the header states its own length in the "file" entry
    body = '\n'.join(lines) + '\n'
    offset = len(body) + len('file: . ') + 12 + len('\nEND\n')
    text = body + f'file: . {offset:<12d}'.rstrip() + '\nEND\n'
    text = text.ljust(offset, '\n') if len(text) < offset else text
    offset = len(text)
    with open(path, 'wb') as f:
        f.write(text.encode('latin1'))
        for c in chunks:
            f.write(c)
    return count


def load_tck(path):
    raw = open(path, 'rb').read()
    head_end = raw.index(b'END\n') + 4
    fields = dict(l.split(': ', 1) for l in raw[:head_end].decode('latin1')
                  .split('\n') if ': ' in l)
    offset = int(fields['file'].split()[1])
    data = np.frombuffer(raw, '<f4', offset=offset).reshape(-1, 3)
    lines, cur = [], 0
    for i in range(len(data)):
        if np.isinf(data[i, 0]):
            break
        if np.isnan(data[i, 0]):
            lines.append(data[cur:i].copy())
            cur = i + 1
    return Tractogram(lines, {}), fields


def save(tractogram, path, header=None):
    """``nib.streamlines.save`` for the two supported formats."""
    lower = str(path).lower()
    if lower.endswith('.trk'):
        if header is None:
            raise ValueError('a .trk file needs a reference header')
        return save_trk(tractogram, path, header)
    if lower.endswith('.tck'):
        return save_tck(tractogram, path, header)
    raise ValueError('output must be .trk or .tck')
