"""SH -> SF evaluation and fODF peak extraction on the GPU.

``BaseEnv._load_files`` (TrackToLearn/environments/env.py:405-432) projects
every voxel's SH coefficients on a hemisphere, takes the spherical function's
local maxima (scilpy ``get_maximas(data, sphere, B, 0.1, 0)`` = dipy
``peak_directions`` with relative threshold 0.1 and 25 degree minimum
separation), keeps at most 5 peaks, scales them by value / first value and
stores 15 floats per voxel.  The reference does this voxel by voxel in Python
(minutes at 145^3); here it is one
hand-written HIP kernel (``k_peaks`` in csrc/ttl_peaks.hip, C ABI
``ttl_peaks_from_sh``): one wavefront per voxel, the SH->SF matrix staged in
LDS, maxima / thresholds / separation decided wave-wide.  (A plain PyTorch
fp32 restatement lives in tests/ref_peaks.py as the numerics reference.)

PARITY UNPINNED: the reference's sphere is dipy's ``repulsion724`` vertex
table (data that cannot be regenerated offline) and the basis / peak code is
dipy + scilpy (absent).  This module uses an icosphere (3 subdivisions, 321
hemisphere vertices) and the published legacy descoteaux07 real basis
(Descoteaux et al. 2007: sqrt(2) Re Y_l^|m| for m < 0, Y_l^0, sqrt(2) Im Y_l^m
for m > 0; even l, m = -l..l).  Peak directions therefore differ from the
reference's by up to the angular resolution of the spheres (~7 degrees).
"""
import numpy as np
import torch


def icosphere(subdivisions=3):
    """Unit icosphere: (vertices (V, 3), faces (F, 3))."""
    t = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
                  [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    verts = [tuple(x) for x in v]
    for _ in range(subdivisions):
        cache = {}
        new_faces = []

        def midpoint(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (np.array(verts[a]) + np.array(verts[b])) / 2.0
                verts.append(tuple(m / np.linalg.norm(m)))
                cache[key] = len(verts) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            new_faces += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = np.array(new_faces)
    return np.array(verts), f


def hemisphere(subdivisions=3):
    """One vertex per antipodal pair of the icosphere, with the neighbour
    table of the identified graph: (vertices (V, 3) float64, neighbours (V, D)
    int64 padded with the vertex itself)."""
    verts, faces = icosphere(subdivisions)
    # representative of each antipodal pair: the one in the upper half
    # (ties broken lexicographically)
    key = np.round(verts, 9)
    upper = [i for i in range(len(verts))
             if tuple(key[i][::-1]) > tuple((-key[i])[::-1])]
    rep_of = {}
    lookup = {tuple(np.round(verts[i], 6)): i for i in range(len(verts))}
    for new, i in enumerate(upper):
        rep_of[i] = new
        rep_of[lookup[tuple(np.round(-verts[i], 6))]] = new
    nbrs = [set() for _ in upper]
    for a, b, c in faces:
        for x, y in ((a, b), (b, c), (c, a)):
            rx, ry = rep_of[x], rep_of[y]
            if rx != ry:
                nbrs[rx].add(ry)
                nbrs[ry].add(rx)
    deg = max(len(s) for s in nbrs)
    table = np.array([sorted(s) + [i] * (deg - len(s))
                      for i, s in enumerate(nbrs)], dtype=np.int64)
    return verts[upper], table


def real_sh_parts(sh_basis, legacy=True):
    """How the real basis function (l, m) of ``sh_basis`` is built from the
    complex harmonic Y_l^|m| (Condon-Shortley phase): returns
    ``f(l, m) -> (part, scale)`` with part 're' | 'im' | 'zonal', meaning
    ``scale * Re/Im Y_l^|m|`` (or ``Y_l^0``).  Definitions as published in
    dipy's ``real_sh_descoteaux_from_index`` / ``real_sh_tournier_from_index``
    docstrings (dipy itself is absent offline -> parity unpinned):

      descoteaux07 legacy      m<0: sqrt2 Re Y_l^|m|        m>0: sqrt2 Im Y_l^m
      descoteaux07 non-legacy  m<0: sqrt2 Re Y_l^m (signed: (-1)^m sqrt2 Re Y_l^|m|)
      tournier07 legacy        m<0: Im Y_l^|m|              m>0: Re Y_l^m   (no sqrt2)
      tournier07 non-legacy    m<0: sqrt2 Im Y_l^|m|        m>0: sqrt2 Re Y_l^m  (MRtrix3)
    """
    r2 = np.sqrt(2.0)
    if sh_basis == 'descoteaux07':
        def parts(l, m):
            if m == 0:
                return 'zonal', 1.0
            if m < 0:
                return 're', r2 * (1.0 if legacy or m % 2 == 0 else -1.0)
            return 'im', r2
    elif sh_basis == 'tournier07':
        def parts(l, m):
            if m == 0:
                return 'zonal', 1.0
            scale = 1.0 if legacy else r2
            return ('im', scale) if m < 0 else ('re', scale)
    else:
        raise ValueError(f'unknown SH basis {sh_basis!r}')
    return parts


def sh_to_sf_matrix(vertices, sh_order, sh_basis='descoteaux07', legacy=True):
    """(n_coef, V) matrix B with SF = SH @ B for a real symmetric basis of
    even orders <= sh_order, coefficients ordered by l then m = -l..l (what
    dipy ``sh_to_sf_matrix(sphere, order, basis)`` returns first).  Default:
    the legacy descoteaux07 basis the environment tracks in."""
    from scipy.special import sph_harm_y
    v = np.asarray(vertices, dtype=np.float64)
    polar = np.arccos(np.clip(v[:, 2], -1.0, 1.0))
    azim = np.arctan2(v[:, 1], v[:, 0])
    parts = real_sh_parts(sh_basis, legacy)
    rows = []
    for l in range(0, int(sh_order) + 1, 2):
        for m in range(-l, l + 1):
            y = sph_harm_y(l, abs(m), polar, azim)
            part, scale = parts(l, m)
            rows.append(scale * (y.imag if part == 'im' else y.real))
    return np.stack(rows)


def peaks_from_sh(sh, npeaks=5, relative_threshold=0.1, absolute_threshold=0.0,
                  min_separation_angle=25.0, subdivisions=3, max_candidates=16):
    """fODF peaks of an SH volume on the GPU (``ttl_peaks_from_sh`` /
    ``k_peaks``: one wavefront per voxel, SH->SF matrix in LDS).

    sh: (X, Y, Z, C) float32 CUDA tensor.  Returns (X, Y, Z, 3*npeaks) float32
    on the same device: up to ``npeaks`` unit directions sorted by decreasing
    SF value, each scaled by value / first value; zeros where a voxel has no
    signal (sum of coefficients == 0, env.py:418) or no peak.
    """
    import ctypes as C

    from tracktolearn_amd import _lib
    lib = _lib.load()                       # raises without the HIP extension
    if not sh.is_cuda:
        raise _lib.TTLError('peaks_from_sh needs a CUDA tensor: there is no CPU path')
    dev = sh.device
    X, Y, Z, n_coef = sh.shape
    order = int(round((-3 + np.sqrt(1 + 8 * n_coef)) / 2))
    verts, nbr = hemisphere(subdivisions)
    B = torch.from_numpy(
        np.ascontiguousarray(sh_to_sf_matrix(verts, order), dtype=np.float32)).to(dev)
    if B.shape[0] != n_coef:
        raise ValueError(f'{n_coef} coefficients are not a full even SH order')
    V = torch.from_numpy(np.ascontiguousarray(verts, dtype=np.float32)).to(dev)
    N = torch.from_numpy(np.ascontiguousarray(nbr, dtype=np.int32)).to(dev)
    flat = sh.reshape(-1, n_coef).contiguous().to(torch.float32)
    out = torch.empty((flat.shape[0], 3 * npeaks), dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        _lib.check(lib.ttl_peaks_from_sh(
            flat.data_ptr(), flat.shape[0], n_coef, B.data_ptr(), V.data_ptr(),
            N.data_ptr(), V.shape[0], N.shape[1], npeaks,
            float(relative_threshold), float(absolute_threshold),
            float(np.cos(np.deg2rad(min_separation_angle))), int(max_candidates),
            out.data_ptr(), stream), 'ttl_peaks_from_sh')
    return out.reshape(X, Y, Z, 3 * npeaks)
