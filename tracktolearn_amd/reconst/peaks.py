"""SH -> SF evaluation and fODF peak extraction on the GPU.

``BaseEnv._load_files`` (TrackToLearn/environments/env.py:405-432) projects
every voxel's SH coefficients on a hemisphere, takes the spherical function's
local maxima (scilpy ``get_maximas(data, sphere, B, 0.1, 0)`` = dipy
``peak_directions`` with relative threshold 0.1 and 25 degree minimum
separation), keeps at most 5 peaks, scales them by value / first value and
stores 15 floats per voxel.  The reference does this voxel by voxel in Python
(minutes at 145^3); here the projection is one (n_vox x C) @ (C x V) GEMM on
the matrix cores (PyTorch-ROCm) and maxima / thresholds / separation are
vectorised torch ops.

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


def sh_to_sf_matrix(vertices, sh_order):
    """(n_coef, V) matrix B with SF = SH @ B for the legacy descoteaux07 real
    basis of even orders <= sh_order (what dipy ``sh_to_sf_matrix(sphere,
    order, 'descoteaux07')`` returns first)."""
    from scipy.special import sph_harm_y
    v = np.asarray(vertices, dtype=np.float64)
    polar = np.arccos(np.clip(v[:, 2], -1.0, 1.0))
    azim = np.arctan2(v[:, 1], v[:, 0])
    rows = []
    for l in range(0, int(sh_order) + 1, 2):
        for m in range(-l, l + 1):
            y = sph_harm_y(l, abs(m), polar, azim)
            if m < 0:
                rows.append(np.sqrt(2.0) * y.real)
            elif m == 0:
                rows.append(y.real)
            else:
                rows.append(np.sqrt(2.0) * y.imag)
    return np.stack(rows)


@torch.no_grad()
def peaks_from_sh(sh, npeaks=5, relative_threshold=0.1, absolute_threshold=0.0,
                  min_separation_angle=25.0, subdivisions=3, chunk=1 << 18,
                  max_candidates=16):
    """fODF peaks of an SH volume.

    sh: (X, Y, Z, C) float32 tensor (any device).  Returns (X, Y, Z, 3*npeaks)
    float32 on the same device: up to ``npeaks`` unit directions sorted by
    decreasing SF value, each scaled by value / first value; zeros where a
    voxel has no signal (sum of coefficients == 0, env.py:418) or no peak.
    """
    dev = sh.device
    X, Y, Z, C = sh.shape
    order = int(round((-3 + np.sqrt(1 + 8 * C)) / 2))
    verts, nbr = hemisphere(subdivisions)
    B = torch.from_numpy(sh_to_sf_matrix(verts, order).astype(np.float32)).to(dev)
    V = torch.from_numpy(verts.astype(np.float32)).to(dev)
    nbr = torch.from_numpy(nbr).to(dev)
    cos_sep = float(np.cos(np.deg2rad(min_separation_angle)))
    flat = sh.reshape(-1, C)
    out = torch.zeros((flat.shape[0], npeaks, 3), dtype=torch.float32, device=dev)
    K = max_candidates
    for lo in range(0, flat.shape[0], chunk):
        part = flat[lo:lo + chunk]
        sf = part @ B                                            # GEMM (MFMA)
        sf = torch.where(sf < absolute_threshold, torch.zeros_like(sf), sf)
        # local maxima on the hemisphere graph: strictly above no neighbour
        # and above at least one (dipy local_maxima), positive
        nb_vals = sf[:, nbr]                                     # (n, V, D)
        is_max = (sf[:, :, None] >= nb_vals).all(dim=2) & \
            (sf[:, :, None] > nb_vals).any(dim=2) & (sf > 0)
        cand = torch.where(is_max, sf, torch.full_like(sf, -1.0))
        vals, idx = cand.topk(K, dim=1)                          # descending
        valid = vals > 0
        # relative threshold on (value - min(odf, floor 0))
        odf_min = sf.min(dim=1, keepdim=True).values.clamp(min=0.0)
        norm = vals - odf_min
        valid &= norm >= relative_threshold * norm[:, :1]
        dirs = V[idx]                                            # (n, K, 3)
        # greedy minimum-separation pruning, antipodally symmetric
        kept = torch.zeros_like(valid)
        for i in range(K):
            ok = valid[:, i].clone()
            if i:
                cosang = (dirs[:, :i] * dirs[:, i:i + 1]).sum(dim=2).abs()
                ok &= ~((cosang > cos_sep) & kept[:, :i]).any(dim=1)
            kept[:, i] = ok
        # first npeaks kept candidates, in order
        rank = kept.cumsum(dim=1) - 1
        take = kept & (rank < npeaks)
        rows = torch.nonzero(take)
        res = torch.zeros((part.shape[0], npeaks, 3), dtype=torch.float32, device=dev)
        first = torch.where(valid[:, :1], vals[:, :1], torch.ones_like(vals[:, :1]))
        scale = vals / first
        res[rows[:, 0], rank[rows[:, 0], rows[:, 1]]] = \
            dirs[rows[:, 0], rows[:, 1]] * scale[rows[:, 0], rows[:, 1], None]
        has_signal = part.sum(dim=1) != 0
        out[lo:lo + chunk] = res * has_signal[:, None, None]
    return out.reshape(X, Y, Z, 3 * npeaks)
