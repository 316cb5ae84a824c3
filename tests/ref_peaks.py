"""Plain PyTorch fp32 restatement of the fODF peak extraction (numerics
reference of the HIP kernel ``k_peaks``; test infrastructure only).  Same
semantics as tracktolearn_amd.reconst.peaks.peaks_from_sh."""
import numpy as np
import torch

from tracktolearn_amd.reconst.peaks import hemisphere, sh_to_sf_matrix


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
