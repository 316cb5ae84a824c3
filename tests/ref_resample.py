"""Plain PyTorch restatement of the arc-length resampler (numerics reference
of the HIP kernel ``k_resample``; test infrastructure only).  Same semantics
as tracktolearn_amd.oracles.oracle.resample_streamlines, any device."""
import torch


def resample_streamlines(points, lengths, nb_points=128):
    """Arc-length resampling of a padded batch.

    points (N, L, 3) float, lengths (N,) number of valid points per row (>= 2)
    -> (N, nb_points, 3): equally spaced along the polyline, first and last
    point kept.  This is what ``dipy.tracking.streamline.set_number_of_points``
    computes (oracle.py:52,70; dipy is absent -> restated from its documented
    behaviour, parity unpinned): cumulative segment lengths in float64, target
    arc length k * total / (nb_points - 1), linear interpolation inside the
    segment that contains it.
    """
    n, L, _ = points.shape
    dev = points.device
    p = points.double()
    seg = (p[:, 1:] - p[:, :-1]).norm(dim=2)                    # (N, L-1)
    steps = torch.arange(L - 1, device=dev)
    seg = seg * (steps[None, :] < (lengths - 1)[:, None])
    cum = torch.cat([torch.zeros(n, 1, dtype=torch.float64, device=dev),
                     seg.cumsum(dim=1)], dim=1)                   # (N, L)
    total = cum.gather(1, (lengths - 1).clamp(min=0)[:, None])   # (N, 1)
    k = torch.arange(nb_points, device=dev, dtype=torch.float64)
    target = total * (k / (nb_points - 1))[None, :]              # (N, nb)
    # segment j with cum[j] <= t < cum[j+1]
    j = torch.searchsorted(cum[:, 1:].contiguous(), target.contiguous(),
                           right=True)
    j = torch.minimum(j, (lengths - 2).clamp(min=0)[:, None])
    c0 = cum.gather(1, j)
    c1 = cum.gather(1, j + 1)
    denom = (c1 - c0)
    ratio = torch.where(denom > 0, (target - c0) / denom,
                        torch.zeros_like(denom))
    a = p.gather(1, j[:, :, None].expand(-1, -1, 3))
    b = p.gather(1, (j + 1)[:, :, None].expand(-1, -1, 3))
    out = a + ratio[:, :, None] * (b - a)
    last = p.gather(1, (lengths - 1).clamp(min=0)[:, None, None].expand(-1, 1, 3))
    out[:, -1:] = last
    return out.to(points.dtype)
