"""OracleSingleton: batched streamline scoring on the device.

Mirror of TrackToLearn/oracles/oracle.py.  The reference resamples every
streamline to 128 points on the host with dipy, stages the 127 segment
vectors through pinned memory in batches of 4096 and runs the transformer
under autocast; here resampling, differencing and batching stay on the GPU.

Reference quirk kept on purpose: with more than one batch,
the final *partial* batch is never evaluated and its scores stay 0
(oracle.py:62-84); ``drop_tail=False`` scores everything.
"""
import contextlib

import torch

from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle


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


class OracleSingleton:
    """One oracle per process (oracle.py:11-37)."""
    _self = None

    def __new__(cls, *args, **kwargs):
        if cls._self is None:
            print('Instanciating new Oracle, should only happen once.')
            cls._self = super().__new__(cls)
        return cls._self

    def __init__(self, checkpoint: str, device, batch_size=4096):
        ckpt = torch.load(checkpoint, map_location=device, weights_only=True)
        models = {'TransformerOracle': TransformerOracle}
        self.model = models[ckpt['hyper_parameters']['name']] \
            .load_from_checkpoint(ckpt).to(device)
        self.model.eval()
        self.batch_size = batch_size
        self.device = torch.device(device)
        self.drop_tail = True

    @classmethod
    def reset(cls):
        """Forget the process-wide instance (tests)."""
        cls._self = None

    def predict(self, points, lengths=None):
        """Scores (N,) float32 on the device for a padded batch of
        streamlines ``points`` (N, L, 3); ``lengths`` defaults to L for every
        row (the env passes equal-length histories, oracle_reward.py:82)."""
        n = points.shape[0]
        result = torch.zeros(n, dtype=torch.float32, device=self.device)
        if n == 0:
            return result
        points = points.to(self.device)
        if lengths is None:
            lengths = torch.full((n,), points.shape[1], dtype=torch.long,
                                 device=self.device)
        bs = self.batch_size
        n_full = n // bs
        if n <= bs:
            spans = [(0, n)]
        else:
            spans = [(i * bs, (i + 1) * bs) for i in range(n_full)]
            if not self.drop_tail and n % bs:
                spans.append((n_full * bs, n))
        autocast = (torch.autocast('cuda') if self.device.type == 'cuda'
                    else contextlib.nullcontext())
        for lo, hi in spans:
            data = resample_streamlines(points[lo:hi], lengths[lo:hi], 128)
            dirs = (data[:, 1:] - data[:, :-1]).float()
            with autocast, torch.no_grad():
                result[lo:hi] = self.model(dirs).float()
        return result
