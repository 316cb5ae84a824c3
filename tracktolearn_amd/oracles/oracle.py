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
import os
import sys

import torch

from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle


def resample_streamlines(points, lengths, nb_points=128):
    """Arc-length resampling of a padded batch on the GPU
    (``ttl_resample_streamlines`` / ``k_resample``, one wavefront per row).

    points (N, L, 3) float32 CUDA tensor, lengths (N,) number of valid points
    per row (>= 2) -> (N, nb_points, 3) float32: equally spaced along the
    polyline, first and last point kept.  This is what
    ``dipy.tracking.streamline.set_number_of_points`` computes (oracle.py:52,70;
    dipy is absent -> restated from its documented behaviour, parity unpinned):
    cumulative segment lengths in float64, target arc length
    k * total / (nb_points - 1), linear interpolation inside the segment that
    contains it.  (Plain PyTorch restatement: tests/ref_resample.py.)
    """
    import ctypes as C

    from tracktolearn_amd import _lib
    lib = _lib.load()
    if not points.is_cuda:
        raise _lib.TTLError('resample_streamlines needs CUDA tensors: there is no CPU path')
    n, L, _ = points.shape
    pts = points.to(torch.float32)
    if pts.stride(2) != 1 or pts.stride(1) != 3:
        pts = pts.contiguous()
    lengths = lengths.to(points.device)
    if lengths.dtype == torch.int32:
        l32, l64 = lengths.contiguous(), None
    else:
        l32, l64 = None, lengths.to(torch.int64).contiguous()
    out = torch.empty((n, nb_points, 3), dtype=torch.float32, device=points.device)
    if n == 0:
        return out
    stream = C.c_void_p(torch.cuda.current_stream(points.device).cuda_stream)
    with torch.cuda.device(points.device):
        _lib.check(lib.ttl_resample_streamlines(
            pts.data_ptr(), pts.stride(0),
            l32.data_ptr() if l32 is not None else None,
            l64.data_ptr() if l64 is not None else None,
            n, L, int(nb_points), out.data_ptr(), stream), 'ttl_resample_streamlines')
    return out


class OracleSingleton:
    """One oracle per process (oracle.py:11-37)."""
    _self = None

    def __new__(cls, *args, **kwargs):
        if cls._self is None:
            # (the reference prints this to stdout, oracle.py:18; stderr here: bench.py's
            # stdout is one JSON line)
            print('Instanciating new Oracle, should only happen once.', file=sys.stderr)
            cls._self = super().__new__(cls)
        return cls._self

    def __init__(self, checkpoint: str, device, batch_size=4096, resample=None):
        ckpt = torch.load(checkpoint, map_location=device, weights_only=True)
        models = {'TransformerOracle': TransformerOracle}
        self.model = models[ckpt['hyper_parameters']['name']] \
            .load_from_checkpoint(ckpt).to(device)
        self.model.eval()
        self.batch_size = batch_size
        self.device = torch.device(device)
        self.drop_tail = True
        #: on the GPU a batch is padded with zero rows to a multiple of this many
        #: rows (0: never): the number of rows scored changes at every step (the
        #: streamlines that just stopped / are still active), and every new row
        #: count is a new set of GEMM shapes for which hipBLASLt picks kernels
        #: afresh; with 512-row buckets a 4 096-row batch size has 8 shapes.
        #: Rows are independent inside the network (attention stays inside a
        #: streamline), so the scores of the real rows do not depend on the padding.
        self.pad_rows = 512 if self.device.type == 'cuda' else 0
        self._pad_buf = None
        #: the network as one hand-written kernel (oracles/fused_net.py) when it has
        #: the reference's architecture; ``TTL_ORACLE_FUSED=0`` keeps the PyTorch
        #: module under autocast (A/B runs; other architectures always do)
        self.net = None
        if self.device.type == 'cuda' and os.environ.get('TTL_ORACLE_FUSED', '1') != '0':
            from tracktolearn_amd.oracles.fused_net import FusedOracleNet
            if FusedOracleNet.supports(self.model):
                self.net = FusedOracleNet(self.model, self.device)
        # the resampler is the HIP kernel; CPU-only tests of the batching
        # logic inject the PyTorch restatement from tests/
        self._resample = resample if resample is not None else resample_streamlines

    @classmethod
    def reset(cls):
        """Forget the process-wide instance (tests)."""
        cls._self = None

    def _spans(self, n):
        """Row ranges the reference evaluates (oracle.py:62-84): one batch when
        n <= batch_size, otherwise the full batches only (the partial last one
        keeps score 0) unless ``drop_tail`` is off."""
        bs = self.batch_size
        if n <= bs:
            return [(0, n)]
        spans = [(i * bs, (i + 1) * bs) for i in range(n // bs)]
        if not self.drop_tail and n % bs:
            spans.append((n // bs * bs, n))
        return spans

    def predict_history(self, history, ids, id_stride, n, n_points, lin=None):
        """``predict`` for rows of the env's history buffer, without the
        intermediate tensors: streamline ``ids[r * id_stride]`` (device int32
        pointer; None: row r) of ``history`` (rows of (max_nb_steps + 1) x 3
        float32), its first ``n_points`` points, mapped by the 3x3 ``lin`` (host
        floats, p @ lin) when given.  Two launches per batch -- gather + map +
        resample + difference (``ttl_oracle_segments``), then the network --
        and only with the fused network.  Returns ``(scores, n_scored)``: rows
        from ``n_scored`` on were not evaluated (the reference's partial last
        batch) and hold 0."""
        import ctypes as C

        from tracktolearn_amd import _lib
        if self.net is None:
            raise _lib.TTLError('predict_history needs the fused network')
        lib = _lib.load()
        spans = self._spans(n)
        n_scored = spans[-1][1]
        dirs = torch.empty((n_scored, 127, 3), dtype=torch.float32, device=self.device)
        lin_c = None
        if lin is not None:
            lin_c = (C.c_float * 9)(*[float(v) for v in lin])
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            _lib.check(lib.ttl_oracle_segments(
                history.data_ptr(), history.stride(0), ids, id_stride, n_scored, n_points,
                lin_c, 128, dirs.data_ptr(), stream), 'ttl_oracle_segments')
        if len(spans) == 1 and n_scored == n:
            return self.net(dirs), n
        result = torch.zeros(n, dtype=torch.float32, device=self.device)
        for lo, hi in spans:
            result[lo:hi] = self.net(dirs[lo:hi])
        return result, n_scored

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
        spans = self._spans(n)
        autocast = (torch.autocast('cuda') if self.device.type == 'cuda'
                    else contextlib.nullcontext())
        for lo, hi in spans:
            data = self._resample(points[lo:hi], lengths[lo:hi], 128)
            dirs = (data[:, 1:] - data[:, :-1]).float()
            if self.net is not None:
                result[lo:hi] = self.net(dirs)
                continue
            rows = hi - lo
            padded = -(-rows // self.pad_rows) * self.pad_rows if self.pad_rows else rows
            if padded != rows:
                if self._pad_buf is None or self._pad_buf.shape[0] < padded or \
                        self._pad_buf.shape[1:] != dirs.shape[1:]:
                    self._pad_buf = torch.zeros((max(padded, self.batch_size),) + tuple(dirs.shape[1:]),
                                                dtype=dirs.dtype, device=dirs.device)
                self._pad_buf[:rows].copy_(dirs)
                self._pad_buf[rows:padded].zero_()
                dirs = self._pad_buf[:padded]
            with autocast, torch.no_grad():
                result[lo:hi] = self.model(dirs).float()[:rows]
        return result
