"""CPU tests of the TractOracle-Net port and of the device-side resampler /
batching."""
import numpy as np
import pytest
import torch

from helpers import load_trace


def test_transformer_matches_reference_forward():
    from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle
    z = load_trace('oracle_transformer')
    model = TransformerOracle(int(z['input_size']), 1, int(z['n_head']),
                              int(z['n_layers']), 1e-4)
    sd = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files
          if k.startswith('sd/')}
    assert set(sd) == set(model.state_dict())          # same checkpoint keys
    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        y = model(torch.from_numpy(z['x'])).numpy()
    assert np.abs(y - z['y']).max() <= 1e-5


@pytest.mark.gpu
def test_transformer_matches_reference_forward_on_the_gpu():
    """The reference's `TransformerOracle.forward` vector (recorded on the CPU
    in float32) on the MI355X: float32 within 2e-5; under `torch.autocast` --
    how `OracleSingleton.predict` runs it on a GPU, oracles/oracle.py:74-81 --
    within 5e-3 of the float32 scores (fp16 GEMMs)."""
    from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle
    z = load_trace('oracle_transformer')
    model = TransformerOracle(int(z['input_size']), 1, int(z['n_head']),
                              int(z['n_layers']), 1e-4)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k].astype(np.float32))
                           for k in z.files if k.startswith('sd/')})
    model = model.cuda().eval()
    x = torch.from_numpy(z['x']).cuda()
    with torch.no_grad():
        y32 = model(x).float().cpu().numpy()
        with torch.autocast('cuda'):
            y16 = model(x).float().cpu().numpy()
    assert np.abs(y32 - z['y']).max() <= 2e-5
    assert np.abs(y16 - z['y']).max() <= 5e-3


def _resample_reference(s, n):
    """Straightforward per-streamline arc-length resampling (float64)."""
    s = s.astype(np.float64)
    seg = np.sqrt((np.diff(s, axis=0) ** 2).sum(1))
    cum = np.concatenate([[0.0], np.cumsum(seg)])
    t = np.linspace(0.0, cum[-1], n)
    out = np.stack([np.interp(t, cum, s[:, c]) for c in range(3)], axis=1)
    out[-1] = s[-1]
    return out


def _check_resampler(resample, device):
    rng = np.random.RandomState(0)
    L = 40
    pts = np.cumsum(rng.standard_normal((50, L, 3)) * 0.3 + 0.2, axis=1).astype(np.float32)
    lengths = rng.randint(2, L + 1, 50)
    lengths[:3] = [2, L, 3]
    got = resample(torch.from_numpy(pts).to(device), torch.from_numpy(lengths).to(device),
                   128).cpu().numpy()
    for i in range(50):
        want = _resample_reference(pts[i, :lengths[i]], 128)
        assert np.abs(got[i] - want).max() <= 2e-5
        assert np.array_equal(got[i, 0], pts[i, 0])
        assert np.array_equal(got[i, -1], pts[i, lengths[i] - 1])
    # equally spaced along the curve
    d = np.sqrt((np.diff(got[1].astype(np.float64), axis=0) ** 2).sum(1))
    assert d.std() / d.mean() < 0.15
    return got


def test_reference_resampler_against_numpy():
    import ref_resample
    _check_resampler(ref_resample.resample_streamlines, 'cpu')


@pytest.mark.gpu
def test_hip_resampler_against_numpy_and_the_torch_reference():
    import ref_resample
    from tracktolearn_amd.oracles.oracle import resample_streamlines
    got = _check_resampler(resample_streamlines, 'cuda')
    want = _check_resampler(ref_resample.resample_streamlines, 'cuda')
    assert np.abs(got - want).max() <= 1e-5
    # strided rows (a view of a longer history buffer), int32 lengths, ragged
    rng = np.random.RandomState(3)
    hist = torch.from_numpy(np.cumsum(rng.standard_normal((3000, 267, 3)) * 0.3, 1)
                            .astype(np.float32)).cuda()
    lengths = torch.from_numpy(rng.randint(2, 101, 3000).astype(np.int32)).cuda()
    view = hist[:, :100]
    a = resample_streamlines(view, lengths, 128)
    b = ref_resample.resample_streamlines(view, lengths.long(), 128)
    assert (a - b).abs().max().item() <= 1e-5
    with pytest.raises(RuntimeError, match='no CPU'):
        resample_streamlines(view.cpu(), lengths.cpu(), 128)


def test_predict_batching_keeps_the_reference_tail_quirk(tmp_path):
    from tracktolearn_amd.oracles.oracle import OracleSingleton
    from tracktolearn_amd.oracles.transformer_oracle import save_random_checkpoint
    ck = save_random_checkpoint(str(tmp_path / 'o.ckpt'), n_head=2, n_layers=1)
    OracleSingleton.reset()
    import ref_resample
    ref = ref_resample.resample_streamlines   # CPU run of the batching logic
    oracle = OracleSingleton(ck, torch.device('cpu'), batch_size=8, resample=ref)
    assert OracleSingleton(ck, torch.device('cpu'), resample=ref) is oracle      # singleton
    oracle.batch_size = 8
    rng = np.random.RandomState(1)
    pts = torch.from_numpy(np.cumsum(rng.standard_normal((21, 12, 3)), 1).astype(np.float32))
    full = oracle.predict(pts[:8])
    assert full.shape == (8,) and (full > 0).all() and (full < 1).all()
    part = oracle.predict(pts[:5])                 # N < batch: the only batch runs
    assert torch.allclose(part, full[:5], atol=1e-6)
    two = oracle.predict(pts[:16])                 # exact multiple: all scored
    assert (two > 0).all() and torch.allclose(two[:8], full, atol=1e-6)
    tail = oracle.predict(pts)                     # 21 = 2 * 8 + 5: tail dropped
    assert (tail[:16] > 0).all() and (tail[16:] == 0).all()
    oracle.drop_tail = False
    assert (oracle.predict(pts) > 0).all()
    OracleSingleton.reset()


def _random_oracle(n_head, n_layers, seed, ff=None):
    """A TransformerOracle with weights of a trained network's size (the
    default initialisation leaves the attention nearly uniform and the scores
    within 1e-2 of each other: too easy a target)."""
    from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle
    torch.manual_seed(seed)
    model = TransformerOracle(381, 1, n_head, n_layers, 1e-4)
    if ff is not None:
        for layer in model.bert.layers:
            layer.linear1 = torch.nn.Linear(32, ff)
            layer.linear2 = torch.nn.Linear(ff, 32)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'in_proj_weight' in name:
                p.mul_(3.0)
            elif name.endswith('bias') or 'norm' in name:
                p.add_(0.2 * torch.randn_like(p))
        model.head.weight.mul_(6.0)
    return model.eval()


def test_oracle_net_packing_layout():
    """Host-side packing of the fused network (oracles/fused_net.py): fragment
    element j of lane (r, h) of k-step s is w[r][16 s + 8 (j >> 2) + 4 h + (j & 3)],
    per-row vectors follow the accumulator's row order."""
    from tracktolearn_amd.oracles.fused_net import FusedOracleNet, pack32, pack_oracle_net, rowpack
    w = torch.arange(1024, dtype=torch.float32).view(32, 32)
    f = pack32(w)
    assert f.shape == (2, 64, 8)
    for s, lane, j in ((0, 0, 0), (1, 37, 5), (0, 63, 7), (1, 31, 3)):
        r, h = lane & 31, lane >> 5
        assert float(f[s, lane, j]) == float(w[r, 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)])
    # every k index appears once per (lane row) over the two steps and two halves
    assert sorted(f[:, [5, 37]].reshape(-1).tolist()) == w[5].tolist()
    v = torch.arange(32.0)
    rp = rowpack(v)
    assert rp.shape == (2, 16) and sorted(rp.reshape(-1).tolist()) == v.tolist()
    assert float(rp[1, 6]) == 8 * (6 >> 2) + 4 + (6 & 3)
    model = _random_oracle(4, 2, 0)
    assert FusedOracleNet.supports(model)
    p = pack_oracle_net(model, 'cpu')
    assert p['wh'].shape == (2, (8 + 4 * 64) * 512) and p['wf'].shape == (2, 288 + 2048)
    assert p['embed'].shape == (2, 16, 4) and p['pe'].shape == (4, 64, 16) and p['head'].shape == (33,)
    # an architecture the kernel does not implement keeps the PyTorch module
    from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle
    assert not FusedOracleNet.supports(TransformerOracle(381, 1, 8, 2, 1e-4))
    assert not FusedOracleNet.supports(TransformerOracle(300, 1, 4, 2, 1e-4))
    # ... and so does a feed-forward block wider than the kernels' LDS staging of b_1
    wide = _random_oracle(4, 1, 0, ff=8192 + 32)
    assert not FusedOracleNet.supports(wide) and FusedOracleNet.supports(_random_oracle(4, 1, 0, ff=8192))


@pytest.mark.gpu
@pytest.mark.parametrize('n_head,n_layers,ff', [(4, 4, None), (4, 1, None), (2, 2, 96), (1, 3, 64)])
def test_fused_oracle_net_matches_the_module(n_head, n_layers, ff):
    """`ttl_oracle_net_forward` (one wavefront per streamline, fp16 MFMA,
    everything in registers) against the PyTorch module: under
    `torch.autocast` -- the arithmetic it restates -- and in float32.  The
    fused scores may be no further from float32 than 3 x what autocast itself
    is, and within 4e-3 of autocast's."""
    from tracktolearn_amd.oracles.fused_net import FusedOracleNet
    model = _random_oracle(n_head, n_layers, 7 + n_head, ff).cuda()
    net = FusedOracleNet(model)
    g = torch.Generator().manual_seed(11)
    for n in (1, 3, 4, 257, 4096):
        # segment vectors of resampled streamlines: smooth random walks, ~0.3 long
        steps = torch.randn(n, 127, 3, generator=g) * 0.05
        base = torch.randn(n, 1, 3, generator=g) * 0.3
        dirs = (base + torch.cumsum(steps, 1) * 0.3).cuda()
        with torch.no_grad():
            y32 = model(dirs).float()
            with torch.autocast('cuda'):
                y16 = model(dirs).float()
        got = net(dirs)
        assert got.shape == (n,) and bool(((got > 0) & (got < 1)).all())
        e_auto = float((y16 - y32).abs().max())
        e_fused = float((got - y32).abs().max())
        d = float((got - y16).abs().max())
        print(f'heads {n_head} layers {n_layers} n {n}: scores {float(y32.min()):.3f}..'
              f'{float(y32.max()):.3f}, autocast err {e_auto:.2e}, fused err {e_fused:.2e}, '
              f'fused vs autocast {d:.2e}')
        assert float(y32.max() - y32.min()) > 0.02 or n < 257    # the scores do spread
        assert e_fused <= 3 * e_auto + 2e-3 and d <= 4e-3
    # rows are independent: a row's score does not depend on its batch ...
    few = net(dirs[:512])
    assert torch.equal(net(dirs[100:101]), few[100:101])
    assert torch.equal(net(dirs[512:1100]), got[512:1100])
    # ... and the two kernels (<= 512 rows: one workgroup per streamline -- keys / values
    # through LDS, the feed-forward block split over the hidden units with its four partial
    # sums added in wave order; more: one wavefront per streamline, one accumulator) agree
    # to the rounding of the fp16 score: the order of a float32 sum is all that differs
    d = float((few - got[:512]).abs().max())
    print(f'heads {n_head} layers {n_layers}: workgroup kernel vs wave kernel {d:.2e}')
    assert d <= 1e-3


@pytest.mark.gpu
def test_reference_transformer_vector_through_the_fused_net():
    """The reference's own `TransformerOracle.forward` vector
    (tests/golden/oracle_transformer.npz) through the fused kernel: within
    the 5e-3 the autocast module is held to."""
    from tracktolearn_amd.oracles.fused_net import FusedOracleNet
    from tracktolearn_amd.oracles.transformer_oracle import TransformerOracle
    z = load_trace('oracle_transformer')
    model = TransformerOracle(int(z['input_size']), 1, int(z['n_head']), int(z['n_layers']), 1e-4)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k].astype(np.float32))
                           for k in z.files if k.startswith('sd/')})
    model = model.cuda().eval()
    if not FusedOracleNet.supports(model):
        pytest.skip('golden vector of another architecture')
    y = FusedOracleNet(model)(torch.from_numpy(z['x']).cuda()).cpu().numpy()
    assert np.abs(y - z['y']).max() <= 5e-3


@pytest.mark.gpu
def test_oracle_segments_kernel():
    """`ttl_oracle_segments` (history rows -> network input in one launch)
    against the separate steps it replaces: index gather of the rows' first L
    points, `@ lin`, `ttl_resample_streamlines`, difference.  Without the 3x3
    map the result is bit-identical; with it, the map's roundings may differ
    from the BLAS's by an ulp of a coordinate."""
    import ctypes as C

    from tracktolearn_amd import _lib
    from tracktolearn_amd.oracles.oracle import resample_streamlines
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(2)
    n_hist, max_pts = 700, 267
    hist = (torch.randn(n_hist, max_pts, 3, generator=g).cumsum(1) * 0.3 + 40).to(dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    m = torch.tensor([[0.9, 0.05, 0.0], [-0.03, 1.1, 0.02], [0.01, 0.0, 0.8]])
    for n, L, stride in ((1, 2, 1), (5, 3, 2), (300, 11, 1), (700, 128, 1), (257, 267, 2),
                         (64, 200, 1)):
        ids = torch.randperm(n_hist, generator=g)[:n].int()
        pairs = torch.stack([torch.arange(n, dtype=torch.int32), ids], 1).contiguous().to(dev)
        ids_dev = ids.to(dev)
        for lin in (None, m):
            pts = hist[ids_dev.long(), :L]
            if lin is not None:
                pts = pts @ lin.to(dev)
            data = resample_streamlines(pts, torch.full((n,), L, dtype=torch.long, device=dev))
            want = data[:, 1:] - data[:, :-1]
            got = torch.empty(n, 127, 3, device=dev)
            src = pairs.data_ptr() + 4 if stride == 2 else ids_dev.data_ptr()
            lin_c = None if lin is None else (C.c_float * 9)(*[float(v) for v in lin.ravel()])
            _lib.check(lib.ttl_oracle_segments(hist.data_ptr(), hist.stride(0), src, stride, n, L,
                                               lin_c, 128, got.data_ptr(), stream), 'segments')
            if lin is None:
                assert torch.equal(got, want), (n, L)
            else:
                assert torch.allclose(got, want, rtol=0, atol=2e-5), (n, L)
    # ids = NULL: the first n rows
    got = torch.empty(9, 127, 3, device=dev)
    _lib.check(lib.ttl_oracle_segments(hist.data_ptr(), hist.stride(0), None, 1, 9, 50, None, 128,
                                       got.data_ptr(), stream), 'segments')
    data = resample_streamlines(hist[:9, :50], torch.full((9,), 50, dtype=torch.long, device=dev))
    assert torch.equal(got, data[:, 1:] - data[:, :-1])
    # refused, not launched
    assert lib.ttl_oracle_segments(hist.data_ptr(), 3 * 10, None, 1, 9, 50, None, 128,
                                   got.data_ptr(), stream) == _lib.ERR_INVALID


@pytest.mark.gpu
def test_oracle_bonus_kernel():
    """`ttl_oracle_bonus`: term = 0 everywhere, `bonus` at the listed rows whose
    score is > 0.5 -- rows from n_scored on were never scored (the reference's
    partial last batch, oracle.py:62-84) and get nothing --, reward += term
    (oracle_reward.py:84-93)."""
    import ctypes as C

    from tracktolearn_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(8)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for n, n_stop, n_scored in ((1, 1, 1), (1000, 300, 300), (5000, 4500, 4096), (70, 0, 0),
                                (513, 513, 0)):
        rows = torch.sort(torch.randperm(n, generator=g)[:n_stop]).values
        ids = torch.randint(0, 10 ** 6, (n_stop,), generator=g)
        pairs = torch.stack([rows, ids], 1).int().contiguous().to(dev)
        scores = torch.rand(max(n_stop, 1), generator=g)
        scores[::7] = 0.5                                   # exactly 0.5 earns nothing
        reward0 = torch.randn(n, generator=g, dtype=torch.float64)
        reward = reward0.clone().to(dev)
        term = torch.full((n,), 3.25, dtype=torch.float64, device=dev)
        sc = scores.to(dev)
        _lib.check(lib.ttl_oracle_bonus(sc.data_ptr(), n_scored, pairs.data_ptr(), n_stop, 10.0,
                                        n, term.data_ptr(), reward.data_ptr(), stream), 'bonus')
        want = torch.zeros(n, dtype=torch.float64)
        hit = (scores[:n_stop] > 0.5)
        hit[n_scored:] = False
        want[rows] = hit.double() * 10.0
        assert torch.equal(term.cpu(), want)
        assert torch.equal(reward.cpu(), reward0 + want)
    assert lib.ttl_oracle_bonus(sc.data_ptr(), 5, pairs.data_ptr(), 4, 10.0, 10, term.data_ptr(),
                                reward.data_ptr(), stream) == _lib.ERR_INVALID

