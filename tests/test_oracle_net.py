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
