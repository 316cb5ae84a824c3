"""The hand-scheduled SAC / SACAuto update (tracktolearn_amd/algorithms/shared/
fused.py) against the autograd formulation of the reference's update
(sac_auto.py:139-250, sac.py:135-232).

CPU part: the schedule (arenas, batching, manual backward, Adam/Polyak over the
arenas, the gradient autograd leaves on log_alpha) with the kernels replaced
by their plain-torch restatement (tests/ref_learner_ops.py), in float64: any
difference beyond rounding is a wrong formula.
GPU part (-m gpu): every HIP kernel against its restatement on the same
inputs, and the fused update on cuda:0 against the autograd update on cuda:0
and against float64."""
import copy

import pytest
import torch

from ref_learner_ops import TorchOps

CPU = torch.device('cpu')
DEV = 'cuda:0'


def _params(alg):
    out = []
    for net in (alg.agent.actor, alg.agent.critic, alg.target.actor, alg.target.critic):
        out += list(net.named_parameters())
    return out


def _pair(cls, hidden, W, B, dtype, device=CPU, ops=None, seed=1):
    torch.manual_seed(seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        a = cls(W, 3, hidden, n_actors=8, batch_size=B, replay_size=100, rng=None,
                device=torch.device(device))
        b = cls(W, 3, hidden, n_actors=8, batch_size=B, replay_size=100, rng=None,
                device=torch.device(device))
    finally:
        torch.set_default_dtype(old)
    b.agent.load_state_dict(a.agent.state_dict())
    b.target.load_state_dict(a.target.state_dict())
    if hasattr(a, 'log_alpha'):
        b.log_alpha.data.copy_(a.log_alpha.data)
    b._fused_ops = ops
    return a, b


def _batches(n, B, W, dtype, device=CPU, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        batch = [torch.randn(B, W, generator=g), torch.tanh(torch.randn(B, 3, generator=g)),
                 torch.randn(B, W, generator=g), torch.rand(B, generator=g),
                 (torch.rand(B, generator=g) > 0.2).float()]
        eps = [torch.randn(B, 3, generator=g) for _ in range(2)]
        out.append(([t.to(device, dtype) for t in batch], [e.to(device, dtype) for e in eps]))
    return out


def _inject(alg, eps):
    it = iter(eps)
    alg.noise_fn = lambda like: next(it)


@pytest.mark.parametrize('cls_name,hidden', [('SACAuto', '32-32'), ('SACAuto', '16'),
                                             ('SACAuto', '24-20-12'), ('SAC', '32-32')])
def test_schedule_equals_autograd_in_float64(cls_name, hidden):
    from tracktolearn_amd.algorithms.sac import SAC
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    cls = {'SAC': SAC, 'SACAuto': SACAuto}[cls_name]
    W, B = 27, 64
    ref, fused = _pair(cls, hidden, W, B, torch.float64, ops=TorchOps())
    for batch, eps in _batches(3, B, W, torch.float64):
        _inject(ref, eps)
        _inject(fused, eps)
        l_ref, l_fused = ref.update(batch), fused.update(batch)
        assert fused._fused is not None and ref._fused is None
        for (name, p), (_, q) in zip(_params(ref), _params(fused)):
            assert torch.allclose(p, q, rtol=0, atol=1e-12), name
            if p.grad is not None:
                assert torch.allclose(p.grad, q.grad, rtol=1e-9, atol=1e-13), name
        if cls is SACAuto:
            assert l_ref == l_fused == {}
            assert abs(float(ref.log_alpha.detach()) - float(fused.log_alpha.detach())) < 1e-13
            # what autograd leaves on log_alpha.grad: d alpha_loss + d actor_loss
            assert abs(float(ref.log_alpha.grad) - float(fused.log_alpha.grad)) < 1e-12
        else:
            assert set(l_ref) == set(l_fused)
            for k in l_ref:
                assert abs(float(l_ref[k]) - float(l_fused[k])) < 1e-12, k
    assert fused.total_it == ref.total_it == 3


def test_arena_views_keep_the_reference_surface(tmp_path):
    """Parameters, gradients and optimizer state are views of the arenas: the
    checkpoint keys, ``state_dict`` round trips, ``optimizer.state_dict`` and
    ``zero_grad()`` behave as with the plain modules."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    W, B = 27, 64
    ref, fused = _pair(SACAuto, '32-32', W, B, torch.float64, ops=TorchOps())
    (batch, eps), (batch2, eps2) = _batches(2, B, W, torch.float64)
    for alg in (ref, fused):
        _inject(alg, eps)
        alg.update(batch)
    fu = fused._fused
    w0 = fused.agent.actor.layers[0].weight
    assert w0.data_ptr() == fu.arena_a.online.data_ptr()
    assert w0.grad.data_ptr() == fu.arena_a.grad.data_ptr()
    assert list(fused.agent.actor.state_dict()) == list(ref.agent.actor.state_dict())
    assert list(fused.agent.critic.state_dict()) == list(ref.agent.critic.state_dict())
    sd_ref, sd_fused = ref.actor_optimizer.state_dict(), fused.actor_optimizer.state_dict()
    assert sd_ref['param_groups'][0]['params'] == sd_fused['param_groups'][0]['params']
    for k in sd_ref['state']:
        assert float(sd_fused['state'][k]['step']) == float(sd_ref['state'][k]['step']) == 1.0
        assert torch.allclose(sd_ref['state'][k]['exp_avg'], sd_fused['state'][k]['exp_avg'],
                              rtol=1e-9, atol=1e-14)
    # save / load through the reference's file names, zero_grad, a foreign
    # optimizer state: the next update still equals the autograd one
    fused.agent.save(str(tmp_path), 'm')
    fused.agent.load(str(tmp_path), 'm')
    fused.actor_optimizer.zero_grad()
    # (deep copy: load_state_dict aliases tensors that need no cast)
    fused.critic_optimizer.load_state_dict(copy.deepcopy(ref.critic_optimizer.state_dict()))
    assert w0.grad is None
    for alg in (ref, fused):
        _inject(alg, eps2)
        alg.update(batch2)
    assert w0.grad is not None
    for (name, p), (_, q) in zip(_params(ref), _params(fused)):
        assert torch.allclose(p, q, rtol=0, atol=1e-12), name
    # a module moved / re-typed by the caller is re-homed, not silently stale
    fused.agent.actor.double()
    fused.agent.actor.layers[0].weight.data = fused.agent.actor.layers[0].weight.data.clone()
    for alg in (ref, fused):
        _inject(alg, eps)
        alg.update(batch)
    for (name, p), (_, q) in zip(_params(ref), _params(fused)):
        assert torch.allclose(p, q, rtol=0, atol=1e-12), name


def test_fused_path_needs_the_library_on_a_gpu_and_is_off_on_the_cpu(monkeypatch):
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    alg = SACAuto(27, 3, '32-32', n_actors=8, batch_size=16, replay_size=100, rng=None,
                  device=CPU)
    assert not alg._use_fused()                  # CPU: the autograd formulation
    # a CUDA learner without the library fails loudly (no fallback)
    from tracktolearn_amd import _lib
    from tracktolearn_amd.algorithms.shared import fused
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libttl_hip.so')
    with pytest.raises(_lib.TTLError):
        fused.HipOps('cuda:0')


# --------------------------------------------------------------------------
# GPU: kernels against their restatements, update against autograd
# --------------------------------------------------------------------------
def _close(a, b, rtol, atol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), (what, float(err.max()), float((err / tol).max()))


def _no_worse(hip, ref32, ref64, what, factor=4.0, atol=1e-7):
    """|hip - float64| <= factor * |torch float32 restatement - float64| + atol,
    in the max norm: the kernel is as accurate as the same formula through
    torch's own float32 kernels (whose GEMM sums in another order)."""
    hip, ref32, ref64 = (t.detach().double().cpu() for t in (hip, ref32, ref64))
    e_hip, e_32 = float((hip - ref64).abs().max()), float((ref32 - ref64).abs().max())
    assert e_hip <= factor * e_32 + atol * max(1.0, float(ref64.abs().max())), \
        (what, e_hip, e_32)


@pytest.mark.gpu
@pytest.mark.parametrize('M,H,A', [(8192, 1024, 3), (4096, 1024, 3), (100, 68, 2), (37, 45, 3),
                                   (16, 2050, 4), (5, 7, 1)])
def test_thin_forward_kernels(M, H, A):
    from tracktolearn_amd.algorithms.shared.fused import (HEAD_PLAIN, HEAD_SAC, HEAD_TANH,
                                                          THIN_FWD_ROWS, HipOps)
    hip, ref = HipOps(DEV), TorchOps()
    g = torch.Generator().manual_seed(M + H)
    # strided rows (ld > H) as the side-by-side critic activations have
    a = torch.randn(M, H + 8, generator=g).to(DEV)[:, :H]
    w = (torch.randn(2 * A, H, generator=g) / H ** 0.5).to(DEV)
    w[A:] *= 4                                             # some log_std beyond the clamp
    b = torch.randn(2 * A, generator=g).to(DEV)
    b[A:] += torch.tensor([-1.0, 3.0, -25.0, 0.0][:A], device=DEV)
    eps = torch.randn(M, A, generator=g).to(DEV)
    ent_rows = M // 2 + 3
    outs = []
    for ops, dt in ((hip, torch.float32), (ref, torch.float32), (ref, torch.float64)):
        z = dict(device=DEV, dtype=dt)
        out = torch.zeros(M, A + 5, **z)
        logp, raw = torch.zeros(M, **z), torch.zeros(M, A, **z)
        part = torch.zeros(-(-M // THIN_FWD_ROWS), 1, **z)
        ops.thin_forward(a.to(dt), w.to(dt), b.to(dt), 2 * A, False, HEAD_SAC, out[:, 2:], A + 5,
                         eps=eps.to(dt), entropy_rows=ent_rows, logp=logp, ls_raw=raw,
                         ent_part=part)
        outs.append((out, logp, raw, part))
    for k, what in enumerate(('pi', 'logp', 'log_std_raw', 'entropy partials')):
        _no_worse(outs[0][k], outs[1][k], outs[2][k], what)
    o1 = outs[0][0]
    assert float(o1[:, :2].abs().max()) == 0 and float(o1[:, 2 + A:].abs().max()) == 0
    # plain / tanh heads, dense and block-diagonal
    for n_out, head in ((2 * A, HEAD_PLAIN), (A, HEAD_TANH), (1, HEAD_PLAIN)):
        res = []
        for ops, dt in ((hip, torch.float32), (ref, torch.float32), (ref, torch.float64)):
            out = torch.zeros(M, n_out + 1, device=DEV, dtype=dt)
            ops.thin_forward(a.to(dt), w[:n_out].contiguous().to(dt),
                             b[:n_out].contiguous().to(dt), n_out, False, head, out, n_out + 1)
            res.append(out)
        _no_worse(*res, (n_out, head))
    a2 = torch.randn(M, 2 * H, generator=g).to(DEV)
    res = []
    for ops, dt in ((hip, torch.float32), (ref, torch.float32), (ref, torch.float64)):
        out = torch.zeros(M, 2, device=DEV, dtype=dt)
        ops.thin_forward(a2.to(dt), w[:2].contiguous().to(dt), b[:2].contiguous().to(dt), 2, True,
                         HEAD_PLAIN, out, 2)
        res.append(out)
    _no_worse(*res, 'block diagonal')
    # ... and with the two networks in planes of their own [2 x M x H]
    planes = torch.stack([a2[:, :H], a2[:, H:]]).contiguous()
    out_p = torch.zeros(M, 2, device=DEV)
    hip.thin_forward(planes, w[:2].contiguous(), b[:2].contiguous(), 2, True, HEAD_PLAIN, out_p, 2)
    assert torch.equal(out_p, res[0])


def _three(fn):
    """fn(ops, dtype) -> tuple of tensors, for the HIP kernels in float32, the
    restatement in float32 and the restatement in float64."""
    from tracktolearn_amd.algorithms.shared.fused import HipOps
    return [fn(ops, dt) for ops, dt in ((HipOps(DEV), torch.float32), (TorchOps(), torch.float32),
                                        (TorchOps(), torch.float64))]


def _check_three(res, names):
    for k, what in enumerate(names):
        _no_worse(res[0][k], res[1][k], res[2][k], what)


@pytest.mark.gpu
@pytest.mark.parametrize('M,H', [(8192, 1024), (4096, 1024), (1000, 68), (37, 45), (9, 2050)])
def test_backward_kernels(M, H):
    from tracktolearn_amd.algorithms.shared.fused import _rows_per_block
    g = torch.Generator().manual_seed(M * 7 + H)
    R = -(-M // _rows_per_block(M))
    r0, r1 = M // 3, M - 2
    # dense thin layer (actor head: 6 outputs)
    a = torch.relu(torch.randn(M, H, generator=g)).to(DEV)
    d_out = torch.randn(M, 6, generator=g).to(DEV)
    w = torch.randn(6, H, generator=g).to(DEV)

    def dense(ops, dt):
        dz = torch.zeros(M, H, device=DEV, dtype=dt)
        part = torch.full((R, H + 6 * H + 6), 7.0, device=DEV, dtype=dt)
        ops.thin_backward(d_out.to(dt), a.to(dt), w.to(dt), 6, False, r0, r1, dz, part)
        return dz, part, part.sum(0)
    _check_three(_three(dense), ('thin dense dz', 'thin dense slab', 'thin dense sums'))
    # block diagonal (the two critics side by side, ld > 2H)
    a2 = torch.relu(torch.randn(M, 2 * H + 4, generator=g)).to(DEV)
    dq = torch.randn(M, 2, generator=g).to(DEV)
    w2 = torch.randn(2, H, generator=g).to(DEV)

    def blockdiag(ops, dt):
        dz = torch.zeros(M, 2 * H, device=DEV, dtype=dt)
        part = torch.full((R, 4 * H + 2), 7.0, device=DEV, dtype=dt)
        ops.thin_backward(dq.to(dt), a2.to(dt)[:, :2 * H], w2.to(dt), 2, True, r0, r1, dz, part)
        return dz, part
    res_bd = _three(blockdiag)
    _check_three(res_bd, ('thin bd dz', 'thin bd slab'))
    # the same with activations and gradients in planes [2 x M x H]: same bits
    from tracktolearn_amd.algorithms.shared.fused import HipOps
    a_pl = torch.stack([a2[:, :H], a2[:, H:2 * H]]).contiguous()
    dz_pl = torch.zeros(2, M, H, device=DEV)
    part_pl = torch.full((R, 4 * H + 2), 7.0, device=DEV)
    HipOps(DEV).thin_backward(dq, a_pl, w2, 2, True, r0, r1, dz_pl, part_pl)
    assert torch.equal(torch.cat([dz_pl[0], dz_pl[1]], dim=1), res_bd[0][0])
    assert torch.equal(part_pl, res_bd[0][1])
    # ReLU backward + bias partials
    d0 = torch.randn(M, 2 * H, generator=g).to(DEV)

    def relu(ops, dt):
        dz = d0.to(dt).clone()
        part = torch.full((R, 2 * H), 7.0, device=DEV, dtype=dt)
        ops.relu_backward_bias(dz, a2.to(dt)[:, :2 * H], r0, r1, part)
        return dz, part
    res = _three(relu)
    assert torch.equal(res[0][0], res[1][0])
    _check_three(res, ('relu dz', 'relu slab'))
    dz_pl = torch.stack([d0[:, :H], d0[:, H:]]).contiguous()
    part_pl = torch.full((R, 2 * H), 7.0, device=DEV)
    HipOps(DEV).relu_backward_bias(dz_pl, a_pl, r0, r1, part_pl)
    assert torch.equal(torch.cat([dz_pl[0], dz_pl[1]], dim=1), res[0][0])
    assert torch.equal(part_pl, res[0][1])
    # finalize: wide, narrow and scaled segments in one launch
    part = res[0][1]
    narrow = torch.randn(300, 8, generator=g).to(DEV)

    def finalize(ops, dt):
        o1, o2, o3 = (torch.zeros(k, device=DEV, dtype=dt) for k in (2 * H, 8, 1))
        ops.colsum_finalize([(part.to(dt), 0, 2 * H, o1, 1.0), (narrow.to(dt), 0, 8, o2, 0.25),
                             (narrow.to(dt)[:77], 3, 1, o3, 1.0 / 77)])
        return o1, o2, o3
    _check_three(_three(finalize), ('finalize wide', 'finalize narrow', 'finalize scalar'))
    # actor-loss gradient through the critics' first layer + head backward
    dh = torch.randn(M, 2 * H, generator=g).to(DEV)
    wa = torch.randn(3, 2 * H, generator=g).to(DEV)
    pi = torch.tanh(torch.randn(M, 9, generator=g)).to(DEV)
    eps = torch.randn(M, 3, generator=g).to(DEV)
    raw = (torch.randn(M, 3, generator=g) * 8).to(DEV)
    la = torch.tensor([-1.3], device=DEV)
    for log_alpha, const in ((la, 0.0), (None, 0.2)):
        def head(ops, dt):
            d_head = torch.zeros(M, 6, device=DEV, dtype=dt)
            ops.actor_head_backward(dh.to(dt), a2.to(dt)[:, :2 * H], wa.to(dt), 3,
                                    pi.to(dt)[:, 4:], 9, eps.to(dt), raw.to(dt),
                                    None if log_alpha is None else log_alpha.to(dt), const, d_head)
            return (d_head,)
        _check_three(_three(head), ('head backward',))


@pytest.mark.gpu
def test_losses_adam_and_input_kernels():
    from tracktolearn_amd.algorithms.shared.fused import LOSS_BLOCK, HipOps
    hip, ref = HipOps(DEV), TorchOps()
    g = torch.Generator().manual_seed(5)
    z = dict(device=DEV)
    for n in (4096, 1000, 7):
        q_on = torch.randn(2 * n, 2, generator=g).to(DEV)
        q_on[n + 1, 1] = q_on[n + 1, 0]                      # a tie of the two critics
        q_tg = torch.randn(n, 2, generator=g).to(DEV)
        logp = torch.randn(2 * n, generator=g).to(DEV)
        r, nd = torch.rand(n, generator=g).to(DEV), (torch.rand(n, generator=g) > 0.2).float().to(DEV)
        la = torch.tensor([-1.6], device=DEV)
        for log_alpha, const, mask in ((la, 0.0, 0b111), (None, 0.2, 0b110)):
            res = []
            for ops in (hip, ref):
                dq = torch.zeros(2 * n, 2, **z)
                part = torch.zeros(-(-n // LOSS_BLOCK), 8, **z)
                steps = torch.tensor([4.0, 9.0, 0.0], **z)
                pows = torch.tensor([0.9 ** 4, 0.999 ** 4, 0.9 ** 9, 0.999 ** 9, 1.0, 1.0],
                                    dtype=torch.float64, device=DEV)
                consts = torch.zeros(6, **z)
                ops.sac_losses(q_on, q_tg, logp, r, nd, log_alpha, const, 0.99, dq, part, steps,
                               consts, pows, mask, 3e-4)
                res.append((dq, part.sum(0), steps, consts, pows))
            assert torch.equal(res[0][2], res[1][2])
            _close(res[0][0], res[1][0], 1e-6, 1e-9, 'dq')
            _close(res[0][1], res[1][1], 1e-5, 1e-3, 'loss sums')
            _close(res[0][3], res[1][3], 1e-6, 0, 'adam scalars')
            _close(res[0][4], res[1][4], 1e-14, 0, 'beta powers')
    # Adam + Polyak over an arena whose length is not a multiple of 4
    n = 1000003
    p0, g0 = torch.randn(n + 1, generator=g).to(DEV)[:n], torch.randn(n + 1, generator=g).to(DEV)[:n]
    m0, v0 = 0.1 * torch.randn(n + 1, generator=g).to(DEV)[:n], torch.rand(n + 1, generator=g).to(DEV)[:n]
    t0 = torch.randn(n + 1, generator=g).to(DEV)[:n]
    consts = torch.tensor([3e-4 / (1 - 0.9 ** 3), (1 - 0.999 ** 3) ** 0.5], **z)
    res = []
    for ops in (hip, ref):
        p, m, v, t = p0.clone(), m0.clone(), v0.clone(), t0.clone()
        ops.adam_polyak(p, g0, m, v, t, consts, 0.005)
        res.append((p, m, v, t))
    for x, y, what in zip(res[0], res[1], 'pmvt'):
        _close(x, y, 1e-6, 1e-7, 'adam ' + what)
    # torch.optim.Adam itself, three steps from scratch
    w = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([w], lr=3e-4)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    steps, consts = torch.zeros(1, **z), torch.zeros(2, **z)
    pows = torch.ones(2, dtype=torch.float64, device=DEV)
    dummy = torch.zeros(2, 2, **z)
    for it in range(3):
        gi = torch.randn(n, generator=g).to(DEV)
        w.grad = gi.clone()
        opt.step()
        hip.sac_losses(dummy, dummy[:1], dummy.view(-1)[:2], dummy[0, :1], dummy[0, :1], None, 0.2,
                       0.99, torch.zeros(2, 2, **z), None, steps, consts, pows, 0b1, 3e-4)
        hip.adam_polyak(p, gi, m, v, None, consts, 0.0)
    assert float(steps) == 3.0
    _close(p, w.data, 2e-7, 1e-7, 'three Adam steps vs torch.optim.Adam')     # <= 1 ulp
    # temperature step
    res = []
    for ops in (hip, ref):
        la, gr = torch.tensor([-1.6], **z), torch.zeros(1, **z)
        m, v = torch.tensor([0.3], **z), torch.tensor([2.0], **z)
        ops.alpha_step(la, gr, m, v, torch.tensor([-2.2], **z), -3.0,
                       torch.tensor([3e-4 / (1 - 0.9 ** 2), (1 - 0.999 ** 2) ** 0.5], **z))
        res.append(torch.cat([la, gr, m, v]))
    _close(res[0], res[1], 1e-6, 1e-8, 'alpha step')
    # network input rows
    n, S, A = 1000, 327, 3
    s, a, s2 = (torch.randn(n, k, generator=g).to(DEV) for k in (S, A, S))
    w1 = torch.randn(2048, S + A, generator=g).to(DEV)
    res = []
    for ops in (hip, ref):
        xs, wa = torch.full((3 * n, 332), 9.0, **z), torch.zeros(A, 2048, **z)
        ops.build_inputs(s, a, s2, xs, S, A, w1, wa)
        res.append((xs, wa))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def _grad_err(g, g64):
    g, g64 = g.detach().double().cpu(), g64.detach().double().cpu()
    return float((g - g64).norm() / (g64.norm() + 1e-300))


def _sync(dst, src, dtype, device):
    """dst <- src: weights, targets, temperature and optimizer state."""
    for name in ('agent', 'target'):
        sd_a, sd_c = getattr(src, name).state_dict()
        getattr(dst, name).load_state_dict(({k: v.to(device, dtype) for k, v in sd_a.items()},
                                            {k: v.to(device, dtype) for k, v in sd_c.items()}))
    if hasattr(src, 'log_alpha'):
        dst.log_alpha.data.copy_(src.log_alpha.data.to(device, dtype))
    for od, os_ in zip(dst._optimizers(), src._optimizers()):
        od.load_state_dict(copy.deepcopy(os_.state_dict()))
    dst.total_it = src.total_it


@pytest.mark.gpu
@pytest.mark.parametrize('cls_name,hidden,W,B', [('SACAuto', '1024-1024', 327, 4096),
                                                 ('SACAuto', '256-192-128', 615, 1000),
                                                 ('SAC', '128', 45, 333)])
def test_fused_update_on_the_gpu_matches_autograd(cls_name, hidden, W, B):
    """cuda:0: FusedSACUpdate (HIP kernels + GEMMs) against the autograd
    update on the same device and against a float64 CPU run, two updates (the
    second with Adam moments in place; before it the two referees are synced to
    the fused learner's state, optimizers included): gradients within 5e-3 in
    L2 of float64 (ReLU / min decisions within rounding of their boundary route
    a row differently; typical tensors 1e-6), parameters >= 99.8 % of the
    entries within 1e-5 of the autograd result, none beyond 2 lr."""
    from tracktolearn_amd.algorithms.sac import SAC
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    cls = {'SAC': SAC, 'SACAuto': SACAuto}[cls_name]
    ref64, _ = _pair(cls, hidden, W, B, torch.float64)
    plain, fused = _pair(cls, hidden, W, B, torch.float32, device=DEV)
    plain.use_fused_learner = False
    for alg in (plain, fused):
        _sync(alg, ref64, torch.float32, DEV)
    lr = 3e-4
    for u, (batch, eps) in enumerate(_batches(2, B, W, torch.float64)):
        if u:
            _sync(plain, fused, torch.float32, DEV)
            _sync(ref64, fused, torch.float64, 'cpu')
        _inject(ref64, eps)
        ref64.update(batch)
        b32 = [t.float().to(DEV) for t in batch]
        for alg in (plain, fused):
            _inject(alg, [e.float().to(DEV) for e in eps])
            alg.update(b32)
        assert fused._fused is not None and plain._fused is None
        worst = 0.0
        for (name, p64), (_, pp), (_, pf) in zip(_params(ref64), _params(plain), _params(fused)):
            if p64.grad is not None:
                e_f, e_p = _grad_err(pf.grad, p64.grad), _grad_err(pp.grad, p64.grad)
                worst = max(worst, e_f)
                assert e_f <= 5e-3, (u, name, e_f, e_p)
            d = (pf.detach() - pp.detach()).abs()
            assert float(d.max()) <= 2 * lr, (u, name, float(d.max()))
            assert float((d <= 1e-5).float().mean()) >= 0.998, (u, name)
        if cls is SACAuto:
            assert abs(float(fused.log_alpha.detach()) - float(plain.log_alpha.detach())) <= 1e-6
            assert _grad_err(fused.log_alpha.grad, ref64.log_alpha.grad) <= 1e-5
        print(f'{cls_name} {hidden} update {u}: worst gradient L2 error vs float64 {worst:.2e}')
    assert fused.total_it == plain.total_it == 2


@pytest.mark.gpu
def test_fused_sac_losses_dict_matches_autograd():
    from tracktolearn_amd.algorithms.sac import SAC
    W, B = 64, 512
    plain, fused = _pair(SAC, '64-64', W, B, torch.float32, device=DEV)
    plain.use_fused_learner = False
    (batch, eps), = _batches(1, B, W, torch.float32, device=DEV)
    _inject(plain, eps)
    _inject(fused, eps)
    lp, lf = plain.update(batch), fused.update(batch)
    assert set(lp) == set(lf)
    for k in lp:
        assert abs(float(lp[k]) - float(lf[k])) <= 1e-5 * max(1.0, abs(float(lp[k]))), k


@pytest.mark.gpu
def test_replay_add_kernel_fills_the_ring_like_the_torch_path(monkeypatch):
    """`add_partitioned` with what `env.step_device()` hands out (float32 rows,
    int32 row_dest, float64 reward, uint8 done) is one launch of
    `ttl_replay_add`; the ring it leaves equals the torch scatter path's, also
    across the wrap and with float32 rewards / bool dones."""
    from tracktolearn_amd.algorithms.shared import replay as rp
    W, A, cap = 41, 3, 1000
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(4)
    calls = []
    real = rp.OffPolicyReplayBuffer._add_device

    def spy(self, *a):
        done = real(self, *a)
        calls.append(done)
        return done
    monkeypatch.setattr(rp.OffPolicyReplayBuffer, '_add_device', spy)
    fast = rp.OffPolicyReplayBuffer(W, A, max_size=cap, device=dev)
    slow = rp.OffPolicyReplayBuffer(W, A, max_size=cap, device=dev)
    for i, n in enumerate((300, 1, 450, 400, 1000, 77)):          # 4th add wraps
        s, a = torch.randn(n, W, generator=g).to(dev), torch.randn(n, A, generator=g).to(dev)
        ns = torch.randn(n, W, generator=g).to(dev)
        dest = torch.randperm(n, generator=g).to(dev)
        r = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
        d = (torch.rand(n, generator=g) > 0.5).to(dev)
        ns_part = torch.empty_like(ns)
        ns_part[dest] = ns
        if i % 2:
            r_in, d_in = r.float(), d
        else:
            r_in, d_in = r, d.to(torch.uint8)
        fast.add_partitioned(s, a, ns_part, dest.int(), r_in, d_in)
        slow.add(s, a, ns, r_in, d_in)
        assert fast.ptr == slow.ptr and fast.size == slow.size
        for name in ('state', 'action', 'next_state', 'reward', 'not_done'):
            assert torch.equal(getattr(fast, name), getattr(slow, name)), (i, name)
    assert calls == [True] * 6
    # int64 row_dest (not what the env returns) keeps the torch path, same ring
    fast.add_partitioned(s, a, ns_part, dest, r, d)
    slow.add(s, a, ns, r, d)
    assert calls[-1] is False and torch.equal(fast.next_state, slow.next_state)
    # more rows than the ring holds cannot be one launch
    big = rp.OffPolicyReplayBuffer(W, A, max_size=50, device=dev)
    assert not real(big, s, a, ns_part, dest.int(), r, d.to(torch.uint8))


@pytest.mark.gpu
def test_replay_sample_kernel_draws_distinct_uniform_rows():
    """`OffPolicyReplayBuffer.sample` on the GPU (`ttl_replay_sample`): what
    `randperm(size)[:batch]` + five `index_select`s give -- distinct ring rows,
    uniformly drawn, the five tensors of the same rows -- from a keyed
    permutation evaluated for the first `batch` positions only."""
    from tracktolearn_amd.algorithms.shared.replay import OffPolicyReplayBuffer
    W, A = 37, 3
    g = torch.Generator().manual_seed(9)
    for size, batch in ((1000, 1000), (1000, 64), (5, 3), (1, 1), (65537, 4096), (300000, 4096)):
        buf = OffPolicyReplayBuffer(W, A, max_size=size + 7, device=torch.device(DEV))
        n = size
        buf.add(torch.randn(n, W, generator=g), torch.randn(n, A, generator=g),
                torch.randn(n, W, generator=g), torch.rand(n, generator=g),
                (torch.rand(n, generator=g) > 0.5).float())
        assert len(buf) == size
        torch.manual_seed(123)
        s, a, ns, r, d = buf.sample(batch)
        ind = buf.last_indices
        assert s.shape == (batch, W) and a.shape == (batch, A) and r.shape == d.shape == (batch,)
        assert int(ind.min()) >= 0 and int(ind.max()) < size
        assert len(torch.unique(ind)) == batch                          # no replacement
        assert torch.equal(s, buf.state[ind]) and torch.equal(ns, buf.next_state[ind])
        assert torch.equal(a, buf.action[ind])
        assert torch.equal(r, buf.reward[ind, 0]) and torch.equal(d, buf.not_done[ind, 0])
        # torch's seed makes the draw repeatable; the next call draws afresh
        torch.manual_seed(123)
        assert torch.equal(buf.sample(batch)[0], s)
        if batch < size:
            buf.sample(batch)
            assert not torch.equal(buf.last_indices, ind)
        # asking for more than the ring holds returns every row once
        assert buf.sample(size + 100)[0].shape[0] == size
        assert torch.equal(torch.sort(buf.last_indices).values,
                           torch.arange(size, device=DEV))
    # uniform marginals: 3 000 draws of 64 from 1 000 rows -> 192 hits per row on
    # average; a chi-square far outside its range would mean a biased permutation
    buf = OffPolicyReplayBuffer(4, A, max_size=1000, device=torch.device(DEV))
    buf.add(torch.zeros(1000, 4), torch.zeros(1000, A), torch.zeros(1000, 4), torch.zeros(1000),
            torch.zeros(1000))
    counts = torch.zeros(1000, dtype=torch.int64, device=DEV)
    first = torch.zeros(1000, dtype=torch.int64, device=DEV)
    torch.manual_seed(5)
    for _ in range(3000):
        buf.sample(64)
        counts += torch.bincount(buf.last_indices, minlength=1000)
        first[buf.last_indices[0]] += 1
    expected = 3000 * 64 / 1000
    chi2 = float(((counts.double() - expected) ** 2 / expected).sum())
    assert 850 < chi2 < 1150, chi2                      # 999 degrees of freedom: 999 +- 45
    chi2_first = float(((first.double() - 3.0) ** 2 / 3.0).sum())
    assert 850 < chi2_first < 1150, chi2_first          # position 0 alone is uniform too


@pytest.mark.gpu
def test_policy_sampling_with_the_fused_head_equals_the_module_path(monkeypatch):
    """`SACActorCritic.select_action` on the GPU runs the actor's head as one
    `ttl_thin_forward` launch: same actions as the Linear + clamp + exp + randn +
    tanh chain (same draw: one `randn` of the same shape) to float32 rounding,
    for probabilistic 0, 1 and in between, and a row's action does not depend
    on the rows it shares a batch with."""
    from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic
    torch.manual_seed(3)
    ac = SACActorCritic(327, 3, '256-128', torch.device(DEV))
    x = torch.randn(1000, 327, device=DEV)
    for prob in (0.0, 1.0, 0.4):
        monkeypatch.setenv('TTL_FUSED_POLICY_HEAD', '1')
        torch.manual_seed(9)
        with torch.no_grad():
            a_fused = ac.select_action(x, prob)
        monkeypatch.setenv('TTL_FUSED_POLICY_HEAD', '0')
        torch.manual_seed(9)
        with torch.no_grad():
            a_plain = ac.select_action(x, prob)
        assert a_fused.shape == a_plain.shape == (1000, 3)
        assert float((a_fused - a_plain).abs().max()) <= 2e-6, prob
    monkeypatch.setenv('TTL_FUSED_POLICY_HEAD', '1')
    with torch.no_grad():
        whole = ac.select_action(x, 0.0)
        part = ac.select_action(x[17:29].contiguous(), 0.0)
    # hidden-layer GEMMs may pick other kernels for other batch sizes; the head is row-local
    assert float((whole[17:29] - part).abs().max()) <= 2e-6
    assert ac.select_action(x[:0], 1.0).shape == (0, 3)


# --------------------------------------------------------------------------
# TD3 / DDPG
# --------------------------------------------------------------------------
def _det_pair(cls_name, hidden, W, B, dtype, device=CPU, ops=None, seed=2):
    from tracktolearn_amd.algorithms.ddpg import DDPG
    from tracktolearn_amd.algorithms.td3 import TD3
    cls = {'TD3': TD3, 'DDPG': DDPG}[cls_name]
    torch.manual_seed(seed)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        a = cls(W, 3, hidden, action_std=0.3, n_actors=8, batch_size=B, replay_size=100,
                rng=None, device=torch.device(device))
        b = cls(W, 3, hidden, action_std=0.3, n_actors=8, batch_size=B, replay_size=100,
                rng=None, device=torch.device(device))
    finally:
        torch.set_default_dtype(old)
    b.agent.load_state_dict(a.agent.state_dict())
    b.target.load_state_dict(a.target.state_dict())
    b._fused_ops = ops
    return a, b


@pytest.mark.parametrize('cls_name,hidden', [('TD3', '32-32'), ('TD3', '16'), ('TD3', '24-20-12'),
                                             ('DDPG', '32-32'), ('DDPG', '12-20-16')])
def test_td3_ddpg_schedule_equals_autograd_in_float64(cls_name, hidden, monkeypatch):
    """FusedTD3Update's schedule (critic step first, then -- every agent_freq-th
    update -- the actor through the UPDATED first critic, then the Polyak
    averages) with the kernels' torch restatement, against the autograd
    `update` of td3.py:130-230 / ddpg.py:234-319 in float64: four updates (two
    with an actor step for TD3), every parameter, target and loss."""
    W, B = 27, 64
    ref, fused = _det_pair(cls_name, hidden, W, B, torch.float64, ops=TorchOps())
    g = torch.Generator().manual_seed(4)
    for u, (batch, _) in enumerate(_batches(4, B, W, torch.float64)):
        noise = torch.randn(B, 3, generator=g, dtype=torch.float64)
        monkeypatch.setattr(torch, 'randn_like', lambda t, **kw: noise)
        l_ref = ref.update(batch)
        l_fused = fused.update(batch)
        assert fused._fused is not None and ref._fused is None
        for (name, p), (_, q) in zip(_params(ref), _params(fused)):
            assert torch.allclose(p, q, rtol=0, atol=1e-12), (u, name)
        assert set(l_ref) == set(l_fused), (l_ref.keys(), l_fused.keys())
        for k in l_ref:
            assert abs(float(l_ref[k]) - float(l_fused[k])) < 1e-12, (u, k)
    assert fused.total_it == ref.total_it == 4


@pytest.mark.gpu
@pytest.mark.parametrize('cls_name,hidden,W,B', [('TD3', '1024-1024', 327, 4096),
                                                 ('TD3', '96-64-48', 45, 500),
                                                 ('DDPG', '256-256', 615, 1000)])
def test_fused_td3_ddpg_update_on_the_gpu_matches_autograd(cls_name, hidden, W, B, monkeypatch):
    """cuda:0: the fused TD3 / DDPG update (HIP kernels + GEMMs) against the
    autograd update on the same device, two updates from identical states:
    >= 99.8 % of the parameters within 1e-5, none beyond 2 lr; losses within
    1e-5 relative."""
    plain, fused = _det_pair(cls_name, hidden, W, B, torch.float32, device=DEV)
    plain.use_fused_learner = False
    g = torch.Generator().manual_seed(6)
    lr = 3e-4
    for u, (batch, _) in enumerate(_batches(2, B, W, torch.float32, device=DEV)):
        if u:
            _sync_det(plain, fused)
        noise = torch.randn(B, 3, generator=g).to(DEV)
        monkeypatch.setattr(torch, 'randn_like', lambda t, **kw: noise)
        l_plain = plain.update(batch)
        l_fused = fused.update(batch)
        assert fused._fused is not None and plain._fused is None
        for (name, pp), (_, pf) in zip(_params(plain), _params(fused)):
            d = (pf.detach() - pp.detach()).abs()
            assert float(d.max()) <= 2 * lr, (u, name, float(d.max()))
            assert float((d <= 1e-5).float().mean()) >= 0.998, (u, name)
        for k in l_plain:
            a, b = float(l_plain[k]), float(l_fused[k])
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (u, k, a, b)
    assert fused.total_it == plain.total_it == 2


def _sync_det(dst, src):
    for name in ('agent', 'target'):
        getattr(dst, name).load_state_dict(getattr(src, name).state_dict())
    for od, os_ in ((dst.actor_optimizer, src.actor_optimizer),
                    (dst.critic_optimizer, src.critic_optimizer)):
        od.load_state_dict(copy.deepcopy(os_.state_dict()))
    dst.total_it = src.total_it
