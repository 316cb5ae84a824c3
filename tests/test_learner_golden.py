"""Known-answer tests of the learner against vectors
captured from the reference's SACAuto / TD3 / OffPolicyReplayBuffer
(tests/golden/make_golden_learner.py).  CPU: the learner is plain PyTorch."""
import numpy as np
import pytest
import torch

from helpers import load_trace

CPU = torch.device('cpu')


def _load_sd(z, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(z[k])
            for k in z.files if k.startswith(prefix + '/')}


def _check_sd(module, z, prefix, tol):
    want = _load_sd(z, prefix)
    got = module.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.allclose(got[k], want[k], rtol=tol, atol=tol), (prefix, k)


def _batch(z):
    return [torch.from_numpy(z[f'batch/{n}']) for n in
            ('state', 'action', 'next_state', 'reward', 'not_done')]


def test_sac_auto_update_matches_reference():
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    z = load_trace('learner_sac_auto')
    alg = SACAuto(27, 3, '32-32', lr=3e-4, gamma=0.99, alpha=0.2, n_actors=8,
                  batch_size=64, replay_size=1000, rng=None, device=CPU)
    # same parameter names as the reference checkpoints
    alg.agent.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    alg.target.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    assert alg.start_timesteps == 80000 and alg.tau == 0.005
    eps = iter(torch.from_numpy(z['eps']).reshape(-1, 64, 3))
    alg.noise_fn = lambda like: next(eps)
    batch = _batch(z)
    for u in range(int(z['n_updates'])):
        assert alg.update(batch) == {}
        _check_sd(alg.agent.actor, z, f'u{u}/actor', 2e-6)
        _check_sd(alg.agent.critic, z, f'u{u}/critic', 2e-6)
        _check_sd(alg.target.actor, z, f'u{u}/target_actor', 2e-6)
        _check_sd(alg.target.critic, z, f'u{u}/target_critic', 2e-6)
        assert np.allclose(alg.log_alpha.detach().numpy(), z[f'u{u}/log_alpha'],
                           rtol=1e-6, atol=1e-7)
    assert alg.total_it == 3
    # policy outputs
    st = batch[0]
    with torch.no_grad():
        det = alg.agent.select_action(st, probabilistic=0.0)
        a, logp = alg.agent.act(st, probabilistic=1.0,
                                eps=torch.from_numpy(z['eps']).reshape(-1, 64, 3)[0])
    assert np.allclose(det.numpy(), z['act_det'], atol=2e-6)
    assert np.allclose(a.numpy(), z['act_sto'], atol=2e-6)
    assert np.allclose(logp.numpy(), z['act_sto_logp'], atol=2e-5)


def test_td3_update_matches_reference(monkeypatch):
    from tracktolearn_amd.algorithms.td3 import TD3
    z = load_trace('learner_td3')
    alg = TD3(27, 3, '32-32', action_std=float(z['action_std']), lr=3e-4,
              gamma=0.99, n_actors=8, batch_size=64, replay_size=1000, rng=None,
              device=CPU)
    alg.agent.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    alg.target.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    eps = iter(torch.from_numpy(z['eps']))
    monkeypatch.setattr(torch, 'randn_like', lambda t, **kw: next(eps))
    batch = _batch(z)
    for u in range(int(z['n_updates'])):
        losses = alg.update(batch)
        assert set(losses) >= {'actor_loss', 'critic_loss', 'Q1', 'Q2'}
        _check_sd(alg.agent.actor, z, f'u{u}/actor', 2e-6)
        _check_sd(alg.agent.critic, z, f'u{u}/critic', 2e-6)
        _check_sd(alg.target.actor, z, f'u{u}/target_actor', 2e-6)
        _check_sd(alg.target.critic, z, f'u{u}/target_critic', 2e-6)


def test_replay_ring_matches_reference():
    """Integer-exact ring arithmetic incl. wrap-around and a batch larger than
    the free tail; contents identical after every add."""
    from tracktolearn_amd.algorithms.shared.replay import OffPolicyReplayBuffer
    z = load_trace('learner_replay')
    buf = OffPolicyReplayBuffer(5, 3, max_size=10, device=CPU)
    twin = OffPolicyReplayBuffer(5, 3, max_size=10, device=CPU)
    rng = np.random.RandomState(0)
    for i in range(int(z['n_adds'])):
        args = [torch.from_numpy(z[f'add{i}/{k}']) for k in ('s', 'a', 'ns', 'r', 'd')]
        buf.add(*args)
        # the partition-order entry point must fill the ring identically
        n = len(args[0])
        dest = torch.from_numpy(rng.permutation(n))
        ns_part = torch.empty_like(args[2])
        ns_part[dest] = args[2]
        twin.add_partitioned(args[0], args[1], ns_part, dest, args[3], args[4])
        for b in (buf, twin):
            assert b.ptr == int(z[f'after{i}/ptr']) and b.size == int(z[f'after{i}/size'])
            assert len(b) == b.size
            for name in ('state', 'action', 'next_state', 'reward', 'not_done'):
                assert np.array_equal(getattr(b, name).numpy(), z[f'after{i}/{name}'])
    s, a, ns, r, d = buf.sample(4)
    assert s.shape == (4, 5) and a.shape == (4, 3) and r.shape == (4,) and d.shape == (4,)
    s, *_ = buf.sample(4096)
    assert s.shape[0] == 10                      # min(size, batch), no replacement
    assert len({tuple(row) for row in s.numpy().round(6).tolist()}) == 10


def test_checkpoint_round_trip(tmp_path):
    from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic
    a = SACActorCritic(27, 3, '32-32', CPU)
    keys_a, keys_c = a.state_dict()
    assert list(keys_a) == ['layers.0.weight', 'layers.0.bias', 'layers.2.weight',
                            'layers.2.bias', 'layers.4.weight', 'layers.4.bias']
    assert [k for k in keys_c if k.startswith('q2')][0] == 'q2.0.weight'
    a.save(str(tmp_path), 'last_model_state')
    b = SACActorCritic(27, 3, '32-32', CPU)
    b.load(str(tmp_path), 'last_model_state')
    x = torch.randn(5, 27)
    assert torch.equal(a.select_action(x, 0.0), b.select_action(x, 0.0))


def test_sac_fixed_alpha_update_matches_reference():
    from tracktolearn_amd.algorithms.sac import SAC
    z = load_trace('learner_sac')
    alg = SAC(27, 3, '32-32', lr=3e-4, gamma=0.99, alpha=0.2, n_actors=8,
              batch_size=64, replay_size=1000, rng=None, device=CPU)
    alg.agent.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    alg.target.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    assert alg.start_timesteps == 1000
    eps = iter(torch.from_numpy(z['eps']).reshape(-1, 64, 3))
    alg.noise_fn = lambda like: next(eps)
    batch = _batch(z)
    for u in range(int(z['n_updates'])):
        losses = alg.update(batch)
        assert abs(float(losses['critic_loss']) - float(z[f'u{u}/critic_loss'])) < 1e-5
        assert abs(float(losses['actor_loss']) - float(z[f'u{u}/actor_loss'])) < 1e-5
        _check_sd(alg.agent.actor, z, f'u{u}/actor', 2e-6)
        _check_sd(alg.agent.critic, z, f'u{u}/critic', 2e-6)
        _check_sd(alg.target.critic, z, f'u{u}/target_critic', 2e-6)


def test_ddpg_update_matches_reference(monkeypatch):
    from tracktolearn_amd.algorithms.ddpg import DDPG
    z = load_trace('learner_ddpg')
    alg = DDPG(27, 3, '32-32', action_std=0.35, lr=3e-4, gamma=0.99, n_actors=8,
               batch_size=64, replay_size=1000, rng=None, device=CPU)
    alg.agent.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    alg.target.load_state_dict((_load_sd(z, 'init/actor'), _load_sd(z, 'init/critic')))
    eps = iter(torch.from_numpy(z['eps']))
    monkeypatch.setattr(torch, 'randn_like', lambda t, **kw: next(eps))
    batch = _batch(z)
    for u in range(int(z['n_updates'])):
        alg.update(batch)
        _check_sd(alg.agent.actor, z, f'u{u}/actor', 2e-6)
        _check_sd(alg.agent.critic, z, f'u{u}/critic', 2e-6)
        _check_sd(alg.target.actor, z, f'u{u}/target_actor', 2e-6)


def test_select_action_without_log_prob_equals_the_full_forward():
    """SACActorCritic.select_action samples through MaxEntropyActor.sample (no
    log-probability, 9 launches instead of ~35): the action has the bits of
    ``act()``'s for prob 0 (no draw), 1 and in between (same single draw)."""
    from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic
    torch.manual_seed(0)
    ac = SACActorCritic(50, 3, '64-64', torch.device('cpu'))
    x = torch.randn(17, 50)
    state = torch.get_rng_state()
    assert torch.equal(ac.select_action(x, 0.0), ac.act(x, 0.0)[0])
    torch.set_rng_state(state)
    ac.select_action(x, 0.0)
    assert torch.equal(torch.get_rng_state(), state)        # nothing drawn
    for prob in (1.0, 0.3):
        torch.manual_seed(5)
        a = ac.select_action(x, prob)
        torch.manual_seed(5)
        assert torch.equal(a, ac.act(x, prob)[0])


def test_mlp_inference_fuses_relu_without_changing_a_bit():
    """offpolicy.mlp_inference: Linear + ReLU pairs through the GEMM's ReLU
    epilogue under no_grad; the ordinary path with autograd on."""
    from tracktolearn_amd.algorithms.shared.offpolicy import mlp_inference
    from tracktolearn_amd.algorithms.shared.utils import make_fc_network
    torch.manual_seed(1)
    net = make_fc_network([64, 48], 30, 6)
    x = torch.randn(19, 30)
    want = net(x)
    with torch.no_grad():
        assert torch.equal(mlp_inference(net, x), want)
    got = mlp_inference(net, x)            # autograd on: the plain stack
    assert got.requires_grad and torch.equal(got, want)
