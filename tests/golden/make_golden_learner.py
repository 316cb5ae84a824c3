#!/usr/bin/env python3
"""Known-answer vectors for the learner rows of SURVEY 8(a) (a19-a21), from the
REFERENCE's own SACAuto / TD3 / OffPolicyReplayBuffer run on CPU in the build
container (same import harness as make_golden.py).

Gaussian noise is injected: ``torch.distributions.normal._standard_normal``
(what ``Normal.rsample`` draws) and ``torch.randn_like`` are replaced by
functions that replay recorded tensors, so an independent implementation can
consume the same noise.  Tiny networks (hidden_dims '32-32', state width 27).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_reference, _save  # noqa: E402


def _flat(prefix, sd, out):
    for k, v in sd.items():
        out[f'{prefix}/{k}'] = v.detach().cpu().numpy().copy()


def sac_auto(ref_mods):
    from TrackToLearn.algorithms.sac_auto import SACAuto
    import torch.distributions.normal as tdn
    W, A, B = 27, 3, 64
    torch.manual_seed(0)
    alg = SACAuto(W, A, '32-32', lr=3e-4, gamma=0.99, alpha=0.2, n_actors=8,
                  batch_size=B, replay_size=1000,
                  rng=np.random.RandomState(0), device=torch.device('cpu'))
    out = {}
    _flat('init/actor', alg.agent.actor.state_dict(), out)
    _flat('init/critic', alg.agent.critic.state_dict(), out)
    rng = np.random.RandomState(5)
    batch = [rng.standard_normal((B, W)).astype(np.float32),
             np.tanh(rng.standard_normal((B, A))).astype(np.float32),
             rng.standard_normal((B, W)).astype(np.float32),
             rng.uniform(-1, 1, B).astype(np.float32),
             (rng.uniform(size=B) > 0.2).astype(np.float32)]
    for n, b in zip(['state', 'action', 'next_state', 'reward', 'not_done'], batch):
        out[f'batch/{n}'] = b
    n_updates = 3
    eps = rng.standard_normal((n_updates, 2, B, A)).astype(np.float32)
    out['eps'] = eps
    calls = {'i': 0}
    orig = tdn._standard_normal

    def replay(shape, dtype, device):
        e = torch.from_numpy(eps.reshape(-1, B, A)[calls['i']])
        calls['i'] += 1
        assert tuple(shape) == tuple(e.shape)
        return e
    tdn._standard_normal = replay
    try:
        tb = [torch.from_numpy(b) for b in batch]
        for u in range(n_updates):
            alg.update(tb)
            _flat(f'u{u}/actor', alg.agent.actor.state_dict(), out)
            _flat(f'u{u}/critic', alg.agent.critic.state_dict(), out)
            _flat(f'u{u}/target_actor', alg.target.actor.state_dict(), out)
            _flat(f'u{u}/target_critic', alg.target.critic.state_dict(), out)
            out[f'u{u}/log_alpha'] = alg.log_alpha.detach().numpy().copy()
    finally:
        tdn._standard_normal = orig
    # deterministic policy outputs (probabilistic = 0) and one stochastic draw
    st = torch.from_numpy(batch[0])
    with torch.no_grad():
        out['act_det'] = alg.agent.select_action(st, probabilistic=0.0).numpy()
        calls['i'] = 0
        tdn._standard_normal = replay
        try:
            a, logp = alg.agent.act(st, probabilistic=1.0)
        finally:
            tdn._standard_normal = orig
        out['act_sto'] = a.numpy()
        out['act_sto_logp'] = logp.numpy()
    out['n_updates'] = n_updates
    _save('learner_sac_auto', out)


def td3(ref_mods):
    from TrackToLearn.algorithms.td3 import TD3
    W, A, B = 27, 3, 64
    torch.manual_seed(1)
    alg = TD3(W, A, '32-32', action_std=0.35, lr=3e-4, gamma=0.99, n_actors=8,
              batch_size=B, replay_size=1000, rng=np.random.RandomState(0),
              device=torch.device('cpu'))
    out = {}
    _flat('init/actor', alg.agent.actor.state_dict(), out)
    _flat('init/critic', alg.agent.critic.state_dict(), out)
    rng = np.random.RandomState(6)
    batch = [rng.standard_normal((B, W)).astype(np.float32),
             np.tanh(rng.standard_normal((B, A))).astype(np.float32),
             rng.standard_normal((B, W)).astype(np.float32),
             rng.uniform(-1, 1, B).astype(np.float32),
             (rng.uniform(size=B) > 0.2).astype(np.float32)]
    for n, b in zip(['state', 'action', 'next_state', 'reward', 'not_done'], batch):
        out[f'batch/{n}'] = b
    n_updates = 4
    eps = rng.standard_normal((n_updates, B, A)).astype(np.float32)
    out['eps'] = eps
    calls = {'i': 0}
    orig = torch.randn_like

    def replay(t, **kw):
        e = torch.from_numpy(eps[calls['i']])
        calls['i'] += 1
        return e
    torch.randn_like = replay
    try:
        tb = [torch.from_numpy(b) for b in batch]
        for u in range(n_updates):
            alg.update(tb)
            _flat(f'u{u}/actor', alg.agent.actor.state_dict(), out)
            _flat(f'u{u}/critic', alg.agent.critic.state_dict(), out)
            _flat(f'u{u}/target_actor', alg.target.actor.state_dict(), out)
            _flat(f'u{u}/target_critic', alg.target.critic.state_dict(), out)
    finally:
        torch.randn_like = orig
    out['n_updates'] = n_updates
    out['action_std'] = 0.35
    _save('learner_td3', out)


def sac_fixed_alpha(ref_mods):
    from TrackToLearn.algorithms.sac import SAC
    import torch.distributions.normal as tdn
    W, A, B = 27, 3, 64
    torch.manual_seed(2)
    alg = SAC(W, A, '32-32', lr=3e-4, gamma=0.99, alpha=0.2, n_actors=8,
              batch_size=B, replay_size=1000, rng=np.random.RandomState(0),
              device=torch.device('cpu'))
    out = {}
    _flat('init/actor', alg.agent.actor.state_dict(), out)
    _flat('init/critic', alg.agent.critic.state_dict(), out)
    rng = np.random.RandomState(8)
    batch = [rng.standard_normal((B, W)).astype(np.float32),
             np.tanh(rng.standard_normal((B, A))).astype(np.float32),
             rng.standard_normal((B, W)).astype(np.float32),
             rng.uniform(-1, 1, B).astype(np.float32),
             (rng.uniform(size=B) > 0.2).astype(np.float32)]
    for n, b in zip(['state', 'action', 'next_state', 'reward', 'not_done'], batch):
        out[f'batch/{n}'] = b
    n_updates = 3
    eps = rng.standard_normal((n_updates, 2, B, A)).astype(np.float32)
    out['eps'] = eps
    calls = {'i': 0}
    orig = tdn._standard_normal

    def replay(shape, dtype, device):
        e = torch.from_numpy(eps.reshape(-1, B, A)[calls['i']])
        calls['i'] += 1
        return e
    tdn._standard_normal = replay
    try:
        tb = [torch.from_numpy(b) for b in batch]
        for u in range(n_updates):
            losses = alg.update(tb)
            out[f'u{u}/critic_loss'] = np.float64(losses['critic_loss'])
            out[f'u{u}/actor_loss'] = np.float64(losses['actor_loss'])
            _flat(f'u{u}/actor', alg.agent.actor.state_dict(), out)
            _flat(f'u{u}/critic', alg.agent.critic.state_dict(), out)
            _flat(f'u{u}/target_critic', alg.target.critic.state_dict(), out)
    finally:
        tdn._standard_normal = orig
    out['n_updates'] = n_updates
    _save('learner_sac', out)


def ddpg(ref_mods):
    from TrackToLearn.algorithms.ddpg import DDPG
    W, A, B = 27, 3, 64
    torch.manual_seed(4)
    alg = DDPG(W, A, '32-32', action_std=0.35, lr=3e-4, gamma=0.99, n_actors=8,
               batch_size=B, replay_size=1000, rng=np.random.RandomState(0),
               device=torch.device('cpu'))
    out = {}
    _flat('init/actor', alg.agent.actor.state_dict(), out)
    _flat('init/critic', alg.agent.critic.state_dict(), out)
    rng = np.random.RandomState(9)
    batch = [rng.standard_normal((B, W)).astype(np.float32),
             np.tanh(rng.standard_normal((B, A))).astype(np.float32),
             rng.standard_normal((B, W)).astype(np.float32),
             rng.uniform(-1, 1, B).astype(np.float32),
             (rng.uniform(size=B) > 0.2).astype(np.float32)]
    for n, b in zip(['state', 'action', 'next_state', 'reward', 'not_done'], batch):
        out[f'batch/{n}'] = b
    n_updates = 3
    eps = rng.standard_normal((n_updates, B, A)).astype(np.float32)
    out['eps'] = eps
    calls = {'i': 0}
    orig = torch.randn_like

    def replay(t, **kw):
        e = torch.from_numpy(eps[calls['i']])
        calls['i'] += 1
        return e
    torch.randn_like = replay
    try:
        tb = [torch.from_numpy(b) for b in batch]
        for u in range(n_updates):
            alg.update(tb)
            _flat(f'u{u}/actor', alg.agent.actor.state_dict(), out)
            _flat(f'u{u}/critic', alg.agent.critic.state_dict(), out)
            _flat(f'u{u}/target_actor', alg.target.actor.state_dict(), out)
    finally:
        torch.randn_like = orig
    out['n_updates'] = n_updates
    _save('learner_ddpg', out)


def replay_buffer(ref_mods):
    from TrackToLearn.algorithms.shared.replay import OffPolicyReplayBuffer
    buf = OffPolicyReplayBuffer(5, 3, max_size=10)
    rng = np.random.RandomState(2)
    out = {}
    sizes = [4, 4, 4, 7, 1]
    for i, n in enumerate(sizes):
        s = torch.from_numpy(rng.standard_normal((n, 5)).astype(np.float32))
        a = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32))
        ns = torch.from_numpy(rng.standard_normal((n, 5)).astype(np.float32))
        r = torch.from_numpy(rng.standard_normal((n, 1)).astype(np.float32))
        d = torch.from_numpy((rng.uniform(size=(n, 1)) > 0.5).astype(np.float32))
        buf.add(s, a, ns, r, d)
        for nm, t in zip(['s', 'a', 'ns', 'r', 'd'], [s, a, ns, r, d]):
            out[f'add{i}/{nm}'] = t.numpy()
        out[f'after{i}/ptr'] = np.int64(buf.ptr)
        out[f'after{i}/size'] = np.int64(buf.size)
        out[f'after{i}/state'] = buf.state.numpy().copy()
        out[f'after{i}/action'] = buf.action.numpy().copy()
        out[f'after{i}/next_state'] = buf.next_state.numpy().copy()
        out[f'after{i}/reward'] = buf.reward.numpy().copy()
        out[f'after{i}/not_done'] = buf.not_done.numpy().copy()
    out['n_adds'] = len(sizes)
    _save('learner_replay', out)


if __name__ == '__main__':
    mods = import_reference()
    sac_auto(mods)
    td3(mods)
    sac_fixed_alpha(mods)
    ddpg(mods)
    replay_buffer(mods)
