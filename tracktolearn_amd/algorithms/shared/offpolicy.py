"""Actor / critic networks of the off-policy learners, on PyTorch-ROCm (the
MLP GEMMs run on the MI355X matrix cores through rocBLAS/hipBLASLt).

Same module names, constructor arguments, forward contracts and state_dict
keys (``layers.N.*``, ``q1.N.*``, ``q2.N.*``) as
TrackToLearn/algorithms/shared/offpolicy.py, so checkpoints are
interchangeable.  The gaussian draw of the max-entropy actor can be injected
(``eps=``) for known-answer tests.
"""
import math
import os
from os.path import join as pjoin

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from tracktolearn_amd.algorithms.shared.utils import (format_widths,
                                                      make_fc_network)

LOG_STD_MAX = 2
LOG_STD_MIN = -20
_HALF_LOG_2PI = math.log(math.sqrt(2 * math.pi))


def mlp_inference(layers, x):
    """``layers(x)`` for a Linear / ReLU stack without autograd, with every
    Linear + ReLU pair as ONE launch (the GEMM's ReLU epilogue,
    ``torch._addmm_activation``: the same bits as ``relu(linear(x))`` on this
    stack -- checked in tests -- and one launch less per hidden layer; a tracking
    step at a few hundred rows is bound by its launches).  Anything else in the
    stack, or a build without that entry point, takes the ordinary path.

    ``TTL_POLICY_TILE_ROWS=R`` (off by default) evaluates the stack in tiles of
    exactly R rows (the last one zero padded): every GEMM then has the same
    shape whatever the batch size, so a row's action no longer depends on how
    many other rows share its batch -- a tractogram tracked in shards (one seed
    batch split over several GPUs, runners/ttl_track.py) is then identical to
    the one-process tractogram bit for bit, at the price of the padding."""
    tile = int(os.environ.get('TTL_POLICY_TILE_ROWS', '0') or 0)
    if tile > 0 and x.dim() == 2 and not torch.is_grad_enabled():
        n = x.shape[0]
        pad = (-n) % tile
        if pad:
            x = torch.cat([x, x.new_zeros((pad, x.shape[1]))])
        out = [_mlp_inference(layers, x[i:i + tile]) for i in range(0, n + pad, tile)]
        return (out[0] if len(out) == 1 else torch.cat(out))[:n]
    return _mlp_inference(layers, x)


def _mlp_inference(layers, x):
    fused = getattr(torch, '_addmm_activation', None)
    if fused is None or x.dim() != 2 or torch.is_grad_enabled():
        return layers(x)
    mods = list(layers)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear) and m.bias is not None and i + 1 < len(mods) \
                and type(mods[i + 1]) is nn.ReLU:
            x = fused(m.bias, x, m.weight.t(), use_gelu=False)
            i += 2
        else:
            x = m(x)
            i += 1
    return x


class Actor(nn.Module):
    """Deterministic policy: state -> tanh(MLP(state)) (offpolicy.py:18-60)."""

    def __init__(self, state_dim, action_dim, hidden_dims, output_activation=nn.Tanh):
        super().__init__()
        self.action_dim = action_dim
        self.hidden_layers = format_widths(hidden_dims)
        self.layers = make_fc_network(self.hidden_layers, state_dim, action_dim)
        self.output_activation = output_activation()

    def forward(self, state):
        return self.output_activation(mlp_inference(self.layers, state))


class MaxEntropyActor(Actor):
    """Squashed-gaussian policy of SAC (offpolicy.py:62-140): the MLP emits
    mean || log_std; returns (tanh(u), log pi(u)) with the tanh correction."""

    def __init__(self, state_dim, action_dim, hidden_dims):
        super().__init__(state_dim, action_dim, hidden_dims)
        self.layers = make_fc_network(self.hidden_layers, state_dim,
                                      action_dim * 2)

    def forward(self, state, probabilistic, eps=None):
        p = mlp_inference(self.layers, state)
        mu = p[:, :self.action_dim]
        log_std = torch.clamp(p[:, self.action_dim:], LOG_STD_MIN, LOG_STD_MAX)
        std = torch.exp(log_std) * probabilistic
        if eps is None:
            eps = torch.randn_like(mu)
        u = mu + eps * std                       # reparametrised sample
        # log N(u; mu, std), summed over the action, then the tanh correction
        # 2 (log 2 - u - softplus(-2u))  (SAC, arXiv 1801.01290 app. C)
        var = std ** 2
        logp = (-((u - mu) ** 2) / (2 * var) - std.log() - _HALF_LOG_2PI).sum(axis=-1)
        logp = logp - (2 * (np.log(2) - u - F.softplus(-2 * u))).sum(axis=1)
        return self.output_activation(u), logp

    def sample(self, state, probabilistic):
        """The action of ``forward`` without its log-probability (tracking and
        sample gathering only use the action, offpolicy.py:467-482): the same
        ``tanh(mu + eps * std)``, bit for bit, in 9 launches instead of ~35 --
        with the default networks a tracking step is bound by exactly those
        launches.  ``probabilistic == 0`` gives ``tanh(mu)`` (``eps * 0`` adds
        nothing) and draws no random numbers."""
        if state.is_cuda and state.dim() == 2 and not torch.is_grad_enabled() and \
                state.dtype == torch.float32 and self.action_dim <= 4 and \
                os.environ.get('TTL_FUSED_POLICY_HEAD', '1') != '0' and \
                not int(os.environ.get('TTL_POLICY_TILE_ROWS', '0') or 0):
            return self._sample_fused_head(state, probabilistic)
        p = mlp_inference(self.layers, state)
        mu = p[:, :self.action_dim]
        if not probabilistic:
            return self.output_activation(mu)
        log_std = torch.clamp(p[:, self.action_dim:], LOG_STD_MIN, LOG_STD_MAX)
        std = torch.exp(log_std) * probabilistic
        return self.output_activation(mu + torch.randn_like(mu) * std)

    def _sample_fused_head(self, state, probabilistic):
        """``sample`` on the GPU with the head -- the 2A-wide Linear, clamp, exp, the
        gaussian draw and tanh -- as ONE launch of ``ttl_thin_forward`` (include/
        ttl_learner.h; the kernel the learner's update uses) instead of a GEMM
        with N = 6 and seven element-wise kernels.  Same formula: tanh(mu + eps
        * probabilistic * exp(clamp(log_std))), one ``randn`` of the same shape
        (none for probabilistic = 0); every row is computed on its own, so a
        row's action does not depend on its batch."""
        from tracktolearn_amd.algorithms.shared.fused import HEAD_SAC, HipOps
        mods = list(self.layers)
        head = mods[-1]
        x = _mlp_inference(nn.Sequential(*mods[:-1]), state) if len(mods) > 1 else state
        if x.stride(1) != 1:
            x = x.contiguous()
        n, a = x.shape[0], self.action_dim
        ops = getattr(self, '_head_ops', None)
        if ops is None or ops.index != (x.device.index or 0):
            ops = self._head_ops = HipOps(x.device)
        if probabilistic:
            eps = torch.randn((n, a), dtype=torch.float32, device=x.device)
            if probabilistic != 1.0:
                eps = eps * probabilistic
        else:
            eps = torch.zeros((n, a), dtype=torch.float32, device=x.device)
        out = torch.empty((n, a), dtype=torch.float32, device=x.device)
        scratch = torch.empty(n * (a + 1), dtype=torch.float32, device=x.device)
        if n:
            ops.thin_forward(x, head.weight.detach(), head.bias.detach(), 2 * a, False, HEAD_SAC,
                             out, a, eps=eps, logp=scratch[:n], ls_raw=scratch[n:].view(n, a))
        return out


class Critic(nn.Module):
    """Q(s, a) (offpolicy.py:143-181)."""

    def __init__(self, state_dim, action_dim, hidden_dims):
        super().__init__()
        self.hidden_layers = format_widths(hidden_dims)
        self.q1 = make_fc_network(self.hidden_layers, state_dim + action_dim, 1)

    def forward(self, state, action):
        return mlp_inference(self.q1, torch.cat([state, action], -1)).squeeze(-1)


class DoubleCritic(Critic):
    """Two independent Q networks (offpolicy.py:183-238)."""

    def __init__(self, state_dim, action_dim, hidden_dims, critic_size_factor=1):
        super().__init__(state_dim, action_dim, hidden_dims)
        self.hidden_layers = format_widths(hidden_dims) * critic_size_factor
        self.q1 = make_fc_network(self.hidden_layers, state_dim + action_dim, 1)
        self.q2 = make_fc_network(self.hidden_layers, state_dim + action_dim, 1)

    def forward(self, state, action):
        sa = torch.cat([state, action], -1)
        return mlp_inference(self.q1, sa).squeeze(-1), mlp_inference(self.q2, sa).squeeze(-1)

    def Q1(self, state, action):
        return mlp_inference(self.q1, torch.cat([state, action], -1)).squeeze(-1)


class ActorCritic(object):
    """Actor + critic pair with the save/load conventions of
    offpolicy.py:240-376 (``<name>_actor.pth`` / ``<name>_critic.pth``)."""

    actor_cls = Actor
    critic_cls = Critic
    #: select_action is plain torch code without a host round trip: it can run
    #: under stream capture (RLAlgorithm.validation_episode's graphed loop)
    graph_safe = True

    def __init__(self, state_dim, action_dim, hidden_dims, device):
        self.device = device
        self.actor = self.actor_cls(state_dim, action_dim, hidden_dims).to(device)
        self.critic = self.critic_cls(state_dim, action_dim, hidden_dims).to(device)

    def act(self, state):
        return self.actor(state)

    def select_action(self, state, probabilistic=0.0):
        if len(state.shape) < 2:
            state = state[None, :]
        return self.act(state)

    def parameters(self):
        return self.actor.parameters()

    def load_state_dict(self, state_dict):
        actor_state_dict, critic_state_dict = state_dict
        self.actor.load_state_dict(actor_state_dict)
        self.critic.load_state_dict(critic_state_dict)

    def state_dict(self):
        return self.actor.state_dict(), self.critic.state_dict()

    def save(self, path, filename):
        torch.save(self.critic.state_dict(), pjoin(path, filename + '_critic.pth'))
        torch.save(self.actor.state_dict(), pjoin(path, filename + '_actor.pth'))

    def load(self, path, filename):
        self.critic.load_state_dict(torch.load(
            pjoin(path, filename + '_critic.pth'), map_location=self.device,
            weights_only=True))
        self.actor.load_state_dict(torch.load(
            pjoin(path, filename + '_actor.pth'), map_location=self.device,
            weights_only=True))

    def eval(self):
        self.actor.eval()
        self.critic.eval()

    def train(self):
        self.actor.train()
        self.critic.train()


class TD3ActorCritic(ActorCritic):
    """Deterministic actor + double critic (offpolicy.py:379-412)."""
    critic_cls = DoubleCritic


class SACActorCritic(ActorCritic):
    """Max-entropy actor + double critic (offpolicy.py:415-482)."""
    actor_cls = MaxEntropyActor
    critic_cls = DoubleCritic

    def act(self, state, probabilistic=1.0, eps=None):
        return self.actor(state, probabilistic, eps=eps)

    def select_action(self, state, probabilistic=1.0):
        if len(state.shape) < 2:
            state = state[None, :]
        return self.actor.sample(state, probabilistic)
