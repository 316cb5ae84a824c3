"""TD3 (Fujimoto et al. 2018) -- mirror of TrackToLearn/algorithms/td3.py.  On
a CUDA device ``update`` runs as shared/fused.py:FusedTD3Update (as DDPG's)."""
import torch
import torch.nn.functional as F

from tracktolearn_amd.algorithms.ddpg import DDPG
from tracktolearn_amd.algorithms.shared.offpolicy import TD3ActorCritic


class TD3(DDPG):
    """Double critic, clipped target-policy noise, delayed actor/target
    updates (td3.py:20-230)."""

    agent_cls = TD3ActorCritic

    def __init__(self, input_size, action_size, hidden_dims, action_std=0.35,
                 lr=3e-4, gamma=0.99, n_actors=4096, batch_size=2 ** 12,
                 replay_size=1e6, rng=None, device=None):
        super().__init__(input_size, action_size, hidden_dims, action_std, lr,
                         gamma, n_actors, batch_size, replay_size, rng, device)
        self.noise_clip = 1.
        self.agent_freq = 2

    def sample_action(self, state):
        """Policy action + exploration noise, clipped to the action range
        (td3.py:111-128; the reference draws the noise from its numpy rng on
        the host, here it is drawn on the device)."""
        with torch.no_grad():
            a = self.agent.select_action(state)
            noise = torch.randn_like(a) * (self.max_action * self.action_std)
            return (a + noise).clamp(-self.max_action, self.max_action)

    def update(self, batch):
        """td3.py:130-230."""
        self.total_it += 1
        state, action, next_state, reward, not_done = batch
        if self._use_fused():
            with torch.no_grad():
                noise = (torch.randn_like(action) * (self.action_std * 2)).clamp(
                    -self.noise_clip, self.noise_clip)
                return self._fused.update(batch, noise,
                                          update_actor=self.total_it % self.agent_freq == 0)
        with torch.no_grad():
            noise = (torch.randn_like(action) * (self.action_std * 2)).clamp(
                -self.noise_clip, self.noise_clip)
            next_action = (self.target.actor(next_state) + noise).clamp(
                -self.max_action, self.max_action)
            tq1, tq2 = self.target.critic(next_state, next_action)
            target_Q = reward + not_done * self.gamma * torch.min(tq1, tq2)
        q1, q2 = self.agent.critic(state, action)
        loss_q1 = F.mse_loss(q1, target_Q)
        loss_q2 = F.mse_loss(q2, target_Q)
        critic_loss = loss_q1 + loss_q2
        losses = {'actor_loss': 0.0, 'critic_loss': critic_loss.detach(),
                  'loss_q1': loss_q1.detach(), 'loss_q2': loss_q2.detach(),
                  'Q1': q1.mean().detach(), 'Q2': q2.mean().detach(),
                  'Q\'': target_Q.mean().detach()}
        self.critic_optimizer.zero_grad()
        critic_loss.backward()
        self.critic_optimizer.step()
        if self.total_it % self.agent_freq == 0:
            actor_loss = -self.agent.critic.Q1(
                state, self.agent.actor(state)).mean()
            losses['actor_loss'] = actor_loss.detach()
            self.actor_optimizer.zero_grad()
            actor_loss.backward()
            self.actor_optimizer.step()
            self._polyak()
        return losses
