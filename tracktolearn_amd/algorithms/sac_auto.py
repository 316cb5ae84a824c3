"""SAC with automatic temperature -- mirror of
TrackToLearn/algorithms/sac_auto.py (the learner of sac_auto_train.py)."""
import numpy as np
import torch

from tracktolearn_amd.algorithms.sac import SAC


class SACAuto(SAC):
    """Soft actor-critic whose entropy coefficient alpha = exp(log_alpha) is
    learned towards a target entropy of -|A| (sac_auto.py:20-250)."""

    def __init__(self, input_size, action_size, hidden_dims, lr=3e-4,
                 gamma=0.99, alpha=0.2, n_actors=4096, batch_size=2 ** 12,
                 replay_size=1e6, rng=None, device=None):
        super().__init__(input_size, action_size, hidden_dims, lr, gamma, alpha,
                         n_actors, batch_size, replay_size, rng, device)
        self.target_entropy = -np.prod(action_size).item()
        self.log_alpha = torch.full((1,), np.log(alpha), requires_grad=True,
                                    device=self.device)
        self.alpha_optimizer = torch.optim.Adam([self.log_alpha], lr=lr)
        self.on_agent = False
        self.start_timesteps = 80000
        self.agent_freq = 1

    def _optimizers(self):
        return [self.alpha_optimizer, self.actor_optimizer, self.critic_optimizer]

    def _update_impl(self, batch):
        """sac_auto.py:139-250: temperature, actor, critic steps in that
        order, then Polyak averaging of critic and actor targets.  Returns an
        empty dict, as the reference does (all its entries are commented)."""
        if self._use_fused():
            return self._update_fused(batch, want_losses=False)
        state = batch[0]
        pi, logp_pi = self.agent.act(state, probabilistic=1.0,
                                     eps=self._eps(batch[1]))
        alpha_loss = -(self.log_alpha *
                       (logp_pi + self.target_entropy).detach()).mean()
        alpha = self.log_alpha.exp()
        actor_loss, critic_loss, _ = self._soft_q_losses(batch, alpha, pi, logp_pi)
        self.alpha_optimizer.zero_grad()
        alpha_loss.backward()
        self._sync_grads([self.log_alpha])
        self.alpha_optimizer.step()
        self._step_actor_critic(actor_loss, critic_loss)
        return {}
