"""SAC with a fixed temperature -- mirror of TrackToLearn/algorithms/sac.py."""
import torch

from tracktolearn_amd.algorithms.ddpg import DDPG
from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic


class SAC(DDPG):
    """Soft actor-critic (Haarnoja et al. 2018), entropy coefficient alpha
    fixed (sac.py:20-232)."""

    agent_cls = SACActorCritic

    def __init__(self, input_size, action_size, hidden_dims, lr=3e-4,
                 gamma=0.99, alpha=0.2, n_actors=4096, batch_size=2 ** 12,
                 replay_size=1e6, rng=None, device=None):
        super().__init__(input_size, action_size, hidden_dims, 0.0, lr, gamma,
                         n_actors, batch_size, replay_size, rng, device)
        self.alpha = alpha
        #: optional hook returning the N(0,1) draws of the actor (tests)
        self.noise_fn = None

    def sample_action(self, state):
        """sac.py:123-133: a fully stochastic action."""
        with torch.no_grad():
            return self.agent.select_action(state, probabilistic=1.0)

    def _eps(self, like):
        return self.noise_fn(like) if self.noise_fn is not None else None

    def _soft_q_losses(self, batch, alpha, pi, logp_pi):
        """Actor loss and critic loss shared by SAC and SACAuto."""
        state, action, next_state, reward, not_done = batch
        q1_pi, q2_pi = self.agent.critic(state, pi)
        actor_loss = (alpha * logp_pi - torch.min(q1_pi, q2_pi)).mean()
        with torch.no_grad():
            # target actions come from the *current* policy
            next_action, logp_next = self.agent.act(
                next_state, probabilistic=1.0, eps=self._eps(action))
            tq1, tq2 = self.target.critic(next_state, next_action)
            backup = reward + self.gamma * not_done * (
                torch.min(tq1, tq2) - alpha * logp_next)
        q1, q2 = self.agent.critic(state, action)
        loss_q1 = ((q1 - backup) ** 2).mean()
        loss_q2 = ((q2 - backup) ** 2).mean()
        return actor_loss, loss_q1 + loss_q2, (loss_q1, loss_q2, q1, q2, backup)

    def _step_actor_critic(self, actor_loss, critic_loss):
        self.actor_optimizer.zero_grad()
        actor_loss.backward()
        self.actor_optimizer.step()
        # the actor backward also left gradients on the critic; they are
        # discarded here, as in the reference
        self.critic_optimizer.zero_grad()
        critic_loss.backward()
        self.critic_optimizer.step()
        self._polyak()

    def update(self, batch):
        """sac.py:135-232."""
        self.total_it += 1
        state = batch[0]
        pi, logp_pi = self.agent.act(state, probabilistic=1.0,
                                     eps=self._eps(batch[1]))
        actor_loss, critic_loss, (l1, l2, q1, q2, backup) = \
            self._soft_q_losses(batch, self.alpha, pi, logp_pi)
        losses = {'actor_loss': actor_loss.detach(),
                  'critic_loss': critic_loss.detach(),
                  'loss_q1': l1.detach(), 'loss_q2': l2.detach(),
                  'Q1': q1.mean().detach(), 'Q2': q2.mean().detach(),
                  'backup': backup.mean().detach()}
        self._step_actor_critic(actor_loss, critic_loss)
        return losses
