"""SAC with a fixed temperature -- mirror of TrackToLearn/algorithms/sac.py.

On a CUDA device ``update`` runs as ``shared/fused.py``'s hand-scheduled
forward/backward (GEMMs on PyTorch-ROCm + the HIP learner kernels of
libttl_hip.so, include/ttl_learner.h); on the CPU (known-answer tests) as the
autograd formulation below, which is also what ``TTL_FUSED_LEARNER=0`` keeps on
the GPU for A/B measurements."""
import os

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
        self._graph = None
        self._graph_batch = None
        # data-parallel learner (``enable_data_parallel``): every rank samples
        # its own replay ring and the gradients are averaged over the process
        # group before each optimizer step; DDPG._schedule keeps the number of
        # updates per rank equal

    def enable_data_parallel(self, group=None):
        """One learner replica per GPU (the reference has a single learner;
        SURVEY 8e): weights are broadcast from rank 0 once, then every
        update averages the gradients with one flattened all-reduce per
        network, so the replicas stay bit-identical while each consumes the
        transitions of its own streamline shard."""
        import torch.distributed as dist
        from tracktolearn_amd.parallel import broadcast_parameters
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('enable_data_parallel needs torch.distributed')
        if self._graph is not None:
            raise RuntimeError('the graphed update is single-GPU only')
        self._dp, self._dp_group = True, group
        mods = [self.agent.actor, self.agent.critic, self.target.actor,
                self.target.critic]
        broadcast_parameters(mods, 0, group)
        if hasattr(self, 'log_alpha'):
            dist.broadcast(self.log_alpha.data, src=0, group=group)

    def _sync_grads(self, params):
        if self._dp:
            from tracktolearn_amd.parallel import all_reduce_gradients
            all_reduce_gradients(list(params), self._dp_group)

    # ------------------------------------------------------------------ #
    # HIP-graph replay of the update.  One update is ~150 small kernels
    # (3 MLP forwards/backwards, 3 Adam steps, Polyak); at batch 4096 their
    # launch overhead is comparable to the GEMM time, so the whole update is
    # captured once into a HIP graph and replayed with static input buffers.
    def _optimizers(self):
        return [self.actor_optimizer, self.critic_optimizer]

    def enable_graph(self, warmup=3):
        """Capture ``update`` into a HIP graph at its first call.  Must be
        called before the first update (the Adam states are made capturable)."""
        if self.total_it != 0:
            raise RuntimeError('enable_graph() must precede the first update')
        for opt in self._optimizers():
            for group in opt.param_groups:
                group['capturable'] = True
        self._graph = 'pending'
        self._graph_warmup = int(warmup)

    def _capture(self, batch):
        self._graph_batch = [torch.empty_like(b) for b in batch]
        for dst, src in zip(self._graph_batch, batch):
            dst.copy_(src)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(self._graph_warmup):
                self._update_impl(self._graph_batch)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._graph_losses = self._update_impl(self._graph_batch)
        self._graph = graph
        # capturing records the kernels without running them: replay once so
        # that this call has performed warmup + 1 real updates
        graph.replay()
        return self._graph_warmup + 1

    def update(self, batch):
        """One gradient update (see ``_update_impl``); replayed from a HIP
        graph after ``enable_graph()``."""
        if self._graph is None:
            self.total_it += 1
            return self._update_impl(batch)
        if self._graph == 'pending':
            n = self._capture(batch)
            self.total_it += n
            return self._graph_losses
        if batch[0].shape != self._graph_batch[0].shape:
            raise RuntimeError('graphed update needs a fixed batch shape')
        for dst, src in zip(self._graph_batch, batch):
            dst.copy_(src)
        self._graph.replay()
        self.total_it += 1
        return self._graph_losses

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
        self._sync_grads(self.agent.actor.parameters())
        self.actor_optimizer.step()
        # the actor backward also left gradients on the critic; they are
        # discarded here, as in the reference
        self.critic_optimizer.zero_grad()
        critic_loss.backward()
        self._sync_grads(self.agent.critic.parameters())
        self.critic_optimizer.step()
        self._polyak()

    def _fused_class(self):
        from tracktolearn_amd.algorithms.shared.fused import FusedSACUpdate
        return FusedSACUpdate

    def _update_fused(self, batch, want_losses):
        # the two gaussian draws in the reference's order: pi(s), then pi(s')
        eps_pi, eps_next = self._eps(batch[1]), self._eps(batch[1])
        with torch.no_grad():
            return self._fused.update(batch, eps_pi, eps_next, want_losses=want_losses)

    def _update_impl(self, batch):
        """sac.py:135-232."""
        if self._use_fused():
            return self._update_fused(batch, want_losses=True)
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
