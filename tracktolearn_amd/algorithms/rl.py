"""RLAlgorithm: the tracking (validation) episode loop.

Mirror of TrackToLearn/algorithms/rl.py.  ``validation_episode`` is one of the
two callers of ``env.step`` / ``env.harvest``; here the loop is
device resident: the policy reads the state tensor the env kernels wrote,
actions never visit the host, and the only per-step host traffic is the 8-byte
survivor count.
"""
import os

import torch

from tracktolearn_amd.utils.torch_utils import get_device


class RLAlgorithm(object):
    """Abstract sample-gathering and training algorithm (rl.py:7-56)."""

    def __init__(self, input_size, action_size=3, hidden_size=256, lr=3e-4,
                 gamma=0.99, batch_size=10000, rng=None, device=None):
        self.max_action = 1.
        self.t = 1
        self.action_size = action_size
        self.lr = lr
        self.gamma = gamma
        self.device = device if device is not None else get_device()
        self.batch_size = batch_size
        self.rng = rng

    #: the graphed tracking loop evaluates the policy on the whole batch at every
    #: step; it is taken when one such evaluation costs at most this many
    #: microseconds (measured once per batch size), i.e. while the step-by-step
    #: loop would be bound by its ~15 launches and the survivor count per step
    #: (measured, benchmarks/bench_tracking_loop.py: a '64-64' policy on 4 096
    #: rows costs 80 us and the graph wins 1.56x; '256-256' on 10 000 rows costs
    #: 95 us and loses 0.73x; the reference's default '1024-1024-1024' on 4 096
    #: rows costs 227 us -- fp32 GEMMs -- and loses 2x)
    graph_policy_us = float(os.environ.get('TTL_GRAPH_POLICY_US', '80'))

    def _can_run_free(self, env):
        """The policy is one of this package's networks (torch code without a
        host round trip, safe under stream capture) and the env can advance on
        its own (``TrackingEnvironment.freerun_supported``).  ``TTL_GRAPH_EPISODE=0``
        keeps the step-by-step loop."""
        agent = getattr(self, 'agent', None)
        return (os.environ.get('TTL_GRAPH_EPISODE', '1') != '0'
                and getattr(type(agent), 'graph_safe', False)
                and 'select_action' not in vars(agent)      # not wrapped / replaced
                and hasattr(env, 'freerun_supported') and env.freerun_supported())

    def validation_episode(self, initial_state, env, prob=1.):
        """Run the policy until every streamline of the batch has stopped
        (rl.py:58-106).  Returns the summed reward (0 when the env computes
        none, as the reference's ``sum(zeros)``)."""
        if RLAlgorithm._can_run_free(self, env):
            # small batches (the default --n_actor of ttl_track among them) are
            # bound by the host's launches and by the survivor count it waits
            # for every step: run policy + step as one replayed HIP graph
            agent = self.agent
            out = env.run_free(
                lambda s: agent.select_action(s, probabilistic=prob), initial_state,
                key=(id(agent), float(prob)), owner=agent,
                max_policy_us=getattr(self, 'graph_policy_us', RLAlgorithm.graph_policy_us))
            if out is None and not prob and os.environ.get('TTL_FREE_RUNNING_EAGER', '0') == '1':
                # Opt-in: the policy is too expensive to run on the full batch at
                # every step, so launch it on the newest reported survivor count,
                # still without ever waiting for a step (2-10 % faster than the
                # loop below).  Not the default because how many rows a launch
                # covers then depends on when the GPU reported its counts: a
                # sampling policy would draw a different random stream from run
                # to run, and even the deterministic one may round differently
                # when the GEMM heuristics switch kernels with the row count --
                # tractograms would not be reproducible bit for bit.
                out = env.run_free_eager(
                    lambda s: agent.select_action(s, probabilistic=prob), initial_state)
            if out is not None:
                return float(out[0]) if out[0] is not None else 0.0
        running_reward = None
        state = initial_state
        while state.shape[0] > 0:
            with torch.no_grad():
                action = self.agent.select_action(state, probabilistic=prob)
            _, reward, _, _ = env.step_device(action)
            if reward is not None:
                r = reward.sum()
                running_reward = r if running_reward is None else running_reward + r
            # harvesting drops the finished streamlines from the state
            state, _ = env.harvest()
        return float(running_reward) if running_reward is not None else 0.0
