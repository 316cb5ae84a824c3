"""RLAlgorithm: the tracking (validation) episode loop.

Mirror of TrackToLearn/algorithms/rl.py.  ``validation_episode`` is one of the
two callers of ``env.step`` / ``env.harvest``; here the loop is
device resident: the policy reads the state tensor the env kernels wrote,
actions never visit the host, and the only per-step host traffic is the 8-byte
survivor count.
"""
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

    def validation_episode(self, initial_state, env, prob=1.):
        """Run the policy until every streamline of the batch has stopped
        (rl.py:58-106).  Returns the summed reward (0 when the env computes
        none, as the reference's ``sum(zeros)``)."""
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
