"""Small helpers of the learner (TrackToLearn/algorithms/shared/utils.py)."""
import numpy as np
import torch
from torch import nn


def add_item_to_means(means, dic):
    """Append each value of ``dic`` to the list kept under its key."""
    return {k: means[k] + [dic[k]] for k in dic.keys()}


def add_to_means(means, dic):
    return {k: means[k] + dic[k] for k in dic.keys()}


def mean_losses(dic):
    return {k: np.mean(torch.stack([torch.as_tensor(v) for v in dic[k]])
                       .cpu().numpy(), axis=0) for k in dic.keys()}


def mean_rewards(dic):
    return {k: np.mean(np.asarray(dic[k]), axis=0) for k in dic.keys()}


def format_widths(widths_str):
    """'1024-1024-1024' -> array([1024, 1024, 1024])."""
    return np.asarray([int(i) for i in str(widths_str).split('-')])


def make_fc_network(widths, input_size, output_size, activation=nn.ReLU):
    """Linear/ReLU stack ending in a bare Linear.  The Sequential indices
    (0, 2, 4, ...) are the checkpoint keys of the reference's
    ``make_fc_network`` (shared/utils.py:41-51), so ``*_actor.pth`` /
    ``*_critic.pth`` files load unchanged."""
    sizes = [int(input_size)] + [int(w) for w in widths]
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [nn.Linear(a, b), activation()]
    layers.append(nn.Linear(sizes[-1], int(output_size)))
    return nn.Sequential(*layers)
