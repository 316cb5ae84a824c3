#!/usr/bin/env python
"""sac_auto_train.py -- train a tracking agent with SAC + automatic entropy
adjustment on the MI355X (mirror of TrackToLearn/trainers/sac_auto_train.py).

    sac_auto_train.py path experiment id dataset_file [options]

The oracle reward needs TractOracle-Net weights that the reference does not
ship; train with ``--oracle_bonus 0`` (BASELINE config 3).
"""
import argparse
from argparse import RawTextHelpFormatter

from tracktolearn_amd.algorithms.sac_auto import SACAuto
from tracktolearn_amd.trainers.train import (TrackToLearnTraining,
                                             add_training_args)
from tracktolearn_amd.utils.torch_utils import get_device


class SACAutoTrackToLearnTraining(TrackToLearnTraining):
    """sac_auto_train.py:19-74."""

    def __init__(self, sac_auto_train_dto, comet_experiment=None):
        super().__init__(sac_auto_train_dto, comet_experiment)
        self.alpha = sac_auto_train_dto['alpha']
        self.batch_size = sac_auto_train_dto['batch_size']
        self.replay_size = sac_auto_train_dto['replay_size']

    def save_hyperparameters(self):
        self.hyperparameters.update(
            {'algorithm': 'SACAuto', 'alpha': self.alpha,
             'batch_size': self.batch_size, 'replay_size': self.replay_size})
        super().save_hyperparameters()

    def get_alg(self, max_nb_steps):
        return SACAuto(self.input_size, self.action_size, self.hidden_dims,
                       self.lr, self.gamma, self.alpha, self.n_actor,
                       self.batch_size, self.replay_size, self.rng, get_device())


def add_sac_auto_args(parser):
    parser.add_argument('--alpha', default=0.2, type=float,
                        help='Initial temperature parameter')
    parser.add_argument('--batch_size', default=2**12, type=int,
                        help='How many tuples to sample from the replay buffer.')
    parser.add_argument('--replay_size', default=1e6, type=int,
                        help='How many tuples to store in the replay buffer.')


def parse_args(argv=None):
    """ Train a tracking agent with SAC (automatic entropy adjustment). """
    parser = argparse.ArgumentParser(description=parse_args.__doc__,
                                     formatter_class=RawTextHelpFormatter)
    add_training_args(parser)
    add_sac_auto_args(parser)
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    print(args)
    SACAutoTrackToLearnTraining(vars(args)).run()


if __name__ == '__main__':
    main()
