#!/usr/bin/env python3
"""ttl_track_from_hdf5.py -- track a subject of a dataset file with a trained
agent on the MI355X.

Same command line as TrackToLearn/runners/ttl_track_from_hdf5.py
(`path experiment id dataset_file agent subject_id hyperparameters` plus the
model / reward / tractometer / oracle / tracking option groups).  Upstream this
script cannot run as shipped (it reads a `prob` key argparse never sets and
calls `env.set_step_size`, which no environment defines -- SURVEY App. E.7; the
reference's own test only runs `--help`); here `run()` does what the script
evidently intends: the noisy tracking env over the dataset, the step rescaled
to the subject's voxel size, `Tracker.track`, a .tck under `path`.
"""
import argparse
import json
import os
import random
from argparse import RawTextHelpFormatter
from os.path import join

import numpy as np
import torch

from tracktolearn_amd.trainers.train import (add_experiment_args,
                                             add_model_args, add_oracle_args,
                                             add_reward_args,
                                             add_tracking_args,
                                             add_tractometer_args)


class TrackToLearnValidation(object):
    """ttl_track_from_hdf5.py:28-176 of the reference."""

    def __init__(self, valid_dto):
        g = valid_dto
        self.experiment_path = g['path']
        self.experiment = g['experiment']
        self.id = g['id']
        self.dataset_file = g['dataset_file']
        self.subject_id = g['subject_id']
        self.prob = g.get('prob', 0.0)
        self.noise = g['noise']
        self.agent = g['agent']
        self.n_actor = g['n_actor']
        self.npv = g['npv']
        self.min_length = g['min_length']
        self.max_length = g['max_length']
        self.alignment_weighting = g['alignment_weighting']
        if g.get('fa_map'):
            raise NotImplementedError('FA-scaled noise is not supported '
                                      '(noisy_tracking_env.py)')
        with open(g['hyperparameters'], 'r') as json_file:
            hp = json.load(json_file)
        self.algorithm = hp['algorithm']
        self.step_size = float(hp['step_size'])
        self.voxel_size = float(hp.get('voxel_size', 2.0))
        self.theta = hp['max_angle']
        self.hidden_dims = hp['hidden_dims']
        self.n_dirs = hp['n_dirs']
        self.target_sh_order = hp.get('target_sh_order')
        self.binary_stopping_threshold = hp.get('binary_stopping_threshold', 0.5)
        self.random_seed = g['rng_seed']
        torch.manual_seed(self.random_seed)
        np.random.seed(self.random_seed)
        self.rng = np.random.RandomState(seed=self.random_seed)
        random.seed(self.random_seed)

    def get_valid_env(self):
        from tracktolearn_amd.environments.noisy_tracking_env import \
            NoisyTrackingEnvironment
        from tracktolearn_amd.utils.torch_utils import get_device
        self.device = get_device()
        env_dto = {
            'dataset_file': self.dataset_file, 'fa_map': None,
            'n_dirs': self.n_dirs, 'step_size': self.step_size,
            'theta': self.theta, 'min_length': self.min_length,
            'max_length': self.max_length, 'noise': self.noise, 'npv': self.npv,
            'rng': self.rng, 'alignment_weighting': self.alignment_weighting,
            'oracle_bonus': 0.0, 'oracle_stopping_criterion': False,
            'oracle_checkpoint': None, 'scoring_data': None,
            'binary_stopping_threshold': self.binary_stopping_threshold,
            'compute_reward': False, 'device': self.device,
            'target_sh_order': self.target_sh_order,
        }
        env = NoisyTrackingEnvironment.from_dataset(env_dto, 'training')
        # keep drawing subjects until the requested one is loaded
        if self.subject_id is not None and hasattr(env, 'dataset'):
            if self.subject_id not in env.dataset.subjects:
                raise ValueError(f'subject {self.subject_id!r} is not in '
                                 f'{self.dataset_file}')
            for _ in range(4 * len(env.dataset) + 4):
                if env.subject_id == self.subject_id:
                    break
                env.load_subject()
        return env

    def run(self):
        from tracktolearn_amd.algorithms.sac_auto import SACAuto
        from tracktolearn_amd.io import streamlines as sio
        from tracktolearn_amd.tracking.tracker import Tracker, detect_format
        env = self.get_valid_env()
        example_state = env.reset(0, 1)
        input_size = example_state.shape[1]
        tracking_voxel_size = env.get_voxel_size()
        step_size_mm = (tracking_voxel_size / self.voxel_size) * self.step_size
        print('Agent was trained on a voxel size of {}mm and a step size of '
              '{}mm.'.format(self.voxel_size, self.step_size))
        print('Subject has a voxel size of {}mm, setting step size to '
              '{}mm.'.format(tracking_voxel_size, step_size_mm))
        env.set_step_size(step_size_mm)
        alg = {'SACAuto': SACAuto}[self.algorithm](
            input_size, env.get_action_size(), self.hidden_dims,
            n_actors=self.n_actor, rng=self.rng, device=self.device,
            replay_size=1)
        alg.agent.load(self.agent, 'last_model_state')
        tracker = Tracker(alg, self.n_actor, compress=0.0,
                          min_length=self.min_length,
                          max_length=self.max_length, save_seeds=False)
        os.makedirs(self.experiment_path, exist_ok=True)
        out = join(self.experiment_path, 'tractogram_{}_{}_{}.tck'.format(
            self.experiment, self.id, env.subject_id))
        tractogram = tracker.track(env, detect_format(out))
        ref = env.reference if isinstance(env.reference, dict) else {}
        shape = ref.get('shape', env.tracking_mask.data.shape[:3])
        zooms = (float(tracking_voxel_size),) * 3
        header = sio.create_tractogram_header(
            ref.get('affine', env.affine_vox2rasmm), shape, zooms)
        n = sio.save(tractogram, out, header=header)
        print('Saved {} streamlines to {}'.format(n, out))
        return out


def add_valid_args(parser):
    parser.add_argument('dataset_file',
                        help='Path to preprocessed datset file (.hdf5)')
    parser.add_argument('agent', help='Path to the policy')
    parser.add_argument('subject_id', type=str, default=None,
                        help='Subject in HDF5 to track on.')
    parser.add_argument('hyperparameters',
                        help='File containing the hyperparameters for the '
                             'experiment')
    parser.add_argument('--fa_map', type=str, default=None,
                        help='FA map to influence STD for probabilistic '
                             'tracking (unsupported)')


def parse_args(argv=None):
    """ Generate a tractogram from a trained model, on a dataset subject. """
    parser = argparse.ArgumentParser(description=parse_args.__doc__,
                                     formatter_class=RawTextHelpFormatter)
    add_experiment_args(parser)
    add_model_args(parser)
    add_reward_args(parser)
    add_valid_args(parser)
    add_tractometer_args(parser)
    add_oracle_args(parser)
    add_tracking_args(parser)
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    print(args)
    return TrackToLearnValidation(vars(args)).run()


if __name__ == '__main__':
    main()
