#!/usr/bin/env python3
"""ttl_track.py -- generate a tractogram from a trained agent on the MI355X.

Same command line as TrackToLearn/runners/ttl_track.py:
    ttl_track.py in_odf in_seed in_mask out_tractogram [--input_wm]
        [--sh_basis B] [--compress T] [-f] [--save_seeds] [--agent DIR]
        [--hyperparameters JSON] [--n_actor N] [--npv N] [--min_length m]
        [--max_length M] [--noise s] [--fa_map F]
        [--binary_stopping_threshold t] [--rng_seed S]

Launch with ``torchrun --nproc-per-node R`` to shard every seed batch over R
GPUs (volumes replicated, RCCL gather-to-root of the finished tracts, rank 0
writes the file).
"""
import argparse
import json
import os
import random
from argparse import RawTextHelpFormatter
from os.path import join

import numpy as np
import torch

from tracktolearn_amd.algorithms.sac_auto import SACAuto
from tracktolearn_amd.environments.noisy_tracking_env import \
    NoisyTrackingEnvironment
from tracktolearn_amd.io import nifti
from tracktolearn_amd.io import streamlines as sio
from tracktolearn_amd.tracking.tracker import Tracker, detect_format
from tracktolearn_amd.utils.torch_utils import get_device

_ROOT = os.sep.join(os.path.normpath(
    os.path.dirname(os.path.abspath(__file__))).split(os.sep)[:-2])
DEFAULT_MODEL = os.path.join(_ROOT, 'models')


def per_rank_noise_rng(rng_seed, rank):
    """Generator of the exploration noise of one rank of a sharded run.  The
    ranks must share ``rng_seed`` for seed generation and shuffling (every
    rank derives the same seed list and takes its slice of each batch), so
    the noise gets its own stream per rank; with one rank the env keeps the
    reference's single stream."""
    return np.random.RandomState((int(rng_seed) + 1 + int(rank)) % (2 ** 32))


#: track_dto keys copied onto the experiment as they are (argparse names)
_PASS_THROUGH = ('in_odf', 'in_seed', 'in_mask', 'out_tractogram', 'noise',
                 'binary_stopping_threshold', 'n_actor', 'npv', 'min_length',
                 'max_length', 'sh_basis', 'save_seeds', 'agent', 'hyperparameters')

#: hyperparameters.json key -> attribute (the file a training run writes,
#: trainers/train.py; the shipped one is models/hyperparameters.json)
_HYPER = {'algorithm': 'algorithm', 'max_angle': 'theta', 'hidden_dims': 'hidden_dims',
          'n_dirs': 'n_dirs', 'target_sh_order': 'target_sh_order'}


class TrackToLearnTrack(object):
    """Tracking experiment: the noisy env built from files, a trained SACAuto
    policy, `Tracker.track`, a .trk / .tck file (runners/ttl_track.py:38-186 of
    the reference: same track_dto keys, same order of the side effects on the
    random generators)."""

    def __init__(self, track_dto):
        for key in _PASS_THROUGH:
            setattr(self, key, track_dto[key])
        self.input_wm = track_dto.get('input_wm', False)
        self.reference_file = self.in_mask
        self.compress = track_dto['compress'] or 0.0
        # tracking never computes rewards and never consults the oracle
        self.compute_reward, self.alignment_weighting = False, 0.0
        self.oracle_checkpoint, self.oracle_bonus = None, 0.0
        self.oracle_stopping_criterion = False
        self.fa_map = None
        self.device = torch.device('cuda', torch.cuda.current_device()) \
            if torch.cuda.is_available() else get_device()
        with open(self.hyperparameters, 'r') as json_file:
            hyper = json.load(json_file)
        for key, attr in _HYPER.items():
            setattr(self, attr, hyper[key])
        self.step_size = float(hyper['step_size'])
        self.voxel_size = hyper.get('voxel_size', 2.0)
        self.random_seed = track_dto['rng_seed']
        torch.manual_seed(self.random_seed)
        np.random.seed(self.random_seed)
        random.seed(self.random_seed)
        self.rng = np.random.RandomState(seed=self.random_seed)

    def get_tracking_env(self):
        """The noisy env over the input files (experiment.py:177-204)."""
        env_dto = {key: getattr(self, key) for key in (
            'fa_map', 'n_dirs', 'theta', 'min_length', 'max_length', 'noise', 'npv',
            'rng', 'alignment_weighting', 'oracle_bonus', 'oracle_stopping_criterion',
            'oracle_checkpoint', 'binary_stopping_threshold', 'compute_reward',
            'device', 'target_sh_order', 'in_odf', 'in_seed', 'in_mask', 'sh_basis',
            'input_wm')}
        env_dto.update(dataset_file=None, scoring_data=None, step_size=self.step_size,
                       reference=self.in_odf)
        return NoisyTrackingEnvironment.from_files(env_dto)

    def _step_for_subject(self, subject_voxel_size):
        """Keep the number of voxels traversed per step of the training: an agent
        trained at another voxel size steps proportionally (ttl_track.py:145-157)."""
        trained = float(self.voxel_size)
        if abs(float(subject_voxel_size) - trained) < 0.1:
            return self.step_size
        step_size_mm = float(subject_voxel_size) / trained * self.step_size
        print('Agent was trained on a voxel size of {}mm and a step size '
              'of {}mm.'.format(self.voxel_size, self.step_size))
        print('Subject has a voxel size of {}mm, setting step size to '
              '{}mm.'.format(subject_voxel_size, step_size_mm))
        return step_size_mm

    def _load_policy(self, env):
        input_size = env.reset(0, 1).shape[1]
        print('Tracking with {} agent.'.format(self.algorithm))
        alg = {'SACAuto': SACAuto}[self.algorithm](
            input_size, env.get_action_size(), self.hidden_dims, n_actors=self.n_actor,
            rng=self.rng, device=self.device, replay_size=1)
        alg.agent.load(self.agent, 'last_model_state')
        return alg

    def run(self):
        ref_img = nifti.load(self.reference_file)
        env = self.get_tracking_env()
        env.step_size_mm = self._step_for_subject(ref_img.get_zooms()[0])
        alg = self._load_policy(env)
        tracker = Tracker(alg, self.n_actor, compress=self.compress,
                          min_length=self.min_length, max_length=self.max_length,
                          save_seeds=self.save_seeds)
        # re-derives the step in voxels, the step counts and the neighbourhood
        # radius from the rescaled step (environments/env.py:196-212)
        env.load_subject()
        if tracker.group_size > 1:
            env.noise_rng = per_rank_noise_rng(self.random_seed, tracker.rank)
        tractogram = tracker.track(env, detect_format(self.out_tractogram))
        if tracker.rank != 0:
            for _ in tractogram:        # take part in the collectives only
                pass
            return
        header = sio.create_tractogram_header(
            ref_img.affine, ref_img.shape[:3], ref_img.get_zooms()[:3])
        n = sio.save(tractogram, self.out_tractogram, header=header)
        print('Saved {} streamlines to {}'.format(n, self.out_tractogram))


def add_mandatory_options_tracking(p):
    p.add_argument('in_odf',
                   help='File containing the orientation diffusion function \n'
                        'as spherical harmonics file (.nii.gz). Ex: ODF or '
                        'fODF.')
    p.add_argument('in_seed', help='Seeding mask (.nii.gz).')
    p.add_argument('in_mask', help='Tracking mask (.nii.gz).\nTracking will '
                                   'stop outside this mask.')
    p.add_argument('out_tractogram',
                   help='Tractogram output file (must be .trk or .tck).')
    p.add_argument('--input_wm', action='store_true',
                   help='If set, append the WM mask to the input signal.')


def add_out_options(p):
    out_g = p.add_argument_group('Output options')
    out_g.add_argument('--compress', type=float, metavar='thresh',
                       help='If set, will compress streamlines. The parameter '
                            'value is the \ndistance threshold.')
    out_g.add_argument('-f', dest='overwrite', action='store_true',
                       help='Force overwriting of the output files.')
    out_g.add_argument('--save_seeds', action='store_true',
                       help='If set, save the seeds used for the tracking \n '
                            'in the data_per_streamline property.')
    return out_g


def add_track_args(parser):
    add_mandatory_options_tracking(parser)
    basis_group = parser.add_argument_group('Basis options')
    basis_group.add_argument('--sh_basis', default='descoteaux07',
                             choices=['descoteaux07', 'tournier07'],
                             help='Spherical harmonics basis used for the SH '
                                  'coefficients. [%(default)s]')
    add_out_options(parser)
    agent_group = parser.add_argument_group('Tracking agent options')
    agent_group.add_argument('--agent', type=str,
                             help='Path to the folder containing .pth files.\n'
                                  '[{}]'.format(DEFAULT_MODEL))
    agent_group.add_argument('--hyperparameters', type=str,
                             help='Path to the .json file containing the '
                                  'hyperparameters of your tracking agent.')
    agent_group.add_argument('--n_actor', type=int, default=10000, metavar='N',
                             help='Number of streamlines to track simultaneous'
                                  'ly. [%(default)s]')
    seed_group = parser.add_argument_group('Seeding options')
    seed_group.add_argument('--npv', type=int, default=1,
                            help='Number of seeds per voxel [%(default)s].')
    track_g = parser.add_argument_group('Tracking options')
    track_g.add_argument('--min_length', type=float, default=10., metavar='m',
                         help='Minimum length of a streamline in mm. '
                              '[%(default)s]')
    track_g.add_argument('--max_length', type=float, default=300., metavar='M',
                         help='Maximum length of a streamline in mm. '
                              '[%(default)s]')
    track_g.add_argument('--noise', default=0.0, type=float, metavar='sigma',
                         help='Add noise ~ N (0, `noise`) to the agent\'s\n'
                              'output to make tracking more probabilistic.'
                              '[%(default)s]')
    track_g.add_argument('--fa_map', type=str, default=None,
                         help='Scale the added noise according to an FA map '
                              '(unsupported, see noisy_tracking_env.py).')
    track_g.add_argument('--binary_stopping_threshold', type=float, default=0.1,
                         help='Lower limit for interpolation of tracking mask '
                              'value.\nTracking will stop below this '
                              'threshold.')
    parser.add_argument('--rng_seed', default=1337, type=int,
                        help='Random number generator seed [%(default)s].')


def verify_agent_option(parser, args):
    if (args.agent is not None and args.hyperparameters is None) or \
       (args.agent is None and args.hyperparameters is not None):
        parser.error('You must specify both --agent and --hyperparameters '
                     'arguments or use the default model.')
    if args.agent is None:
        args.agent = DEFAULT_MODEL
        args.hyperparameters = join(DEFAULT_MODEL, 'hyperparameters.json')


def parse_args(argv=None):
    """ Generate a tractogram from a trained model. """
    parser = argparse.ArgumentParser(description=parse_args.__doc__,
                                     formatter_class=RawTextHelpFormatter)
    add_track_args(parser)
    args = parser.parse_args(argv)
    for f in (args.in_odf, args.in_seed, args.in_mask):
        if not os.path.isfile(f):
            parser.error('Input file {} does not exist.'.format(f))
    if os.path.isfile(args.out_tractogram) and not args.overwrite:
        parser.error('Output file {} exists. Use -f to force overwriting.'
                     .format(args.out_tractogram))
    if detect_format(args.out_tractogram) is None:
        parser.error('Invalid output streamline file format (must be trk or '
                     'tck): {0}'.format(args.out_tractogram))
    if args.min_length < 0 or args.max_length < args.min_length:
        parser.error('min_length must be >= 0 and <= max_length.')
    if args.compress is not None and not 0.001 <= args.compress <= 1:
        parser.error('The compression threshold must be in [0.001, 1] mm.')
    verify_agent_option(parser, args)
    return args


def main(argv=None):
    """ Main tracking script """
    import torch.distributed as dist
    args = parse_args(argv)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not dist.is_initialized():
        # one process per GPU over RCCL.  Rehearsal on a single-GPU box:
        # TTL_ONE_DEVICE=1 keeps every rank on cuda:0 and TTL_DIST_BACKEND=gloo
        # replaces RCCL, which refuses two ranks on one device.
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if os.environ.get('TTL_ONE_DEVICE') == '1':
            local_rank = 0
        torch.cuda.set_device(local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('TTL_DIST_BACKEND', 'nccl'))
    experiment = TrackToLearnTrack(vars(args))
    experiment.run()


if __name__ == '__main__':
    main()
