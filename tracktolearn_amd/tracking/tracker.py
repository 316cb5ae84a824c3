"""Tracker: batches seeds through the environment and the agent.

Mirror of TrackToLearn/tracking/tracker.py (track / track_and_train /
track_and_validate).  The per-streamline Python work of the reference's
generator (length filter, optional compression, voxel->file space) is kept for
the yielded items, but the length filter itself runs on the GPU over the whole
batch so that rejected streamlines are never downloaded.
"""
from collections import defaultdict

import numpy as np
import torch
from tqdm import tqdm

from tracktolearn_amd.algorithms.shared.utils import add_to_means
from tracktolearn_amd.tractogram import (LazyTractogram, TractogramItem,
                                         compress_streamline)


class TrkFile:
    """Format tag for '.trk' (stands in for nibabel.streamlines.TrkFile at
    tracker.py:127)."""
    EXT = '.trk'


class TckFile:
    """Format tag for '.tck'."""
    EXT = '.tck'


def detect_format(filename):
    """nibabel.streamlines.detect_format by extension."""
    lower = str(filename).lower()
    if lower.endswith('.trk'):
        return TrkFile
    if lower.endswith('.tck'):
        return TckFile
    return None


def to_file_space(streamline, tracts_format, affine, vox_size):
    """Voxel space -> the space the file format expects, as the reference
    writes it (tracker.py:127-136; pinned by tests/golden/tracker_*.npz):
    .trk: voxmm with corner origin, ``(s + 0.5) * vox_size`` evaluated IN
    PLACE on the float32 points (the reference edits the env's history view:
    the sum rounds to float32, the product is taken in float64 and rounded to
    float32); .tck: world space as ``s @ A[:3, :3] + A[:3, 3]`` in float64 --
    a row vector times the matrix, i.e. A's transpose applied, as upstream."""
    if tracts_format is TrkFile:
        out = np.array(streamline, dtype=np.float32)       # a copy
        out += 0.5
        out *= vox_size
        return out
    return np.dot(streamline, affine[:3, :3]) + affine[:3, 3]


class Tracker(object):
    """Generates streamlines with an agent, with or without training it
    (tracker.py:19-60)."""

    def __init__(self, alg, n_actor, prob=0., compress=0.0, min_length=20,
                 max_length=200, save_seeds=False):
        self.alg = alg
        self.n_actor = n_actor
        self.prob = prob
        self.compress = compress
        self.min_length = min_length
        self.max_length = max_length
        self.save_seeds = save_seeds
        #: one process per GPU: every rank tracks its contiguous shard of each
        #: seed batch and rank 0 yields the collated streamlines
        self.rank, self.group_size = 0, 1
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.rank, self.group_size = dist.get_rank(), dist.get_world_size()

    # ------------------------------------------------------------------ #
    def _batch_arrays(self, env, scaled_min, scaled_max):
        """Streamlines of the finished batch whose arc length (voxels) is in
        [scaled_min, scaled_max]: the filter of tracker.py:120-121 evaluated
        on the device.  Returns (packed points (M, 3) f32, kept lengths (k,)
        i64, seeds (k, 3) f64) as device tensors."""
        n = env._n_total
        from tracktolearn_amd.parallel import kept_lengths, pack_points
        if n == 0:              # an empty shard of a sharded batch
            dev = env.device
            return (torch.zeros((0, 3), dtype=torch.float32, device=dev),
                    torch.zeros(0, dtype=torch.int64, device=dev),
                    torch.zeros((0, 3), dtype=torch.float64, device=dev))
        keep_len = kept_lengths(env._buf_lengths[:n], env._buf_flags[:n])
        hist = env._buf_streamlines[:n]
        seg = (hist[:, 1:] - hist[:, :-1]).double()
        seg_len = torch.sqrt((seg ** 2).sum(dim=2))
        steps = torch.arange(seg_len.shape[1], device=hist.device)
        valid = steps[None, :] < (keep_len - 1)[:, None]
        arc = (seg_len * valid).sum(dim=1)
        ok = (arc >= scaled_min) & (arc <= scaled_max)
        sel = torch.nonzero(ok).squeeze(1)
        keep_sel = keep_len[sel]
        points = pack_points(hist[sel], keep_sel)
        seeds = torch.from_numpy(np.ascontiguousarray(
            env.initial_points, dtype=np.float64)).to(hist.device)[sel]
        return points, keep_sel, seeds

    def _batch_items(self, env, scaled_min, scaled_max, transform=None):
        """(streamline, seed) pairs of the finished batch on this process;
        with a process group, every rank's pairs, on rank 0 only.
        ``transform`` (packed (M, 3) points -> packed points) is applied to the
        whole batch before it is cut into streamlines."""
        points, keep_sel, seeds = self._batch_arrays(env, scaled_min, scaled_max)
        if self.group_size > 1:
            # gather-to-root of exact sizes: only rank 0 consumes the tracts
            from tracktolearn_amd.parallel import gather_ragged_to_root
            points, _ = gather_ragged_to_root(points)
            keep_sel, _ = gather_ragged_to_root(keep_sel)
            seeds, _ = gather_ragged_to_root(seeds)
            if self.rank != 0:
                return
        points = points.cpu().numpy()
        keep_np = keep_sel.cpu().numpy()
        seeds = seeds.cpu().numpy()
        if transform is not None:
            # the whole batch at once: the same element-wise arithmetic as per
            # streamline, a few numpy calls instead of a few per streamline
            points = transform(points)
        offsets = np.concatenate(([0], np.cumsum(keep_np)))
        for k in range(len(keep_np)):
            yield points[offsets[k]:offsets[k + 1]], seeds[k]

    def track(self, env, tracts_format):
        """Tracking only; a lazy tractogram whose iteration does the work
        (tracker.py:62-150).  Streamlines come out in the space the format
        expects: voxmm with corner origin for .trk ((s + 0.5) * voxel size),
        world space for .tck (``s @ A[:3,:3] + A[:3,3]`` as the reference
        writes it)."""
        batch_size = self.n_actor
        self.alg.agent.eval()
        affine = env.affine_vox2rasmm
        # shuffle so that partial displays of huge tractograms look uniform
        np.random.shuffle(env.seeds)

        def tracking_generator():
            vox_size = np.mean(np.abs(affine)[np.diag_indices(4)][:3])
            scaled_min_length = self.min_length / vox_size
            scaled_max_length = self.max_length / vox_size
            compress_th_vox = self.compress / vox_size
            for start in tqdm(range(0, len(env.seeds), batch_size),
                              disable=self.rank != 0):
                end = min(start + batch_size, len(env.seeds))
                if self.group_size > 1:
                    from tracktolearn_amd.parallel import shard_bounds
                    lo, hi = shard_bounds(end - start, self.rank, self.group_size)
                    start, end = start + lo, start + hi
                if end > start:
                    state = env.reset(start, end)
                    self.alg.validation_episode(state, env, self.prob)
                else:           # an empty shard still joins the collectives
                    env._n_total = 0
                # .trk without compression (ttl_track's default): the file-space
                # conversion is element-wise, so it runs once over the packed
                # batch -- bit for bit what the per-streamline call gives
                whole_batch = tracts_format is TrkFile and not self.compress
                for streamline, seed in self._batch_items(
                        env, scaled_min_length, scaled_max_length,
                        transform=(lambda p: to_file_space(p, TrkFile, affine, vox_size))
                        if whole_batch else None):
                    if self.compress:
                        streamline = compress_streamline(
                            streamline, compress_th_vox)
                    if not whole_batch:
                        streamline = to_file_space(streamline, tracts_format,
                                                   affine, vox_size)
                    seed_dict = {}
                    if self.save_seeds:
                        seed_dict = {'seeds': seed - 0.5}
                    yield TractogramItem(streamline, seed_dict, {})

        tractogram = LazyTractogram.from_data_func(tracking_generator)
        tractogram.affine_to_rasmm = affine
        return tractogram

    def track_and_train(self, env):
        """One training "epoch": n_actor random seeds tracked while learning
        (tracker.py:152-202)."""
        self.alg.agent.train()
        mean_losses = defaultdict(list)
        mean_reward_factors = defaultdict(list)
        state = env.nreset(self.n_actor)
        reward, losses, length, reward_factors = self.alg._episode(state, env)
        train_tractogram = env.get_streamlines()
        if len(losses.keys()) > 0:
            mean_losses = add_to_means(mean_losses, losses)
        if len(reward_factors.keys()) > 0:
            mean_reward_factors = add_to_means(mean_reward_factors,
                                               reward_factors)
        return train_tractogram, mean_losses, reward, mean_reward_factors

    def track_and_validate(self, env):
        """Track every seed without training, still summing the reward
        (tracker.py:204-259)."""
        self.alg.agent.eval()
        tractogram = None
        cummulative_reward = 0
        for start in tqdm(range(0, len(env.seeds), self.n_actor)):
            end = min(start + self.n_actor, len(env.seeds))
            state = env.reset(start, end)
            reward = self.alg.validation_episode(state, env, self.prob)
            batch = env.get_streamlines()
            if tractogram is None and len(batch) > 0:
                tractogram = batch
            elif len(batch) > 0:
                tractogram += batch
            cummulative_reward += reward
        return tractogram, cummulative_reward
