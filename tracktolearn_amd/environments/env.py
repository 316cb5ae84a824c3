"""BaseEnv: per-subject setup of the MI355X tractography environment.

Host-side mirror of TrackToLearn/environments/env.py (``BaseEnv``).  It keeps
the constructor signature ``(subject_data, split_id, env_dto)``, the env_dto
keys of TrackToLearn/experiment/experiment.py:107-129, ``load_subject()``,
``get_state_size/get_action_size/get_voxel_size/get_target_sh_order`` and the
attributes the tracker and trainers read (``seeds``, ``affine_vox2rasmm``,
``reference``, ``tracking_mask``, ``max_nb_steps``, ``step_size`` ...).

All per-step arithmetic lives in the HIP library (libttl_hip.so, C ABI in
include/ttl_hip.h).  There is no CPU path: a non-CUDA device or a missing
library raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from tracktolearn_amd import _lib
from tracktolearn_amd.datasets.utils import convert_length_mm2vox
from tracktolearn_amd.environments.stopping_criteria import (
    curvature_dot_threshold, mask_spline_coefficients)


def random_seeds_from_mask(mask, seeds_count=1, rng=None):
    """``seeds_count`` seeds per non-zero voxel, uniformly jittered inside the
    voxel, in voxel space with voxel centres at integer coordinates.

    Stand-in for ``dipy.tracking.utils.random_seeds_from_mask(mask, np.eye(4),
    seeds_count=npv)`` called at TrackToLearn/environments/env.py:216-219
    (dipy is absent here; seeds are *inputs* of the step path, so only their
    distribution matters -- SURVEY 8c, "seeds": parity unpinned).
    """
    rng = rng if rng is not None else np.random.RandomState()
    vox = np.argwhere(np.asarray(mask) > 0)
    vox = np.repeat(vox, int(seeds_count), axis=0)
    return vox + rng.uniform(size=vox.shape) - 0.5


try:                                    # fast path: no Stream object per call
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                  # pragma: no cover
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


class BaseEnv(object):
    """Abstract tracking environment (see TrackingEnvironment)."""

    #: direction arithmetic forced to float64 by subclasses that add float64
    #: noise to the action (NoisyTrackingEnvironment)
    _force_f64_directions = False

    def __init__(self, subject_data, split_id: str, env_dto: dict):
        if type(subject_data) is str:
            # env.py:85-94: a dataset file; subjects are visited in a fresh
            # random order every pass (DataLoader(batch_size=1, shuffle=True))
            from tracktolearn_amd.datasets.SubjectDataset import SubjectDataset
            self.dataset_file = subject_data
            self.split = split_id
            self.dataset = SubjectDataset(self.dataset_file, self.split)
            self._subject_order = iter(())
        else:
            self.subject_data = subject_data
            self.split = split_id

        self.normalize_obs = False
        self.obs_rms = None
        self._state_size = None

        # env.py:105-138
        self.n_dirs = env_dto['n_dirs']
        self.theta = env_dto['theta']
        self.npv = env_dto['npv']
        self.binary_stopping_threshold = env_dto['binary_stopping_threshold']
        self.step_size_mm = env_dto['step_size']
        self.min_length_mm = env_dto['min_length']
        self.max_length_mm = env_dto['max_length']
        self.oracle_checkpoint = env_dto.get('oracle_checkpoint')
        self.oracle_stopping_criterion = env_dto.get(
            'oracle_stopping_criterion', False)
        self.scoring_data = env_dto.get('scoring_data')
        self.compute_reward = env_dto['compute_reward']
        self.alignment_weighting = env_dto['alignment_weighting']
        self.oracle_bonus = env_dto.get('oracle_bonus', 0)
        self.rng = env_dto['rng']
        self.device = torch.device(env_dto['device'])
        self.target_sh_order = env_dto.get('target_sh_order')
        #: offset added to coordinates before the SH gather (0: voxel i sits at
        #: coordinate i); exposed because the third-party
        #: interpolation it replaces could not be inspected offline
        self.sh_coord_shift = float(env_dto.get('sh_coord_shift', 0.0))

        self._device_index = self.device.index if self.device.index is not None \
            else (torch.cuda.current_device() if self.device.type == 'cuda' else 0)
        if self.device.type != 'cuda':
            raise RuntimeError(
                'tracktolearn_amd environments run on an MI355X only '
                f"(env_dto['device']={self.device}); there is no CPU fallback")
        #: keep the reference's tail-batch quirk of OracleSingleton.predict
        #: unless told otherwise
        self.oracle_drop_tail = bool(env_dto.get('oracle_drop_tail', True))
        self._use_oracle_reward = bool(
            self.compute_reward and self.oracle_bonus and self.oracle_bonus > 0)
        self._use_oracle_stopping = bool(
            self.oracle_checkpoint and self.oracle_stopping_criterion)
        if self._use_oracle_reward and not self.oracle_checkpoint:
            # the reference only fails later, with `w * None`, once a
            # streamline longer than min_nb_steps is done
            raise ValueError(
                'oracle_bonus > 0 needs an oracle_checkpoint; use '
                "oracle_bonus=0 / --oracle_bonus 0 to train without the oracle")

        self._lib = _lib.load()
        self._handle = None
        self._n_max = 0
        # episode bookkeeping of the concrete envs (nothing tracked yet)
        self._pending = None
        self._n_active = 0
        self._n_total = 0
        self._cur = 0
        self.length = 0
        self.not_stopping = None
        self.load_subject()

    # ------------------------------------------------------------------ #
    def load_subject(self):
        """Per-subject setup, TrackToLearn/environments/env.py:143-281."""
        if hasattr(self, 'dataset_file'):
            if hasattr(self, 'subject_id') and len(self.dataset) == 1:
                return
            try:
                index = next(self._subject_order)
            except StopIteration:
                self._subject_order = iter(
                    torch.randperm(len(self.dataset)).tolist())
                index = next(self._subject_order)
            (self.subject_id, input_volume, tracking_mask, seeding_mask,
             peaks, reference) = self.dataset[index]
        else:
            # the runners call load_subject() again right after the constructor
            # (ttl_track.py:178, train.py:261): the same volumes are already
            # packed on the device; only what derives from step_size_mm /
            # min/max length is recomputed when the caller changed them in
            # between (ttl_track.py:146 rescales the step to the subject's
            # voxel size, env.py:196-212 then re-derives the step in voxels)
            if getattr(self, '_loaded_subject', None) is self.subject_data:
                if self._tracking_params_key() != self._loaded_params_key:
                    self._derive_tracking_params()
                    self._destroy_handle()
                    self._n_max = 0
                return
            (input_volume, tracking_mask, seeding_mask, peaks,
             reference) = self.subject_data
            self._loaded_subject = self.subject_data

        self.affine_vox2rasmm = input_volume.affine_vox2rasmm
        self.affine_rasmm2vox = np.linalg.inv(self.affine_vox2rasmm)
        self.reference = reference

        sh = np.ascontiguousarray(input_volume.data, dtype=np.float32)
        if sh.ndim != 4:
            raise ValueError('input volume must be (X, Y, Z, C)')
        self.data_volume = torch.from_numpy(sh).to(self.device)
        if self.target_sh_order is None:
            # even, symmetric basis: C = (n+1)(n+2)/2
            n_coefs = sh.shape[-1]
            order = int(round((-3 + np.sqrt(1 + 8 * n_coefs)) / 2))
            self.target_sh_order = order

        self.tracking_mask = tracking_mask
        self.peaks = peaks
        mask_data = tracking_mask.data.astype(np.uint8)
        self.seeding_data = seeding_mask.data.astype(np.uint8)

        self._derive_tracking_params()

        self.seeds = random_seeds_from_mask(
            self.seeding_data, seeds_count=self.npv, rng=self.rng)

        # --- device-side volumes ------------------------------------- #
        C_ = sh.shape[-1]
        pitch = (C_ + 3) // 4 * 4
        self._sh_dim = tuple(int(d) for d in sh.shape[:3])
        # record order of the packed volume: 4x4x4-voxel bricks keep a
        # streamline's neighbourhood in a few contiguous runs (ttl_hip.h)
        self._sh_layout = _lib.SH_BRICK4 \
            if os.environ.get('TTL_SH_LAYOUT', 'brick4') == 'brick4' else _lib.SH_LINEAR
        dims = (C.c_int32 * 3)(*self._sh_dim)
        n_rec = int(self._lib.ttl_sh_volume_records(dims, self._sh_layout))
        self._sh_packed, self._sh_memory = self._place_sh_volume(
            dims, C_, pitch, n_rec, mask_data)
        stream = self._stream()
        self._n_coef, self._coef_pitch = C_, pitch

        # cubic B-spline coefficients: scipy on the host at load time, as
        # BinaryStoppingCriterion.__init__ does (stopping_criteria.py:58-59)
        coef = mask_spline_coefficients(mask_data)
        self._mask_coef = torch.from_numpy(coef).to(self.device)
        self._mask_dim = tuple(int(d) for d in coef.shape)
        # per-cell shortcut of the mask test (decisions unchanged)
        self._mask_cls = None
        if os.environ.get('TTL_MASK_CLASSES', '1') != '0':
            self._mask_cls = torch.empty(self._mask_dim, dtype=torch.uint8,
                                         device=self.device)
            dims = (C.c_int32 * 3)(*self._mask_dim)
            _lib.check(self._lib.ttl_mask_classes(
                self._mask_coef.data_ptr(), dims,
                float(self.binary_stopping_threshold),
                self._mask_cls.data_ptr(), stream), 'ttl_mask_classes')

        self._peaks_dev = None
        self._peaks_dim = (0, 0, 0)
        if self.compute_reward:
            pk = np.ascontiguousarray(peaks.data, dtype=np.float32)
            if pk.ndim != 4 or pk.shape[-1] != 15:
                raise ValueError('peaks volume must be (X, Y, Z, 15)')
            self._peaks_dev = torch.from_numpy(pk).to(self.device)
            self._peaks_dim = tuple(int(d) for d in pk.shape[:3])

        self._curv_dot_max, self._curv_enabled = \
            curvature_dot_threshold(self.theta)

        # --- oracle (reward.py / oracle_reward.py / stopping_criteria.py) - #
        self._oracle = None
        if self._use_oracle_reward or self._use_oracle_stopping:
            from tracktolearn_amd.oracles.oracle import OracleSingleton
            self._oracle = OracleSingleton(self.oracle_checkpoint, self.device)
            self._oracle.drop_tail = self.oracle_drop_tail
            # voxel space of the tracked volume -> voxel space of the
            # reference anatomy (oracle_reward.py:82-90: vox -> rasmm -> the
            # reference's vox, corner origin; the shift drops out of the
            # segment vectors the oracle consumes)
            ref_affine = None
            if isinstance(reference, dict):
                ref_affine = reference.get('affine')
            elif hasattr(reference, 'affine'):
                ref_affine = reference.affine
            lin = np.eye(3)
            if ref_affine is not None:
                lin = np.linalg.inv(np.asarray(ref_affine, np.float64))[:3, :3] @ \
                    np.asarray(self.affine_vox2rasmm, np.float64)[:3, :3]
            self._oracle_lin = None if np.allclose(lin, np.eye(3)) else \
                torch.from_numpy(lin.T.astype(np.float32)).to(self.device)
            #: the same matrix as host floats, row-major (ttl_oracle_segments)
            self._oracle_lin_host = None if self._oracle_lin is None else \
                [float(v) for v in lin.T.astype(np.float32).ravel()]

        self._destroy_handle()
        self._n_max = 0
        self._state_width = 7 * C_ + 3 * int(self.n_dirs)
        #: floats between consecutive state rows (= the width: contiguous rows)
        self._state_pitch = self._state_width

    # ------------------------------------------------------------------ #
    def _tracking_params_key(self):
        return (float(self.step_size_mm), float(self.min_length_mm),
                float(self.max_length_mm))

    #: placements tried at the first large reset: this many allocations of the
    #: packed SH volume x this many allocations of the state ring
    #: (TTL_VOLUME_CANDIDATES, TTL_STATE_RING_CANDIDATES; 1 and 0 = keep the first
    #: allocation of the volume and fresh state tensors per step)
    VOLUME_CANDIDATES = 3
    STATE_RING_CANDIDATES = 16
    #: device memory the search may hold on top of the env's own (the extra
    #: copies of the volume + the candidate state buffers): at most this much and
    #: at most PLACEMENT_SEARCH_FREE_FRACTION of what hipMemGetInfo reports free
    PLACEMENT_SEARCH_BYTES = 8 << 30
    PLACEMENT_SEARCH_FREE_FRACTION = 0.10
    #: the search stops after its first three pairs when their gather times agree
    #: within this fraction (a box without placement classes: nothing to find)
    PLACEMENT_EARLY_EXIT = 0.02
    #: volumes below this sit in the caches wherever they are
    VOLUME_TUNE_MIN_BYTES = 64 << 20
    #: batches below this are bound by launches, not by the gather
    VOLUME_TUNE_MIN_ROWS = 65536

    def _place_sh_volume(self, dims, n_coef, pitch, n_rec, mask_data):
        """Pack the SH volume into device memory of its own (``ttl_volume_alloc``,
        not the caching allocator: ``_tune_placement`` may want to re-roll where
        it lands).  Returns ``(tensor (n_rec, pitch) float32, owner of its
        memory)``."""
        mem = _lib.DeviceVolume(self._device_index, n_rec * pitch * 4, False)
        vol = torch.as_tensor(mem, device=self.device).view(torch.float32).view(n_rec, pitch)
        _lib.check(self._lib.ttl_pack_sh_volume(
            self.data_volume.data_ptr(), vol.data_ptr(), dims, n_coef, pitch,
            self._sh_layout, self._stream()), 'ttl_pack_sh_volume')
        self._sh_tuned = None          # the matrix of candidate times once tuned
        self._placement_search = None  # what the search did (see _tune_placement)
        self._state_ring, self._state_ring_pos, self._state_ring_memory = None, 0, None
        return vol, mem

    def _tune_placement(self, seeds):
        """Where the gather's source (the 170 MB packed SH volume of the bench)
        and its destination (the state rows) land in device memory moves the
        kernel between 0.18 and 0.20 ms on the same GPU -- as a PAIR: with five
        allocations of each, volumes fall into two classes and so do the
        allocations of the rows, one combination is the fastest (0.1805 ms), the
        same volume with the other class of rows the slowest (0.197), the other
        volumes sit in between with either (0.187-0.189)
        (``benchmarks/placement_probe13.py``; the counters show the same
        requests at a longer memory-side latency; address, alignment,
        contiguity, which XCD reads what: none of it predicts the class --
        DESIGN.md 3.3).  So the host measures, once per subject, at the first
        reset of at least VOLUME_TUNE_MIN_ROWS streamlines: the volume is copied
        into VOLUME_CANDIDATES allocations, STATE_RING_CANDIDATES state buffers
        (for the device-resident loop) are allocated one by one, every pair
        (volume, buffer) runs four steps of the real loop on up to
        262 144 of the given seeds with the scripted policy, and the pair with
        the fastest gather wins (51 pairs, ~0.3 s; the same bytes at other
        addresses: no result changes): its volume is kept, and the STATE_RING
        buffers that were fastest with that volume form the ring.  The caching
        allocator's own blocks (no ring: a fresh state tensor per step) are the
        17th candidate for the rows when the probe runs at the batch's own size.

        Bounded (round 3): the extra volume copies and the candidate buffers
        together take at most PLACEMENT_SEARCH_BYTES and at most
        PLACEMENT_SEARCH_FREE_FRACTION of the device memory that is free when
        the search starts (fewer candidates otherwise, none when even
        STATE_RING buffers do not fit); when the first three pairs agree within
        PLACEMENT_EARLY_EXIT the box has no placement classes worth a search
        and it stops there.  ``_placement_search`` records what was done
        (seconds, bytes held, pairs timed, early exit)."""
        import time
        t_search = time.perf_counter()
        kv = int(os.environ.get('TTL_VOLUME_CANDIDATES', self.VOLUME_CANDIDATES))
        kr = int(os.environ.get('TTL_STATE_RING_CANDIDATES', self.STATE_RING_CANDIDATES))
        ring_len = int(os.environ.get('TTL_STATE_RING', self.STATE_RING))
        vol0, mem0 = self._sh_packed, self._sh_memory
        nbytes = vol0.numel() * 4
        radius = float(np.float32(self.add_neighborhood_vox or 0.0))
        self._sh_tuned = []
        if kv < 2 or nbytes < self.VOLUME_TUNE_MIN_BYTES or not 0.0 < radius < 1.0 \
                or self._use_oracle_stopping or self._use_oracle_reward:
            return
        if ring_len < 2:
            kr = 0
        n_ring = len(seeds)                     # the batch that triggered the tuning
        n = min(n_ring, 262144)
        buf_bytes = n_ring * self._state_pitch * 4
        # memory budget of the search, from what the device reports free now
        try:
            free_bytes = int(torch.cuda.mem_get_info(self._device_index)[0])
        except Exception:           # pragma: no cover
            free_bytes = 0
        budget = int(min(self.PLACEMENT_SEARCH_BYTES,
                         self.PLACEMENT_SEARCH_FREE_FRACTION * free_bytes))
        budget = int(os.environ.get('TTL_PLACEMENT_SEARCH_BYTES', budget))
        kv = max(1, min(kv, 1 + budget // max(2 * nbytes, 1)))   # copies: half of it at most
        left = budget - (kv - 1) * nbytes
        if kr > 0:
            kr = min(kr, left // max(buf_bytes, 1))
            if kr < ring_len:
                kr = 0
        self._placement_search = dict(budget_bytes=budget, free_bytes_at_start=free_bytes,
                                      volume_candidates=kv, state_buffer_candidates=kr,
                                      pairs_timed=0, early_exit=False, seconds=0.0)
        if kv < 2 and kr == 0:
            return
        keep = dict(initial_points=getattr(self, 'initial_points', None),
                    noise=getattr(self, 'noise', None))
        if keep['noise'] is not None:
            self.noise = 0.0           # the probe steps must not draw from the env's generator
        vols, rings, candidates = [(vol0, mem0)], [], [(None, None)]
        best = None
        try:
            for c in range(1, kv):
                try:
                    mem = _lib.DeviceVolume(self._device_index, nbytes, False)
                except _lib.TTLError:
                    break               # no room for another copy
                vol = torch.as_tensor(mem, device=self.device).view(torch.float32) \
                    .view(vol0.shape)
                vol.copy_(vol0)
                vols.append((vol, mem))
            pitch, width = self._state_pitch, self._state_width
            for c in range(kr):     # state buffers, one allocation each
                try:
                    mem = _lib.DeviceVolume(self._device_index, buf_bytes, False)
                except _lib.TTLError:
                    break
                flat = torch.as_tensor(mem, device=self.device).view(torch.float32)
                rings.append(([flat.view(n_ring, pitch)[:, :width]], mem))
            if len(rings) < ring_len:
                rings = []
            # the caching allocator's own blocks (fresh state tensors per step, as
            # without a ring) compete too when the probe runs at the batch's own
            # size: a step then gets the very blocks it will get later
            candidates = rings + ([(None, None)] if n == n_ring or not rings else [])
            self._placement_search['bytes_held'] = (len(vols) - 1) * nbytes + \
                len(rings) * buf_bytes
            timed, settled = [], False
            for vi, (vol, mem) in enumerate(vols):
                if settled:
                    break
                self._sh_packed, self._sh_memory = vol, mem
                self._destroy_handle()
                self._n_max = 0
                row = []
                for ri, (ring, rmem) in enumerate(candidates):
                    # a candidate is a "ring" of one buffer that every step of the
                    # probe must write, referenced or not
                    self._state_ring_search = True
                    self._state_ring, self._state_ring_pos = ring, 0
                    state = self._start(seeds[:n])
                    self.profile_begin(16, classes=('state',))
                    for step in range(4):
                        if not self._n_active:
                            break
                        self.step_device(self.scripted_actions(state, step, 1, 0.05))
                        state, _ = self.harvest()
                    total_ms, launches = self.profile_end()['state']
                    ms = total_ms / max(launches, 1)
                    row.append(round(ms, 5))
                    if best is None or ms < best[0]:
                        best = (ms, vi, ri)
                    timed.append(ms)
                    if len(timed) == 3 and min(timed) > 0 and \
                            max(timed) - min(timed) <= self.PLACEMENT_EARLY_EXIT * min(timed) \
                            and os.environ.get('TTL_PLACEMENT_EARLY_EXIT', '1') != '0':
                        settled = True      # no classes on this box: nothing to find
                        break
                self._sh_tuned.append(row)
            self._placement_search.update(pairs_timed=len(timed), early_exit=settled)
        finally:
            self._state_ring_search = False
            vi, ri = (best[1], best[2]) if best else (0, 0)
            self._sh_packed, self._sh_memory = vols[vi]
            self._state_ring, self._state_ring_memory = None, None
            if best and candidates[ri][0] is not None:
                # the ring: the ring_len buffers that were fastest with this volume
                # (buffers the search never timed -- early exit -- rank last, in
                # allocation order)
                row = self._sh_tuned[vi]
                order = sorted(range(len(rings)),
                               key=lambda r: (row[r] if r < len(row) else float('inf'), r))
                picked = [rings[r] for r in order[:ring_len]]
                self._state_ring = [bufs[0] for bufs, _ in picked]
                self._state_ring_memory = [m for _, m in picked]
            self._state_ring_pos = 0
            self._destroy_handle()
            self._n_max = 0
            self.initial_points = keep['initial_points']
            if keep['noise'] is not None:
                self.noise = keep['noise']
            del vols, rings, candidates          # the losers go back to the driver
            if getattr(self, '_placement_search', None) is not None:
                self._placement_search['seconds'] = time.perf_counter() - t_search

    def _derive_tracking_params(self):
        """env.py:196-213: step size in voxels, step counts and the
        neighbourhood radius from step_size_mm / min / max length.  Also
        picks the direction arithmetic (depends on the step's numpy type)."""
        self.step_size = convert_length_mm2vox(
            self.step_size_mm, self.affine_vox2rasmm)
        self.min_length = self.min_length_mm
        self.max_length = self.max_length_mm
        self.max_nb_steps = int(self.max_length / self.step_size_mm)
        self.min_nb_steps = int(self.min_length / self.step_size_mm)
        self.add_neighborhood_vox = convert_length_mm2vox(
            self.step_size_mm, self.affine_vox2rasmm)
        r = np.float32(self.add_neighborhood_vox)
        eye = torch.eye(3)
        self.neighborhood_directions = torch.cat(
            (torch.zeros((1, 3)), eye * float(r), -eye * float(r))
        ).to(self.device)
        # float32 vs float64 direction arithmetic: whatever this host's numpy
        # gives `float32_array * step_size`
        promoted = (np.zeros(1, np.float32) * self.step_size).dtype
        if self._force_f64_directions:
            self._mode = _lib.MODE_F64DIR
        elif promoted == np.float64:
            self._mode = _lib.MODE_F32NORM
        else:
            self._mode = _lib.MODE_F32
        self._loaded_params_key = self._tracking_params_key()

    def _new_state(self, n):
        """(n, state width) float32 rows on the device, `_state_pitch` apart."""
        if self._state_pitch == self._state_width:
            return torch.empty((n, self._state_width), dtype=torch.float32,
                               device=self.device)
        return torch.empty((n, self._state_pitch), dtype=torch.float32,
                           device=self.device)[:, :self._state_width]

    #: state rows of large batches live in memory whose placement was measured: a
    #: pool of this many buffers (TTL_STATE_RING=0: a fresh ``torch.empty`` per
    #: step).  A buffer is handed out again only when no tensor refers to it any
    #: more (the use count of its storage), so every state tensor stays intact
    #: for as long as the caller holds it, exactly like a fresh allocation; when
    #: the caller holds them all, the step falls back to a fresh allocation.
    STATE_RING = 4
    #: TTL_STATE_RING_ROTATE=1 (and a torch without the storage use count): the
    #: round-2 behaviour, a plain ring -- a tensor is overwritten STATE_RING steps
    #: after it was handed out, referenced or not, and ``step()`` allocates
    STATE_RING_ROTATE = os.environ.get('TTL_STATE_RING_ROTATE', '0') == '1' or \
        not hasattr(torch._C, '_storage_Use_Count')

    @property
    def state_ring_len(self):
        """Number of placed state buffers the env rotates through (0: none, every
        state tensor is a fresh allocation).  With ``state_ring_rotates`` False
        (the default) this is an implementation detail: buffers are only reused
        once nothing refers to them.  With it True, a tensor returned by
        ``reset`` / ``step_device`` / ``harvest`` is overwritten after
        ``state_ring_len - 1`` further state tensors were handed out: clone what
        must live longer."""
        return len(self._state_ring) if getattr(self, '_state_ring', None) else 0

    @property
    def state_ring_rotates(self):
        return bool(self.state_ring_len) and (self.STATE_RING_ROTATE or
                                              getattr(self, '_state_ring_search', False))

    @staticmethod
    def _storage_users(t):
        # TensorImpls + Python storage objects alive on this storage (the probe's
        # own temporary included, in the baseline as well)
        return torch._C._storage_Use_Count(t.untyped_storage()._cdata)

    def _ring_state(self, n):
        """State rows for a reset / step of a large batch: a placed buffer that
        nothing refers to any more when there is one that fits, a fresh tensor
        otherwise."""
        ring = self._state_ring
        if ring is None or n > ring[0].shape[0]:
            return self._new_state(n)
        if self.STATE_RING_ROTATE or getattr(self, '_state_ring_search', False):
            self._state_ring_pos = (self._state_ring_pos + 1) % len(ring)
            return ring[self._state_ring_pos][:n]
        if getattr(self, '_state_ring_idle_of', None) is not ring:
            # first use of this pool: nothing has been handed out yet
            self._state_ring_idle = [self._storage_users(b) for b in ring]
            self._state_ring_idle_of = ring
        for k in range(1, len(ring) + 1):
            pos = (self._state_ring_pos + k) % len(ring)
            if self._storage_users(ring[pos]) <= self._state_ring_idle[pos]:
                self._state_ring_pos = pos
                return ring[pos][:n]
        return self._new_state(n)       # the caller still holds a tensor on every buffer

    def _stream(self):
        """Raw HIP stream torch currently launches on for this device (the
        private C getter costs ~0.3 us against ~2 us for the Stream object)."""
        return C.c_void_p(_raw_stream(self._device_index))

    def _destroy_handle(self):
        # graphs captured over the handle's buffers die with it
        self._free_runs = {}
        self._free_bufs = {}
        if getattr(self, '_handle', None):
            self._lib.ttl_env_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    def _ensure_capacity(self, n):
        """(Re)allocate the per-streamline buffers for ``n`` streamlines and
        create the library handle over them (all buffers are torch tensors the
        library borrows)."""
        if self._handle and n <= self._n_max:
            return
        self._destroy_handle()
        dev = self.device
        n_max = int(n)
        self._buf_streamlines = torch.empty(
            (n_max, self.max_nb_steps + 1, 3), dtype=torch.float32, device=dev)
        self._buf_flags = torch.zeros(n_max, dtype=torch.int32, device=dev)
        self._buf_lengths = torch.zeros(n_max, dtype=torch.int32, device=dev)
        self._buf_dones = torch.zeros(n_max, dtype=torch.uint8, device=dev)
        self._buf_idx = torch.zeros((2, n_max), dtype=torch.int32, device=dev)
        ws = int(self._lib.ttl_env_workspace_bytes(n_max))
        self._buf_ws = torch.zeros(ws + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (self._buf_ws.data_ptr() + 255) // 256 * 256

        d = _lib.EnvDesc()
        d.abi_version = _lib.ABI_VERSION
        d.mode = self._mode
        d.sh_dim[:] = self._sh_dim
        d.n_coef = self._n_coef
        d.coef_pitch = self._coef_pitch
        d.sh_packed = self._sh_packed.data_ptr()
        d.sh_coord_shift = self.sh_coord_shift
        d.sh_layout = self._sh_layout
        d.mask_dim[:] = self._mask_dim
        d.mask_coef = self._mask_coef.data_ptr()
        d.mask_threshold = float(self.binary_stopping_threshold)
        d.mask_classes = self._mask_cls.data_ptr() \
            if self._mask_cls is not None else None
        d.peaks_dim[:] = self._peaks_dim
        d.peaks = self._peaks_dev.data_ptr() if self._peaks_dev is not None else None
        d.compute_reward = 1 if self.compute_reward else 0
        d.alignment_weighting = float(self.alignment_weighting)
        d.n_dirs = int(self.n_dirs)
        d.max_nb_steps = int(self.max_nb_steps)
        d.step_size_vox = float(self.step_size)
        d.neigh_radius_vox = float(np.float32(self.add_neighborhood_vox))
        d.curvature_enabled = 1 if self._curv_enabled else 0
        d.curv_dot_max = float(self._curv_dot_max)
        d.n_max = n_max
        d.streamlines = self._buf_streamlines.data_ptr()
        d.flags = self._buf_flags.data_ptr()
        d.lengths = self._buf_lengths.data_ptr()
        d.dones = self._buf_dones.data_ptr()
        d.idx_a = self._buf_idx[0].data_ptr()
        d.idx_b = self._buf_idx[1].data_ptr()
        d.workspace = ws_ptr
        d.workspace_bytes = ws
        handle = C.c_void_p()
        _lib.check(self._lib.ttl_env_create(C.byref(d), C.byref(handle)),
                   'ttl_env_create')
        self._handle = handle
        self._n_max = n_max
        self._buf_idx_rows = (self._buf_idx[0], self._buf_idx[1])
        # int32 view of the library's active-row -> state-row map
        idx_p, dest_p, length = C.c_void_p(), C.c_void_p(), C.c_int32()
        _lib.check(self._lib.ttl_env_view(handle, C.byref(idx_p), C.byref(dest_p),
                                          C.byref(length)), 'ttl_env_view')
        off = dest_p.value - self._buf_ws.data_ptr()
        self._row_dest_all = self._buf_ws[off:off + 4 * n_max].view(torch.int32)
        self._host_counts = torch.zeros(4, dtype=torch.int32).pin_memory()
        self._host_counts_np = self._host_counts.numpy()
        # staging of the reference's host contract (step(numpy) -> host dones /
        # reward), allocated at the first such step (TrackingEnvironment._host_io)
        self._io = None
        self._n_continue_out = C.c_int32()

    # ------------------------------------------------------------------ #
    @classmethod
    def from_dataset(cls, env_dto: dict, split: str):
        """env.py:284-309."""
        return cls(env_dto['dataset_file'], split, env_dto)

    @classmethod
    def from_files(cls, env_dto: dict):
        """Environment from NIfTI files, for tracking with a trained agent
        (env.py:311-347)."""
        (input_volume, peaks_volume, tracking_mask, seeding_mask) = \
            BaseEnv._load_files(
                env_dto['in_odf'], env_dto['in_seed'], env_dto['in_mask'],
                env_dto['sh_basis'], env_dto['target_sh_order'],
                need_peaks=bool(env_dto.get('compute_reward')))
        subj_files = (input_volume, tracking_mask, seeding_mask, peaks_volume,
                      env_dto.get('reference'))
        return cls(subj_files, 'testing', env_dto)

    @classmethod
    def _load_files(cls, signal_file, in_seed, in_mask, sh_basis,
                    target_sh_order=6, need_peaks=False):
        """env.py:350-449.  The reference always extracts fODF peaks here, but
        only the alignment reward consumes them and ``ttl_track.py`` runs with
        ``compute_reward=False``, so they are computed on demand
        (``need_peaks``), on the GPU (tracktolearn_amd/reconst/peaks.py; own
        sphere -> parity unpinned)."""
        from tracktolearn_amd.datasets.utils import (MRIDataVolume,
                                                     set_sh_order_basis)
        from tracktolearn_amd.io import nifti
        signal = nifti.load(signal_file)
        zooms = signal.get_zooms()[:3]
        if not np.allclose(np.mean(zooms), zooms[0], atol=1e-03):
            print('WARNING: ODF SH file is not isotropic. Tracking cannot be '
                  'ran robustly. You are entering undefined behavior '
                  'territory.')
        data = set_sh_order_basis(signal.get_fdata(dtype=np.float32), sh_basis,
                                  target_order=target_sh_order,
                                  target_basis='descoteaux07')
        seeding = nifti.load(in_seed)
        tracking = nifti.load(in_mask)
        signal_volume = MRIDataVolume(data, signal.affine)
        peaks_volume = None
        if need_peaks:
            from tracktolearn_amd.reconst.peaks import peaks_from_sh
            from tracktolearn_amd.utils.torch_utils import get_device
            sh_dev = torch.from_numpy(np.ascontiguousarray(
                data, dtype=np.float32)).to(get_device())
            peaks_volume = MRIDataVolume(
                peaks_from_sh(sh_dev).cpu().numpy(), signal.affine)
        seeding_volume = MRIDataVolume(seeding.get_fdata(), seeding.affine)
        tracking_volume = MRIDataVolume(tracking.get_fdata(), tracking.affine)
        return (signal_volume, peaks_volume, tracking_volume, seeding_volume)

    def set_step_size(self, step_size_mm):
        """Change the step size (mm) of a loaded subject and re-derive what
        depends on it (step in voxels, step counts, neighbourhood radius).
        ``ttl_track_from_hdf5.py:134`` of the reference calls this name;
        ``ttl_track.py:146`` assigns ``step_size_mm`` and reloads instead."""
        self.step_size_mm = step_size_mm
        if self._tracking_params_key() != self._loaded_params_key:
            self._derive_tracking_params()
            self._destroy_handle()
            self._n_max = 0

    def get_state_size(self):
        """env.py:451-463."""
        example_state = self.reset(0, 1)
        self._state_size = example_state.shape[1]
        return self._state_size

    def get_action_size(self):
        return 3

    def get_target_sh_order(self):
        return self.target_sh_order

    def get_voxel_size(self):
        """env.py:478-491."""
        diag = np.diagonal(self.affine_vox2rasmm)[:3]
        return np.mean(np.abs(diag))
