"""TrackingEnvironment: reset / nreset / step / harvest / get_streamlines on
the MI355X.

Host-side mirror of TrackToLearn/environments/tracking_env.py.  Method names,
arguments and return shapes follow the reference; the arithmetic runs in
libttl_hip.so:

  reference call                      C ABI entry point (include/ttl_hip.h)
  reset / nreset   (:91-133/:47-89)   ttl_env_reset
  step             (:135-221)         ttl_env_step
  harvest          (:223-245)         ttl_env_harvest
  get_streamlines  (:247-294)         device-side ragged pack (torch) + D2H

Two flavours of the step loop are offered:

  * ``step()`` / ``harvest()``: the reference's contract.  ``state`` is a
    float32 ``torch.Tensor`` on the GPU; ``reward`` (float64) and ``dones``
    (bool) are host numpy arrays as in the reference, which costs one small
    device->host copy + stream sync per step.
  * ``step_device()``: everything stays on the device (torch tensors), state
    rows are written survivors-first so that ``harvest()`` returns a view and
    copies nothing.  ``RLAlgorithm.validation_episode``-style loops that only
    use the harvested state get identical results from either flavour.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from tracktolearn_amd import _lib
from tracktolearn_amd.environments.env import BaseEnv
from tracktolearn_amd.environments.stopping_criteria import StoppingFlags
from tracktolearn_amd.tractogram import Tractogram


class _StepInfo(dict):
    """``info`` dict of ``step``; 'continue_idx' is downloaded on first use
    (the reference puts the host index array there, tracking_env.py:220)."""

    def __init__(self, env, n_active, reward_info):
        super().__init__(reward_info=reward_info)
        self._env, self._n = env, n_active
        self._idx_dev = env._idx_view(n_active)

    def __missing__(self, key):
        if key == 'continue_idx':
            val = self._idx_dev.to('cpu').numpy().astype(np.int64)
            self[key] = val
            return val
        raise KeyError(key)

    def __contains__(self, key):
        return key == 'continue_idx' or super().__contains__(key)


class _LazyStateRows(torch.Tensor):
    """``step()``'s state rows in the reference's row order (one row per active
    streamline, continue_idx order), gathered on first use.

    The step writes its state rows survivors first, so that ``harvest()`` --
    whose result the policy needs at once -- is a view and copies nothing.  The
    reference's own tracking loop never looks at the rows ``step()`` returns
    (rl.py:95-101 overwrites them with ``harvest()``'s); its training loop does
    (ddpg.py: the replay buffer's ``next_state``).  This tensor serves both: it
    has the shape, dtype and device of the real thing and becomes
    ``rows[row_dest]`` (one gather, what ``harvest()`` used to pay as a copy)
    the first time anything touches its values: torch operations (through
    ``__torch_dispatch__``) and the accessors that bypass the dispatcher --
    ``numpy()``, ``tolist()``, ``__array__``, ``data_ptr()``, storages, dlpack,
    ``__cuda_array_interface__`` -- alike; conversions that would be no-ops on
    a real tensor (``float()``, ``contiguous()``, ``to(same device)``,
    ``detach()``) return the gathered PLAIN tensor.  ``materialize()`` is the
    explicit form."""

    @staticmethod
    def __new__(cls, rows, row_dest):
        t = torch.Tensor._make_wrapper_subclass(
            cls, (int(row_dest.shape[0]), int(rows.shape[1])), dtype=rows.dtype,
            device=rows.device)
        t._rows, t._row_dest, t._value = rows, row_dest, None
        return t

    def materialize(self):
        """The gathered rows as a plain ``torch.Tensor`` (cached)."""
        if self._value is None:
            self._value = self._rows.index_select(0, self._row_dest.long())
            self._rows = self._row_dest = None
        return self._value

    def __repr__(self):
        return f'_LazyStateRows(shape={tuple(self.shape)}, materialized={self._value is not None})'

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map

        def real(x):
            return x.materialize() if isinstance(x, _LazyStateRows) else x
        return func(*tree_map(real, args), **tree_map(real, kwargs or {}))

    @property
    def data(self):
        return self.materialize().data

    @property
    def __cuda_array_interface__(self):
        return self.materialize().__cuda_array_interface__


def _forward_to_gathered(name):
    def method(self, *args, **kwargs):
        return getattr(self.materialize(), name)(*args, **kwargs)
    method.__name__ = name
    method.__doc__ = f'``torch.Tensor.{name}`` of the gathered rows.'
    return method


# accessors implemented below the dispatcher (a wrapper subclass has no storage
# of its own: data_ptr() would be 0, numpy() / storage access would raise) and
# conversions that short-circuit to ``self`` before dispatching
for _name in ('numpy', 'tolist', '__array__', 'data_ptr', 'untyped_storage', 'storage',
              '_typed_storage', '__dlpack__', '__dlpack_device__', 'contiguous', 'float',
              'double', 'half', 'bfloat16', 'to', 'type', 'cuda', 'cpu', 'detach', 'clone',
              'pin_memory', 'record_stream', 'share_memory_', 'is_pinned', 'item',
              'storage_offset', 'is_shared', '__reduce_ex__', '__deepcopy__'):
    setattr(_LazyStateRows, _name, _forward_to_gathered(_name))
del _name


class _FreeRun:
    """The captured graph of one free-running step (policy + env launches) and
    the fixed buffers it works on."""

    def __init__(self, env, n, policy, record_actions, owner=None):
        dev = env.device
        # the captured graph holds raw pointers to the policy's weights: keep
        # the owner alive and remember which storages were captured, so that a
        # replaced network (or another agent recycled at the same id()) is
        # noticed instead of replayed on freed memory
        self.owner = owner
        self.fingerprint = _policy_fingerprint(owner)
        self.state = env._new_state(n)
        self.state.zero_()
        self.done = torch.empty(n, dtype=torch.uint8, device=dev)
        self.reward = self.reward_sum = None
        if env.compute_reward:
            self.reward = torch.empty(n, dtype=torch.float64, device=dev)
            self.reward_sum = torch.zeros((), dtype=torch.float64, device=dev)
        self.actions_log = self.step_no = None
        if record_actions:
            self.actions_log = torch.zeros((env.max_nb_steps + 2, n, 3),
                                           dtype=torch.float32, device=dev)
            self.step_no = torch.zeros(1, dtype=torch.int64, device=dev)
        # the policy's lazy initialisations (GEMM heuristics, workspaces) must
        # not happen under capture
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side), torch.no_grad():
            policy(self.state)
            policy(self.state)
            t0.record(side)
            for _ in range(3):
                policy(self.state)
            t1.record(side)
        cur.wait_stream(side)
        t1.synchronize()
        #: GPU + launch time of one policy evaluation on all n rows, microseconds
        self.policy_us = t0.elapsed_time(t1) / 3 * 1e3
        self.graph = None

    def capture(self, env, policy):
        """Between ttl_env_freerun_begin and _end: record policy + step."""
        n, record_actions = self.state.shape[0], self.actions_log is not None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            with torch.no_grad():
                a = policy(self.state)
            a = a.to(torch.float32).contiguous()
            if a.shape != (n, 3):
                raise ValueError(f'policy returned {tuple(a.shape)}, expected ({n}, 3)')
            if record_actions:
                self.actions_log.index_copy_(0, self.step_no, a[None])
                self.step_no += 1
            _lib.check(env._lib.ttl_env_freerun_step(
                env._handle, a.data_ptr(), n, self.state.data_ptr(), env._state_pitch,
                self.reward.data_ptr() if self.reward is not None else None,
                self.done.data_ptr(), env._stream()), 'ttl_env_freerun_step')
            if self.reward is not None:
                self.reward_sum += self.reward.sum()
            self.actions = a
        self.graph = graph          # only a capture that went through is replayed


def _policy_fingerprint(owner):
    """Storage addresses of the parameters a captured policy reads (the owner's
    own and its ``actor``'s when those are torch modules)."""
    if owner is None:
        return None
    mods = []
    for m in (owner, getattr(owner, 'actor', None), getattr(owner, 'agent', None)):
        if isinstance(m, torch.nn.Module) and all(m is not k for k in mods):
            mods.append(m)
    return tuple(p.data_ptr() for m in mods for p in m.parameters())


class TrackingEnvironment(BaseEnv):
    """Tracking environment; also the "tracker" (see the reference class)."""

    # ------------------------------------------------------------------ #
    def _idx_view(self, n):
        """Device view (int32) of the current continue_idx."""
        return self._buf_idx_rows[self._cur][:n]

    def _start(self, initial_points):
        n = int(initial_points.shape[0])
        if n < 1:
            raise ValueError('need at least one seed')
        if n >= self.VOLUME_TUNE_MIN_ROWS and self._sh_tuned is None:
            self._tune_placement(initial_points)
        self.initial_points = initial_points
        self._ensure_capacity(n)
        seeds32 = torch.from_numpy(
            np.ascontiguousarray(initial_points, dtype=np.float32)
        ).to(self.device)
        state = self._ring_state(n)
        # batches of SPATIAL_ORDER_MIN rows and more are gathered in a
        # spatially sorted processing order (built by the library)
        sort_rows = n >= self.SPATIAL_ORDER_MIN and getattr(self, 'spatial_order', True)
        _lib.check(self._lib.ttl_env_reset(
            self._handle, seeds32.data_ptr(), n,
            _lib.ORDER_BY_POSITION if sort_rows else None, state.data_ptr(),
            self._state_pitch, self._stream()), 'ttl_env_reset')
        self._n_total = n
        self._n_active = n
        self._order_slots = n if sort_rows else 0
        self._cur = 0
        self.length = 1
        self._pending = None
        self.not_stopping = None
        return state

    #: batches at least this large get a spatially sorted processing order
    #: (TTL_FUSE_MAX_ROWS, clamped as ttl_env_create clamps it: [256, 65536])
    SPATIAL_ORDER_MIN = min(max(int(os.environ.get('TTL_FUSE_MAX_ROWS', '16384')), 256), 65536)

    #: the order is rebuilt from the current positions every this many steps
    #: (a streamline crosses an 8-voxel brick in about ten 0.75-voxel steps;
    #: the gather costs 0.72 ns per streamline in fresh order, 1.15 ns once
    #: the order has decayed; measured best of 2/4/8/16/32); 0 = never
    SPATIAL_ORDER_REFRESH = int(os.environ.get('TTL_ORDER_REFRESH', '16'))

    def _refresh_processing_order(self, force=False):
        """Every SPATIAL_ORDER_REFRESH steps: re-sort the active rows by where
        their streamlines are now (``ttl_env_refresh_processing_order``: a key
        kernel + a counting sort over the bricks inside the library, on
        workspace memory); in between, the library re-sorts every 256-slot
        block of the order by current voxel at each step."""
        every = self.SPATIAL_ORDER_REFRESH
        n = self._n_active
        if n < self.SPATIAL_ORDER_MIN or self._pending is not None or \
                not getattr(self, 'spatial_order', True):
            return
        # between refreshes the order keeps its length (the step's fused tail does
        # not compact it: stopped streamlines leave holes at the end of their
        # 256-slot block): refresh early once a fifth of it is holes
        slots = getattr(self, '_order_slots', 0)
        sparse = bool(every) and self.length > 1 and slots and n < self.ORDER_MIN_FILL * slots \
            and slots <= self.TAIL_FUSED_MAX_ROWS and self._order_keeps_holes()
        if not force and not sparse and \
                (not every or self.length <= 1 or (self.length - 1) % every):
            return
        _lib.check(self._lib.ttl_env_refresh_processing_order(
            self._handle, self._stream()), 'ttl_env_refresh_processing_order')
        self._order_slots = n

    def _order_keeps_holes(self):
        """Whether the library runs the one-launch tail on this handle (only then
        does the order keep holes between refreshes): the gather must read the
        per-slot records, i.e. the register-deduplicating kernel (TTL_STATE_KERNEL
        not 0 or 2) with a neighbourhood radius inside (0, 1) voxel -- the
        conditions of ``fused_tail`` in ttl_env_step."""
        return os.environ.get('TTL_STATE_KERNEL', '') not in ('0', '2') and \
            bool(self.add_neighborhood_vox) and \
            0.0 < float(np.float32(self.add_neighborhood_vox)) < 1.0

    #: refresh the processing order early when fewer than this share of its slots
    #: still hold a streamline
    ORDER_MIN_FILL = float(os.environ.get('TTL_ORDER_MIN_FILL', '0.8'))
    #: the library's fused tail (and with it the holes) is used for orders of at
    #: most this many slots (mirrors TTL_TAIL_FUSED / TTL_TAIL_FUSED_MAX_ROWS)
    TAIL_FUSED_MAX_ROWS = 0 if os.environ.get('TTL_TAIL_FUSED', '1') == '0' else \
        min(int(os.environ.get('TTL_TAIL_FUSED_MAX_ROWS', '262144')), 4096 * 256)

    def nreset(self, n_seeds: int):
        """N random seeds among all seeds (tracking_env.py:47-89; global
        numpy RNG, as the reference)."""
        replace = n_seeds > len(self.seeds)
        picks = np.random.choice(
            np.arange(len(self.seeds)), size=n_seeds, replace=replace)
        return self._start(self.seeds[picks])

    def reset(self, start: int, end: int):
        """A given batch of seeds (tracking_env.py:91-133)."""
        return self._start(self.seeds[start:end])

    # ------------------------------------------------------------------ #
    def _host_io(self):
        """Persistent buffers of the reference's host contract (``step(numpy
        actions)`` -> host ``reward`` / ``dones``): the action batch's device
        buffer, pinned host memory for the dones and the reward, an event."""
        if self._io is None:
            n = self._n_max
            io = dict(
                act_dev=torch.empty((n, 3), dtype=torch.float32, device=self.device),
                done_pin=torch.empty(n, dtype=torch.uint8).pin_memory(),
                reward_pin=torch.empty(n, dtype=torch.float64).pin_memory()
                if self.compute_reward else None,
                event=torch.cuda.Event())
            io['done_np'] = io['done_pin'].numpy()
            io['reward_np'] = io['reward_pin'].numpy() if io['reward_pin'] is not None else None
            self._io = io
        return self._io

    def _actions_to_device(self, actions):
        if isinstance(actions, torch.Tensor):
            a = actions
            if a.dtype is not torch.float32 or a.device != self.device:
                a = a.to(device=self.device, dtype=torch.float32)
        else:
            # a host array (the reference's contract, rl.py:93-94): straight from
            # the caller's pageable memory into a persistent device buffer.  The
            # runtime's own chunked staging overlaps the host copy with the DMA
            # (0.076 ms for 262 144 x 3 floats; through a pinned buffer of ours
            # 0.047 ms host copy + 0.073 ms DMA one after the other:
            # benchmarks/micro/h2d_probe.py)
            actions = np.ascontiguousarray(actions, dtype=np.float32)
            n = self._n_active
            if actions.shape != (n, 3):
                raise ValueError(
                    f'actions must be ({n}, 3), got {tuple(actions.shape)}')
            a = self._host_io()['act_dev'][:n]
            a.copy_(torch.from_numpy(actions))
        a = a.contiguous()
        if a.shape != (self._n_active, 3):
            raise ValueError(
                f'actions must be ({self._n_active}, 3), got {tuple(a.shape)}')
        return a

    def _noise_for(self, actions):
        """float64 noise added to the action before normalisation, or None.
        Only NoisyTrackingEnvironment returns something."""
        return None

    def _launch_step(self, actions, order, host_outputs=False):
        """``host_outputs``: the caller wants ``dones`` (and ``reward``) on the
        host (``step()``): they are final after the step's first kernel, so
        their copies into pinned memory are queued right behind it, in front of
        the index compaction and the state gather, and an event marks them --
        ``step()`` returns while the gather is still running."""
        if self._pending is not None:
            raise RuntimeError('harvest() the previous step first')
        n = self._n_active
        if n < 1:
            raise RuntimeError('no active streamline left; reset first')
        a = self._actions_to_device(actions)
        noise = self._noise_for(a)
        self._refresh_processing_order()
        # rows written survivors first land in a placed buffer that nothing refers
        # to any more (env.py:_ring_state; with the round-2 plain ring, step() keeps
        # allocating: its callers may hold states for any number of steps)
        pooled = order == _lib.ORDER_PARTITION and \
            not (host_outputs and self.state_ring_rotates)
        state = self._ring_state(n) if pooled else self._new_state(n)
        done = torch.empty(n, dtype=torch.uint8, device=self.device)
        reward = None
        if self.compute_reward:
            reward = torch.empty(n, dtype=torch.float64, device=self.device)
        noise_ptr = noise.data_ptr() if noise is not None else None
        reward_ptr = reward.data_ptr() if reward is not None else None
        early = host_outputs and not self._use_oracle_stopping and not self._use_oracle_reward
        if early:
            io = self._host_io()
            _lib.check(self._lib.ttl_env_step_begin(
                self._handle, a.data_ptr(), noise_ptr, n, reward_ptr,
                done.data_ptr(), self._stream()), 'ttl_env_step_begin')
            io['done_pin'][:n].copy_(done, non_blocking=True)
            if reward is not None:
                io['reward_pin'][:n].copy_(reward, non_blocking=True)
            io['event'].record()
            _lib.check(self._lib.ttl_env_step_end(
                self._handle, None, order, state.data_ptr(), self._state_pitch,
                self._host_counts.data_ptr(), self._stream()), 'ttl_env_step_end')
        elif not self._use_oracle_stopping:
            _lib.check(self._lib.ttl_env_step(
                self._handle, a.data_ptr(), noise_ptr, n, order,
                state.data_ptr(), self._state_pitch, reward_ptr,
                done.data_ptr(), self._host_counts.data_ptr(), self._stream()),
                'ttl_env_step')
        else:
            # the oracle criterion sits between the point update and the
            # compaction (criteria order LENGTH, CURVATURE, ORACLE, MASK;
            # env.py:233-260 -- the flags are OR-ed, so the order is moot)
            _lib.check(self._lib.ttl_env_step_begin(
                self._handle, a.data_ptr(), noise_ptr, n, reward_ptr,
                done.data_ptr(), self._stream()), 'ttl_env_step_begin')
            extra = self._oracle_stopping_flags(n, self.length + 1)
            _lib.check(self._lib.ttl_env_step_end(
                self._handle, extra.data_ptr() if extra is not None else None,
                order, state.data_ptr(), self._state_pitch,
                self._host_counts.data_ptr(), self._stream()),
                'ttl_env_step_end')
            self._keep_alive = extra
        self._last_oracle_term = None
        if self._use_oracle_reward:
            self._last_oracle_term = self._oracle_reward(
                n, self.length + 1, done, reward)
        self.length += 1
        self._pending = dict(order=order, state=state, n=n, done=done,
                             keep=(a, noise), early=early)
        return state, reward, done

    # -- oracle criterion / reward: torch ops between the library calls ----- #
    def _oracle_points(self, rows, n_points):
        """Histories (len(rows), n_points, 3) of the given active rows, in the
        reference anatomy's voxel space (oracle_reward.py:82-90)."""
        g = self._idx_view(self._n_active)[rows].long() if rows is not None \
            else self._idx_view(self._n_active).long()
        pts = self._buf_streamlines[g, :n_points]
        if self._oracle_lin is not None:
            pts = pts @ self._oracle_lin
        return pts

    def _oracle_stopping_flags(self, n, n_points):
        """OracleStoppingCriterion.__call__ (stopping_criteria.py:115-154) on
        every active streamline once they have more than 5 * min_nb_steps
        points: ORACLE bit where the score is < 0.5.  None before that."""
        if not n_points > self.min_nb_steps * 5:
            return None
        if self._oracle_fast():
            # the active rows' histories straight from the library's buffer
            scores, _ = self._oracle.predict_history(
                self._buf_streamlines, self._idx_view(n).data_ptr(), 1, n, n_points,
                self._oracle_lin_host)
        else:
            scores = self._oracle.predict(self._oracle_points(None, n_points))
        from tracktolearn_amd.environments.stopping_criteria import StoppingFlags
        return (scores < 0.5).to(torch.uint8) * \
            StoppingFlags.STOPPING_ORACLE.value

    def _oracle_reward(self, n, n_points, done, reward):
        """OracleReward.__call__ (oracle_reward.py:70-93): +oracle_bonus for the
        streamlines that stopped in this step with a score > 0.5, once the
        streamlines are longer than min_nb_steps.  Adds into ``reward`` and
        returns the oracle term (or None when nothing was scored)."""
        if not n_points > self.min_nb_steps:
            return None
        if self._oracle_fast():
            return self._oracle_reward_fast(n, n_points, reward)
        rows = torch.nonzero(done).squeeze(1)         # host sync: count of dones
        if rows.numel() == 0:
            return None
        scores = self._oracle.predict(self._oracle_points(rows, n_points))
        term = torch.zeros(n, dtype=torch.float64, device=self.device)
        term[rows] = (scores > 0.5).double() * float(self.oracle_bonus)
        reward += term
        return term

    #: the oracle path as library calls (``ttl_env_stopped`` ->
    #: ``ttl_oracle_segments`` -> the fused network -> ``ttl_oracle_bonus``: four
    #: launches and the wait for the 8-byte counts that ``harvest()`` needs
    #: anyway) instead of some twenty torch ops behind a ``nonzero`` sync;
    #: ``TTL_ORACLE_FAST=0`` or a network the fused kernel does not support keep
    #: the torch path
    oracle_fast = os.environ.get('TTL_ORACLE_FAST', '1') != '0'

    def _oracle_fast(self):
        return self.oracle_fast and getattr(self._oracle, 'net', None) is not None

    def _oracle_reward_fast(self, n, n_points, reward):
        """``_oracle_reward`` on the rows ``ttl_env_stopped`` lists (the rows
        ``nonzero(done)`` finds, in the same order)."""
        import ctypes as C
        lst, n_stop = C.c_void_p(), C.c_int32()
        _lib.check(self._lib.ttl_env_stopped(self._handle, C.byref(lst), C.byref(n_stop)),
                   'ttl_env_stopped')
        n_stop = n_stop.value
        if n_stop == 0:
            return None
        # ids: the odd words of {row, id} pairs
        scores, n_scored = self._oracle.predict_history(
            self._buf_streamlines, lst.value + 4, 2, n_stop, n_points, self._oracle_lin_host)
        term = torch.empty(n, dtype=torch.float64, device=self.device)
        _lib.check(self._lib.ttl_oracle_bonus(
            scores.data_ptr(), n_scored, lst.value, n_stop, float(self.oracle_bonus), n,
            term.data_ptr(), reward.data_ptr(), self._stream()), 'ttl_oracle_bonus')
        self._keep_alive_reward = scores
        return term

    def step(self, actions):
        """Apply actions, grow every active streamline by one step, test the
        stopping criteria, compute the reward and the new state
        (tracking_env.py:135-221).

        Returns ``(state, reward, dones, info)``: state rows for all currently
        active streamlines (incl. the ones that just stopped) as a float32 GPU
        tensor in continue_idx order; ``reward`` float64 numpy (zeros of size
        N_total when rewards are off, tracking_env.py:204); ``dones`` bool
        numpy; ``info`` = {'continue_idx', 'reward_info'}.
        """
        lazy = self.lazy_step_state
        state, reward, done = self._launch_step(
            actions, _lib.ORDER_PARTITION if lazy else _lib.ORDER_ACTIVE, host_outputs=True)
        n = self._pending['n']
        if lazy:
            # the rows were written survivors first (harvest() will be a view);
            # the reference's row order is one gather away, paid only by callers
            # that look (the map is copied: the library rewrites it every step)
            state = _LazyStateRows(state, self._row_dest_view(n).clone())
        if self._pending['early']:
            # dones / reward were copied to pinned memory right behind the
            # step's first kernel: wait for those copies only
            io = self._io
            io['event'].synchronize()
            dones = io['done_np'][:n].astype(bool)
            reward_np = io['reward_np'][:n].copy() if reward is not None else None
        else:
            dones = done.to('cpu').numpy().astype(bool)      # syncs the stream
            reward_np = reward.to('cpu').numpy() if reward is not None else None
        self._pending['dones_host'] = dones
        reward_info = {}
        if reward is not None:
            # reward.py:73-75: mean of each weighted factor
            oracle_np = np.zeros_like(reward_np)
            if self._last_oracle_term is not None:
                oracle_np = self._last_oracle_term.to('cpu').numpy()
            reward_info = {'peaks_reward': np.mean(reward_np - oracle_np),
                           'oracle_reward': np.mean(oracle_np)}
        else:
            # tracking_env.py:204: np.zeros(N_total) -- the same zeros every
            # step, so one array per batch size (never written by the callers,
            # which sum it: rl.py:97, ddpg.py:209)
            reward_np = self._zero_reward(self._n_total)
        info = _StepInfo(self, self._pending['n'], reward_info)
        return state, reward_np, dones, info

    def _zero_reward(self, n):
        z = getattr(self, '_zero_reward_np', None)
        if z is None or z.shape[0] != n:
            z = self._zero_reward_np = np.zeros(n)
            z.flags.writeable = False       # shared between steps: writing it raises
        return z

    #: ``step()`` returns its state rows as a ``_LazyStateRows`` (gathered into the
    #: reference's row order on first use) and ``harvest()`` copies nothing;
    #: False (``TTL_LAZY_STEP_STATE=0``): the rows are written in the reference's
    #: order and ``harvest()`` copies the survivors' rows
    lazy_step_state = os.environ.get('TTL_LAZY_STEP_STATE', '1') != '0'

    def step_device(self, actions):
        """Device-resident step: no host copy, no sync.  ``state`` rows are
        written survivors first (stable), then the streamlines that stopped in
        this step (stable); ``info['row_dest'][i]`` is the state row of active
        row ``i``.  ``reward`` (float64 or None) and ``dones`` (uint8) are GPU
        tensors in active-row order.  ``harvest()`` afterwards returns the
        leading rows of ``state`` without copying."""
        state, reward, done = self._launch_step(actions, _lib.ORDER_PARTITION)
        n = self._pending['n']
        info = {'row_dest': self._row_dest_view(n), 'reward_info': {}}
        return state, reward, done, info

    def _row_dest_view(self, n):
        """int32 view of the library's active-row -> state-row map."""
        return self._row_dest_all[:n]

    def harvest(self):
        """Drop the streamlines that stopped in the last step
        (tracking_env.py:223-245).  Returns ``(state, not_stopping)``: the
        state rows of the streamlines still being tracked, and the boolean
        mask over the previous active rows (host numpy, as the reference).
        After ``step_device()`` the mask is ``None``: the caller already holds
        the ``dones`` GPU tensor of that step (``not_stopping == (dones == 0)``).

        Only the 8-byte survivor count is waited for, not the step itself: the
        returned tensor is stream-ordered like any other torch result."""
        if self._pending is None:
            raise RuntimeError('no step to harvest')
        pend = self._pending
        n = pend['n']
        order = pend['order']
        state_in = pend['state']
        out = None
        if order == _lib.ORDER_ACTIVE:
            out = self._new_state(state_in.shape[0])
        # the survivor count left the GPU right after the stopping decisions
        # (written by the step's kernel into pinned memory): this wait does not
        # cover the state gather
        n_out = self._n_continue_out
        _lib.check(self._lib.ttl_env_harvest_wait(
            self._handle, state_in.data_ptr(),
            out.data_ptr() if out is not None else None, self._state_pitch,
            self._stream(), n_out), 'ttl_env_harvest')
        n_cont = n_out.value
        if order == _lib.ORDER_ACTIVE:
            new_state = out[:n_cont]
        else:
            new_state = state_in[:n_cont]
        if 'dones_host' in pend:
            self.not_stopping = np.logical_not(pend['dones_host'])
        else:
            self.not_stopping = None
        self._cur ^= 1
        self._n_active = n_cont
        self._pending = None
        return new_state, self.not_stopping

    # ------------------------------------------------------------------ #
    # free-running episode: policy + step in one HIP graph, no host in the loop
    #: largest batch the free-running step takes (the one-launch step tail;
    #: TTL_FUSE_MAX_ROWS clamped as ttl_env_create clamps it)
    FREERUN_MAX = min(max(int(os.environ.get('TTL_FUSE_MAX_ROWS', '16384')), 256), 65536)

    def _has_action_noise(self):
        return False

    def freerun_supported(self):
        """Whether ``run_free`` can take the episode from here: a batch of at
        most FREERUN_MAX rows between steps, no oracle criterion / bonus
        (torch code between the step's launches) and no Gaussian action noise
        (drawn on the host per step)."""
        return (self._pending is None and 1 <= self._n_active <= self.FREERUN_MAX
                and not self._use_oracle_stopping and not self._use_oracle_reward
                and not self._has_action_noise()
                and self.add_neighborhood_vox
                and 0.0 < float(np.float32(self.add_neighborhood_vox)) < 1.0)

    @staticmethod
    def _wait_for_gpu(counts, step_no, timeout_s=20.0):
        """Spin until the GPU has reported step ``step_no`` in the pinned words
        (free-running loops: keeps the host a bounded number of steps ahead).
        Bounded: a GPU that stopped answering surfaces as an error in
        ``ttl_env_freerun_end`` instead of a hang here."""
        if int(counts[2]) >= step_no:
            return
        t0 = time.perf_counter()
        spins = 0
        while int(counts[2]) < step_no:
            spins += 1
            if not spins & 0xfff and time.perf_counter() - t0 > timeout_s:
                raise RuntimeError('free-running step: no progress reported by the GPU '
                                   f'for {timeout_s:.0f} s')

    @staticmethod
    def _no_grad():
        # inference_mode skips autograd's view / version bookkeeping on top of
        # no_grad (measurement knob: TTL_INFERENCE_MODE=0)
        if os.environ.get('TTL_INFERENCE_MODE', '1') != '0':
            return torch.inference_mode()
        return torch.no_grad()

    def run_free_eager(self, policy, state, lookahead=2):
        """The same episode as ``run_free`` without a graph, for policies whose
        cost grows with the batch (the reference's default 1024-wide networks
        are bound by their GEMMs): the host launches policy + free-running step
        for ``cap`` rows, where ``cap`` is the newest survivor count the GPU has
        reported in pinned memory -- an upper bound of the rows active now, since
        counts only fall -- and never waits for a step to finish, only for the
        GPU not to fall more than ``lookahead`` steps behind (so that ``cap``
        follows the survivors closely).  The GPU runs back to back; the policy's
        batches shrink as in the step-by-step loop.  Returns ``(summed reward
        as a 0-d float64 tensor or None, number of steps)``."""
        if not self.freerun_supported():
            raise RuntimeError('run_free_eager: not available for this configuration')
        n = self._n_active
        dev = self.device
        buf = self._free_bufs.get(n) if hasattr(self, '_free_bufs') else None
        if buf is None:
            if not hasattr(self, '_free_bufs'):
                self._free_bufs = {}
            reward = torch.empty(n, dtype=torch.float64, device=dev) \
                if self.compute_reward else None
            if len(self._free_bufs) >= 4:       # batch sizes normally repeat
                self._free_bufs.clear()
            buf = self._free_bufs[n] = (self._new_state(n),
                                        torch.empty(n, dtype=torch.uint8, device=dev), reward)
        state_buf, done, reward = buf
        state_buf[:n].copy_(state)
        reward_sum = torch.zeros((), dtype=torch.float64, device=dev) \
            if reward is not None else None
        _lib.check(self._lib.ttl_env_freerun_begin(
            self._handle, self._host_counts.data_ptr(), self._stream()),
            'ttl_env_freerun_begin')
        try:
            counts = self._host_counts_np
            step_fn, handle, pitch = self._lib.ttl_env_freerun_step, self._handle, self._state_pitch
            reward_ptr = reward.data_ptr() if reward is not None else None
            state_ptr, done_ptr = state_buf.data_ptr(), done.data_ptr()
            limit = self.max_nb_steps + 2 - self.length
            steps, cap = 0, n
            while steps < limit:
                c = int(counts[0])
                if c == 0:
                    break
                cap = min(cap, c)
                with self._no_grad():
                    a = policy(state_buf[:cap])
                if a.dtype is not torch.float32 or not a.is_contiguous():
                    a = a.to(torch.float32).contiguous()
                _lib.check(step_fn(handle, a.data_ptr(), cap, state_ptr, pitch, reward_ptr,
                                   done_ptr, self._stream()), 'ttl_env_freerun_step')
                if reward is not None:
                    reward_sum += reward[:cap].sum()
                steps += 1
                self._wait_for_gpu(counts, steps - lookahead)
        finally:
            n_left, length, done_steps = C.c_int32(), C.c_int32(), C.c_int32()
            _lib.check(self._lib.ttl_env_freerun_end(
                self._handle, C.byref(n_left), C.byref(length), C.byref(done_steps),
                self._stream()), 'ttl_env_freerun_end')
        n_steps = length.value - self.length
        if n_steps & 1:
            self._cur ^= 1
        self.length = length.value
        self._n_active = n_left.value
        self._pending = None
        self.not_stopping = None
        return reward_sum, n_steps

    def run_free(self, policy, state, key=None, record_actions=False, max_policy_us=None,
                 owner=None):
        """Track the current batch to exhaustion without the host in the loop
        (what ``RLAlgorithm.validation_episode`` does with ``step_device`` /
        ``harvest``, rl.py:58-106): ``policy(state) -> actions`` and the step's
        launches (``ttl_env_freerun_step``) are captured once in a HIP graph
        over fixed ``(n, .)`` buffers and replayed until the pinned survivor
        count reads zero.  Row count, length and continue_idx parity advance in
        device memory; rows that have stopped keep stale state rows whose
        actions nothing reads.

        ``policy`` must be torch code that can run under stream capture (no
        host round trip) and treat rows independently.  ``key`` identifies it
        for the graph cache (default ``id(policy)``); ``owner`` (the agent whose
        networks the policy evaluates) is kept alive with the cached graph, and
        the graph is captured again when another owner turns up under the same
        key or its parameters have moved to other storage.  Returns ``(summed reward
        as a 0-d float64 tensor or None, number of steps)``; with
        ``record_actions`` also the ``(steps, n, 3)`` action batches.

        A replayed graph evaluates the policy on all ``n`` rows at every step,
        while the step-by-step loop's batches shrink as streamlines stop: the
        graph only wins while the loop is bound by launches.  With
        ``max_policy_us`` the policy is timed once on ``n`` rows (cached per
        key); if one evaluation takes longer, nothing is run and ``None`` is
        returned -- the caller keeps its step-by-step loop.
        """
        if not self.freerun_supported():
            raise RuntimeError('run_free: not available for this configuration')
        n = self._n_active
        key = (n, id(policy) if key is None else key, bool(record_actions))
        fr = self._free_runs.get(key)
        if fr is not None and (fr.owner is not owner or
                               fr.fingerprint != _policy_fingerprint(owner)):
            del self._free_runs[key]        # captured over other weights
            fr = None
        if fr is not None and max_policy_us is not None and fr.policy_us > max_policy_us:
            return None         # a replayed graph would lose to the shrinking batches
        rc = self._lib.ttl_env_freerun_begin(
            self._handle, self._host_counts.data_ptr(), self._stream())
        if rc == _lib.ERR_UNSUPPORTED:
            return None         # the library has no free-running path here: the caller
            #                     keeps its step-by-step loop (as for a slow policy)
        _lib.check(rc, 'ttl_env_freerun_begin')
        try:
            if fr is None:      # buffers + one timing of the policy on n rows (the policy
                # may itself read the free-running words: timed after begin)
                if len(self._free_runs) >= 8:       # keys normally repeat (batch size, agent)
                    self._free_runs.clear()
                fr = self._free_runs[key] = _FreeRun(self, n, policy, record_actions, owner)
                if max_policy_us is not None and fr.policy_us > max_policy_us:
                    return None
            if fr.graph is None:
                try:
                    fr.capture(self, policy)
                except BaseException:
                    self._free_runs.pop(key, None)
                    raise
            fr.state[:n].copy_(state)
            if fr.reward_sum is not None:
                fr.reward_sum.zero_()
            if fr.step_no is not None:
                fr.step_no.zero_()
            counts = self._host_counts_np
            limit = self.max_nb_steps + 2 - self.length
            steps = 0
            while steps < limit:
                fr.graph.replay()
                steps += 1
                if counts[0] == 0:
                    break
                # the host runs ahead of the GPU; keep the distance bounded so
                # that few empty steps are queued behind the last real one
                if counts[0] != 0:
                    self._wait_for_gpu(counts, steps - 24)
        finally:
            n_left, length, done = C.c_int32(), C.c_int32(), C.c_int32()
            _lib.check(self._lib.ttl_env_freerun_end(
                self._handle, C.byref(n_left), C.byref(length), C.byref(done),
                self._stream()), 'ttl_env_freerun_end')
        # the host view catches up with what the device did; steps replayed
        # after the last streamline stopped changed nothing
        n_steps = length.value - self.length
        if n_steps & 1:
            self._cur ^= 1
        self.length = length.value
        self._n_active = n_left.value
        self._pending = None
        self.not_stopping = None
        reward = fr.reward_sum.clone() if fr.reward_sum is not None else None
        if record_actions:
            return reward, n_steps, fr.actions_log[:n_steps].clone()
        return reward, n_steps

    def _compute_stopping_flags(self, streamlines, stopping_criteria=None):
        """Which of the given streamlines should stop, and why
        (TrackToLearn/environments/env.py:567-603) -- evaluated on the GPU for
        arbitrary ``streamlines`` (N, L, 3); the env state is not touched.
        Returns ``(should_stop bool (N,), flags int (N,))`` on the host."""
        pts = np.ascontiguousarray(streamlines, dtype=np.float32)
        n, n_points = pts.shape[0], pts.shape[1]
        tail = np.zeros((n, 3, 3), dtype=np.float32)
        k = min(3, n_points)
        tail[:, 3 - k:] = pts[:, n_points - k:]
        self._ensure_capacity(1)
        tail_dev = torch.from_numpy(tail).to(self.device)
        out = torch.empty(n, dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.ttl_env_stopping_flags(
            self._handle, tail_dev.data_ptr(), n, n_points, out.data_ptr(),
            self._stream()), 'ttl_env_stopping_flags')
        flags = out.to('cpu').numpy().astype(int)
        return flags != 0, flags

    def _is_stopping(self, streamlines):
        """tracking_env.py:22-45."""
        return self._compute_stopping_flags(streamlines)

    # ------------------------------------------------------------------ #
    # measurement support
    def scripted_actions(self, state, step, seed=0, wobble=0.05):
        """Policy stand-in for "env.step only" runs: (n_active, 3)
        float32 actions on the GPU from ``ttl_scripted_actions``."""
        n = self._n_active
        out = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ttl_scripted_actions(
            state.data_ptr(), state.stride(0), 7 * self._n_coef,
            self._idx_view(n).data_ptr(), n, int(seed) & 0xffffffff,
            int(step), float(wobble), out.data_ptr(), self._stream()),
            'ttl_scripted_actions')
        return out

    def scripted_actions_free(self, state, seed=0, wobble=0.05):
        """``scripted_actions`` as the policy of a free-running episode
        (``run_free_eager`` / ``run_free``): row count and step number are read
        on the device; actions for ``len(state)`` rows."""
        n = int(state.shape[0])
        out = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ttl_env_freerun_scripted_actions(
            self._handle, state.data_ptr(), state.stride(0), 7 * self._n_coef, n,
            int(seed) & 0xffffffff, float(wobble), out.data_ptr(), self._stream()),
            'ttl_env_freerun_scripted_actions')
        return out

    #: kernel classes `profile_begin` can bracket (ttl_env_profile_begin's mask bits)
    PROFILE_CLASSES = ('advance', 'prefix', 'state', 'proc_scatter')

    def profile_begin(self, max_launches=4096, classes=('state',)):
        """Bracket the step kernels of the given classes (PROFILE_CLASSES) with
        HIP events on the launch stream."""
        mask = sum(1 << self.PROFILE_CLASSES.index(c) for c in classes)
        _lib.check(self._lib.ttl_env_profile_begin(self._handle, max_launches,
                                                   mask), 'ttl_env_profile_begin')

    def profile_end(self):
        """{class: (total_ms, n_launches)} for every class of PROFILE_CLASSES."""
        import ctypes as C
        k = len(self.PROFILE_CLASSES)
        ms = (C.c_double * k)()
        n = (C.c_int32 * k)()
        _lib.check(self._lib.ttl_env_profile_end(self._handle, ms, n),
                   'ttl_env_profile_end')
        return {c: (ms[i], n[i]) for i, c in enumerate(self.PROFILE_CLASSES)}

    # ------------------------------------------------------------------ #
    # host views of the per-streamline state (reference attribute names)
    @property
    def continue_idx(self):
        return self._idx_view(self._n_active).to('cpu').numpy().astype(np.int64)

    @property
    def flags(self):
        return self._buf_flags[:self._n_total].to('cpu').numpy().astype(np.int64)

    @property
    def lengths(self):
        return self._buf_lengths[:self._n_total].to('cpu').numpy()

    @property
    def dones(self):
        return self._buf_dones[:self._n_total].to('cpu').numpy().astype(bool)

    @property
    def streamlines(self):
        """(N, max_nb_steps + 1, 3) float32 history, downloaded."""
        return self._buf_streamlines[:self._n_total].to('cpu').numpy()

    def get_streamlines(self):
        """Tracked streamlines in voxel space (tracking_env.py:247-294): each
        streamline's ``lengths[i]`` points, minus the last one if it raised the
        CURVATURE or MASK flag; ``data_per_streamline`` = seeds and flags.  The
        ragged list is packed on the device and downloaded once."""
        n = self._n_total
        flags_dev = self._buf_flags[:n]
        lengths_dev = self._buf_lengths[:n].to(torch.int64)
        cut = (StoppingFlags.STOPPING_CURVATURE.value |
               StoppingFlags.STOPPING_MASK.value)
        keep_len = lengths_dev - ((flags_dev & cut) != 0).to(torch.int64)
        from tracktolearn_amd.parallel import pack_points
        points = pack_points(self._buf_streamlines[:n], keep_len).to('cpu').numpy()
        keep_len_np = keep_len.to('cpu').numpy()
        offsets = np.concatenate(([0], np.cumsum(keep_len_np)))
        stopped_streamlines = [points[offsets[i]:offsets[i + 1]]
                               for i in range(n)]
        return Tractogram(
            streamlines=stopped_streamlines,
            data_per_streamline={'seeds': self.initial_points,
                                 'flags': self.flags})
