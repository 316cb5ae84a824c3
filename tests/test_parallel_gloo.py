"""world_size-2 gloo tests (CPU) of the sharding helpers and of the ragged
all-gather that collates finished tracts."""
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tracktolearn_amd import parallel


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 262144, 1048576 + 3):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d


def _fake_env(rank):
    """A host-tensor stand-in for the env buffers: rank r owns 3 + 2r
    streamlines of up to 6 points."""
    rng = np.random.RandomState(10 + rank)
    n, T = 3 + 2 * rank, 6
    hist = torch.from_numpy(rng.standard_normal((n, T, 3)).astype(np.float32))
    lengths = torch.from_numpy(rng.randint(1, T + 1, n).astype(np.int32))
    flags = torch.from_numpy(rng.choice([1, 2, 4, 5], n).astype(np.int32))
    return SimpleNamespace(_n_total=n, _buf_streamlines=hist,
                           _buf_lengths=lengths, _buf_flags=flags,
                           initial_points=rng.uniform(size=(n, 3)))


def _expected():
    lines, flags = [], []
    for r in range(2):
        e = _fake_env(r)
        for i in range(e._n_total):
            k = int(e._buf_lengths[i]) - (1 if int(e._buf_flags[i]) & 5 else 0)
            lines.append(e._buf_streamlines[i, :k].numpy())
            flags.append(int(e._buf_flags[i]))
    return lines, flags


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        env = _fake_env(rank)
        pieces = parallel.all_gather_ragged(env._buf_lengths)
        assert [p.shape[0] for p in pieces] == [3, 5]
        lengths, flags = parallel.all_gather_tract_index(env)
        lines, want_flags = _expected()
        # the collate the Tracker uses: exact-size gather to rank 0 only
        rows, counts = parallel.gather_ragged_to_root(env._buf_lengths)
        assert counts == [3, 5]
        if rank == 0:
            assert torch.equal(rows, lengths)
        else:
            assert rows is None
        empty, counts = parallel.gather_ragged_to_root(
            env._buf_streamlines[:0] if rank == 1 else env._buf_streamlines[:2])
        assert counts == [2, 0] and (empty is None) == (rank == 1)
        got = parallel.gather_tract_arrays(env)
        if rank == 0:
            keep_all, flags_all, pts_all, moved = got
            assert pts_all.shape[0] == int(keep_all.sum()) == sum(len(l) for l in lines)
            assert moved == sum(len(l) for l in lines[3:]) * 12 + 5 * (8 + 4)
        else:
            assert got is None
        for tg in (parallel.all_gather_tractogram(env),
                   parallel.gather_tractogram(env)):
            if tg is None:
                assert rank == 1
                continue
            assert len(tg) == 8 and lengths.shape[0] == 8
            assert list(tg.data_per_streamline['flags']) == want_flags
            assert tg.data_per_streamline['seeds'].shape == (8, 3)
            for got_line, want in zip(tg.streamlines, lines):
                assert np.array_equal(got_line, want)
        q.put((rank, 'ok'))
    except Exception as exc:          # pragma: no cover
        q.put((rank, repr(exc)))
    finally:
        dist.destroy_process_group()


def test_ragged_all_gather_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, 'ok'), (1, 'ok')], results


def _dp_worker(rank, world, port, q, fused=False, overlap=True):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from tracktolearn_amd.algorithms.sac_auto import SACAuto
        W, B = 27, 32
        torch.manual_seed(100 + rank)            # different initial weights
        alg = SACAuto(W, 3, '32-32', n_actors=8, batch_size=B, replay_size=100,
                      rng=None, device=torch.device('cpu'))
        if fused:
            # the GPU learner's schedule (shared/fused.py: one all-reduce per gradient
            # arena) with the kernels replaced by their torch restatement
            from ref_learner_ops import TorchOps
            from tracktolearn_amd.algorithms.shared import fused as fused_mod
            alg._fused_ops = TorchOps()
            # critics' average beside the actor's backward / both after the backward
            fused_mod._FusedNets.dp_overlap = overlap
        alg.enable_data_parallel()
        g = torch.Generator().manual_seed(7)
        full = [torch.randn(2 * B, W, generator=g),
                torch.tanh(torch.randn(2 * B, 3, generator=g)),
                torch.randn(2 * B, W, generator=g), torch.rand(2 * B, generator=g),
                (torch.rand(2 * B, generator=g) > 0.2).float()]
        eps = [torch.randn(2 * B, 3, generator=g) for _ in range(2)]
        calls = {'i': 0}

        def noise(like):
            calls['i'] += 1
            return eps[calls['i'] % 2][rank * B:(rank + 1) * B]
        alg.noise_fn = noise
        mine = [t[rank * B:(rank + 1) * B] for t in full]     # this rank's half
        for _ in range(3):
            alg.update(mine)
        assert (alg._fused is not None) == fused
        flat = torch.cat([p.detach().reshape(-1) for p in
                          list(alg.agent.actor.parameters()) +
                          list(alg.agent.critic.parameters()) +
                          list(alg.target.critic.parameters()) + [alg.log_alpha]])
        q.put((rank, flat.numpy()))
    except Exception as exc:          # pragma: no cover
        q.put((rank, repr(exc)))
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize('fused', [False, True, 'sequential'])
def test_data_parallel_learner_world2_gloo(fused):
    """Two learner replicas on half batches each == one learner on the whole
    batch: identical replicas, and equal (to rounding) to the single-process
    update started from rank 0's weights.  `fused`: the replicas run the GPU
    learner's hand-scheduled update (arena all-reduce -- overlapped with the
    backward, or 'sequential': both after it --, deterministic reductions), the
    single process the autograd formulation."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    overlap = fused != 'sequential'
    fused = bool(fused)
    port = 31500 + os.getpid() % 2000 + (37 if fused else 0) + (0 if overlap else 11)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, fused, overlap))
             for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(v, np.ndarray) for v in results.values()), results
    assert np.array_equal(results[0], results[1])          # replicas identical
    # single process, whole batch, rank 0's initial weights
    W, B = 27, 32
    torch.manual_seed(100)
    alg = SACAuto(W, 3, '32-32', n_actors=8, batch_size=2 * B, replay_size=100,
                  rng=None, device=torch.device('cpu'))
    g = torch.Generator().manual_seed(7)
    full = [torch.randn(2 * B, W, generator=g),
            torch.tanh(torch.randn(2 * B, 3, generator=g)),
            torch.randn(2 * B, W, generator=g), torch.rand(2 * B, generator=g),
            (torch.rand(2 * B, generator=g) > 0.2).float()]
    eps = [torch.randn(2 * B, 3, generator=g) for _ in range(2)]
    calls = {'i': 0}

    def noise(like):
        calls['i'] += 1
        return eps[calls['i'] % 2]
    alg.noise_fn = noise
    for _ in range(3):
        alg.update(full)
    flat = torch.cat([p.detach().reshape(-1) for p in
                      list(alg.agent.actor.parameters()) +
                      list(alg.agent.critic.parameters()) +
                      list(alg.target.critic.parameters()) + [alg.log_alpha]]).numpy()
    assert np.abs(flat - results[0]).max() < 5e-6


class _ToyEnv:
    """Just enough of the env's device loop for DDPG._episode: `n` rows, a
    fixed fraction stops at every step, all stop at `max_steps`."""

    def __init__(self, n, width, max_steps, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.n, self.width, self.max_steps, self.t = n, width, max_steps, 0

    def reset(self):
        self.t = 0
        return torch.randn(self.n, self.width, generator=self.g)

    def step_device(self, action):
        n = action.shape[0]
        self.t += 1
        done = (torch.rand(n, generator=self.g) < 0.3).to(torch.uint8)
        if self.t >= self.max_steps:
            done[:] = 1
        keep = done == 0
        n_keep = int(keep.sum())
        dest = torch.empty(n, dtype=torch.int32)
        dest[keep] = torch.arange(n_keep, dtype=torch.int32)
        dest[~keep] = torch.arange(n_keep, n, dtype=torch.int32)
        self._state = torch.randn(n, self.width, generator=self.g)
        self._n_keep = n_keep
        reward = torch.rand(n, generator=self.g).double()
        return self._state, reward, done, {'row_dest': dest}

    def harvest(self):
        return self._state[:self._n_keep], None


def _dp_episode_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from tracktolearn_amd.algorithms.sac_auto import SACAuto
        W, B = 27, 16
        torch.manual_seed(200 + rank)
        alg = SACAuto(W, 3, '32-32', n_actors=8, batch_size=B, replay_size=500,
                      rng=None, device=torch.device('cpu'))
        alg.enable_data_parallel()
        # unequal shards: rank 0 tracks 12 rows for <= 3 steps and crosses
        # start_timesteps late, rank 1 tracks 40 rows for <= 9 steps
        alg.start_timesteps = 20
        env = _ToyEnv(12 if rank == 0 else 40, W, 3 if rank == 0 else 9, 5 + rank)
        lengths = []
        for _ in range(2):                      # two episodes back to back
            _, _, length, _ = alg._episode(env.reset(), env)
            lengths.append(length)
        flat = torch.cat([p.detach().reshape(-1) for p in
                          list(alg.agent.actor.parameters()) +
                          list(alg.agent.critic.parameters()) + [alg.log_alpha]])
        q.put((rank, (flat.numpy(), alg.total_it, lengths)))
    except Exception as exc:          # pragma: no cover
        q.put((rank, repr(exc)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_episode_schedule_world2_gloo():
    """`_episode` with data-parallel replicas whose shards differ in size and
    episode length: no rank is left alone in an all-reduce, every
    rank performs the same number of updates, the replicas stay identical."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_episode_worker, args=(r, 2, port, q))
             for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(v, tuple) for v in results.values()), results
    (w0, it0, len0), (w1, it1, len1) = results[0], results[1]
    assert it0 == it1 > 0
    assert len0 != len1                         # the shards really differed
    assert np.array_equal(w0, w1)
