"""Multi-GPU layout of the environment path: one process per GPU, streamlines
sharded, volumes replicated, no collective on the step path.

The reference is single-process.  Streamlines are independent
given the read-only volumes, so rank r of R tracks the contiguous slice
``shard_bounds(n, r, R)`` of every seed batch and never talks to its peers
while stepping.  The only exchange is collating the finished tracts at
``get_streamlines()`` time.  Only one rank consumes them (rank 0 writes the
file), so the collate is a gather-to-root of exact sizes: one all-gather of
the per-rank row counts (8 bytes each), then every rank sends its kept
lengths / flags / seeds / packed points straight into its slice of the root's
buffers (batched point-to-point over RCCL/xGMI -- xGMI is point to point, each
sender uses its own link to the root; gloo in the CPU tests).  No padding, and
1/R of the bytes an all-gather of the same data would land on every GPU.
``all_gather_ragged`` remains for callers that need the data everywhere.
"""
import numpy as np
import torch
import torch.distributed as dist

from tracktolearn_amd.environments.stopping_criteria import StoppingFlags
from tracktolearn_amd.tractogram import Tractogram


def shard_bounds(n, rank, world):
    """[start, end) of rank's contiguous shard of n seeds (ceil split; the
    last shards may be short or empty)."""
    per = -(-n // world)
    start = min(rank * per, n)
    return start, min(start + per, n)


def kept_lengths(lengths, flags):
    """Points kept per streamline: the last point is dropped when CURVATURE or
    MASK stopped it (TrackToLearn/environments/tracking_env.py:263-284)."""
    cut = (StoppingFlags.STOPPING_CURVATURE.value |
           StoppingFlags.STOPPING_MASK.value)
    return lengths.to(torch.int64) - ((flags & cut) != 0).to(torch.int64)


def pack_points(history, keep_len):
    """Ragged pack: (n, T, 3) history + kept lengths -> (sum(keep_len), 3)
    points, streamline-major.  On the GPU: one wave per streamline
    (``ttl_pack_streamlines``; the boolean-mask indexing it replaces took
    56 ms for the 68 M points of a 1 048 576-streamline tractogram, the kernel
    moves them at memory speed).  Host tensors (the gloo tests) index."""
    if not history.is_cuda:
        steps = torch.arange(history.shape[1], device=history.device)
        return history[steps[None, :] < keep_len[:, None]]
    from tracktolearn_amd import _lib
    from tracktolearn_amd.environments.env import _raw_stream
    n = int(history.shape[0])
    keep = keep_len.to(torch.int64).contiguous()
    ends = torch.cumsum(keep, 0)
    total = int(ends[-1].item()) if n else 0
    out = torch.empty((total, 3), dtype=torch.float32, device=history.device)
    if total:
        # rows may be further apart than their points (a slice of a larger buffer)
        hist = history if history.stride(2) == 1 and history.stride(1) == 3 \
            else history.contiguous()
        offsets = ends - keep
        import ctypes as C
        _lib.check(_lib.load().ttl_pack_streamlines(
            hist.data_ptr(), hist.stride(0), keep.data_ptr(), offsets.data_ptr(), n,
            out.data_ptr(), C.c_void_p(_raw_stream(history.device.index or 0))),
            'ttl_pack_streamlines')
    return out


def all_gather_counts(values, group=None):
    """All-gather one int64 per rank."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(values)], dtype=torch.int64,
                        device=_coll_device(group))
    out = torch.empty(world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.tolist()


def _coll_device(group):
    backend = dist.get_backend(group)
    if backend == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_gather_ragged(rows, group=None):
    """All-gather tensors whose first dimension differs per rank: pad to the
    longest shard, one all_gather_into_tensor, cut the padding.  Returns the
    per-rank pieces in rank order."""
    world = dist.get_world_size(group)
    dev = _coll_device(group)
    rows = rows.to(dev).contiguous()
    counts = all_gather_counts(rows.shape[0], group)
    longest = max(max(counts), 1)
    padded = torch.zeros((longest,) + tuple(rows.shape[1:]), dtype=rows.dtype,
                         device=dev)
    padded[:rows.shape[0]] = rows
    out = torch.empty((world * longest,) + tuple(rows.shape[1:]),
                      dtype=rows.dtype, device=dev)
    dist.all_gather_into_tensor(out, padded, group=group)
    return [out[r * longest:r * longest + counts[r]] for r in range(world)]


def gather_ragged_to_root(rows, dst=0, group=None):
    """Gather tensors whose first dimension differs per rank on rank ``dst``
    only, exact sizes, no padding: the root allocates sum(counts) rows and
    every other rank's send lands directly in its slice.  Returns
    ``(all_rows, counts)`` on the root (rank order) and ``(None, counts)``
    elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = _coll_device(group)
    rows = rows.to(dev).contiguous()
    counts = all_gather_counts(rows.shape[0], group)
    ops, out = [], None
    if rank == dst:
        out = torch.empty((sum(counts),) + tuple(rows.shape[1:]),
                          dtype=rows.dtype, device=dev)
        offs = np.concatenate(([0], np.cumsum(counts)))
        out[offs[dst]:offs[dst + 1]] = rows
        for r in range(world):
            if r != dst and counts[r] > 0:
                peer = dist.get_global_rank(group, r) if group is not None else r
                ops.append(dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]],
                                      peer, group))
    elif rows.shape[0] > 0:
        peer = dist.get_global_rank(group, dst) if group is not None else dst
        ops.append(dist.P2POp(dist.isend, rows, peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts


def tract_arrays(env):
    """This rank's finished tracts as device arrays: (kept lengths int64 (n,),
    flags int32 (n,), packed points float32 (sum(kept), 3))."""
    n = env._n_total
    lengths, flags = env._buf_lengths[:n], env._buf_flags[:n]
    keep = kept_lengths(lengths, flags)
    return keep, flags, pack_points(env._buf_streamlines[:n], keep)


def gather_tract_arrays(env, dst=0, group=None):
    """Every rank's ``tract_arrays`` on rank ``dst`` (rank order): (keep,
    flags, points, bytes received) there, None elsewhere.  This is the
    path's one exchange step; bench.py times it as ``collate_ms``."""
    keep, flags, points = tract_arrays(env)
    keep_all, _ = gather_ragged_to_root(keep, dst, group)
    flags_all, _ = gather_ragged_to_root(flags, dst, group)
    pts_all, _ = gather_ragged_to_root(points, dst, group)
    if keep_all is None:
        return None
    moved = sum(int(a.numel() - b.numel()) * a.element_size() for a, b in
                ((keep_all, keep), (flags_all, flags), (pts_all, points)))
    return keep_all, flags_all, pts_all, moved


def gather_tractogram(env, dst=0, group=None):
    """The sharded ``get_streamlines()``: every rank's finished tracts as one
    Tractogram (rank order, then streamline order) on rank ``dst``; None on
    the other ranks."""
    seeds = torch.from_numpy(np.ascontiguousarray(env.initial_points,
                                                  dtype=np.float64))
    seeds_all, _ = gather_ragged_to_root(seeds, dst, group)
    got = gather_tract_arrays(env, dst, group)
    if got is None:
        return None
    keep_all, flags_all, pts_all, _ = got
    keep_np = keep_all.cpu().numpy()
    pts_np = pts_all.cpu().numpy()
    offsets = np.concatenate(([0], np.cumsum(keep_np)))
    lines = [pts_np[offsets[i]:offsets[i + 1]] for i in range(len(keep_np))]
    return Tractogram(streamlines=lines,
                      data_per_streamline={
                          'seeds': seeds_all.cpu().numpy(),
                          'flags': flags_all.cpu().numpy().astype(np.int64)})


def all_gather_tract_index(env, group=None):
    """(lengths, flags) of every rank's streamlines, concatenated in rank
    order (device tensors)."""
    n = env._n_total
    lengths = torch.cat(all_gather_ragged(env._buf_lengths[:n], group))
    flags = torch.cat(all_gather_ragged(env._buf_flags[:n], group))
    return lengths, flags


def all_gather_tractogram(env, group=None):
    """Every rank's finished tracts as one Tractogram (rank order, then
    streamline order) -- the sharded ``get_streamlines()``."""
    n = env._n_total
    lengths, flags = env._buf_lengths[:n], env._buf_flags[:n]
    keep = kept_lengths(lengths, flags)
    points = pack_points(env._buf_streamlines[:n], keep)
    keep_all = torch.cat(all_gather_ragged(keep, group)).cpu().numpy()
    flags_all = torch.cat(all_gather_ragged(flags, group)).cpu().numpy()
    seeds = torch.from_numpy(np.ascontiguousarray(env.initial_points,
                                                  dtype=np.float64))
    seeds_all = torch.cat(all_gather_ragged(seeds, group)).cpu().numpy()
    pts_all = torch.cat(all_gather_ragged(points, group)).cpu().numpy()
    offsets = np.concatenate(([0], np.cumsum(keep_all)))
    lines = [pts_all[offsets[i]:offsets[i + 1]] for i in range(len(keep_all))]
    return Tractogram(streamlines=lines,
                      data_per_streamline={'seeds': seeds_all,
                                           'flags': flags_all.astype(np.int64)})


# --------------------------------------------------------------------------
# data-parallel learner (BASELINE config 5: training on 8 GPUs)
# --------------------------------------------------------------------------
def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank ``src``'s weights (and buffers)."""
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def all_reduce_gradients(params, group=None):
    """Average the gradients of ``params`` over the ranks with ONE flattened
    all-reduce (RCCL ring over xGMI is per-link bound, so a single ~10-20 MB
    bucket per network beats one collective per tensor).  Parameters without
    a gradient are skipped on every rank alike."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    offset = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[offset:offset + n].view_as(g))
        offset += n
