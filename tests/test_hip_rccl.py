"""The RCCL branch of tracktolearn_amd/parallel.py on hardware.

A 1-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on one device),
so the multi-rank tests run over gloo (tests/test_parallel_gloo.py, the bench
rehearsal).  What those never execute is the `nccl` branch itself: device
tensors through `_coll_device`, `init_process_group('nccl', device_id=...)`,
`all_gather_into_tensor` / `batch_isend_irecv` / `all_reduce` on cuda tensors.
This test runs every collective of the path on a WORLD-SIZE-1 nccl group on
cuda:0 (in a child process: the process group is global state), against the
values the single-rank path gives without a group.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from tracktolearn_amd import parallel
from tracktolearn_amd.environments import TrackingEnvironment
from tracktolearn_amd.utils.synthetic import synthetic_seeds, synthetic_subject

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
assert dist.get_backend() == 'nccl'
assert parallel._coll_device(None) == torch.device('cuda', 0)

subject = synthetic_subject(24, 45, seed=7, peaks=False, affine_dtype=np.float32)
dto = dict(n_dirs=4, theta=30.0, npv=1, binary_stopping_threshold=0.1, step_size=0.75,
           min_length=2.0, max_length=30.0, compute_reward=False, alignment_weighting=1.0,
           oracle_bonus=0.0, rng=np.random.RandomState(0), device=torch.device('cuda:0'),
           target_sh_order=8)
env = TrackingEnvironment(subject, 'testing', dto)
env.seeds = synthetic_seeds(subject[1].data, 3000, seed=5)
state = env.reset(0, 3000)
step = 0
while env._n_active:
    env.step_device(env.scripted_actions(state, step, 1, 0.05))
    state, _ = env.harvest()
    step += 1

# the collate, through RCCL (device tensors end to end)
keep, flags, points = parallel.tract_arrays(env)
got = parallel.gather_tract_arrays(env)
assert got is not None
keep_all, flags_all, pts_all, moved = got
assert keep_all.is_cuda and pts_all.is_cuda and moved == 0
assert torch.equal(keep_all, keep) and torch.equal(flags_all, flags)
assert torch.equal(pts_all, points)
tract = parallel.gather_tractogram(env)
ref = env.get_streamlines()
assert len(tract.streamlines) == len(ref.streamlines) == 3000
for a, b in zip(tract.streamlines, ref.streamlines):
    assert np.array_equal(a, b)
assert np.array_equal(tract.data_per_streamline['flags'], ref.data_per_streamline['flags'])
assert np.array_equal(tract.data_per_streamline['seeds'], ref.data_per_streamline['seeds'])
everywhere = parallel.all_gather_tractogram(env)
assert all(np.array_equal(a, b) for a, b in zip(everywhere.streamlines, ref.streamlines))
lengths, flags2 = parallel.all_gather_tract_index(env)
assert lengths.is_cuda and torch.equal(lengths, env._buf_lengths[:3000])
assert parallel.all_gather_counts(17) == [17]

# the learner's gradient exchange and weight broadcast, through RCCL
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3)).cuda()
net(torch.randn(11, 7, device='cuda')).square().sum().backward()
before = [p.grad.clone() for p in net.parameters()]
parallel.all_reduce_gradients(list(net.parameters()))
for g, p in zip(before, net.parameters()):
    assert torch.equal(g, p.grad)          # world size 1: sum / 1
weights = [p.detach().clone() for p in net.parameters()]
parallel.broadcast_parameters([net])
assert all(torch.equal(w, p) for w, p in zip(weights, net.parameters()))

# the fused learner's gradient exchange through RCCL: the critics' arena averaged
# beside the actor's backward (async all_reduce on the device), or both arenas after
# the backward -- on one rank either equals the update without a group, bit for bit
from tracktolearn_amd.algorithms.sac_auto import SACAuto
from tracktolearn_amd.algorithms.shared import fused as fused_mod
g = torch.Generator().manual_seed(3)
Wd, Bt = 37, 256
batch = [torch.randn(Bt, Wd, generator=g).cuda(), torch.tanh(torch.randn(Bt, 3, generator=g)).cuda(),
         torch.randn(Bt, Wd, generator=g).cuda(), torch.rand(Bt, generator=g).cuda(),
         (torch.rand(Bt, generator=g) > 0.2).float().cuda()]
results = []
for mode in ('none', 'overlap', 'sequential'):
    torch.manual_seed(11)
    alg = SACAuto(Wd, 3, '64-64', n_actors=8, batch_size=Bt, replay_size=100, rng=None,
                  device=torch.device('cuda:0'))
    if mode != 'none':
        fused_mod._FusedNets.dp_overlap = mode == 'overlap'
        alg.enable_data_parallel()
    for _ in range(3):
        alg.update(batch)
    assert alg._fused is not None
    results.append(torch.cat([p.detach().reshape(-1) for p in
                              list(alg.agent.parameters()) + list(alg.target.parameters())
                              + [alg.log_alpha]]))
torch.cuda.synchronize()
assert torch.equal(results[0], results[1]) and torch.equal(results[0], results[2])

# bench.py's reductions on the device
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device='cuda:0')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert t.tolist() == [1.5, 2.5]
dist.destroy_process_group()
print('rccl-ok', step)
'''


def test_collate_and_gradient_exchange_on_a_world_size_1_nccl_group():
    port = 29900 + os.getpid() % 90
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               PYTHONPATH=ROOT)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    out = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert 'rccl-ok' in out.stdout


def test_pack_streamlines_kernel_equals_the_masked_indexing():
    """`ttl_pack_streamlines` (ABI v8; `parallel.pack_points` on CUDA tensors,
    `get_streamlines`): the ragged pack of tracked streamlines, bit for bit what
    boolean-mask indexing of the history gives, incl. rows that keep nothing and
    a non-contiguous slice of the history buffer."""
    import torch
    from tracktolearn_amd import parallel
    torch.manual_seed(1)
    for n, T in ((1, 5), (257, 33), (5000, 268)):
        hist = torch.randn((n, T + 2, 3), device='cuda:0')[:, :T]   # rows further apart than T points
        keep = torch.randint(0, T + 1, (n,), device='cuda:0')
        keep[0] = 0 if n > 1 else T
        got = parallel.pack_points(hist, keep)
        steps = torch.arange(T, device='cuda:0')
        want = hist[steps[None, :] < keep[:, None]]
        assert got.shape == want.shape and torch.equal(got, want)
    empty = parallel.pack_points(torch.randn((4, 6, 3), device='cuda:0'),
                                 torch.zeros(4, dtype=torch.int64, device='cuda:0'))
    assert empty.shape == (0, 3)
    # host tensors (the gloo tests) keep the indexing path
    h = torch.randn(7, 5, 3)
    k = torch.tensor([5, 0, 3, 1, 5, 2, 4])
    assert parallel.pack_points(h, k).shape == (int(k.sum()), 3)
