"""CPU-only tests of the host logic and of the C-ABI library surface (no
compute call: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import load_trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tracktolearn_amd import _lib
    declared = set()
    for name in ('ttl_hip.h', 'ttl_learner.h'):
        header = open(os.path.join(ROOT, 'include', name)).read()
        body = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
        declared |= set(re.findall(r'\b(ttl_[a-z0-9_]+)\s*\(', body))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ttl_abi_version() == _lib.ABI_VERSION
    assert lib.ttl_env_workspace_bytes(1024) > 1024 * 13
    # built with -fvisibility=hidden: nothing but the TTL_API entry points (and
    # the toolchain's own __hip_* / std:: data symbols) leaves the library
    import shutil
    import subprocess
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    if os.path.exists(nm) or shutil.which(nm):
        out = subprocess.run([nm, '-D', '--defined-only', _lib.LIB_PATH],
                             capture_output=True, text=True, check=True).stdout
        rows = [r.split() for r in out.splitlines() if r.strip()]
        functions = {r[-1] for r in rows if r[-2] in ('T', 'W', 't', 'w')}
        assert functions == declared, functions ^ declared
        assert not [r[-1] for r in rows if 'ttl_detail' in r[-1]]


def test_descriptor_layout_matches_header():
    """ctypes struct mirrors the C struct field for field (names, order)."""
    from tracktolearn_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'ttl_hip.h')).read()
    struct = header[header.index('typedef struct ttl_env_desc {'):
                    header.index('} ttl_env_desc;')]
    struct = re.sub(r'/\*.*?\*/', '', struct, flags=re.S)
    names = re.findall(r'\b\**([a-z_0-9]+)(?:\[3\])?;', struct)
    assert names == [f[0] for f in _lib.EnvDesc._fields_]


def test_create_rejects_bad_descriptors():
    """Argument validation happens on the host, before any device work."""
    from tracktolearn_amd import _lib
    lib = _lib.load()
    d = _lib.EnvDesc()
    h = ctypes.c_void_p()
    assert lib.ttl_env_create(ctypes.byref(d), ctypes.byref(h)) == -1
    assert b'abi_version' in lib.ttl_last_error()
    d.abi_version = _lib.ABI_VERSION
    d.mode = 7
    assert lib.ttl_env_create(ctypes.byref(d), ctypes.byref(h)) == -1
    assert b'mode' in lib.ttl_last_error()
    assert lib.ttl_env_step(None, None, None, 1, 0, None, 0, None, None, None,
                            None) == -1
    assert lib.ttl_pack_sh_volume(None, None, None, 1, 4, 0, None) == -1
    dims = (ctypes.c_int32 * 3)(5, 6, 7)
    assert lib.ttl_sh_volume_records(dims, 0) == 210
    assert lib.ttl_sh_volume_records(dims, 1) == 2 * 2 * 2 * 64


def test_missing_library_fails_loudly(monkeypatch):
    from tracktolearn_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libttl_hip.so')
    with pytest.raises(_lib.TTLError, match='no CPU fallback'):
        _lib.load()


def test_env_refuses_cpu_device():
    import torch
    from tracktolearn_amd.datasets.utils import MRIDataVolume
    from tracktolearn_amd.environments import TrackingEnvironment
    vol = MRIDataVolume(np.zeros((4, 4, 4, 45), np.float32), np.eye(4))
    dto = dict(n_dirs=4, theta=30, npv=1, binary_stopping_threshold=0.1,
               step_size=0.75, min_length=20, max_length=200,
               compute_reward=False, alignment_weighting=1, rng=None,
               device=torch.device('cpu'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        TrackingEnvironment((vol, vol, vol, None, None), 'testing', dto)


def test_curvature_threshold_equals_numpy_arccos_decision():
    """The dot-product threshold reproduces the reference's golden curvature
    decisions (2048 triples, most of them within 1e-6 rad of theta)."""
    from tracktolearn_amd.environments.stopping_criteria import \
        curvature_dot_threshold
    z = load_trace('isolated_functions')
    tri = z['curvy_in']
    with np.errstate(all='ignore'):
        d1 = tri[:, 2] - tri[:, 1]
        d0 = tri[:, 1] - tri[:, 0]
        u = d1 / np.sqrt(np.einsum('...i,...i', d1, d1))[..., None]
        v = d0 / np.sqrt(np.einsum('...i,...i', d0, d0))[..., None]
        dot = np.einsum('ij,ij->i', u, v)
        c, enabled = curvature_dot_threshold(float(z['curvy_theta']))
        mine = (dot <= c) & (dot >= -1)
    assert enabled
    assert np.array_equal(mine, z['curvy_out'])
    # exhaustive check of the step property near the boundary
    bits = np.array([c], np.float32).view(np.int32)[0]
    xs = (bits + np.arange(-5000, 5000)).astype(np.int32).view(np.float32)
    assert np.array_equal(np.arccos(xs) > np.deg2rad(float(z['curvy_theta'])),
                          xs <= c)
    assert curvature_dot_threshold(200)[1] is False


def test_flag_helpers_and_length_conversion():
    from tracktolearn_amd.environments.stopping_criteria import (
        StoppingFlags, count_flags, is_flag_set)
    from tracktolearn_amd.datasets.utils import convert_length_mm2vox
    flags = np.array([0, 1, 2, 4, 5, 6, 64])
    assert list(is_flag_set(flags, StoppingFlags.STOPPING_MASK)) == \
        [False, True, False, False, True, False, False]
    assert count_flags(flags, StoppingFlags.STOPPING_CURVATURE) == 3
    s32 = convert_length_mm2vox(0.75, np.eye(4, dtype=np.float32))
    s64 = convert_length_mm2vox(0.75, np.eye(4))
    assert s32.dtype == np.float32 and s64.dtype == np.float64
    with pytest.raises(ValueError):
        convert_length_mm2vox(1.0, np.diag([1.0, 2.0, 1.0, 1.0]))


def test_tractogram_container():
    from tracktolearn_amd.tractogram import Tractogram
    a = Tractogram([np.zeros((3, 3), np.float32), np.ones((2, 3), np.float32)],
                   {'flags': np.array([1, 4]), 'seeds': np.zeros((2, 3))})
    b = Tractogram([np.full((4, 3), 2, np.float32)],
                   {'flags': np.array([2]), 'seeds': np.ones((1, 3))})
    a += b
    assert len(a) == 3 and list(a.data_per_streamline['flags']) == [1, 4, 2]
    items = list(a)
    assert items[2].streamline.shape == (4, 3)
    assert items[1].data_for_streamline['flags'] == 4
    A = np.diag([2.0, 2.0, 2.0, 1.0])
    A[:3, 3] = 1
    a.apply_affine(A)
    assert np.allclose(a.streamlines[1], 3.0)


def test_library_binds_to_torchs_hip_runtime():
    """libttl_hip.so must share the HIP runtime PyTorch loaded (one runtime per
    process), whatever the import order was: tracktolearn_amd._lib imports
    torch before it dlopens the library."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from tracktolearn_amd import _lib\n"       # before any `import torch`
        "_lib.load()\n"
        "import torch\n"
        "maps = open('/proc/self/maps').read()\n"
        "libs = sorted({l.split()[-1] for l in maps.splitlines() "
        "if 'libamdhip64' in l})\n"
        "print(len(libs), libs)\n") % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == '1', out.stdout


def test_bench_algorithmic_bytes_match_the_survey_figures():
    """A(45, 4) = 11 986 B and A(45, 100) = 14 290 B per
    streamline-step; the dominant kernel's share is gather + history + row."""
    import bench
    assert bench.algorithmic_bytes(45, 4) == (11986, 4 * 56 * 45 + 12 * 5 + 4 * 327)
    assert bench.algorithmic_bytes(45, 100)[0] == 14290
    assert bench.algorithmic_bytes(45, 4)[1] == 11448


def test_sharded_tracking_draws_independent_noise_per_rank():
    """every rank of a sharded `ttl_track` run shares the seed for
    seed generation / shuffling but must not inject the same exploration noise
    into row i of every shard."""
    from types import SimpleNamespace
    import torch
    from tracktolearn_amd.environments.noisy_tracking_env import \
        NoisyTrackingEnvironment
    from tracktolearn_amd.runners.ttl_track import per_rank_noise_rng
    actions = torch.zeros((5, 3))
    draws = []
    for rank in (0, 1):
        shared = np.random.RandomState(1337)
        env = SimpleNamespace(noise=0.1, device_noise=False, rng=shared,
                              noise_rng=per_rank_noise_rng(1337, rank),
                              device=torch.device('cpu'))
        draws.append(NoisyTrackingEnvironment._noise_for(env, actions).numpy())
        # the shared stream was not consumed: seeds / shuffles stay in step
        assert shared.randint(1 << 30) == np.random.RandomState(1337).randint(1 << 30)
    assert draws[0].shape == (5, 3) and draws[0].dtype == np.float64
    assert not np.allclose(draws[0], draws[1])
    # single process: the reference's single stream (noisy_tracking_env.py:73)
    env = SimpleNamespace(noise=0.1, device_noise=False, noise_rng=None,
                          rng=np.random.RandomState(7), device=torch.device('cpu'))
    one = NoisyTrackingEnvironment._noise_for(env, actions).numpy()
    assert np.array_equal(one, np.random.RandomState(7).normal(0., 0.1, (5, 3)))


def test_lazy_state_rows_behave_like_the_gathered_tensor():
    """`step()` hands its state rows out as a `_LazyStateRows`: shape, dtype and
    device of the real thing, gathered into the reference's row order on first
    use; every torch operation then sees the gathered values."""
    import torch
    from tracktolearn_amd.environments.tracking_env import _LazyStateRows
    rows = torch.arange(20, dtype=torch.float32).view(5, 4)
    row_dest = torch.tensor([3, 0, 4, 1, 2], dtype=torch.int32)
    want = rows[row_dest.long()]
    lazy = _LazyStateRows(rows, row_dest)
    assert isinstance(lazy, torch.Tensor) and lazy.shape == (5, 4)
    assert lazy.dtype == torch.float32 and lazy.device == rows.device
    assert lazy._value is None                                   # nothing gathered yet
    assert torch.equal(lazy.clone(), want) and lazy._value is not None
    lazy = _LazyStateRows(rows, row_dest)
    assert torch.equal(lazy[torch.tensor([True, False, True, False, False])], want[[0, 2]])
    assert torch.equal(torch.cat([lazy, lazy]), torch.cat([want, want]))
    assert float(lazy.sum()) == float(want.sum()) and len(lazy) == 5
    assert np.array_equal(lazy.double().numpy(), want.double().numpy())
    buf = torch.zeros(5, 4)
    buf.copy_(lazy)
    assert torch.equal(buf, want)
    # the source buffers may be reused once the rows were gathered
    lazy2 = _LazyStateRows(rows, row_dest)
    got = lazy2 + 0
    rows.zero_()
    assert torch.equal(got, want) and torch.equal(lazy2 * 1, want)


def test_lazy_state_rows_are_safe_for_a_caller_who_changes_nothing():
    """The object `step()` returns, used exactly as the reference's callers use
    a state tensor (ddpg.py:187-224 feeding replay.py:38-89) and through every
    accessor that bypasses the torch dispatcher: nothing raises, nothing reads a
    null pointer, no-op conversions give a plain tensor."""
    import copy
    import pickle
    import torch
    from tracktolearn_amd.environments.tracking_env import _LazyStateRows
    rows = torch.arange(28, dtype=torch.float32).view(7, 4)
    row_dest = torch.tensor([3, 0, 4, 1, 6, 5, 2], dtype=torch.int32)
    want = rows[row_dest.long()]

    def lazy():
        return _LazyStateRows(rows, row_dest)
    # --- the reference's consumer lines -------------------------------------
    max_size, ptr = 16, 12                               # the ring wraps
    ring = torch.zeros((max_size, 4), dtype=torch.float32)        # replay.py:38-47
    next_state = lazy()
    ind = (np.arange(0, len(next_state)) + ptr) % max_size         # replay.py:80
    ring[ind] = next_state.to('cpu', copy=True)                    # ddpg.py:205, replay.py:84
    assert np.array_equal(ring[ind].numpy(), want.numpy())
    assert np.array_equal(lazy().cpu().numpy(), want.numpy())      # `state.cpu().numpy()`
    assert np.array_equal(lazy().to(device='cpu', copy=True).numpy(), want.numpy())
    # --- accessors below the dispatcher -------------------------------------
    assert np.array_equal(lazy().numpy(), want.numpy())
    assert np.array_equal(np.asarray(lazy()), want.numpy())
    assert np.array_equal(np.array(lazy(), dtype=np.float64), want.double().numpy())
    assert lazy().tolist() == want.tolist()
    x = lazy()
    assert x.data_ptr() != 0 and x.data_ptr() == x.materialize().data_ptr()
    assert x.untyped_storage().nbytes() == want.untyped_storage().nbytes()
    assert x.storage_offset() == 0 and x.stride() == want.stride() and x.is_contiguous()
    assert torch.equal(torch.from_dlpack(lazy()), want)
    assert torch.equal(lazy().data, want)
    assert torch.equal(copy.deepcopy(lazy()), want)
    assert torch.equal(pickle.loads(pickle.dumps(lazy())), want)
    assert float(lazy()[2, 1].item()) == float(want[2, 1])
    # --- conversions a real tensor answers with `self` ----------------------
    for conv in (lambda t: t.float(), lambda t: t.contiguous(), lambda t: t.to('cpu'),
                 lambda t: t.to(torch.float32), lambda t: t.detach(), lambda t: t.cpu(),
                 lambda t: t.type(torch.float32), lambda t: t.clone()):
        got = conv(lazy())
        assert type(got) is torch.Tensor and torch.equal(got, want)
        assert np.array_equal(got.numpy(), want.numpy())
    # views and slices of it are plain tensors too
    assert type(lazy()[1:3]) is torch.Tensor and type(lazy().view(-1)) is torch.Tensor
    assert np.array_equal(lazy().T.contiguous().numpy(), want.T.numpy())


def test_policy_tiles_make_actions_independent_of_the_batch_shape(monkeypatch):
    """TTL_POLICY_TILE_ROWS: the networks run in tiles of a fixed row count, so a
    row's action does not depend on which other rows share its batch (what lets
    a sharded tracking run equal the one-process run bit for bit)."""
    import torch
    from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic
    torch.manual_seed(0)
    agent = SACActorCritic(27, 3, '64-64', torch.device('cpu'))
    x = torch.randn(1500, 27)
    with torch.no_grad():
        plain = agent.select_action(x, 0.0)
        monkeypatch.setenv('TTL_POLICY_TILE_ROWS', '512')
        whole = agent.select_action(x, 0.0)
        shard = agent.select_action(x[750:], 0.0)
        single = agent.select_action(x[1499:], 0.0)
    assert whole.shape == plain.shape and torch.allclose(whole, plain, atol=1e-6)
    assert torch.equal(whole[750:], shard) and torch.equal(whole[1499:], single)


def test_bench_refuses_unknown_legs():
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--legs', 'weak,nope'],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and 'unknown leg' in out.stderr


def test_header_is_plain_c99(tmp_path):
    """include/ttl_hip.h is the C ABI: it must compile as pedantic C99 on its
    own (no C++, no HIP, no torch types in any signature)."""
    import shutil
    import subprocess
    cc = shutil.which('gcc')
    if cc is None:
        pytest.skip('no gcc')
    src = tmp_path / 'hdr.c'
    src.write_text('#include "ttl_hip.h"\n'
                   'int main(void) { ttl_env_desc d; (void)d; return (int)sizeof(ttl_env *) * 0; }\n')
    out = subprocess.run([cc, '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic',
                          '-fsyntax-only', '-I', os.path.join(ROOT, 'include'), str(src)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_c_host_example_builds_against_the_library():
    """examples/ttl_track_c.c -- the C ABI driven from plain C, no Python / torch
    in the process -- compiles with gcc -Wall -Wextra and links against the
    in-tree libttl_hip.so (it runs under -m gpu, tests/test_runners.py)."""
    import subprocess
    from tracktolearn_amd.csrc import build as hip_build
    try:
        hip_build.find_hipcc()
    except RuntimeError:
        pytest.skip('no ROCm toolchain here')
    if not os.path.exists(hip_build.OUTPUT):
        hip_build.build()
    exe = hip_build.build_example(verbose=False)
    assert os.access(exe, os.X_OK)
    needed = subprocess.run(['readelf', '-d', exe], capture_output=True, text=True).stdout
    assert 'libttl_hip.so' in needed and 'libamdhip64' in needed
    assert 'torch' not in needed and 'python' not in needed
