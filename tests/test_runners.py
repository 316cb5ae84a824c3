"""Entry-point tests.  The reference's own tests are three `--help` smoke
tests (tests/test_runners.py there); the GPU test below runs BASELINE.json
config 1 end to end: ttl_track.py on a 32^3 synthetic descoteaux07 order-8
fODF + WM mask with n_actor = 4096."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ttl_track_help():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'ttl_track.py'),
                          '--help'], capture_output=True, text=True)
    assert out.returncode == 0
    for word in ('in_odf', 'in_seed', 'in_mask', 'out_tractogram', '--n_actor',
                 '--npv', '--noise', '--binary_stopping_threshold', '--rng_seed',
                 '--sh_basis', '--compress', '--save_seeds', '--agent',
                 '--hyperparameters', '--min_length', '--max_length'):
        assert word in out.stdout


def _write_inputs(tmp_path, D=32, order=8):
    from tracktolearn_amd.io import nifti
    from tracktolearn_amd.utils.synthetic import synthetic_volumes
    sh, mask, _ = synthetic_volumes(D, (order + 1) * (order + 2) // 2, peaks=False)
    aff = np.diag([1.0, 1.0, 1.0, 1.0])
    aff[:3, 3] = [-16.0, -20.0, 5.0]
    paths = {k: str(tmp_path / f'{k}.nii.gz') for k in ('odf', 'seed', 'mask')}
    nifti.save(paths['odf'], sh, aff)
    nifti.save(paths['seed'], mask, aff)
    nifti.save(paths['mask'], mask, aff)
    return paths, aff


def _write_agent(tmp_path, input_size, hidden='64-64', n_dirs=4):
    import torch
    from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic
    torch.manual_seed(3)
    agent = SACActorCritic(input_size, 3, hidden, torch.device('cpu'))
    agent_dir = tmp_path / 'model'
    agent_dir.mkdir()
    agent.save(str(agent_dir), 'last_model_state')
    hp = {'algorithm': 'SACAuto', 'step_size': 0.75, 'voxel_size': '1.0',
          'max_angle': 30, 'hidden_dims': hidden, 'n_dirs': n_dirs,
          'target_sh_order': 8.0}
    hp_path = agent_dir / 'hyperparameters.json'
    hp_path.write_text(json.dumps(hp))
    return str(agent_dir), str(hp_path)


@pytest.mark.gpu
def test_ttl_track_config1_end_to_end(tmp_path):
    from tracktolearn_amd.io import streamlines as sio
    from tracktolearn_amd.runners import ttl_track
    from tracktolearn_amd.tractogram import streamline_length
    paths, aff = _write_inputs(tmp_path)
    # the shipped model's own hyperparameters.json (a data file of the
    # reference, models/hyperparameters.json: K = 100 previous directions,
    # W = 615, hidden 1024-1024-1024, voxel_size "0.9987237", step 0.75) with
    # random weights of that architecture (the trained .pth files are not part
    # of the reference tree)
    import shutil
    import torch
    from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic
    shipped = json.loads(open(os.path.join(ROOT, 'tests', 'golden',
                                           'reference_model_hyperparameters.json')).read())
    assert shipped['input_size'] == 7 * 45 + 3 * shipped['n_dirs'] == 615
    torch.manual_seed(3)
    agent = SACActorCritic(shipped['input_size'], 3, shipped['hidden_dims'],
                           torch.device('cpu'))
    agent_dir = tmp_path / 'model'
    agent_dir.mkdir()
    agent.save(str(agent_dir), 'last_model_state')
    hp = str(agent_dir / 'hyperparameters.json')
    shutil.copy(os.path.join(ROOT, 'tests', 'golden',
                             'reference_model_hyperparameters.json'), hp)
    agent_dir = str(agent_dir)
    out = str(tmp_path / 'out.trk')
    argv = [paths['odf'], paths['seed'], paths['mask'], out, '--agent', agent_dir,
            '--hyperparameters', hp, '--n_actor', '4096', '--npv', '1',
            '--min_length', '3', '--max_length', '40', '--save_seeds',
            '--rng_seed', '11']
    ttl_track.main(argv)
    tg, header = sio.load_trk(out)
    assert header['nb_streamlines'] == len(tg) > 100
    assert np.allclose(header['voxel_to_rasmm'], aff)
    lo = aff[:3, 3] - 2.0
    hi = aff[:3, 3] + 32.0 + 2.0
    for s in tg.streamlines[:500]:
        assert (s >= lo).all() and (s <= hi).all()
        assert 3.0 - 1e-3 <= streamline_length(s) <= 40.0 + 1e-3
    assert tg.data_per_streamline['seeds'].shape == (len(tg), 3)
    # refuses to overwrite without -f, writes .tck too
    with pytest.raises(SystemExit):
        ttl_track.main(argv)
    out2 = str(tmp_path / 'out.tck')
    ttl_track.main(argv[:3] + [out2] + argv[4:])
    tck, fields = sio.load_tck(out2)
    assert int(fields['count']) == len(tck) == len(tg)


@pytest.mark.gpu
def test_ttl_track_rescales_the_step_to_the_subject_voxel_size(tmp_path, monkeypatch):
    """runners/ttl_track.py:145-178 of the reference: an agent trained at
    another voxel size tracks with step_size * subject_voxel / training_voxel,
    and the *second* load_subject() re-derives step / max steps / neighbourhood
    radius from it (environments/env.py:196-212)."""
    from tracktolearn_amd.io import streamlines as sio
    from tracktolearn_amd.runners import ttl_track
    from tracktolearn_amd.tracking.tracker import Tracker
    paths, aff = _write_inputs(tmp_path, D=24)
    agent_dir, hp = _write_agent(tmp_path, 7 * 45 + 3 * 4)
    hyper = json.loads(open(hp).read())
    hyper['voxel_size'] = '2.0'            # trained at 2 mm, subject is 1 mm
    open(hp, 'w').write(json.dumps(hyper))
    seen = {}
    real_track = Tracker.track

    def spy(self, env, fmt):
        seen['env'] = env
        return real_track(self, env, fmt)
    monkeypatch.setattr(Tracker, 'track', spy)
    out = str(tmp_path / 'out.trk')
    ttl_track.main([paths['odf'], paths['seed'], paths['mask'], out, '--agent',
                    agent_dir, '--hyperparameters', hp, '--n_actor', '2000',
                    '--min_length', '2', '--max_length', '30', '--rng_seed', '3'])
    env = seen['env']
    assert env.step_size_mm == 0.375
    assert float(env.step_size) == 0.375            # 1 mm voxels
    assert env.max_nb_steps == int(30 / 0.375) == 80
    assert env.min_nb_steps == int(2 / 0.375)
    assert abs(float(env.add_neighborhood_vox) - 0.375) < 1e-7
    assert env._buf_streamlines.shape[1] == 81
    tg, _ = sio.load_trk(out)
    assert len(tg) > 50
    for s in tg.streamlines[:200]:
        seg = np.linalg.norm(np.diff(s, axis=0), axis=1)
        assert np.abs(seg - 0.375).max() < 1e-4


def test_shipped_hyperparameters_file_is_understood(tmp_path):
    """ttl_track.py's constructor reads the reference's own
    models/hyperparameters.json (string voxel size, float SH order, three
    hidden layers) as runners/ttl_track.py:73-101 of the reference does."""
    from tracktolearn_amd.runners.ttl_track import TrackToLearnTrack
    hp = os.path.join(ROOT, 'tests', 'golden', 'reference_model_hyperparameters.json')
    exp = TrackToLearnTrack(dict(
        in_odf='a', in_seed='b', in_mask='c', out_tractogram='o.trk', noise=0.0,
        binary_stopping_threshold=0.1, n_actor=4096, npv=1, min_length=10.,
        max_length=300., compress=None, sh_basis='descoteaux07', save_seeds=False,
        agent=str(tmp_path), hyperparameters=hp, rng_seed=1337))
    assert exp.algorithm == 'SACAuto' and exp.n_dirs == 100
    assert exp.hidden_dims == '1024-1024-1024' and exp.step_size == 0.75
    assert abs(float(exp.voxel_size) - 0.9987237) < 1e-9 and exp.theta == 30
    assert int(exp.target_sh_order) == 8


def test_set_sh_order_basis_orders_and_fullness():
    from tracktolearn_amd.datasets.utils import set_sh_order_basis
    sh = np.zeros((2, 2, 2, 28), np.float32)
    sh[..., :] = np.arange(28)
    up = set_sh_order_basis(sh, 'descoteaux07', target_order=8)
    assert up.shape[-1] == 45 and (up[..., 28:] == 0).all()
    down = set_sh_order_basis(np.zeros((2, 2, 2, 45), np.float32), 'descoteaux07',
                              target_order=6)
    assert down.shape[-1] == 28
    full = np.tile(np.arange(81, dtype=np.float32), (2, 2, 2, 1))
    even = set_sh_order_basis(full, 'descoteaux07', target_order=8)
    assert even.shape[-1] == 45 and even[0, 0, 0, 1] == 4.0   # l=2 starts at index 4


def test_tournier07_to_descoteaux07_conversion():
    """`--sh_basis tournier07` (TrackToLearn/datasets/utils.py:172-175, scilpy
    convert_sh_basis): the converted coefficients describe the same spherical
    function -- checked by evaluating both bases on the package's hemisphere
    -- and the conversion round-trips.  scilpy / dipy are absent, the basis
    definitions are the published ones: PARITY UNPINNED."""
    from tracktolearn_amd.datasets.utils import (convert_sh_basis,
                                                 set_sh_order_basis)
    from tracktolearn_amd.reconst.peaks import hemisphere, sh_to_sf_matrix
    verts, _ = hemisphere(3)
    rng = np.random.RandomState(0)
    for order in (4, 6, 8):
        n = (order + 1) * (order + 2) // 2
        c = rng.standard_normal((3, 2, 2, n)).astype(np.float32)
        for legacy_in in (True, False):
            for out_basis, legacy_out in (('descoteaux07', True),
                                          ('descoteaux07', False),
                                          ('tournier07', not legacy_in)):
                d = convert_sh_basis(c, 'tournier07', out_basis, legacy_in, legacy_out)
                assert d.dtype == np.float32 and d.shape == c.shape
                sf_in = c @ sh_to_sf_matrix(verts, order, 'tournier07', legacy_in)
                sf_out = d @ sh_to_sf_matrix(verts, order, out_basis, legacy_out)
                assert np.abs(sf_in - sf_out).max() <= 1e-6 * np.abs(sf_in).max()
                back = convert_sh_basis(d, out_basis, 'tournier07', legacy_out, legacy_in)
                assert np.abs(back - c).max() <= 1e-6
    # the zonal (m = 0) coefficients are common to every basis; legacy
    # tournier07 carries no sqrt(2) on the others
    c = np.zeros((1, 1, 1, 6), np.float32)
    c[..., 0], c[..., 3], c[..., 1], c[..., 5] = 1.0, 2.0, 3.0, 4.0
    d = convert_sh_basis(c, 'tournier07', 'descoteaux07')
    assert d[0, 0, 0, 0] == 1.0 and d[0, 0, 0, 3] == 2.0
    # (l=2, m=-2) tournier = Im Y_2^2 -> descoteaux (l=2, m=+2) = sqrt2 Im Y_2^2
    assert np.isclose(d[0, 0, 0, 5], 3.0 / np.sqrt(2.0))
    assert np.isclose(d[0, 0, 0, 1], 4.0 / np.sqrt(2.0))
    # through the entry point's helper: order change + basis change
    t = rng.standard_normal((2, 2, 2, 28)).astype(np.float32)
    out = set_sh_order_basis(t, 'tournier07', target_order=8)
    assert out.shape[-1] == 45 and (out[..., 28:] == 0).all()
    assert np.allclose(out[..., :28], convert_sh_basis(t, 'tournier07'))


def test_sac_auto_train_help():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'sac_auto_train.py'),
                          '--help'], capture_output=True, text=True)
    assert out.returncode == 0
    for word in ('path', 'experiment', 'id', 'dataset_file', '--n_actor',
                 '--hidden_dims', '--max_ep', '--log_interval', '--lr', '--gamma',
                 '--alpha', '--batch_size', '--replay_size', '--oracle_bonus',
                 '--n_dirs', '--npv', '--theta', '--step_size'):
        assert word in out.stdout


def test_ttl_track_from_hdf5_help():
    """The reference's own test for this entry point (tests/test_runners.py:
    9-14 there) is `--help`."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'ttl_track_from_hdf5.py'),
                          '--help'], capture_output=True, text=True)
    assert out.returncode == 0
    for word in ('path', 'experiment', 'id', 'dataset_file', 'agent', 'subject_id',
                 'hyperparameters', '--n_actor', '--npv', '--min_length',
                 '--max_length', '--noise', '--fa_map', '--oracle_checkpoint'):
        assert word in out.stdout


def _write_dataset(path, D=20):
    from tracktolearn_amd.datasets.SubjectDataset import write_npz_dataset
    from tracktolearn_amd.utils.synthetic import synthetic_volumes
    subs = {}
    for i, sid in enumerate(('sub-a', 'sub-b')):
        sh, mask, pk = synthetic_volumes(D, 45, seed=50 + i)
        aff = np.eye(4, dtype=np.float32)
        subs[sid] = {'input_volume': (sh, aff), 'peaks_volume': (pk, aff),
                     'tracking_volume': (mask, aff), 'seeding_volume': (mask, aff)}
    write_npz_dataset(path, {'training': subs})


def test_npz_dataset_layout(tmp_path):
    from tracktolearn_amd.datasets.SubjectDataset import SubjectDataset
    p = str(tmp_path / 'ds.npz')
    _write_dataset(p, D=8)
    ds = SubjectDataset(p, 'training')
    assert len(ds) == 2 and ds.subjects == ['sub-a', 'sub-b']
    sid, vol, tracking, seeding, peaks, reference = ds[1]
    assert sid == 'sub-b' and vol.data.dtype == np.float32
    assert vol.shape == (8, 8, 8, 45) and peaks.shape == (8, 8, 8, 15)
    assert vol.affine_vox2rasmm.dtype == np.float32     # float32 affine -> F32 mode
    assert reference['shape'] == (8, 8, 8)              # anat falls back to tracking


@pytest.mark.gpu
def test_sac_auto_train_config3_smoke(tmp_path):
    """sac_auto_train.py end to end on a two-subject synthetic dataset: two
    training episodes with validation, model + hyperparameters + tractogram
    written (BASELINE config 3 at toy size; no oracle)."""
    from tracktolearn_amd.trainers import sac_auto_train
    ds = str(tmp_path / 'ds.npz')
    _write_dataset(ds)
    exp = tmp_path / 'exp'
    sac_auto_train.main([
        str(exp), 'toy', 'run1', ds, '--max_ep', '2', '--log_interval', '1',
        '--n_actor', '512', '--hidden_dims', '32-32', '--batch_size', '64',
        '--replay_size', '20000', '--npv', '1', '--min_length', '2',
        '--max_length', '20', '--oracle_bonus', '0', '--oracle_checkpoint', '',
        '--rng_seed', '4'])
    model = exp / 'model'
    assert (model / 'last_model_state_actor.pth').exists()
    assert (model / 'last_model_state_critic.pth').exists()
    hp = json.loads((model / 'hyperparameters.json').read_text())
    assert hp['algorithm'] == 'SACAuto' and hp['input_size'] == 7 * 45 + 12
    assert hp['n_dirs'] == 4 and hp['target_sh_order'] == 8
    assert list(exp.glob('tractogram_toy_run1_*.trk'))
    assert (exp / 'plots' / 'train_reward.npy').exists()
    # the saved agent loads back into the tracking entry point's agent class
    import torch
    from tracktolearn_amd.algorithms.shared.offpolicy import SACActorCritic
    agent = SACActorCritic(hp['input_size'], 3, hp['hidden_dims'], torch.device('cpu'))
    agent.load(str(model), 'last_model_state')


@pytest.mark.gpu
def test_ttl_track_two_ranks_match_one_rank(tmp_path):
    """The sharded tracking path of config 4 rehearsed with 2 processes (both
    on cuda:0, gloo instead of RCCL): every seed batch is split over the
    ranks, rank 0 collates and writes.  Same streamlines as the 1-process run
    (same order: shards are contiguous and gathered in rank order)."""
    from tracktolearn_amd.io import streamlines as sio
    paths, aff = _write_inputs(tmp_path, D=24)
    agent_dir, hp = _write_agent(tmp_path, 7 * 45 + 3 * 4)
    common = [paths['odf'], paths['seed'], paths['mask']]
    opts = ['--agent', agent_dir, '--hyperparameters', hp, '--n_actor', '1500',
            '--min_length', '2', '--max_length', '40', '--save_seeds',
            '--rng_seed', '5']
    one, two = str(tmp_path / 'one.trk'), str(tmp_path / 'two.trk')
    script = os.path.join(ROOT, 'ttl_track.py')
    # the policy in fixed 512-row tiles: its GEMMs then have the same shape in
    # both runs, so a row's action does not depend on how many rows share its
    # batch and the sharded tractogram must EQUAL the one-process tractogram
    env = dict(os.environ, PYTHONPATH=ROOT, TTL_POLICY_TILE_ROWS='512')
    r1 = subprocess.run([sys.executable, script] + common + [one] + opts,
                        capture_output=True, text=True, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    env2 = dict(env, TTL_ONE_DEVICE='1', TTL_DIST_BACKEND='gloo')
    port = 29600 + os.getpid() % 300
    r2 = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
         str(port), script] + common + [two] + opts,
        capture_output=True, text=True, env=env2)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a, _ = sio.load_trk(one)
    b, _ = sio.load_trk(two)
    assert len(a) > 500
    # streamline for streamline, in the same order (shards are contiguous and
    # gathered in rank order), bit for bit
    assert len(a) == len(b)
    assert np.array_equal(a.data_per_streamline['seeds'], b.data_per_streamline['seeds'])
    for sa, sb in zip(a.streamlines, b.streamlines):
        assert sa.shape == sb.shape and np.array_equal(sa, sb)


@pytest.mark.gpu
def test_ttl_track_from_hdf5_end_to_end(tmp_path):
    """Tracks the requested subject of a (npz-layout) dataset file and writes
    the .tck the reference names `tractogram_<experiment>_<id>_<subject>.tck`."""
    from tracktolearn_amd.io import streamlines as sio
    from tracktolearn_amd.runners import ttl_track_from_hdf5
    ds = str(tmp_path / 'ds.npz')
    _write_dataset(ds)
    agent_dir, hp = _write_agent(tmp_path, 7 * 45 + 3 * 4)
    out = ttl_track_from_hdf5.main([
        str(tmp_path / 'exp'), 'toy', 'v1', ds, agent_dir, 'sub-b', hp,
        '--n_actor', '1000', '--npv', '1', '--min_length', '2',
        '--max_length', '20', '--rng_seed', '3'])
    assert out.endswith('tractogram_toy_v1_sub-b.tck')
    tck, fields = sio.load_tck(out)
    assert int(fields['count']) == len(tck) > 50


def _last_json_line(text):
    rows = [r for r in text.splitlines() if r.startswith('{') and '"metric"' in r]
    assert rows, text[-2000:]
    return json.loads(rows[-1])


def test_bench_self_launch_reports_failed_ranks():
    """`python bench.py --gpus 2` without a launcher starts its ranks itself;
    here (no GPU) the ranks fail, and the parent must come back with a
    non-zero exit code instead of a line or a hang."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                          '--windows', '1', '--no-cpu-baseline', '--no-whole-episode'],
                         capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES='',
                                  HIP_VISIBLE_DEVICES=''))
    assert out.returncode != 0
    assert '"metric"' not in out.stdout
    assert 'failed' in out.stderr


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """The N > 1 path of bench.py as far as a 1-GPU box can take it: the bare
    `--gpus 2` call launches two ranks itself (both on cuda:0, gloo instead of
    RCCL, which refuses two ranks on one device), shards, times the windows of
    every leg, gathers the finished tracts to rank 0 and prints ONE line that
    carries the weak-scaling value, the strong-scaling leg (262144 streamlines
    in total) and config 4 (145^3, sharded, collate included end to end).
    Config 4 is shrunk to 262144 streamlines in total for the rehearsal: the
    shard is then the 131072 rows one GPU holds at N = 8."""
    env = dict(os.environ, PYTHONPATH=ROOT, TTL_BENCH_ONE_DEVICE='1',
               TTL_BENCH_BACKEND='gloo', TTL_BENCH_C4_TOTAL='262144',
               TTL_BENCH_C5_TOTAL='16384')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                          '--windows', '3', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=1100, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _last_json_line(out.stdout)
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak'
    assert line['streamline_steps'] > 2 * 12 * 200000          # both shards counted
    assert line['collate_ms'] > 0 and line['collate_bytes_to_root'] > 0
    assert 'collate_error' not in line
    assert line['windows']['n'] == 3
    assert line['roofline']['frac'] is None or line['roofline']['frac'] <= 1.0
    # strong scaling: one global batch of 262144 seeds split over the two ranks
    strong = line['strong']
    assert strong['scaling'] == 'strong' and strong['n_actor_total'] == 262144
    assert strong['n_actor_per_gpu'] == 131072 and not strong['same_run_as_value']
    assert 12 * 200000 < strong['streamline_steps'] <= 12 * 262144
    assert strong['value'] > 0
    # config 4: sharded 145^3 run, the collate inside the end-to-end figure
    c4 = line['config4']
    assert c4['n_actor_total'] == 262144 and c4['n_actor_per_gpu'] == 131072
    assert c4['step_only']['value'] > 0
    e2e = c4['end_to_end']
    assert 'collate_error' not in e2e
    assert e2e['collate_ms'] > 0 and e2e['collate_bytes_to_root'] > 0
    assert e2e['end_to_end_ms'] >= e2e['track_ms'] + 0.5 * e2e['collate_ms']
    assert 0 < e2e['value_end_to_end'] < e2e['value_step_only']
    # every streamline of both shards was tracked to exhaustion
    assert e2e['streamline_steps'] > 262144 * 10
    assert line['roofline_hbm_regime']['units_per_launch'] > 100000
    assert line['cpu_baseline'].startswith('N=1 only') if 'cpu_baseline' in line else True
    assert e2e['collate_first_call_ms'] > 0         # the untimed warm-up collate
    assert line['other_shapes'] == 'N=1 only'       # the `shapes` leg is a one-GPU leg
    assert line['config3_training'] == 'N=1 only'   # and so is the `learner` leg
    # config 5: every rank trains on its shard, the learner is data-parallel -- the fused
    # update's gradient exchange (critics' arena beside the actor's backward) ran on both
    c5 = line['config5']
    assert 'error' not in c5, c5
    assert c5['n_actor'] == 8192 and c5['n_actor_total'] == 16384 and c5['data_parallel']
    assert c5['fused_learner'] and c5['phases_ms_per_step']['update_all_reduce'] > 0
    assert c5['value'] > 0 and line['config5_value'] == c5['value']


@pytest.mark.gpu
def test_bench_other_shapes_leg():
    """`bench.py --legs shapes`: the other BASELINE shapes in the line (one GPU),
    the reference's own calling contract among them."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--legs', 'shapes',
                          '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _last_json_line(out.stdout)
    assert 'config3_training' not in line           # only the legs that were asked for
    shapes = line['other_shapes']
    assert set(shapes) == {'c2_K100', 'c3_env', 'c1_shape', 'c2_host_contract'}
    for name, o in shapes.items():
        assert o['value'] > 0 and o['windows'] == 3, name
    assert shapes['c2_K100']['n_dirs'] == 100 and shapes['c3_env']['n_actor'] == 65536
    assert shapes['c2_host_contract']['loop'].startswith('step(numpy)')
    # PCIe inclusive: slower than the device-resident loop on the same shape
    assert shapes['c2_host_contract']['value'] < 1.3e9


@pytest.mark.gpu
def test_bench_learner_leg():
    """`bench.py --legs learner`: BASELINE config 3's training step (SAC, hidden
    1024-1024, n_actor 65536) in the line."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--legs', 'learner',
                          '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _last_json_line(out.stdout)
    t = line['config3_training']
    assert t['n_actor'] == 65536 and t['hidden'] == '1024-1024' and t['batch'] == 4096
    assert 0 < t['update_ms'] < t['train_step_ms']
    assert t['train_streamline_steps_per_s'] > 0
    assert t['fused_learner'] is True
    # the update's FLOP roofline and the per-phase brackets
    roof = t['roofline']
    assert roof['bound'] == 'mfma' and roof['unit'] == 'TFLOP/s' and roof['peak'] == 157.3
    assert abs(roof['flop_per_update_issued'] - 1.68e11) < 2e9
    assert 0.2 < roof['frac'] < 1.0
    assert abs(roof['achieved'] - roof['flop_per_update_issued'] / t['update_ms'] / 1e9) < 1e-6
    ph = t['phases_ms_per_step']
    assert set(ph) >= {'policy', 'env_step', 'replay_add', 'replay_sample', 'update', 'harvest'}
    assert 0.5 * t['train_step_ms'] < sum(ph[k] for k in ('policy', 'env_step', 'replay_add',
                                                          'replay_sample', 'update',
                                                          'harvest')) < 1.5 * t['train_step_ms']
    assert line['config3_update_ms'] == t['update_ms']       # short top-level scalars


@pytest.mark.gpu
def test_bench_config5_leg():
    """`bench.py --legs config5`: BASELINE config 5's training step (oracle
    bonus + oracle stopping) at one GPU's shard, with both oracle paths live
    and timed."""
    env = dict(os.environ, PYTHONPATH=ROOT, TTL_BENCH_C5_TOTAL='32768')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--legs', 'config5',
                          '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _last_json_line(out.stdout)
    c5 = line['config5']
    assert 'error' not in c5, c5
    assert c5['n_actor'] == 4096 and c5['policy'] == 'straight' and c5['value'] > 0
    ph = c5['phases_ms_per_step']
    # (the resampler is a phase of its own only with TTL_ORACLE_FAST=0: by default it
    # is fused with the history gather into ttl_oracle_segments, part of env_step)
    assert ph['oracle_transformer'] > 0 and 'oracle_resample' not in ph
    assert ph['env_step'] > ph['oracle_transformer']
    # the stopping criterion scores every active streamline of a step, the
    # bonus the ones that stopped: more rows scored than one batch per episode
    assert c5['oracle_rows_scored_per_step'] > 20
    assert c5['whole_batch_on_one_gpu']['n_actor'] == 32768
    assert 0 < c5['policy_forward']['frac'] < 1 and c5['policy_forward']['dtype'] == 'f32'
    # the network alone: the batch a step scores (one workgroup per streamline) and a large
    # one (one wavefront per streamline) with its fraction of the fp16 MFMA peak
    on = c5['oracle_net_alone']
    assert [b['rows'] for b in on['per_batch']] == [256, 16384]
    assert on['per_batch'][0]['kernel'].startswith('workgroup') and \
        on['per_batch'][1]['kernel'].startswith('wavefront')
    assert 0.05 < on['roofline']['frac'] < 1.0 and on['roofline']['dtype'] == 'f16'
    assert line['oracle_net_16384_ms'] == on['per_batch'][1]['fused_ms']
    assert line['config5_value'] == c5['value']


@pytest.mark.gpu
def test_c_host_example_tracks_an_episode():
    """The C ABI without Python in the process: examples/ttl_track_c (plain C99,
    hipMalloc-ed buffers) tracks 20 000 streamlines to exhaustion through the
    large-batch path (processing order, partly filled last workgroup) and
    checks segment lengths, flags, lengths and the reported survivor counts."""
    from tracktolearn_amd.csrc import build as hip_build
    exe = hip_build.EXAMPLE_BIN
    if not os.path.exists(exe):
        exe = hip_build.build_example(verbose=False)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith('ok: 20000 streamlines'), out.stdout
