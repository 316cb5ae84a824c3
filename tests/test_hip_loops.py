"""GPU tests of the callers of the env path: validation_episode, the training
episode (_episode) with the HBM replay ring, and the Tracker."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5
DEV = 'cuda:0'


def _env(D, N, K, *, noisy, reward, max_length=30.0, seed=3):
    from tracktolearn_amd.environments import (NoisyTrackingEnvironment,
                                               TrackingEnvironment)
    from tracktolearn_amd.utils.synthetic import (synthetic_seeds,
                                                  synthetic_subject)
    subject = synthetic_subject(D, 45, seed=1234, peaks=True,
                                affine_dtype=np.float32)
    dto = dict(n_dirs=K, theta=30.0, npv=1, binary_stopping_threshold=0.1,
               step_size=0.75, min_length=2.0, max_length=max_length,
               compute_reward=reward, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device(DEV),
               target_sh_order=8, noise=0.0, fa_map=None)
    cls = NoisyTrackingEnvironment if noisy else TrackingEnvironment
    env = cls(subject, 'testing', dto)
    env.seeds = synthetic_seeds(subject[1].data, N, seed=seed)
    return env, subject


def _oracle(env, subject, *, noisy, K, reward):
    from oracle import env_oracle as orc
    kw = dict(n_dirs=K, theta=30.0, step_size=env.step_size,
              max_nb_steps=env.max_nb_steps, mask_threshold=0.1,
              peaks=subject[3].data, compute_reward=reward,
              alignment_weighting=1.0)
    if noisy:
        return orc.OracleNoisyTrackingEnv(subject[0].data, subject[1].data,
                                          env.seeds, noise=0.0, **kw)
    return orc.OracleTrackingEnv(subject[0].data, subject[1].data, env.seeds, **kw)


class _Recorder:
    """Wraps an agent's select_action and keeps every action batch."""

    def __init__(self, agent):
        self.agent, self.actions = agent, []
        self._orig = agent.select_action
        agent.select_action = self

    def __call__(self, state, probabilistic=1.0):
        a = self._orig(state, probabilistic=probabilistic)
        self.actions.append(a.detach().cpu().numpy().copy())
        return a


def test_validation_episode_tracks_like_the_oracle():
    """A random-init SACAuto policy tracks 2048 streamlines with the noisy
    (tracking) env; replaying the recorded actions through the CPU oracle
    gives the same tractogram, bit for bit."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    from tracktolearn_amd.tracking.tracker import Tracker
    torch.manual_seed(0)
    N, K = 2048, 4
    env, subject = _env(20, N, K, noisy=True, reward=True)
    alg = SACAuto(env.get_state_size(), 3, '64-64', n_actors=N, rng=None,
                  device=torch.device(DEV))
    rec = _Recorder(alg.agent)
    tracker = Tracker(alg, n_actor=N, prob=0.0)
    tractogram, reward = tracker.track_and_validate(env)
    assert len(tractogram) == N and np.isfinite(reward)

    ref = _oracle(env, subject, noisy=True, K=K, reward=True)
    ref.reset(0, N)
    total = 0.0
    for a in rec.actions:
        _, r, _, _ = ref.step(a)
        total += float(np.sum(r))
        ref.harvest()
    assert len(ref.continue_idx) == 0
    lines, seeds, flags = ref.get_streamlines()
    assert np.array_equal(tractogram.data_per_streamline['flags'], flags)
    for got, want in zip(tractogram.streamlines, lines):
        assert np.array_equal(got, want)
    assert abs(reward - total) <= 1e-5 * max(1.0, abs(total))


def test_training_episode_fills_the_replay_ring_correctly():
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    from tracktolearn_amd.tracking.tracker import Tracker
    torch.manual_seed(1)
    np.random.seed(1)
    N, K = 1024, 4
    env, subject = _env(20, N, K, noisy=False, reward=True)
    W = env.get_state_size()
    alg = SACAuto(W, 3, '64-64', n_actors=N, batch_size=256, replay_size=50000,
                  rng=None, device=torch.device(DEV))
    alg.start_timesteps = N               # one step of pure collection first
    before = [p.detach().clone() for p in alg.agent.actor.parameters()]
    rec = _Recorder(alg.agent)
    tracker = Tracker(alg, n_actor=N)
    tractogram, losses, reward, factors = tracker.track_and_train(env)
    n_steps = len(rec.actions)
    n_transitions = sum(len(a) for a in rec.actions)
    assert len(tractogram) == N
    assert alg.t == 1 + n_transitions
    assert len(alg.replay_buffer) == n_transitions
    assert alg.total_it > 0
    assert any(not torch.equal(b, p.detach())
               for b, p in zip(before, alg.agent.actor.parameters()))
    assert len(factors['peaks_reward']) == n_steps

    # the transitions of the first steps, against the CPU oracle fed with the
    # same (stochastic) actions: rows aligned as (s, a, s', r, 1 - done)
    ref = _oracle(env, subject, noisy=False, K=K, reward=True)
    ref.seeds = np.asarray(env.initial_points)
    s_ref = ref.reset(0, N)
    ptr = 0
    buf = alg.replay_buffer
    total = 0.0
    for step, a in enumerate(rec.actions):
        n = len(a)
        ns_ref, r_ref, d_ref, _ = ref.step(a.copy())
        total += float(r_ref.sum())
        if step < 4:
            sl = slice(ptr, ptr + n)
            assert np.abs(buf.state[sl].cpu().numpy() - s_ref).max() <= TOL
            assert np.array_equal(buf.action[sl].cpu().numpy(), a)
            assert np.abs(buf.next_state[sl].cpu().numpy() - ns_ref).max() <= TOL
            assert np.abs(buf.reward[sl, 0].cpu().numpy() - r_ref).max() <= TOL
            assert np.array_equal(buf.not_done[sl, 0].cpu().numpy(),
                                  1.0 - d_ref.astype(np.float32))
        ptr += n
        s_ref, _ = ref.harvest()
    assert abs(reward - total) <= 1e-4 * max(1.0, abs(total))
    assert np.array_equal(env.flags, ref.flags)


def test_tracker_track_filters_and_converts():
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    from tracktolearn_amd.tracking.tracker import TckFile, Tracker, TrkFile
    from tracktolearn_amd.tractogram import streamline_length
    torch.manual_seed(2)
    N, K = 512, 4
    env, _ = _env(16, N, K, noisy=True, reward=False)
    alg = SACAuto(env.get_state_size(), 3, '32-32', n_actors=200, rng=None,
                  device=torch.device(DEV))
    for fmt in (TrkFile, TckFile):
        np.random.seed(5)
        tracker = Tracker(alg, n_actor=200, prob=0.0, min_length=3.0,
                          max_length=8.0, save_seeds=True)
        lazy = tracker.track(env, fmt)           # 3 batches: 200, 200, 112
        items = list(lazy)
        assert 0 < len(items) < N
        for it in items:
            s = it.streamline
            vox = s / 1.0 - 0.5 if fmt is TrkFile else s
            assert 3.0 - 1e-6 <= streamline_length(vox) <= 8.0 + 1e-6
            assert it.data_for_streamline['seeds'].shape == (3,)
        assert np.array_equal(lazy.affine_to_rasmm, env.affine_vox2rasmm)


def test_graphed_update_equals_eager_update():
    """SACAuto.update replayed from a HIP graph: bit-identical to the eager
    update run with the same (capturable) Adam arithmetic, and within 2e-5 of
    the default eager update after 6 steps (fixed gaussian draws so every path
    is deterministic)."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    dev = torch.device(DEV)
    W, B = 327, 512
    g = torch.Generator(device='cpu').manual_seed(0)
    batches = [[torch.randn(B, W, generator=g).to(dev),
                torch.tanh(torch.randn(B, 3, generator=g)).to(dev),
                torch.randn(B, W, generator=g).to(dev),
                torch.rand(B, generator=g).to(dev),
                (torch.rand(B, generator=g) > 0.2).float().to(dev)]
               for _ in range(3)]
    eps = [torch.randn(B, 3, generator=g).to(dev) for _ in range(2)]

    def make(capturable=False):
        torch.manual_seed(7)
        alg = SACAuto(W, 3, '128-128', n_actors=64, batch_size=B,
                      replay_size=1000, rng=None, device=dev)
        calls = {'i': 0}

        def noise(like):
            calls['i'] += 1
            return eps[calls['i'] % 2]
        alg.noise_fn = noise
        if capturable:
            for opt in alg._optimizers():
                for group in opt.param_groups:
                    group['capturable'] = True
        return alg

    def params(alg):
        return (list(alg.agent.actor.parameters()) +
                list(alg.agent.critic.parameters()) +
                list(alg.target.critic.parameters()) +
                list(alg.target.actor.parameters()) + [alg.log_alpha])
    eager, twin, graphed = make(), make(True), make()
    graphed.enable_graph(warmup=2)
    # the first graphed call runs the update warmup + 1 = 3 times
    for _ in range(3):
        eager.update(batches[0])
        twin.update(batches[0])
    graphed.update(batches[0])
    assert graphed.total_it == eager.total_it == 3
    for b in (batches[1], batches[2], batches[1]):
        eager.update(b)
        twin.update(b)
        graphed.update(b)
    for pe, pt, pg in zip(params(eager), params(twin), params(graphed)):
        assert torch.equal(pt, pg)
        assert torch.allclose(pe, pg, rtol=0, atol=2e-5)
    with pytest.raises(RuntimeError):
        graphed.update([b[:100] for b in batches[0]])


def test_oracle_reward_and_oracle_stopping(tmp_path):
    """Sparse oracle bonus (oracle_reward.py) and the oracle stopping
    criterion (stopping_criteria.py:85-154) on the GPU env, re-derived on the
    CPU from the downloaded streamlines with a float32 copy of the network
    (resampler checked against numpy in tests/test_oracle_net.py)."""
    from tracktolearn_amd.environments import TrackingEnvironment
    from ref_resample import resample_streamlines     # CPU re-derivation
    from tracktolearn_amd.oracles.oracle import OracleSingleton
    from tracktolearn_amd.oracles.transformer_oracle import (
        TransformerOracle, save_random_checkpoint)
    from tracktolearn_amd.utils.synthetic import (synthetic_seeds,
                                                  synthetic_subject)
    ck = save_random_checkpoint(str(tmp_path / 'o.ckpt'), n_head=2, n_layers=1,
                                seed=5)
    N, K = 1500, 4
    subject = synthetic_subject(24, 45, seed=1234, peaks=True)
    seeds = synthetic_seeds(subject[1].data, N, seed=2)

    def make_env(ckpt):
        OracleSingleton.reset()
        dto = dict(n_dirs=K, theta=60.0, npv=1, binary_stopping_threshold=0.1,
                   step_size=0.75, min_length=1.5, max_length=30.0,
                   compute_reward=True, alignment_weighting=1.0,
                   oracle_bonus=10.0, oracle_checkpoint=ckpt,
                   oracle_stopping_criterion=True, rng=np.random.RandomState(0),
                   device=torch.device(DEV), target_sh_order=8)
        env = TrackingEnvironment(subject, 'training', dto)
        env._oracle.batch_size = 512      # 1500 = 2 * 512 + 476: tail quirk live
        env.seeds = seeds
        return env

    def cpu_logits(model, lines):
        pts = torch.from_numpy(np.stack(lines))
        lengths = torch.full((len(lines),), pts.shape[1], dtype=torch.long)
        data = resample_streamlines(pts, lengths, 128)
        with torch.no_grad():
            p = model(data[:, 1:] - data[:, :-1]).double()
        return torch.log(p / (1 - p)).numpy()

    # calibrate the random network so that its scores straddle 0.5 on the
    # kind of streamlines this env produces
    env = make_env(ck)
    assert env.min_nb_steps == 2
    blob = torch.load(ck, map_location='cpu', weights_only=True)
    model = TransformerOracle.load_from_checkpoint(blob)
    state = env.reset(0, N)
    for step in range(8):
        state, _ = (env.step(env.scripted_actions(state, step, 4, 0.1)),
                    env.harvest())[1]
    idx = env.continue_idx
    hist = env.streamlines
    logits = cpu_logits(model, [hist[g, :env.length] for g in idx])
    blob['state_dict']['head.bias'] -= float(np.median(logits))
    ck2 = str(tmp_path / 'o2.ckpt')
    torch.save(blob, ck2)
    cpu_model = TransformerOracle.load_from_checkpoint(
        torch.load(ck2, map_location='cpu', weights_only=True))

    def cpu_scores(lines, n_rows):
        sc = 1.0 / (1.0 + np.exp(-cpu_logits(cpu_model, lines)))
        full = (n_rows // 512) * 512 if n_rows > 512 else n_rows
        sc[full:] = 0.0                       # unevaluated tail -> score 0
        return sc

    env = make_env(ck2)
    state = env.reset(0, N)
    saw_bonus = saw_oracle_stop = saw_oracle_keep = False
    for step in range(16):
        n = env._n_active
        if n == 0:
            break
        idx = env.continue_idx
        a = env.scripted_actions(state, step, seed=4, wobble=0.1)
        nstate, reward, done, info = env.step(a)
        L = env.length
        hist = env.streamlines
        flags = env.flags
        # --- oracle stopping: every active row once L > 5 * min_nb_steps
        if L > 5 * env.min_nb_steps:
            sc = cpu_scores([hist[g, :L] for g in idx], n)
            sure = np.abs(sc - 0.5) > 0.03
            got = (flags[idx] & 64) != 0
            assert np.array_equal(got[sure], (sc < 0.5)[sure])
            assert done[got].all()
            saw_oracle_stop |= bool(got.any())
            saw_oracle_keep |= bool((~done).any())
        else:
            assert not (flags[idx] & 64).any()
        # --- oracle reward on the rows that just stopped
        term = env._last_oracle_term
        if L > env.min_nb_steps and done.any():
            rows = np.nonzero(done)[0]
            sc = cpu_scores([hist[idx[r], :L] for r in rows], len(rows))
            sure = np.abs(sc - 0.5) > 0.03
            t = term.cpu().numpy()
            assert np.array_equal((t[rows] == 10.0)[sure], (sc > 0.5)[sure])
            assert (t[~done] == 0).all() and set(np.unique(t)) <= {0.0, 10.0}
            assert abs(info['reward_info']['oracle_reward'] - t.mean()) < 1e-12
            saw_bonus |= bool((t == 10.0).any())
        state, _ = env.harvest()
    assert saw_bonus and saw_oracle_stop and saw_oracle_keep
    OracleSingleton.reset()
