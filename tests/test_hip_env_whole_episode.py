"""Whole episodes at the headline sizes against the CPU oracle
(TrackToLearn/environments/tracking_env.py:135-245 restated in
oracle/env_oracle.py), with the knobs bench.py runs: the one-launch step tail
(`k_tail`, holes in the processing order), the order refresh every 16 steps,
the early refresh at 20 % holes and the hand-over to free-running steps at
16 384 rows.

Streamlines are independent, so the oracle follows a random sample of 4 096 of
them; it draws its own actions from the numpy twin of the scripted policy
(oracle/scripted_policy.py: same counter-based hash, keyed by the GLOBAL
streamline id), which are asserted to be the GPU's bit for bit on every
stepped step.  Every step: dones, membership of continue_idx, flags, and the
sampled state rows within 1e-5; at the end: lengths, flags and every point of
the sampled streamlines bit-identical."""
import numpy as np
import pytest
import torch

from test_hip_env_fullsize import _make

pytestmark = pytest.mark.gpu

TOL = 1e-5
SAMPLE = 4096


def _sampled_oracle(env, subject, sample, *, noisy, K):
    from oracle import env_oracle as orc
    kw = dict(n_dirs=K, theta=30.0, step_size=env.step_size, max_nb_steps=env.max_nb_steps,
              mask_threshold=0.1, peaks=None, compute_reward=False, alignment_weighting=1.0,
              spline_eval='scipy')
    if noisy:
        return orc.OracleNoisyTrackingEnv(subject[0].data, subject[1].data, env.seeds[sample],
                                          noise=0.0, **kw)
    return orc.OracleTrackingEnv(subject[0].data, subject[1].data, env.seeds[sample], **kw)


def _follow(env, ref, sample, N, K, seed, wobble, max_stepped, free_tail):
    """Track env (all N rows, bench.py's loop) and ref (the sample) side by side.
    Returns (stepped steps compared, free-running steps, order refreshes seen)."""
    from oracle.scripted_policy import scripted_actions
    dir_off = 7 * 45
    sample_dev = torch.from_numpy(sample).cuda()
    state = env.reset(0, N)
    s_ref = ref.reset(0, len(sample))
    assert np.abs(state[sample_dev].cpu().numpy() - s_ref).max() <= TOL
    step, free_steps = 0, 0
    while env._n_active:
        if free_tail and env.freerun_supported():
            # bench.py's hand-over: the rest of the episode on free-running steps
            _, free_steps = env.run_free_eager(
                lambda st: env.scripted_actions_free(st, seed, wobble), state)
            break
        if step >= max_stepped:
            break
        ids = env.continue_idx                                   # global ids, ascending
        want_ids = sample[ref.continue_idx]
        rows = np.searchsorted(ids, want_ids)
        assert np.array_equal(ids[rows], want_ids), (step, 'continue_idx membership')
        assert len(np.intersect1d(ids, sample)) == len(want_ids), (step, 'extra survivors')
        rows_dev = torch.from_numpy(rows).cuda()
        a = env.scripted_actions(state, step, seed, wobble)
        a_ref = scripted_actions(s_ref, dir_off, want_ids, seed, step, wobble)
        assert np.array_equal(a[rows_dev].cpu().numpy(), a_ref), (step, 'actions')
        nstate, _, done, info = env.step_device(a)
        ns_ref, _, d_ref, _ = ref.step(a_ref.copy())
        dest = info['row_dest'].long()
        assert np.array_equal(done[rows_dev].cpu().numpy().astype(bool), d_ref), (step, 'dones')
        got = nstate[dest[rows_dev]].cpu().numpy()
        assert np.abs(got - ns_ref).max() <= TOL, (step, 'state rows')
        state, _ = env.harvest()
        s_ref, _ = ref.harvest()
        assert np.array_equal(env._buf_flags[sample_dev].cpu().numpy(), ref.flags), (step, 'flags')
        step += 1
    # the oracle finishes the episode on its own (same policy, same ids)
    ref_steps = step
    if free_steps:
        while len(ref.continue_idx):
            a_ref = scripted_actions(s_ref, dir_off, sample[ref.continue_idx], seed, ref_steps,
                                     wobble)
            ref.step(a_ref)
            s_ref, _ = ref.harvest()
            ref_steps += 1
    return step, free_steps, ref_steps


def _compare_final(env, ref, sample, n_points):
    sel = torch.from_numpy(sample).cuda()
    assert np.array_equal(env.lengths[sample], ref.lengths)
    assert np.array_equal(env.flags[sample], ref.flags)
    assert np.array_equal(env._buf_streamlines[sel, :n_points].cpu().numpy(),
                          ref.streamlines[:, :n_points])             # bit-identical points


def test_config2_whole_episode_follows_the_oracle_to_exhaustion():
    """262 144 rows, 96^3, K = 4, float32 directions, the knobs and the loop of
    bench.py's headline (`track_to_exhaustion`), to exhaustion."""
    import bench
    N, K, seed = 262144, 4, 1
    env, subject = _make(96, N, K, noisy=False, reward=False, max_length=200.0)
    assert env.SPATIAL_ORDER_REFRESH == 16 and env.TAIL_FUSED_MAX_ROWS >= N   # bench's defaults
    rng = np.random.RandomState(41)
    sample = np.sort(rng.choice(N, SAMPLE, replace=False))
    ref = _sampled_oracle(env, subject, sample, noisy=False, K=K)
    stepped, free_steps, ref_steps = _follow(env, ref, sample, N, K, seed, bench.WOBBLE,
                                             max_stepped=10 ** 6, free_tail=True)
    assert env._n_active == 0 and len(ref.continue_idx) == 0
    # the machinery that only switches on later was live: several order
    # refreshes, holes in the order, the free-running tail
    assert stepped > 3 * env.SPATIAL_ORDER_REFRESH and free_steps > 0
    assert stepped + free_steps >= ref_steps           # the GPU tracked at least as long
    assert bool(env.dones.all()) and (env.flags != 0).all()
    L = int(env.lengths.max())
    assert L == int(env.length) or L <= env.max_nb_steps
    _compare_final(env, ref, sample, L)
    print(f'config 2 whole episode: {stepped} stepped + {free_steps} free-running steps, '
          f'sample of {SAMPLE}: lengths, flags and points identical')
    # the loop bench.py times (no per-step host reads) gives the same tractogram
    flags, lengths = env.flags.copy(), env.lengths.copy()
    hist = env._buf_streamlines[torch.from_numpy(sample).cuda(), :L].clone()
    steps_b, free_b = bench.track_to_exhaustion(env, env.reset(0, N), seed, True)
    assert np.array_equal(env.flags, flags) and np.array_equal(env.lengths, lengths)
    assert torch.equal(env._buf_streamlines[torch.from_numpy(sample).cuda(), :L], hist)
    assert steps_b >= stepped


def test_config4_shard_forty_eight_steps_follow_the_oracle():
    """One GPU's shard of config 4: 131 072 rows on the 145^3 volume, K = 100,
    float64 directions (NoisyTrackingEnvironment, sigma 0), 48 steps."""
    import bench
    N, K, seed = 131072, 100, 2
    env, subject = _make(145, N, K, noisy=True, reward=False, max_length=300.0,
                         affine=np.float64)
    rng = np.random.RandomState(43)
    sample = np.sort(rng.choice(N, SAMPLE, replace=False))
    ref = _sampled_oracle(env, subject, sample, noisy=True, K=K)
    stepped, _, _ = _follow(env, ref, sample, N, K, seed, bench.WOBBLE, max_stepped=48,
                            free_tail=False)
    assert stepped == 48 and 0 < env._n_active < N
    _compare_final(env, ref, sample, env.length)
    print(f'config 4 shard: 48 steps, {env._n_active} of {N} still active, sample identical')


def test_one_launch_tail_at_its_cap_of_4096_row_blocks(monkeypatch):
    """`k_tail` at the largest order it accepts (1 048 576 slots = 4 096 row
    blocks of 256, 16 counts per thread of its scan): a few steps at N =
    1 048 576 with TTL_TAIL_FUSED_MAX_ROWS = 1 048 576 against the two-kernel
    tail (TTL_TAIL_FUSED = 0) on the same seeds and actions -- survivors, row
    map, a sample of state rows, flags, lengths and points all equal -- and a
    sample of 4 096 streamlines against the oracle."""
    import bench
    from tracktolearn_amd.environments import TrackingEnvironment
    N, K, seed, n_steps = 1 << 20, 4, 5, 5
    rng = np.random.RandomState(47)
    sample = np.sort(rng.choice(N, SAMPLE, replace=False))
    sample_dev = torch.from_numpy(sample).cuda()
    runs = {}
    for tail in ('1', '0'):
        monkeypatch.setenv('TTL_TAIL_FUSED', tail)
        monkeypatch.setenv('TTL_TAIL_FUSED_MAX_ROWS', str(N))
        monkeypatch.setattr(TrackingEnvironment, 'TAIL_FUSED_MAX_ROWS', N if tail == '1' else 0)
        env, subject = _make(96, N, K, noisy=False, reward=False, max_length=200.0)
        state = env.reset(0, N)
        log = []
        for step in range(n_steps):
            a = env.scripted_actions(state, step, seed, bench.WOBBLE)
            nstate, _, done, info = env.step_device(a)
            ids = env._idx_view(env._n_active).clone()
            dest = info['row_dest'].long()
            rows = torch.arange(0, env._n_active, 257, device='cuda')
            log.append((ids, done.clone(), dest.clone(), nstate[dest[rows]].clone()))
            state, _ = env.harvest()
        runs[tail] = (log, env.flags, env.lengths,
                      env._buf_streamlines[sample_dev, :env.length].clone(), env._n_active)
        if tail == '1':
            # ... and the one-launch tail against the oracle on the sample
            ref = _sampled_oracle(env, subject, sample, noisy=False, K=K)
            from oracle.scripted_policy import scripted_actions
            s_ref = ref.reset(0, SAMPLE)
            for step in range(n_steps):
                a_ref = scripted_actions(s_ref, 7 * 45, sample[ref.continue_idx], seed, step,
                                         bench.WOBBLE)
                ref.step(a_ref)
                s_ref, _ = ref.harvest()
            _compare_final(env, ref, sample, env.length)
            assert np.array_equal(np.intersect1d(env.continue_idx, sample),
                                  sample[ref.continue_idx])
        del env
        torch.cuda.empty_cache()
    (log1, f1, l1, h1, n1), (log0, f0, l0, h0, n0) = runs['1'], runs['0']
    assert n1 == n0 and 0 < n1 < N
    for step, (x, y) in enumerate(zip(log1, log0)):
        for u, v, what in zip(x, y, ('continue_idx', 'done', 'row_dest', 'state rows')):
            assert torch.equal(u, v), (step, what)
    assert np.array_equal(f1, f0) and np.array_equal(l1, l0) and torch.equal(h1, h0)
