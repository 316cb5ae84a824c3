"""GPU tests of the free-running step (ttl_env_freerun_*, ABI v7) and of the
graphed tracking loop built on it (TrackingEnvironment.run_free,
RLAlgorithm.validation_episode): the tractogram must be the one the CPU oracle
produces from the same actions, bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _env(D, N, K, *, noisy, reward, max_length=30.0, seed=3):
    from test_hip_loops import _env as make
    return make(D, N, K, noisy=noisy, reward=reward, max_length=max_length, seed=seed)


def _oracle(env, subject, *, noisy, K, reward):
    from test_hip_loops import _oracle as make
    return make(env, subject, noisy=noisy, K=K, reward=reward)


def _rowwise_policy(K):
    """Element-wise torch code only: every row's action has the same bits
    whatever the batch size, so the step-by-step loop, the graphed loop and the
    oracle replay can be compared exactly."""
    bias = torch.tensor([0.31, -0.22, 0.13], device=DEV)

    def policy(state):
        last_dir = state[:, -3 * K:-3 * K + 3]           # most recent segment (zeros at first)
        sh = state[:, 1:4]
        return last_dir * 0.9 + sh * 0.25 + bias
    return policy


def _replay_through_oracle(ref, actions_log, n):
    ref.reset(0, n)
    total = 0.0
    for a in actions_log:
        k = len(ref.continue_idx)
        assert k > 0
        _, r, _, _ = ref.step(a[:k])
        total += float(np.sum(r))
        ref.harvest()
    assert len(ref.continue_idx) == 0
    return total


@pytest.mark.parametrize('noisy,reward', [(False, False), (True, True)])
def test_graphed_episode_equals_the_oracle(noisy, reward):
    """run_free (policy + step captured in one HIP graph, replayed until the
    pinned survivor count reads zero) against the oracle fed with the recorded
    action batches: flags, lengths and points bit for bit, summed reward."""
    N, K = 3000, 4
    env, subject = _env(20, N, K, noisy=noisy, reward=reward)
    policy = _rowwise_policy(K)
    for rep in range(2):            # the second run replays the cached graph
        state = env.reset(0, N)
        assert env.freerun_supported()
        got_reward, n_steps, log = env.run_free(policy, state, key='rowwise',
                                                record_actions=True)
        assert env._n_active == 0 and n_steps == len(log) >= 2
        assert env.length == 1 + n_steps
        tract = env.get_streamlines()
        ref = _oracle(env, subject, noisy=noisy, K=K, reward=reward)
        total = _replay_through_oracle(ref, log.cpu().numpy(), N)
        lines, _, flags = ref.get_streamlines()
        assert np.array_equal(tract.data_per_streamline['flags'], flags)
        assert np.array_equal(env.lengths, ref.lengths)
        for a, b in zip(tract.streamlines, lines):
            assert np.array_equal(a, b)
        if reward:
            assert abs(float(got_reward) - total) <= 1e-9 * max(1.0, abs(total))
        else:
            assert got_reward is None
    assert len(env._free_runs) == 1


def test_graphed_loop_equals_the_step_by_step_loop():
    """The same row-wise policy through step_device()/harvest() and through the
    graph: identical tractograms (the env arithmetic does not depend on how many
    stale rows ride along)."""
    N, K = 4096, 4
    env, _ = _env(24, N, K, noisy=True, reward=False)
    policy = _rowwise_policy(K)
    state = env.reset(0, N)
    while state.shape[0] > 0:
        env.step_device(policy(state))
        state, _ = env.harvest()
    eager = env.get_streamlines()
    eager_len = env.lengths.copy()
    state = env.reset(0, N)
    env.run_free(policy, state)
    graphed = env.get_streamlines()
    assert np.array_equal(eager_len, env.lengths)
    assert np.array_equal(eager.data_per_streamline['flags'],
                          graphed.data_per_streamline['flags'])
    for a, b in zip(eager.streamlines, graphed.streamlines):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('noisy,reward', [(False, True), (True, False)])
def test_eager_free_running_episode_equals_the_oracle(noisy, reward):
    """run_free_eager: policy + free-running step launched for the newest
    reported survivor count, the host never waiting for a step.  The recorded
    action batches (of whatever row count the host happened to use) replayed
    through the oracle give the same tractogram, bit for bit."""
    N, K = 5000, 4
    env, subject = _env(24, N, K, noisy=noisy, reward=reward)
    inner = _rowwise_policy(K)
    log = []

    def policy(state):
        a = inner(state)
        log.append(a.clone())
        return a
    state = env.reset(0, N)
    got_reward, n_steps = env.run_free_eager(policy, state)
    assert env._n_active == 0 and env.length == 1 + n_steps
    caps = [len(a) for a in log]
    assert caps[0] == N and caps == sorted(caps, reverse=True) and caps[-1] < N // 4
    tract = env.get_streamlines()
    ref = _oracle(env, subject, noisy=noisy, K=K, reward=reward)
    total = _replay_through_oracle(ref, [a.cpu().numpy() for a in log[:n_steps]], N)
    lines, _, flags = ref.get_streamlines()
    assert np.array_equal(tract.data_per_streamline['flags'], flags)
    assert np.array_equal(env.lengths, ref.lengths)
    for a, b in zip(tract.streamlines, lines):
        assert np.array_equal(a, b)
    if reward:
        assert abs(float(got_reward) - total) <= 1e-9 * max(1.0, abs(total))


def test_validation_episode_takes_the_graph_with_a_network(monkeypatch):
    """Tracker.track_and_validate with a SACAuto policy: the graphed loop is
    taken (default) and tracks like the step-by-step loop (TTL_GRAPH_EPISODE=0).
    The MLP's GEMMs may round differently for different batch sizes, so a few
    streamlines may part ways: at least 95 % must have the same length."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    from tracktolearn_amd.tracking.tracker import Tracker
    torch.manual_seed(0)
    N, K = 2500, 4
    env, _ = _env(20, 2 * N + 100, K, noisy=True, reward=True)
    alg = SACAuto(env.get_state_size(), 3, '64-64', n_actors=N, rng=None,
                  device=torch.device(DEV))
    tracker = Tracker(alg, n_actor=N, prob=0.0)
    monkeypatch.setattr(type(alg), 'graph_policy_us', 1e9)    # whatever the policy costs
    tract_g, reward_g = tracker.track_and_validate(env)
    assert all(fr.graph is not None for fr in env._free_runs.values())
    assert len(env._free_runs) == 2          # full batches and the last short one
    monkeypatch.setenv('TTL_GRAPH_EPISODE', '0')
    tract_e, reward_e = tracker.track_and_validate(env)
    assert len(tract_g) == len(tract_e) == 2 * N + 100
    len_g = np.array([len(s) for s in tract_g.streamlines])
    len_e = np.array([len(s) for s in tract_e.streamlines])
    same = np.mean(len_g == len_e)
    assert same >= 0.95, same
    assert abs(reward_g - reward_e) <= 0.05 * max(1.0, abs(reward_e))


def test_free_running_steps_hand_the_episode_back():
    """Raw C ABI: three free-running steps launched eagerly (no graph), then
    ttl_env_freerun_end and the ordinary step()/harvest() loop to the end; the
    result equals the oracle's.  Also the call-order errors."""
    from tracktolearn_amd import _lib
    N, K = 1500, 4
    env, subject = _env(20, N, K, noisy=False, reward=True)
    lib, h = env._lib, env._handle
    policy = _rowwise_policy(K)
    state = env.reset(0, N)
    lib, h = env._lib, env._handle
    done = torch.empty(N, dtype=torch.uint8, device=DEV)
    reward = torch.empty(N, dtype=torch.float64, device=DEV)
    buf = env._new_state(N)
    buf[:N].copy_(state)
    a = policy(buf).contiguous()
    step_args = (h, a.data_ptr(), N, buf.data_ptr(), env._state_pitch, reward.data_ptr(),
                 done.data_ptr(), env._stream())
    assert lib.ttl_env_freerun_step(*step_args) == _lib.ERR_STATE        # begin first
    assert lib.ttl_env_freerun_end(h, None, None, None, env._stream()) == _lib.ERR_STATE
    _lib.check(lib.ttl_env_freerun_begin(h, env._host_counts.data_ptr(), env._stream()))
    assert lib.ttl_env_freerun_begin(h, None, env._stream()) == _lib.ERR_STATE
    with pytest.raises(_lib.TTLError):
        env.step_device(a)                    # the handle is free-running
    assert lib.ttl_env_freerun_step(h, a.data_ptr(), N + 1, buf.data_ptr(), env._state_pitch,
                                    reward.data_ptr(), done.data_ptr(),
                                    env._stream()) == _lib.ERR_INVALID
    # a launch that does not cover the active rows only counts itself
    _lib.check(lib.ttl_env_freerun_step(h, a.data_ptr(), N - 7, buf.data_ptr(), env._state_pitch,
                                        reward.data_ptr(), done.data_ptr(), env._stream()))
    torch.cuda.synchronize()
    assert env._host_counts_np[:3].tolist() == [N, 0, 1]
    log, rewards = [], 0.0
    for _ in range(3):
        a = policy(buf).contiguous()
        log.append(a.cpu().numpy())
        _lib.check(lib.ttl_env_freerun_step(h, a.data_ptr(), N, buf.data_ptr(), env._state_pitch,
                                            reward.data_ptr(), done.data_ptr(), env._stream()))
        rewards += float(reward.sum())
    n_left, length, steps = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(lib.ttl_env_freerun_end(h, C.byref(n_left), C.byref(length), C.byref(steps),
                                       env._stream()))
    assert steps.value == 4 and length.value == 4        # the skipped launch counted itself
    assert int(env._host_counts_np[0]) == n_left.value and int(env._host_counts_np[2]) == 4
    # hand back to the host loop
    env.length, env._n_active = length.value, n_left.value
    env._cur ^= 1                                  # three steps: odd
    state = buf[:n_left.value]
    while state.shape[0] > 0:
        a = policy(state)
        log.append(a.cpu().numpy())
        _, r, _, _ = env.step_device(a)
        rewards += float(r.sum())
        state, _ = env.harvest()
    tract = env.get_streamlines()
    ref = _oracle(env, subject, noisy=False, K=K, reward=True)
    total = _replay_through_oracle(ref, log, N)
    lines, _, flags = ref.get_streamlines()
    assert np.array_equal(tract.data_per_streamline['flags'], flags)
    for x, y in zip(tract.streamlines, lines):
        assert np.array_equal(x, y)
    assert abs(rewards - total) <= 1e-9 * max(1.0, abs(total))


def test_expensive_policies_run_free_without_a_graph(monkeypatch):
    """run_free times the policy once on the full batch; above max_policy_us it
    declines (None) and validation_episode keeps its step-by-step loop -- or,
    with TTL_FREE_RUNNING_EAGER=1, launches policy + free-running step itself on
    batches that shrink with the reported survivor count."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    N, K = 1200, 4
    env, _ = _env(20, N, K, noisy=True, reward=False)
    alg = SACAuto(env.get_state_size(), 3, '64-64', n_actors=N, rng=None,
                  device=torch.device(DEV))
    monkeypatch.setattr(type(alg), 'graph_policy_us', 0.0)
    state = env.reset(0, N)
    assert alg._can_run_free(env)
    alg.validation_episode(state, env, 0.0)              # default: step by step
    assert env._n_active == 0 and env.length > 2
    (fr,) = env._free_runs.values()
    assert fr.graph is None and fr.policy_us > 0.0
    assert not env._free_bufs
    lengths = env.lengths.copy()
    monkeypatch.setenv('TTL_FREE_RUNNING_EAGER', '1')   # opt-in: never wait for a step
    state = env.reset(0, N)
    alg.validation_episode(state, env, 0.0)
    assert list(env._free_bufs) == [N]
    assert np.mean(env.lengths == lengths) >= 0.95


def test_free_running_refuses_large_batches_and_noise():
    from tracktolearn_amd import _lib
    env, _ = _env(24, 20000, 4, noisy=True, reward=False)
    env.reset(0, 20000)
    assert not env.freerun_supported()
    rc = env._lib.ttl_env_freerun_begin(env._handle, None, env._stream())
    assert rc == _lib.ERR_UNSUPPORTED
    env.reset(0, 1000)
    assert env.freerun_supported()
    env.noise = 0.1
    assert not env.freerun_supported()


def test_scripted_policy_episode_with_a_free_running_tail():
    """bench.py's whole-episode loop: step by step down to 16 384 rows, then
    free-running steps with the scripted policy reading row count and step
    number on the device.  Same tractogram as the all-step-by-step episode."""
    N, K = 20000, 4
    env, _ = _env(28, N, K, noisy=False, reward=False, max_length=40.0)

    def episode(free_tail):
        state = env.reset(0, N)
        step, free_steps = 0, 0
        while env._n_active:
            if free_tail and env.freerun_supported():
                _, free_steps = env.run_free_eager(
                    lambda st: env.scripted_actions_free(st, 7, 0.05), state)
                break
            env.step_device(env.scripted_actions(state, step, 7, 0.05))
            state, _ = env.harvest()
            step += 1
        return step, free_steps, env.lengths.copy(), env.flags.copy(), env.streamlines.copy()

    s0, f0, len0, fl0, pts0 = episode(False)
    s1, f1, len1, fl1, pts1 = episode(True)
    assert f0 == 0 and f1 > 0 and s1 >= 1 and s1 + f1 == s0
    assert np.array_equal(len0, len1) and np.array_equal(fl0, fl1)
    assert np.array_equal(pts0, pts1)


@pytest.mark.parametrize('hidden', ['64-64', '512-512-512'])
def test_default_tracking_loops_are_reproducible_bit_for_bit(hidden):
    """Two runs of Tracker.track_and_validate with the same weights give the same
    tractogram, whichever default loop the policy's cost selects (the graph for
    launch-bound policies, step by step otherwise)."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    from tracktolearn_amd.tracking.tracker import Tracker
    torch.manual_seed(0)
    N = 3000
    env, _ = _env(20, N, 4, noisy=True, reward=True)
    alg = SACAuto(env.get_state_size(), 3, hidden, n_actors=N, rng=None,
                  device=torch.device(DEV))
    tracker = Tracker(alg, n_actor=N, prob=0.0)
    first, r1 = tracker.track_and_validate(env)
    second, r2 = tracker.track_and_validate(env)
    assert r1 == r2
    assert np.array_equal(first.data_per_streamline['flags'], second.data_per_streamline['flags'])
    for a, b in zip(first.streamlines, second.streamlines):
        assert np.array_equal(a, b)
