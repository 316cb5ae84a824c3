#!/usr/bin/env python3
"""SAC training steps of BASELINE configs[2] ("config 3") and configs[4]
("config 5") timed phase by phase with HIP events (bench.py legs `learner`
and `config5` call `measure`; standalone: one JSON object on stdout).

  config 3  sac_auto_train.py: SAC (automatic entropy), hidden 1024-1024,
            n_actor 65536, batch 4096, 96^3 x 45 volume, n_dirs 4, alignment
            reward, no oracle.
  config 5  the same learner with oracle_bonus 10 and the oracle stopping
            criterion on (TractOracle-Net transformer, random-init weights of
            the checkpoint's architecture -- the shipped checkpoint is absent
            offline), n_actor 131072 in total: one GPU's shard (16384 at
            N = 8) per rank, learner replicas data-parallel
            (`enable_data_parallel`: one all-reduce per parameter arena and
            update).

One training step = policy forward (sample_action) -> env.step_device (advance,
stopping tests, alignment reward [, k_resample -> transformer -> sparse bonus,
k_resample -> transformer -> k_restop], state gather) -> replay add -> replay
sample -> SACAuto.update -> harvest (TrackToLearn/algorithms/ddpg.py:141-232).
Timed twice: plain wall clock over `steps` steps (the figure), then once more
with every phase bracketed by events on the launch stream (the breakdown; the
records themselves cost a few us each, so the phases add up to slightly more
than the plain step).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


class _Phases:
    """Sums of event-bracketed GPU time per phase name."""

    def __init__(self):
        self.pairs = {}

    def span(self, name):
        return _Span(self, name)

    def totals_ms(self):
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.pairs.items()}


class _Span:
    def __init__(self, owner, name):
        self.owner, self.name = owner, name

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.owner.pairs.setdefault(self.name, []).append((self.a, self.b))


def _instrument_oracle(env, phases):
    """Bracket the oracle's resampler and network (OracleSingleton.predict)."""
    orc = env._oracle
    if orc is None:
        return lambda: None
    resample, forward = orc._resample, orc.model.forward
    rows = {'scored': 0, 'batches': 0}

    def timed_resample(points, lengths, nb):
        with phases.span('oracle_resample'):
            out = resample(points, lengths, nb)
        rows['scored'] += int(points.shape[0])
        rows['batches'] += 1
        return out

    def timed_forward(x):
        with phases.span('oracle_transformer'):
            return forward(x)
    net = orc.net                     # the fused kernel (oracles/fused_net.py), if in use

    def timed_net(x):
        # (with the oracle path as library calls -- env.oracle_fast -- the resampler
        # above is not called: ttl_oracle_segments is part of the env step's rest)
        if env._oracle_fast():
            rows['scored'] += int(x.shape[0])
            rows['batches'] += 1
        with phases.span('oracle_transformer'):
            return net(x)
    orc._resample, orc.model.forward = timed_resample, timed_forward
    if net is not None:
        orc.net = timed_net

    def restore():
        orc._resample, orc.model.forward, orc.net = resample, forward, net
        return rows
    return restore


def straight_policy_(actor, dir_offset, log_std=-3.0, rest=0.05):
    """Give a freshly initialised MaxEntropyActor the behaviour of a policy
    that has learned the first thing every tracking policy learns: keep going.
    A skip path through the first six units of every hidden layer carries the
    newest direction of the state (columns dir_offset..+2, positive and
    negative parts through the ReLUs) to the mean; the other weights of the
    head keep their random initialisation scaled by `rest`, the log-std head
    starts at `log_std`.  A random-init policy ends every streamline within a
    few steps (curvature), so the oracle -- which only scores streamlines longer
    than min_nb_steps -- would never run."""
    lins = [m for m in actor.layers if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        w0 = lins[0].weight
        w0[:6].zero_()
        lins[0].bias[:6].zero_()
        for i in range(3):
            w0[i, dir_offset + i] = 1.0
            w0[3 + i, dir_offset + i] = -1.0
        for lin in lins[1:-1]:
            lin.weight[:6].zero_()
            lin.weight[:, :6].zero_()
            lin.bias[:6].zero_()
            for j in range(6):
                lin.weight[j, j] = 1.0
        head = lins[-1]
        head.weight.mul_(rest)
        head.bias.zero_()
        head.weight[:, :6].zero_()
        for i in range(3):
            head.weight[i, i] = 1.0
            head.weight[i, 3 + i] = -1.0
        head.bias[3:] = log_std


def measure(config='c3', n_actor=None, hidden='1024-1024', batch=4096, steps=None, graph=False,
            device='cuda:0', data_parallel=False, seed_offset=0, policy=None, lr=None):
    """Timings of `steps` training steps as a dict (see the module docstring).
    `policy`: 'random' (fresh initialisation; config 3's convention since round
    2) or 'straight' (`straight_policy_`; config 5's default: the oracle needs
    streamlines that live).  `lr`: 3e-4 (the trainers' default) for 'random';
    1e-7 for 'straight' -- Adam moves every weight by about lr per update
    whatever the gradient, and a few hundred updates at 3e-4 on the synthetic
    reward un-learn the hand-built policy before the episode ends; the work of
    an update does not depend on lr."""
    from tracktolearn_amd.algorithms.sac_auto import SACAuto
    from tracktolearn_amd.environments import TrackingEnvironment
    from tracktolearn_amd.utils.synthetic import synthetic_seeds, synthetic_subject
    dev = torch.device(device)
    oracle = config == 'c5'
    if n_actor is None:
        n_actor = 16384 if oracle else 65536
    if policy is None:
        policy = 'straight' if oracle else 'random'
    if lr is None:
        lr = 1e-7 if policy == 'straight' else 3e-4
    if steps is None:
        # config 5: one whole episode and a bit (the longest chord of the ball mask
        # is ~107 steps): the oracle only scores streamlines longer than
        # min_nb_steps (bonus) / 5 min_nb_steps (stopping criterion)
        steps = 112 if oracle else 24
    subject = synthetic_subject(96, 45, seed=1234, peaks=True)
    dto = dict(n_dirs=4, theta=30.0, npv=1, binary_stopping_threshold=0.1,
               step_size=0.75, min_length=20.0, max_length=200.0,
               compute_reward=True, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=dev, target_sh_order=8)
    tmp = None
    if oracle:
        from tracktolearn_amd.oracles.oracle import OracleSingleton
        from tracktolearn_amd.oracles.transformer_oracle import save_random_checkpoint
        OracleSingleton.reset()
        tmp = tempfile.TemporaryDirectory()
        ck = save_random_checkpoint(os.path.join(tmp.name, 'oracle.ckpt'), n_head=4,
                                    n_layers=4, seed=5)
        # min_length 10 mm (default 20): the oracle scores stopped streamlines of
        # more than min_nb_steps = 13 steps and all active ones beyond 65 steps;
        # the longest chord of the synthetic ball mask is ~107 steps
        dto.update(oracle_bonus=10.0, oracle_checkpoint=ck, oracle_stopping_criterion=True,
                   min_length=10.0)
    env = TrackingEnvironment(subject, 'training', dto)
    env.seeds = synthetic_seeds(subject[1].data, n_actor, seed=1 + seed_offset)
    W = env.get_state_size()
    torch.manual_seed(0)
    alg = SACAuto(W, 3, hidden, lr=lr, n_actors=n_actor, batch_size=batch,
                  replay_size=int(1e6), rng=None, device=dev)
    if policy == 'straight':
        straight_policy_(alg.agent.actor, 7 * 45)
        alg.target.actor.load_state_dict(alg.agent.actor.state_dict())
    if data_parallel:
        alg.enable_data_parallel()
    if graph:
        alg.enable_graph()
    out = {'config': 'BASELINE configs[4] (oracle bonus + oracle stopping)' if oracle
           else 'BASELINE configs[2]', 'W': W, 'hidden': hidden, 'n_actor': n_actor,
           'batch': batch, 'graph': bool(graph), 'data_parallel': bool(data_parallel),
           'policy': policy, 'lr': lr,
           'oracle_net': ('fused kernel' if getattr(env._oracle, 'net', None) is not None
                          else 'torch module under autocast') if oracle else None,
           'fused_learner': os.environ.get('TTL_FUSED_LEARNER', '1') != '0'}

    def reset():
        return env.reset(0, n_actor)

    def one_step(state, ph=None):
        span = ph.span if ph is not None else (lambda name: _Null)
        if state.shape[0] == 0:              # a random policy ends episodes fast
            state = reset()
        with span('policy'):
            with torch.no_grad():
                a = alg.sample_action(state)
        n = a.shape[0]
        with span('env_step'):
            ns, r, d, info = env.step_device(a)
        with span('replay_add'):
            alg.replay_buffer.add_partitioned(state, a, ns, info['row_dest'], r, d)
        with span('replay_sample'):
            b = alg.replay_buffer.sample(batch)
        with span('update'):
            alg.update(b)
        with span('harvest'):
            state, _ = env.harvest()
        return state, n

    # fill the ring, first-call costs, graph capture: the same `steps` steps once
    # untimed -- the policy's (and the oracle's) GEMM shapes follow the number of
    # active rows, and hipBLASLt picks its kernel for every new row count once
    # (a few hundred us of host time each; a training run meets every count again)
    state = reset()
    for _ in range(steps):
        state, _ = one_step(state)
    # SACAuto.update alone
    b = alg.replay_buffer.sample(batch)
    for _ in range(3):
        alg.update(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        alg.update(b)
    torch.cuda.synchronize()
    out['update_ms'] = (time.perf_counter() - t0) / 30 * 1e3
    fl = getattr(alg, '_fused', None)
    if fl is not None:
        flops = fl.flops_per_update(batch)
        tf = flops['issued'] / (out['update_ms'] * 1e-3) / 1e12
        out['roofline'] = {
            'bound': 'mfma', 'kernel': 'SACAuto.update (16 fp32 GEMMs on hipBLASLt + the '
                                       'learner kernels of libttl_hip.so)',
            'achieved': tf, 'peak': FP32_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
            'frac': tf / FP32_MFMA_PEAK_TF, 'traffic': None, 'dtype': 'f32',
            'flop_per_update_issued': flops['issued'],
            'flop_per_update_reference_autograd': flops['autograd'],
            'frac_on_reference_flop': flops['autograd'] / (out['update_ms'] * 1e-3) / 1e12
            / FP32_MFMA_PEAK_TF,
            'how': 'FLOP counted from the layer shapes (2 M N K per GEMM, thin layers '
                   'included) / update_ms (wall clock, 30 updates, synchronised) / the dense '
                   'fp32 MFMA peak'}
    # the plain figure
    state = reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    units = 0
    for _ in range(steps):
        state, n = one_step(state)
        units += n
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out.update(train_steps=steps, train_step_ms=dt / steps * 1e3,
               train_rows_per_step=units / steps, train_streamline_steps_per_s=units / dt)
    # the breakdown
    ph = _Phases()
    restore = _instrument_oracle(env, ph)
    ar_orig = None
    if fl is not None and data_parallel:
        # the part of the gradient exchange the compute stream waits for: with
        # TTL_DP_OVERLAP (default) the critics' all-reduce runs beside the actor's
        # backward and only `_all_reduce_end` (the waits + the 1/world scaling)
        # shows; without, `_all_reduce` is the whole exchange
        ar_orig = (fl._all_reduce, fl._all_reduce_end)

        def timed_all_reduce(*extra):
            with ph.span('update_all_reduce'):
                ar_orig[0](*extra)

        def timed_all_reduce_end(handle):
            with ph.span('update_all_reduce'):
                ar_orig[1](handle)
        fl._all_reduce, fl._all_reduce_end = timed_all_reduce, timed_all_reduce_end
    state = reset()
    torch.cuda.synchronize()
    units = 0
    for _ in range(steps):
        state, n = one_step(state, ph)
        units += n
    tot = ph.totals_ms()
    rows = restore()
    if ar_orig is not None:
        fl._all_reduce, fl._all_reduce_end = ar_orig
    phases = {k: v / steps for k, v in tot.items()}
    if oracle:
        phases['env_step_without_oracle_network'] = phases['env_step'] - \
            phases.get('oracle_resample', 0.0) - phases.get('oracle_transformer', 0.0)
        out['oracle_rows_scored_per_step'] = rows['scored'] / steps
        out['oracle_batches_per_step'] = rows['batches'] / steps
    out['phases_ms_per_step'] = phases
    out['phases_rows_per_step'] = units / steps
    # the policy phase against the fp32 MFMA peak: the actor's forward on the rows of a step
    # (2 M N K per layer, head included), summed over the episode / the summed phase time
    if phases.get('policy'):
        dims = [W] + [int(h) for h in str(hidden).split('-')] + [6]
        macs = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        tf = 2.0 * macs * units / (tot['policy'] * 1e-3) / 1e12
        out['policy_forward'] = {'bound': 'mfma', 'achieved': tf, 'peak': FP32_MFMA_PEAK_TF,
                                 'unit': 'TFLOP/s', 'frac': tf / FP32_MFMA_PEAK_TF,
                                 'dtype': 'f32', 'rows_per_step': units / steps,
                                 'how': 'actor forward FLOP of the rows stepped / policy phase time '
                                        '(HIP events): two hipBLASLt GEMMs with bias + ReLU '
                                        'epilogue whose row count changes every step, and the '
                                        'head as one ttl_thin_forward launch'}
    if tmp is not None:
        from tracktolearn_amd.oracles.oracle import OracleSingleton
        OracleSingleton.reset()
        tmp.cleanup()
    return out


class _NullSpan:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_Null = _NullSpan()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c3', choices=('c3', 'c5'))
    ap.add_argument('--n_actor', type=int, default=None)
    ap.add_argument('--hidden', default='1024-1024')
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--graph', action='store_true')
    ap.add_argument('--policy', default=None, choices=('random', 'straight'))
    args = ap.parse_args()
    print(json.dumps(measure(args.config, args.n_actor, args.hidden, args.batch, args.steps,
                             args.graph, policy=args.policy)))


if __name__ == '__main__':
    main()
