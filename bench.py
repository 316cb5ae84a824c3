#!/usr/bin/env python3
"""bench.py -- streamline-steps/s of the MI355X environment step.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): "env.step only" on a
synthetic 96^3 x 45-SH volume, n_actor = 262144 streamlines per GPU, ball
mask, step 0.75 mm, theta 30 deg, max_length 200 mm, n_dirs = 4, reward off,
float32 (training-env) arithmetic, scripted policy-free actions generated on
the GPU (counter-based; `ttl_scripted_actions`).

One "step" = one pass of the hot path over the batch: scripted actions ->
env.step_device() -> env.harvest().  A streamline-step = one active
streamline advanced by one step (the reference's `t += n_active`,
TrackToLearn/algorithms/ddpg.py:219).  W warm-up steps run on their own
episode, the env is reset, then exactly K steps are timed between
barrier + torch.cuda.synchronize() pairs; value = streamline-steps of all
ranks / max-over-ranks time.  If an episode runs out of streamlines inside the
timed region the reset is timed too.

With N > 1 the driver launches one process per GPU (torch.distributed.run);
streamlines shard across ranks with the volumes replicated and no collective
on the step path ("scaling": "weak": n_actor per GPU is fixed).  The one
exchange the path has -- collating finished tracts (lengths + flags
all-gather over RCCL) -- runs after the timed region and is reported as
`collate_ms`.

Extra objects on the JSON line:
  roofline     dominant kernel (k_state: 7-point SH gather + state row write):
               algorithmic bytes per launch / average launch duration measured
               with HIP events on the launch stream inside the timed region.
  cpu_baseline the CPU oracle (oracle/env_oracle.py, a port of the reference's
               NumPy env) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D = 96
C = 45
N_ACTOR = 262144
N_DIRS = 4
STEP_MM = 0.75
THETA = 30.0
MAX_LENGTH = 200.0
WOBBLE = 0.05
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes(c, k):
    """bytes per streamline-step of the whole step, and the share
    of the dominant kernel (gather + position history + state row write)."""
    gather = 4 * 56 * c
    hist = 12 * (k + 1)
    state_row = 4 * (7 * c + 3 * k)
    whole = gather + 64 * 8 + 12 + hist + state_row + 12 + 2
    return whole, gather + hist + state_row


def make_env(device, seed_offset):
    import torch
    from tracktolearn_amd.environments import TrackingEnvironment
    from tracktolearn_amd.utils.synthetic import (synthetic_seeds,
                                                  synthetic_subject)
    subject = synthetic_subject(D, C, seed=1234, peaks=False,
                                affine_dtype=np.float32)
    dto = dict(n_dirs=N_DIRS, theta=THETA, npv=1, binary_stopping_threshold=0.1,
               step_size=STEP_MM, min_length=20.0, max_length=MAX_LENGTH,
               compute_reward=False, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device(device),
               target_sh_order=8)
    env = TrackingEnvironment(subject, 'testing', dto)
    env.seeds = synthetic_seeds(subject[1].data, N_ACTOR, seed=100 + seed_offset)
    return env, subject


def run_steps(env, n_steps, seed, counter):
    """n_steps passes of the hot path; returns streamline-steps processed."""
    state = counter['state']
    total = 0
    for _ in range(n_steps):
        if env._n_active == 0:
            state = env.reset(0, N_ACTOR)
            counter['step'] = 0
            counter['resets'] += 1
        n = env._n_active
        actions = env.scripted_actions(state, counter['step'], seed, WOBBLE)
        env.step_device(actions)
        state, _ = env.harvest()
        total += n
        counter['step'] += 1
    counter['state'] = state
    return total


def cpu_baseline(mask_data, sh, n_sample=65536, n_steps=12):
    """The CPU oracle (a port of the reference's NumPy env) on a bounded
    sample of the same workload: n_sample streamlines, first n_steps steps."""
    from oracle import env_oracle as orc
    from oracle.scripted_policy import scripted_actions
    from tracktolearn_amd.utils.synthetic import synthetic_seeds
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
        single_thread = threadpool_limits(limits=1)
    except Exception:          # pragma: no cover
        single_thread = contextlib.nullcontext()
    seeds = synthetic_seeds(mask_data, n_sample, seed=100)
    env = orc.OracleTrackingEnv(
        sh, mask_data, seeds, n_dirs=N_DIRS, theta=THETA,
        step_size=np.float32(STEP_MM), max_nb_steps=int(MAX_LENGTH / STEP_MM),
        mask_threshold=0.1, compute_reward=False, spline_eval='scipy')
    rates, spent = [], 0.0
    with single_thread:
        for _ in range(3):                  # median of three repetitions
            total, elapsed = 0, 0.0
            state = env.reset(0, n_sample)
            for step in range(n_steps):
                idx = env.continue_idx
                if len(idx) == 0:
                    break
                a = scripted_actions(state, 7 * C, idx, 1, step, WOBBLE)
                t0 = time.perf_counter()
                env.step(a)
                state, _ = env.harvest()
                elapsed += time.perf_counter() - t0
                total += len(idx)
            rates.append(total / elapsed)
            spent += elapsed
    return {'value': float(np.median(rates)), 'unit': 'streamline-steps/s', 'cores': 1,
            'kind': 'port',
            'sample': f'oracle/env_oracle.py (numpy/scipy port of the reference '
                      f'env), {n_sample} of the {N_ACTOR} streamlines, first '
                      f'{n_steps} steps, step()+harvest() timed, median of 3 '
                      f'repetitions, 1 thread of {os.cpu_count()} host cpus, '
                      f'{spent:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--whole-episode', action='store_true',
                    help='also run one episode to exhaustion after the timed '
                         'region (SURVEY 8d (ii)) and report it as whole_episode')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with torch.distributed.run for --gpus > 1')
    # rehearsal on a 1-GPU box: TTL_BENCH_ONE_DEVICE=1 puts every rank on
    # cuda:0 and TTL_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks
    # on one device); the driver's real runs use neither
    if os.environ.get('TTL_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
    backend = os.environ.get('TTL_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    env, subject = make_env(device, seed_offset=rank)
    seed = 1 + rank

    # ---- warm-up on its own episode -------------------------------------
    counter = {'state': env.reset(0, N_ACTOR), 'step': 0, 'resets': 0}
    run_steps(env, args.warmup, seed, counter)
    # exercise the periodic re-sort of the processing order (every 16 steps)
    # once outside the timed regions
    if env._n_active:
        env._refresh_processing_order(force=True)
    torch.cuda.synchronize()

    # ---- timed region -----------------------------------------------------
    counter = {'state': env.reset(0, N_ACTOR), 'step': 0, 'resets': 0}
    env.profile_begin(max_launches=max(16, args.steps + 8), classes=('state',))
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_units = run_steps(env, args.steps, seed, counter)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = env.profile_end()

    # untimed replay of the same steps with every kernel class bracketed, for
    # the per-kernel breakdown (the timed region only brackets the dominant
    # kernel to keep the event records out of the other launch gaps)
    counter2 = {'state': env.reset(0, N_ACTOR), 'step': 0, 'resets': 0}
    env.profile_begin(max_launches=max(16, args.steps + 8),
                      classes=('advance', 'prefix', 'state'))
    run_steps(env, args.steps, seed, counter2)
    torch.cuda.synchronize()
    prof_all = env.profile_end()

    # ---- optional: whole episode to exhaustion, outside the
    # K-step region of the contract ----------------------------------------
    ep = None
    if args.whole_episode:
        # two episodes, the second one reported (the first one of a process
        # runs ~30 % slower: allocator growth and first-use effects a tracking
        # run pays once, on its first seed batch)
        for attempt in range(2):
            state = env.reset(0, N_ACTOR)
            torch.cuda.synchronize()
            t_ep = time.perf_counter()
            ep_units, ep_steps = 0, 0
            while env._n_active:
                ep_units += env._n_active
                actions = env.scripted_actions(state, ep_steps, seed, WOBBLE)
                env.step_device(actions)
                state, _ = env.harvest()
                ep_steps += 1
            torch.cuda.synchronize()
            t_ep = time.perf_counter() - t_ep
            first = ep
            ep = {'streamline_steps_per_s_rank0': ep_units / t_ep, 'steps': ep_steps,
                  'streamline_steps': ep_units, 'ms': t_ep * 1e3,
                  'order_refresh_every': env.SPATIAL_ORDER_REFRESH}
            if first is not None:
                ep['first_episode_ms'] = first['ms']

    # ---- collate finished tracts (the path's only exchange step) ----------
    collate_ms, collate_error = None, None
    if world > 1:
        try:
            from tracktolearn_amd.parallel import all_gather_tract_index
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            all_gather_tract_index(env)
            torch.cuda.synchronize()
            collate_ms = (time.perf_counter() - t1) * 1e3
        except Exception as exc:      # never lose the bench line to the collate
            collate_error = repr(exc)

    red_dev = device if backend == 'nccl' else 'cpu'
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    units = torch.tensor([float(n_units)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    t_max = float(t_max.item())
    total_units = float(units.item())

    if rank == 0:
        whole_b, kern_b = algorithmic_bytes(C, N_DIRS)
        state_ms, state_n = prof['state']
        adv_ms, adv_n = prof_all['advance']
        pre_ms, _ = prof_all['prefix']
        avg_launch_s = state_ms / max(state_n, 1) * 1e-3
        units_per_launch = n_units / max(state_n, 1)
        achieved = kern_b * units_per_launch / avg_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get('k_state_hbm_bytes_per_launch')
            except Exception:
                traffic = None
        value = total_units / t_max
        line = {
            'metric': 'streamline-steps/s at n_actor=262144',
            'value': value,
            'unit': 'streamline-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': t_max / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'env.step only (scripted actions -> step -> '
                            'harvest), 96^3x45-SH synthetic volume, '
                            'n_actor=262144 per GPU, 1xMI355X per rank',
                'n_actor_per_gpu': N_ACTOR, 'volume': [D, D, D, C],
                'n_dirs': N_DIRS, 'state_width': 7 * C + 3 * N_DIRS,
                'reward': False, 'arithmetic': 'float32 directions (train env)',
                'loop': 'step_device + harvest (survivors-first rows, 8-byte '
                        'count readback per step)',
                'resets_in_timed_region': counter['resets'],
                'parallelism': f'streamlines sharded over {world} GPU(s), '
                               'volumes replicated',
            },
            'streamline_steps': total_units,
            'roofline': {
                'bound': 'hbm', 'kernel': 'k_state_dd<12,4,false,true>',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                # memory-side view of the same launch (PMC bytes / duration)
                'traffic_GBs': (traffic / avg_launch_s / 1e9) if traffic else None,
                'traffic_frac': (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS)
                if traffic else None,
                'bytes_per_unit': kern_b,
                'units_per_launch': units_per_launch,
                'avg_launch_ms': avg_launch_s * 1e3,
                'launches': state_n,
                'whole_step_bytes_per_unit': whole_b,
                'whole_step_GBs': whole_b * value / world / 1e9,
                'other_kernels_ms_per_step': {
                    'advance': adv_ms / max(adv_n, 1),
                    'prefix': pre_ms / max(adv_n, 1)},
            },
        }
        if ep is not None:
            line['whole_episode'] = ep
        if collate_ms is not None:
            line['collate_ms'] = collate_ms
        if collate_error is not None:
            line['collate_error'] = collate_error
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(subject[1].data, subject[0].data)
        elif not args.no_cpu_baseline:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
