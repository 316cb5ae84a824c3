#!/usr/bin/env python3
"""bench.py -- streamline-steps/s of the MI355X environment step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--windows M]

Workload (BASELINE.json configs[1]): "env.step only" on a
synthetic 96^3 x 45-SH volume, n_actor = 262144 streamlines per GPU, ball
mask, step 0.75 mm, theta 30 deg, max_length 200 mm, n_dirs = 4, reward off,
float32 (training-env) arithmetic, scripted policy-free actions generated on
the GPU (counter-based; `ttl_scripted_actions`).

One "step" = one pass of the hot path over the batch: scripted actions ->
env.step_device() -> env.harvest().  A streamline-step = one active
streamline advanced by one step (the reference's `t += n_active`,
TrackToLearn/algorithms/ddpg.py:219).  W warm-up steps run on their own
episode; then M windows (default 11) of EXACTLY K steps each are timed, every
window bracketed by barrier + torch.cuda.synchronize() on both sides and
started from a fresh (untimed) reset, so all windows time the same K steps.
`value` / `ms_per_step` are those of the MEDIAN window (max over ranks per
window); min / max over the windows are printed next to them (`windows`).
Every second window also brackets the dominant kernel with HIP events (for
`roofline`); those records cost a few percent, both medians are printed.
One 12-step window is ~3 ms of GPU time, which is why one window alone is a
fragile figure.

With N > 1 the driver launches one process per GPU (torch.distributed.run);
called bare with --gpus N > 1 this script launches those N ranks itself (as
child processes, before this process touches the GPU) and relays rank 0's
line.  Streamlines shard across ranks with the volumes replicated and no
collective on the step path ("scaling": "weak": n_actor per GPU is fixed).
The one exchange the path has -- collating finished tracts on rank 0 (exact-
size gather to root over RCCL) -- runs after the timed region and is reported
as `collate_ms`.

Extra objects on the JSON line:
  roofline     dominant kernel (k_state_dd: 7-point SH gather + state row
               write).  `achieved` = HBM bytes the kernel moves per launch /
               its average launch duration (HIP events on the launch stream,
               inside the timed windows); the bytes come from PMC counters
               (`profiles/pmc_traffic.json`: FETCH_SIZE / WRITE_SIZE passes of
               this same command, bytes PER UNIT x this run's units per
               launch) -- they are not measured in this run and the line says
               so.  `frac` = achieved / 8 TB/s.  Next to it: the compulsory
               traffic (what an ideal kernel must move) and SURVEY 8(d)'s
               algorithmic figure (counts all 56 corner fetches per unit, an
               upper bound on naive traffic, can exceed the peak).
  cpu_baseline the CPU oracle (oracle/env_oracle.py, a port of the reference's
               NumPy env) timed on this box's host cores on a bounded sample:
               1 thread, and all the cores this process may use (the
               streamlines sharded over worker processes).
"""
import argparse
import json
import os
import socket
import subprocess
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
DOMINANT_KERNEL = 'k_state_dd<12,4,false,true>'


def algorithmic_bytes(c, k):
    """bytes per streamline-step of the whole step, and the share
    of the dominant kernel (gather + position history + state row write)."""
    gather = 4 * 56 * c
    hist = 12 * (k + 1)
    state_row = 4 * (7 * c + 3 * k)
    whole = gather + 64 * 8 + 12 + hist + state_row + 12 + 2
    return whole, gather + hist + state_row


def compulsory_bytes(c, k, n_mask_voxels, units_per_launch):
    """What an ideal state gather must move per unit: the row written once,
    the per-streamline inputs read once (slot record 16 B + output row index
    4 B + the K+1 history points of the direction block) and every SH record
    (padded to 16-byte columns) inside the tracking mask read ONCE per launch,
    shared by all the streamlines of the launch."""
    row = 4 * (7 * c + 3 * k)
    per_streamline = 16 + 4 + 12 * (k + 1)
    record = 4 * ((c + 3) // 4 * 4)
    return row + per_streamline + n_mask_voxels * record / max(units_per_launch, 1.0)


def make_subject():
    from tracktolearn_amd.utils.synthetic import synthetic_subject
    return synthetic_subject(D, C, seed=1234, peaks=False, affine_dtype=np.float32)


def make_env(subject, device, seed_offset):
    import torch
    from tracktolearn_amd.environments import TrackingEnvironment
    from tracktolearn_amd.utils.synthetic import synthetic_seeds
    dto = dict(n_dirs=N_DIRS, theta=THETA, npv=1, binary_stopping_threshold=0.1,
               step_size=STEP_MM, min_length=20.0, max_length=MAX_LENGTH,
               compute_reward=False, alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device(device),
               target_sh_order=8)
    env = TrackingEnvironment(subject, 'testing', dto)
    env.seeds = synthetic_seeds(subject[1].data, N_ACTOR, seed=100 + seed_offset)
    return env


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


# --------------------------------------------------------------------------
# CPU baseline: the oracle on the host cores (rank 0, N = 1 only; runs before
# this process touches the GPU, so that worker processes can be forked)
# --------------------------------------------------------------------------
def _oracle_leg(sh, mask, seeds, n_steps):
    """One oracle run over `seeds`: (streamline-steps, seconds in step+harvest)."""
    from oracle import env_oracle as orc
    from oracle.scripted_policy import scripted_actions
    env = orc.OracleTrackingEnv(
        sh, mask, seeds, n_dirs=N_DIRS, theta=THETA,
        step_size=np.float32(STEP_MM), max_nb_steps=int(MAX_LENGTH / STEP_MM),
        mask_threshold=0.1, compute_reward=False, spline_eval='scipy')
    total, elapsed = 0, 0.0
    state = env.reset(0, len(seeds))
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
    return total, elapsed


_FORK_SHARED = {}


def _oracle_worker(job):
    lo, hi, n_steps = job
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            return _oracle_leg(_FORK_SHARED['sh'], _FORK_SHARED['mask'],
                               _FORK_SHARED['seeds'][lo:hi], n_steps)
    except ImportError:      # pragma: no cover
        return _oracle_leg(_FORK_SHARED['sh'], _FORK_SHARED['mask'],
                           _FORK_SHARED['seeds'][lo:hi], n_steps)


def cpu_baseline(mask_data, sh, n_sample=65536, n_steps=12):
    """The CPU oracle (a port of the reference's NumPy env) on a bounded
    sample of the same workload: n_sample streamlines, first n_steps steps.
    Two figures: one thread (median of 3 repetitions), and every core this
    process may use -- the sample sharded over worker processes, each a
    single-threaded oracle on its slice (streamlines are independent; the
    oracle's own NumPy/SciPy calls are single-threaded apart from one small
    BLAS product, so lifting the BLAS thread limit alone changes nothing)."""
    import contextlib
    import multiprocessing as mp
    from tracktolearn_amd.utils.synthetic import synthetic_seeds
    try:
        from threadpoolctl import threadpool_limits
        single_thread = threadpool_limits(limits=1)
    except Exception:          # pragma: no cover
        single_thread = contextlib.nullcontext()
    seeds = synthetic_seeds(mask_data, n_sample, seed=100)
    rates, spent = [], 0.0
    with single_thread:
        for _ in range(3):                  # median of three repetitions
            total, elapsed = _oracle_leg(sh, mask_data, seeds, n_steps)
            rates.append(total / elapsed)
            spent += elapsed
    one = {'value': float(np.median(rates)), 'unit': 'streamline-steps/s', 'cores': 1,
           'kind': 'port',
           'sample': f'oracle/env_oracle.py (numpy/scipy port of the reference '
                     f'env), {n_sample} of the {N_ACTOR} streamlines, first '
                     f'{n_steps} steps, step()+harvest() timed, median of 3 '
                     f'repetitions, 1 thread of {os.cpu_count()} host cpus, '
                     f'{spent:.1f} s'}
    # all usable cores: at most 16 workers (a 1-GPU box's CPU share)
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:     # pragma: no cover
        usable = os.cpu_count() or 1
    workers = max(1, min(16, usable))
    multi = None
    try:
        _FORK_SHARED.update(sh=sh, mask=mask_data, seeds=seeds)
        per = -(-n_sample // workers)
        jobs = [(w * per, min((w + 1) * per, n_sample), n_steps)
                for w in range(workers) if w * per < n_sample]
        t0 = time.perf_counter()
        with mp.get_context('fork').Pool(len(jobs)) as pool:
            parts = pool.map(_oracle_worker, jobs)
        wall = time.perf_counter() - t0
        units = sum(p[0] for p in parts)
        busy = max(p[1] for p in parts)
        multi = {'value': units / busy, 'unit': 'streamline-steps/s',
                 'cores': len(jobs), 'kind': 'port',
                 'sample': f'the same sample sharded over {len(jobs)} forked worker '
                           f'processes (one single-threaded oracle each, '
                           f'{usable} usable of {os.cpu_count()} host cpus); units / '
                           f'slowest worker\'s step()+harvest() time; {wall:.1f} s '
                           f'wall incl. process start and per-worker setup'}
    except Exception as exc:   # never lose the bench line to the baseline
        multi = {'error': repr(exc)}
    finally:
        _FORK_SHARED.clear()
    one['all_cores'] = multi
    return one


# --------------------------------------------------------------------------
# bare `python bench.py --gpus N` (N > 1): start the N ranks as children
# --------------------------------------------------------------------------
def self_launch(args, argv):
    """Runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a child process and relays rank 0's JSON line.  The parent never
    imports torch.cuda / touches the GPU."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for row in proc.stdout.splitlines():
        if row.startswith('{') and '"metric"' in row:
            line = row
    if line is not None:
        print(line, flush=True)
    if proc.returncode != 0 or line is None:
        sys.stderr.write(f'bench.py: the {args.gpus}-rank run failed '
                         f'(exit code {proc.returncode})\n')
        if line is None:
            sys.stderr.write(proc.stdout[-4000:])
        return proc.returncode or 1
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--windows', type=int, default=11,
                    help='timed windows of --steps steps each (median reported)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-whole-episode', action='store_true',
                    help='skip the episode-to-exhaustion figure (SURVEY 8d (ii))')
    ap.add_argument('--whole-episode', action='store_true',
                    help='(default now; kept for older command lines)')
    args = ap.parse_args(argv)

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args, argv)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    subject = make_subject()
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        # before the first GPU call of this process (forked workers)
        cpu = cpu_baseline(subject[1].data, subject[0].data)

    import torch
    import torch.distributed as dist

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
    red_dev = device if backend == 'nccl' else 'cpu'

    def barrier():
        if world > 1:
            dist.barrier()

    env = make_env(subject, device, seed_offset=rank)
    seed = 1 + rank

    # ---- warm-up on its own episode -------------------------------------
    counter = {'state': env.reset(0, N_ACTOR), 'step': 0, 'resets': 0}
    run_steps(env, args.warmup, seed, counter)
    # exercise the periodic re-sort of the processing order once outside the
    # timed regions
    if env._n_active:
        env._refresh_processing_order(force=True)
    torch.cuda.synchronize()

    # ---- timed windows: each EXACTLY --steps steps from a fresh reset ------
    # The dominant kernel is bracketed with HIP events in every second window
    # only: an event record costs ~6 us of GPU idle time on either side of the
    # kernel (rocprofv3 trace, benchmarks/trace_gaps.py), ~5 % of a step.  All
    # windows are timed alike and `value` is the median over all of them.
    n_win = max(1, args.windows)
    times, n_units, resets = [], 0, 0
    state_ms, state_n, evented = 0.0, 0, []
    for w in range(n_win):
        counter = {'state': env.reset(0, N_ACTOR), 'step': 0, 'resets': 0}
        with_events = (w % 2 == 1) or n_win == 1
        if with_events:
            env.profile_begin(max_launches=args.steps + 8, classes=('state',))
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_units = run_steps(env, args.steps, seed, counter)
        torch.cuda.synchronize()
        barrier()
        times.append(time.perf_counter() - t0)
        resets = max(resets, counter['resets'])
        if with_events:
            ms, cnt = env.profile_end()['state']
            state_ms += ms
            state_n += cnt
            evented.append(w)

    # untimed replay of the same steps with every kernel class bracketed, for
    # the per-kernel breakdown (the timed windows only bracket the dominant
    # kernel to keep the event records out of the other launch gaps)
    counter2 = {'state': env.reset(0, N_ACTOR), 'step': 0, 'resets': 0}
    env.profile_begin(max_launches=max(16, args.steps + 8),
                      classes=('advance', 'prefix', 'state'))
    run_steps(env, args.steps, seed, counter2)
    torch.cuda.synchronize()
    prof_all = env.profile_end()

    # ---- whole episode to exhaustion, outside the K-step
    # windows of the contract ------------------------------------------------
    ep = None
    free_tail = os.environ.get('TTL_BENCH_FREE_TAIL', '1') != '0'
    if not args.no_whole_episode:
        # two episodes, the second one reported (the first one of a process
        # runs slower: allocator growth and first-use effects a tracking run
        # pays once, on its first seed batch)
        for attempt in range(2):
            state = env.reset(0, N_ACTOR)
            torch.cuda.synchronize()
            t_ep = time.perf_counter()
            ep_units, ep_steps, free_steps = 0, 0, 0
            while env._n_active:
                if free_tail and env.freerun_supported():
                    # from 16 384 rows down a step is bound by the host waiting
                    # for its survivor count: free-running steps, launched for
                    # the newest count the GPU has reported, never waited for
                    left = env._n_active
                    _, free_steps = env.run_free_eager(
                        lambda st: env.scripted_actions_free(st, seed, WOBBLE), state)
                    ep_steps += free_steps
                    break
                ep_units += env._n_active
                actions = env.scripted_actions(state, ep_steps, seed, WOBBLE)
                env.step_device(actions)
                state, _ = env.harvest()
                ep_steps += 1
            torch.cuda.synchronize()
            t_ep = time.perf_counter() - t_ep
            # streamline-steps = points added = sum(lengths - 1)
            ep_units = int(env._buf_lengths[:N_ACTOR].sum().item()) - N_ACTOR
            first = ep
            ep = {'streamline_steps_per_s_rank0': ep_units / t_ep, 'steps': ep_steps,
                  'streamline_steps': ep_units, 'ms': t_ep * 1e3,
                  'free_running_tail_steps': free_steps,
                  'order_refresh_every': env.SPATIAL_ORDER_REFRESH}
            if first is not None:
                ep['first_episode_ms'] = first['ms']

    # ---- collate finished tracts on rank 0 (the path's only exchange) ------
    collate_ms, collate_bytes, collate_error = None, None, None
    if world > 1:
        try:
            from tracktolearn_amd.parallel import gather_tract_arrays
            barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            got = gather_tract_arrays(env)
            torch.cuda.synchronize()
            barrier()
            collate_ms = (time.perf_counter() - t1) * 1e3
            if got is not None:
                collate_bytes = got[3]
        except Exception as exc:      # never lose the bench line to the collate
            collate_error = repr(exc)

    t_win = torch.tensor(times, dtype=torch.float64, device=red_dev)
    units = torch.tensor([float(n_units)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t_win, op=dist.ReduceOp.MAX)      # per window, over ranks
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    t_all = t_win.cpu().numpy()          # per window, max over the ranks
    t_win = np.sort(t_all)
    total_units = float(units.item())        # of ONE window, all ranks
    t_med = float(t_win[len(t_win) // 2])

    if rank == 0:
        whole_b, kern_b = algorithmic_bytes(C, N_DIRS)
        adv_ms, adv_n = prof_all['advance']
        pre_ms, _ = prof_all['prefix']
        avg_launch_s = state_ms / max(state_n, 1) * 1e-3
        units_per_launch = n_units / max(args.steps, 1)
        algorithmic_gbs = kern_b * units_per_launch / avg_launch_s / 1e9
        n_mask = int(np.count_nonzero(subject[1].data))
        comp_b = compulsory_bytes(C, N_DIRS, n_mask, units_per_launch)
        pmc, pmc_src = None, None
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath):
            try:
                js = json.load(open(tpath))
                pmc = js.get('k_state_hbm_bytes_per_unit')
                pmc_src = (f"profiles/pmc_traffic.json <- {js.get('source')}: PMC "
                           f"FETCH_SIZE/WRITE_SIZE passes of this command, "
                           f"(2*FETCH+WRITE)*1024 bytes per unit x this run's units "
                           f"per launch; NOT measured in this run")
            except Exception:
                pmc = None
        traffic = pmc * units_per_launch if pmc else None
        achieved = traffic / avg_launch_s / 1e9 if traffic else None
        value = total_units / t_med
        line = {
            'metric': 'streamline-steps/s at n_actor=262144',
            'value': value,
            'unit': 'streamline-steps/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': t_med / args.steps * 1e3,
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
                'resets_in_timed_region': resets,
                'parallelism': f'streamlines sharded over {world} GPU(s), '
                               'volumes replicated',
            },
            'streamline_steps': total_units,
            'windows': {
                'n': len(t_win), 'timed': 'each window = exactly --steps steps '
                'from a fresh untimed reset; value/ms_per_step = median window',
                'with_kernel_events': evented,
                'value_median_with_events': (total_units / float(np.median(
                    [t_all[i] for i in evented]))) if evented else None,
                'value_median_without_events': (total_units / float(np.median(
                    [t_all[i] for i in range(len(t_all)) if i not in evented])))
                if len(evented) < len(t_all) else None,
                'value_min': total_units / float(t_win[-1]),
                'value_median': value,
                'value_max': total_units / float(t_win[0]),
                'ms_per_step_min': float(t_win[0]) / args.steps * 1e3,
                'ms_per_step_max': float(t_win[-1]) / args.steps * 1e3,
            },
            'roofline': {
                'bound': 'hbm', 'kernel': DOMINANT_KERNEL,
                # HBM bytes per launch (PMC counters) / measured launch time
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': (achieved / HBM_PEAK_GBS) if achieved else None,
                'traffic': traffic,
                'traffic_bytes_per_unit': pmc,
                'traffic_source': pmc_src,
                # what an ideal kernel must move (rows once + per-streamline
                # inputs + every in-mask SH record once per launch)
                'compulsory_bytes_per_unit': comp_b,
                'compulsory_GBs': comp_b * units_per_launch / avg_launch_s / 1e9,
                'compulsory_frac': comp_b * units_per_launch / avg_launch_s / 1e9
                / HBM_PEAK_GBS,
                # all 56 corner fetches charged per unit -- an
                # upper bound on naive traffic, not a roofline fraction
                'algorithmic_bytes_per_unit': kern_b,
                'algorithmic_GBs': algorithmic_gbs,
                'units_per_launch': units_per_launch,
                'avg_launch_ms': avg_launch_s * 1e3,
                'launches': state_n,
                'whole_step_algorithmic_bytes_per_unit': whole_b,
                'whole_step_algorithmic_GBs': whole_b * value / world / 1e9,
                'other_kernels_ms_per_step': {
                    'advance': adv_ms / max(adv_n, 1),
                    'prefix': pre_ms / max(adv_n, 1)},
            },
        }
        if ep is not None:
            line['whole_episode'] = ep
        if getattr(env, '_sh_tuned', None):
            # gather time (ms per launch at 131 072 streamlines) of every pair
            # (allocation of the SH volume, allocation of the state ring) tried at the
            # first large reset; the fastest pair was kept (env.py:_tune_placement)
            line['placement_candidates_ms'] = env._sh_tuned
        if collate_ms is not None:
            line['collate_ms'] = collate_ms
            line['collate_bytes_to_root'] = collate_bytes
        if collate_error is not None:
            line['collate_error'] = collate_error
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
