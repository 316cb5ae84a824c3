#!/usr/bin/env python3
"""bench.py -- streamline-steps/s of the MI355X environment step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--windows M] [--legs ...]

Headline workload (BASELINE.json configs[1]): "env.step only" on
a synthetic 96^3 x 45-SH volume, n_actor = 262144 streamlines per GPU, ball
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
Every fourth window also brackets the dominant kernel with HIP events (for
`roofline`); those records cost a few percent, both medians are printed.

With N > 1 the driver launches one process per GPU (torch.distributed.run);
called bare with --gpus N > 1 this script launches those N ranks itself (as
child processes, before this process touches the GPU) and relays rank 0's
line.  Streamlines shard across ranks with the volumes replicated and no
collective on the step path.  The one exchange the path has -- collating
finished tracts on rank 0 (exact-size gather to root over RCCL) -- is timed
as `collate_ms`.

Legs (all of them by default; `--legs weak,hbm` etc. restricts, which is how
the rocprofv3 passes under profiles/ look at one workload at a time):

  weak     the headline: 262144 streamlines PER GPU ("scaling": "weak").
           -> `value`, `windows`, `roofline`, `whole_episode`, `collate_ms`.
  strong   BASELINE's metric as stated, "at n_actor = 262144": 262144
           streamlines in TOTAL, rank r tracks the contiguous shard
           shard_bounds(262144, r, N) of one global seed batch.  -> `strong`.
           (At N = 1 weak and strong are the same run.)
  config4  BASELINE configs[3]: 145^3 x 45 volume (585 MB packed: does not fit
           the 256 MB Infinity Cache), 1048576 streamlines in total sharded
           over the N ranks (131072 per GPU at N = 8), n_dirs = 100, float64
           directions (NoisyTrackingEnvironment, sigma 0), max_length 300 mm.
           Step-only windows as above, and one whole tractogram end to end:
           every streamline tracked to exhaustion, then the finished tracts
           collated on rank 0 -- `collate_ms` INCLUDED in the end-to-end
           streamline-steps/s printed next to the step-only one.  -> `config4`.
  pipelined  the headline batch as TWO half-batches of 131072 software-pipelined
           on two HIP streams: the caller evaluates the policy per half, so the
           latency-bound small kernels of one half (policy, advance, index
           compaction) run under the state gather of the other.  Same streamlines,
           same kernels, same results; only the schedule differs.  NOT the
           headline (`value` stays the plain one-batch loop of the reference's
           contract) -> `pipelined_halves`.
  shapes   (N = 1 only) the other BASELINE shapes through the same loop, three
           windows each, so that they are driver-timed too: config 2 at K = 100
           (the shipped model's state width), config 3's env side (65536,
           reward on), config 1's shape (32^3, 4096, K = 100, float64
           directions) and config 2 through the reference's OWN calling contract
           (`env.step(numpy)` -> host reward / dones -> `harvest()`, PCIe
           inclusive; never `value`).  -> `other_shapes`.
  learner  (N = 1 only) BASELINE configs[2]: one SAC training step at
           n_actor = 65536, hidden 1024-1024, batch 4096, float32 (policy
           forward -> env step with the alignment reward -> replay add ->
           sample -> SACAuto.update -> harvest), phase by phase (HIP events),
           and the update alone with its FLOP roofline.  The GEMMs are
           PyTorch-ROCm's (hipBLASLt fp32 MFMA); everything else of the update
           and the env step are this library's kernels.  -> `config3_training`.
  config5  BASELINE configs[4]: the same training step with oracle_bonus 10 and
           the oracle stopping criterion on (TractOracle-Net transformer,
           random-init weights), 131072 streamlines in total sharded over the N
           ranks (16384 per GPU at N = 8), learner replicas data-parallel (one
           all-reduce per parameter arena and update over RCCL).  At N = 1 the
           N = 8 shard (16384) and the whole batch (131072) are both timed.
           Phase by phase: policy, env step (of which k_resample and the
           transformer), replay add / sample, update (of which the gradient
           all-reduce), harvest.  -> `config5`.
  hbm      the `roofline` object again in the regime where HBM binds: one
           GPU's shard of config 4 at N = 8 (131072 streamlines on the 145^3
           volume).  -> `roofline_hbm_regime` (rank 0's kernel times).

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
               streamlines sharded over worker processes).  N = 1 only.
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

C = 45
STEP_MM = 0.75
THETA = 30.0
WOBBLE = 0.05
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DOMINANT_KERNEL = 'k_state_dd<12,4,false,true>'

#: the two volumes the legs run on (BASELINE.json configs[1] and configs[3])
WORKLOADS = {
    'c2': dict(D=96, n_total=262144, n_dirs=4, noisy=False, max_length=200.0,
               what='96^3x45-SH synthetic volume, n_dirs=4, float32 directions '
                    '(train env), max_length 200 mm'),
    'c4': dict(D=145, n_total=1048576, n_dirs=100, noisy=True, max_length=300.0,
               what='145^3x45-SH synthetic volume (ISMRM2015-shaped), n_dirs=100, float64 '
                    'directions (NoisyTrackingEnvironment, sigma 0), max_length 300 mm'),
}
D = WORKLOADS['c2']['D']
N_ACTOR = WORKLOADS['c2']['n_total']
N_DIRS = WORKLOADS['c2']['n_dirs']
MAX_LENGTH = WORKLOADS['c2']['max_length']
#: rows of the HBM-regime roofline leg: one GPU's shard of config 4 at N = 8
HBM_LEG_ROWS = 131072
LEGS = ('weak', 'strong', 'config4', 'hbm', 'pipelined', 'shapes', 'learner', 'config5')


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


#: leg `shapes`: the other BASELINE shapes (same loop; `host`: the reference's
#: own calling contract instead of the device-resident loop)
OTHER_SHAPES = {
    'c2_K100': dict(D=96, n_total=262144, n_dirs=100, noisy=False, max_length=200.0,
                    what='config 2 at n_dirs=100 (state width 615, the shipped model\'s)'),
    'c3_env': dict(D=96, n_total=65536, n_dirs=4, noisy=False, max_length=200.0, reward=True,
                   what='config 3, env side: 65536 streamlines, alignment reward on'),
    'c1_shape': dict(D=32, n_total=4096, n_dirs=100, noisy=True, max_length=300.0,
                     what='config 1\'s shape: 32^3, 4096 streamlines, n_dirs=100, float64 '
                          'directions'),
    'c2_host_contract': dict(D=96, n_total=262144, n_dirs=4, noisy=False, max_length=200.0,
                             host=True,
                             what='config 2 through env.step(numpy actions) -> host reward / '
                                  'dones -> env.harvest() (rl.py:93-102), PCIe inclusive'),
}


def make_subject(workload='c2'):
    from tracktolearn_amd.utils.synthetic import synthetic_subject
    w = WORKLOADS.get(workload) or OTHER_SHAPES[workload]
    return synthetic_subject(w['D'], C, seed=1234, peaks=bool(w.get('reward')),
                             affine_dtype=np.float64 if w['noisy'] else np.float32)


def make_env(subject, device, workload='c2'):
    """The environment of a workload; the caller sets `env.seeds`."""
    import torch
    from tracktolearn_amd.environments import (NoisyTrackingEnvironment,
                                               TrackingEnvironment)
    w = WORKLOADS.get(workload) or OTHER_SHAPES[workload]
    dto = dict(n_dirs=w['n_dirs'], theta=THETA, npv=1, binary_stopping_threshold=0.1,
               step_size=STEP_MM, min_length=20.0, max_length=w['max_length'],
               compute_reward=bool(w.get('reward')), alignment_weighting=1.0, oracle_bonus=0.0,
               rng=np.random.RandomState(0), device=torch.device(device),
               target_sh_order=8, noise=0.0, fa_map=None)
    cls = NoisyTrackingEnvironment if w['noisy'] else TrackingEnvironment
    return cls(subject, 'testing', dto)


def shard_seeds(mask_data, n_total, rank, world, seed=100):
    """Rank's contiguous shard of ONE global batch of n_total seeds
    (tracktolearn_amd.parallel.shard_bounds)."""
    from tracktolearn_amd.parallel import shard_bounds
    from tracktolearn_amd.utils.synthetic import synthetic_seeds
    seeds = synthetic_seeds(mask_data, n_total, seed=seed)
    lo, hi = shard_bounds(n_total, rank, world)
    return seeds[lo:hi]


def run_steps(env, n_steps, seed, counter, rows, host_contract=False):
    """n_steps passes of the hot path; returns streamline-steps processed.
    `host_contract`: the reference's own calling sequence (rl.py:93-102): the
    action batch goes to the host and into `env.step` as a numpy array."""
    state = counter['state']
    total = 0
    for _ in range(n_steps):
        if env._n_active == 0:
            state = env.reset(0, rows)
            counter['step'] = 0
            counter['resets'] += 1
        n = env._n_active
        actions = env.scripted_actions(state, counter['step'], seed, WOBBLE)
        if host_contract:
            env.step(actions.to(device='cpu', copy=True).numpy())
        else:
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
    # every core this process may use (capped by memory: ~0.3 GB per worker
    # beside the shared volume); the sample grows with the workers so that
    # each one has work for about as long as the single-thread run
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:     # pragma: no cover
        usable = os.cpu_count() or 1
    try:
        avail_gb = os.sysconf('SC_AVPHYS_PAGES') * os.sysconf('SC_PAGE_SIZE') / 2 ** 30
    except (ValueError, OSError):      # pragma: no cover
        avail_gb = 16.0
    workers = max(1, min(usable, int(avail_gb / 0.5)))
    cpu_model = ''
    try:
        for row in open('/proc/cpuinfo'):
            if row.startswith('model name'):
                cpu_model = row.split(':', 1)[1].strip()
                break
    except OSError:            # pragma: no cover
        pass
    one['cpu_model'] = cpu_model
    one['usable_cpus'] = usable
    def sharded(n_workers):
        """The sample sharded over n_workers forked single-threaded oracles."""
        n_multi = int(min(N_ACTOR, max(n_sample, 1024 * n_workers)))
        seeds = synthetic_seeds(mask_data, n_multi, seed=100)
        _FORK_SHARED.update(sh=sh, mask=mask_data, seeds=seeds)
        per = -(-n_multi // n_workers)
        jobs = [(w * per, min((w + 1) * per, n_multi), n_steps)
                for w in range(n_workers) if w * per < n_multi]
        t0 = time.perf_counter()
        with mp.get_context('fork').Pool(len(jobs)) as pool:
            parts = pool.map(_oracle_worker, jobs)
        wall = time.perf_counter() - t0
        units = sum(p[0] for p in parts)
        busy = max(p[1] for p in parts)
        return {'workers': len(jobs), 'streamlines': n_multi, 'value': units / busy,
                'value_wall': units / wall, 'wall_s': wall}

    multi = None
    try:
        # one worker per usable cpu, and -- because that many single-threaded
        # numpy / scipy oracles contend for memory bandwidth and SMT siblings --
        # a quarter and a sixteenth of them; the best of the three is the figure
        counts = sorted({workers, max(1, workers // 4), max(1, min(16, workers))})
        tried = [sharded(k) for k in counts]
        best = max(tried, key=lambda t: t['value'])
        multi = {'value': best['value'], 'unit': 'streamline-steps/s',
                 'cores': best['workers'], 'kind': 'port', 'cpu_model': cpu_model,
                 'usable_cpus': usable, 'tried': tried,
                 'sample': f"{best['streamlines']} of the {N_ACTOR} streamlines, first "
                           f"{n_steps} steps, sharded over forked worker processes (one "
                           f"single-threaded oracle each) -- {', '.join(str(t['workers']) for t in tried)} "
                           f"workers tried on {usable} usable of {os.cpu_count()} host cpus, the "
                           f"best kept ({best['workers']}); units / slowest worker's "
                           f"step()+harvest() time; value_wall includes process start and "
                           f"per-worker setup"}
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


# --------------------------------------------------------------------------
# one leg's measurements
# --------------------------------------------------------------------------
class Dist:
    """The process group as the bench uses it: barrier, max / sum over ranks."""

    def __init__(self, world, red_dev):
        self.world, self.red_dev = world, red_dev

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    def reduce(self, values, op):
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64,
                         device=self.red_dev)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 'max' else dist.ReduceOp.SUM)
        return t.cpu().numpy()


def timed_windows(env, rows, steps, warmup, n_win, seed, grp, host_contract=False):
    """`warmup` untimed steps, then `n_win` windows of exactly `steps` steps,
    each from a fresh untimed reset and bracketed by barrier + synchronize.
    Returns a dict: per-window wall times (max over ranks), streamline-steps of
    one window (sum over ranks), the dominant kernel's event time on THIS rank."""
    import torch
    counter = {'state': env.reset(0, rows), 'step': 0, 'resets': 0}
    run_steps(env, warmup, seed, counter, rows, host_contract)
    # exercise the periodic re-sort of the processing order once outside the
    # timed regions
    if env._n_active:
        env._refresh_processing_order(force=True)
    torch.cuda.synchronize()
    # The dominant kernel is bracketed with HIP events in every fourth window
    # only (3 of the default 11: 36 launches): an event record costs ~6 us of
    # GPU idle time on either side of the kernel (rocprofv3 trace,
    # benchmarks/trace_gaps.py), ~3-5 % of a step.  All windows are timed alike
    # and the value is the median over all of them; both sub-medians are printed.
    times, n_units, resets = [], 0, 0
    state_ms, state_n, evented = 0.0, 0, []
    for w in range(n_win):
        counter = {'state': env.reset(0, rows), 'step': 0, 'resets': 0}
        with_events = (w % 4 == 1) or n_win == 1
        if with_events:
            env.profile_begin(max_launches=steps + 8, classes=('state',))
        grp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_units = run_steps(env, steps, seed, counter, rows, host_contract)
        torch.cuda.synchronize()
        grp.barrier()
        times.append(time.perf_counter() - t0)
        resets = max(resets, counter['resets'])
        if with_events:
            ms, cnt = env.profile_end()['state']
            state_ms += ms
            state_n += cnt
            evented.append(w)
    t_all = grp.reduce(times, 'max')            # per window, max over the ranks
    total_units = float(grp.reduce([n_units], 'sum')[0])     # of ONE window, all ranks
    t_sorted = np.sort(t_all)
    t_med = float(t_sorted[len(t_sorted) // 2])
    rest = [i for i in range(len(t_all)) if i not in evented]
    return {
        'value': total_units / t_med, 'ms_per_step': t_med / steps * 1e3,
        'streamline_steps': total_units, 'rank0_units': n_units, 'resets': resets,
        'state_ms': state_ms, 'state_n': state_n,
        'windows': {
            'n': len(t_all), 'timed': 'each window = exactly --steps steps '
            'from a fresh untimed reset; value/ms_per_step = median window',
            'with_kernel_events': evented,
            'value_median_with_events': (total_units / float(np.median(
                [t_all[i] for i in evented]))) if evented else None,
            'value_median_without_events': (total_units / float(np.median(
                [t_all[i] for i in rest]))) if rest else None,
            'value_min': total_units / float(t_sorted[-1]),
            'value_median': total_units / t_med,
            'value_max': total_units / float(t_sorted[0]),
            'ms_per_step_min': float(t_sorted[0]) / steps * 1e3,
            'ms_per_step_max': float(t_sorted[-1]) / steps * 1e3,
        },
    }


def pipelined_windows(subject, device, seeds, parts, steps, warmup, n_win, seed, grp):
    """`parts` envs over the same volumes, each with 1/parts of the seeds and a
    HIP stream of its own, stepped alternately: windows of exactly `steps` steps
    of EVERY part, bracketed like the headline's."""
    import torch
    rows = len(seeds) // parts
    envs, streams = [], []
    for k in range(parts):
        env = make_env(subject, device, 'c2')
        env.seeds = seeds[k * rows:(k + 1) * rows]
        envs.append(env)
        streams.append(torch.cuda.Stream())

    def window(n_steps):
        states = []
        for env, s in zip(envs, streams):
            with torch.cuda.stream(s):
                states.append(env.reset(0, rows))
        grp.barrier()
        torch.cuda.synchronize()
        total, t0 = 0, time.perf_counter()
        for step in range(n_steps):
            for k, (env, s) in enumerate(zip(envs, streams)):
                with torch.cuda.stream(s):
                    total += env._n_active
                    env.step_device(env.scripted_actions(states[k], step, seed, WOBBLE))
            for k, (env, s) in enumerate(zip(envs, streams)):
                with torch.cuda.stream(s):
                    states[k], _ = env.harvest()
        torch.cuda.synchronize()
        grp.barrier()
        return total, time.perf_counter() - t0

    window(max(warmup, 1))
    times, units = [], 0
    for _ in range(n_win):
        units, dt = window(steps)
        times.append(dt)
    t_all = np.sort(grp.reduce(times, 'max'))
    total_units = float(grp.reduce([units], 'sum')[0])
    t_med = float(t_all[len(t_all) // 2])
    return {'parts': parts, 'rows_per_part': rows, 'value': total_units / t_med,
            'ms_per_step': t_med / steps * 1e3, 'streamline_steps': total_units,
            'value_min': total_units / float(t_all[-1]),
            'value_max': total_units / float(t_all[0]), 'windows': len(t_all)}


def other_shape(name, w, subject, device, args, seed, grp):
    """One entry of the `shapes` leg: three windows of OTHER_SHAPES[name]."""
    from tracktolearn_amd.utils.synthetic import synthetic_seeds
    subj = subject if (subject is not None and w['D'] == D and not w.get('reward')
                       and not w['noisy']) else make_subject(name)
    env = make_env(subj, device, name)
    env.seeds = synthetic_seeds(subj[1].data, w['n_total'], seed=100)
    r = timed_windows(env, w['n_total'], args.steps, args.warmup, 3, seed, grp,
                      host_contract=bool(w.get('host')))
    return {
        'what': w['what'], 'volume': [w['D']] * 3 + [C], 'n_actor': w['n_total'],
        'n_dirs': w['n_dirs'],
        'loop': 'step(numpy) + harvest' if w.get('host') else 'step_device + harvest',
        'value': r['value'], 'ms_per_step': r['ms_per_step'],
        'value_min': r['windows']['value_min'], 'value_max': r['windows']['value_max'],
        'k_state_ms': r['state_ms'] / max(r['state_n'], 1), 'windows': 3,
    }


def kernel_breakdown(env, rows, steps, seed):
    """Untimed replay of a window with every kernel class bracketed (the timed
    windows only bracket the dominant kernel, to keep the event records out of
    the other launch gaps): average ms per step of the other kernels -- the
    step's own (advance, prefix, proc_scatter: library events) and the scripted
    policy in front of it (torch events on the same stream)."""
    import torch
    state = env.reset(0, rows)
    classes = tuple(env.PROFILE_CLASSES)
    env.profile_begin(max_launches=max(16, steps + 8), classes=classes)
    policy_ev = []
    for step in range(steps):
        if env._n_active == 0:
            break
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        actions = env.scripted_actions(state, step, seed, WOBBLE)
        e1.record()
        policy_ev.append((e0, e1))
        env.step_device(actions)
        state, _ = env.harvest()
    torch.cuda.synchronize()
    prof = env.profile_end()
    n_steps = max(prof[classes[0]][1], 1)
    out = {k: prof[k][0] / n_steps for k in classes if k != 'state'}
    out['scripted_policy'] = sum(a.elapsed_time(b) for a, b in policy_ev) / max(len(policy_ev), 1)
    out['sum'] = sum(out.values())
    # which tail ran: `prefix` is k_tail (rows + slots in one launch, no
    # proc_scatter launches) or k_prefix next to k_proc_scatter
    out['step_tail'] = 'k_tail' if prof['proc_scatter'][1] == 0 else 'k_prefix + k_proc_scatter'
    return out


def track_to_exhaustion(env, state, seed, free_tail):
    """Steps a freshly reset env until no streamline is active; returns (steps,
    how many of them were free-running)."""
    ep_steps, free_steps = 0, 0
    while env._n_active:
        if free_tail and env.freerun_supported():
            # from 16 384 rows down a step is bound by the host waiting
            # for its survivor count: free-running steps, launched for
            # the newest count the GPU has reported, never waited for
            _, free_steps = env.run_free_eager(
                lambda st: env.scripted_actions_free(st, seed, WOBBLE), state)
            ep_steps += free_steps
            break
        actions = env.scripted_actions(state, ep_steps, seed, WOBBLE)
        env.step_device(actions)
        state, _ = env.harvest()
        ep_steps += 1
    return ep_steps, free_steps


def whole_episode(env, rows, seed, free_tail, grp=None):
    """One episode to exhaustion, the second of two (the first
    one of a process runs slower: allocator growth and first-use effects a
    tracking run pays once, on its first seed batch).  With `grp` the episode is
    bracketed by barriers and the time is the max over ranks."""
    import torch
    ep = None
    for attempt in range(2):
        state = env.reset(0, rows)
        torch.cuda.synchronize()
        if grp is not None:
            grp.barrier()
        t_ep = time.perf_counter()
        ep_steps, free_steps = track_to_exhaustion(env, state, seed, free_tail)
        torch.cuda.synchronize()
        t_ep = time.perf_counter() - t_ep
        # streamline-steps = points added = sum(lengths - 1)
        ep_units = int(env._buf_lengths[:rows].sum().item()) - rows
        first = ep
        ep = {'streamline_steps_per_s_rank0': ep_units / t_ep, 'steps': ep_steps,
              'streamline_steps': ep_units, 'ms': t_ep * 1e3,
              'free_running_tail_steps': free_steps,
              'order_refresh_every': env.SPATIAL_ORDER_REFRESH}
        if first is not None:
            ep['first_episode_ms'] = first['ms']
    return ep


def collate(env, grp):
    """The path's one exchange: every rank's finished tracts gathered on rank 0
    (exact sizes, RCCL point-to-point).  With one rank there is nothing to
    exchange: the device-side ragged pack (`tract_arrays`) is what is timed.
    Returns (ms, bytes received by the root, error or None)."""
    import torch
    from tracktolearn_amd.parallel import gather_tract_arrays, tract_arrays
    try:
        grp.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        moved = 0
        if grp.world > 1:
            got = gather_tract_arrays(env)
            if got is not None:
                moved = got[3]
        else:
            tract_arrays(env)
        torch.cuda.synchronize()
        grp.barrier()
        return (time.perf_counter() - t1) * 1e3, moved, None
    except Exception as exc:      # never lose the bench line to the collate
        return None, None, repr(exc)


def roofline_object(win, workload, mask_data, pmc_path, cache_note):
    """The `roofline` object of a leg from its timed windows (rank 0's kernel
    events) and the per-unit PMC bytes of `pmc_path`."""
    w = WORKLOADS[workload]
    k = w['n_dirs']
    whole_b, kern_b = algorithmic_bytes(C, k)
    steps = max(win['steps'], 1)
    avg_launch_s = win['state_ms'] / max(win['state_n'], 1) * 1e-3
    units_per_launch = win['rank0_units'] / steps
    n_mask = int(np.count_nonzero(mask_data))
    comp_b = compulsory_bytes(C, k, n_mask, units_per_launch)
    pmc, pmc_src, kernel = None, None, DOMINANT_KERNEL
    for path in pmc_path:
        full = os.path.join(ROOT, 'profiles', path)
        if not os.path.exists(full):
            continue
        try:
            js = json.load(open(full))
            pmc = js.get('k_state_hbm_bytes_per_unit')
            kernel = js.get('k_state_kernel', kernel)
            pmc_src = (f"profiles/{path} <- {js.get('source')}: PMC FETCH_SIZE/WRITE_SIZE "
                       f"passes of this command (`--legs` restricted to this leg), "
                       f"(2*FETCH+WRITE)*1024 bytes per unit x this run's units per "
                       f"launch; NOT measured in this run.  {cache_note}")
            break
        except Exception:
            pmc = None
    if not avg_launch_s:
        return None
    traffic = pmc * units_per_launch if pmc else None
    achieved = traffic / avg_launch_s / 1e9 if traffic else None
    return {
        'bound': 'hbm', 'kernel': kernel,
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
        'compulsory_frac': comp_b * units_per_launch / avg_launch_s / 1e9 / HBM_PEAK_GBS,
        # all 56 corner fetches charged per unit -- an
        # upper bound on naive traffic, not a roofline fraction
        'algorithmic_bytes_per_unit': kern_b,
        'algorithmic_GBs': kern_b * units_per_launch / avg_launch_s / 1e9,
        'units_per_launch': units_per_launch,
        'avg_launch_ms': avg_launch_s * 1e3,
        'launches': win['state_n'],
        'whole_step_algorithmic_bytes_per_unit': whole_b,
        'workload': w['what'],
    }


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--windows', type=int, default=11,
                    help='timed windows of --steps steps each (median reported)')
    ap.add_argument('--legs', default='all',
                    help='comma list of ' + ','.join(LEGS) + ' (default: all)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-whole-episode', action='store_true',
                    help='skip the episode-to-exhaustion figure (SURVEY 8d (ii))')
    ap.add_argument('--whole-episode', action='store_true',
                    help='(default now; kept for older command lines)')
    args = ap.parse_args(argv)
    legs = set(LEGS) if args.legs == 'all' else set(args.legs.split(','))
    if not legs <= set(LEGS):
        sys.exit(f'--legs: unknown leg in {sorted(legs)} (known: {LEGS})')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args, argv)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    need_c2 = bool(legs & {'weak', 'strong', 'pipelined'})
    need_c4 = bool(legs & {'config4', 'hbm'})
    subject = make_subject('c2') if need_c2 or not args.no_cpu_baseline else None
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
    grp = Dist(world, device if backend == 'nccl' else 'cpu')
    seed = 1 + rank
    free_tail = os.environ.get('TTL_BENCH_FREE_TAIL', '1') != '0'
    n_win = max(1, args.windows)
    # rehearsals on small boxes may shrink config 4 (never the driver's runs)
    c4_total = int(os.environ.get('TTL_BENCH_C4_TOTAL', WORKLOADS['c4']['n_total']))

    line = {
        'metric': 'streamline-steps/s at n_actor=262144',
        'value': None, 'unit': 'streamline-steps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
    }
    out = {}          # rank 0 assembles the line from this at the end

    # ======================= 96^3 volume: weak + strong ====================
    if need_c2:
        from tracktolearn_amd.utils.synthetic import synthetic_seeds
        env = make_env(subject, device, 'c2')
        mask_c2 = subject[1].data
        if 'weak' in legs:
            env.seeds = synthetic_seeds(mask_c2, N_ACTOR, seed=100 + rank)
            weak = timed_windows(env, N_ACTOR, args.steps, args.warmup, n_win, seed, grp)
            weak['steps'] = args.steps
            weak['other_kernels_ms_per_step'] = kernel_breakdown(env, N_ACTOR, args.steps, seed)
            if not args.no_whole_episode:
                weak['whole_episode'] = whole_episode(env, N_ACTOR, seed, free_tail)
            if world > 1:
                weak['collate'] = collate(env, grp)
            out['weak'] = weak
        if 'strong' in legs:
            if world == 1 and 'weak' in out:
                # one rank: the shard IS the whole batch -- the same run
                strong = dict(out['weak'], same_run_as_value=True)
            else:
                env.seeds = shard_seeds(mask_c2, N_ACTOR, rank, world)
                rows = len(env.seeds)
                strong = timed_windows(env, rows, args.steps, args.warmup, n_win, seed, grp)
                strong['rows_rank0'] = rows
            out['strong'] = strong
        out['placement'] = (getattr(env, '_sh_tuned', None),
                            getattr(env, '_placement_search', None))
        del env
        torch.cuda.empty_cache()
        if 'pipelined' in legs:
            out['pipelined'] = pipelined_windows(
                subject, device, synthetic_seeds(mask_c2, N_ACTOR, seed=100 + rank), 2,
                args.steps, args.warmup, n_win, seed, grp)
            torch.cuda.empty_cache()

    # ======================= the other BASELINE shapes (N = 1) =============
    if 'shapes' in legs and world == 1:
        shapes = {}
        for name, w in OTHER_SHAPES.items():
            try:          # auxiliary one-GPU leg: never lose the headline line to it
                shapes[name] = other_shape(name, w, subject, device, args, seed, grp)
            except Exception as exc:
                shapes[name] = {'what': w['what'], 'error': repr(exc)}
            torch.cuda.empty_cache()
        out['shapes'] = shapes

    # ======================= config 3: one SAC training step (N = 1) =======
    if 'learner' in legs and world == 1:
        # an auxiliary one-GPU leg: never lose the headline line to it
        try:
            from benchmarks.bench_training import measure as training_measure
            out['learner'] = training_measure('c3', device=device)
            torch.cuda.empty_cache()
            # the same with SACAuto.update replayed from a HIP graph (enable_graph())
            graphed = training_measure('c3', device=device, graph=True)
            out['learner']['graphed_update'] = {k: graphed[k] for k in
                                                ('update_ms', 'train_step_ms',
                                                 'train_streamline_steps_per_s')}
        except Exception as exc:
            out['learner'] = dict(out.get('learner') or {}, error=repr(exc))
        torch.cuda.empty_cache()

    # ======================= config 5: training with the oracle, data parallel
    if 'config5' in legs:
        try:
            from benchmarks.bench_training import measure as training_measure
            c5_total = int(os.environ.get('TTL_BENCH_C5_TOTAL', 131072))
            shard = c5_total // 8 if world == 1 else -(-c5_total // world)
            grp.barrier()
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):     # stdout is the one JSON line
                c5 = training_measure('c5', n_actor=shard, device=device,
                                      data_parallel=world > 1, seed_offset=rank)
            grp.barrier()
            # whole job: the slowest rank's step, every rank's streamline-steps
            ms = float(grp.reduce([c5['train_step_ms']], 'max')[0])
            rows = float(grp.reduce([c5['train_rows_per_step']], 'sum')[0])
            c5['train_step_ms_max_over_ranks'] = ms
            c5['value'] = rows / (ms * 1e-3)
            c5['n_actor_total'] = shard * world
            if world == 1:
                torch.cuda.empty_cache()
                with contextlib.redirect_stdout(sys.stderr):
                    whole = training_measure('c5', n_actor=c5_total, device=device)
                c5['whole_batch_on_one_gpu'] = {
                    k: whole[k] for k in ('n_actor', 'update_ms', 'train_step_ms',
                                          'train_rows_per_step', 'train_streamline_steps_per_s',
                                          'phases_ms_per_step', 'oracle_rows_scored_per_step',
                                          'oracle_batches_per_step', 'policy_forward')
                    if k in whole}
                # TractOracle-Net alone (the checkpoint's architecture, random weights): the
                # batch a training step scores and a large one, against the reference's
                # formulation (the PyTorch module under autocast) at the large one
                from benchmarks.bench_oracle_net import measure as oracle_net_measure
                torch.cuda.empty_cache()
                nets = oracle_net_measure([256], with_module=False, device=device) + \
                    oracle_net_measure([16384], with_module=True, device=device)
                big = nets[-1]
                c5['oracle_net_alone'] = {
                    'per_batch': nets,
                    'roofline': {'bound': 'mfma', 'kernel': 'k_oracle_net<4> (16384 streamlines)',
                                 'achieved': big['fused_TFLOPs_issued'], 'peak': 2500.0,
                                 'unit': 'TFLOP/s', 'frac': big['fused_frac_of_fp16_mfma_peak'],
                                 'traffic': None, 'dtype': 'f16',
                                 'how': 'FLOP the kernel issues (119.9 MFLOP per streamline; the '
                                        'module: 146.8) / wall clock of 5 launches / the dense '
                                        'fp16 MFMA peak'}}
            out['config5'] = c5
        except Exception as exc:
            out['config5'] = {'error': repr(exc)}
        torch.cuda.empty_cache()

    # ======================= 145^3 volume: config 4 + HBM regime ===========
    if need_c4:
        from tracktolearn_amd.utils.synthetic import synthetic_seeds
        t_setup = time.perf_counter()
        subject4 = make_subject('c4')
        mask_c4 = subject4[1].data
        env4 = make_env(subject4, device, 'c4')
        setup_s = time.perf_counter() - t_setup
        from tracktolearn_amd.parallel import shard_bounds
        lo, hi = shard_bounds(c4_total, rank, world)
        rows4 = hi - lo
        hbm = None
        if 'hbm' in legs and not ('config4' in legs and rows4 == HBM_LEG_ROWS):
            # the shard a GPU holds at N = 8, first: its reset is what places
            # the volume and the ring of state buffers
            env4.seeds = synthetic_seeds(mask_c4, HBM_LEG_ROWS, seed=100 + rank)
            hbm = timed_windows(env4, HBM_LEG_ROWS, args.steps, args.warmup, n_win, seed, grp)
            hbm['steps'] = args.steps
        if 'config4' in legs:
            env4.seeds = shard_seeds(mask_c4, c4_total, rank, world)
            c4 = timed_windows(env4, rows4, args.steps, args.warmup, n_win, seed, grp)
            c4['steps'] = args.steps
            c4['rows_rank0'] = rows4
            if 'hbm' in legs and hbm is None:
                hbm = c4
            # one whole tractogram end to end: track every streamline to
            # exhaustion, then collate on rank 0 -- timed as one region
            warm = whole_episode(env4, rows4, seed, free_tail, grp)
            # ... and one untimed collate of that warm-up tractogram, like the W
            # warm-up steps: the first call pays for loading the code objects of
            # the pack / index kernels (about 0.2 s at N = 1) and, with N > 1,
            # for RCCL setting up its point-to-point connections
            first_ms, _, _ = collate(env4, grp)
            grp.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ep_steps, _ = track_to_exhaustion(env4, env4.reset(0, rows4), seed, free_tail)
            torch.cuda.synchronize()
            t_track = time.perf_counter() - t0
            c_ms, c_bytes, c_err = collate(env4, grp)
            t_total = time.perf_counter() - t0
            ep_units = int(env4._buf_lengths[:rows4].sum().item()) - rows4
            units = float(grp.reduce([ep_units], 'sum')[0])
            t_track_max = float(grp.reduce([t_track], 'max')[0])
            t_total_max = float(grp.reduce([t_total], 'max')[0])
            c4['end_to_end'] = {
                'what': 'reset -> every streamline tracked to exhaustion (scripted policy) '
                        '-> finished tracts collated on rank 0; one timed region, max '
                        'over ranks',
                'streamline_steps': units, 'episode_steps_rank0': ep_steps,
                'track_ms': t_track_max * 1e3, 'collate_ms': c_ms,
                'collate_first_call_ms': first_ms,
                'collate_bytes_to_root': c_bytes, 'end_to_end_ms': t_total_max * 1e3,
                'value_step_only': units / t_track_max,
                'value_end_to_end': units / t_total_max,
                'warm_episode_rank0': warm,
            }
            if c_err:
                c4['end_to_end']['collate_error'] = c_err
            if world == 1:
                c4['end_to_end']['collate'] = ('one rank: nothing to exchange; collate_ms '
                                               'is the device-side ragged pack of the tracts')
            out['config4'] = c4
        if hbm is not None:
            out['hbm'] = hbm
        out['c4_setup_s'] = setup_s
        out['c4_placement'] = (getattr(env4, '_sh_tuned', None),
                               getattr(env4, '_placement_search', None))
        out['mask_c4'] = mask_c4

    if rank == 0:
        note_c2 = ('FETCH_SIZE counts Infinity-Cache hits: the 170 MB packed 96^3 volume '
                   'fits the 256 MB cache, so this is fabric traffic, of which the read '
                   'share is mostly cache-served, not DRAM traffic (see '
                   'roofline_hbm_regime for a volume that does not fit).')
        note_c4 = ('The 585 MB packed 145^3 volume does not fit the 256 MB Infinity '
                   'Cache: reads go to HBM.')
        weak = out.get('weak')
        head = weak or out.get('strong') or out.get('config4') or out.get('hbm')
        if weak:
            line['value'] = weak['value']
            line['ms_per_step'] = weak['ms_per_step']
        line['config'] = {
            'workload': 'env.step only (scripted actions -> step -> '
                        'harvest), 96^3x45-SH synthetic volume, '
                        'n_actor=262144 per GPU, 1xMI355X per rank',
            'n_actor_per_gpu': N_ACTOR, 'volume': [D, D, D, C],
            'n_dirs': N_DIRS, 'state_width': 7 * C + 3 * N_DIRS,
            'reward': False, 'arithmetic': 'float32 directions (train env)',
            'loop': 'step_device + harvest (survivors-first rows, 8-byte '
                    'count readback per step)',
            'resets_in_timed_region': weak['resets'] if weak else None,
            'parallelism': f'streamlines sharded over {world} GPU(s), '
                           'volumes replicated',
            'legs': sorted(legs),
        }
        if weak:
            line['streamline_steps'] = weak['streamline_steps']
            line['windows'] = weak['windows']
            roof = roofline_object(weak, 'c2', subject[1].data, ['pmc_traffic.json'], note_c2)
            if roof is not None:
                roof['whole_step_algorithmic_GBs'] = \
                    roof['whole_step_algorithmic_bytes_per_unit'] * weak['value'] / world / 1e9
                roof['other_kernels_ms_per_step'] = weak['other_kernels_ms_per_step']
            line['roofline'] = roof
            if weak.get('whole_episode'):
                line['whole_episode'] = weak['whole_episode']
            if 'collate' in weak:
                line['collate_ms'], line['collate_bytes_to_root'] = weak['collate'][:2]
                if weak['collate'][2]:
                    line['collate_error'] = weak['collate'][2]
        if 'strong' in out:
            s = out['strong']
            line['strong'] = {
                'what': 'BASELINE metric as stated: 262144 streamlines in TOTAL, one global '
                        'seed batch sharded contiguously over the ranks '
                        '(parallel.shard_bounds), same 96^3 volume and loop as `value`',
                'scaling': 'strong', 'n_actor_total': N_ACTOR,
                'n_actor_per_gpu': -(-N_ACTOR // world),
                'value': s['value'], 'ms_per_step': s['ms_per_step'],
                'streamline_steps': s['streamline_steps'],
                'value_min': s['windows']['value_min'], 'value_max': s['windows']['value_max'],
                'windows': s['windows']['n'],
                'same_run_as_value': bool(s.get('same_run_as_value', False)),
            }
        if 'pipelined' in out:
            pl = out['pipelined']
            line['pipelined_halves'] = dict(
                pl, what='the same 262144 streamlines per GPU as two half-batches, each with '
                         'its own env handle and HIP stream, stepped alternately (policy per '
                         'half): the small kernels of one half run under the gather of the '
                         'other; not the headline loop',
                vs_value=(pl['value'] / weak['value']) if weak else None)
        if 'learner' in out:
            line['config3_training'] = dict(
                out['learner'],
                what='BASELINE configs[2]: SAC (automatic entropy), hidden 1024-1024, '
                     'n_actor=65536, batch 4096, 96^3x45 volume, n_dirs=4, alignment reward; '
                     'train_step_ms = policy forward + env step + replay add + sample + '
                     'update + harvest, 24 steps after warm-up (freshly initialised policy); '
                     'update_ms = SACAuto.update alone (fused schedule: 16 fp32 GEMMs + the '
                     'learner kernels of libttl_hip.so); roofline = its FLOP / update_ms / '
                     'fp32 MFMA peak; graphed_update = the same replayed from a HIP graph; '
                     'phases_ms_per_step = HIP-event brackets; fp32')
        elif 'learner' in legs and world > 1:
            line['config3_training'] = 'N=1 only'
        if 'config5' in out:
            line['config5'] = dict(
                out['config5'],
                what='BASELINE configs[4] on synthetic data: SAC training with oracle_bonus 10 '
                     'and the oracle stopping criterion (TransformerOracle, random-init '
                     'weights, fp16 autocast as the reference), 131072 streamlines in total '
                     f'sharded over {world} GPU(s)' + (' -- at N = 1 the N = 8 shard (16384) '
                     'is the headline of this object and the whole batch is timed beside it'
                     if world == 1 else ', learner replicas data-parallel over RCCL') +
                     '; min_length 10 mm so that both oracle paths are live inside the '
                     'synthetic ball mask; policy initialised to keep going straight '
                     '(benchmarks/bench_training.py); value = streamline-steps of all ranks '
                     '/ slowest rank\'s step time, one whole episode (112 steps)')
        if 'shapes' in out:
            line['other_shapes'] = out['shapes']
        elif 'shapes' in legs and world > 1:
            line['other_shapes'] = 'N=1 only'
        if 'config4' in out:
            c4 = out['config4']
            line['config4'] = {
                'what': 'BASELINE configs[3] on synthetic data: ' + WORKLOADS['c4']['what']
                        + f', {c4_total} streamlines in total sharded over {world} GPU(s)',
                'scaling': 'strong', 'n_actor_total': c4_total,
                'n_actor_per_gpu': -(-c4_total // world),
                'step_only': {'value': c4['value'], 'ms_per_step': c4['ms_per_step'],
                              'streamline_steps': c4['streamline_steps'],
                              'value_min': c4['windows']['value_min'],
                              'value_max': c4['windows']['value_max'],
                              'windows': c4['windows']['n'], 'steps': args.steps},
                'end_to_end': c4['end_to_end'],
                'setup_s': out['c4_setup_s'],
            }
        if 'hbm' in out:
            roof4 = roofline_object(out['hbm'], 'c4', out['mask_c4'],
                                    ['pmc_traffic_c4shard.json',
                                     'r02_c4shard_pmc_traffic.json'], note_c4)
            if roof4 is not None:
                roof4['value_rank0_shard'] = out['hbm']['rank0_units'] / \
                    (out['hbm']['ms_per_step'] * 1e-3 * args.steps)
                roof4['ms_per_step'] = out['hbm']['ms_per_step']
            line['roofline_hbm_regime'] = roof4
        tuned, search = out.get('placement', (None, None))
        if tuned:
            # gather time (ms per launch) of every pair (allocation of the SH
            # volume, allocation of a state buffer) tried at the first large
            # reset; the fastest pair was kept (env.py:_tune_placement)
            line['placement_candidates_ms'] = tuned
        if search:
            line['placement_search'] = search
        if out.get('c4_placement', (None, None))[0]:
            line['placement_candidates_ms_config4'] = out['c4_placement'][0]
        if out.get('c4_placement', (None, None))[1]:
            line['placement_search_config4'] = out['c4_placement'][1]
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu if world == 1 else \
                'N=1 only (the oracle is timed on rank 0 of a 1-GPU run)'
        # short scalars of the other configurations, so that a parsed record
        # shows them without opening the nested objects
        def _get(d, *keys):
            for k in keys:
                if isinstance(d, list) and isinstance(k, int) and d:
                    d = d[k]
                    continue
                if not isinstance(d, dict) or k not in d:
                    return None
                d = d[k]
            return d
        scalars = {
            'config3_train_step_ms': _get(line, 'config3_training', 'train_step_ms'),
            'config3_update_ms': _get(line, 'config3_training', 'update_ms'),
            'config3_update_frac_fp32_mfma': _get(line, 'config3_training', 'roofline', 'frac'),
            'config3_value': _get(line, 'config3_training', 'train_streamline_steps_per_s'),
            'config5_train_step_ms': _get(line, 'config5', 'train_step_ms_max_over_ranks'),
            'config5_update_ms': _get(line, 'config5', 'update_ms'),
            'config5_value': _get(line, 'config5', 'value'),
            'oracle_net_16384_ms': _get(line, 'config5', 'oracle_net_alone', 'per_batch', -1,
                                        'fused_ms'),
            'oracle_net_frac_fp16_mfma': _get(line, 'config5', 'oracle_net_alone', 'roofline', 'frac'),
            'config4_step_only_value': _get(line, 'config4', 'step_only', 'value'),
            'config4_end_to_end_value': _get(line, 'config4', 'end_to_end', 'value_end_to_end'),
            'roofline_hbm_regime_frac': _get(line, 'roofline_hbm_regime', 'frac'),
            'strong_value': _get(line, 'strong', 'value'),
            'whole_episode_value_rank0': _get(line, 'whole_episode',
                                              'streamline_steps_per_s_rank0'),
            'c2_K100_value': _get(line, 'other_shapes', 'c2_K100', 'value'),
            'c3_env_value': _get(line, 'other_shapes', 'c3_env', 'value'),
            'c1_shape_ms_per_step': _get(line, 'other_shapes', 'c1_shape', 'ms_per_step'),
            'c2_host_contract_value': _get(line, 'other_shapes', 'c2_host_contract', 'value'),
        }
        line['summary'] = {k: v for k, v in scalars.items() if v is not None}
        for k, v in line['summary'].items():
            line[k] = v
        if head is None:
            line['error'] = 'no leg ran'
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
