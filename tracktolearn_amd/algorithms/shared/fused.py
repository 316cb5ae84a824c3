"""The MI355X form of one SAC / SACAuto gradient update.

The reference's update (TrackToLearn/algorithms/sac_auto.py:139-250,
sac.py:135-232) is PyTorch autograd over three MLPs + three
``torch.optim.Adam`` steps + a per-parameter Polyak loop.  At config 3's
shapes (W = 327, hidden 1024-1024, batch 4 096) that is ~40 GEMMs and ~110
small kernels, and 40 % of the GPU time is in the small ones
(``profiles/r03``: 600 bias-gradient reductions, 2 000 adds, ReLU-backward
masks, unfused Adam).  ``FusedSACUpdate`` computes the same update as a
hand-scheduled forward/backward:

* **memory**: every network's parameters live in ONE flat fp32 arena (online,
  target, gradient, Adam first and second moment: five arenas with the same
  layout), the ``nn.Parameter``s, their ``.grad`` and the optimizer's state
  tensors are views into them, so checkpoints, ``state_dict`` and
  ``optimizer.state_dict`` read as before, while Adam + Polyak is one kernel
  over the arena and the data-parallel gradient average one all-reduce per
  network without a flatten copy.  The two critics are interleaved layer by
  layer so that their first layers are ONE GEMM ([2h x (W+3)] stacked weights)
  and their activations sit side by side ([rows x 2h]).
* **batching**: the rows the three forwards consume come from one buffer
  ``xs`` [3B x ld]: rows [0,B) = (s, a), [B,2B) = (s, pi(s)), [2B,3B) =
  (s', pi(s')); the actor runs once on rows [B,3B) (2B rows), the online
  critics once on rows [0,2B), the target critics on rows [2B,3B): 16 GEMMs
  instead of ~40, all with 4 096-8 192 rows.
* **everything that is not a dense GEMM** is a hand-written HIP kernel of
  libttl_hip.so (include/ttl_learner.h): the 6-wide actor head with the
  squashed-gaussian sample / log-probability, the 1-wide critic heads, the
  per-row losses, the thin layers' backward fused with the ReLU mask and the
  bias / weight-gradient column sums, ReLU-backward + bias gradient, the
  actor-loss gradient through the critics' first layer, Adam + Polyak.
  Reductions are deterministic (slab partials, fixed order).

The GEMMs stay on PyTorch-ROCm (hipBLASLt fp32 MFMA, north_star) with the
bias + ReLU epilogue (``torch._addmm_activation``) and preallocated outputs.
fp32 throughout.  There is no CPU form of this path: on a CUDA device the
kernels are required (``_lib.load()`` raises without the library); on the
CPU (the known-answer tests) SAC keeps the plain autograd update.
"""
import ctypes as C
import os

import torch
from torch import nn

from tracktolearn_amd import _lib

try:                                    # no Stream object per call
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                  # pragma: no cover
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

HEAD_PLAIN, HEAD_SAC, HEAD_TANH = 0, 1, 2
THIN_FWD_ROWS = 4
LOSS_BLOCK = 256
BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-8
#: slab rows (row blocks) of the backward kernels: enough workgroups to fill
#: the 256 CUs, few enough for the fixed-order finalize to stay a few us
SLAB_ROWS = 128


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _rows_per_block(n_rows):
    return max(4, -(-n_rows // SLAB_ROWS))


class HipOps:
    """The learner kernels of libttl_hip.so, called with torch tensors.
    (The tests substitute a plain-torch restatement, tests/ref_learner_ops.py,
    to check the schedule below on the CPU against autograd.)"""

    def __init__(self, device):
        self.lib = _lib.load()
        self.index = torch.device(device).index or 0

    def _s(self):
        return C.c_void_p(_raw_stream(self.index))

    @staticmethod
    def _blocks(t, n_out, block_diagonal):
        """(n_rows, n_in, row stride, block stride) of a thin layer's input: one
        matrix [M x n_in]; or, block diagonal, n_out networks side by side
        [M x n_out * n_in] or in planes [n_out x M x n_in]."""
        if not block_diagonal:
            assert t.dim() == 2 and t.stride(1) == 1
            return t.shape[0], t.shape[1], t.stride(0), 0
        if t.dim() == 3:
            assert t.shape[0] == n_out and t.stride(2) == 1
            return t.shape[1], t.shape[2], t.stride(1), t.stride(0)
        assert t.stride(1) == 1 and t.shape[1] % n_out == 0
        return t.shape[0], t.shape[1] // n_out, t.stride(0), t.shape[1] // n_out

    def thin_forward(self, a, w, b, n_out, block_diagonal, head, out, ld_out, eps=None,
                     entropy_rows=0, logp=None, ls_raw=None, ent_part=None):
        n_rows, n_in, lda, a_bs = self._blocks(a, n_out, block_diagonal)
        assert w.is_contiguous() and w.numel() == n_out * n_in
        _lib.check(self.lib.ttl_thin_forward(
            _ptr(a), lda, a_bs, _ptr(w), _ptr(b), n_rows, n_in, n_out, int(block_diagonal),
            head, _ptr(eps) if eps is not None else None, entropy_rows, _ptr(out), ld_out,
            _ptr(logp) if logp is not None else None,
            _ptr(ls_raw) if ls_raw is not None else None,
            _ptr(ent_part) if ent_part is not None else None, self._s()), 'ttl_thin_forward')

    def sac_losses(self, q_on, q_tg, logp, reward, not_done, log_alpha, alpha_const, gamma,
                   dq, loss_part, steps, consts, beta_pows, tick_mask, lr):
        n = reward.shape[0]
        _lib.check(self.lib.ttl_sac_losses(
            _ptr(q_on), _ptr(q_tg), _ptr(logp), _ptr(reward), _ptr(not_done), n,
            _ptr(log_alpha) if log_alpha is not None else None, float(alpha_const),
            float(gamma), _ptr(dq), _ptr(loss_part) if loss_part is not None else None,
            _ptr(steps), _ptr(consts), _ptr(beta_pows), steps.numel(), tick_mask, float(lr),
            BETA1, BETA2, self._s()), 'ttl_sac_losses')

    def thin_backward(self, d_out, a, w, n_out, block_diagonal, r0, r1, dz, part):
        n_rows, n_in, lda, a_bs = self._blocks(a, n_out, block_diagonal)
        _, n_in_dz, ld_dz, dz_bs = self._blocks(dz, n_out, block_diagonal)
        assert n_in_dz == n_in and d_out.stride(1) == 1
        _lib.check(self.lib.ttl_thin_backward(
            _ptr(d_out), d_out.stride(0), _ptr(a), lda, a_bs, _ptr(w), n_rows, n_in, n_out,
            int(block_diagonal), r0, r1, _rows_per_block(n_rows), _ptr(dz), ld_dz, dz_bs,
            _ptr(part), part.stride(0), self._s()), 'ttl_thin_backward')

    def relu_backward_bias(self, dz, a, r0, r1, part):
        """dz, a: [M x n_cols], or planes [P x M x n_cols] (slab columns P * n_cols)."""
        assert a.shape == dz.shape and a.stride(-1) == 1 and dz.stride(-1) == 1
        planes = dz.shape[0] if dz.dim() == 3 else 1
        n_rows, n_cols = dz.shape[-2], dz.shape[-1]
        _lib.check(self.lib.ttl_relu_backward_bias(
            _ptr(dz), dz.stride(-2), dz.stride(0) if dz.dim() == 3 else 0, _ptr(a),
            a.stride(-2), a.stride(0) if a.dim() == 3 else 0, planes, n_rows, n_cols, r0, r1,
            _rows_per_block(n_rows), _ptr(part), part.stride(0), self._s()),
            'ttl_relu_backward_bias')

    def colsum_finalize(self, segs):
        """segs: list of (part [R x ld] tensor, column offset, n, out tensor, scale)."""
        arr = (_lib.ColsumSeg * len(segs))()
        for k, (part, off, n, out, scale) in enumerate(segs):
            assert out.numel() >= n and out.is_contiguous() and off + n <= part.shape[1]
            arr[k].part = part.data_ptr() + 4 * off
            arr[k].ld = part.stride(0)
            arr[k].n_part = part.shape[0]
            arr[k].n = n
            arr[k].out = out.data_ptr()
            arr[k].scale = scale
            arr[k].accumulate = 0
        _lib.check(self.lib.ttl_colsum_finalize(arr, len(segs), self._s()),
                   'ttl_colsum_finalize')

    def actor_head_backward(self, dh, h, wa, n_act, pi, ld_pi, eps, ls_raw, log_alpha,
                            alpha_const, d_head, head=HEAD_SAC):
        n_rows, n_cols = dh.shape
        assert h.shape == dh.shape and wa.is_contiguous() and wa.shape == (n_act, n_cols)
        _lib.check(self.lib.ttl_sac_actor_head_backward(
            _ptr(dh), dh.stride(0), _ptr(h), h.stride(0), _ptr(wa), n_rows, n_cols, n_act,
            head, _ptr(pi), ld_pi, _ptr(eps) if eps is not None else None,
            _ptr(ls_raw) if ls_raw is not None else None,
            _ptr(log_alpha) if log_alpha is not None else None, float(alpha_const),
            _ptr(d_head), self._s()), 'ttl_sac_actor_head_backward')

    def td3_losses(self, q_on, q_tg, reward, not_done, gamma, dq, loss_part, steps, consts,
                   beta_pows, tick_mask, lr):
        n, n_q = q_on.shape
        _lib.check(self.lib.ttl_td3_losses(
            _ptr(q_on), _ptr(q_tg), _ptr(reward), _ptr(not_done), n, n_q, float(gamma),
            _ptr(dq), _ptr(loss_part) if loss_part is not None else None, _ptr(steps),
            _ptr(consts), _ptr(beta_pows), steps.numel(), tick_mask, float(lr), BETA1, BETA2,
            self._s()), 'ttl_td3_losses')

    def polyak(self, target, p, tau):
        _lib.check(self.lib.ttl_polyak_average(_ptr(target), _ptr(p), p.numel(), float(tau),
                                               self._s()), 'ttl_polyak_average')

    def adam_polyak(self, p, g, m, v, target, consts, tau):
        _lib.check(self.lib.ttl_adam_polyak(
            _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(target) if target is not None else None,
            p.numel(), _ptr(consts), BETA1, BETA2, ADAM_EPS, float(tau), self._s()),
            'ttl_adam_polyak')

    def alpha_step(self, log_alpha, grad, m, v, mean_logp, target_entropy, consts):
        _lib.check(self.lib.ttl_sac_alpha_step(
            _ptr(log_alpha), _ptr(grad), _ptr(m), _ptr(v), _ptr(mean_logp),
            float(target_entropy), _ptr(consts), BETA1, BETA2, ADAM_EPS, self._s()),
            'ttl_sac_alpha_step')

    def build_inputs(self, state, action, next_state, xs, n_state, n_act, w1, wa):
        n = state.shape[0]
        assert state.stride(1) == 1 and action.stride(1) == 1 and next_state.stride(1) == 1
        _lib.check(self.lib.ttl_build_learner_inputs(
            _ptr(state), state.stride(0), _ptr(action), action.stride(0), _ptr(next_state),
            next_state.stride(0), n, n_state, n_act, _ptr(xs), xs.stride(0), _ptr(w1),
            w1.stride(0), w1.shape[0], _ptr(wa), self._s()), 'ttl_build_learner_inputs')


def _linears(seq):
    """The Linear layers of a make_fc_network stack (Linear/ReLU ... Linear)."""
    mods = list(seq)
    lin = [m for m in mods if isinstance(m, nn.Linear)]
    ok = all(isinstance(m, (nn.Linear, nn.ReLU)) for m in mods) and \
        len(mods) == 2 * len(lin) - 1 and all(m.bias is not None for m in lin) and \
        all(type(mods[2 * i + 1]) is nn.ReLU for i in range(len(lin) - 1))
    return lin if ok and len(lin) >= 2 else None


class _Arena:
    """Flat fp32 storage of one network (actor, or the two critics interleaved)
    in five copies -- online, target, gradient, Adam m, Adam v -- with views
    per tensor.  ``slots``: [(name, shape)] in arena order; every slot starts
    on a 16-byte boundary (the pads stay zero through Adam)."""

    def __init__(self, slots, device, dtype=torch.float32):
        self.offsets, off = {}, 0
        for name, shape in slots:
            n = 1
            for d in shape:
                n *= d
            self.offsets[name] = (off, tuple(shape))
            off += (n + 3) // 4 * 4
        self.n = off
        z = dict(dtype=dtype, device=device)
        self.online = torch.zeros(off, **z)
        self.target = torch.zeros(off, **z)
        self.grad = torch.zeros(off, **z)
        self.m = torch.zeros(off, **z)
        self.v = torch.zeros(off, **z)

    def view(self, flat, name):
        off, shape = self.offsets[name]
        n = 1
        for d in shape:
            n *= d
        return flat[off:off + n].view(shape)


class _FusedNets:
    """What the hand-scheduled updates share: the actor and the critic(s) of
    ``alg.agent`` / ``alg.target`` re-homed into arenas (parameters, gradients,
    Adam moments as views), the optimizers' state bound to them, the device-side
    Adam step counters."""

    #: (optimizer attribute, arena attribute) in the order of the step counters
    OPTIMIZERS = (('actor_optimizer', 'arena_a'), ('critic_optimizer', 'arena_q'))

    def __init__(self, alg, ops, head_out):
        self.alg = alg
        self.device = torch.device(alg.device)
        self.ops = ops if ops is not None else HipOps(self.device)
        actor, critic = alg.agent.actor, alg.agent.critic
        tcritic = alg.target.critic
        self.a_lin = _linears(actor.layers)
        self.ta_lin = _linears(alg.target.actor.layers)
        names = ['q1'] + (['q2'] if hasattr(critic, 'q2') else [])
        self.q_lin = [_linears(getattr(critic, n)) for n in names]
        self.tq_lin = [_linears(getattr(tcritic, n)) for n in names]
        self.NQ = len(names)
        if self.a_lin is None or None in self.q_lin:
            raise ValueError('the fused update needs Linear/ReLU stacks (make_fc_network)')
        self.S = self.a_lin[0].in_features
        self.A = actor.action_dim
        self.L = len(self.a_lin) - 1                         # hidden layers
        self.ha = [lin.out_features for lin in self.a_lin[:-1]]
        self.hq = [lin.out_features for lin in self.q_lin[0][:-1]]
        ok = all(len(q) == self.L + 1 for q in self.q_lin) and \
            self.a_lin[-1].out_features == head_out and self.A <= 4 and \
            self.q_lin[0][0].in_features == self.S + self.A and \
            all(a.weight.shape == b.weight.shape for q in self.q_lin[1:]
                for a, b in zip(self.q_lin[0], q)) and \
            self.q_lin[0][-1].out_features == 1
        if not ok:
            raise ValueError('the fused update: unsupported network shapes')
        #: float32 on the GPU; the CPU schedule test also runs it in float64
        self.dtype = self.a_lin[0].weight.dtype
        if isinstance(self.ops, HipOps) and self.dtype != torch.float32:
            raise ValueError('the fused update: the HIP kernels are float32')
        self._build_arenas()
        self._batch = None
        self.steps = torch.zeros(3, dtype=self.dtype, device=self.device)
        self.consts = torch.zeros(6, dtype=self.dtype, device=self.device)
        #: beta1^step, beta2^step per optimizer (float64, advanced on the device)
        self.beta_pows = torch.ones(6, dtype=torch.float64, device=self.device)
        self._bind_optimizers()
        self.loss_out = torch.zeros(8, dtype=self.dtype, device=self.device)

    # ------------------------------------------------------------------ #
    # arenas
    def _build_arenas(self):
        dev = self.device
        a_slots = []
        for l, lin in enumerate(self.a_lin):
            a_slots += [(f'w{l}', lin.weight.shape), (f'b{l}', lin.bias.shape)]
        self.arena_a = _Arena(a_slots, dev, self.dtype)
        q_slots = []
        for l, lin in enumerate(self.q_lin[0]):
            o, i = lin.weight.shape
            q_slots += [(f'w{l}', (self.NQ, o, i)), (f'b{l}', (self.NQ, o))]
        self.arena_q = _Arena(q_slots, dev, self.dtype)
        self._rehome()

    def _tables(self):
        """(arena, which, networks, stacked) of the four parameter sets."""
        return ((self.arena_a, 'online', [self.a_lin], False),
                (self.arena_a, 'target', [self.ta_lin], False),
                (self.arena_q, 'online', self.q_lin, True),
                (self.arena_q, 'target', self.tq_lin, True))

    def _home(self, arena, which, nets, stacked):
        """Move the parameters of ``nets`` (one list of Linears per network)
        into ``arena.<which>`` and make them views of it; online parameters
        get their ``.grad`` as a view of ``arena.grad``."""
        flat = getattr(arena, which)
        for k, lins in enumerate(nets):
            for l, lin in enumerate(lins):
                for name, p in ((f'w{l}', lin.weight), (f'b{l}', lin.bias)):
                    v = arena.view(flat, name)
                    v = v[k] if stacked else v
                    v.copy_(p.data)
                    p.data = v
                    if which == 'online':
                        g = arena.view(arena.grad, name)
                        p.grad = g[k] if stacked else g

    def _homed(self):
        """Whether the parameters still are views of the arenas (``.to()`` /
        ``.float()`` on a module re-allocates them)."""
        a, q = self.arena_a, self.arena_q
        k = self.NQ - 1
        return (self.a_lin[0].weight.data_ptr() == a.view(a.online, 'w0').data_ptr() and
                self.q_lin[k][-1].bias.data_ptr() == q.view(q.online, f'b{self.L}')[k].data_ptr()
                and self.ta_lin[0].weight.data_ptr() == a.view(a.target, 'w0').data_ptr() and
                self.tq_lin[k][-1].bias.data_ptr() ==
                q.view(q.target, f'b{self.L}')[k].data_ptr())

    def _rehome(self):
        for arena, which, nets, stacked in self._tables():
            self._home(arena, which, nets, stacked)

    def _attach_grads(self):
        """``optimizer.zero_grad()`` (set_to_none) detaches ``.grad``."""
        for arena, which, nets, stacked in self._tables():
            if which != 'online':
                continue
            for k, lins in enumerate(nets):
                for l, lin in enumerate(lins):
                    for name, p in ((f'w{l}', lin.weight), (f'b{l}', lin.bias)):
                        if p.grad is None or p.grad.data_ptr() == 0:
                            g = arena.view(arena.grad, name)
                            p.grad = g[k] if stacked else g

    # ------------------------------------------------------------------ #
    # optimizer state as views of the arenas
    def _bind_optimizers(self):
        alg = self.alg
        table = [(alg.actor_optimizer, self.arena_a, [self.a_lin], 1, False),
                 (alg.critic_optimizer, self.arena_q, self.q_lin, 2, True)]
        for opt, arena, nets, k_opt, stacked in table:
            for k, lins in enumerate(nets):
                for l, lin in enumerate(lins):
                    for name, p in ((f'w{l}', lin.weight), (f'b{l}', lin.bias)):
                        self._bind_state(opt, p, k_opt,
                                         *(arena.view(f, name)[k] if stacked
                                           else arena.view(f, name) for f in (arena.m, arena.v)))
        if getattr(self, 'auto', False):
            if not hasattr(self, 'alpha_m'):
                self.alpha_m = torch.zeros(1, dtype=self.dtype, device=self.device)
                self.alpha_v = torch.zeros(1, dtype=self.dtype, device=self.device)
            self._bind_state(alg.alpha_optimizer, alg.log_alpha, 0, self.alpha_m, self.alpha_v)
            if alg.log_alpha.grad is None:
                alg.log_alpha.grad = torch.zeros_like(alg.log_alpha)

    @staticmethod
    def _export_own_steps(optimizer, state_dict):
        """``optimizer.state_dict()``: every parameter gets a ``step`` tensor of
        its own (here they all are one view of the device counter; a torch
        optimizer that loads the dict increments them one by one)."""
        state_dict['state'] = {
            k: dict(v, step=v['step'].clone()) if torch.is_tensor(v.get('step')) else v
            for k, v in state_dict['state'].items()}
        return state_dict

    def _bind_state(self, opt, p, k_opt, m_view, v_view):
        if not getattr(opt, '_ttl_step_hook', False):
            opt.register_state_dict_post_hook(self._export_own_steps)
            opt._ttl_step_hook = True
        st = opt.state[p]
        if 'exp_avg' in st and st['exp_avg'].data_ptr() != m_view.data_ptr():
            # state made by torch's own step() or load_state_dict(): import it
            m_view.copy_(st['exp_avg'])
            v_view.copy_(st['exp_avg_sq'])
            step = float(st['step'])
            self.steps[k_opt] = step
            self.beta_pows[2 * k_opt] = BETA1 ** step
            self.beta_pows[2 * k_opt + 1] = BETA2 ** step
        st['step'] = self.steps[k_opt]
        st['exp_avg'] = m_view
        st['exp_avg_sq'] = v_view

    def _optimizers_bound(self):
        alg = self.alg
        a, q = self.arena_a, self.arena_q
        k = self.NQ - 1
        st_a = alg.actor_optimizer.state.get(self.a_lin[0].weight, {})
        st_q = alg.critic_optimizer.state.get(self.q_lin[k][-1].bias, {})
        ok = ('exp_avg' in st_a and st_a['exp_avg'].data_ptr() == a.view(a.m, 'w0').data_ptr()
              and 'exp_avg' in st_q and
              st_q['exp_avg'].data_ptr() == q.view(q.m, f'b{self.L}')[k].data_ptr())
        if ok and getattr(self, 'auto', False):
            st = alg.alpha_optimizer.state.get(alg.log_alpha, {})
            ok = 'exp_avg' in st and st['exp_avg'].data_ptr() == self.alpha_m.data_ptr()
        return ok

    def _prepare(self, B):
        """Before an update: workspaces of the batch size, parameters and
        optimizer state still at home, gradients attached."""
        if self._batch != B:
            self._alloc(B)
            self._batch = B
        if not self._homed():
            self._rehome()
        if not self._optimizers_bound():
            self._bind_optimizers()
        self._attach_grads()

    def _all_reduce(self, extra=()):
        import torch.distributed as dist
        group = self.alg._dp_group
        world = dist.get_world_size(group)
        for t in (self.arena_a.grad, self.arena_q.grad) + tuple(extra):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            t /= world

    #: data-parallel replicas: the critics' gradient arena is averaged WHILE the
    #: actor's backward runs (it is complete before the actor-loss rows leave the
    #: critics), the actor's while the critics' Adam step runs; ``TTL_DP_OVERLAP=0``:
    #: both after the backward, one after the other (``_all_reduce``)
    dp_overlap = os.environ.get('TTL_DP_OVERLAP', '1') != '0'

    def _all_reduce_begin(self, tensors):
        """Start averaging ``tensors`` over the replicas: the collective is
        ordered after everything queued on the current stream so far and runs
        beside what is queued next (RCCL's own stream).  Returns a handle for
        ``_all_reduce_end``."""
        import torch.distributed as dist
        group = self.alg._dp_group
        works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
                 for t in tensors]
        return works, tensors, dist.get_world_size(group)

    def _all_reduce_end(self, handle):
        """The current stream waits for the collectives of ``handle``; sums ->
        means."""
        works, tensors, world = handle
        for w in works:
            w.wait()
        for t in tensors:
            t /= world


class FusedSACUpdate(_FusedNets):
    """One SAC / SACAuto update on ``alg``'s networks (see the module
    docstring).  ``alg`` is a ``tracktolearn_amd.algorithms.sac.SAC`` (or
    ``SACAuto``); the parameters of ``alg.agent`` / ``alg.target`` and the
    state of its optimizers are re-homed into the arenas on construction."""

    def __init__(self, alg, ops=None):
        self.auto = hasattr(alg, 'log_alpha')
        super().__init__(alg, ops, 2 * alg.agent.actor.action_dim)
        if self.NQ != 2:
            raise ValueError('FusedSACUpdate needs the double critic')
        self.mean_logp = torch.zeros(1, dtype=self.dtype, device=self.device)

    # ------------------------------------------------------------------ #
    # workspaces of a batch size
    def _alloc(self, B):
        dev, S, A, L = self.device, self.S, self.A, self.L
        z = dict(dtype=self.dtype, device=dev)
        ld = (S + A + 3) // 4 * 4
        self.B, self.ld = B, ld
        self.xs = torch.zeros(3 * B, ld, **z)
        self.eps = torch.zeros(2 * B, A, **z)
        self.act_a = [torch.empty(2 * B, h, **z) for h in self.ha]
        self.logp = torch.empty(2 * B, **z)
        self.ls_raw = torch.empty(2 * B, A, **z)
        self.ent_part = torch.zeros(-(-2 * B // THIN_FWD_ROWS), 1, **z)
        # the critics' activations: layer 0 side by side [rows x 2h] (one stacked
        # GEMM), the layers above in planes [2 x rows x h] -- contiguous GEMM outputs
        # keep the bias + ReLU epilogue (a strided `out` costs torch's addmm a bias
        # broadcast copy and a separate ReLU pass, ~25 us per GEMM here)
        def crit(rows):
            return [torch.empty(rows, 2 * h, **z) if l == 0 else torch.empty(2, rows, h, **z)
                    for l, h in enumerate(self.hq)]
        self.hc, self.ht = crit(2 * B), crit(B)
        self.q_on = torch.empty(2 * B, 2, **z)
        self.q_tg = torch.empty(B, 2, **z)
        self.dq = torch.empty(2 * B, 2, **z)
        self.loss_part = torch.zeros(-(-B // LOSS_BLOCK), 8, **z)
        self.dzc = crit(2 * B)
        self.dza = [torch.empty(B, h, **z) for h in self.ha]
        self.wa = torch.empty(A, 2 * self.hq[0], **z)
        self.d_head = torch.empty(B, 2 * A, **z)
        R2, R1 = -(-2 * B // _rows_per_block(2 * B)), -(-B // _rows_per_block(B))
        hqL, haL = self.hq[-1], self.ha[-1]
        # slabs: [db below | dW thin | db thin] of the thin layers, [db] of the others
        self.part_q_top = torch.zeros(R2, 2 * hqL + 2 * hqL + 2, **z)
        self.part_a_top = torch.zeros(R1, haL + 2 * A * haL + 2 * A, **z)
        # lower critic layers: rows [0,B) of the slab are the critic-loss rows;
        # layer 0 runs the ReLU backward on those B rows only
        self.part_q = [torch.zeros(R1 if l == 0 else R2, 2 * h, **z)
                       for l, h in enumerate(self.hq[:-1])]
        self.part_a = [torch.zeros(R1, h, **z) for h in self.ha[:-1]]

    # ------------------------------------------------------------------ #
    def update(self, batch, eps_pi=None, eps_next=None, want_losses=False):
        """One update from ``batch`` = (state, action, next_state, reward,
        not_done).  ``eps_*``: the N(0, 1) draws of the two policy samples
        (drawn here, in the reference's order, when None)."""
        alg, ops = self.alg, self.ops
        state, action, next_state, reward, not_done = batch
        B = state.shape[0]
        self._prepare(B)
        S, A, L, ld = self.S, self.A, self.L, self.ld
        aa, aq = self.arena_a, self.arena_q
        W = lambda arena, flat, l: arena.view(flat, f'w{l}')      # noqa: E731
        Bv = lambda arena, flat, l: arena.view(flat, f'b{l}')     # noqa: E731
        fused = torch._addmm_activation

        # ---- inputs
        xs = self.xs
        ops.build_inputs(state, action, next_state, xs, S, A,
                         W(aq, aq.online, 0).view(2 * self.hq[0], S + A), self.wa)
        if eps_pi is None:
            torch.randn((B, A), out=self.eps[:B])
        else:
            self.eps[:B].copy_(eps_pi)
        if eps_next is None:
            torch.randn((B, A), out=self.eps[B:])
        else:
            self.eps[B:].copy_(eps_next)

        # ---- actor on rows [B, 3B): pi(s) and pi(s')
        x = xs[B:, :S]
        for l in range(L):
            fused(Bv(aa, aa.online, l), x, W(aa, aa.online, l).t(), use_gelu=False,
                  out=self.act_a[l])
            x = self.act_a[l]
        ops.thin_forward(x, W(aa, aa.online, L), Bv(aa, aa.online, L), 2 * A, False, HEAD_SAC,
                         xs[B:, S:], ld, eps=self.eps, entropy_rows=B, logp=self.logp,
                         ls_raw=self.ls_raw, ent_part=self.ent_part)

        # ---- critics: online on rows [0, 2B), target on rows [2B, 3B)
        for flat, rows, hbuf, qout in ((aq.online, xs[:2 * B], self.hc, self.q_on),
                                       (aq.target, xs[2 * B:], self.ht, self.q_tg)):
            x = rows[:, :S + A]
            h0 = self.hq[0]
            fused(Bv(aq, flat, 0).view(2 * h0), x, W(aq, flat, 0).view(2 * h0, S + A).t(),
                  use_gelu=False, out=hbuf[0])
            for l in range(1, L):
                hp = self.hq[l - 1]
                for k in range(2):
                    prev = hbuf[0][:, k * hp:(k + 1) * hp] if l == 1 else hbuf[l - 1][k]
                    fused(Bv(aq, flat, l)[k], prev, W(aq, flat, l)[k].t(), use_gelu=False,
                          out=hbuf[l][k])
            ops.thin_forward(hbuf[L - 1], W(aq, flat, L), Bv(aq, flat, L).view(2), 2, True,
                             HEAD_PLAIN, qout, 2)

        # ---- per-row losses, d loss / d q, Adam step counters
        log_alpha = alg.log_alpha if self.auto else None
        ops.sac_losses(self.q_on, self.q_tg, self.logp, reward, not_done, log_alpha,
                       0.0 if self.auto else alg.alpha, alg.gamma, self.dq,
                       self.loss_part if want_losses else None, self.steps, self.consts,
                       self.beta_pows, 0b111 if self.auto else 0b110, alg.lr)

        # ---- critics backward: rows [0,B) train the critics, rows [B,2B)
        #      carry the actor loss down to pi(s)
        ops.thin_backward(self.dq, self.hc[L - 1], W(aq, aq.online, L), 2, True, 0, B,
                          self.dzc[L - 1], self.part_q_top)
        for l in range(L - 1, 0, -1):
            hp = self.hq[l - 1]
            for k in range(2):
                dz = self.dzc[l][k]
                if l == 1:
                    cols = slice(k * hp, (k + 1) * hp)
                    a_prev, dz_prev = self.hc[0][:B, cols], self.dzc[0][:, cols]
                else:
                    a_prev, dz_prev = self.hc[l - 1][k][:B], self.dzc[l - 1][k]
                torch.mm(dz[:B].t(), a_prev, out=W(aq, aq.grad, l)[k])
                torch.mm(dz, W(aq, aq.online, l)[k], out=dz_prev)
            if l == 1:
                ops.relu_backward_bias(self.dzc[0][:B], self.hc[0][:B], 0, B, self.part_q[0])
            else:
                ops.relu_backward_bias(self.dzc[l - 1], self.hc[l - 1], 0, B,
                                       self.part_q[l - 1])
        h0 = self.hq[0]
        torch.mm(self.dzc[0][:B].t(), xs[:B, :S + A],
                 out=W(aq, aq.grad, 0).view(2 * h0, S + A))
        # the critics' gradients are complete: slabs -> arena (fixed order), and with
        # data-parallel replicas their average starts here, beside the actor's backward
        haL, hqL = self.ha[-1], self.hq[-1]
        segs = [(self.part_q_top, 0, 2 * hqL, Bv(aq, aq.grad, L - 1).view(-1), 1.0),
                (self.part_q_top, 2 * hqL, 2 * hqL, W(aq, aq.grad, L).view(-1), 1.0),
                (self.part_q_top, 4 * hqL, 2, Bv(aq, aq.grad, L).view(-1), 1.0)]
        segs += [(self.part_q[l], 0, 2 * self.hq[l], Bv(aq, aq.grad, l).view(-1), 1.0)
                 for l in range(L - 1)]
        ops.colsum_finalize(segs)
        dp = getattr(alg, '_dp', False)
        overlap = dp and self.dp_overlap
        pending_q = self._all_reduce_begin((aq.grad,)) if overlap else None
        ops.actor_head_backward(self.dzc[0][B:], self.hc[0][B:], self.wa, A, xs[B:2 * B, S:],
                                ld, self.eps, self.ls_raw, log_alpha,
                                0.0 if self.auto else alg.alpha, self.d_head)

        # ---- actor backward (rows [0,B) of its batch = pi(s))
        ops.thin_backward(self.d_head, self.act_a[L - 1][:B], W(aa, aa.online, L), 2 * A, False,
                          0, B, self.dza[L - 1], self.part_a_top)
        for l in range(L - 1, 0, -1):
            torch.mm(self.dza[l].t(), self.act_a[l - 1][:B], out=W(aa, aa.grad, l))
            torch.mm(self.dza[l], W(aa, aa.online, l), out=self.dza[l - 1])
            ops.relu_backward_bias(self.dza[l - 1], self.act_a[l - 1][:B], 0, B,
                                   self.part_a[l - 1])
        torch.mm(self.dza[0].t(), xs[B:2 * B, :S], out=W(aa, aa.grad, 0))

        # ---- the actor's slabs -> gradients (fixed order)
        segs = [(self.part_a_top, 0, haL, Bv(aa, aa.grad, L - 1), 1.0),
                (self.part_a_top, haL, 2 * A * haL, W(aa, aa.grad, L).view(-1), 1.0),
                (self.part_a_top, haL + 2 * A * haL, 2 * A, Bv(aa, aa.grad, L), 1.0)]
        segs += [(self.part_a[l], 0, self.ha[l], Bv(aa, aa.grad, l), 1.0) for l in range(L - 1)]
        segs.append((self.ent_part[:B // THIN_FWD_ROWS + (B % THIN_FWD_ROWS > 0)], 0, 1,
                     self.mean_logp, 1.0 / B))
        if want_losses:
            segs.append((self.loss_part, 0, 8, self.loss_out, 1.0 / B))
        ops.colsum_finalize(segs)

        # ---- data-parallel replicas: one all-reduce per arena
        extra = (self.mean_logp,) if self.auto else ()
        if overlap:
            # the actor's average runs beside the critics' Adam step
            pending_a = self._all_reduce_begin((aa.grad,) + extra)
            self._all_reduce_end(pending_q)
            ops.adam_polyak(aq.online, aq.grad, aq.m, aq.v, aq.target, self.consts[4:6], alg.tau)
            self._all_reduce_end(pending_a)
        elif dp:
            self._all_reduce(extra)

        # ---- temperature, actor, critics: Adam (+ Polyak)
        if self.auto:
            ops.alpha_step(alg.log_alpha.data, alg.log_alpha.grad, self.alpha_m, self.alpha_v,
                           self.mean_logp, alg.target_entropy, self.consts[0:2])
        ops.adam_polyak(aa.online, aa.grad, aa.m, aa.v, aa.target, self.consts[2:4], alg.tau)
        if not overlap:
            ops.adam_polyak(aq.online, aq.grad, aq.m, aq.v, aq.target, self.consts[4:6], alg.tau)
        if want_losses:
            lo = self.loss_out
            return {'actor_loss': lo[0], 'critic_loss': lo[1] + lo[2], 'loss_q1': lo[1],
                    'loss_q2': lo[2], 'Q1': lo[3], 'Q2': lo[4], 'backup': lo[5]}
        return {}

    def flops_per_update(self, B):
        """FLOP of one update at batch B: ``issued`` = what this schedule runs
        (GEMMs 2 M N K each + the thin layers), ``autograd`` = what the
        reference's formulation runs (sac_auto.py:139-250: three separate
        forwards per network and, in ``actor_loss.backward()``, the critics'
        weight gradients that ``critic_optimizer.zero_grad()`` then discards)."""
        S, A = self.S, self.A
        da = [S] + self.ha
        dq = [S + A] + self.hq
        fa = sum(a * b for a, b in zip(da[:-1], da[1:]))            # actor hidden MACs / row
        fq = sum(a * b for a, b in zip(dq[:-1], dq[1:]))            # one critic
        ta, tq = self.ha[-1] * 2 * A, self.hq[-1]                   # thin layers
        mac_dgrad_a = sum(a * b for a, b in zip(da[1:-1], da[2:]))  # no dgrad into the input
        mac_dgrad_q = sum(a * b for a, b in zip(dq[1:-1], dq[2:]))
        issued = (2 * B * (fa + ta)                 # actor forward on s and s'
                  + 2 * B * 2 * (fq + tq)           # online critics on (s,a), (s,pi)
                  + B * 2 * (fq + tq)               # target critics
                  + 2 * B * 2 * (mac_dgrad_q + tq)  # critics: data gradients, 2B rows
                  + B * 2 * (fq + tq)               # critics: weight gradients, B rows
                  + B * 2 * A * self.hq[0]          # action columns of the critics' layer 0
                  + B * (mac_dgrad_a + ta)          # actor: data gradients
                  + B * (fa + ta))                  # actor: weight gradients
        autograd = (2 * B * (fa + ta) + 3 * B * 2 * (fq + tq)
                    + B * 2 * (mac_dgrad_q + tq + (S + A) * self.hq[0])     # actor loss -> pi
                    + B * 2 * (fq + tq)                                      # ... discarded dW
                    + B * 2 * (mac_dgrad_q + tq) + B * 2 * (fq + tq)         # critic loss
                    + B * (mac_dgrad_a + ta) + B * (fa + ta))
        return {'issued': 2.0 * issued, 'autograd': 2.0 * autograd}


class FusedTD3Update(_FusedNets):
    """One TD3 update (td3.py:130-230) -- or, with the single critic and
    ``agent_freq`` 1, one DDPG update (ddpg.py:234-319) -- as the same kind of
    hand-scheduled forward/backward: critic regression on the smoothed target
    action first, then (every ``agent_freq``-th update) policy ascent through
    the UPDATED first critic and the Polyak averages.  Rows of ``xs``: [0,B) =
    (s, a), [B,2B) = (s, pi(s)) (actor pass), [2B,3B) = (s', target action)."""

    def __init__(self, alg, ops=None):
        super().__init__(alg, ops, alg.agent.actor.action_dim)
        self.actor_loss = torch.zeros(1, dtype=self.dtype, device=self.device)

    def _alloc(self, B):
        dev, S, A = self.device, self.S, self.A
        z = dict(dtype=self.dtype, device=dev)
        ld = (S + A + 3) // 4 * 4
        self.B, self.ld = B, ld
        nq = self.NQ
        self.xs = torch.zeros(3 * B, ld, **z)
        self.act_a = [torch.empty(B, h, **z) for h in self.ha]

        def crit(rows, nets):
            return [torch.empty(rows, nets * h, **z) if l == 0 or nets == 1
                    else torch.empty(nets, rows, h, **z) for l, h in enumerate(self.hq)]
        self.hc, self.ht, self.dzc = crit(B, nq), crit(B, nq), crit(B, nq)
        self.h1 = [torch.empty(B, h, **z) for h in self.hq]      # first critic on (s, pi(s))
        self.dz1 = [torch.empty(B, h, **z) for h in self.hq]
        self.q_on, self.q_tg = torch.empty(B, nq, **z), torch.empty(B, nq, **z)
        self.q_pi = torch.empty(B, 1, **z)
        self.dq = torch.empty(B, nq, **z)
        self.dq_pi = torch.full((B, 1), -1.0 / B, **z)
        self.loss_part = torch.zeros(-(-B // LOSS_BLOCK), 8, **z)
        self.dza = [torch.empty(B, h, **z) for h in self.ha]
        self.wa = torch.empty(A, self.hq[0], **z)
        self.d_head = torch.empty(B, A, **z)
        R = -(-B // _rows_per_block(B))
        hqL, haL = self.hq[-1], self.ha[-1]
        self.part_q_top = torch.zeros(R, 2 * nq * hqL + nq, **z)
        self.part_q = [torch.zeros(R, nq * h, **z) for h in self.hq[:-1]]
        self.part_1 = torch.zeros(R, max(2 * hqL + 1, max(self.hq)), **z)   # actor-pass scratch
        self.part_a_top = torch.zeros(R, haL + A * haL + A, **z)
        self.part_a = [torch.zeros(R, h, **z) for h in self.ha[:-1]]

    def _critics_forward(self, flat, rows, hbuf, qout):
        aq, S, A, L, nq = self.arena_q, self.S, self.A, self.L, self.NQ
        fused = torch._addmm_activation
        h0 = self.hq[0]
        fused(aq.view(flat, 'b0').view(nq * h0), rows[:, :S + A],
              aq.view(flat, 'w0').view(nq * h0, S + A).t(), use_gelu=False, out=hbuf[0])
        for l in range(1, L):
            hp = self.hq[l - 1]
            for k in range(nq):
                if nq == 1:
                    prev, out = hbuf[l - 1], hbuf[l]
                else:
                    prev = hbuf[0][:, k * hp:(k + 1) * hp] if l == 1 else hbuf[l - 1][k]
                    out = hbuf[l][k]
                fused(aq.view(flat, f'b{l}')[k], prev, aq.view(flat, f'w{l}')[k].t(),
                      use_gelu=False, out=out)
        self.ops.thin_forward(hbuf[L - 1], aq.view(flat, f'w{L}'), aq.view(flat, f'b{L}').view(nq),
                              nq, nq > 1, HEAD_PLAIN, qout, nq)

    def update(self, batch, noise, update_actor, want_losses=True):
        """``noise``: the target-policy smoothing noise (already scaled and, for
        TD3, clipped); ``update_actor``: whether this update also steps the
        actor and the targets."""
        alg, ops = self.alg, self.ops
        state, action, next_state, reward, not_done = batch
        B = state.shape[0]
        self._prepare(B)
        S, A, L, ld, nq = self.S, self.A, self.L, self.ld, self.NQ
        aa, aq = self.arena_a, self.arena_q
        W = lambda arena, flat, l: arena.view(flat, f'w{l}')      # noqa: E731
        Bv = lambda arena, flat, l: arena.view(flat, f'b{l}')     # noqa: E731
        fused = torch._addmm_activation
        xs = self.xs
        h0 = self.hq[0]
        clip = getattr(alg, 'noise_clip', None) is not None          # TD3 clamps, DDPG does not

        # ---- inputs; target action = target_actor(s') + noise
        ops.build_inputs(state, action, next_state, xs, S, A, W(aq, aq.online, 0)[0], self.wa)
        x = xs[2 * B:, :S]
        for l in range(L):
            fused(Bv(aa, aa.target, l), x, W(aa, aa.target, l).t(), use_gelu=False,
                  out=self.act_a[l])
            x = self.act_a[l]
        ops.thin_forward(x, W(aa, aa.target, L), Bv(aa, aa.target, L), A, False, HEAD_TANH,
                         xs[2 * B:, S:], ld)
        nxt = xs[2 * B:, S:S + A]
        nxt.add_(noise)
        if clip:
            nxt.clamp_(-alg.max_action, alg.max_action)

        # ---- critics: target on (s', a'), online on (s, a); loss and d loss / d q
        self._critics_forward(aq.target, xs[2 * B:], self.ht, self.q_tg)
        self._critics_forward(aq.online, xs[:B], self.hc, self.q_on)
        ops.td3_losses(self.q_on, self.q_tg, reward, not_done, alg.gamma, self.dq,
                       self.loss_part if want_losses else None, self.steps, self.consts,
                       self.beta_pows, 0b110 if update_actor else 0b100, alg.lr)

        # ---- critics backward
        ops.thin_backward(self.dq, self.hc[L - 1], W(aq, aq.online, L), nq, nq > 1, 0, B,
                          self.dzc[L - 1], self.part_q_top)
        for l in range(L - 1, 0, -1):
            hp = self.hq[l - 1]
            for k in range(nq):
                if nq == 1:
                    dz, a_prev, dz_prev = self.dzc[l], self.hc[l - 1], self.dzc[l - 1]
                elif l == 1:
                    cols = slice(k * hp, (k + 1) * hp)
                    dz, a_prev, dz_prev = self.dzc[l][k], self.hc[0][:, cols], self.dzc[0][:, cols]
                else:
                    dz, a_prev, dz_prev = self.dzc[l][k], self.hc[l - 1][k], self.dzc[l - 1][k]
                torch.mm(dz.t(), a_prev, out=W(aq, aq.grad, l)[k])
                torch.mm(dz, W(aq, aq.online, l)[k], out=dz_prev)
            ops.relu_backward_bias(self.dzc[l - 1], self.hc[l - 1], 0, B, self.part_q[l - 1])
        torch.mm(self.dzc[0].t(), xs[:B, :S + A], out=W(aq, aq.grad, 0).view(nq * h0, S + A))
        hqL = self.hq[-1]
        segs = [(self.part_q_top, 0, nq * hqL, Bv(aq, aq.grad, L - 1).view(-1), 1.0),
                (self.part_q_top, nq * hqL, nq * hqL, W(aq, aq.grad, L).view(-1), 1.0),
                (self.part_q_top, 2 * nq * hqL, nq, Bv(aq, aq.grad, L).view(-1), 1.0)]
        segs += [(self.part_q[l], 0, nq * self.hq[l], Bv(aq, aq.grad, l).view(-1), 1.0)
                 for l in range(L - 1)]
        if want_losses:
            segs.append((self.loss_part, 0, 8, self.loss_out, 1.0 / B))
        ops.colsum_finalize(segs)
        dp = getattr(alg, '_dp', False)
        if dp and not update_actor:
            self._all_reduce_one(aq.grad)
        if not update_actor:
            ops.adam_polyak(aq.online, aq.grad, aq.m, aq.v, None, self.consts[4:6], alg.tau)
            return self._losses(want_losses, False)
        if dp:
            self._all_reduce_one(aq.grad)
        ops.adam_polyak(aq.online, aq.grad, aq.m, aq.v, None, self.consts[4:6], alg.tau)

        # ---- actor pass: pi(s) on rows [B,2B), the UPDATED first critic on (s, pi(s))
        self.wa.copy_(W(aq, aq.online, 0)[0][:, S:S + A].t())     # its action columns, now
        x = xs[B:2 * B, :S]
        for l in range(L):
            fused(Bv(aa, aa.online, l), x, W(aa, aa.online, l).t(), use_gelu=False,
                  out=self.act_a[l])
            x = self.act_a[l]
        ops.thin_forward(x, W(aa, aa.online, L), Bv(aa, aa.online, L), A, False, HEAD_TANH,
                         xs[B:2 * B, S:], ld)
        x = xs[B:2 * B, :S + A]
        for l in range(L):
            fused(Bv(aq, aq.online, l)[0], x, W(aq, aq.online, l)[0].t(), use_gelu=False,
                  out=self.h1[l])
            x = self.h1[l]
        ops.thin_forward(x, W(aq, aq.online, L)[0], Bv(aq, aq.online, L)[0], 1, False,
                         HEAD_PLAIN, self.q_pi, 1)
        # d(-mean q1) / d q1 = -1 / B down to the action columns of the first layer
        ops.thin_backward(self.dq_pi, self.h1[L - 1], W(aq, aq.online, L)[0], 1, False, 0, 0,
                          self.dz1[L - 1], self.part_1)
        for l in range(L - 1, 0, -1):
            torch.mm(self.dz1[l], W(aq, aq.online, l)[0], out=self.dz1[l - 1])
            if l > 1:
                ops.relu_backward_bias(self.dz1[l - 1], self.h1[l - 1], 0, 0,
                                       self.part_1[:, :self.hq[l - 1]])
        ops.actor_head_backward(self.dz1[0], self.h1[0], self.wa, A, xs[B:2 * B, S:], ld, None,
                                None, None, 0.0, self.d_head, head=HEAD_TANH)

        # ---- actor backward
        haL = self.ha[-1]
        ops.thin_backward(self.d_head, self.act_a[L - 1], W(aa, aa.online, L), A, False, 0, B,
                          self.dza[L - 1], self.part_a_top)
        for l in range(L - 1, 0, -1):
            torch.mm(self.dza[l].t(), self.act_a[l - 1], out=W(aa, aa.grad, l))
            torch.mm(self.dza[l], W(aa, aa.online, l), out=self.dza[l - 1])
            ops.relu_backward_bias(self.dza[l - 1], self.act_a[l - 1], 0, B, self.part_a[l - 1])
        torch.mm(self.dza[0].t(), xs[B:2 * B, :S], out=W(aa, aa.grad, 0))
        segs = [(self.part_a_top, 0, haL, Bv(aa, aa.grad, L - 1), 1.0),
                (self.part_a_top, haL, A * haL, W(aa, aa.grad, L).view(-1), 1.0),
                (self.part_a_top, haL + A * haL, A, Bv(aa, aa.grad, L), 1.0)]
        segs += [(self.part_a[l], 0, self.ha[l], Bv(aa, aa.grad, l), 1.0) for l in range(L - 1)]
        segs.append((self.q_pi, 0, 1, self.actor_loss, -1.0 / B))
        ops.colsum_finalize(segs)
        if dp:
            self._all_reduce_one(aa.grad)
        ops.adam_polyak(aa.online, aa.grad, aa.m, aa.v, aa.target, self.consts[2:4], alg.tau)
        ops.polyak(aq.target, aq.online, alg.tau)
        return self._losses(want_losses, True)

    def _all_reduce_one(self, t):
        import torch.distributed as dist
        group = self.alg._dp_group
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t /= dist.get_world_size(group)

    def _losses(self, want, actor):
        if not want:
            return {}
        lo = self.loss_out
        al = self.actor_loss[0] if actor else 0.0
        if self.NQ == 2:
            return {'actor_loss': al, 'critic_loss': lo[1] + lo[2], 'loss_q1': lo[1],
                    'loss_q2': lo[2], 'Q1': lo[3], 'Q2': lo[4], "Q'": lo[5]}
        return {'actor_loss': al, 'critic_loss': lo[1], 'Q': lo[3], "Q'": lo[5]}
