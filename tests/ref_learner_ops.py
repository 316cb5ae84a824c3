"""Plain-torch restatement of the learner kernels (include/ttl_learner.h), with
the call signatures of ``tracktolearn_amd.algorithms.shared.fused.HipOps``.

TEST INFRASTRUCTURE ONLY: (i) on the CPU it lets ``FusedSACUpdate``'s schedule
(arenas, batching, manual backward) be checked against the autograd update of
the reference's formulas without a GPU; (ii) on the GPU every HIP kernel is
compared with its method here on the same inputs.  Nothing in the package
imports this file."""
import math

import torch
import torch.nn.functional as F

HEAD_PLAIN, HEAD_SAC, HEAD_TANH = 0, 1, 2
THIN_FWD_ROWS = 4
LOSS_BLOCK = 256
BETA1, BETA2, ADAM_EPS = 0.9, 0.999, 1e-8
HALF_LOG_2PI = math.log(math.sqrt(2 * math.pi))


def _as_blocks(t, n_out):
    """[n_out x M x n_in] view of block-diagonal activations given side by
    side [M x n_out * n_in] or already in planes."""
    if t.dim() == 3:
        return t
    return t.view(t.shape[0], n_out, -1).transpose(0, 1)


class TorchOps:

    def thin_forward(self, a, w, b, n_out, block_diagonal, head, out, ld_out, eps=None,
                     entropy_rows=0, logp=None, ls_raw=None, ent_part=None):
        w = w.reshape(n_out, -1)
        b = b.reshape(n_out)
        if block_diagonal:
            blk = _as_blocks(a, n_out)
            n_rows = blk.shape[1]
            y = torch.stack([blk[o] @ w[o] + b[o] for o in range(n_out)], dim=1)
        else:
            n_rows = a.shape[0]
            y = a @ w.t() + b
        if head == HEAD_PLAIN:
            out[:, :n_out] = y
        elif head == HEAD_TANH:
            out[:, :n_out] = torch.tanh(y)
        else:
            na = n_out // 2
            mu, raw = y[:, :na], y[:, na:]
            std = torch.exp(torch.clamp(raw, -20, 2))
            u = mu + eps * std
            lp = (-((u - mu) ** 2) / (2 * std ** 2) - std.log() - HALF_LOG_2PI).sum(-1)
            lp = lp - (2 * (math.log(2) - u - F.softplus(-2 * u))).sum(-1)
            out[:, :na] = torch.tanh(u)
            logp[:n_rows] = lp
            ls_raw[:n_rows] = raw
            if ent_part is not None:
                ent_part.zero_()
                masked = torch.where(torch.arange(n_rows, device=a.device) < entropy_rows, lp,
                                     torch.zeros_like(lp))
                pad = (-n_rows) % THIN_FWD_ROWS
                masked = torch.cat([masked, masked.new_zeros(pad)])
                sums = masked.view(-1, THIN_FWD_ROWS).sum(1)
                ent_part.view(-1)[:len(sums)] = sums

    def sac_losses(self, q_on, q_tg, logp, reward, not_done, log_alpha, alpha_const, gamma,
                   dq, loss_part, steps, consts, beta_pows, tick_mask, lr):
        n = reward.shape[0]
        alpha = torch.exp(log_alpha.detach()) if log_alpha is not None else alpha_const
        tq = torch.min(q_tg[:, 0], q_tg[:, 1])
        backup = reward + gamma * not_done * (tq - alpha * logp[n:])
        e = q_on[:n] - backup[:, None]
        dq[:n] = 2 * e / n
        p1, p2 = q_on[n:, 0], q_on[n:, 1]
        tie = p1 == p2
        dq[n:, 0] = torch.where(p1 < p2, -1.0 / n, 0.0) + torch.where(tie, -0.5 / n, 0.0)
        dq[n:, 1] = torch.where(p2 < p1, -1.0 / n, 0.0) + torch.where(tie, -0.5 / n, 0.0)
        if loss_part is not None:
            terms = torch.stack([alpha * logp[:n] - torch.min(p1, p2), e[:, 0] ** 2,
                                 e[:, 1] ** 2, q_on[:n, 0], q_on[:n, 1], backup,
                                 torch.zeros_like(backup), torch.zeros_like(backup)], dim=1)
            pad = (-n) % LOSS_BLOCK
            terms = torch.cat([terms, terms.new_zeros(pad, 8)])
            loss_part[:] = terms.view(-1, LOSS_BLOCK, 8).sum(1)
        for k in range(steps.numel()):
            if (tick_mask >> k) & 1:
                steps[k] += 1
                beta_pows[2 * k] *= BETA1
                beta_pows[2 * k + 1] *= BETA2
                consts[2 * k] = lr / (1 - float(beta_pows[2 * k]))
                consts[2 * k + 1] = math.sqrt(1 - float(beta_pows[2 * k + 1]))

    @staticmethod
    def _slab_sum(x, part, col0, n, r0, r1, rpb):
        """Column sums of the rows [r0, r1) of x per block of rpb rows."""
        rows = torch.arange(x.shape[0], device=x.device)
        x = torch.where(((rows >= r0) & (rows < r1))[:, None], x, torch.zeros_like(x))
        pad = (-x.shape[0]) % rpb
        x = torch.cat([x, x.new_zeros(pad, x.shape[1])])
        part[:, col0:col0 + n] = x.view(-1, rpb, x.shape[1]).sum(1)

    def thin_backward(self, d_out, a, w, n_out, block_diagonal, r0, r1, dz, part):
        from tracktolearn_amd.algorithms.shared.fused import _rows_per_block
        w = w.reshape(n_out, -1)
        n_in = w.shape[1]
        if block_diagonal:
            a_blk, dz_blk = _as_blocks(a, n_out), _as_blocks(dz, n_out)
            n_rows = a_blk.shape[1]
            a_flat = torch.cat([a_blk[o] for o in range(n_out)], dim=1)
            g = torch.cat([d_out[:, o:o + 1] * w[o][None, :] for o in range(n_out)], dim=1)
            dw = torch.cat([d_out[:, o:o + 1] * a_blk[o] for o in range(n_out)], dim=1)
        else:
            n_rows = a.shape[0]
            a_flat = a
            g = d_out @ w
            dw = torch.cat([d_out[:, o:o + 1] * a for o in range(n_out)], dim=1)
        rpb = _rows_per_block(n_rows)
        g = torch.where(a_flat > 0, g, torch.zeros_like(g))
        if block_diagonal:
            for o in range(n_out):
                dz_blk[o][:] = g[:, o * n_in:(o + 1) * n_in]
        else:
            dz[:] = g
        n_cols = a_flat.shape[1]
        self._slab_sum(g, part, 0, n_cols, r0, r1, rpb)
        self._slab_sum(dw, part, n_cols, n_out * n_in, r0, r1, rpb)
        self._slab_sum(d_out[:, :n_out], part, n_cols + n_out * n_in, n_out, r0, r1, rpb)

    def relu_backward_bias(self, dz, a, r0, r1, part):
        from tracktolearn_amd.algorithms.shared.fused import _rows_per_block
        g = torch.where(a > 0, dz, torch.zeros_like(dz))
        dz[:] = g
        rpb = _rows_per_block(dz.shape[-2])
        if dz.dim() == 3:
            for z in range(dz.shape[0]):
                self._slab_sum(g[z], part, z * dz.shape[2], dz.shape[2], r0, r1, rpb)
        else:
            self._slab_sum(g, part, 0, dz.shape[1], r0, r1, rpb)

    def colsum_finalize(self, segs):
        for part, off, n, out, scale in segs:
            out.view(-1)[:n] = part[:, off:off + n].sum(0) * scale

    def actor_head_backward(self, dh, h, wa, n_act, pi, ld_pi, eps, ls_raw, log_alpha,
                            alpha_const, d_head, head=HEAD_SAC):
        n = dh.shape[0]
        if head == HEAD_TANH:
            g = torch.where(h > 0, dh, torch.zeros_like(dh))
            t = pi[:, :n_act]
            d_head[:, :n_act] = (g @ wa.t()) * (1 - t * t)
            return
        alpha = torch.exp(log_alpha.detach()) if log_alpha is not None else alpha_const
        g = torch.where(h > 0, dh, torch.zeros_like(dh))
        dpi = g @ wa.t()
        t = pi[:, :n_act]
        an = alpha / n
        du = an * (2 * t) + dpi * (1 - t * t)
        raw = ls_raw[:n]
        inside = (raw >= -20) & (raw <= 2)
        std = torch.exp(torch.clamp(raw, -20, 2))
        d_head[:, :n_act] = du
        d_head[:, n_act:] = torch.where(inside, du * (eps[:n] * std) - an,
                                        torch.zeros_like(du))

    def td3_losses(self, q_on, q_tg, reward, not_done, gamma, dq, loss_part, steps, consts,
                   beta_pows, tick_mask, lr):
        n, n_q = q_on.shape
        tq = q_tg.min(dim=1).values
        target = reward + not_done * gamma * tq
        e = q_on - target[:, None]
        dq[:] = 2 * e / n
        if loss_part is not None:
            z = torch.zeros_like(target)
            e2 = e[:, 1] ** 2 if n_q == 2 else z
            q2 = q_on[:, 1] if n_q == 2 else z
            terms = torch.stack([z, e[:, 0] ** 2, e2, q_on[:, 0], q2, target, z, z], dim=1)
            pad = (-n) % LOSS_BLOCK
            terms = torch.cat([terms, terms.new_zeros(pad, 8)])
            loss_part[:] = terms.view(-1, LOSS_BLOCK, 8).sum(1)
        for k in range(steps.numel()):
            if (tick_mask >> k) & 1:
                steps[k] += 1
                beta_pows[2 * k] *= BETA1
                beta_pows[2 * k + 1] *= BETA2
                consts[2 * k] = lr / (1 - float(beta_pows[2 * k]))
                consts[2 * k + 1] = math.sqrt(1 - float(beta_pows[2 * k + 1]))

    def polyak(self, target, p, tau):
        target.mul_(1 - tau).add_(p * tau)

    @staticmethod
    def _adam(p, g, m, v, consts):
        m += (g - m) * (1 - BETA1)
        v.mul_(BETA2).add_((1 - BETA2) * g * g)
        denom = v.sqrt() / consts[1] + ADAM_EPS
        p += -consts[0] * (m / denom)

    def adam_polyak(self, p, g, m, v, target, consts, tau):
        self._adam(p, g, m, v, consts)
        if target is not None:
            target.mul_(1 - tau).add_(p * tau)

    def alpha_step(self, log_alpha, grad, m, v, mean_logp, target_entropy, consts):
        g = -(mean_logp + target_entropy)
        alpha_before = torch.exp(log_alpha)
        self._adam(log_alpha, g, m, v, consts)
        grad[:] = g + alpha_before * mean_logp

    def build_inputs(self, state, action, next_state, xs, n_state, n_act, w1, wa):
        n = state.shape[0]
        xs[:n, :n_state] = state
        xs[:n, n_state:n_state + n_act] = action
        xs[n:2 * n, :n_state] = state
        xs[2 * n:, :n_state] = next_state
        wa[:] = w1[:, n_state:n_state + n_act].t()
