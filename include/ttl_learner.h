/*
 * ttl_learner.h -- C ABI of the learner kernels of libttl_hip.so: everything
 * of one SAC / SACAuto gradient update that is NOT a dense GEMM.
 *
 * The reference's learner (TrackToLearn/algorithms/sac_auto.py:139-250,
 * sac.py:135-232, shared/offpolicy.py:62-238) is PyTorch autograd +
 * torch.optim.Adam: at batch 4096 with 1024-1024 networks one update is ~150
 * launches of which 40 % of the GPU time are small element-wise / reduction
 * kernels (bias-gradient reductions, ReLU backward, the squashed-gaussian head
 * and its log-probability, three unfused Adam steps, Polyak averaging).  The
 * MI355X learner (tracktolearn_amd/algorithms/shared/fused.py) keeps the dense
 * layers on PyTorch-ROCm's fp32 MFMA GEMMs (north_star) and issues the rest as
 * the hand-written kernels below: a manual forward/backward with the same
 * arithmetic, 16 GEMMs + 15 of these launches per update at
 * two hidden layers.
 *
 * Conventions: as include/ttl_hip.h -- plain C, device pointers are borrowed,
 * every call is asynchronous on `hip_stream` (capturable in a HIP graph: no
 * call synchronises or allocates), 0 on success / negative TTL_ERR_* with
 * ttl_last_error().  All matrices are row-major f32; `ld*` are row strides in
 * floats.  Reductions are deterministic (per-block partial sums in a slab,
 * summed in a fixed order by ttl_colsum_finalize: no float atomics), so an
 * update replayed from a HIP graph equals the eager one bit for bit.
 */
#ifndef TTL_LEARNER_H
#define TTL_LEARNER_H

#include <stddef.h>
#include <stdint.h>

#include "ttl_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TTL_THIN_MAX_OUT 8      /* widest "thin" layer (SAC head: 2 * 3)   */
#define TTL_HEAD_PLAIN 0        /* out = a W^T + b                          */
#define TTL_HEAD_SAC 1          /* squashed gaussian (offpolicy.py:94-140)  */
#define TTL_HEAD_TANH 2         /* tanh(a W^T + b) (offpolicy.py:54-60)     */

/* The last layer of an MLP when it is only a few units wide (actor head: 6,
 * critic heads: 1 each): out[m][o] = sum_j A_o[m][j] w[o][j] + b[o], j < n_in.
 * block_diagonal = 0: one network, A_o = a ([M][n_in], row stride lda);
 * block_diagonal = 1: n_out networks (the double critic: q1, q2), A_o[m][j] =
 * a[o * a_block_stride + m * lda + j] -- side by side in one row
 * (a_block_stride = n_in, lda >= n_out * n_in) or in planes of their own
 * (a_block_stride = plane size, lda >= n_in).
 *
 * head = TTL_HEAD_PLAIN: out[m * ld_out + o].
 * head = TTL_HEAD_TANH : out[m * ld_out + o] = tanh(.)  (Actor.forward).
 * head = TTL_HEAD_SAC (n_out = 2 * n_act, n_act <= 4; MaxEntropyActor.forward,
 * shared/offpolicy.py:94-140 = Normal(mu, std).rsample(), its log_prob and the
 * tanh correction): mu = first n_act outputs, log_std = clamp(rest, -20, 2),
 * u = mu + eps * exp(log_std), out[m * ld_out + i] = tanh(u_i) (ld_out may be
 * the row stride of the critics' input rows: the action lands in its columns),
 * logp[m] = sum_i(-(u-mu)^2 / (2 var) - log std - log sqrt(2 pi)) - sum_i 2
 * (log 2 - u - softplus(-2u)), log_std_raw[m][n_act] = the unclamped log_std
 * (kept for the backward).  entropy_part[block] = sum over the block's rows
 * m < entropy_rows of logp[m] (for the temperature loss, sac_auto.py:172-174),
 * block = TTL_THIN_FWD_ROWS rows; NULL to skip. */
TTL_API int ttl_thin_forward(const float *a, int64_t lda, int64_t a_block_stride,
                             const float *w, const float *b, int32_t n_rows, int32_t n_in,
                             int32_t n_out, int32_t block_diagonal, int32_t head, const float *eps,
                             int32_t entropy_rows, float *out, int64_t ld_out, float *logp,
                             float *log_std_raw, float *entropy_part, void *hip_stream);

/* Rows of one block of ttl_thin_forward (entropy_part has ceil(n_rows / this)
 * entries). */
#define TTL_THIN_FWD_ROWS 4

/* Per-row terms of the SAC losses and their gradients w.r.t. the critic
 * outputs (sac_auto.py:177-205 / sac.py:168-200).  q_online: [2 * n][2], rows
 * [0, n) = Q1, Q2(s, a), rows [n, 2n) = Q1, Q2(s, pi); q_target: [n][2] =
 * target critics at (s', a'); logp: [2 * n] (log pi(a|s), then log pi(a'|s'));
 * reward, not_done: [n].  alpha = exp(*log_alpha) when log_alpha != NULL, else
 * alpha_const.
 *   backup = r + gamma not_done (min(tq1, tq2) - alpha logp')
 *   dq[i][k]     = 2 (q_k - backup) / n          (critic loss, both means)
 *   dq[n + i][k] = -(1/n) [q_k is the smaller one; 1/2 each on a tie]
 * loss_part[block][8] = the block's sums of {alpha logp - min q, (q1-backup)^2,
 * (q2-backup)^2, q1, q2, backup, 0, 0} (block = 256 rows), NULL to skip.
 *
 * One thread also advances the Adam step counters: for every optimizer k <
 * n_opt with bit k of `tick_mask` set, steps[k] += 1, beta_pows[2k] *= beta1,
 * beta_pows[2k+1] *= beta2 (= beta^step, float64, device memory: 1.0 before the
 * first step) and adam_consts[2k], [2k+1] = lr / (1 - beta1^step), sqrt(1 -
 * beta2^step) (float64 arithmetic, stored as f32: torch.optim.Adam's scalars). */
TTL_API int ttl_sac_losses(const float *q_online, const float *q_target, const float *logp,
                           const float *reward, const float *not_done, int32_t n,
                           const float *log_alpha, float alpha_const, float gamma,
                           float *dq, float *loss_part, float *steps, float *adam_consts,
                           double *beta_pows, int32_t n_opt, uint32_t tick_mask, double lr,
                           double beta1, double beta2, void *hip_stream);

/* Backward of a thin last layer and of the ReLU in front of it, one pass over
 * the activations: for every row m and input column j (A_o, n_in as above; dz is
 * addressed like a, with its own ld_dz / dz_block_stride)
 *   dZ_o[m][j] = (A_o[m][j] > 0) * sum_o' d_out[m][o'] w[o'][j]   (dense: all o';
 *                                                     block diagonal: o' = o only)
 * and, summed over the rows r0 <= m < r1 only (the rows whose loss trains THIS
 * layer; the others only pass the gradient through), per block of `rows_per_block`
 * rows into part[block][...]:
 *   [0, n_cols)                 column sums of dz        (bias grad of the layer below)
 *   [n_cols, n_cols + n_w)      d_out^T a                (weight grad, [n_out][n_in];
 *                               block-diagonal: [n_out][n_in] too, = the two critics' rows)
 *   [n_cols + n_w, + n_out)     column sums of d_out     (bias grad of this layer)
 * n_cols = n_in (dense) or n_out * n_in (block diagonal), n_w = n_out * n_in;
 * ld_part >= n_cols + n_w + n_out.  Blocks without a row in [r0, r1) write
 * zeros.  a and dz may not alias. */
TTL_API int ttl_thin_backward(const float *d_out, int64_t ld_dout, const float *a, int64_t lda,
                              int64_t a_block_stride, const float *w, int32_t n_rows,
                              int32_t n_in, int32_t n_out, int32_t block_diagonal, int32_t r0,
                              int32_t r1, int32_t rows_per_block, float *dz, int64_t ld_dz,
                              int64_t dz_block_stride, float *part, int64_t ld_part,
                              void *hip_stream);

/* ReLU backward in place + bias gradient, for n_planes matrices of [n_rows][n_cols]
 * (plane z at dz + z * dz_plane_stride / a + z * a_plane_stride: the two critics'
 * activations when they are kept in planes; 1 otherwise): dz[m][j] *= (a[m][j] > 0);
 * part[block][z * n_cols + j] = the block's column sums of the result over its
 * rows r0 <= m < r1 (block = rows_per_block rows; zeros for a block without such
 * a row). */
TTL_API int ttl_relu_backward_bias(float *dz, int64_t ld_dz, int64_t dz_plane_stride,
                                   const float *a, int64_t lda, int64_t a_plane_stride,
                                   int32_t n_planes, int32_t n_rows, int32_t n_cols,
                                   int32_t r0, int32_t r1, int32_t rows_per_block,
                                   float *part, int64_t ld_part, void *hip_stream);

/* One segment of ttl_colsum_finalize: out[j] = scale * sum_{r < n_part}
 * part[r * ld + j] (+ out[j] if accumulate), j < n, summed in the fixed order
 * r = 0, 4, 8, ... | 1, 5, ... | ... then across the four (n > 8), or r = 0, 32,
 * ... | 1, 33, ... | ... then across the 32 (n <= 8). */
typedef struct ttl_colsum_seg {
    const float *part;
    int64_t ld;
    int32_t n_part;
    int32_t n;
    float *out;
    float scale;
    int32_t accumulate;
} ttl_colsum_seg;
#define TTL_COLSUM_MAX_SEGS 12

TTL_API int ttl_colsum_finalize(const ttl_colsum_seg *segs, int32_t n_segs, void *hip_stream);

/* Gradient of the actor loss w.r.t. the actor's head outputs, through the
 * first layer of the critics (sac_auto.py:177-181 backward): with
 * g[m][j] = (h[m][j] > 0) * dh[m][j] (ReLU backward of the critics' first
 * layer, n_cols = both critics side by side),
 *   dpi[m][i] = sum_j g[m][j] wa[i][j]                         i < n_act
 * (wa: [n_act][n_cols] = the action columns of the stacked first-layer weights,
 * transposed), then the squashed-gaussian head backward with pi = tanh(u) read
 * from pi[m * ld_pi + i]:
 *   du_i   = (alpha / n) 2 pi_i + dpi_i (1 - pi_i^2)
 *   d_head[m][i]         = du_i                                    (d mu)
 *   d_head[m][n_act + i] = [-20 <= log_std_raw <= 2] (du_i eps_i std_i - alpha / n)
 * alpha as in ttl_sac_losses, n = n_rows.  d_head: [n_rows][2 * n_act].
 *
 * head = TTL_HEAD_TANH (the deterministic actor of TD3 / DDPG, td3.py:213-216,
 * ddpg.py:285-288: actor_loss = -Q1(s, tanh(actor(s))).mean(), d_out of the
 * critic's head = -1 / n): d_head[m][i] = dpi_i (1 - pi_i^2), d_head:
 * [n_rows][n_act]; eps, log_std_raw, log_alpha unused. */
TTL_API int ttl_sac_actor_head_backward(const float *dh, int64_t ld_dh, const float *h,
                                        int64_t ld_h, const float *wa, int32_t n_rows,
                                        int32_t n_cols, int32_t n_act, int32_t head,
                                        const float *pi, int64_t ld_pi, const float *eps,
                                        const float *log_std_raw, const float *log_alpha,
                                        float alpha_const, float *d_head, void *hip_stream);

/* The critic loss of TD3 / DDPG (td3.py:147-176, ddpg.py:254-270): q_online,
 * q_target: [n][n_q] (n_q = 2: the double critic, 1: DDPG's single one),
 *   target = r + not_done gamma min_k q_target_k
 *   dq[i][k] = 2 (q_k - target) / n
 * loss_part[block][8] = {0, (q1-target)^2, (q2-target)^2, q1, q2, target, 0, 0}
 * sums per block of 256 rows (NULL to skip); Adam step counters as in
 * ttl_sac_losses. */
TTL_API int ttl_td3_losses(const float *q_online, const float *q_target, const float *reward,
                           const float *not_done, int32_t n, int32_t n_q, float gamma,
                           float *dq, float *loss_part, float *steps, float *adam_consts,
                           double *beta_pows, int32_t n_opt, uint32_t tick_mask, double lr,
                           double beta1, double beta2, void *hip_stream);

/* target = target (1 - tau) + p tau over a flat arena (ddpg.py:300-317) when the
 * Polyak average is not taken in the same pass as the Adam step (TD3 averages
 * the critics' targets only every `agent_freq`-th update, td3.py:205-230). */
TTL_API int ttl_polyak_average(float *target, const float *p, int64_t n, double tau,
                               void *hip_stream);

/* torch.optim.Adam's step (amsgrad off, weight decay 0, maximize off) over a
 * flat arena of n parameters, fused with the Polyak average of the target
 * copy (ddpg.py:300-317, tau * online + (1 - tau) * target) when target !=
 * NULL:
 *   m += (g - m) (1 - beta1);  v = v beta2 + (1 - beta2) g g
 *   p += -step_size * (m / (sqrt(v) / bc2_sqrt + eps))
 *   target = target (1 - tau) + p tau
 * step_size, bc2_sqrt = consts[0], consts[1] (device; written by
 * ttl_sac_losses).  beta1, beta2, eps, tau are torch's Python scalars: 1 - beta1,
 * 1 - beta2 and 1 - tau are formed in float64 and rounded to f32 once, as torch
 * does. */
TTL_API int ttl_adam_polyak(float *p, float *g, float *m, float *v, float *target, int64_t n,
                            const float *consts, double beta1, double beta2, double eps,
                            double tau, void *hip_stream);

/* The temperature step of SACAuto (sac_auto.py:172-174, 219-221): with
 * mean_logp = mean log pi(a|s) over the batch,
 *   g = -(mean_logp + target_entropy)        d alpha_loss / d log_alpha
 *   Adam step on the scalar log_alpha (as ttl_adam_polyak)
 *   grad[0] = g + exp(log_alpha before the step) * mean_logp
 * -- the last line is what autograd leaves in log_alpha.grad: the actor loss
 * (alpha logp - min q).mean() is back-propagated after the temperature has
 * stepped and adds its own d/d log_alpha to the same buffer (sac_auto.py:223). */
TTL_API int ttl_sac_alpha_step(float *log_alpha, float *grad, float *m, float *v,
                               const float *mean_logp, float target_entropy,
                               const float *consts, double beta1, double beta2, double eps,
                               void *hip_stream);

/* The network input rows of one update from a sampled batch, one buffer
 * xs: [3 n][ld], ld >= n_state + n_act:
 *   rows [0, n)    = [s  | a ]   critic loss
 *   rows [n, 2n)   = [s  | . ]   actor loss: pi(s) is written by ttl_thin_forward
 *   rows [2n, 3n)  = [s' | . ]   backup: pi(s') likewise
 * so that rows [0, 2n) are the online critics' batch, rows [n, 3n) (first
 * n_state columns) the actor's and rows [2n, 3n) the target critics'.
 * Optionally (w1 != NULL) wa[i][j] = w1[j * ld_w1 + n_state + i], i < n_act,
 * j < n_w1_rows: the action columns of the critics' stacked first-layer weights,
 * transposed, for ttl_sac_actor_head_backward. */
TTL_API int ttl_build_learner_inputs(const float *state, int64_t ld_s, const float *action,
                                     int64_t ld_a, const float *next_state, int64_t ld_s2,
                                     int32_t n, int32_t n_state, int32_t n_act, float *xs,
                                     int64_t ld, const float *w1, int64_t ld_w1,
                                     int32_t n_w1_rows, float *wa, void *hip_stream);

/* OffPolicyReplayBuffer.add (TrackToLearn/algorithms/shared/replay.py:56-92) for a
 * batch of n transitions already on the device, one launch: transition i goes to
 * ring slot (ptr + i) mod max_size -- state[i], action[i], next_state[row_dest[i]]
 * (row_dest: where env.step_device() wrote the state row of active row i; NULL:
 * row i), reward (float64 as the env returns it, or float32; exactly one array),
 * not_done = 1 - done (uint8).  The caller advances ptr and size. */
TTL_API int ttl_replay_add(const float *state, const float *action, const float *next_state,
                           const int32_t *row_dest, const double *reward_f64,
                           const float *reward_f32, const uint8_t *done, int32_t n,
                           int32_t n_state, int32_t n_act, int64_t ptr, int64_t max_size,
                           float *ring_state, float *ring_action, float *ring_next_state,
                           float *ring_reward, float *ring_not_done, void *hip_stream);

/* OffPolicyReplayBuffer.sample (TrackToLearn/algorithms/shared/replay.py:94-143:
 * `ind = torch.randperm(size)[:batch]`, five `index_select`s) in one launch:
 * `batch` (<= size) DISTINCT ring rows drawn uniformly and gathered into the
 * five output tensors.  Instead of permuting all `size` rows (a 10^6-key sort
 * per training step) position j of a keyed pseudo-random permutation of
 * [0, size) is evaluated for j < batch only: a six-round balanced Feistel
 * network over the next even power of two, cycle-walked into the range.  The
 * caller draws (key0, key1) from its generator at every call.  state /
 * next_state: [size][n_state], action: [size][n_act], reward / not_done:
 * [size] (contiguous rows); out_index (int64, optional): the ring rows taken. */
TTL_API int ttl_replay_sample(const float *state, const float *action,
                              const float *next_state, const float *reward,
                              const float *not_done, int64_t size, int32_t n_state,
                              int32_t n_act, int32_t batch, uint32_t key0, uint32_t key1,
                              float *out_state, float *out_action, float *out_next_state,
                              float *out_reward, float *out_not_done, int64_t *out_index,
                              void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
