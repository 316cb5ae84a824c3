/*
 * ttl_hip.h -- C ABI of libttl_hip.so: the MI355X (gfx950) tractography
 * environment step.
 *
 * The reference (levje/TrackToLearn @ 2024-10-24) is pure Python and has no
 * FFI layer; the drop-in boundary is the Python class surface of
 * TrackToLearn/environments/{env,tracking_env,noisy_tracking_env}.py.  The
 * Python host classes in tracktolearn_amd/environments keep that surface and
 * bind the entry points below through ctypes (INTEGRATION.md shows the stub).
 * Each entry point cites the reference code it replaces (paths relative to the
 * reference root, TTL = TrackToLearn).
 *
 * Conventions
 *   - plain C: pointers, sizes, no C++/torch types; return 0 on success, a
 *     negative TTL_ERR_* otherwise, message via ttl_last_error() (thread
 *     local).  Nothing throws or aborts across the ABI.
 *   - every device pointer is BORROWED: the caller (torch, or any hipMalloc
 *     owner) allocates and outlives the handle.  The library allocates no
 *     device memory.
 *   - every call is asynchronous on the caller's HIP stream (`hip_stream` is
 *     a hipStream_t, NULL = default stream).  The only host-visible outputs
 *     are the 4-byte counters written through `host_counts` (pinned memory
 *     supplied by the caller), valid after the stream is synchronised.
 *   - one host thread per handle; handles share nothing.
 */
#ifndef TTL_HIP_H
#define TTL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: only the entry points marked
 * TTL_API below are exported (nm -D shows exactly these). */
#if defined(__GNUC__) || defined(__clang__)
#define TTL_API __attribute__((visibility("default")))
#else
#define TTL_API
#endif

#define TTL_ABI_VERSION 11

#define TTL_OK 0
#define TTL_ERR_INVALID (-1) /* bad argument / shape / alignment             */
#define TTL_ERR_HIP (-2)     /* a HIP runtime call failed                    */
#define TTL_ERR_STATE (-3)   /* call order violated (e.g. step before reset) */
#define TTL_ERR_UNSUPPORTED (-4) /* the configuration has no such path (free-running steps) */

/* Direction arithmetic, TTL/environments/env.py:493-502:
 *   F32      normalise, scale and add in float32 (training env, float32
 *            affine; also any affine under numpy 1.x).
 *   F64DIR   NoisyTrackingEnvironment (noisy_tracking_env.py:65-77): float64
 *            noise is added first, so normalise/scale/add run in float64 and
 *            the new point is float32(float64(p) + d).
 *   F32NORM  float32 normalise, float64 scale/add: what numpy >= 2 computes
 *            for the plain TrackingEnvironment when step_size is np.float64
 *            (float64 affine). */
#define TTL_MODE_F32 0
#define TTL_MODE_F64DIR 1
#define TTL_MODE_F32NORM 2

/* StoppingFlags, TTL/environments/stopping_criteria.py:10-20 */
#define TTL_FLAG_MASK 1
#define TTL_FLAG_LENGTH 2
#define TTL_FLAG_CURVATURE 4
#define TTL_FLAG_ORACLE 64

/* Row order of the state rows written by ttl_env_step():
 *   ORDER_ACTIVE    row i of the output = i-th active streamline (the order
 *                   of continue_idx), exactly tracking_env.py:214-221.
 *   ORDER_PARTITION survivors first (stable), then the streamlines that just
 *                   stopped (stable): rows [0, n_continue) are already the
 *                   harvest() result, so harvest copies nothing.  row_dest[i]
 *                   maps active row i to its output row. */
#define TTL_ORDER_ACTIVE 0
#define TTL_ORDER_PARTITION 1

typedef struct ttl_env_desc {
    uint32_t abi_version;  /* TTL_ABI_VERSION */
    int32_t mode;          /* TTL_MODE_*      */

    /* SH volume, packed by ttl_pack_sh_volume(): voxel records of coef_pitch
     * f32, in the order sh_layout says */
    int32_t sh_dim[3];
    int32_t n_coef;        /* C (45 for SH order 8)                          */
    int32_t coef_pitch;    /* floats per voxel record, multiple of 4, >= C   */
    const float *sh_packed;
    float sh_coord_shift;  /* added to coordinates before the gather; 0.0    */
                           /* (voxel i sits at coordinate i)  */
    int32_t sh_layout;     /* TTL_SH_LINEAR or TTL_SH_BRICK4                 */

    /* tracking mask: cubic B-spline coefficients, [X][Y][Z] f64
     * (scipy.ndimage.spline_filter(order=3), stopping_criteria.py:58-59)    */
    int32_t mask_dim[3];
    const double *mask_coef;
    double mask_threshold; /* env_dto['binary_stopping_threshold']           */
    /* optional [X][Y][Z] u8 from ttl_mask_classes(mask_coef, mask_dim,
     * mask_threshold): cells where the spline cannot cross the threshold are
     * decided without evaluating it (same decisions, bit for bit); or NULL  */
    const uint8_t *mask_classes;

    /* fODF peaks for the alignment reward: [X][Y][Z][15] f32, or NULL       */
    int32_t peaks_dim[3];
    const float *peaks;
    int32_t compute_reward;
    double alignment_weighting;

    /* tracking parameters (env.py:196-213) */
    int32_t n_dirs;        /* K previous directions in the state             */
    int32_t max_nb_steps;  /* int(max_length / step_size_mm)                 */
    double step_size_vox;  /* convert_length_mm2vox(step_size_mm, affine)    */
    float neigh_radius_vox;/* neighbourhood radius (float32, env.py:207-213) */
    int32_t curvature_enabled;
    float curv_dot_max;    /* too curvy  <=>  -1 <= dot <= curv_dot_max; the
                              host derives it from numpy's own float32 arccos
                              (utils.py:172-173), see curvature_threshold()  */

    /* per-streamline state, capacity n_max rows (tracking_env.py:110-124)   */
    int32_t n_max;
    float *streamlines;    /* [n_max][max_nb_steps+1][3] f32                 */
    int32_t *flags;        /* [n_max] StoppingFlags bitmask                  */
    int32_t *lengths;      /* [n_max]                                        */
    uint8_t *dones;        /* [n_max]                                        */
    int32_t *idx_a;        /* [n_max] continue_idx, double buffered          */
    int32_t *idx_b;        /* [n_max]                                        */
    void *workspace;       /* ttl_env_workspace_bytes(n_max) bytes           */
    size_t workspace_bytes;
} ttl_env_desc;

typedef struct ttl_env ttl_env;

/* Bytes of device scratch the handle needs for n_max streamlines. */
TTL_API size_t ttl_env_workspace_bytes(int32_t n_max);

/* Order of the packed voxel records:
 *   TTL_SH_LINEAR  [X][Y][Z] as the source volume;
 *   TTL_SH_BRICK4  4x4x4-voxel bricks, each 64 consecutive records
 *                  ([X/4][Y/4][Z/4] bricks, [4][4][4] voxels inside; dims
 *                  rounded up to multiples of 4, padding records zero): the
 *                  4x4x4 neighbourhood one streamline gathers then lives in
 *                  at most 8 contiguous 64-record runs instead of 16 runs of
 *                  4 records spread over Y*Z- and Z-record strides. */
#define TTL_SH_LINEAR 0
#define TTL_SH_BRICK4 1

/* Number of voxel records ttl_pack_sh_volume() writes for a volume. */
TTL_API int64_t ttl_sh_volume_records(const int32_t *dim /*[3]*/, int32_t layout);

/* Device memory for a volume the step gathers from (the packed SH volume):
 * try_contiguous 0: plain hipMalloc; 1: physically contiguous
 * (hipExtMallocWithFlags, hipDeviceMallocContiguous) when the driver grants it;
 * 2: through the virtual-memory API (hipMemCreate + hipMemMap at the
 * recommended granularity), hipMalloc when that fails; *contiguous_out says
 * which was granted (measurement showed no kind to be better placed than
 * another; the host classes use 0).  device < 0: the calling thread's
 * current device.  These two entry points are the only ones that own device
 * memory; everything else is borrowed.  They exist because WHERE the 170 MB
 * volume of the bench lands in physical memory moves the state gather between
 * 0.18 and 0.20 ms on the same GPU, and so does where the state rows land
 * (DESIGN.md 3.3): the host classes put a subject's volume and a ring of state
 * buffers into a few allocations obtained here, time the step's own gather on
 * every pair at the first large reset and keep the fastest; the caching
 * allocator would hand the same block back every time. */
TTL_API int ttl_volume_alloc(int32_t device, size_t bytes, int32_t try_contiguous, void **out,
                     int32_t *contiguous_out);
TTL_API int ttl_volume_free(void *ptr);


/* Repack an SH volume [X][Y][Z][C] f32 (the layout of
 * TTL/environments/env.py:169-180 `data_volume`) into 16-byte aligned voxel
 * records of coef_pitch floats, zero padded, in `layout` order; dst holds
 * ttl_sh_volume_records(dim, layout) records.  Once per subject. */
TTL_API int ttl_pack_sh_volume(const float *src, float *dst, const int32_t *dim /*[3]*/,
                       int32_t n_coef, int32_t coef_pitch, int32_t layout,
                       void *hip_stream);

/* Per-cell shortcut table for the mask test (see ttl_env_desc.mask_classes):
 * 1 = all 64 spline taps of the cell are >= threshold, 2 = all are below,
 * 0 = undecided.  Once per subject and threshold. */
TTL_API int ttl_mask_classes(const double *mask_coef, const int32_t *dim /*[3]*/,
                     double threshold, uint8_t *classes_out, void *hip_stream);

/* Validates the descriptor and creates a handle (no device work).
 * Replaces the per-subject setup of BaseEnv.load_subject, env.py:143-281. */
TTL_API int ttl_env_create(const ttl_env_desc *desc, ttl_env **out);
TTL_API void ttl_env_destroy(ttl_env *env);

/* TrackingEnvironment.reset / nreset (tracking_env.py:91-133 / 47-89):
 * n seeds (float32 [n][3], voxel space) become streamlines 0..n-1 of one
 * point; flags 0, lengths 1, dones 0, continue_idx = arange(n); writes the
 * first state rows ([n][state_pitch] f32, state_pitch >= 7*C + 3*K).
 * processing_order (device int32 [n], a permutation of 0..n-1, or NULL) does
 * not change any result: it only chooses which streamlines one workgroup of
 * the state gather handles together.  Passing the seeds sorted by spatial
 * brick lets neighbouring workgroups share voxels through L2; the library
 * keeps the order compacted as streamlines stop.  TTL_ORDER_BY_POSITION asks
 * the library to build that order itself (as ttl_env_refresh_processing_order
 * does later in the episode). */
#define TTL_ORDER_BY_POSITION ((const int32_t *)(uintptr_t)1)
TTL_API int ttl_env_reset(ttl_env *env, const float *seeds, int32_t n,
                  const int32_t *processing_order, float *state_out,
                  int64_t state_pitch, void *hip_stream);

/* TrackingEnvironment.step (tracking_env.py:135-221) for the n_active
 * streamlines of continue_idx (n_active must equal the count the last
 * reset/harvest reported):
 *   actions   [n_active][3] f32 (row i belongs to continue_idx[i])
 *   noise     [n_active][3] f64 added to the action first (F64DIR mode,
 *             noisy_tracking_env.py:73-77), or NULL for +0.0
 *   state_out [n_active][state_pitch] f32, row order per `order`
 *   reward_out[n_active] f64 or NULL  (row i = active row i, always)
 *   done_out  [n_active] u8           (row i = active row i, always)
 *   host_counts pinned host memory, 4 x int32, or NULL: receives
 *             {n_continue, n_stopped, sequence, reserved}.  When the buffer is
 *             device-visible (hipHostMalloc / torch pin_memory) the kernel
 *             that computes the counts -- k_prefix, before the state gather
 *             has even started, or the one-launch tail of batches of at most
 *             16384 rows -- writes them and then a process-unique sequence
 *             number (bit 30 set, so that it can never equal the "steps done"
 *             count a free-running episode leaves in the same word) straight
 *             into it, and ttl_env_wait_counts() polls that
 *             word: the host can queue the next step while this one's gather
 *             is still running, with no copy kernel competing for the CUs.
 *             Otherwise (or with TTL_POLL_COUNTS=0) the counts are copied on
 *             a side stream as soon as the stopping decisions are final.
 * Normalise + scale the action, first-step flip, grow by one point, LENGTH /
 * CURVATURE / MASK stopping tests, flags and dones, alignment reward, new
 * state.  continue_idx itself only changes in ttl_env_harvest(). */
TTL_API int ttl_env_step(ttl_env *env, const float *actions, const double *noise,
                 int32_t n_active, int32_t order, float *state_out,
                 int64_t state_pitch, double *reward_out, uint8_t *done_out,
                 int32_t *host_counts, void *hip_stream);

/* The same step in two halves, for stopping criteria evaluated outside the
 * library (OracleStoppingCriterion, stopping_criteria.py:85-154, needs the new
 * points and a transformer forward):
 *   ttl_env_step_begin  actions -> new points, LENGTH/CURVATURE/MASK
 *                       decisions, reward, done_out;
 *   (caller computes extra_flags[n_active] u8, e.g. TTL_FLAG_ORACLE per row)
 *   ttl_env_step_end    ORs extra_flags (may be NULL) into flags / dones /
 *                       done_out, compacts, writes the state rows.
 * ttl_env_step() == begin + end(NULL). */
TTL_API int ttl_env_step_begin(ttl_env *env, const float *actions, const double *noise,
                       int32_t n_active, double *reward_out, uint8_t *done_out,
                       void *hip_stream);
TTL_API int ttl_env_step_end(ttl_env *env, const uint8_t *extra_flags, int32_t order,
                     float *state_out, int64_t state_pitch, int32_t *host_counts,
                     void *hip_stream);

/* Blocks the calling host thread until the host_counts of the last
 * ttl_env_step() have landed (not until the step has finished).  Called after
 * ttl_env_harvest() it also tells the handle the exact survivor count, and
 * the next ttl_env_step() is then refused unless n_active equals it. */
TTL_API int ttl_env_wait_counts(ttl_env *env);

/* Between a step that was given host_counts and its harvest: waits for the
 * counts like ttl_env_wait_counts() (without consuming them) and returns the
 * rows that stopped in that step, compacted in active-row order -- what
 * OracleReward scores (oracle_reward.py:78-90: idx = arange(N)[dones]):
 * stop_list[2 q] = active row, stop_list[2 q + 1] = streamline id (row of the
 * history buffer) of the q-th of them, q < *n_stopped.  Device memory owned by
 * the handle, rewritten by the next step. */
TTL_API int ttl_env_stopped(ttl_env *env, const int32_t **stop_list, int32_t *n_stopped);

/* TrackingEnvironment.harvest (tracking_env.py:223-245): lengths of the
 * streamlines that stopped in the last step, continue_idx <- survivors
 * (stable).  If the last step used ORDER_ACTIVE and state_out != NULL the
 * survivors' rows are copied from state_in (the step's output) to the first
 * n_continue rows of state_out.  After an ORDER_PARTITION step nothing is
 * launched (the lengths were written by the step itself). */
TTL_API int ttl_env_harvest(ttl_env *env, const float *state_in, float *state_out,
                    int64_t state_pitch, void *hip_stream);

/* ttl_env_harvest() followed by ttl_env_wait_counts() in one call (one trip
 * through the FFI instead of two; small batches are host bound): returns the
 * survivor count through n_continue_out.  The last step must have been given
 * host_counts. */
TTL_API int ttl_env_harvest_wait(ttl_env *env, const float *state_in, float *state_out,
                         int64_t state_pitch, void *hip_stream,
                         int32_t *n_continue_out);

/* Free-running steps (ABI v7).  The tracking loops of the reference
 * (RLAlgorithm.validation_episode, rl.py:58-106, driven by Tracker.track,
 * tracker.py:110-116) run policy -> step -> harvest until every streamline of
 * the batch has stopped; with the default --n_actor (10 000) a step is a few
 * microseconds of GPU work and the loop is bound by the host's launches and by
 * the survivor count it fetches every step.  Between ttl_env_freerun_begin()
 * and ttl_env_freerun_end() the number of active rows, the current length and
 * the live continue_idx buffer are kept in device memory and advanced by the
 * step's own kernels, so ttl_env_freerun_step() is nothing but launches with
 * fixed arguments: it may be captured in a HIP graph (together with the policy
 * network that turns state rows into actions) and replayed until the pinned
 * words report no survivor.  Results are those of ttl_env_step(ORDER_PARTITION)
 * + ttl_env_harvest() on the same actions, row for row.
 *
 * begin: after a reset or a harvested step whose count was read; batches of at
 *   most 16 384 rows (TTL_ERR_UNSUPPORTED otherwise, also for a neighbourhood
 *   radius outside (0, 1) voxel).  host_counts (may be NULL): 4 int32 of
 *   pinned, device-visible memory; every step writes {n_continue, n_stopped,
 *   steps done since begin} there.
 * step: covers rows 0..n_rows-1: actions [n_rows][3] f32, state_out
 *   [n_rows][state_pitch] f32, reward_out [n_rows] f64 or NULL, done_out
 *   [n_rows] u8.  n_rows is at most the row count at begin and must not be less
 *   than the number of rows active now -- the count a previous step reported
 *   through host_counts is such a bound (counts only fall), so a host loop may
 *   shrink its action batches with the survivors without ever waiting for the
 *   GPU; a step launched for too few rows does nothing but count itself.
 *   Rows 0..n_active-1 (n_active as the previous step left it) are the active
 *   rows; state rows come out survivors first (stable), then the rows that
 *   stopped in this step; rows that left earlier get done = 1, reward = 0 and
 *   keep their old state row.  No Gaussian action noise (TTL_MODE_F64DIR adds
 *   +0.0).
 * end: waits for the stream, reads the device words back into the handle
 *   (which then continues as after a harvest) and returns them. */
TTL_API int ttl_env_freerun_begin(ttl_env *env, int32_t *host_counts, void *hip_stream);
TTL_API int ttl_env_freerun_step(ttl_env *env, const float *actions, int32_t n_rows,
                         float *state_out, int64_t state_pitch, double *reward_out,
                         uint8_t *done_out, void *hip_stream);
TTL_API int ttl_env_freerun_end(ttl_env *env, int32_t *n_active_out, int32_t *length_out,
                        int32_t *steps_out, void *hip_stream);
/* Measurement only: ttl_scripted_actions() for a free-running step -- the row
 * count, the step number (length - 1) and the live continue_idx buffer are
 * read from the device words; rows 0..min(n_active, n_rows)-1 are written. */
TTL_API int ttl_env_freerun_scripted_actions(ttl_env *env, const float *state, int64_t state_pitch,
                                     int32_t dir_offset, int32_t n_rows, uint32_t seed,
                                     float wobble, float *actions_out, void *hip_stream);

/* BaseEnv._compute_stopping_flags (env.py:567-603) on caller-supplied points:
 * tail [n][3][3] f32 holds the last three points (oldest first) of n
 * streamlines that have n_points points each (with n_points == 2 the first
 * entry of a tail is ignored, with n_points == 1 the first two are).
 * flags_out[n] u8 = OR of TTL_FLAG_* (0 = keeps going).  Touches no env state. */
TTL_API int ttl_env_stopping_flags(ttl_env *env, const float *tail, int32_t n,
                           int32_t n_points, uint8_t *flags_out,
                           void *hip_stream);

/* Replaces the processing order of the state gather between a harvest and the
 * next step: order[n] (device, int32) is a permutation of the n currently
 * active rows.  Like the order given to ttl_env_reset it is a scheduling hint
 * only (which rows a workgroup gathers together; tracking_env.py fixes the
 * ROW order, not this): results do not depend on it.  The host classes
 * refresh it every few steps from the streamlines' current positions, because
 * an order sorted by seed position decays as the streamlines travel. */
TTL_API int ttl_env_set_processing_order(ttl_env *env, const int32_t *order, int32_t n,
                                 void *hip_stream);

/* The same, computed by the library: active rows sorted by the 8^3-voxel brick
 * of their streamline's newest point (key kernel + counting sort on workspace
 * memory, no allocation).  Between a harvest and the next step, after the
 * survivor count has been read back. */
TTL_API int ttl_env_refresh_processing_order(ttl_env *env, void *hip_stream);

/* Current continue_idx buffer (device, int32 [n_active]) and, after a step,
 * the active-row -> output-row map (device, int32 [n_active]). */
TTL_API int ttl_env_view(ttl_env *env, const int32_t **continue_idx,
                 const int32_t **row_dest, int32_t *length);

/* Measurement support (bench.py): while profiling is on, every kernel launched
 * by ttl_env_step() is bracketed by HIP events on the caller's stream.
 * ttl_env_profile_end() synchronises those events and returns, per kernel
 * class {0: advance, 1: prefix, 2: state gather, 3: processing-order
 * compaction (ABI v9)}, the summed duration in ms and the number of launches,
 * then switches profiling off. */
#define TTL_PROFILE_CLASSES 4
TTL_API int ttl_env_profile_begin(ttl_env *env, int32_t max_launches,
                          int32_t class_mask /* bit k = time class k */);
TTL_API int ttl_env_profile_end(ttl_env *env, double *total_ms /*[TTL_PROFILE_CLASSES]*/,
                        int32_t *n_launches /*[TTL_PROFILE_CLASSES]*/);

/* Scripted, policy-free actions for "env.step only" runs (stands in
 * for agent.select_action, TTL/algorithms/rl.py:91): step 0 -> a random
 * vector; later -> unit(previous segment) + wobble * noise, the previous
 * segment being state[i][dir_offset .. dir_offset+2] (dir_offset = 7*C).  The
 * noise is a counter-based hash of (seed, step, continue_idx[i], component),
 * reproduced bit for bit by oracle/scripted_policy.py. */
TTL_API int ttl_scripted_actions(const float *state, int64_t state_pitch,
                         int32_t dir_offset, const int32_t *continue_idx,
                         int32_t n, uint32_t seed, uint32_t step, float wobble,
                         float *actions_out, void *hip_stream);

/* fODF peaks of an SH volume (replaces the per-voxel Python loop of
 * TrackToLearn/environments/env.py:405-432: sh_to_sf_matrix + get_maximas per
 * voxel, 5 peaks scaled by value / first value).  sh: [n_voxels][n_coef] f32;
 * sf_matrix: [n_coef][n_vertices] f32 (SF = SH @ matrix); vertices:
 * [n_vertices][3] unit vectors of one hemisphere; neighbours:
 * [n_vertices][degree] vertex indices of the hemisphere graph (rows padded with
 * the vertex itself).  A direction is a peak if its SF value (values below
 * absolute_threshold count as 0) is >= all neighbours, > one of them and > 0;
 * peaks are taken in decreasing order among the max_candidates largest, kept
 * while value - max(min SF, 0) >= relative_threshold * that of the first and
 * |cos| to every kept peak <= min_separation_cos.  peaks_out:
 * [n_voxels][3*npeaks] f32, zeros for voxels whose coefficients sum to 0 and
 * for missing peaks.  All pointers are device memory. */
TTL_API int ttl_peaks_from_sh(const float *sh, int64_t n_voxels, int32_t n_coef,
                      const float *sf_matrix, const float *vertices,
                      const int32_t *neighbours, int32_t n_vertices, int32_t degree,
                      int32_t npeaks, float relative_threshold, float absolute_threshold,
                      float min_separation_cos, int32_t max_candidates, float *peaks_out,
                      void *hip_stream);

/* TractOracle-Net forward (TrackToLearn/oracles/transformer_oracle.py:77-92 under the
 * autocast of oracles/oracle.py:76): scores[i] = sigmoid(head(encoder(embed([CLS; dirs[i]]))[0]))
 * for n sequences of 127 segment vectors, ONE launch (csrc/ttl_oracle_net.hip): one
 * workgroup per streamline for n <= 512, one wavefront per streamline above (the environment
 * variable TTL_ORACLE_NET_WG = 1 / 0 forces the first / the second).  Each kernel is
 * deterministic and scores a row independently of its batch; the two agree to the rounding of
 * the fp16 score (the order of one float32 sum differs).  d_model 32, 128 tokens, n_head in
 * {1, 2, 4}, ReLU post-norm layers, ff_dim a multiple of 32, at most 8192 (TTL_ERR_UNSUPPORTED
 * otherwise).  The weights come packed by
 * tracktolearn_amd/oracles/fused_net.py:pack_oracle_net (fp16 MFMA fragments in the k order an
 * accumulator tile presents, per-row vectors in accumulator row order):
 *   packed_half  [n_layers][8 + 4 ff_dim / 32][64][8] f16: W_q, W_k, W_v, W_o, W_1 chunks, W_2 chunks
 *   packed_float [n_layers][288 + ff_dim] f32: b_q, b_k, b_v, b_o, LN1 gain / bias, b_2,
 *                LN2 gain / bias, b_1 chunks
 *   embed [2][16][4], cls [3], pos_enc [4][64][16], head [33].
 * dirs: [n][127][3] f32; scores: [n] f32.  Device pointers. */
TTL_API int ttl_oracle_net_forward(const float *dirs, int64_t n, const void *packed_half,
                                   const float *packed_float, const float *embed,
                                   const float *cls, const float *pos_enc, const float *head,
                                   int32_t n_layers, int32_t n_head, int32_t ff_dim,
                                   float *scores, void *hip_stream);

/* Arc-length resampling of a padded batch of streamlines to nb_points points
 * each (the oracle's input, TrackToLearn/oracles/oracle.py:52,70: dipy
 * set_number_of_points).  points: [n] rows of row_pitch floats holding up to
 * max_len xyz points; lengths32 or lengths64 (exactly one non-null): valid
 * points per row; out: [n][nb_points][3] f32.  First and last point are kept,
 * the others sit at arc length k * total / (nb_points - 1) (float64), linearly
 * interpolated inside their segment.  Device pointers. */
TTL_API int ttl_resample_streamlines(const float *points, int64_t row_pitch,
                             const int32_t *lengths32, const int64_t *lengths64,
                             int32_t n, int32_t max_len, int32_t nb_points, float *out,
                             void *hip_stream);

/* The env's oracle path in one launch (oracle_reward.py:78-90,
 * stopping_criteria.py:132-150 -> oracles/oracle.py:52-72): for streamline
 * ids[r * id_stride] (NULL: r) of the history buffer, r < n, take its first
 * n_points points, map them with the row-major 3x3 matrix lin (host memory,
 * p' = p @ lin; NULL: none), resample to nb_points along the arc length as
 * ttl_resample_streamlines() does, and write the nb_points - 1 segment vectors:
 * dirs_out [n][nb_points - 1][3] float32, the network's input. */
TTL_API int ttl_oracle_segments(const float *history, int64_t row_pitch, const int32_t *ids,
                                int32_t id_stride, int32_t n, int32_t n_points,
                                const float *lin, int32_t nb_points, float *dirs_out,
                                void *hip_stream);

/* OracleReward's sparse bonus (oracle_reward.py:84-93) for the rows of
 * ttl_env_stopped(): term[0 .. n_active) = 0, then term[row_q] = bonus where
 * scores[q] > 0.5 (q < n_scored <= n_stopped; the rest was not scored), and
 * reward[row_q] += term[row_q]. */
TTL_API int ttl_oracle_bonus(const float *scores, int32_t n_scored, const int32_t *stop_list,
                             int32_t n_stopped, double bonus, int32_t n_active, double *term,
                             double *reward, void *hip_stream);

/* Ragged pack of tracked streamlines (ABI v8), the device side of
 * TrackingEnvironment.get_streamlines (tracking_env.py:263-284) and of the
 * multi-GPU collate: history [n] rows of row_pitch floats (the env's
 * `streamlines` buffer, row_pitch = 3 * (max_nb_steps + 1)); keep[i] points of
 * row i (int64; lengths minus the point a CURVATURE / MASK stop drops) are
 * copied to points_out + 3 * offsets[i] (offsets = exclusive prefix of keep,
 * int64).  One wave per streamline, coalesced; device pointers. */
TTL_API int ttl_pack_streamlines(const float *history, int64_t row_pitch, const int64_t *keep,
                         const int64_t *offsets, int32_t n, float *points_out,
                         void *hip_stream);

TTL_API const char *ttl_last_error(void);
TTL_API uint32_t ttl_abi_version(void);
/* sizeof(ttl_env_desc) as compiled: a binding checks its own struct against it */
TTL_API size_t ttl_env_desc_size(void);

#ifdef __cplusplus
}
#endif
#endif /* TTL_HIP_H */
