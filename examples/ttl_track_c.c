/*
 * ttl_track_c.c -- the C ABI of libttl_hip.so driven from plain C: no Python,
 * no torch.  Tracks a batch of streamlines through a small synthetic volume
 * with the library's scripted policy until every streamline has stopped, the
 * way RLAlgorithm.validation_episode does (TrackToLearn/algorithms/rl.py:58-106:
 * reset -> [policy -> step -> harvest]* -> get_streamlines), and checks what
 * the reference guarantees about the result:
 *   - every stored segment has the step size as its length (env.py:493-502,
 *     tracking_env.py:181-183);
 *   - a streamline stops for a reason (flags != 0), its length is the number of
 *     points it had when it stopped, and LENGTH stops sit at max_nb_steps;
 *   - the survivor counts the steps report fall monotonically to zero.
 *
 *   hipcc -x c examples/ttl_track_c.c -I include -L tracktolearn_amd -lttl_hip \
 *         -Wl,-rpath,'$ORIGIN/../tracktolearn_amd' -lm -o examples/ttl_track_c
 *   (tracktolearn_amd/csrc/build.py:build_example() does exactly that)
 *
 * The mask here is a box: inside a box the cubic B-spline coefficients of
 * scipy.ndimage.spline_filter are not needed exactly -- the example sets the
 * coefficient table to 1 inside / 0 outside, which is a valid coefficient
 * table of SOME smooth mask; the parity of the mask test itself against SciPy
 * is the test suite's business (tests/test_hip_env_parity.py).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ttl_hip.h"

#define D 24          /* volume edge (voxels)          */
#define C 45          /* SH coefficients (order 8)     */
#define PITCH 48      /* floats per packed record      */
#define K 4           /* previous directions in a state */
#define N 20000       /* streamlines                   */
#define MAXS 40       /* max_nb_steps                  */
#define STEP 0.75     /* step size, voxels             */

#define HIP(call)                                                             \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));        \
            return 2;                                                         \
        }                                                                     \
    } while (0)
#define TTL(call)                                                             \
    do {                                                                      \
        int rc_ = (call);                                                     \
        if (rc_ != TTL_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ttl_last_error());  \
            return 3;                                                         \
        }                                                                     \
    } while (0)

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }
static double unif(uint32_t *s) { return (lcg(s) >> 8) / 16777216.0; }

int main(void) {
    if (ttl_abi_version() != TTL_ABI_VERSION) {
        fprintf(stderr, "header ABI %d, library ABI %u\n", TTL_ABI_VERSION, ttl_abi_version());
        return 1;
    }
    const int32_t dim[3] = {D, D, D};
    const size_t n_vox = (size_t)D * D * D;
    uint32_t rng = 12345u;

    /* ---- host volumes: SH coefficients, mask coefficient table, seeds ---- */
    float *sh = (float *)malloc(n_vox * C * sizeof(float));
    double *coef = (double *)malloc(n_vox * sizeof(double));
    float *seeds = (float *)malloc((size_t)N * 3 * sizeof(float));
    if (!sh || !coef || !seeds) return 1;
    for (size_t i = 0; i < n_vox * C; ++i) sh[i] = (float)(unif(&rng) - 0.5);
    const int lo = 5, hi = D - 6;                       /* the box [lo, hi]^3 is "white matter" */
    for (int x = 0; x < D; ++x)
        for (int y = 0; y < D; ++y)
            for (int z = 0; z < D; ++z)
                coef[((size_t)x * D + y) * D + z] =
                    (x >= lo && x <= hi && y >= lo && y <= hi && z >= lo && z <= hi) ? 1.0 : 0.0;
    for (int i = 0; i < N; ++i)
        for (int a = 0; a < 3; ++a)                     /* well inside the box */
            seeds[3 * i + a] = (float)(lo + 2.5 + unif(&rng) * (hi - lo - 4.0));

    /* ---- device buffers (all borrowed by the library) ---- */
    const int64_t n_rec = ttl_sh_volume_records(dim, TTL_SH_BRICK4);
    const int W = 7 * C + 3 * K;
    float *d_sh_src, *d_sh, *d_seeds, *d_hist, *d_state[2], *d_actions;
    double *d_coef;
    uint8_t *d_cls, *d_dones, *d_done_step;
    int32_t *d_flags, *d_lengths, *d_idx_a, *d_idx_b, *h_counts;
    void *d_ws;
    const size_t ws_bytes = ttl_env_workspace_bytes(N);
    HIP(hipMalloc((void **)&d_sh_src, n_vox * C * sizeof(float)));
    HIP(hipMalloc((void **)&d_sh, (size_t)n_rec * PITCH * sizeof(float)));
    HIP(hipMalloc((void **)&d_coef, n_vox * sizeof(double)));
    HIP(hipMalloc((void **)&d_cls, n_vox));
    HIP(hipMalloc((void **)&d_seeds, (size_t)N * 3 * sizeof(float)));
    HIP(hipMalloc((void **)&d_hist, (size_t)N * (MAXS + 1) * 3 * sizeof(float)));
    HIP(hipMalloc((void **)&d_state[0], (size_t)N * W * sizeof(float)));
    HIP(hipMalloc((void **)&d_state[1], (size_t)N * W * sizeof(float)));
    HIP(hipMalloc((void **)&d_actions, (size_t)N * 3 * sizeof(float)));
    HIP(hipMalloc((void **)&d_dones, N));
    HIP(hipMalloc((void **)&d_done_step, N));
    HIP(hipMalloc((void **)&d_flags, (size_t)N * sizeof(int32_t)));
    HIP(hipMalloc((void **)&d_lengths, (size_t)N * sizeof(int32_t)));
    HIP(hipMalloc((void **)&d_idx_a, (size_t)N * sizeof(int32_t)));
    HIP(hipMalloc((void **)&d_idx_b, (size_t)N * sizeof(int32_t)));
    HIP(hipMalloc(&d_ws, ws_bytes));
    HIP(hipHostMalloc((void **)&h_counts, 4 * sizeof(int32_t), 0));   /* pinned, device-visible */
    memset(h_counts, 0, 4 * sizeof(int32_t));
    HIP(hipMemcpy(d_sh_src, sh, n_vox * C * sizeof(float), hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_coef, coef, n_vox * sizeof(double), hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_seeds, seeds, (size_t)N * 3 * sizeof(float), hipMemcpyHostToDevice));
    HIP(hipMemset(d_hist, 0, (size_t)N * (MAXS + 1) * 3 * sizeof(float)));

    /* ---- per-subject setup (BaseEnv.load_subject, env.py:143-282) ---- */
    const double thr = 0.1;
    TTL(ttl_pack_sh_volume(d_sh_src, d_sh, dim, C, PITCH, TTL_SH_BRICK4, NULL));
    TTL(ttl_mask_classes(d_coef, dim, thr, d_cls, NULL));

    ttl_env_desc desc;
    memset(&desc, 0, sizeof(desc));
    if (ttl_env_desc_size() != sizeof(desc)) {
        fprintf(stderr, "descriptor size mismatch\n");
        return 1;
    }
    desc.abi_version = TTL_ABI_VERSION;
    desc.mode = TTL_MODE_F32;
    memcpy(desc.sh_dim, dim, sizeof(dim));
    desc.n_coef = C;
    desc.coef_pitch = PITCH;
    desc.sh_packed = d_sh;
    desc.sh_layout = TTL_SH_BRICK4;
    memcpy(desc.mask_dim, dim, sizeof(dim));
    desc.mask_coef = d_coef;
    desc.mask_threshold = thr;
    desc.mask_classes = d_cls;
    desc.n_dirs = K;
    desc.max_nb_steps = MAXS;
    desc.step_size_vox = STEP;
    desc.neigh_radius_vox = (float)STEP;
    desc.curvature_enabled = 1;
    desc.curv_dot_max = 0.8660254f;            /* theta = 30 degrees */
    desc.n_max = N;
    desc.streamlines = d_hist;
    desc.flags = d_flags;
    desc.lengths = d_lengths;
    desc.dones = d_dones;
    desc.idx_a = d_idx_a;
    desc.idx_b = d_idx_b;
    desc.workspace = d_ws;
    desc.workspace_bytes = ws_bytes;
    ttl_env *env = NULL;
    TTL(ttl_env_create(&desc, &env));

    /* ---- one episode: reset -> [policy -> step -> harvest]* ---- */
    int cur = 0, n_active = N, steps = 0, prev = N;
    long long streamline_steps = 0;
    TTL(ttl_env_reset(env, d_seeds, N, TTL_ORDER_BY_POSITION, d_state[cur], W, NULL));
    while (n_active > 0) {
        const int32_t *d_idx = NULL;
        TTL(ttl_env_view(env, &d_idx, NULL, NULL));
        /* the policy: state rows -> actions (here the library's scripted one) */
        TTL(ttl_scripted_actions(d_state[cur], W, 7 * C, d_idx, n_active, 7u, (uint32_t)steps,
                                 0.3f, d_actions, NULL));
        TTL(ttl_env_step(env, d_actions, NULL, n_active, TTL_ORDER_PARTITION, d_state[cur ^ 1],
                         W, NULL, d_done_step, h_counts, NULL));
        int32_t n_continue = -1;
        TTL(ttl_env_harvest_wait(env, NULL, NULL, W, NULL, &n_continue));
        if (n_continue != h_counts[0] || h_counts[0] + h_counts[1] != n_active ||
            n_continue > prev) {
            fprintf(stderr, "step %d: counts {%d, %d} for %d active rows\n", steps, h_counts[0],
                    h_counts[1], n_active);
            return 4;
        }
        streamline_steps += n_active;
        prev = n_active = n_continue;
        cur ^= 1;                               /* the survivors' rows lead the new state */
        if (++steps > MAXS + 1) {
            fprintf(stderr, "episode did not end\n");
            return 4;
        }
    }
    HIP(hipDeviceSynchronize());

    /* ---- get_streamlines (tracking_env.py:247-294) on the host, and the checks ---- */
    float *hist = (float *)malloc((size_t)N * (MAXS + 1) * 3 * sizeof(float));
    int32_t *flags = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    int32_t *lengths = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    if (!hist || !flags || !lengths) return 1;
    HIP(hipMemcpy(hist, d_hist, (size_t)N * (MAXS + 1) * 3 * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(flags, d_flags, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(lengths, d_lengths, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost));
    long long points = 0, n_mask = 0, n_len = 0, n_curv = 0;
    double worst = 0.0;
    for (int i = 0; i < N; ++i) {
        const int L = lengths[i];
        if (flags[i] == 0 || L < 2 || L > MAXS + 1) {
            fprintf(stderr, "streamline %d: flags %d, length %d\n", i, flags[i], L);
            return 5;
        }
        if ((flags[i] & TTL_FLAG_LENGTH) && L != MAXS) {
            fprintf(stderr, "streamline %d: LENGTH stop at %d points\n", i, L);
            return 5;
        }
        n_mask += (flags[i] & TTL_FLAG_MASK) != 0;
        n_len += (flags[i] & TTL_FLAG_LENGTH) != 0;
        n_curv += (flags[i] & TTL_FLAG_CURVATURE) != 0;
        const float *p = hist + (size_t)i * (MAXS + 1) * 3;
        if (p[0] != seeds[3 * i] || p[1] != seeds[3 * i + 1] || p[2] != seeds[3 * i + 2]) {
            fprintf(stderr, "streamline %d does not start at its seed\n", i);
            return 5;
        }
        for (int s = 1; s < L; ++s) {
            const double dx = (double)p[3 * s] - p[3 * s - 3], dy = (double)p[3 * s + 1] - p[3 * s - 2],
                         dz = (double)p[3 * s + 2] - p[3 * s - 1];
            const double err = fabs(sqrt(dx * dx + dy * dy + dz * dz) - STEP);
            if (err > worst) worst = err;
        }
        points += L;
    }
    if (worst > 1e-5 || points != streamline_steps + N) {
        fprintf(stderr, "segment length off by %.3g; %lld points for %lld streamline-steps\n", worst,
                points, streamline_steps);
        return 5;
    }
    printf("ok: %d streamlines, %d steps, %lld streamline-steps, stops mask/length/curvature "
           "%lld/%lld/%lld, worst |segment| - step %.2g\n",
           N, steps, streamline_steps, n_mask, n_len, n_curv, worst);
    ttl_env_destroy(env);
    return 0;
}
