// ttl_hip.hip -- MI355X (gfx950 / CDNA4) kernels and C ABI for the vectorised
// tractography environment step.  Written for wave64 only; built with
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// (contraction off: the stopping decisions must round exactly like the
// NumPy/SciPy arithmetic of the reference, which never fuses a*b+c).
//
// What runs where (reference = levje/TrackToLearn, TTL = TrackToLearn):
//   k_advance  : TTL/environments/env.py:493-502 (_format_actions),
//                noisy_tracking_env.py:73-77 (f64 noise add),
//                tracking_env.py:165-183 (first-step flip, position update),
//                env.py:567-603 + utils.py:127-173 + stopping_criteria.py:79-82
//                (LENGTH / CURVATURE / MASK tests, flags, dones),
//                reward.py:46-79 + local_reward.py:29-107 (alignment reward),
//                and the wave-ballot survivor ranks for the compaction.
//   k_prefix   : tracking_env.py:192-195 / 238-241 (stable index compaction:
//                new_continue_idx = continue_idx[~stopping]).
//   k_state*   : env.py:504-565 (_format_state): 7-point trilinear gather of
//                the SH volume + last K segment vectors -> ttl_state.hip.
//   k_finish   : tracking_env.py:236 (lengths[stopping_idx] = length).
//   k_copy_rows: tracking_env.py:245 (state[continue_idx]).
// Other translation units of libttl_hip.so: ttl_order.hip (processing order of
// the gather), ttl_peaks.hip (fODF peaks), ttl_resample.hip (oracle input).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "ttl_hip.h"

#include "ttl_internal.h"

#include <map>
#include <mutex>

namespace {
thread_local char g_err[512] = "";
}

// the one error slot of the library (ttl_last_error), shared with the other
// translation units through ttl_internal.h
int ttl_detail_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

constexpr int BLOCK = TTL_BLOCK;


// ---------------------------------------------------------------------------
// mask test: scipy.ndimage.map_coordinates(coef, p - 0.5, order=3,
// mode='constant', cval=0, prefilter=False) < thr, restated tap for tap
// (oracle/env_oracle.py: spline3_sample).  float64, sequential
// accumulation, no FMA.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int mirror_fold(int i, int n) {
    if (n <= 1) return 0;
    const int s2 = 2 * n - 2;
    if (i < 0) {
        i = s2 * ((-i) / s2) + i;
        i = (i <= 1 - n) ? i + s2 : -i;
    } else if (i >= n) {
        i -= s2 * (i / s2);
        if (i >= n) i = s2 - i;
    }
    return i;
}

__device__ __forceinline__ void cubic_weights(double c, double fl, double *w) {
    const double y = c - fl;
    const double z = 1.0 - y;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = z * z * z / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}

__device__ bool outside_mask(const EnvParams &P, float px, float py, float pz) {
    // coords = streamlines[:, -1, :].T - 0.5 : float32 subtraction
    const double cx = (double)(px - 0.5f);
    const double cy = (double)(py - 0.5f);
    const double cz = (double)(pz - 0.5f);
    const int nx = P.mask_dim[0], ny = P.mask_dim[1], nz = P.mask_dim[2];
    // outside [0, n-1] on any axis (or NaN) -> constant 0.0
    const bool inside = (cx >= 0.0 && cx <= (double)(nx - 1)) &&
                        (cy >= 0.0 && cy <= (double)(ny - 1)) &&
                        (cz >= 0.0 && cz <= (double)(nz - 1));
    if (!inside) return 0.0 < P.mask_thr;
    const double fx = floor(cx), fy = floor(cy), fz = floor(cz);
    if (P.mask_cls) {
        // cubic B-spline weights are non-negative and sum to 1, so the value
        // lies between the smallest and the largest of the 64 taps: cells
        // whose taps are all on one side of the threshold (with a rounding
        // margin, k_mask_classes) are decided by this one byte
        const uint8_t cls =
            P.mask_cls[((size_t)(int)fx * ny + (int)fy) * nz + (int)fz];
        if (cls == 1) return false;
        if (cls == 2) return true;
    }
    double wx[4], wy[4], wz[4];
    cubic_weights(cx, fx, wx);
    cubic_weights(cy, fy, wy);
    cubic_weights(cz, fz, wz);
    const int sx = (int)fx - 1, sy = (int)fy - 1, sz = (int)fz - 1;
    double t = 0.0;
    if (sx >= 0 && sy >= 0 && sz >= 0 && sx + 3 < nx && sy + 3 < ny && sz + 3 < nz) {
        // interior: no border folding, the 4 z-taps of a (x, y) line are 32
        // contiguous bytes (8-byte aligned) -> two 16-byte loads
        struct __attribute__((aligned(8))) taps4 {
            double v[4];
        };
        const double *base = P.mask_coef + ((size_t)sx * ny + sy) * nz + sz;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const taps4 q = *reinterpret_cast<const taps4 *>(
                    base + ((size_t)a * ny + b) * nz);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    double v = q.v[d];
                    v = v * wx[a];
                    v = v * wy[b];
                    v = v * wz[d];
                    t = t + v;
                }
            }
        }
        return t < P.mask_thr;
    }
    int ix[4], iy[4], iz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ix[j] = mirror_fold(sx + j, nx) * ny;
        iy[j] = mirror_fold(sy + j, ny);
        iz[j] = mirror_fold(sz + j, nz);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double *line = P.mask_coef + (size_t)(ix[a] + iy[b]) * nz;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                double v = line[iz[d]];
                v = v * wx[a];
                v = v * wy[b];
                v = v * wz[d];
                t = t + v;
            }
        }
    }
    return t < P.mask_thr;
}

// tracking_env.py:165-178: at the first step all criteria run on the 2-point
// trial streamline; only LENGTH (2 >= max_nb_steps) and MASK can fire there.
// NOTE (ROCm 7.2 hipcc): written as `a || outside_mask(..)` followed by
// `if (flip) d = -d`, the divergent arm lost its body in the ISA (the
// direction was never negated).  Keep the non-short-circuit form below and
// apply the flip with selects; tests/test_hip_env_parity.py covers it.
__device__ __forceinline__ bool first_step_flips(const EnvParams &P, float tx,
                                                 float ty, float tz) {
    const bool out = outside_mask(P, tx, ty, tz);
    return out | (2 >= P.max_nb_steps);
}

// normalize_vectors on one float32 3-vector: v / sqrt((x*x + y*y) + z*z)
__device__ __forceinline__ void unit3(float x, float y, float z, float &ox,
                                      float &oy, float &oz) {
    const float s = sqrtf((x * x + y * y) + z * z);
    ox = x / s;
    oy = y / s;
    oz = z / s;
}

// numpy.nan_to_num on float32
__device__ __forceinline__ float nan_to_num(float v) {
    if (v != v) return 0.0f;
    if (v == INFINITY) return 3.4028234663852886e38f;
    if (v == -INFINITY) return -3.4028234663852886e38f;
    return v;
}

// BaseEnv._compute_stopping_flags (env.py:567-603) for one streamline of n_pts
// points whose last three points are p0, p1, p2 (oldest first): LENGTH
// (utils.py:142), CURVATURE (utils.py:162-173), MASK
// (stopping_criteria.py:79-82).  Also returns the unit last segment u and
// unit previous segment w (zero when n_pts < 3) for the reward.
__device__ __forceinline__ int stopping_bits(
    const EnvParams &P, float p0x, float p0y, float p0z, float p1x, float p1y,
    float p1z, float p2x, float p2y, float p2z, int n_pts, float &ux, float &uy,
    float &uz, float &wx, float &wy, float &wz) {
    int bits = 0;
    if (n_pts >= P.max_nb_steps) bits |= TTL_FLAG_LENGTH;
    // segments are recomputed in float32 from the stored positions
    unit3(p2x - p1x, p2y - p1y, p2z - p1z, ux, uy, uz);
    wx = wy = wz = 0.f;
    if (n_pts >= 3) {
        unit3(p1x - p0x, p1y - p0y, p1z - p0z, wx, wy, wz);
        if (P.curv_enabled) {
            const float dot = (ux * wx + uy * wy) + uz * wz;
            // arccos(dot) > theta  <=>  -1 <= dot <= curv_dot_max (NaN and
            // |dot| > 1 give NaN angles -> not curvy)
            if (dot <= P.curv_dot_max && dot >= -1.0f) bits |= TTL_FLAG_CURVATURE;
        }
    }
    if (outside_mask(P, p2x, p2y, p2z)) bits |= TTL_FLAG_MASK;
    return bits;
}

// Stable survivor ranks of one 256-thread block: 64-bit ballot per wavefront,
// popcount of the lanes below, then the 4 wave totals through LDS.  Writes
// rank[i] (survivors before row i inside the block) and the block's count.
__device__ __forceinline__ void block_ranks(int *__restrict__ rank_out,
                                            int *__restrict__ counts_out, int i,
                                            bool active, bool keep) {
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int below = __popcll(m & ((1ull << lane) - 1ull));
    __shared__ int wave_total[BLOCK / 64];
    if (lane == 0) wave_total[wave] = __popcll(m);
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w)
        if (w < wave) before += wave_total[w];
    if (active) rank_out[i] = before + below;
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) tot += wave_total[w];
        counts_out[blockIdx.x] = tot;
    }
}

__device__ __forceinline__ void block_survivor_ranks(const EnvParams &P, int i,
                                                     bool active, bool keep) {
    block_ranks(P.rank, P.block_counts, i, active, keep);
}

// ---------------------------------------------------------------------------
// k_advance: one thread per active streamline.
// ---------------------------------------------------------------------------
// One active row of a step: streamline g = idx[i] grows by one point, the
// stopping criteria are tested on it, reward / done / head records are written.
// Returns whether the streamline stops.
template <int MODE>
__device__ __forceinline__ bool advance_row(
    const EnvParams &P, int g, int i, const float *__restrict__ actions,
    const double *__restrict__ noise, int L, double *__restrict__ reward_out,
    uint8_t *__restrict__ done_out) {
    bool stop = false;
    {
        float *h = P.hist + (size_t)g * (size_t)(P.max_nb_steps + 1) * 3;
        // the two newest points come from a compact per-streamline array (32 B
        // each, walked in ascending id order = nearly sequentially), not from
        // the history rows (3.2 KB apart: one scattered DRAM sector per
        // streamline, which is what bounded this kernel)
        float4 *l2 = reinterpret_cast<float4 *>(P.last2 + 8 * (size_t)g);
        const float4 q0 = l2[0], q1 = l2[1];
        const float p0x = q0.x, p0y = q0.y, p0z = q0.z;     // zeros while L == 1
        const float p1x = q1.x, p1y = q1.y, p1z = q1.z;
        const float ax = actions[(size_t)i * 3 + 0];
        const float ay = actions[(size_t)i * 3 + 1];
        const float az = actions[(size_t)i * 3 + 2];

        float p2x, p2y, p2z;
        if (MODE == TTL_MODE_F32) {
            // directions = normalize_vectors(actions) * step_size   (float32)
            const float s = sqrtf((ax * ax + ay * ay) + az * az);
            float dx = (ax / s) * P.step32;
            float dy = (ay / s) * P.step32;
            float dz = (az / s) * P.step32;
            if (L == 1) {
                // first step: a trial step that would stop is reversed
                const float tx = p1x + dx, ty = p1y + dy, tz = p1z + dz;
                const bool flip = first_step_flips(P, tx, ty, tz);
                dx = flip ? -dx : dx;
                dy = flip ? -dy : dy;
                dz = flip ? -dz : dz;
            }
            p2x = p1x + dx;
            p2y = p1y + dy;
            p2z = p1z + dz;
        } else {
            // float64 directions; the new point is float32(float64(p) + d)
            double dx, dy, dz;
            if (MODE == TTL_MODE_F32NORM) {
                // float32 normalise, then float32_array * np.float64 step
                // (plain TrackingEnvironment with a float64 affine, numpy >= 2)
                const float s = sqrtf((ax * ax + ay * ay) + az * az);
                dx = (double)(ax / s) * P.step64;
                dy = (double)(ay / s) * P.step64;
                dz = (double)(az / s) * P.step64;
            } else {
                // (actions + noise) in float64 -> normalise -> * step_size
                double a0 = (double)ax, a1 = (double)ay, a2 = (double)az;
                if (noise) {
                    a0 = a0 + noise[(size_t)i * 3 + 0];
                    a1 = a1 + noise[(size_t)i * 3 + 1];
                    a2 = a2 + noise[(size_t)i * 3 + 2];
                } else {
                    a0 = a0 + 0.0;
                    a1 = a1 + 0.0;
                    a2 = a2 + 0.0;
                }
                const double s = sqrt((a0 * a0 + a1 * a1) + a2 * a2);
                dx = (a0 / s) * P.step64;
                dy = (a1 / s) * P.step64;
                dz = (a2 / s) * P.step64;
            }
            if (L == 1) {
                const float tx = (float)((double)p1x + dx);
                const float ty = (float)((double)p1y + dy);
                const float tz = (float)((double)p1z + dz);
                const bool flip = first_step_flips(P, tx, ty, tz);
                dx = flip ? -dx : dx;
                dy = flip ? -dy : dy;
                dz = flip ? -dz : dz;
            }
            p2x = (float)((double)p1x + dx);
            p2y = (float)((double)p1y + dy);
            p2z = (float)((double)p1z + dz);
        }
        h[L * 3 + 0] = p2x;
        h[L * 3 + 1] = p2y;
        h[L * 3 + 2] = p2z;
        l2[0] = q1;
        l2[1] = float4{p2x, p2y, p2z, 0.0f};

        const int n_pts = L + 1;
        float ux, uy, uz, wx, wy, wz;
        const int bits = stopping_bits(P, p0x, p0y, p0z, p1x, p1y, p1z, p2x, p2y,
                                       p2z, n_pts, ux, uy, uz, wx, wy, wz);

        stop = bits != 0;
        if (stop) {
            P.flags[g] = bits;
            P.dones[g] = 1;
        }
        done_out[i] = stop ? 1 : 0;

        if (reward_out) {
            double rew = 0.0;
            if (P.compute_reward && P.align_w > 0.0f) {
                // peaks at int32(p[-2]) (truncation), clipped
                int vi = (int)p1x, vj = (int)p1y, vk = (int)p1z;
                vi = min(max(vi, 0), P.peaks_dim[0] - 1);
                vj = min(max(vj, 0), P.peaks_dim[1] - 1);
                vk = min(max(vk, 0), P.peaks_dim[2] - 1);
                const float *pk =
                    P.peaks +
                    (((size_t)vi * P.peaks_dim[1] + vj) * P.peaks_dim[2] + vk) * 15;
                const float u0 = nan_to_num(ux), u1 = nan_to_num(uy),
                            u2 = nan_to_num(uz);
                float best = 0.f;  // np.amax over the 5 peaks (NaN sticks)
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    float vx, vy, vz;
                    unit3(pk[3 * k], pk[3 * k + 1], pk[3 * k + 2], vx, vy, vz);
                    vx = nan_to_num(vx);
                    vy = nan_to_num(vy);
                    vz = nan_to_num(vz);
                    const float d = fabsf((vx * u0 + vy * u1) + vz * u2);
                    if (k == 0)
                        best = d;
                    else if (best == best && (d != d || d > best))
                        best = d;
                }
                float r32 = best;
                if (n_pts >= 3) {
                    const double f = ((double)u0 * (double)nan_to_num(wx) +
                                      (double)u1 * (double)nan_to_num(wy)) +
                                     (double)u2 * (double)nan_to_num(wz);
                    r32 = (float)((double)r32 * f);
                }
                rew = (double)(P.align_w * r32);
            }
            reward_out[i] = rew;
        }
        P.stop[i] = stop ? 1 : 0;
        // newest point, in row order: the state gather reads it without the
        // idx -> history indirection (one dependent memory round trip less)
        *reinterpret_cast<float4 *>(P.head + 4 * (size_t)i) =
            float4{p2x, p2y, p2z, __int_as_float(g)};
    }
    return stop;
}

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_advance(
    EnvParams P, const int *__restrict__ idx, const float *__restrict__ actions,
    const double *__restrict__ noise, int n_active, int L,
    double *__restrict__ reward_out, uint8_t *__restrict__ done_out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n_active;
    bool stop = false;
    if (active)
        stop = advance_row<MODE>(P, idx[i], i, actions, noise, L, reward_out, done_out);
    block_survivor_ranks(P, i, active, active && !stop);
}

// ---------------------------------------------------------------------------
// Free-running step (ttl_env_freerun_*): the number of active rows, the
// current length and which of the two continue_idx buffers is live come from
// device memory, so that a step is a fixed sequence of launches that can be
// captured in a HIP graph and replayed without the host in the loop.
// P.counts + TTL_FR_LIVE  = {n_active, length, cur, steps done}: read by
// k_advance_fr, rewritten by the step's last kernel (k_prefix_state_fr);
// P.counts + TTL_FR_SNAP  = the same four words as this step found them:
// written by k_advance_fr, read by k_prefix_state_fr.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_advance_fr(
    EnvParams P, const int *__restrict__ idx_a, const int *__restrict__ idx_b,
    const float *__restrict__ actions, int n_cap, double *__restrict__ reward_out,
    uint8_t *__restrict__ done_out) {
    const int *live = P.counts + TTL_FR_LIVE;
    int n_active = live[0];
    const int L = live[1], cur = live[2];
    // A launch that does not cover the active rows, or a full history (cannot
    // happen while rows are active: the LENGTH criterion has stopped them;
    // guards the history write): the step does nothing but count itself.
    const bool skip = L < 1 || L > P.max_nb_steps || n_active > n_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int *snap = P.counts + TTL_FR_SNAP;
        snap[0] = n_active;
        snap[1] = L;
        snap[2] = cur;
        snap[3] = live[3];
        snap[4] = skip ? 1 : 0;
    }
    if (skip) n_active = 0;
    const int *idx = cur ? idx_b : idx_a;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n_active;
    bool stop = false;
    if (active) {
        stop = advance_row<MODE>(P, idx[i], i, actions, nullptr, L, reward_out, done_out);
    } else if (i < n_cap) {      // rows that left the episode earlier
        done_out[i] = 1;
        if (reward_out) reward_out[i] = 0.0;
    }
    block_survivor_ranks(P, i, active, active && !stop);
}

__global__ void k_fr_init(EnvParams P, int n_active, int length, int cur) {
    int *live = P.counts + TTL_FR_LIVE;
    live[0] = n_active;
    live[1] = length;
    live[2] = cur;
    live[3] = 0;
}

// ---------------------------------------------------------------------------
// k_restop: OR externally computed stopping bits (the oracle criterion,
// stopping_criteria.py:85-154, evaluated by the host-side oracle between
// k_advance and k_prefix) into the step's decisions and redo the ranks.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_restop(EnvParams P,
                                                  const int *__restrict__ idx,
                                                  const uint8_t *__restrict__ extra,
                                                  int n_active,
                                                  uint8_t *__restrict__ done_out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = i < n_active;
    bool stop = false;
    if (active) {
        stop = P.stop[i] != 0;
        const int bits = extra[i];
        if (bits) {
            const int g = idx[i];
            P.flags[g] = (stop ? P.flags[g] : 0) | bits;
            P.dones[g] = 1;
            P.stop[i] = 1;
            done_out[i] = 1;
            stop = true;
        }
    }
    block_survivor_ranks(P, i, active, active && !stop);
}

// ---------------------------------------------------------------------------
// k_prefix: exclusive prefix over the per-block survivor counts (each block
// sums its predecessors; <= a few thousand ints from L2), then the stable
// scatter of continue_idx and the active-row -> state-row map.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_prefix(EnvParams P,
                                                  const int *__restrict__ idx,
                                                  int *__restrict__ idx_next,
                                                  const int *__restrict__ proc,
                                                  int n_active, int n_blocks,
                                                  int order, int n_pts,
                                                  int *__restrict__ host_word, int seq) {
    if (proc) {
        // first half of the processing-order compaction (slot order, same
        // grid): which slots survive, ranked inside their block
        const int j = blockIdx.x * BLOCK + threadIdx.x;
        const bool active = j < n_active;
        const bool keep = active && P.stop[proc[active ? j : 0]] == 0;
        block_ranks(P.proc_rank, P.proc_counts, j, active, keep);
    }
    __shared__ int red[2][BLOCK / 64];
    int before = 0, total = 0;
    for (int b = threadIdx.x; b < n_blocks; b += BLOCK) {
        const int c = P.block_counts[b];
        total += c;
        if (b < (int)blockIdx.x) before += c;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        before += __shfl_down(before, off);
        total += __shfl_down(total, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = before;
        red[1][wave] = total;
    }
    __syncthreads();
    before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        before += red[0][w];
        total += red[1][w];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.counts[0] = total;
        P.counts[1] = n_active - total;
        if (host_word) {
            // straight into the caller's pinned buffer, sequence number last:
            // the host polls it (ttl_env_wait_counts) and can queue the next
            // step while the state gather of this one is still running
            __hip_atomic_store(host_word + 0, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_word + 1, n_active - total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_word + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_active) return;
    const int pos = before + P.rank[i];
    const bool stop = P.stop[i] != 0;
    const int g = idx[i];
    if (!stop) idx_next[pos] = g;
    // ORDER_PARTITION has no separate harvest kernel: record the final length
    // of the streamlines that just stopped here (tracking_env.py:236)
    if (stop && order == TTL_ORDER_PARTITION) P.lengths[g] = n_pts;
    P.surv_pos[i] = stop ? -1 : pos;
    // the rows that stopped, compacted in row order (ttl_env_stopped: the
    // oracle reward scores exactly these, oracle_reward.py:78-90)
    if (stop) *reinterpret_cast<int2 *>(P.stop_list + 2 * (size_t)(i - pos)) = int2{i, g};
    int dest = i;
    if (order == TTL_ORDER_PARTITION) dest = stop ? total + (i - pos) : pos;
    P.row_dest[i] = dest;
    *reinterpret_cast<int2 *>(P.pos_dest + 2 * (size_t)i) = int2{stop ? -1 : pos, dest};
}

// ---------------------------------------------------------------------------
// Processing order of the state gather.  Row order (continue_idx order) is
// fixed by the reference, but WHICH rows a workgroup gathers is free: proc[j]
// lists the active rows sorted by the 8^3-voxel brick of their seed, so that a
// workgroup -- and its neighbours in time on the same XCD -- fetch voxels that
// are already in L2 instead of going to the Infinity Cache / HBM for each
// streamline separately.  Each step proc is compacted (stable, in proc order)
// and renumbered with the survivors' new row ids.
// ---------------------------------------------------------------------------
// low 6 bits of v spread to every third bit (Morton interleave helper)
__device__ __forceinline__ unsigned spread3(unsigned v) {
    v &= 63u;
    v = (v | (v << 8)) & 0x300Fu;
    v = (v | (v << 4)) & 0x30C3u;
    v = (v | (v << 2)) & 0x9249u;
    return v;
}

__global__ __launch_bounds__(BLOCK) void k_proc_scatter(EnvParams P,
                                                        const int *__restrict__ idx,
                                                        const int *__restrict__ proc,
                                                        int *__restrict__ proc_next,
                                                        int n_active, int n_blocks,
                                                        int local_sort) {
    __shared__ int red[BLOCK / 64];
    __shared__ unsigned s_key[BLOCK];
    __shared__ int s_pos[BLOCK];
    __shared__ int s_rank[BLOCK];
    int before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += BLOCK) before += P.proc_counts[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = before;
    __syncthreads();
    before = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) before += red[w];
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    const bool active = j < n_active;
    float4 hd = float4{0.f, 0.f, 0.f, 0.f};
    int2 pd = int2{-1, 0};
    if (active) {
        const int row = proc[j];
        pd = *reinterpret_cast<const int2 *>(P.pos_dest + 2 * (size_t)row);
        // two scattered loads per slot: the packed {surv_pos, row_dest} above
        // and the head record, whose .w already carries idx[row]
        hd = *reinterpret_cast<const float4 *>(P.head + 4 * (size_t)row);
    }
    if (!local_sort) {
        if (!active) return;
        if (pd.x >= 0) proc_next[before + P.proc_rank[j]] = pd.x;
        // everything this step's state gather needs to know about slot j, in
        // slot order: one thread per slot resolves the row indirections here,
        // so the gather (12 lanes per slot) starts from two coalesced loads
        *reinterpret_cast<float4 *>(P.slot_head + 4 * (size_t)j) = hd;
        P.slot_dest[j] = pd.y;
        return;
    }
    // Local re-sort: the 256 slots of this block are put in Morton order of
    // the voxel their streamline sits in now.  The global order (8^3 bricks,
    // rebuilt every few steps) decays slowly; the order INSIDE a brick decays
    // within two steps (a step is 0.75 voxel) and decides how many of a wave's
    // five streamlines share voxel records.  A block's slots stay the block's
    // (the kept-slot count per block is permutation invariant), so this is a
    // 256-key rank sort in LDS, no extra launch.
    unsigned key = 0xFFFFFFFFu;
    if (active) {
        const unsigned vx = (unsigned)(int)fminf(fmaxf(floorf(hd.x), 0.0f), 1023.0f);
        const unsigned vy = (unsigned)(int)fminf(fmaxf(floorf(hd.y), 0.0f), 1023.0f);
        const unsigned vz = (unsigned)(int)fminf(fmaxf(floorf(hd.z), 0.0f), 1023.0f);
        // bits above the low 6 per axis first (coarse), then the Morton code
        const unsigned coarse = (((vx >> 6) & 3u) << 4) | (((vy >> 6) & 3u) << 2) | ((vz >> 6) & 3u);
        const unsigned m = (spread3(vx) << 2) | (spread3(vy) << 1) | spread3(vz);
        // (the all-ones code -- voxel 255 / 511 / .. on every axis -- gives way by
        // one, so that no live key equals the idle threads' 0xFFFFFFFF)
        key = (min(coarse << 18 | m, 0xFFFFFEu) << 8) | threadIdx.x;   // unique inside the block
    }
    // Rank of this slot's key among the block's 256 keys (unique: the thread id
    // sits in the low byte; idle threads of a partly filled last block share
    // 0xFFFFFFFF and all rank behind the live ones).  Each wave sorts its 64 keys
    // in registers (bitonic network over cross-lane exchanges, 21 stages), the
    // four sorted runs go to LDS, and every key counts the smaller keys of the
    // other three runs by binary search: 21 exchanges + 21 LDS reads per thread
    // instead of 256 comparisons.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned sk = key;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned other = (unsigned)__shfl_xor((int)sk, stride);
            const bool up = (lane & size) == 0;          // this block of `size` lanes ascends
            const bool low = (lane & stride) == 0;       // the lower lane of the pair
            const unsigned mn = min(sk, other), mx = max(sk, other);
            sk = (up == low) ? mn : mx;
        }
    }
    s_key[threadIdx.x] = sk;                 // run `wave`, ascending over the lanes
    // sorted positions nobody ranks into (a partly filled last block: its idle
    // threads all rank to the same position) must read as "no survivor": LDS
    // keeps whatever an earlier workgroup left there
    s_pos[threadIdx.x] = -1;
    __syncthreads();
    // position of the sorted key `sk`: its lane + the smaller keys of the other runs
    int srank = lane;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        if (w == wave) continue;
        const unsigned *run = s_key + 64 * w;
        int c = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (run[c + step - 1] < sk) c += step;
        if (c == 63 && run[63] < sk) c = 64;
        srank += c;
    }
    // hand the position to the thread the key came from
    if (sk != 0xFFFFFFFFu) s_rank[sk & 255u] = srank;
    __syncthreads();
    int rank = 0;
    if (active) rank = s_rank[threadIdx.x];
    // sorted position `rank` of this block receives this slot's record
    if (active) {
        const size_t o = (size_t)blockIdx.x * BLOCK + rank;
        *reinterpret_cast<float4 *>(P.slot_head + 4 * o) = hd;
        P.slot_dest[o] = pd.y;
    }
    if (active) s_pos[rank] = pd.x;              // surv_pos (or -1) in sorted order
    __syncthreads();
    // next step's order: the survivors, in this sorted order
    const int pos = s_pos[threadIdx.x];
    const bool keep = pos >= 0;
    const unsigned long long mk = __ballot(keep);
    const int below = __popcll(mk & ((1ull << lane) - 1ull));
    if (lane == 0) red[wave] = __popcll(mk);
    __syncthreads();
    int wave_before = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w)
        if (w < wave) wave_before += red[w];
    if (keep) proc_next[before + wave_before + below] = pos;
}

// ---------------------------------------------------------------------------
// k_tail: k_prefix and k_proc_scatter in ONE launch (batches with a processing
// order of at most TTL_TAIL_FUSED_MAX_ROWS slots, default 262 144: 0-3 % of a step
// at 262 144 rows depending on the box, 3-4 % at 32 768-65 536 rows; at 1 048 576
// rows, where every workgroup scans 4 096 counts, it is 3 % SLOWER:
// profiles/r03_tail_fused_ab.log; TTL_TAIL_FUSED=0: never).  What kept them apart was the stable
// compaction of the processing order: a slot's new position needs the number
// of surviving slots in every earlier block -- a second grid-wide prefix, over
// slots, behind the first one over rows.  Here the order is NOT compacted
// between its periodic refreshes: it keeps the length it had at the last
// refresh (n_slots), a slot whose streamline has stopped becomes a hole (-1),
// and the per-block re-sort moves the holes to the end of their 256-slot block,
// where whole lane groups / waves of the gather exit at once.  Everything a
// slot needs -- the row's position among the survivors, its state row -- then
// follows from what k_advance left (stop, in-block rank, per-block counts): every
// workgroup scans ALL block counts itself (<= 4096 ints through LDS), no
// workgroup waits for another.
// ---------------------------------------------------------------------------
constexpr int TTL_TAIL_MAX_BLOCKS = 4096;

__global__ __launch_bounds__(BLOCK) void k_tail(
    EnvParams P, const int *__restrict__ idx, int *__restrict__ idx_next,
    const int *__restrict__ proc, int *__restrict__ proc_next, int n_active, int n_slots,
    int nb_rows, int order, int n_pts, int *__restrict__ host_word, int seq, int local_sort) {
    __shared__ int s_scan[TTL_TAIL_MAX_BLOCKS];
    __shared__ int s_wave[BLOCK / 64];
    __shared__ unsigned s_key[BLOCK];
    __shared__ int s_pos[BLOCK];
    __shared__ int s_rank[BLOCK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- exclusive scan of the per-block survivor counts (rows) ----
    const int per = (nb_rows + BLOCK - 1) / BLOCK;          // <= 16 counts per thread
    const int lo = tid * per;
    int sum = 0;
    for (int k = 0; k < per; ++k)
        if (lo + k < nb_rows) sum += P.block_counts[lo + k];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int run = incl - sum, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        if (w < wave) run += s_wave[w];
        total += s_wave[w];
    }
    for (int k = 0; k < per; ++k)
        if (lo + k < nb_rows) {
            s_scan[lo + k] = run;
            run += P.block_counts[lo + k];
        }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        P.counts[0] = total;
        P.counts[1] = n_active - total;
        if (host_word) {       // see k_prefix
            __hip_atomic_store(host_word + 0, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_word + 1, n_active - total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_word + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- rows: continue_idx of the next step, lengths, the row maps ----
    const int i = blockIdx.x * BLOCK + tid;
    if (i < n_active) {
        const int pos = s_scan[blockIdx.x] + P.rank[i];
        const bool stop = P.stop[i] != 0;
        const int g = idx[i];
        if (!stop) idx_next[pos] = g;
        if (stop && order == TTL_ORDER_PARTITION) P.lengths[g] = n_pts;
        P.surv_pos[i] = stop ? -1 : pos;
        if (stop) *reinterpret_cast<int2 *>(P.stop_list + 2 * (size_t)(i - pos)) = int2{i, g};
        int dest = i;
        if (order == TTL_ORDER_PARTITION) dest = stop ? total + (i - pos) : pos;
        P.row_dest[i] = dest;
    }
    // ---- slots: this step's records for the gather, next step's order ----
    const int j = i;
    const int row = j < n_slots ? proc[j] : -1;
    const bool live = row >= 0;
    float4 hd = float4{0.f, 0.f, 0.f, 0.f};
    int dest = -1, next = -1;
    if (live) {
        const bool stop = P.stop[row] != 0;
        const int pos = s_scan[row / BLOCK] + P.rank[row];
        dest = row;
        if (order == TTL_ORDER_PARTITION) dest = stop ? total + (row - pos) : pos;
        next = stop ? -1 : pos;
        hd = *reinterpret_cast<const float4 *>(P.head + 4 * (size_t)row);
    }
    if (!local_sort) {         // slots keep their places, holes included
        if (j < n_slots) {
            *reinterpret_cast<float4 *>(P.slot_head + 4 * (size_t)j) = hd;
            P.slot_dest[j] = dest;
            proc_next[j] = next;
        }
        return;
    }
    // per-block re-sort by the voxel the streamline sits in now (see
    // k_proc_scatter); holes carry the largest key and end up behind the live slots
    unsigned key = 0xFFFFFFFFu;
    if (live) {
        const unsigned vx = (unsigned)(int)fminf(fmaxf(floorf(hd.x), 0.0f), 1023.0f);
        const unsigned vy = (unsigned)(int)fminf(fmaxf(floorf(hd.y), 0.0f), 1023.0f);
        const unsigned vz = (unsigned)(int)fminf(fmaxf(floorf(hd.z), 0.0f), 1023.0f);
        const unsigned coarse = (((vx >> 6) & 3u) << 4) | (((vy >> 6) & 3u) << 2) | ((vz >> 6) & 3u);
        const unsigned m = (spread3(vx) << 2) | (spread3(vy) << 1) | spread3(vz);
        key = (min(coarse << 18 | m, 0xFFFFFEu) << 8) | (unsigned)tid;
    }
    unsigned sk = key;
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const unsigned other = (unsigned)__shfl_xor((int)sk, stride);
            const bool up = (lane & size) == 0;
            const bool low = (lane & stride) == 0;
            const unsigned mn = min(sk, other), mx = max(sk, other);
            sk = (up == low) ? mn : mx;
        }
    }
    s_key[tid] = sk;
    s_pos[tid] = -1;
    const int n_live = __syncthreads_count(live);
    int srank = lane;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        if (w == wave) continue;
        const unsigned *srun = s_key + 64 * w;
        int c = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (srun[c + step - 1] < sk) c += step;
        if (c == 63 && srun[63] < sk) c = 64;
        srank += c;
    }
    if (sk != 0xFFFFFFFFu) s_rank[sk & 255u] = srank;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * BLOCK;
    if (live) {
        const int rank = s_rank[tid];
        *reinterpret_cast<float4 *>(P.slot_head + 4 * (base + rank)) = hd;
        P.slot_dest[base + rank] = dest;
        s_pos[rank] = next;
    }
    __syncthreads();
    if (j < n_slots) {
        if (tid >= n_live) P.slot_dest[j] = -1;        // the holes, behind the live slots
        proc_next[j] = s_pos[tid];
    }
}

// stopping flags of caller-supplied tails (n_pts points per streamline)
__global__ __launch_bounds__(BLOCK) void k_probe_flags(
    EnvParams P, const float *__restrict__ tail, int n, int n_pts,
    uint8_t *__restrict__ out) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *t = tail + (size_t)i * 9;
    float ux, uy, uz, wx, wy, wz;
    int bits;
    if (n_pts >= 2) {
        bits = stopping_bits(P, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7],
                             t[8], n_pts, ux, uy, uz, wx, wy, wz);
    } else {  // a single point: no segment, only LENGTH and MASK can fire
        bits = (n_pts >= P.max_nb_steps) ? TTL_FLAG_LENGTH : 0;
        if (outside_mask(P, t[6], t[7], t[8])) bits |= TTL_FLAG_MASK;
    }
    out[i] = (uint8_t)bits;
}

// lengths[stopping_idx] = length
__global__ __launch_bounds__(BLOCK) void k_finish(EnvParams P,
                                                  const int *__restrict__ idx,
                                                  int n_active, int n_pts) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_active) return;
    if (P.stop[i]) P.lengths[idx[i]] = n_pts;
}

// state[continue_idx]: copy the survivors' rows to their compacted position
__global__ __launch_bounds__(BLOCK) void k_copy_rows(
    EnvParams P, int n_active, int width, const float *__restrict__ in,
    float *__restrict__ out, long long pitch) {
    constexpr int LPS = 64;
    const int row = blockIdx.x * (BLOCK / LPS) + threadIdx.x / LPS;
    const int sub = threadIdx.x % LPS;
    if (row >= n_active) return;
    const int pos = P.surv_pos[row];
    if (pos < 0) return;
    const float *src = in + (size_t)row * (size_t)pitch;
    float *dst = out + (size_t)pos * (size_t)pitch;
    for (int f = sub; f < width; f += LPS) dst[f] = src[f];
}

// ragged pack of the tracked streamlines (tracking_env.py:263-284: the first
// keep[i] points of streamline i, one after the other): one wave per streamline
__global__ __launch_bounds__(BLOCK) void k_pack_streamlines(
    const float *__restrict__ hist, long long row_pitch, const long long *__restrict__ keep,
    const long long *__restrict__ offsets, int n, float *__restrict__ out) {
    const int i = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    const int floats = (int)keep[i] * 3;
    const float *src = hist + (size_t)i * (size_t)row_pitch;
    float *dst = out + (size_t)offsets[i] * 3;
    for (int f = lane; f < floats; f += 64) dst[f] = src[f];
}

__global__ __launch_bounds__(BLOCK) void k_reset(EnvParams P, int *idx,
                                                 const float *__restrict__ seeds,
                                                 int n) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float *h = P.hist + (size_t)i * (size_t)(P.max_nb_steps + 1) * 3;
    h[0] = seeds[(size_t)i * 3 + 0];
    h[1] = seeds[(size_t)i * 3 + 1];
    h[2] = seeds[(size_t)i * 3 + 2];
    float4 *l2 = reinterpret_cast<float4 *>(P.last2 + 8 * (size_t)i);
    l2[0] = float4{0.0f, 0.0f, 0.0f, 0.0f};
    l2[1] = float4{h[0], h[1], h[2], 0.0f};
    P.flags[i] = 0;
    P.lengths[i] = 1;
    P.dones[i] = 0;
    idx[i] = i;
}

__global__ __launch_bounds__(BLOCK) void k_pack_sh(const float *__restrict__ src,
                                                   float *__restrict__ dst, int X, int Y,
                                                   int Z, int C, int pitch, int brick,
                                                   unsigned sx, unsigned sy) {
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long total = (long long)X * Y * Z * pitch;
    if (t >= total) return;
    const long long v = t / pitch;
    const int c = (int)(t - v * pitch);
    const int z = (int)(v % Z), y = (int)((v / Z) % Y), x = (int)(v / ((long long)Z * Y));
    size_t rec;
    if (brick)
        rec = (size_t)(x >> 2) * sx + (size_t)(x & 3) * 16 + (size_t)(y >> 2) * sy +
              (size_t)(y & 3) * 4 + (size_t)(z >> 2) * 64 + (size_t)(z & 3);
    else
        rec = (size_t)v;
    dst[rec * pitch + c] = (c < C) ? src[v * C + c] : 0.0f;
}

// ---------------------------------------------------------------------------
// Scripted, policy-free actions for "env.step only" measurements and tests
// (SURVEY 8d): step 0 -> a random direction; later steps -> unit(previous
// segment, read from the state row) + wobble * noise.  The noise is a
// counter-based hash of (seed, step, global streamline id, component), so the
// CPU twin (oracle/scripted_policy.py) produces the same bits regardless of
// compaction.  Noise = centred sum of 4 uniforms * sqrt(3) (unit variance).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ float scripted_noise(unsigned seed, unsigned step,
                                                unsigned gid, unsigned comp) {
    const unsigned base = mix32(seed * 0x9E3779B1u + step) ^
                          mix32(gid * 0x27D4EB2Fu + comp * 0x165667B1u + 0x1234567u);
    float acc = 0.0f;
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
        const unsigned h = mix32(base + k * 0x9E3779B9u);
        acc = acc + (float)(h >> 8) * 5.9604644775390625e-08f;  // 2^-24, exact
    }
    return (acc - 2.0f) * 1.7320508075688772f;
}

__global__ __launch_bounds__(BLOCK) void k_scripted_actions(
    const float *__restrict__ state, long long pitch, int dir_offset,
    const int *__restrict__ idx, int n, unsigned seed, unsigned step,
    float wobble, float *__restrict__ actions) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned gid = (unsigned)idx[i];
    const float n0 = scripted_noise(seed, step, gid, 0);
    const float n1 = scripted_noise(seed, step, gid, 1);
    const float n2 = scripted_noise(seed, step, gid, 2);
    float a0 = n0, a1 = n1, a2 = n2;
    if (step > 0) {
        const float *row = state + (size_t)i * (size_t)pitch + dir_offset;
        const float px = row[0], py = row[1], pz = row[2];
        float s = sqrtf((px * px + py * py) + pz * pz);
        if (!(s > 0.0f)) s = 1.0f;
        a0 = px / s + wobble * n0;
        a1 = py / s + wobble * n1;
        a2 = pz / s + wobble * n2;
    }
    actions[(size_t)i * 3 + 0] = a0;
    actions[(size_t)i * 3 + 1] = a1;
    actions[(size_t)i * 3 + 2] = a2;
}

// the same policy for a free-running step: row count, step number and the live
// continue_idx buffer from the device words (see k_advance_fr)
__global__ __launch_bounds__(BLOCK) void k_scripted_actions_fr(
    EnvParams P, const float *__restrict__ state, long long pitch, int dir_offset,
    const int *__restrict__ idx_a, const int *__restrict__ idx_b, int n_rows,
    unsigned seed, float wobble, float *__restrict__ actions) {
    const int *live = P.counts + TTL_FR_LIVE;
    const int n = min(live[0], n_rows);
    const unsigned step = (unsigned)(live[1] - 1);
    const int *idx = live[2] ? idx_b : idx_a;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned gid = (unsigned)idx[i];
    const float n0 = scripted_noise(seed, step, gid, 0);
    const float n1 = scripted_noise(seed, step, gid, 1);
    const float n2 = scripted_noise(seed, step, gid, 2);
    float a0 = n0, a1 = n1, a2 = n2;
    if (step > 0) {
        const float *row = state + (size_t)i * (size_t)pitch + dir_offset;
        const float px = row[0], py = row[1], pz = row[2];
        float s = sqrtf((px * px + py * py) + pz * pz);
        if (!(s > 0.0f)) s = 1.0f;
        a0 = px / s + wobble * n0;
        a1 = py / s + wobble * n1;
        a2 = pz / s + wobble * n2;
    }
    actions[(size_t)i * 3 + 0] = a0;
    actions[(size_t)i * 3 + 1] = a1;
    actions[(size_t)i * 3 + 2] = a2;
}

// ---------------------------------------------------------------------------
// k_mask_classes: for every cell (the integer part of a sample coordinate)
// the min / max of the 64 mirror-folded coefficient taps a sample in that
// cell would read.  1 = every sample there is >= thr (never stops), 2 = every
// sample is < thr (always stops), 0 = evaluate the spline.  The margin covers
// the rounding of the float64 evaluation (weights sum to 1 within a few ulp,
// 64 products and additions: error << 1e-12 * max|c|).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_mask_classes(
    const double *__restrict__ coef, int nx, int ny, int nz, double thr,
    uint8_t *__restrict__ cls) {
    const long long t = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const long long total = (long long)nx * ny * nz;
    if (t >= total) return;
    const int fz = (int)(t % nz), fy = (int)((t / nz) % ny), fx = (int)(t / ((long long)nz * ny));
    double lo = INFINITY, hi = -INFINITY;
    for (int a = 0; a < 4; ++a) {
        const int ia = mirror_fold(fx - 1 + a, nx);
        for (int b = 0; b < 4; ++b) {
            const int ib = mirror_fold(fy - 1 + b, ny);
            const double *line = coef + ((size_t)ia * ny + ib) * nz;
            for (int d = 0; d < 4; ++d) {
                const double v = line[mirror_fold(fz - 1 + d, nz)];
                lo = fmin(lo, v);
                hi = fmax(hi, v);
            }
        }
    }
    const double margin = 1e-9 * (1.0 + fabs(thr) + fmax(fabs(lo), fabs(hi)));
    uint8_t c = 0;
    if (lo >= thr + margin) c = 1;
    else if (hi < thr - margin) c = 2;
    if (!(lo == lo) || !(hi == hi)) c = 0;   // NaN coefficients: evaluate
    cls[t] = c;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct ttl_env {
    ttl_env_desc d;
    EnvParams P;
    int length;      // points per active streamline (reference: self.length)
    int n_active;    // rows of the current continue_idx
    int cur;         // 0: idx_a is continue_idx, 1: idx_b
    int stepped;     // 0: idle, 2: step begun (advanced), 1: step done, awaiting harvest
    int last_order;
    int last_n;      // n_active of the pending step
    uint8_t *last_done;  // done_out of the pending step (k_restop updates it)
    int *proc[2];        // processing order of the state gather, double buffered
    char *order_ws;      // scratch of the in-library order refresh (ttl_order.hip)
    size_t order_ws_bytes;
    int proc_cur;        // which proc buffer is current
    int use_proc;        // a processing order was installed for this episode
    int n_slots;         // length of the processing order (>= n_active: with the fused tail it keeps
                         // the length of its last refresh, stopped streamlines leave holes)
    int tail_fused;      // k_tail instead of k_prefix + k_proc_scatter (TTL_TAIL_FUSED) ...
    int tail_fused_max;  // ... for processing orders of at most this many slots (TTL_TAIL_FUSED_MAX_ROWS)
    // optional per-kernel timing with HIP events on the caller's stream
    int state_kernel; // 0: k_state (all 56 corner fetches), 3: k_state_dd with scalar tail stores, else k_state_dd
    hipStream_t side;      // carries the early device->host copy of the counts
    hipEvent_t ev_prefix;  // main stream: k_prefix done (counts are final)
    hipEvent_t ev_counts;  // side stream: counts have landed in host memory
    int counts_pending;    // 1: an early copy is in flight / unread, 2: the step's own
                           // kernel writes the caller's pinned buffer, host polls
    const int32_t *host_counts;  // where the counts land (caller's pinned memory)
    const int32_t *host_probe;   // pinned buffer whose device address is cached below
    int *host_dev;         // device address of host_probe, null if it has none
    int poll_seq;          // sequence number the polled word must reach
    hipStream_t poll_stream;
    int fuse_small;        // batches <= 16384 rows: one launch for prefix + gather
    int poll_counts;       // counts written by the kernel into the pinned buffer (TTL_POLL_COUNTS)
    int local_sort;        // k_proc_scatter re-sorts each block's slots by current voxel
    int n_exact;           // n_active is the exact survivor count (read back)
    int fr_cap;            // > 0: free-running steps are being enqueued for this many rows
    int *fr_host_word;     // device address of the pinned words a free-running step reports to
    int prof_on;
    int prof_mask;    // bit k: time kernel class k
    int prof_cap;     // event pairs available per kernel class
    int prof_n[TTL_PROFILE_CLASSES];    // launches recorded: advance, prefix, state, proc_scatter
    hipEvent_t *prof_ev[TTL_PROFILE_CLASSES];  // [2 * prof_cap] start/stop pairs
};

// Measurement support (profiles/pmc_summary.py): with TTL_GATHER_ROWS_LOG=<file>
// every launch of the state gather appends the number of state rows it really
// writes -- with the fused tail the gather's grid covers the slots of an
// uncompacted processing order, more than the rows -- so that counter bytes can
// be divided by units without reading them off the grid size.
static void log_gather_rows(int n_rows) {
    static FILE *f = []() -> FILE * {
        const char *path = getenv("TTL_GATHER_ROWS_LOG");
        return path && *path ? fopen(path, "a") : nullptr;
    }();
    if (f) {
        fprintf(f, "%d\n", n_rows);
        fflush(f);
    }
}

static void prof_mark(ttl_env *e, int which, int stop, hipStream_t s) {
    if (!e->prof_on || !((e->prof_mask >> which) & 1) ||
        e->prof_n[which] >= e->prof_cap)
        return;
    (void)hipEventRecord(e->prof_ev[which][2 * e->prof_n[which] + stop], s);
    if (stop) e->prof_n[which]++;
}

extern "C" {

const char *ttl_last_error(void) { return g_err; }
uint32_t ttl_abi_version(void) { return TTL_ABI_VERSION; }
size_t ttl_env_desc_size(void) { return sizeof(ttl_env_desc); }

size_t ttl_env_workspace_bytes(int32_t n_max) {
    if (n_max < 0) return 0;
    const size_t n = (size_t)n_max;
    const size_t nb = (n + BLOCK - 1) / BLOCK + 1;
    size_t b = 0;
    b += align_up(n, 256);                    // stop
    b += 6 * align_up(n * sizeof(int), 256);  // rank, surv_pos, row_dest, proc_rank, proc x2
    b += 2 * align_up(n * 4 * sizeof(float), 256); // head, slot_head
    b += align_up(n * 8 * sizeof(float), 256); // last2
    b += 2 * align_up(n * 2 * sizeof(int), 256); // pos_dest, stop_list
    b += align_up(n * sizeof(int), 256);      // slot_dest
    b += 2 * align_up(nb * sizeof(int), 256); // block_counts, proc_counts
    b += 256;                                 // counts
    b += ttl_detail_order_workspace_bytes(n); // order refresh scratch
    return b;
}

int64_t ttl_sh_volume_records(const int32_t *dim, int32_t layout) {
    if (!dim || dim[0] < 1 || dim[1] < 1 || dim[2] < 1) return 0;
    if (layout == TTL_SH_BRICK4)
        return (int64_t)((dim[0] + 3) / 4) * ((dim[1] + 3) / 4) * ((dim[2] + 3) / 4) * 64;
    return (int64_t)dim[0] * dim[1] * dim[2];
}

// Device memory for a gathered volume: on request physically contiguous when
// the driver can give that, ordinary hipMalloc otherwise.
// allocations made through the virtual-memory API (try_contiguous == 2): what
// ttl_volume_free needs to undo them
struct VmmBlock {
    hipMemGenericAllocationHandle_t handle;
    size_t size;
};
static std::mutex g_vmm_mutex;
static std::map<void *, VmmBlock> g_vmm_blocks;

static hipError_t vmm_alloc(int device, size_t bytes, void **out) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop,
                                                  hipMemAllocationGranularityRecommended);
    if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorInvalidValue;
    const size_t size = (bytes + gran - 1) / gran * gran;
    VmmBlock b{};
    b.size = size;
    if ((e = hipMemCreate(&b.handle, size, &prop, 0)) != hipSuccess) return e;
    void *va = nullptr;
    if ((e = hipMemAddressReserve(&va, size, gran, nullptr, 0)) != hipSuccess) {
        (void)hipMemRelease(b.handle);
        return e;
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemMap(va, size, 0, b.handle, 0)) != hipSuccess ||
        (e = hipMemSetAccess(va, size, &acc, 1)) != hipSuccess) {
        (void)hipMemUnmap(va, size);
        (void)hipMemAddressFree(va, size);
        (void)hipMemRelease(b.handle);
        return e;
    }
    std::lock_guard<std::mutex> lock(g_vmm_mutex);
    g_vmm_blocks[va] = b;
    *out = va;
    return hipSuccess;
}

int ttl_volume_alloc(int32_t device, size_t bytes, int32_t try_contiguous, void **out,
                     int32_t *contiguous_out) {
    if (!out || bytes == 0) return fail(TTL_ERR_INVALID, "ttl_volume_alloc: bad arguments");
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    if (device >= 0 && device != prev) HIP_TRY(hipSetDevice(device));
    void *p = nullptr;
    int contiguous = 0;
    if (try_contiguous == 2) {          // experiment: the virtual-memory API
        if (vmm_alloc(device >= 0 ? device : prev, bytes, &p) == hipSuccess && p)
            contiguous = 2;
        else {
            (void)hipGetLastError();
            p = nullptr;
        }
    } else if (try_contiguous) {
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) == hipSuccess && p)
            contiguous = 1;
        else {
            (void)hipGetLastError();
            p = nullptr;
        }
    }
    hipError_t e = hipSuccess;
    if (!p) e = hipMalloc(&p, bytes);
    if (device >= 0 && device != prev) (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        // callers probe for room with this call (BaseEnv._tune_placement): the
        // runtime keeps the last error until it is read, and the next launch
        // check -- ours or torch's -- would report this one as its own
        (void)hipGetLastError();
        return fail(TTL_ERR_HIP, "ttl_volume_alloc: %s", hipGetErrorString(e));
    }
    *out = p;
    if (contiguous_out) *contiguous_out = contiguous;
    return TTL_OK;
}

int ttl_volume_free(void *ptr) {
    if (!ptr) return TTL_OK;
    {
        std::unique_lock<std::mutex> lock(g_vmm_mutex);
        auto it = g_vmm_blocks.find(ptr);
        if (it != g_vmm_blocks.end()) {
            const VmmBlock b = it->second;
            g_vmm_blocks.erase(it);
            lock.unlock();
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemUnmap(ptr, b.size));
            HIP_TRY(hipMemAddressFree(ptr, b.size));
            HIP_TRY(hipMemRelease(b.handle));
            return TTL_OK;
        }
    }
    HIP_TRY(hipFree(ptr));
    return TTL_OK;
}

int ttl_pack_sh_volume(const float *src, float *dst, const int32_t *dim, int32_t n_coef,
                       int32_t coef_pitch, int32_t layout, void *hip_stream) {
    if (!src || !dst || !dim || dim[0] < 1 || dim[1] < 1 || dim[2] < 1 || n_coef <= 0 ||
        coef_pitch < n_coef || (coef_pitch & 3) ||
        (layout != TTL_SH_LINEAR && layout != TTL_SH_BRICK4))
        return fail(TTL_ERR_INVALID, "ttl_pack_sh_volume: bad arguments");
    if (((uintptr_t)dst) & 15)
        return fail(TTL_ERR_INVALID, "ttl_pack_sh_volume: dst must be 16B aligned");
    const long long total = (long long)dim[0] * dim[1] * dim[2] * coef_pitch;
    const long long blocks = (total + BLOCK - 1) / BLOCK;
    if (blocks > 0x7fffffffLL)
        return fail(TTL_ERR_INVALID, "ttl_pack_sh_volume: volume too large");
    hipStream_t s = (hipStream_t)hip_stream;
    const int brick = layout == TTL_SH_BRICK4;
    unsigned sx = 0, sy = 0;
    if (brick) {
        // padding records (dims rounded up to 4) are never written below
        HIP_TRY(hipMemsetAsync(dst, 0, (size_t)ttl_sh_volume_records(dim, layout) *
                                           coef_pitch * sizeof(float), s));
        sy = (unsigned)((dim[2] + 3) / 4) * 64u;
        sx = (unsigned)((dim[1] + 3) / 4) * sy;
    }
    hipLaunchKernelGGL(k_pack_sh, dim3((unsigned)blocks), dim3(BLOCK), 0, s, src, dst, dim[0],
                       dim[1], dim[2], n_coef, coef_pitch, brick, sx, sy);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_mask_classes(const double *mask_coef, const int32_t *dim, double threshold,
                     uint8_t *classes_out, void *hip_stream) {
    if (!mask_coef || !dim || !classes_out || dim[0] < 1 || dim[1] < 1 || dim[2] < 1)
        return fail(TTL_ERR_INVALID, "ttl_mask_classes: bad arguments");
    const long long total = (long long)dim[0] * dim[1] * dim[2];
    hipLaunchKernelGGL(k_mask_classes, dim3((unsigned)((total + BLOCK - 1) / BLOCK)),
                       dim3(BLOCK), 0, (hipStream_t)hip_stream, mask_coef, dim[0],
                       dim[1], dim[2], threshold, classes_out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_env_create(const ttl_env_desc *desc, ttl_env **out) {
    if (!desc || !out) return fail(TTL_ERR_INVALID, "ttl_env_create: null argument");
    const ttl_env_desc &d = *desc;
    if (d.abi_version != TTL_ABI_VERSION)
        return fail(TTL_ERR_INVALID, "ttl_env_create: abi_version %u != %u",
                    d.abi_version, TTL_ABI_VERSION);
    if (d.mode != TTL_MODE_F32 && d.mode != TTL_MODE_F64DIR &&
        d.mode != TTL_MODE_F32NORM)
        return fail(TTL_ERR_INVALID, "ttl_env_create: bad mode %d", d.mode);
    for (int a = 0; a < 3; ++a) {
        if (d.sh_dim[a] <= 0 || d.mask_dim[a] <= 0)
            return fail(TTL_ERR_INVALID, "ttl_env_create: non-positive volume dim");
        if (d.compute_reward && d.peaks_dim[a] <= 0)
            return fail(TTL_ERR_INVALID, "ttl_env_create: non-positive peaks dim");
    }
    if (d.sh_layout != TTL_SH_LINEAR && d.sh_layout != TTL_SH_BRICK4)
        return fail(TTL_ERR_INVALID, "ttl_env_create: bad sh_layout %d", d.sh_layout);
    if (d.n_coef <= 0 || d.coef_pitch < d.n_coef || (d.coef_pitch & 3))
        return fail(TTL_ERR_INVALID, "ttl_env_create: coef_pitch must be a multiple of 4 >= n_coef");
    if (!d.sh_packed || (((uintptr_t)d.sh_packed) & 15))
        return fail(TTL_ERR_INVALID, "ttl_env_create: sh_packed null or not 16B aligned");
    if (!d.mask_coef || (((uintptr_t)d.mask_coef) & 7))
        return fail(TTL_ERR_INVALID, "ttl_env_create: mask_coef null or misaligned");
    if (d.compute_reward && !d.peaks)
        return fail(TTL_ERR_INVALID, "ttl_env_create: compute_reward without peaks");
    if (d.n_dirs < 0 || d.max_nb_steps < 1 || d.n_max < 1)
        return fail(TTL_ERR_INVALID, "ttl_env_create: bad n_dirs/max_nb_steps/n_max");
    if (!(d.step_size_vox > 0.0))
        return fail(TTL_ERR_INVALID, "ttl_env_create: step_size_vox must be > 0");
    if (!d.streamlines || !d.flags || !d.lengths || !d.dones || !d.idx_a || !d.idx_b)
        return fail(TTL_ERR_INVALID, "ttl_env_create: null state buffer");
    if (!d.workspace || d.workspace_bytes < ttl_env_workspace_bytes(d.n_max) ||
        (((uintptr_t)d.workspace) & 255))
        return fail(TTL_ERR_INVALID, "ttl_env_create: workspace too small or not 256B aligned");

    ttl_env *e = new (std::nothrow) ttl_env;
    if (!e) return fail(TTL_ERR_INVALID, "ttl_env_create: out of host memory");
    e->d = d;
    EnvParams &P = e->P;
    memset(&P, 0, sizeof(P));
    P.mode = d.mode;
    for (int a = 0; a < 3; ++a) {
        P.sh_dim[a] = d.sh_dim[a];
        P.mask_dim[a] = d.mask_dim[a];
        P.peaks_dim[a] = d.peaks_dim[a];
    }
    P.n_coef = d.n_coef;
    P.coef_pitch = d.coef_pitch;
    P.sh = d.sh_packed;
    P.sh_shift = d.sh_coord_shift;
    P.sh_brick = d.sh_layout == TTL_SH_BRICK4;
    if (P.sh_brick) {
        P.sh_sy = (unsigned)((d.sh_dim[2] + 3) / 4) * 64u;
        P.sh_sx = (unsigned)((d.sh_dim[1] + 3) / 4) * P.sh_sy;
    } else {
        P.sh_sy = (unsigned)d.sh_dim[2];
        P.sh_sx = (unsigned)d.sh_dim[1] * (unsigned)d.sh_dim[2];
    }
    P.mask_coef = d.mask_coef;
    P.mask_cls = d.mask_classes;
    P.mask_thr = d.mask_threshold;
    P.peaks = d.peaks;
    P.compute_reward = d.compute_reward;
    P.align_w = (float)d.alignment_weighting;
    P.n_dirs = d.n_dirs;
    P.max_nb_steps = d.max_nb_steps;
    P.step64 = d.step_size_vox;
    P.step32 = (float)d.step_size_vox;
    P.radius = d.neigh_radius_vox;
    P.curv_enabled = d.curvature_enabled;
    P.curv_dot_max = d.curv_dot_max;
    P.hist = d.streamlines;
    P.flags = d.flags;
    P.lengths = d.lengths;
    P.dones = d.dones;
    const size_t n = (size_t)d.n_max;
    const size_t nb = (n + BLOCK - 1) / BLOCK + 1;
    char *w = (char *)d.workspace;
    P.stop = (uint8_t *)w;        w += align_up(n, 256);
    P.rank = (int *)w;            w += align_up(n * sizeof(int), 256);
    P.surv_pos = (int *)w;        w += align_up(n * sizeof(int), 256);
    P.row_dest = (int *)w;        w += align_up(n * sizeof(int), 256);
    P.block_counts = (int *)w;    w += align_up(nb * sizeof(int), 256);
    P.proc_rank = (int *)w;       w += align_up(n * sizeof(int), 256);
    P.proc_counts = (int *)w;     w += align_up(nb * sizeof(int), 256);
    e->proc[0] = (int *)w;        w += align_up(n * sizeof(int), 256);
    e->proc[1] = (int *)w;        w += align_up(n * sizeof(int), 256);
    P.head = (float *)w;          w += align_up(n * 4 * sizeof(float), 256);
    P.slot_head = (float *)w;     w += align_up(n * 4 * sizeof(float), 256);
    P.last2 = (float *)w;         w += align_up(n * 8 * sizeof(float), 256);
    P.pos_dest = (int *)w;        w += align_up(n * 2 * sizeof(int), 256);
    P.stop_list = (int *)w;       w += align_up(n * 2 * sizeof(int), 256);
    P.slot_dest = (int *)w;       w += align_up(n * sizeof(int), 256);
    P.counts = (int *)w;          w += 256;
    e->order_ws = w;
    e->order_ws_bytes = ttl_detail_order_workspace_bytes(n);
    e->length = 0;
    e->n_active = 0;
    e->cur = 0;
    e->stepped = 0;
    e->last_order = TTL_ORDER_ACTIVE;
    e->last_n = 0;
    e->last_done = nullptr;
    e->proc_cur = 0;
    e->use_proc = 0;
    e->state_kernel = 4;
    if (const char *v = getenv("TTL_STATE_KERNEL")) e->state_kernel = atoi(v);
    P.slot_rec = e->state_kernel != 2;
    P.xcd_remap = 1;
    if (const char *v = getenv("TTL_XCD_REMAP")) P.xcd_remap = atoi(v);
    P.xcd_rot = 0;
    if (const char *v = getenv("TTL_XCD_ROTATE")) P.xcd_rot = atoi(v) & 7;
    P.store_flavour = 0;
    if (const char *v = getenv("TTL_STORE_FLAVOUR")) P.store_flavour = atoi(v);
    e->side = nullptr;
    e->ev_prefix = nullptr;
    e->ev_counts = nullptr;
    e->counts_pending = 0;
    e->host_counts = nullptr;
    e->host_probe = nullptr;
    e->host_dev = nullptr;
    e->poll_seq = 0;
    e->poll_stream = nullptr;
    e->fuse_small = 1;
    if (const char *v = getenv("TTL_FUSE_SMALL")) e->fuse_small = atoi(v);
    P.fuse_max_rows = 64 * BLOCK;
    if (const char *v = getenv("TTL_FUSE_MAX_ROWS")) {
        const int rows = atoi(v);
        P.fuse_max_rows = rows < BLOCK ? BLOCK : rows > TTL_FUSE_MAX_BLOCKS * BLOCK
                                                     ? TTL_FUSE_MAX_BLOCKS * BLOCK : rows;
    }
    e->local_sort = 1;
    P.persist_rows = 0;
    if (const char *v = getenv("TTL_GATHER_PERSIST_ROWS")) P.persist_rows = atoi(v);
    e->tail_fused = 1;
    if (const char *v = getenv("TTL_TAIL_FUSED")) e->tail_fused = atoi(v);
    e->tail_fused_max = 262144;
    if (const char *v = getenv("TTL_TAIL_FUSED_MAX_ROWS")) e->tail_fused_max = atoi(v);
    if (e->tail_fused_max > TTL_TAIL_MAX_BLOCKS * BLOCK) e->tail_fused_max = TTL_TAIL_MAX_BLOCKS * BLOCK;
    e->n_slots = 0;
    if (const char *v = getenv("TTL_LOCAL_SORT")) e->local_sort = atoi(v);
    e->poll_counts = 1;
    if (const char *v = getenv("TTL_POLL_COUNTS")) e->poll_counts = atoi(v);
    e->n_exact = 0;
    e->fr_cap = 0;
    e->fr_host_word = nullptr;
    e->prof_mask = (1 << TTL_PROFILE_CLASSES) - 1;
    e->prof_on = 0;
    e->prof_cap = 0;
    for (int k = 0; k < TTL_PROFILE_CLASSES; ++k) {
        e->prof_n[k] = 0;
        e->prof_ev[k] = nullptr;
    }
    *out = e;
    return TTL_OK;
}

static void prof_free(ttl_env *env) {
    for (int k = 0; k < TTL_PROFILE_CLASSES; ++k) {
        if (env->prof_ev[k]) {
            for (int j = 0; j < 2 * env->prof_cap; ++j) (void)hipEventDestroy(env->prof_ev[k][j]);
            delete[] env->prof_ev[k];
            env->prof_ev[k] = nullptr;
        }
        env->prof_n[k] = 0;
    }
    env->prof_cap = 0;
    env->prof_on = 0;
}

void ttl_env_destroy(ttl_env *env) {
    if (!env) return;
    prof_free(env);
    if (env->ev_prefix) (void)hipEventDestroy(env->ev_prefix);
    if (env->ev_counts) (void)hipEventDestroy(env->ev_counts);
    if (env->side) (void)hipStreamDestroy(env->side);
    delete env;
}

int ttl_env_profile_begin(ttl_env *env, int32_t max_launches, int32_t class_mask) {
    const int all = (1 << TTL_PROFILE_CLASSES) - 1;
    if (!env || max_launches < 1 || max_launches > (1 << 20) || !(class_mask & all))
        return fail(TTL_ERR_INVALID, "ttl_env_profile_begin: bad arguments");
    prof_free(env);
    env->prof_mask = class_mask & all;
    for (int k = 0; k < TTL_PROFILE_CLASSES; ++k) {
        env->prof_ev[k] = new (std::nothrow) hipEvent_t[2 * (size_t)max_launches];
        if (!env->prof_ev[k]) return fail(TTL_ERR_INVALID, "ttl_env_profile_begin: out of host memory");
        for (int j = 0; j < 2 * max_launches; ++j) HIP_TRY(hipEventCreate(&env->prof_ev[k][j]));
    }
    env->prof_cap = max_launches;
    env->prof_on = 1;
    return TTL_OK;
}

int ttl_env_profile_end(ttl_env *env, double *total_ms, int32_t *n_launches) {
    if (!env || !total_ms || !n_launches)
        return fail(TTL_ERR_INVALID, "ttl_env_profile_end: null argument");
    if (!env->prof_cap) return fail(TTL_ERR_STATE, "ttl_env_profile_end: profiling is off");
    for (int k = 0; k < TTL_PROFILE_CLASSES; ++k) {
        double acc = 0.0;
        for (int j = 0; j < env->prof_n[k]; ++j) {
            float ms = 0.f;
            HIP_TRY(hipEventSynchronize(env->prof_ev[k][2 * j + 1]));
            HIP_TRY(hipEventElapsedTime(&ms, env->prof_ev[k][2 * j], env->prof_ev[k][2 * j + 1]));
            acc += ms;
        }
        total_ms[k] = acc;
        n_launches[k] = env->prof_n[k];
    }
    prof_free(env);
    return TTL_OK;
}

int ttl_scripted_actions(const float *state, int64_t state_pitch, int32_t dir_offset,
                         const int32_t *continue_idx, int32_t n, uint32_t seed,
                         uint32_t step, float wobble, float *actions_out,
                         void *hip_stream) {
    if (!state || !continue_idx || !actions_out || n < 1 || dir_offset < 0 ||
        state_pitch < dir_offset + 3)
        return fail(TTL_ERR_INVALID, "ttl_scripted_actions: bad arguments");
    hipLaunchKernelGGL(k_scripted_actions, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0,
                       (hipStream_t)hip_stream, state, (long long)state_pitch,
                       dir_offset, continue_idx, n, seed, step, wobble, actions_out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_env_reset(ttl_env *env, const float *seeds, int32_t n,
                  const int32_t *processing_order, float *state_out,
                  int64_t state_pitch, void *hip_stream) {
    if (!env || !seeds || !state_out)
        return fail(TTL_ERR_INVALID, "ttl_env_reset: null argument");
    const ttl_env_desc &d = env->d;
    if (n < 1 || n > d.n_max)
        return fail(TTL_ERR_INVALID, "ttl_env_reset: n=%d outside [1, %d]", n, d.n_max);
    const int64_t width = 7LL * d.n_coef + 3LL * d.n_dirs;
    if (state_pitch < width)
        return fail(TTL_ERR_INVALID, "ttl_env_reset: state_pitch %lld < %lld",
                    (long long)state_pitch, (long long)width);
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t hist_bytes = (size_t)n * (size_t)(d.max_nb_steps + 1) * 3 * sizeof(float);
    HIP_TRY(hipMemsetAsync(d.streamlines, 0, hist_bytes, s));
    hipLaunchKernelGGL(k_reset, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s,
                       env->P, d.idx_a, seeds, n);
    HIP_TRY(hipGetLastError());
    env->cur = 0;
    env->length = 1;
    env->n_active = n;
    env->n_exact = 1;
    env->stepped = 0;
    env->fr_cap = 0;
    env->proc_cur = 0;
    env->use_proc = processing_order != nullptr;
    env->n_slots = n;
    if (processing_order == TTL_ORDER_BY_POSITION) {
        // the library's own order: rows sorted by the brick of their seed
        const int rc = ttl_detail_refresh_order(env->P, d.idx_a, n, env->order_ws,
                                                env->order_ws_bytes, env->proc[0], s);
        if (rc != TTL_OK) return rc;
    } else if (env->use_proc) {
        HIP_TRY(hipMemcpyAsync(env->proc[0], processing_order, (size_t)n * sizeof(int32_t),
                               hipMemcpyDeviceToDevice, s));
    }
    log_gather_rows(n);
    return ttl_detail_launch_state(env->P, env->state_kernel, nullptr, nullptr,
                                   env->use_proc ? env->proc[0] : nullptr, n, 1, state_out,
                                   state_pitch, s);
}

// side-stream copy of {n_continue, n_stopped} to the caller's pinned buffer,
// ordered after everything queued on `s` so far
static hipError_t ttl_copy_counts(ttl_env *env, int32_t *host_counts, hipStream_t s) {
    hipError_t e;
    if (!env->side) {
        if ((e = hipStreamCreateWithFlags(&env->side, hipStreamNonBlocking)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&env->ev_prefix, hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&env->ev_counts, hipEventDisableTiming)) != hipSuccess) return e;
    }
    if ((e = hipEventRecord(env->ev_prefix, s)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(env->side, env->ev_prefix, 0)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(host_counts, env->P.counts, 2 * sizeof(int32_t),
                            hipMemcpyDeviceToHost, env->side)) != hipSuccess) return e;
    if ((e = hipEventRecord(env->ev_counts, env->side)) != hipSuccess) return e;
    env->counts_pending = 1;
    env->host_counts = host_counts;
    return hipSuccess;
}

int ttl_env_step_begin(ttl_env *env, const float *actions, const double *noise,
                       int32_t n_active, double *reward_out, uint8_t *done_out,
                       void *hip_stream) {
    if (!env || !actions || !done_out)
        return fail(TTL_ERR_INVALID, "ttl_env_step: null argument");
    if (env->length < 1) return fail(TTL_ERR_STATE, "ttl_env_step: reset first");
    if (env->fr_cap) return fail(TTL_ERR_STATE, "ttl_env_step: free-running steps are enqueued, call ttl_env_freerun_end first");
    if (env->stepped) return fail(TTL_ERR_STATE, "ttl_env_step: harvest the previous step first");
    if (n_active < 1 || n_active > env->n_active ||
        (env->n_exact && n_active != env->n_active))
        return fail(TTL_ERR_INVALID, "ttl_env_step: n_active=%d, but %s%d streamlines are active",
                    n_active, env->n_exact ? "" : "at most ", env->n_active);
    const ttl_env_desc &d = env->d;
    if (env->length > d.max_nb_steps)
        return fail(TTL_ERR_STATE, "ttl_env_step: streamline history is full");
    if (noise && d.mode != TTL_MODE_F64DIR)
        return fail(TTL_ERR_INVALID, "ttl_env_step: noise needs TTL_MODE_F64DIR");
    hipStream_t s = (hipStream_t)hip_stream;
    const int *idx = env->cur ? d.idx_b : d.idx_a;
    const int L = env->length;
    const int nb = (n_active + BLOCK - 1) / BLOCK;
    prof_mark(env, 0, 0, s);
#define TTL_LAUNCH_ADVANCE(M)                                                  \
    hipLaunchKernelGGL((k_advance<M>), dim3(nb), dim3(BLOCK), 0, s, env->P, idx, \
                       actions, noise, n_active, L, reward_out, done_out)
    if (d.mode == TTL_MODE_F32) TTL_LAUNCH_ADVANCE(TTL_MODE_F32);
    else if (d.mode == TTL_MODE_F64DIR) TTL_LAUNCH_ADVANCE(TTL_MODE_F64DIR);
    else TTL_LAUNCH_ADVANCE(TTL_MODE_F32NORM);
#undef TTL_LAUNCH_ADVANCE
    prof_mark(env, 0, 1, s);
    HIP_TRY(hipGetLastError());
    env->length = L + 1;
    env->stepped = 2;          // advanced, not yet committed
    env->last_n = n_active;
    env->last_done = done_out;
    return TTL_OK;
}

int ttl_env_step_end(ttl_env *env, const uint8_t *extra_flags, int32_t order,
                     float *state_out, int64_t state_pitch, int32_t *host_counts,
                     void *hip_stream) {
    if (!env || !state_out)
        return fail(TTL_ERR_INVALID, "ttl_env_step: null argument");
    if (env->stepped != 2)
        return fail(TTL_ERR_STATE, "ttl_env_step_end: call ttl_env_step_begin first");
    if (order != TTL_ORDER_ACTIVE && order != TTL_ORDER_PARTITION)
        return fail(TTL_ERR_INVALID, "ttl_env_step: bad order %d", order);
    const ttl_env_desc &d = env->d;
    const int64_t width = 7LL * d.n_coef + 3LL * d.n_dirs;
    if (state_pitch < width)
        return fail(TTL_ERR_INVALID, "ttl_env_step: state_pitch too small");
    hipStream_t s = (hipStream_t)hip_stream;
    const int *idx = env->cur ? d.idx_b : d.idx_a;
    int *idx_next = env->cur ? d.idx_a : d.idx_b;
    const int n_active = env->last_n;
    const int n_pts = env->length;
    const int nb = (n_active + BLOCK - 1) / BLOCK;
    if (extra_flags) {
        hipLaunchKernelGGL(k_restop, dim3(nb), dim3(BLOCK), 0, s, env->P, idx,
                           extra_flags, n_active, env->last_done);
        HIP_TRY(hipGetLastError());
    }
    // a few thousand streamlines fit the caches in any order: stop paying for
    // the processing order in the episode's tail (from 16 384 rows down the
    // one-launch tail below takes over)
    if (env->use_proc && (n_active < 8192 ||
                          (env->fuse_small && ttl_detail_can_fuse_tail(env->P, n_active))))
        env->use_proc = 0;
    const int *proc = env->use_proc ? env->proc[env->proc_cur] : nullptr;
    env->stepped = 1;
    env->last_order = order;
    // The survivor count goes straight into the caller's pinned buffer
    // ({n_continue, n_stopped, sequence number}, written by the kernel that
    // computes it) when that buffer is device-visible: no side stream, no copy
    // kernel waiting for free CUs behind the gather, no API calls between the
    // step's launches; the host polls the sequence word.
    int *host_word = nullptr;
    int seq = 0;
    if (host_counts && env->poll_counts) {
        if (env->host_probe != host_counts) {
            void *dev = nullptr;
            env->host_probe = host_counts;
            env->host_dev = nullptr;
            if (hipHostGetDevicePointer(&dev, host_counts, 0) == hipSuccess)
                env->host_dev = static_cast<int *>(dev);
            else
                (void)hipGetLastError();      // not pinned: copy path below
        }
        host_word = env->host_dev;
        // sequence numbers carry bit 30: word [2] of the same pinned buffer is
        // the "steps done" counter of a free-running episode (small values), and
        // a leftover count must never read as this step's sequence number
        static std::atomic<unsigned> g_seq{1};
        seq = (int)((g_seq.fetch_add(1, std::memory_order_relaxed) & 0x3fffffffu) | 0x40000000u);
    }
    if (host_word) {
        env->counts_pending = 2;
        env->host_counts = host_counts;
        env->poll_seq = seq;
        env->poll_stream = s;
    }
    if (!proc && env->fuse_small && env->state_kernel != 0 &&
        ttl_detail_can_fuse_tail(env->P, n_active)) {
        // small batch: prefix + compaction + gather in ONE launch
        prof_mark(env, 2, 0, s);
        const int rc = ttl_detail_launch_fused_tail(env->P, idx, idx_next, n_active, order,
                                                    n_pts, state_out, state_pitch,
                                                    host_word, seq, s);
        prof_mark(env, 2, 1, s);
        if (rc != TTL_OK) return rc;
        if (host_counts && !host_word) HIP_TRY(ttl_copy_counts(env, host_counts, s));
        return TTL_OK;
    }
    // batches with a processing order: rows and slots in one launch (k_tail)
    // when the gather reads per-slot records and the block counts fit its scan;
    // the order then keeps its length between refreshes (holes), so the gather
    // is launched for n_slots slots
    int n_gather = n_active;
    const bool fused_tail = proc && env->tail_fused && env->P.slot_rec &&
                            ttl_detail_state_dedupes(env->P, env->state_kernel) &&
                            (env->n_slots < 0 ? n_active : env->n_slots) <= env->tail_fused_max;
    if (fused_tail) {
        if (env->n_slots < 0) env->n_slots = n_active;     // the order was dense so far
        const int nbs = (env->n_slots + BLOCK - 1) / BLOCK;
        prof_mark(env, 1, 0, s);
        hipLaunchKernelGGL(k_tail, dim3(nbs), dim3(BLOCK), 0, s, env->P, idx, idx_next, proc,
                           env->proc[env->proc_cur ^ 1], n_active, env->n_slots, nb, order,
                           n_pts, host_word, seq, env->local_sort);
        prof_mark(env, 1, 1, s);
        HIP_TRY(hipGetLastError());
        if (host_counts && !host_word) HIP_TRY(ttl_copy_counts(env, host_counts, s));
        n_gather = env->n_slots;
    } else {
        // the two-kernel tail reads proc[j] for j < n_active without a hole check:
        // an order that still holds holes (left by k_tail) must never reach it.
        // The knobs that choose the tail are fixed per handle, so this cannot
        // happen today; a setter that flips one mid-episode fails here instead
        // of reading stop[-1]
        if (proc && env->n_slots > n_active)
            return fail(TTL_ERR_STATE, "ttl_env_step: the processing order holds %d holes but "
                                       "the step tail was switched to the two-kernel form; "
                                       "refresh the order first", env->n_slots - n_active);
        prof_mark(env, 1, 0, s);
        hipLaunchKernelGGL(k_prefix, dim3(nb), dim3(BLOCK), 0, s, env->P, idx, idx_next,
                           proc, n_active, nb, order, n_pts, host_word, seq);
        prof_mark(env, 1, 1, s);
        HIP_TRY(hipGetLastError());
        if (host_counts && !host_word) {
            // fallback (buffer not device-visible): ship the count on a side stream
            // as soon as k_prefix has run
            HIP_TRY(ttl_copy_counts(env, host_counts, s));
        }
        if (proc) {
            // next step's processing order: this one, compacted in its own order
            // (ranks from k_prefix) and renumbered with the survivors' new row
            // ids; plus this step's per-slot records for the gather
            prof_mark(env, 3, 0, s);
            hipLaunchKernelGGL(k_proc_scatter, dim3(nb), dim3(BLOCK), 0, s, env->P, idx, proc,
                               env->proc[env->proc_cur ^ 1], n_active, nb,
                               env->local_sort && env->P.slot_rec);
            prof_mark(env, 3, 1, s);
            HIP_TRY(hipGetLastError());
            env->n_slots = -1;         // compacted: as long as the next step's active rows
        }
    }
    prof_mark(env, 2, 0, s);
    log_gather_rows(n_active);
    const int rc = ttl_detail_launch_state(env->P, env->state_kernel, idx, env->P.row_dest,
                                           proc, n_gather, n_pts, state_out, state_pitch, s);
    prof_mark(env, 2, 1, s);
    return rc;
}

int ttl_env_step(ttl_env *env, const float *actions, const double *noise,
                 int32_t n_active, int32_t order, float *state_out,
                 int64_t state_pitch, double *reward_out, uint8_t *done_out,
                 int32_t *host_counts, void *hip_stream) {
    if (!state_out) return fail(TTL_ERR_INVALID, "ttl_env_step: null argument");
    if (order != TTL_ORDER_ACTIVE && order != TTL_ORDER_PARTITION)
        return fail(TTL_ERR_INVALID, "ttl_env_step: bad order %d", order);
    if (env) {
        const int64_t width = 7LL * env->d.n_coef + 3LL * env->d.n_dirs;
        if (state_pitch < width)
            return fail(TTL_ERR_INVALID, "ttl_env_step: state_pitch too small");
    }
    const int rc = ttl_env_step_begin(env, actions, noise, n_active, reward_out,
                                      done_out, hip_stream);
    if (rc != TTL_OK) return rc;
    return ttl_env_step_end(env, nullptr, order, state_out, state_pitch,
                            host_counts, hip_stream);
}

int ttl_env_harvest(ttl_env *env, const float *state_in, float *state_out,
                    int64_t state_pitch, void *hip_stream) {
    if (!env) return fail(TTL_ERR_INVALID, "ttl_env_harvest: null handle");
    if (env->stepped != 1) return fail(TTL_ERR_STATE, "ttl_env_harvest: no finished step to harvest");
    const ttl_env_desc &d = env->d;
    hipStream_t s = (hipStream_t)hip_stream;
    const int *idx = env->cur ? d.idx_b : d.idx_a;
    const int n = env->last_n;
    const int nb = (n + BLOCK - 1) / BLOCK;
    if (env->last_order == TTL_ORDER_ACTIVE) {
        hipLaunchKernelGGL(k_finish, dim3(nb), dim3(BLOCK), 0, s, env->P, idx, n,
                           env->length);
        HIP_TRY(hipGetLastError());
        if (state_out) {
            if (!state_in)
                return fail(TTL_ERR_INVALID, "ttl_env_harvest: state_in needed to compact rows");
            const int width = 7 * d.n_coef + 3 * d.n_dirs;
            hipLaunchKernelGGL(k_copy_rows, dim3((n + 3) / 4), dim3(BLOCK), 0, s,
                               env->P, n, width, state_in, state_out,
                               (long long)state_pitch);
            HIP_TRY(hipGetLastError());
        }
    }
    env->cur ^= 1;
    if (env->use_proc) env->proc_cur ^= 1;
    env->stepped = 0;
    // until ttl_env_wait_counts() has read the survivor count back, the
    // previous count is only an upper bound
    env->n_active = n;
    env->n_exact = 0;
    return TTL_OK;
}

// blocks until the host_counts of the last step are in the pinned buffer;
// leaves counts_pending as it is (waiting twice is harmless)
static int await_counts(ttl_env *env) {
    if (env->counts_pending == 2) {
        // the step's kernel writes {n_continue, n_stopped, seq} into the pinned
        // buffer: poll the sequence word (bounded; then fall back to waiting
        // for the stream, after which the word must be there)
        const volatile int32_t *w = env->host_counts;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spin = 0;; ++spin) {
            if (w[2] == env->poll_seq) {
                seen = true;
                break;
            }
            if ((spin & 1023u) == 1023u &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200))
                break;
        }
        if (!seen) {
            HIP_TRY(hipStreamSynchronize(env->poll_stream));
            if (w[2] != env->poll_seq)
                return fail(TTL_ERR_HIP, "ttl_env_wait_counts: the survivor count never "
                                         "reached the pinned buffer");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        HIP_TRY(hipEventSynchronize(env->ev_counts));
    }
    return TTL_OK;
}

int ttl_env_wait_counts(ttl_env *env) {
    if (!env) return fail(TTL_ERR_INVALID, "ttl_env_wait_counts: null handle");
    if (!env->counts_pending)
        return fail(TTL_ERR_STATE, "ttl_env_wait_counts: the last step had no host_counts");
    const int rc = await_counts(env);
    if (rc != TTL_OK) return rc;
    env->counts_pending = 0;
    // the handle now knows the exact number of survivors: the next step must
    // be launched for exactly that many rows
    if (env->host_counts && !env->stepped) {
        env->n_active = env->host_counts[0];
        env->n_exact = 1;
    }
    return TTL_OK;
}

int ttl_env_stopped(ttl_env *env, const int32_t **stop_list, int32_t *n_stopped) {
    if (!env || !stop_list || !n_stopped)
        return fail(TTL_ERR_INVALID, "ttl_env_stopped: null argument");
    if (!env->stepped || !env->counts_pending || !env->host_counts)
        return fail(TTL_ERR_STATE, "ttl_env_stopped: call it between a step that was given "
                                   "host_counts and its harvest");
    const int rc = await_counts(env);
    if (rc != TTL_OK) return rc;
    *stop_list = env->P.stop_list;
    *n_stopped = env->host_counts[1];
    return TTL_OK;
}

int ttl_env_harvest_wait(ttl_env *env, const float *state_in, float *state_out,
                         int64_t state_pitch, void *hip_stream, int32_t *n_continue_out) {
    if (!n_continue_out) return fail(TTL_ERR_INVALID, "ttl_env_harvest_wait: null argument");
    int rc = ttl_env_harvest(env, state_in, state_out, state_pitch, hip_stream);
    if (rc != TTL_OK) return rc;
    rc = ttl_env_wait_counts(env);
    if (rc != TTL_OK) return rc;
    *n_continue_out = env->n_active;
    return TTL_OK;
}

// ---------------------------------------------------------------------------
// Free-running steps: a step as a fixed sequence of launches whose row count,
// length and buffer parity live in device memory, for capture in a HIP graph.
// ---------------------------------------------------------------------------
int ttl_env_freerun_begin(ttl_env *env, int32_t *host_counts, void *hip_stream) {
    if (!env) return fail(TTL_ERR_INVALID, "ttl_env_freerun_begin: null handle");
    if (env->length < 1) return fail(TTL_ERR_STATE, "ttl_env_freerun_begin: reset first");
    if (env->stepped) return fail(TTL_ERR_STATE, "ttl_env_freerun_begin: harvest the previous step first");
    if (env->fr_cap) return fail(TTL_ERR_STATE, "ttl_env_freerun_begin: already free-running");
    if (!env->n_exact)
        return fail(TTL_ERR_STATE, "ttl_env_freerun_begin: the survivor count of the last step has not been read");
    if (env->n_active < 1) return fail(TTL_ERR_STATE, "ttl_env_freerun_begin: no active streamline");
    if (env->state_kernel == 0 || !ttl_detail_can_fuse_tail(env->P, env->n_active))
        return fail(TTL_ERR_UNSUPPORTED, "ttl_env_freerun_begin: needs a batch of at most %d rows, "
                    "a neighbourhood radius in (0, 1) voxel and a volume below 4 GiB", env->P.fuse_max_rows);
    hipStream_t s = (hipStream_t)hip_stream;
    env->fr_host_word = nullptr;
    if (host_counts) {
        void *dev = nullptr;
        if (hipHostGetDevicePointer(&dev, host_counts, 0) != hipSuccess) {
            (void)hipGetLastError();
            return fail(TTL_ERR_INVALID, "ttl_env_freerun_begin: host_counts must be pinned, device-visible memory");
        }
        env->fr_host_word = static_cast<int *>(dev);
        host_counts[0] = env->n_active;
        host_counts[1] = 0;
        host_counts[2] = 0;
    }
    hipLaunchKernelGGL(k_fr_init, dim3(1), dim3(1), 0, s, env->P, env->n_active, env->length,
                       env->cur);
    HIP_TRY(hipGetLastError());
    env->use_proc = 0;
    env->counts_pending = 0;
    env->fr_cap = env->n_active;
    return TTL_OK;
}

int ttl_env_freerun_step(ttl_env *env, const float *actions, int32_t n_rows, float *state_out,
                         int64_t state_pitch, double *reward_out, uint8_t *done_out,
                         void *hip_stream) {
    if (!env || !actions || !state_out || !done_out)
        return fail(TTL_ERR_INVALID, "ttl_env_freerun_step: null argument");
    if (!env->fr_cap) return fail(TTL_ERR_STATE, "ttl_env_freerun_step: call ttl_env_freerun_begin first");
    if (n_rows < 1 || n_rows > env->fr_cap)
        return fail(TTL_ERR_INVALID, "ttl_env_freerun_step: n_rows=%d outside [1, %d]", n_rows,
                    env->fr_cap);
    const ttl_env_desc &d = env->d;
    const int64_t width = 7LL * d.n_coef + 3LL * d.n_dirs;
    if (state_pitch < width)
        return fail(TTL_ERR_INVALID, "ttl_env_freerun_step: state_pitch too small");
    hipStream_t s = (hipStream_t)hip_stream;
    const int n_cap = n_rows;
    const int nb = (n_cap + BLOCK - 1) / BLOCK;
    // launches only: nothing below waits, copies or asks the runtime anything,
    // so the call may run under stream capture
#define TTL_LAUNCH_ADVANCE_FR(M)                                                     \
    hipLaunchKernelGGL((k_advance_fr<M>), dim3(nb), dim3(BLOCK), 0, s, env->P, d.idx_a, \
                       d.idx_b, actions, n_cap, reward_out, done_out)
    if (d.mode == TTL_MODE_F32) TTL_LAUNCH_ADVANCE_FR(TTL_MODE_F32);
    else if (d.mode == TTL_MODE_F64DIR) TTL_LAUNCH_ADVANCE_FR(TTL_MODE_F64DIR);
    else TTL_LAUNCH_ADVANCE_FR(TTL_MODE_F32NORM);
#undef TTL_LAUNCH_ADVANCE_FR
    HIP_TRY(hipGetLastError());
    return ttl_detail_launch_fused_tail_fr(env->P, d.idx_a, d.idx_b, n_cap, state_out,
                                           state_pitch, env->fr_host_word, s);
}

int ttl_env_freerun_scripted_actions(ttl_env *env, const float *state, int64_t state_pitch,
                                     int32_t dir_offset, int32_t n_rows, uint32_t seed,
                                     float wobble, float *actions_out, void *hip_stream) {
    if (!env || !state || !actions_out || n_rows < 1 || dir_offset < 0 ||
        state_pitch < dir_offset + 3)
        return fail(TTL_ERR_INVALID, "ttl_env_freerun_scripted_actions: bad arguments");
    if (!env->fr_cap || n_rows > env->fr_cap)
        return fail(TTL_ERR_STATE, "ttl_env_freerun_scripted_actions: not free-running for %d rows", n_rows);
    hipLaunchKernelGGL(k_scripted_actions_fr, dim3((n_rows + BLOCK - 1) / BLOCK), dim3(BLOCK), 0,
                       (hipStream_t)hip_stream, env->P, state, (long long)state_pitch, dir_offset,
                       env->d.idx_a, env->d.idx_b, n_rows, seed, wobble, actions_out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_env_freerun_end(ttl_env *env, int32_t *n_active_out, int32_t *length_out,
                        int32_t *steps_out, void *hip_stream) {
    if (!env) return fail(TTL_ERR_INVALID, "ttl_env_freerun_end: null handle");
    if (!env->fr_cap) return fail(TTL_ERR_STATE, "ttl_env_freerun_end: not free-running");
    hipStream_t s = (hipStream_t)hip_stream;
    int live[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(live, env->P.counts + TTL_FR_LIVE, sizeof(live),
                           hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    env->n_active = live[0];
    env->n_exact = 1;
    env->length = live[1];
    env->cur = live[2];
    env->stepped = 0;
    env->fr_cap = 0;
    if (n_active_out) *n_active_out = live[0];
    if (length_out) *length_out = live[1];
    if (steps_out) *steps_out = live[3];
    return TTL_OK;
}

int ttl_env_stopping_flags(ttl_env *env, const float *tail, int32_t n,
                           int32_t n_points, uint8_t *flags_out,
                           void *hip_stream) {
    if (!env || !tail || !flags_out)
        return fail(TTL_ERR_INVALID, "ttl_env_stopping_flags: null argument");
    if (n < 1 || n_points < 1)
        return fail(TTL_ERR_INVALID, "ttl_env_stopping_flags: n and n_points must be >= 1");
    hipLaunchKernelGGL(k_probe_flags, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0,
                       (hipStream_t)hip_stream, env->P, tail, n, n_points,
                       flags_out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_env_set_processing_order(ttl_env *env, const int32_t *order, int32_t n,
                                 void *hip_stream) {
    if (!env || !order) return fail(TTL_ERR_INVALID, "ttl_env_set_processing_order: null argument");
    if (env->length < 1 || env->stepped)
        return fail(TTL_ERR_STATE, "ttl_env_set_processing_order: between harvest and step only");
    if (!env->n_exact || n != env->n_active)
        return fail(TTL_ERR_INVALID, "ttl_env_set_processing_order: n=%d, %d rows are active",
                    n, env->n_active);
    HIP_TRY(hipMemcpyAsync(env->proc[env->proc_cur], order, (size_t)n * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    env->use_proc = 1;
    env->n_slots = n;
    return TTL_OK;
}

int ttl_env_refresh_processing_order(ttl_env *env, void *hip_stream) {
    if (!env) return fail(TTL_ERR_INVALID, "ttl_env_refresh_processing_order: null handle");
    if (env->length < 1 || env->stepped)
        return fail(TTL_ERR_STATE, "ttl_env_refresh_processing_order: between harvest and step only");
    if (!env->n_exact)
        return fail(TTL_ERR_STATE, "ttl_env_refresh_processing_order: survivor count not read back yet");
    const int *idx = env->cur ? env->d.idx_b : env->d.idx_a;
    const int rc = ttl_detail_refresh_order(env->P, idx, env->n_active, env->order_ws,
                                            env->order_ws_bytes, env->proc[env->proc_cur],
                                            (hipStream_t)hip_stream);
    if (rc == TTL_OK) {
        env->use_proc = 1;
        env->n_slots = env->n_active;
    }
    return rc;
}

int ttl_pack_streamlines(const float *history, int64_t row_pitch, const int64_t *keep,
                         const int64_t *offsets, int32_t n, float *points_out,
                         void *hip_stream) {
    if (!history || !keep || !offsets || !points_out || n < 1 || row_pitch < 3)
        return fail(TTL_ERR_INVALID, "ttl_pack_streamlines: bad arguments");
    const int per_block = BLOCK / 64;
    hipLaunchKernelGGL(k_pack_streamlines, dim3((n + per_block - 1) / per_block), dim3(BLOCK), 0,
                       (hipStream_t)hip_stream, history, (long long)row_pitch,
                       (const long long *)keep, (const long long *)offsets, n, points_out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_env_view(ttl_env *env, const int32_t **continue_idx,
                 const int32_t **row_dest, int32_t *length) {
    if (!env) return fail(TTL_ERR_INVALID, "ttl_env_view: null handle");
    if (continue_idx) *continue_idx = env->cur ? env->d.idx_b : env->d.idx_a;
    if (row_dest) *row_dest = env->P.row_dest;
    if (length) *length = env->length;
    return TTL_OK;
}

}  // extern "C"
