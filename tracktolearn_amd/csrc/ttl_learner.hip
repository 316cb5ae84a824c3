// ttl_learner.hip -- the learner kernels of libttl_hip.so (include/ttl_learner.h):
// everything of one SAC / SACAuto update that is not a dense GEMM.
//
// Shapes: batch rows M = 4096..8192, hidden width 1024 per network, "thin"
// layers of 1..6 units.  Every kernel here streams its matrices once (HBM /
// Infinity-Cache bound, no reuse to tile for), 64-wide waves reading 1 KB of a
// row per instruction (float4 per lane), four waves per workgroup taking
// interleaved rows.  Column reductions (bias / thin-layer weight gradients) are
// block partials in a slab + ttl_colsum_finalize: fixed summation order, no
// float atomics, so a graph replay reproduces the eager update bit for bit.
#include <cmath>

#include "ttl_internal.h"
#include "ttl_learner.h"

namespace {

constexpr int LB = 256;   // threads per workgroup
constexpr int NW = 4;     // waves per workgroup
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;   // log(sqrt(2 pi))
constexpr float LOG_2 = 0.69314718055994530942f;
constexpr float LOG_STD_MIN = -20.f, LOG_STD_MAX = 2.f;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    // v + (v of the lane CTRL selects; 0 where it selects none / the row is masked)
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(t);
}

__device__ __forceinline__ float wave_sum(float v) {
    // 64-lane sum on the VALU's DPP path (no LDS crossbar: a ds_bpermute
    // butterfly of the 24 sums a thin layer makes per wave costs more than its
    // loads): prefix sums inside each row of 16 lanes (row_shr 1, 2, 4, 8),
    // then lane 15 of a row into the next row (row_bcast:15, rows 1 and 3),
    // lane 31 into rows 2 and 3 (row_bcast:31); lane 63 holds the total.
    // Fixed order: the same bits on every launch.  Wave-uniform result.
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int V>
__device__ __forceinline__ void ldv(float (&x)[V], const float *p) {
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
    } else {
        x[0] = *p;
    }
}
template <int V>
__device__ __forceinline__ void stv(float *p, const float (&x)[V]) {
    if constexpr (V == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
        *p = x[0];
    }
}

__device__ __forceinline__ float softplus_t(float x) {
    // torch.nn.functional.softplus, beta 1, threshold 20
    return x > 20.f ? x : log1pf(expf(x));
}

// ------------------------------------------------------------------------
// thin forward (+ squashed-gaussian head)
// ------------------------------------------------------------------------
struct ThinFwd {
    const float *a; int64_t lda; int64_t a_bs;
    const float *w; const float *b;
    int n_rows, n_in;
    int head; const float *eps; int ent_rows;
    float *out; int64_t ld_out;
    float *logp; float *ls_raw; float *ent_part;
};

constexpr int FWD_ROWS = TTL_THIN_FWD_ROWS;     // rows per workgroup
static_assert(FWD_ROWS == 4, "one epilogue thread per row, 4 x NOUT combine threads");

// A workgroup takes FWD_ROWS rows; its four waves split the columns (each wave
// a contiguous quarter, rounded up to whole 64 * V chunks), so 8192 rows are
// 8192 waves with a handful of 16-byte loads in flight each, and the thin
// layer's weights are read once per workgroup.  Per-wave partial sums meet in
// LDS in wave order.
template <int NOUT, bool BD, int V>
__global__ __launch_bounds__(LB) void k_thin_forward(ThinFwd P) {
    __shared__ float part[NW][FWD_ROWS][NOUT];
    __shared__ float y[FWD_ROWS][NOUT];
    __shared__ float ent[FWD_ROWS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m0 = blockIdx.x * FWD_ROWS;
    const int chunk = 64 * V;
    const int per_wave = ((P.n_in + NW * chunk - 1) / (NW * chunk)) * chunk;
    const int c_lo = wv * per_wave, c_hi = min(c_lo + per_wave, P.n_in);
    float acc[FWD_ROWS][NOUT];
#pragma unroll
    for (int r = 0; r < FWD_ROWS; ++r)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[r][o] = 0.f;
    const float *arow[FWD_ROWS];
#pragma unroll
    for (int r = 0; r < FWD_ROWS; ++r) {
        const int m = min(m0 + r, P.n_rows - 1);      // clamped: loads stay in bounds
        arow[r] = P.a + (int64_t)m * P.lda;
    }
    if constexpr (!BD) {
        for (int c = c_lo + lane * V; c < c_hi; c += chunk) {
            float av[FWD_ROWS][V];
#pragma unroll
            for (int r = 0; r < FWD_ROWS; ++r) ldv<V>(av[r], arow[r] + c);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float wq[V];
                ldv<V>(wq, P.w + (int64_t)o * P.n_in + c);
#pragma unroll
                for (int r = 0; r < FWD_ROWS; ++r)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[r][o] += av[r][v] * wq[v];
            }
        }
    } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            for (int c = c_lo + lane * V; c < c_hi; c += chunk) {
                float wq[V];
                ldv<V>(wq, P.w + (int64_t)o * P.n_in + c);
#pragma unroll
                for (int r = 0; r < FWD_ROWS; ++r) {
                    float av[V];
                    ldv<V>(av, arow[r] + (int64_t)o * P.a_bs + c);
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[r][o] += av[v] * wq[v];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < FWD_ROWS; ++r)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float t = wave_sum(acc[r][o]);
            if (lane == 0) part[wv][r][o] = t;
        }
    __syncthreads();
    if (threadIdx.x < FWD_ROWS * NOUT) {
        const int r = threadIdx.x / NOUT, o = threadIdx.x - r * NOUT;
        y[r][o] = (((part[0][r][o] + part[1][r][o]) + part[2][r][o]) + part[3][r][o]) + P.b[o];
    }
    __syncthreads();
    const int r = threadIdx.x, m = m0 + r;
    if (P.head != TTL_HEAD_SAC) {
        if (r < FWD_ROWS && m < P.n_rows) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
                P.out[(int64_t)m * P.ld_out + o] = P.head == TTL_HEAD_TANH ? tanhf(y[r][o]) : y[r][o];
        }
        return;
    }
    if constexpr (NOUT % 2 == 0 && NOUT >= 2) {
        constexpr int NA = NOUT / 2;
        if (r < FWD_ROWS) {
            float lp = 0.f;
            if (m < P.n_rows) {
                float lg = 0.f, corr = 0.f;
                float *dst = P.out + (int64_t)m * P.ld_out;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const float mu = y[r][i], raw = y[r][NA + i];
                    const float ls = fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX);
                    const float sd = expf(ls);
                    const float u = mu + P.eps[(int64_t)m * NA + i] * sd;
                    const float var = sd * sd;
                    const float d = u - mu;
                    const float g = -(d * d) / (2.f * var) - logf(sd) - HALF_LOG_2PI;
                    const float cr = 2.f * (LOG_2 - u - softplus_t(-2.f * u));
                    lg = i == 0 ? g : lg + g;
                    corr = i == 0 ? cr : corr + cr;
                    dst[i] = tanhf(u);
                    P.ls_raw[(int64_t)m * NA + i] = raw;
                }
                lp = lg - corr;
                P.logp[m] = lp;
            }
            ent[r] = m < P.ent_rows && m < P.n_rows ? lp : 0.f;
        }
        if (P.ent_part) {
            __syncthreads();
            if (threadIdx.x == 0)
                P.ent_part[blockIdx.x] = ((ent[0] + ent[1]) + ent[2]) + ent[3];
        }
    }
}

// ------------------------------------------------------------------------
// per-row loss terms, gradients w.r.t. the critic outputs, Adam step counters
// ------------------------------------------------------------------------
struct LossArgs {
    const float *q_on, *q_tg, *logp, *reward, *not_done;
    int n;
    const float *log_alpha; float alpha_const, gamma;
    float *dq, *loss_part;
    float *steps, *consts; double *beta_pows; int n_opt; unsigned tick_mask;
    double lr, beta1, beta2;
};

__global__ __launch_bounds__(LB) void k_sac_losses(LossArgs P) {
    const int i = blockIdx.x * LB + threadIdx.x;
    const float alpha = P.log_alpha ? expf(P.log_alpha[0]) : P.alpha_const;
    const float inv_n = 1.f / (float)P.n;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < P.n) {
        const float tq = fminf(P.q_tg[2 * i], P.q_tg[2 * i + 1]);
        const float backup = P.reward[i] +
                             (P.gamma * P.not_done[i]) * (tq - alpha * P.logp[P.n + i]);
        const float q1 = P.q_on[2 * i], q2 = P.q_on[2 * i + 1];
        const float e1 = q1 - backup, e2 = q2 - backup;
        P.dq[2 * i] = 2.f * e1 * inv_n;
        P.dq[2 * i + 1] = 2.f * e2 * inv_n;
        const float p1 = P.q_on[2 * (P.n + i)], p2 = P.q_on[2 * (P.n + i) + 1];
        // d(-min(p1, p2) / n): the smaller one takes it, a tie splits it
        P.dq[2 * (P.n + i)] = p1 < p2 ? -inv_n : (p1 == p2 ? -0.5f * inv_n : 0.f);
        P.dq[2 * (P.n + i) + 1] = p2 < p1 ? -inv_n : (p1 == p2 ? -0.5f * inv_n : 0.f);
        s[0] = alpha * P.logp[i] - fminf(p1, p2);
        s[1] = e1 * e1; s[2] = e2 * e2; s[3] = q1; s[4] = q2; s[5] = backup;
    }
    if (P.loss_part) {
        __shared__ float red[NW][6];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float t = wave_sum(s[k]);
            if (lane == 0) red[wv][k] = t;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int k = threadIdx.x;
            P.loss_part[blockIdx.x * 8 + k] =
                k < 6 ? ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k] : 0.f;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int k = 0; k < P.n_opt; ++k) {
            if (!((P.tick_mask >> k) & 1u)) continue;
            P.steps[k] = P.steps[k] + 1.f;
            // beta^step kept as running products (float64): no pow() here
            const double p1 = P.beta_pows[2 * k] * P.beta1, p2 = P.beta_pows[2 * k + 1] * P.beta2;
            P.beta_pows[2 * k] = p1;
            P.beta_pows[2 * k + 1] = p2;
            P.consts[2 * k] = (float)(P.lr / (1.0 - p1));
            P.consts[2 * k + 1] = (float)sqrt(1.0 - p2);
        }
    }
}

struct Td3LossArgs {
    const float *q_on, *q_tg, *reward, *not_done;
    int n, n_q;
    float gamma;
    float *dq, *loss_part;
    float *steps, *consts; double *beta_pows; int n_opt; unsigned tick_mask;
    double lr, beta1, beta2;
};

__global__ __launch_bounds__(LB) void k_td3_losses(Td3LossArgs P) {
    const int i = blockIdx.x * LB + threadIdx.x;
    const float inv_n = 1.f / (float)P.n;
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < P.n) {
        float tq = P.q_tg[P.n_q * i];
        if (P.n_q == 2) tq = fminf(tq, P.q_tg[2 * i + 1]);
        // td3.py:160-163 / ddpg.py:262-264: reward + not_done * gamma * target_Q
        const float target = P.reward[i] + (P.not_done[i] * P.gamma) * tq;
        const float q1 = P.q_on[P.n_q * i], e1 = q1 - target;
        P.dq[P.n_q * i] = 2.f * e1 * inv_n;
        s[1] = e1 * e1; s[3] = q1; s[5] = target;
        if (P.n_q == 2) {
            const float q2 = P.q_on[2 * i + 1], e2 = q2 - target;
            P.dq[2 * i + 1] = 2.f * e2 * inv_n;
            s[2] = e2 * e2; s[4] = q2;
        }
    }
    if (P.loss_part) {
        __shared__ float red[NW][6];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float t = wave_sum(s[k]);
            if (lane == 0) red[wv][k] = t;
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            const int k = threadIdx.x;
            P.loss_part[blockIdx.x * 8 + k] =
                k < 6 ? ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k] : 0.f;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int k = 0; k < P.n_opt; ++k) {
            if (!((P.tick_mask >> k) & 1u)) continue;
            P.steps[k] = P.steps[k] + 1.f;
            const double p1 = P.beta_pows[2 * k] * P.beta1, p2 = P.beta_pows[2 * k + 1] * P.beta2;
            P.beta_pows[2 * k] = p1;
            P.beta_pows[2 * k + 1] = p2;
            P.consts[2 * k] = (float)(P.lr / (1.0 - p1));
            P.consts[2 * k + 1] = (float)sqrt(1.0 - p2);
        }
    }
}

__global__ __launch_bounds__(LB) void k_polyak(float *target, const float *p, long long n,
                                               float tau, float om_tau) {
    const long long i4 = ((long long)blockIdx.x * LB + threadIdx.x) * 4;
    if (i4 + 3 < n) {
        float t[4], x[4];
        ldv<4>(t, target + i4);
        ldv<4>(x, p + i4);
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = t[k] * om_tau + x[k] * tau;
        stv<4>(target + i4, t);
    } else {
        for (long long i = i4; i < n; ++i) target[i] = target[i] * om_tau + p[i] * tau;
    }
}

// ------------------------------------------------------------------------
// thin backward / ReLU backward with column partials
// ------------------------------------------------------------------------
struct ThinBwd {
    const float *d_out; int64_t ld_dout;
    const float *a; int64_t lda; int64_t a_bs;    // block-diagonal: network o at a + o * a_bs
    const float *w;
    int n_rows, n_in, n_cols, r0, r1, rpb;
    float *dz; int64_t ld_dz; int64_t dz_bs;
    float *part; int64_t ld_part;
};

// sum the NW per-wave copies of NACC accumulator rows (64 * V columns each) in
// wave order and write them to the slab row: accumulator k goes to
// dst + off[k] + column
template <int NACC, int V>
__device__ __forceinline__ void block_reduce_store(float (*lds)[NACC][64 * V],
                                                   const float (&acc)[NACC][V], float *dst,
                                                   const int64_t (&off)[NACC], int col0,
                                                   int n_cols) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) lds[wv][k][lane * V + v] = acc[k][v];
    __syncthreads();
    for (int e = threadIdx.x; e < NACC * 64 * V; e += LB) {
        const int k = e / (64 * V), c = e - k * (64 * V);
        if (col0 + c < n_cols)
            dst[off[k] + col0 + c] =
                ((lds[0][k][c] + lds[1][k][c]) + lds[2][k][c]) + lds[3][k][c];
    }
}

template <int NOUT, bool BD, int V>
__global__ __launch_bounds__(LB) void k_thin_backward(ThinBwd P) {
    constexpr int NACC = BD ? 2 : 1 + NOUT;
    __shared__ float lds[NW][NACC][64 * V];
    __shared__ float dbo_s[NW][NOUT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col0 = blockIdx.x * 64 * V;
    const int col = col0 + lane * V;
    const bool cok = col < P.n_cols;
    const int colc = cok ? col : 0;
    const int ob = BD ? colc / P.n_in : 0;                 // my critic (block diagonal)
    // element (m, col) of a / dz: networks side by side in a row (bs = n_in) or
    // in planes of their own (bs = plane size, row stride n_in)
    const int64_t a_off = BD ? (int64_t)ob * P.a_bs + (colc - ob * P.n_in) : colc;
    const int64_t dz_off = BD ? (int64_t)ob * P.dz_bs + (colc - ob * P.n_in) : colc;
    float wq[BD ? 1 : NOUT][V];
    if constexpr (BD) {
        ldv<V>(wq[0], P.w + colc);                         // [n_out][n_in] flat == column
    } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) ldv<V>(wq[o], P.w + (int64_t)o * P.n_in + colc);
    }
    float acc[NACC][V];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[k][v] = 0.f;
    float dbo[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) dbo[o] = 0.f;

    const int mb = blockIdx.y * P.rpb;
#pragma unroll 4
    for (int i = wv; i < P.rpb; i += NW) {
        const int m = mb + i;
        if (m >= P.n_rows) break;
        float d[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) d[o] = P.d_out[(int64_t)m * P.ld_dout + o];
        float av[V], g[V];
        ldv<V>(av, P.a + (int64_t)m * P.lda + a_off);
        float dsel = d[0];
        if constexpr (BD) {
#pragma unroll
            for (int o = 1; o < NOUT; ++o) dsel = ob == o ? d[o] : dsel;
#pragma unroll
            for (int v = 0; v < V; ++v) g[v] = dsel * wq[0][v];
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float t = d[0] * wq[0][v];
#pragma unroll
                for (int o = 1; o < NOUT; ++o) t += d[o] * wq[o][v];
                g[v] = t;
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) g[v] = av[v] > 0.f ? g[v] : 0.f;
        if (cok) stv<V>(P.dz + (int64_t)m * P.ld_dz + dz_off, g);
        if (m >= P.r0 && m < P.r1) {
#pragma unroll
            for (int v = 0; v < V; ++v) acc[0][v] += g[v];
            if constexpr (BD) {
#pragma unroll
                for (int v = 0; v < V; ++v) acc[1][v] += dsel * av[v];
            } else {
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[1 + o][v] += d[o] * av[v];
            }
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dbo[o] += d[o];
        }
    }
    float *dst = P.part + (int64_t)blockIdx.y * P.ld_part;
    int64_t off[NACC];
    off[0] = 0;
    if constexpr (BD) {
        off[1] = P.n_cols;
    } else {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) off[1 + o] = P.n_cols + (int64_t)o * P.n_in;
    }
    block_reduce_store<NACC, V>(lds, acc, dst, off, col0, P.n_cols);
    if (blockIdx.x == 0) {
        if (lane == 0) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) dbo_s[wv][o] = dbo[o];
        }
        __syncthreads();
        if (threadIdx.x < NOUT) {
            const int o = threadIdx.x;
            dst[P.n_cols + (int64_t)NOUT * P.n_in + o] =
                ((dbo_s[0][o] + dbo_s[1][o]) + dbo_s[2][o]) + dbo_s[3][o];
        }
    }
}

struct ReluBwd {
    float *dz; int64_t ld_dz; int64_t dz_ps;      // plane z at dz + z * dz_ps
    const float *a; int64_t lda; int64_t a_ps;
    int n_rows, n_cols, r0, r1, rpb;
    float *part; int64_t ld_part;
};

template <int V>
__global__ __launch_bounds__(LB) void k_relu_backward_bias(ReluBwd P) {
    __shared__ float lds[NW][1][64 * V];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col0 = blockIdx.x * 64 * V;
    const int col = col0 + lane * V;
    const bool cok = col < P.n_cols;
    const int colc = cok ? col : 0;
    float *dz = P.dz + (int64_t)blockIdx.z * P.dz_ps;
    const float *a = P.a + (int64_t)blockIdx.z * P.a_ps;
    float acc[1][V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[0][v] = 0.f;
    const int mb = blockIdx.y * P.rpb;
#pragma unroll 4
    for (int i = wv; i < P.rpb; i += NW) {
        const int m = mb + i;
        if (m >= P.n_rows) break;
        float av[V], g[V];
        ldv<V>(av, a + (int64_t)m * P.lda + colc);
        ldv<V>(g, dz + (int64_t)m * P.ld_dz + colc);
#pragma unroll
        for (int v = 0; v < V; ++v) g[v] = av[v] > 0.f ? g[v] : 0.f;
        if (m >= P.r0 && m < P.r1) {
#pragma unroll
            for (int v = 0; v < V; ++v) acc[0][v] += g[v];
        }
        if (cok) stv<V>(dz + (int64_t)m * P.ld_dz + col, g);
    }
    // plane z owns the columns [z * n_cols, (z + 1) * n_cols) of the slab row
    const int64_t off[1] = {(int64_t)blockIdx.z * P.n_cols};
    block_reduce_store<1, V>(lds, acc, P.part + (int64_t)blockIdx.y * P.ld_part, off, col0,
                             P.n_cols);
}

// ------------------------------------------------------------------------
// slab -> gradient
// ------------------------------------------------------------------------
struct Segs {
    ttl_colsum_seg s[TTL_COLSUM_MAX_SEGS];
    int first_block[TTL_COLSUM_MAX_SEGS + 1];
    int n;
};

__global__ __launch_bounds__(LB) void k_colsum_finalize(Segs S) {
    __shared__ float lds[LB];
    int k = 0;
    while (k + 1 < S.n && (int)blockIdx.x >= S.first_block[k + 1]) ++k;
    const ttl_colsum_seg sg = S.s[k];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (sg.n > 8) {
        const int col = (blockIdx.x - S.first_block[k]) * 64 + lane;
        float acc = 0.f;
        if (col < sg.n) {
            // eight independent loads in flight, summed in row order
            const float *src = sg.part + col;
            int r = wv;
            for (; r + 7 * NW < sg.n_part; r += 8 * NW) {
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = src[(int64_t)(r + k * NW) * sg.ld];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += x[k];
            }
            for (; r < sg.n_part; r += NW) acc += src[(int64_t)r * sg.ld];
        }
        lds[threadIdx.x] = acc;
        __syncthreads();
        if (wv == 0 && col < sg.n) {
            float t = ((lds[lane] + lds[64 + lane]) + lds[128 + lane]) + lds[192 + lane];
            t *= sg.scale;
            if (sg.accumulate) t += sg.out[col];
            sg.out[col] = t;
        }
        return;
    }
    // narrow segment (scalars: loss terms, thin-layer bias grads; n <= 8): thread
    // t sums the rows r = t / 8, t / 8 + 32, ... of column t % 8, then one thread
    // per column adds the 32 partial sums in order
    {
        const int c = threadIdx.x & 7, g = threadIdx.x >> 3;
        float acc = 0.f;
        if (c < sg.n)
            for (int r = g; r < sg.n_part; r += LB / 8) acc += sg.part[(int64_t)r * sg.ld + c];
        lds[threadIdx.x] = acc;
        __syncthreads();
        if ((int)threadIdx.x < sg.n) {
            float t = lds[threadIdx.x];
            for (int k = 1; k < LB / 8; ++k) t += lds[k * 8 + threadIdx.x];
            t *= sg.scale;
            if (sg.accumulate) t += sg.out[threadIdx.x];
            sg.out[threadIdx.x] = t;
        }
    }
}

// ------------------------------------------------------------------------
// d(actor loss) / d(actor head outputs)
// ------------------------------------------------------------------------
struct HeadBwd {
    const float *dh; int64_t ld_dh;
    const float *h; int64_t ld_h;
    const float *wa;                 // [n_act][n_cols]: action columns of the stacked W1
    int n_rows, n_cols;
    const float *pi; int64_t ld_pi;
    const float *eps, *ls_raw, *log_alpha; float alpha_const;
    float *d_head;
    int head;                         // TTL_HEAD_SAC or TTL_HEAD_TANH
};

template <int NA, int V>
__global__ __launch_bounds__(LB) void k_sac_actor_head_backward(HeadBwd P) {
    // as k_thin_forward: FWD_ROWS rows per workgroup, the columns split over its waves
    __shared__ float part[NW][FWD_ROWS][NA];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int m0 = blockIdx.x * FWD_ROWS;
    const int chunk = 64 * V;
    const int per_wave = ((P.n_cols + NW * chunk - 1) / (NW * chunk)) * chunk;
    const int c_lo = wv * per_wave, c_hi = min(c_lo + per_wave, P.n_cols);
    float acc[FWD_ROWS][NA];
#pragma unroll
    for (int r = 0; r < FWD_ROWS; ++r)
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[r][i] = 0.f;
    const float *hrow[FWD_ROWS], *drow[FWD_ROWS];
#pragma unroll
    for (int r = 0; r < FWD_ROWS; ++r) {
        const int m = min(m0 + r, P.n_rows - 1);
        hrow[r] = P.h + (int64_t)m * P.ld_h;
        drow[r] = P.dh + (int64_t)m * P.ld_dh;
    }
    for (int c = c_lo + lane * V; c < c_hi; c += chunk) {
        float g[FWD_ROWS][V];
#pragma unroll
        for (int r = 0; r < FWD_ROWS; ++r) {
            float hv[V];
            ldv<V>(hv, hrow[r] + c);
            ldv<V>(g[r], drow[r] + c);
#pragma unroll
            for (int v = 0; v < V; ++v) g[r][v] = hv[v] > 0.f ? g[r][v] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            float wq[V];
            ldv<V>(wq, P.wa + (int64_t)i * P.n_cols + c);
#pragma unroll
            for (int r = 0; r < FWD_ROWS; ++r)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[r][i] += g[r][v] * wq[v];
        }
    }
#pragma unroll
    for (int r = 0; r < FWD_ROWS; ++r)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const float t = wave_sum(acc[r][i]);
            if (lane == 0) part[wv][r][i] = t;
        }
    __syncthreads();
    if (threadIdx.x >= FWD_ROWS * NA) return;
    const int r = threadIdx.x / NA, i = threadIdx.x - r * NA, m = m0 + r;
    if (m >= P.n_rows) return;
    const float dpi = ((part[0][r][i] + part[1][r][i]) + part[2][r][i]) + part[3][r][i];
    if (P.head == TTL_HEAD_TANH) {
        // deterministic actor (offpolicy.py:54-60): d tanh(u) / du = 1 - pi^2
        const float t = P.pi[(int64_t)m * P.ld_pi + i];
        P.d_head[(int64_t)m * NA + i] = dpi * (1.f - t * t);
        return;
    }
    const float alpha = P.log_alpha ? expf(P.log_alpha[0]) : P.alpha_const;
    const float an = alpha / (float)P.n_rows;
    const float t = P.pi[(int64_t)m * P.ld_pi + i];
    const float du = an * (2.f * t) + dpi * (1.f - t * t);
    const float raw = P.ls_raw[(int64_t)m * NA + i];
    const bool in = raw >= LOG_STD_MIN && raw <= LOG_STD_MAX;
    const float sd = expf(fminf(fmaxf(raw, LOG_STD_MIN), LOG_STD_MAX));
    P.d_head[(int64_t)m * 2 * NA + i] = du;
    P.d_head[(int64_t)m * 2 * NA + NA + i] = in ? du * (P.eps[(int64_t)m * NA + i] * sd) - an : 0.f;
}

// ------------------------------------------------------------------------
// Adam + Polyak over a flat arena
// ------------------------------------------------------------------------
struct AdamArgs {
    float *p, *g, *m, *v, *t;
    int64_t n;
    const float *consts;
    // torch's Python scalars, rounded to f32 once: 1 - beta1, beta2, 1 - beta2,
    // eps, tau, 1 - tau
    float om_b1, b2, om_b2, eps, tau, om_tau;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float *t,
                                         const AdamArgs &A, float step_size, float bc2s) {
    m = m + (g - m) * A.om_b1;                          // lerp_(grad, 1 - beta1)
    v = v * A.b2 + (A.om_b2 * g) * g;                   // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) / bc2s + A.eps;
    p = p + (-step_size) * (m / denom);                 // addcdiv_(m, denom, -step_size)
    if (t) *t = *t * A.om_tau + p * A.tau;
}

__global__ __launch_bounds__(LB) void k_adam_polyak(AdamArgs A) {
    const float step_size = A.consts[0], bc2s = A.consts[1];
    const int64_t i4 = ((int64_t)blockIdx.x * LB + threadIdx.x) * 4;
    if (i4 + 3 < A.n) {
        float p[4], g[4], m[4], v[4], t[4];
        ldv<4>(p, A.p + i4); ldv<4>(g, A.g + i4); ldv<4>(m, A.m + i4); ldv<4>(v, A.v + i4);
        if (A.t) ldv<4>(t, A.t + i4);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            adam_one(p[k], g[k], m[k], v[k], A.t ? &t[k] : nullptr, A, step_size, bc2s);
        stv<4>(A.p + i4, p); stv<4>(A.m + i4, m); stv<4>(A.v + i4, v);
        if (A.t) stv<4>(A.t + i4, t);
    } else {
        for (int64_t i = i4; i < A.n; ++i) {
            float p = A.p[i], m = A.m[i], v = A.v[i], t = A.t ? A.t[i] : 0.f;
            adam_one(p, A.g[i], m, v, A.t ? &t : nullptr, A, step_size, bc2s);
            A.p[i] = p; A.m[i] = m; A.v[i] = v;
            if (A.t) A.t[i] = t;
        }
    }
}

struct AlphaArgs {
    float *p, *g, *m, *v;
    const float *mean_logp; float target_entropy;
    const float *consts;
    float om_b1, b2, om_b2, eps;
};

__global__ void k_sac_alpha_step(AlphaArgs P) {
    if (threadIdx.x || blockIdx.x) return;
    const float ml = P.mean_logp[0];
    const float g = -(ml + P.target_entropy);
    const float alpha_before = expf(P.p[0]);
    AdamArgs A{P.p, P.g, P.m, P.v, nullptr, 1, P.consts, P.om_b1, P.b2, P.om_b2, P.eps, 0.f, 1.f};
    float p = P.p[0], m = P.m[0], v = P.v[0];
    adam_one(p, g, m, v, nullptr, A, P.consts[0], P.consts[1]);
    P.p[0] = p; P.m[0] = m; P.v[0] = v;
    P.g[0] = g + alpha_before * ml;
}

// ------------------------------------------------------------------------
// network input rows
// ------------------------------------------------------------------------
struct BuildArgs {
    const float *s; int64_t ld_s; const float *a; int64_t ld_a; const float *s2; int64_t ld_s2;
    int n, n_state, n_act;
    float *xs; int64_t ld;
    const float *w1; int64_t ld_w1; int n_w1; float *wa;
    int row_blocks;
};

__global__ __launch_bounds__(LB) void k_build_learner_inputs(BuildArgs P) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if ((int)blockIdx.x >= P.row_blocks) {
        // the action columns of the critics' first layer, transposed
        const int j = (blockIdx.x - P.row_blocks) * LB + threadIdx.x;
        if (j < P.n_w1)
            for (int i = 0; i < P.n_act; ++i)
                P.wa[(int64_t)i * P.n_w1 + j] = P.w1[(int64_t)j * P.ld_w1 + P.n_state + i];
        return;
    }
    const int m = blockIdx.x * NW + wv;
    if (m >= P.n) return;
    const float *s = P.s + (int64_t)m * P.ld_s, *s2 = P.s2 + (int64_t)m * P.ld_s2;
    float *a_row = P.xs + (int64_t)m * P.ld, *pi_row = P.xs + (int64_t)(P.n + m) * P.ld;
    float *n_row = P.xs + (int64_t)(2 * P.n + m) * P.ld;
    for (int c = lane; c < P.n_state; c += 64) {
        const float x = s[c], y = s2[c];
        a_row[c] = x;
        pi_row[c] = x;
        n_row[c] = y;
    }
    if (lane < P.n_act) a_row[P.n_state + lane] = P.a[(int64_t)m * P.ld_a + lane];
}

// ------------------------------------------------------------------------
// replay ring: a batch of distinct uniformly drawn transitions in one launch
// ------------------------------------------------------------------------
struct SampleArgs {
    const float *state, *action, *next_state, *reward, *not_done;
    long long size; int n_state, n_act, batch;
    unsigned key0, key1;
    float *o_state, *o_action, *o_next, *o_reward, *o_not_done;
    long long *o_index;
    int half_bits;
};

__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// position j of a keyed pseudo-random permutation of [0, size): a six-round
// balanced Feistel network over the next even power of two, cycle-walked back
// into the range (a bijection of [0, 2^2h) restricted to the orbit of j is a
// bijection of [0, size): positions 0..batch-1 are distinct indices)
__device__ __forceinline__ unsigned long long prp_index(unsigned long long j, const SampleArgs &P) {
    const unsigned mask = (1u << P.half_bits) - 1u;
    unsigned long long x = j;
    do {
        unsigned l = (unsigned)(x >> P.half_bits) & mask, r = (unsigned)x & mask;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const unsigned f = mix32(r * 0x9E3779B1u + ((k & 1) ? P.key1 : P.key0) +
                                     (unsigned)k * 0x7F4A7C15u);
            const unsigned t = l ^ (f & mask);
            l = r;
            r = t;
        }
        x = ((unsigned long long)l << P.half_bits) | r;
    } while (x >= (unsigned long long)P.size);
    return x;
}

__global__ __launch_bounds__(LB) void k_replay_sample(SampleArgs P) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * NW + wv;            // one wave per sampled transition
    if (j >= P.batch) return;
    const unsigned long long i = prp_index((unsigned long long)j, P);
    const float *s = P.state + i * P.n_state, *s2 = P.next_state + i * P.n_state;
    float *os = P.o_state + (long long)j * P.n_state, *on = P.o_next + (long long)j * P.n_state;
    for (int c = lane; c < P.n_state; c += 64) {
        os[c] = s[c];
        on[c] = s2[c];
    }
    if (lane < P.n_act) P.o_action[(long long)j * P.n_act + lane] = P.action[i * P.n_act + lane];
    if (lane == 0) {
        P.o_reward[j] = P.reward[i];
        P.o_not_done[j] = P.not_done[i];
        if (P.o_index) P.o_index[j] = (long long)i;
    }
}

// ------------------------------------------------------------------------
// replay ring: append a batch of transitions in one launch
// ------------------------------------------------------------------------
struct AddArgs {
    const float *state, *action, *next_state;
    const int *row_dest;            // next_state row of transition i, or null (= i)
    const double *reward64; const float *reward32;
    const unsigned char *done;
    int n, n_state, n_act;
    long long ptr, max_size;
    float *r_state, *r_action, *r_next, *r_reward, *r_not_done;
};

__global__ __launch_bounds__(LB) void k_replay_add(AddArgs P) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * NW + wv;              // one wave per transition
    if (i >= P.n) return;
    const long long slot = (P.ptr + i) % P.max_size;
    const long long src_next = P.row_dest ? P.row_dest[i] : i;
    const float *s = P.state + (long long)i * P.n_state;
    const float *s2 = P.next_state + src_next * P.n_state;
    float *ds = P.r_state + slot * P.n_state, *dn = P.r_next + slot * P.n_state;
    for (int c = lane; c < P.n_state; c += 64) {
        ds[c] = s[c];
        dn[c] = s2[c];
    }
    if (lane < P.n_act) P.r_action[slot * P.n_act + lane] = P.action[(long long)i * P.n_act + lane];
    if (lane == 0) {
        P.r_reward[slot] = P.reward64 ? (float)P.reward64[i] : P.reward32[i];
        P.r_not_done[slot] = 1.f - (float)P.done[i];
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }
inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

#define LAUNCH_CHECK(name)                                                       \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess)                                                    \
            return fail(TTL_ERR_HIP, name ": %s", hipGetErrorString(e_));        \
    } while (0)

extern "C" {

int ttl_thin_forward(const float *a, int64_t lda, int64_t a_block_stride, const float *w,
                     const float *b, int32_t n_rows, int32_t n_in, int32_t n_out,
                     int32_t block_diagonal, int32_t head, const float *eps, int32_t entropy_rows, float *out,
                     int64_t ld_out, float *logp, float *log_std_raw, float *entropy_part,
                     void *hip_stream) {
    if (!a || !w || !b || !out || n_rows <= 0 || n_in <= 0)
        return fail(TTL_ERR_INVALID, "ttl_thin_forward: null pointer or empty shape");
    if (lda < n_in || (block_diagonal && a_block_stride < n_in))
        return fail(TTL_ERR_INVALID, "ttl_thin_forward: lda %lld / block stride %lld < n_in %d",
                    (long long)lda, (long long)a_block_stride, n_in);
    if (block_diagonal && a_block_stride < lda && lda < (int64_t)n_out * a_block_stride)
        return fail(TTL_ERR_INVALID, "ttl_thin_forward: side-by-side networks need lda >= "
                                     "n_out * block stride");
    if (head == TTL_HEAD_SAC) {
        if (block_diagonal || (n_out != 2 && n_out != 4 && n_out != 6 && n_out != 8))
            return fail(TTL_ERR_INVALID, "ttl_thin_forward: SAC head needs a dense layer of "
                                         "2, 4, 6 or 8 units (got %d)", n_out);
        if (!eps || !logp || !log_std_raw || entropy_rows < 0 || entropy_rows > n_rows)
            return fail(TTL_ERR_INVALID, "ttl_thin_forward: SAC head outputs missing");
        if (ld_out < n_out / 2)
            return fail(TTL_ERR_INVALID, "ttl_thin_forward: output stride too small");
    } else if (head != TTL_HEAD_PLAIN && head != TTL_HEAD_TANH) {
        return fail(TTL_ERR_INVALID, "ttl_thin_forward: unknown head %d", head);
    } else if (ld_out < n_out) {
        return fail(TTL_ERR_INVALID, "ttl_thin_forward: output stride too small");
    }
    ThinFwd P{a, lda, block_diagonal ? a_block_stride : 0, w, b, n_rows, n_in, head, eps,
              entropy_rows, out, ld_out, logp, log_std_raw,
              head == TTL_HEAD_SAC ? entropy_part : nullptr};
    const bool vec = n_in % 4 == 0 && lda % 4 == 0 && aligned16(a) && aligned16(w) &&
                     (!block_diagonal || a_block_stride % 4 == 0);
    const dim3 grid((n_rows + TTL_THIN_FWD_ROWS - 1) / TTL_THIN_FWD_ROWS), block(LB);
    hipStream_t s = S(hip_stream);
#define FWD_CASE(N, BDV)                                                                  \
    if (n_out == N && (block_diagonal != 0) == BDV) {                                     \
        if (vec) k_thin_forward<N, BDV, 4><<<grid, block, 0, s>>>(P);                     \
        else k_thin_forward<N, BDV, 1><<<grid, block, 0, s>>>(P);                         \
        LAUNCH_CHECK("k_thin_forward");                                                   \
        return TTL_OK;                                                                    \
    }
    FWD_CASE(1, false) FWD_CASE(2, false) FWD_CASE(3, false) FWD_CASE(4, false)
    FWD_CASE(6, false) FWD_CASE(8, false) FWD_CASE(2, true) FWD_CASE(1, true)
#undef FWD_CASE
    return fail(TTL_ERR_UNSUPPORTED, "ttl_thin_forward: no kernel for n_out = %d (%s)", n_out,
                block_diagonal ? "block diagonal" : "dense");
}

int ttl_sac_losses(const float *q_online, const float *q_target, const float *logp,
                   const float *reward, const float *not_done, int32_t n,
                   const float *log_alpha, float alpha_const, float gamma, float *dq,
                   float *loss_part, float *steps, float *adam_consts, double *beta_pows,
                   int32_t n_opt, uint32_t tick_mask, double lr, double beta1, double beta2,
                   void *hip_stream) {
    if (!q_online || !q_target || !logp || !reward || !not_done || !dq || n <= 0)
        return fail(TTL_ERR_INVALID, "ttl_sac_losses: null pointer or empty batch");
    if (n_opt < 0 || n_opt > 8 || (n_opt && (!steps || !adam_consts || !beta_pows)))
        return fail(TTL_ERR_INVALID, "ttl_sac_losses: bad optimizer table");
    LossArgs P{q_online, q_target, logp, reward, not_done, n, log_alpha, alpha_const, gamma,
               dq, loss_part, steps, adam_consts, beta_pows, n_opt, tick_mask, lr, beta1, beta2};
    k_sac_losses<<<dim3((n + LB - 1) / LB), dim3(LB), 0, S(hip_stream)>>>(P);
    LAUNCH_CHECK("k_sac_losses");
    return TTL_OK;
}

int ttl_thin_backward(const float *d_out, int64_t ld_dout, const float *a, int64_t lda,
                      int64_t a_block_stride, const float *w, int32_t n_rows, int32_t n_in,
                      int32_t n_out, int32_t block_diagonal, int32_t r0, int32_t r1,
                      int32_t rows_per_block, float *dz, int64_t ld_dz, int64_t dz_block_stride,
                      float *part, int64_t ld_part, void *hip_stream) {
    if (!d_out || !a || !w || !dz || !part || n_rows <= 0 || n_in <= 0 || rows_per_block <= 0)
        return fail(TTL_ERR_INVALID, "ttl_thin_backward: null pointer or empty shape");
    if (a == dz) return fail(TTL_ERR_INVALID, "ttl_thin_backward: a and dz alias");
    const int64_t n_cols = block_diagonal ? (int64_t)n_out * n_in : n_in;
    if (lda < n_in || ld_dz < n_in || ld_dout < n_out ||
        ld_part < n_cols + (int64_t)n_out * n_in + n_out ||
        (block_diagonal && (a_block_stride < n_in || dz_block_stride < n_in)))
        return fail(TTL_ERR_INVALID, "ttl_thin_backward: a stride is too small");
    if (!block_diagonal) a_block_stride = dz_block_stride = 0;
    ThinBwd P{d_out, ld_dout, a, lda, a_block_stride, w, n_rows, n_in, (int)n_cols, r0, r1,
              rows_per_block, dz, ld_dz, dz_block_stride, part, ld_part};
    const bool vec = n_in % 4 == 0 && lda % 4 == 0 && ld_dz % 4 == 0 && aligned16(a) &&
                     aligned16(dz) && aligned16(w) && a_block_stride % 4 == 0 &&
                     dz_block_stride % 4 == 0;
    const int V = vec ? 4 : 1;
    const dim3 grid((unsigned)((n_cols + 64 * V - 1) / (64 * V)),
                    (unsigned)((n_rows + rows_per_block - 1) / rows_per_block)), block(LB);
    hipStream_t s = S(hip_stream);
#define BWD_CASE(N, BDV)                                                                  \
    if (n_out == N && (block_diagonal != 0) == BDV) {                                     \
        if (vec) k_thin_backward<N, BDV, 4><<<grid, block, 0, s>>>(P);                    \
        else k_thin_backward<N, BDV, 1><<<grid, block, 0, s>>>(P);                        \
        LAUNCH_CHECK("k_thin_backward");                                                  \
        return TTL_OK;                                                                    \
    }
    BWD_CASE(1, false) BWD_CASE(2, false) BWD_CASE(3, false) BWD_CASE(4, false)
    BWD_CASE(6, false) BWD_CASE(8, false) BWD_CASE(2, true) BWD_CASE(1, true)
#undef BWD_CASE
    return fail(TTL_ERR_UNSUPPORTED, "ttl_thin_backward: no kernel for n_out = %d (%s)", n_out,
                block_diagonal ? "block diagonal" : "dense");
}

int ttl_relu_backward_bias(float *dz, int64_t ld_dz, int64_t dz_plane_stride, const float *a,
                           int64_t lda, int64_t a_plane_stride, int32_t n_planes,
                           int32_t n_rows, int32_t n_cols, int32_t r0, int32_t r1,
                           int32_t rows_per_block, float *part, int64_t ld_part,
                           void *hip_stream) {
    if (!dz || !a || !part || n_rows <= 0 || n_cols <= 0 || rows_per_block <= 0 ||
        n_planes < 1 || n_planes > 64)
        return fail(TTL_ERR_INVALID, "ttl_relu_backward_bias: null pointer or empty shape");
    if (ld_dz < n_cols || lda < n_cols || ld_part < (int64_t)n_planes * n_cols)
        return fail(TTL_ERR_INVALID, "ttl_relu_backward_bias: a stride is too small");
    ReluBwd P{dz, ld_dz, dz_plane_stride, a, lda, a_plane_stride, n_rows, n_cols, r0, r1,
              rows_per_block, part, ld_part};
    const bool vec = n_cols % 4 == 0 && lda % 4 == 0 && ld_dz % 4 == 0 && aligned16(a) &&
                     aligned16(dz) && dz_plane_stride % 4 == 0 && a_plane_stride % 4 == 0;
    const int V = vec ? 4 : 1;
    const dim3 grid((unsigned)((n_cols + 64 * V - 1) / (64 * V)),
                    (unsigned)((n_rows + rows_per_block - 1) / rows_per_block),
                    (unsigned)n_planes), block(LB);
    if (vec) k_relu_backward_bias<4><<<grid, block, 0, S(hip_stream)>>>(P);
    else k_relu_backward_bias<1><<<grid, block, 0, S(hip_stream)>>>(P);
    LAUNCH_CHECK("k_relu_backward_bias");
    return TTL_OK;
}

int ttl_colsum_finalize(const ttl_colsum_seg *segs, int32_t n_segs, void *hip_stream) {
    if (!segs || n_segs <= 0 || n_segs > TTL_COLSUM_MAX_SEGS)
        return fail(TTL_ERR_INVALID, "ttl_colsum_finalize: 1..%d segments", TTL_COLSUM_MAX_SEGS);
    Segs Sg;
    int blocks = 0;
    for (int k = 0; k < n_segs; ++k) {
        const ttl_colsum_seg &g = segs[k];
        if (!g.part || !g.out || g.n <= 0 || g.n_part <= 0 || g.ld < g.n)
            return fail(TTL_ERR_INVALID, "ttl_colsum_finalize: segment %d is malformed", k);
        Sg.s[k] = g;
        Sg.first_block[k] = blocks;
        blocks += g.n > 8 ? (g.n + 63) / 64 : 1;
    }
    Sg.first_block[n_segs] = blocks;
    Sg.n = n_segs;
    k_colsum_finalize<<<dim3(blocks), dim3(LB), 0, S(hip_stream)>>>(Sg);
    LAUNCH_CHECK("k_colsum_finalize");
    return TTL_OK;
}

int ttl_sac_actor_head_backward(const float *dh, int64_t ld_dh, const float *h, int64_t ld_h,
                                const float *wa, int32_t n_rows, int32_t n_cols, int32_t n_act,
                                int32_t head, const float *pi, int64_t ld_pi, const float *eps,
                                const float *log_std_raw, const float *log_alpha,
                                float alpha_const, float *d_head, void *hip_stream) {
    if (head != TTL_HEAD_SAC && head != TTL_HEAD_TANH)
        return fail(TTL_ERR_INVALID, "ttl_sac_actor_head_backward: unknown head %d", head);
    if (!dh || !h || !wa || !pi || !d_head || n_rows <= 0 || n_cols <= 0 ||
        (head == TTL_HEAD_SAC && (!eps || !log_std_raw)))
        return fail(TTL_ERR_INVALID, "ttl_sac_actor_head_backward: null pointer or empty shape");
    if (ld_dh < n_cols || ld_h < n_cols || ld_pi < n_act)
        return fail(TTL_ERR_INVALID, "ttl_sac_actor_head_backward: a stride is too small");
    HeadBwd P{dh, ld_dh, h, ld_h, wa, n_rows, n_cols, pi, ld_pi, eps, log_std_raw, log_alpha,
              alpha_const, d_head, head};
    const bool vec = n_cols % 4 == 0 && ld_dh % 4 == 0 && ld_h % 4 == 0 && aligned16(dh) &&
                     aligned16(h) && aligned16(wa);
    const dim3 grid((n_rows + FWD_ROWS - 1) / FWD_ROWS), block(LB);
    hipStream_t s = S(hip_stream);
#define HB_CASE(N)                                                                        \
    if (n_act == N) {                                                                     \
        if (vec) k_sac_actor_head_backward<N, 4><<<grid, block, 0, s>>>(P);               \
        else k_sac_actor_head_backward<N, 1><<<grid, block, 0, s>>>(P);                   \
        LAUNCH_CHECK("k_sac_actor_head_backward");                                        \
        return TTL_OK;                                                                    \
    }
    HB_CASE(1) HB_CASE(2) HB_CASE(3) HB_CASE(4)
#undef HB_CASE
    return fail(TTL_ERR_UNSUPPORTED, "ttl_sac_actor_head_backward: n_act = %d", n_act);
}

int ttl_adam_polyak(float *p, float *g, float *m, float *v, float *target, int64_t n,
                    const float *consts, double beta1, double beta2, double eps, double tau,
                    void *hip_stream) {
    if (!p || !g || !m || !v || !consts || n <= 0)
        return fail(TTL_ERR_INVALID, "ttl_adam_polyak: null pointer or empty arena");
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v) ||
        (target && !aligned16(target)))
        return fail(TTL_ERR_INVALID, "ttl_adam_polyak: arenas must be 16-byte aligned");
    AdamArgs A{p, g, m, v, target, n, consts, (float)(1.0 - beta1), (float)beta2,
               (float)(1.0 - beta2), (float)eps, (float)tau, (float)(1.0 - tau)};
    const int64_t threads = (n + 3) / 4;
    k_adam_polyak<<<dim3((unsigned)((threads + LB - 1) / LB)), dim3(LB), 0, S(hip_stream)>>>(A);
    LAUNCH_CHECK("k_adam_polyak");
    return TTL_OK;
}

int ttl_sac_alpha_step(float *log_alpha, float *grad, float *m, float *v,
                       const float *mean_logp, float target_entropy, const float *consts,
                       double beta1, double beta2, double eps, void *hip_stream) {
    if (!log_alpha || !grad || !m || !v || !mean_logp || !consts)
        return fail(TTL_ERR_INVALID, "ttl_sac_alpha_step: null pointer");
    AlphaArgs P{log_alpha, grad, m, v, mean_logp, target_entropy, consts, (float)(1.0 - beta1),
                (float)beta2, (float)(1.0 - beta2), (float)eps};
    k_sac_alpha_step<<<dim3(1), dim3(64), 0, S(hip_stream)>>>(P);
    LAUNCH_CHECK("k_sac_alpha_step");
    return TTL_OK;
}

int ttl_build_learner_inputs(const float *state, int64_t ld_s, const float *action, int64_t ld_a,
                             const float *next_state, int64_t ld_s2, int32_t n, int32_t n_state,
                             int32_t n_act, float *xs, int64_t ld, const float *w1,
                             int64_t ld_w1, int32_t n_w1_rows, float *wa, void *hip_stream) {
    if (!state || !action || !next_state || !xs || n <= 0 || n_state <= 0 || n_act <= 0)
        return fail(TTL_ERR_INVALID, "ttl_build_learner_inputs: null pointer or empty shape");
    if (ld_s < n_state || ld_s2 < n_state || ld_a < n_act || ld < n_state + n_act || n_act > 64)
        return fail(TTL_ERR_INVALID, "ttl_build_learner_inputs: a stride is too small");
    if (w1 && (!wa || n_w1_rows <= 0 || ld_w1 < n_state + n_act))
        return fail(TTL_ERR_INVALID, "ttl_build_learner_inputs: bad first-layer weights");
    const int row_blocks = (n + NW - 1) / NW;
    const int w_blocks = w1 ? (n_w1_rows + LB - 1) / LB : 0;
    BuildArgs P{state, ld_s, action, ld_a, next_state, ld_s2, n, n_state, n_act, xs, ld,
                w1, ld_w1, w1 ? n_w1_rows : 0, wa, row_blocks};
    k_build_learner_inputs<<<dim3(row_blocks + w_blocks), dim3(LB), 0, S(hip_stream)>>>(P);
    LAUNCH_CHECK("k_build_learner_inputs");
    return TTL_OK;
}

int ttl_td3_losses(const float *q_online, const float *q_target, const float *reward,
                   const float *not_done, int32_t n, int32_t n_q, float gamma, float *dq,
                   float *loss_part, float *steps, float *adam_consts, double *beta_pows,
                   int32_t n_opt, uint32_t tick_mask, double lr, double beta1, double beta2,
                   void *hip_stream) {
    if (!q_online || !q_target || !reward || !not_done || !dq || n <= 0 || (n_q != 1 && n_q != 2))
        return fail(TTL_ERR_INVALID, "ttl_td3_losses: null pointer, empty batch or n_q not 1 / 2");
    if (n_opt < 0 || n_opt > 8 || (n_opt && (!steps || !adam_consts || !beta_pows)))
        return fail(TTL_ERR_INVALID, "ttl_td3_losses: bad optimizer table");
    Td3LossArgs P{q_online, q_target, reward, not_done, n, n_q, gamma, dq, loss_part, steps,
                  adam_consts, beta_pows, n_opt, tick_mask, lr, beta1, beta2};
    k_td3_losses<<<dim3((n + LB - 1) / LB), dim3(LB), 0, S(hip_stream)>>>(P);
    LAUNCH_CHECK("k_td3_losses");
    return TTL_OK;
}

int ttl_polyak_average(float *target, const float *p, int64_t n, double tau, void *hip_stream) {
    if (!target || !p || n <= 0)
        return fail(TTL_ERR_INVALID, "ttl_polyak_average: null pointer or empty arena");
    if (!aligned16(target) || !aligned16(p))
        return fail(TTL_ERR_INVALID, "ttl_polyak_average: arenas must be 16-byte aligned");
    const int64_t threads = (n + 3) / 4;
    k_polyak<<<dim3((unsigned)((threads + LB - 1) / LB)), dim3(LB), 0, S(hip_stream)>>>(
        target, p, (long long)n, (float)tau, (float)(1.0 - tau));
    LAUNCH_CHECK("k_polyak");
    return TTL_OK;
}

int ttl_replay_add(const float *state, const float *action, const float *next_state,
                   const int32_t *row_dest, const double *reward_f64, const float *reward_f32,
                   const uint8_t *done, int32_t n, int32_t n_state, int32_t n_act, int64_t ptr,
                   int64_t max_size, float *ring_state, float *ring_action,
                   float *ring_next_state, float *ring_reward, float *ring_not_done,
                   void *hip_stream) {
    if (!state || !action || !next_state || !done || (!reward_f64 == !reward_f32) || !ring_state ||
        !ring_action || !ring_next_state || !ring_reward || !ring_not_done)
        return fail(TTL_ERR_INVALID, "ttl_replay_add: null pointer (exactly one reward array)");
    if (n <= 0 || n > max_size || ptr < 0 || ptr >= max_size || n_state <= 0 || n_act <= 0 ||
        n_act > 64)
        return fail(TTL_ERR_INVALID, "ttl_replay_add: need 0 < n <= max_size and 0 <= ptr < max_size");
    AddArgs P{state, action, next_state, row_dest, reward_f64, reward_f32, done, n, n_state, n_act,
              (long long)ptr, (long long)max_size, ring_state, ring_action, ring_next_state,
              ring_reward, ring_not_done};
    k_replay_add<<<dim3((n + NW - 1) / NW), dim3(LB), 0, S(hip_stream)>>>(P);
    LAUNCH_CHECK("k_replay_add");
    return TTL_OK;
}

int ttl_replay_sample(const float *state, const float *action, const float *next_state,
                      const float *reward, const float *not_done, int64_t size, int32_t n_state,
                      int32_t n_act, int32_t batch, uint32_t key0, uint32_t key1,
                      float *out_state, float *out_action, float *out_next_state,
                      float *out_reward, float *out_not_done, int64_t *out_index,
                      void *hip_stream) {
    if (!state || !action || !next_state || !reward || !not_done || !out_state || !out_action ||
        !out_next_state || !out_reward || !out_not_done)
        return fail(TTL_ERR_INVALID, "ttl_replay_sample: null pointer");
    if (size <= 0 || batch <= 0 || batch > size || n_state <= 0 || n_act <= 0 || n_act > 64 ||
        size > (1ll << 40))
        return fail(TTL_ERR_INVALID, "ttl_replay_sample: need 0 < batch <= size (got %d of %lld)",
                    batch, (long long)size);
    int bits = 2;
    while ((1ll << bits) < size) ++bits;
    if (bits & 1) ++bits;
    SampleArgs P{state, action, next_state, reward, not_done, (long long)size, n_state, n_act,
                 batch, key0, key1, out_state, out_action, out_next_state, out_reward,
                 out_not_done, reinterpret_cast<long long *>(out_index), bits / 2};
    k_replay_sample<<<dim3((batch + NW - 1) / NW), dim3(LB), 0, S(hip_stream)>>>(P);
    LAUNCH_CHECK("k_replay_sample");
    return TTL_OK;
}

}  // extern "C"
