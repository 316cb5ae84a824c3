// ttl_resample.hip -- arc-length resampling of padded streamline batches for
// the TractOracle-Net scoring path (TrackToLearn/oracles/oracle.py:52,70:
// dipy set_number_of_points(streamlines, 128)); part of libttl_hip.so.
// One wavefront per streamline: float64 segment lengths -> blocked wave scan
// of the cumulative arc length in LDS -> each lane places its target points
// by binary search and interpolates linearly inside the segment.
#include "ttl_internal.h"

namespace {
constexpr int BLOCK = TTL_BLOCK;

__global__ __launch_bounds__(BLOCK) void k_resample(
    const float *__restrict__ points, long long row_pitch, const int *__restrict__ lengths32,
    const long long *__restrict__ lengths64, int n, int max_len, int nb,
    float *__restrict__ out) {
    extern __shared__ __align__(16) double cum_all[];
    double *cum = cum_all + (size_t)(threadIdx.x >> 6) * max_len;   // this wave's [max_len]
    const int lane = threadIdx.x & 63;
    const int waves = (BLOCK / 64) * gridDim.x;
    for (int row = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); row < n; row += waves) {
        const float *p = points + (size_t)row * (size_t)row_pitch;
        float *o = out + (size_t)row * (size_t)nb * 3;
        int L = lengths32 ? lengths32[row] : (int)lengths64[row];
        L = min(max(L, 1), max_len);
        const int nseg = L - 1;
        // blocked scan: lane owns the contiguous segments [lo, hi)
        const int per = (nseg + 63) >> 6;
        const int lo = min(lane * per, nseg), hi = min(lo + per, nseg);
        double local = 0.0;
        for (int j = lo; j < hi; ++j) {
            const double dx = (double)p[3 * (j + 1) + 0] - (double)p[3 * j + 0];
            const double dy = (double)p[3 * (j + 1) + 1] - (double)p[3 * j + 1];
            const double dz = (double)p[3 * (j + 1) + 2] - (double)p[3 * j + 2];
            local = local + sqrt((dx * dx + dy * dy) + dz * dz);
            cum[j + 1] = local;                 // within-chunk prefix for now
        }
        double before = local;                  // inclusive scan over lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double up = __shfl_up(before, off);
            if (lane >= off) before = before + up;
        }
        before = before - local;                // exclusive
        for (int j = lo; j < hi; ++j) cum[j + 1] = cum[j + 1] + before;
        if (lane == 0) cum[0] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double total = cum[nseg];
        for (int k = lane; k < nb; k += 64) {
            float x, y, z;
            if (k == nb - 1 || nseg == 0) {     // the last point is kept exactly
                x = p[3 * nseg + 0];
                y = p[3 * nseg + 1];
                z = p[3 * nseg + 2];
            } else {
                const double target = total * ((double)k / (double)(nb - 1));
                // j = #{m in [0, nseg) : cum[m + 1] <= target}, at most nseg - 1
                int a = 0, b = nseg;
                while (a < b) {
                    const int mid = (a + b) >> 1;
                    if (cum[mid + 1] <= target) a = mid + 1;
                    else b = mid;
                }
                const int j = min(a, nseg - 1);
                const double c0 = cum[j], c1 = cum[j + 1];
                const double den = c1 - c0;
                const double r = den > 0.0 ? (target - c0) / den : 0.0;
                const double ax = p[3 * j + 0], ay = p[3 * j + 1], az = p[3 * j + 2];
                const double bx = p[3 * j + 3], by = p[3 * j + 4], bz = p[3 * j + 5];
                x = (float)(ax + r * (bx - ax));
                y = (float)(ay + r * (by - ay));
                z = (float)(az + r * (bz - az));
            }
            o[3 * k + 0] = x;
            o[3 * k + 1] = y;
            o[3 * k + 2] = z;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();        // cum is reused by the next row
    }
}

// History rows -> the network's input in one pass: what the env's oracle path
// does with torch ops (gather the rows' first n_pts points, optional 3x3 map
// into the oracle's voxel space, resample to nb points, difference), one
// wavefront per streamline.  The mapped points are rounded to float32 and the
// resampled points to float32 before differencing, as the separate steps do.
struct Lin { float m[9]; };          // row-major: out = p @ m

__global__ __launch_bounds__(BLOCK) void k_oracle_segments(
    const float *__restrict__ hist, long long row_pitch, const int *__restrict__ ids,
    int id_stride, int n, int n_pts, int use_lin, Lin lin, int nb, float *__restrict__ dirs) {
    extern __shared__ __align__(16) double seg_lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // per wave: cum [n_pts] doubles, pts [3 n_pts] floats, res [3 nb] floats
    const size_t per_wave = (size_t)n_pts * 8 + (size_t)n_pts * 12 + (size_t)nb * 12;
    char *base = reinterpret_cast<char *>(seg_lds) + (size_t)wv * ((per_wave + 15) & ~(size_t)15);
    double *cum = reinterpret_cast<double *>(base);
    float *pts = reinterpret_cast<float *>(base + (size_t)n_pts * 8);
    float *res = pts + 3 * (size_t)n_pts;
    const int waves = (BLOCK / 64) * gridDim.x;
    const int nseg = n_pts - 1;
    for (int row = blockIdx.x * (BLOCK / 64) + wv; row < n; row += waves) {
        const long long g = ids ? ids[(size_t)row * id_stride] : row;
        const float *p = hist + g * row_pitch;
        for (int j = lane; j < n_pts; j += 64) {
            float x = p[3 * j], y = p[3 * j + 1], z = p[3 * j + 2];
            if (use_lin) {
                const float a = x, b = y, c = z;
                x = fmaf(c, lin.m[6], fmaf(b, lin.m[3], a * lin.m[0]));
                y = fmaf(c, lin.m[7], fmaf(b, lin.m[4], a * lin.m[1]));
                z = fmaf(c, lin.m[8], fmaf(b, lin.m[5], a * lin.m[2]));
            }
            pts[3 * j] = x;
            pts[3 * j + 1] = y;
            pts[3 * j + 2] = z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // blocked scan of the segment lengths, as k_resample
        const int per = (nseg + 63) >> 6;
        const int lo = min(lane * per, nseg), hi = min(lo + per, nseg);
        double local = 0.0;
        for (int j = lo; j < hi; ++j) {
            const double dx = (double)pts[3 * (j + 1) + 0] - (double)pts[3 * j + 0];
            const double dy = (double)pts[3 * (j + 1) + 1] - (double)pts[3 * j + 1];
            const double dz = (double)pts[3 * (j + 1) + 2] - (double)pts[3 * j + 2];
            local = local + sqrt((dx * dx + dy * dy) + dz * dz);
            cum[j + 1] = local;
        }
        double before = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double up = __shfl_up(before, off);
            if (lane >= off) before = before + up;
        }
        before = before - local;
        for (int j = lo; j < hi; ++j) cum[j + 1] = cum[j + 1] + before;
        if (lane == 0) cum[0] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double total = cum[nseg];
        for (int k = lane; k < nb; k += 64) {
            float x, y, z;
            if (k == nb - 1 || nseg == 0) {
                x = pts[3 * nseg + 0];
                y = pts[3 * nseg + 1];
                z = pts[3 * nseg + 2];
            } else {
                const double target = total * ((double)k / (double)(nb - 1));
                int a = 0, b = nseg;
                while (a < b) {
                    const int mid = (a + b) >> 1;
                    if (cum[mid + 1] <= target) a = mid + 1;
                    else b = mid;
                }
                const int j = min(a, nseg - 1);
                const double c0 = cum[j], c1 = cum[j + 1];
                const double den = c1 - c0;
                const double r = den > 0.0 ? (target - c0) / den : 0.0;
                const double ax = pts[3 * j + 0], ay = pts[3 * j + 1], az = pts[3 * j + 2];
                const double bx = pts[3 * j + 3], by = pts[3 * j + 4], bz = pts[3 * j + 5];
                x = (float)(ax + r * (bx - ax));
                y = (float)(ay + r * (by - ay));
                z = (float)(az + r * (bz - az));
            }
            res[3 * k + 0] = x;
            res[3 * k + 1] = y;
            res[3 * k + 2] = z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float *o = dirs + (size_t)row * (size_t)(nb - 1) * 3;
        for (int e = lane; e < 3 * (nb - 1); e += 64) o[e] = res[e + 3] - res[e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();        // the LDS rows are reused by the next streamline
    }
}

// OracleReward's sparse bonus (oracle_reward.py:84-93): term = 0 everywhere,
// bonus at the stopped rows whose score is > 0.5 (rows past n_scored were never
// scored: 0); reward += term.
__global__ __launch_bounds__(BLOCK) void k_oracle_bonus(
    const float *__restrict__ scores, int n_scored, const int *__restrict__ stop_list,
    int n_stopped, double bonus, double *__restrict__ term, double *__restrict__ reward) {
    const int q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= n_stopped) return;
    const int row = stop_list[2 * (size_t)q];
    const double t = (q < n_scored && scores[q] > 0.5f) ? bonus : 0.0;
    term[row] = t;
    reward[row] += t;
}
}  // namespace

extern "C" {

int ttl_resample_streamlines(const float *points, int64_t row_pitch, const int32_t *lengths32,
                             const int64_t *lengths64, int32_t n, int32_t max_len,
                             int32_t nb_points, float *out, void *hip_stream) {
    if (!points || !out || (!lengths32 && !lengths64) || n < 1 || max_len < 1 ||
        nb_points < 2 || row_pitch < 3LL * max_len)
        return fail(TTL_ERR_INVALID, "ttl_resample_streamlines: bad arguments");
    const size_t lds = (size_t)(BLOCK / 64) * (size_t)max_len * sizeof(double);
    if (lds > 160u * 1024u)
        return fail(TTL_ERR_INVALID, "ttl_resample_streamlines: %d points per row exceed the LDS",
                    max_len);
    HIP_TRY(hipFuncSetAttribute((const void *)k_resample,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int want = (n + (BLOCK / 64) - 1) / (BLOCK / 64);
    hipLaunchKernelGGL(k_resample, dim3(want < 4096 ? want : 4096), dim3(BLOCK), lds,
                       (hipStream_t)hip_stream, points, (long long)row_pitch, lengths32,
                       (const long long *)lengths64, n, max_len, nb_points, out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_oracle_segments(const float *history, int64_t row_pitch, const int32_t *ids,
                        int32_t id_stride, int32_t n, int32_t n_points, const float *lin,
                        int32_t nb_points, float *dirs_out, void *hip_stream) {
    if (!history || !dirs_out || n < 1 || n_points < 1 || nb_points < 2 ||
        row_pitch < 3LL * n_points || (ids && id_stride < 1))
        return fail(TTL_ERR_INVALID, "ttl_oracle_segments: bad arguments");
    const size_t per_wave =
        (((size_t)n_points * 20 + (size_t)nb_points * 12) + 15) & ~(size_t)15;
    const size_t lds = (size_t)(BLOCK / 64) * per_wave;
    if (lds > 160u * 1024u)
        return fail(TTL_ERR_INVALID, "ttl_oracle_segments: %d points per row exceed the LDS",
                    n_points);
    // (raised when a call needs more than any before it: the attribute is per device and
    // process, the call is on the training step's path)
    static thread_local size_t lds_allowed[64] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || lds > lds_allowed[dev]) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_oracle_segments,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) lds_allowed[dev] = lds;
    }
    Lin L{};
    if (lin)
        for (int k = 0; k < 9; ++k) L.m[k] = lin[k];
    const int want = (n + (BLOCK / 64) - 1) / (BLOCK / 64);
    hipLaunchKernelGGL(k_oracle_segments, dim3(want < 8192 ? want : 8192), dim3(BLOCK), lds,
                       (hipStream_t)hip_stream, history, (long long)row_pitch, ids, id_stride, n,
                       n_points, lin ? 1 : 0, L, nb_points, dirs_out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_oracle_bonus(const float *scores, int32_t n_scored, const int32_t *stop_list,
                     int32_t n_stopped, double bonus, int32_t n_active, double *term,
                     double *reward, void *hip_stream) {
    if (!term || !reward || n_stopped < 0 || n_scored < 0 ||
        (n_stopped > 0 && (!scores || !stop_list)) ||
        n_scored > n_stopped || n_stopped > n_active)
        return fail(TTL_ERR_INVALID, "ttl_oracle_bonus: bad arguments");
    hipStream_t s = (hipStream_t)hip_stream;
    HIP_TRY(hipMemsetAsync(term, 0, (size_t)n_active * sizeof(double), s));
    if (n_stopped > 0) {
        hipLaunchKernelGGL(k_oracle_bonus, dim3((n_stopped + BLOCK - 1) / BLOCK), dim3(BLOCK), 0,
                           s, scores, n_scored, stop_list, n_stopped, bonus, term, reward);
        HIP_TRY(hipGetLastError());
    }
    return TTL_OK;
}

}  // extern "C"
