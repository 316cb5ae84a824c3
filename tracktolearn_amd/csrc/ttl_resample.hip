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

}  // extern "C"
