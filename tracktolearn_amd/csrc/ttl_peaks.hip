// ttl_peaks.hip -- fODF peak extraction (SH -> SF on a hemisphere + local
// maxima) for gfx950; part of libttl_hip.so, C ABI in include/ttl_hip.h.
#include "ttl_internal.h"

namespace {
constexpr int BLOCK = TTL_BLOCK;

// ---------------------------------------------------------------------------
// k_peaks: fODF peaks of every voxel (TrackToLearn/environments/env.py:405-432:
// SH -> SF on a hemisphere, local maxima, relative threshold, minimum
// separation angle, at most npeaks directions scaled by value / first value).
// One wavefront per voxel, the SH->SF matrix B [C][V] staged once per
// workgroup in LDS; a lane owns the directions v = lane + 64 k.  The greedy
// peak selection runs wave-wide: repeated argmax (ties -> lowest vertex index)
// over the remaining local maxima.
// ---------------------------------------------------------------------------
constexpr int PEAKS_MAX_DIRS_PER_LANE = 12;   // V <= 768
constexpr int PEAKS_MAX_KEEP = 8;

__global__ __launch_bounds__(BLOCK) void k_peaks(
    const float *__restrict__ sh, long long n_vox, int C, const float *__restrict__ Bm,
    const float *__restrict__ verts, const int *__restrict__ nbr, int V, int deg,
    int npeaks, float rel_thr, float abs_thr, float cos_sep, int max_cand,
    float *__restrict__ out) {
    extern __shared__ __align__(16) float peaks_lds[];
    float *Bl = peaks_lds;                         // [C][V]
    float *sfw = peaks_lds + (size_t)C * V + (threadIdx.x >> 6) * V;   // this wave's SF
    for (int e = threadIdx.x; e < C * V; e += BLOCK) Bl[e] = Bm[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int waves = (BLOCK / 64) * gridDim.x;
    const int ndl = (V + 63) >> 6;
    for (long long vox = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); vox < n_vox;
         vox += waves) {
        const float *coef = sh + vox * C;
        float acc[PEAKS_MAX_DIRS_PER_LANE];
#pragma unroll
        for (int k = 0; k < PEAKS_MAX_DIRS_PER_LANE; ++k) acc[k] = 0.0f;
        float total = 0.0f;
        for (int c0 = 0; c0 < C; c0 += 64) {
            const float mine = (c0 + lane < C) ? coef[c0 + lane] : 0.0f;
            const int cn = min(64, C - c0);
            for (int c = 0; c < cn; ++c) {
                const float s = __shfl(mine, c);
                total = total + s;
                const float *brow = Bl + (size_t)(c0 + c) * V;
#pragma unroll
                for (int k = 0; k < PEAKS_MAX_DIRS_PER_LANE; ++k) {
                    const int v = lane + 64 * k;
                    if (k < ndl && v < V) acc[k] = acc[k] + s * brow[v];
                }
            }
        }
        float *o = out + vox * (3 * npeaks);
        if (total == 0.0f) {                       // no signal (env.py:418)
            for (int e = lane; e < 3 * npeaks; e += 64) o[e] = 0.0f;
            continue;
        }
        // SF below the absolute threshold counts as 0; publish to the wave
        float lo = 3.0e38f;
#pragma unroll
        for (int k = 0; k < PEAKS_MAX_DIRS_PER_LANE; ++k) {
            const int v = lane + 64 * k;
            if (k < ndl && v < V) {
                if (acc[k] < abs_thr) acc[k] = 0.0f;
                sfw[v] = acc[k];
                lo = fminf(lo, acc[k]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lo = fminf(lo, __shfl_xor(lo, off));
        const float odf_min = fmaxf(lo, 0.0f);
        // same-wave LDS hand-off (no other wave touches sfw): order the
        // lanes' writes before the neighbour reads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // local maxima of the hemisphere graph: >= every neighbour, > at least
        // one, positive.  cand = value or -1
        float cand[PEAKS_MAX_DIRS_PER_LANE];
#pragma unroll
        for (int k = 0; k < PEAKS_MAX_DIRS_PER_LANE; ++k) {
            const int v = lane + 64 * k;
            cand[k] = -1.0f;
            if (k < ndl && v < V) {
                const float x = acc[k];
                bool ge_all = true, gt_any = false;
                for (int d = 0; d < deg; ++d) {
                    const float y = sfw[nbr[v * deg + d]];
                    ge_all = ge_all && (x >= y);
                    gt_any = gt_any || (x > y);
                }
                if (ge_all && gt_any && x > 0.0f) cand[k] = x;
            }
        }
        float kx[PEAKS_MAX_KEEP], ky[PEAKS_MAX_KEEP], kz[PEAKS_MAX_KEEP], kval[PEAKS_MAX_KEEP];
        int n_keep = 0;
        float first_val = 1.0f, first_norm = 0.0f;
        for (int it = 0; it < max_cand && n_keep < npeaks; ++it) {
            // wave argmax over the remaining candidates, ties -> lowest index
            float bv = -1.0f;
            int bi = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < PEAKS_MAX_DIRS_PER_LANE; ++k) {
                const int v = lane + 64 * k;
                if (k < ndl && (cand[k] > bv || (cand[k] == bv && v < bi))) {
                    bv = cand[k];
                    bi = v;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(bv, off);
                const int oi = __shfl_xor(bi, off);
                if (ov > bv || (ov == bv && oi < bi)) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (!(bv > 0.0f)) break;               // no local maximum left
            // retire it
#pragma unroll
            for (int k = 0; k < PEAKS_MAX_DIRS_PER_LANE; ++k)
                if (lane + 64 * k == bi) cand[k] = -1.0f;
            const float norm = bv - odf_min;
            if (it == 0) {
                first_val = bv;
                first_norm = norm;
            }
            if (!(norm >= rel_thr * first_norm)) break;   // descending: the rest fail too
            const float dx = verts[bi * 3 + 0], dy = verts[bi * 3 + 1], dz = verts[bi * 3 + 2];
            bool ok = true;
            for (int q = 0; q < n_keep; ++q) {
                const float ca = fabsf((kx[q] * dx + ky[q] * dy) + kz[q] * dz);
                if (ca > cos_sep) ok = false;
            }
            if (ok) {
                kx[n_keep] = dx;
                ky[n_keep] = dy;
                kz[n_keep] = dz;
                kval[n_keep] = bv;
                ++n_keep;
            }
        }
        if (lane < npeaks) {
            float px = 0.0f, py = 0.0f, pz = 0.0f;
            for (int q = 0; q < PEAKS_MAX_KEEP; ++q)
                if (q == lane && q < n_keep) {
                    const float sc = kval[q] / first_val;
                    px = kx[q] * sc;
                    py = ky[q] * sc;
                    pz = kz[q] * sc;
                }
            o[3 * lane + 0] = px;
            o[3 * lane + 1] = py;
            o[3 * lane + 2] = pz;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();           // sfw is reused by the next voxel
    }
}

}  // namespace

extern "C" {

int ttl_peaks_from_sh(const float *sh, int64_t n_voxels, int32_t n_coef,
                      const float *sf_matrix, const float *vertices,
                      const int32_t *neighbours, int32_t n_vertices, int32_t degree,
                      int32_t npeaks, float relative_threshold, float absolute_threshold,
                      float min_separation_cos, int32_t max_candidates, float *peaks_out,
                      void *hip_stream) {
    if (!sh || !sf_matrix || !vertices || !neighbours || !peaks_out || n_voxels < 1 ||
        n_coef < 1 || degree < 1)
        return fail(TTL_ERR_INVALID, "ttl_peaks_from_sh: bad arguments");
    if (n_vertices < 1 || n_vertices > 64 * PEAKS_MAX_DIRS_PER_LANE)
        return fail(TTL_ERR_INVALID, "ttl_peaks_from_sh: %d vertices (1..%d supported)",
                    n_vertices, 64 * PEAKS_MAX_DIRS_PER_LANE);
    if (npeaks < 1 || npeaks > PEAKS_MAX_KEEP || max_candidates < npeaks)
        return fail(TTL_ERR_INVALID, "ttl_peaks_from_sh: npeaks=%d (1..%d), max_candidates=%d",
                    npeaks, PEAKS_MAX_KEEP, max_candidates);
    const size_t lds = ((size_t)n_coef * n_vertices + (size_t)(BLOCK / 64) * n_vertices) *
                       sizeof(float);
    if (lds > 160u * 1024u)
        return fail(TTL_ERR_INVALID, "ttl_peaks_from_sh: SH->SF matrix (%d x %d) exceeds the LDS",
                    n_coef, n_vertices);
    HIP_TRY(hipFuncSetAttribute((const void *)k_peaks,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long want = (n_voxels + (BLOCK / 64) - 1) / (BLOCK / 64);
    const int grid = (int)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(k_peaks, dim3(grid), dim3(BLOCK), lds, (hipStream_t)hip_stream, sh,
                       (long long)n_voxels, n_coef, sf_matrix, vertices, neighbours,
                       n_vertices, degree, npeaks, relative_threshold, absolute_threshold,
                       min_separation_cos, max_candidates, peaks_out);
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

}  // extern "C"
