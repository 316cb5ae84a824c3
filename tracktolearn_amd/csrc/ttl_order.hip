// ttl_order.hip -- processing order of the state gather rebuilt from the
// streamlines' current positions: a key kernel (8^3-voxel brick of the newest
// point of every active row) + a counting sort over the bricks (rocPRIM radix
// sort for volumes of more than 16 384 bricks or the experimental key flavours)
// on workspace memory.  Scheduling only: results never depend on the order.
// Part of libttl_hip.so.
#include "ttl_internal.h"

#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace {
constexpr int BLOCK = TTL_BLOCK;

__global__ __launch_bounds__(BLOCK) void k_order_keys(const float *__restrict__ last2,
                                                      const int *__restrict__ idx, int n,
                                                      int nbx, int nby, int nbz, int fine,
                                                      unsigned *__restrict__ keys,
                                                      int *__restrict__ rows) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float *p = last2 + 8 * (size_t)idx[i] + 4;     // the newest point of the streamline
    const int nb[3] = {nbx, nby, nbz};
    unsigned b[3], m = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // brick of the voxel the point sits in; NaN and far-away points are
        // clamped into the brick grid
        const float vox = floorf(p[a] + 0.5f);
        const float v = floorf(vox * 0.125f);
        b[a] = (unsigned)fminf(fmaxf(v == v ? v : 0.0f, 0.0f), (float)(nb[a] - 1));
        // voxel inside the brick, Morton-interleaved (fine keys only)
        const unsigned l = (unsigned)fminf(fmaxf(vox == vox ? vox - 8.0f * v : 0.0f, 0.0f), 7.0f);
        m |= ((l & 1u) << (2 - a)) | ((l & 2u) << (4 - a)) | ((l & 4u) << (6 - a));
    }
    unsigned key;
    if (fine & 2) {
        // Morton code of the brick coordinates: consecutive keys are compact
        // 3-D blobs, so the eight XCD ranges of the order are octant-like
        // (small shared surface) instead of slabs along x
        key = 0;
#pragma unroll
        for (int bit = 0; bit < 10; ++bit)
            key |= (((b[0] >> bit) & 1u) << (3 * bit + 2)) | (((b[1] >> bit) & 1u) << (3 * bit + 1)) |
                   (((b[2] >> bit) & 1u) << (3 * bit));
    } else {
        // dense brick index: as few significant bits (= radix passes) as possible
        key = (b[0] * (unsigned)nby + b[1]) * (unsigned)nbz + b[2];
    }
    if (fine & 1) key = (key << 9) | m;
    keys[i] = key;
    rows[i] = i;
}

// ---------------------------------------------------------------------------
// Counting sort by brick (round 2): the keys are dense brick indices (a few
// thousand bins), so three short kernels replace the radix sort's dozen
// launches (0.1 ms -> ~0.02 ms per refresh at 200 k rows):
//   k_count   per-block histogram in LDS, flushed with one global atomic per
//             non-empty bin;
//   k_scan    one workgroup: exclusive scan of the bins -> cursors;
//   k_scatter the block histogram again, each element taking its rank inside
//             its bin from the LDS atomic and the block its share of the bin
//             from a global atomic on the cursor.
// The order INSIDE a bin follows the atomics and is not deterministic; it is a
// scheduling hint only (and k_proc_scatter re-sorts every block by voxel).
// ---------------------------------------------------------------------------
constexpr int SORT_ITEMS = 8;                       // elements per thread
constexpr int SORT_CHUNK = BLOCK * SORT_ITEMS;
constexpr int SORT_MAX_BINS = 16384;                // 64 KB of LDS counters

__global__ __launch_bounds__(BLOCK) void k_count(const unsigned *__restrict__ keys, int n,
                                                 int bins, unsigned *__restrict__ gcount) {
    extern __shared__ unsigned s_bin[];
    for (int b = threadIdx.x; b < bins; b += BLOCK) s_bin[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_CHUNK;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const int i = base + k * BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&s_bin[keys[i]], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += BLOCK) {
        const unsigned c = s_bin[b];
        if (c) atomicAdd(&gcount[b], c);
    }
}

__global__ __launch_bounds__(1024) void k_scan(const unsigned *__restrict__ gcount,
                                               unsigned *__restrict__ gcursor, int bins) {
    // one workgroup of 1024 threads, ceil(bins / 1024) consecutive bins each
    __shared__ unsigned s_part[1024 / 64];
    const int per = (bins + 1023) / 1024;
    const int lo = threadIdx.x * per;
    unsigned mine = 0;
    for (int b = lo; b < lo + per && b < bins; ++b) mine += gcount[b];
    // exclusive scan of `mine` over the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned v = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    if (lane == 63) s_part[wave] = v;
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wave; ++w) before += s_part[w];
    unsigned run = before + v - mine;
    for (int b = lo; b < lo + per && b < bins; ++b) {
        const unsigned c = gcount[b];
        gcursor[b] = run;
        run += c;
    }
}

__global__ __launch_bounds__(BLOCK) void k_scatter(const unsigned *__restrict__ keys,
                                                   const int *__restrict__ rows, int n, int bins,
                                                   unsigned *__restrict__ gcursor,
                                                   int *__restrict__ out) {
    extern __shared__ unsigned s_bin[];
    for (int b = threadIdx.x; b < bins; b += BLOCK) s_bin[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_CHUNK;
    unsigned key[SORT_ITEMS], rank[SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const int i = base + k * BLOCK + threadIdx.x;
        key[k] = 0;
        rank[k] = 0;
        if (i < n) {
            key[k] = keys[i];
            rank[k] = atomicAdd(&s_bin[key[k]], 1u);
        }
    }
    __syncthreads();
    // this block's share of every non-empty bin
    for (int b = threadIdx.x; b < bins; b += BLOCK) {
        const unsigned c = s_bin[b];
        if (c) s_bin[b] = atomicAdd(&gcursor[b], c);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        const int i = base + k * BLOCK + threadIdx.x;
        if (i < n) out[s_bin[key[k]] + rank[k]] = rows[i];
    }
}
}  // namespace

size_t ttl_detail_order_workspace_bytes(size_t n) {
    // keys in/out + rows in (rows out is the processing-order buffer itself),
    // plus rocPRIM's own scratch: depending on the path it picks, histograms
    // and look-back state or a second (key, value) buffer -- 8 B per row and a
    // few KiB measured; 4 MiB + 24 B per row leaves a wide margin (checked
    // against rocPRIM's own size query at run time)
    return 3 * ((n * 4 + 255) / 256 * 256) + (4u << 20) + n * 24;
}

int ttl_detail_refresh_order(const EnvParams &P, const int *idx, int n, char *ws,
                             size_t ws_bytes, int *order_out, hipStream_t s) {
    const size_t slab = ((size_t)n * 4 + 255) / 256 * 256;
    if (ws_bytes < 3 * slab) return fail(TTL_ERR_INVALID, "order refresh: workspace too small");
    unsigned *keys_in = reinterpret_cast<unsigned *>(ws);
    unsigned *keys_out = reinterpret_cast<unsigned *>(ws + slab);
    int *rows_in = reinterpret_cast<int *>(ws + 2 * slab);
    void *temp = ws + 3 * slab;
    const size_t temp_avail = ws_bytes - 3 * slab;
    int nb[3];
    unsigned long long bricks = 1;
    for (int a = 0; a < 3; ++a) {
        nb[a] = (P.sh_dim[a] + 7) / 8 + 1;
        if (nb[a] > 1024) nb[a] = 1024;
        bricks *= (unsigned long long)nb[a];
    }
    unsigned bits = 1;
    while ((1ull << bits) < bricks) ++bits;
    // TTL_ORDER_KEY: bit 0 = voxel-level keys (brick, then Morton code of the
    // voxel inside the brick), bit 1 = Morton order of the bricks themselves
    // instead of the dense x-major brick index
    int fine = 0;     // measured: neither bit pays once k_proc_scatter re-sorts locally
    if (const char *v = getenv("TTL_ORDER_KEY")) fine = atoi(v) & 3;
    if (fine & 2) {
        int mx = nb[0] > nb[1] ? nb[0] : nb[1];
        if (nb[2] > mx) mx = nb[2];
        unsigned per_axis = 1;
        while ((1 << per_axis) < mx) ++per_axis;
        bits = 3 * per_axis;
        if (bits > 23) return fail(TTL_ERR_INVALID, "order refresh: volume too large for Morton keys");
    }
    if (fine & 1) bits += 9;
    hipLaunchKernelGGL(k_order_keys, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, P.last2,
                       idx, n, nb[0], nb[1], nb[2], fine, keys_in, rows_in);
    HIP_TRY(hipGetLastError());
    // dense brick keys, few bins: counting sort (three short launches)
    int counting = 1;
    if (const char *v = getenv("TTL_ORDER_SORT")) counting = atoi(v) != 0;     // 0: rocPRIM
    if (counting && fine == 0 && bricks <= (unsigned long long)SORT_MAX_BINS &&
        temp_avail >= 2 * (size_t)bricks * sizeof(unsigned)) {
        const int bins = (int)bricks;
        unsigned *gcount = reinterpret_cast<unsigned *>(temp);
        unsigned *gcursor = gcount + bins;
        const int nblk = (n + SORT_CHUNK - 1) / SORT_CHUNK;
        const size_t lds = (size_t)bins * sizeof(unsigned);
        HIP_TRY(hipMemsetAsync(gcount, 0, (size_t)bins * sizeof(unsigned), s));
        hipLaunchKernelGGL(k_count, dim3(nblk), dim3(BLOCK), lds, s, keys_in, n, bins, gcount);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, gcount, gcursor, bins);
        hipLaunchKernelGGL(k_scatter, dim3(nblk), dim3(BLOCK), lds, s, keys_in, rows_in, n, bins,
                           gcursor, order_out);
        HIP_TRY(hipGetLastError());
        return TTL_OK;
    }
    size_t need = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, rows_in, order_out,
                                      (size_t)n, 0u, bits, s));
    if (need > temp_avail)
        return fail(TTL_ERR_INVALID, "order refresh: rocPRIM needs %zu B of scratch, %zu available",
                    need, temp_avail);
    HIP_TRY(rocprim::radix_sort_pairs(temp, need, keys_in, keys_out, rows_in, order_out,
                                      (size_t)n, 0u, bits, s));
    return TTL_OK;
}
