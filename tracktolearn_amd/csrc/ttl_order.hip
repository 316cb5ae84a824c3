// ttl_order.hip -- processing order of the state gather rebuilt from the
// streamlines' current positions: a key kernel (8^3-voxel brick of the newest
// point of every active row, bricks in Morton order) + a rocPRIM radix sort
// (12-15 key bits) of (key, row) pairs
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
