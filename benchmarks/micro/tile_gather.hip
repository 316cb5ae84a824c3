// Prototype: brick-tiled state gather (tile + halo staged in LDS) against the
// register-deduplicated direct gather of k_state_dd, same arithmetic, on
// synthetic positions (246 k points uniform in a ball of a 96^3 x 48-float
// volume, radius 0.75 voxel).  Decides whether the tiled design is worth
// integrating.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tile_gather.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int BLOCK = 256;
constexpr int C = 45, C4 = 12, K = 4, W = 7 * C + 3 * K;
struct f4 { float x, y, z, w; };
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_a4 __attribute__((aligned(4)));

__device__ __forceinline__ f4 scale4(f4 a, float w) { return f4{a.x * w, a.y * w, a.z * w, a.w * w}; }
__device__ __forceinline__ f4 axpy4(f4 c, f4 a, float w) { return f4{c.x + a.x * w, c.y + a.y * w, c.z + a.z * w, c.w + a.w * w}; }
__device__ __forceinline__ f4 blend4(f4 v00, f4 v01, f4 v10, f4 v11, float a0, float a1, float b0, float b1) {
    f4 r = scale4(v00, a0 * b0); r = axpy4(r, v01, a0 * b1); r = axpy4(r, v10, a1 * b0); r = axpy4(r, v11, a1 * b1); return r;
}
__device__ __forceinline__ f4 lerp4(f4 lo, f4 hi, float d) { return axpy4(scale4(lo, 1.0f - d), hi, d); }
__device__ __forceinline__ f4 sel4(bool c, f4 a, f4 b) { return f4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w}; }
__device__ __forceinline__ int clipi(int v, int n) { return min(max(v, 0), n - 1); }
__device__ __forceinline__ float from_prev_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ void put4(float *o, f4 a, int c) {
    const f4 p{from_prev_lane(a.x), from_prev_lane(a.y), from_prev_lane(a.z), from_prev_lane(a.w)};
    const int back = (c + 3 < C) ? 0 : 4 - (C - c);
    v4f v{a.x, a.y, a.z, a.w};
    if (back == 1) v = v4f{p.w, a.x, a.y, a.z};
    if (back == 2) v = v4f{p.z, p.w, a.x, a.y};
    if (back == 3) v = v4f{p.y, p.z, p.w, a.x};
    *reinterpret_cast<v4f_a4 *>(o - back) = v;
}

// the 7-point gather of one streamline column; FETCH(xi, yi, zi) returns the
// float4 column of the record at slice indices (0..3 = f-1, f, f+1, f+2)
#define GATHER_BODY(FETCH)                                                              \
    const f4 zero{0.f, 0.f, 0.f, 0.f};                                                  \
    const f4 v000 = FETCH(1, 1, 1), v001 = FETCH(1, 1, 2), v010 = FETCH(1, 2, 1), v011 = FETCH(1, 2, 2); \
    const f4 v100 = FETCH(2, 1, 1), v101 = FETCH(2, 1, 2), v110 = FETCH(2, 2, 1), v111 = FETCH(2, 2, 2); \
    {                                                                                   \
        f4 b0 = zero, b3 = zero;                                                        \
        if (xdn) b0 = blend4(FETCH(0, 1, 1), FETCH(0, 1, 2), FETCH(0, 2, 1), FETCH(0, 2, 2), ey, dy, ez, dz); \
        if (xup) b3 = blend4(FETCH(3, 1, 1), FETCH(3, 1, 2), FETCH(3, 2, 1), FETCH(3, 2, 2), ey, dy, ez, dz); \
        const f4 b1 = blend4(v000, v001, v010, v011, ey, dy, ez, dz);                    \
        const f4 b2 = blend4(v100, v101, v110, v111, ey, dy, ez, dz);                    \
        put4(orow + 0 * C + c, lerp4(b1, b2, dx), c);                                    \
        put4(orow + 1 * C + c, lerp4(sel4(xup, b2, b1), sel4(xup, b3, b2), dxp), c);     \
        put4(orow + 4 * C + c, lerp4(sel4(xdn, b0, b1), sel4(xdn, b1, b2), dxm), c);     \
    }                                                                                   \
    {                                                                                   \
        f4 b0 = zero, b3 = zero;                                                        \
        if (ydn) b0 = blend4(FETCH(1, 0, 1), FETCH(1, 0, 2), FETCH(2, 0, 1), FETCH(2, 0, 2), ex, dx, ez, dz); \
        if (yup) b3 = blend4(FETCH(1, 3, 1), FETCH(1, 3, 2), FETCH(2, 3, 1), FETCH(2, 3, 2), ex, dx, ez, dz); \
        const f4 b1 = blend4(v000, v001, v100, v101, ex, dx, ez, dz);                    \
        const f4 b2 = blend4(v010, v011, v110, v111, ex, dx, ez, dz);                    \
        put4(orow + 2 * C + c, lerp4(sel4(yup, b2, b1), sel4(yup, b3, b2), dyp), c);     \
        put4(orow + 5 * C + c, lerp4(sel4(ydn, b0, b1), sel4(ydn, b1, b2), dym), c);     \
    }                                                                                   \
    {                                                                                   \
        f4 b0 = zero, b3 = zero;                                                        \
        if (zdn) b0 = blend4(FETCH(1, 1, 0), FETCH(1, 2, 0), FETCH(2, 1, 0), FETCH(2, 2, 0), ex, dx, ey, dy); \
        if (zup) b3 = blend4(FETCH(1, 1, 3), FETCH(1, 2, 3), FETCH(2, 1, 3), FETCH(2, 2, 3), ex, dx, ey, dy); \
        const f4 b1 = blend4(v000, v010, v100, v110, ex, dx, ey, dy);                    \
        const f4 b2 = blend4(v001, v011, v101, v111, ex, dx, ey, dy);                    \
        put4(orow + 3 * C + c, lerp4(sel4(zup, b2, b1), sel4(zup, b3, b2), dzp), c);     \
        put4(orow + 6 * C + c, lerp4(sel4(zdn, b0, b1), sel4(zdn, b1, b2), dzm), c);     \
    }

#define POINT_SETUP(px, py, pz)                                                          \
    const float cxp = px + rad, cxm = px + (-rad), cyp = py + rad, cym = py + (-rad);    \
    const float czp = pz + rad, czm = pz + (-rad);                                       \
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);                       \
    const float dx = px - fx, dy = py - fy, dz = pz - fz;                                \
    const float ex = 1.0f - dx, ey = 1.0f - dy, ez = 1.0f - dz;                          \
    const float fxp = floorf(cxp), fxm = floorf(cxm), fyp = floorf(cyp), fym = floorf(cym); \
    const float fzp = floorf(czp), fzm = floorf(czm);                                    \
    const float dxp = cxp - fxp, dxm = cxm - fxm, dyp = cyp - fyp, dym = cym - fym;      \
    const float dzp = czp - fzp, dzm = czm - fzm;                                        \
    const bool xup = fxp > fx, xdn = fxm < fx, yup = fyp > fy, ydn = fym < fy;           \
    const bool zup = fzp > fz, zdn = fzm < fz;                                           \
    const int ix = (int)fminf(fmaxf(fx, -4.0f), (float)X + 4.0f);                        \
    const int iy = (int)fminf(fmaxf(fy, -4.0f), (float)Y + 4.0f);                        \
    const int iz = (int)fminf(fmaxf(fz, -4.0f), (float)Z + 4.0f);

// A: direct gather in a given processing order (what k_state_dd does)
__global__ __launch_bounds__(BLOCK, 4) void k_direct(const char *__restrict__ vol, int X, int Y, int Z,
                                                   const float4 *__restrict__ pos, const int *__restrict__ proc,
                                                   int n, float rad, float *__restrict__ out) {
    int blk = blockIdx.x;
    { const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = blk & 7;
      blk = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blk >> 3); }
    const int lane = threadIdx.x & 63, grp = lane / 12, sub = lane - grp * 12;
    const int slot = blk * 20 + (threadIdx.x >> 6) * 5 + grp;
    if (grp >= 5 || slot >= n) return;
    const int row = proc[slot];
    const float4 hp = pos[row];
    const float px = hp.x, py = hp.y, pz = hp.z;
    float *orow = out + (size_t)row * W;
    POINT_SETUP(px, py, pz)
    const unsigned rec = 192u, sz = rec, sy = rec * Z, sx = sy * Y;
    unsigned xo[4], yo[4], zo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xo[k] = clipi(ix - 1 + k, X) * sx; yo[k] = clipi(iy - 1 + k, Y) * sy; zo[k] = clipi(iz - 1 + k, Z) * sz; }
    const unsigned cb = sub * 16u;
    const int c = sub * 4;
#define FETCH_G(a, b, d) (*reinterpret_cast<const f4 *>(vol + (xo[a] + yo[b] + zo[d] + cb)))
    GATHER_BODY(FETCH_G)
    if (sub < K) { float *od = orow + 7 * C + 3 * sub; od[0] = hp.w; od[1] = hp.w; od[2] = hp.w; }
}

// B: one workgroup per tile: tile + halo -> LDS, then the tile's streamlines
template <int TX, int TY, int TZ, int TB>
__global__ __launch_bounds__(TB) void k_tiled(const char *__restrict__ vol, int X, int Y, int Z,
                                                 const float4 *__restrict__ pos, const int *__restrict__ tile_start,
                                                 const int *__restrict__ slots, int ntx, int nty, int ntz,
                                                 float rad, float *__restrict__ out, int mode) {
    constexpr int RX = TX + 3, RY = TY + 3, RZ = TZ + 3;
    extern __shared__ __align__(16) char lds[];
    constexpr int MAXS = 256;   // streamline records prefetched per pass
    float4 *spos = reinterpret_cast<float4 *>(lds + (size_t)RX * RY * RZ * 192);
    const int t = blockIdx.x;
    const int start = tile_start[t], cnt = tile_start[t + 1] - start;
    if (cnt == 0) return;
    // the tile's slot records ({x, y, z, row bits}, written in slot order by
    // the binning pass): issued first, consumed after the region is staged
    float4 rec0 = float4{0.f, 0.f, 0.f, 0.f};
    if ((int)threadIdx.x < min(cnt, MAXS)) rec0 = pos[start + threadIdx.x];
    const int tz = t % ntz, ty = (t / ntz) % nty, tx = t / (ntz * nty);
    const int ox = tx * TX - 1, oy = ty * TY - 1, oz = tz * TZ - 1;
    // stage the region: consecutive threads take consecutive 16-B columns of
    // consecutive records along z
    constexpr int TOTAL = RX * RY * RZ * C4, NLD = (TOTAL + TB - 1) / TB, BATCH = 8;
#pragma unroll 1
    for (int b0 = 0; b0 < (mode == 1 ? 0 : NLD); b0 += BATCH) {
        f4 tmp[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            // unconditional (clamped) loads: a branch around a load makes the
            // compiler wait for it at the join, serialising the batch
            const int e = min((b0 + k) * TB + (int)threadIdx.x, TOTAL - 1);
            const int r = e / C4, col = e - r * C4;
            const int cz = r % RZ, cy = (r / RZ) % RY, cx = r / (RZ * RY);
            const size_t v = ((size_t)clipi(ox + cx, X) * Y + clipi(oy + cy, Y)) * Z + clipi(oz + cz, Z);
            tmp[k] = *reinterpret_cast<const f4 *>(vol + v * 192 + col * 16);
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int e = (b0 + k) * TB + threadIdx.x;
            if (b0 + k < NLD && e < TOTAL) *reinterpret_cast<f4 *>(lds + (size_t)e * 16) = tmp[k];
        }
    }
    const int lane = threadIdx.x & 63, grp = lane / 12, sub = lane - grp * 12;
    for (int base = 0; base < (mode == 2 ? 0 : cnt); base += MAXS) {
    const int m = min(cnt - base, MAXS);
    if (base) __syncthreads();
    if (base == 0) {
        if ((int)threadIdx.x < m) spos[threadIdx.x] = rec0;
    } else {
        for (int i = threadIdx.x; i < m; i += TB) spos[i] = pos[start + base + i];
    }
    __syncthreads();
    if (grp < 5)
    for (int s = (threadIdx.x >> 6) * 5 + grp; s < m; s += (TB / 64) * 5) {
        const float4 hp = spos[s];
        const int row = __float_as_int(hp.w);
        const float px = hp.x, py = hp.y, pz = hp.z;
        float *orow = out + (size_t)row * W;
        POINT_SETUP(px, py, pz)
        unsigned xo[4], yo[4], zo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xo[k] = (unsigned)(clipi(ix - 1 + k, X) - ox) * (RY * RZ * 192u);
            yo[k] = (unsigned)(clipi(iy - 1 + k, Y) - oy) * (RZ * 192u);
            zo[k] = (unsigned)(clipi(iz - 1 + k, Z) - oz) * 192u;
        }
        const unsigned cb = sub * 16u;
        const int c = sub * 4;
#define FETCH_L(a, b, d) (*reinterpret_cast<const f4 *>(lds + (xo[a] + yo[b] + zo[d] + cb)))
        GATHER_BODY(FETCH_L)
        if (sub < K) { float *od = orow + 7 * C + 3 * sub; od[0] = 0.25f; od[1] = 0.25f; od[2] = 0.25f; }
    }
    }
}



// C: persistent workgroups, one LDS region each, the NEXT tile's records
// prefetched into registers while the current tile is gathered and stored
template <int TX, int TY, int TZ, int TB, int WPC>
__global__ __launch_bounds__(TB, WPC) void k_tiled_persist(
    const char *__restrict__ vol, int X, int Y, int Z, const float4 *__restrict__ srec,
    const int *__restrict__ tile_start, const int *__restrict__ tiles, int n_tiles, int nty, int ntz,
    float rad, float *__restrict__ out) {
    constexpr int RX = TX + 3, RY = TY + 3, RZ = TZ + 3;
    constexpr int TOTAL = RX * RY * RZ * C4, NLD = (TOTAL + TB - 1) / TB;
    constexpr int MAXS = 256;
    extern __shared__ __align__(16) char lds[];
    float4 *spos = reinterpret_cast<float4 *>(lds + (size_t)RX * RY * RZ * 192);
    const int lane = threadIdx.x & 63, grp = lane / 12, sub = lane - grp * 12;
    f4 tmp[NLD];
    float4 rec0;
    int start = 0, cnt = 0, ox = 0, oy = 0, oz = 0;
    auto prefetch = [&](int j) {
        const int t = tiles[j];
        start = tile_start[t];
        cnt = tile_start[t + 1] - start;
        const int tz = t % ntz, ty = (t / ntz) % nty, tx = t / (ntz * nty);
        ox = tx * TX - 1; oy = ty * TY - 1; oz = tz * TZ - 1;
        rec0 = srec[start + min((int)threadIdx.x, cnt - 1)];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = min(k * TB + (int)threadIdx.x, TOTAL - 1);
            const int r = e / C4, col = e - r * C4;
            const int cz = r % RZ, cy = (r / RZ) % RY, cx = r / (RZ * RY);
            const size_t v = ((size_t)clipi(ox + cx, X) * Y + clipi(oy + cy, Y)) * Z + clipi(oz + cz, Z);
            tmp[k] = *reinterpret_cast<const f4 *>(vol + v * 192 + col * 16);
        }
    };
    int j = blockIdx.x;
    if (j >= n_tiles) return;
    prefetch(j);
    while (true) {
        // land the prefetched tile (the previous tile's readers are done: barrier)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = k * TB + threadIdx.x;
            if (e < TOTAL) *reinterpret_cast<f4 *>(lds + (size_t)e * 16) = tmp[k];
        }
        const int c_start = start, c_cnt = cnt, cox = ox, coy = oy, coz = oz;
        if ((int)threadIdx.x < min(c_cnt, MAXS)) spos[threadIdx.x] = rec0;
        __syncthreads();
        const int jn = j + gridDim.x;
        if (jn < n_tiles) prefetch(jn);      // in flight during the gather below
        for (int base = 0; base < c_cnt; base += MAXS) {
            const int m = min(c_cnt - base, MAXS);
            if (base) {
                __syncthreads();
                for (int i = threadIdx.x; i < m; i += TB) spos[i] = srec[c_start + base + i];
                __syncthreads();
            }
            if (grp < 5)
            for (int s = (threadIdx.x >> 6) * 5 + grp; s < m; s += (TB / 64) * 5) {
                const float4 hp = spos[s];
                const int row = __float_as_int(hp.w);
                const float px = hp.x, py = hp.y, pz = hp.z;
                float *orow = out + (size_t)row * W;
                POINT_SETUP(px, py, pz)
                unsigned xo[4], yo[4], zo[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xo[k] = (unsigned)(clipi(ix - 1 + k, X) - cox) * (RY * RZ * 192u);
                    yo[k] = (unsigned)(clipi(iy - 1 + k, Y) - coy) * (RZ * 192u);
                    zo[k] = (unsigned)(clipi(iz - 1 + k, Z) - coz) * 192u;
                }
                const unsigned cb = sub * 16u;
                const int c = sub * 4;
                GATHER_BODY(FETCH_L)
                if (sub < K) { float *od = orow + 7 * C + 3 * sub; od[0] = 0.25f; od[1] = 0.25f; od[2] = 0.25f; }
            }
        }
        if (jn >= n_tiles) break;
        j = jn;
    }
}

template <class F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int TX, int TY, int TZ, int TB>
void run_tiled(const char *d_vol, int D, const std::vector<float4> &pos, const float4 *d_pos, float rad,
               float *d_out, const std::vector<float> &ref, int n, int mode = 0) {
    const int ntx = (D + TX - 1) / TX, nty = (D + TY - 1) / TY, ntz = (D + TZ - 1) / TZ, nt = ntx * nty * ntz;
    std::vector<int> tile(n), start(nt + 1, 0), slots(n);
    for (int i = 0; i < n; ++i) {
        auto cl = [&](float p) { int v = (int)fminf(fmaxf(floorf(p), -4.0f), (float)D + 4.0f); return std::min(std::max(v, 0), D - 1); };
        tile[i] = ((cl(pos[i].x) / TX) * nty + cl(pos[i].y) / TY) * ntz + cl(pos[i].z) / TZ;
        start[tile[i] + 1]++;
    }
    int nonempty = 0, maxc = 0;
    for (int t = 0; t < nt; ++t) { nonempty += start[t + 1] > 0; maxc = std::max(maxc, start[t + 1]); start[t + 1] += start[t]; }
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < n; ++i) slots[fill[tile[i]]++] = i;
    std::vector<float4> srec(n);
    for (int j = 0; j < n; ++j) { srec[j] = pos[slots[j]]; int r = slots[j]; memcpy(&srec[j].w, &r, 4); }
    int *d_start, *d_slots; float4 *d_srec;
    CK(hipMalloc(&d_start, (nt + 1) * 4)); CK(hipMalloc(&d_slots, n * 4)); CK(hipMalloc(&d_srec, n * 16));
    CK(hipMemcpy(d_srec, srec.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_start, start.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_slots, slots.data(), n * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(TX + 3) * (TY + 3) * (TZ + 3) * 192 + 256 * 16;
    CK(hipFuncSetAttribute((const void *)k_tiled<TX, TY, TZ, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(d_out, 0, (size_t)n * W * 4));
    const float ms = timeit([&] { k_tiled<TX, TY, TZ, TB><<<nt, TB, lds>>>(d_vol, D, D, D, d_srec, d_start, d_slots, ntx, nty, ntz, rad, d_out, mode); });
    CK(hipGetLastError());
    std::vector<float> got((size_t)n * W);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(got.data(), ref.data(), got.size() * 4) == 0;
    printf("tiled mode %d %dx%dx%d block %d: %d tiles (%d non-empty, max %d/tile), LDS %zu KB: %.4f ms  %s\n", mode, TX, TY, TZ, TB, nt, nonempty, maxc,
           lds / 1024, ms, same ? "bit-identical to direct" : "MISMATCH");
    CK(hipFree(d_start)); CK(hipFree(d_slots)); CK(hipFree(d_srec));
}

template <int TX, int TY, int TZ, int TB, int WPC>
void run_persist(const char *d_vol, int D, const std::vector<float4> &pos, float rad,
                 float *d_out, const std::vector<float> &ref, int n) {
    const int ntx = (D + TX - 1) / TX, nty = (D + TY - 1) / TY, ntz = (D + TZ - 1) / TZ, nt = ntx * nty * ntz;
    std::vector<int> tile(n), start(nt + 1, 0), slots(n), tiles;
    for (int i = 0; i < n; ++i) {
        auto cl = [&](float p) { int v = (int)fminf(fmaxf(floorf(p), -4.0f), (float)D + 4.0f); return std::min(std::max(v, 0), D - 1); };
        tile[i] = ((cl(pos[i].x) / TX) * nty + cl(pos[i].y) / TY) * ntz + cl(pos[i].z) / TZ;
        start[tile[i] + 1]++;
    }
    for (int t = 0; t < nt; ++t) { if (start[t + 1] > 0) tiles.push_back(t); start[t + 1] += start[t]; }
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < n; ++i) slots[fill[tile[i]]++] = i;
    std::vector<float4> srec(n);
    for (int j = 0; j < n; ++j) { srec[j] = pos[slots[j]]; int r = slots[j]; memcpy(&srec[j].w, &r, 4); }
    int *d_start, *d_tiles; float4 *d_srec;
    CK(hipMalloc(&d_start, (nt + 1) * 4)); CK(hipMalloc(&d_tiles, tiles.size() * 4)); CK(hipMalloc(&d_srec, n * 16));
    CK(hipMemcpy(d_start, start.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_srec, srec.data(), n * 16, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(TX + 3) * (TY + 3) * (TZ + 3) * 192 + 256 * 16;
    CK(hipFuncSetAttribute((const void *)k_tiled_persist<TX, TY, TZ, TB, WPC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(d_out, 0, (size_t)n * W * 4));
    const int grid = std::min((int)tiles.size(), 256 * WPC);
    const float ms = timeit([&] { k_tiled_persist<TX, TY, TZ, TB, WPC><<<grid, TB, lds>>>(d_vol, D, D, D, d_srec, d_start, d_tiles, (int)tiles.size(), nty, ntz, rad, d_out); });
    CK(hipGetLastError());
    std::vector<float> got((size_t)n * W);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(got.data(), ref.data(), got.size() * 4) == 0;
    printf("persistent %dx%dx%d block %d x%d/CU: %zu tiles, grid %d, LDS %zu KB: %.4f ms  %s\n", TX, TY, TZ, TB, WPC, tiles.size(), grid,
           lds / 1024, ms, same ? "bit-identical to direct" : "MISMATCH");
    CK(hipFree(d_start)); CK(hipFree(d_tiles)); CK(hipFree(d_srec));
}

int main() {
    const int D = 96, n = 246360;
    const float rad = 0.75f;
    const size_t nvox = (size_t)D * D * D;
    std::vector<float> vol(nvox * 48);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 0.1f);
    for (auto &v : vol) v = nd(g);
    std::vector<float4> pos(n);
    std::uniform_real_distribution<float> ud(0.f, (float)D);
    for (int i = 0; i < n;) {
        float x = ud(g), y = ud(g), z = ud(g);
        const float c = (D - 1) / 2.0f, r = 0.42f * D;
        if ((x - c) * (x - c) + (y - c) * (y - c) + (z - c) * (z - c) < r * r) pos[i++] = float4{x, y, z, 0.25f};
    }
    // processing order of the direct kernel: sorted by 8^3 brick (its best case)
    std::vector<int> proc(n);
    for (int i = 0; i < n; ++i) proc[i] = i;
    auto key = [&](int i) { return (((int)pos[i].x >> 3) * 64 + ((int)pos[i].y >> 3)) * 64 + ((int)pos[i].z >> 3); };
    std::stable_sort(proc.begin(), proc.end(), [&](int a, int b) { return key(a) < key(b); });
    char *d_vol; float4 *d_pos; int *d_proc; float *d_out;
    CK(hipMalloc(&d_vol, vol.size() * 4)); CK(hipMalloc(&d_pos, n * 16)); CK(hipMalloc(&d_proc, n * 4));
    CK(hipMalloc(&d_out, (size_t)n * W * 4 + 64));
    CK(hipMemcpy(d_vol, vol.data(), vol.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pos, pos.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_proc, proc.data(), n * 4, hipMemcpyHostToDevice));
    const float ms = timeit([&] { k_direct<<<(n + 19) / 20, BLOCK>>>(d_vol, D, D, D, d_pos, d_proc, n, rad, d_out); });
    CK(hipGetLastError());
    std::vector<float> ref((size_t)n * W);
    CK(hipMemcpy(ref.data(), d_out, ref.size() * 4, hipMemcpyDeviceToHost));
    printf("direct, brick-sorted order: %.4f ms\n", ms);
    run_tiled<4, 4, 4, 256>(d_vol, D, pos, d_pos, rad, d_out, ref, n);
    run_tiled<4, 4, 4, 512>(d_vol, D, pos, d_pos, rad, d_out, ref, n);
    run_tiled<4, 4, 4, 512>(d_vol, D, pos, d_pos, rad, d_out, ref, n, 1);
    run_tiled<4, 4, 4, 512>(d_vol, D, pos, d_pos, rad, d_out, ref, n, 2);
    run_tiled<4, 4, 4, 1024>(d_vol, D, pos, d_pos, rad, d_out, ref, n);
    run_tiled<4, 4, 2, 256>(d_vol, D, pos, d_pos, rad, d_out, ref, n);
    run_tiled<4, 4, 8, 512>(d_vol, D, pos, d_pos, rad, d_out, ref, n);
    run_tiled<2, 4, 4, 256>(d_vol, D, pos, d_pos, rad, d_out, ref, n);
    run_persist<4, 4, 4, 1024, 1>(d_vol, D, pos, rad, d_out, ref, n);
    run_persist<4, 4, 4, 512, 2>(d_vol, D, pos, rad, d_out, ref, n);
    run_persist<4, 4, 4, 512, 1>(d_vol, D, pos, rad, d_out, ref, n);
    run_persist<4, 4, 8, 1024, 1>(d_vol, D, pos, rad, d_out, ref, n);
    CK(hipDeviceSynchronize());
    return 0;
}
