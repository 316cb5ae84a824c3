// Prototype 3: brick-tiled state gather with a loader wave and an LDS ring.
// One persistent 1024-thread workgroup per CU; wave 0 streams the 7x7x7-record
// region (4^3 tile + halo) of the workgroup's tiles into a two-slot LDS ring by
// LDS-DMA (global_load_lds_dwordx4, no VGPR staging, no workgroup barrier);
// the 15 other waves take chunks of five streamlines round-robin across the
// tiles and gather their 7-point stencils from LDS.  Hand-offs are LDS words
// (FULL sequence number per slot, DONE counter per slot).  Against the
// register-deduplicated direct gather (k_state_dd's arithmetic) on the same
// synthetic positions as tile_gather.hip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tile_ring.hip -o tile_ring
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

// ---------------------------------------------------------------------------
// B: loader wave + LDS ring
// ---------------------------------------------------------------------------
constexpr int T = 4, R = T + 3;                        // tile edge, region edge
constexpr int REGION_UNITS = R * R * R * C4;           // 16-byte units of a region (4116)
constexpr int PIECES = (REGION_UNITS + 63) / 64;       // 1-KiB LDS-DMA pieces (65)
constexpr int MAXS = 256;                              // streamlines per ring entry
constexpr int SLOT_BYTES = PIECES * 1024 + MAXS * 16;  // region image + slot records
constexpr int NSLOT = 2;
constexpr int SPIN_CAP = 1 << 20;                    // bounded waits: a protocol bug must not hang the GPU
struct Entry { int ox, oy, oz, start, cnt, pad0, pad1, pad2; };   // one ring entry: a tile (or part of a crowded one)
struct Ctrl {
    int full[NSLOT];       // sequence number of the entry the slot holds (k + 1)
    int done[NSLOT];       // consumer waves that have left the slot, cumulative
    int info[NSLOT][8];    // ox, oy, oz, cnt, first consumer of the entry's chunk 0
};

// 16 bytes per lane from a per-lane global address to LDS at wave-uniform base + 16 * lane
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__device__ __forceinline__ void glds16(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

template <int NWAVES, int NLOAD>
__global__ __launch_bounds__(NWAVES * 64) void k_ring(const char *__restrict__ vol, int X, int Y, int Z,
                                                      const float4 *__restrict__ srec,
                                                      const Entry *__restrict__ entries,
                                                      const int *__restrict__ wg_first, float rad,
                                                      float *__restrict__ out, int *__restrict__ err, int mode,
                                                      long long *__restrict__ dbg) {
    constexpr int NCONS = NWAVES - NLOAD;
    constexpr int MYP = (PIECES + NLOAD - 1) / NLOAD;   // pieces per loader wave
    extern __shared__ __align__(1024) char lds[];
    Ctrl *ctrl = reinterpret_cast<Ctrl *>(lds + NSLOT * SLOT_BYTES);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < NSLOT) { ctrl->full[threadIdx.x] = 0; ctrl->done[threadIdx.x] = 0; }
    __syncthreads();
    const int e0 = wg_first[blockIdx.x], e1 = wg_first[blockIdx.x + 1];
    if (wave < NLOAD) {
        // ---------------- loaders: pieces wave, wave + NLOAD, ... of every entry ----------------
        // byte offset of this lane's 16 bytes of piece p from the region's origin
        // record, for regions that lie inside the volume (no clipping)
        unsigned rel[MYP];
#pragma unroll
        for (int j = 0; j < MYP; ++j) {
            const int p = wave + j * NLOAD;
            const int e = min(p * 64 + lane, REGION_UNITS - 1);
            const int r = e / C4, col = e - r * C4;
            const int cz = r % R, t = r / R, cy = t % R, cx = t / R;
            rel[j] = (unsigned)(((cx * Y + cy) * Z + cz) * 192 + col * 16);
        }
        int chunk_base = 0;
        long long t_wait = 0, t_issue = 0, t_land = 0;
        for (int k = 0; k <= e1 - e0; ++k) {
            const long long t0 = __builtin_readcyclecounter();
            const int slot = k & (NSLOT - 1);
            // the slot's previous tenant (entry k - NSLOT) must have been left by every consumer
            const int need = NCONS * (k / NSLOT);
            int spins = 0;
            while (__hip_atomic_load(&ctrl->done[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_CAP) {     // bail out: every wave leaves
                    ctrl->info[0][3] = ctrl->info[1][3] = -1;
                    __hip_atomic_store(&ctrl->full[0], 1 << 30, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&ctrl->full[1], 1 << 30, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    *err = 1;
                    return;
                }
            }
            if (k == e1 - e0) {   // terminator
                if (lane == 0) {
                    if (wave == 0) ctrl->info[slot][3] = -1;
                    __hip_atomic_fetch_add(&ctrl->full[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                break;
            }
            const Entry en = entries[e0 + k];
            const long long t1 = __builtin_readcyclecounter();
            char *base = lds + slot * SLOT_BYTES;
            const bool inside = mode != 1 && en.ox >= 0 && en.oy >= 0 && en.oz >= 0 && en.ox + R <= X && en.oy + R <= Y && en.oz + R <= Z;
            if (inside) {
                const char *origin = vol + (((size_t)en.ox * Y + en.oy) * Z + en.oz) * 192;
#pragma unroll
                for (int j = 0; j < MYP; ++j) {
                    const int p = wave + j * NLOAD;
                    if (p < PIECES) glds16(origin + rel[j], base + p * 1024);
                }
            } else if (mode != 1) {
#pragma unroll 1
                for (int p = wave; p < PIECES; p += NLOAD) {
                    const int e = min(p * 64 + lane, REGION_UNITS - 1);
                    const int r = e / C4, col = e - r * C4;
                    const int cz = r % R, t = r / R, cy = t % R, cx = t / R;
                    const size_t v = ((size_t)clipi(en.ox + cx, X) * Y + clipi(en.oy + cy, Y)) * Z + clipi(en.oz + cz, Z);
                    glds16(vol + v * 192 + col * 16, base + p * 1024);
                }
            }
            // the entry's slot records, 64 per piece
            for (int p = wave; p * 64 < en.cnt; p += NLOAD) {
                const int i = min(p * 64 + lane, en.cnt - 1);
                glds16(srec + en.start + i, base + PIECES * 1024 + p * 1024);
            }
            const long long t2 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t3 = __builtin_readcyclecounter();
            t_wait += t1 - t0; t_issue += t2 - t1; t_land += t3 - t2;
            if (lane == 0) {
                if (wave == 0) {
                    ctrl->info[slot][0] = en.ox; ctrl->info[slot][1] = en.oy; ctrl->info[slot][2] = en.oz;
                    ctrl->info[slot][3] = en.cnt; ctrl->info[slot][4] = chunk_base;
                }
                __hip_atomic_fetch_add(&ctrl->full[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            chunk_base = (chunk_base + (en.cnt + 4) / 5) % NCONS;
        }
        if (wave == 0 && lane == 0) { dbg[blockIdx.x * 8 + 0] = t_wait; dbg[blockIdx.x * 8 + 1] = t_issue; dbg[blockIdx.x * 8 + 2] = t_land; }
        return;
    }
    // ---------------- consumers ----------------
    const int cw = wave - NLOAD;
    const int grp = lane / 12, sub = lane - grp * 12;
    const unsigned cb = sub * 16u;
    const int c = sub * 4;
    long long c_wait = 0, c_work = 0;
    const long long c_begin = __builtin_readcyclecounter();
    for (int k = 0;; ++k) {
        const long long t0 = __builtin_readcyclecounter();
        const int slot = k & (NSLOT - 1);
        int spins = 0;
        // every loader has landed its share of entry k in the slot
        while (__hip_atomic_load(&ctrl->full[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < NLOAD * (k / NSLOT + 1)) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_CAP) { *err = 2; return; }
        }
        const int cnt = ctrl->info[slot][3];
        const long long t1 = __builtin_readcyclecounter();
        c_wait += t1 - t0;
        if (cnt < 0) break;
        const int ox = ctrl->info[slot][0], oy = ctrl->info[slot][1], oz = ctrl->info[slot][2];
        const int first = ctrl->info[slot][4];
        const char *base = lds + slot * SLOT_BYTES;
        const float4 *spos = reinterpret_cast<const float4 *>(base + PIECES * 1024);
        const int nch = (cnt + 4) / 5;
        int ch = cw - first;
        if (ch < 0) ch += NCONS;
        for (; ch < nch; ch += NCONS) {
            const int s = ch * 5 + grp;
            if (mode != 2 && grp < 5 && s < cnt) {
                const float4 hp = spos[s];
                const int row = __float_as_int(hp.w);
                const float px = hp.x, py = hp.y, pz = hp.z;
                float *orow = out + (size_t)row * W;
                POINT_SETUP(px, py, pz)
                unsigned xo[4], yo[4], zo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xo[q] = (unsigned)(clipi(ix - 1 + q, X) - ox) * (R * R * 192u);
                    yo[q] = (unsigned)(clipi(iy - 1 + q, Y) - oy) * (R * 192u);
                    zo[q] = (unsigned)(clipi(iz - 1 + q, Z) - oz) * 192u;
                }
#define FETCH_L(a, b, d) (*reinterpret_cast<const f4 *>(base + (xo[a] + yo[b] + zo[d] + cb)))
                GATHER_BODY(FETCH_L)
                if (sub < K) { float *od = orow + 7 * C + 3 * sub; od[0] = 0.25f; od[1] = 0.25f; od[2] = 0.25f; }
            }
        }
        // this wave's LDS reads of the slot are behind it (ds ops of one wave execute in order)
        if (lane == 0) __hip_atomic_fetch_add(&ctrl->done[slot], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        c_work += __builtin_readcyclecounter() - t1;
    }
    if (cw == 0 && lane == 0) { dbg[blockIdx.x * 8 + 3] = c_wait; dbg[blockIdx.x * 8 + 4] = c_work; dbg[blockIdx.x * 8 + 5] = __builtin_readcyclecounter() - c_begin; }
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

template <int NWAVES, int NLOAD>
void run_ring(const char *d_vol, int D, const std::vector<float4> &pos, float rad, float *d_out,
              const std::vector<float> &ref, int n, int n_wg, int mode = 0) {
    const int nt1 = (D + T - 1) / T, nt = nt1 * nt1 * nt1;
    std::vector<int> tile(n), start(nt + 1, 0), slots(n);
    for (int i = 0; i < n; ++i) {
        auto cl = [&](float p) { int v = (int)fminf(fmaxf(floorf(p), -4.0f), (float)D + 4.0f); return std::min(std::max(v, 0), D - 1); };
        tile[i] = ((cl(pos[i].x) / T) * nt1 + cl(pos[i].y) / T) * nt1 + cl(pos[i].z) / T;
        start[tile[i] + 1]++;
    }
    for (int t = 0; t < nt; ++t) start[t + 1] += start[t];
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < n; ++i) slots[fill[tile[i]]++] = i;
    std::vector<float4> srec(n);
    for (int j = 0; j < n; ++j) { srec[j] = pos[slots[j]]; int r = slots[j]; memcpy(&srec[j].w, &r, 4); }
    std::vector<Entry> entries;
    for (int t = 0; t < nt; ++t) {
        const int tz = t % nt1, ty = (t / nt1) % nt1, tx = t / (nt1 * nt1);
        for (int b = start[t]; b < start[t + 1]; b += MAXS)
            entries.push_back(Entry{tx * T - 1, ty * T - 1, tz * T - 1, b, std::min(MAXS, start[t + 1] - b), 0, 0, 0});
    }
    // workgroup ranges: equal shares of the streamlines, XCD x (= blockIdx % 8) gets the x-th eighth
    std::vector<int> wg_first(n_wg + 1, 0);
    {
        std::vector<int> range_first(n_wg + 1, (int)entries.size());
        int e = 0;
        for (int r = 0; r < n_wg; ++r) {
            const long long lo = (long long)n * r / n_wg;
            while (e < (int)entries.size() && entries[e].start < lo) ++e;
            range_first[r] = e;
        }
        // ranges are consecutive in r; workgroup b takes range (b % 8) * (n_wg / 8) + b / 8
        std::vector<Entry> perm;
        for (int b = 0; b < n_wg; ++b) {
            const int r = (b % 8) * (n_wg / 8) + b / 8;
            wg_first[b] = (int)perm.size();
            for (int q = range_first[r]; q < range_first[r + 1]; ++q) perm.push_back(entries[q]);
        }
        wg_first[n_wg] = (int)perm.size();
        entries.swap(perm);
    }
    Entry *d_entries; int *d_first; float4 *d_srec; int *d_err;
    CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4));
    long long *d_dbg; CK(hipMalloc(&d_dbg, n_wg * 64)); CK(hipMemset(d_dbg, 0, n_wg * 64));
    CK(hipMalloc(&d_entries, entries.size() * sizeof(Entry))); CK(hipMalloc(&d_first, (n_wg + 1) * 4)); CK(hipMalloc(&d_srec, n * 16));
    CK(hipMemcpy(d_entries, entries.data(), entries.size() * sizeof(Entry), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_first, wg_first.data(), (n_wg + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_srec, srec.data(), n * 16, hipMemcpyHostToDevice));
    const size_t lds = (size_t)NSLOT * SLOT_BYTES + sizeof(Ctrl);
    CK(hipFuncSetAttribute((const void *)k_ring<NWAVES, NLOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(d_out, 0, (size_t)n * W * 4));
    const float ms = timeit([&] { k_ring<NWAVES, NLOAD><<<n_wg, NWAVES * 64, lds>>>(d_vol, D, D, D, d_srec, d_entries, d_first, rad, d_out, d_err, mode, d_dbg); });
    CK(hipGetLastError());
    std::vector<float> got((size_t)n * W);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(got.data(), ref.data(), got.size() * 4) == 0;
    int h_err = 0; CK(hipMemcpy(&h_err, d_err, 4, hipMemcpyDeviceToHost));
    if (h_err) printf("WAIT TIMED OUT (code %d)\n", h_err);
    std::vector<long long> dbg(n_wg * 8); CK(hipMemcpy(dbg.data(), d_dbg, n_wg * 64, hipMemcpyDeviceToHost));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < n_wg; ++b) for (int q = 0; q < 6; ++q) acc[q] += (double)dbg[b * 8 + q] / n_wg;
    printf("  mode %d; mean cycles per workgroup: loader wait %.0f issue %.0f land %.0f | consumer wait %.0f work %.0f total %.0f\n", mode, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]);
    printf("ring %d waves (%d loaders), %d workgroups, %zu entries, LDS %zu KB: %.4f ms  %s\n", NWAVES, NLOAD, n_wg, entries.size(), lds / 1024, ms,
           same ? "bit-identical to direct" : "MISMATCH");
    CK(hipFree(d_entries)); CK(hipFree(d_first)); CK(hipFree(d_srec));
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
    run_ring<16, 1>(d_vol, D, pos, rad, d_out, ref, n, 256);
    run_ring<16, 2>(d_vol, D, pos, rad, d_out, ref, n, 256);
    run_ring<16, 2>(d_vol, D, pos, rad, d_out, ref, n, 256, 1);
    run_ring<16, 2>(d_vol, D, pos, rad, d_out, ref, n, 256, 2);
    run_ring<8, 2>(d_vol, D, pos, rad, d_out, ref, n, 256);
    run_ring<8, 2>(d_vol, D, pos, rad, d_out, ref, n, 256, 1);
    run_ring<8, 2>(d_vol, D, pos, rad, d_out, ref, n, 256, 2);
    CK(hipDeviceSynchronize());
    return 0;
}
