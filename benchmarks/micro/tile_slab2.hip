// Prototype 5 (= prototype 4 with a two-tile look-ahead): half-height slabs
// (7x7 records x 2 z-slices), eight ring slots in the same LDS: a tile reads four
// half-slabs, a tile step fills two new ones, and the loaders may run two tiles
// ahead of the slowest consumer instead of one.
// Prototype 4: column-walking tiled state gather.  One persistent workgroup
// per CU walks columns of 4x4x4 tiles along z; LDS holds four z-slabs (7x7
// records in x-y around the column, 4 slices = one brick layer each) as a ring,
// so a tile step loads ONE new slab (37 KB) by LDS-DMA instead of the whole
// 7x7x7 region (66 KB); loader waves issue the DMAs, consumer waves take chunks
// of five streamlines round-robin and gather from LDS; hand-offs are LDS
// counters, no workgroup barrier.  Volume in the bricked record order of the
// library (TTL_SH_BRICK4).  Compared with the register-deduplicated direct
// gather (k_state_dd's arithmetic) on the same synthetic positions.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tile_slab2.hip -o tile_slab2
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


// record index of voxel (x, y, z) in the bricked order: 4x4x4 bricks of 64
// consecutive records, [X/4][Y/4][Z/4] bricks x [4][4][4] voxels
__host__ __device__ __forceinline__ unsigned bx_off(int x, int NB) { return (unsigned)((x >> 2) * NB * NB * 64 + (x & 3) * 16); }
__host__ __device__ __forceinline__ unsigned by_off(int y, int NB) { return (unsigned)((y >> 2) * NB * 64 + (y & 3) * 4); }
__host__ __device__ __forceinline__ unsigned bz_off(int z) { return (unsigned)((z >> 2) * 64 + (z & 3)); }

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
    const int NB = X >> 2;
    unsigned xo[4], yo[4], zo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xo[k] = bx_off(clipi(ix - 1 + k, X), NB) * 192u; yo[k] = by_off(clipi(iy - 1 + k, Y), NB) * 192u;
        zo[k] = bz_off(clipi(iz - 1 + k, Z)) * 192u;
    }
    const unsigned cb = sub * 16u;
    const int c = sub * 4;
#define FETCH_G(a, b, d) (*reinterpret_cast<const f4 *>(vol + (xo[a] + yo[b] + zo[d] + cb)))
    GATHER_BODY(FETCH_G)
    if (sub < K) { float *od = orow + 7 * C + 3 * sub; od[0] = hp.w; od[1] = hp.w; od[2] = hp.w; }
}

// ---------------------------------------------------------------------------
// B: column walk, four-slab LDS ring
// ---------------------------------------------------------------------------
constexpr int R = 7;                                   // region edge in x and y (4 + halo -1, +2)
constexpr int SLAB_UNITS = R * R * 2 * C4;             // 16-byte units of a half-slab (1176)
constexpr int SLAB_PIECES = (SLAB_UNITS + 63) / 64;    // 1-KiB LDS-DMA pieces (19, the last one 24 lanes)
constexpr int SLAB_BYTES = SLAB_UNITS * 16;            // packed: 18 816 B
constexpr int NSLOT = 8;
constexpr int MAXE = 400;                              // tiles per workgroup kept in LDS
constexpr int SPIN_CAP = 1 << 20;                      // bounded waits: a protocol bug must not hang the GPU
struct Entry { int txyz, start, cnt, first; };         // tile (tx | ty << 10 | tz << 20), its slot range, first consumer
struct Ctrl {
    int full[NSLOT];           // loaders that have landed entry k's slabs, cumulative per k & 3
    int done[NSLOT];           // consumers that have left entry k, cumulative per k & 3
    int tag[4][NSLOT];         // per loader: half-slab (column << 8 | h) held by each slot
    int last_use[4][NSLOT];    // per loader: last entry that reads each slot
};

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
// 16 bytes per lane from a per-lane global address to LDS at wave-uniform base + 16 * lane
__device__ __forceinline__ void glds16(const void *src, void *lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

template <int NWAVES, int NLOAD>
__global__ __launch_bounds__(NWAVES * 64) void k_slab(const char *__restrict__ vol, int X, int Y, int Z,
                                                      const float4 *__restrict__ srec,
                                                      const Entry *__restrict__ entries,
                                                      const int *__restrict__ wg_first, float rad,
                                                      float *__restrict__ out, int *__restrict__ err, int mode,
                                                      long long *__restrict__ dbg) {
    constexpr int NCONS = NWAVES - NLOAD;
    constexpr int MYP = (SLAB_PIECES + NLOAD - 1) / NLOAD;   // pieces per loader wave
    extern __shared__ __align__(1024) char lds[];
    Ctrl *ctrl = reinterpret_cast<Ctrl *>(lds + NSLOT * SLAB_BYTES);
    Entry *ent = reinterpret_cast<Entry *>(lds + NSLOT * SLAB_BYTES + sizeof(Ctrl));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e0 = wg_first[blockIdx.x], ne = min(wg_first[blockIdx.x + 1] - e0, MAXE);
    if (threadIdx.x < NSLOT) {
        ctrl->full[threadIdx.x] = 0; ctrl->done[threadIdx.x] = 0;
        for (int w = 0; w < 4; ++w) { ctrl->tag[w][threadIdx.x] = -1; ctrl->last_use[w][threadIdx.x] = -1; }
    }
    for (int i = threadIdx.x; i < ne; i += NWAVES * 64) ent[i] = entries[e0 + i];
    __syncthreads();
    if (threadIdx.x == 0) {     // which consumer takes chunk 0 of every tile: chunks go round-robin across tiles
        int first = 0;
        for (int i = 0; i < ne; ++i) { ent[i].first = first; first = (first + (ent[i].cnt + 4) / 5) % NCONS; }
    }
    __syncthreads();
    const int NB = X >> 2;
    if (wave < NLOAD) {
        // ---------------- loaders: pieces wave, wave + NLOAD, ... of every slab ----------------
        // byte offset of this lane's 16 bytes of piece p from brick (tx - 1, ty - 1, layer)
        // for columns whose 3 x 3 bricks lie inside the volume
        unsigned rel[MYP];
#pragma unroll
        for (int j = 0; j < MYP; ++j) {
            const int p = wave + j * NLOAD;
            const int e = min(p * 64 + lane, SLAB_UNITS - 1);
            const int r = e / C4, col = e - r * C4;
            const int zz = r & 1, t = r >> 1, cy = t % R, cx = t / R;
            rel[j] = (bx_off(cx + 3, NB) + by_off(cy + 3, NB) + (unsigned)zz) * 192u + (unsigned)col * 16u;
        }
        long long t_wait = 0, t_issue = 0, t_land = 0;
        for (int k = 0; k < ne; ++k) {
            const Entry en = ent[k];
            const int tx = en.txyz & 1023, ty = (en.txyz >> 10) & 1023, tz = en.txyz >> 20;
            const int column = tx | ty << 10;
            const bool inside = tx >= 1 && ty >= 1 && tx + 1 < NB && ty + 1 < NB;
            for (int b = max(2 * tz - 1, 0); b <= min(2 * tz + 2, (Z >> 1) - 1); ++b) {   // half-slab index
                const int slot = b & 7;
                const long long t0 = __builtin_readcyclecounter();
                if (ctrl->tag[wave][slot] != (column << 8 | b)) {
                    // the slot's readers: every consumer must have left the last entry that used it
                    const int j = ctrl->last_use[wave][slot];
                    if (j >= 0 && mode < 3) {
                        const int need = NCONS * (j / NSLOT + 1);
                        int spins = 0;
                        while (__hip_atomic_load(&ctrl->done[j & 7], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > SPIN_CAP) { *err = 1; return; }   // consumers time out too
                        }
                    }
                    const long long t1 = __builtin_readcyclecounter();
                    char *base = lds + slot * SLAB_BYTES;
                    if (mode == 1 || mode == 4 || mode == 7) {
                    } else if (inside) {
                        const char *origin = vol + (size_t)(bx_off(4 * (tx - 1), NB) + by_off(4 * (ty - 1), NB) + bz_off(2 * b)) * 192u;
#pragma unroll
                        for (int jj = 0; jj < MYP; ++jj) {
                            const int p = wave + jj * NLOAD;
                            // the last piece of a packed half-slab is 24 lanes wide
                            if (p < SLAB_PIECES && p * 64 + lane < SLAB_UNITS) glds16(origin + rel[jj], base + p * 1024);
                        }
                    } else {
#pragma unroll 1
                        for (int p = wave; p < SLAB_PIECES; p += NLOAD) {
                            const int e = min(p * 64 + lane, SLAB_UNITS - 1);
                            const int r = e / C4, col = e - r * C4;
                            const int zz = r & 1, t = r >> 1, cy = t % R, cx = t / R;
                            // the part of the region outside the volume is never read: any valid address
                            const unsigned v = bx_off(clipi(4 * tx - 1 + cx, X), NB) + by_off(clipi(4 * ty - 1 + cy, Y), NB) + bz_off(2 * b + zz);
                            if (p * 64 + lane < SLAB_UNITS) glds16(vol + (size_t)v * 192u + col * 16, base + p * 1024);
                        }
                    }
                    ctrl->tag[wave][slot] = column << 8 | b;
                    const long long t2 = __builtin_readcyclecounter();
                    t_wait += t1 - t0; t_issue += t2 - t1;
                }
                ctrl->last_use[wave][slot] = k;
            }
            const long long t2 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t_land += __builtin_readcyclecounter() - t2;
            if (lane == 0) __hip_atomic_fetch_add(&ctrl->full[k & 7], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
    for (int k = 0; k < ne; ++k) {
        const long long t0 = __builtin_readcyclecounter();
        int spins = 0;
        // every loader has landed its share of entry k's slabs
        while (mode < 3 && __hip_atomic_load(&ctrl->full[k & 7], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < NLOAD * (k / NSLOT + 1)) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_CAP) { *err = 2; return; }
        }
        const Entry en = ent[k];
        const long long t1 = __builtin_readcyclecounter();
        c_wait += t1 - t0;
        const int x0 = 4 * (en.txyz & 1023) - 1, y0 = 4 * ((en.txyz >> 10) & 1023) - 1;
        const int nch = (en.cnt + 4) / 5;
        int ch = cw - en.first;
        if (ch < 0) ch += NCONS;
        for (; ch < nch; ch += NCONS) {
            const int s = ch * 5 + grp;
            if (mode != 2 && grp < 5 && s < en.cnt) {
                float4 hp;
                if (mode >= 7) {   // timing only: no global load in the consumer
                    hp = float4{(float)x0 + 1.3f + 0.04f * (float)s, (float)y0 + 2.1f + 0.03f * (float)grp, 4.0f * (float)(en.txyz >> 20) + 0.5f + 0.05f * (float)s, __int_as_float(en.start + s)};
                } else {
                    hp = srec[en.start + s];
                }
                const int row = __float_as_int(hp.w);
                const float px = hp.x, py = hp.y, pz = hp.z;
                float *orow = out + (size_t)row * W;
                POINT_SETUP(px, py, pz)
                unsigned xo[4], yo[4], zo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gz = clipi(iz - 1 + q, Z);
                    xo[q] = (unsigned)(clipi(ix - 1 + q, X) - x0) * (R * 2 * 192u);
                    yo[q] = (unsigned)(clipi(iy - 1 + q, Y) - y0) * (2 * 192u);
                    zo[q] = (unsigned)((gz >> 1) & 7) * SLAB_BYTES + (unsigned)(gz & 1) * 192u;
                }
#define FETCH_L(a, b, d) (*reinterpret_cast<const f4 *>(lds + (xo[a] + yo[b] + zo[d] + cb)))
                GATHER_BODY(FETCH_L)
                if (sub < K) { float *od = orow + 7 * C + 3 * sub; od[0] = 0.25f; od[1] = 0.25f; od[2] = 0.25f; }
            }
        }
        // this wave's LDS reads of the entry are behind it (ds ops of one wave execute in order)
        if (lane == 0) __hip_atomic_fetch_add(&ctrl->done[k & 7], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
void run_slab(const char *d_vol, int D, const std::vector<float4> &pos, float rad, float *d_out,
              const std::vector<float> &ref, int n, int n_wg, int mode = 0) {
    const int nt1 = D / 4, nt = nt1 * nt1 * nt1;
    std::vector<int> tile(n), start(nt + 1, 0), slots(n);
    for (int i = 0; i < n; ++i) {
        auto cl = [&](float p) { int v = (int)fminf(fmaxf(floorf(p), -4.0f), (float)D + 4.0f); return std::min(std::max(v, 0), D - 1); };
        tile[i] = ((cl(pos[i].x) / 4) * nt1 + cl(pos[i].y) / 4) * nt1 + cl(pos[i].z) / 4;   // column-major, z fastest
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
        if (start[t + 1] > start[t]) entries.push_back(Entry{tx | ty << 10 | tz << 20, start[t], start[t + 1] - start[t], 0});
    }
    // workgroup ranges: equal shares of the streamlines, XCD x (= blockIdx % 8) gets the x-th eighth
    std::vector<int> wg_first(n_wg + 1, 0);
    int max_e = 0;
    {
        std::vector<int> range_first(n_wg + 1, (int)entries.size());
        int e = 0;
        for (int r = 0; r < n_wg; ++r) {
            const long long lo = (long long)n * r / n_wg;
            while (e < (int)entries.size() && entries[e].start < lo) ++e;
            range_first[r] = e;
        }
        std::vector<Entry> perm;
        for (int b = 0; b < n_wg; ++b) {
            const int r = (b % 8) * (n_wg / 8) + b / 8;
            wg_first[b] = (int)perm.size();
            for (int q = range_first[r]; q < range_first[r + 1]; ++q) perm.push_back(entries[q]);
            max_e = std::max(max_e, range_first[r + 1] - range_first[r]);
        }
        wg_first[n_wg] = (int)perm.size();
        entries.swap(perm);
    }
    if (max_e > MAXE) { printf("too many tiles per workgroup (%d)\n", max_e); return; }
    Entry *d_entries; int *d_first; float4 *d_srec; int *d_err;
    CK(hipMalloc(&d_err, 4)); CK(hipMemset(d_err, 0, 4));
    long long *d_dbg; CK(hipMalloc(&d_dbg, n_wg * 64)); CK(hipMemset(d_dbg, 0, n_wg * 64));
    CK(hipMalloc(&d_entries, entries.size() * sizeof(Entry))); CK(hipMalloc(&d_first, (n_wg + 1) * 4)); CK(hipMalloc(&d_srec, n * 16));
    CK(hipMemcpy(d_entries, entries.data(), entries.size() * sizeof(Entry), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_first, wg_first.data(), (n_wg + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_srec, srec.data(), n * 16, hipMemcpyHostToDevice));
    const size_t lds = (size_t)NSLOT * SLAB_BYTES + sizeof(Ctrl) + MAXE * sizeof(Entry);
    CK(hipFuncSetAttribute((const void *)k_slab<NWAVES, NLOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemset(d_out, 0, (size_t)n * W * 4));
    const float ms = timeit([&] { k_slab<NWAVES, NLOAD><<<n_wg, NWAVES * 64, lds>>>(d_vol, D, D, D, d_srec, d_entries, d_first, rad, d_out, d_err, mode, d_dbg); });
    CK(hipGetLastError());
    std::vector<float> got((size_t)n * W);
    CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
    const bool same = memcmp(got.data(), ref.data(), got.size() * 4) == 0;
    int h_err = 0; CK(hipMemcpy(&h_err, d_err, 4, hipMemcpyDeviceToHost));
    if (h_err) printf("WAIT TIMED OUT (code %d)\n", h_err);
    std::vector<long long> dbg(n_wg * 8); CK(hipMemcpy(dbg.data(), d_dbg, n_wg * 64, hipMemcpyDeviceToHost));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < n_wg; ++b) for (int q = 0; q < 6; ++q) acc[q] += (double)dbg[b * 8 + q] / n_wg;
    printf("slab %d waves (%d loaders), mode %d, %d workgroups, %zu tiles (max %d per workgroup), LDS %zu KB: %.4f ms  %s\n", NWAVES, NLOAD, mode,
           n_wg, entries.size(), max_e, lds / 1024, ms, same ? "bit-identical to direct" : "MISMATCH");
    printf("    mean cycles per workgroup: loader wait %.0f issue %.0f land %.0f | consumer wait %.0f work %.0f total %.0f\n", acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]);
    CK(hipFree(d_entries)); CK(hipFree(d_first)); CK(hipFree(d_srec)); CK(hipFree(d_err)); CK(hipFree(d_dbg));
}

int main() {
    const int D = 96, n = 246360, NB = D / 4;
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
    (void)NB;
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
    printf("direct, brick-sorted order, bricked volume: %.4f ms\n", ms);
    run_slab<16, 2>(d_vol, D, pos, rad, d_out, ref, n, 256);
    run_slab<16, 1>(d_vol, D, pos, rad, d_out, ref, n, 256);
    run_slab<16, 3>(d_vol, D, pos, rad, d_out, ref, n, 256);
    run_slab<12, 2>(d_vol, D, pos, rad, d_out, ref, n, 256);
    run_slab<16, 2>(d_vol, D, pos, rad, d_out, ref, n, 256, 1);
    run_slab<16, 2>(d_vol, D, pos, rad, d_out, ref, n, 256, 2);
    run_slab<16, 2>(d_vol, D, pos, rad, d_out, ref, n, 256, 3);   // no hand-offs at all: both sides free-running
    CK(hipDeviceSynchronize());
    return 0;
}
