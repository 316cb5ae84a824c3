// ttl_state.hip -- the state gather of the environment step
// (TrackToLearn/environments/env.py:504-565, _format_state) for gfx950:
// k_state_dd (register-deduplicated 7-point stencil, the dominant kernel of a
// step) and k_state (all 56 corner fetches; radius >= 1 voxel or volumes of
// 4 GiB and more).  Part of libttl_hip.so.
#include "ttl_internal.h"

// The state rows are compared with the reference at 1e-5, not bit for bit (the
// stopping decisions and positions, which are, live in ttl_hip.hip): let the
// blends contract to FMAs here although the library is built with
// -ffp-contract=off.  The direction block is plain float32 differences of
// stored positions and is not affected.
#pragma clang fp contract(fast)

namespace {
constexpr int BLOCK = TTL_BLOCK;

// ---------------------------------------------------------------------------
// k_state: LPS lanes per streamline, lane = one float4 column of the padded
// voxel record, so a group reads each 16B-aligned voxel record as one
// contiguous coef_pitch*4-byte segment.  7 points x 8 corners accumulate in
// registers (no cross-lane traffic); the previous-direction block is written
// by the same lanes.
// ---------------------------------------------------------------------------
template <int LPS>
__global__ __launch_bounds__(BLOCK) void k_state(
    EnvParams P, const int *__restrict__ idx, const int *__restrict__ row_dest,
    const int *__restrict__ proc, int n_rows, int L, float *__restrict__ out,
    long long pitch) {
    constexpr int GPW = 64 / LPS;              // streamlines per wave
    constexpr int ROWS = (BLOCK / 64) * GPW;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPS;
    const int slot = blockIdx.x * ROWS + (threadIdx.x >> 6) * GPW + grp;
    const int sub = lane - grp * LPS;
    if (grp >= GPW || slot >= n_rows) return;
    const int row = proc ? proc[slot] : slot;
    if (row < 0) return;        // a hole of an uncompacted processing order
    const int g = idx ? idx[row] : row;
    const int r = row_dest ? row_dest[row] : row;
    const float *h = P.hist + (size_t)g * (size_t)(P.max_nb_steps + 1) * 3;
    const float px = h[(L - 1) * 3 + 0];
    const float py = h[(L - 1) * 3 + 1];
    const float pz = h[(L - 1) * 3 + 2];
    float *orow = out + (size_t)r * (size_t)pitch;
    const int C = P.n_coef;
    const int C4 = P.coef_pitch >> 2;
    const int X = P.sh_dim[0], Y = P.sh_dim[1], Z = P.sh_dim[2];
    const float4 *vol = reinterpret_cast<const float4 *>(P.sh);

    for (int c4 = sub; c4 < C4; c4 += LPS) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            // neighbourhood point: [0, +x, +y, +z, -x, -y, -z] * radius
            const float ox = (k == 1) ? P.radius : (k == 4) ? -P.radius : 0.0f;
            const float oy = (k == 2) ? P.radius : (k == 5) ? -P.radius : 0.0f;
            const float oz = (k == 3) ? P.radius : (k == 6) ? -P.radius : 0.0f;
            float x = px + ox, y = py + oy, z = pz + oz;
            if (P.sh_shift != 0.0f) {
                x += P.sh_shift;
                y += P.sh_shift;
                z += P.sh_shift;
            }
            const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
            const float dx = x - fx, dy = y - fy, dz = z - fz;
            // clip the corner indices, not the weights (edge replication);
            // the float clamp also tames NaN / huge coordinates
            const int ix0 = (int)fminf(fmaxf(fx, -1.0f), (float)X);
            const int iy0 = (int)fminf(fmaxf(fy, -1.0f), (float)Y);
            const int iz0 = (int)fminf(fmaxf(fz, -1.0f), (float)Z);
            const int xa = min(max(ix0, 0), X - 1), xb = min(max(ix0 + 1, 0), X - 1);
            const int ya = min(max(iy0, 0), Y - 1), yb = min(max(iy0 + 1, 0), Y - 1);
            const int za = min(max(iz0, 0), Z - 1), zb = min(max(iz0 + 1, 0), Z - 1);
            const float ex = 1.0f - dx, ey = 1.0f - dy, ez = 1.0f - dz;
            const size_t ra = (size_t)vox_x(P, xa) + vox_y(P, ya), rb = (size_t)vox_x(P, xa) + vox_y(P, yb);
            const size_t rc = (size_t)vox_x(P, xb) + vox_y(P, ya), rd = (size_t)vox_x(P, xb) + vox_y(P, yb);
            // corner order 000,001,010,011,100,101,110,111 (x,y,z bits)
            const size_t qa = vox_z(P, za), qb = vox_z(P, zb);
            const float4 v0 = vol[(ra + qa) * C4 + c4];
            const float4 v1 = vol[(ra + qb) * C4 + c4];
            const float4 v2 = vol[(rb + qa) * C4 + c4];
            const float4 v3 = vol[(rb + qb) * C4 + c4];
            const float4 v4 = vol[(rc + qa) * C4 + c4];
            const float4 v5 = vol[(rc + qb) * C4 + c4];
            const float4 v6 = vol[(rd + qa) * C4 + c4];
            const float4 v7 = vol[(rd + qb) * C4 + c4];
            const float w0 = (ex * ey) * ez, w1 = (ex * ey) * dz;
            const float w2 = (ex * dy) * ez, w3 = (ex * dy) * dz;
            const float w4 = (dx * ey) * ez, w5 = (dx * ey) * dz;
            const float w6 = (dx * dy) * ez, w7 = (dx * dy) * dz;
            float4 a;
#define TTL_ACC(comp)                                                        \
    a.comp = v0.comp * w0;                                                   \
    a.comp = a.comp + v1.comp * w1;                                          \
    a.comp = a.comp + v2.comp * w2;                                          \
    a.comp = a.comp + v3.comp * w3;                                          \
    a.comp = a.comp + v4.comp * w4;                                          \
    a.comp = a.comp + v5.comp * w5;                                          \
    a.comp = a.comp + v6.comp * w6;                                          \
    a.comp = a.comp + v7.comp * w7;
            TTL_ACC(x) TTL_ACC(y) TTL_ACC(z) TTL_ACC(w)
#undef TTL_ACC
            const int c = c4 * 4;
            float *o = orow + k * C + c;
            if (c + 0 < C) o[0] = a.x;
            if (c + 1 < C) o[1] = a.y;
            if (c + 2 < C) o[2] = a.z;
            if (c + 3 < C) o[3] = a.w;
        }
    }
    // previous directions, most recent first, zero padded (np.diff of the
    // stored float32 positions)
    // previous directions, most recent first, zero padded (np.diff of the
    // stored float32 positions): one whole segment (3 floats from 6
    // contiguous ones) per lane and iteration
    float *od = orow + 7 * C;
    const int n_seg = L - 1;
    for (int j = sub; j < P.n_dirs; j += LPS) {
        float vx = 0.0f, vy = 0.0f, vz = 0.0f;
        if (j < n_seg) {
            const float *a = h + (L - 2 - j) * 3;   // points L-2-j and L-1-j
            const float ax = a[0], ay = a[1], az = a[2];
            const float bx = a[3], by = a[4], bz = a[5];
            vx = bx - ax;
            vy = by - ay;
            vz = bz - az;
        }
        od[3 * j + 0] = vx;
        od[3 * j + 1] = vy;
        od[3 * j + 2] = vz;
    }
}

// ---------------------------------------------------------------------------
// k_state_dd: same result as k_state (up to float32 summation order, ~1e-7)
// with the 56 corner fetches of the 7-point stencil deduplicated in registers.
// For a neighbourhood radius 0 < r < 1 voxel the shifted points (+-r along one
// axis) sit in the centre cell or in the adjacent one, so along each axis only
// the slices f-1 .. f+2 are touched and the other two axes keep the centre's
// 2x2 footprint and weights.  Per axis: B_s = bilinear blend (other two axes)
// of slice s; centre / plus / minus points are then 1-D lerps of two B_s.
// Loads: 8 centre-cell voxels + 4 per outer slice that is really needed
// (f-1 iff the minus point crosses down, f+2 iff the plus point crosses up):
// 20..32 records per streamline (26 on average) instead of 56.
// ---------------------------------------------------------------------------
struct f4 {
    float x, y, z, w;
};
// 16-byte load at a 32-bit byte offset from a wave-uniform base: lets the
// compiler use the SGPR-base + VGPR-offset addressing form (one VGPR per
// address instead of two)
__device__ __forceinline__ f4 ld4(const char *base, unsigned byte_off) {
    const float4 v = *reinterpret_cast<const float4 *>(base + byte_off);
    return f4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ f4 scale4(f4 a, float w) {
    return f4{a.x * w, a.y * w, a.z * w, a.w * w};
}
__device__ __forceinline__ f4 axpy4(f4 acc, f4 a, float w) {  // acc + a*w
    return f4{acc.x + a.x * w, acc.y + a.y * w, acc.z + a.z * w, acc.w + a.w * w};
}
// bilinear blend of 4 records with weights (a0,a1) x (b0,b1)
__device__ __forceinline__ f4 blend4(f4 v00, f4 v01, f4 v10, f4 v11, float a0,
                                     float a1, float b0, float b1) {
    f4 r = scale4(v00, a0 * b0);
    r = axpy4(r, v01, a0 * b1);
    r = axpy4(r, v10, a1 * b0);
    r = axpy4(r, v11, a1 * b1);
    return r;
}
__device__ __forceinline__ f4 lerp4(f4 lo, f4 hi, float d) {
    return axpy4(scale4(lo, 1.0f - d), hi, d);
}
__device__ __forceinline__ f4 sel4(bool c, f4 a, f4 b) {
    return f4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w};
}
__device__ __forceinline__ int clipi(int v, int n) { return min(max(v, 0), n - 1); }
// store the float4 column c..c+3 of one point's C coefficients (the last
// column of a padded record may be partial).  State rows are only 4-byte
// aligned (W = 7C + 3K floats), so the full column goes out as ONE dword-
// aligned 16-byte store (gfx950 global stores need dword alignment only)
// instead of four strided dword stores: the row-per-lane-group epilogue is
// store-issue bound otherwise.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_dword_aligned __attribute__((aligned(4)));
// the value lane-1 holds (DPP row_shr:1).  Only used between the last two
// lanes of one lane group, which never straddle a 16-lane DPP row for group
// sizes 4, 8, 12, 16; both lanes are active together.
__device__ __forceinline__ float from_prev_lane(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
}
// MERGE_TAIL: when C is not a multiple of 4 the last column holds 1..3 valid
// floats.  Instead of separate dword stores (one more store instruction per
// point with a handful of live lanes: the epilogue is bound by the number of
// store instructions the texture-address unit has to take), the last lane
// writes the 16 bytes that END at the row's last coefficient, borrowing the
// leading floats from its left neighbour; the overlap rewrites equal values.
// 16-byte dword-aligned store of a row fragment.  flavour 0: plain (write-back,
// the line stays in the XCD's L2); 1: sc1 (write-through, the line is dropped
// from L2: rows are written once and never read by this kernel, so they need
// not evict the voxel records the neighbours are about to gather); 2: nt;
// 3: sc0 sc1.  TTL_STORE_FLAVOUR selects (measurement knob).
__device__ __forceinline__ void store16(float *o, v4f v, int flavour) {
    if (flavour == 0) {
        *reinterpret_cast<v4f_dword_aligned *>(o) = v;
    } else if (flavour == 1) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(o), "v"(v) : "memory");
    } else if (flavour == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(o), "v"(v) : "memory");
    } else {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(o), "v"(v) : "memory");
    }
}

template <bool MERGE_TAIL>
__device__ __forceinline__ void put4(int flavour, float *o, f4 a, int c, int C) {
    if (MERGE_TAIL && (flavour & 8)) {
        // C mod 4 is wave-uniform: a scalar branch picks the one shifted vector
        // the last lane needs (3 DPP moves and 4 selects at most, none when C is
        // a multiple of 4) instead of building all three
        const int rem = C & 3;
        const bool last = c + 3 >= C;
        v4f v{a.x, a.y, a.z, a.w};
        int back = 0;
        if (rem == 1) {
            const float py = from_prev_lane(a.y), pz = from_prev_lane(a.z), pw = from_prev_lane(a.w);
            if (last) v = v4f{py, pz, pw, a.x};
            back = last ? 3 : 0;
        } else if (rem == 2) {
            const float pz = from_prev_lane(a.z), pw = from_prev_lane(a.w);
            if (last) v = v4f{pz, pw, a.x, a.y};
            back = last ? 2 : 0;
        } else if (rem == 3) {
            const float pw = from_prev_lane(a.w);
            if (last) v = v4f{pw, a.x, a.y, a.z};
            back = last ? 1 : 0;
        }
        store16(o - back, v, flavour & 7);
        return;
    }
    if (MERGE_TAIL) {
        const f4 p{from_prev_lane(a.x), from_prev_lane(a.y), from_prev_lane(a.z),
                   from_prev_lane(a.w)};
        const int back = (c + 3 < C) ? 0 : 4 - (C - c);   // 0 (full column), 1..3
        v4f v{a.x, a.y, a.z, a.w};
        if (back == 1) v = v4f{p.w, a.x, a.y, a.z};
        if (back == 2) v = v4f{p.z, p.w, a.x, a.y};
        if (back == 3) v = v4f{p.y, p.z, p.w, a.x};
        store16(o - back, v, flavour);
        return;
    }
    if (c + 3 < C) {
        store16(o, v4f{a.x, a.y, a.z, a.w}, flavour);
    } else {
        if (c + 0 < C) o[0] = a.x;
        if (c + 1 < C) o[1] = a.y;
        if (c + 2 < C) o[2] = a.z;
    }
}

// One state row of the register-deduplicated gather: the lane group's `sub`-th
// float4 column of the 7 stencil points + this lane's share of the direction
// block.  (px, py, pz) = newest point, h = the streamline's history, orow = the
// output row.  Shared by k_state_dd and the fused small-batch kernel.
template <int LPS, bool LOOP, bool MERGE_TAIL>
__device__ __forceinline__ void state_row_dd(const EnvParams &P, float px, float py,
                                             float pz, const float *__restrict__ h, int L,
                                             int sub, float *__restrict__ orow) {
    // this lane's first direction segment (np.diff of the stored float32
    // positions): a scattered sector of the history, i.e. the longest latency
    // of the wave -- fetched now, used after the gather
    const int n_seg = L - 1;
    float dvx = 0.0f, dvy = 0.0f, dvz = 0.0f;
    if (sub < P.n_dirs && sub < n_seg) {
        const float *a = h + (L - 2 - sub) * 3;   // points L-2-sub and L-1-sub
        const float ax = a[0], ay = a[1], az = a[2];
        const float bx = a[3], by = a[4], bz = a[5];
        dvx = bx - ax;
        dvy = by - ay;
        dvz = bz - az;
    }
    const int C = P.n_coef;
    const int C4 = P.coef_pitch >> 2;
    const int X = P.sh_dim[0], Y = P.sh_dim[1], Z = P.sh_dim[2];
    const char *vol = reinterpret_cast<const char *>(P.sh);
    const float rad = P.radius;

    // centre / plus / minus coordinates per axis (float32 adds as the
    // reference's `coords + neighbourhood`), floors and fractions
    float cxp = px + rad, cxm = px + (-rad);
    float cyp = py + rad, cym = py + (-rad);
    float czp = pz + rad, czm = pz + (-rad);
    if (P.sh_shift != 0.0f) {
        px += P.sh_shift; py += P.sh_shift; pz += P.sh_shift;
        cxp += P.sh_shift; cxm += P.sh_shift;
        cyp += P.sh_shift; cym += P.sh_shift;
        czp += P.sh_shift; czm += P.sh_shift;
    }
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const float dx = px - fx, dy = py - fy, dz = pz - fz;
    const float ex = 1.0f - dx, ey = 1.0f - dy, ez = 1.0f - dz;
    const float fxp = floorf(cxp), fxm = floorf(cxm);
    const float fyp = floorf(cyp), fym = floorf(cym);
    const float fzp = floorf(czp), fzm = floorf(czm);
    const float dxp = cxp - fxp, dxm = cxm - fxm;
    const float dyp = cyp - fyp, dym = cym - fym;
    const float dzp = czp - fzp, dzm = czm - fzm;
    // does the plus point sit in the next cell / the minus point in the
    // previous one?  (0 < r < 1 guarantees one cell at most)
    const bool xup = fxp > fx, xdn = fxm < fx;
    const bool yup = fyp > fy, ydn = fym < fy;
    const bool zup = fzp > fz, zdn = fzm < fz;
    // the float clamp tames NaN / huge coordinates before the int conversion
    const int ix = (int)fminf(fmaxf(fx, -4.0f), (float)X + 4.0f);
    const int iy = (int)fminf(fmaxf(fy, -4.0f), (float)Y + 4.0f);
    const int iz = (int)fminf(fmaxf(fz, -4.0f), (float)Z + 4.0f);
    // byte offset of a voxel record = ox[.] + oy[.] + oz[.] (32-bit; the host
    // guarantees the volume < 4 GiB), slices f-1, f, f+1, f+2 clipped per axis
    const unsigned rec = (unsigned)C4 * 16u;
    const unsigned x0 = vox_x(P, clipi(ix - 1, X)) * rec, x1 = vox_x(P, clipi(ix, X)) * rec,
                   x2 = vox_x(P, clipi(ix + 1, X)) * rec, x3 = vox_x(P, clipi(ix + 2, X)) * rec;
    const unsigned y0 = vox_y(P, clipi(iy - 1, Y)) * rec, y1 = vox_y(P, clipi(iy, Y)) * rec,
                   y2 = vox_y(P, clipi(iy + 1, Y)) * rec, y3 = vox_y(P, clipi(iy + 2, Y)) * rec;
    const unsigned z0 = vox_z(P, clipi(iz - 1, Z)) * rec, z1 = vox_z(P, clipi(iz, Z)) * rec,
                   z2 = vox_z(P, clipi(iz + 1, Z)) * rec, z3 = vox_z(P, clipi(iz + 2, Z)) * rec;
#define TTL_VOX(xo, yo, zo) ((xo) + (yo) + (zo))
    // one float4 column per lane when the record fits the lane group (LOOP =
    // false, the usual case: nothing is hoisted and kept live across columns)
    for (int c4 = sub; c4 < C4; c4 += LPS) {
        const unsigned cb = (unsigned)c4 * 16u;
        // centre cell
        const f4 v000 = ld4(vol, TTL_VOX(x1, y1, z1) + cb), v001 = ld4(vol, TTL_VOX(x1, y1, z2) + cb);
        const f4 v010 = ld4(vol, TTL_VOX(x1, y2, z1) + cb), v011 = ld4(vol, TTL_VOX(x1, y2, z2) + cb);
        const f4 v100 = ld4(vol, TTL_VOX(x2, y1, z1) + cb), v101 = ld4(vol, TTL_VOX(x2, y1, z2) + cb);
        const f4 v110 = ld4(vol, TTL_VOX(x2, y2, z1) + cb), v111 = ld4(vol, TTL_VOX(x2, y2, z2) + cb);
        const f4 zero{0.f, 0.f, 0.f, 0.f};
        const int c = c4 * 4;
        // outer slices are fetched (and reduced to one blended record at once)
        // only where a shifted point really reaches them
        // --- x axis: slices blended over (y, z) ---
        {
            f4 b0 = zero, b3 = zero;
            if (xdn)
                b0 = blend4(ld4(vol, TTL_VOX(x0, y1, z1) + cb), ld4(vol, TTL_VOX(x0, y1, z2) + cb),
                            ld4(vol, TTL_VOX(x0, y2, z1) + cb), ld4(vol, TTL_VOX(x0, y2, z2) + cb),
                            ey, dy, ez, dz);
            if (xup)
                b3 = blend4(ld4(vol, TTL_VOX(x3, y1, z1) + cb), ld4(vol, TTL_VOX(x3, y1, z2) + cb),
                            ld4(vol, TTL_VOX(x3, y2, z1) + cb), ld4(vol, TTL_VOX(x3, y2, z2) + cb),
                            ey, dy, ez, dz);
            const f4 b1 = blend4(v000, v001, v010, v011, ey, dy, ez, dz);
            const f4 b2 = blend4(v100, v101, v110, v111, ey, dy, ez, dz);
            put4<MERGE_TAIL>(P.store_flavour, orow + 0 * C + c, lerp4(b1, b2, dx), c, C);
            put4<MERGE_TAIL>(P.store_flavour, orow + 1 * C + c, lerp4(sel4(xup, b2, b1), sel4(xup, b3, b2), dxp), c, C);
            put4<MERGE_TAIL>(P.store_flavour, orow + 4 * C + c, lerp4(sel4(xdn, b0, b1), sel4(xdn, b1, b2), dxm), c, C);
        }
        // --- y axis: slices blended over (x, z) ---
        {
            f4 b0 = zero, b3 = zero;
            if (ydn)
                b0 = blend4(ld4(vol, TTL_VOX(x1, y0, z1) + cb), ld4(vol, TTL_VOX(x1, y0, z2) + cb),
                            ld4(vol, TTL_VOX(x2, y0, z1) + cb), ld4(vol, TTL_VOX(x2, y0, z2) + cb),
                            ex, dx, ez, dz);
            if (yup)
                b3 = blend4(ld4(vol, TTL_VOX(x1, y3, z1) + cb), ld4(vol, TTL_VOX(x1, y3, z2) + cb),
                            ld4(vol, TTL_VOX(x2, y3, z1) + cb), ld4(vol, TTL_VOX(x2, y3, z2) + cb),
                            ex, dx, ez, dz);
            const f4 b1 = blend4(v000, v001, v100, v101, ex, dx, ez, dz);
            const f4 b2 = blend4(v010, v011, v110, v111, ex, dx, ez, dz);
            put4<MERGE_TAIL>(P.store_flavour, orow + 2 * C + c, lerp4(sel4(yup, b2, b1), sel4(yup, b3, b2), dyp), c, C);
            put4<MERGE_TAIL>(P.store_flavour, orow + 5 * C + c, lerp4(sel4(ydn, b0, b1), sel4(ydn, b1, b2), dym), c, C);
        }
        // --- z axis: slices blended over (x, y) ---
        {
            f4 b0 = zero, b3 = zero;
            if (zdn)
                b0 = blend4(ld4(vol, TTL_VOX(x1, y1, z0) + cb), ld4(vol, TTL_VOX(x1, y2, z0) + cb),
                            ld4(vol, TTL_VOX(x2, y1, z0) + cb), ld4(vol, TTL_VOX(x2, y2, z0) + cb),
                            ex, dx, ey, dy);
            if (zup)
                b3 = blend4(ld4(vol, TTL_VOX(x1, y1, z3) + cb), ld4(vol, TTL_VOX(x1, y2, z3) + cb),
                            ld4(vol, TTL_VOX(x2, y1, z3) + cb), ld4(vol, TTL_VOX(x2, y2, z3) + cb),
                            ex, dx, ey, dy);
            const f4 b1 = blend4(v000, v010, v100, v110, ex, dx, ey, dy);
            const f4 b2 = blend4(v001, v011, v101, v111, ex, dx, ey, dy);
            put4<MERGE_TAIL>(P.store_flavour, orow + 3 * C + c, lerp4(sel4(zup, b2, b1), sel4(zup, b3, b2), dzp), c, C);
            put4<MERGE_TAIL>(P.store_flavour, orow + 6 * C + c, lerp4(sel4(zdn, b0, b1), sel4(zdn, b1, b2), dzm), c, C);
        }
        if (!LOOP) break;
    }
#undef TTL_VOX
    // previous directions, most recent first, zero padded (np.diff of the
    // stored float32 positions): one whole segment (3 floats from 6
    // contiguous ones) per lane and iteration
    float *od = orow + 7 * C;
    if (sub < P.n_dirs) {
        od[3 * sub + 0] = dvx;
        od[3 * sub + 1] = dvy;
        od[3 * sub + 2] = dvz;
    }
    for (int j = sub + LPS; j < P.n_dirs; j += LPS) {   // K > lane group size
        float vx = 0.0f, vy = 0.0f, vz = 0.0f;
        if (j < n_seg) {
            const float *a = h + (L - 2 - j) * 3;   // points L-2-j and L-1-j
            const float ax = a[0], ay = a[1], az = a[2];
            const float bx = a[3], by = a[4], bz = a[5];
            vx = bx - ax;
            vy = by - ay;
            vz = bz - az;
        }
        od[3 * j + 0] = vx;
        od[3 * j + 1] = vy;
        od[3 * j + 2] = vz;
    }
}

template <int LPS, int MINW, bool LOOP, bool MERGE_TAIL>
__global__ __launch_bounds__(BLOCK, MINW) void k_state_dd(
    EnvParams P, const int *__restrict__ idx, const int *__restrict__ row_dest,
    const int *__restrict__ proc, int n_rows, int L, float *__restrict__ out,
    long long pitch, int n_vblocks) {
    // LPS lanes per streamline, 64 / LPS streamlines per wave (LPS need not be
    // a power of two: with 12 float4 columns per record a wave serves 5
    // streamlines on 60 lanes instead of 4 on 48)
    constexpr int GPW = 64 / LPS;
    constexpr int ROWS = (BLOCK / 64) * GPW;
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPS;
    const int sub = lane - grp * LPS;
    if (grp >= GPW) return;
    // n_vblocks == gridDim.x: one workgroup per block of ROWS slots.  A launch
    // with fewer workgroups (TTL_GATHER_PERSIST_ROWS: one resident round) walks
    // the blocks with a stride of gridDim.x.
    for (int vb = blockIdx.x; vb < n_vblocks; vb += gridDim.x) {
        int blk = vb;
        if (proc && P.xcd_remap) {
            // workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8
            // share one; speed only, never correctness): give every XCD one
            // contiguous range of the spatially sorted processing order, so that
            // a voxel is fetched into ONE XCD's L2 instead of all eight.
            // Bijective for any grid size (cdna_hip_programming.md, T1).
            const int nwg = n_vblocks, q = nwg >> 3, rr = nwg & 7, xcd = blk & 7;
            if (P.xcd_rot == 0) {
                blk = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (blk >> 3);
            } else {
                // XCD x takes range (x + rot) & 7 of the order; a range has as many
                // workgroups as its XCD receives (q or q + 1), so the map stays
                // bijective for any grid size
                const int mine = (xcd + P.xcd_rot) & 7;
                int start = 0;
                for (int r = 0; r < mine; ++r)
                    start += q + ((((r - P.xcd_rot) & 7) < rr) ? 1 : 0);
                blk = start + (blk >> 3);
            }
        }
        const int slot = blk * ROWS + (threadIdx.x >> 6) * GPW + grp;
        if (slot >= n_rows) continue;
        int row, g, r;
        float px, py, pz;
        const bool slot_records = proc && idx && P.slot_rec;   // a step in processing order
        if (slot_records) {     // k_proc_scatter resolved row, idx[row], row_dest[row]
            const float4 hp = *reinterpret_cast<const float4 *>(P.slot_head + 4 * (size_t)slot);
            px = hp.x;
            py = hp.y;
            pz = hp.z;
            g = __float_as_int(hp.w);
            r = P.slot_dest[slot];
            row = 0;
            if (r < 0) continue;    // a hole of the uncompacted order (k_tail): nothing to gather
        } else {
            row = proc ? proc[slot] : slot;
            if (row < 0) continue;  // a hole of an uncompacted processing order
            g = idx ? idx[row] : row;
            r = row_dest ? row_dest[row] : row;
        }
        const float *h = P.hist + (size_t)g * (size_t)(P.max_nb_steps + 1) * 3;
        if (slot_records) {
        } else if (idx) {      // a step: k_advance left the new point in row order
            const float4 hp = *reinterpret_cast<const float4 *>(P.head + 4 * (size_t)row);
            px = hp.x;
            py = hp.y;
            pz = hp.z;
        } else {        // reset: the seed
            px = h[(L - 1) * 3 + 0];
            py = h[(L - 1) * 3 + 1];
            pz = h[(L - 1) * 3 + 2];
        }
        state_row_dd<LPS, LOOP, MERGE_TAIL>(P, px, py, pz, h, L, sub,
                                            out + (size_t)r * (size_t)pitch);
    }
}

// ---------------------------------------------------------------------------
// k_prefix_state: the small-batch step tail in ONE launch (batches of at most
// P.fuse_max_rows <= 256 * 256 streamlines without a processing order): what k_prefix and
// k_state_dd do in two.  Every workgroup scans the <= 64 per-block survivor
// counts k_advance left (tracking_env.py:192-195 stable compaction), resolves
// its own rows (continue_idx of the next step, row_dest, lengths of the
// streamlines that just stopped) and gathers their state rows.  Workgroup 0
// also hands the survivor count to the host: into device memory and, when the
// caller's pinned buffer is device-visible, straight into it, followed by the
// step's sequence number -- the host polls that word instead of waiting for a
// copy on a side stream.
// ---------------------------------------------------------------------------
template <int LPS, bool MERGE_TAIL, bool FR>
__device__ __forceinline__ void prefix_state_body(
    const EnvParams &P, const int *__restrict__ idx, int *__restrict__ idx_next, int n_active,
    int n_blocks, int order, int n_pts, float *__restrict__ out, long long pitch,
    int *__restrict__ host_word, int seq, int cur) {
    constexpr int GPW = 64 / LPS;
    constexpr int ROWS = (BLOCK / 64) * GPW;
    // exclusive prefix over the <= TTL_FUSE_MAX_BLOCKS per-block survivor counts
    // of k_advance: one wave, four counts per lane
    __shared__ int s_before[TTL_FUSE_MAX_BLOCKS + 1];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        int c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            c[k] = 4 * lane + k < n_blocks ? P.block_counts[4 * lane + k] : 0;
        const int local = (c[0] + c[1]) + (c[2] + c[3]);
        int v = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(v, off);
            if (lane >= off) v += t;
        }
        int run = v - local;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s_before[4 * lane + k] = run;
            run += c[k];
        }
        if (lane == 63) s_before[TTL_FUSE_MAX_BLOCKS] = v;
    }
    __syncthreads();
    const int total = s_before[TTL_FUSE_MAX_BLOCKS];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        P.counts[0] = total;
        P.counts[1] = n_active - total;
        if (FR) {
            // the next step's words; a step without active rows changes nothing
            // but the step counter
            int *live = P.counts + TTL_FR_LIVE;
            live[0] = total;
            live[1] = n_active > 0 ? n_pts : n_pts - 1;
            live[2] = n_active > 0 ? cur ^ 1 : cur;
            live[3] = seq;
        }
        if (host_word) {
            __hip_atomic_store(host_word + 0, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_word + 1, n_active - total, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_word + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const int grp = lane / LPS;
    const int row = blockIdx.x * ROWS + (threadIdx.x >> 6) * GPW + grp;
    const int sub = lane - grp * LPS;
    if (grp >= GPW || row >= n_active) return;
    const float4 hp = *reinterpret_cast<const float4 *>(P.head + 4 * (size_t)row);
    const int g = __float_as_int(hp.w);
    const bool stop = P.stop[row] != 0;
    const int pos = s_before[row >> 8] + P.rank[row];
    int dest = row;
    if (order == TTL_ORDER_PARTITION) dest = stop ? total + (row - pos) : pos;
    if (sub == 0) {
        if (!stop) idx_next[pos] = g;
        if (stop && order == TTL_ORDER_PARTITION) P.lengths[g] = n_pts;
        P.surv_pos[row] = stop ? -1 : pos;
        P.row_dest[row] = dest;
        if (stop)       // see k_prefix: the stopped rows in row order, for ttl_env_stopped
            *reinterpret_cast<int2 *>(P.stop_list + 2 * (size_t)(row - pos)) = int2{row, g};
    }
    const float *h = P.hist + (size_t)g * (size_t)(P.max_nb_steps + 1) * 3;
    state_row_dd<LPS, false, MERGE_TAIL>(P, hp.x, hp.y, hp.z, h, n_pts, sub,
                                         out + (size_t)dest * (size_t)pitch);
}

template <int LPS, bool MERGE_TAIL>
__global__ __launch_bounds__(BLOCK, 4) void k_prefix_state(
    EnvParams P, const int *__restrict__ idx, int *__restrict__ idx_next, int n_active,
    int n_blocks, int order, int n_pts, float *__restrict__ out, long long pitch,
    int *__restrict__ host_word, int seq) {
    prefix_state_body<LPS, MERGE_TAIL, false>(P, idx, idx_next, n_active, n_blocks, order,
                                              n_pts, out, pitch, host_word, seq, 0);
}

// the tail of a free-running step (see k_advance_fr): everything that changes
// from step to step is read from the snapshot k_advance_fr left
template <int LPS, bool MERGE_TAIL>
__global__ __launch_bounds__(BLOCK, 4) void k_prefix_state_fr(
    EnvParams P, int *__restrict__ idx_a, int *__restrict__ idx_b, int n_blocks,
    float *__restrict__ out, long long pitch, int *__restrict__ host_word) {
    const int *snap = P.counts + TTL_FR_SNAP;
    const int n_active = snap[0], L = snap[1], cur = snap[2], seq = snap[3] + 1;
    if (snap[4]) {      // k_advance_fr skipped the step: only the step counter moves
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            P.counts[TTL_FR_LIVE + 3] = seq;
            if (host_word)
                __hip_atomic_store(host_word + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    prefix_state_body<LPS, MERGE_TAIL, true>(P, cur ? idx_b : idx_a, cur ? idx_a : idx_b,
                                             n_active, n_blocks, TTL_ORDER_PARTITION, L + 1,
                                             out, pitch, host_word, seq, cur);
}

}  // namespace

// The fused small-batch tail (k_prefix_state) applies when the deduplicated
// gather does, the record fits one lane group of 4..16 lanes and the batch has
// at most 64 advance blocks.
bool ttl_detail_can_fuse_tail(const EnvParams &P, int n_active) {
    const int C4 = P.coef_pitch >> 2;
    const size_t vol_bytes = ttl_detail_sh_records(P) * P.coef_pitch * sizeof(float);
    return n_active <= P.fuse_max_rows && P.radius > 0.0f && P.radius < 1.0f &&
           vol_bytes < (1ull << 32) && C4 <= 16 && P.n_coef >= 4;
}

int ttl_detail_launch_fused_tail(const EnvParams &P, const int *idx, int *idx_next,
                                 int n_active, int order, int n_pts, float *out,
                                 int64_t pitch, int *host_word, int seq, hipStream_t s) {
    const int C4 = P.coef_pitch >> 2;
    const int n_blocks = (n_active + BLOCK - 1) / BLOCK;
#define TTL_LAUNCH_FUSED(LPS)                                                        \
    do {                                                                             \
        const int rows_per_block = (BLOCK / 64) * (64 / LPS);                        \
        const dim3 grid((n_active + rows_per_block - 1) / rows_per_block);           \
        hipLaunchKernelGGL((k_prefix_state<LPS, true>), grid, dim3(BLOCK), 0, s, P, idx, \
                           idx_next, n_active, n_blocks, order, n_pts, out,          \
                           (long long)pitch, host_word, seq);                        \
    } while (0)
    if (C4 <= 4) TTL_LAUNCH_FUSED(4);
    else if (C4 <= 8) TTL_LAUNCH_FUSED(8);
    else if (C4 <= 12) TTL_LAUNCH_FUSED(12);
    else TTL_LAUNCH_FUSED(16);
#undef TTL_LAUNCH_FUSED
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

int ttl_detail_launch_fused_tail_fr(const EnvParams &P, int *idx_a, int *idx_b, int n_cap,
                                    float *out, int64_t pitch, int *host_word,
                                    hipStream_t s) {
    const int C4 = P.coef_pitch >> 2;
    const int n_blocks = (n_cap + BLOCK - 1) / BLOCK;
#define TTL_LAUNCH_FUSED_FR(LPS)                                                     \
    do {                                                                             \
        const int rows_per_block = (BLOCK / 64) * (64 / LPS);                        \
        const dim3 grid((n_cap + rows_per_block - 1) / rows_per_block);              \
        hipLaunchKernelGGL((k_prefix_state_fr<LPS, true>), grid, dim3(BLOCK), 0, s, P, idx_a, \
                           idx_b, n_blocks, out, (long long)pitch, host_word);       \
    } while (0)
    if (C4 <= 4) TTL_LAUNCH_FUSED_FR(4);
    else if (C4 <= 8) TTL_LAUNCH_FUSED_FR(8);
    else if (C4 <= 12) TTL_LAUNCH_FUSED_FR(12);
    else TTL_LAUNCH_FUSED_FR(16);
#undef TTL_LAUNCH_FUSED_FR
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

// the register-deduplicated kernel (the one that reads the per-slot records of a
// processing order) needs the shifted points to stay within one cell of the
// centre -- 0 < radius < 1 voxel -- and 32-bit byte offsets into the volume
bool ttl_detail_state_dedupes(const EnvParams &P, int state_kernel) {
    const size_t vol_bytes = ttl_detail_sh_records(P) * P.coef_pitch * sizeof(float);
    return state_kernel != 0 && P.radius > 0.0f && P.radius < 1.0f && vol_bytes < (1ull << 32);
}

int ttl_detail_launch_state(const EnvParams &P, int state_kernel, const int *idx,
                            const int *row_dest, const int *proc, int n_rows, int L,
                            float *out, int64_t pitch, hipStream_t s) {
    const int C4 = P.coef_pitch >> 2;
    const bool dedupe = ttl_detail_state_dedupes(P, state_kernel);
#define TTL_LAUNCH_STATE(LPS)                                                 \
    do {                                                                      \
        const int rows_per_block = (BLOCK / 64) * (64 / LPS);                 \
        const int n_vb = (n_rows + rows_per_block - 1) / rows_per_block;      \
        /* TTL_GATHER_PERSIST_ROWS (experiment, off by default): batches of at */ \
        /* most that many rows that need more than one resident round (4       */ \
        /* workgroups per CU x 256 CUs) run as ONE round of workgroups that    */ \
        /* walk the blocks with a stride                                       */ \
        const int resident = 1024;                                            \
        const bool persist = dedupe && n_rows <= P.persist_rows && n_vb > resident; \
        const dim3 grid(persist ? resident : n_vb);                           \
        if (!dedupe)                                                          \
            hipLaunchKernelGGL((k_state<LPS>), grid, dim3(BLOCK), 0, s, P, \
                               idx, row_dest, proc, n_rows, L, out,           \
                               (long long)pitch);                             \
        else if (LPS < 32 && P.n_coef >= 4 && state_kernel != 3)    \
            hipLaunchKernelGGL((k_state_dd<LPS, 4, (LPS >= 32), (LPS < 32)>), grid, dim3(BLOCK), 0, s, \
                               P, idx, row_dest, proc, n_rows, L, out,   \
                               (long long)pitch, n_vb);                       \
        else                                                                  \
            hipLaunchKernelGGL((k_state_dd<LPS, (LPS >= 32 ? 2 : 4), (LPS >= 32), false>), grid, dim3(BLOCK), 0, s, \
                               P, idx, row_dest, proc, n_rows, L, out,   \
                               (long long)pitch, n_vb);                       \
    } while (0)
    if (C4 <= 4) TTL_LAUNCH_STATE(4);
    else if (C4 <= 8) TTL_LAUNCH_STATE(8);
    else if (C4 <= 12) TTL_LAUNCH_STATE(12);
    else if (C4 <= 16) TTL_LAUNCH_STATE(16);
    else TTL_LAUNCH_STATE(32);
#undef TTL_LAUNCH_STATE
    HIP_TRY(hipGetLastError());
    return TTL_OK;
}

