// Micro-benchmark: how fast can 246 k state rows of 327 floats be written,
// depending on the store pattern?  (decides whether k_state_dd's epilogue is
// worth restructuring).  hipcc --offload-arch=gfx950 -O3 store_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int BLOCK = 256;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef v4f v4f_a4 __attribute__((aligned(4)));

// A: what k_state_dd does now: 12 lanes / row, 7 blocks of C floats, one
// dword-aligned 16-B store per lane and block
__global__ __launch_bounds__(BLOCK) void k_cols(const int *dest, int n, float *out, long long pitch, int C, float v) {
    const int lane = threadIdx.x & 63, grp = lane / 12, sub = lane - grp * 12;
    const int slot = blockIdx.x * 20 + (threadIdx.x >> 6) * 5 + grp;
    if (grp >= 5 || slot >= n) return;
    float *o = out + (size_t)dest[slot] * pitch;
    const int c = sub * 4;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        float *p = o + j * C + c;
        if (c + 3 < C) *reinterpret_cast<v4f_a4 *>(p) = v4f{v, v + 1, v + 2, v + 3};
        else { if (c < C) p[0] = v; if (c + 1 < C) p[1] = v; if (c + 2 < C) p[2] = v; }
    }
    if (sub < 4) { float *p = o + 7 * C + 3 * sub; p[0] = v; p[1] = v; p[2] = v; }
}

// B: row-linear 16-B aligned chunks (needs pitch % 4 == 0), 12 lanes / row
__global__ __launch_bounds__(BLOCK) void k_linear12(const int *dest, int n, float *out, long long pitch, int W, float v) {
    const int lane = threadIdx.x & 63, grp = lane / 12, sub = lane - grp * 12;
    const int slot = blockIdx.x * 20 + (threadIdx.x >> 6) * 5 + grp;
    if (grp >= 5 || slot >= n) return;
    float *o = out + (size_t)dest[slot] * pitch;
    const int nq = (W + 3) >> 2;
    for (int q = sub; q < nq; q += 12) {
        if (q * 4 + 3 < W) *reinterpret_cast<v4f *>(o + q * 4) = v4f{v, v + 1, v + 2, v + 3};
        else for (int k = q * 4; k < W; ++k) o[k] = v;
    }
}

// E: pitch 327, per-row peel to 16-B alignment then aligned chunks, 12 lanes / row
__global__ __launch_bounds__(BLOCK) void k_peel12(const int *dest, int n, float *out, long long pitch, int W, float v) {
    const int lane = threadIdx.x & 63, grp = lane / 12, sub = lane - grp * 12;
    const int slot = blockIdx.x * 20 + (threadIdx.x >> 6) * 5 + grp;
    if (grp >= 5 || slot >= n) return;
    float *o = out + (size_t)dest[slot] * pitch;
    const int head = (int)((4 - (((size_t)o >> 2) & 3)) & 3);
    if (sub < head) o[sub] = v;
    const int nq = (W - head) >> 2;
    float *a = o + head;
    for (int q = sub; q < nq; q += 12) *reinterpret_cast<v4f *>(a + q * 4) = v4f{v, v + 1, v + 2, v + 3};
    const int done = head + nq * 4;
    if (sub < W - done) o[done + sub] = v;
}

// F: one wave per row, linear aligned chunks (pitch % 4 == 0)
__global__ __launch_bounds__(BLOCK) void k_linear64(const int *dest, int n, float *out, long long pitch, int W, float v) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= n) return;
    float *o = out + (size_t)dest[slot] * pitch;
    const int nq = (W + 3) >> 2;
    for (int q = lane; q < nq; q += 64) {
        if (q * 4 + 3 < W) *reinterpret_cast<v4f *>(o + q * 4) = v4f{v, v + 1, v + 2, v + 3};
        else for (int k = q * 4; k < W; ++k) o[k] = v;
    }
}

// C: plain linear fill
__global__ __launch_bounds__(BLOCK) void k_fill(float *out, size_t n4, float v) {
    size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n4) reinterpret_cast<v4f *>(out)[i] = v4f{v, v + 1, v + 2, v + 3};
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

int main() {
    const int n = 246360, C = 45, W = 327;
    float *out; CK(hipMalloc(&out, (size_t)n * 384 * 4 + 256));
    std::vector<int> ident(n), perm(n);
    for (int i = 0; i < n; ++i) ident[i] = perm[i] = i;
    std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
    int *d_ident, *d_perm;
    CK(hipMalloc(&d_ident, n * 4)); CK(hipMalloc(&d_perm, n * 4));
    CK(hipMemcpy(d_ident, ident.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice));
    const double bytes = (double)n * W * 4;
    const int g12 = (n + 19) / 20, g64 = (n + 3) / 4;
    auto rep = [&](const char *name, float ms) { printf("%-34s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6); };
    for (int pass = 0; pass < 2; ++pass) {
        const int *d = pass ? d_perm : d_ident;
        printf("--- rows %s ---\n", pass ? "randomly permuted" : "in order");
        rep("cols, pitch 327 (current)", timeit([&] { k_cols<<<g12, BLOCK>>>(d, n, out, 327, C, 1.f); }));
        rep("cols, pitch 328", timeit([&] { k_cols<<<g12, BLOCK>>>(d, n, out, 328, C, 1.f); }));
        rep("linear12 aligned, pitch 328", timeit([&] { k_linear12<<<g12, BLOCK>>>(d, n, out, 328, W, 1.f); }));
        rep("cols, pitch 352 (line aligned)", timeit([&] { k_cols<<<g12, BLOCK>>>(d, n, out, 352, C, 1.f); }));
        rep("cols, pitch 384", timeit([&] { k_cols<<<g12, BLOCK>>>(d, n, out, 384, C, 1.f); }));
        rep("linear12 aligned, pitch 352", timeit([&] { k_linear12<<<g12, BLOCK>>>(d, n, out, 352, W, 1.f); }));
        rep("linear12 aligned, pitch 336", timeit([&] { k_linear12<<<g12, BLOCK>>>(d, n, out, 336, W, 1.f); }));
        rep("peel12, pitch 327", timeit([&] { k_peel12<<<g12, BLOCK>>>(d, n, out, 327, W, 1.f); }));
        rep("linear64 aligned, pitch 328", timeit([&] { k_linear64<<<g64, BLOCK>>>(d, n, out, 328, W, 1.f); }));
    }
    const size_t n4 = (size_t)n * W / 4;
    rep("plain fill", timeit([&] { k_fill<<<(unsigned)((n4 + BLOCK - 1) / BLOCK), BLOCK>>>(out, n4, 1.f); }));
    CK(hipDeviceSynchronize());
    return 0;
}
