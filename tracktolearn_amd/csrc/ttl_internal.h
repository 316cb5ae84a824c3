// ttl_internal.h -- shared by the translation units of libttl_hip.so (not
// part of the ABI).
#ifndef TTL_INTERNAL_H
#define TTL_INTERNAL_H
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ttl_hip.h"

// records the message returned by ttl_last_error() (thread local) and
// returns `code`
int ttl_detail_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail ttl_detail_fail

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess)                                                \
            return fail(TTL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int TTL_BLOCK = 256;   // threads per workgroup of every kernel

// Device-side view of the environment (a flat copy of the descriptor).
struct EnvParams {
    int mode;
    int sh_dim[3];
    int n_coef;
    int coef_pitch;
    const float *sh;
    float sh_shift;
    int sh_brick;          // TTL_SH_BRICK4 record order (see ttl_hip.h)
    unsigned sh_sx, sh_sy; // records per unit of x / y: linear Y*Z, Z; bricked: per brick step
    int mask_dim[3];
    const double *mask_coef;
    const uint8_t *mask_cls;  // per-cell class (see k_mask_classes) or null
    double mask_thr;
    int peaks_dim[3];
    const float *peaks;
    int compute_reward;
    float align_w;
    int n_dirs;
    int max_nb_steps;
    double step64;
    float step32;
    float radius;
    int curv_enabled;
    float curv_dot_max;
    float *hist;
    int *flags;
    int *lengths;
    uint8_t *dones;
    // workspace
    uint8_t *stop;     // [n_max] 1 = stopped in the last step
    float *head;       // [n_max][4] newest point of every active row (row order), .w = bits of idx[row]
    int *pos_dest;     // [n_max][2] {surv_pos, row_dest} of every active row, packed for k_proc_scatter
    float *last2;      // [n_max][8] per streamline id: {p[L-2], pad, p[L-1], pad}, the two newest points
    int *rank;         // [n_max] survivors before this row inside its block
    int *surv_pos;     // [n_max] position among survivors, -1 if stopped
    int *row_dest;     // [n_max] state row written for this active row
    int *stop_list;    // [n_max][2] {active row, streamline id} of the rows that stopped in the last step, in row order
    int *block_counts; // [ceil(n_max/BLOCK)] survivors per block
    int *proc_rank;    // [n_max] rank of a kept slot of the processing order
    int *proc_counts;  // [ceil(n_max/BLOCK)] kept slots per block
    float *slot_head;  // [n_max][4] per slot of the processing order: newest point, .w = bits of idx[row]
    int *slot_dest;    // [n_max] per slot of the processing order: row_dest[row]
    int slot_rec;      // the gather reads the slot records (0: resolves proc -> idx/row_dest/head itself)
    int store_flavour; // cache policy of the state-row stores (TTL_STORE_FLAVOUR, see store16)
    int xcd_remap;     // XCD-contiguous ranges of the processing order (TTL_XCD_REMAP)
    int xcd_rot;       // XCD x gathers range (x + xcd_rot) & 7 of the processing order
    int *counts;       // {n_continue, n_stopped}; 64 ints: the free-running step's words live here too
    int fuse_max_rows; // largest batch of the one-launch step tail (TTL_FUSE_MAX_ROWS, <= TTL_FUSE_MAX_BLOCKS * 256)
    int persist_rows;  // gathers of at most this many rows run as one resident round (TTL_GATHER_PERSIST_ROWS, 0 = off)
};

// the one-launch tail scans at most this many per-block counts (one wave, four per lane)
constexpr int TTL_FUSE_MAX_BLOCKS = 256;

// free-running step (ttl_env_freerun_*): int offsets into EnvParams::counts of
// {n_active, length, cur, steps done} -- live (between steps) and the snapshot
// a step works from
constexpr int TTL_FR_LIVE = 8;
constexpr int TTL_FR_SNAP = 16;

// Record index of voxel (x, y, z) = vox_x(x) + vox_y(y) + vox_z(z), for both
// record orders (separable, so the gather keeps per-axis partial offsets).
__device__ __forceinline__ unsigned vox_x(const EnvParams &P, int x) {
    return P.sh_brick ? (unsigned)(x >> 2) * P.sh_sx + (unsigned)(x & 3) * 16u
                      : (unsigned)x * P.sh_sx;
}
__device__ __forceinline__ unsigned vox_y(const EnvParams &P, int y) {
    return P.sh_brick ? (unsigned)(y >> 2) * P.sh_sy + (unsigned)(y & 3) * 4u
                      : (unsigned)y * P.sh_sy;
}
__device__ __forceinline__ unsigned vox_z(const EnvParams &P, int z) {
    return P.sh_brick ? (unsigned)(z >> 2) * 64u + (unsigned)(z & 3) : (unsigned)z;
}

// records of the packed SH volume (padding records of the bricked order included)
inline size_t ttl_detail_sh_records(const EnvParams &P) {
    if (!P.sh_brick) return (size_t)P.sh_dim[0] * P.sh_dim[1] * P.sh_dim[2];
    return (size_t)((P.sh_dim[0] + 3) / 4) * ((P.sh_dim[1] + 3) / 4) *
           ((P.sh_dim[2] + 3) / 4) * 64;
}

// ttl_state.hip: gathers the state rows of `n_rows` active rows (a step when
// idx != nullptr, the reset otherwise) on stream s
int ttl_detail_launch_state(const EnvParams &P, int state_kernel, const int *idx,
                            const int *row_dest, const int *proc, int n_rows, int L,
                            float *out, int64_t pitch, hipStream_t s);
// ttl_state.hip: whether ttl_detail_launch_state() takes the register-deduplicated
// gather (k_state_dd: reads the per-slot records of a processing order)
bool ttl_detail_state_dedupes(const EnvParams &P, int state_kernel);
// ttl_state.hip: the small-batch step tail (prefix + compaction + gather) in
// one launch; host_word = device-visible pinned {n_continue, n_stopped, seq}
// or null
bool ttl_detail_can_fuse_tail(const EnvParams &P, int n_active);
int ttl_detail_launch_fused_tail(const EnvParams &P, const int *idx, int *idx_next,
                                 int n_active, int order, int n_pts, float *out,
                                 int64_t pitch, int *host_word, int seq, hipStream_t s);
// ttl_state.hip: the same tail for a free-running step: n_active, the length
// and the live continue_idx buffer are read from P.counts + TTL_FR_SNAP, the
// next step's words are written to P.counts + TTL_FR_LIVE; rows in partition
// order; n_cap = rows the launch covers
int ttl_detail_launch_fused_tail_fr(const EnvParams &P, int *idx_a, int *idx_b, int n_cap,
                                    float *out, int64_t pitch, int *host_word,
                                    hipStream_t s);
// ttl_order.hip: rows 0..n-1 sorted by the 8^3-voxel brick of their newest
// point (P.last2 of streamline idx[row]) -> order_out[n]; ws = scratch of
// ttl_detail_order_workspace_bytes(n_max) bytes
size_t ttl_detail_order_workspace_bytes(size_t n);
int ttl_detail_refresh_order(const EnvParams &P, const int *idx, int n, char *ws,
                             size_t ws_bytes, int *order_out, hipStream_t s);
#endif
