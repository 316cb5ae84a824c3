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
    float *last2;      // [n_max][8] per streamline id: {p[L-2], pad, p[L-1], pad}, the two newest points
    int *rank;         // [n_max] survivors before this row inside its block; sign bit: the row stopped
    int *surv_pos;     // [n_max] position among survivors, -1 if stopped
    int *row_dest;     // [n_max] state row written for this active row
    int *block_counts; // [ceil(n_max/BLOCK)] survivors per block of k_advance
    int *block_before; // [ceil(n_max/BLOCK) + 1] survivors in the blocks before this one (k_advance's last block)
    unsigned long long *block_tagged; // [ceil(n_max/BLOCK)] {epoch, count} granules of a scanning launch
    float *slot_head;  // [n_max + 256][4] per slot of the processing order: newest point, .w = bits of idx[row]
    int *slot_dest;    // [n_max + 256] per slot of the processing order: row_dest[row]
    int seg_slots;     // slots per segment of the processing order (see TTL_SEG_* below)
    int store_flavour; // cache policy of the state-row stores (TTL_STORE_FLAVOUR, see store16)
    int xcd_remap;     // XCD-contiguous ranges of the processing order (TTL_XCD_REMAP)
    int xcd_rot;       // XCD x gathers range (x + xcd_rot) & 7 of the processing order
    int *counts;       // {n_continue, n_stopped}; 64 ints: the free-running step's words and
                       // the arrival ticket of k_advance's blocks (TTL_SCAN_TICKET) live here too
};

// Processing order in SEGMENTS (round 3).  The order of the state gather is
// kept as segments of seg_slots slots (the largest multiple of the gather's
// rows per workgroup that fits a 256-thread workgroup: 240 for 12 lanes per
// streamline, 256 otherwise), stored at stride seg_slots: segment s owns
// proc[s * seg_slots ...] and holds seg_cnt[s] live slots at its front.  A
// streamline that stops leaves its segment; nothing moves between segments
// until the next global refresh (ttl_order.hip) rebuilds a dense order.  So
// compacting the order needs no prefix over workgroups -- one workgroup per
// segment does it alone (k_slots) -- and the gather's workgroups, each inside
// one segment, skip the slots past seg_cnt.
__host__ __device__ inline int ttl_detail_lanes_per_streamline(int coef_pitch) {
    const int c4 = coef_pitch >> 2;
    return c4 <= 4 ? 4 : c4 <= 8 ? 8 : c4 <= 12 ? 12 : c4 <= 16 ? 16 : 32;
}
__host__ __device__ inline int ttl_detail_seg_slots(int coef_pitch) {
    const int rows = (TTL_BLOCK / 64) * (64 / ttl_detail_lanes_per_streamline(coef_pitch));
    return TTL_BLOCK / rows * rows;
}
// int offset into EnvParams::counts of the arrival ticket of k_advance's /
// k_restop's workgroups (the last one to arrive scans the per-block counts)
constexpr int TTL_SCAN_TICKET = 32;

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
// seg_cnt != nullptr (a step in processing order): n_rows = segments *
// P.seg_slots slots, of which segment s has seg_cnt[s] live ones; the gather
// reads the per-slot records k_slots left (P.slot_head / P.slot_dest)
int ttl_detail_launch_state(const EnvParams &P, int state_kernel, const int *idx,
                            const int *row_dest, const int *proc, const int *seg_cnt,
                            int n_rows, int L, float *out, int64_t pitch, hipStream_t s);
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
