// ttl_oracle_net.hip -- TractOracle-Net (TrackToLearn/oracles/transformer_oracle.py:38-118)
// as ONE kernel: the streamline-scoring transformer of the oracle reward / oracle stopping
// criterion (oracle_reward.py:70-93, stopping_criteria.py:85-154, oracles/oracle.py:39-89).
//
// The network is tiny and oddly shaped for a GEMM library -- d_model 32, 128 tokens, 4 heads of
// 8, a 2 048-wide feed-forward block, 4 post-norm layers: 147 MFLOP per streamline, almost all
// of it two GEMMs with K or N = 32 -- and PyTorch's fp16-autocast path (what the reference runs)
// reaches ~45 TFLOP/s on it, 2 % of the MI355X's dense fp16 MFMA rate.  Here one wavefront owns
// one streamline and keeps the whole sequence in registers, transposed: h^T [32 features x 128
// tokens] as four 32x32 accumulator tiles (column = token = lane & 31, rows = features in the 16
// registers of the two lane halves).  Every product of the layer is arranged so that it sums
// over the ROW index of an accumulator tile, which `v_mfma_f32_32x32x16_f16` can take as its A or
// B operand straight from the registers (cdna_hip_programming.md, "an accumulator tile as the
// next MFMA's operand"): no LDS, no lane movement except one half-swap per softmax / LayerNorm
// reduction.  The k order inside such a fragment is permuted (element j of lane half h is row
// 16 s + 8 (j >> 2) + 4 h + (j & 3)), so the weights are packed on the host once per model in
// exactly that order (oracles/fused_net.py:pack_oracle_net).
//
//   Q^T, K^T = W_q,k . h^T            A = packed weights, B = h^T fragments
//   V        = h . W_v^T              A = h^T fragments (as h), B = packed weights
//   S_h^T    = K . (Q^T masked to head h)         A = K^T fragments (as K), B = Q^T fragments
//   O^T     += (V^T masked to head h) . P_h^T      A = V fragments (as V^T), B = P^T fragments
//   o^T      = W_o . O^T,  f^T = W_2 . relu(W_1 . h^T)   (the 2 048-wide intermediate lives in
//                                      16 registers per 32-unit chunk and is never stored)
//
// The last layer only needs token 0 (the CLS position feeds the head): its attention output,
// feed-forward block and LayerNorms run on the first token tile only.
//
// Arithmetic as under torch.autocast(fp16): fp16 operands, fp32 accumulation, Linear outputs
// rounded to fp16, softmax and LayerNorm in fp32.
#include <cstdlib>

#include "ttl_internal.h"
#include "ttl_learner.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));

constexpr int NT = 4;            // token tiles of 32 (128 tokens)
// batches of at most this many streamlines take the workgroup-per-streamline kernel
constexpr long long TTL_ORACLE_NET_WG_MAX_ROWS = 512;   // one resident round of 2 workgroups per CU
constexpr float LN_EPS = 1e-5f;

struct NetArgs {
    const float *dirs;           // [n][127][3]
    long long n;
    const _Float16 *wh;          // packed half weights, all layers
    const float *wf;             // packed float vectors, all layers
    const float *embed;          // [2][16][4]: w0, w1, w2, b of feature fperm(a, hi)
    const float *cls;            // [3]
    const float *pe;             // [4][64][16]
    const float *head;           // [2][16] weights + [1] bias
    int n_layers, ff_chunks;
    long long wh_stride, wf_stride;      // per layer, in h8 fragments-of-lane / floats
    float *out;
};

__device__ __forceinline__ float swap_halves(float v) { return __shfl_xor(v, 32); }

__device__ __forceinline__ void to_frags(const f16x &x, h8 (&f)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) f[s][j] = (_Float16)x[8 * s + j];
}

__device__ __forceinline__ f16x mfma(const h8 &a, const h8 &b, const f16x &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f16x zero16() {
    f16x z;
#pragma unroll
    for (int a = 0; a < 16; ++a) z[a] = 0.f;
    return z;
}

// x[a] += v[a] for a per-row vector packed [2 halves][16]
__device__ __forceinline__ void add_rows(f16x &x, const float *p, int hi) {
    const float *q = p + hi * 16;
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] += q[a];
}

// Linear outputs are fp16 under autocast
__device__ __forceinline__ void round_fp16(f16x &x) {
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = (float)(_Float16)x[a];
}

__device__ __forceinline__ void layer_norm(f16x &x, const float *g, const float *b, int hi) {
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < 16; ++a) sum += x[a];
    sum += swap_halves(sum);
    const float mean = sum * (1.f / 32.f);
    float sq = 0.f;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const float d = x[a] - mean;
        sq += d * d;
    }
    sq += swap_halves(sq);
    const float rstd = 1.f / sqrtf(sq * (1.f / 32.f) + LN_EPS);
    const float *gg = g + hi * 16, *bb = b + hi * 16;
#pragma unroll
    for (int a = 0; a < 16; ++a) x[a] = (x[a] - mean) * rstd * gg[a] + bb[a];
}

// S <- 2^((S - m) scale) for the 64 scores of a lane (its query against 64 of the 128 keys),
// returns their sum over all 128 keys.  Two scores per instruction: S scale - m scale as one
// v_pk_fma_f32, the running sum as v_pk_add_f32 (the library is built with -ffp-contract=off,
// so the compiler may not form the fma itself; the softmax is the largest block of VALU work
// in the kernel, and the VALU, not the MFMA pipe, is what the kernel waits for).
typedef float f2 __attribute__((ext_vector_type(2)));

// (x[8 s .. 8 s + 7] * inv) as an fp16 operand fragment, two elements per instruction in the
// register pairs the accumulator tile already has (written out so: left to itself the
// vectoriser pairs elements (1,2), (3,4), (5,6), and pays for it in register copies and
// v_alignbit shuffles -- a quarter of the attention block's instructions)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h8 scaled_frag(const f16x &x, int s, float inv) {
    const f2 i2 = {inv, inv};
    h2 q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f2 v = f2{x[8 * s + 2 * j], x[8 * s + 2 * j + 1]} * i2;
        q[j] = __builtin_convertvector(v, h2);
    }
    return h8{q[0].x, q[0].y, q[1].x, q[1].y, q[2].x, q[2].y, q[3].x, q[3].y};
}

__device__ __forceinline__ float softmax_numerators(f16x (&S)[NT], float m, float scale) {
    const f2 sc = {scale, scale};
    const float nm = -m * scale;
    const f2 off = {nm, nm};
    f2 l2 = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int a = 0; a < 16; a += 2) {
            const f2 s2 = {S[mt][a], S[mt][a + 1]};
            const f2 e = __builtin_elementwise_fma(s2, sc, off);
            f2 p;
            p.x = __builtin_amdgcn_exp2f(e.x);
            p.y = __builtin_amdgcn_exp2f(e.y);
            S[mt][a] = p.x;
            S[mt][a + 1] = p.y;
            l2 += p;
        }
    float l = l2.x + l2.y;
    l += swap_halves(l);
    return l;
}

// b_1 of a layer (ff floats, [chunk][2 halves][16]) staged in LDS by the whole workgroup:
// the feed-forward loop reads 64 bytes of it per lane and chunk, the same bytes in every lane
// of a half -- through the vector L1 that costs as much of its 64 B/clk as the weights
// themselves (a dwordx4 load occupies it for 16 clocks wherever its lanes point), from LDS it
// is a broadcast read on a pipe the kernel leaves idle.  Two buffers, one barrier per layer:
// a wave can only reach the write of layer l + 2 after every wave has passed the barrier of
// layer l + 1, i.e. finished reading layer l.
__device__ __forceinline__ const float *stage_b1(float *lds, const NetArgs &P, int layer) {
    const int ff = P.ff_chunks * 32;
    float *dst = lds + (layer & 1) * ff;
    const float *b1 = P.wf + (long long)layer * P.wf_stride + 288;
    const float4 *src = reinterpret_cast<const float4 *>(b1);
    for (int i = threadIdx.x; i < ff / 4; i += blockDim.x)
        reinterpret_cast<float4 *>(dst)[i] = src[i];
    __syncthreads();
    return dst;
}

__device__ __forceinline__ void load_bias(const float *b1s, int c, int hi, float (&bias)[16]) {
    const float4 *q = reinterpret_cast<const float4 *>(b1s + c * 32 + hi * 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 v = q[k];
        bias[4 * k + 0] = v.x; bias[4 * k + 1] = v.y; bias[4 * k + 2] = v.z; bias[4 * k + 3] = v.w;
    }
}

// One encoder layer on the wave's streamline.  NQ = 4: all four token tiles; NQ = 1 (the
// last layer): keys and values of all tokens, but queries, out-projection, feed-forward and
// LayerNorms of the first tile only (the head reads token 0).  A compile-time NQ keeps the
// tile loops free of branches, so the scheduler interleaves the tiles' MFMA chains.
template <int NHEAD, int NQ>
__device__ __forceinline__ void encoder_layer(f16x (&hT)[NT], const NetArgs &P, int layer,
                                              int lane, int n, int hi, float *lds) {
    constexpr int DH = 32 / NHEAD;
    const h8 *WH = reinterpret_cast<const h8 *>(P.wh) + (long long)layer * P.wh_stride;
    const float *WF = P.wf + (long long)layer * P.wf_stride;
    const h8 *Wqk = WH;                     // [2 mt][2 s][64]
    const h8 *Wv = WH + 4 * 64;             // [2 s][64]
    const h8 *Wo = WH + 6 * 64;             // [2 s][64]
    const h8 *W1 = WH + 8 * 64;             // [C][2 s][64]
    const h8 *W2 = W1 + (long long)P.ff_chunks * 2 * 64;
    const float *bqk = WF;                  // [2][2][16]
    const float *bv = WF + 64;              // [32]
    const float *bo = WF + 96;              // [2][16]
    const float *g1 = WF + 128, *be1 = WF + 160;
    const float *b2 = WF + 192, *g2 = WF + 224, *be2 = WF + 256;
    const float *b1s = stage_b1(lds, P, layer);     // b_1 [C][2][16], in LDS

    // ---- Q^T, K^T [features x tokens] and V [tokens x features]
    h8 QB[NQ][2], KA[NT][2], VA[NT][2];
    {
        const h8 wq0 = Wqk[0 * 64 + lane], wq1 = Wqk[1 * 64 + lane];
        const h8 wk0 = Wqk[2 * 64 + lane], wk1 = Wqk[3 * 64 + lane];
        const h8 wv0 = Wv[0 * 64 + lane], wv1 = Wv[1 * 64 + lane];
        const float bvc = bv[n];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // h^T as fp16 operand fragments (k = feature, permuted order)
            h8 hB[2];
            to_frags(hT[nt], hB);
            if (nt < NQ) {
                f16x q = mfma(wq1, hB[1], mfma(wq0, hB[0], zero16()));
                add_rows(q, bqk, hi);
                to_frags(q, QB[nt < NQ ? nt : 0]);
            }
            f16x k = mfma(wk1, hB[1], mfma(wk0, hB[0], zero16()));
            add_rows(k, bqk + 32, hi);
            to_frags(k, KA[nt]);
            // V tile: rows = tokens of this tile, column (lane) = feature n
            f16x v = mfma(hB[1], wv1, mfma(hB[0], wv0, zero16()));
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] += bvc;
            to_frags(v, VA[nt]);
        }
    }

    // ---- attention, one query tile at a time; out-projection, residual, LayerNorm 1
    const float scale = 1.4426950408889634f / sqrtf((float)DH);     // log2(e) / sqrt(dh)
#pragma unroll
    for (int nt = 0; nt < NQ; ++nt) {
        f16x OT = zero16();
#pragma unroll
        for (int h = 0; h < NHEAD; ++h) {
            // scores S^T [keys x queries] of head h
            f16x S[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                if constexpr (NHEAD == 1) {
                    S[mt] = mfma(KA[mt][1], QB[nt][1], mfma(KA[mt][0], QB[nt][0], zero16()));
                } else if constexpr (NHEAD == 2) {
                    S[mt] = mfma(KA[mt][h], QB[nt][h], zero16());
                } else {
                    // head h = features 8h..8h+7 = elements 4 (h & 1) .. +3 of k-step h >> 1
                    h8 qm = QB[nt][h >> 1];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if ((j >> 2) != (h & 1)) qm[j] = (_Float16)0.f;
                    S[mt] = mfma(KA[mt][h >> 1], qm, zero16());
                }
            }
            // softmax over the keys of each query (= over rows, per lane column)
            float m = S[0][0];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int a = 0; a < 16; ++a) m = fmaxf(m, S[mt][a]);
            m = fmaxf(m, swap_halves(m));
            const float l = softmax_numerators(S, m, scale);
            const float inv = 1.f / l;
            // O^T += V^T (rows of head h only) . P^T
            const bool mine = (n / DH) == h;        // this lane's V column belongs to head h
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                h8 PB[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) PB[s] = scaled_frag(S[mt], s, inv);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    h8 va = VA[mt][s];
                    if constexpr (NHEAD > 1) {
                        if (!mine) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) va[j] = (_Float16)0.f;
                        }
                    }
                    OT = mfma(va, PB[s], OT);
                }
            }
        }
        // out-projection on the fp16 attention output, residual, LayerNorm 1
        h8 OB[2];
        to_frags(OT, OB);
        f16x o = mfma(Wo[64 + lane], OB[1], mfma(Wo[lane], OB[0], zero16()));
        add_rows(o, bo, hi);
        round_fp16(o);
#pragma unroll
        for (int a = 0; a < 16; ++a) hT[nt][a] += o[a];
        layer_norm(hT[nt], g1, be1, hi);
    }

    // ---- feed-forward: f^T = W_2 . relu(W_1 . h^T + b_1) + b_2, 32 hidden units at a time
    h8 fB[NQ][2];
    f16x D2[NQ];
#pragma unroll
    for (int nt = 0; nt < NQ; ++nt) {
        to_frags(hT[nt], fB[nt]);
        D2[nt] = zero16();
    }
    h8 w1a = W1[lane], w1b = W1[64 + lane], w2a = W2[lane], w2b = W2[64 + lane];
    float bias[16];
    load_bias(b1s, 0, hi, bias);
    for (int c = 0; c < P.ff_chunks; ++c) {
        // prefetch the next chunk's weights and bias while this one is multiplied
        const int cn = c + 1 < P.ff_chunks ? c + 1 : c;
        const h8 n1a = W1[(long long)cn * 128 + lane], n1b = W1[(long long)cn * 128 + 64 + lane];
        const h8 n2a = W2[(long long)cn * 128 + lane], n2b = W2[(long long)cn * 128 + 64 + lane];
        float nbias[16];
        load_bias(b1s, cn, hi, nbias);
        // the tiles' chains side by side: GEMM 1 of every tile (the bias rides in as the
        // accumulator's initial value), the fp16 round + ReLU of every tile, GEMM 2
        f16x d1[NQ];
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt) {
#pragma unroll
            for (int a = 0; a < 16; ++a) d1[nt][a] = bias[a];
            d1[nt] = mfma(w1a, fB[nt][0], d1[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt) d1[nt] = mfma(w1b, fB[nt][1], d1[nt]);
        h8 F[NQ][2];
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // Linear output in fp16, then ReLU
                    const _Float16 u = (_Float16)d1[nt][8 * s + j];
                    F[nt][s][j] = u > (_Float16)0.f ? u : (_Float16)0.f;
                }
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt) D2[nt] = mfma(w2a, F[nt][0], D2[nt]);
#pragma unroll
        for (int nt = 0; nt < NQ; ++nt) D2[nt] = mfma(w2b, F[nt][1], D2[nt]);
        w1a = n1a; w1b = n1b; w2a = n2a; w2b = n2b;
#pragma unroll
        for (int a = 0; a < 16; ++a) bias[a] = nbias[a];
    }
#pragma unroll
    for (int nt = 0; nt < NQ; ++nt) {
        add_rows(D2[nt], b2, hi);
        round_fp16(D2[nt]);
#pragma unroll
        for (int a = 0; a < 16; ++a) hT[nt][a] += D2[nt][a];
        layer_norm(hT[nt], g2, be2, hi);
    }
}

template <int NHEAD>
__global__ __launch_bounds__(256, 1) void k_oracle_net(NetArgs P) {
    extern __shared__ __align__(16) float b1_lds[];              // [2][ff]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long row = (long long)blockIdx.x * 4 + wv;              // one wave per streamline
    // (a wave past the end works on the last streamline again and stores nothing: the
    // workgroup's barriers need every wave)
    const bool live = row < P.n;
    if (!live) row = P.n - 1;
    const int n = lane & 31, hi = lane >> 5;
    const float *dirs = P.dirs + row * (127 * 3);

    // ---- embedding: relu(W_e x + b_e) * sqrt(32) + positional encoding
    f16x hT[NT];
    {
        const float *E = P.embed + hi * 64;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int t = 32 * nt + n;
            float x0, x1, x2;
            if (t == 0) {
                x0 = P.cls[0]; x1 = P.cls[1]; x2 = P.cls[2];
            } else {
                const float *d = dirs + (t - 1) * 3;
                x0 = d[0]; x1 = d[1]; x2 = d[2];
            }
            // autocast: inputs and weights of the Linear in fp16, fp32 accumulation,
            // fp16 result; ReLU and the sqrt(32) scale in fp16; the table add in fp32
            x0 = (float)(_Float16)x0; x1 = (float)(_Float16)x1; x2 = (float)(_Float16)x2;
            const float *pe = P.pe + ((long long)nt * 64 + lane) * 16;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                float e = E[4 * a + 0] * x0 + E[4 * a + 1] * x1 + E[4 * a + 2] * x2 + E[4 * a + 3];
                e = (float)(_Float16)e;
                e = e > 0.f ? e : 0.f;
                e = (float)(_Float16)(e * (float)(_Float16)5.656854249492381f);
                hT[nt][a] = e + pe[a];
            }
        }
    }

    for (int layer = 0; layer + 1 < P.n_layers; ++layer)
        encoder_layer<NHEAD, NT>(hT, P, layer, lane, n, hi, b1_lds);
    encoder_layer<NHEAD, 1>(hT, P, P.n_layers - 1, lane, n, hi, b1_lds);

    // ---- head on the CLS position (token 0 = tile 0, lane column 0 of both halves)
    float dot = 0.f;
    {
        const float *w = P.head + hi * 16;
#pragma unroll
        for (int a = 0; a < 16; ++a)
            dot += (float)(_Float16)hT[0][a] * (float)(_Float16)w[a];
    }
    dot += swap_halves(dot);
    if (lane == 0 && live) {
        float y = (float)(_Float16)(dot + (float)(_Float16)P.head[32]);
        y = 1.f / (1.f + expf(-y));
        P.out[row] = (float)(_Float16)y;
    }
}

// ------------------------------------------------------------------------
// The same network with one WORKGROUP per streamline: each of its four waves
// owns one 32-token tile (16 accumulator registers of h^T instead of 64), the
// keys and values of all tiles meet in LDS once per layer (16 KB), everything
// else stays wave-private.  A quarter of the dependent work per wave: the
// latency of one streamline drops ~3x, and at <= 256 registers two workgroups
// share a CU, so one wave's conversions and softmax run under another's MFMAs.
// In the last layer only wave 0 (the tile of token 0) goes on after K / V.
// ------------------------------------------------------------------------
template <int NHEAD>
__device__ __forceinline__ void attention_tile(f16x &hT, const h8 (&QB)[2],
                                               const h8 (*kv)[NT][2][64], const NetArgs &P,
                                               int layer, int lane, int n, int hi) {
    constexpr int DH = 32 / NHEAD;
    const h8 *WH = reinterpret_cast<const h8 *>(P.wh) + (long long)layer * P.wh_stride;
    const float *WF = P.wf + (long long)layer * P.wf_stride;
    const h8 *Wo = WH + 6 * 64;
    const float *bo = WF + 96;
    const float *g1 = WF + 128, *be1 = WF + 160;

    h8 KA[NT][2], VA[NT][2];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            KA[mt][s] = kv[0][mt][s][lane];
            VA[mt][s] = kv[1][mt][s][lane];
        }
    const float scale = 1.4426950408889634f / sqrtf((float)DH);     // log2(e) / sqrt(dh)
    f16x OT = zero16();
#pragma unroll
    for (int h = 0; h < NHEAD; ++h) {
        f16x S[NT];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            if constexpr (NHEAD == 1) {
                S[mt] = mfma(KA[mt][1], QB[1], mfma(KA[mt][0], QB[0], zero16()));
            } else if constexpr (NHEAD == 2) {
                S[mt] = mfma(KA[mt][h], QB[h], zero16());
            } else {
                h8 qm = QB[h >> 1];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if ((j >> 2) != (h & 1)) qm[j] = (_Float16)0.f;
                S[mt] = mfma(KA[mt][h >> 1], qm, zero16());
            }
        }
        float m = S[0][0];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int a = 0; a < 16; ++a) m = fmaxf(m, S[mt][a]);
        m = fmaxf(m, swap_halves(m));
        const float l = softmax_numerators(S, m, scale);
        const float inv = 1.f / l;
        const bool mine = (n / DH) == h;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            h8 PB[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) PB[s] = scaled_frag(S[mt], s, inv);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                h8 va = VA[mt][s];
                if constexpr (NHEAD > 1) {
                    if (!mine) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) va[j] = (_Float16)0.f;
                    }
                }
                OT = mfma(va, PB[s], OT);
            }
        }
    }
    h8 OB[2];
    to_frags(OT, OB);
    f16x o = mfma(Wo[64 + lane], OB[1], mfma(Wo[lane], OB[0], zero16()));
    add_rows(o, bo, hi);
    round_fp16(o);
#pragma unroll
    for (int a = 0; a < 16; ++a) hT[a] += o[a];
    layer_norm(hT, g1, be1, hi);
}

// Feed-forward block of the workgroup's streamline, split over the HIDDEN units: wave w takes
// the chunks c = w, w + 4, ... for ALL token tiles (NQT = 4; 1 in the last layer, which only
// feeds token 0's tile) and the four partial sums of a tile meet in LDS.  With the block split
// over token tiles instead (one tile per wave) every weight fragment fed exactly one MFMA and
// each wave streamed the whole 256 KB of the layer through the vector L1 -- 1 KB per 32-cycle
// MFMA and SIMD, twice what it delivers; here a fragment feeds NQT MFMAs and the workgroup
// reads every weight once.  The partial sums are added in wave order (a fixed order: the
// scores do not depend on scheduling), which is not the order of the wave-per-streamline
// kernel's single accumulator -- the two kernels agree to an fp16 ulp of the score, not bit
// for bit.
template <int NQT>
__device__ __forceinline__ void ff_hidden_split(f16x &hT, const h8 (*fbuf)[2][64],
                                                float4 (*pbuf)[4][64], const NetArgs &P,
                                                int layer, int lane, int hi, int w,
                                                const float *b1s) {
    const h8 *WH = reinterpret_cast<const h8 *>(P.wh) + (long long)layer * P.wh_stride;
    const float *WF = P.wf + (long long)layer * P.wf_stride;
    const h8 *W1 = WH + 8 * 64;
    const h8 *W2 = W1 + (long long)P.ff_chunks * 2 * 64;
    const float *b2 = WF + 192, *g2 = WF + 224, *be2 = WF + 256;
    const int C = P.ff_chunks;

    h8 fB[NQT][2];
    f16x D2[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        fB[t][0] = fbuf[t][0][lane];
        fB[t][1] = fbuf[t][1][lane];
        D2[t] = zero16();
    }
    auto load = [&](int c, h8 (&wt)[4]) {
        wt[0] = W1[(long long)c * 128 + lane];
        wt[1] = W1[(long long)c * 128 + 64 + lane];
        wt[2] = W2[(long long)c * 128 + lane];
        wt[3] = W2[(long long)c * 128 + 64 + lane];
    };
    h8 wt[4];
    if (w < C) load(w, wt);
    for (int c = w; c < C; c += 4) {
        h8 nw[4];
        load(c + 4 < C ? c + 4 : c, nw);        // this wave's next chunk, one ahead
        // two token tiles at a time: two independent MFMA chains, half the transient registers
        // (the bias comes from LDS again for each pair: four broadcast reads, no registers held)
#pragma unroll
        for (int t0 = 0; t0 < NQT; t0 += 2) {
            constexpr int one = 1;
            const int t1 = t0 + one < NQT ? t0 + one : t0;
            float bias[16];
            load_bias(b1s, c, hi, bias);
            f16x da, db;
#pragma unroll
            for (int a = 0; a < 16; ++a) {
                da[a] = bias[a];
                db[a] = bias[a];
            }
            da = mfma(wt[0], fB[t0][0], da);
            if (NQT > 1) db = mfma(wt[0], fB[t1][0], db);
            da = mfma(wt[1], fB[t0][1], da);
            if (NQT > 1) db = mfma(wt[1], fB[t1][1], db);
            h8 FA[2], FB2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // Linear output in fp16, then ReLU
                    const _Float16 u = (_Float16)da[8 * s + j], v = (_Float16)db[8 * s + j];
                    FA[s][j] = u > (_Float16)0.f ? u : (_Float16)0.f;
                    FB2[s][j] = v > (_Float16)0.f ? v : (_Float16)0.f;
                }
            D2[t0] = mfma(wt[2], FA[0], D2[t0]);
            if (NQT > 1) D2[t1] = mfma(wt[2], FB2[0], D2[t1]);
            D2[t0] = mfma(wt[3], FA[1], D2[t0]);
            if (NQT > 1) D2[t1] = mfma(wt[3], FB2[1], D2[t1]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) wt[k] = nw[k];
    }
    // partial sums of the tiles this wave does not own -> LDS (slot 3 w + k: tile t != w in
    // rising order), then every owner adds the four partials of its tile in wave order
#pragma unroll
    for (int t = 0; t < NQT; ++t)
        if (t != w) {
            float4 (*dst)[64] = pbuf[3 * w + (t < w ? t : t - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[q][lane] = float4{D2[t][4 * q], D2[t][4 * q + 1], D2[t][4 * q + 2],
                                      D2[t][4 * q + 3]};
        }
    __syncthreads();
    if (w < NQT) {
        f16x sum = zero16();
#pragma unroll
        for (int src = 0; src < 4; ++src) {
            if (src == w) {
                // (NQT is a compile-time bound of t; the wave's own tile is D2[w])
#pragma unroll
                for (int t = 0; t < NQT; ++t)
                    if (t == w) {
#pragma unroll
                        for (int a = 0; a < 16; ++a) sum[a] += D2[t][a];
                    }
            } else {
                const float4 (*from)[64] = pbuf[3 * src + (w < src ? w : w - 1)];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = from[q][lane];
                    sum[4 * q] += v.x; sum[4 * q + 1] += v.y;
                    sum[4 * q + 2] += v.z; sum[4 * q + 3] += v.w;
                }
            }
        }
        add_rows(sum, b2, hi);
        round_fp16(sum);
#pragma unroll
        for (int a = 0; a < 16; ++a) hT[a] += sum[a];
        layer_norm(hT, g2, be2, hi);
    }
    __syncthreads();            // the partials' memory is the next layer's keys and values
}

template <int NHEAD>
__global__ __launch_bounds__(256, 2) void k_oracle_net_wg(NetArgs P) {
    // 48 KB: during the attention the keys | values of all tiles ([2][tile][k-step][lane] fp16
    // fragments, 16 KB), during the feed-forward block the partial sums of the hidden split
    // ([12 slots][4][lane] float4); + 8 KB of h^T fragments of all tiles; two workgroups per CU
    __shared__ float4 pbuf[12][4][64];
    __shared__ h8 fbuf[NT][2][64];
    h8 (*kv)[NT][2][64] = reinterpret_cast<h8 (*)[NT][2][64]>(&pbuf[0][0][0]);
    extern __shared__ __align__(16) float b1_lds[]; // [2][ff], see stage_b1
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;      // wave w owns token tile w
    const long long row = blockIdx.x;               // one workgroup per streamline
    const int n = lane & 31, hi = lane >> 5;
    const float *dirs = P.dirs + row * (127 * 3);

    // A training step calls this kernel right after an update that has swept the caches: the
    // first touch of each layer's weights would otherwise be a chain of HBM round trips, one
    // chunk ahead of its use.  The workgroups of this XCD (blockIdx.x mod 8, round-robin
    // dispatch) read the whole packed set once, up front and in parallel, into their L2;
    // the values are folded into a word that is only looked at after the last layer.
    unsigned warm = 0;
    {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 *wp = reinterpret_cast<const u4 *>(P.wh);
        const long long total = (long long)P.n_layers * P.wh_stride;       // 16-byte units
        const long long per_xcd = (gridDim.x + 7) / 8;
        const long long i0 = (long long)(blockIdx.x >> 3) * 256 + threadIdx.x;
        // sixteen independent 16-byte loads per thread, all in flight together (one wait):
        // 35 workgroups per XCD -- what a step of config 5 scores -- cover the 2.1 MB set
        u4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            long long i = i0 + k * per_xcd * 256;
            i = i < total ? i : total - 1;
            v[k] = wp[i];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) warm ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        // fold them HERE: left to the scheduler the sixteen results wait (spilled) for the
        // word's only reader at the end of the kernel
        asm volatile("" : "+v"(warm));
    }

    f16x hT;
    {
        const float *E = P.embed + hi * 64;
        const int t = 32 * w + n;
        float x0, x1, x2;
        if (t == 0) {
            x0 = P.cls[0]; x1 = P.cls[1]; x2 = P.cls[2];
        } else {
            const float *d = dirs + (t - 1) * 3;
            x0 = d[0]; x1 = d[1]; x2 = d[2];
        }
        x0 = (float)(_Float16)x0; x1 = (float)(_Float16)x1; x2 = (float)(_Float16)x2;
        const float *pe = P.pe + ((long long)w * 64 + lane) * 16;
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            float e = E[4 * a + 0] * x0 + E[4 * a + 1] * x1 + E[4 * a + 2] * x2 + E[4 * a + 3];
            e = (float)(_Float16)e;
            e = e > 0.f ? e : 0.f;
            e = (float)(_Float16)(e * (float)(_Float16)5.656854249492381f);
            hT[a] = e + pe[a];
        }
    }

    for (int layer = 0; layer < P.n_layers; ++layer) {
        const bool last = layer == P.n_layers - 1;
        const bool goes_on = !last || w == 0;       // the last layer only feeds token 0's tile
        const h8 *WH = reinterpret_cast<const h8 *>(P.wh) + (long long)layer * P.wh_stride;
        const float *WF = P.wf + (long long)layer * P.wf_stride;
        const h8 *Wqk = WH, *Wv = WH + 4 * 64;
        const float *bqk = WF, *bv = WF + 64;
        h8 hB[2], QB[2];
        to_frags(hT, hB);
        {
            f16x k = mfma(Wqk[3 * 64 + lane], hB[1], mfma(Wqk[2 * 64 + lane], hB[0], zero16()));
            add_rows(k, bqk + 32, hi);
            h8 f[2];
            to_frags(k, f);
            kv[0][w][0][lane] = f[0];
            kv[0][w][1][lane] = f[1];
            f16x v = mfma(hB[1], Wv[64 + lane], mfma(hB[0], Wv[lane], zero16()));
            const float bvc = bv[n];
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] += bvc;
            to_frags(v, f);
            kv[1][w][0][lane] = f[0];
            kv[1][w][1][lane] = f[1];
        }
        if (goes_on) {
            f16x q = mfma(Wqk[64 + lane], hB[1], mfma(Wqk[lane], hB[0], zero16()));
            add_rows(q, bqk, hi);
            to_frags(q, QB);
        }
        const float *b1s = stage_b1(b1_lds, P, layer);      // its barrier also publishes kv
        if (goes_on) {
            attention_tile<NHEAD>(hT, QB, kv, P, layer, lane, n, hi);
            // h^T of this tile after LayerNorm 1, as operand fragments, for every wave
            h8 f[2];
            to_frags(hT, f);
            fbuf[w][0][lane] = f[0];
            fbuf[w][1][lane] = f[1];
        }
        __syncthreads();                            // everybody is done with kv as well
        if (last) ff_hidden_split<1>(hT, fbuf, pbuf, P, layer, lane, hi, w, b1s);
        else ff_hidden_split<NT>(hT, fbuf, pbuf, P, layer, lane, hi, w, b1s);
    }

    if (w == 0) {
        float dot = 0.f;
        const float *wh = P.head + hi * 16;
#pragma unroll
        for (int a = 0; a < 16; ++a) dot += (float)(_Float16)hT[a] * (float)(_Float16)wh[a];
        dot += swap_halves(dot);
        if (lane == 0) {
            float y = (float)(_Float16)(dot + (float)(_Float16)P.head[32]);
            y = 1.f / (1.f + expf(-y));
            P.out[row] = (float)(_Float16)y;
        }
    }
    // (keeps the warming loads alive; weights are finite halves, the pattern cannot occur)
    if (warm == 0xfff1fff2u && P.n < 0) P.out[row] = 0.f;
}

}  // namespace

extern "C" {

int ttl_oracle_net_forward(const float *dirs, int64_t n, const void *packed_half,
                           const float *packed_float, const float *embed, const float *cls,
                           const float *pos_enc, const float *head, int32_t n_layers,
                           int32_t n_head, int32_t ff_dim, float *scores, void *hip_stream) {
    if (!dirs || !packed_half || !packed_float || !embed || !cls || !pos_enc || !head || !scores)
        return fail(TTL_ERR_INVALID, "ttl_oracle_net_forward: null pointer");
    if (n <= 0 || n_layers <= 0 || ff_dim <= 0 || ff_dim % 32)
        return fail(TTL_ERR_INVALID, "ttl_oracle_net_forward: bad shape (n %lld, layers %d, ff %d)",
                    (long long)n, n_layers, ff_dim);
    if (ff_dim > 8192)      // b_1 of two layers in LDS: 8 ff bytes of the default 64 KB
        return fail(TTL_ERR_UNSUPPORTED, "ttl_oracle_net_forward: feed-forward width %d > 8192",
                    ff_dim);
    if (n_head != 1 && n_head != 2 && n_head != 4)
        return fail(TTL_ERR_UNSUPPORTED, "ttl_oracle_net_forward: 1, 2 or 4 heads (got %d)", n_head);
    if (((uintptr_t)packed_half & 15u) || ((uintptr_t)packed_float & 15u))
        return fail(TTL_ERR_INVALID, "ttl_oracle_net_forward: packed weights must be 16-byte aligned");
    const int chunks = ff_dim / 32;
    NetArgs P{dirs, (long long)n, reinterpret_cast<const _Float16 *>(packed_half), packed_float,
              embed, cls, pos_enc, head, n_layers, chunks,
              (long long)(8 + 4 * chunks) * 64, (long long)288 + 32 * chunks, scores};
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    // one workgroup per streamline (a quarter of the latency; TTL_ORACLE_NET_WG = 1 always,
    // 0 never) or one wavefront per streamline
    static const int wg_mode = [] {
        const char *v = getenv("TTL_ORACLE_NET_WG");
        return v ? atoi(v) : -1;
    }();
    const bool wg = wg_mode == 1 || (wg_mode != 0 && n <= TTL_ORACLE_NET_WG_MAX_ROWS);
    const size_t lds = (size_t)2 * ff_dim * sizeof(float);          // b_1, two layers
    if (wg) {
        const dim3 grid((unsigned)n), block(256);
        if (n_head == 1) k_oracle_net_wg<1><<<grid, block, lds, s>>>(P);
        else if (n_head == 2) k_oracle_net_wg<2><<<grid, block, lds, s>>>(P);
        else k_oracle_net_wg<4><<<grid, block, lds, s>>>(P);
    } else {
        const dim3 grid((unsigned)((n + 3) / 4)), block(256);
        if (n_head == 1) k_oracle_net<1><<<grid, block, lds, s>>>(P);
        else if (n_head == 2) k_oracle_net<2><<<grid, block, lds, s>>>(P);
        else k_oracle_net<4><<<grid, block, lds, s>>>(P);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(TTL_ERR_HIP, "k_oracle_net: %s", hipGetErrorString(e));
    return TTL_OK;
}

}  // extern "C"
