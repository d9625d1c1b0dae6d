// Soft-attention front end on MI355X (gfx950): text x mel log-likelihood matrix.
//
//   logit[b,i,j] = -T * sum_c (Q[b,c,j] - K[b,c,i])^2        (ALIGNER_SIM_L2)
//               =  T * sum_c  Q[b,c,j] * K[b,c,i]            (ALIGNER_SIM_DOT)
//   logp [b,i,j] = log_softmax over the text axis i (rows i >= t_x masked to -inf)
//                  (+ log(prior + 1e-8))
//
// This is the build-defined spec of SURVEY.md 7.4 (the reference snapshot only
// links the OTA paper, README.md:50); parity is against oracle/softattn_oracle.py.
//
// Design (HBM-bound: 4*B*C*(Tx+Ty) bytes in, 4*B*Tx*Ty out, 2*B*Tx*Ty*C flops):
//  * L2 distance expands to |q|^2 + |k|^2 - 2 k.q, so the only O(Tx*Ty*C) work is
//    the [Tx,C]x[C,Ty] contraction -> MFMA.  fp32-input MFMA runs at the vector
//    rate (too slow to hide under the HBM time), so each fp32 operand is split
//    x = hi + lo into two bf16 halves and the product is three bf16 MFMAs
//    (hi*hi + hi*lo + lo*hi, fp32 accumulate): error ~2^-16 relative per product,
//    far inside the 1e-4 tolerance, at ~5x the fp32-MFMA rate.
//  * One wave owns a strip of 32 mel frames and ALL text rows of a row group, so
//    the softmax over the text axis is a reduction over its own accumulator
//    registers plus one cross-half shuffle; logits never leave registers.
//  * A workgroup = 4 waves = 128 consecutive frames of one utterance.  The text
//    operand (shared by the 4 waves) is split to bf16 once per workgroup into LDS
//    in MFMA fragment order (one ds_read_b128 per fragment, conflict-free); the mel
//    operand is read once, coalesced, straight into B fragments.
//  * Text longer than G*32 rows is processed in row groups with a two-pass
//    (running max / sum, then normalise) sweep; the contraction is simply redone
//    in the second pass -- MFMA time is not the bound.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct SoftAttnParams {
    const float *keys;      // [B,C,Tx]
    const float *queries;   // [B,C,Ty]
    const int   *t_xs;      // nullable
    const float *prior;     // nullable [B,Tx,Ty]
    float *logp;            // [B,Tx,Ty]
    float *soft;            // nullable
    const uint4 *frag_hi;   // [B][RT][KS][64] text operand, bf16 high halves in MFMA A-fragment order
    const uint4 *frag_lo;   // same, low halves
    const float *knorm;     // [B][RT*32] |k_i|^2
    int RT;                 // row tiles of 32 text rows
    unsigned long long *stamps;   // debug (nullable): [blocks][4 waves][8] shader clock
    int B, C, Tx, Ty;
    float temperature;
    int sim;
};

constexpr int SA_WAVES = 8;                       // waves per workgroup: 8 x 32 = 256 mel frames share one staged text operand
constexpr int SA_THREADS = SA_WAVES * 64;
constexpr float NEG_INF_F = -__builtin_huge_valf();
constexpr float LOG2E_F = 1.4426950408889634f, LN2_F = 0.6931471805599453f;

__device__ __forceinline__ void split_bf16(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// Text operand prep (once per utterance, not once per workgroup): split K to bf16 hi/lo in
// MFMA A-fragment order (A[i = lane&31][k = 8*(lane>>5)+jj] for row tile r, k-step s) and
// |k_i|^2, into the workspace.  One workgroup per (row tile, utterance).
template <int KS>
__global__ __launch_bounds__(256) void softattn_prep_kernel(const float *__restrict__ keys, uint4 *__restrict__ frag_hi,
                                                            uint4 *__restrict__ frag_lo, float *__restrict__ knorm,
                                                            int C, int Tx, int RT) {
    __shared__ float part[32][2 * KS + 1];
    const int tid = threadIdx.x;
    const int r = blockIdx.x, b = blockIdx.y;
    const float *Kb = keys + (size_t)b * C * Tx;
    for (int idx = tid; idx < KS * 64; idx += 256) {
        const int ln = idx & 63, s = idx >> 6;
        const int i = 32 * r + (ln & 31);
        const int c0 = 16 * s + 8 * (ln >> 5);
        bf16x8 h, l;
        float sq = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            float v = 0.f;
            if (i < Tx && c0 + jj < C) v = Kb[(size_t)(c0 + jj) * Tx + i];
            sq += v * v;
            __bf16 hh, ll;
            split_bf16(v, hh, ll);
            h[jj] = hh;
            l[jj] = ll;
        }
        const size_t o = ((size_t)(b * RT + r) * KS + s) * 64 + ln;
        frag_hi[o] = __builtin_bit_cast(uint4, h);
        frag_lo[o] = __builtin_bit_cast(uint4, l);
        part[ln & 31][2 * s + (ln >> 5)] = sq;
    }
    __syncthreads();
    if (tid < 32) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * KS; ++j) acc += part[tid][j];      // fixed order: deterministic
        knorm[(size_t)(b * RT + r) * 32 + tid] = acc;
    }
}

// Row group -> LDS: plain 16-byte copies of the prepared fragments (+ the norms).
template <int KS, int G>
__device__ __forceinline__ void stage_text_group(const SoftAttnParams &p, int b, int g, uint4 *Ahi, uint4 *Alo,
                                                 float *kn, int tx, bool l2, float s2) {
    const int tid = threadIdx.x;
    const int r0 = G * g;
    const size_t base = ((size_t)(b * p.RT + r0) * KS) * 64;
    const int nvalid = (p.RT - r0 < G ? p.RT - r0 : G) * KS * 64;      // fragments that exist
    for (int idx = tid; idx < G * KS * 64; idx += SA_THREADS) {
        uint4 h = make_uint4(0, 0, 0, 0), l = h;
        if (idx < nvalid) { h = p.frag_hi[base + idx]; l = p.frag_lo[base + idx]; }
        Ahi[idx] = h;
        Alo[idx] = l;
    }
    // per-row additive term of the base-2 logit: s2*|k_i|^2 (L2) or 0 (dot); -inf masks rows >= t_x
    for (int il = tid; il < G * 32; il += SA_THREADS) {
        const int i = r0 * 32 + il;
        float v = NEG_INF_F;
        if (i < tx) v = l2 ? s2 * p.knorm[(size_t)(b * p.RT + r0) * 32 + il] : 0.f;
        kn[il] = v;
    }
}

// Single-row-group path: split the text operand straight from K into LDS (no prep launch).
// Each workgroup redoes the split for its utterance -- ~100 VALU per thread, cheaper than the
// extra kernel boundary and the workspace round trip.  `part` collects per-(k-step, half) partial
// |k|^2 sums so the norm is added in a fixed order.
template <int KS, int G>
__device__ __forceinline__ void stage_text_direct(const SoftAttnParams &p, const float *Kb, uint4 *Ahi, uint4 *Alo,
                                                  float *part) {
    const int tid = threadIdx.x;
    constexpr int NIT = (G * KS * 64 + SA_THREADS - 1) / SA_THREADS;
    // issue every load of this thread first (NIT x 8 strided dwords in flight), convert afterwards
    float raw[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * SA_THREADS;
        const int ln = idx & 63;
        const int s = (idx >> 6) % KS;
        const int r = (idx >> 6) / KS;
        const int i = 32 * r + (ln & 31);
        const int c0 = 16 * s + 8 * (ln >> 5);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
            raw[it][jj] = (idx < G * KS * 64 && i < p.Tx && c0 + jj < p.C) ? Kb[(size_t)(c0 + jj) * p.Tx + i] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * SA_THREADS;
        if (idx < G * KS * 64) {
            const int ln = idx & 63;
            const int s = (idx >> 6) % KS;
            const int r = (idx >> 6) / KS;
            bf16x8 h, l;
            float sq = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float v = raw[it][jj];
                sq += v * v;
                __bf16 hh, ll;
                split_bf16(v, hh, ll);
                h[jj] = hh;
                l[jj] = ll;
            }
            Ahi[idx] = __builtin_bit_cast(uint4, h);
            Alo[idx] = __builtin_bit_cast(uint4, l);
            part[(32 * r + (ln & 31)) * (2 * KS) + 2 * s + (ln >> 5)] = sq;
        }
    }
}

// dot products of one 32-row tile with this wave's 32 frames: 3x bf16 MFMA per k-step
// (hi*hi + hi*lo + lo*hi).  C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
template <int KS>
__device__ __forceinline__ f32x16 tile_dot(const uint4 *Ahi_r, const uint4 *Alo_r, const bf16x8 (&bhi)[KS],
                                           const bf16x8 (&blo)[KS], int lane) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, Ahi_r[s * 64 + lane]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, Alo_r[s * 64 + lane]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[s], acc, 0, 0, 0);
    }
    return acc;
}

// logits of one 32-row tile for this lane's column: 3x bf16 MFMA + distance/scale/mask.
// C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
template <int KS>
__device__ __forceinline__ void tile_logits(float (&lg)[16], const uint4 *Ahi_r, const uint4 *Alo_r,
                                            const bf16x8 (&bhi)[KS], const bf16x8 (&blo)[KS],
                                            const float *kn_r, float qn, float s2, bool l2,
                                            int i_lane0, int tx, int lane) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, Ahi_r[s * 64 + lane]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, Alo_r[s * 64 + lane]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[s], acc, 0, 0, 0);
    }
    const int half4 = 4 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int il = (e & 3) + 8 * (e >> 2) + half4;
        const float d = acc[e];
        // kn_r holds the base-2 row term (s2*|k|^2 or 0; -inf for masked rows): back to natural log
        const float v = (l2 ? fmaf(d, -2.0f * s2, kn_r[il] + s2 * qn) : fmaf(d, s2, kn_r[il])) * LN2_F;
        lg[e] = v;
    }
}

// mel operand: B fragments (k = channel, col = frame) + |q_j|^2, read once per wave
template <int KS>
__device__ __forceinline__ float load_mel_fragments(const SoftAttnParams &p, const float *Qb, int col, bool col_ok,
                                                    int half, bf16x8 (&bhi)[KS], bf16x8 (&blo)[KS]) {
    float qn = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int c = 16 * s + 8 * half + jj;
            float v = 0.f;
            if (c < p.C && col_ok) v = Qb[(size_t)c * p.Ty + col];
            qn += v * v;
            __bf16 h, l;
            split_bf16(v, h, l);
            bhi[s][jj] = h;
            blo[s][jj] = l;
        }
    }
    return qn + __shfl_xor(qn, 32);
}

// MULTI == false: all text rows fit one row group (Tx <= 32*G): logits stay in registers.
// MULTI == true : row groups, two sweeps (running max/sum, then normalise + store).
#define SA_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (p.stamps && (threadIdx.x & 63) == 0)                                                      \
            p.stamps[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SA_WAVES + (threadIdx.x >> 6)) * 8 + (k)] = \
                __builtin_amdgcn_s_memtime();                                                         \
    } while (0)

template <int KS, int G, bool MULTI>
__global__ __launch_bounds__(SA_THREADS, (KS == 16 && MULTI) ? 1 : 2) void softattn_kernel(SoftAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *Ahi = reinterpret_cast<uint4 *>(smem);              // [G][KS][64] bf16x8 fragments
    uint4 *Alo = Ahi + G * KS * 64;                            // [G][KS][64]
    float *kn = reinterpret_cast<float *>(Alo + G * KS * 64);  // [G*32]
    float *part = kn + G * 32;                                 // [G*32][2*KS] (single-group path only)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;
    const int b = blockIdx.y;
    const int col = blockIdx.x * (32 * SA_WAVES) + wave * 32 + (lane & 31);
    const bool col_ok = col < p.Ty;
    int tx = p.Tx;
    if (p.t_xs) {
        tx = p.t_xs[b];
        tx = tx < 0 ? 0 : (tx > p.Tx ? p.Tx : tx);
    }
    const float *Qb = p.queries + (size_t)b * p.C * p.Ty;
    const bool l2 = (p.sim == ALIGNER_SIM_L2);
    const float scale = l2 ? -p.temperature : p.temperature;
    const float s2 = scale * LOG2E_F;                  // logits are kept in base 2 (v_exp_f32 is exp2)

    SA_STAMP(0);
    // mel operand: strided loads, converted after the text operand is staged.  The waves of a
    // workgroup are split in two groups: the first issues its mel loads before the staging, the
    // second only after the staging barrier, so that the second group's loads and MFMA/softmax phase
    // overlap the first group's MFMA/softmax and store phases (all workgroups start together, so
    // without the stagger the whole GPU alternates between an HBM-read, a compute and an HBM-write phase)
    const bool early = MULTI || wave < SA_WAVES / 2;
    float qraw[KS][8];
    if (early) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int c = 16 * s + 8 * half + jj;
                qraw[s][jj] = (c < p.C && col_ok) ? Qb[(size_t)c * p.Ty + col] : 0.f;
            }
    }
    if (!MULTI) stage_text_direct<KS, G>(p, p.keys + (size_t)b * p.C * p.Tx, Ahi, Alo, part);
    if (!early) {
        __syncthreads();                      // (the staging barrier, taken early by this group)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int c = 16 * s + 8 * half + jj;
                qraw[s][jj] = (c < p.C && col_ok) ? Qb[(size_t)c * p.Ty + col] : 0.f;
            }
    }
    bf16x8 bhi[KS], blo[KS];
    float qn = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float v = qraw[s][jj];
            qn += v * v;
            __bf16 h, l;
            split_bf16(v, h, l);
            bhi[s][jj] = h;
            blo[s][jj] = l;
        }
    qn += __shfl_xor(qn, 32);
    SA_STAMP(1);

    if (!MULTI) {
        if (early) __syncthreads();
        // per-row additive term of the base-2 logit (fixed summation order); -inf masks rows >= t_x
        for (int il = tid; il < G * 32; il += SA_THREADS) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 2 * KS; ++j) acc += part[il * (2 * KS) + j];
            kn[il] = (il < tx) ? (l2 ? s2 * acc : 0.f) : NEG_INF_F;
        }
        __syncthreads();
        SA_STAMP(2);
        // lg = logit * log2(e) = acc*dmul + (bias[row] + qterm); bias carries the row mask (-inf)
        const float qterm = l2 ? s2 * qn : 0.f;
        const float dmul = l2 ? -2.0f * s2 : s2;
        float lg[G][16];
        float m = NEG_INF_F, l = 0.f;                 // running max / sum of 2^(lg - m) over this lane's rows
        f32x16 acc = tile_dot<KS>(Ahi, Alo, bhi, blo, lane);
#pragma unroll
        for (int r = 0; r < G; ++r) {
            // issue the next tile's MFMAs before this tile's VALU epilogue so the two overlap
            f32x16 acc_next;
            if (r + 1 < G) acc_next = tile_dot<KS>(Ahi + (r + 1) * KS * 64, Alo + (r + 1) * KS * 64, bhi, blo, lane);
            float tmax = NEG_INF_F;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                // rows 8*gq + 4*half + (0..3) of the tile: one ds_read_b128 of their bias
                const float4 bz = *reinterpret_cast<const float4 *>(kn + 32 * r + 8 * gq + 4 * half);
                lg[r][4 * gq + 0] = fmaf(acc[4 * gq + 0], dmul, bz.x + qterm);
                lg[r][4 * gq + 1] = fmaf(acc[4 * gq + 1], dmul, bz.y + qterm);
                lg[r][4 * gq + 2] = fmaf(acc[4 * gq + 2], dmul, bz.z + qterm);
                lg[r][4 * gq + 3] = fmaf(acc[4 * gq + 3], dmul, bz.w + qterm);
                tmax = fmaxf(fmaxf(tmax, fmaxf(lg[r][4 * gq + 0], lg[r][4 * gq + 1])),
                             fmaxf(lg[r][4 * gq + 2], lg[r][4 * gq + 3]));
            }
            const float mn = fmaxf(m, tmax);
            const float ms = (mn == NEG_INF_F) ? 0.f : mn;
            float ts = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) ts += __builtin_amdgcn_exp2f(lg[r][e] - ms);
            l = l * __builtin_amdgcn_exp2f((m == NEG_INF_F ? ms : m) - ms) + ts;
            m = mn;
            if (r + 1 < G) acc = acc_next;
        }
        SA_STAMP(3);
        // merge the two half-waves (rows 4*half offset) of each column
        const float m_o = __shfl_xor(m, 32), l_o = __shfl_xor(l, 32);
        const float m_all = fmaxf(m, m_o);
        const float m_fin = (m_all == NEG_INF_F) ? 0.f : m_all;
        const float l_all = (m == NEG_INF_F ? 0.f : l * __builtin_amdgcn_exp2f(m - m_fin)) +
                            (m_o == NEG_INF_F ? 0.f : l_o * __builtin_amdgcn_exp2f(m_o - m_fin));
        const float lse2 = m_fin + __builtin_amdgcn_logf(l_all);   // v_log_f32 = log2
        SA_STAMP(4);

        // per-lane base + wave-uniform row offsets keep the addresses out of VGPRs; bounds tests
        // are hoisted: the column test once per lane, the row test only for the last partial tile
        const size_t lane_off = ((size_t)b * p.Tx + 4 * half) * p.Ty + col;
        const int i_lane = 4 * half;
        const float c0 = -lse2 * LN2_F;
        float m2 = NEG_INF_F;
        if (col_ok) {
#pragma unroll
            for (int r = 0; r < G; ++r) {
                if (32 * r >= p.Tx) break;                                   // uniform
                const bool full = 32 * r + 32 <= p.Tx;                       // uniform
                if (p.prior) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                        float v = fmaf(lg[r][e], LN2_F, c0);
                        if (full || i_lane + iu < p.Tx) {
                            v += __logf(p.prior[lane_off + (size_t)iu * p.Ty] + 1e-8f);
                            p.logp[lane_off + (size_t)iu * p.Ty] = v;
                        }
                        lg[r][e] = v;
                        m2 = fmaxf(m2, v);
                    }
                } else if (full) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                        const float v = fmaf(lg[r][e], LN2_F, c0);
                        lg[r][e] = v;
                        p.logp[lane_off + (size_t)iu * p.Ty] = v;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                        const float v = fmaf(lg[r][e], LN2_F, c0);
                        lg[r][e] = v;
                        if (i_lane + iu < p.Tx) p.logp[lane_off + (size_t)iu * p.Ty] = v;
                    }
                }
            }
        }
        SA_STAMP(5);
        if (p.soft) {
            // softmax over text of the final log-probs (== exp(logp) when there is no prior)
            float lse2 = 0.f;
            if (p.prior) {
                m2 = fmaxf(m2, __shfl_xor(m2, 32));
                const float m2m = (m2 == NEG_INF_F) ? 0.f : m2;
                float s2 = 0.f;
#pragma unroll
                for (int r = 0; r < G; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s2 += __expf(lg[r][e] - m2m);
                s2 += __shfl_xor(s2, 32);
                lse2 = m2m + __logf(s2);
            }
#pragma unroll
            for (int r = 0; r < G; ++r) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                    if ((i_lane + iu < p.Tx) && col_ok)
                        p.soft[lane_off + (size_t)iu * p.Ty] = __expf(lg[r][e] - lse2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        const int RT = (p.Tx + 31) / 32;
        const int NG = (RT + G - 1) / G;
        float m_run = NEG_INF_F, l_run = 0.f;
        for (int g = 0; g < NG; ++g) {
            const int row0 = 32 * G * g;
            __syncthreads();
            stage_text_group<KS, G>(p, b, g, Ahi, Alo, kn, tx, l2, s2);
            __syncthreads();
            for (int r = 0; r < G; ++r) {
                float lg[16];
                tile_logits<KS>(lg, Ahi + r * KS * 64, Alo + r * KS * 64, bhi, blo, kn + 32 * r, qn, s2, l2,
                                row0 + 32 * r + 4 * half, tx, lane);
                float tm = m_run;
#pragma unroll
                for (int e = 0; e < 16; ++e) tm = fmaxf(tm, lg[e]);
                if (tm != NEG_INF_F) {
                    float ls = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ls += __expf(lg[e] - tm);
                    l_run = (m_run == NEG_INF_F ? 0.f : l_run * __expf(m_run - tm)) + ls;
                    m_run = tm;
                }
            }
        }
        float lse;
        {
            const float m_o = __shfl_xor(m_run, 32), l_o = __shfl_xor(l_run, 32);
            const float m_all = fmaxf(m_run, m_o);
            const float m_fin = (m_all == NEG_INF_F) ? 0.f : m_all;
            const float l_all = (m_run == NEG_INF_F ? 0.f : l_run * __expf(m_run - m_fin)) +
                                (m_o == NEG_INF_F ? 0.f : l_o * __expf(m_o - m_fin));
            lse = m_fin + __logf(l_all);
        }
        for (int g = 0; g < NG; ++g) {
            const int row0 = 32 * G * g;
            __syncthreads();
            stage_text_group<KS, G>(p, b, g, Ahi, Alo, kn, tx, l2, s2);
            __syncthreads();
            for (int r = 0; r < G; ++r) {
                float lg[16];
                const int i_lane = row0 + 32 * r + 4 * half;
                tile_logits<KS>(lg, Ahi + r * KS * 64, Alo + r * KS * 64, bhi, blo, kn + 32 * r, qn, s2, l2,
                                i_lane, tx, lane);
                const size_t lane_off = ((size_t)b * p.Tx + i_lane) * p.Ty + col;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int iu = (e & 3) + 8 * (e >> 2);
                    const bool ok = (i_lane + iu < p.Tx) && col_ok;
                    float v = lg[e] - lse;
                    if (p.prior && ok) v += __logf(p.prior[lane_off + (size_t)iu * p.Ty] + 1e-8f);
                    if (ok) {
                        p.logp[lane_off + (size_t)iu * p.Ty] = v;
                        if (p.soft) p.soft[lane_off + (size_t)iu * p.Ty] = __expf(v);   // no prior here (host checks)
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------------
// 1-D convolution of the text / mel encoders ("same" zero padding, K odd) on the matrix cores:
// y[b,o,t] = act(bias[o] + sum_{i,k} w[o,i,k] x[b,i,t+k-K/2]) as a GEMM with
// M = out channels, N = frames, reduction over (in channel, tap).  fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, same 64 FLOP/clk/SIMD as the vector ALU but one
// VGPR per operand and no VALU issue per FMA).  Workgroup tile 128 x 128 (4 waves, each 2x2 MFMA
// tiles of 32x32), input channels staged through LDS 16 at a time: x rows with their halo once,
// w as [ (i,k) ][ o ] so that both fragment reads are one conflict-free ds_read_b32 per lane.
// --------------------------------------------------------------------------
constexpr int CM_CI = 16;

// WO x WT waves, each AO x AT MFMA tiles of 32x32: workgroup tile (32*WO*AO) out channels x (32*WT*AT) frames.
// <2,2,2,2> = 128x128 for wide layers; <3,2,1,1> = 96x64 (six waves) for the narrow (<= 96 channel) ones.
template <int K, int WO, int WT, int AO, int AT>
__global__ __launch_bounds__(WO * WT * 64) void conv1d_mfma_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ w,
                                                                     const float *__restrict__ bias,
                                                                     float *__restrict__ y, int Cin, int Cout, int T,
                                                                     int relu) {
    constexpr int CM_TO = 32 * WO * AO, CM_TT = 32 * WT * AT, NTHR = WO * WT * 64;
    constexpr int HALO = K / 2;
    constexpr int XLD = CM_TT + 2 * HALO + 1;                 // odd-ish row stride
    constexpr int WLD = CM_TO + 4;                            // [ik][o], padded
    __shared__ float xs[CM_CI * XLD];
    __shared__ float wsm[CM_CI * K * WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * CM_TO, t0 = blockIdx.x * CM_TT;
    const int wo = (wave / WT) * (32 * AO), wt = (wave % WT) * (32 * AT);   // this wave's corner inside the tile
    const float *xb = x + (size_t)b * Cin * T;
    f32x16 acc[AO][AT];
#pragma unroll
    for (int a = 0; a < AO; ++a)
#pragma unroll
        for (int c = 0; c < AT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

    // register-staged pipeline: the next chunk's global loads are in flight while the MFMA loop
    // runs on the current LDS image (issue early / write late)
    constexpr int XN = (CM_CI * (CM_TT + 2 * HALO) + NTHR - 1) / NTHR;
    constexpr int WN = (CM_CI * K * CM_TO + NTHR - 1) / NTHR;
    float xr[XN], wr[WN];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int j = 0; j < XN; ++j) {
            const int idx = tid + NTHR * j;
            const int ii = idx / (CM_TT + 2 * HALO), tl = idx - ii * (CM_TT + 2 * HALO);
            const int t = t0 + tl - HALO, i = i0 + ii;
            xr[j] = (idx < CM_CI * (CM_TT + 2 * HALO) && i < Cin && t >= 0 && t < T) ? xb[(size_t)i * T + t] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int idx = tid + NTHR * j;
            const int ol = idx / (CM_CI * K), ik = idx - ol * (CM_CI * K);      // ik = ii*K + k: contiguous in w
            const int o = o0 + ol, i = i0 + ik / K;
            wr[j] = (idx < CM_CI * K * CM_TO && o < Cout && i < Cin) ? w[((size_t)o * Cin + i0) * K + ik] : 0.f;
        }
    };
    fetch(0);
    for (int i0 = 0; i0 < Cin; i0 += CM_CI) {
        __syncthreads();                                       // previous chunk's fragment reads are done
#pragma unroll
        for (int j = 0; j < XN; ++j) {
            const int idx = tid + NTHR * j;
            const int ii = idx / (CM_TT + 2 * HALO), tl = idx - ii * (CM_TT + 2 * HALO);
            if (idx < CM_CI * (CM_TT + 2 * HALO)) xs[ii * XLD + tl] = xr[j];
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int idx = tid + NTHR * j;
            const int ol = idx / (CM_CI * K), ik = idx - ol * (CM_CI * K);
            if (idx < CM_CI * K * CM_TO) wsm[ik * WLD + ol] = wr[j];
        }
        __syncthreads();
        if (i0 + CM_CI < Cin) fetch(i0 + CM_CI);
#pragma unroll
        for (int kk = 0; kk < CM_CI * K; kk += 2) {
            // this lane's reduction index: kk + half  ->  (in channel, tap)
            const int ik = kk + half;
            const int ii = ik / K, tap = ik - ii * K;
            float af[AO], bf[AT];
#pragma unroll
            for (int a = 0; a < AO; ++a) af[a] = wsm[ik * WLD + wo + 32 * a + l31];         // A[o][ik]
#pragma unroll
            for (int c = 0; c < AT; ++c) bf[c] = xs[ii * XLD + wt + 32 * c + l31 + tap];     // B[ik][t]
#pragma unroll
            for (int a = 0; a < AO; ++a)
#pragma unroll
                for (int c = 0; c < AT; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[c], acc[a][c], 0, 0, 0);
        }
    }
    // C/D layout: col = lane&31 (frame), row = (e&3) + 8*(e>>2) + 4*half (out channel)
#pragma unroll
    for (int a = 0; a < AO; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int o = o0 + wo + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * half;
            if (o >= Cout) continue;
            const float bv = bias ? bias[o] : 0.f;
#pragma unroll
            for (int c = 0; c < AT; ++c) {
                const int t = t0 + wt + 32 * c + l31;
                if (t < T) {
                    float v = acc[a][c][e] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    y[((size_t)b * Cout + o) * T + t] = v;
                }
            }
        }
}

struct SaLayout { size_t hi_off, lo_off, kn_off, total; int RT, KS; };

static SaLayout sa_layout(int B, int C, int Tx) {
    SaLayout L;
    L.RT = (Tx + 31) / 32;
    const int ks = (C + 15) / 16;
    L.KS = ks <= 5 ? 5 : ks <= 8 ? 8 : 16;
    const size_t frag = (size_t)B * L.RT * L.KS * 64 * sizeof(uint4);
    L.hi_off = 0;
    L.lo_off = align_up(frag, 256);
    L.kn_off = L.lo_off + align_up(frag, 256);
    L.total = L.kn_off + align_up((size_t)B * L.RT * 32 * sizeof(float), 256);
    return L;
}

template <int KS, int G, bool MULTI>
static int launch_softattn(const SoftAttnParams &p, unsigned char *ws, const SaLayout &L, hipStream_t s) {
    if (MULTI) {          // row-group path restages the text operand per group: prepare it once
        hipLaunchKernelGGL(softattn_prep_kernel<KS>, dim3(L.RT, p.B), dim3(256), 0, s, p.keys,
                           reinterpret_cast<uint4 *>(ws + L.hi_off), reinterpret_cast<uint4 *>(ws + L.lo_off),
                           reinterpret_cast<float *>(ws + L.kn_off), p.C, p.Tx, L.RT);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    const size_t lds = (size_t)2 * G * KS * 64 * sizeof(uint4) + (size_t)G * 32 * sizeof(float) +
                       (MULTI ? 0 : (size_t)G * 32 * 2 * KS * sizeof(float));
    auto kern = softattn_kernel<KS, G, MULTI>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    dim3 grid((p.Ty + 32 * SA_WAVES - 1) / (32 * SA_WAVES), p.B), block(SA_THREADS);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

}  // namespace aligner

using namespace aligner;

extern "C" {

size_t aligner_softattn_workspace_bytes(int B, int C, int Tx) {
    if (B < 0 || C < 1 || Tx < 1) return 0;
    return sa_layout(B, C, Tx).total;
}

int aligner_softattn_f32(const float *keys, const float *queries, const int32_t *t_xs, const float *prior,
                         float *logp_out, float *soft_out, void *workspace, size_t workspace_bytes, int B, int C,
                         int Tx, int Ty, float temperature, int sim, void *stream) {
    if (!keys || !queries || !logp_out || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || C < 1 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d C=%d Tx=%d Ty=%d", B, C, Tx, Ty);
    if (sim != ALIGNER_SIM_L2 && sim != ALIGNER_SIM_DOT) return fail(ALIGNER_EINVAL, "bad sim %d", sim);
    if (C > 256) return fail(ALIGNER_EDOM, "C=%d exceeds 256 attention channels", C);
    if (B > 65535) return fail(ALIGNER_EDOM, "B=%d too large", B);
    if (B == 0) return ALIGNER_OK;
    const SaLayout L = sa_layout(B, C, Tx);
    if (workspace_bytes < L.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, L.total);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    SoftAttnParams p{keys, queries, t_xs, prior, logp_out, soft_out,
                     reinterpret_cast<const uint4 *>(ws + L.hi_off), reinterpret_cast<const uint4 *>(ws + L.lo_off),
                     reinterpret_cast<const float *>(ws + L.kn_off), L.RT, g_debug_stamps, B, C, Tx, Ty, temperature,
                     sim};
    const int G = L.KS <= 8 ? 7 : 4;
    const bool multi = L.RT > G;
    if (soft_out && prior && multi)
        return fail(ALIGNER_EDOM, "soft output with a prior needs Tx <= %d", 32 * G);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (L.KS == 5) return multi ? launch_softattn<5, 7, true>(p, ws, L, s) : launch_softattn<5, 7, false>(p, ws, L, s);
    if (L.KS == 8) return multi ? launch_softattn<8, 7, true>(p, ws, L, s) : launch_softattn<8, 7, false>(p, ws, L, s);
    return multi ? launch_softattn<16, 4, true>(p, ws, L, s) : launch_softattn<16, 4, false>(p, ws, L, s);
}

int aligner_conv1d_f32(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int Cout,
                       int T, int K, int relu, void *stream) {
    if (!x || !w || !y) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Cin < 1 || Cout < 1 || T < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (B == 0) return ALIGNER_OK;
    if (B > 65535) return fail(ALIGNER_EDOM, "grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (K != 1 && K != 3 && K != 5) return fail(ALIGNER_EDOM, "kernel size %d not supported (1, 3, 5)", K);
    if (Cout > 96) {
        dim3 grid((T + 127) / 128, (Cout + 127) / 128, B), block(256);
        if (grid.y > 65535) return fail(ALIGNER_EDOM, "grid too large");
        if (K == 1) hipLaunchKernelGGL((conv1d_mfma_kernel<1, 2, 2, 2, 2>), grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu);
        if (K == 3) hipLaunchKernelGGL((conv1d_mfma_kernel<3, 2, 2, 2, 2>), grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu);
        if (K == 5) hipLaunchKernelGGL((conv1d_mfma_kernel<5, 2, 2, 2, 2>), grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu);
    } else {
        dim3 grid((T + 63) / 64, 1, B), block(384);
        if (K == 1) hipLaunchKernelGGL((conv1d_mfma_kernel<1, 3, 2, 1, 1>), grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu);
        if (K == 3) hipLaunchKernelGGL((conv1d_mfma_kernel<3, 3, 2, 1, 1>), grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu);
        if (K == 5) hipLaunchKernelGGL((conv1d_mfma_kernel<5, 3, 2, 1, 1>), grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu);
    }
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

}  // extern "C"
