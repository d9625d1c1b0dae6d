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
    int B, C, Tx, Ty;
    float temperature;
    int sim;
};

__device__ __forceinline__ void split_bf16(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

constexpr float NEG_INF_F = -__builtin_huge_valf();

// Row group -> LDS: split the text operand to bf16 hi/lo in MFMA A-fragment order
// (A[i = lane&31][k = 8*(lane>>5)+jj] for row tile r, k-step s) and |k_i|^2.
template <int KS, int G>
__device__ __forceinline__ void stage_text_group(const SoftAttnParams &p, const float *Kb, int row0,
                                                 bf16x8 *Ahi, bf16x8 *Alo, float *kn) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < G * KS * 64; idx += 256) {
        const int ln = idx & 63;
        const int s = (idx >> 6) % KS;
        const int r = (idx >> 6) / KS;
        const int i = row0 + 32 * r + (ln & 31);
        const int c0 = 16 * s + 8 * (ln >> 5);
        bf16x8 h, l;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            float v = 0.f;
            if (i < p.Tx && c0 + jj < p.C) v = Kb[(size_t)(c0 + jj) * p.Tx + i];
            __bf16 hh, ll;
            split_bf16(v, hh, ll);
            h[jj] = hh;
            l[jj] = ll;
        }
        Ahi[idx] = h;
        Alo[idx] = l;
    }
    for (int il = tid; il < G * 32; il += 256) {
        const int i = row0 + il;
        float sacc = 0.f;
        if (i < p.Tx)
            for (int c = 0; c < p.C; ++c) {
                const float v = Kb[(size_t)c * p.Tx + i];
                sacc += v * v;
            }
        kn[il] = sacc;
    }
}

// logits of one 32-row tile for this lane's column: 3x bf16 MFMA + distance/scale/mask.
// C/D layout: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
template <int KS>
__device__ __forceinline__ void tile_logits(float (&lg)[16], const bf16x8 *Ahi_r, const bf16x8 *Alo_r,
                                            const bf16x8 (&bhi)[KS], const bf16x8 (&blo)[KS],
                                            const float *kn_r, float qn, float scale, bool l2,
                                            int i_lane0, int tx, int lane) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 ah = Ahi_r[s * 64 + lane];
        const bf16x8 al = Alo_r[s * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[s], acc, 0, 0, 0);
    }
    const int half4 = 4 * (lane >> 5);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int il = (e & 3) + 8 * (e >> 2) + half4;
        const float d = acc[e];
        const float v = l2 ? scale * ((kn_r[il] + qn) - 2.0f * d) : scale * d;
        lg[e] = (i_lane0 + (e & 3) + 8 * (e >> 2) < tx) ? v : NEG_INF_F;
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
template <int KS, int G, bool MULTI>
__global__ __launch_bounds__(256) void softattn_kernel(SoftAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8 *Ahi = reinterpret_cast<bf16x8 *>(smem);            // [G][KS][64]
    bf16x8 *Alo = Ahi + G * KS * 64;                           // [G][KS][64]
    float *kn = reinterpret_cast<float *>(Alo + G * KS * 64);  // [G*32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;
    const int b = blockIdx.y;
    const int col = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool col_ok = col < p.Ty;
    int tx = p.Tx;
    if (p.t_xs) {
        tx = p.t_xs[b];
        tx = tx < 0 ? 0 : (tx > p.Tx ? p.Tx : tx);
    }
    const float *Kb = p.keys + (size_t)b * p.C * p.Tx;
    const float *Qb = p.queries + (size_t)b * p.C * p.Ty;
    const bool l2 = (p.sim == ALIGNER_SIM_L2);
    const float scale = l2 ? -p.temperature : p.temperature;

    bf16x8 bhi[KS], blo[KS];
    const float qn = load_mel_fragments<KS>(p, Qb, col, col_ok, half, bhi, blo);

    if (!MULTI) {
        stage_text_group<KS, G>(p, Kb, 0, Ahi, Alo, kn);
        __syncthreads();
        float lg[G][16];
        float m = NEG_INF_F;
#pragma unroll
        for (int r = 0; r < G; ++r) {
            tile_logits<KS>(lg[r], Ahi + r * KS * 64, Alo + r * KS * 64, bhi, blo, kn + 32 * r, qn, scale, l2,
                            32 * r + 4 * half, tx, lane);
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, lg[r][e]);
            __builtin_amdgcn_sched_barrier(0);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mm = (m == NEG_INF_F) ? 0.f : m;
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < G; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) l += __expf(lg[r][e] - mm);
        l += __shfl_xor(l, 32);
        const float lse = mm + __logf(l);

        // per-lane base + wave-uniform row offsets keep the addresses out of VGPRs
        const size_t lane_off = ((size_t)b * p.Tx + 4 * half) * p.Ty + col;
        const int i_lane = 4 * half;
        float m2 = NEG_INF_F;
#pragma unroll
        for (int r = 0; r < G; ++r) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                const bool ok = (i_lane + iu < p.Tx) && col_ok;
                float v = lg[r][e] - lse;
                if (p.prior && ok) v += __logf(p.prior[lane_off + (size_t)iu * p.Ty] + 1e-8f);
                lg[r][e] = v;
                m2 = fmaxf(m2, v);
                if (ok) p.logp[lane_off + (size_t)iu * p.Ty] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
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
            stage_text_group<KS, G>(p, Kb, row0, Ahi, Alo, kn);
            __syncthreads();
            for (int r = 0; r < G; ++r) {
                float lg[16];
                tile_logits<KS>(lg, Ahi + r * KS * 64, Alo + r * KS * 64, bhi, blo, kn + 32 * r, qn, scale, l2,
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
            stage_text_group<KS, G>(p, Kb, row0, Ahi, Alo, kn);
            __syncthreads();
            for (int r = 0; r < G; ++r) {
                float lg[16];
                const int i_lane = row0 + 32 * r + 4 * half;
                tile_logits<KS>(lg, Ahi + r * KS * 64, Alo + r * KS * 64, bhi, blo, kn + 32 * r, qn, scale, l2,
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
// 1-D convolution of the text / mel encoders ("same" zero padding, K odd):
//   y[b,o,t] = act(bias[o] + sum_{i,k} w[o,i,k] x[b,i,t+k-K/2])
// LDS-tiled fp32: a workgroup computes a 64(out-channels) x 64(frames) tile of
// one utterance; x rows (with halo) and w slices are staged through LDS in
// chunks of 16 input channels; each thread owns a 4x4 register tile.
// --------------------------------------------------------------------------
constexpr int CV_TO = 64, CV_TT = 64, CV_CI = 16;

template <int K>
__global__ __launch_bounds__(256) void conv1d_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ bias, float *__restrict__ y,
                                                      int Cin, int Cout, int T, int relu) {
    constexpr int HALO = K / 2;
    constexpr int XW = CV_TT + 2 * HALO;
    __shared__ float xs[CV_CI][XW + 1];
    __shared__ float wsm[CV_CI * K][CV_TO + 1];     // [i*K+k][o]
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * CV_TO, t0 = blockIdx.x * CV_TT;
    const int to = (tid >> 4) * 4, tt = (tid & 15) * 4;   // 16x16 threads, 4x4 each
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    const float *xb = x + (size_t)b * Cin * T;

    for (int i0 = 0; i0 < Cin; i0 += CV_CI) {
        __syncthreads();
        for (int idx = tid; idx < CV_CI * XW; idx += 256) {
            const int ii = idx / XW, tl = idx - ii * XW;
            const int t = t0 + tl - HALO, i = i0 + ii;
            xs[ii][tl] = (i < Cin && t >= 0 && t < T) ? xb[(size_t)i * T + t] : 0.f;
        }
        for (int idx = tid; idx < CV_CI * K * CV_TO; idx += 256) {
            const int ol = idx / (CV_CI * K), ik = idx - ol * (CV_CI * K);   // ik = ii*K + k (contiguous in w)
            const int o = o0 + ol, i = i0 + ik / K;
            wsm[ik][ol] = (o < Cout && i < Cin) ? w[((size_t)o * Cin + i0) * K + ik] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int ii = 0; ii < CV_CI; ++ii) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float wv[4], xv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) wv[a] = wsm[ii * K + k][to + a];
#pragma unroll
                for (int c = 0; c < 4; ++c) xv[c] = xs[ii][tt + c + k];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] = fmaf(wv[a], xv[c], acc[a][c]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int o = o0 + to + a;
        if (o >= Cout) continue;
        const float bv = bias ? bias[o] : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t = t0 + tt + c;
            if (t < T) {
                float v = acc[a][c] + bv;
                if (relu) v = fmaxf(v, 0.f);
                y[((size_t)b * Cout + o) * T + t] = v;
            }
        }
    }
}

template <int KS, int G, bool MULTI>
static int launch_softattn(const SoftAttnParams &p, hipStream_t s) {
    const size_t lds = (size_t)2 * G * KS * 64 * sizeof(bf16x8) + (size_t)G * 32 * sizeof(float);
    auto kern = softattn_kernel<KS, G, MULTI>;
    if (lds > 64 * 1024)
        ALIGNER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((p.Ty + 127) / 128, p.B), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

}  // namespace aligner

using namespace aligner;

extern "C" {

int aligner_softattn_f32(const float *keys, const float *queries, const int32_t *t_xs, const float *prior,
                         float *logp_out, float *soft_out, int B, int C, int Tx, int Ty, float temperature,
                         int sim, void *stream) {
    if (!keys || !queries || !logp_out) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || C < 1 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d C=%d Tx=%d Ty=%d", B, C, Tx, Ty);
    if (sim != ALIGNER_SIM_L2 && sim != ALIGNER_SIM_DOT) return fail(ALIGNER_EINVAL, "bad sim %d", sim);
    if (C > 256) return fail(ALIGNER_EDOM, "C=%d exceeds 256 attention channels", C);
    if (B > 65535) return fail(ALIGNER_EDOM, "B=%d too large", B);
    if (B == 0) return ALIGNER_OK;
    SoftAttnParams p{keys, queries, t_xs, prior, logp_out, soft_out, B, C, Tx, Ty, temperature, sim};
    const int RT = (Tx + 31) / 32;
    const int ks = (C + 15) / 16;
    const int G = ks <= 8 ? 7 : 4;
    const bool multi = RT > G;
    if (soft_out && prior && multi)
        return fail(ALIGNER_EDOM, "soft output with a prior needs Tx <= %d", 32 * G);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ks <= 5) return multi ? launch_softattn<5, 7, true>(p, s) : launch_softattn<5, 7, false>(p, s);
    if (ks <= 8) return multi ? launch_softattn<8, 7, true>(p, s) : launch_softattn<8, 7, false>(p, s);
    return multi ? launch_softattn<16, 4, true>(p, s) : launch_softattn<16, 4, false>(p, s);
}

int aligner_conv1d_f32(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int Cout,
                       int T, int K, int relu, void *stream) {
    if (!x || !w || !y) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Cin < 1 || Cout < 1 || T < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (B == 0) return ALIGNER_OK;
    if (B > 65535 || (Cout + CV_TO - 1) / CV_TO > 65535) return fail(ALIGNER_EDOM, "grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((T + CV_TT - 1) / CV_TT, (Cout + CV_TO - 1) / CV_TO, B), block(256);
    switch (K) {
        case 1: hipLaunchKernelGGL(conv1d_kernel<1>, grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu); break;
        case 3: hipLaunchKernelGGL(conv1d_kernel<3>, grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu); break;
        case 5: hipLaunchKernelGGL(conv1d_kernel<5>, grid, block, 0, s, x, w, bias, y, Cin, Cout, T, relu); break;
        default: return fail(ALIGNER_EDOM, "kernel size %d not supported (1, 3, 5)", K);
    }
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

}  // extern "C"
