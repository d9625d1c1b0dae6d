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
#include <cstdlib>
#include <type_traits>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

// convgemm.hip: the wide layers' GEMM-structured convolution (operands pre-split, LDS-DMA staging)
bool conv_gemm_applies(int Cin, int Cout, int K);
size_t conv_gemm_prepared_bytes(int Cout, int Cin, int K);
size_t conv_gemm_workspace_bytes(int B, int Cin, int Cout, int T, int K);
int conv_gemm_prepare(const float *w, void *prepared, int Cout, int Cin, int K, hipStream_t s);
int conv_gemm_run(const float *x, const void *prepared, const float *bias, float *y, void *workspace, size_t workspace_bytes,
                  int B, int Cin, int Cout, int T, int K, int relu, hipStream_t s);
struct ConvStackLayer { const void *prepared; const float *bias; int Cin, Cout, K, relu; };
size_t conv_stack_workspace_bytes(const ConvStackLayer *L, int n, int B, int T);
int conv_stack_run(const float *x, const ConvStackLayer *L, int n, float *y, void *workspace, size_t workspace_bytes, int B, int T,
                   hipStream_t s);

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct SoftAttnParams {
    const float *keys;      // [B,C,Tx]
    const float *queries;   // [B,C,Ty]
    const int   *t_xs;      // nullable
    const float *prior;     // nullable [B,Tx,Ty]
    float *logp;            // [B,Tx,Ty] fp32, or bf16 when out16 (same shape)
    float *soft;            // nullable
    const uint4 *frag_hi;   // [B][RT][KS][64] text operand, bf16 high halves in MFMA A-fragment order
    const uint4 *frag_lo;   // same, low halves
    const float *knorm;     // [B][RT*32] |k_i|^2
    int RT;                 // row tiles of 32 text rows
    unsigned long long *stamps;   // debug (nullable): [blocks][4 waves][8] shader clock
    int B, C, Tx, Ty;
    float temperature;
    int sim;
    int out16;              // logp is written as bf16 (round to nearest even)
    int pair;               // row-group form on a small batch: 2 or 4 waves share a 32-frame strip, each a part of the row tiles (0: one wave)
    int ldo;                // row pitch of logp in elements (>= Ty; the row-tile form only: aligner_softattn_ld)
};

constexpr int SA_WAVES = 8;                       // waves per workgroup: 8 x 32 = 256 mel frames share one staged text operand
constexpr int SA_THREADS = SA_WAVES * 64;
constexpr float NEG_INF_F = -__builtin_huge_valf();
constexpr float LOG2E_F = 1.4426950408889634f, LN2_F = 0.6931471805599453f;

// v or +0.0 without a select on the loaded value (a select fed by a load is turned back into a
// branch around the load, which serialises the loads)
__device__ __forceinline__ float and_mask(float v, unsigned m) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m);
}

// Hand-issued loads for the two operand fetches: with compiler-issued loads the scheduler wraps every
// masked load in "load, s_waitcnt vmcnt(0), select" and the 40 loads of a lane run one after the other.
// The compiler does not see these loads' vmcnt; consumers go through wait_regs8, which carries the
// data dependence ("+v") behind an explicit s_waitcnt (the compiler's own waits stay conservative:
// vmcnt retires in order).
__device__ __forceinline__ void gload_dword(float &dst, const float *sbase, unsigned voff_bytes) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
template <int CNT>
__device__ __forceinline__ void wait_regs8(float (&r)[8]) {       // CNT < 0: dependence only, no wait
    if (CNT >= 0)
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                     : "n"(CNT < 0 ? 0 : (CNT > 63 ? 63 : CNT)) : "memory");
    else
        asm volatile(""
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
}

__device__ __forceinline__ void split_bf16(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// one log-prob to logp[idx]: fp32, or bf16 (RNE) when the caller asked for 16-bit log-probs
__device__ __forceinline__ void store_logp(const SoftAttnParams &p, size_t idx, float v) {
    if (p.out16) reinterpret_cast<__bf16 *>(p.logp)[idx] = (__bf16)v;
    else         p.logp[idx] = v;
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
// Wave r owns row tile r: lane (i = lane&31, hh = lane>>5) converts one half of the channels of text
// row 32r+i, so every global load instruction reads two 128-byte runs of K[c][.], the norm |k_i|^2 is
// one shuffle away (fixed summation order) and a single barrier publishes fragments + bias.
// Loads and conversion are separate calls so that the caller can put its mel loads between them:
// vmcnt retires in order, so the text data (issued first) is converted while the mel loads fly.
template <int KS>
struct TextStage {
    static constexpr int CHG = KS <= 8 ? KS : 8;          // 8-channel groups per thread and pass
    static constexpr int NCH = KS / CHG;                   // passes (2 for KS == 16)
    float raw[CHG][8];
    float nrm = 0.f;

    __device__ __forceinline__ void load(const SoftAttnParams &p, const float *Kb, int pass, int wave, int lane) {
        const int i = 32 * wave + (lane & 31), hh = lane >> 5;
        const int ic = i < p.Tx ? i : p.Tx - 1;            // clamped: every load is unconditional (no exec branches)
#pragma unroll
        for (int gi = 0; gi < CHG; ++gi)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int c = 8 * (hh * KS + pass * CHG + gi) + jj;
                const float v = Kb[(size_t)(c < p.C ? c : p.C - 1) * p.Tx + ic];
                raw[gi][jj] = and_mask(v, (i < p.Tx && c < p.C) ? ~0u : 0u);
            }
    }
    // C == 16*KS (no channel padding), single pass: uniform row base + one lane offset for all loads
    __device__ __forceinline__ void load_fast(const SoftAttnParams &p, const float *Kb, int wave, int lane) {
        const int i = 32 * wave + (lane & 31), hh = lane >> 5;
        const int ic = i < p.Tx ? i : p.Tx - 1;
        const unsigned voff = 4u * (unsigned)(hh * 8 * KS * p.Tx + ic);
#pragma unroll
        for (int gi = 0; gi < CHG; ++gi)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) gload_dword(raw[gi][jj], Kb + (size_t)(8 * gi + jj) * p.Tx, voff);
    }
    // wait until at most YOUNGER younger loads are outstanding, then zero the rows >= Tx
    template <int YOUNGER>
    __device__ __forceinline__ void settle_fast(const SoftAttnParams &p, int wave, int lane) {
        const unsigned m = (32 * wave + (lane & 31) < p.Tx) ? ~0u : 0u;
        wait_regs8<YOUNGER>(raw[0]);
#pragma unroll
        for (int gi = 1; gi < CHG; ++gi) wait_regs8<-1>(raw[gi]);
#pragma unroll
        for (int gi = 0; gi < CHG; ++gi)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) raw[gi][jj] = and_mask(raw[gi][jj], m);
    }
    __device__ __forceinline__ void convert(uint4 *Ahi, uint4 *Alo, int pass, int wave, int lane) {
        const int hh = lane >> 5;
#pragma unroll
        for (int gi = 0; gi < CHG; ++gi) {
            const int g = hh * KS + pass * CHG + gi;       // 8-channel group of this row: k-step g>>1, half g&1
            bf16x8 h, l;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float v = raw[gi][jj];
                nrm = fmaf(v, v, nrm);
                __bf16 hv, lv;
                split_bf16(v, hv, lv);
                h[jj] = hv;
                l[jj] = lv;
            }
            const int o = (wave * KS + (g >> 1)) * 64 + (lane & 31) + 32 * (g & 1);
            Ahi[o] = __builtin_bit_cast(uint4, h);
            Alo[o] = __builtin_bit_cast(uint4, l);
        }
    }
};

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
            p.stamps[((size_t)blockIdx.x * SA_WAVES + (threadIdx.x >> 6)) * 8 + (k)] = \
                __builtin_amdgcn_s_memtime();                                                         \
    } while (0)

template <int KS, int G, bool MULTI>
__global__ __launch_bounds__(SA_THREADS, (KS == 16 && MULTI) ? 1 : 2) void softattn_kernel(SoftAttnParams p) {
    static_assert(G <= SA_WAVES, "one staging wave per row tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *Ahi = reinterpret_cast<uint4 *>(smem);              // [G][KS][64] bf16x8 fragments
    uint4 *Alo = Ahi + G * KS * 64;                            // [G][KS][64]
    float *kn = reinterpret_cast<float *>(Alo + G * KS * 64);  // [G*32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5;
    // 1-D grid of NQ frame ranges x B utterances.  Workgroups are dealt round-robin over the 8 XCDs
    // (observed, speed only): keep the NQ workgroups that re-read one utterance's text operand on the
    // same XCD so its L2 fetches that operand once instead of once per XCD.
    // (row-group form, p.pair: the workgroup covers 128 frames, waves w and w + 4 share strip w and take the even / odd
    // row tiles of every group -- a strip of long text is a long serial job of ONE wave otherwise, and a small batch
    // leaves the chip's other SIMDs idle meanwhile)
    const int nsp = (MULTI && p.pair > 1) ? p.pair : 1;            // waves per strip
    const bool pair = nsp > 1;
    const int swv = SA_WAVES / nsp;                               // strips per workgroup
    const int NQ = (p.Ty + 32 * swv - 1) / (32 * swv);
    int b, fq;
    if ((p.B & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        fq = slot % NQ;
        b = (slot / NQ) * 8 + xcd;
    } else {
        b = blockIdx.x / NQ;
        fq = blockIdx.x % NQ;
    }
    const int col = fq * (32 * swv) + (wave % swv) * 32 + (lane & 31);
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
    if (p.stamps && (threadIdx.x & 63) == 0)          // slot 7: entry on the 100 MHz clock all XCDs share
        p.stamps[((size_t)blockIdx.x * SA_WAVES + (threadIdx.x >> 6)) * 8 + 7] =
            __builtin_amdgcn_s_memrealtime();
    // The waves of a workgroup form two groups.  Group A (waves 0..3, one per SIMD) issues its mel
    // loads right behind its share of the text loads and computes at raised priority; group B issues
    // its mel loads only after the staging barrier and fills the issue slots A leaves, so that B's
    // loads and MFMA/softmax phase overlap A's compute and A's store phase (all workgroups start
    // together: without the stagger the whole GPU alternates between an HBM-read, a compute and an
    // HBM-write phase).
    const bool early = MULTI || wave < SA_WAVES / 2;
    float qraw[KS][8];
    const int colc = col_ok ? col : p.Ty - 1;              // clamped: unconditional loads, select afterwards
    auto load_mel = [&]() {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int c = 16 * s + 8 * half + jj;
                const float v = Qb[(size_t)(c < p.C ? c : p.C - 1) * p.Ty + colc];
                qraw[s][jj] = and_mask(v, (c < p.C && col_ok) ? ~0u : 0u);
            }
    };
    // fast operand fetch: no channel padding and few enough loads for the 6-bit vmcnt
    const bool fastld = !MULTI && KS <= 8 && p.C == 16 * KS;              // uniform
    auto load_mel_fast = [&]() {
        const unsigned voff = 4u * (unsigned)(8 * half * p.Ty + colc);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) gload_dword(qraw[s][jj], Qb + (size_t)(16 * s + jj) * p.Ty, voff);
    };
    auto settle_mel_fast = [&]() {
        const unsigned mk = col_ok ? ~0u : 0u;
        wait_regs8<0>(qraw[0]);
#pragma unroll
        for (int s = 1; s < KS; ++s) wait_regs8<-1>(qraw[s]);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) qraw[s][jj] = and_mask(qraw[s][jj], mk);
    };
    if (MULTI) {
        load_mel();
    } else {
        TextStage<KS> ts;
        const float *Kb = p.keys + (size_t)b * p.C * p.Tx;
        if (fastld) {
            if (KS <= 8) {
                if (wave < G) ts.load_fast(p, Kb, wave, lane);
                if (early) {
                    load_mel_fast();
                    if (wave < G) ts.template settle_fast<8 * KS>(p, wave, lane);    // text in, mel still in flight
                } else if (wave < G) {
                    ts.template settle_fast<0>(p, wave, lane);
                }
            }
        } else {
            if (wave < G) ts.load(p, Kb, 0, wave, lane);
            if (early) load_mel();
        }
        if (wave < G) {
            ts.convert(Ahi, Alo, 0, wave, lane);
            if (TextStage<KS>::NCH == 2) {
                ts.load(p, Kb, 1, wave, lane);
                ts.convert(Ahi, Alo, 1, wave, lane);
            }
            // per-row additive term of the base-2 logit: s2*|k_i|^2 (L2) or 0 (dot); -inf masks rows >= t_x
            const float nrm = ts.nrm + __shfl_xor(ts.nrm, 32);
            const int i = 32 * wave + (lane & 31);
            if (lane < 32) kn[i] = (i < tx) ? (l2 ? s2 * nrm : 0.f) : NEG_INF_F;
        }
        __syncthreads();
        SA_STAMP(1);
        // group B fetches its mel strip now and parks at a second barrier until group A has finished
        // its matrix + softmax phase: a SIMD's two waves would otherwise split the matrix pipe and A's
        // stores -- the first bytes this workgroup can write -- would start twice as late
        if (fastld) {
            if (KS <= 8) {
                if (!early) {
                    load_mel_fast();
                    __syncthreads();
                }
                settle_mel_fast();
            }
        } else if (!early) {
            load_mel();
            __syncthreads();
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

    if (!MULTI) {
        SA_STAMP(2);
        // lg = logit * log2(e) up to a per-column constant (s2*|q_j|^2 cancels in the softmax over the
        // text axis): acc*dmul + bias[row]; bias carries the row mask (-inf)
        const float dmul = l2 ? -2.0f * s2 : s2;
        (void)qn;
        float lg[G][16];
        float m = NEG_INF_F, l = 0.f;                 // running max / sum of 2^(lg - m) over this lane's rows
        // One wave per SIMD has to overlap its own matrix and vector work: left alone the compiler runs
        // all MFMAs and then all VALU (and sched_group_barrier pipelines came out lumpy), so the issue
        // order is written out: MFMA k of tile r+1, then slice k of tile r's epilogue, with a scheduling
        // fence after every slot.  Fragments are fetched one k-step ahead, the bias one tile ahead.
        auto frag = [&](const uint4 *A, int r, int s) { return __builtin_bit_cast(bf16x8, A[(r * KS + s) * 64 + lane]); };
        auto bias4 = [&](int r, int gq) { return *reinterpret_cast<const float4 *>(kn + 32 * r + 8 * gq + 4 * half); };
        constexpr int NS = 3 * KS;                    // MFMA slots per tile
        // slot 0 is left to the last MFMA of the previous tile (its result is not readable yet), slots
        // 1..4: fma + max of one accumulator quarter each, slot 5: tile statistics, E0..NS-1: exp2 of the
        // 16 logits, each sum one slot behind its exp (no back-to-back dependent transcendental)
        constexpr int F0 = 1, E0 = 6;
        bf16x8 ah = frag(Ahi, 0, 0), al = frag(Alo, 0, 0), ahn = ah, aln = al;
        float4 bz[4] = {bias4(0, 0), bias4(0, 1), bias4(0, 2), bias4(0, 3)};
        f32x16 acc, d;
        auto mfma_slot = [&](int rn, int k) {         // MFMA k of tile rn into d
            const int s = k / 3, j = k % 3;
            if (j == 0) {
                if (s + 1 < KS) { ahn = frag(Ahi, rn, s + 1); aln = frag(Alo, rn, s + 1); }
                else if (rn + 1 < G) { ahn = frag(Ahi, rn + 1, 0); aln = frag(Alo, rn + 1, 0); }
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[s], d, 0, 0, 0);
            } else if (j == 1) {
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[s], d, 0, 0, 0);
            } else {
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[s], d, 0, 0, 0);
                ah = ahn; al = aln;
            }
        };
#pragma unroll
        for (int e = 0; e < 16; ++e) d[e] = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) mfma_slot(0, k);              // tile 0: overlaps the mel operand split
        acc = d;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const bool nxt = r + 1 < G;
            float tmax = NEG_INF_F, ms = 0.f, ts = 0.f, tp = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = 0.f;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (nxt) mfma_slot(r + 1, k);
                if (k >= F0 && k < F0 + 4) {
                    const int gq = k - F0;
                    lg[r][4 * gq + 0] = fmaf(acc[4 * gq + 0], dmul, bz[gq].x);
                    lg[r][4 * gq + 1] = fmaf(acc[4 * gq + 1], dmul, bz[gq].y);
                    lg[r][4 * gq + 2] = fmaf(acc[4 * gq + 2], dmul, bz[gq].z);
                    lg[r][4 * gq + 3] = fmaf(acc[4 * gq + 3], dmul, bz[gq].w);
                    tmax = fmaxf(fmaxf(tmax, lg[r][4 * gq + 0]), lg[r][4 * gq + 1]);      // v_max3_f32
                    tmax = fmaxf(fmaxf(tmax, lg[r][4 * gq + 2]), lg[r][4 * gq + 3]);
                    if (nxt) bz[gq] = bias4(r + 1, gq);
                } else if (k == F0 + 4) {
                    const float mn = fmaxf(m, tmax);
                    ms = (mn == NEG_INF_F) ? 0.f : mn;
                    l = l * __builtin_amdgcn_exp2f((m == NEG_INF_F ? ms : m) - ms);
                    m = mn;
                }
                if (k >= E0) {
                    if (k > E0) ts = (k == E0 + 1) ? tp : ts + tp;      // the previous slot's exps
                    bool first = true;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (E0 + e * (NS - E0) / 16 == k) {
                            const float x = __builtin_amdgcn_exp2f(lg[r][e] - ms);
                            tp = first ? x : tp + x;
                            first = false;
                        }
                    if (first) tp = 0.f;               // a slot without a logit (NS - E0 > 16)
                }
                if (k == NS - 1) l += ts + tp;
                asm volatile("" : "+v"(l), "+v"(ts), "+v"(tp), "+v"(tmax));   // keep the slice here (the IR sink pass would move it)
                __builtin_amdgcn_sched_barrier(0);
            }
            acc = d;
        }
        if (early) __syncthreads();                   // releases group B (see above)
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
        const bool plain = !p.prior && !p.soft;                              // uniform
        if (plain) {
            // the common case: every store unconditional.  The utterance's [Tx,Ty] block is a buffer resource, so a
            // store past it (the text rows >= Tx of the last tile) is dropped by the hardware; a lane whose frame does
            // not exist (col >= Ty) carries an offset beyond any block.  No bounds test, no branch per tile or per
            // store (a uniform branch costs a wave ~20 cycles, an exec-masked store a save / branch / restore):
            // a uniform row offset (scalar, a compile-time multiple of Ty) plus one 32-bit lane offset.
            const unsigned esz = p.out16 ? 2u : 4u;
            const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<unsigned char *>(p.logp) + (size_t)b * p.Tx * p.Ty * esz, 0,
                (unsigned)p.Tx * (unsigned)p.Ty * esz, 0x00020000);
            const unsigned lane_byte = col_ok ? (unsigned)(4 * half * p.Ty + col) * esz : 0x80000000u;
            const unsigned row_bytes = (unsigned)p.Ty * esz;
            if (!p.out16) {                                                  // uniform, once
#pragma unroll
                for (int r = 0; r < G; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaf(lg[r][e], LN2_F, c0)), out_rs,
                                                              lane_byte, (unsigned)iu * row_bytes, 0);
                    }
            } else {                                                         // bf16 log-probs: half the write traffic
#pragma unroll
                for (int r = 0; r < G; ++r)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                        const __bf16 hv = (__bf16)fmaf(lg[r][e], LN2_F, c0);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), out_rs, lane_byte,
                                                              (unsigned)iu * row_bytes, 0);
                    }
            }
        }
        if (col_ok && !plain) {
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const bool full = 32 * r + 32 <= p.Tx;                       // uniform
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
                    float v = fmaf(lg[r][e], LN2_F, c0);
                    if (full || i_lane + iu < p.Tx) {
                        if (p.prior) v += __logf(p.prior[lane_off + (size_t)iu * p.Ty] + 1e-8f);
                        store_logp(p, lane_off + (size_t)iu * p.Ty, v);
                    }
                    lg[r][e] = v;
                    m2 = fmaxf(m2, v);
                }
            }
        }
        SA_STAMP(5);
        if (p.stamps) {                       // debug only: when have this wave's stores left the CU?
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SA_STAMP(6);
            if ((threadIdx.x & 63) == 0)              // slot 4 (overwritten): drained, on the 100 MHz clock
                p.stamps[((size_t)blockIdx.x * SA_WAVES + (threadIdx.x >> 6)) * 8 + 4] =
                    __builtin_amdgcn_s_memrealtime();
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
            stage_text_group<KS, G>(p, b, g, Ahi, Alo, kn, tx, l2, s2);
            __syncthreads();
            for (int r = wave / swv; r < G; r += nsp) {
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
        if (pair) {                                   // the strip's other wave: its running (max, sum) through LDS
            __syncthreads();                          // (the last group's fragments are read)
            float2 *sx = reinterpret_cast<float2 *>(Ahi);
            sx[threadIdx.x] = make_float2(m_run, l_run);
            __syncthreads();
            // every wave of the strip combines the parts in the same order: they must agree on the normaliser to the bit
            float m_all = NEG_INF_F;
            for (int q = 0; q < nsp; ++q) m_all = fmaxf(m_all, sx[(q * swv + wave % swv) * 64 + lane].x);
            const float m_fin = (m_all == NEG_INF_F) ? 0.f : m_all;
            float l_all = 0.f;
            for (int q = 0; q < nsp; ++q) {
                const float2 o = sx[(q * swv + wave % swv) * 64 + lane];
                l_all += (o.x == NEG_INF_F) ? 0.f : o.y * __expf(o.x - m_fin);
            }
            l_run = l_all;
            m_run = m_all;
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
            for (int r = wave / swv; r < G; r += nsp) {
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
                        store_logp(p, lane_off + (size_t)iu * p.Ty, v);
                        if (p.soft) p.soft[lane_off + (size_t)iu * p.Ty] = __expf(v);   // no prior here (host checks)
                    }
                }
            }
        }
    }
}

// --------------------------------------------------------------------------
// Row-tile form of the one-row-group case (Tx <= 32*NT): the kernel BASELINE configs[1] and [2] run.
//
// softattn_kernel<.,.,false> above gives a wave a strip of 32 frames and ALL row tiles: its first byte leaves the CU
// after the text staging AND a whole strip's arithmetic, and since every workgroup of the grid starts together the
// chip's memory system sees a read phase, a silent phase and a write phase (profiles/r02g_softattn_stamps.txt: the
// first store at 15 k of 29.5 k cycles).  Here the roles are turned by 90 degrees:
//   * wave w < NT owns ROW TILE w (32 text rows) for the whole kernel: its A fragments (bf16 hi / lo of K) live in
//     registers -- the text operand never touches LDS -- and its 16 bias values (s2*|k_i|^2, -inf for masked rows) too;
//   * the workgroup walks its 256 frames strip by strip (32 frames); for a strip every compute wave does the 3*KS
//     MFMAs of ITS tile, the strip's softmax statistics (max, sum of 2^(x - max) per frame) meet through LDS: one
//     barrier per strip, and each wave stores its own 32 x 32 tile.  The first stores leave one tile's arithmetic
//     after the text operand arrived, and from then on stores, arithmetic and the next strips' loads overlap;
//   * wave 0 is the loader: the mel strips come global -> LDS by LDS-DMA (raw fp32, all of them in flight at once, no
//     registers) and the loader splits a landed strip into the B fragments (bf16 hi / lo) of a ring slot.  The compute
//     waves are bound by their VALU issue (a SIMD issues one vector instruction per 4 cycles whichever of its two waves
//     it comes from: 2 x ~290 instructions a strip were 2.4 k cycles), the loader's SIMD hosts one compute wave only:
//     everything that is not a tile's own arithmetic belongs to the loader.
// Barrier k (k = -1 .. NS-1) is the only synchronisation: before barrier k the loader has written strip k + 2's
// fragments into its ring slot and every compute wave has published its statistics of strip k; after it the compute
// waves read strip k + 1's fragments (ring of three slots: strip k + 2 is being written meanwhile) and strip k's
// statistics (two buffers).
// --------------------------------------------------------------------------
constexpr int RT_STRIPS = 8;      // 32-frame strips per workgroup
// raw strips in LDS (in flight by LDS-DMA).  16 channel groups of 5 k-steps: all of a workgroup's strips -- they are
// fetched while the memory system is still idle (before the first stores), and the store phase is a pure write stream
constexpr int rt_raw_slots(int KS) { return KS <= 5 ? RT_STRIPS : 4; }
constexpr int RT_RING = 3;        // split strips in LDS
// the log-probs are stored sc1: written through the XCD's L2, the line dropped from it.  Measured (rocprofv3, one batch at
// a time, same box): plain 17.5 us, sc0 17.7, sc1 16.4, sc0 sc1 16.4, nt 29.3 (rows are 4000 bytes apart: a 128-byte
// run straddles two lines); the search that reads them next is not slower (32.3 against 32.5 us: Infinity Cache)
constexpr int RT_ST_AUX = 16;

__device__ __forceinline__ void rt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses (sbase + voff) to LDS [lds_dst + 16*lane].  The
// instruction's offset field moves BOTH addresses, so up to four 1 KB pieces share one M0 setting: piece i of a group
// lands at lds_dst + 1024 i and its base pointer is passed less 1024 i (M0 save / restore and the wait states around
// it were most of a piece's issue cost: ~50 cycles each, 60 pieces before the first strip could be waited for)
template <int N>
__device__ __forceinline__ void rt_dma_group(unsigned lds_dst, unsigned voff, const unsigned char *b0, const unsigned char *b1,
                                             const unsigned char *b2, const unsigned char *b3) {
    unsigned keep;
    if (N == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %5 offset:2048\n\tglobal_load_lds_dwordx4 %1, %6 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(b0), "s"(b1 - 1024), "s"(b2 - 2048), "s"(b3 - 3072) : "memory");
    else if (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\tglobal_load_lds_dwordx4 %1, %4 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(b0), "s"(b1 - 1024) : "memory");
}
// a [16 KS channels][32 positions] fp32 block global -> LDS: piece n = channels 8n .. 8n+7 (lane: channel 8n + lane/8,
// positions 4 (lane%8) .. +3), `pstep` bytes between two pieces' channel groups in memory
template <int KS>
__device__ __forceinline__ void rt_dma_block(unsigned dst, unsigned voff, const unsigned char *base, size_t pstep) {
    static_assert((2 * KS) % 4 == 0 || (2 * KS) % 4 == 2, "pieces in groups of four, then two");
#pragma unroll
    for (int g = 0; g + 4 <= 2 * KS; g += 4)
        rt_dma_group<4>(dst + g * 1024, voff, base + g * pstep, base + (g + 1) * pstep, base + (g + 2) * pstep, base + (g + 3) * pstep);
    if ((2 * KS) % 4 == 2)
        rt_dma_group<2>(dst + (2 * KS - 2) * 1024, voff, base + (2 * KS - 2) * pstep, base + (2 * KS - 1) * pstep, nullptr, nullptr);
}
template <int CNT>
__device__ __forceinline__ void rt_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); }

// both halves of a wave combined: every lane gets (lower half's value, upper half's value) of its column
__device__ __forceinline__ void rt_halves(float v, float &lo, float &hi) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned r0 = r[0], r1 = r[1];              // (bit_cast of a vector ELEMENT reads element 0: DESIGN.md rule 10)
    lo = __builtin_bit_cast(float, r0);
    hi = __builtin_bit_cast(float, r1);
}

// a word of LDS for the polls (a volatile access through a generic pointer is a FLAT load: it waits for vmcnt(0),
// i.e. for every store the wave has in flight)
typedef __attribute__((address_space(3))) unsigned rt_lds_u32;
// "the frame's constant has not been published yet" (a quiet NaN no arithmetic produces)
constexpr unsigned RT_PENDING = 0x7FC0DEADu;
// the NT tiles' (max, sum of 2^(x - max)) of a frame merged: -(max + log2 sum) * ln2
template <int NT>
__device__ __forceinline__ float rt_merge(const float2 (&st)[NT]) {
    float M = st[0].x;
#pragma unroll
    for (int w = 1; w < NT; ++w) M = fmaxf(M, st[w].x);
    const float Mf = (M == NEG_INF_F) ? 0.f : M;
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < NT; ++w) L = fmaf(st[w].y, __builtin_amdgcn_exp2f(st[w].x - Mf), L);
    return -(Mf + __builtin_amdgcn_logf(L)) * LN2_F;           // v_log_f32 = log2
}

// debug stamps: [workgroup][wave][64]: 0 entry, 1 operands staged, 2 first strip's MFMAs done, 5 end, 6 stores drained,
// 4 / 7 drained / entry on the 100 MHz clock; 8 + 4j, 9 + 4j: arrival at / release from barrier j
#define RT_STAMP(k)                                                                                           \
    do {                                                                                                      \
        if (STAMPS && p.stamps && (threadIdx.x & 63) == 0)                                                    \
            p.stamps[((size_t)blockIdx.x * (NT + 1) + (threadIdx.x >> 6)) * 64 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// STAMPS: the development build with the cycle stamps (aligner_debug_set_stamps); in the product's instantiation they are
// compiled out -- four "is the stamp buffer set" branches a strip were ~130 cycles of every wave's 2 400
template <int KS, int NT, bool OUT16, bool STAMPS = false>
__global__ __launch_bounds__((NT + 1) * 64) void softattn_rt_kernel(SoftAttnParams p) {
    constexpr int SLOTB = KS * 2048;                  // one strip: raw [16 KS channels][32 frames] fp32 == split [KS][hi, lo][64] x 16 B
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RT_RAW = rt_raw_slots(KS);
    unsigned char *raw = smem;                                                     // [RT_RAW][SLOTB]
    unsigned char *frg = smem + RT_RAW * SLOTB;                                    // [RT_RING][SLOTB]
    float2 *stat = reinterpret_cast<float2 *>(frg + RT_RING * SLOTB);              // [2][NT][32] (max, sum) per frame
    float *knl = reinterpret_cast<float *>(stat + 2 * NT * 32);                    // [NT][32] row terms
    float *c0buf = knl + NT * 32;                                                  // [2][32] -lse*ln2 per frame, or RT_PENDING

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int NQ = (p.Ty + 32 * RT_STRIPS - 1) / (32 * RT_STRIPS);
    int b, fq;
    if ((p.B & 7) == 0) {                             // the workgroups of one utterance on one XCD (see softattn_kernel)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        fq = slot % NQ;
        b = (slot / NQ) * 8 + xcd;
    } else {
        b = blockIdx.x / NQ;
        fq = blockIdx.x % NQ;
    }
    // the utterance's strips dealt evenly to its NQ workgroups (T_mel = 900: 29 strips as 8, 7, 7, 7 -- not 8, 8, 8, 5: the
    // launch ends with its longest workgroup)
    const int nstrips = (p.Ty + 31) / 32, sbase = nstrips / NQ, srem = nstrips - sbase * NQ;
    const int NS = sbase + (fq < srem ? 1 : 0);                                    // strips of this workgroup (>= 1, <= RT_STRIPS)
    const int f0 = 32 * (fq * sbase + (fq < srem ? fq : srem));
    const float *Qb = p.queries + (size_t)b * p.C * p.Ty;
    RT_STAMP(0);
    if (STAMPS && p.stamps && lane == 0)
        p.stamps[((size_t)blockIdx.x * (NT + 1) + wave) * 64 + 7] = __builtin_amdgcn_s_memrealtime();

    // wave 0 is the loader (the first wave's loads are the first the CU's load path serves: the first strip is out
    // while the text tiles stream in); wave t + 1 owns row tile t
    const int tile = wave - 1;
    if (wave == 0) {
        // ---------------- loader ----------------
        const unsigned raw0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)raw;
        const int cl = lane >> 3, q4 = lane & 7;      // DMA piece n: channel 8n + cl, frames 4*q4 .. 4*q4+3 of the strip
        int issued = 0;                               // strips whose DMAs have been issued
        auto issue_upto = [&](int n) {
            for (; issued < n && issued < NS; ++issued) {
                int fr = f0 + 32 * issued + 4 * q4;
                fr = fr < p.Ty - 4 ? fr : p.Ty - 4;   // Ty % 4 == 0: a quad is all in or all out; frames past Ty: any valid quad
                rt_dma_block<KS>(raw0 + (unsigned)(issued % RT_RAW) * SLOTB, 4u * (unsigned)(cl * p.Ty + fr),
                                 reinterpret_cast<const unsigned char *>(Qb), (size_t)32 * p.Ty);
            }
        };
        // raw strip s has landed: the DMAs of the strips issued after it may stay in flight (vmcnt counts instructions,
        // in order; its field has 6 bits)
        auto wait_for = [&](int s) {
            const int later = issued - 1 - s;         // uniform
            constexpr int P = 2 * KS;
            if (later >= 5) rt_wait_vm<(5 * P < 63 ? 5 * P : 63)>();
            else if (later == 4) rt_wait_vm<(4 * P < 63 ? 4 * P : 63)>();
            else if (later == 3) rt_wait_vm<(3 * P < 63 ? 3 * P : 63)>();
            else if (later == 2) rt_wait_vm<2 * P>();
            else if (later == 1) rt_wait_vm<P>();
            else rt_wait_vm<0>();
        };
        // raw strip s -> B fragments in ring slot s % RT_RING: lane (frame l31, channel half) takes channels
        // 16k + 8 half + jj of k-step k.  All the LDS reads first, then the arithmetic
        auto split = [&](int s) {
            const float *R = reinterpret_cast<const float *>(raw + (s % RT_RAW) * SLOTB) + half * 8 * 32 + l31;
            uint4 *F = reinterpret_cast<uint4 *>(frg + (s % RT_RING) * SLOTB) + lane;
            float v[KS][8];
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[k][jj] = R[(16 * k + jj) * 32];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                bf16x8 h, l;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    __bf16 hv, lv;
                    split_bf16(v[k][jj], hv, lv);
                    h[jj] = hv;
                    l[jj] = lv;
                }
                F[(2 * k) * 64] = __builtin_bit_cast(uint4, h);
                F[(2 * k + 1) * 64] = __builtin_bit_cast(uint4, l);
            }
        };
        // the NT tiles' statistics of strip s merged into the frame's constant -lse*ln2 (what the compute waves add
        // to ln2 * logit): every wave needs it, one wave works it out
        auto merge = [&](int s) {
            if (p.pair < 0) return;                   // testing ("softattn_rt_drop_merge"): every compute wave falls back to its own merge
            float2 st[NT];
#pragma unroll
            for (int w = 0; w < NT; ++w) st[w] = stat[((s & 1) * NT + w) * 32 + l31];
            c0buf[(s & 1) * 32 + l31] = rt_merge<NT>(st);
        };
        static_assert(RT_RAW <= 8, "wait_for's cases: at most 6 strips in flight");
        c0buf[lane] = __builtin_bit_cast(float, RT_PENDING);      // both buffers
        // The first two strips come through registers, in fragment order (lane: frame l31, channel half; channels
        // 16k + 8 half + jj), and the loader converts them as they are: plain loads return 1.2 k cycles after the
        // kernel's start, the first LDS-DMA instructions of a wave took 3 k cycles to issue.
        float q0[KS][8], q1[KS][8];
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(Qb), 0, (unsigned)p.C * (unsigned)p.Ty * 4u, 0x00020000);
        auto load_strip = [&](float (&v)[KS][8], int s) {
            int cf = f0 + 32 * s + l31;
            cf = cf < p.Ty ? cf : p.Ty - 1;
            const unsigned vo = 4u * (unsigned)(8 * half * p.Ty + cf);
#pragma unroll
            for (int k = 0; k < KS; ++k)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    v[k][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, vo, 4u * (unsigned)((16 * k + jj) * p.Ty), 0));
        };
        auto to_ring = [&](const float (&v)[KS][8], int s) {
            uint4 *F = reinterpret_cast<uint4 *>(frg + (s % RT_RING) * SLOTB) + lane;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                bf16x8 h, l;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    __bf16 hv, lv;
                    split_bf16(v[k][jj], hv, lv);
                    h[jj] = hv;
                    l[jj] = lv;
                }
                F[(2 * k) * 64] = __builtin_bit_cast(uint4, h);
                F[(2 * k + 1) * 64] = __builtin_bit_cast(uint4, l);
            }
        };
        load_strip(q0, 0);
        to_ring(q0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_strip(q1, 1);                            // (behind strip 0's conversion: nothing of it in front of the first strip)
        RT_STAMP(1);
        rt_barrier();                                 // barrier A: strip 0 is in the ring
        issued = 2;
        issue_upto(3);                                // strip 2 by LDS-DMA while strip 1's loads land; the others behind barrier B
        RT_STAMP(11);
        to_ring(q1, 1);
        RT_STAMP(14);
        rt_barrier();                                 // barrier B (= -1): strip 1 is in the ring
        for (int j = 0; j < NS; ++j) {
            if (j > 0) merge(j - 1);                  // first: the compute waves' stores of strip j-1 wait for it
            if (lane < 32) c0buf[(j & 1) * 32 + lane] = __builtin_bit_cast(float, RT_PENDING);   // (strip j-2's: read before barrier j-1)
            // strips < j + 2 have been split: their raw slots are free, and every slot is filled as soon as it is (with
            // eight slots: the whole mel block is asked for in iteration 0, before the first store leaves the CU; on one box,
            // seven alternating runs each: 16.0 against 16.3 us for two more strips an iteration; one more: 17.2)
            issue_upto(j + 2 + RT_RAW);
            if (j + 2 < NS) {
                wait_for(j + 2);
                split(j + 2);
            }
            RT_STAMP(8 + 4 * j);
            rt_barrier();                             // barrier j
            RT_STAMP(9 + 4 * j);
        }
        merge(NS - 1);
        RT_STAMP(5);
        return;
    }

    // ---------------- compute wave: row tile `tile` ----------------
    int tx = p.Tx;
    if (p.t_xs) {
        tx = p.t_xs[b];
        tx = tx < 0 ? 0 : (tx > p.Tx ? p.Tx : tx);
    }
    const bool l2 = (p.sim == ALIGNER_SIM_L2);
    const float scale = l2 ? -p.temperature : p.temperature;
    const float s2 = scale * LOG2E_F;                 // logits are kept in base 2
    const float dmul = l2 ? -2.0f * s2 : s2;
    bf16x8 ah[KS], al[KS];
    float4 bz[4];
    // A fragments: lane (row i = 32 tile + l31, channel half) holds channels 16s + 8 half + jj.  The utterance's
    // [C,Tx] block is a buffer resource: channels >= C read as zero.  Rows >= Tx read row Tx - 1 again (finite): their
    // logits are -inf through the bias whatever the products are.
    const int row_i = 32 * tile + l31;
    float kv[KS][8];
    {
        const float *Kb = p.keys + (size_t)b * p.C * p.Tx;
        {
            const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(Kb), 0, (unsigned)p.C * (unsigned)p.Tx * 4u, 0x00020000);
            const int ic = row_i < p.Tx ? row_i : p.Tx - 1;
            const unsigned voff = 4u * (unsigned)(8 * half * p.Tx + ic);
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    kv[s][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(krs, voff, 4u * (unsigned)((16 * s + jj) * p.Tx), 0));
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                __bf16 hv, lv;
                split_bf16(kv[s][jj], hv, lv);
                ah[s][jj] = hv;
                al[s][jj] = lv;
            }
        }
    }
    {
        // |k_i|^2 and the bias: the compute waves wait for the loader's first strip here anyway
        float nrm = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) nrm = fmaf(kv[k][jj], kv[k][jj], nrm);
        float n0, n1;
        rt_halves(nrm, n0, n1);
        nrm = n0 + n1;
        // per-row additive term of the base-2 logit: s2*|k_i|^2 (L2) or 0 (dot); -inf masks rows >= t_x.  Through LDS
        // into the accumulator layout (rows (e&3) + 8*(e>>2) + 4*half): this wave's words only, no barrier
        if (lane < 32) knl[tile * 32 + lane] = (row_i < tx) ? (l2 ? s2 * nrm : 0.f) : NEG_INF_F;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bz[gq] = *reinterpret_cast<const float4 *>(knl + tile * 32 + 8 * gq + 4 * half);
    }
    RT_STAMP(1);
    rt_barrier();                                     // barrier A: strip 0 is in the ring

    bf16x8 bh[KS], bl[KS];
    auto read_b = [&](int s) {
        const uint4 *F = reinterpret_cast<const uint4 *>(frg + (s % RT_RING) * SLOTB) + lane;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            bh[k] = __builtin_bit_cast(bf16x8, F[(2 * k) * 64]);
            bl[k] = __builtin_bit_cast(bf16x8, F[(2 * k + 1) * 64]);
        }
    };
    auto read_bk = [&](int s, int k) {                // one k-step of strip s
        const uint4 *F = reinterpret_cast<const uint4 *>(frg + (s % RT_RING) * SLOTB) + lane;
        bh[k] = __builtin_bit_cast(bf16x8, F[(2 * k) * 64]);
        bl[k] = __builtin_bit_cast(bf16x8, F[(2 * k + 1) * 64]);
    };
    // the utterance's [Tx,Ty] block as a buffer resource: stores to rows >= Tx of the last tile are dropped by the
    // hardware, a lane whose frame does not exist carries an offset beyond any block (softattn_kernel's store path)
    constexpr unsigned esz = OUT16 ? 2u : 4u;
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<unsigned char *>(p.logp) + (size_t)b * p.Tx * p.ldo * esz, 0, (unsigned)p.Tx * (unsigned)p.ldo * esz,
        0x00020000);
    const unsigned row_bytes = (unsigned)p.ldo * esz;
    const unsigned tile_bytes = (unsigned)(32 * tile) * row_bytes;
    auto lane_byte_of = [&](int j) {
        const int col = f0 + 32 * j + l31;
        return col < p.Ty ? (unsigned)(4 * half * p.ldo + col) * esz : 0x80000000u;
    };
    auto out_store = [&](float v, unsigned lane_byte, int e) {
        const int iu = (e & 3) + 8 * (e >> 2);
        if (!OUT16) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rs, lane_byte, tile_bytes + (unsigned)iu * row_bytes, RT_ST_AUX);
        } else {
            const __bf16 hv = (__bf16)v;
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), out_rs, lane_byte,
                                                  tile_bytes + (unsigned)iu * row_bytes, RT_ST_AUX);
        }
    };

    f32x16 acc;
    {
        read_b(0);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[k], bh[k], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[k], bl[k], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[k], bh[k], acc, 0, 0, 0);
        }
    }
    rt_barrier();                                     // barrier B (= -1): strip 1 is in the ring
    RT_STAMP(2);
    // Strip j's iteration (between barriers j-1 and j) carries three strips: the MFMAs of strip j+1, the softmax
    // statistics of strip j (logits from the accumulators of the previous iteration's MFMAs) and the STORES of strip
    // j-1, whose normaliser the statistics published before barrier j-1 give.  The issue order is written out, one
    // MFMA per slot with a slice of the vector work and at most two stores behind it (a wave whose stores and
    // arithmetic alternate in phases leaves the CU's store path idle two thirds of the time: 4 k cycles a strip).
    // lgP / lgC: base-2 logits (up to the frame's constant) of strip j-1 / j; the two arrays swap roles from one
    // iteration to the next (the loop is unrolled by two: no copies).
    constexpr int NSL = 3 * KS;                       // MFMA slots of a strip
    constexpr int SS0 = 4;                            // first slot with a store of strip j-1 (the loader needs ~400 cycles for its constant)
    unsigned lane_byteP = 0x80000000u;                // strip j-1's lane offsets (no strip -1: the stores are dropped)
    // strip s's constant -lse*ln2 from the loader (c0buf), `first` = an earlier read of the word.  Polled a few times; a
    // loader that is late (or a reader that is early) costs the merge this wave would otherwise have done itself --
    // never a hang: the statistics are in LDS for every wave
    auto frame_constant = [&](int s, unsigned first) -> float {
        const volatile rt_lds_u32 *cp = (const volatile rt_lds_u32 *)(c0buf + (s & 1) * 32 + l31);
        unsigned cv = first;
        for (int t = 0; t < 8 && __builtin_amdgcn_ballot_w64(cv == RT_PENDING) != 0; ++t) cv = *cp;
        if (__builtin_amdgcn_ballot_w64(cv == RT_PENDING) != 0) {
            float2 st[NT];
#pragma unroll
            for (int w = 0; w < NT; ++w) st[w] = stat[((s & 1) * NT + w) * 32 + l31];
            return rt_merge<NT>(st);
        }
        return __builtin_bit_cast(float, cv);
    };
    auto body = [&](int j, float (&lgP)[16], float (&lgC)[16], auto more_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr bool MORE = decltype(more_tag)::value;      // strip j+1 exists: its MFMAs run here
        constexpr bool FIRST = decltype(first_tag)::value;    // strip 0: there is no strip -1 to store
        // ---- LDS reads first: strip j-1's constant (if it is there already), strip j+1's B fragments
        unsigned cfirst = RT_PENDING;
        if (!FIRST) cfirst = *(const volatile rt_lds_u32 *)(c0buf + ((j + 1) & 1) * 32 + l31);
        if (MORE) read_bk(j + 1, 0);                  // (the other k-steps' fragments one k-step ahead of their MFMAs: seven waves' reads
                                                      // of a whole strip at once are 70 KB the LDS serves for ~500 cycles behind the barrier)
        // ---- strip j: logits and their maximum
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            lgC[4 * gq + 0] = fmaf(acc[4 * gq + 0], dmul, bz[gq].x);
            lgC[4 * gq + 1] = fmaf(acc[4 * gq + 1], dmul, bz[gq].y);
            lgC[4 * gq + 2] = fmaf(acc[4 * gq + 2], dmul, bz[gq].z);
            lgC[4 * gq + 3] = fmaf(acc[4 * gq + 3], dmul, bz[gq].w);
        }
        float tmax = fmaxf(fmaxf(lgC[0], lgC[1]), lgC[2]);
#pragma unroll
        for (int e = 3; e < 15; e += 2) tmax = fmaxf(fmaxf(tmax, lgC[e]), lgC[e + 1]);
        tmax = fmaxf(tmax, lgC[15]);
        float t0, t1;
        rt_halves(tmax, t0, t1);
        asm("v_max_f32_e32 %0, %1, %2" : "=v"(tmax) : "v"(t0), "v"(t1));      // (fmaxf would first canonicalise both operands)
        const float ms = (tmax == NEG_INF_F) ? 0.f : tmax;
        RT_STAMP(10 + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
        // ---- the slots
        float c0P = 0.f, ls = 0.f, xe = 0.f;
        bool have_xe = false;
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            if (MORE) {
                const int s3 = k / 3, q3 = k % 3;
                if (q3 == 0 && s3 + 1 < KS) read_bk(j + 1, s3 + 1);
                if (k == 0) {
                    f32x16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], bh[0], z, 0, 0, 0);
                } else if (q3 == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s3], bh[s3], acc, 0, 0, 0);
                else if (q3 == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s3], bl[s3], acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s3], bh[s3], acc, 0, 0, 0);
            }
            if (!FIRST && k == SS0) c0P = frame_constant(j - 1, cfirst);
            // stores of strip j-1
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (!FIRST && SS0 + e * (NSL - SS0) / 16 == k) out_store(fmaf(lgP[e], LN2_F, c0P), lane_byteP, e);
            // strip j: 2^(x - max), each sum one slot behind its exp (no transcendental waited for)
            if (have_xe) { ls += xe; have_xe = false; }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (e * NSL / 16 == k) {
                    const float x = __builtin_amdgcn_exp2f(lgC[e] - ms);
                    xe = have_xe ? xe + x : x;
                    have_xe = true;
                }
            asm volatile("" : "+v"(ls), "+v"(xe));    // keep the slice here (the IR sink pass would move it)
            __builtin_amdgcn_sched_barrier(0);
        }
        RT_STAMP(11 + 4 * j);
        if (have_xe) ls += xe;
        rt_halves(ls, t0, t1);
        ls = t0 + t1;
        if (lane < 32) stat[((j & 1) * NT + tile) * 32 + lane] = make_float2(tmax, ls);
        lane_byteP = lane_byte_of(j);
        RT_STAMP(8 + 4 * j);
        rt_barrier();
        RT_STAMP(9 + 4 * j);
    };
    // the last strip's stores
    auto finish = [&](float (&lgP)[16]) __attribute__((always_inline)) {
        const unsigned cfirst = *(const volatile rt_lds_u32 *)(c0buf + ((NS - 1) & 1) * 32 + l31);
        const float c0 = frame_constant(NS - 1, cfirst);
#pragma unroll
        for (int e = 0; e < 16; ++e) out_store(fmaf(lgP[e], LN2_F, c0), lane_byteP, e);
    };
    float lgA[16], lgB[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) lgA[e] = 0.f;
    constexpr std::true_type T{};
    constexpr std::false_type F{};
    if (NS == 1) {
        body(0, lgA, lgB, F, T);
        finish(lgB);
    } else {
        body(0, lgA, lgB, T, T);
        int j = 1;
        for (; j + 2 < NS; j += 2) {
            body(j, lgB, lgA, T, F);
            body(j + 1, lgA, lgB, T, F);
        }
        if (j + 1 < NS) {
            body(j, lgB, lgA, T, F);
            body(j + 1, lgA, lgB, F, F);
            finish(lgB);
        } else {
            body(j, lgB, lgA, F, F);
            finish(lgA);
        }
    }
    RT_STAMP(5);
    if (STAMPS && p.stamps) {             // debug only: when have this wave's stores left the CU?
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RT_STAMP(6);
        if (lane == 0) p.stamps[((size_t)blockIdx.x * (NT + 1) + wave) * 64 + 4] = __builtin_amdgcn_s_memrealtime();
    }
}

// --------------------------------------------------------------------------
// Exact-product form of the same front end: the contraction on v_mfma_f32_32x32x2_f32 (fp32 operands, fp32
// fma chain -- no operand splitting), a quarter of the bf16x3 kernel's matrix rate.  The bf16x3 product
// (hi*hi + hi*lo + lo*hi, lo itself rounded to bf16) carries ~2^-16.5 relative error per product; the logit
// multiplies the summed products by 2*temperature (L2) or temperature (dot), so the 1e-4 bound on logp holds
// for the default temperature with a wide margin but not for sharp ones on large encodings (measured 5e-4 at
// temperature 0.05, |k|,|q| ~ 3 per channel).  Host rule (aligner_softattn): L2 with temperature > 0.002, or
// dot with temperature > 0.2, takes this kernel; so does the "softattn_exact" debug option.
// Plain structure (a correctness path): a workgroup = 4 waves = 128 frames of one utterance; text rows in
// groups of GE 32-row tiles staged to LDS as fp32 A fragments; two sweeps over the groups (running max / sum,
// then normalise and store), the contraction simply redone in the second.
// --------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void softattn_exact_kernel(SoftAttnParams p, int GE) {
    constexpr int S2 = 8 * KS;                                   // k-steps of 2 channels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Af = reinterpret_cast<float *>(smem);                 // [GE][S2][64]: K[2s + (lane>>5)][32(r0+r) + (lane&31)]
    float *kn = Af + (size_t)GE * S2 * 64;                       // [GE*32] row term (natural log units), -inf = masked
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int NQ = (p.Ty + 127) / 128;
    const int b = blockIdx.x / NQ, fq = blockIdx.x % NQ;
    const int col = fq * 128 + wave * 32 + (lane & 31);
    const bool col_ok = col < p.Ty;
    int tx = p.Tx;
    if (p.t_xs) {
        tx = p.t_xs[b];
        tx = tx < 0 ? 0 : (tx > p.Tx ? p.Tx : tx);
    }
    const bool l2 = (p.sim == ALIGNER_SIM_L2);
    const float scale = l2 ? -p.temperature : p.temperature;
    const float *Kb = p.keys + (size_t)b * p.C * p.Tx;
    const float *Qb = p.queries + (size_t)b * p.C * p.Ty;
    // mel operand: B fragment k = 2s + half, column = this lane's frame
    float qx[S2];
    float qn = 0.f;
#pragma unroll
    for (int s = 0; s < S2; ++s) {
        const int c = 2 * s + half;
        const float v = (c < p.C && col_ok) ? Qb[(size_t)c * p.Ty + col] : 0.f;
        qx[s] = v;
        qn = fmaf(v, v, qn);
    }
    qn += __shfl_xor(qn, 32);
    const int RT = (p.Tx + 31) / 32, NG = (RT + GE - 1) / GE;
    auto stage = [&](int g) {
        const int r0 = GE * g;
        for (int idx = tid; idx < GE * S2 * 64; idx += 256) {
            const int ln = idx & 63, sr = idx >> 6, st = sr % S2, r = sr / S2;
            const int i = 32 * (r0 + r) + (ln & 31), c = 2 * st + (ln >> 5);
            Af[idx] = (i < p.Tx && c < p.C) ? Kb[(size_t)c * p.Tx + i] : 0.f;
        }
        for (int il = tid; il < GE * 32; il += 256) {
            const int i = 32 * r0 + il;
            float v = NEG_INF_F;
            if (i < tx) {
                v = 0.f;
                if (l2) {
                    float nrm = 0.f;
                    for (int c = 0; c < p.C; ++c) { const float k = Kb[(size_t)c * p.Tx + i]; nrm = fmaf(k, k, nrm); }
                    v = scale * nrm;
                }
            }
            kn[il] = v;
        }
    };
    auto logits = [&](float (&lg)[16], int r) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float *A = Af + (size_t)r * S2 * 64 + lane;
#pragma unroll
        for (int s = 0; s < S2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s * 64], qx[s], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int il = 32 * r + (e & 3) + 8 * (e >> 2) + 4 * half;
            lg[e] = l2 ? fmaf(acc[e], -2.0f * scale, kn[il] + scale * qn) : fmaf(acc[e], scale, kn[il]);
        }
    };
    float m_run = NEG_INF_F, l_run = 0.f;
    for (int g = 0; g < NG; ++g) {
        __syncthreads();
        stage(g);
        __syncthreads();
        const int nt = (RT - GE * g < GE) ? RT - GE * g : GE;
        for (int r = 0; r < nt; ++r) {
            float lg[16];
            logits(lg, r);
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
    // with a prior the soft output needs a second normalisation: running max / sum of the final log-probs
    float m2 = NEG_INF_F, l2s = 0.f;
    const int npass = (p.soft && p.prior) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        float lse2 = 0.f;
        if (pass == 1) {
            const float m_o = __shfl_xor(m2, 32), l_o = __shfl_xor(l2s, 32);
            const float m_all = fmaxf(m2, m_o);
            const float m_fin = (m_all == NEG_INF_F) ? 0.f : m_all;
            lse2 = m_fin + __logf((m2 == NEG_INF_F ? 0.f : l2s * __expf(m2 - m_fin)) +
                                  (m_o == NEG_INF_F ? 0.f : l_o * __expf(m_o - m_fin)));
        }
        for (int g = 0; g < NG; ++g) {
            __syncthreads();
            stage(g);
            __syncthreads();
            const int nt = (RT - GE * g < GE) ? RT - GE * g : GE;
            for (int r = 0; r < nt; ++r) {
                float lg[16];
                logits(lg, r);
                const int i_lane = 32 * (GE * g + r) + 4 * half;
                const size_t lane_off = ((size_t)b * p.Tx + i_lane) * p.Ty + col;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int iu = (e & 3) + 8 * (e >> 2);
                    const bool ok = (i_lane + iu < p.Tx) && col_ok;
                    float v = lg[e] - lse;
                    if (p.prior && ok) v += __logf(p.prior[lane_off + (size_t)iu * p.Ty] + 1e-8f);
                    if (pass == 0) {
                        if (ok) {
                            store_logp(p, lane_off + (size_t)iu * p.Ty, v);
                            if (p.soft && !p.prior) p.soft[lane_off + (size_t)iu * p.Ty] = __expf(v);
                        }
                        if (npass == 2 && ok && v != NEG_INF_F) {
                            const float mn = fmaxf(m2, v);
                            l2s = (m2 == NEG_INF_F ? 0.f : l2s * __expf(m2 - mn)) + __expf(v - mn);
                            m2 = mn;
                        }
                    } else if (ok) {
                        p.soft[lane_off + (size_t)iu * p.Ty] = __expf(v - lse2);
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

// --------------------------------------------------------------------------
// The same convolution on the bf16 matrix cores with split operands: x = hi + lo (two bf16 halves of the
// fp32 value), product = hi*hi + hi*lo + lo*hi in fp32 accumulators -- the similarity kernel's trick,
// ~2^-16 relative per product (far inside the 1e-4 the encoders are held to) at 16/3 times the fp32 MFMA
// rate.  Reduction index k = (tap, in channel): one v_mfma_f32_32x32x16_bf16 k-step is one tap of 16 input
// channels, so a chunk of 16 input channels is K k-steps.  Both operands are split once per workgroup and
// chunk on their way into LDS, in fragment order: a fragment (8 consecutive channels of one frame / one
// output channel) is one 16-byte LDS write and one conflict-free ds_read_b128.  Chunks are double-buffered
// in LDS and the next chunk's global loads are in flight during the MFMAs (one barrier per chunk).
// --------------------------------------------------------------------------
template <int K, int WO, int WT, int AO, int AT>
__global__ __launch_bounds__(WO * WT * 64) void conv1d_bf16x3_kernel(const float *__restrict__ x,
                                                                       const float *__restrict__ w,
                                                                       const float *__restrict__ bias,
                                                                       float *__restrict__ y, int Cin, int Cout, int T,
                                                                       int relu) {
    constexpr int TO = 32 * WO * AO, TT = 32 * WT * AT, NTHR = WO * WT * 64;
    constexpr int HALO = K / 2, XF = TT + 2 * HALO;
    constexpr int XN = XF * 2;                    // x fragments per chunk: [frame][channel half]
    constexpr int WN = K * TO * 2;                // w fragments per chunk: [tap][out channel][channel half]
    constexpr int BUF = 2 * XN + 2 * WN;          // uint4 per buffer: Xhi, Xlo, Whi, Wlo
    extern __shared__ __attribute__((aligned(16))) unsigned char cv_smem[];
    uint4 *lds = reinterpret_cast<uint4 *>(cv_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * TO, t0 = blockIdx.x * TT;
    const int wo = (wave / WT) * (32 * AO), wt = (wave % WT) * (32 * AT);
    const float *xb = x + (size_t)b * Cin * T;
    f32x16 acc[AO][AT];
#pragma unroll
    for (int a = 0; a < AO; ++a)
#pragma unroll
        for (int c = 0; c < AT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

    constexpr int XTASK = (XN + NTHR - 1) / NTHR, WTASK = (WN + NTHR - 1) / NTHR;
    float xr[XTASK][8], wr[WTASK][8];
    // every load is unconditional (indices clamped, value masked afterwards): see and_mask()
    auto fetch = [&](int i0) {
#pragma unroll
        for (int j = 0; j < XTASK; ++j) {
            int task = tid + NTHR * j;
            task = task < XN ? task : XN - 1;
            const int f = task >> 1, h = task & 1;
            const int t = t0 + f - HALO;
            const int tc = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int i = i0 + 8 * h + jj;
                const float v = xb[(size_t)(i < Cin ? i : Cin - 1) * T + tc];
                xr[j][jj] = and_mask(v, (i < Cin && t >= 0 && t < T) ? ~0u : 0u);
            }
        }
#pragma unroll
        for (int j = 0; j < WTASK; ++j) {
            int task = tid + NTHR * j;            // tap fastest: the lanes of one output channel share two cache lines
            task = task < WN ? task : WN - 1;
            const int tap = task % K, rest = task / K;
            const int h = rest & 1, o = o0 + (rest >> 1);
            const size_t orow = (size_t)(o < Cout ? o : Cout - 1) * Cin;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int i = i0 + 8 * h + jj;
                const float v = w[(orow + (i < Cin ? i : Cin - 1)) * K + tap];
                wr[j][jj] = and_mask(v, (o < Cout && i < Cin) ? ~0u : 0u);
            }
        }
    };
    auto pack_split = [&](const float (&r)[8], uint4 &hi, uint4 &lo) {
        bf16x8 h, l;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            __bf16 hh, ll;
            split_bf16(r[jj], hh, ll);
            h[jj] = hh;
            l[jj] = ll;
        }
        hi = __builtin_bit_cast(uint4, h);
        lo = __builtin_bit_cast(uint4, l);
    };
    auto stash = [&](uint4 *bufp) {
#pragma unroll
        for (int j = 0; j < XTASK; ++j) {
            const int task = tid + NTHR * j;
            if (task < XN) pack_split(xr[j], bufp[task], bufp[XN + task]);
        }
#pragma unroll
        for (int j = 0; j < WTASK; ++j) {
            const int task = tid + NTHR * j;
            if (task < WN) {
                const int tap = task % K, rest = task / K;
                const int idx = (tap * TO + (rest >> 1)) * 2 + (rest & 1);
                pack_split(wr[j], bufp[2 * XN + idx], bufp[2 * XN + WN + idx]);
            }
        }
    };
    fetch(0);
    int it = 0;
    for (int i0 = 0; i0 < Cin; i0 += 16, ++it) {
        uint4 *bufp = lds + (it & 1) * BUF;
        stash(bufp);                              // (the other buffer may still be read by slower waves)
        __syncthreads();
        if (i0 + 16 < Cin) fetch(i0 + 16);
        const uint4 *Xhi = bufp, *Xlo = bufp + XN, *Whi = bufp + 2 * XN, *Wlo = bufp + 2 * XN + WN;
#pragma unroll
        for (int tap = 0; tap < K; ++tap) {
            bf16x8 ah[AO], al[AO], bh[AT], bl[AT];
#pragma unroll
            for (int a = 0; a < AO; ++a) {
                const int idx = (tap * TO + wo + 32 * a + l31) * 2 + half;
                ah[a] = __builtin_bit_cast(bf16x8, Whi[idx]);
                al[a] = __builtin_bit_cast(bf16x8, Wlo[idx]);
            }
#pragma unroll
            for (int c = 0; c < AT; ++c) {
                const int idx = (wt + 32 * c + l31 + tap) * 2 + half;
                bh[c] = __builtin_bit_cast(bf16x8, Xhi[idx]);
                bl[c] = __builtin_bit_cast(bf16x8, Xlo[idx]);
            }
#pragma unroll
            for (int a = 0; a < AO; ++a)
#pragma unroll
                for (int c = 0; c < AT; ++c) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[c], acc[a][c], 0, 0, 0);
                }
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

// ---- prepared weights: the split, fragment-ordered form of w, built once per weight tensor ----
// Layout (both halves): [chunk = ceil(Cin/16)][tap][Cout padded to 128][channel half] of uint4 (8 bf16):
// for one (chunk, tap) the fragments of consecutive output channels are contiguous, so a workgroup's
// weight tile is K contiguous runs that it copies with coalesced 16-byte loads -- no gather, no VALU.
// (Gathering w[o][i][tap] per workgroup cost ~300 scattered 4-byte wave-loads per chunk and CU: the
// address coalescer, not the matrix pipe, set the pace: 466 us for the 512->1024 k=3 layer.)
struct ConvPrep { size_t lo_off, total; int nch, cpad; };
static ConvPrep conv_prep_layout(int Cout, int Cin, int K) {
    ConvPrep L;
    L.nch = (Cin + 15) / 16;
    L.cpad = (Cout + 127) / 128 * 128;
    const size_t half = (size_t)L.nch * K * L.cpad * 2 * sizeof(uint4);
    L.lo_off = half;
    L.total = 2 * half;
    return L;
}

__global__ __launch_bounds__(256) void conv_prep_kernel(const float *__restrict__ w, uint4 *__restrict__ phi,
                                                        uint4 *__restrict__ plo, int Cout, int Cin, int K, int cpad,
                                                        int nfrag) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // fragment index: ((chunk*K + tap)*cpad + o)*2 + h
    if (idx >= nfrag) return;
    const int h = idx & 1, o = (idx >> 1) % cpad, ct = (idx >> 1) / cpad;
    const int tap = ct % K, ch = ct / K;
    bf16x8 hv, lv;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int i = 16 * ch + 8 * h + jj;
        const float v = (o < Cout && i < Cin) ? w[((size_t)o * Cin + i) * K + tap] : 0.f;
        __bf16 hh, ll;
        split_bf16(v, hh, ll);
        hv[jj] = hh;
        lv[jj] = ll;
    }
    phi[idx] = __builtin_bit_cast(uint4, hv);
    plo[idx] = __builtin_bit_cast(uint4, lv);
}

// The convolution proper on prepared weights.  XV: T % 4 == 0, so the input rows can be read as aligned
// 16-byte quads (a task = 8 channels x 4 frames -> four fragments); otherwise one frame per task.
template <int K, int WO, int WT, int AO, int AT, bool XV, int SUB>
__global__ __launch_bounds__(WO * WT * 64) void conv1d_prepared_kernel(const float *__restrict__ x,
                                                                         const uint4 *__restrict__ phi,
                                                                         const uint4 *__restrict__ plo,
                                                                         const float *__restrict__ bias,
                                                                         float *__restrict__ y, int Cin, int Cout, int T,
                                                                         int cpad, int relu) {
    constexpr int TO = 32 * WO * AO, TT = 32 * WT * AT, NTHR = WO * WT * 64;
    constexpr int HALO = K / 2;
    constexpr int F0 = XV ? 4 : HALO;             // LDS frame 0 <-> input frame t0 - F0
    constexpr int XF = XV ? TT + 8 : TT + 2 * HALO;
    // a chunk = SUB sub-chunks of 16 input channels (SUB = 4 for k = 1: one tap is too little work per barrier)
    constexpr int XN1 = XF * 2, WN1 = K * TO * 2;
    constexpr int XN = XN1 * SUB;                 // x fragments per chunk: [sub][frame][channel half]
    constexpr int WN = WN1 * SUB;                 // w fragments per chunk: [sub][tap][out channel][channel half]
    constexpr int BUF = 2 * XN + 2 * WN;          // uint4 per buffer: Xhi, Xlo, Whi, Wlo
    extern __shared__ __attribute__((aligned(16))) unsigned char cv_smem[];
    uint4 *lds = reinterpret_cast<uint4 *>(cv_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // (an XCD-aware tile map that keeps one output tile's weights in one XCD's L2 changed nothing on the
    // 512->1024 layer and cost 50 % on the narrow ones: not kept)
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * TO, t0 = blockIdx.x * TT;
    const int wo = (wave / WT) * (32 * AO), wt = (wave % WT) * (32 * AT);
    const float *xb = x + (size_t)b * Cin * T;
    f32x16 acc[AO][AT];
#pragma unroll
    for (int a = 0; a < AO; ++a)
#pragma unroll
        for (int c = 0; c < AT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][c][e] = 0.f;

    constexpr int WTASK = (WN + NTHR - 1) / NTHR;
    constexpr int XQ1 = XV ? (XF / 4) * 2 : XN1;  // x tasks per sub-chunk
    constexpr int XQ = XQ1 * SUB;
    constexpr int XTASK = (XQ + NTHR - 1) / NTHR;
    // one chunk's operands in registers, on their way to LDS: chunk c+1 is fetched while chunk c is multiplied
    // (fetching two chunks ahead with two register sets was slower: 376 vs 325 us on the 512->1024 layer)
    // One chunk's operands in registers on their way to LDS.  Native vector types: register arrays of HIP's
    // uint4/float4 CLASSES stayed in scratch memory (320 instead of 201 us on the 512->1024 layer).
    typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
    typedef float __attribute__((ext_vector_type(4))) f32x4v;
    struct Stage {
        u32x4 wh[WTASK], wl[WTASK];
        f32x4v xq[XV ? XTASK : 1][8];
        float xr[XV ? 1 : XTASK][8];
    };
    const int nch16 = (Cin + 15) / 16;
    auto fetch = [&](Stage &R, int ch) {
#pragma unroll
        for (int j = 0; j < WTASK; ++j) {
            int task = tid + NTHR * j;
            task = task < WN ? task : WN - 1;
            const int sub = task / WN1, t1 = task - sub * WN1;
            const int tap = t1 / (TO * 2), r = t1 - tap * (TO * 2);
            int c16 = ch * SUB + sub;                                     // past the last sub-chunk: any finite data
            c16 = c16 < nch16 ? c16 : nch16 - 1;                          // (the x operand is zero there)
            const size_t src = ((size_t)(c16 * K + tap) * cpad + o0) * 2 + r;
            R.wh[j] = *reinterpret_cast<const u32x4 *>(phi + src);
            R.wl[j] = *reinterpret_cast<const u32x4 *>(plo + src);
        }
        if (XV) {
#pragma unroll
            for (int j = 0; j < XTASK; ++j) {
                const int task = tid + NTHR * j;
                if (task < XQ) {
                    const int sub = task / XQ1, t1 = task - sub * XQ1;
                    const int i0 = 16 * (ch * SUB + sub);
                    const int q = t1 >> 1, h = t1 & 1;
                    const int t = t0 - 4 + 4 * q;                         // aligned quad: all in or all out
                    const bool in = t >= 0 && t < T;
                    const int tc = t < 0 ? 0 : (t > T - 4 ? T - 4 : t);   // unconditional loads, masked afterwards
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int i = i0 + 8 * h + jj;
                        const f32x4v v = *reinterpret_cast<const f32x4v *>(xb + (size_t)(i < Cin ? i : Cin - 1) * T + tc);
                        const unsigned mk = (in && i < Cin) ? ~0u : 0u;
                        f32x4v mv;
                        mv.x = and_mask(v.x, mk); mv.y = and_mask(v.y, mk); mv.z = and_mask(v.z, mk); mv.w = and_mask(v.w, mk);
                        R.xq[j][jj] = mv;
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < XTASK; ++j) {
                int task = tid + NTHR * j;
                task = task < XN ? task : XN - 1;
                const int sub = task / XN1, t1 = task - sub * XN1;
                const int i0 = 16 * (ch * SUB + sub);
                const int f = t1 >> 1, h = t1 & 1;
                const int t = t0 + f - F0;
                const int tc = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int i = i0 + 8 * h + jj;
                    const float v = xb[(size_t)(i < Cin ? i : Cin - 1) * T + tc];
                    R.xr[j][jj] = and_mask(v, (i < Cin && t >= 0 && t < T) ? ~0u : 0u);
                }
            }
        }
    };
    auto pack_split = [&](const float (&r)[8], uint4 &hi, uint4 &lo) {
        bf16x8 h, l;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            __bf16 hh, ll;
            split_bf16(r[jj], hh, ll);
            h[jj] = hh;
            l[jj] = ll;
        }
        hi = __builtin_bit_cast(uint4, h);
        lo = __builtin_bit_cast(uint4, l);
    };
    auto multiply = [&](const uint4 *bufp) {
        const uint4 *Xhi = bufp, *Xlo = bufp + XN, *Whi = bufp + 2 * XN, *Wlo = bufp + 2 * XN + WN;
#pragma unroll
        for (int st = 0; st < SUB * K; ++st) {
            const int sub = st / K, tap = st - sub * K;
            bf16x8 ah[AO], al[AO], bh[AT], bl[AT];
#pragma unroll
            for (int a = 0; a < AO; ++a) {
                const int idx = sub * WN1 + half * (K * TO) + tap * TO + wo + 32 * a + l31;       // [sub][channel half][tap][out channel]
                ah[a] = __builtin_bit_cast(bf16x8, Whi[idx]);
                al[a] = __builtin_bit_cast(bf16x8, Wlo[idx]);
            }
#pragma unroll
            for (int c = 0; c < AT; ++c) {
                const int idx = sub * XN1 + half * XF + (wt + 32 * c + l31 + tap - HALO + F0);     // [sub][channel half][frame]
                bh[c] = __builtin_bit_cast(bf16x8, Xhi[idx]);
                bl[c] = __builtin_bit_cast(bf16x8, Xlo[idx]);
            }
#pragma unroll
            for (int a = 0; a < AO; ++a)
#pragma unroll
                for (int c = 0; c < AT; ++c) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[c], acc[a][c], 0, 0, 0);
                }
        }
    };
    auto stash = [&](const Stage &R, uint4 *bufp) {
#pragma unroll
        for (int j = 0; j < WTASK; ++j) {
            const int task = tid + NTHR * j;
            if (task < WN) {
                // LDS keeps the two channel halves of a fragment row apart ([half][tap][out channel]): the 32 lanes of a
                // half then read CONSECUTIVE 16-byte slots (interleaved, every ds_read_b128 was a 2-4-way bank conflict:
                // 45 % of the LDS's busy cycles)
                const int sub = task / WN1, t1 = task - sub * WN1;
                const int dsti = sub * WN1 + (t1 & 1) * (K * TO) + (t1 >> 1);
                *reinterpret_cast<u32x4 *>(bufp + 2 * XN + dsti) = R.wh[j];
                *reinterpret_cast<u32x4 *>(bufp + 2 * XN + WN + dsti) = R.wl[j];
            }
        }
        if (XV) {
#pragma unroll
            for (int j = 0; j < XTASK; ++j) {
                const int task = tid + NTHR * j;
                if (task < XQ) {
                    const int sub = task / XQ1, t1 = task - sub * XQ1;
                    const int q = t1 >> 1, h = t1 & 1;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float r[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj)
                            r[jj] = u == 0 ? R.xq[j][jj].x : u == 1 ? R.xq[j][jj].y : u == 2 ? R.xq[j][jj].z : R.xq[j][jj].w;
                        const int fi = sub * XN1 + h * XF + (4 * q + u);
                        pack_split(r, bufp[fi], bufp[XN + fi]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < XTASK; ++j) {
                const int task = tid + NTHR * j;
                if (task < XN) {
                    const int sub = task / XN1, t1 = task - sub * XN1;
                    const int fi = sub * XN1 + (t1 & 1) * XF + (t1 >> 1);
                    pack_split(R.xr[j], bufp[fi], bufp[XN + fi]);
                }
            }
        }
    };
    // software pipeline: chunk c+1 is fetched into registers while chunk c is multiplied out of LDS
    // (fetching two chunks ahead with a second register set was slower: 304 vs 201 us on the 512->1024 layer)
    const int nch = (Cin + 16 * SUB - 1) / (16 * SUB);
    Stage R;
    fetch(R, 0);
    for (int ch = 0; ch < nch; ++ch) {
        uint4 *bufp = lds + (ch & 1) * BUF;
        stash(R, bufp);                           // (the other buffer may still be read by slower waves)
        __syncthreads();
        if (ch + 1 < nch) fetch(R, ch + 1);
        multiply(bufp);
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

template <int K, int WO, int WT, int AO, int AT, int SUB>
static int launch_conv_prepared_sub(dim3 grid, hipStream_t s, const float *x, const uint4 *phi, const uint4 *plo,
                                    const float *bias, float *y, int Cin, int Cout, int T, int cpad, int relu) {
    constexpr int TO = 32 * WO * AO, TT = 32 * WT * AT;
    const bool xv = (T % 4) == 0;
    const size_t xf = xv ? TT + 8 : TT + 2 * (K / 2);
    const size_t lds = (size_t)2 * SUB * (2 * xf * 2 + 2 * K * TO * 2) * sizeof(uint4);
    if (xv) {
        auto kern = conv1d_prepared_kernel<K, WO, WT, AO, AT, true, SUB>;
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
        hipLaunchKernelGGL(kern, grid, dim3(WO * WT * 64), lds, s, x, phi, plo, bias, y, Cin, Cout, T, cpad, relu);
    } else {
        auto kern = conv1d_prepared_kernel<K, WO, WT, AO, AT, false, SUB>;
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
        hipLaunchKernelGGL(kern, grid, dim3(WO * WT * 64), lds, s, x, phi, plo, bias, y, Cin, Cout, T, cpad, relu);
    }
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

template <int K, int WO, int WT, int AO, int AT>
static int launch_conv_prepared(dim3 grid, hipStream_t s, const float *x, const uint4 *phi, const uint4 *plo,
                                const float *bias, float *y, int Cin, int Cout, int T, int cpad, int relu) {
    // k = 1 over many input channels: 64-channel chunks (one tap of 16 channels is too little work per barrier;
    // 1024->80 on [64,.,200]: 67 -> 36 us).  Narrow inputs keep 16-channel chunks (padding to 64 would waste them).
    if (K == 1 && Cin >= 256)
        return launch_conv_prepared_sub<K, WO, WT, AO, AT, (K == 1 ? 4 : 1)>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, cpad, relu);
    return launch_conv_prepared_sub<K, WO, WT, AO, AT, 1>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, cpad, relu);
}

template <int K, int WO, int WT, int AO, int AT>
static int launch_conv_bf16x3(dim3 grid, hipStream_t s, const float *x, const float *w, const float *bias, float *y,
                              int Cin, int Cout, int T, int relu) {
    constexpr int TO = 32 * WO * AO, TT = 32 * WT * AT;
    const size_t lds = (size_t)2 * (2 * (TT + 2 * (K / 2)) * 2 + 2 * K * TO * 2) * sizeof(uint4);
    auto kern = conv1d_bf16x3_kernel<K, WO, WT, AO, AT>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    hipLaunchKernelGGL(kern, grid, dim3(WO * WT * 64), lds, s, x, w, bias, y, Cin, Cout, T, relu);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
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
    const size_t lds = (size_t)2 * G * KS * 64 * sizeof(uint4) + (size_t)G * 32 * sizeof(float);
    auto kern = softattn_kernel<KS, G, MULTI>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    SoftAttnParams q = p;
    q.pair = 0;
    if (MULTI && !g_opt_softattn_no_pair) {
        const long long cus = device_cu_count();
        auto wgs = [&](int nsp) { return (long long)((p.Ty + 32 * (SA_WAVES / nsp) - 1) / (32 * (SA_WAVES / nsp))) * p.B; };
        if (g_opt_softattn_split > 1) q.pair = g_opt_softattn_split;
        // as many waves per strip as still leave every workgroup a CU of its own (measured, one / two / four waves per
        // strip: [8,500,4000] 81 / 55 / 69 us, [4,500,4000] 78 / 52 / 39, [16,400,2000] 61 / 42 / 54, [2,300,1000] 51 / 35 / 26)
        else if (wgs(4) <= cus && p.Ty > 64 && g_opt_softattn_split != 1) q.pair = 4;
        else if (wgs(2) <= cus && p.Ty > 128) q.pair = 2;
    }
    const int swv = q.pair > 1 ? SA_WAVES / q.pair : SA_WAVES;
    dim3 grid((unsigned)((p.Ty + 32 * swv - 1) / (32 * swv)) * (unsigned)p.B), block(SA_THREADS);
    hipLaunchKernelGGL(kern, grid, block, lds, s, q);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

// row-tile form: one compute wave per 32-row tile + the loader wave; 256 frames per workgroup
template <int KS, int NT, bool OUT16>
static int launch_softattn_rt(const SoftAttnParams &p, hipStream_t s) {
    const size_t lds = (size_t)(rt_raw_slots(KS) + RT_RING) * KS * 2048 + (size_t)2 * NT * 32 * sizeof(float2) + (size_t)(NT + 2) * 32 * sizeof(float);
    static_assert(((rt_raw_slots(KS) + RT_RING) * KS * 2048 + 2 * NT * 32 * 8 + (NT + 2) * 32 * 4) <= 160 * 1024, "LDS");
    auto kern = softattn_rt_kernel<KS, NT, OUT16>;
    if (p.stamps && KS == 5 && !OUT16) kern = softattn_rt_kernel<(KS == 5 ? KS : 5), NT, false, true>;     // (development: tools/sa_rt_stamps.py)
    SoftAttnParams q = p;
    q.pair = g_opt_softattn_rt_drop_merge ? -1 : 0;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    dim3 grid((unsigned)((p.Ty + 32 * RT_STRIPS - 1) / (32 * RT_STRIPS)) * (unsigned)p.B), block((NT + 1) * 64);
    hipLaunchKernelGGL(kern, grid, block, lds, s, q);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

template <int KS>
static int launch_softattn_exact(const SoftAttnParams &p, hipStream_t s) {
    const size_t per_tile = (size_t)8 * KS * 64 * sizeof(float);
    const int RT = (p.Tx + 31) / 32;
    int GE = (int)((60 * 1024) / per_tile);
    GE = GE < 1 ? 1 : (GE > RT ? RT : GE);
    const size_t lds = GE * per_tile + (size_t)GE * 32 * sizeof(float);
    auto kern = softattn_exact_kernel<KS>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    dim3 grid((unsigned)((p.Ty + 127) / 128) * (unsigned)p.B), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, s, p, GE);
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
    return aligner_softattn(keys, queries, t_xs, prior, logp_out, ALIGNER_DT_F32, soft_out, workspace, workspace_bytes, B,
                            C, Tx, Ty, temperature, sim, stream);
}

int aligner_softattn(const float *keys, const float *queries, const int32_t *t_xs, const float *prior,
                     void *logp_out, int logp_dtype, float *soft_out, void *workspace, size_t workspace_bytes, int B,
                     int C, int Tx, int Ty, float temperature, int sim, void *stream) {
    return aligner_softattn_ld(keys, queries, t_xs, prior, logp_out, logp_dtype, Ty, soft_out, workspace, workspace_bytes, B, C,
                               Tx, Ty, temperature, sim, stream);
}

int aligner_softattn_ld(const float *keys, const float *queries, const int32_t *t_xs, const float *prior,
                        void *logp_out, int logp_dtype, int ld_logp, float *soft_out, void *workspace, size_t workspace_bytes,
                        int B, int C, int Tx, int Ty, float temperature, int sim, void *stream) {
    if (!keys || !queries || !logp_out || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (ld_logp < Ty) return fail(ALIGNER_EINVAL, "ld_logp=%d < Ty=%d", ld_logp, Ty);
    if (logp_dtype != ALIGNER_DT_F32 && logp_dtype != ALIGNER_DT_BF16)
        return fail(ALIGNER_EINVAL, "logp dtype %d not supported (F32 or BF16)", logp_dtype);
    if (B < 0 || C < 1 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d C=%d Tx=%d Ty=%d", B, C, Tx, Ty);
    if (sim != ALIGNER_SIM_L2 && sim != ALIGNER_SIM_DOT) return fail(ALIGNER_EINVAL, "bad sim %d", sim);
    if (C > 256) return fail(ALIGNER_EDOM, "C=%d exceeds 256 attention channels", C);
    if (B > 65535) return fail(ALIGNER_EDOM, "B=%d too large", B);
    if ((size_t)Tx * (size_t)ld_logp >= (1ull << 29))     // an utterance's block is one buffer resource (32-bit offsets)
        return fail(ALIGNER_EDOM, "Tx*ld=%zu exceeds 2^29", (size_t)Tx * ld_logp);
    if (B == 0) return ALIGNER_OK;
    const SaLayout L = sa_layout(B, C, Tx);
    if (workspace_bytes < L.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, L.total);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    SoftAttnParams p{keys, queries, t_xs, prior, static_cast<float *>(logp_out), soft_out,
                     reinterpret_cast<const uint4 *>(ws + L.hi_off), reinterpret_cast<const uint4 *>(ws + L.lo_off),
                     reinterpret_cast<const float *>(ws + L.kn_off), L.RT, g_debug_stamps, B, C, Tx, Ty, temperature,
                     sim, logp_dtype == ALIGNER_DT_BF16 ? 1 : 0, 0, ld_logp};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a row pitch of its own (rows that start on a 128-byte line: see aligner_amd.h) is the row-tile form's
    const bool rt_form = L.KS <= 8 && L.RT <= 7 && C == 16 * L.KS && (Ty & 3) == 0 && !prior && !soft_out && !g_opt_softattn_strips &&
                         (reinterpret_cast<uintptr_t>(queries) & 15) == 0;
    if (ld_logp != Ty) {
        const bool sharp0 = (sim == ALIGNER_SIM_L2) ? temperature > 0.002f : temperature > 0.2f;
        if (!rt_form || sharp0 || g_opt_softattn_exact)
            return fail(ALIGNER_EDOM, "ld_logp != Ty needs the row-tile form (Tx <= 224, C = 80 or 128, Ty %% 4 == 0, no prior / soft "
                                      "output, a temperature the bf16x3 products hold)");
        if ((ld_logp * (logp_dtype == ALIGNER_DT_BF16 ? 2 : 4)) % 16 != 0)
            return fail(ALIGNER_EINVAL, "ld_logp=%d: rows must start on 16-byte boundaries", ld_logp);
    }
    // sharp temperatures multiply the bf16x3 product error past the 1e-4 bound: exact fp32 products instead
    // (rule and reasoning: softattn_exact_kernel)
    const bool sharp = (sim == ALIGNER_SIM_L2) ? temperature > 0.002f : temperature > 0.2f;
    if (sharp || g_opt_softattn_exact) {
        if (L.KS == 5) return launch_softattn_exact<5>(p, s);
        if (L.KS == 8) return launch_softattn_exact<8>(p, s);
        return launch_softattn_exact<16>(p, s);
    }
    const int G = L.KS <= 8 ? 7 : 4;
    const bool multi = L.RT > G;
    if (soft_out && prior && multi)
        return fail(ALIGNER_EDOM, "soft output with a prior needs Tx <= %d", 32 * G);
    // the row-tile form (softattn_rt_kernel): one row group, no channel padding (the loader's LDS-DMA reads whole
    // 8-channel pieces), frames in aligned quads, log-probs only
    if (!multi && rt_form) {
        if (L.KS == 5) return p.out16 ? launch_softattn_rt<5, 7, true>(p, s) : launch_softattn_rt<5, 7, false>(p, s);
        return p.out16 ? launch_softattn_rt<8, 7, true>(p, s) : launch_softattn_rt<8, 7, false>(p, s);
    }
    if (L.KS == 5) return multi ? launch_softattn<5, 7, true>(p, ws, L, s) : launch_softattn<5, 7, false>(p, ws, L, s);
    if (L.KS == 8) return multi ? launch_softattn<8, 7, true>(p, ws, L, s) : launch_softattn<8, 7, false>(p, ws, L, s);
    return multi ? launch_softattn<16, 4, true>(p, ws, L, s) : launch_softattn<16, 4, false>(p, ws, L, s);
}

// prepared weights = [conv1d_prepared_kernel's image][for wide layers (conv_gemm_applies): conv_gemm_kernel's image]
static size_t conv_prep_first_bytes(int Cout, int Cin, int K) { return align_up(conv_prep_layout(Cout, Cin, K).total, 256); }

size_t aligner_conv1d_prepared_bytes(int Cout, int Cin, int K) {
    if (Cout < 1 || Cin < 1 || (K != 1 && K != 3 && K != 5)) return 0;
    return conv_prep_first_bytes(Cout, Cin, K) + (conv_gemm_applies(Cin, Cout, K) ? conv_gemm_prepared_bytes(Cout, Cin, K) : 0);
}

size_t aligner_conv1d_workspace_bytes(int B, int Cin, int Cout, int T, int K) {
    if (B < 1 || Cout < 1 || Cin < 1 || T < 1 || !conv_gemm_applies(Cin, Cout, K)) return 0;
    return conv_gemm_workspace_bytes(B, Cin, Cout, T, K);
}

int aligner_conv1d_prepare_f32(const float *w, void *prepared, size_t prepared_bytes, int Cout, int Cin, int K,
                               void *stream) {
    if (!w || !prepared) return fail(ALIGNER_EINVAL, "null pointer");
    if (Cout < 1 || Cin < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (K != 1 && K != 3 && K != 5) return fail(ALIGNER_EDOM, "kernel size %d not supported (1, 3, 5)", K);
    const ConvPrep L = conv_prep_layout(Cout, Cin, K);
    const size_t need = aligner_conv1d_prepared_bytes(Cout, Cin, K);
    if (prepared_bytes < need) return fail(ALIGNER_ENOSPC, "prepared buffer %zu < %zu bytes", prepared_bytes, need);
    unsigned char *pp = static_cast<unsigned char *>(prepared);
    const int nfrag = L.nch * K * L.cpad * 2;
    hipLaunchKernelGGL(conv_prep_kernel, dim3((nfrag + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       reinterpret_cast<uint4 *>(pp), reinterpret_cast<uint4 *>(pp + L.lo_off), Cout, Cin, K, L.cpad, nfrag);
    ALIGNER_HIP_CHECK(hipGetLastError());
    if (conv_gemm_applies(Cin, Cout, K))
        return conv_gemm_prepare(w, pp + conv_prep_first_bytes(Cout, Cin, K), Cout, Cin, K, static_cast<hipStream_t>(stream));
    return ALIGNER_OK;
}

int aligner_conv1d_prepared_ws_f32(const float *x, const void *prepared, const float *bias, float *y, void *workspace,
                                   size_t workspace_bytes, int B, int Cin, int Cout, int T, int K, int relu, void *stream) {
    if (!x || !prepared || !y) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Cin < 1 || Cout < 1 || T < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (K != 1 && K != 3 && K != 5) return fail(ALIGNER_EDOM, "kernel size %d not supported (1, 3, 5)", K);
    if (B == 0) return ALIGNER_OK;
    static const bool no_gemm = [] { const char *e = getenv("ALIGNER_CONV_NO_GEMM"); return e && e[0] == '1'; }();
    const size_t nws = (conv_gemm_applies(Cin, Cout, K) && !no_gemm) ? conv_gemm_workspace_bytes(B, Cin, Cout, T, K) : 0;
    if (nws == 0)                                          // no GEMM form for this layer: conv1d_prepared_kernel, no workspace
        return aligner_conv1d_prepared_f32(x, prepared, bias, y, B, Cin, Cout, T, K, relu, stream);
    if (!workspace) return fail(ALIGNER_EINVAL, "this layer needs aligner_conv1d_workspace_bytes() of workspace");
    const unsigned char *pp = static_cast<const unsigned char *>(prepared);
    return conv_gemm_run(x, pp + conv_prep_first_bytes(Cout, Cin, K), bias, y, workspace, workspace_bytes, B, Cin, Cout, T, K,
                         relu, static_cast<hipStream_t>(stream));
}

// A whole encoder stack in one call: the first layer's input is split once, every k = 1 layer reads the image its
// producer's epilogue wrote (no fp32 round trip between layers), the last layer writes fp32 [B, Cout, T].
static int conv_stack_convert(const aligner_conv_layer *layers, int n, ConvStackLayer *L) {
    for (int i = 0; i < n; ++i) {
        const aligner_conv_layer &a = layers[i];
        if (a.Cin < 1 || a.Cout < 1 || (a.K != 1 && a.K != 3 && a.K != 5)) return fail(ALIGNER_EINVAL, "layer %d: bad shape", i);
        if (i > 0 && a.Cin != layers[i - 1].Cout) return fail(ALIGNER_EINVAL, "layer %d: %d input channels after %d outputs", i, a.Cin, layers[i - 1].Cout);
        const unsigned char *pp = static_cast<const unsigned char *>(a.prepared);
        L[i] = ConvStackLayer{pp ? pp + conv_prep_first_bytes(a.Cout, a.Cin, a.K) : nullptr, a.bias, a.Cin, a.Cout, a.K, a.relu};
    }
    return ALIGNER_OK;
}

size_t aligner_conv_stack_workspace_bytes(const aligner_conv_layer *layers, int n_layers, int B, int T) {
    if (!layers || n_layers < 1 || n_layers > 16 || B < 1 || T < 1) return 0;
    ConvStackLayer L[16];
    if (conv_stack_convert(layers, n_layers, L) != ALIGNER_OK) return 0;
    for (int i = 0; i < n_layers; ++i)
        if (!conv_gemm_applies(L[i].Cin, L[i].Cout, L[i].K)) return 0;
    return conv_stack_workspace_bytes(L, n_layers, B, T);
}

int aligner_conv_stack_f32(const float *x, const aligner_conv_layer *layers, int n_layers, float *y, void *workspace,
                           size_t workspace_bytes, int B, int T, void *stream) {
    if (!x || !layers || !y || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (n_layers < 1 || n_layers > 16) return fail(ALIGNER_EINVAL, "1..16 layers");
    if (B < 0 || T < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (B == 0) return ALIGNER_OK;
    ConvStackLayer L[16];
    const int rc = conv_stack_convert(layers, n_layers, L);
    if (rc != ALIGNER_OK) return rc;
    for (int i = 0; i < n_layers; ++i)
        if (!layers[i].prepared) return fail(ALIGNER_EINVAL, "layer %d: null prepared weights", i);
    return conv_stack_run(x, L, n_layers, y, workspace, workspace_bytes, B, T, static_cast<hipStream_t>(stream));
}

int aligner_conv1d_prepared_f32(const float *x, const void *prepared, const float *bias, float *y, int B, int Cin,
                                int Cout, int T, int K, int relu, void *stream) {
    if (!x || !prepared || !y) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Cin < 1 || Cout < 1 || T < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (K != 1 && K != 3 && K != 5) return fail(ALIGNER_EDOM, "kernel size %d not supported (1, 3, 5)", K);
    if (B == 0) return ALIGNER_OK;
    if (B > 65535) return fail(ALIGNER_EDOM, "grid too large");
    const ConvPrep L = conv_prep_layout(Cout, Cin, K);
    const unsigned char *pp = static_cast<const unsigned char *>(prepared);
    const uint4 *phi = reinterpret_cast<const uint4 *>(pp), *plo = reinterpret_cast<const uint4 *>(pp + L.lo_off);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (Cout > 96) {
        dim3 grid((T + 127) / 128, (Cout + 127) / 128, B);
        if (grid.y > 65535) return fail(ALIGNER_EDOM, "grid too large");
        if (K == 1) return launch_conv_prepared<1, 2, 2, 2, 2>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, L.cpad, relu);
        if (K == 3) return launch_conv_prepared<3, 2, 2, 2, 2>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, L.cpad, relu);
        return launch_conv_prepared<5, 2, 2, 2, 2>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, L.cpad, relu);
    }
    dim3 grid((T + 63) / 64, 1, B);
    if (K == 1) return launch_conv_prepared<1, 3, 2, 1, 1>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, L.cpad, relu);
    if (K == 3) return launch_conv_prepared<3, 3, 2, 1, 1>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, L.cpad, relu);
    return launch_conv_prepared<5, 3, 2, 1, 1>(grid, s, x, phi, plo, bias, y, Cin, Cout, T, L.cpad, relu);
}

int aligner_conv1d_f32(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int Cout,
                       int T, int K, int relu, void *stream) {
    if (!x || !w || !y) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Cin < 1 || Cout < 1 || T < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (B == 0) return ALIGNER_OK;
    if (B > 65535) return fail(ALIGNER_EDOM, "grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (K != 1 && K != 3 && K != 5) return fail(ALIGNER_EDOM, "kernel size %d not supported (1, 3, 5)", K);
    // ALIGNER_CONV_FP32=1 selects the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32); the default splits the
    // operands into bf16 halves (three bf16 MFMAs per product, ~2^-16 relative)
    static const bool exact_fp32 = [] { const char *e = getenv("ALIGNER_CONV_FP32"); return e && e[0] == '1'; }();
    if (!exact_fp32) {
        if (Cout > 96) {
            dim3 grid((T + 127) / 128, (Cout + 127) / 128, B);
            if (grid.y > 65535) return fail(ALIGNER_EDOM, "grid too large");
            if (K == 1) return launch_conv_bf16x3<1, 2, 2, 2, 2>(grid, s, x, w, bias, y, Cin, Cout, T, relu);
            if (K == 3) return launch_conv_bf16x3<3, 2, 2, 2, 2>(grid, s, x, w, bias, y, Cin, Cout, T, relu);
            return launch_conv_bf16x3<5, 2, 2, 2, 2>(grid, s, x, w, bias, y, Cin, Cout, T, relu);
        }
        dim3 grid((T + 63) / 64, 1, B);
        if (K == 1) return launch_conv_bf16x3<1, 3, 2, 1, 1>(grid, s, x, w, bias, y, Cin, Cout, T, relu);
        if (K == 3) return launch_conv_bf16x3<3, 3, 2, 1, 1>(grid, s, x, w, bias, y, Cin, Cout, T, relu);
        return launch_conv_bf16x3<5, 3, 2, 1, 1>(grid, s, x, w, bias, y, Cin, Cout, T, relu);
    }
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
