// Wide 1-D convolutions of the alignment encoders as a matrix-core GEMM that never waits for its operands
// (gfx950).  y[b,o,t] = act(bias[o] + sum_{i,k} w[o,i,k] * x[b,i,t+k-K/2]) -- the build-defined encoder stack of
// SURVEY.md 7.4 (no reference source: /root/reference/README.md:21-25,50 only names the branch).
//
// Round 3's kernel (softattn.hip, conv1d_prepared_kernel) moved every 16-channel chunk global -> VGPR -> split -> LDS
// -> barrier: 36 MFMAs (~0.5 us) per ~1 us load round trip, 42 % of its wave-cycles in s_waitcnt, 0.075 of the
// bf16 peak on the 512 -> 1024 k=3 layer.  This one is built around the load path instead:
//
//  * Both operands are split into bf16 halves BEFORE the GEMM (x = hi + lo; products hi*hi + hi*lo + lo*hi in fp32
//    accumulators, ~2^-16 relative per product, as before): the weights once per weight tensor
//    (conv_gemm_wprep_kernel), the activations by a streaming pass (conv_split_kernel, ~2 x 27 MB at C3) that also
//    transposes them to "channels-last fragments": 16 bytes = 8 consecutive input channels of one frame, zero
//    frames either side of the utterance -- the convolution's padding is data, not a branch.
//  * v_mfma_f32_16x16x32_bf16, a chunk = 32 input channels.  A workgroup = 4 waves = 128 output channels x all
//    frames of one utterance (<= 16*FT; FT = 13: 208 frames for T = 200, 4 % padding where 32-wide tiles had 12 %);
//    wave w owns output channels 32w .. 32w+31 and every frame tile: 2 x FT accumulator tiles (104 VGPRs).
//  * The activations of a chunk are shared by the four waves: they go global -> LDS by LDS-DMA
//    (global_load_lds_dwordx4, 1 KB per wave-instruction, no VGPRs, no ds_write), double-buffered, the next chunk in
//    flight under this chunk's 234 MFMAs per wave.  LDS image [plane][channel quarter][frame slot] with rows a multiple
//    of 256 bytes apart: every ds_read_b128 of a fragment is conflict-free for every tap shift.
//  * A wave's weight fragments are its own (nobody shares them), so they skip LDS: prepared in fragment order,
//    1 KB per (chunk, tap, plane, 16-channel tile), they are loaded straight into VGPRs one chunk ahead (a
//    three-slot ring, one slot per tap).
//  * All loads are issued by hand and retired with counted s_waitcnt vmcnt(N) (vmcnt retires in order: the counts
//    below are exact), one raw s_barrier per chunk: nothing in the loop drains the memory queue.
//  * Two workgroups per CU (2 x 56 KB LDS, <= 256 VGPRs): one wave's barrier / wait is the other's matrix time.
//  * Output orientation: frames are the MFMA's M axis, so a lane holds 4 consecutive frames of one output channel:
//    16-byte stores into [B, Cout, T].
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

typedef __attribute__((ext_vector_type(8))) __bf16 cg_bf16x8;
typedef __attribute__((ext_vector_type(4))) float cg_f32x4;
typedef unsigned __attribute__((ext_vector_type(4))) cg_u32x4;

constexpr int CG_CH = 32;            // input channels per chunk (one k-step of v_mfma_f32_16x16x32_bf16)
constexpr int CG_TO = 128;           // output channels per workgroup (4 waves x 2 tiles of 16)

__host__ __device__ constexpr int cg_lrow(int KT, int FT) { return (16 * FT + 2 * (KT / 2) + 31) / 32 * 32; }

struct ConvGemmLayout {
    int nch, cpad, S;                // chunks of 32 input channels; Cout padded to 128; slots per (b, chunk, quarter) row
    size_t w_bytes;                  // prepared weights: [chunk][tap][16-channel tile][plane][lane] of 16 bytes
    size_t xs_plane;                 // uint4 per plane of the split activations: [b][chunk][quarter][S]
    size_t xs_bytes;                 // both planes + read slack
};

static ConvGemmLayout conv_gemm_layout(int B, int Cin, int Cout, int T, int K) {
    ConvGemmLayout L;
    L.nch = (Cin + CG_CH - 1) / CG_CH;
    L.cpad = (Cout + CG_TO - 1) / CG_TO * CG_TO;
    L.S = (T + 2 * (K / 2) + 15) / 16 * 16;
    L.w_bytes = (size_t)L.nch * K * (L.cpad / 16) * 2 * 64 * sizeof(uint4);
    L.xs_plane = (size_t)B * L.nch * 4 * L.S;
    L.xs_bytes = (2 * L.xs_plane + 512) * sizeof(uint4);     // slack: a staging piece may read past the last row
    return L;
}

__device__ __forceinline__ void cg_split(float v, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

// weights -> fragment order.  B operand of the MFMA (k = input channel, n = output channel): lane l holds
// w[o = 16*tile + (l & 15)][i = 32*chunk + 8*(l >> 4) + j][tap], j = 0..7.
__global__ __launch_bounds__(256) void conv_gemm_wprep_kernel(const float *__restrict__ w, uint4 *__restrict__ wp, int Cout,
                                                              int Cin, int K, int cpad, int nfrag) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // ((chunk*K + tap)*(cpad/16) + tile)*64 + lane
    if (idx >= nfrag) return;
    const int lane = idx & 63, tile = (idx >> 6) % (cpad / 16), ct = (idx >> 6) / (cpad / 16);
    const int tap = ct % K, ch = ct / K;
    const int o = 16 * tile + (lane & 15);
    cg_bf16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = CG_CH * ch + 8 * (lane >> 4) + j;
        const float v = (o < Cout && i < Cin) ? w[((size_t)o * Cin + i) * K + tap] : 0.f;
        __bf16 hh, ll;
        cg_split(v, hh, ll);
        hv[j] = hh;
        lv[j] = ll;
    }
    const size_t dst = ((size_t)(idx >> 6) * 2) * 64 + lane;
    wp[dst] = __builtin_bit_cast(uint4, hv);
    wp[dst + 64] = __builtin_bit_cast(uint4, lv);
}

// activations [B, Cin, T] fp32 -> split channels-last fragments: xs[plane][b][chunk][quarter][slot], slot s = frame
// s - HALO; slots outside the utterance and channels >= Cin are zero.  One thread per slot: its 8 channel loads are
// coalesced across the wave (consecutive frames), its two stores are 16 bytes.
__global__ __launch_bounds__(256) void conv_split_kernel(const float *__restrict__ x, uint4 *__restrict__ xs, size_t xs_plane,
                                                         int Cin, int T, int S, int nch, int halo, size_t total) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;       // ((b*nch + chunk)*4 + quarter)*S + slot
    if (idx >= total) return;
    const int s = (int)(idx % S);
    const size_t r = idx / S;
    const int qq = (int)(r & 3), c = (int)((r >> 2) % nch);
    const size_t b = (r >> 2) / nch;
    const int t = s - halo;
    const bool in = t >= 0 && t < T;
    const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = CG_CH * c + 8 * qq + j;
        const float ld = x[((size_t)b * Cin + (i < Cin ? i : Cin - 1)) * T + tc];   // unconditional, masked afterwards
        v[j] = (in && i < Cin) ? ld : 0.f;
    }
    cg_bf16x8 hv, lv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 hh, ll;
        cg_split(v[j], hh, ll);
        hv[j] = hh;
        lv[j] = ll;
    }
    xs[idx] = __builtin_bit_cast(uint4, hv);
    xs[xs_plane + idx] = __builtin_bit_cast(uint4, lv);
}

struct ConvGemmParams {
    const uint4 *xs;        // split activations, plane 0 (plane 1 at + xs_plane)
    const uint4 *wp;        // prepared weights
    const float *bias;      // nullable
    float *y;               // [B, Cout, T]
    unsigned long long xs_plane;
    int B, Cout, T, S, nch, cpad, relu, nx;
};

// ---- hand-issued memory operations (the compiler neither counts nor waits for them: every wait below is ours) ----
// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses (sbase + voff) to LDS [lds_dst + 16*lane]
__device__ __forceinline__ void cg_dma16(unsigned lds_dst, unsigned voff, const void *sbase) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int OFF>
__device__ __forceinline__ void cg_wload(cg_u32x4 &dst, unsigned voff, const void *sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void cg_wait4(cg_u32x4 &a, cg_u32x4 &b, cg_u32x4 &c, cg_u32x4 &d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT) : "memory");
}

template <int KT, int FT>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmParams p) {
    constexpr int HALO = KT / 2;
    constexpr int LROW = cg_lrow(KT, FT);             // LDS slots per (plane, quarter) row: >= 16*FT + 2*HALO, multiple of 32
    constexpr int NSLOT = 8 * LROW;                   // per stage: 2 planes x 4 channel quarters
    constexpr int STAGE = NSLOT * 16;                 // bytes
    constexpr int NP = LROW / 32;                     // 1 KB staging pieces per wave and chunk
    constexpr int NWL = 4 * KT;                       // weight fragment loads per wave and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char cg_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, j16 = lane & 15;

    // workgroup -> (frame tile, output-channel tile, utterance).  Workgroups are dealt round-robin over the 8 XCDs:
    // give each XCD a contiguous run of utterances with ALL their output-channel tiles, so that an utterance's
    // activations are fetched into one L2 only (speed, never correctness).
    const int ny = p.cpad / CG_TO;
    const unsigned total = gridDim.x;
    unsigned lin = blockIdx.x;
    if ((total & 7u) == 0u) lin = (blockIdx.x & 7u) * (total >> 3) + (blockIdx.x >> 3);
    const int by = (int)(lin % (unsigned)ny);
    const int bx = (int)((lin / (unsigned)ny) % (unsigned)p.nx);
    const int b = (int)(lin / (unsigned)(ny * p.nx));
    const int o0 = by * CG_TO, f0 = bx * 16 * FT;

    // staging: linear LDS slot i = 64*piece + lane <-> (plane, quarter, slot) -> this lane's source offset (constant
    // over the chunks: the chunk moves the scalar base)
    unsigned xvoff[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int i = 64 * (wave * NP + k) + lane;
        const int plane = i / (4 * LROW), qq = (i / LROW) & 3, s = i % LROW;
        xvoff[k] = (unsigned)(((size_t)plane * p.xs_plane + (size_t)qq * p.S + f0 + s) * 16);
    }
    const unsigned char *xbase = reinterpret_cast<const unsigned char *>(p.xs + (size_t)b * p.nch * 4 * p.S);
    const size_t xchunk = (size_t)4 * p.S * 16;                              // bytes from one chunk to the next
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)cg_smem;   // LDS byte address
    auto stage_x = [&](int c, int st) {
        const unsigned char *sb = xbase + (size_t)c * xchunk;
#pragma unroll
        for (int k = 0; k < NP; ++k) cg_dma16(lds0 + st * STAGE + (wave * NP + k) * 1024, xvoff[k], sb);
    };
    // weights: this wave's two 16-channel tiles, both planes, are 4 KB in a row per (chunk, tap)
    const unsigned wvoff = (unsigned)lane * 16u + (unsigned)((o0 / 16 + 2 * wave) * 2) * 1024u;
    const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.wp);
    const size_t wtap = (size_t)(p.cpad / 16) * 2 * 1024;                    // bytes per (chunk, tap)
    cg_u32x4 W[KT][4];                                                       // [tap][tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo]
    auto load_w = [&](int c, int tap, cg_u32x4 (&w4)[4]) {
        const unsigned char *sb = wbase + ((size_t)c * KT + tap) * wtap;
        cg_wload<0>(w4[0], wvoff, sb);
        cg_wload<1024>(w4[1], wvoff, sb);
        cg_wload<2048>(w4[2], wvoff, sb);
        cg_wload<3072>(w4[3], wvoff, sb);
    };

    cg_f32x4 acc[2][FT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int n = 0; n < FT; ++n) acc[a][n] = cg_f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: chunk 0's activations and weights
    stage_x(0, 0);
#pragma unroll
    for (int t = 0; t < KT; ++t) load_w(0, t, W[t]);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWL) : "memory");             // this wave's staging pieces have landed ...
    __builtin_amdgcn_s_barrier();                                            // ... and everybody's
    const unsigned lbase = (unsigned)(q * LROW + j16) * 16u;

    const int nch = p.nch;
    for (int c = 0; c < nch; ++c) {
        const int cn = c + 1 < nch ? c + 1 : c;                              // always issue: the counts below stay exact
        stage_x(cn, (c + 1) & 1);                                            // (the last chunk again, into the dead stage)
        const unsigned char *st = cg_smem + (c & 1) * STAGE + lbase;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            // W[t] of this chunk is the oldest operation in flight; younger: the other taps (4 each), the staging pieces
            cg_wait4<NP + 4 * (KT - 1)>(W[t][0], W[t][1], W[t][2], W[t][3]);
            const cg_bf16x8 wh0 = __builtin_bit_cast(cg_bf16x8, W[t][0]), wl0 = __builtin_bit_cast(cg_bf16x8, W[t][1]);
            const cg_bf16x8 wh1 = __builtin_bit_cast(cg_bf16x8, W[t][2]), wl1 = __builtin_bit_cast(cg_bf16x8, W[t][3]);
            // fragments one frame tile ahead of their MFMAs (left alone, hipcc issues a tile's two reads and waits for the
            // first at once: an LDS round trip per six MFMAs)
            cg_bf16x8 xh = *reinterpret_cast<const cg_bf16x8 *>(st + t * 16);
            cg_bf16x8 xl = *reinterpret_cast<const cg_bf16x8 *>(st + (4 * LROW + t) * 16);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);              // (tile 0's reads: a group of their own)
#pragma unroll
            for (int n = 0; n < FT; ++n) {
                cg_bf16x8 nh = xh, nl = xl;
                if (n + 1 < FT) {
                    nh = *reinterpret_cast<const cg_bf16x8 *>(st + (16 * (n + 1) + t) * 16);
                    nl = *reinterpret_cast<const cg_bf16x8 *>(st + (4 * LROW + 16 * (n + 1) + t) * 16);
                }
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh0, acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, wh1, acc[1][n], 0, 0, 0);
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl0, acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wl1, acc[1][n], 0, 0, 0);
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh0, acc[0][n], 0, 0, 0);
                acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, wh1, acc[1][n], 0, 0, 0);
                // 2 reads, then 6 MFMAs: keep this tile's reads (for the next tile) ahead of this tile's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);          // MFMA
                xh = nh;
                xl = nl;
            }
            // the MFMAs above have read W[t] before the loads below may overwrite it
            asm volatile("" : "+v"(acc[0][FT - 1]), "+v"(acc[1][FT - 1]));
            __builtin_amdgcn_sched_barrier(0);
            load_w(cn, t, W[t]);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWL) : "memory");         // the next chunk's staging pieces have landed
        __builtin_amdgcn_s_barrier();                                        // everybody's; and this chunk's stage is free
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // (the clamped re-loads of the last chunk)

    // epilogue.  C/D layout of 16x16: column = lane & 15 (output channel), rows 4*(lane >> 4) + r (frames)
    const bool vec = (p.T & 3) == 0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int o = o0 + 32 * wave + 16 * a + j16;
        if (o >= p.Cout) continue;
        const float bv = p.bias ? p.bias[o] : 0.f;
        float *yr = p.y + ((size_t)b * p.Cout + o) * p.T;
#pragma unroll
        for (int n = 0; n < FT; ++n) {
            const int f = f0 + 16 * n + 4 * q;
            cg_f32x4 v = acc[a][n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] += bv;
                if (p.relu) v[r] = fmaxf(v[r], 0.f);
            }
            if (vec) {
                if (f < p.T) *reinterpret_cast<cg_f32x4 *>(yr + f) = v;       // T % 4 == 0: a quad is all in or all out
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f + r < p.T) yr[f + r] = v[r];
            }
        }
    }
}

template <int KT, int FT>
static int launch_conv_gemm(const ConvGemmParams &p, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 8 * cg_lrow(KT, FT) * 16;
    auto kern = conv_gemm_kernel<KT, FT>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    const unsigned grid = (unsigned)p.nx * (unsigned)(p.cpad / CG_TO) * (unsigned)p.B;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

// ---- what softattn.hip's entry points call ----
bool conv_gemm_applies(int Cin, int Cout, int K) {
    // wide layers only: >= one full workgroup of output channels and enough input channels to amortise the pipeline;
    // the narrow mel / projection layers keep conv1d_prepared_kernel
    return Cout >= 128 && Cin >= 64 && (K == 1 || K == 3 || K == 5);
}
size_t conv_gemm_prepared_bytes(int Cout, int Cin, int K) { return conv_gemm_layout(1, Cin, Cout, 16, K).w_bytes; }
size_t conv_gemm_workspace_bytes(int B, int Cin, int Cout, int T, int K) { return conv_gemm_layout(B, Cin, Cout, T, K).xs_bytes; }

int conv_gemm_prepare(const float *w, void *prepared, int Cout, int Cin, int K, hipStream_t s) {
    const ConvGemmLayout L = conv_gemm_layout(1, Cin, Cout, 16, K);
    const int nfrag = L.nch * K * (L.cpad / 16) * 64;
    hipLaunchKernelGGL(conv_gemm_wprep_kernel, dim3((nfrag + 255) / 256), dim3(256), 0, s, w, static_cast<uint4 *>(prepared),
                       Cout, Cin, K, L.cpad, nfrag);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

int conv_gemm_run(const float *x, const void *prepared, const float *bias, float *y, void *workspace, size_t workspace_bytes,
                  int B, int Cin, int Cout, int T, int K, int relu, hipStream_t s) {
    const ConvGemmLayout L = conv_gemm_layout(B, Cin, Cout, T, K);
    if (workspace_bytes < L.xs_bytes) return fail(ALIGNER_ENOSPC, "conv workspace %zu < %zu bytes", workspace_bytes, L.xs_bytes);
    if (L.xs_bytes >= (1ull << 32)) return fail(ALIGNER_EDOM, "split activations of %zu bytes exceed 32-bit offsets", L.xs_bytes);
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(ALIGNER_EINVAL, "conv workspace must be 16-byte aligned");
    uint4 *xs = static_cast<uint4 *>(workspace);
    const size_t total = L.xs_plane;
    hipLaunchKernelGGL(conv_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, xs, L.xs_plane, Cin, T, L.S,
                       L.nch, K / 2, total);
    ALIGNER_HIP_CHECK(hipGetLastError());
    ConvGemmParams p{xs, static_cast<const uint4 *>(prepared), bias, y, (unsigned long long)L.xs_plane, B, Cout, T, L.S, L.nch,
                     L.cpad, relu, 1};
    // frames per workgroup: one utterance's whole T when it fits 13 tiles of 16 (T = 200 -> 208), else the tile count that
    // wastes least
    int FT = 13;
    if (T > 16 * 13) {
        const int w13 = (T + 207) / 208 * 208, w8 = (T + 127) / 128 * 128;
        FT = w8 < w13 ? 8 : 13;
    } else if (T <= 128) {
        FT = 8;
    }
    p.nx = (T + 16 * FT - 1) / (16 * FT);
    if ((unsigned long long)p.nx * (L.cpad / CG_TO) * B >= (1ull << 31)) return fail(ALIGNER_EDOM, "grid too large");
#define CG_LAUNCH(KT)                                                              \
    return FT == 13 ? launch_conv_gemm<KT, 13>(p, s) : launch_conv_gemm<KT, 8>(p, s)
    if (K == 1) { CG_LAUNCH(1); }
    if (K == 3) { CG_LAUNCH(3); }
    CG_LAUNCH(5);
#undef CG_LAUNCH
}

}  // namespace aligner
