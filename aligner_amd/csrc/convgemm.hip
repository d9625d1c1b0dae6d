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
//    16-byte stores into [B, Cout, T] -- or, when the next layer of the stack is a k = 1 convolution, the operands swap
//    roles (output channels on the M axis: a lane holds 4 consecutive channels of one frame) and the epilogue writes
//    the NEXT layer's split channels-last fragments directly (8-byte stores): a stack costs one split pass, for its
//    first layer only.
//  * Narrow layers (fewer than 128 output channels: the mel encoder, the projections down to the attention
//    channels) turn the roles around -- conv_narrow_kernel: every wave owns ALL frames of the workgroup's tile and
//    16 or 32 output channels; the activations of ALL input chunks are staged at once (one barrier, no ring: these
//    layers are bound by HBM, not by the matrix cores), the weight fragments stream through a four-step register ring.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

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

__device__ __forceinline__ float and_maskf(float v, unsigned m) {            // v or +0.0 without a branch
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m);
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
// CS_U slots a thread, 256 apart (every load and store instruction of a wave stays one contiguous run), all their loads in
// flight before the first is used: with one slot a thread a workgroup lived for one memory round trip and the pass was bound
// by that, not by bytes (C3, 52 MB: one slot a thread 14.1 us, four 12.1, eight 12.6).
constexpr int CS_U = 4;
__global__ __launch_bounds__(256) void conv_split_kernel(const float *__restrict__ x, uint4 *__restrict__ xs, size_t xs_plane,
                                                         int Cin, int T, int S, int nch, int halo, size_t total) {
    const size_t idx0 = (size_t)blockIdx.x * (256 * CS_U) + threadIdx.x;       // ((b*nch + chunk)*4 + quarter)*S + slot
    float v[CS_U][8];
    bool in[CS_U];
#pragma unroll
    for (int u = 0; u < CS_U; ++u) {
        const size_t idx = idx0 + (size_t)u * 256;
        const size_t ic = idx < total ? idx : total - 1;
        const int s = (int)(ic % S);
        const size_t r = ic / S;
        const int qq = (int)(r & 3), c = (int)((r >> 2) % nch);
        const size_t b = (r >> 2) / nch;
        const int t = s - halo;
        in[u] = t >= 0 && t < T;
        const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = CG_CH * c + 8 * qq + j;
            const float ld = x[((size_t)b * Cin + (i < Cin ? i : Cin - 1)) * T + tc];   // unconditional, masked afterwards
            v[u][j] = i < Cin ? ld : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < CS_U; ++u) {
        const size_t idx = idx0 + (size_t)u * 256;
        if (idx >= total) break;
        cg_bf16x8 hv, lv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 hh, ll;
            cg_split(in[u] ? v[u][j] : 0.f, hh, ll);
            hv[j] = hh;
            lv[j] = ll;
        }
        xs[idx] = __builtin_bit_cast(uint4, hv);
        xs[xs_plane + idx] = __builtin_bit_cast(uint4, lv);
    }
}

struct ConvGemmParams {
    const uint4 *xs;        // split activations, plane 0 (plane 1 at + xs_plane)
    const uint4 *wp;        // prepared weights
    const float *bias;      // nullable
    float *y;               // [B, Cout, T] fp32 output (SPLIT = false)
    uint4 *ys;              // SPLIT: the next (k = 1) layer's split activations, plane 0
    unsigned long long xs_plane, ys_plane;
    int B, Cout, T, S, nch, cpad, relu, nx;
    int S2, nch2;           // SPLIT: slots per row and 32-channel chunks of the next layer's image
    const float *xf;        // narrow forms: the activations as fp32 [B, Cin, T] instead of xs (staged and split by the kernel)
    int Cin;                // ... and their channel count
};

#ifndef CG_DMA_BITS
#define CG_DMA_BITS ""
#endif
// ---- hand-issued memory operations (the compiler neither counts nor waits for them: every wait below is ours) ----
// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses (sbase + voff) to LDS [lds_dst + 16*lane]
__device__ __forceinline__ void cg_dma16(unsigned lds_dst, unsigned voff, const void *sbase) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 " CG_DMA_BITS "\n\ts_mov_b32 m0, %0"
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
template <int CNT>
__device__ __forceinline__ void cg_wait2(cg_u32x4 &a, cg_u32x4 &b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT) : "memory");
}

// ---- epilogues, shared by both kernels: NT output-channel tiles x FT frame tiles of 16x16 accumulators ----
// fp32 [B, Cout, T]: the accumulators were built with frames on the M axis (mfma(x, w)): column = lane & 15 is the
// output channel, rows 4*(lane >> 4) + r are consecutive frames -> one 16-byte store per tile
template <int NT, int FT>
__device__ __forceinline__ void cg_store_f32(const ConvGemmParams &p, const cg_f32x4 (&acc)[NT][FT], int b, int otile0, int f0,
                                             int lane) {
    const int q = lane >> 4, j16 = lane & 15;
    const bool vec = (p.T & 3) == 0;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int o = 16 * (otile0 + a) + j16;
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
// the next layer's split image (that layer has k = 1: no halo, slot = frame): the accumulators were built with output
// channels on the M axis (mfma(w, x)): column = lane & 15 is the frame, rows 4*(lane >> 4) + r are consecutive output
// channels = half of one 16-byte fragment (8 channels) -> one 8-byte store per plane and tile; lanes q and q ^ 1 fill
// the two halves of a fragment, 16 consecutive frames are 256 bytes in a row.  Frames >= T inside the tile and
// channels >= Cout (zero weights, no bias) are written as zeros: the consumer's padding is data.
template <int NT, int FT>
__device__ __forceinline__ void cg_store_split(const ConvGemmParams &p, const cg_f32x4 (&acc)[NT][FT], int b, int otile0, int f0,
                                               int lane) {
    const int q = lane >> 4, j16 = lane & 15;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int ob = 16 * (otile0 + a) + 4 * q;                            // this lane's 4 channels: ob .. ob+3
        if (ob >= 32 * p.nch2) continue;                                     // (a tile past the consumer's last chunk)
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {                                                        // (uniform) clamped loads, masked: no branch per element
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ob + r;
                bv[r] = and_maskf(p.bias[o < p.Cout ? o : p.Cout - 1], o < p.Cout ? ~0u : 0u);
            }
        }
        // image [b][chunk = ob / 32][quarter = (ob % 32) / 8][slot] of 16 bytes; this lane's half at + 8 * ((ob % 8) / 4)
        const size_t row = (((size_t)b * p.nch2 + (ob >> 5)) * 4 + ((ob & 31) >> 3)) * p.S2;
        unsigned char *dst0 = reinterpret_cast<unsigned char *>(p.ys + row) + 8 * ((ob & 7) >> 2);
#pragma unroll
        for (int n = 0; n < FT; ++n) {
            const int f = f0 + 16 * n + j16;
            if (f >= p.S2) continue;
            bf16x4 hv, lv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[a][n][r] + bv[r];
                if (p.relu) v = fmaxf(v, 0.f);
                v = and_maskf(v, (f < p.T && ob + r < p.Cout) ? ~0u : 0u);
                __bf16 hh, ll;
                cg_split(v, hh, ll);
                hv[r] = hh;
                lv[r] = ll;
            }
            unsigned char *d = dst0 + (size_t)f * 16;
            *reinterpret_cast<bf16x4 *>(d) = hv;
            *reinterpret_cast<bf16x4 *>(d + p.ys_plane * 16) = lv;
        }
    }
}

// 6 MFMAs of one (frame tile, output tile pair): products lo*hi, hi*lo, hi*hi; SPLIT swaps the operands' roles
template <bool SPLIT>
__device__ __forceinline__ cg_f32x4 cg_mfma(cg_bf16x8 x, cg_bf16x8 w, cg_f32x4 c) {
    return SPLIT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, w, c, 0, 0, 0);
}

template <int KT, int FT, bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmParams p) {
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
        // fragments one frame tile ahead of their MFMAs, ACROSS the taps of the chunk (left alone, hipcc issues a tile's two
        // reads and waits for the first at once: an LDS round trip per six MFMAs; with the look-ahead restarted per tap a
        // round trip per tap was exposed: three per chunk)
        cg_bf16x8 xh = *reinterpret_cast<const cg_bf16x8 *>(st);
        cg_bf16x8 xl = *reinterpret_cast<const cg_bf16x8 *>(st + (4 * LROW) * 16);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                  // (the chunk's first reads: a group of their own)
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            // W[t] of this chunk is the oldest operation in flight; younger: the other taps (4 each), the staging pieces
            cg_wait4<NP + 4 * (KT - 1)>(W[t][0], W[t][1], W[t][2], W[t][3]);
            const cg_bf16x8 wh0 = __builtin_bit_cast(cg_bf16x8, W[t][0]), wl0 = __builtin_bit_cast(cg_bf16x8, W[t][1]);
            const cg_bf16x8 wh1 = __builtin_bit_cast(cg_bf16x8, W[t][2]), wl1 = __builtin_bit_cast(cg_bf16x8, W[t][3]);
#pragma unroll
            for (int n = 0; n < FT; ++n) {
                cg_bf16x8 nh = xh, nl = xl;
                if (n + 1 < FT) {
                    nh = *reinterpret_cast<const cg_bf16x8 *>(st + (16 * (n + 1) + t) * 16);
                    nl = *reinterpret_cast<const cg_bf16x8 *>(st + (4 * LROW + 16 * (n + 1) + t) * 16);
                } else if (t + 1 < KT) {                                     // the next tap's first tile
                    nh = *reinterpret_cast<const cg_bf16x8 *>(st + (t + 1) * 16);
                    nl = *reinterpret_cast<const cg_bf16x8 *>(st + (4 * LROW + t + 1) * 16);
                }
                acc[0][n] = cg_mfma<SPLIT>(xl, wh0, acc[0][n]);
                acc[1][n] = cg_mfma<SPLIT>(xl, wh1, acc[1][n]);
                acc[0][n] = cg_mfma<SPLIT>(xh, wl0, acc[0][n]);
                acc[1][n] = cg_mfma<SPLIT>(xh, wl1, acc[1][n]);
                acc[0][n] = cg_mfma<SPLIT>(xh, wh0, acc[0][n]);
                acc[1][n] = cg_mfma<SPLIT>(xh, wh1, acc[1][n]);
                // 2 reads, then 6 MFMAs: keep this tile's reads (for the next tile) ahead of this tile's MFMAs
                if (n + 1 < FT || t + 1 < KT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
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
    // the clamped re-loads of the last chunk are still in flight INTO W: the wait names every register of the ring, or
    // hipcc -- for whom they are dead after the last MFMA -- reuses them in the epilogue under the landing loads
#pragma unroll
    for (int t = 0; t < KT; ++t) cg_wait4<0>(W[t][0], W[t][1], W[t][2], W[t][3]);

    if (SPLIT) cg_store_split<2, FT>(p, acc, b, o0 / 16 + 2 * wave, f0, lane);
    else       cg_store_f32<2, FT>(p, acc, b, o0 / 16 + 2 * wave, f0, lane);
}

// ---- narrow layers: Cout <= 256 (in practice the 80 / 160 channels of the mel encoder and the attention projections) ----
// A workgroup = NW waves = one utterance's 16*FT frames x ALL output channels; wave w owns output tiles NT*w .. NT*w+NT-1
// (16 channels each) and every frame tile.  The activations of ALL chunks are staged at once (LDS image
// [chunk][plane][quarter][slot]): these layers have 3 to 32 chunks of a few hundred MFMA cycles each -- a ring with a
// barrier per chunk would be all latency; what overlaps staging and arithmetic here is the other workgroup of the CU.
// The weight fragments stream through a ring of four (chunk, tap) steps in registers, as in the wide kernel.
constexpr int CN_RING = 4;           // weight-ring depth of the one-layer kernel; the fused kernel picks per layer (4 or 8)

template <int I, int N, class F>
__device__ __forceinline__ void cg_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cg_static_for<I + 1, N>(f);
    }
}

// the weight ring of one wave: RD (chunk, tap) steps of NT output tiles x 2 planes, loaded straight into registers
template <int NT, int RD = CN_RING>
struct CnRing {
    cg_u32x4 W[RD][2 * NT];
    unsigned wvoff;
    const unsigned char *wbase;
    size_t wstep;
    int nstep, rot;
    // wt: which pair / tile of output channels this wave owns.  rot: the workgroup walks the steps from a start of its own
    // (step s of the loop is step (s + rot) % nstep of the layer: a sum's order, nothing else)
    __device__ __forceinline__ void init(const uint4 *wp, int cpad, int nstep_, int wt, int lane, int rot_ = 0) {
        wvoff = (unsigned)lane * 16u + (unsigned)(NT * wt * 2) * 1024u;      // this wave's tiles: 2 KB each, in a row
        wbase = reinterpret_cast<const unsigned char *>(wp);
        wstep = (size_t)(cpad / 16) * 2 * 1024;                             // bytes per step
        nstep = nstep_;
        rot = rot_ % nstep_;
    }
    __device__ __forceinline__ int phys(int s) const {                       // loop step -> the layer's step
        int ps = (s < nstep ? s : nstep - 1) + rot;                          // clamped: the counts stay exact
        return ps >= nstep ? ps - nstep : ps;
    }
    template <int I>
    __device__ __forceinline__ void load(int s) {
        const unsigned char *sb = wbase + (size_t)phys(s) * wstep;
        cg_wload<0>(W[I][0], wvoff, sb);
        cg_wload<1024>(W[I][1], wvoff, sb);
        if (NT == 2) {
            cg_wload<2048>(W[I][2 * NT - 2], wvoff, sb);
            cg_wload<3072>(W[I][2 * NT - 1], wvoff, sb);
        }
    }
    __device__ __forceinline__ void prime() {                               // steps 0 .. RD-1
        cg_static_for<0, RD>([&](auto ic) { load<decltype(ic)::value>(decltype(ic)::value); });
    }
    template <int I, int CNT>
    __device__ __forceinline__ void wait() {                                // ... until at most CNT younger operations are in flight
        if (NT == 2) cg_wait4<CNT>(W[I][0], W[I][1], W[I][2], W[I][3]);
        else         cg_wait2<CNT>(W[I][0], W[I][1]);
    }
    // The clamped re-loads of the last steps are still in flight INTO the ring's registers when the steps are done: this wait
    // names every one of them (PEND = operations issued since that may stay in flight), or hipcc -- for whom they are dead
    // after the last MFMA -- hands them to the next instructions and the landing loads overwrite those (seen: a lane's
    // channel index in the epilogue, i.e. stores to another wave's channels, one run in a few).
    template <int PEND>
    __device__ __forceinline__ void drain() {
        cg_static_for<0, RD>([&](auto ic) { wait<decltype(ic)::value, PEND>(); });
    }
};

// one step of one layer for one wave: X fragments from the LDS image at `xl` ([chunk][plane][quarter][LROW] of 16 bytes;
// the caller has added the wave's first frame tile), W from slot I of the (primed) ring; FTW frame tiles.  WA: the
// weights are the MFMA's A operand (output channels on the M axis: split-format epilogues), else the activations are
// (frames on the M axis: fp32 epilogue)
template <int KT, int LROW, int FTW, int NT, int RD, bool WA, int I>
__device__ __forceinline__ void cn_step(CnRing<NT, RD> &R, const unsigned char *xl, int s, int lane, cg_f32x4 (&acc)[NT][FTW]) {
    constexpr int CHSLOT = 8 * LROW;
    R.template wait<I, (RD - 1) * 2 * NT>();
    if (s < R.nstep) {                                                       // (uniform; only the last group is partial)
        const int ps = R.phys(s);
        const int c = ps / KT, t = ps - c * KT;
        const unsigned char *st = xl + (size_t)c * CHSLOT * 16 + (unsigned)((lane >> 4) * LROW + (lane & 15)) * 16u + t * 16;
        const cg_bf16x8 wh0 = __builtin_bit_cast(cg_bf16x8, R.W[I][0]), wl0 = __builtin_bit_cast(cg_bf16x8, R.W[I][1]);
        const cg_bf16x8 wh1 = __builtin_bit_cast(cg_bf16x8, R.W[I][2 * NT - 2]), wl1 = __builtin_bit_cast(cg_bf16x8, R.W[I][2 * NT - 1]);
        // fragments one frame tile ahead of their MFMAs (as in the wide kernel: left alone, hipcc issues a tile's reads and
        // waits for them at once -- an LDS round trip per 3 * NT MFMAs, with one or two waves per SIMD to hide it)
        cg_bf16x8 xh = *reinterpret_cast<const cg_bf16x8 *>(st);
        cg_bf16x8 xlo = *reinterpret_cast<const cg_bf16x8 *>(st + (4 * LROW) * 16);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int n = 0; n < FTW; ++n) {
            cg_bf16x8 nh = xh, nl = xlo;
            if (n + 1 < FTW) {
                nh = *reinterpret_cast<const cg_bf16x8 *>(st + (16 * (n + 1)) * 16);
                nl = *reinterpret_cast<const cg_bf16x8 *>(st + (4 * LROW + 16 * (n + 1)) * 16);
            }
            acc[0][n] = cg_mfma<WA>(xlo, wh0, acc[0][n]);
            if (NT == 2) acc[NT - 1][n] = cg_mfma<WA>(xlo, wh1, acc[NT - 1][n]);
            acc[0][n] = cg_mfma<WA>(xh, wl0, acc[0][n]);
            if (NT == 2) acc[NT - 1][n] = cg_mfma<WA>(xh, wl1, acc[NT - 1][n]);
            acc[0][n] = cg_mfma<WA>(xh, wh0, acc[0][n]);
            if (NT == 2) acc[NT - 1][n] = cg_mfma<WA>(xh, wh1, acc[NT - 1][n]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);              // DS read (the next tile's)
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NT, 0);         // MFMA (this tile's)
            xh = nh;
            xlo = nl;
        }
    }
    // the MFMAs above have read the slot before the loads below may overwrite it
    if (NT == 2) asm volatile("" : "+v"(acc[0][FTW - 1]), "+v"(acc[NT - 1][FTW - 1]));
    else         asm volatile("" : "+v"(acc[0][FTW - 1]));
    __builtin_amdgcn_sched_barrier(0);
    R.template load<I>(s + RD);
}
template <int KT, int LROW, int FTW, int NT, int RD, bool WA>
__device__ __forceinline__ void cn_steps(CnRing<NT, RD> &R, const unsigned char *xl, int lane, cg_f32x4 (&acc)[NT][FTW]) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int n = 0; n < FTW; ++n) acc[a][n] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < R.nstep; s0 += RD)
        cg_static_for<0, RD>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            cn_step<KT, LROW, FTW, NT, RD, WA, I>(R, xl, s0 + I, lane, acc);
        });
}

// stage every chunk of one frame tile into LDS: piece = 64 consecutive slots, dealt round-robin to the waves
template <int KT, int FT>
__device__ __forceinline__ void cn_stage_all(const uint4 *xs, unsigned long long xs_plane, int S, int nch, int b, int f0, int wave,
                                             int nw, int lane, unsigned lds_dst) {
    constexpr int LROW = cg_lrow(KT, FT), CHSLOT = 8 * LROW;
    const unsigned char *xbase = reinterpret_cast<const unsigned char *>(xs + (size_t)b * nch * 4 * S);
    const int npiece = nch * CHSLOT / 64;
    for (int pc = wave; pc < npiece; pc += nw) {
        const int i = 64 * pc + lane;
        const int c = i / CHSLOT, r = i - c * CHSLOT;
        const int plane = r / (4 * LROW), qq = (r / LROW) & 3, s = r % LROW;
        const unsigned voff = (unsigned)(((size_t)plane * xs_plane + ((size_t)c * 4 + qq) * S + f0 + s) * 16);
        cg_dma16(lds_dst + (unsigned)pc * 1024u, voff, xbase);
    }
}

// The same LDS image straight from fp32 activations [B, Cin, T]: a narrow layer at the head of a stack (the mel encoder's
// first) needs no split pass over HBM -- its workgroup reads each input element once anyway.  One item = 8 input channels
// of one slot (what conv_split_kernel's thread does), four items' loads in flight per thread; consecutive lanes take
// consecutive frames (256-byte row segments in, 16-byte ds_writes out, 1 KB contiguous per wave and plane).
template <int KT, int FT>
__device__ __forceinline__ void cn_stage_f32(const float *__restrict__ x, int Cin, int T, int nch, int b, int f0, int tid,
                                             int nthreads, unsigned char *lds) {
    constexpr int LROW = cg_lrow(KT, FT), HALO = KT / 2, U = 4;
    const int items = nch * 4 * LROW;
    const float *xb = x + (size_t)b * Cin * T;
    uint4 *img = reinterpret_cast<uint4 *>(lds);
    for (int it0 = tid; it0 < items; it0 += U * nthreads) {
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * nthreads;
            const int itc = it < items ? it : items - 1;
            const int sl = itc % LROW, cq = itc / LROW;
            const int t = f0 + sl - HALO;
            const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 8 * cq + j;                                       // 32 * chunk + 8 * quarter + j
                v[u][j] = xb[(size_t)(i < Cin ? i : Cin - 1) * T + tc];        // unconditional, masked below
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * nthreads;
            if (it < items) {
                const int sl = it % LROW, cq = it / LROW;
                const int t = f0 + sl - HALO;
                const bool in = t >= 0 && t < T;
                cg_bf16x8 hv, lv;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 hh, ll;
                    cg_split((in && 8 * cq + j < Cin) ? v[u][j] : 0.f, hh, ll);
                    hv[j] = hh;
                    lv[j] = ll;
                }
                uint4 *d = img + ((size_t)(cq >> 2) * 8 + (cq & 3)) * LROW + sl;   // [chunk][plane][quarter][slot]
                d[0] = __builtin_bit_cast(uint4, hv);
                d[4 * LROW] = __builtin_bit_cast(uint4, lv);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          // the ds_writes, before the caller's barrier
}

template <int KT, int FT, int NT, bool SPLIT>
__global__ __launch_bounds__(512) void conv_narrow_kernel(ConvGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cg_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)(blockDim.x >> 6);
    const int bx = (int)(blockIdx.x % (unsigned)p.nx), b = (int)(blockIdx.x / (unsigned)p.nx);
    const int f0 = bx * 16 * FT;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)cg_smem;
    if (p.xf) cn_stage_f32<KT, FT>(p.xf, p.Cin, p.T, p.nch, b, f0, tid, (int)blockDim.x, cg_smem);
    else      cn_stage_all<KT, FT>(p.xs, p.xs_plane, p.S, p.nch, b, f0, wave, nw, lane, lds0);
    CnRing<NT> R;
    R.init(p.wp, p.cpad, p.nch * KT, wave, lane, (int)blockIdx.x);
    R.prime();
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CN_RING * 2 * NT) : "memory");   // this wave's staging pieces have landed ...
    __builtin_amdgcn_s_barrier();                                               // ... and everybody's
    cg_f32x4 acc[NT][FT];
    cn_steps<KT, cg_lrow(KT, FT), FT, NT, CN_RING, SPLIT>(R, cg_smem, lane, acc);
    R.template drain<0>();
    if (SPLIT) cg_store_split<NT, FT>(p, acc, b, NT * wave, f0, lane);
    else       cg_store_f32<NT, FT>(p, acc, b, NT * wave, f0, lane);
}

// ---- a narrow k = 1 layer with MANY input chunks (the text stack's 1024 -> 80 projection: 32 chunks) ----
// conv_narrow_kernel stages ALL chunks before its first MFMA: 128 KB for a 32-frame tile -- one workgroup a CU, six
// waves, staging and arithmetic one after the other (30 us for 52 MB at C3: 1.7 TB/s; this form: 17 us).  Here the chunks come in GROUPS of
// CNR_G through CNR_NB buffers of LDS (64 KB: two workgroups a CU): a loader wave keeps three groups in flight by LDS-DMA
// and counts them in (vmcnt retires in order, and nothing else of that wave is in flight), the compute waves' only
// loads are their weight rings (whose counted waits stay exact), one raw s_barrier per group: "group g has landed" and
// "group g - 1 has been read" in one (the loader refills that buffer right behind it).  A group is CNR_G steps = one
// turn of the four-step weight ring.
constexpr int CNR_G = 4, CNR_NB = 4;
static_assert(CNR_G == CN_RING, "a group of chunks is one turn of the weight ring (k = 1: a step per chunk)");

template <int FT, int NT, bool SPLIT>
__global__ __launch_bounds__(64 * 9) void conv_narrow_ring_kernel(ConvGemmParams p) {
    constexpr int LROW = cg_lrow(1, FT), CHSLOT = 8 * LROW;
    constexpr int GB = CNR_G * CHSLOT * 16;                      // bytes of a group in LDS
    constexpr int PG = GB / 1024;                                // its 1 KB LDS-DMA pieces
    constexpr int W2 = 2 * PG < 63 ? 2 * PG : 63;                // (the counter's field has 6 bits: a wait for one piece more than needed)
    extern __shared__ __attribute__((aligned(16))) unsigned char cg_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)(blockDim.x >> 6) - 1;                   // compute waves; wave nw is the loader
    const int bx = (int)(blockIdx.x % (unsigned)p.nx), b = (int)(blockIdx.x / (unsigned)p.nx);
    const int f0 = bx * 16 * FT;
    const int ng = p.nch / CNR_G;                                // (the host takes this form for whole groups only)
    if (wave == nw) {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)cg_smem;
        const unsigned char *xbase = reinterpret_cast<const unsigned char *>(p.xs + (size_t)b * p.nch * 4 * p.S);
        auto issue = [&](int g) {
#pragma unroll
            for (int q = 0; q < PG; ++q) {
                const int i = 64 * (g * PG + q) + lane;          // cn_stage_all's piece g * PG + q
                const int c = i / CHSLOT, r = i - c * CHSLOT;
                const int plane = r / (4 * LROW), qq = (r / LROW) & 3, sl = r % LROW;
                const unsigned voff = (unsigned)(((size_t)plane * p.xs_plane + ((size_t)c * 4 + qq) * p.S + f0 + sl) * 16);
                cg_dma16(lds0 + (unsigned)(g % CNR_NB) * GB + (unsigned)q * 1024u, voff, xbase);
            }
        };
        for (int g = 0; g < CNR_NB - 1 && g < ng; ++g) issue(g);
        for (int g = 0; g < ng; ++g) {
            const int last = g + CNR_NB - 2 < ng - 1 ? g + CNR_NB - 2 : ng - 1;     // the youngest group in flight
            const int later = last - g;                                             // uniform: 0 .. CNR_NB - 2
            if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W2) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PG) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                        // group g is in LDS; group g - 1's buffer has been read
            if (g + CNR_NB - 1 < ng) issue(g + CNR_NB - 1);      // ... and is refilled
        }
        return;
    }
    static_assert(CNR_NB - 2 == 2, "the loader's three wait cases");
    CnRing<NT> R;
    R.init(p.wp, p.cpad, p.nch, wave, lane, 0);                  // (no rotation: the groups arrive in order)
    R.prime();
    cg_f32x4 acc[NT][FT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int n = 0; n < FT; ++n) acc[a][n] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < ng; ++g) {
        __builtin_amdgcn_s_barrier();
        // cn_step addresses chunk c at xl + c * CHSLOT * 16: chunk CNR_G * g + i sits in buffer g % CNR_NB at i
        const unsigned char *xl = cg_smem + (size_t)(g % CNR_NB) * GB - (size_t)g * GB;
        cg_static_for<0, CN_RING>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            cn_step<1, LROW, FT, NT, CN_RING, SPLIT, I>(R, xl, CNR_G * g + I, lane, acc);
        });
    }
    R.template drain<0>();
    if (SPLIT) cg_store_split<NT, FT>(p, acc, b, NT * wave, f0, lane);
    else       cg_store_f32<NT, FT>(p, acc, b, NT * wave, f0, lane);
}

// ---- a run of narrow layers in ONE kernel (the mel encoder: 80 -> 160 k=3, 160 -> 80, 80 -> 80) ----
// Layers after the first have k = 1, so a frame tile of layer l+1 needs exactly the same frames of layer l: the
// workgroup keeps its tile on the CU.  Layer l's epilogue writes the split image of layer l+1's input into a second LDS
// buffer (same [chunk][plane][quarter][slot] layout the global images have: 8-byte ds_writes, a lane's 4 channels of one
// frame), one barrier, and the same step loop runs again on it with the next layer's weight ring -- primed before the
// epilogue, so the ring's first loads fly under it.  Nothing between the layers touches HBM: at [64, 80, 900] the three
// kernels moved 23 + 37 | 37 + 18 | 18 + 18 MB, this one 23 + 18 MB.
// Waves: 2 * nwt.  Wave (wt, fh) owns output tiles N * wt .. of every layer and frame tiles fh * FT/2 .. (fh+1) * FT/2 - 1:
// the first version had one wave per channel group and all FT frame tiles -- five waves for the mel encoder, i.e. one SIMD
// with two of them while three ran one (layer 0: 11.7 k cycles alone, 17.7 k paired; the workgroup waits for the pair).
// With ten waves (3, 3, 2, 2 per SIMD) every layer's arithmetic and every epilogue is spread over all of them.
// Ring depth: layer 0 four steps; the k = 1 layers' steps are 12 or 24 MFMAs (200-400 cycles) -- shorter than the
// weights' trip from L2 --, so their rings hold eight steps: with nch <= 8 that is ALL of the layer's weights, in flight
// from the moment the previous layer's steps end.
struct FusedLayer { const uint4 *wp; const float *bias; int nch, cpad, Cout, relu; };
struct ConvFusedParams {
    const uint4 *xs; unsigned long long xs_plane; int S, nx;
    const float *xf; int Cin0;   // the first layer's input as fp32 [B, Cin0, T] instead of xs (see cn_stage_f32)
    FusedLayer L[3];
    ConvGemmParams out;       // the last layer's epilogue (bias, relu, Cout, T, y)
    int ldsB;                 // byte offset of the second LDS buffer
    unsigned long long *stamps;   // debug (nullable): [workgroup][wave][8] shader clock (aligner_debug_set_stamps)
};
#define CF_STAMP(k)                                                                                    \
    do {                                                                                               \
        if (p.stamps && (threadIdx.x & 63) == 0)                                                       \
            p.stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// the accumulators (output channels on the M axis) of wave (wt, frame tiles n0 .. n0+FTW-1) -> the next layer's LDS image
// (rows of LROWN slots); frames >= T and channels >= Cout are zeros, and so are the tiles of the image's last chunk that
// no wave owns
template <int NT, int FTW, int LROWN>
__device__ __forceinline__ void cg_store_lds(unsigned char *dst, int nch_next, const FusedLayer &L, int T, int f0, int n0,
                                             const cg_f32x4 (&acc)[NT][FTW], int wt, int nwt, int lane) {
    const int q = lane >> 4, j16 = lane & 15;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    auto slot_ptr = [&](int ob, int slot, int plane) {
        return dst + ((size_t)(((ob >> 5) * 8 + plane * 4 + ((ob & 31) >> 3)) * LROWN + slot)) * 16 + 8 * ((ob & 7) >> 2);
    };
    // (no branch in here: a wave's tiles always lie inside the next image -- fused_plan checks NT * nwt * 16 <= 32 * nch_next --,
    // the bias comes as clamped loads, masks are selects; the first version's per-element branches and four dependent bias
    // loads per tile made this epilogue 11 k cycles, more than the layer's arithmetic)
    const bool has_bias = L.bias != nullptr;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int ob = 16 * (NT * wt + a) + 4 * q;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_bias) {                                                      // (uniform)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ob + r;
                bv[r] = and_maskf(L.bias[o < L.Cout ? o : L.Cout - 1], o < L.Cout ? ~0u : 0u);
            }
        }
        const unsigned keep[4] = {ob + 0 < L.Cout ? ~0u : 0u, ob + 1 < L.Cout ? ~0u : 0u, ob + 2 < L.Cout ? ~0u : 0u,
                                  ob + 3 < L.Cout ? ~0u : 0u};
#pragma unroll
        for (int n = 0; n < FTW; ++n) {
            const int slot = 16 * (n0 + n) + j16;
            const unsigned fin = f0 + slot < T ? ~0u : 0u;
            bf16x4 hv, lv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[a][n][r] + bv[r];
                if (L.relu) v = fmaxf(v, 0.f);
                v = and_maskf(v, keep[r] & fin);
                __bf16 hh, ll;
                cg_split(v, hh, ll);
                hv[r] = hh;
                lv[r] = ll;
            }
            *reinterpret_cast<bf16x4 *>(slot_ptr(ob, slot, 0)) = hv;
            *reinterpret_cast<bf16x4 *>(slot_ptr(ob, slot, 1)) = lv;
        }
    }
    for (int tz = NT * nwt + wt; tz < 2 * nch_next; tz += nwt) {            // tiles nobody computed: the image's padding
        const int ob = 16 * tz + 4 * q;
        const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
#pragma unroll
        for (int n = 0; n < FTW; ++n) {
            *reinterpret_cast<bf16x4 *>(slot_ptr(ob, 16 * (n0 + n) + j16, 0)) = z;
            *reinterpret_cast<bf16x4 *>(slot_ptr(ob, 16 * (n0 + n) + j16, 1)) = z;
        }
    }
}

template <int KT0, int FT, int N0, int N1, int N2>
__global__ __launch_bounds__(768) void conv_narrow_fused_kernel(ConvFusedParams p) {   // <= 12 waves: 168 VGPRs (16 waves: 128, and the rings spill)
    constexpr int FTW = FT / 2;                          // frame tiles per wave
    constexpr int LROW0 = cg_lrow(KT0, FT), LROW1 = cg_lrow(1, FT);
    constexpr int RDA = N1 == 2 ? 4 : 8, RDB = N2 == 2 ? 4 : 8;   // ring depths of the k = 1 layers (32 registers either way)
    extern __shared__ __attribute__((aligned(16))) unsigned char cg_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)(blockDim.x >> 6), nwt = nw >> 1;
    const int wt = wave < nwt ? wave : wave - nwt, fh = wave < nwt ? 0 : 1;
    const int n0 = fh * FTW;
    const int bx = (int)(blockIdx.x % (unsigned)p.nx), b = (int)(blockIdx.x / (unsigned)p.nx);
    const int f0 = bx * 16 * FT, T = p.out.T;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)cg_smem;
    unsigned char *bufA = cg_smem, *bufB = cg_smem + p.ldsB;
    const unsigned xoff = (unsigned)n0 * 256u;           // this wave's first frame tile inside an image row
    CF_STAMP(0);
    if (p.xf) cn_stage_f32<KT0, FT>(p.xf, p.Cin0, p.out.T, p.L[0].nch, b, f0, tid, (int)blockDim.x, cg_smem);
    else      cn_stage_all<KT0, FT>(p.xs, p.xs_plane, p.S, p.L[0].nch, b, f0, wave, nw, lane, lds0);
    CnRing<N0, CN_RING> R0;
    R0.init(p.L[0].wp, p.L[0].cpad, p.L[0].nch * KT0, wt, lane, (int)blockIdx.x);
    R0.prime();
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CN_RING * 2 * N0) : "memory");
    __builtin_amdgcn_s_barrier();
    CF_STAMP(1);
    CnRing<N1, RDA> R1;
    R1.init(p.L[1].wp, p.L[1].cpad, p.L[1].nch, wt, lane, (int)blockIdx.x);
    {
        cg_f32x4 acc[N0][FTW];
        cn_steps<KT0, LROW0, FTW, N0, CN_RING, true>(R0, bufA + xoff, lane, acc);
        CF_STAMP(2);
        R1.prime();                                                          // the next layer's weights fly under the epilogue
        R0.template drain<RDA * 2 * N1>();
        cg_store_lds<N0, FTW, LROW1>(bufB, p.L[1].nch, p.L[0], T, f0, n0, acc, wt, nwt, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CF_STAMP(3);
    if (N2 == 0) {
        cg_f32x4 acc[N1][FTW];
        cn_steps<1, LROW1, FTW, N1, RDA, false>(R1, bufB + xoff, lane, acc);
        R1.template drain<0>();
        CF_STAMP(6);
        cg_store_f32<N1, FTW>(p.out, acc, b, N1 * wt, f0 + 16 * n0, lane);
        CF_STAMP(7);
    } else {
        constexpr int M2 = N2 > 0 ? N2 : 1;
        CnRing<M2, RDB> R2;
        R2.init(p.L[2].wp, p.L[2].cpad, p.L[2].nch, wt, lane, (int)blockIdx.x);
        {
            cg_f32x4 acc[N1][FTW];
            cn_steps<1, LROW1, FTW, N1, RDA, true>(R1, bufB + xoff, lane, acc);
            CF_STAMP(4);
            R2.prime();
            R1.template drain<RDB * 2 * M2>();
            cg_store_lds<N1, FTW, LROW1>(bufA, p.L[2].nch, p.L[1], T, f0, n0, acc, wt, nwt, lane);   // (layer 0's image is dead: everybody passed the barrier)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        CF_STAMP(5);
        cg_f32x4 acc[M2][FTW];
        cn_steps<1, LROW1, FTW, M2, RDB, false>(R2, bufA + xoff, lane, acc);
        R2.template drain<0>();
        CF_STAMP(6);
        cg_store_f32<M2, FTW>(p.out, acc, b, M2 * wt, f0 + 16 * n0, lane);
        CF_STAMP(7);
    }
}

template <int KT, int FT, bool SPLIT>
static int launch_conv_gemm(const ConvGemmParams &p, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 8 * cg_lrow(KT, FT) * 16;
    auto kern = conv_gemm_kernel<KT, FT, SPLIT>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    const unsigned grid = (unsigned)p.nx * (unsigned)(p.cpad / CG_TO) * (unsigned)p.B;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}
template <int KT, int FT, int NT, bool SPLIT>
static int launch_conv_narrow(const ConvGemmParams &p, int nw, hipStream_t s) {
    const size_t lds = (size_t)p.nch * 8 * cg_lrow(KT, FT) * 16;
    auto kern = conv_narrow_kernel<KT, FT, NT, SPLIT>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.nx * (unsigned)p.B), dim3(64 * nw), lds, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}
template <int FT, int NT, bool SPLIT>
static int launch_conv_narrow_ring(const ConvGemmParams &p, int nw, hipStream_t s) {
    constexpr size_t lds = (size_t)CNR_NB * CNR_G * 8 * cg_lrow(1, FT) * 16;
    auto kern = conv_narrow_ring_kernel<FT, NT, SPLIT>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.nx * (unsigned)p.B), dim3(64 * (nw + 1)), lds, s, p);   // + the loader wave
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}
template <int KT0, int FT, int N0, int N1, int N2>
static int launch_conv_fused(const ConvFusedParams &p, int nw, int B, size_t lds, hipStream_t s) {
    auto kern = conv_narrow_fused_kernel<KT0, FT, N0, N1, N2>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.nx * (unsigned)B), dim3(64 * 2 * nw), lds, s, p);   // two frame halves
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

// ---- host side: which form a layer takes, its prepared weights, its launch ----
enum { CG_NONE = 0, CG_WIDE = 1, CG_NARROW = 2 };
struct ConvPlan { int form, FT, NT, nw, nx, cpad; size_t lds; };

static int conv_form(int Cin, int Cout, int K) {
    if (K != 1 && K != 3 && K != 5) return CG_NONE;
    if (Cout >= 128 && Cin >= 128) return CG_WIDE;    // enough chunks to amortise the ring's prologue, full 128-channel tiles
    if (Cout <= 256 && Cin >= 16) return CG_NARROW;
    return CG_NONE;
}
static int conv_cpad(int form, int Cout) { return form == CG_WIDE ? (Cout + 127) / 128 * 128 : (Cout + 31) / 32 * 32; }

static ConvPlan conv_plan(int B, int Cin, int Cout, int T, int K) {
    ConvPlan P{conv_form(Cin, Cout, K), 0, 0, 0, 0, 0, 0};
    if (P.form == CG_NONE) return P;
    P.cpad = conv_cpad(P.form, Cout);
    const int nch = (Cin + CG_CH - 1) / CG_CH;
    if (P.form == CG_WIDE) {
        // frames per workgroup: one utterance's whole T when it fits 13 tiles of 16 (T = 200 -> 208), else the tile count
        // that wastes least
        P.FT = 13;
        if (T > 16 * 13) {
            const int w13 = (T + 207) / 208 * 208, w8 = (T + 127) / 128 * 128;
            P.FT = w8 < w13 ? 8 : 13;
        } else if (T <= 128) {
            P.FT = 8;
        }
        P.NT = 2; P.nw = 4;
        P.lds = (size_t)2 * 8 * cg_lrow(K, P.FT) * 16;
    } else {
        // output tiles per wave: 32 channels from 144 output channels on (at most 8 waves); frames per workgroup: the
        // largest tile that still gives the chip two workgroups per CU and whose activations fit in LDS
        P.NT = P.cpad > 128 ? 2 : 1;
        P.nw = P.cpad / (16 * P.NT);
        const size_t lds_max = (size_t)device_lds_limit();
        const int cus = device_cu_count();
        // Every workgroup streams ALL the layer's weights through its waves' registers, so frames per workgroup are
        // weight reuse: the largest tile that fits in LDS and still gives the chip two workgroups per CU; a batch too small
        // for that takes the smallest tile (most workgroups).  Measured at [64, ., 900]: 80 -> 80 k=1 13.2 us with 8-tile
        // workgroups, 18.0 us with 2-tile ones.
        for (int ft : {8, 4, 2}) {
            if (g_opt_conv_narrow_ft && ft != g_opt_conv_narrow_ft) continue;   // testing: aligner_debug_set_option
            const size_t lds = (size_t)nch * 8 * cg_lrow(K, ft) * 16;
            if (lds > lds_max) continue;
            P.FT = ft;
            P.lds = lds;
            if ((long long)((T + 16 * ft - 1) / (16 * ft)) * B >= 2LL * cus) break;
        }
        if (P.FT == 0) { P.form = CG_NONE; return P; }                       // too many input channels for one LDS
    }
    P.nx = (T + 16 * P.FT - 1) / (16 * P.FT);
    return P;
}

static int conv_launch(const ConvPlan &P, ConvGemmParams &p, int K, bool split, hipStream_t s) {
    p.nx = P.nx;
    p.cpad = P.cpad;
    if ((unsigned long long)p.nx * (p.cpad / 16) * p.B >= (1ull << 31)) return fail(ALIGNER_EDOM, "grid too large");
#define CG_W(KT, FT) (split ? launch_conv_gemm<KT, FT, true>(p, s) : launch_conv_gemm<KT, FT, false>(p, s))
#define CG_N(KT, FT, NT) (split ? launch_conv_narrow<KT, FT, NT, true>(p, P.nw, s) : launch_conv_narrow<KT, FT, NT, false>(p, P.nw, s))
#define CG_NF(KT, NT) (P.FT == 8 ? CG_N(KT, 8, NT) : P.FT == 4 ? CG_N(KT, 4, NT) : CG_N(KT, 2, NT))
#define CG_NK(NT) (K == 1 ? CG_NF(1, NT) : K == 3 ? CG_NF(3, NT) : CG_NF(5, NT))
    if (P.form == CG_WIDE) {
        if (K == 1) return P.FT == 13 ? CG_W(1, 13) : CG_W(1, 8);
        if (K == 3) return P.FT == 13 ? CG_W(3, 13) : CG_W(3, 8);
        return P.FT == 13 ? CG_W(5, 13) : CG_W(5, 8);
    }
    // many chunks of a split image into a k = 1 layer: the group-ring form (conv_narrow_ring_kernel)
    if (K == 1 && P.FT == 2 && !p.xf && p.nch >= 4 * CNR_G && p.nch % CNR_G == 0 && P.nw <= 8 && !g_opt_conv_no_ring) {
        // (fp32 epilogue: a wave whose 16 or 32 channels are all padding has nothing to store -- and no weights to stream:
        // 1024 -> 80 with five compute waves 17.0 us, with the sixth 18.1.  64-frame tiles, one workgroup a CU: 21.8 us)
        const int nw = split ? P.nw : (p.Cout + 16 * P.NT - 1) / (16 * P.NT);
        if (P.NT == 2) return split ? launch_conv_narrow_ring<2, 2, true>(p, nw, s) : launch_conv_narrow_ring<2, 2, false>(p, nw, s);
        return split ? launch_conv_narrow_ring<2, 1, true>(p, nw, s) : launch_conv_narrow_ring<2, 1, false>(p, nw, s);
    }
    return P.NT == 2 ? CG_NK(2) : CG_NK(1);
#undef CG_W
#undef CG_N
#undef CG_NF
#undef CG_NK
}

// ---- what softattn.hip's entry points call ----
bool conv_gemm_applies(int Cin, int Cout, int K) { return conv_form(Cin, Cout, K) != CG_NONE; }

size_t conv_gemm_prepared_bytes(int Cout, int Cin, int K) {
    const int form = conv_form(Cin, Cout, K);
    if (form == CG_NONE) return 0;
    const int nch = (Cin + CG_CH - 1) / CG_CH;
    return (size_t)nch * K * (conv_cpad(form, Cout) / 16) * 2 * 64 * sizeof(uint4);
}

int conv_gemm_prepare(const float *w, void *prepared, int Cout, int Cin, int K, hipStream_t s) {
    const int form = conv_form(Cin, Cout, K);
    const int cpad = conv_cpad(form, Cout), nch = (Cin + CG_CH - 1) / CG_CH;
    const int nfrag = nch * K * (cpad / 16) * 64;
    hipLaunchKernelGGL(conv_gemm_wprep_kernel, dim3((nfrag + 255) / 256), dim3(256), 0, s, w, static_cast<uint4 *>(prepared),
                       Cout, Cin, K, cpad, nfrag);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

// A stack of layers y = act(conv(...act(conv(x)))) on split activations.  Workspace: two split images that take turns
// (each sized for the largest one of the stack) + an fp32 temporary for a layer whose consumer is not a k = 1 layer
// of one of these forms.  Returns the index of the first layer the stack runner cannot take (== n: all of them).
struct ConvStackLayer { const void *prepared; const float *bias; int Cin, Cout, K, relu; };

static size_t split_image_bytes(int B, int Cin, int T, int K) {
    const size_t nch = (Cin + CG_CH - 1) / CG_CH, S = (T + 2 * (K / 2) + 15) / 16 * 16;
    return align_up((2 * (size_t)B * nch * 4 * S + 512) * sizeof(uint4), 256);
}

// A run of narrow layers at the END of a stack as one kernel (conv_narrow_fused_kernel): how many layers (0, 2 or 3),
// with the launch's numbers.
struct FusedPlan { int count, FT, nw, N[3], nx; size_t lds, ldsB; };
static FusedPlan fused_plan(const ConvStackLayer *L, int n, int B, int T) {
    FusedPlan F{};
    if (g_opt_conv_no_fuse) return F;
    for (int cnt : {3, 2}) {
        if (cnt > n) continue;
        const ConvStackLayer *G = L + (n - cnt);
        bool ok = conv_form(G[0].Cin, G[0].Cout, G[0].K) == CG_NARROW;
        for (int j = 1; j < cnt && ok; ++j) ok = G[j].K == 1 && conv_form(G[j].Cin, G[j].Cout, 1) == CG_NARROW;
        if (!ok) continue;
        const int cpad0 = conv_cpad(CG_NARROW, G[0].Cout);
        F.N[0] = cpad0 > 128 ? 2 : 1;
        F.nw = cpad0 / (16 * F.N[0]);
        F.N[2] = 0;
        if (F.nw > 6) continue;                                               // two wave groups of at most six waves (register budget)
        for (int j = 1; j < cnt && ok; ++j) {
            const int tiles = (G[j].Cout + 15) / 16;
            F.N[j] = (tiles + F.nw - 1) / F.nw;
            ok = F.N[j] <= 2 && F.N[j] * F.nw * 16 <= conv_cpad(CG_NARROW, G[j].Cout);   // every wave's tiles exist in the prepared image
            // ... and the producer's tiles lie inside this layer's input image (cg_store_lds has no range check)
            ok = ok && F.N[j - 1] * F.nw * 16 <= 32 * ((G[j].Cin + CG_CH - 1) / CG_CH);
        }
        const int key = F.N[0] * 100 + F.N[1] * 10 + F.N[2];
        ok = ok && (key == 110 || key == 111 || key == 210 || key == 211 || key == 220);   // the instantiated forms
        if (!ok) continue;
        const size_t lds_max = (size_t)device_lds_limit();
        for (int ft : {8, 4}) {
            if (g_opt_conv_narrow_ft && ft != g_opt_conv_narrow_ft) continue;           // testing: aligner_debug_set_option
            const size_t nch0 = (G[0].Cin + CG_CH - 1) / CG_CH, nch1 = (G[1].Cin + CG_CH - 1) / CG_CH;
            size_t a = nch0 * 8 * cg_lrow(G[0].K, ft) * 16;
            if (cnt == 3) {
                const size_t a2 = (size_t)((G[2].Cin + CG_CH - 1) / CG_CH) * 8 * cg_lrow(1, ft) * 16;
                a = a2 > a ? a2 : a;
            }
            const size_t bsz = nch1 * 8 * cg_lrow(1, ft) * 16;
            if (a + bsz > lds_max) continue;
            F.count = cnt; F.FT = ft; F.ldsB = a; F.lds = a + bsz;
            F.nx = (T + 16 * ft - 1) / (16 * ft);
            return F;
        }
    }
    return FusedPlan{};
}

size_t conv_stack_workspace_bytes(const ConvStackLayer *L, int n, int B, int T) {
    size_t img = 0, tmp = 0;
    for (int i = 0; i < n; ++i) {
        if (conv_plan(B, L[i].Cin, L[i].Cout, T, L[i].K).form == CG_NONE) return 0;
        const size_t v = split_image_bytes(B, L[i].Cin, T, L[i].K);
        img = v > img ? v : img;
        if (i + 1 < n && L[i + 1].K != 1) {
            const size_t t = align_up((size_t)B * L[i].Cout * T * sizeof(float), 256);
            tmp = t > tmp ? t : tmp;
        }
    }
    return 2 * img + tmp;
}

int conv_stack_run(const float *x, const ConvStackLayer *L, int n, float *y, void *workspace, size_t workspace_bytes, int B, int T,
                   hipStream_t s) {
    const size_t need = conv_stack_workspace_bytes(L, n, B, T);
    if (need == 0) return fail(ALIGNER_EDOM, "a layer of this stack has no GEMM form");
    if (workspace_bytes < need) return fail(ALIGNER_ENOSPC, "conv workspace %zu < %zu bytes", workspace_bytes, need);
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(ALIGNER_EINVAL, "conv workspace must be 16-byte aligned");
    size_t img = 0;
    for (int i = 0; i < n; ++i) {
        const size_t v = split_image_bytes(B, L[i].Cin, T, L[i].K);
        img = v > img ? v : img;
    }
    if (img >= (1ull << 32)) return fail(ALIGNER_EDOM, "split activations of %zu bytes exceed 32-bit offsets", img);
    unsigned char *wsb = static_cast<unsigned char *>(workspace);
    uint4 *bufs[2] = {reinterpret_cast<uint4 *>(wsb), reinterpret_cast<uint4 *>(wsb + img)};
    float *tmp = reinterpret_cast<float *>(wsb + 2 * img);
    auto split_pass = [&](const float *src, uint4 *dst, int Cin, int K) -> int {
        const size_t nch = (Cin + CG_CH - 1) / CG_CH, S = (T + 2 * (K / 2) + 15) / 16 * 16;
        const size_t plane = (size_t)B * nch * 4 * S;
        hipLaunchKernelGGL(conv_split_kernel, dim3((unsigned)((plane + 256 * CS_U - 1) / (256 * CS_U))), dim3(256), 0, s, src, dst, plane,
                           Cin, T, (int)S, (int)nch, K / 2, plane);
        ALIGNER_HIP_CHECK(hipGetLastError());
        return ALIGNER_OK;
    };
    int cur = 0;
    int rc = ALIGNER_OK;
    // the current activations are either fp32 (`f32`: the caller's x, or a layer's fp32 output) or a split image in
    // bufs[cur]; a narrow layer stages fp32 itself (cn_stage_f32), a wide one needs the split pass first
    const float *f32 = x;
    const bool stage_f32 = !g_opt_conv_split_always;
    const FusedPlan F = fused_plan(L, n, B, T);
    for (int i = 0; i < n; ++i) {
        const bool fused_here = F.count && i == n - F.count;
        const bool narrow_here = fused_here || conv_plan(B, L[i].Cin, L[i].Cout, T, L[i].K).form == CG_NARROW;
        if (f32 && !(narrow_here && stage_f32)) {
            rc = split_pass(f32, bufs[cur], L[i].Cin, L[i].K);               // (a layer that read bufs[cur] is behind us: stream order)
            if (rc != ALIGNER_OK) return rc;
            f32 = nullptr;
        }
        if (fused_here) {                                                    // the stack's last layers: one kernel, tiles kept on the CU
            const ConvStackLayer *G = L + i;
            ConvFusedParams fp{};
            const int nch0 = (G[0].Cin + CG_CH - 1) / CG_CH, S0 = (T + 2 * (G[0].K / 2) + 15) / 16 * 16;
            fp.stamps = g_debug_stamps;
            fp.xf = f32; fp.Cin0 = G[0].Cin;
            fp.xs = bufs[cur]; fp.xs_plane = (unsigned long long)B * nch0 * 4 * S0; fp.S = S0; fp.nx = F.nx; fp.ldsB = (int)F.ldsB;
            for (int j = 0; j < F.count; ++j)
                fp.L[j] = FusedLayer{static_cast<const uint4 *>(G[j].prepared), G[j].bias, (G[j].Cin + CG_CH - 1) / CG_CH,
                                     conv_cpad(CG_NARROW, G[j].Cout), G[j].Cout, G[j].relu};
            const ConvStackLayer &Z = G[F.count - 1];
            fp.out.bias = Z.bias; fp.out.y = y; fp.out.B = B; fp.out.Cout = Z.Cout; fp.out.T = T; fp.out.relu = Z.relu;
            const int key = F.N[0] * 100 + F.N[1] * 10 + F.N[2], K0 = G[0].K;
#define CF(KT0, FTV, A, Bn, C) launch_conv_fused<KT0, FTV, A, Bn, C>(fp, F.nw, B, F.lds, s)
#define CF_FT(KT0, A, Bn, C) (F.FT == 8 ? CF(KT0, 8, A, Bn, C) : CF(KT0, 4, A, Bn, C))
#define CF_K(A, Bn, C) (K0 == 1 ? CF_FT(1, A, Bn, C) : K0 == 3 ? CF_FT(3, A, Bn, C) : CF_FT(5, A, Bn, C))
            rc = key == 110 ? CF_K(1, 1, 0) : key == 111 ? CF_K(1, 1, 1) : key == 210 ? CF_K(2, 1, 0) : key == 211 ? CF_K(2, 1, 1)
                                                                                                       : CF_K(2, 2, 0);
#undef CF
#undef CF_FT
#undef CF_K
            return rc;
        }
        const ConvPlan P = conv_plan(B, L[i].Cin, L[i].Cout, T, L[i].K);
        const int nch = (L[i].Cin + CG_CH - 1) / CG_CH, S = (T + 2 * (L[i].K / 2) + 15) / 16 * 16;
        const bool last = i + 1 == n;
        const bool split = !last && L[i + 1].K == 1;                         // write the consumer's image directly
        ConvGemmParams p{};
        p.xf = f32; p.Cin = L[i].Cin;
        f32 = nullptr;
        p.xs = bufs[cur]; p.wp = static_cast<const uint4 *>(L[i].prepared); p.bias = L[i].bias;
        p.xs_plane = (unsigned long long)B * nch * 4 * S;
        p.B = B; p.Cout = L[i].Cout; p.T = T; p.S = S; p.nch = nch; p.relu = L[i].relu;
        if (split) {
            p.nch2 = (L[i].Cout + CG_CH - 1) / CG_CH;
            p.S2 = (T + 15) / 16 * 16;
            p.ys = bufs[cur ^ 1];
            p.ys_plane = (unsigned long long)B * p.nch2 * 4 * p.S2;
        } else {
            p.y = last ? y : tmp;
        }
        rc = conv_launch(P, p, L[i].K, split, s);
        if (rc != ALIGNER_OK) return rc;
        if (split) cur ^= 1;
        else if (!last) f32 = tmp;
    }
    return ALIGNER_OK;
}

size_t conv_gemm_workspace_bytes(int B, int Cin, int Cout, int T, int K) {
    ConvStackLayer l{nullptr, nullptr, Cin, Cout, K, 0};
    return conv_stack_workspace_bytes(&l, 1, B, T);
}

int conv_gemm_run(const float *x, const void *prepared, const float *bias, float *y, void *workspace, size_t workspace_bytes,
                  int B, int Cin, int Cout, int T, int K, int relu, hipStream_t s) {
    ConvStackLayer l{prepared, bias, Cin, Cout, K, relu};
    return conv_stack_run(x, &l, 1, y, workspace, workspace_bytes, B, T, s);
}

}  // namespace aligner
