// Forward-sum alignment objective on MI355X (gfx950): the log-likelihood of ALL monotonic alignments
// and its gradient (the OTA objective the reference's README.md:21-25,50 points at --
// not in the snapshot, so the spec is build-defined: DESIGN.md 5 / DESIGN_HISTORY.md 7, oracle/forward_sum_oracle.py).
//
//   alpha[x,y] = logaddexp(alpha[x,y-1], alpha[x-1,y-1]) + logp[x,y]      alpha[0,0] = logp[0,0]
//   beta [x,y] = logaddexp(beta[x,y+1] + logp[x,y+1], beta[x+1,y+1] + logp[x+1,y+1])   beta[tx-1,ty-1] = 0
//   loss       = -alpha[tx-1,ty-1]             d loss / d logp[x,y] = -exp(alpha + beta - log Z)
//
// i.e. the column recurrence of maximum_path_each (reference core.pyx:17-30) with log-sum-exp for max.
// Same shape of computation as the DP: a mel frame only couples to the previous one, text rows are
// independent within a frame -- and the same remedy for the 1000-step dependent chain: no workgroup
// barrier inside it.  One workgroup per utterance; ONE wave sweeps the frames, each lane owning R
// consecutive text rows (R = 4, 8, 16 for T_text <= 256, 512, 1024), so the neighbour row is the
// lane's own register except for one DPP wave shift per frame, and the R independent cells per lane
// cover each other's transcendental latency.  The other three waves only move data: the [Tx,Ty]
// operands are row-major with the mel axis contiguous, so tiles of TW frames are staged through LDS
// (coalesced row segments from/to HBM, frame-major in LDS so that a lane's R rows of one frame are one
// ds_read_b128 per four rows), double-buffered, one barrier per tile.
//
// Everything inside the kernels is in BASE-2 logs (the stagers scale the log-probs by log2 e on their
// way into LDS): v_exp_f32 / v_log_f32 are exp2 / log2, so logaddexp is sub, exp2, add, log2, add, max --
// six issues -- and "minus infinity" is the finite FS_NEG, which absorbs every addend (no NaN from
// inf - inf, no select).
//
// Numerics: fp32 in log space cannot carry alpha ~ -5000 over 1000 frames and still resolve the
// posterior (an error of 1e-2 in the exponent is 1 % of the value; rounding at magnitude 50 is 2e-6 per
// add and 2000 adds walk 1e-4 away).  The running column is therefore kept near 0: every frame a
// uniform drift estimate is subtracted, every FS_RB frames the column maximum (a wave reduction), and
// both go into a double offset (C forward, one value per frame in the workspace; D backward).  The
// posterior is exp(alpha_hat[x,y] + beta_hat[x,y] + float(C_y + D_y - log Z)).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

constexpr float FS_NEG_INF = -__builtin_huge_valf();
constexpr float FS_NEG = -1e30f;                  // "log 0" inside the kernels: absorbs any finite addend
constexpr float FS_LOG2E = 1.4426950408889634f;
constexpr double FS_LN2 = 0.6931471805599453;
constexpr int FS_RB = 8;                          // frames between re-basings (divides every tile width)
constexpr int FS_THREADS = 256;                   // wave 0 sweeps, waves 1..3 stage tiles
constexpr int FS_STAGERS = FS_THREADS - 64;
constexpr int SY_NW_MAX = 8;                      // ... offsets in the workspace: one per (wave, frame)

struct FwdSumParams {
    const float *logp;      // [B,Tx,Ty]
    const int   *t_xs, *t_ys;
    float  *alpha;          // workspace [B,Tx,Ty]: log2 alpha of frame y relative to offs[y]
    double *offs;           // workspace [B,NT = Ty]: C_y (base 2)
    double *logz;           // workspace [B]: log2 Z
    float  *loss;           // [B]
    float  *grad;           // [B,Tx,Ty] (backward only)
    int B, Tx, Ty, NT;
    double *doffs;          // workspace [B,SY_NW_MAX,NT]: D_w per (wave, frame) -- the sweeps-side-by-side form only
    unsigned long long *stamps;   // development (aligner_debug_set_stamps): per (workgroup, wave) 8 words of phase cycle totals
    int no_grad_stager;     // A-B / testing ("fwdsum_no_grad_stager"): the gradient-making backward kernel keeps its compiler-scheduled stager
};

// log2(2^a + 2^b).  The log term is in (0, 1]: its absolute error (~1 ulp of the hardware log2/exp2)
// is below the rounding of the sum itself.  FS_NEG operands: a - b = 0 -> m + 1 = FS_NEG (absorbed).
__device__ __forceinline__ float fs_lae2(float a, float b) {
    float m;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));      // (fmaxf would first canonicalise both operands)
    return m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a - b)));
}

// a log-prob on its way into LDS: base 2, and never below FS_NEG (a -inf score would meet FS_NEG as NaN)
__device__ __forceinline__ float fs_in(float lp) { return fmaxf(lp * FS_LOG2E, FS_NEG); }
// the systolic kernels' stagers leave the scaling to the sweeper (an fma where it had a subtraction: free there, while a
// stager's VALU issues land on the SIMD its sweeper's dependent chain runs on): natural logs in LDS, never below
// FS_NEG_NAT (x log2 e = FS_NEG), rows past the text exactly that
constexpr float FS_NEG_NAT = -6.9314718e29f;
__device__ __forceinline__ float fs_nat(float lp) { return fmaxf(lp, FS_NEG_NAT); }

// workgroup barrier for LDS traffic only: __syncthreads() would also wait for the stagers' global
// stores (vmcnt(0)), a memory round trip per tile
__device__ __forceinline__ void fs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float fs_wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// the same maximum without LDS traffic (__shfl_xor is a ds_bpermute: ~100 cycles a step): the usual DPP
// ladder, result in lane 63, broadcast through an SGPR
__device__ __forceinline__ float fs_wave_max_dpp(float v) {
#define FS_DPP_MAX(CTRL, RMASK)                                                                          \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v),       \
                                                                        __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false)))
    FS_DPP_MAX(0xB1, 0xf);      // quad_perm [1,0,3,2]
    FS_DPP_MAX(0x4E, 0xf);      // quad_perm [2,3,0,1]
    FS_DPP_MAX(0x141, 0xf);     // row_half_mirror
    FS_DPP_MAX(0x140, 0xf);     // row_mirror
    FS_DPP_MAX(0x142, 0xa);     // row_bcast:15 into rows 1 and 3
    FS_DPP_MAX(0x143, 0xc);     // row_bcast:31 into rows 2 and 3
#undef FS_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// lane i <- src[lane i-1]; lane 0 gets `edge`          (wave_shr:1, bound_ctrl off keeps `old`)
__device__ __forceinline__ float fs_from_lane_below(float edge, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}
// lane i <- src[lane i+1]; lane 63 gets `edge`         (wave_shl:1)
__device__ __forceinline__ float fs_from_lane_above(float edge, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, src), 0x130, 0xf, 0xf, false));
}

// The stagers' global accesses as raw buffer operations over the utterance's [Tx,Ty] block: the per-lane byte offset of
// an element (row * Ty + frame-in-tile) never changes, the tile's first frame goes into the SCALAR offset -- no address
// arithmetic per element and tile.  (Computed per element, it was ~400 VALU issues a phase on the SIMD the sweeper's
// dependent chain runs on: the phase took 3 300 cycles where the chain needs 1 400.)  Elements that must not be written
// carry an offset beyond the block: the hardware drops the store.
constexpr unsigned FS_DROP = 0xFFFFFF00u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fs_rsrc(const void *base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (unsigned)(bytes < 0xFFFFFF00ull ? bytes : 0xFFFFFF00ull),
                                             0x00020000);
}
__device__ __forceinline__ float fs_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void fs_bstore(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// ---- a stager wave of the systolic kernels, with every global access issued by hand ----
// Measured on the way here (fwdsum at [64,200,1000], 68 phases of ~3 300 cycles): neither the sweepers' instruction
// count (halved: no change), nor the stagers' address arithmetic (removed: no change), nor their stores (removed: no
// change) set the phase -- the compiler's waits did: written as loads into arrays, every tile was waited for with
// vmcnt(15..0), which also waits for every NEWER load, so however many tiles were "in flight" a phase was one memory
// round trip (~1.4 us through 400 TLB-unfriendly rows).  Here a tile is four (TW = 16) or two 16-byte loads per lane,
// FS_DEPTH tiles are in flight in FS_DEPTH register sets, and the wait is counted: the order inside a phase is
// wait - LDS write - stores - loads, so behind a tile's loads there are at least 4 (D - 1) newer operations (the later
// tiles' loads; stores only add to that) and vmcnt(4 (D - 1)) is always enough, whatever the stores do.
// Requires T_mel % 4 == 0 and 16-byte aligned tensors (else the callers keep the compiler-scheduled stager).
typedef unsigned fs_u32x4 __attribute__((ext_vector_type(4)));
constexpr int FS_DEPTH = 4;
__device__ __forceinline__ void fs_aload4(fs_u32x4 &dst, unsigned voff, const void *sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// (exec = mask for the store only; s_nop: nothing may write the data registers while the store still reads them)
__device__ __forceinline__ void fs_astore4(unsigned voff, fs_u32x4 data, void *sbase, unsigned long long mask) {
    unsigned long long keep;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\tglobal_store_dwordx4 %1, %2, %3\n\ts_mov_b64 exec, %0\n\ts_nop 1"
                 : "=&s"(keep) : "v"(voff), "v"(data), "s"(sbase), "s"(mask) : "memory");
}
template <int NJ, int CNT> struct FsWait;
template <int CNT> struct FsWait<4, CNT> {
    static __device__ __forceinline__ void on(fs_u32x4 (&q)[4]) {
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(CNT) : "memory");
    }
};
template <int CNT> struct FsWait<2, CNT> {
    static __device__ __forceinline__ void on(fs_u32x4 (&q)[2]) {
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q[0]), "+v"(q[1]) : "n"(CNT) : "memory");
    }
};

// in_g / out_g: the utterance's [Tx,Ty] blocks (log-probs in, alpha or beta out); offs_g: this wave's per-frame offsets in
// the workspace; tin_w / tout_w: this wave's two LDS tiles in / out; toff_w: its two rows of per-frame offsets in LDS.
// Slot r of the tile is text row 63 w + r - 1 (forward: slot 0 is the ghost of the wave above) or 63 w + r (backward:
// slot 63 is the ghost of the wave below); `lag`: the phase this wave's first tile is due in.
template <int SY_TW, bool BACKWARD>
__device__ __forceinline__ void fs_stager_by_hand(const float *in_g, float *out_g, double *offs_g, float *tin_w, const float *tout_w,
                                                  const double *toff_w, int lag, int w, int lane, int tx, int ty, int Tx, int Ty,
                                                  int ntl, int nph, unsigned long long *st = nullptr) {
    constexpr int PITCH = SY_TW + 4, SY_TILE = 64 * PITCH, QR = SY_TW / 4, NJ = SY_TW / 4, D = FS_DEPTH;
    fs_u32x4 q[D][NJ];
    unsigned vo[NJ], vt[NJ], so[NJ];
    unsigned long long mfull[NJ], mtail[NJ];
    int rr[NJ], qd[NJ];
    bool rowok[NJ];
    const int tailq = (Ty - (ntl - 1) * SY_TW) / 4;          // quads of the last tile that lie inside the tensor (>= QR: all)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e4 = lane + 64 * j;
        rr[j] = e4 / QR;
        qd[j] = e4 - rr[j] * QR;
        const int rs = BACKWARD ? 63 * w + rr[j] : 63 * w + rr[j] - 1;
        const int rg = rs < 0 ? 0 : (rs < tx ? rs : tx - 1);
        const int qc = qd[j] < tailq ? qd[j] : tailq - 1;    // the last tile: quads past the tensor re-read the last one inside
        vo[j] = (unsigned)(((size_t)rg * Ty + 4 * qd[j]) * sizeof(float));
        vt[j] = (unsigned)(((size_t)rg * Ty + 4 * qc) * sizeof(float));
        const bool sv = (BACKWARD ? rr[j] < 63 : rr[j] >= 1) && rs < Tx && rs >= 0;
        so[j] = (unsigned)(((size_t)(rs < 0 ? 0 : rs) * Ty + 4 * qd[j]) * sizeof(float));
        mfull[j] = __builtin_amdgcn_ballot_w64(sv);
        mtail[j] = __builtin_amdgcn_ballot_w64(sv && qd[j] < tailq);
        rowok[j] = rs < tx;                                   // rows past the text are staged as log 0
    }
    const bool allok = (BACKWARD ? 63 * w + 63 : 63 * w + 62) < tx;           // (uniform: most waves select nothing)
    auto tile_of = [&](int k) { return BACKWARD ? ntl - 1 - k : k; };          // k-th tile in sweep order
    auto issue = [&](int k, fs_u32x4 (&s)[NJ]) {
        const int t = tile_of(k < ntl ? k : ntl - 1);
        const char *base = reinterpret_cast<const char *>(in_g) + (size_t)t * SY_TW * sizeof(float);
        const bool last = t == ntl - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fs_aload4(s[j], last ? vt[j] : vo[j], base);
    };
    unsigned long long sa[4] = {0, 0, 0, 0};
    auto phase = [&](int ph, fs_u32x4 (&s)[NJ]) {
        const int k = ph - lag, ks = ph - 2 - lag;
        const unsigned long long c0 = st ? __builtin_amdgcn_s_memtime() : 0;
        FsWait<NJ, NJ * (D - 1)>::on(s);                                       // tile k has landed (see above)
        const unsigned long long c1 = st ? __builtin_amdgcn_s_memtime() : 0;
        if (k < ntl) {
            float *dst = tin_w + (k & 1) * SY_TILE;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float f[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned raw = s[j][jj];        // (bit_cast straight from the vector element reads element 0)
                    f[jj] = (allok || rowok[j]) ? fs_nat(__builtin_bit_cast(float, raw)) : FS_NEG_NAT;
                }
                *reinterpret_cast<float4 *>(dst + rr[j] * PITCH + 4 * qd[j]) = make_float4(f[0], f[1], f[2], f[3]);   // one ds_write_b128
            }
        }
        if (ks >= 0 && ks < ntl) {
            const float *src = tout_w + (ks & 1) * SY_TILE;
            const int t = tile_of(ks), y0 = t * SY_TW;
            char *base = reinterpret_cast<char *>(out_g) + (size_t)y0 * sizeof(float);
            const bool last = t == ntl - 1;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 f = *reinterpret_cast<const float4 *>(src + rr[j] * PITCH + 4 * qd[j]);       // one ds_read_b128
                fs_u32x4 d;
                d[0] = __builtin_bit_cast(unsigned, f.x); d[1] = __builtin_bit_cast(unsigned, f.y);
                d[2] = __builtin_bit_cast(unsigned, f.z); d[3] = __builtin_bit_cast(unsigned, f.w);
                fs_astore4(so[j], d, base, last ? mtail[j] : mfull[j]);
            }
            if (lane < SY_TW && y0 + lane < ty) offs_g[y0 + lane] = toff_w[(ks & 1) * SY_TW + lane];
        }
        unsigned long long c2 = 0;
        if (st) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c2 = __builtin_amdgcn_s_memtime(); }
        issue(k + D, s);                                                       // (behind the stores: the wait's count)
        const unsigned long long c3 = st ? __builtin_amdgcn_s_memtime() : 0;
        fs_lds_barrier();
        if (st) { sa[0] += c1 - c0; sa[1] += c2 - c1; sa[2] += c3 - c2; sa[3] += __builtin_amdgcn_s_memtime() - c3; }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, q[d]);
    for (int ph = 0; ph < lag; ++ph) fs_lds_barrier();                         // (this stager's first tile is due in phase `lag`)
    for (int ph = lag; ph < nph; ph += D) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (ph + d < nph) phase(ph + d, q[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) FsWait<NJ, 0>::on(q[d]);                       // the loads nobody consumed: every set named
    if (st && lane == 0) { st[0] = sa[0]; st[1] = sa[1]; st[2] = sa[2]; st[3] = sa[3]; st[4] = (unsigned long long)(nph - lag); }
}

// ---- the gradient-making backward kernel's stager (batches past half the CU count: forward, then backward) ----
// A tile here is three things: the log-probs, ALPHA (same rows and frames: same per-lane offsets, another base) and the
// wave's forward offsets C_w of the tile's frames (SY_TW doubles: one 8-byte load, lanes < SY_TW).  2 NJ + 1 loads a tile,
// all hand-issued, FS_GDEPTH tiles in flight in FS_GDEPTH register sets; the order inside a phase is again wait - LDS
// writes - stores - loads, so behind a tile's loads there are at least (2 NJ + 1)(D - 1) newer operations and
// vmcnt((2 NJ + 1)(D - 1)) is always enough.  EVERY register of a set is named by the wait that retires it ("+v"): a
// destination the compiler believes dead gets reassigned or spilled while the load is still landing (DESIGN.md rule 1;
// round 4's first version of this stager faulted on its first run -- it carried four sets beside a kernel that already
// spilled).  Three sets: the kernel stays inside its 256 registers without a spill
// (tests/test_abi.py checks the compiler's resource report).
constexpr int FS_GDEPTH = 3;
__device__ __forceinline__ void fs_aload2(unsigned long long &dst, unsigned voff, const void *sbase) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <int NJ, int CNT> struct FsWaitG;
template <int CNT> struct FsWaitG<4, CNT> {
    static __device__ __forceinline__ void on(fs_u32x4 (&a)[4], fs_u32x4 (&b)[4], unsigned long long &o) {
        asm volatile("s_waitcnt vmcnt(%9)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]),
                     "+v"(b[3]), "+v"(o) : "n"(CNT) : "memory");
    }
};
template <int CNT> struct FsWaitG<2, CNT> {
    static __device__ __forceinline__ void on(fs_u32x4 (&a)[2], fs_u32x4 (&b)[2], unsigned long long &o) {
        asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(o) : "n"(CNT) : "memory");
    }
};

// lp_g / al_g / gr_g: the utterance's [Tx,Ty] blocks (log-probs and alpha in, gradient out); offs_g: this wave's forward
// offsets per frame; tlp_w / tal_w / tgr_w: this wave's two LDS tiles each; toff_w: its two rows of offsets in LDS.
// Slot r of a tile is text row 63 w + r (slot 63 is the ghost of the wave below); tiles run from the last to the first.
template <int SY_TW>
__device__ __forceinline__ void fs_grad_stager_by_hand(const float *lp_g, const float *al_g, float *gr_g, const double *offs_g,
                                                       float *tlp_w, float *tal_w, const float *tgr_w, double *toff_w, int lag,
                                                       int w, int lane, int tx, int ty, int Tx, int Ty, int ntl, int nph) {
    constexpr int PITCH = SY_TW + 4, SY_TILE = 64 * PITCH, QR = SY_TW / 4, NJ = SY_TW / 4, D = FS_GDEPTH, OPS = 2 * NJ + 1;
    static_assert(OPS * (D - 1) <= 63, "the counted wait's field");
    fs_u32x4 ql[D][NJ], qa[D][NJ];
    unsigned long long qo[D];
    unsigned vo[NJ], vt[NJ], so[NJ];
    unsigned long long mfull[NJ], mtail[NJ];
    int rr[NJ], qd[NJ];
    bool rowok[NJ];
    const int tailq = (Ty - (ntl - 1) * SY_TW) / 4;          // quads of the last tile that lie inside the tensor (>= QR: all)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int e4 = lane + 64 * j;
        rr[j] = e4 / QR;
        qd[j] = e4 - rr[j] * QR;
        const int rs = 63 * w + rr[j];
        const int rg = rs < tx ? rs : tx - 1;
        const int qc = qd[j] < tailq ? qd[j] : tailq - 1;    // the last tile: quads past the tensor re-read the last one inside
        vo[j] = (unsigned)(((size_t)rg * Ty + 4 * qd[j]) * sizeof(float));
        vt[j] = (unsigned)(((size_t)rg * Ty + 4 * qc) * sizeof(float));
        const bool sv = rr[j] < 63 && rs < Tx;
        so[j] = (unsigned)(((size_t)(rs < Tx ? rs : 0) * Ty + 4 * qd[j]) * sizeof(float));
        mfull[j] = __builtin_amdgcn_ballot_w64(sv);
        mtail[j] = __builtin_amdgcn_ballot_w64(sv && qd[j] < tailq);
        rowok[j] = rs < tx;                                   // rows past the text are staged as log 0
    }
    const bool allok = 63 * w + 63 < tx;                      // (uniform: most waves select nothing)
    const int fl = lane & (SY_TW - 1);                        // this lane's frame of a tile's offsets
    auto issue = [&](int k, fs_u32x4 (&sl)[NJ], fs_u32x4 (&sa)[NJ], unsigned long long &o) {
        const int t = ntl - 1 - (k < ntl ? k : ntl - 1);      // k-th tile from the end, clamped
        const char *bl = reinterpret_cast<const char *>(lp_g) + (size_t)t * SY_TW * sizeof(float);
        const char *ba = reinterpret_cast<const char *>(al_g) + (size_t)t * SY_TW * sizeof(float);
        const bool last = t == ntl - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) fs_aload4(sl[j], last ? vt[j] : vo[j], bl);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fs_aload4(sa[j], last ? vt[j] : vo[j], ba);
        const int yo = t * SY_TW + fl;
        fs_aload2(o, (unsigned)((yo < ty ? yo : ty - 1) * sizeof(double)), offs_g);
    };
    auto phase = [&](int ph, fs_u32x4 (&sl)[NJ], fs_u32x4 (&sa)[NJ], unsigned long long &o) {
        const int k = ph - lag, ks = ph - 2 - lag;
        FsWaitG<NJ, OPS * (D - 1)>::on(sl, sa, o);            // tile k has landed (see above)
        if (k < ntl) {
            float *dl = tlp_w + (k & 1) * SY_TILE, *da = tal_w + (k & 1) * SY_TILE;
            const int y0 = (ntl - 1 - k) * SY_TW;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float f[4], a[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned rl = sl[j][jj], ra = sa[j][jj];        // (bit_cast straight from the vector element reads element 0)
                    f[jj] = (allok || rowok[j]) ? fs_nat(__builtin_bit_cast(float, rl)) : FS_NEG_NAT;
                    a[jj] = __builtin_bit_cast(float, ra);
                }
                *reinterpret_cast<float4 *>(dl + rr[j] * PITCH + 4 * qd[j]) = make_float4(f[0], f[1], f[2], f[3]);   // one ds_write_b128
                *reinterpret_cast<float4 *>(da + rr[j] * PITCH + 4 * qd[j]) = make_float4(a[0], a[1], a[2], a[3]);
            }
            if (lane < SY_TW) toff_w[(k & 1) * SY_TW + lane] = (y0 + lane < ty) ? __builtin_bit_cast(double, o) : 0.0;
        }
        if (ks >= 0 && ks < ntl) {
            const float *src = tgr_w + (ks & 1) * SY_TILE;
            const int t = ntl - 1 - ks, y0 = t * SY_TW;
            char *base = reinterpret_cast<char *>(gr_g) + (size_t)y0 * sizeof(float);
            const bool last = t == ntl - 1;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 f = *reinterpret_cast<const float4 *>(src + rr[j] * PITCH + 4 * qd[j]);       // one ds_read_b128
                fs_u32x4 d;
                d[0] = __builtin_bit_cast(unsigned, f.x); d[1] = __builtin_bit_cast(unsigned, f.y);
                d[2] = __builtin_bit_cast(unsigned, f.z); d[3] = __builtin_bit_cast(unsigned, f.w);
                fs_astore4(so[j], d, base, last ? mtail[j] : mfull[j]);
            }
        }
        issue(k + D, sl, sa, o);                              // (behind the stores: the wait's count)
        fs_lds_barrier();
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, ql[d], qa[d], qo[d]);
    for (int ph = 0; ph < lag; ++ph) fs_lds_barrier();        // (this stager's first tile is due in phase `lag`)
    for (int ph = lag; ph < nph; ph += D) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (ph + d < nph) phase(ph + d, ql[d], qa[d], qo[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) FsWaitG<NJ, 0>::on(ql[d], qa[d], qo[d]);     // the loads nobody consumed: every set named
}

// ---- forward: alpha (frame y relative to C_y), the offsets, log Z, loss ----
template <int R>
__global__ __launch_bounds__(FS_THREADS) void fwdsum_forward_kernel(FwdSumParams p) {
    // tile: TW frames x 64R rows = 32 KB.  Frame-major in LDS with the row stride padded by 4 floats: the
    // sweeper's 16-byte accesses stay aligned and conflict-free, and the stagers' transposing accesses
    // (consecutive lanes = consecutive frames) spread over 8 banks instead of hitting one
    constexpr int TW = 128 / R, ROWS = 64 * R + 4;
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tin = fs_smem;                         // [2][TW][ROWS] log-probs, frame-major
    float *tout = tin + 2 * TW * ROWS;            // [2][TW][ROWS] alpha
    double *toff = reinterpret_cast<double *>(tout + 2 * TW * ROWS);   // [2][TW]
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const bool sweeper = tid < 64;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    if (!(tx >= 1 && tx <= ty)) {                 // no monotonic alignment exists: loss = +inf
        if (tid == 0) { p.loss[b] = -FS_NEG_INF; p.logz[b] = (double)FS_NEG_INF; }
        for (int t = tid; t < p.NT; t += FS_THREADS) p.offs[(size_t)b * p.NT + t] = 0.0;
        return;
    }
    const int ntl = (ty + TW - 1) / TW;
    float prev[R];
#pragma unroll
    for (int j = 0; j < R; ++j) prev[j] = FS_NEG;
    float drift = 0.f;
    double C = 0.0;
    // phase ph: stagers load tile ph and store tile ph-2; the sweeper computes tile ph-1
    for (int ph = 0; ph < ntl + 2; ++ph) {
        if (!sweeper) {
            const int s = tid - 64;
            if (ph < ntl) {
                // Every load is unconditional (row and frame clamped into the utterance: the duplicates land
                // in cells the sweep overrides) -- a load behind a bounds test is compiled to "load, wait,
                // select" and the tile would arrive one element at a time.
                float *dst = tin + (ph & 1) * TW * ROWS;
                const int y0 = ph * TW;
                constexpr int NEL = TW * 64 * R, NIT = (NEL + FS_STAGERS - 1) / FS_STAGERS;
                constexpr int BATCH = NIT;               // the whole tile in flight: one memory latency per tile
                for (int i0 = 0; i0 < NIT; i0 += BATCH) {
                    float v[BATCH];
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) {
                        int e = s + (i0 + i) * FS_STAGERS;
                        e = e < NEL ? e : NEL - 1;
                        const int r = e / TW, c = e - r * TW;
                        const int rc = r < tx ? r : tx - 1, yc = y0 + c < ty ? y0 + c : ty - 1;
                        v[i] = p.logp[ubase + (size_t)rc * p.Ty + yc];
                    }
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) {
                        const int e = s + (i0 + i) * FS_STAGERS;
                        if (e < NEL) { const int r = e / TW, c = e - r * TW; dst[c * ROWS + r] = fs_in(v[i]); }
                    }
                }
            }
            if (ph >= 2) {
                const float *src = tout + (ph & 1) * TW * ROWS;
                const int y0 = (ph - 2) * TW;
                for (int e = s; e < TW * 64 * R; e += FS_STAGERS) {
                    const int r = e / TW, c = e - r * TW;
                    if (r < p.Tx && y0 + c < p.Ty) p.alpha[ubase + (size_t)r * p.Ty + y0 + c] = src[c * ROWS + r];
                }
                if (s < TW && y0 + s < ty) p.offs[(size_t)b * p.NT + y0 + s] = toff[(ph & 1) * TW + s];
            }
        } else if (ph >= 1 && ph <= ntl) {
            const int t = ph - 1, y0 = t * TW;
            const float *src = tin + (t & 1) * TW * ROWS + R * lane;
            float *dst = tout + (t & 1) * TW * ROWS + R * lane;
            float4 nx[R / 4];                          // the next frame's log-probs: the LDS read of frame c+1
#pragma unroll                                         // is in flight while frame c is computed
            for (int j = 0; j < R; j += 4) nx[j / 4] = *reinterpret_cast<const float4 *>(src + j);
            for (int c = 0; c < TW; ++c) {
                const int y = y0 + c;
                float lp[R];
#pragma unroll
                for (int j = 0; j < R; j += 4) {
                    lp[j] = nx[j / 4].x; lp[j + 1] = nx[j / 4].y; lp[j + 2] = nx[j / 4].z; lp[j + 3] = nx[j / 4].w;
                }
                const int cn = c + 1 < TW ? c + 1 : c;
#pragma unroll
                for (int j = 0; j < R; j += 4) nx[j / 4] = *reinterpret_cast<const float4 *>(src + cn * ROWS + j);
                // row -1 is log 1 = 0 for the very first frame (alpha[0,0] = logp[0,0]) and "log 0" afterwards
                const float up0 = fs_from_lane_below(y == 0 ? 0.f : FS_NEG, prev[R - 1]);
                float a[R];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const float v = fs_lae2(prev[j], j ? prev[j - 1] : up0) + (lp[j] - drift);
                    a[j] = (R * lane + j < tx && y < ty) ? v : FS_NEG;
                }
                C += (double)drift;
                if (lane == 0) toff[(t & 1) * TW + c] = C;
#pragma unroll
                for (int j = 0; j < R; j += 4)
                    *reinterpret_cast<float4 *>(dst + c * ROWS + j) = make_float4(a[j], a[j + 1], a[j + 2], a[j + 3]);
#pragma unroll
                for (int j = 0; j < R; ++j) prev[j] = a[j];
                if (y == ty - 1) {                                                  // uniform
#pragma unroll
                    for (int j = 0; j < R; ++j)
                        if (R * lane + j == tx - 1) {
                            const double lz = (double)a[j] + C;                     // log2 Z
                            p.logz[b] = lz;
                            p.loss[b] = (float)(-lz * FS_LN2);
                        }
                }
                if ((c & (FS_RB - 1)) == FS_RB - 1) {
                    // re-base the running column on its maximum and learn the per-frame drift
                    float m = prev[0];
#pragma unroll
                    for (int j = 1; j < R; ++j) m = fmaxf(m, prev[j]);
                    m = fs_wave_max(m);
                    if (m < 0.5f * FS_NEG) m = 0.f;                                  // an all-"log 0" column
                    C += (double)m;
                    drift += m * (1.0f / FS_RB);
#pragma unroll
                    for (int j = 0; j < R; ++j) prev[j] = fmaxf(prev[j] - m, FS_NEG);
                }
            }
        }
        fs_lds_barrier();
    }
}

// ---- backward: beta on the fly, gradient = -posterior ----
template <int R>
__global__ __launch_bounds__(FS_THREADS) void fwdsum_backward_kernel(FwdSumParams p) {
    constexpr int TW = 64 / R, ROWS = 64 * R + 4; // tile: TW frames x 64R rows = 16 KB (three operands, double-buffered)
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tlp = fs_smem;                         // [2][TW][ROWS] log-probs
    float *tal = tlp + 2 * TW * ROWS;             // [2][TW][ROWS] alpha (relative to C_y)
    float *tgr = tal + 2 * TW * ROWS;             // [2][TW][ROWS] gradient out
    double *toff = reinterpret_cast<double *>(tgr + 2 * TW * ROWS);    // [2][TW] C_y
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const bool sweeper = tid < 64;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const bool ok = tx >= 1 && tx <= ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    const int ntl = ok ? (ty + TW - 1) / TW : 0;
    // frames past the utterance's last tile (everything when no alignment exists): gradient 0
    for (int r = 0; r < p.Tx; ++r)
        for (int y = ntl * TW + tid; y < p.Ty; y += FS_THREADS) p.grad[ubase + (size_t)r * p.Ty + y] = 0.f;
    if (!ok) return;
    const double logz = p.logz[b];
    float g_prev[R];                              // beta[x,y+1] + logp[x,y+1], relative to D
#pragma unroll
    for (int j = 0; j < R; ++j) g_prev[j] = FS_NEG;
    float drift = 0.f;
    double D = 0.0;
    // phase ph: stagers load tile ntl-1-ph and store tile ntl-1-(ph-2); the sweeper computes tile ntl-1-(ph-1)
    for (int ph = 0; ph < ntl + 2; ++ph) {
        if (!sweeper) {
            const int s = tid - 64;
            if (ph < ntl) {
                const int t = ntl - 1 - ph, y0 = t * TW;
                float *dlp = tlp + (ph & 1) * TW * ROWS, *dal = tal + (ph & 1) * TW * ROWS;
                constexpr int NEL = TW * 64 * R, NIT = (NEL + FS_STAGERS - 1) / FS_STAGERS;
                constexpr int BATCH = NIT;
                for (int i0 = 0; i0 < NIT; i0 += BATCH) {
                    float v[BATCH], w[BATCH];
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) {
                        int e = s + (i0 + i) * FS_STAGERS;
                        e = e < NEL ? e : NEL - 1;
                        const int r = e / TW, c = e - r * TW;
                        const int rc = r < tx ? r : tx - 1, yc = y0 + c < ty ? y0 + c : ty - 1;
                        v[i] = p.logp[ubase + (size_t)rc * p.Ty + yc];
                        w[i] = p.alpha[ubase + (size_t)rc * p.Ty + yc];
                    }
#pragma unroll
                    for (int i = 0; i < BATCH; ++i) {
                        const int e = s + (i0 + i) * FS_STAGERS;
                        if (e < NEL) {
                            const int r = e / TW, c = e - r * TW;
                            dlp[c * ROWS + r] = fs_in(v[i]);
                            dal[c * ROWS + r] = w[i];
                        }
                    }
                }
                if (s < TW) toff[(ph & 1) * TW + s] = (y0 + s < ty) ? p.offs[(size_t)b * p.NT + y0 + s] : 0.0;
            }
            if (ph >= 2) {
                const float *src = tgr + (ph & 1) * TW * ROWS;
                const int y0 = (ntl - 1 - (ph - 2)) * TW;
                for (int e = s; e < TW * 64 * R; e += FS_STAGERS) {
                    const int r = e / TW, c = e - r * TW;
                    if (r < p.Tx && y0 + c < p.Ty) p.grad[ubase + (size_t)r * p.Ty + y0 + c] = src[c * ROWS + r];
                }
            }
        } else if (ph >= 1 && ph <= ntl) {
            const int buf = (ph - 1) & 1, y0 = (ntl - 1 - (ph - 1)) * TW;
            const float *slp = tlp + buf * TW * ROWS + R * lane, *sal = tal + buf * TW * ROWS + R * lane;
            float *dst = tgr + buf * TW * ROWS + R * lane;
            for (int c = TW - 1; c >= 0; --c) {
                const int y = y0 + c;
                if (y >= ty) {                                                       // uniform: padding frames
#pragma unroll
                    for (int j = 0; j < R; j += 4) *reinterpret_cast<float4 *>(dst + c * ROWS + j) = make_float4(0.f, 0.f, 0.f, 0.f);
                    continue;
                }
                float lp[R], al[R];
#pragma unroll
                for (int j = 0; j < R; j += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(slp + c * ROWS + j);
                    const float4 w = *reinterpret_cast<const float4 *>(sal + c * ROWS + j);
                    lp[j] = v.x; lp[j + 1] = v.y; lp[j + 2] = v.z; lp[j + 3] = v.w;
                    al[j] = w.x; al[j + 1] = w.y; al[j + 2] = w.z; al[j + 3] = w.w;
                }
                const float st = (float)(toff[buf * TW + c] + D - logz);            // uniform
                const float dnR = fs_from_lane_above(FS_NEG, g_prev[0]);             // row below this lane's last one
                float g[R], gr[R];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int row = R * lane + j;
                    float beta;
                    if (y == ty - 1) beta = (row == tx - 1) ? 0.f : FS_NEG;          // uniform branch
                    else             beta = fs_lae2(g_prev[j], j + 1 < R ? g_prev[j + 1] : dnR);
                    if (row >= tx) beta = FS_NEG;
                    gr[j] = -__builtin_amdgcn_exp2f(al[j] + beta + st);              // 2^(-1e30) = 0
                    g[j] = fmaxf(beta + (lp[j] - drift), FS_NEG);
                }
                D += (double)drift;
#pragma unroll
                for (int j = 0; j < R; j += 4)
                    *reinterpret_cast<float4 *>(dst + c * ROWS + j) = make_float4(gr[j], gr[j + 1], gr[j + 2], gr[j + 3]);
#pragma unroll
                for (int j = 0; j < R; ++j) g_prev[j] = g[j];
                if ((c & (FS_RB - 1)) == 0) {
                    float m = g_prev[0];
#pragma unroll
                    for (int j = 1; j < R; ++j) m = fmaxf(m, g_prev[j]);
                    m = fs_wave_max(m);
                    if (m < 0.5f * FS_NEG) m = 0.f;
                    D += (double)m;
                    drift += m * (1.0f / FS_RB);
#pragma unroll
                    for (int j = 0; j < R; ++j) g_prev[j] = fmaxf(g_prev[j] - m, FS_NEG);
                }
            }
        }
        fs_lds_barrier();
    }
}

// --------------------------------------------------------------------------
// Systolic form for T_text <= 252 (the usual TTS range): FOUR sweeping waves instead of one.
//
// One wave with R = 4 rows per lane spends ~530 cycles per frame, most of it waiting on its own
// exp2 -> log2 chains.  Here wave w owns text rows 63w .. 63w+62, one per lane, and the four waves form
// the same tile pipeline as the path search (maxpath.hip): in phase p wave w sweeps tile p-1-w of SY_TW
// frames, a ghost lane replays the neighbouring wave's boundary row (forward: lane 0 = row 63w-1, read
// from the upper wave's alpha tile; backward: lane 63 = row 63w+63, read from the lower wave's g tile),
// one barrier per tile.  Waves 4..7 stage: stager w moves wave w's tiles.
//
// Each wave keeps its OWN running offset (C_w forward, D_w backward; per-wave drift and one re-basing per tile keep
// every row block near 0 whatever the others do).  Inside a tile the offset is linear, C_w(y) = Cg + k * drift, so a
// boundary value travels with its wave's (Cg, drift) of that tile and is converted on arrival with one fma:
// ghost = value + (float(Cg_sender - Cg_receiver) + k * (drift_sender - drift_receiver)).  The workspace holds one offset
// per (wave, frame); posterior = exp2(alpha + beta + C_w[y] + D_w[y] - log Z).
// Round 4: tiles slot-major in LDS (16-byte accesses on both sides), stagers issued by hand with counted
// waits (fs_stager_by_hand), no per-row select and no double arithmetic in the frame loop, log2 e folded into the
// sweeper's fma; with the gradient, both sweeps in one launch and a combining pass (fwdsum_both_sys_kernel).
// --------------------------------------------------------------------------
// NW sweeping waves (T_text <= 63 NW) and TW frames per tile: <4, 16> up to 252 rows, <8, 8> up to 504 (the
// 16-wave workgroup has 128 VGPRs per lane and the backward kernel three tile arrays in LDS)

template <int SY_NW, int SY_TW>
__device__ __forceinline__ void fwdsum_forward_sys_body(const FwdSumParams &p, const int b) {
    // tiles are SLOT-major, a slot's TW frames contiguous and slots TW + 4 floats apart: every wave moves its tile with
    // 16-byte LDS accesses (a sweeper lane's operands are TW / 4 ds_read_b128, conflict-free at this pitch; frame-major,
    // they were 2 x TW ds_read_b32 and 850 of a phase's 3 200 cycles: tools/fwdsum_stamps.py)
    constexpr int PITCH = SY_TW + 4, SY_TILE = 64 * PITCH;
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tin = fs_smem;                                     // [NW][2][64][PITCH] log-probs (slot = lane)
    float *tout = tin + SY_NW * 2 * SY_TILE;                  // [NW][2][64][PITCH] alpha
    double *toff = reinterpret_cast<double *>(tout + SY_NW * 2 * SY_TILE);   // [NW][2][TW] C_w per frame
    constexpr int RB = SY_TW, NG = SY_TW / RB;                // frames between re-basings: once per tile (the rebasing
    // ladder is ~170 cycles of the sweeper's chain; fp32 holds the column near 0 over 16 frames as well as over 8)
    double *tcg = toff + SY_NW * 2 * SY_TW;                   // [NW][2][NG] C_w at the start of each group ...
    float *tdr = reinterpret_cast<float *>(tcg + SY_NW * 2 * NG);            // [NW][2][NG] ... and the group's drift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & (SY_NW - 1);
    const bool sweeper = wave < SY_NW;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    double *offs = p.offs + ((size_t)b * SY_NW_MAX + w) * p.NT;
    if (!(tx >= 1 && tx <= ty)) {                             // no monotonic alignment exists: loss = +inf
        if (tid == 0) { p.loss[b] = -FS_NEG_INF; p.logz[b] = (double)FS_NEG_INF; }
        if (!sweeper) for (int t = lane; t < p.NT; t += 64) offs[t] = 0.0;
        return;
    }
    const int ntl = (ty + SY_TW - 1) / SY_TW;
    const int row = 63 * w + lane - 1;                        // sweeper: lane 0 is the ghost (row 63w-1)
    float prev = (w == 0 && lane == 0) ? 0.f : FS_NEG;        // row -1 is log 1 before the first frame
    const bool ghost = lane == 0;
    float drift = 0.f;
    // Two loops, one per role (same phase count): the stagers' tile in flight is then live in their loop only
    const bool by_hand = p.Ty % 4 == 0 && ((reinterpret_cast<uintptr_t>(p.logp) | reinterpret_cast<uintptr_t>(p.alpha)) & 15) == 0 &&
                         (size_t)p.Tx * p.Ty * sizeof(float) < (1ull << 32);
    if (!sweeper && by_hand) {
        fs_stager_by_hand<SY_TW, false>(p.logp + ubase, p.alpha + ubase, offs, tin + w * 2 * SY_TILE, tout + w * 2 * SY_TILE,
                                        toff + w * 2 * SY_TW, w, w, lane, tx, ty, p.Tx, p.Ty, ntl, ntl + SY_NW + 1,
                                        p.stamps ? p.stamps + ((size_t)b * 16 + wave) * 8 : nullptr);
    } else if (!sweeper) {
        // stagers: TWO tiles in flight (even tiles in one register set, odd ones in the other; the phase loop is unrolled
        // by two so that nothing is copied -- a copy would wait for the newest loads).  With one tile in flight a phase
        // could not be shorter than a memory round trip, whatever the sweepers did.
        float vA[SY_TW], vB[SY_TW];
        const size_t ubytes = (size_t)p.Tx * p.Ty * sizeof(float);
        const bool fastio = ubytes < 0xFFFFFF00ull;
        const __amdgpu_buffer_rsrc_t rs_in = fs_rsrc(p.logp + ubase, ubytes), rs_out = fs_rsrc(p.alpha + ubase, ubytes);
        unsigned vo[SY_TW], so[SY_TW];                        // element offsets inside the block: loads (rows clamped), stores
#pragma unroll
        for (int i = 0; i < SY_TW; ++i) {
            const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
            const int rs = 63 * w + r - 1;
            const int rg = rs < 0 ? 0 : (rs < tx ? rs : tx - 1);
            vo[i] = (unsigned)(((size_t)rg * p.Ty + c) * sizeof(float));
            so[i] = (r >= 1 && rs < p.Tx) ? (unsigned)(((size_t)rs * p.Ty + c) * sizeof(float)) : FS_DROP;
        }
        auto stage_issue = [&](int tl, float (&v)[SY_TW]) {      // unconditional loads (row and frame clamped into the utterance)
            const int tc = tl < ntl ? tl : ntl - 1;
            const int y0 = tc * SY_TW;
            if (fastio && y0 + SY_TW <= p.Ty) {                   // a whole tile inside the tensor: no address arithmetic
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) v[i] = fs_bload(rs_in, vo[i], (unsigned)y0 * 4u);
                return;
            }
    #pragma unroll
            for (int i = 0; i < SY_TW; ++i) {
                const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;     // slot r, frame c
                int rg = 63 * w + r - 1;
                rg = rg < 0 ? 0 : (rg < tx ? rg : tx - 1);
                const int yc = y0 + c < ty ? y0 + c : ty - 1;
                v[i] = p.logp[ubase + (size_t)rg * p.Ty + yc];
            }
        };
        auto phase = [&](int ph, float (&v)[SY_TW]) {
            const int tl = ph - w, ts = ph - 2 - w;
            if (tl >= 0 && tl < ntl) {
                float *dst = tin + (w * 2 + (tl & 1)) * SY_TILE;
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) {
                    const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                    dst[r * PITCH + c] = (63 * w + r - 1 < tx) ? fs_nat(v[i]) : FS_NEG_NAT;   // rows past the text: log 0
                }
                stage_issue(tl + 2, v);
            }
            if (ts >= 0 && ts < ntl) {
                const float *src = tout + (w * 2 + (ts & 1)) * SY_TILE;
                const int y0 = ts * SY_TW;
                if (fastio && y0 + SY_TW <= p.Ty) {
#pragma unroll
                    for (int i = 0; i < SY_TW; ++i) {
                        const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                        fs_bstore(src[r * PITCH + c], rs_out, so[i], (unsigned)y0 * 4u);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < SY_TW; ++i) {
                        const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                        const int rg = 63 * w + r - 1;
                        if (r >= 1 && rg < p.Tx && y0 + c < p.Ty) p.alpha[ubase + (size_t)rg * p.Ty + y0 + c] = src[r * PITCH + c];
                    }
                }
                if (lane < SY_TW && y0 + lane < ty) offs[y0 + lane] = toff[(w * 2 + (ts & 1)) * SY_TW + lane];
            }
            fs_lds_barrier();
        };
        stage_issue(0, vA);
        stage_issue(1, vB);
        const int nph = ntl + SY_NW + 1;
        for (int ph = 0; ph < w; ++ph) fs_lds_barrier();          // (this stager's first tile is due in phase w)
        for (int ph = w; ph < nph; ph += 2) {
            phase(ph, vA);
            if (ph + 1 < nph) phase(ph + 1, vB);
        }
    } else {
        // Per frame the chain is dpp, sub, exp2, add, log2, add (tools/microbench_fwdsum.hip: 75 cycles) and everything
        // else is kept off it and cheap: no select per row (the stagers wrote log 0 into rows past the text, and log 0
        // absorbs), no double arithmetic (the offset is linear inside a re-basing group: C(y) = Cg + k * drift, so a
        // boundary value is converted with one fma from the two waves' (Cg, drift) pairs, published once per group).
        double Cg = 0.0;                                      // this wave's offset at the start of the current group
        unsigned long long sa[4] = {0, 0, 0, 0}, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int ph = 0; ph < ntl + SY_NW + 1; ++ph) {
            const int t = ph - 1 - w;
            if (p.stamps) c0 = c1 = c2 = c3 = __builtin_amdgcn_s_memtime();
            if (t >= 0 && t < ntl) {
                const int y0 = t * SY_TW, buf = t & 1;
                const float4 *src4 = reinterpret_cast<const float4 *>(tin + (w * 2 + buf) * SY_TILE + lane * PITCH);
                float4 *dst4 = reinterpret_cast<float4 *>(tout + (w * 2 + buf) * SY_TILE + lane * PITCH);
                double *myoff = toff + (w * 2 + buf) * SY_TW;
                // the wave above: its last row's alpha (slot 63) and its (Cg, drift) of the same groups
                const int wu = w ? w - 1 : 0;
                const float4 *ring4 = reinterpret_cast<const float4 *>(tout + (wu * 2 + buf) * SY_TILE + 63 * PITCH);
                float lpv[SY_TW], rgv[SY_TW], av[4];
                double scg[NG];
                float sdr[NG];
#pragma unroll
                for (int i = 0; i < SY_TW / 4; ++i) {
                    const float4 l4 = src4[i], r4 = ring4[i];
                    lpv[4 * i] = l4.x; lpv[4 * i + 1] = l4.y; lpv[4 * i + 2] = l4.z; lpv[4 * i + 3] = l4.w;
                    rgv[4 * i] = w ? r4.x : FS_NEG; rgv[4 * i + 1] = w ? r4.y : FS_NEG;
                    rgv[4 * i + 2] = w ? r4.z : FS_NEG; rgv[4 * i + 3] = w ? r4.w : FS_NEG;
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) { scg[g] = tcg[(wu * 2 + buf) * NG + g]; sdr[g] = tdr[(wu * 2 + buf) * NG + g]; }
                if (p.stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c1 = __builtin_amdgcn_s_memtime(); }
                {   // what the wave below converts with, and the offsets of the tile's frames for the workspace: known at the
                    // tile's start (one re-basing per tile), so the LDS writes land under the frames, not before the barrier
                    static_assert(NG == 1, "published before the frames: one group per tile");
                    if (lane == 0) { tcg[(w * 2 + buf) * NG] = Cg; tdr[(w * 2 + buf) * NG] = drift; }
                    if (lane < SY_TW) myoff[lane] = Cg + (double)(lane + 1) * (double)drift;
                }
                auto frames = [&](auto tail) {
                    constexpr bool TAIL = decltype(tail)::value;
                    float D0 = 0.f, dl = 0.f;
#pragma unroll
                    for (int c = 0; c < SY_TW; ++c) {
                        const int y = y0 + c, g = c / RB, k = (c & (RB - 1)) + 1;
                        if (k == 1) {                                                // a group starts (uniform values)
                            D0 = w ? (float)(scg[g] - Cg) : 0.f;
                            dl = w ? sdr[g] - drift : 0.f;
                        }
                        const float up = fs_from_lane_below(FS_NEG, prev);
                        float m;
                        asm("v_max_f32_e32 %0, %1, %2" : "=v"(m) : "v"(prev), "v"(up));
                        const float base = m + fmaf(lpv[c], FS_LOG2E, -drift);             // (the staged value is a natural log)
                        float v = base + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(prev - up)));
                        if (TAIL && y >= ty) v = FS_NEG;                             // uniform
                        const float gh = rgv[c] + (D0 + (float)k * dl);              // the sender's value, on this wave's offset
                        const float a = ghost ? gh : v;
                        av[c & 3] = a;
                        if ((c & 3) == 3) dst4[c >> 2] = make_float4(av[0], av[1], av[2], av[3]);
                        prev = a;
                        if (TAIL && y == ty - 1) {                                   // uniform
                            if (row == tx - 1 && !ghost) {
                                const double lz = (double)a + (Cg + (double)k * (double)drift);   // log2 Z
                                p.logz[b] = lz;
                                p.loss[b] = (float)(-lz * FS_LN2);
                            }
                        }
                        if (k == RB) {
                            // re-base this wave's running column on its maximum and learn the per-frame drift
                            float mx = fs_wave_max_dpp(prev);
                            if (mx < 0.5f * FS_NEG) mx = 0.f;                        // an all-"log 0" column
                            Cg += (double)RB * (double)drift + (double)mx;
                            drift += mx * (1.0f / RB);
                            prev = fmaxf(prev - mx, FS_NEG);
                        }
                    }
                };
                if (y0 + SY_TW < ty) frames(std::false_type{});
                else                 frames(std::true_type{});
                if (p.stamps) { asm volatile("" :: "v"(prev)); c2 = __builtin_amdgcn_s_memtime(); }
                if (p.stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c3 = __builtin_amdgcn_s_memtime(); }
            }
            fs_lds_barrier();
            if (p.stamps) { sa[0] += c1 - c0; sa[1] += c2 - c1; sa[2] += c3 - c2; sa[3] += __builtin_amdgcn_s_memtime() - c3; }
        }
        if (p.stamps && lane == 0) {
            unsigned long long *st = p.stamps + ((size_t)b * 16 + wave) * 8;
            st[0] = sa[0]; st[1] = sa[1]; st[2] = sa[2]; st[3] = sa[3]; st[4] = (unsigned long long)(ntl + SY_NW + 1);
        }
    }
}

template <int SY_NW, int SY_TW>
__global__ __launch_bounds__(2 * SY_NW * 64) void fwdsum_forward_sys_kernel(FwdSumParams p) {
    fwdsum_forward_sys_body<SY_NW, SY_TW>(p, blockIdx.x);
}

// BETA_ONLY: the sweep alone, for the form in which it runs BESIDE the forward sweep (fwdsum_both_sys_kernel: alpha and
// beta do not depend on each other, only the posterior needs both): no alpha, no offsets, no log Z come in; what goes out
// through the gradient tile is beta itself, relative to the wave's offset of that frame (D_w[y], to p.doffs), into
// p.grad -- which fwdsum_combine_kernel then turns into the gradient in place.
template <int SY_NW, int SY_TW, bool BETA_ONLY, bool GRADHAND = false>
__device__ __forceinline__ void fwdsum_backward_sys_body(const FwdSumParams &p, const int b) {
    constexpr int PITCH = SY_TW + 4, SY_TILE = 64 * PITCH, SY_THREADS = 2 * SY_NW * 64;   // slot-major tiles (see the forward kernel)
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tlp = fs_smem;                                     // [NW][2][64][PITCH] log-probs (slot = lane)
    float *tal = tlp + SY_NW * 2 * SY_TILE;                   // alpha (relative to C_w)
    float *tgr = tal + SY_NW * 2 * SY_TILE;                   // gradient out
    float *tg = tgr + SY_NW * 2 * SY_TILE;                    // [NW][2][TW] g = beta + logp (relative to D_w) of a wave's FIRST row: feeds the wave above
    float *tgd = tg + SY_NW * 2 * SY_TW;                      // [NW][64][4] where the other lanes write theirs
    double *toff = reinterpret_cast<double *>(tgd + SY_NW * 64 * 4);       // [NW][2][TW] C_w per frame
    double *tdof = toff + SY_NW * 2 * SY_TW;                  // [NW][2][TW] D_w per frame
    constexpr int RB = SY_TW, NG = SY_TW / RB;                // frames between re-basings: once per tile (the rebasing
    // ladder is ~170 cycles of the sweeper's chain; fp32 holds the column near 0 over 16 frames as well as over 8)
    double *tdg = tdof + SY_NW * 2 * SY_TW;                   // [NW][2][NG] D_w at the start of each group ...
    float *tdr = reinterpret_cast<float *>(tdg + SY_NW * 2 * NG);            // [NW][2][NG] ... and the group's drift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & (SY_NW - 1), wr = SY_NW - 1 - w;     // wave SY_NW-1 (the last rows) leads
    const bool sweeper = wave < SY_NW;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const bool ok = tx >= 1 && tx <= ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    const int ntl = ok ? (ty + SY_TW - 1) / SY_TW : 0;
    // frames past the utterance's last tile (everything when no alignment exists): gradient 0
    // (BETA_ONLY: fwdsum_combine_kernel writes every element of the gradient)
    if (!BETA_ONLY)
        for (int r = 0; r < p.Tx; ++r)
            for (int y = ntl * SY_TW + tid; y < p.Ty; y += SY_THREADS) p.grad[ubase + (size_t)r * p.Ty + y] = 0.f;
    if (!ok) return;
    const double logz = BETA_ONLY ? 0.0 : p.logz[b];
    const double *offs = p.offs + ((size_t)b * SY_NW_MAX + w) * p.NT;
    double *doffs = p.doffs + ((size_t)b * SY_NW_MAX + w) * p.NT;
    const int row = 63 * w + lane;                            // sweeper: lane 63 is the ghost (row 63w+63)
    float g_prev = FS_NEG;                                    // beta[x,y+1] + logp[x,y+1], relative to D
    const bool ghost = lane == 63;
    float drift = 0.f;
    // Two loops, one per role (the phase count is the same): written as one loop with the role tested inside, the
    // stagers' tile in flight was live through the sweepers' code as well and the kernel spilled (236 -> 493 us).
    const bool by_hand = BETA_ONLY && p.Ty % 4 == 0 &&
                         ((reinterpret_cast<uintptr_t>(p.logp) | reinterpret_cast<uintptr_t>(p.grad)) & 15) == 0 &&
                         (size_t)p.Tx * p.Ty * sizeof(float) < (1ull << 32);
    // GRADHAND (the host checked: T_mel % 4 == 0, 16-byte aligned tensors, 32-bit offsets): the gradient-making kernel
    // with the hand-issued stager and NOTHING of the compiler-scheduled one in it -- that one spills (9 registers), and
    // a kernel that spills must not have hand-issued loads in flight
    if (GRADHAND) {
        if (!sweeper)
            fs_grad_stager_by_hand<SY_TW>(p.logp + ubase, p.alpha + ubase, p.grad + ubase, offs, tlp + w * 2 * SY_TILE,
                                          tal + w * 2 * SY_TILE, tgr + w * 2 * SY_TILE, toff + w * 2 * SY_TW, wr, w, lane, tx, ty,
                                          p.Tx, p.Ty, ntl, ntl + SY_NW + 1);
    } else if (!sweeper && by_hand) {
        fs_stager_by_hand<SY_TW, true>(p.logp + ubase, p.grad + ubase, doffs, tlp + w * 2 * SY_TILE, tgr + w * 2 * SY_TILE,
                                       tdof + w * 2 * SY_TW, wr, w, lane, tx, ty, p.Tx, p.Ty, ntl, ntl + SY_NW + 1);
    } else if (!sweeper) {
        // two tiles (log-probs, alpha, offsets) in flight, as in the forward kernel: even tiles in set A, odd ones in set B
        struct Set { float v[SY_TW], u[SY_TW]; double o; };
        Set sA, sB;
        const size_t ubytes = (size_t)p.Tx * p.Ty * sizeof(float);
        const bool fastio = ubytes < 0xFFFFFF00ull;
        const __amdgpu_buffer_rsrc_t rs_in = fs_rsrc(p.logp + ubase, ubytes), rs_al = fs_rsrc(p.alpha + ubase, ubytes),
                                     rs_out = fs_rsrc(p.grad + ubase, ubytes);
        unsigned vo[SY_TW], so[SY_TW];                        // element offsets inside the block: loads (rows clamped), stores
#pragma unroll
        for (int i = 0; i < SY_TW; ++i) {
            const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
            const int rs = 63 * w + r;
            const int rg = rs < tx ? rs : tx - 1;
            vo[i] = (unsigned)(((size_t)rg * p.Ty + c) * sizeof(float));
            so[i] = (r < 63 && rs < p.Tx) ? (unsigned)(((size_t)rs * p.Ty + c) * sizeof(float)) : FS_DROP;
        }
        auto stage_issue = [&](int kl, Set &q) {              // tile counted from the end, clamped
            const int kc = kl < ntl ? kl : ntl - 1;
            const int y0 = (ntl - 1 - kc) * SY_TW;
            if (fastio && y0 + SY_TW <= p.Ty) {                   // a whole tile inside the tensor: no address arithmetic
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) {
                    q.v[i] = fs_bload(rs_in, vo[i], (unsigned)y0 * 4u);
                    if (!BETA_ONLY) q.u[i] = fs_bload(rs_al, vo[i], (unsigned)y0 * 4u);
                }
                const int yo = y0 + (lane & (SY_TW - 1));
                if (!BETA_ONLY) q.o = offs[yo < ty ? yo : ty - 1];
                return;
            }
#pragma unroll
            for (int i = 0; i < SY_TW; ++i) {
                const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                int rg = 63 * w + r;
                rg = rg < tx ? rg : tx - 1;
                const int yc = y0 + c < ty ? y0 + c : ty - 1;
                q.v[i] = p.logp[ubase + (size_t)rg * p.Ty + yc];
                if (!BETA_ONLY) q.u[i] = p.alpha[ubase + (size_t)rg * p.Ty + yc];
            }
            const int yo = y0 + (lane & (SY_TW - 1));
            if (!BETA_ONLY) q.o = offs[yo < ty ? yo : ty - 1];
        };
        auto phase = [&](int ph, Set &q) {
            const int kl = ph - wr, ks = ph - 2 - wr;         // tile counted from the end
            if (kl >= 0 && kl < ntl) {
                const int t = ntl - 1 - kl, y0 = t * SY_TW;
                float *dlp = tlp + (w * 2 + (kl & 1)) * SY_TILE, *dal = tal + (w * 2 + (kl & 1)) * SY_TILE;
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) {
                    const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                    dlp[r * PITCH + c] = (63 * w + r < tx) ? fs_nat(q.v[i]) : FS_NEG_NAT;     // rows past the text: log 0
                    if (!BETA_ONLY) dal[r * PITCH + c] = q.u[i];
                }
                if (!BETA_ONLY && lane < SY_TW) toff[(w * 2 + (kl & 1)) * SY_TW + lane] = (y0 + lane < ty) ? q.o : 0.0;
                stage_issue(kl + 2, q);
            }
            if (ks >= 0 && ks < ntl) {
                const float *src = tgr + (w * 2 + (ks & 1)) * SY_TILE;
                const int y0 = (ntl - 1 - ks) * SY_TW;
                if (fastio && y0 + SY_TW <= p.Ty) {
#pragma unroll
                    for (int i = 0; i < SY_TW; ++i) {
                        const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                        fs_bstore(src[r * PITCH + c], rs_out, so[i], (unsigned)y0 * 4u);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < SY_TW; ++i) {
                        const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                        const int rg = 63 * w + r;
                        if (r < 63 && rg < p.Tx && y0 + c < p.Ty) p.grad[ubase + (size_t)rg * p.Ty + y0 + c] = src[r * PITCH + c];
                    }
                }
                if (BETA_ONLY && lane < SY_TW && y0 + lane < ty) doffs[y0 + lane] = tdof[(w * 2 + (ks & 1)) * SY_TW + lane];
            }
            fs_lds_barrier();
        };
        stage_issue(0, sA);
        stage_issue(1, sB);
        const int nph = ntl + SY_NW + 1;
        for (int ph = 0; ph < wr; ++ph) fs_lds_barrier();         // (this stager's first tile is due in phase wr)
        for (int ph = wr; ph < nph; ph += 2) {
            phase(ph, sA);
            if (ph + 1 < nph) phase(ph + 1, sB);
        }
    }
    if (sweeper) {
        // (the frame's work is arranged as in fwdsum_forward_sys_body: no per-row select, offsets linear inside a group)
        double Dg = 0.0;                                      // this wave's offset at the start of the current group
        for (int ph = 0; ph < ntl + SY_NW + 1; ++ph) {
            const int kt = ph - 1 - wr;
            if (kt >= 0 && kt < ntl) {
                const int buf = kt & 1, y0 = (ntl - 1 - kt) * SY_TW;
                const float4 *slp4 = reinterpret_cast<const float4 *>(tlp + (w * 2 + buf) * SY_TILE + lane * PITCH);
                const float4 *sal4 = reinterpret_cast<const float4 *>(tal + (w * 2 + buf) * SY_TILE + lane * PITCH);
                float4 *dgr4 = reinterpret_cast<float4 *>(tgr + (w * 2 + buf) * SY_TILE + lane * PITCH);
                // lane 0's g goes to the ring the wave above reads, everybody else's to a slot of their own
                float4 *dg4 = reinterpret_cast<float4 *>(lane == 0 ? tg + (w * 2 + buf) * SY_TW : tgd + (w * 64 + lane) * 4);
                const int dgs = lane == 0 ? 1 : 0;                                  // (quads advance in the ring only)
                const double *myoff = toff + (w * 2 + buf) * SY_TW;
                double *mydof = tdof + (w * 2 + buf) * SY_TW;
                // the wave below: its first row's g and its (Dg, drift) of the same groups
                const bool has = w + 1 < SY_NW;
                const int wb = has ? w + 1 : w;
                const float4 *ring4 = reinterpret_cast<const float4 *>(tg + (wb * 2 + buf) * SY_TW);
                // the whole tile's operands up front (one LDS latency per tile, not one per frame)
                float lpv[SY_TW], alv[SY_TW], rgv[SY_TW], ov[4], gq[4];
                double cov[SY_TW];
                double sdg[NG];
                float sdr[NG];
#pragma unroll
                for (int i = 0; i < SY_TW / 4; ++i) {
                    const float4 l4 = slp4[i], r4 = ring4[i];
                    lpv[4 * i] = l4.x; lpv[4 * i + 1] = l4.y; lpv[4 * i + 2] = l4.z; lpv[4 * i + 3] = l4.w;
                    rgv[4 * i] = has ? r4.x : FS_NEG; rgv[4 * i + 1] = has ? r4.y : FS_NEG;
                    rgv[4 * i + 2] = has ? r4.z : FS_NEG; rgv[4 * i + 3] = has ? r4.w : FS_NEG;
                    if (!BETA_ONLY) {
                        const float4 a4 = sal4[i];
                        alv[4 * i] = a4.x; alv[4 * i + 1] = a4.y; alv[4 * i + 2] = a4.z; alv[4 * i + 3] = a4.w;
                    }
                }
                if (!BETA_ONLY) {
#pragma unroll
                    for (int c = 0; c < SY_TW; ++c) cov[c] = myoff[c];
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) { sdg[g] = tdg[(wb * 2 + buf) * NG + g]; sdr[g] = tdr[(wb * 2 + buf) * NG + g]; }
                {   // (published at the tile's start: see the forward kernel)
                    static_assert(NG == 1, "published before the frames: one group per tile");
                    if (lane == 0) { tdg[(w * 2 + buf) * NG] = Dg; tdr[(w * 2 + buf) * NG] = drift; }
                    if (lane < SY_TW) mydof[lane] = Dg + (double)(RB - (lane & (RB - 1))) * (double)drift;
                }
                auto frames = [&](auto tail) {
                    constexpr bool TAIL = decltype(tail)::value;
                    float D0 = 0.f, dl = 0.f;
                    double Dlz = 0.0;
#pragma unroll
                    for (int c = SY_TW - 1; c >= 0; --c) {
                        const int y = y0 + c, g = c / RB, k = RB - (c & (RB - 1));   // k-th frame of its group
                        if (k == 1) {                                                // a group starts (uniform values)
                            D0 = has ? (float)(sdg[g] - Dg) : 0.f;
                            dl = has ? sdr[g] - drift : 0.f;
                            Dlz = Dg - logz;
                        }
                        if (TAIL && y >= ty) {                                       // uniform: padding frames (the state is all
                            ov[c & 3] = 0.f;                                         // log 0 and the drift 0: nothing moves)
                            gq[c & 3] = FS_NEG;
                            if ((c & 3) == 0) {
                                dgr4[c >> 2] = make_float4(ov[0], ov[1], ov[2], ov[3]);
                                dg4[dgs * (c >> 2)] = make_float4(gq[0], gq[1], gq[2], gq[3]);
                            }
                            continue;
                        }
                        const float dn = fs_from_lane_above(FS_NEG, g_prev);          // row below
                        float beta;
                        if (TAIL && y == ty - 1) {                                   // uniform branch
                            beta = (row == tx - 1) ? 0.f : FS_NEG;
                        } else {
                            float m;
                            asm("v_max_f32_e32 %0, %1, %2" : "=v"(m) : "v"(g_prev), "v"(dn));
                            beta = m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(g_prev - dn)));
                        }
                        if (BETA_ONLY) {
                            ov[c & 3] = beta - drift;                 // relative to the D of this frame (Dg + k * drift)
                        } else {
                            // C_w[y] + D_w - log Z with D_w = Dg + (k - 1) * drift: the offset beta is relative to
                            const float st = (float)(cov[c] + Dlz) + (float)(k - 1) * drift;
                            ov[c & 3] = -__builtin_amdgcn_exp2f(alv[c] + beta + st);        // 2^(-1e30) = 0
                        }
                        const float gv = beta + fmaf(lpv[c], FS_LOG2E, -drift);
                        const float gh = rgv[c] + (D0 + (float)k * dl);              // the sender's value, on this wave's offset
                        const float gn = ghost ? gh : gv;
                        gq[c & 3] = gn;
                        if ((c & 3) == 0) {                                          // a quad of frames is complete (c runs down)
                            dgr4[c >> 2] = make_float4(ov[0], ov[1], ov[2], ov[3]);
                            dg4[dgs * (c >> 2)] = make_float4(gq[0], gq[1], gq[2], gq[3]);
                        }
                        g_prev = gn;
                        if (k == RB) {
                            float mx = fs_wave_max_dpp(g_prev);
                            if (mx < 0.5f * FS_NEG) mx = 0.f;
                            Dg += (double)RB * (double)drift + (double)mx;
                            drift += mx * (1.0f / RB);
                            g_prev = fmaxf(g_prev - mx, FS_NEG);
                        }
                    }
                };
                if (kt != 0) frames(std::false_type{});
                else         frames(std::true_type{});
            }
            fs_lds_barrier();
        }
    }
}

template <int SY_NW, int SY_TW, bool GRADHAND>
__global__ __launch_bounds__(2 * SY_NW * 64) void fwdsum_backward_sys_kernel(FwdSumParams p) {
    fwdsum_backward_sys_body<SY_NW, SY_TW, false, GRADHAND>(p, blockIdx.x);
}

// The two sweeps side by side (2B workgroups <= the CUs): alpha runs forward and beta backward through the same log-probs
// and neither reads the other -- only the posterior needs both.  Even blocks sweep alpha (loss, log Z, alpha and its
// offsets into the workspace, exactly fwdsum_forward_sys_kernel), odd blocks beta (into the gradient buffer, offsets
// into p.doffs); fwdsum_combine_kernel then makes the gradient of it in place.  One launch of ~max(forward, backward)
// plus one streaming pass instead of forward + backward one after the other on a quarter of the chip.
template <int SY_NW, int SY_TW>
__global__ __launch_bounds__(2 * SY_NW * 64) void fwdsum_both_sys_kernel(FwdSumParams p) {
    const int b = blockIdx.x >> 1;
    if (blockIdx.x & 1) fwdsum_backward_sys_body<SY_NW, SY_TW, true>(p, b);
    else                fwdsum_forward_sys_body<SY_NW, SY_TW>(p, b);
}

// gradient = -posterior = -2^(alpha + beta - log Z), in place over the beta the sweep left in p.grad.  A workgroup takes
// 256 frames of one wave's 63 text rows: the per-frame term C_w[y] + D_w[y] - log Z (three doubles) is worked out ONCE
// into LDS, then four rows at a time stream through -- 16-byte loads of alpha and beta, one exp2, a 16-byte store.
// (One thread per four cells with its own offsets read 64 bytes of offsets per 32 bytes of operands.)  CTC: the same with
// the softmax term 2^(x - n_y) in front (fwdsum_ctc_backward_sys_body's line).  Rows >= t_x, frames >= t_y, utterances
// without an alignment: 0.
template <bool CTC>
__device__ __forceinline__ void fwdsum_combine_body(const FwdSumParams &p, const float *nrm) {
    __shared__ float st[256], ny[256];
    const int b = blockIdx.z, w = blockIdx.y, tid = threadIdx.x;
    const int yb = blockIdx.x * 256;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const bool ok = tx >= 1 && tx <= ty;
    {
        const int y = yb + tid;
        float v = 0.f, n = 0.f;
        if (ok && y < ty) {
            const size_t oi = ((size_t)b * SY_NW_MAX + w) * p.NT + y;
            v = (float)(p.offs[oi] + p.doffs[oi] - p.logz[b]);
            if (CTC) n = nrm[(size_t)b * p.Ty + y];
        }
        st[tid] = v;
        if (CTC) ny[tid] = n;
    }
    __syncthreads();
    const int tq = tid & 63, rl = tid >> 6;
    const int y0 = yb + 4 * tq;
    if (y0 >= p.Ty) return;
    const int n4 = p.Ty - y0 < 4 ? p.Ty - y0 : 4;
    const int r1 = 63 * w + 63 < p.Tx ? 63 * w + 63 : p.Tx;
    const bool al16 = n4 == 4 && p.Ty % 4 == 0 &&
                      ((reinterpret_cast<uintptr_t>(p.alpha) | reinterpret_cast<uintptr_t>(p.grad) | reinterpret_cast<uintptr_t>(p.logp)) & 15) == 0;
#pragma unroll 2
    for (int r = 63 * w + rl; r < r1; r += 4) {
        const size_t o = ((size_t)b * p.Tx + r) * p.Ty + y0;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok && r < tx && y0 < ty) {
            float al[4], be[4], x[4] = {0.f, 0.f, 0.f, 0.f};
            if (al16) {
                const float4 a4 = *reinterpret_cast<const float4 *>(p.alpha + o), b4 = *reinterpret_cast<const float4 *>(p.grad + o);
                al[0] = a4.x; al[1] = a4.y; al[2] = a4.z; al[3] = a4.w;
                be[0] = b4.x; be[1] = b4.y; be[2] = b4.z; be[3] = b4.w;
                if (CTC) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(p.logp + o);
                    x[0] = x4.x; x[1] = x4.y; x[2] = x4.z; x[3] = x4.w;
                }
            } else {
                for (int j = 0; j < n4; ++j) { al[j] = p.alpha[o + j]; be[j] = p.grad[o + j]; if (CTC) x[j] = p.logp[o + j]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < n4 && y0 + j < ty) {
                    const float occ = __builtin_amdgcn_exp2f(al[j] + be[j] + st[4 * tq + j]);
                    g[j] = CTC ? __builtin_amdgcn_exp2f(fs_in(x[j]) - ny[4 * tq + j]) - occ : -occ;
                }
        }
        if (al16) *reinterpret_cast<float4 *>(p.grad + o) = make_float4(g[0], g[1], g[2], g[3]);
        else for (int j = 0; j < n4; ++j) p.grad[o + j] = g[j];
    }
}
__global__ __launch_bounds__(256) void fwdsum_combine_kernel(FwdSumParams p) { fwdsum_combine_body<false>(p, nullptr); }

// --------------------------------------------------------------------------
// The CTC form of the objective (the published one: OTA's ForwardSumLoss, README.md:21-25,50 -- a blank column
// at log-prob `blank` is put before the text, every frame is renormalised over blank + text (log_softmax), and
// the loss is the CTC loss of the token sequence 1..t_x): between two tokens, before the first and after the
// last a frame may be "blank".  States per token row r: B_r (blank before token r) and T_r (token r); row t_x
// holds the blank after the last token.
//     B_r(y) = blank  + lse( B_r(y-1), T_{r-1}(y-1) )
//     T_r(y) = x[r,y] + lse( T_r(y-1), B_r(y-1), T_{r-1}(y-1) )          start: B_0(-1) = log 1
//     Z = T_{tx-1}(ty-1) + B_tx(ty-1)
// The per-frame log_softmax normaliser n_y = lse(blank, x[.,y]) multiplies every path by the same factor, so the
// sweeps run on the raw scores and  loss = -(log Z - sum_y n_y);  d loss / d x[r,y] = softmax_y(r) - occupancy of
// T_r at y.  n_y comes from a row-parallel kernel of its own (ctc_colnorm_kernel).  Same one-sweeping-wave design
// and numerics (base-2 logs, FS_NEG, drift / re-basing into double offsets) as the kernels above; pinned to
// torch.nn.functional.ctc_loss in float64 (tests/test_objective.py) -- the one externally pinned row of 8f.
// --------------------------------------------------------------------------
__device__ __forceinline__ float fs_lae3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) +
                                     __builtin_amdgcn_exp2f(c - m));
}

struct CtcParams {
    FwdSumParams f;         // logp = the raw scores x; alpha = log2 alpha of the TOKEN states
    float *nrm;             // workspace [B,Ty]: n_y, base 2
    float blank2;           // blank score, base 2
    int fused_norm;         // the sweeps-side-by-side launch: the normalisers come from extra workgroups of that same launch
                            // (nobody needs them before the combining pass, which then also finishes the loss)
};

// n_y = log2( 2^blank + sum_{r < t_x} 2^x[r,y] ): 64 frames per workgroup, the rows dealt to its four waves (wave g takes
// rows g, g+4, ...: four loads in flight per thread, a running (max, sum) pair), the four partial pairs met through LDS.
// (One thread per frame over all rows was a chain of t_x dependent loads: 56 us at [64,200,1000], 1 TB/s.)
__device__ __forceinline__ void ctc_colnorm_body(const CtcParams &q, float (*sm)[64], float (*ss)[64], const int bx, const int b) {
    const FwdSumParams &p = q.f;
    const int fx = threadIdx.x & 63, g = threadIdx.x >> 6, y = bx * 64 + fx;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    float m = FS_NEG, s = 0.f;
    if (y < ty) {
        const float *col = p.logp + (size_t)b * p.Tx * p.Ty + y;
        int r = g;
        for (; r + 12 < tx; r += 16) {
            const float v0 = fs_in(col[(size_t)r * p.Ty]), v1 = fs_in(col[(size_t)(r + 4) * p.Ty]);
            const float v2 = fs_in(col[(size_t)(r + 8) * p.Ty]), v3 = fs_in(col[(size_t)(r + 12) * p.Ty]);
            const float mn = fmaxf(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)), m);
            s = s * __builtin_amdgcn_exp2f(m - mn) + ((__builtin_amdgcn_exp2f(v0 - mn) + __builtin_amdgcn_exp2f(v1 - mn)) +
                                                      (__builtin_amdgcn_exp2f(v2 - mn) + __builtin_amdgcn_exp2f(v3 - mn)));
            m = mn;
        }
        for (; r < tx; r += 4) {
            const float v = fs_in(col[(size_t)r * p.Ty]);
            const float mn = fmaxf(m, v);
            s = s * __builtin_amdgcn_exp2f(m - mn) + __builtin_amdgcn_exp2f(v - mn);
            m = mn;
        }
    }
    sm[g][fx] = m;
    ss[g][fx] = s;
    __syncthreads();
    if (g == 0 && y < p.Ty) {
        float M = q.blank2, S = 1.f;                                       // the blank column
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mk = sm[k][fx], sk = ss[k][fx];
            const float mn = fmaxf(M, mk);
            S = S * __builtin_amdgcn_exp2f(M - mn) + sk * __builtin_amdgcn_exp2f(mk - mn);
            M = mn;
        }
        q.nrm[(size_t)b * p.Ty + y] = M + __builtin_amdgcn_logf(S);
    }
}
__global__ __launch_bounds__(256) void ctc_colnorm_kernel(CtcParams q) {
    __shared__ float sm[4][64], ss[4][64];
    ctc_colnorm_body(q, sm, ss, blockIdx.x, blockIdx.y);
}

template <int R>
__global__ __launch_bounds__(FS_THREADS) void fwdsum_ctc_forward_kernel(CtcParams q) {
    const FwdSumParams &p = q.f;
    constexpr int TW = 128 / R, ROWS = 64 * R + 4;
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tin = fs_smem;                         // [2][TW][ROWS] scores, frame-major
    float *tout = tin + 2 * TW * ROWS;            // [2][TW][ROWS] alpha of the token states
    double *toff = reinterpret_cast<double *>(tout + 2 * TW * ROWS);   // [2][TW]
    float *tnrm = reinterpret_cast<float *>(toff + 2 * TW);            // [2][TW] n_y
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const bool sweeper = tid < 64;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    if (!(tx >= 1 && tx <= ty)) {                 // fewer frames than tokens: no labelling exists, loss = +inf
        if (tid == 0) { p.loss[b] = -FS_NEG_INF; p.logz[b] = (double)FS_NEG_INF; }
        for (int t = tid; t < p.NT; t += FS_THREADS) p.offs[(size_t)b * p.NT + t] = 0.0;
        return;
    }
    const int ntl = (ty + TW - 1) / TW;
    float pB[R], pT[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { pB[j] = FS_NEG; pT[j] = FS_NEG; }
    if (lane == 0) pB[0] = 0.f;                   // B_0 before the first frame: log 1
    float drift = 0.f;
    double C = 0.0, NS = 0.0;                     // NS: sum of the frames' normalisers
    for (int ph = 0; ph < ntl + 2; ++ph) {
        if (!sweeper) {
            const int s = tid - 64;
            if (ph < ntl) {
                float *dst = tin + (ph & 1) * TW * ROWS;
                const int y0 = ph * TW;
                constexpr int NEL = TW * 64 * R, NIT = (NEL + FS_STAGERS - 1) / FS_STAGERS;
                float v[NIT];
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    int e = s + i * FS_STAGERS;
                    e = e < NEL ? e : NEL - 1;
                    const int r = e / TW, c = e - r * TW;
                    const int rc = r < tx ? r : tx - 1, yc = y0 + c < ty ? y0 + c : ty - 1;
                    v[i] = p.logp[ubase + (size_t)rc * p.Ty + yc];
                }
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int e = s + i * FS_STAGERS;
                    if (e < NEL) { const int r = e / TW, c = e - r * TW; dst[c * ROWS + r] = fs_in(v[i]); }
                }
                if (s < TW) tnrm[(ph & 1) * TW + s] = (y0 + s < ty) ? q.nrm[(size_t)b * p.Ty + y0 + s] : 0.f;
            }
            if (ph >= 2) {
                const float *src = tout + (ph & 1) * TW * ROWS;
                const int y0 = (ph - 2) * TW;
                for (int e = s; e < TW * 64 * R; e += FS_STAGERS) {
                    const int r = e / TW, c = e - r * TW;
                    if (r < p.Tx && y0 + c < p.Ty) p.alpha[ubase + (size_t)r * p.Ty + y0 + c] = src[c * ROWS + r];
                }
                if (s < TW && y0 + s < ty) p.offs[(size_t)b * p.NT + y0 + s] = toff[(ph & 1) * TW + s];
            }
        } else if (ph >= 1 && ph <= ntl) {
            const int t = ph - 1, y0 = t * TW;
            const float *src = tin + (t & 1) * TW * ROWS + R * lane;
            float *dst = tout + (t & 1) * TW * ROWS + R * lane;
            for (int c = 0; c < TW; ++c) {
                const int y = y0 + c;
                if (y >= ty) break;                                                  // uniform: padding frames
                float x[R];
#pragma unroll
                for (int j = 0; j < R; j += 4) {
                    const float4 w = *reinterpret_cast<const float4 *>(src + c * ROWS + j);
                    x[j] = w.x; x[j + 1] = w.y; x[j + 2] = w.z; x[j + 3] = w.w;
                }
                NS += (double)tnrm[(t & 1) * TW + c];
                const float up0 = fs_from_lane_below(FS_NEG, pT[R - 1]);             // T of the row below this lane's first
                const float bl = q.blank2 - drift;
                float nB[R], nT[R];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int row = R * lane + j;
                    const float upT = j ? pT[j - 1] : up0;
                    const float vb = fs_lae2(pB[j], upT) + bl;
                    const float vt = fs_lae3(pT[j], pB[j], upT) + (x[j] - drift);
                    nB[j] = (row <= tx) ? fmaxf(vb, FS_NEG) : FS_NEG;
                    nT[j] = (row < tx) ? fmaxf(vt, FS_NEG) : FS_NEG;
                }
                C += (double)drift;
                if (lane == 0) toff[(t & 1) * TW + c] = C;
#pragma unroll
                for (int j = 0; j < R; j += 4)
                    *reinterpret_cast<float4 *>(dst + c * ROWS + j) = make_float4(nT[j], nT[j + 1], nT[j + 2], nT[j + 3]);
#pragma unroll
                for (int j = 0; j < R; ++j) { pB[j] = nB[j]; pT[j] = nT[j]; }
                if (y == ty - 1) {                                                  // uniform: Z = T_{tx-1} + B_tx
                    const float upn = fs_from_lane_below(FS_NEG, pT[R - 1]);
#pragma unroll
                    for (int j = 0; j < R; ++j)
                        if (R * lane + j == tx) {
                            const double lz = (double)fs_lae2(pB[j], j ? pT[j - 1] : upn) + C;     // log2 Z of the raw scores
                            p.logz[b] = lz;
                            p.loss[b] = (float)(-(lz - NS) * FS_LN2);
                        }
                }
                if ((c & (FS_RB - 1)) == FS_RB - 1) {
                    float m = fmaxf(pB[0], pT[0]);
#pragma unroll
                    for (int j = 1; j < R; ++j) m = fmaxf(m, fmaxf(pB[j], pT[j]));
                    m = fs_wave_max_dpp(m);
                    if (m < 0.5f * FS_NEG) m = 0.f;
                    C += (double)m;
                    drift += m * (1.0f / FS_RB);
#pragma unroll
                    for (int j = 0; j < R; ++j) { pB[j] = fmaxf(pB[j] - m, FS_NEG); pT[j] = fmaxf(pT[j] - m, FS_NEG); }
                }
            }
        }
        fs_lds_barrier();
    }
}

template <int R>
__global__ __launch_bounds__(FS_THREADS) void fwdsum_ctc_backward_kernel(CtcParams q) {
    const FwdSumParams &p = q.f;
    constexpr int TW = 64 / R, ROWS = 64 * R + 4;
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tlp = fs_smem;                         // [2][TW][ROWS] scores
    float *tal = tlp + 2 * TW * ROWS;             // [2][TW][ROWS] alpha of the token states (relative to C_y)
    float *tgr = tal + 2 * TW * ROWS;             // [2][TW][ROWS] gradient out
    double *toff = reinterpret_cast<double *>(tgr + 2 * TW * ROWS);    // [2][TW] C_y
    float *tnrm = reinterpret_cast<float *>(toff + 2 * TW);            // [2][TW] n_y
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const bool sweeper = tid < 64;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const bool ok = tx >= 1 && tx <= ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    const int ntl = ok ? (ty + TW - 1) / TW : 0;
    for (int r = 0; r < p.Tx; ++r)
        for (int y = ntl * TW + tid; y < p.Ty; y += FS_THREADS) p.grad[ubase + (size_t)r * p.Ty + y] = 0.f;
    if (!ok) return;
    const double logz = p.logz[b];
    float gB[R], gT[R];                           // beta + emission of frame y+1, relative to D
#pragma unroll
    for (int j = 0; j < R; ++j) { gB[j] = FS_NEG; gT[j] = FS_NEG; }
    float drift = 0.f;
    double D = 0.0;
    for (int ph = 0; ph < ntl + 2; ++ph) {
        if (!sweeper) {
            const int s = tid - 64;
            if (ph < ntl) {
                const int t = ntl - 1 - ph, y0 = t * TW;
                float *dlp = tlp + (ph & 1) * TW * ROWS, *dal = tal + (ph & 1) * TW * ROWS;
                constexpr int NEL = TW * 64 * R, NIT = (NEL + FS_STAGERS - 1) / FS_STAGERS;
                float v[NIT], w[NIT];
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    int e = s + i * FS_STAGERS;
                    e = e < NEL ? e : NEL - 1;
                    const int r = e / TW, c = e - r * TW;
                    const int rc = r < tx ? r : tx - 1, yc = y0 + c < ty ? y0 + c : ty - 1;
                    v[i] = p.logp[ubase + (size_t)rc * p.Ty + yc];
                    w[i] = p.alpha[ubase + (size_t)rc * p.Ty + yc];
                }
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    const int e = s + i * FS_STAGERS;
                    if (e < NEL) {
                        const int r = e / TW, c = e - r * TW;
                        dlp[c * ROWS + r] = fs_in(v[i]);
                        dal[c * ROWS + r] = w[i];
                    }
                }
                if (s < TW) {
                    toff[(ph & 1) * TW + s] = (y0 + s < ty) ? p.offs[(size_t)b * p.NT + y0 + s] : 0.0;
                    tnrm[(ph & 1) * TW + s] = (y0 + s < ty) ? q.nrm[(size_t)b * p.Ty + y0 + s] : 0.f;
                }
            }
            if (ph >= 2) {
                const float *src = tgr + (ph & 1) * TW * ROWS;
                const int y0 = (ntl - 1 - (ph - 2)) * TW;
                for (int e = s; e < TW * 64 * R; e += FS_STAGERS) {
                    const int r = e / TW, c = e - r * TW;
                    if (r < p.Tx && y0 + c < p.Ty) p.grad[ubase + (size_t)r * p.Ty + y0 + c] = src[c * ROWS + r];
                }
            }
        } else if (ph >= 1 && ph <= ntl) {
            const int buf = (ph - 1) & 1, y0 = (ntl - 1 - (ph - 1)) * TW;
            const float *slp = tlp + buf * TW * ROWS + R * lane, *sal = tal + buf * TW * ROWS + R * lane;
            float *dst = tgr + buf * TW * ROWS + R * lane;
            for (int c = TW - 1; c >= 0; --c) {
                const int y = y0 + c;
                if (y >= ty) {                                                       // uniform: padding frames
#pragma unroll
                    for (int j = 0; j < R; j += 4) *reinterpret_cast<float4 *>(dst + c * ROWS + j) = make_float4(0.f, 0.f, 0.f, 0.f);
                    continue;
                }
                float x[R], al[R];
#pragma unroll
                for (int j = 0; j < R; j += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(slp + c * ROWS + j);
                    const float4 w = *reinterpret_cast<const float4 *>(sal + c * ROWS + j);
                    x[j] = v.x; x[j + 1] = v.y; x[j + 2] = v.z; x[j + 3] = v.w;
                    al[j] = w.x; al[j + 1] = w.y; al[j + 2] = w.z; al[j + 3] = w.w;
                }
                const float st = (float)(toff[buf * TW + c] + D - logz);            // uniform
                const float ny = tnrm[buf * TW + c];
                const float aB = fs_from_lane_above(FS_NEG, gB[0]);                  // row above this lane's last one
                const float aT = fs_from_lane_above(FS_NEG, gT[0]);
                float nB[R], nT[R], gr[R];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int row = R * lane + j;
                    float bT, bB;
                    if (y == ty - 1) {                                               // uniform branch
                        bT = (row == tx - 1) ? 0.f : FS_NEG;
                        bB = (row == tx) ? 0.f : FS_NEG;
                    } else {
                        bT = fs_lae3(gT[j], j + 1 < R ? gB[j + 1] : aB, j + 1 < R ? gT[j + 1] : aT);
                        bB = fs_lae2(gB[j], gT[j]);
                    }
                    if (row >= tx) bT = FS_NEG;
                    if (row > tx) bB = FS_NEG;
                    const float occ = __builtin_amdgcn_exp2f(al[j] + bT + st);       // 2^(-1e30) = 0
                    gr[j] = (row < tx) ? __builtin_amdgcn_exp2f(x[j] - ny) - occ : 0.f;
                    nT[j] = fmaxf(bT + (x[j] - drift), FS_NEG);
                    nB[j] = fmaxf(bB + (q.blank2 - drift), FS_NEG);
                }
                D += (double)drift;
#pragma unroll
                for (int j = 0; j < R; j += 4)
                    *reinterpret_cast<float4 *>(dst + c * ROWS + j) = make_float4(gr[j], gr[j + 1], gr[j + 2], gr[j + 3]);
#pragma unroll
                for (int j = 0; j < R; ++j) { gB[j] = nB[j]; gT[j] = nT[j]; }
                if ((c & (FS_RB - 1)) == 0) {
                    float m = fmaxf(gB[0], gT[0]);
#pragma unroll
                    for (int j = 1; j < R; ++j) m = fmaxf(m, fmaxf(gB[j], gT[j]));
                    m = fs_wave_max_dpp(m);
                    if (m < 0.5f * FS_NEG) m = 0.f;
                    D += (double)m;
                    drift += m * (1.0f / FS_RB);
#pragma unroll
                    for (int j = 0; j < R; ++j) { gB[j] = fmaxf(gB[j] - m, FS_NEG); gT[j] = fmaxf(gT[j] - m, FS_NEG); }
                }
            }
        }
        fs_lds_barrier();
    }
}

// --------------------------------------------------------------------------
// The CTC form on the systolic pipeline (T_text + 1 <= 252 / 504 rows): wave w owns rows 63w .. 63w+62, one per lane,
// with the two states of a row (blank before the token, the token) in two registers -- their log-sums are
// independent chains of one lane, so a frame costs about what the plain form's costs.  Only TOKEN states cross a
// lane or a wave (B_r reads T_{r-1}; T_r reads T_{r-1}), so the forward ghost lane replays the upper wave's last token
// state exactly as in fwdsum_forward_sys_kernel; backward, a token state reads BOTH states of the row below, which
// the lower wave leaves per frame in a two-float ring.  Same alpha (token states) / per-(wave, frame) offsets in the
// workspace, same numerics as the one-wave kernels above.
// --------------------------------------------------------------------------
template <int SY_NW, int SY_TW>
__device__ __forceinline__ void fwdsum_ctc_forward_sys_body(const CtcParams &q, const int b) {
    const FwdSumParams &p = q.f;
    // (tiles slot-major at a pitch of TW + 4 floats, stagers issued by hand, one re-basing per tile, offsets linear inside
    // it: everything fwdsum_forward_sys_body says; what differs is the frame's arithmetic)
    constexpr int PITCH = SY_TW + 4, SY_TILE = 64 * PITCH, RB = SY_TW;
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tin = fs_smem;                                     // [NW][2][64][PITCH] scores (slot = lane)
    float *tout = tin + SY_NW * 2 * SY_TILE;                  // [NW][2][64][PITCH] alpha of the token states
    double *toff = reinterpret_cast<double *>(tout + SY_NW * 2 * SY_TILE);   // [NW][2][TW] C_w per frame
    double *tcg = toff + SY_NW * 2 * SY_TW;                   // [NW][2] C_w at the start of the tile ...
    double *tns = tcg + SY_NW * 2;                            // [64] partial sums of the frames' normalisers
    float *tdr = reinterpret_cast<float *>(tns + 64);         // [NW][2] ... and the tile's drift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & (SY_NW - 1);
    const bool sweeper = wave < SY_NW;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    double *offs = p.offs + ((size_t)b * SY_NW_MAX + w) * p.NT;
    if (!(tx >= 1 && tx <= ty)) {                             // fewer frames than tokens: no labelling exists, loss = +inf
        if (tid == 0) { p.loss[b] = -FS_NEG_INF; p.logz[b] = (double)FS_NEG_INF; }
        if (!sweeper) for (int t = lane; t < p.NT; t += 64) offs[t] = 0.0;
        return;
    }
    const int ntl = (ty + SY_TW - 1) / SY_TW, nph = ntl + SY_NW + 1;
    const int row = 63 * w + lane - 1;                        // sweeper: lane 0 is the ghost (row 63w-1)
    const bool ghost = lane == 0;
    if (!sweeper) {
        if (w == 0 && !q.fused_norm) {                        // the normalisers' sum, 64 partial sums (read in the last phase)
            double ns = 0.0;
            for (int y = lane; y < ty; y += 64) ns += (double)q.nrm[(size_t)b * p.Ty + y];
            tns[lane] = ns;
        }
        const bool by_hand = p.Ty % 4 == 0 && ((reinterpret_cast<uintptr_t>(p.logp) | reinterpret_cast<uintptr_t>(p.alpha)) & 15) == 0 &&
                             (size_t)p.Tx * p.Ty * sizeof(float) < (1ull << 32);
        if (by_hand) {
            fs_stager_by_hand<SY_TW, false>(p.logp + ubase, p.alpha + ubase, offs, tin + w * 2 * SY_TILE, tout + w * 2 * SY_TILE,
                                            toff + w * 2 * SY_TW, w, w, lane, tx, ty, p.Tx, p.Ty, ntl, nph);
            return;
        }
        // any T_mel / alignment: one tile in flight, scheduled by the compiler (a phase is then a memory round trip)
        float vnext[SY_TW];
        auto stage_issue = [&](int tl) {
            const int tc = tl < ntl ? tl : ntl - 1;
            const int y0 = tc * SY_TW;
#pragma unroll
            for (int i = 0; i < SY_TW; ++i) {
                const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;     // slot r, frame c
                int rg = 63 * w + r - 1;
                rg = rg < 0 ? 0 : (rg < tx ? rg : tx - 1);
                const int yc = y0 + c < ty ? y0 + c : ty - 1;
                vnext[i] = p.logp[ubase + (size_t)rg * p.Ty + yc];
            }
        };
        stage_issue(0);
        for (int ph = 0; ph < nph; ++ph) {
            const int tl = ph - w, ts = ph - 2 - w;
            if (tl >= 0 && tl < ntl) {
                float *dst = tin + (w * 2 + (tl & 1)) * SY_TILE;
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) {
                    const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                    dst[r * PITCH + c] = (63 * w + r - 1 < tx) ? fs_nat(vnext[i]) : FS_NEG_NAT;   // rows past the text: log 0
                }
                stage_issue(tl + 1);
            }
            if (ts >= 0 && ts < ntl) {
                const float *src = tout + (w * 2 + (ts & 1)) * SY_TILE;
                const int y0 = ts * SY_TW;
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) {
                    const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                    const int rg = 63 * w + r - 1;
                    if (r >= 1 && rg < p.Tx && y0 + c < p.Ty) p.alpha[ubase + (size_t)rg * p.Ty + y0 + c] = src[r * PITCH + c];
                }
                if (lane < SY_TW && y0 + lane < ty) offs[y0 + lane] = toff[(w * 2 + (ts & 1)) * SY_TW + lane];
            }
            fs_lds_barrier();
        }
        return;
    }
    // ---- sweepers.  A frame: L = lse(B, T of the row above) serves both states --
    //     B' = blank + L,   T' = x + lse(T, L)
    // four transcendentals where the two independent sums took six.  No select per row: rows past the text were staged as
    // log 0 (the token state dies there), and a blank state is alive exactly where its inputs are (row t_x: from the last
    // token; below it nothing is).
    float pT = FS_NEG, pB = (row == 0) ? 0.f : FS_NEG;        // B_0 before the first frame: log 1
    float drift = 0.f;
    double Cg = 0.0;
    for (int ph = 0; ph < nph; ++ph) {
        const int t = ph - 1 - w;
        if (t >= 0 && t < ntl) {
            const int y0 = t * SY_TW, buf = t & 1;
            const float4 *src4 = reinterpret_cast<const float4 *>(tin + (w * 2 + buf) * SY_TILE + lane * PITCH);
            float4 *dst4 = reinterpret_cast<float4 *>(tout + (w * 2 + buf) * SY_TILE + lane * PITCH);
            const int wu = w ? w - 1 : 0;                     // the wave above: its last row's token state, its (Cg, drift)
            const float4 *ring4 = reinterpret_cast<const float4 *>(tout + (wu * 2 + buf) * SY_TILE + 63 * PITCH);
            float xv[SY_TW], rgv[SY_TW], av[4];
#pragma unroll
            for (int i = 0; i < SY_TW / 4; ++i) {
                const float4 l4 = src4[i], r4 = ring4[i];
                xv[4 * i] = l4.x; xv[4 * i + 1] = l4.y; xv[4 * i + 2] = l4.z; xv[4 * i + 3] = l4.w;
                rgv[4 * i] = w ? r4.x : FS_NEG; rgv[4 * i + 1] = w ? r4.y : FS_NEG;
                rgv[4 * i + 2] = w ? r4.z : FS_NEG; rgv[4 * i + 3] = w ? r4.w : FS_NEG;
            }
            const double scg = tcg[wu * 2 + buf];
            const float sdr = tdr[wu * 2 + buf];
            const double mcg = Cg;
            const float mdr = drift;
            const float D0 = w ? (float)(scg - Cg) : 0.f, dl = w ? sdr - drift : 0.f;
            const float bl = q.blank2 - drift;
            if (lane == 0) { tcg[w * 2 + buf] = mcg; tdr[w * 2 + buf] = mdr; }           // (at the tile's start: see the plain form)
            if (lane < SY_TW) toff[(w * 2 + buf) * SY_TW + lane] = mcg + (double)(lane + 1) * (double)mdr;
            auto frames = [&](auto tail) {
                constexpr bool TAIL = decltype(tail)::value;
#pragma unroll
                for (int c = 0; c < SY_TW; ++c) {
                    const int y = y0 + c, k = c + 1;
                    const float upT = fs_from_lane_below(FS_NEG, pT);
                    const float L = fs_lae2(pB, upT);
                    float vb = L + bl;
                    float vt = fs_lae2(pT, L) + fmaf(xv[c], FS_LOG2E, -drift);       // (the staged score is a natural log)
                    if (TAIL && y >= ty) { vb = FS_NEG; vt = FS_NEG; }                  // uniform
                    const float gh = rgv[c] + (D0 + (float)k * dl);                    // the sender's value, on this wave's offset
                    const float nT = ghost ? gh : vt;
                    pB = ghost ? FS_NEG : vb;
                    pT = nT;
                    av[c & 3] = nT;
                    if ((c & 3) == 3) dst4[c >> 2] = make_float4(av[0], av[1], av[2], av[3]);
                    if (TAIL && y == ty - 1) {                                           // uniform: Z = T_{tx-1} + B_tx
                        const float below = fs_from_lane_below(FS_NEG, pT);
                        if (row == tx && !ghost) {
                            const double lz = (double)fs_lae2(pB, below) + (Cg + (double)k * (double)drift);   // log2 Z of the raw scores
                            p.logz[b] = lz;
                            if (!q.fused_norm) {                                         // (else: fwdsum_ctc_combine_kernel)
                                double ns = 0.0;
                                for (int j = 0; j < 64; ++j) ns += tns[j];
                                p.loss[b] = (float)(-(lz - ns) * FS_LN2);
                            }
                        }
                    }
                }
            };
            if (y0 + SY_TW < ty) frames(std::false_type{});
            else                 frames(std::true_type{});
            {   // re-base on the column's maximum, learn the per-frame drift
                float mx = fs_wave_max_dpp(fmaxf(pT, pB));
                if (mx < 0.5f * FS_NEG) mx = 0.f;
                Cg += (double)RB * (double)drift + (double)mx;
                drift += mx * (1.0f / RB);
                pT = fmaxf(pT - mx, FS_NEG);
                pB = fmaxf(pB - mx, FS_NEG);
            }
        }
        fs_lds_barrier();
    }
}

// (fused_norm: workgroups beyond the B sweeping ones work out the frames' normalisers -- the sweep never reads them, the
// loss does: fwdsum_ctc_loss_kernel behind this launch)
template <int SY_NW, int SY_TW>
__global__ __launch_bounds__(2 * SY_NW * 64) void fwdsum_ctc_forward_sys_kernel(CtcParams q) {
    if (q.fused_norm && (int)blockIdx.x >= q.f.B) {
        extern __shared__ __attribute__((aligned(16))) float fs_smem[];
        if (threadIdx.x >= 256) return;
        const int ncx = (q.f.Ty + 63) / 64, cb = (int)blockIdx.x - q.f.B;
        ctc_colnorm_body(q, reinterpret_cast<float (*)[64]>(fs_smem), reinterpret_cast<float (*)[64]>(fs_smem + 256), cb % ncx, cb / ncx);
        return;
    }
    fwdsum_ctc_forward_sys_body<SY_NW, SY_TW>(q, blockIdx.x);
}

// BETA_ONLY as in fwdsum_backward_sys_body: the token states' beta (relative to D_w[y], which goes to p.doffs) leaves
// through the gradient tile into p.grad; fwdsum_ctc_combine_kernel makes the gradient of it.
template <int SY_NW, int SY_TW, bool BETA_ONLY>
__device__ __forceinline__ void fwdsum_ctc_backward_sys_body(const CtcParams &q, const int b) {
    const FwdSumParams &p = q.f;
    constexpr int PITCH = SY_TW + 4, SY_TILE = 64 * PITCH, SY_THREADS = 2 * SY_NW * 64, RB = SY_TW;
    extern __shared__ __attribute__((aligned(16))) float fs_smem[];
    float *tlp = fs_smem;                                     // [NW][2][64][PITCH] scores (slot = lane)
    float *tal = tlp + SY_NW * 2 * SY_TILE;                   // alpha of the token states (relative to C_w)
    float *tgr = tal + SY_NW * 2 * SY_TILE;                   // gradient out
    double *toff = reinterpret_cast<double *>(tgr + SY_NW * 2 * SY_TILE);   // [NW][2][TW] C_w per frame
    double *tdof = toff + SY_NW * 2 * SY_TW;                  // [NW][2][TW] D_w per frame
    double *tdg = tdof + SY_NW * 2 * SY_TW;                   // [NW][2] D_w at the start of the tile ...
    float2 *tg = reinterpret_cast<float2 *>(tdg + SY_NW * 2); // [NW][2][TW] (g_B, g_T) of a wave's FIRST row
    float2 *dump = tg + SY_NW * 2 * SY_TW;                    // [NW][64] where the other lanes write
    float *tnrm = reinterpret_cast<float *>(dump + SY_NW * 64);            // [NW][2][TW] n_y
    float *tdr = tnrm + SY_NW * 2 * SY_TW;                    // [NW][2] ... and the tile's drift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wave & (SY_NW - 1), wr = SY_NW - 1 - w;     // wave SY_NW-1 (the last rows) leads
    const bool sweeper = wave < SY_NW;
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    const bool ok = tx >= 1 && tx <= ty;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    const int ntl = ok ? (ty + SY_TW - 1) / SY_TW : 0, nph = ntl + SY_NW + 1;
    if (!BETA_ONLY)
        for (int r = 0; r < p.Tx; ++r)
            for (int y = ntl * SY_TW + tid; y < p.Ty; y += SY_THREADS) p.grad[ubase + (size_t)r * p.Ty + y] = 0.f;
    if (!ok) return;
    const double logz = BETA_ONLY ? 0.0 : p.logz[b];
    const double *offs = p.offs + ((size_t)b * SY_NW_MAX + w) * p.NT;
    double *doffs = p.doffs + ((size_t)b * SY_NW_MAX + w) * p.NT;
    const int row = 63 * w + lane;                            // sweeper: lane 63 is the ghost (row 63w+63)
    const bool ghost = lane == 63;
    if (!sweeper) {
        const bool by_hand = BETA_ONLY && p.Ty % 4 == 0 &&
                             ((reinterpret_cast<uintptr_t>(p.logp) | reinterpret_cast<uintptr_t>(p.grad)) & 15) == 0 &&
                             (size_t)p.Tx * p.Ty * sizeof(float) < (1ull << 32);
        if (by_hand) {
            fs_stager_by_hand<SY_TW, true>(p.logp + ubase, p.grad + ubase, doffs, tlp + w * 2 * SY_TILE, tgr + w * 2 * SY_TILE,
                                           tdof + w * 2 * SY_TW, wr, w, lane, tx, ty, p.Tx, p.Ty, ntl, nph);
            return;
        }
        float vnext[SY_TW], unext[SY_TW];
        double onext = 0.0;
        float nnext = 0.f;
        auto stage_issue = [&](int kl) {                      // tile counted from the end, clamped
            const int kc = kl < ntl ? kl : ntl - 1;
            const int y0 = (ntl - 1 - kc) * SY_TW;
#pragma unroll
            for (int i = 0; i < SY_TW; ++i) {
                const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                int rg = 63 * w + r;
                rg = rg < tx ? rg : tx - 1;
                const int yc = y0 + c < ty ? y0 + c : ty - 1;
                vnext[i] = p.logp[ubase + (size_t)rg * p.Ty + yc];
                if (!BETA_ONLY) unext[i] = p.alpha[ubase + (size_t)rg * p.Ty + yc];
            }
            const int yo = y0 + (lane & (SY_TW - 1));
            if (!BETA_ONLY) {
                onext = offs[yo < ty ? yo : ty - 1];
                nnext = q.nrm[(size_t)b * p.Ty + (yo < ty ? yo : ty - 1)];
            }
        };
        stage_issue(0);
        for (int ph = 0; ph < nph; ++ph) {
            const int kl = ph - wr, ks = ph - 2 - wr;         // tile counted from the end
            if (kl >= 0 && kl < ntl) {
                const int t = ntl - 1 - kl, y0 = t * SY_TW;
                float *dlp = tlp + (w * 2 + (kl & 1)) * SY_TILE, *dal = tal + (w * 2 + (kl & 1)) * SY_TILE;
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) {
                    const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                    dlp[r * PITCH + c] = (63 * w + r < tx) ? fs_nat(vnext[i]) : FS_NEG_NAT;   // rows past the text: log 0
                    if (!BETA_ONLY) dal[r * PITCH + c] = unext[i];
                }
                if (!BETA_ONLY && lane < SY_TW) {
                    toff[(w * 2 + (kl & 1)) * SY_TW + lane] = (y0 + lane < ty) ? onext : 0.0;
                    tnrm[(w * 2 + (kl & 1)) * SY_TW + lane] = (y0 + lane < ty) ? nnext : 0.f;
                }
                stage_issue(kl + 1);
            }
            if (ks >= 0 && ks < ntl) {
                const float *src = tgr + (w * 2 + (ks & 1)) * SY_TILE;
                const int y0 = (ntl - 1 - ks) * SY_TW;
#pragma unroll
                for (int i = 0; i < SY_TW; ++i) {
                    const int e = lane + 64 * i, r = e / SY_TW, c = e - r * SY_TW;
                    const int rg = 63 * w + r;
                    if (r < 63 && rg < p.Tx && y0 + c < p.Ty) p.grad[ubase + (size_t)rg * p.Ty + y0 + c] = src[r * PITCH + c];
                }
                if (BETA_ONLY && lane < SY_TW && y0 + lane < ty) doffs[y0 + lane] = tdof[(w * 2 + (ks & 1)) * SY_TW + lane];
            }
            fs_lds_barrier();
        }
        return;
    }
    // ---- sweepers (no select per row: see the forward kernel; below row t_x every state stays log 0 by itself) ----
    float gT = FS_NEG, gB = FS_NEG;                           // beta + emission of frame y+1, relative to D
    float drift = 0.f;
    double Dg = 0.0;
    for (int ph = 0; ph < nph; ++ph) {
        const int kt = ph - 1 - wr;
        if (kt >= 0 && kt < ntl) {
            const int buf = kt & 1, y0 = (ntl - 1 - kt) * SY_TW;
            const float4 *slp4 = reinterpret_cast<const float4 *>(tlp + (w * 2 + buf) * SY_TILE + lane * PITCH);
            const float4 *sal4 = reinterpret_cast<const float4 *>(tal + (w * 2 + buf) * SY_TILE + lane * PITCH);
            float4 *dgr4 = reinterpret_cast<float4 *>(tgr + (w * 2 + buf) * SY_TILE + lane * PITCH);
            const double *myoff = toff + (w * 2 + buf) * SY_TW;
            const float *mynrm = tnrm + (w * 2 + buf) * SY_TW;
            float2 *myg = (lane == 0) ? tg + (w * 2 + buf) * SY_TW : dump + w * 64 + lane;
            const int mgs = lane == 0 ? 1 : 0;                // (frames advance in the ring only)
            const bool has = w + 1 < SY_NW;
            const int wb = has ? w + 1 : w;                   // the wave below: its first row's states, its (Dg, drift)
            const float2 *ring = tg + (wb * 2 + buf) * SY_TW;
            float xv[SY_TW], alv[SY_TW], nyv[SY_TW], ov[4];
            float2 rgv[SY_TW];
            double cov[SY_TW];
#pragma unroll
            for (int i = 0; i < SY_TW / 4; ++i) {
                const float4 l4 = slp4[i];
                xv[4 * i] = l4.x; xv[4 * i + 1] = l4.y; xv[4 * i + 2] = l4.z; xv[4 * i + 3] = l4.w;
                if (!BETA_ONLY) {
                    const float4 a4 = sal4[i];
                    alv[4 * i] = a4.x; alv[4 * i + 1] = a4.y; alv[4 * i + 2] = a4.z; alv[4 * i + 3] = a4.w;
                }
            }
#pragma unroll
            for (int c = 0; c < SY_TW; ++c) {
                rgv[c] = has ? ring[c] : make_float2(FS_NEG, FS_NEG);
                if (!BETA_ONLY) { cov[c] = myoff[c]; nyv[c] = mynrm[c]; }
            }
            const double sdg = tdg[wb * 2 + buf];
            const float sdr = tdr[wb * 2 + buf];
            const double mdg = Dg;
            const float mdr = drift;
            const float D0 = has ? (float)(sdg - Dg) : 0.f, dl = has ? sdr - drift : 0.f;
            const double Dlz = Dg - logz;
            const float bl = q.blank2 - drift;
            if (lane == 0) { tdg[w * 2 + buf] = mdg; tdr[w * 2 + buf] = mdr; }           // (at the tile's start: see the plain form)
            if (lane < SY_TW) tdof[(w * 2 + buf) * SY_TW + lane] = mdg + (double)(SY_TW - lane) * (double)mdr;
            auto frames = [&](auto tail) {
                constexpr bool TAIL = decltype(tail)::value;
#pragma unroll
                for (int c = SY_TW - 1; c >= 0; --c) {
                    const int y = y0 + c, k = SY_TW - c;                               // k-th frame of the tile
                    if (TAIL && y >= ty) {                                           // uniform: padding frames (all log 0, drift 0)
                        ov[c & 3] = 0.f;
                        if ((c & 3) == 0) dgr4[c >> 2] = make_float4(ov[0], ov[1], ov[2], ov[3]);
                        myg[mgs * c] = make_float2(FS_NEG, FS_NEG);
                        continue;
                    }
                    const float x = xv[c];
                    float bT, bB;
                    if (TAIL && y == ty - 1) {                                       // uniform branch
                        bT = (row == tx - 1) ? 0.f : FS_NEG;
                        bB = (row == tx) ? 0.f : FS_NEG;
                    } else {
                        // beta_B(r) = lse(g_B(r), g_T(r)) -- and that is also everything a token state of the row ABOVE can
                        // continue into below itself: beta_T(r) = lse(g_T(r), beta_B(r+1)).  Four transcendentals a frame
                        // (the three-way sum over the neighbour's two states took six), one lane shift instead of two.
                        bB = fs_lae2(gB, gT);
                        bT = fs_lae2(gT, fs_from_lane_above(FS_NEG, bB));
                    }
                    if (BETA_ONLY) {
                        ov[c & 3] = bT - drift;                       // relative to the D of this frame (Dg + k * drift)
                    } else {
                        const float st = (float)(cov[c] + Dlz) + (float)(k - 1) * drift;   // C_w[y] + D_w - log Z
                        const float occ = __builtin_amdgcn_exp2f(alv[c] + bT + st);  // 2^(-1e30) = 0
                        ov[c & 3] = __builtin_amdgcn_exp2f(fmaf(x, FS_LOG2E, -nyv[c])) - occ;   // (rows past the text: 0 - 0)
                    }
                    if ((c & 3) == 0) dgr4[c >> 2] = make_float4(ov[0], ov[1], ov[2], ov[3]);
                    const float conv = D0 + (float)k * dl;
                    const float nT = ghost ? rgv[c].y + conv : bT + fmaf(x, FS_LOG2E, -drift);
                    const float nB = ghost ? rgv[c].x + conv : bB + bl;
                    myg[mgs * c] = make_float2(nB, nT);
                    gT = nT;
                    gB = nB;
                }
            };
            if (kt != 0) frames(std::false_type{});
            else         frames(std::true_type{});
            {
                float mx = fs_wave_max_dpp(fmaxf(gT, gB));
                if (mx < 0.5f * FS_NEG) mx = 0.f;
                Dg += (double)RB * (double)drift + (double)mx;
                drift += mx * (1.0f / RB);
                gT = fmaxf(gT - mx, FS_NEG);
                gB = fmaxf(gB - mx, FS_NEG);
            }
        }
        fs_lds_barrier();
    }
}

template <int SY_NW, int SY_TW>
__global__ __launch_bounds__(2 * SY_NW * 64) void fwdsum_ctc_backward_sys_kernel(CtcParams q) {
    fwdsum_ctc_backward_sys_body<SY_NW, SY_TW, false>(q, blockIdx.x);
}

// the CTC form's sweeps side by side (see fwdsum_both_sys_kernel) ...
// Workgroups beyond the 2B sweeping ones work out the frames' normalisers on the CUs the sweeps leave idle (the sweeps
// never read them: the combining pass does, and finishes the loss).
template <int SY_NW, int SY_TW>
__global__ __launch_bounds__(2 * SY_NW * 64) void fwdsum_ctc_both_sys_kernel(CtcParams q) {
    if ((int)blockIdx.x >= 2 * q.f.B) {
        extern __shared__ __attribute__((aligned(16))) float fs_smem[];
        if (threadIdx.x >= 256) return;
        const int ncx = (q.f.Ty + 63) / 64, cb = (int)blockIdx.x - 2 * q.f.B;
        ctc_colnorm_body(q, reinterpret_cast<float (*)[64]>(fs_smem), reinterpret_cast<float (*)[64]>(fs_smem + 256), cb % ncx, cb / ncx);
        return;
    }
    const int b = blockIdx.x >> 1;
    if (blockIdx.x & 1) fwdsum_ctc_backward_sys_body<SY_NW, SY_TW, true>(q, b);
    else                fwdsum_ctc_forward_sys_body<SY_NW, SY_TW>(q, b);
}

// ... and its gradient, in place over the token states' beta: softmax over blank + text of the frame minus the token's
// occupancy (fwdsum_combine_body<true>)
// loss = -(log Z - sum of the frames' normalisers), summed in a fixed order (256 partial sums)
__device__ __forceinline__ void fwdsum_ctc_finish_loss(const CtcParams &q, const int b) {
    const FwdSumParams &p = q.f;
    __shared__ double part[256];
    int tx = p.t_xs[b], ty = p.t_ys[b];
    tx = tx > p.Tx ? p.Tx : tx;
    ty = ty > p.Ty ? p.Ty : ty;
    double ns = 0.0;
    for (int y = threadIdx.x; y < ty; y += 256) ns += (double)q.nrm[(size_t)b * p.Ty + y];
    part[threadIdx.x] = ns;
    __syncthreads();
    if (threadIdx.x == 0 && tx >= 1 && tx <= ty) {
        double t = 0.0;
        for (int j = 0; j < 256; ++j) t += part[j];
        p.loss[b] = (float)(-(p.logz[b] - t) * FS_LN2);
    }
    __syncthreads();
}
__global__ __launch_bounds__(256) void fwdsum_ctc_loss_kernel(CtcParams q) { fwdsum_ctc_finish_loss(q, blockIdx.x); }
__global__ __launch_bounds__(256) void fwdsum_ctc_combine_kernel(CtcParams q) {
    if (q.fused_norm && blockIdx.x == 0 && blockIdx.y == 0) fwdsum_ctc_finish_loss(q, blockIdx.z);
    fwdsum_combine_body<true>(q.f, q.nrm);
}

struct FsLayout { size_t alpha_off, offs_off, logz_off, doffs_off, total; int NT, R; };

static FsLayout fs_layout(int B, int Tx, int Ty) {
    FsLayout L;
    L.R = Tx <= 256 ? 4 : (Tx <= 512 ? 8 : 16);   // text rows per lane of the sweeping wave
    L.NT = Ty;                                    // one offset per frame
    L.alpha_off = 0;
    L.offs_off = align_up((size_t)B * Tx * Ty * sizeof(float), 256);
    L.logz_off = L.offs_off + align_up((size_t)B * SY_NW_MAX * L.NT * sizeof(double), 256);   // per (wave, frame)
    L.doffs_off = L.logz_off + align_up((size_t)B * sizeof(double), 256);
    L.total = L.doffs_off + align_up((size_t)B * SY_NW_MAX * L.NT * sizeof(double), 256);   // backward offsets, sweeps side by side
    return L;
}

template <int R>
static int fs_launch(const FwdSumParams &p, bool backward, hipStream_t s) {
    constexpr int ROWS = 64 * R + 4;
    const size_t lds_f = (size_t)4 * (128 / R) * ROWS * sizeof(float) + 2 * (128 / R) * sizeof(double);
    const size_t lds_b = (size_t)6 * (64 / R) * ROWS * sizeof(float) + 2 * (64 / R) * sizeof(double);
    auto kf = fwdsum_forward_kernel<R>;
    auto kb = fwdsum_backward_kernel<R>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kf), lds_f));
    hipLaunchKernelGGL(kf, dim3(p.B), dim3(FS_THREADS), lds_f, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    if (backward) {
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kb), lds_b));
        hipLaunchKernelGGL(kb, dim3(p.B), dim3(FS_THREADS), lds_b, s, p);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    return ALIGNER_OK;
}

// Both sweeps in one launch pay while all 2B workgroups can be resident at once, one per CU (tools/fwdsum_batch_sweep.py at
// [B,200,1000], side by side / one after the other: B = 96 117 / 187 us, 104 125 / 188, 112 169 / 198, 128 183 / 214; CTC form
// 187 / 266 ... 258 / 275 -- past ~7/8 of the CUs some workgroup waits for a CU and the launch stretches, but it still wins).
static bool fs_side_by_side(int B) { return 2 * B <= device_cu_count(); }

template <int SY_NW, int SY_TW>
static int fs_launch_sys(const FwdSumParams &p, bool backward, hipStream_t s) {
    constexpr int SY_THREADS = 2 * SY_NW * 64;
    constexpr size_t grp = (size_t)SY_NW * 2 * (SY_TW / FS_RB) * (sizeof(double) + sizeof(float));   // (offset, drift) per group
    constexpr size_t tile = (size_t)64 * (SY_TW + 4) * sizeof(float);                                // slot-major, pitch TW + 4
    const size_t lds_f = 2 * SY_NW * 2 * tile + (size_t)SY_NW * 2 * SY_TW * sizeof(double) + grp;
    const size_t lds_b = 3 * SY_NW * 2 * tile + (size_t)SY_NW * (2 * SY_TW + 64 * 4) * sizeof(float) +
                         (size_t)2 * SY_NW * 2 * SY_TW * sizeof(double) + grp;
    auto kf = fwdsum_forward_sys_kernel<SY_NW, SY_TW>;
    const bool gradhand = backward && p.Ty % 4 == 0 && !p.no_grad_stager &&
                          ((reinterpret_cast<uintptr_t>(p.logp) | reinterpret_cast<uintptr_t>(p.grad) |
                            reinterpret_cast<uintptr_t>(p.alpha)) & 15) == 0 &&
                          (size_t)p.Tx * p.Ty * sizeof(float) < (1ull << 32);
    auto kb = gradhand ? fwdsum_backward_sys_kernel<SY_NW, SY_TW, true> : fwdsum_backward_sys_kernel<SY_NW, SY_TW, false>;
    // with the gradient, on a batch that leaves half the CUs idle: both sweeps in one launch, then the combining pass
    if (backward && g_opt_fwdsum_serial <= 0 && fs_side_by_side(p.B) && p.doffs) {
        auto k2 = fwdsum_both_sys_kernel<SY_NW, SY_TW>;
        const size_t lds = lds_f > lds_b ? lds_f : lds_b;
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(k2), lds));
        hipLaunchKernelGGL(k2, dim3(2 * p.B), dim3(SY_THREADS), lds, s, p);
        ALIGNER_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(fwdsum_combine_kernel, dim3((p.Ty + 255) / 256, (p.Tx + 62) / 63, p.B), dim3(256), 0, s, p);
        ALIGNER_HIP_CHECK(hipGetLastError());
        return ALIGNER_OK;
    }
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kf), lds_f));
    hipLaunchKernelGGL(kf, dim3(p.B), dim3(SY_THREADS), lds_f, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    if (backward) {
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kb), lds_b));
        hipLaunchKernelGGL(kb, dim3(p.B), dim3(SY_THREADS), lds_b, s, p);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    return ALIGNER_OK;
}

template <int R>
static int fs_launch_ctc(const CtcParams &q, bool backward, hipStream_t s) {
    constexpr int ROWS = 64 * R + 4;
    const size_t lds_f = (size_t)4 * (128 / R) * ROWS * sizeof(float) + 2 * (128 / R) * (sizeof(double) + sizeof(float));
    const size_t lds_b = (size_t)6 * (64 / R) * ROWS * sizeof(float) + 2 * (64 / R) * (sizeof(double) + sizeof(float));
    hipLaunchKernelGGL(ctc_colnorm_kernel, dim3((q.f.Ty + 63) / 64, q.f.B), dim3(256), 0, s, q);
    ALIGNER_HIP_CHECK(hipGetLastError());
    auto kf = fwdsum_ctc_forward_kernel<R>;
    auto kb = fwdsum_ctc_backward_kernel<R>;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kf), lds_f));
    hipLaunchKernelGGL(kf, dim3(q.f.B), dim3(FS_THREADS), lds_f, s, q);
    ALIGNER_HIP_CHECK(hipGetLastError());
    if (backward) {
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kb), lds_b));
        hipLaunchKernelGGL(kb, dim3(q.f.B), dim3(FS_THREADS), lds_b, s, q);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    return ALIGNER_OK;
}

template <int SY_NW, int SY_TW>
static int fs_launch_ctc_sys(const CtcParams &q, bool backward, hipStream_t s) {
    constexpr int SY_THREADS = 2 * SY_NW * 64;
    constexpr size_t tile = (size_t)64 * (SY_TW + 4) * sizeof(float);                                // slot-major, pitch TW + 4
    const size_t lds_f = 2 * SY_NW * 2 * tile + (size_t)SY_NW * 2 * SY_TW * sizeof(double) + (size_t)SY_NW * 2 * sizeof(double) +
                         64 * sizeof(double) + (size_t)SY_NW * 2 * sizeof(float);
    const size_t lds_b = 3 * SY_NW * 2 * tile + (size_t)2 * SY_NW * 2 * SY_TW * sizeof(double) + (size_t)SY_NW * 2 * sizeof(double) +
                         (size_t)SY_NW * 2 * SY_TW * (sizeof(float2) + sizeof(float)) + (size_t)SY_NW * 64 * sizeof(float2) +
                         (size_t)SY_NW * 2 * sizeof(float);
    auto kf = fwdsum_ctc_forward_sys_kernel<SY_NW, SY_TW>;
    auto kb = fwdsum_ctc_backward_sys_kernel<SY_NW, SY_TW>;
    if (backward && g_opt_fwdsum_serial <= 0 && fs_side_by_side(q.f.B) && q.f.doffs) {
        auto k2 = fwdsum_ctc_both_sys_kernel<SY_NW, SY_TW>;
        const size_t lds = lds_f > lds_b ? lds_f : lds_b;
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(k2), lds));
        CtcParams qf = q;
        qf.fused_norm = 1;
        const unsigned ncol = (unsigned)((q.f.Ty + 63) / 64) * (unsigned)q.f.B;     // normaliser workgroups, behind the sweeps
        hipLaunchKernelGGL(k2, dim3(2 * q.f.B + ncol), dim3(SY_THREADS), lds, s, qf);
        ALIGNER_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(fwdsum_ctc_combine_kernel, dim3((q.f.Ty + 255) / 256, (q.f.Tx + 62) / 63, q.f.B), dim3(256), 0, s, qf);
        ALIGNER_HIP_CHECK(hipGetLastError());
        return ALIGNER_OK;
    }
    if (!backward && g_opt_fwdsum_serial <= 0) {
        // the loss alone: the normalisers from extra workgroups of the sweep's launch, the loss finished behind it
        CtcParams qf = q;
        qf.fused_norm = 1;
        const unsigned ncol = (unsigned)((q.f.Ty + 63) / 64) * (unsigned)q.f.B;
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kf), lds_f));
        hipLaunchKernelGGL(kf, dim3(q.f.B + ncol), dim3(SY_THREADS), lds_f, s, qf);
        ALIGNER_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(fwdsum_ctc_loss_kernel, dim3(q.f.B), dim3(256), 0, s, qf);
        ALIGNER_HIP_CHECK(hipGetLastError());
        return ALIGNER_OK;
    }
    hipLaunchKernelGGL(ctc_colnorm_kernel, dim3((q.f.Ty + 63) / 64, q.f.B), dim3(256), 0, s, q);
    ALIGNER_HIP_CHECK(hipGetLastError());
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kf), lds_f));
    hipLaunchKernelGGL(kf, dim3(q.f.B), dim3(SY_THREADS), lds_f, s, q);
    ALIGNER_HIP_CHECK(hipGetLastError());
    if (backward) {
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kb), lds_b));
        hipLaunchKernelGGL(kb, dim3(q.f.B), dim3(SY_THREADS), lds_b, s, q);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    return ALIGNER_OK;
}

}  // namespace aligner

using namespace aligner;

extern "C" {

size_t aligner_forward_sum_workspace_bytes(int B, int Tx, int Ty) {
    if (B < 0 || Tx < 1 || Ty < 1 || Tx > 1024) return 0;
    return fs_layout(B, Tx, Ty).total;
}

int aligner_forward_sum_f32(const float *logp, const int32_t *t_xs, const int32_t *t_ys, float *loss_out,
                            float *grad_out, void *workspace, size_t workspace_bytes, int B, int Tx, int Ty,
                            void *stream) {
    if (!logp || !t_xs || !t_ys || !loss_out || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (Tx > 1024) return fail(ALIGNER_EDOM, "Tx=%d exceeds 1024 text rows", Tx);
    if (B == 0) return ALIGNER_OK;
    const FsLayout L = fs_layout(B, Tx, Ty);
    if (workspace_bytes < L.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, L.total);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    FwdSumParams p{logp, t_xs, t_ys, reinterpret_cast<float *>(ws + L.alpha_off),
                   reinterpret_cast<double *>(ws + L.offs_off), reinterpret_cast<double *>(ws + L.logz_off),
                   loss_out, grad_out, B, Tx, Ty, L.NT, reinterpret_cast<double *>(ws + L.doffs_off), g_debug_stamps, g_opt_fwdsum_no_grad_stager};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool bwd = grad_out != nullptr;
    if (!g_opt_fwdsum_one_wave) {
        if (Tx <= 63 * 4) return fs_launch_sys<4, 16>(p, bwd, s);
        if (Tx <= 63 * 8) return fs_launch_sys<8, 8>(p, bwd, s);
    }
    if (L.R == 4) return fs_launch<4>(p, bwd, s);
    if (L.R == 8) return fs_launch<8>(p, bwd, s);
    return fs_launch<16>(p, bwd, s);
}

size_t aligner_forward_sum_ctc_workspace_bytes(int B, int Tx, int Ty) {
    if (B < 0 || Tx < 1 || Ty < 1 || Tx > 1023) return 0;
    return fs_layout(B, Tx, Ty).total + align_up((size_t)B * Ty * sizeof(float), 256);
}

int aligner_forward_sum_ctc_f32(const float *scores, const int32_t *t_xs, const int32_t *t_ys, float blank_logprob,
                                float *loss_out, float *grad_out, void *workspace, size_t workspace_bytes, int B, int Tx,
                                int Ty, void *stream) {
    if (!scores || !t_xs || !t_ys || !loss_out || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (Tx > 1023) return fail(ALIGNER_EDOM, "Tx=%d exceeds 1023 text rows (the blank after the last token takes a row)", Tx);
    if (!(blank_logprob - blank_logprob == 0.0f)) return fail(ALIGNER_EINVAL, "blank_logprob must be finite");
    if (B == 0) return ALIGNER_OK;
    if (B > 65535) return fail(ALIGNER_EDOM, "grid too large");
    const FsLayout L = fs_layout(B, Tx, Ty);
    const size_t total = L.total + align_up((size_t)B * Ty * sizeof(float), 256);
    if (workspace_bytes < total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, total);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    CtcParams q{{scores, t_xs, t_ys, reinterpret_cast<float *>(ws + L.alpha_off), reinterpret_cast<double *>(ws + L.offs_off),
                 reinterpret_cast<double *>(ws + L.logz_off), loss_out, grad_out, B, Tx, Ty, L.NT,
                 reinterpret_cast<double *>(ws + L.doffs_off)},
                reinterpret_cast<float *>(ws + L.total), blank_logprob * FS_LOG2E, 0};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool bwd = grad_out != nullptr;
    // rows 0..t_x: the blank after the last token needs a row of its own
    if (!g_opt_fwdsum_one_wave) {
        if (Tx + 1 <= 63 * 4) return fs_launch_ctc_sys<4, 16>(q, bwd, s);
        if (Tx + 1 <= 63 * 8) return fs_launch_ctc_sys<8, 8>(q, bwd, s);
    }
    if (Tx + 1 <= 256) return fs_launch_ctc<4>(q, bwd, s);
    if (Tx + 1 <= 512) return fs_launch_ctc<8>(q, bwd, s);
    return fs_launch_ctc<16>(q, bwd, s);
}

}  // extern "C"
