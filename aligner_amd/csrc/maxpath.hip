// Monotonic alignment search on MI355X (gfx950): forward sweep + backtrack.
//
// Replaces, bit-for-bit on the integer path, the reference's
//   maximum_path_each  (monotonic_align/core.pyx:7-35)
//   maximum_path_c     (monotonic_align/core.pyx:38-45)
// and the marshalling of monotonic_align/__init__.py:11-21, with the score
// tensor resident in HBM.  See DESIGN.md 3 (and DESIGN_HISTORY.md 3 for the derivation); the short form:
//
//  * One workgroup per utterance (grid = batch), like the reference's prange.
//  * The recurrence Q[x,y] = max(Q[x,y-1], Q[x-1,y-1]) + value[x,y] only couples
//    a mel frame to the previous one, so a frame is one parallel step: text rows
//    live on lanes (63 rows per wave + one ghost lane that replays the row above),
//    the running column Q stays in a VGPR and the "row above" operand is a DPP
//    wave-shift fused into v_max_f32 -- never LDS.
//  * Waves of a workgroup form a systolic pipeline over 32-frame tiles: in phase
//    p wave w sweeps tile p-w; the last row of wave w-1 reaches wave w's ghost
//    lane through a 128-byte LDS ring slot per tile; one s_barrier per phase.
//  * The score tensor is row-major with the mel axis contiguous, so a lane-per-
//    row frame read is strided.  Dedicated loader waves (one per compute wave)
//    stream it with coalesced 16-byte loads, DEPTH tiles in flight in registers,
//    and transpose through a padded LDS tile that the compute lanes read back
//    conflict-free with ds_read_b128 (4 frames per read).
//  * Q is never stored.  Each cell leaves one decision bit
//        dec = (x == y) | (Q[x-1,y-1] > Q[x,y-1])
//    which is exactly the reference's backtrack predicate (core.pyx:34); 32
//    frames make one word per lane, kept in LDS when the utterance fits.  The
//    backtrack walks text rows (not frames): row x ending at frame e starts at the
//    highest set bit <= e of its own bit string.
//  * Finite scores take a hand-scheduled sweep built on v_max_f32 (4 VALU issues per
//    frame).  v_max differs from the reference's `(a > b) ? a : b` only when a NaN
//    is involved, and a NaN can only arise from a non-finite input, so the loader
//    waves watch for non-finite scores and the utterance is redone with the exact
//    compare/select sweep if one is seen.
//
// Only cells inside the reference's band (core.pyx:18) influence the result;
// cells outside it are either skipped (whole tiles) or computed and ignored --
// they are provably never read by an in-band cell.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "aligner_amd.h"
#include "common.h"
#include "maxpath_sweep_asm.inc"
#include "maxpath_walk_asm.inc"

namespace aligner {

constexpr int TC       = 32;  // frames per tile == decision bits per word
constexpr int TILE_LD  = 36;  // dwords per LDS tile row: 32 + 4 pad -> 16B-slot stride 9 (odd)
constexpr int RING_T   = 4;   // boundary ring depth in tiles
constexpr int RING_LD  = 64;  // floats per ring slot: 32 used, padded so that all 64 lanes can store unmasked
constexpr int RPW      = 63;  // text rows per compute wave (lane 0 is the ghost lane)
constexpr int WS_HDR_BYTES = 256;
constexpr int MV_CHUNKS = 32;  // mask_verify_kernel: pieces (row ranges) per utterance, one flag word each (eight pieces: 3.3 TB/s)

enum Mode { MODE_NORMAL = 0, MODE_EMPTY = 1, MODE_COMPAT = 2 };

// LDS-qualified float: keeps per-lane selected addresses as ds_* instructions
// (a generic pointer would turn them into flat_* accesses).
typedef __attribute__((address_space(3))) float lds_float;
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;    // 16-byte aligned: ds_read_b128

struct MaxpathParams {
    const void  *value;     // scores [B,Tx,Ty]: fp32, or bf16 / fp16 (template VT) up-cast on the fly
    const void  *mask;      // strict-mask operand (nullable)
    const int   *t_xs;
    const int   *t_ys;
    int         *starts;    // [B,Tx+1] first frame of every token (workspace)
    int         *tok;       // [B,Ty] nullable
    int         *dur;       // [B,Tx] nullable
    unsigned    *bits;      // [B,NT,ROWS] decision words in global memory
    int         *status;
    int         *mflag;     // [B][MV_CHUNKS] nonzero = that piece of the utterance's mask is not all ones inside [0,t_x) x [0,t_y)
    int          verify;    // the launch's zero workgroups also verify p.mask (strict mask, optimistic search: see the kernel)
    int          redo;      // MASKMODE 1: only the utterances a verdict in mflag names, their earlier ones erased first
    int          ldv;       // row pitch of `value` (and of `mask`) in elements: Ty, or more when the producer padded its rows to whole
                            // 128-byte lines (aligner_maxpath_ld)
    int B, Tx, Ty, NT, ROWS;
    int WT;                 // tiles per backtrack window when the words live in global memory
    int bits_in_lds;        // pipelined kernel: decision words stay in LDS
    int lds_bits_off;       // byte offset of that LDS region
    int lds_prev_off;       // byte offset of the "previous start" table P beside it (0: not kept)
    int force_exact;        // skip the v_max sweep (non-finite max_neg_val)
    float neg;
    int flags;
    float *qout;            // ALIGNER_F_WRITE_Q: the running scores Q go back into the score block, as core.pyx:30 (nullable)
    int lds_total;          // bytes of LDS the launch asked for (store_outputs: is there room for its scratch?)
    unsigned    *xring;     // two workgroups per utterance: [B][NT][32] boundary row between the halves (workspace, 0xFF-filled)
    int         *xflag;     // ... [B] first half done (1; 2 = it met a non-finite score), 0xFFFFFFFF until then
    unsigned    *xwalk;     // ... [B] the backtrack's hand-over from the second half to the first (0xFFFFFFFF until then)
    int          split_walk;    // ... each half walks its own rows out of its own LDS (its decision words all fit there)
    unsigned long long *stamps;   // debug: [B][16 waves][16] shader-clock stamps (nullable; [2B] when utterances are split)
    float       *dump;      // (the fused similarity -> search experiment's: tools/experiments/fused_align.patch) unused
    void        *path1;     // ALIGNER_F_PATH_PREZEROED: the caller's all-zero dense path; the kernel writes its ones (nullable)
    int          path1_es;  // ... element size in bytes (1, 2, 4, 8)
    unsigned long long path1_one;   // ... the bits of a 1 in that dtype
    int          zero_blocks;   // the launch's first `zero_blocks` workgroups write the path's zeros while the others search (0: none)
    int          zero_nt;       // ... with non-temporal stores
    unsigned long long zero_n16;    // ... 16-byte pieces of the path
    int         *zsync;         // ... workspace: [0] zero workgroups done, [1] workgroups finished (both 0 between launches)
};

// Development aid: lane 0 of every wave drops a shader-clock stamp (slot 6/7 the
// 100 MHz wall clock at entry/exit) when a stamp buffer was installed with
// aligner_debug_set_stamps().  One uniform branch per stamp when disabled.
#define ALIGNER_STAMP(k)                                                                        \
    do {                                                                                        \
        if (p.stamps && (threadIdx.x & 63) == 0)                                                \
            p.stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (k)] =                \
                ((k) == 6 || (k) == 7) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();   \
    } while (0)

constexpr int ZERO_SPIN_LIMIT = 1 << 22;           // polls of the zero-fill count before ALIGNER_ST_INTERNAL (seconds)

// --------------------------------------------------------------------------
// small device helpers
// --------------------------------------------------------------------------
// Score element types (template VT).  The reference computes in fp32 whatever the input dtype
// (__init__.py:14 astype(np.float32)); 16-bit scores are up-cast exactly, in the loaders.
enum { VT_F32 = 0, VT_BF16 = 1, VT_F16 = 2 };

template <int VT> __device__ __forceinline__ float half_to_f32(unsigned h16) {          // h16: the 16 bits, zero-extended
    if (VT == VT_BF16) return __builtin_bit_cast(float, h16 << 16);
    return (float)__builtin_bit_cast(_Float16, (unsigned short)h16);
}
// value * mask as torch evaluates it for two tensors of this dtype (__init__.py:11): fp32 product, rounded to
// nearest-even in the tensors' dtype; then the fp32 up-cast of __init__.py:14
template <int VT> __device__ __forceinline__ float mul_in_dtype(float v, float m) {
    const float r = v * m;
    if (VT == VT_BF16) return (float)(__bf16)r;
    if (VT == VT_F16) return (float)(_Float16)r;
    return r;
}
template <int VT> __device__ __forceinline__ float load_score(const void *base, size_t idx) {
    if (VT == VT_F32) return static_cast<const float *>(base)[idx];
    return half_to_f32<VT>(static_cast<const unsigned short *>(base)[idx]);
}
__device__ __forceinline__ float dpp_wave_shr1(float old_lane0, float src) {
    // lane i <- src[lane i-1]; lane 0 keeps `old_lane0` (bound_ctrl off).
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old_lane0),
                                           __builtin_bit_cast(int, src),
                                           0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

__device__ __forceinline__ float comp(const float4 &v, int i) {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;   // folds once the frame loop is unrolled
}

// Lengths as the reference uses them, made memory-safe.  Uniform per block.
__device__ __forceinline__ int classify_lengths(const MaxpathParams &p, int b, int &tx, int &ty) {
    tx = p.t_xs[b];
    ty = p.t_ys[b];
    int st = 0;
    if (tx > p.Tx) { tx = p.Tx; st |= ALIGNER_ST_CLAMPED; }
    if (ty > p.Ty) { ty = p.Ty; st |= ALIGNER_ST_CLAMPED; }
    int mode;
    if (tx >= 1 && tx <= ty) {
        mode = MODE_NORMAL;
    } else if (ty <= 0) {
        mode = MODE_EMPTY;                      // both reference loops are empty (core.pyx:17,32)
        ty = 0;
        if (tx < 0) tx = 0;
    } else if (tx > ty && (p.flags & ALIGNER_F_COMPAT_TXGTTY)) {
        mode = MODE_COMPAT;                     // reference: raw-score backtrack from row t_x-1 (write_degenerate)
    } else {
        mode = MODE_EMPTY;
        st |= ALIGNER_ST_BAD_LENGTHS;
        if (tx < 0) tx = 0;
    }
    if (st && threadIdx.x == 0) atomicOr(p.status, st);
    return mode;
}

// Is tile t (frames 32t..32t+31) inside the band of any of rows [r0, r0+nrows)?
// Band of row x: x <= y <= t_y - t_x + x  (core.pyx:18).
__device__ __forceinline__ bool tile_in_band(int t, int r0, int nrows, int tx, int ty) {
    const int y_lo = r0;
    const int y_hi = ty - tx + r0 + nrows - 1;
    return (TC * t + TC - 1 >= y_lo) && (TC * t <= y_hi);
}

// Token starts -> the caller's outputs.  starts[x] = first frame of token x for x < t_x and t_y for t_x <= x <= Tx,
// so durations are plain differences.  A frame's token is the number of tokens 1 .. t_x-1 that start at or before
// it: the starts are marked in a bit string over the frames (LDS scratch behind the starts), a single wave turns
// the words' population counts into running totals, and a frame costs one masked popcount -- no search.  (Without
// room for the bit string -- a very long mel axis on a small workgroup -- each frame bisects the starts.)
__device__ __forceinline__ int starts_words_of(int Tx) { return ((Tx + 1 + 63) / 64) * 64 + 64; }

// Row / frame ranges (the two-workgroup form, each half storing what it walked): rows [x_lo, x_hi) of `starts` and
// `dur` (x_hi < 0: through the padding, Tx + 1 / Tx), frames [y_lo, y_hi) of `tok` and the path (y_hi < 0: Ty); startsL
// must hold rows x_lo .. x_hi (the first row after the range bounds its last token).
__device__ __forceinline__ void store_outputs(const MaxpathParams &p, int b, int tx, int ty, int *startsL,
                                              bool distinct_starts = true, int x_lo = 0, int x_hi = -1, int y_lo = 0,
                                              int y_hi = -1) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    if (p.path1 && p.zero_blocks > 0) {
        // The zeros of the dense path come from the launch's zero workgroups: all of them must have reported before a 1 is
        // written (a 1 written earlier could be wiped by a zero store still on its way).  The wait is bounded; a
        // workgroup that gives up leaves a DEFINED result -- no 1 anywhere in its block (which the zero workgroups, who
        // never wait for anybody, turn into all zeros whenever they do run), zero durations, no token on any frame,
        // ALIGNER_ST_INTERNAL in the status word -- never ones that a late zero workgroup may or may not erase.
        int gave_up = 0;
        if (tid == 0) {
            const int limit = (p.flags & ALIGNER_F_TEST_DROP_ZERO_REPORTS) ? (1 << 10) : ZERO_SPIN_LIMIT;
            int spins = 0;
            while (__hip_atomic_load(p.zsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.zero_blocks) {   // (relaxed: see the kernel)
                if (++spins > limit) { atomicOr(p.status, ALIGNER_ST_INTERNAL); gave_up = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (__syncthreads_or(gave_up)) {
            for (int x = tid; x <= p.Tx; x += nthreads) startsL[x] = 0;
            __syncthreads();
            tx = 0;
            ty = 0;
        }
    }
    const int xs_hi = x_hi < 0 ? p.Tx + 1 : x_hi, xd_hi = x_hi < 0 ? p.Tx : x_hi;
    const int ye = y_hi < 0 ? p.Ty : y_hi;
    for (int x = x_lo + tid; x < xs_hi; x += nthreads) p.starts[(size_t)b * (p.Tx + 1) + x] = startsL[x];
    if (p.dur)
        for (int x = x_lo + tid; x < xd_hi; x += nthreads) p.dur[(size_t)b * p.Tx + x] = startsL[x + 1] - startsL[x];
    if (!p.tok && !p.path1) return;
    // core.pyx:33 `path[index, y] = 1` on a path the caller zeroed (np.zeros, __init__.py:15): one element per frame
    auto mark_path = [&](int t, int y) {
        const size_t idx = ((size_t)b * p.Tx + t) * p.Ty + y;
        switch (p.path1_es) {
            case 1: static_cast<unsigned char *>(p.path1)[idx] = (unsigned char)p.path1_one; break;
            case 2: static_cast<unsigned short *>(p.path1)[idx] = (unsigned short)p.path1_one; break;
            case 4: static_cast<unsigned *>(p.path1)[idx] = (unsigned)p.path1_one; break;
            default: static_cast<unsigned long long *>(p.path1)[idx] = p.path1_one; break;
        }
    };
    const int nmw = (ty + 31) >> 5;                        // words of the bit string
    unsigned *mark = reinterpret_cast<unsigned *>(startsL + starts_words_of(p.Tx));
    int *before = reinterpret_cast<int *>(mark + nmw);    // set bits in the words before word j
    // (the t_x > t_y compatibility result has rows that own no frame: equal starts, so it takes the search)
    const bool room = distinct_starts && (size_t)(starts_words_of(p.Tx) + 2 * nmw) * 4 <= (size_t)p.lds_total;
    const int xt_hi = (x_hi < 0 || x_hi > tx) ? tx : x_hi;  // tokens x_lo .. xt_hi-1 own the frames of the range
    if (room) {
        for (int j = tid; j < nmw; j += nthreads) mark[j] = 0u;
        __syncthreads();
        for (int x = x_lo + 1 + tid; x < xt_hi; x += nthreads) {
            const int st = startsL[x];                      // strictly increasing in x: every bit is set once
            atomicOr(&mark[st >> 5], 1u << (st & 31));
        }
        __syncthreads();
        if (tid < 64) {
            const int per = (nmw + 63) >> 6, j0 = tid * per;
            int mine = 0;
            for (int j = j0; j < j0 + per && j < nmw; ++j) mine += __builtin_popcount(mark[j]);
            int run = mine;                                 // inclusive scan over the 64 lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(run, d);
                if (tid >= d) run += up;
            }
            run -= mine;
            for (int j = j0; j < j0 + per && j < nmw; ++j) { before[j] = run; run += __builtin_popcount(mark[j]); }
        }
        __syncthreads();
        for (int y = y_lo + tid; y < ye; y += nthreads) {
            int t = -1;
            if (y < ty) t = x_lo + before[y >> 5] + __builtin_popcount(mark[y >> 5] & ((2u << (y & 31)) - 1u));
            if (p.tok) p.tok[(size_t)b * p.Ty + y] = t;
            if (p.path1 && t >= 0) mark_path(t, y);
        }
    } else {
        for (int y = y_lo + tid; y < ye; y += nthreads) {
            int t = -1;
            if (y < ty) {
                int lo = x_lo, hi = xt_hi - 1;            // last x with starts[x] <= y
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (startsL[mid] <= y) lo = mid; else hi = mid - 1;
                }
                t = lo;
            }
            if (p.tok) p.tok[(size_t)b * p.Ty + y] = t;
            if (p.path1 && t >= 0) mark_path(t, y);
        }
    }
}

// Outputs for the degenerate modes (whole block, uniform): no ones at all, or the reference's t_x > t_y
// result.  For t_x > t_y the reference's forward band is empty (core.pyx:18: x_lo = t_x + y - t_y > y),
// so `value` stays raw and its backtrack (core.pyx:32-35) walks up from row t_x-1 comparing RAW scores:
//     index -= 1  iff  index != 0 and (index == y or value[index,y-1] < value[index-1,y-1]).
// The comparison at y == 0 reads one float before each row (wraparound off) but happens after the last
// path write, so the output is a function of in-bounds data only; it is reproduced here exactly (one
// thread, t_y dependent steps: a correctness path).  The walk is monotone, so the result still has the
// token-start form: rows the walk never reaches own no frame.
template <int MASKMODE, int VT>
__device__ __forceinline__ void write_degenerate(const MaxpathParams &p, int b, int mode, int tx, int ty, int *startsL) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    for (int x = tid; x <= p.Tx; x += nthreads) startsL[x] = (mode == MODE_COMPAT && x >= tx) ? ty : 0;
    __syncthreads();
    if (mode == MODE_COMPAT && tid == 0) {
        const size_t ub = (size_t)b * p.Tx * p.ldv;
        int index = tx - 1;
        for (int y = ty - 1; y >= 1; --y) {
            if (index == 0) break;
            bool up = (index == y);
            if (!up) {
                const size_t ia = ub + (size_t)index * p.ldv + (y - 1), ic = ub + (size_t)(index - 1) * p.ldv + (y - 1);
                float a = load_score<VT>(p.value, ia), c = load_score<VT>(p.value, ic);
                if (MASKMODE == 1) {
                    a = mul_in_dtype<VT>(a, load_score<VT>(p.mask, ia));
                    c = mul_in_dtype<VT>(c, load_score<VT>(p.mask, ic));
                }
                up = a < c;
            }
            if (up) { startsL[index] = y; index -= 1; }        // row `index` owns frames from y on
        }
        // rows 0..index: `index` owns frames [0, start of index+1), the rows above it nothing
    }
    __syncthreads();
    store_outputs(p, b, tx, mode == MODE_COMPAT ? ty : 0, startsL, false);   // no frame has a token in the empty modes
}

// --------------------------------------------------------------------------
// Backtrack over decision words (shared by both forward kernels).
//
// Word (tile t, row x) holds the decisions of text row x for frames 32t..32t+31, frame 32t+c at bit
// 31-c.  Wave 0 walks rows from t_x-1 down: row x ending at frame e starts at the highest set bit <= e
// of its own bit string (the move at that frame is the reference's `index -= 1`, core.pyx:34-35).
// Lane j holds the row's word of window tile j; the walk state e lives in an SGPR.
//
// The walk is a chain of t_x dependent steps on ONE wave, so its time is (cycles per step) x t_x and a
// step is priced by what sits on the chain (tools/microbench_walk.hip, gfx950): a dependent SALU op 4.1
// cycles, a v_readlane on the chain +27 (the VALU -> SGPR -> SALU crossing; the same off the chain: the
// wave issues in order), s_cmp + s_cbranch +17..21 even when not taken, v_writelane 9.  The fast step
// therefore has no branch at all (walk_row_fast):
//     je = e >> 5;  vcc = {readlane(w, je-1) : readlane(w, je)}   the row's words of e's tile and the one before
//     vcc >>= 31 - (e & 31)                       frame e at bit 0, frame e-k at bit k; SCC = (vcc != 0)
//     k  = ff1(vcc)                               distance back to the row's first frame, -1 if none
//     e' = e - k - SCC                            s_subb: (first frame) - 1 = last frame of the row above
//     flag |= k                                   bit 31 <=> some row of the group found no bit
//     startv[lane K] = e'
// 67 cycles a row (the branching form it replaced: 138), and a group of 16 rows is checked once: if a row's
// first frame lay more than a tile before e's tile (a token of 33+ frames) the group is redone from its saved
// entry state by walk_rows_slow, which searches all the row's earlier words with a ballot.
//
// LDS overlay once the forward sweep is done: startsL at offset 0, then (words in global memory only) a
// window of WT tiles x row_pitch(ROWS) words.
// --------------------------------------------------------------------------
__host__ __device__ constexpr int row_pitch(int ROWS) { return ROWS + 4; }   // 4 x odd words: the walker's 16-byte
                                                                            // reads (lane = tile) are conflict-free

// One chunk of fast steps: see maxpath_walk_asm.inc (generated by tools/gen_walk_asm.py).  `lds_addr` = this
// lane's LDS byte address of (window tile = lane, row 64c); rows kent .. 0 of the chunk are walked.
template <bool WINDOW>
__device__ __forceinline__ void walk_chunk_fast(unsigned lds_addr, int kent, int &e, int &flag, int &startv) {
    int je, jm, t, k;
    // (no-ops on values the compiler already knows to be uniform; they keep every "s" operand provably scalar)
    e = __builtin_amdgcn_readfirstlane(e);
    flag = __builtin_amdgcn_readfirstlane(flag);
    const int skip = __builtin_amdgcn_readfirstlane(63 - kent);
    if (WINDOW)
        asm volatile(ALIGNER_WALK64_W
                     : [e] "+s"(e), [flag] "+s"(flag), [sv] "+v"(startv), [je] "=&s"(je), [jm] "=&s"(jm), [t] "=&s"(t),
                       [k] "=&s"(k)
                     : [addr] "v"(lds_addr), [skip] "s"(skip)
                     : ALIGNER_WALK64_CLOBBERS, "memory");
    else
        asm volatile(ALIGNER_WALK64
                     : [e] "+s"(e), [flag] "+s"(flag), [sv] "+v"(startv), [je] "=&s"(je), [jm] "=&s"(jm), [t] "=&s"(t),
                       [k] "=&s"(k)
                     : [addr] "v"(lds_addr), [skip] "s"(skip)
                     : ALIGNER_WALK64_CLOBBERS, "memory");
}

// The never-failing form over (W, P) tables (ALIGNER_WALK64_P): rows kent .. 0 of the chunk, no flag.
__device__ __forceinline__ void walk_chunk_prev(unsigned lds_addr, unsigned lds_paddr, int kent, int &e, int &startv) {
    int je, jm, t, k;
    e = __builtin_amdgcn_readfirstlane(e);
    const int skip = __builtin_amdgcn_readfirstlane(63 - kent);
    asm volatile(ALIGNER_WALK64_P
                 : [e] "+s"(e), [sv] "+v"(startv), [je] "=&s"(je), [jm] "=&s"(jm), [t] "=&s"(t), [k] "=&s"(k)
                 : [addr] "v"(lds_addr), [paddr] "v"(lds_paddr), [skip] "s"(skip)
                 : ALIGNER_WALK64_P_CLOBBERS, "memory");
}

// Rows xhi .. xlo (descending) with every case handled: the row's word of e's tile first, then a ballot over
// its earlier words in the window.  `wrow` = this lane's (= window tile's) words, indexed by row.  A row whose
// frame e lies below the window, or whose start is not in it, stops the walk: the next (earlier) window
// resumes at row x with the same e.  Returns with (x, e) = the next row to walk and its last frame.
__device__ __forceinline__ void walk_rows_slow(const unsigned *wrow, int xhi, int xlo, int &x, int &e, int &startv,
                                            int lane, int jb, int ntw, int &stop, int *status, bool window) {
    for (int xr = xhi; xr >= xlo; --xr) {
        const unsigned w = wrow[xr];
        const int jr = __builtin_amdgcn_readfirstlane((e >> 5) - jb);
        int jlim = jr, en = 0;
        bool found = false;
        if ((unsigned)jr < (unsigned)ntw) {
            const unsigned word = (unsigned)__builtin_amdgcn_readlane((int)w, jr) >> ((~e) & 31);
            if (word != 0u) { en = e - __builtin_ctz(word) - 1; found = true; }
        } else if (jr >= ntw) {
            jlim = ntw;                                          // e lies past this window: every word qualifies
        }
        if (!found && jlim > 0) {
            const unsigned long long bal = __ballot(lane < jlim && w != 0u);
            if (bal != 0ull) {
                const int js = 63 - __builtin_clzll(bal);
                const unsigned wsel = (unsigned)__builtin_amdgcn_readlane((int)w, js);
                en = (((jb + js) << 5) | (TC - 1)) - __builtin_ctz(wsel) - 1;
                found = true;
            }
        }
        if (!found) {
            if (window && jb > 0) { stop = 1; x = xr; return; }  // the row continues in an earlier window
            en = xr - 1;                                         // cannot happen: the diagonal bit is always set
            if (lane == 0) atomicOr(status, ALIGNER_ST_INTERNAL);
        }
        en = __builtin_amdgcn_readfirstlane(en);
        startv = (lane == (xr & 63)) ? en : startv;
        e = en;
    }
    x = xlo - 1;
}

// One chunk of 64 rows (64c .. 64c+63), entered at row x.  Fast steps; if the flag came up, the FIRST row that
// failed is the highest one whose step did not move e down (a failed step yields e + 1) or (WINDOW) whose e was
// not past the window's first tile; everything above it is good.  That row takes the general search
// (walk_rows_slow) and the fast steps are re-entered right below it.  WINDOW: tiles [jb, jb + ntw) with jb > 0; the
// fast steps run on window-relative frames (maxpath_walk_asm.inc) and their results are shifted back here.  An
// entry row whose e lies above the window (its token runs on from a later window) takes the general search too.
// Leaves x = the next row to walk (64c - 1, or 0 when chunk 0 completed: row 0 is never walked, it starts at frame
// 0 -- the fast steps run over it and its outcome is ignored), or the row to resume at in an earlier window (stop).
template <bool WINDOW>
__device__ __forceinline__ void walk_chunk(const unsigned *wrow, int c, int &x, int &e, int &startv, int lane, int jb,
                                           int ntw, int &stop, int *status, int klo0 = 1) {
    typedef __attribute__((address_space(3))) const unsigned lds_cu32;
    const unsigned lds_addr = (unsigned)(unsigned long long)(lds_cu32 *)(wrow + 64 * c);
    // klo0 = 1: row 0 of these words is the utterance's row 0 (it starts at frame 0: not walked); 0: it is an ordinary row
    // (the second workgroup's first row) and leaves e = the last frame of the row before it
    const int klo = (c == 0) ? klo0 : 0;
    const int off = WINDOW ? (jb << 5) : 0;
    int kent = x - 64 * c;
    while (kent >= klo) {
        if ((e >> 5) - jb >= ntw) {                              // only ever the row a window change resumes at
            x = 64 * c + kent;
            walk_rows_slow(wrow, x, x, x, e, startv, lane, jb, ntw, stop, status, jb > 0);
            if (stop) return;
            --kent;
            continue;
        }
        if (WINDOW && e - off < 32) { x = 64 * c + kent; stop = 1; return; }   // the next window's row
        const int top = kent, e_in = e;
        int flag = 0;
        e -= off;
        walk_chunk_fast<WINDOW>(lds_addr, kent, e, flag, startv);
        e += off;
        if (WINDOW) startv = (lane <= top) ? startv + off : startv;   // (rows it did not reach are walked again)
        if (__builtin_expect(flag >= 0, 1)) break;
        // e entering each row: the step result of the row above it (lane + 1), e_in for the entry row
        int prev = __builtin_amdgcn_update_dpp(0, startv, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
        prev = (lane == top) ? e_in : prev;
        bool bad = startv >= prev;
        if (WINDOW) bad = bad || prev - off < 32;
        const unsigned long long m = __ballot(bad && lane <= top && lane >= klo);
        if (m == 0ull) break;                                    // raised by the last row's outcome only: all rows good
        const int kf = 63 - __builtin_clzll(m);                  // first (highest) failed row of the chunk
        e = __builtin_amdgcn_readfirstlane((kf == top) ? e_in : __builtin_amdgcn_readlane(startv, (kf + 1) & 63));
        x = 64 * c + kf;
        if (WINDOW && e - off < 32) { stop = 1; return; }        // e is in the window's first tile: the next window's row
        walk_rows_slow(wrow, x, x, x, e, startv, lane, jb, ntw, stop, status, jb > 0);
        if (stop) return;
        kent = kf - 1;
    }
    x = 64 * c - 1;
    if (c == 0) x = klo0 ? 0 : -1;
}

// Decision words of `ntw` tiles, rows [0, rows_used), from the workspace into the LDS window: 16-byte loads,
// eight in flight per thread (one element at a time this copy was a memory round trip per 4 bytes and a
// quarter of the long-form kernel's time).  rows_used, ROWS and the window's row pitch are multiples of 4.
__device__ __forceinline__ void load_window(unsigned *win, int RP, const unsigned *gbits, int ROWS, int rows_used,
                                            int ntw, int tid, int nthreads) {
    typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
    const int q = rows_used >> 2, n4 = ntw * q;
    for (int base = 0; base < n4; base += 8 * nthreads) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int idx = base + k * nthreads + tid;
            idx = idx < n4 ? idx : n4 - 1;
            const int j = idx / q, r4 = idx - j * q;
            v[k] = *reinterpret_cast<const u32x4 *>(gbits + (size_t)j * ROWS + 4 * r4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = base + k * nthreads + tid;
            if (idx < n4) {
                const int j = idx / q, r4 = idx - j * q;
                *reinterpret_cast<u32x4 *>(win + j * RP + 4 * r4) = v[k];
            }
        }
    }
}

// The two-workgroup form's walk: ALL of this half's decision words are in LDS ([ntb tiles][RP], rows numbered from the
// half's first row), so a window of 64 tiles is a pointer offset -- nothing is loaded between windows, nobody waits at a
// barrier.  One wave.  Walks rows x .. klo0 from frame e; leaves (first frame) of every walked row in startsLoc[row] and
// e = the last frame of the row before the lowest one.
__device__ __forceinline__ void walk_half(const MaxpathParams &p, const unsigned *win, int RP, int ntb, int &x, int &e, int klo0,
                                          int *startsLoc, int lane) {
    int startv = 0;
    for (int jhi = ntb, jb = 0; jhi > 0; jhi = (jb > 0) ? jb + 1 : 0) {
        jb = (jhi - 64 > 0) ? (jhi - 64) : 0;
        const int ntw = jhi - jb;
        if (x >= klo0) {
            const unsigned *wrow = win + (size_t)(jb + (lane < ntw ? lane : 0)) * RP;
            int stop = 0;
            for (int c = x >> 6; c >= 0 && stop == 0; --c) {
                if (jb == 0) walk_chunk<false>(wrow, c, x, e, startv, lane, 0, ntw, stop, p.status, klo0);
                else         walk_chunk<true>(wrow, c, x, e, startv, lane, jb, ntw, stop, p.status, klo0);
                if (stop == 0) startsLoc[64 * c + lane] = startv + 1;
            }
        }
    }
}

template <bool USE_PREV>
__device__ __forceinline__ void backtrack_and_store(const MaxpathParams &p, int b, int tx, int ty, unsigned char *smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthreads = blockDim.x;
    const int RP = row_pitch(p.ROWS);
    int *startsL = reinterpret_cast<int *>(smem);
    const int starts_words = starts_words_of(p.Tx);
    unsigned *win = reinterpret_cast<unsigned *>(smem) + starts_words;
    const bool in_lds = p.bits_in_lds != 0;
    if (in_lds) win = reinterpret_cast<unsigned *>(smem + p.lds_bits_off);

    const int ntb = (ty + TC - 1) / TC;            // tiles this utterance uses
    const int rows_used = ((tx + 63) / 64) * 64;
    const unsigned *gbits = p.bits + (size_t)b * p.NT * p.ROWS;
    const int WT = in_lds ? ntb : p.WT;

    // Walk state lives in SGPRs (a VGPR-carried loop costs an exec-mask dance per branch).
    const bool walker = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    int x = __builtin_amdgcn_readfirstlane(tx - 1);   // core.pyx:15
    int e = __builtin_amdgcn_readfirstlane(ty - 1);
    int startv = 0;   // (first frame - 1) of rows 64c..64c+63 of the chunk being walked, one per lane

    // windows of WT tiles from the last one down, each sharing its first tile with the next one's last: a row whose
    // last frame is in a window's first tile has no tile before it there and is resumed as an ordinary row of the
    // next window (WT >= 2 whenever the tiles do not fit one window, pick_window)
    for (int jhi = ntb, jb = 0; jhi > 0; jhi = (jb > 0) ? jb + 1 : 0) {
        jb = (jhi - WT > 0) ? (jhi - WT) : 0;
        const int ntw = jhi - jb;
        if (!in_lds) {
            __syncthreads();                           // previous window fully consumed
            load_window(win, RP, gbits + (size_t)jb * p.ROWS, p.ROWS, rows_used, ntw, tid, nthreads);
            __syncthreads();
        }
        ALIGNER_STAMP(2);
        if (walker && x >= 1) {
            // lanes past the window read tile 0's words; they can never be selected (lane select < ntw)
            const unsigned *wrow = win + (lane < ntw ? lane : 0) * RP;
            int stop = 0;
            for (int c = x >> 6; c >= 0 && stop == 0; --c) {
                if (USE_PREV && p.lds_prev_off != 0) {
                    // decision words and the P table both in LDS (single window by construction): cannot fail
                    typedef __attribute__((address_space(3))) const unsigned lds_cu32;
                    const unsigned *prow = reinterpret_cast<const unsigned *>(smem + p.lds_prev_off) +
                                           (lane < ntw ? lane : 0) * RP;
                    walk_chunk_prev((unsigned)(unsigned long long)(lds_cu32 *)(wrow + 64 * c),
                                    (unsigned)(unsigned long long)(lds_cu32 *)(prow + 64 * c), x - 64 * c, e, startv);
                    x = (c == 0) ? 0 : 64 * c - 1;
                } else if (jb == 0) {                  // (every tile in one window, or the last of several)
                    walk_chunk<false>(wrow, c, x, e, startv, lane, 0, ntw, stop, p.status);
                } else {
                    walk_chunk<true>(wrow, c, x, e, startv, lane, jb, ntw, stop, p.status);
                }
                // chunk finished (a stopped one is finished from the next window; startv carries over)
                if (stop == 0) startsL[64 * c + lane] = (c == 0 && lane == 0) ? 0 : startv + 1;
            }
        }
    }
    ALIGNER_STAMP(3);
    // x == 0 here for every valid input (forced diagonal, core.pyx:34): row 0 starts at frame 0.
    if (walker && x != 0 && lane == 0) atomicOr(p.status, ALIGNER_ST_INTERNAL);
    if (walker && lane == 0) startsL[0] = 0;       // (t_x == 1: nothing was walked)
    __syncthreads();
    for (int r = tx + tid; r <= p.Tx; r += nthreads) startsL[r] = ty;
    __syncthreads();
    store_outputs(p, b, tx, ty, startsL);
}

// --------------------------------------------------------------------------
// Generic forward kernel: any Tx <= 256*R, one barrier per frame.  Slow but
// shape-agnostic; also the independent cross-check of the pipelined kernel.
// --------------------------------------------------------------------------
template <int R, int MASKMODE, int VT>
__global__ __launch_bounds__(256) void maxpath_generic_kernel(MaxpathParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    int tx, ty;
    const int mode = classify_lengths(p, b, tx, ty);
    if (mode != MODE_NORMAL) { write_degenerate<MASKMODE, VT>(p, b, mode, tx, ty, reinterpret_cast<int *>(smem)); return; }

    float *qcol = reinterpret_cast<float *>(smem);   // [2][256*R + 1], index x+1
    const int QLD = 256 * R + 1;
    const size_t ubase = (size_t)b * p.Tx * p.ldv;
    unsigned *gbits = p.bits + (size_t)b * p.NT * p.ROWS;

    float q[R];
    unsigned bits[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { q[r] = p.neg; bits[r] = 0u; }
    if (tid == 0) qcol[0] = 0.0f;                      // v_prev for (x=0,y=0): core.pyx:24-25
    __syncthreads();

    for (int y = 0; y < ty; ++y) {
        const float *src = qcol + (y & 1) * QLD;
        float *dst = qcol + ((y + 1) & 1) * QLD;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int x = tid + 256 * r;
            if (x < tx) {
                const float up = src[x];                           // Q[x-1,y-1] (or the x==0 edge)
                const float cur = (x == y) ? p.neg : q[r];         // core.pyx:19-22
                const bool adv = up > cur;                         // core.c:19384
                float v = load_score<VT>(p.value, ubase + (size_t)x * p.ldv + y);
                if (MASKMODE == 1) v = mul_in_dtype<VT>(v, load_score<VT>(p.mask, ubase + (size_t)x * p.ldv + y));
                q[r] = (adv ? up : cur) + v;                       // core.pyx:30
                // the reference overwrites value[x, y] with Q inside its band only (core.pyx:18,30); cells outside
                // it keep their scores.  In place: this thread alone reads and writes the cell.
                if (p.qout && x <= y && x >= tx + y - ty) p.qout[ubase + (size_t)x * p.ldv + y] = q[r];
                // backtrack predicate (core.pyx:34): the diagonal move is forced whatever the scores
                bits[r] = (bits[r] << 1) | ((adv || x == y) ? 1u : 0u);
                dst[x + 1] = q[r];
            }
        }
        if (tid == 0) dst[0] = p.neg;                              // core.pyx:27 for y+1 >= 1
        if ((y & (TC - 1)) == TC - 1 || y == ty - 1) {
            const int t = y / TC;
            const int sh = (TC - 1) - (y & (TC - 1));              // left-align a partial last tile
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int x = tid + 256 * r;
                if (x < p.ROWS) gbits[(size_t)t * p.ROWS + x] = bits[r] << sh;
                bits[r] = 0u;
            }
        }
        __syncthreads();
    }
    __threadfence_block();
    ALIGNER_STAMP(1);
    backtrack_and_store<false>(p, b, tx, ty, smem);
}

// --------------------------------------------------------------------------
// Pipelined forward kernel: NW compute waves + NW loader waves.
// --------------------------------------------------------------------------
// 4 consecutive frames of one text row, always in bounds and never predicated: `rowoff` is
// the element offset of a clamped row, the frame index is clamped to the row's last piece.
// Clamped duplicates only land in cells that cannot matter (rows >= Tx, frames >= Ty) and keep
// every load unconditional, which is what lets hipcc pipeline them with counted vmcnt waits.
template <bool VEC, int MASKMODE>
__device__ __forceinline__ float4 load_tile_piece(const float *ub, const float *mb, unsigned rowoff, int col, int Ty) {
    float4 v;
    if (VEC) {
        const unsigned off = rowoff + (unsigned)(col < Ty - 4 ? col : Ty - 4);
        v = *reinterpret_cast<const float4 *>(ub + off);
        if (MASKMODE == 1) {
            const float4 m = *reinterpret_cast<const float4 *>(mb + off);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
    } else {
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned off = rowoff + (unsigned)(col + i < Ty - 1 ? col + i : Ty - 1);
            t[i] = ub[off];
            if (MASKMODE == 1) t[i] *= mb[off];
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
    }
    return v;
}

__device__ __forceinline__ unsigned absbits(float f) { return __builtin_bit_cast(unsigned, f) & 0x7FFFFFFFu; }

// Finiteness scan of the loader waves: nf = maximum(|a|, |b|, |c|, |d|, nf) with gfx950's IEEE-754-2019
// `maximum` (v_maximum3_f32 propagates NaN, unlike v_max3_f32), two issues per 16 bytes.  Afterwards
// the scores seen so far are all finite  <=>  absbits(nf) < 0x7F800000.
__device__ __forceinline__ void scan4(float &nf, const float4 &v) {
    asm("v_maximum3_f32 %0, |%1|, |%2|, %0\n\tv_maximum3_f32 %0, |%3|, |%4|, %0"
        : "+v"(nf) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
}

// Buffer resource over one utterance's [Tx,Ty] fp32 block: loads take a per-lane byte offset that never
// changes (row and column group) plus a scalar tile offset, so a tile costs the loader no address
// arithmetic; reads past the block (last rows, last partial tile) return 0 instead of faulting.
typedef unsigned __attribute__((ext_vector_type(4))) u32x4r;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t utterance_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buffer_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    // (cast the whole vector: hipcc 7.2 turns per-component reads of the builtin's result into a
    // one-dword load splatted over the four components)
    const u32x4r raw = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const f32x4 f = __builtin_bit_cast(f32x4, raw);
    return make_float4(f.x, f.y, f.z, f.w);
}

// Finite-score sweep of one tile: see maxpath_sweep_asm.inc (generated by tools/gen_sweep_asm.py).
// DIAG: the tile contains cells with row == frame; `rrel` = row - first frame of the tile.
template <bool PUBLISH, bool DIAG>
__device__ __forceinline__ void sweep_tile_fast(float &q, float &m, unsigned &bits, int &coll,
                                                const float4 (&vv)[8], int rrel, float negv) {
    float qb, cur;
    unsigned long long sA, sB, sM0, sM1;
    int sk0, sk1;
#define ALIGNER_SWEEP_OPERANDS(G)                                                                       \
    : [qa] "+v"(q), [qb] "=&v"(qb), [m] "+v"(m), [bits] "+v"(bits), [coll] "+v"(coll), [cur] "=&v"(cur), \
      [sA] "=&s"(sA), [sB] "=&s"(sB), [sM0] "=&s"(sM0), [sM1] "=&s"(sM1), [sk0] "=&s"(sk0),             \
      [sk1] "=&s"(sk1)                                                                                   \
    : [rrel] "v"(rrel), [neg] "v"(negv),                                                                 \
      [v0] "v"(vv[G].x), [v1] "v"(vv[G].y), [v2] "v"(vv[G].z), [v3] "v"(vv[G].w),                       \
      [v4] "v"(vv[G + 1].x), [v5] "v"(vv[G + 1].y), [v6] "v"(vv[G + 1].z), [v7] "v"(vv[G + 1].w),       \
      [v8] "v"(vv[G + 2].x), [v9] "v"(vv[G + 2].y), [v10] "v"(vv[G + 2].z), [v11] "v"(vv[G + 2].w),     \
      [v12] "v"(vv[G + 3].x), [v13] "v"(vv[G + 3].y), [v14] "v"(vv[G + 3].z), [v15] "v"(vv[G + 3].w)    \
    : "vcc"
    if (DIAG) {
        if (PUBLISH) {
            asm volatile(ALIGNER_SWEEP16_DIAG_PUB_0 ALIGNER_SWEEP_OPERANDS(0));
            asm volatile(ALIGNER_SWEEP16_DIAG_PUB_16 ALIGNER_SWEEP_OPERANDS(4));
        } else {
            asm volatile(ALIGNER_SWEEP16_DIAG_NOPUB_0 ALIGNER_SWEEP_OPERANDS(0));
            asm volatile(ALIGNER_SWEEP16_DIAG_NOPUB_16 ALIGNER_SWEEP_OPERANDS(4));
        }
        // the diagonal move is forced whatever the scores (core.pyx:34): frame c <-> bit 31-c
        if (rrel >= 0 && rrel < TC) bits |= 0x80000000u >> rrel;
    } else {
        if (PUBLISH) {
            asm volatile(ALIGNER_SWEEP16_PUB_0 ALIGNER_SWEEP_OPERANDS(0));
            asm volatile(ALIGNER_SWEEP16_PUB_16 ALIGNER_SWEEP_OPERANDS(4));
        } else {
            asm volatile(ALIGNER_SWEEP16_NOPUB_0 ALIGNER_SWEEP_OPERANDS(0));
            asm volatile(ALIGNER_SWEEP16_NOPUB_16 ALIGNER_SWEEP_OPERANDS(4));
        }
    }
#undef ALIGNER_SWEEP_OPERANDS
}

// Exact (reference select) sweep of a whole utterance inside the pipelined kernel's
// workgroup: one text row per thread, one barrier per frame, decision words written in
// the same (tile, row) layout so the shared backtrack reads them unchanged.  Only taken for
// utterances whose scores contain a NaN/infinity -- correctness path, not a fast path.
template <int MASKMODE, int VT>
__device__ __forceinline__ void exact_fallback_sweep(const MaxpathParams &p, int b, int tx, int ty, unsigned char *smem,
                                                   unsigned *bitsL, int RPB) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    float *qcol = reinterpret_cast<float *>(smem);            // [2][nthreads + 1], index x+1
    const int QLD = nthreads + 1;
    const size_t ubase = (size_t)b * p.Tx * p.ldv;
    const int x = tid;                                        // tx <= 63*NW < nthreads
    const int slot = x;
    unsigned *gw = p.bits + (size_t)b * p.NT * p.ROWS + slot;
    float q = p.neg;
    unsigned bits = 0u;
    int pprev = -2;
    unsigned *prevL = reinterpret_cast<unsigned *>(smem + p.lds_prev_off);
    __syncthreads();
    if (tid == 0) qcol[0] = 0.0f;                             // core.pyx:24-25
    __syncthreads();
    for (int y = 0; y < ty; ++y) {
        const float *src = qcol + (y & 1) * QLD;
        float *dst = qcol + ((y + 1) & 1) * QLD;
        if (x < tx) {
            const float up = src[x];
            const float cur = (x == y) ? p.neg : q;             // core.pyx:19-22
            const bool adv = up > cur;                          // core.c:19384
            float v = load_score<VT>(p.value, ubase + (size_t)x * p.ldv + y);
            if (MASKMODE == 1) v = mul_in_dtype<VT>(v, load_score<VT>(p.mask, ubase + (size_t)x * p.ldv + y));
            q = (adv ? up : cur) + v;                           // core.pyx:30
            bits = (bits << 1) | ((adv || x == y) ? 1u : 0u);   // core.pyx:34
            dst[x + 1] = q;
        }
        if (tid == 0) dst[0] = p.neg;                           // core.pyx:27
        if ((y & (TC - 1)) == TC - 1 || y == ty - 1) {
            const int t = y / TC;
            const unsigned wv = bits << ((TC - 1) - (y & (TC - 1)));
            if (x < tx) {
                if (p.bits_in_lds) {
                    bitsL[t * RPB + slot] = wv;
                    if (p.lds_prev_off) {
                        pprev = wv ? (TC * t + TC - 2 - __builtin_ctz(wv)) : pprev;
                        prevL[(t + 1) * RPB + slot] = (unsigned)pprev;        // (slot t + 1: the walk indexes P with e's tile)
                    }
                } else {
                    gw[(size_t)t * p.ROWS] = wv;
                }
            }
            bits = 0u;
        }
        __syncthreads();
    }
}

// Development aid: -DALIGNER_EXP_NOBARRIER drops the phase barriers (results are garbage); the waves' own
// instruction streams can then be timed apart (stamp 4, tools/stamps.py, tools/exp_build.sh).
#ifdef ALIGNER_EXP_NOBARRIER
#define PHASE_BARRIER() do {} while (0)
#else
#define PHASE_BARRIER() __syncthreads()
#endif
// PAIR: two workgroups -- two CUs -- per utterance (grid 2B: block b takes compute waves 0..NW-1 of utterance b,
// block B + b waves NW..2NW-1).  A CU's VALU is what a long text (more than 4 x 63 rows: two compute waves per
// SIMD) is bound by, and a small batch leaves most CUs idle.  The boundary row between the halves goes through
// global memory: the first half's last compute wave publishes into LDS as if a fifth wave followed, its loader wave
// forwards those 32 values per tile with agent-scope stores into a ring that xring_fill_kernel set to 0xFFFFFFFF (the compute
// waves' code is the one-workgroup code: any store of theirs that hipcc could not tell from a load made every tile
// wait for the previous tile's stores), the second half's first loader wave reads them four tiles ahead and polls until no
// word is the filler (a running score is finite; a NaN with that bit pattern is rewritten before it is stored) --
// the data is its own flag, nothing waits for a store to complete.  The first half never waits for the second, so
// the pair cannot deadlock once both are resident; that lower block indices are dispatched first is how the
// hardware behaves, not a HIP guarantee, so the host only takes this form when all 2B workgroups fit the chip at
// once (2B <= CU count), every wait is bounded, and a second half that gave up leaves an all-zero path, zero durations
// and ALIGNER_ST_INTERNAL behind -- never a path walked over filler.  The second half waits once more, for
// the first half's decision words (release / acquire on xflag), and then runs the backtrack for the utterance.
constexpr unsigned XRING_EMPTY = 0xFFFFFFFFu;
constexpr int XRING_SPIN_LIMIT = 1 << 18;          // polls before giving up for good with ALIGNER_ST_INTERNAL (~0.3 s)

template <int NW, int DEPTH, bool VEC, int MASKMODE, int VT, bool PAIR>
__device__ __forceinline__ void maxpath_pipelined_body(const MaxpathParams &p, const int blk) {
    static_assert(!PAIR || NW == 4, "the split form is built on the four-wave workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ldv = PAIR ? p.Ty : p.ldv;            // (the two-workgroup form is out of scalar registers: contiguous scores only, forward_impl)
    const int half = PAIR ? (int)(blk >= p.B) : 0;
    const int b = PAIR ? blk - half * p.B : blk;
    int tx, ty;
    const int mode = classify_lengths(p, b, tx, ty);
    if (mode != MODE_NORMAL) {
        if (half == 0) write_degenerate<MASKMODE, VT>(p, b, mode, tx, ty, reinterpret_cast<int *>(smem));
        return;
    }
    // paired: this utterance really has rows for the second workgroup; otherwise the first one does everything
    const bool paired = PAIR && (tx + RPW - 1) / RPW > NW;
    if (PAIR && half == 1 && !paired) return;
    if (PAIR && paired && half == 0 && (p.flags & ALIGNER_F_TEST_DROP_FIRST_HALF)) return;   // testing: never delivers
    const int wbase = NW * half;                                // this workgroup's waves are wbase .. wbase + NW - 1
    unsigned *xr = PAIR ? p.xring + (size_t)b * p.NT * TC : nullptr;
    ALIGNER_STAMP(0);
    ALIGNER_STAMP(6);

    float *tiles = reinterpret_cast<float *>(smem);              // [NW][2][64][TILE_LD]
    float *ring  = tiles + NW * 2 * 64 * TILE_LD;                // [NW][RING_T][RING_LD]: row 63w-1 for wave w
    int   *flagp = reinterpret_cast<int *>(ring + (NW + (PAIR ? 1 : 0)) * RING_T * RING_LD);   // [4] non-finite score seen
    unsigned *bitsL = reinterpret_cast<unsigned *>(smem + p.lds_bits_off);   // [NT][ROWS+1] when in LDS
    const int RPB = row_pitch(p.ROWS);
    const int ntb = (ty + TC - 1) / TC;
    const size_t ubase = (size_t)b * p.Tx * ldv;
    const int nw_act = (tx + RPW - 1) / RPW;                    // waves that own at least one real row

    // wave 0's ghost lane replays "row -1": max_neg_val for every frame (core.pyx:27)
    for (int i = tid; i < RING_T * RING_LD; i += NW * 128) ring[i] = p.neg;
    if (tid == 0) { flagp[0] = 0; flagp[1] = 0; }
    __syncthreads();

    if (!p.force_exact) {
        // Tiles a wave needs form one contiguous range [t_lo, t_hi] (band of rows 63w..63w+62,
        // core.pyx:18); every wave still takes part in all ntb + NW phase barriers.
        const int w = (wave < NW) ? wave : wave - NW;           // index inside the workgroup (LDS buffers, phases)
        const int gw = wbase + w;                               // index inside the utterance (rows)
        const bool active = gw < nw_act;
        const int t_lo = active ? (RPW * gw) / TC : 0;
        int t_hi = active ? (ty - tx + RPW * gw + RPW - 1) / TC : -1;
        if (t_hi > ntb - 1) t_hi = ntb - 1;
        const int ntiles = t_hi - t_lo + 1;
        if (wave < NW) {
            // ------------------------------ compute wave ------------------------------
            const int row = RPW * gw + lane - 1;                // lane 0: ghost (row 63gw-1)
            const bool publish = gw + 1 < nw_act;
            float q = (gw == 0 && lane == 0) ? 0.0f : p.neg;    // Q[-1,-1] = 0 (core.pyx:24-25)
            float m = 0.0f;                                     // lane 0 stays 0 through the asm sweep
            unsigned bits = 0u;
            int coll = 0;
            int pprev = -2;                                     // P of this row so far: (last decision frame) - 1
            unsigned *prevL = reinterpret_cast<unsigned *>(smem + p.lds_prev_off);
            const float *mytiles = tiles + w * 2 * 64 * TILE_LD + lane * TILE_LD;
            const float *myring = ring + w * RING_T * RING_LD;
            float *outring = ring + (publish ? w + 1 : w) * RING_T * RING_LD;   // (PAIR: slot set NW is the loaders' to forward)
            const int brow = (lane == 0) ? p.ROWS - 1 : row;            // decision word column (ghost lane: padding)
            unsigned *gbits = p.bits + (size_t)b * p.NT * p.ROWS + brow;

            // One tile.  What does not change from tile to tile is decided outside the loop (a uniform branch costs a
            // wave about 20 cycles taken or not, and the loop had a dozen): PUB = this wave's last row feeds another
            // wave; DG = the tile may touch the diagonal (only a wave's first one to three tiles can); BM = where the
            // decision words go (0 workspace, 1 LDS, 2 LDS + the P table).
            typedef __attribute__((address_space(3))) const float lds_cf32;
            lds_cf32 *tile_l = (lds_cf32 *)mytiles, *ring_l = (lds_cf32 *)myring;
            auto tile = [&](auto PUB, auto DG, auto BM, int t) {
                constexpr bool pub = decltype(PUB)::value;
                constexpr int bm = decltype(BM)::value;
                // one ds_read_b128 per 4 frames; the padded row stride (9 x 16 B) makes the 16-lane groups of a b128
                // read hit 16 different 16-byte slots: conflict-free.  Lane 0 reads the row above from the ring.
                lds_cf32 *tl = tile_l + (t & 1) * 64 * TILE_LD, *rg = ring_l + (t & (RING_T - 1)) * RING_LD;
                const lds_f32x4 *src = (const lds_f32x4 *)((lane == 0) ? rg : tl);
                float4 vv[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 r = src[g];
                    vv[g] = make_float4(r.x, r.y, r.z, r.w);
                }
                const int y0 = t * TC;
                const int rrel = row - y0;
                if (decltype(DG)::value && (y0 <= RPW * gw + RPW - 1) && (y0 + TC - 1 >= RPW * gw))
                    sweep_tile_fast<pub, true>(q, m, bits, coll, vv, rrel, p.neg);
                else
                    sweep_tile_fast<pub, false>(q, m, bits, coll, vv, rrel, p.neg);
                // unmasked stores: lanes >= 32 hit the slot's padding, the ghost lane a padding column
                if (pub) outring[(t & (RING_T - 1)) * RING_LD + lane] = __builtin_bit_cast(float, coll);
                if (bm >= 1) {
                    bitsL[t * RPB + brow] = bits;                              // frame 32t+c <-> bit 31-c
                    if (bm == 2) {
                        // P[t][row]: where the backtrack goes from a token of this row still running at the
                        // end of tile t = (last frame <= 32t+31 with a decision bit) - 1  (walk_chunk_prev)
                        pprev = bits ? (TC * t + TC - 2 - __builtin_ctz(bits)) : pprev;
                        prevL[(t + 1) * RPB + brow] = (unsigned)pprev;        // (kept one tile up: slot t + 1)
                    }
                } else {
                    gbits[(size_t)t * p.ROWS] = bits;
                }
                bits = 0u;
                PHASE_BARRIER();
            };
            // the tiles the diagonal crosses come first (rows 63w .. 63w+62 meet frames of the same numbers)
            const int t_dend = (RPW * gw + RPW - 1) / TC;
            auto run = [&](auto PUB, auto BM) {
                int t = t_lo;
                for (; t <= t_hi && t <= t_dend; ++t) tile(PUB, std::integral_constant<bool, true>(), BM, t);
                for (; t <= t_hi; ++t) tile(PUB, std::integral_constant<bool, false>(), BM, t);
            };
            for (int i = 0; i < t_lo + w + 1; ++i) PHASE_BARRIER();
            {
                std::integral_constant<bool, true> yes;
                std::integral_constant<bool, false> no;
                const int bmode = p.bits_in_lds ? (p.lds_prev_off ? 2 : 1) : 0;
                if (publish) {
                    if (bmode == 2) run(yes, std::integral_constant<int, 2>());
                    else if (bmode == 1) run(yes, std::integral_constant<int, 1>());
                    else run(yes, std::integral_constant<int, 0>());
                } else {
                    if (bmode == 2) run(no, std::integral_constant<int, 2>());
                    else if (bmode == 1) run(no, std::integral_constant<int, 1>());
                    else run(no, std::integral_constant<int, 0>());
                }
            }
            for (int i = 0; i < ntb + NW - w - t_hi - 2; ++i) PHASE_BARRIER();
        } else {
            // ------------------------------ loader wave -------------------------------
            if (active) {
                float nf = 0.f;                             // maximum |score| seen, NaN-propagating (scan4)
                // PAIR, first loader wave of the second workgroup: the row above this workgroup's first row comes
                // from the other CU (see the kernel's header).  Read four tiles ahead (XAHEAD), polled when its
                // tile is staged, handed to compute wave 0 through the ring slot its ghost lane reads anyway.
                const bool xsub = PAIR && paired && half == 1 && w == 0;
                int xt_hi = (ty - tx + RPW * (gw - 1) + RPW - 1) / TC;      // last tile wave gw - 1 publishes
                if (xt_hi > ntb - 1) xt_hi = ntb - 1;
                const bool xfw = PAIR && paired && half == 0 && w == NW - 1;   // forwards its compute wave's last row
                auto xforward = [&](int t) {                                   // tile t is complete in the ring
                    if constexpr (PAIR) {
                        if (xfw && t >= t_lo) {
                            const unsigned cv0 = __builtin_bit_cast(unsigned, ring[NW * RING_T * RING_LD + (t & (RING_T - 1)) * RING_LD + (lane & (TC - 1))]);
                            const unsigned cv = (cv0 == XRING_EMPTY) ? 0x7FC00000u : cv0;
                            if (lane < TC) __hip_atomic_store(xr + TC * t + lane, cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                };
                // after the staging loop: tile t_hi - 1 is in the ring now, t_hi after one more phase (returns the
                // number of phase barriers it took)
                auto xforward_tail = [&]() -> int {
                    if constexpr (PAIR) {
                        if (xfw) {
                            xforward(t_hi - 1);
                            PHASE_BARRIER();
                            xforward(t_hi);
                            return 1;
                        }
                    }
                    return 0;
                };
                // An agent-scope load is a trip past the L2 (the halves may sit on different XCDs, ~1.3 us): it is issued
                // XAHEAD tiles early, further than its latency, into a register of its own -- the staging loop is
                // unrolled XAHEAD times for that (a queue shifted with moves made every tile wait for the load issued
                // the tile before: a move reads its source).  A read that came too early finds the filler and is
                // repeated when its tile is staged; that stall drops this workgroup back until its early reads succeed.
                constexpr int XAHEAD = 2 * DEPTH;
                constexpr int UNR = PAIR ? XAHEAD : DEPTH;                   // tiles per trip of the staging loop
                unsigned xq[XAHEAD];
                bool xdead = false;                                         // gave up once: do not wait again
                auto xload = [&](int t) -> unsigned {
                    const int tc = t < xt_hi ? t : xt_hi;
                    return __hip_atomic_load(xr + TC * tc + (lane & (TC - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                };
                auto xprime = [&]() {
                    if constexpr (PAIR) {
                        if (xsub) {
#pragma unroll
                            for (int i = 0; i < XAHEAD; ++i) xq[i] = xload(t_lo + i);
                        }
                    }
                };
                auto xput = [&](int u, int t) {                             // tile t is being staged (u: its queue slot)
                    if constexpr (PAIR) {
                        if (xsub) {
                            unsigned cur = xq[u];
                            xq[u] = xload(t + XAHEAD);
                            if (t <= xt_hi) {
                                // (the first test stands outside the loop, and the loop starts with its load: a loop
                                // header that also tested the queued value made hipcc wait for EVERY load in flight
                                // there -- s_waitcnt vmcnt(0) on the path that never polls)
                                if (__builtin_expect(__ballot(cur == XRING_EMPTY) != 0ull, 0) && !xdead) {
                                    int spins = 0;
                                    do {
                                        if (++spins > XRING_SPIN_LIMIT) {
                                            // the first half never waits for this one, but nothing promises that it
                                            // has been dispatched: workgroups are dealt to eight XCDs with their own
                                            // dispatchers, and other streams' kernels may hold the CU it is queued
                                            // for.  Give up for good: the utterance's outputs become all-zero (below)
                                            if (lane == 0) { atomicOr(p.status, ALIGNER_ST_INTERNAL); flagp[1] = 1; }
                                            xdead = true;
                                            break;
                                        }
                                        __builtin_amdgcn_s_sleep(8);
                                        cur = xload(t);
                                    } while (__ballot(cur == XRING_EMPTY) != 0ull);
                                }
                                ring[(t & (RING_T - 1)) * RING_LD + lane] = __builtin_bit_cast(float, cur);   // (w == 0)
                            }
                        }
                    }
                };
                if (VT == VT_F32) {
                    const int rr = lane >> 3, cg = lane & 7;
                    float *mytiles = tiles + w * 2 * 64 * TILE_LD + rr * TILE_LD + 4 * cg;
                    const float *ub = static_cast<const float *>(p.value) + ubase;
                    const float *mb = (MASKMODE == 1) ? reinterpret_cast<const float *>(p.mask) + ubase : nullptr;
                    // LDS row slot 8k+rr <-> text row 63w + 8k + rr - 1 (slot 0 is the ghost lane's: never read)
                    unsigned rowoff[8];
    #pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // rows past the utterance's own text (padding: often -inf log-probs) are replaced
                        // by its last row: they are never read by the DP and must not trip the finiteness scan
                        int r = RPW * gw + 8 * k + rr - 1;
                        r = r < 0 ? 0 : (r > tx - 1 ? tx - 1 : r);
                        rowoff[k] = (unsigned)r * (unsigned)ldv;
                    }
                    float4 buf[DEPTH][8];
                    // VEC: buffer loads, per-lane byte offsets fixed for the whole sweep + a scalar tile offset
                    const unsigned ubytes = (unsigned)p.Tx * (unsigned)ldv * 4u;
                    const __amdgpu_buffer_rsrc_t urs = utterance_rsrc(ub, ubytes);
                    const __amdgpu_buffer_rsrc_t mrs = utterance_rsrc(MASKMODE == 1 ? mb : ub, ubytes);
                    unsigned voff[8];
    #pragma unroll
                    for (int k = 0; k < 8; ++k) voff[k] = (rowoff[k] + 4u * (unsigned)cg) * 4u;
                    // every refill is issued unconditionally (tile index clamped to t_hi: duplicates hit
                    // L2) so the number of loads in flight at each register->LDS pass is a constant
                    auto issue = [&](float4 (&dst)[8], int t) {
                        const int tc = t < t_hi ? t : t_hi;
                        if (VEC) {
                            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(TC * tc * 4);
    #pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                dst[k] = buffer_load4(urs, voff[k], soff);
                                if (MASKMODE == 1) {
                                    const float4 mk = buffer_load4(mrs, voff[k], soff);
                                    dst[k].x *= mk.x; dst[k].y *= mk.y; dst[k].z *= mk.z; dst[k].w *= mk.w;
                                }
                            }
                        } else {
    #pragma unroll
                            for (int k = 0; k < 8; ++k)
                                dst[k] = load_tile_piece<false, MASKMODE>(ub, mb, rowoff[k], TC * tc + 4 * cg, p.Ty);
                        }
                    };
    #pragma unroll
                    for (int d = 0; d < DEPTH; ++d) issue(buf[d], t_lo + d);
                    xprime();
                    for (int i = 0; i < t_lo + w; ++i) PHASE_BARRIER();
                    for (int i0 = 0; i0 < ntiles; i0 += UNR) {
    #pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            const int d = u % DEPTH;
                            const int t = t_lo + i0 + u;
                            if (t <= t_hi) {
                                xput(u, t);
                                xforward(t - 2);
                                float *dst = mytiles + (t & 1) * 64 * TILE_LD;
                                if (t == ntb - 1) {
                                    // frames >= t_y (mel padding, or past the row's end) never matter: zero them
                                    const int c0 = TC * t + 4 * cg;
    #pragma unroll
                                    for (int k = 0; k < 8; ++k) {
                                        float4 v = buf[d][k];
                                        v.x = (c0 + 0 < ty) ? v.x : 0.f; v.y = (c0 + 1 < ty) ? v.y : 0.f;
                                        v.z = (c0 + 2 < ty) ? v.z : 0.f; v.w = (c0 + 3 < ty) ? v.w : 0.f;
                                        *reinterpret_cast<float4 *>(dst + 8 * k * TILE_LD) = v;
                                        scan4(nf, v);
                                    }
                                } else {
    #pragma unroll
                                    for (int k = 0; k < 8; ++k) {
                                        const float4 v = buf[d][k];
                                        *reinterpret_cast<float4 *>(dst + 8 * k * TILE_LD) = v;
                                        scan4(nf, v);
                                    }
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            issue(buf[d], t + DEPTH);
                            __builtin_amdgcn_sched_barrier(0);
                            if (t <= t_hi) PHASE_BARRIER();
                        }
                    }
                } else {
                    // 16-bit scores (bf16 / fp16): 16 bytes = 8 frames, so one instruction covers 16 rows x 32 frames
                    // and a tile is 4 loads; the up-cast to fp32 (exact) happens on the way into LDS, where the
                    // compute waves find the same padded fp32 tile as for fp32 scores.
                    const int rr = lane >> 2, cg = lane & 3;
                    float *mytiles = tiles + w * 2 * 64 * TILE_LD + rr * TILE_LD + 8 * cg;
                    const unsigned short *ub = static_cast<const unsigned short *>(p.value) + ubase;
                    const unsigned short *mb = (MASKMODE == 1) ? static_cast<const unsigned short *>(p.mask) + ubase : ub;
                    const unsigned ubytes = (unsigned)p.Tx * (unsigned)ldv * 2u;
                    const __amdgpu_buffer_rsrc_t urs = utterance_rsrc(ub, ubytes);
                    const __amdgpu_buffer_rsrc_t mrs = utterance_rsrc(mb, ubytes);
                    unsigned voff[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // LDS row slot 16k+rr <-> text row 63w + 16k + rr - 1; rows outside the utterance's own text
                        // are replaced by its last row (never read by the DP, must not trip the finiteness scan)
                        int r = RPW * gw + 16 * k + rr - 1;
                        r = r < 0 ? 0 : (r > tx - 1 ? tx - 1 : r);
                        voff[k] = ((unsigned)r * (unsigned)ldv + 8u * (unsigned)cg) * 2u;
                    }
                    u32x4r buf[DEPTH][4], mbuf[MASKMODE == 1 ? DEPTH : 1][4];
                    auto issue = [&](int d, int t) {
                        const int tc = t < t_hi ? t : t_hi;
                        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(TC * tc * 2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            buf[d][k] = __builtin_amdgcn_raw_buffer_load_b128(urs, voff[k], soff, 0);
                            if (MASKMODE == 1) mbuf[d][k] = __builtin_amdgcn_raw_buffer_load_b128(mrs, voff[k], soff, 0);
                        }
                    };
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) issue(d, t_lo + d);
                    xprime();
                    for (int i = 0; i < t_lo + w; ++i) PHASE_BARRIER();
                    for (int i0 = 0; i0 < ntiles; i0 += UNR) {
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            const int d = u % DEPTH;
                            const int t = t_lo + i0 + u;
                            if (t <= t_hi) {
                                xput(u, t);
                                xforward(t - 2);
                                float *dst = mytiles + (t & 1) * 64 * TILE_LD;
                                const int c0 = TC * t + 8 * cg;
                                const bool tail = (t == ntb - 1);   // frames >= t_y never matter: zero them
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    float f[8];
#pragma unroll
                                    for (int i = 0; i < 4; ++i) {
                                        const unsigned u = buf[d][k][i];
                                        f[2 * i] = half_to_f32<VT>(u & 0xFFFFu);
                                        f[2 * i + 1] = half_to_f32<VT>(u >> 16);
                                        if (MASKMODE == 1) {
                                            const unsigned mu = mbuf[MASKMODE == 1 ? d : 0][k][i];
                                            f[2 * i] = mul_in_dtype<VT>(f[2 * i], half_to_f32<VT>(mu & 0xFFFFu));
                                            f[2 * i + 1] = mul_in_dtype<VT>(f[2 * i + 1], half_to_f32<VT>(mu >> 16));
                                        }
                                    }
                                    if (tail) {
#pragma unroll
                                        for (int i = 0; i < 8; ++i) f[i] = (c0 + i < ty) ? f[i] : 0.f;
                                    }
                                    const float4 v0 = make_float4(f[0], f[1], f[2], f[3]), v1 = make_float4(f[4], f[5], f[6], f[7]);
                                    *reinterpret_cast<float4 *>(dst + 16 * k * TILE_LD) = v0;
                                    *reinterpret_cast<float4 *>(dst + 16 * k * TILE_LD + 4) = v1;
                                    scan4(nf, v0);
                                    scan4(nf, v1);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            issue(d, t + DEPTH);
                            __builtin_amdgcn_sched_barrier(0);
                            if (t <= t_hi) PHASE_BARRIER();
                        }
                    }
                }
                if (absbits(nf) >= 0x7F800000u) flagp[0] = 1;   // benign race: every writer stores 1
                const int xb = xforward_tail();
                for (int i = xb; i < ntb + NW - w - t_hi - 1; ++i) PHASE_BARRIER();
            } else {
                for (int i = 0; i < ntb + NW; ++i) PHASE_BARRIER();
            }
        }
        ALIGNER_STAMP(4);
        __syncthreads();
    }
    if (PAIR && paired) {
        // Each half walks its OWN rows (round 4).  The tiles are dead after the sweep and a half's decision words -- 125
        // tiles x 256 rows at [8,500,4000] -- fit in their place, so: the second half walks rows t_x-1 .. 252 out of its
        // LDS and hands the frame it arrives at to the first half (one word, xwalk), which has had its own words in
        // LDS since its sweep ended (it finishes ~20 phases before the second half) and walks rows 251 .. 0; each half
        // stores the outputs of its rows and frames.  Before, the second half walked all 500 rows over two windows of
        // 132 KB copied from the workspace between barriers: 28 of the kernel's 106 us.  Not taken (XWALK_SOLO: the
        // second half does everything, as before) when a non-finite score asks for the exact sweep or the words do
        // not fit (p.split_walk).  The hand-over word is claimed with compare-and-swap from both sides, so that a first
        // half that gave up waiting and a second half that arrives late cannot both write outputs.
        constexpr unsigned XWALK_SOLO = 0xFFFFFFFEu, XWALK_CANCEL = 0xFFFFFFFDu;
        const int RP2 = row_pitch(256);
        int *startsL = reinterpret_cast<int *>(smem);
        unsigned *win2 = reinterpret_cast<unsigned *>(smem) + starts_words_of(p.Tx);
        int *bcast = startsL + starts_words_of(p.Tx) - 1;               // a word no row owns (the flags' LDS is about to hold decision words)
        const unsigned *gb = p.bits + (size_t)b * p.NT * p.ROWS;
        if (half == 0) {
            // this half's decision words are written (the barrier waited for every wave's stores): hand them over, with
            // what its loaders saw
            if (tid == 0) __hip_atomic_store(p.xflag + b, flagp[0] != 0 ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (!p.split_walk) return;                                   // the other workgroup finishes the utterance
            __threadfence_block();
            __syncthreads();                                             // (thread 0 has read the flags this copy overwrites)
            load_window(win2, RP2, gb, p.ROWS, 256, ntb, tid, NW * 128);  // rows 0 .. 255 of every tile
            if (tid == 0) {
                unsigned v;
                int spins = 0;
                while ((v = __hip_atomic_load(p.xwalk + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == XRING_EMPTY) {
                    if (++spins > ((p.flags & ALIGNER_F_TEST_IMPATIENT_FIRST_HALF) ? 0 : XRING_SPIN_LIMIT)) {
                        unsigned expect = XRING_EMPTY;                   // give up -- unless the word arrives this instant
                        if (__hip_atomic_compare_exchange_strong(p.xwalk + b, &expect, XWALK_CANCEL, __ATOMIC_RELAXED,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            atomicOr(p.status, ALIGNER_ST_INTERNAL);
                            v = XWALK_CANCEL;
                        } else {
                            v = expect;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                *bcast = (int)v;
            }
            __syncthreads();
            const unsigned hv = (unsigned)*bcast;
            if (hv >= XWALK_CANCEL) return;      // SOLO: the other half did everything; CANCEL: it will, when it arrives (below)
            ALIGNER_STAMP(1);
            if (wave == 0) {
                int x = RPW * NW - 1, e = (int)hv;                       // row 251 ends at the frame handed over
                walk_half(p, win2, RP2, ntb, x, e, 1, startsL, lane);
                if (x != 0 && lane == 0) atomicOr(p.status, ALIGNER_ST_INTERNAL);
                if (lane == 0) { startsL[0] = 0; startsL[RPW * NW] = (int)hv + 1; }
            }
            ALIGNER_STAMP(3);
            __syncthreads();
            store_outputs(p, b, tx, ty, startsL, true, 0, RPW * NW, 0, (int)hv + 1);
            ALIGNER_STAMP(5);
            ALIGNER_STAMP(7);
            return;
        }
        if (tid == 0) {
            int v, spins = 0;
            while ((v = __hip_atomic_load(p.xflag + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == (int)XRING_EMPTY) {
                if (++spins > XRING_SPIN_LIMIT) { atomicOr(p.status, ALIGNER_ST_INTERNAL); flagp[1] = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (v == 2) flagp[0] = 1;
        }
        // (thread 0's acquire has invalidated this CU's L1 and the XCD's L2 for everybody; the barrier orders the
        // other waves' loads behind it -- a full fence by all eight waves here cost the kernel 2 us)
        __syncthreads();
        const int gave_up = flagp[1], nonfinite = flagp[0];
        __syncthreads();                                                 // (everybody has read the flags: their LDS may be reused)
        if (gave_up != 0) {
            // gave up waiting for the first half (ALIGNER_ST_INTERNAL is set): a defined result instead of a path
            // walked over filler -- all-zero path, zero durations, no token on any frame
            if (p.split_walk && tid == 0) __hip_atomic_store(p.xwalk + b, XWALK_SOLO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            write_degenerate<MASKMODE, VT>(p, b, MODE_EMPTY, tx, ty, reinterpret_cast<int *>(smem));
            return;
        }
        if (p.split_walk && !p.force_exact && nonfinite == 0) {
            const int r0 = RPW * NW;                                     // this half's first row (252)
            const int rows_loc = ((tx - r0 + 63) / 64) * 64;
            __threadfence_block();
            load_window(win2, RP2, gb + r0, p.ROWS, rows_loc, ntb, tid, NW * 128);
            __syncthreads();
            ALIGNER_STAMP(1);
            if (wave == 0) {
                int x = tx - 1 - r0, e = ty - 1;                         // core.pyx:15
                walk_half(p, win2, RP2, ntb, x, e, 0, startsL + r0, lane);
                e = __builtin_amdgcn_readfirstlane(e);
                if (lane == 0) {
                    // hand the first half its entry frame -- unless it has given up waiting (then nobody walks its rows)
                    unsigned expect = XRING_EMPTY;
                    const bool ok = x == -1 && e >= r0 - 1 &&
                                    __hip_atomic_compare_exchange_strong(p.xwalk + b, &expect, (unsigned)e, __ATOMIC_RELAXED,
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // The first half has given up waiting (XWALK_CANCEL: the CUs were contended, e.g. by another stream's
                    // kernels; it has set ALIGNER_ST_INTERNAL and left without storing anything): its decision words are in
                    // the workspace all the same (xflag's acquire above), so this half walks every row itself -- a late
                    // answer, not a lost one.
                    const bool cancelled = !ok && expect == XWALK_CANCEL;
                    if (!ok && !cancelled) {
                        if (expect == XRING_EMPTY)                       // (an inconsistent walk: cannot happen)
                            __hip_atomic_store(p.xwalk + b, XWALK_SOLO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicOr(p.status, ALIGNER_ST_INTERNAL);
                    }
                    *bcast = ok ? 1 : cancelled ? 2 : 0;
                }
            }
            ALIGNER_STAMP(3);
            __syncthreads();
            const int handed = *bcast;
            __syncthreads();                                             // (everybody has read the word: its LDS may be reused)
            if (handed == 0) {                                           // the defined failure
                write_degenerate<MASKMODE, VT>(p, b, MODE_EMPTY, tx, ty, reinterpret_cast<int *>(smem));
                return;
            }
            if (handed == 1) {
                for (int r = tx + tid; r <= p.Tx; r += NW * 128) startsL[r] = ty;
                __syncthreads();
                store_outputs(p, b, tx, ty, startsL, true, r0, -1, startsL[r0], -1);
                ALIGNER_STAMP(5);
                ALIGNER_STAMP(7);
                return;
            }
            // handed == 2: the one-walker form, over both halves' words in the workspace (no non-finite score was seen:
            // that was asked above).  (A call of its own rather than a fall-through to the one below: what that one
            // needs would otherwise stay live in scalar registers across the generated walk above -- 64 of them spilled.)
            __threadfence_block();
            backtrack_and_store<(NW <= 4)>(p, b, tx, ty, smem);
            return;
        }
        if (p.split_walk && tid == 0) __hip_atomic_store(p.xwalk + b, XWALK_SOLO, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // A NaN or an infinity among the scores (or max_neg_val): v_max no longer equals the
    // reference's select, so redo this utterance with the exact barrier-per-frame sweep.
    if (p.force_exact || flagp[0] != 0) exact_fallback_sweep<MASKMODE, VT>(p, b, tx, ty, smem, bitsL, RPB);
    // decision words: in LDS, or in global memory written by this CU
    __threadfence_block();
    ALIGNER_STAMP(1);
    backtrack_and_store<(NW <= 4)>(p, b, tx, ty, smem);
    ALIGNER_STAMP(5);
    ALIGNER_STAMP(7);
}

// One piece (a range of rows of one utterance) of the mask's rectangle [0,t_x) x [0,t_y): all ONES of the mask's
// dtype?  (mask_verify_kernel's test with the lengths known; here run by the search launch's zero workgroups.)
template <int VT, int NTHREADS>
__device__ __forceinline__ void verify_mask_piece(const MaxpathParams &p, int q) {
    constexpr int ES = VT == VT_F32 ? 4 : 2, EPV = 16 / ES, NWV = NTHREADS / 64;
    typedef typename std::conditional<ES == 4, unsigned, unsigned short>::type U;
    const unsigned one_bits = VT == VT_F32 ? 0x3F800000u : VT == VT_BF16 ? 0x3F80u : 0x3C00u;
    const unsigned w1 = ES == 4 ? one_bits : (one_bits | (one_bits << 16));
    const int b = q / MV_CHUNKS, c = q - b * MV_CHUNKS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = p.t_xs[b], ty = p.t_ys[b];
    int bad = 0;
    if (!(tx >= 1 && tx <= ty && tx <= p.Tx && ty <= p.Ty)) {
        bad = 1;                                                     // not the normal mode: the multiply stays
    } else {
        const U *mu = static_cast<const U *>(p.mask) + (size_t)b * p.Tx * p.Ty;
        const int per = (tx + MV_CHUNKS - 1) / MV_CHUNKS;
        const int r0 = c * per, r1 = r0 + per < tx ? r0 + per : tx;
        const bool vec = (p.Ty % EPV == 0) && ((reinterpret_cast<uintptr_t>(p.mask) & 15) == 0);
        const int nq = vec ? ty / EPV : 0;
        for (int x = r0 + wave; x < r1; x += NWV) {                  // a wave per row: coalesced 1 KB runs
            const uint4 *row = reinterpret_cast<const uint4 *>(mu + (size_t)x * p.Ty);
            int k = lane;
            for (; k + 192 < nq; k += 256) {                         // four loads in flight per lane
                const uint4 v0 = row[k], v1 = row[k + 64], v2 = row[k + 128], v3 = row[k + 192];
                bad |= (v0.x != w1) | (v0.y != w1) | (v0.z != w1) | (v0.w != w1) | (v1.x != w1) | (v1.y != w1) |
                       (v1.z != w1) | (v1.w != w1) | (v2.x != w1) | (v2.y != w1) | (v2.z != w1) | (v2.w != w1) |
                       (v3.x != w1) | (v3.y != w1) | (v3.z != w1) | (v3.w != w1);
            }
            for (; k < nq; k += 64) {
                const uint4 v = row[k];
                bad |= (v.x != w1) | (v.y != w1) | (v.z != w1) | (v.w != w1);
            }
            for (int y = nq * EPV + lane; y < ty; y += 64) bad |= mu[(size_t)x * p.Ty + y] != (U)one_bits;
        }
    }
    const int any = __syncthreads_or(bad);
    if (tid == 0) p.mflag[q] = any;
}

// The redo of an utterance whose optimistic search ran on unmasked scores: the ones that search wrote into the dense path
// (frames [starts[x], starts[x+1]) of row x, as store_outputs marked them) become zeros again before the strict search
// writes its own.  The erasing stores are drained and the workgroup meets before anything else is stored.
__device__ __forceinline__ void erase_path_ones(const MaxpathParams &p, int b) {
    const int *st = p.starts + (size_t)b * (p.Tx + 1);
    for (int x = threadIdx.x; x < p.Tx; x += blockDim.x) {
        const int s0 = st[x], s1 = st[x + 1];
        for (int y = s0; y < s1 && y < p.Ty; ++y) {
            const size_t idx = ((size_t)b * p.Tx + x) * p.Ty + y;
            switch (p.path1_es) {
                case 1: static_cast<unsigned char *>(p.path1)[idx] = 0; break;
                case 2: static_cast<unsigned short *>(p.path1)[idx] = 0; break;
                case 4: static_cast<unsigned *>(p.path1)[idx] = 0u; break;
                default: static_cast<unsigned long long *>(p.path1)[idx] = 0ull; break;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// The launch.  With p.zero_blocks the grid is zero_blocks + B workgroups: the first ones -- on the CUs a batch of B
// utterances leaves idle -- write the zeros of the dense path (np.zeros, __init__.py:15) while the others search, and
// an utterance's workgroup writes its ones (core.pyx:33) once the zero workgroups have reported (store_outputs):
// the dense path costs no launch of its own and nothing on the step's serial chain.  (A second stream or graph
// branch for the zeros was measured first: the kernels overlap as hoped, but a forked HIP graph costs 15-20 us per
// replay on this stack.)  Zero workgroups never wait, and they come first in dispatch order.
template <int NW, int DEPTH, bool VEC, int MASKMODE, int VT, bool PAIR = false>
__global__ __launch_bounds__(NW * 128, 1) void maxpath_pipelined_kernel(MaxpathParams p) {
    const int Z = p.zero_blocks;
    if ((int)blockIdx.x < Z) {
        // Agent-scope stores (sc1: written through this XCD's L2), then wait for them, then a RELAXED count: no
        // release / acquire anywhere.  A release at agent scope writes the whole L2 back and an acquire invalidates it --
        // per workgroup, under the other workgroups' sweeps: the first version of this, with a fence and a release
        // increment per zero workgroup and an acquire poll per utterance, took 96 us instead of 33.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *d = reinterpret_cast<u32x4 *>(p.path1);
        const u32x4 z = {0u, 0u, 0u, 0u};
        const unsigned long long per = (p.zero_n16 + Z - 1) / Z;
        const unsigned long long lo = per * blockIdx.x, hi = lo + per < p.zero_n16 ? lo + per : p.zero_n16;
        if (p.zero_nt) {
            for (unsigned long long i = lo + threadIdx.x; i < hi; i += NW * 128)
                asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(d + i), "v"(z) : "memory");
        } else {
            for (unsigned long long i = lo + threadIdx.x; i < hi; i += NW * 128)
                asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(d + i), "v"(z) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's zeros have left for memory ...
        __syncthreads();
        if (threadIdx.x == 0 && !(p.flags & ALIGNER_F_TEST_DROP_ZERO_REPORTS))                  // (testing: the zeros are written, never reported)
            __hip_atomic_fetch_add(p.zsync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the workgroup says so
        // Strict mask, optimistic form: the searching workgroups of this launch took the scores as they are (no mask
        // stream through their loaders, which sit at the CU's fetch limit); the zero workgroups, done with the zeros long
        // before the search ends, now check that the multiply would have been the identity (verify_mask_piece) and leave a
        // verdict per piece.  A second launch redoes, with the multiply, exactly the utterances a verdict names.
        if (MASKMODE == 0 && p.verify)
            for (int q = blockIdx.x; q < p.B * MV_CHUNKS; q += Z) verify_mask_piece<VT, NW * 128>(p, q);
    } else if (MASKMODE == 1 && p.redo) {
        const int b = (int)blockIdx.x;
        static_assert(MV_CHUNKS <= 64, "one flag word per lane");
        const int mine = (threadIdx.x & 63) < MV_CHUNKS ? p.mflag[b * MV_CHUNKS + (threadIdx.x & 63)] : 0;
        if (__builtin_amdgcn_ballot_w64(mine != 0) != 0) {           // (uniform: every wave reads the same words)
            if (p.path1) erase_path_ones(p, b);
            maxpath_pipelined_body<NW, DEPTH, VEC, 1, VT, PAIR>(p, b);
        }
    } else if (MASKMODE == 2) {
        // strict mask, decided per utterance on the device: mask_verify_kernel found the mask all ones wherever the
        // search reads a score (a 0/1 prefix rectangle, the usual x_mask * y_mask) -> value * mask IS value there and
        // the mask stream is skipped; anything else takes the multiply (__init__.py:11).  Uniform, outside every loop.
        const int blk = (int)blockIdx.x - Z;
        const int b = PAIR ? (blk >= p.B ? blk - p.B : blk) : blk;
        static_assert(MV_CHUNKS <= 64, "one flag word per lane");
        const int mine = (threadIdx.x & 63) < MV_CHUNKS ? p.mflag[b * MV_CHUNKS + (threadIdx.x & 63)] : 0;
        if (__builtin_amdgcn_ballot_w64(mine != 0) != 0) maxpath_pipelined_body<NW, DEPTH, VEC, 1, VT, PAIR>(p, blk);
        else maxpath_pipelined_body<NW, DEPTH, VEC, 0, VT, PAIR>(p, blk);
    } else {
        maxpath_pipelined_body<NW, DEPTH, VEC, MASKMODE, VT, PAIR>(p, (int)blockIdx.x - Z);
    }
    if (Z > 0) {
        // the last workgroup to finish leaves both counters at 0 for the next launch (launches that share a workspace
        // are ordered on their stream)
        __syncthreads();
        if (threadIdx.x == 0) {
            const int f = __hip_atomic_fetch_add(p.zsync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f == Z + (PAIR ? 2 : 1) * p.B - 1) {
                __hip_atomic_store(p.zsync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.zsync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// --------------------------------------------------------------------------
// starts -> dense 0/1 path in the caller's dtype (the reference's return value,
// __init__.py:21).  Pure streaming store: path[b,x,y] = starts[x] <= y < starts[x+1].
// --------------------------------------------------------------------------
// STREAM: the dense path leaves with non-temporal stores -- 51 MB that nothing on the GPU reads back soon do not push
// the score tensor of the batches in flight out of the caches (bench.py with three batches in flight: 39.9 -> 33.8 us
// per step; alone the kernel is slower, 9.2 -> 11.6 us, so it is the caller's choice: ALIGNER_F_STREAM_PATH)
template <typename T, bool VEC, bool STREAM>
__global__ __launch_bounds__(256) void expand_kernel(const int *__restrict__ starts, T *__restrict__ path,
                                                      int Tx, int Ty, int rows_per_block, T one) {
    const int b = blockIdx.z;
    const int x0 = blockIdx.y * rows_per_block;
    const int y0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y0 >= Ty) return;
    const int x1 = (x0 + rows_per_block < Tx) ? x0 + rows_per_block : Tx;
    const int *st = starts + (size_t)b * (Tx + 1);
    struct alignas(sizeof(T) * 4) Vec4 { T v[4]; };
    int s = st[x0];                                       // block-uniform: scalar loads
    for (int x = x0; x < x1; ++x) {
        const int e = st[x + 1];
        T *dst = path + ((size_t)b * Tx + x) * Ty + y0;
        Vec4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[i] = (y0 + i >= s && y0 + i < e) ? one : T(0);
        if (VEC) {
            if (STREAM) {
                typedef T TV __attribute__((ext_vector_type(4)));
                TV ov;
                ov.x = o.v[0]; ov.y = o.v[1]; ov.z = o.v[2]; ov.w = o.v[3];
                __builtin_nontemporal_store(ov, reinterpret_cast<TV *>(dst));
            } else {
                *reinterpret_cast<Vec4 *>(dst) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (y0 + i < Ty) dst[i] = o.v[i];
        }
        s = e;
    }
}

// --------------------------------------------------------------------------
// lengths from the mask (__init__.py:18-19)
// --------------------------------------------------------------------------
template <typename T, int VT> __device__ __forceinline__ float mask_value(T v) {
    if (VT == VT_F32) return (float)v;
    return half_to_f32<VT>((unsigned)v);                          // T = unsigned short holding bf16 / fp16 bits
}

template <typename T, int VT = VT_F32>
__global__ __launch_bounds__(256) void lengths_kernel(const T *__restrict__ mask, int Tx, int Ty,
                                                       int *__restrict__ t_xs, int *__restrict__ t_ys) {
    __shared__ float red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const T *m = mask + (size_t)b * Tx * Ty;
    float sx = 0.f, sy = 0.f;
    for (int x = tid; x < Tx; x += 256) sx += mask_value<T, VT>(m[(size_t)x * Ty]);   // mask[b, x, 0]
    for (int y = tid; y < Ty; y += 256) sy += mask_value<T, VT>(m[y]);                // mask[b, 0, y]
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sx += __shfl_down(sx, off);
        sy += __shfl_down(sy, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = sx; red[1][tid >> 6] = sy; }
    __syncthreads();
    if (tid == 0) {
        t_xs[b] = (int)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);   // astype(np.int32)
        t_ys[b] = (int)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// Is `value * mask` (__init__.py:11) the identity on every score the search reads?  The search of a normal utterance
// (1 <= t_x <= t_y) reads the cells [0,t_x) x [0,t_y) and nothing else, so it is when the mask holds a ONE
// of its dtype in all of them -- which the usual x_mask[:, :, None] * y_mask[:, None, :] does.  One full-rate pass over
// those cells; piece c of utterance b (a range of rows) leaves its verdict in flag[b][c] (a plain store: every word is
// rewritten by every call, nothing to reset).  The verdict may err on the safe side only ("multiply"): a piece does
// not know t_x (the strided column sum of __init__.py:18 is one round trip too many for 2048 workgroups: only piece
// 0 takes it, for the search's lengths), it takes a row for inside the rectangle when the row's first element is
// ONE and asks that column 0 over its rows (and the row before them) is ONES THEN ZEROS -- then "first element ONE" is
// "x < t_x" and the test is exact; any other column 0 is flagged.  t_y is the reference's row sum.  Utterances in the
// reference's degenerate modes (t_x > t_y walks raw scores outside the rectangle; empty ones) keep the multiply: piece 0
// sees the lengths and says so.
template <int ES, int VT>
__global__ __launch_bounds__(256) void mask_verify_kernel(const void *__restrict__ maskv, int Tx, int Ty,
                                                           const int *__restrict__ t_xs_in, const int *__restrict__ t_ys_in,
                                                           int *__restrict__ t_xs_out, int *__restrict__ t_ys_out,
                                                           int *__restrict__ flag, unsigned one_bits) {
    __shared__ float red[2][4];
    __shared__ int anybad;
    typedef typename std::conditional<ES == 4, float, unsigned short>::type E;      // (mask_value's element types)
    typedef typename std::conditional<ES == 4, unsigned, unsigned short>::type U;
    constexpr int EPV = 16 / ES;                                     // elements per 16-byte load
    constexpr int NQ = 4;                                            // 16-byte loads in flight per lane
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const E *m = static_cast<const E *>(maskv) + (size_t)b * Tx * Ty;
    const U *mu = reinterpret_cast<const U *>(m);
    const U one = (U)one_bits;
    const unsigned w1 = ES == 4 ? one_bits : (one_bits | (one_bits << 16));
    if (tid == 0) anybad = 0;
    const bool given = t_xs_in != nullptr;
    int tx = given ? t_xs_in[b] : -1, ty = given ? t_ys_in[b] : 0;
    const int rows = given ? (tx < Tx ? (tx > 0 ? tx : 0) : Tx) : Tx;       // given lengths: rows [0, t_x) exactly
    const int per = (rows + MV_CHUNKS - 1) / MV_CHUNKS;
    const int r0 = c * per, r1 = r0 + per < rows ? r0 + per : rows;
    const bool vec = (Ty % EPV == 0) && ((reinterpret_cast<uintptr_t>(maskv) & 15) == 0);
    const int nqrow = Ty / EPV;                                      // 16-byte pieces of a whole row
    // One row of the piece per wave and round: its first NQ x 64 pieces are asked for BEFORE anything is known about
    // the lengths (the loads' addresses do not depend on them) -- a piece is two dependent memory round trips
    // (row sum -> t_y, rows), not five.
    auto fetch = [&](int x, uint4 (&v)[NQ], U &before) {
        const uint4 *row = reinterpret_cast<const uint4 *>(mu + (size_t)x * Ty);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = lane + 64 * i;
            v[i] = row[q < nqrow ? q : nqrow - 1];                  // (clamped: unconditional loads)
        }
        before = x > 0 ? mu[(size_t)(x - 1) * Ty] : one;
    };
    uint4 v[NQ];
    U before = one;
    int x = r0 + wave;
    if (vec && x < r1) fetch(x, v, before);
    if (!given) {
        // t_y: the row sum (every piece: 4 KB, contiguous); t_x: the column sum (piece 0 only, for the search)
        float sx = 0.f, sy = 0.f;
        for (int y = tid; y < Ty; y += 256) sy += mask_value<E, VT>(m[y]);                      // mask[b, 0, y]
        if (c == 0)
            for (int xx = tid; xx < Tx; xx += 256) sx += mask_value<E, VT>(m[(size_t)xx * Ty]);   // mask[b, x, 0]
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sx += __shfl_down(sx, off);
            sy += __shfl_down(sy, off);
        }
        if (lane == 0) { red[0][wave] = sx; red[1][wave] = sy; }
    }
    __syncthreads();
    if (!given) {
        ty = (int)(red[1][0] + red[1][1] + red[1][2] + red[1][3]);                             // astype(np.int32),
        if (c == 0) {                                                                           // summed as lengths_kernel does
            tx = (int)(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            if (tid == 0) { t_xs_out[b] = tx; t_ys_out[b] = ty; }
        }
    }
    int bad = 0;
    if (tx >= 0 && !(tx >= 1 && tx <= ty && tx <= Tx && ty <= Ty)) bad = 1;     // (who knows t_x) not the normal mode
    if (ty < 1 || ty > Ty) bad = 1;
    if (!bad) {
        const int nq = vec ? ty / EPV : 0;                           // whole 16-byte pieces inside the row's valid part
        for (; x < r1; x += 4) {                                     // a wave per row: coalesced 1 KB runs
            bool inside = true;
            if (vec) {
                const unsigned f0 = __builtin_amdgcn_readfirstlane(v[0].x);     // the row's first element (s)
                const U first = (U)(ES == 4 ? f0 : (f0 & 0xFFFFu));
                if (!given) {
                    // column 0 over the piece: ONE or zero, never a ONE behind a zero (the row before the piece included)
                    bad |= (first != one && first != 0) | (first == one && before != one);
                    inside = first == one;                           // (wave-uniform)
                }
                if (inside) {
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
                        if (lane + 64 * i < nq) bad |= (v[i].x != w1) | (v[i].y != w1) | (v[i].z != w1) | (v[i].w != w1);
                    const uint4 *row = reinterpret_cast<const uint4 *>(mu + (size_t)x * Ty);
                    for (int q = lane + 64 * NQ; q < nq; q += 64) {  // rows longer than NQ x 64 pieces
                        const uint4 w = row[q];
                        bad |= (w.x != w1) | (w.y != w1) | (w.z != w1) | (w.w != w1);
                    }
                }
                if (x + 4 < r1) fetch(x + 4, v, before);
            } else if (!given) {
                const U first = mu[(size_t)x * Ty], bf = x > 0 ? mu[(size_t)(x - 1) * Ty] : one;
                bad |= (first != one && first != 0) | (first == one && bf != one);
                inside = first == one;
            }
            if (inside)
                for (int y = nq * EPV + lane; y < ty; y += 64) bad |= mu[(size_t)x * Ty + y] != one;
        }
    }
    if (bad) anybad = 1;
    __syncthreads();
    if (tid == 0) flag[b * MV_CHUNKS + c] = anybad;
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------

struct WsLayout {
    size_t status_off, len_off, mflag_off, starts_off, bits_off, dump_off, xring_off, xring_bytes, total;
    int NT, ROWS;
};

static WsLayout ws_layout(int B, int Tx, int Ty) {
    WsLayout L;
    L.NT = (Ty + TC - 1) / TC;
    // one word per text row and tile; rounded so whole waves / blocks of lanes can store
    // without a bounds test (pipelined: 63*NW rows, generic: 256*R rows)
    const int rows_pipe = (Tx + RPW - 1) / RPW * RPW + 1;
    const int rows_plain = (Tx + 255) / 256 * 256;
    L.ROWS = ((rows_pipe > rows_plain ? rows_pipe : rows_plain) + 63) / 64 * 64;
    L.status_off = 0;
    L.len_off = WS_HDR_BYTES;
    L.mflag_off = L.len_off + (size_t)2 * B * sizeof(int);          // [B][MV_CHUNKS] mask_verify_kernel's verdicts
    L.starts_off = align_up(L.mflag_off + (size_t)B * MV_CHUNKS * sizeof(int), 256);
    L.bits_off = align_up(L.starts_off + (size_t)B * (Tx + 1) * sizeof(int), 256);
    L.dump_off = align_up(L.bits_off + (size_t)B * L.NT * L.ROWS * sizeof(unsigned), 256);
    // two workgroups per utterance: the boundary row between them, 32 words per tile, and a done word (PAIR)
    L.xring_off = align_up(L.dump_off + (size_t)B * 4 * 64 * sizeof(float), 256);
    L.xring_bytes = align_up((size_t)B * L.NT * TC * sizeof(unsigned) + (size_t)2 * B * sizeof(int), 256);   // ring, done words, walk hand-over words
    L.total = L.xring_off + L.xring_bytes;
    return L;
}

static int lds_limit() { return device_lds_limit(); }

static size_t starts_bytes(int Tx) { return (size_t)(((Tx + 1 + 63) / 64) * 64 + 64) * 4; }

// Backtrack overlay size for a window of WT tiles (decision words in global memory).
static size_t walk_bytes(int WT, int ROWS, int Tx) { return starts_bytes(Tx) + (size_t)WT * row_pitch(ROWS) * 4; }

static int pick_window(int NT, int ROWS, int Tx, size_t budget) {
    int WT = NT < 64 ? NT : 64;                      // (successive windows share a tile: at least two, or all of them)
    const int least = NT < 2 ? NT : 2;
    while (WT > least && walk_bytes(WT, ROWS, Tx) > budget) --WT;
    return walk_bytes(WT, ROWS, Tx) <= budget ? WT : 0;
}

template <typename K>
static int launch_with_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, MaxpathParams p) {
    p.lds_total = (int)lds;
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), lds));
    hipLaunchKernelGGL(kernel, grid, block, lds, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

template <int NW, int DEPTH, bool VEC, int VT>
static int launch_pipelined_mm(MaxpathParams p, int maskmode, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    if (maskmode == 0) return launch_with_lds(maxpath_pipelined_kernel<NW, DEPTH, VEC, 0, VT>, grid, block, lds, s, p);
    if (NW <= 4) {
        constexpr int N4 = NW <= 4 ? NW : 4;
        if (maskmode == 2) return launch_with_lds(maxpath_pipelined_kernel<N4, DEPTH, VEC, 2, VT>, grid, block, lds, s, p);
        return launch_with_lds(maxpath_pipelined_kernel<N4, DEPTH, VEC, 1, VT>, grid, block, lds, s, p);
    }
    // eight waves (128 registers a lane) with the mask stream: one tile in flight per loader (with two the loaders' 2 x 2 x 16
    // registers spilled: 30-41 VGPRs, 104-116 bytes of scratch a lane)
    return launch_with_lds(maxpath_pipelined_kernel<NW, 1, VEC, 1, VT>, grid, block, lds, s, p);
}

template <int NW, int DEPTH>
static int launch_pipelined(MaxpathParams p, bool vec, int maskmode, int vt, size_t lds, hipStream_t s) {
    dim3 grid(p.B + p.zero_blocks), block(NW * 128);
    if (vt != VT_F32) {
        // 16-bit scores: 16-byte loaders only (the caller routes everything else to the generic kernel),
        // and only the two wide workgroup shapes are built (narrow text runs on NW = 4 with idle waves)
        if (NW >= 4) {
            if (vt == VT_BF16) return launch_pipelined_mm<(NW >= 4 ? NW : 4), DEPTH, true, VT_BF16>(p, maskmode, grid, block, lds, s);
            return launch_pipelined_mm<(NW >= 4 ? NW : 4), DEPTH, true, VT_F16>(p, maskmode, grid, block, lds, s);
        }
        return fail(ALIGNER_EINVAL, "internal: 16-bit scores on a narrow workgroup");
    }
    if (vec) return launch_pipelined_mm<NW, DEPTH, true, VT_F32>(p, maskmode, grid, block, lds, s);
    return launch_pipelined_mm<NW, DEPTH, false, VT_F32>(p, maskmode, grid, block, lds, s);
}

__global__ __launch_bounds__(256) void xring_fill_kernel(uint4 *dst, int n16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = make_uint4(XRING_EMPTY, XRING_EMPTY, XRING_EMPTY, XRING_EMPTY);
}

// two workgroups per utterance (grid 2B), 16-byte loaders only
template <int VT>
static int launch_pair_mm(MaxpathParams p, int maskmode, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    if (maskmode == 0) return launch_with_lds(maxpath_pipelined_kernel<4, 2, true, 0, VT, true>, grid, block, lds, s, p);
    return launch_with_lds(maxpath_pipelined_kernel<4, 2, true, 1, VT, true>, grid, block, lds, s, p);
}
static int launch_pair(MaxpathParams p, int maskmode, int vt, size_t lds, hipStream_t s) {
    dim3 grid(2 * p.B + p.zero_blocks), block(4 * 128);
    if (vt == VT_BF16) return launch_pair_mm<VT_BF16>(p, maskmode, grid, block, lds, s);
    if (vt == VT_F16) return launch_pair_mm<VT_F16>(p, maskmode, grid, block, lds, s);
    return launch_pair_mm<VT_F32>(p, maskmode, grid, block, lds, s);
}

template <int R, int VT>
static int launch_generic_vt(MaxpathParams p, int maskmode, size_t lds, hipStream_t s) {
    dim3 grid(p.B), block(256);
    if (maskmode == 0) return launch_with_lds(maxpath_generic_kernel<R, 0, VT>, grid, block, lds, s, p);
    return launch_with_lds(maxpath_generic_kernel<R, 1, VT>, grid, block, lds, s, p);
}

template <int R>
static int launch_generic(MaxpathParams p, int maskmode, int vt, size_t lds, hipStream_t s) {
    if (vt == VT_BF16) return launch_generic_vt<R, VT_BF16>(p, maskmode, lds, s);
    if (vt == VT_F16) return launch_generic_vt<R, VT_F16>(p, maskmode, lds, s);
    return launch_generic_vt<R, VT_F32>(p, maskmode, lds, s);
}

static void set_path_ones(MaxpathParams &p, void *path, int path_dtype) {
    p.path1 = path;
    p.path1_es = dtype_size(path_dtype);
    switch (path_dtype) {
        case ALIGNER_DT_F32: p.path1_one = 0x3F800000ull; break;
        case ALIGNER_DT_F64: p.path1_one = 0x3FF0000000000000ull; break;
        case ALIGNER_DT_F16: p.path1_one = 0x3C00ull; break;
        case ALIGNER_DT_BF16: p.path1_one = 0x3F80ull; break;
        default: p.path1_one = 1ull; break;
    }
}

static int forward_impl(const void *value, int value_dtype, const void *mask, int mask_dtype, const int32_t *t_xs,
                        const int32_t *t_ys, int32_t *tok_out, int32_t *dur_out, void *ws,
                        size_t ws_bytes, int B, int Tx, int Ty, float neg, int flags, hipStream_t s,
                        void *path_out = nullptr, int path_dtype = 0, bool path_is_zero = false, bool *path_done = nullptr,
                        int ldv = 0) {
    // path_out: the dense path this launch may finish itself -- its ones always (path_is_zero: the caller's zeros,
    // ALIGNER_F_PATH_PREZEROED), its zeros too where the pipelined kernel runs with zero workgroups; *path_done says
    // whether it did (otherwise the caller launches expand)
    if (path_done) *path_done = false;
    void *path_prezeroed = path_is_zero ? path_out : nullptr;
    if (!value || !ws) return fail(ALIGNER_EINVAL, "value/workspace pointer is null");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if ((size_t)Tx * (size_t)Ty >= (1ull << 31))
        return fail(ALIGNER_EDOM, "Tx*Ty=%zu exceeds 2^31", (size_t)Tx * Ty);
    if ((!t_xs || !t_ys) && !mask)
        return fail(ALIGNER_EINVAL, "need either lengths (t_xs,t_ys) or a mask to derive them from");
    if ((flags & ALIGNER_F_STRICT_MASK) && !mask)
        return fail(ALIGNER_EINVAL, "ALIGNER_F_STRICT_MASK needs a mask");
    const int vt = value_dtype == ALIGNER_DT_F32 ? VT_F32 : value_dtype == ALIGNER_DT_BF16 ? VT_BF16
                 : value_dtype == ALIGNER_DT_F16 ? VT_F16 : -1;
    if (vt < 0) return fail(ALIGNER_EINVAL, "value dtype %d not supported (F32, BF16, F16)", value_dtype);
    if ((flags & ALIGNER_F_STRICT_MASK) && mask_dtype != value_dtype)
        return fail(ALIGNER_EINVAL, "strict mask must have the scores' dtype (mask %d, value %d): the product is "
                                    "rounded in that dtype like torch's value * mask", mask_dtype, value_dtype);
    // ldv: the scores' row pitch (aligner_maxpath_ld).  A pitch of its own comes with lengths and without a mask: it is
    // the layout of the pipeline's own intermediate, not of the reference's tensors
    if (ldv == 0) ldv = Ty;
    if (ldv < Ty) return fail(ALIGNER_EINVAL, "ld_value=%d < Ty=%d", ldv, Ty);
    if (ldv != Ty && (mask || !t_xs || !t_ys || (flags & ALIGNER_F_WRITE_Q)))
        return fail(ALIGNER_EINVAL, "a row pitch of its own (ld_value=%d, Ty=%d) takes lengths, no mask and no ALIGNER_F_WRITE_Q", ldv, Ty);
    if ((size_t)Tx * (size_t)ldv >= (1ull << 31)) return fail(ALIGNER_EDOM, "Tx*ld_value=%zu exceeds 2^31", (size_t)Tx * ldv);
    if (B == 0) return ALIGNER_OK;
    const WsLayout L = ws_layout(B, Tx, Ty);
    if (ws_bytes < L.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", ws_bytes, L.total);
    unsigned char *wsb = static_cast<unsigned char *>(ws);

    // Strict mask on the kernels that can decide per utterance (the pipelined forms): one pass over the mask's
    // rectangle verifies that the multiply is the identity there (and derives the lengths on the way), and the search
    // then skips the mask stream (MASKMODE 2)
    // (up to four waves of text rows: the eight-wave workgroup has 128 registers a lane and its strict form keeps one
    // tile in flight instead of two to stay inside them; the two-workgroup form keeps the plain multiply)
    bool dyn_mask = (flags & ALIGNER_F_STRICT_MASK) && !(flags & (ALIGNER_F_FORCE_GENERIC | ALIGNER_F_WRITE_Q)) &&
                    (Tx + RPW - 1) / RPW <= 4 && !g_opt_maxpath_no_mask_verify;
    // ... and where the search launch has zero workgroups (it writes the dense path on the CUs the batch leaves idle),
    // THEY verify the mask, beside an optimistic search, and a second launch redoes the utterances they flag: no pass
    // over the mask in front of the search at all
    const bool opt_mask = dyn_mask && path_out && !path_is_zero && !(flags & ALIGNER_F_SEPARATE_EXPAND) &&
                          device_cu_count() - B >= 32 && (reinterpret_cast<uintptr_t>(path_out) & 15) == 0 &&
                          ((size_t)B * Tx * Ty * dtype_size(path_dtype)) % 16 == 0 && dtype_size(path_dtype) > 0 &&
                          (vt == VT_F32 || (Ty % 8 == 0 && (reinterpret_cast<uintptr_t>(value) & 15) == 0)) &&
                          !g_opt_maxpath_no_optimistic_mask;
    if (opt_mask) dyn_mask = false;
    if (dyn_mask) {
        int *lx = reinterpret_cast<int *>(wsb + L.len_off), *ly = lx + B;
        int *mf = reinterpret_cast<int *>(wsb + L.mflag_off);
        const bool derive = !t_xs || !t_ys;
        dim3 grid(MV_CHUNKS, B), block(256);
        if (vt == VT_F32)
            hipLaunchKernelGGL((mask_verify_kernel<4, VT_F32>), grid, block, 0, s, mask, Tx, Ty, derive ? nullptr : t_xs,
                               derive ? nullptr : t_ys, lx, ly, mf, 0x3F800000u);
        else if (vt == VT_BF16)
            hipLaunchKernelGGL((mask_verify_kernel<2, VT_BF16>), grid, block, 0, s, mask, Tx, Ty, derive ? nullptr : t_xs,
                               derive ? nullptr : t_ys, lx, ly, mf, 0x3F80u);
        else
            hipLaunchKernelGGL((mask_verify_kernel<2, VT_F16>), grid, block, 0, s, mask, Tx, Ty, derive ? nullptr : t_xs,
                               derive ? nullptr : t_ys, lx, ly, mf, 0x3C00u);
        ALIGNER_HIP_CHECK(hipGetLastError());
        if (derive) { t_xs = lx; t_ys = ly; }
    } else if (!t_xs || !t_ys) {
        int *lx = reinterpret_cast<int *>(wsb + L.len_off), *ly = lx + B;
        int rc = aligner_lengths_from_mask(mask, mask_dtype, B, Tx, Ty, lx, ly, s);
        if (rc) return rc;
        t_xs = lx; t_ys = ly;
    }

    MaxpathParams p;
    p.value = value;
    p.mask = (flags & ALIGNER_F_STRICT_MASK) ? mask : nullptr;
    p.t_xs = t_xs; p.t_ys = t_ys;
    p.starts = reinterpret_cast<int *>(wsb + L.starts_off);
    p.tok = tok_out;
    p.dur = dur_out;
    p.bits = reinterpret_cast<unsigned *>(wsb + L.bits_off);
    p.status = reinterpret_cast<int *>(wsb + L.status_off);
    p.mflag = reinterpret_cast<int *>(wsb + L.mflag_off);
    p.verify = 0; p.redo = 0; p.ldv = ldv;
    p.B = B; p.Tx = Tx; p.Ty = Ty; p.NT = L.NT; p.ROWS = L.ROWS;
    p.neg = neg; p.flags = flags;
    p.bits_in_lds = 0; p.lds_bits_off = 0; p.lds_prev_off = 0;
    p.force_exact = !(neg - neg == 0.0f);            // NaN / inf max_neg_val
    p.stamps = g_debug_stamps;
    p.dump = nullptr;
    p.qout = nullptr;
    p.path1 = nullptr; p.path1_es = 0; p.path1_one = 0;
    if (path_prezeroed) set_path_ones(p, path_prezeroed, path_dtype);
    p.zero_blocks = 0; p.zero_nt = 0; p.zero_n16 = 0;
    p.zsync = reinterpret_cast<int *>(wsb + L.status_off + 16);
    if (path_prezeroed && path_done) *path_done = true;
    p.xring = reinterpret_cast<unsigned *>(wsb + L.xring_off);
    p.xflag = reinterpret_cast<int *>(wsb + L.xring_off + (size_t)B * L.NT * TC * sizeof(unsigned));
    p.xwalk = reinterpret_cast<unsigned *>(p.xflag + B);
    p.split_walk = 0;
    if (flags & ALIGNER_F_WRITE_Q) {
        if (vt != VT_F32 || (flags & ALIGNER_F_STRICT_MASK))
            return fail(ALIGNER_EINVAL, "ALIGNER_F_WRITE_Q takes fp32 scores without a strict mask (maximum_path_c's contract)");
        p.qout = const_cast<float *>(static_cast<const float *>(value));
        flags |= ALIGNER_F_FORCE_GENERIC;           // the pipelined kernel never materialises Q
        p.flags = flags;
    }
    const int maskmode = (flags & ALIGNER_F_STRICT_MASK) ? 1 : 0;
    const int pipemask = dyn_mask ? 2 : opt_mask ? 0 : maskmode;        // the pipelined launches: decided per utterance on the device
    const size_t lds_max = (size_t)lds_limit();
    // vec: 16-byte loads; the pipelined kernel's loaders address an utterance with 32-bit byte offsets
    const int per16 = vt == VT_F32 ? 4 : 8;                      // scores per 16-byte load
    const bool vec = (Ty % per16 == 0) && (ldv % per16 == 0) && ((reinterpret_cast<uintptr_t>(value) & 15) == 0) &&
                     (!maskmode || (reinterpret_cast<uintptr_t>(mask) & 15) == 0) &&
                     (size_t)Tx * (size_t)ldv * 4 < (1ull << 31);

    const int nw_need = (Tx + RPW - 1) / RPW;
    // Long text (5..8 waves of rows) on a batch that leaves CUs idle: two workgroups per utterance, each the
    // four-wave form with one compute wave per SIMD (maxpath_pipelined_kernel<.., PAIR>).  A full machine gains
    // nothing from the split, and the second half runs some twenty phases behind the first (read-ahead, skew,
    // forwarding, the boundary row's trip through memory).  Measured on full-length batches (in-kernel, two
    // workgroups / one): [8,500,4000] 106 / 134 us, [8,500,2048] 68 / 76, [8,500,1536] 59 / 63, [8,500,1024] 50 / 50,
    // [8,300,1536] 49 / 48; [16,400,2000] 65 / 75, [32,400,2000] 68 / 75, [64,400,2000] 74 / 76, [64,500,2048] 79 / 79:
    // taken from 56 tiles on, for batches of at most an eighth of the CU count.
    if (!(flags & (ALIGNER_F_FORCE_GENERIC | ALIGNER_F_ONE_CU)) && nw_need > 4 && nw_need <= 8 && vec && ldv == Ty &&
        2 * B <= device_cu_count() &&      // both halves of every utterance resident at once (see the kernel)
        ((flags & ALIGNER_F_TWO_CUS) || (8 * B <= device_cu_count() && L.NT >= 56))) {
        const size_t fwd = align_up(((size_t)4 * (2 * 64 * TILE_LD + RING_T * RING_LD) + RING_T * RING_LD) * 4 + 16, 16);
        p.WT = pick_window(L.NT, L.ROWS, Tx, lds_max);
        if (p.WT > 0 && fwd <= lds_max && starts_bytes(Tx) <= fwd) {
            size_t lds = walk_bytes(p.WT, L.ROWS, Tx);
            if (lds < fwd) lds = fwd;
            // each half walks its own rows when all its decision words (NT tiles x 256 rows) fit in LDS beside the token
            // starts (see the kernel); "maxpath_no_split_walk" keeps the one-walker form for A/B runs
            const size_t split_lds = starts_bytes(Tx) + (size_t)L.NT * row_pitch(256) * 4;
            if (split_lds <= lds_max && !g_opt_maxpath_no_split_walk) {
                p.split_walk = 1;
                if (lds < split_lds) lds = split_lds;
            }
            // the boundary ring and the done words start every launch as 0xFFFFFFFF (a kernel of our own: a captured
            // hipMemsetAsync node did not refill the ring when the graph was replayed)
            const int n16 = (int)(L.xring_bytes / 16);
            hipLaunchKernelGGL(xring_fill_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s,
                               reinterpret_cast<uint4 *>(wsb + L.xring_off), n16);
            ALIGNER_HIP_CHECK(hipGetLastError());
            {   // the dense path inside this launch (see the kernel): zero workgroups on the CUs the 2B halves leave idle
                const int cus = device_cu_count();
                const size_t pbytes = path_out ? (size_t)B * Tx * Ty * dtype_size(path_dtype) : 0;
                if (path_out && !path_is_zero && !(flags & ALIGNER_F_SEPARATE_EXPAND) && cus - 2 * B >= 32 &&
                    (reinterpret_cast<uintptr_t>(path_out) & 15) == 0 && pbytes % 16 == 0 && dtype_size(path_dtype) > 0) {
                    // 48 zero workgroups, not one per idle CU: the zeros are not needed before the outputs (~90 us into the
                    // launch), and 240 workgroups flooding the write path for the first 13 us kept the 16 searching
                    // workgroups from priming their loaders.  [8,500,4000] bf16, int32 path, HIP events: 117.3 us with
                    // all idle CUs, 113.6 us with any count from 16 to 128 (durations only: 110.0 us)
                    p.zero_blocks = cus - 2 * B < 48 ? cus - 2 * B : 48;
                    if (g_opt_maxpath_zero_blocks > 0 && g_opt_maxpath_zero_blocks <= cus - 2 * B) p.zero_blocks = g_opt_maxpath_zero_blocks;
                    p.zero_nt = (flags & ALIGNER_F_STREAM_PATH) ? 1 : 0;
                    p.zero_n16 = pbytes / 16;
                    set_path_ones(p, path_out, path_dtype);
                    if (path_done) *path_done = true;
                }
            }
            return launch_pair(p, pipemask, vt, lds, s);
        }
    }
    if (!(flags & ALIGNER_F_FORCE_GENERIC) && nw_need <= 8 && (vt == VT_F32 || vec)) {
        const int NW = (nw_need <= 4 && vt != VT_F32) ? 4 : nw_need <= 1 ? 1 : nw_need <= 2 ? 2 : nw_need <= 4 ? 4 : 8;
        const size_t fwd = align_up((size_t)NW * (2 * 64 * TILE_LD + RING_T * RING_LD) * 4 + 16, 16);
        if (fwd <= lds_max && starts_bytes(Tx) <= fwd) {
            size_t lds = 0;
            const size_t bits_lds = (size_t)L.NT * row_pitch(L.ROWS) * 4;
            if (L.NT <= 64 && fwd + bits_lds <= lds_max) {
                p.bits_in_lds = 1;
                p.lds_bits_off = (int)fwd;
                p.WT = L.NT;
                lds = fwd + bits_lds;
                // room for the P table as well (and a walk with (W, P) in 128 + 64 VGPRs: NW <= 4): the
                // backtrack's steps then cannot fail (walk_chunk_prev)
                const size_t prev_lds = bits_lds + (size_t)row_pitch(L.ROWS) * 4;     // NT + 1 slots: tile t's entry in slot t + 1
                if (NW <= 4 && L.NT < 64 && lds + prev_lds <= lds_max && !(flags & ALIGNER_F_NO_PREV_TABLE)) {
                    p.lds_prev_off = (int)lds;
                    lds += prev_lds;
                }
            } else {
                p.WT = pick_window(L.NT, L.ROWS, Tx, lds_max);
                if (p.WT > 0) {
                    lds = walk_bytes(p.WT, L.ROWS, Tx);
                    if (lds < fwd) lds = fwd;
                }
            }
            if (lds) {
                // the dense path inside this launch: zero workgroups on the CUs the batch leaves idle (see the kernel)
                const int cus = device_cu_count();
                const size_t pbytes = path_out ? (size_t)B * Tx * Ty * dtype_size(path_dtype) : 0;
                if (path_out && !path_is_zero && !(flags & ALIGNER_F_SEPARATE_EXPAND) && cus - B >= 32 &&
                    (reinterpret_cast<uintptr_t>(path_out) & 15) == 0 && pbytes % 16 == 0 && dtype_size(path_dtype) > 0) {
                    // As many zero workgroups as write the zeros in about half the search's time (a CU streams ~50 GB/s, a
                    // frame of the search takes ~31 ns: bytes / (775 * Ty)), not one per idle CU: the zeros are not needed
                    // before the outputs, and the CUs they leave alone are other batches' (bench.py, fused form, three batches
                    // in flight: 36.8-37.7 us a step with all 192, 33.7 with 64 or 128; ONE batch at a time: 55.7 -> 53.7 us;
                    // 32 are too few: 61 us).  With the mask to verify they all come: the pieces are their work too.
                    p.zero_blocks = cus - B;
                    if (!opt_mask) {
                        const long long need = (long long)(pbytes / ((size_t)775 * (size_t)Ty)) + 1;
                        const int zc = need < 48 ? 48 : (int)(need > cus - B ? cus - B : need);
                        if (zc < p.zero_blocks) p.zero_blocks = zc;
                    }
                    if (g_opt_maxpath_zero_blocks >= 32 && g_opt_maxpath_zero_blocks <= cus - B) p.zero_blocks = g_opt_maxpath_zero_blocks;
                    p.zero_nt = (flags & ALIGNER_F_STREAM_PATH) ? 1 : 0;
                    p.zero_n16 = pbytes / 16;
                    set_path_ones(p, path_out, path_dtype);
                    if (path_done) *path_done = true;
                }
                auto launch = [&](const MaxpathParams &q, int mm) {
                    switch (NW) {
                        case 1: return launch_pipelined<1, 4>(q, vec, mm, vt, lds, s);
                        case 2: return launch_pipelined<2, 4>(q, vec, mm, vt, lds, s);
                        case 4: return launch_pipelined<4, 2>(q, vec, mm, vt, lds, s);
                        default: return launch_pipelined<8, 2>(q, vec, mm, vt, lds, s);
                    }
                };
                if (opt_mask && p.zero_blocks > 0) {
                    p.mask = mask;
                    p.verify = 1;
                    int rc = launch(p, 0);                         // optimistic search + zeros + verification
                    if (rc != ALIGNER_OK) return rc;
                    MaxpathParams r = p;                           // the redo of the flagged utterances: ones only, no zero workgroups
                    r.verify = 0; r.redo = 1; r.zero_blocks = 0; r.zero_n16 = 0;
                    return launch(r, 1);
                }
                return launch(p, opt_mask ? maskmode : pipemask);   // (no zero workgroups after all: the plain multiply)
            }
        }
    }
    // generic path
    const int R = (Tx + 255) / 256;
    if (R > 8) return fail(ALIGNER_EDOM, "Tx=%d exceeds the 2048 text rows the kernels support", Tx);
    const int RR = R <= 1 ? 1 : R <= 2 ? 2 : R <= 4 ? 4 : 8;
    const size_t fwd = (size_t)2 * (256 * RR + 1) * 4;
    p.WT = pick_window(L.NT, L.ROWS, Tx, lds_max);
    if (p.WT <= 0) return fail(ALIGNER_EDOM, "Tx=%d/Ty=%d too large for the backtrack window", Tx, Ty);
    size_t lds = walk_bytes(p.WT, L.ROWS, Tx);
    if (lds < fwd) lds = fwd;
    switch (RR) {
        case 1: return launch_generic<1>(p, maskmode, vt, lds, s);
        case 2: return launch_generic<2>(p, maskmode, vt, lds, s);
        case 4: return launch_generic<4>(p, maskmode, vt, lds, s);
        default: return launch_generic<8>(p, maskmode, vt, lds, s);
    }
}

template <typename T>
static int launch_expand(const int *starts, void *path, int B, int Tx, int Ty, T one, bool stream_path, hipStream_t s) {
    // Rows per workgroup.  Alone, many small workgroups are fastest (8 rows: 8.4 us for the 51 MB of [64,200,1000] fp32; 50 rows:
    // 9.6).  With ALIGNER_F_STREAM_PATH the caller says that other batches are in flight: then about one workgroup per CU
    // (50 rows at that shape) -- the same write stream from a quarter of the wave slots, the rest left to the other
    // batches' kernels: bench.py's three-batch step 34.6-34.9 -> 32.9-33.0 us.
    int rpb = 8;
    if (stream_path) {
        const long long cells = (long long)B * ((Ty + 1023) / 1024) * Tx;
        const int want = (int)((cells + device_cu_count() - 1) / device_cu_count());
        rpb = want < 8 ? 8 : (want > Tx ? Tx : want);
    }
    dim3 grid((Ty + 1023) / 1024, (Tx + rpb - 1) / rpb, B), block(256);
    const bool vec = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(path) % (sizeof(T) * 4)) == 0);
    if (vec && stream_path)
        hipLaunchKernelGGL((expand_kernel<T, true, true>), grid, block, 0, s, starts, static_cast<T *>(path), Tx, Ty, rpb, one);
    else if (vec)
        hipLaunchKernelGGL((expand_kernel<T, true, false>), grid, block, 0, s, starts, static_cast<T *>(path), Tx, Ty, rpb, one);
    else
        hipLaunchKernelGGL((expand_kernel<T, false, false>), grid, block, 0, s, starts, static_cast<T *>(path), Tx, Ty, rpb, one);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

// --------------------------------------------------------------------------
// The dense path the way the reference builds it: np.zeros (__init__.py:15), then core.pyx:33's `path[index, y] = 1`.
// The zeros do not depend on the alignment, so a caller with a second stream (or a graph branch) writes them
// WHILE the search runs, and all that is left behind the search are t_y stores per utterance -- the 51 MB write of
// expand_kernel leaves the serial chain of a step (bench.py: one batch at a time).
// --------------------------------------------------------------------------
template <bool STREAM>
__global__ __launch_bounds__(256) void zero_path_kernel(uint4 *__restrict__ dst, size_t n16, unsigned char *__restrict__ tail,
                                                         int ntail) {
    const size_t stride = (size_t)gridDim.x * 256;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 z = {0u, 0u, 0u, 0u};
    u32x4 *d = reinterpret_cast<u32x4 *>(dst);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        if (STREAM) __builtin_nontemporal_store(z, d + i);
        else d[i] = z;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

// one thread per frame: the token whose [start, next start) holds it (bisection over the starts, staged in LDS)
template <typename T>
__global__ __launch_bounds__(256) void scatter_path_kernel(const int *__restrict__ starts, T *__restrict__ path, int Tx, int Ty,
                                                            T one) {
    extern __shared__ int st[];                                    // [Tx + 1]
    const int b = blockIdx.y, y = blockIdx.x * 256 + threadIdx.x;
    const int *sb = starts + (size_t)b * (Tx + 1);
    for (int x = threadIdx.x; x <= Tx; x += 256) st[x] = sb[x];
    __syncthreads();
    if (y >= Ty || y >= st[Tx] || y < st[0]) return;               // past the utterance's last frame: no token
    int lo = 0, hi = Tx;                                           // invariant: st[lo] <= y < st[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (st[mid] <= y) lo = mid;
        else hi = mid;
    }
    path[((size_t)b * Tx + lo) * Ty + y] = one;
}

template <typename T>
static int launch_scatter(const int *starts, void *path, int B, int Tx, int Ty, T one, hipStream_t s) {
    const size_t lds = (size_t)(Tx + 1) * sizeof(int);
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(scatter_path_kernel<T>), lds));
    hipLaunchKernelGGL(scatter_path_kernel<T>, dim3((Ty + 255) / 256, B), dim3(256), lds, s, starts, static_cast<T *>(path),
                       Tx, Ty, one);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

static int expand_impl(const int *starts, void *path, int path_dtype, int B, int Tx, int Ty, bool stream_path, hipStream_t s) {
    if (B > 65535 || (Tx + 7) / 8 > 65535) return fail(ALIGNER_EDOM, "grid too large");
    switch (path_dtype) {
        case ALIGNER_DT_F32: return launch_expand<float>(starts, path, B, Tx, Ty, 1.0f, stream_path, s);
        case ALIGNER_DT_F64: return launch_expand<double>(starts, path, B, Tx, Ty, 1.0, stream_path, s);
        case ALIGNER_DT_I32: return launch_expand<int32_t>(starts, path, B, Tx, Ty, 1, stream_path, s);
        case ALIGNER_DT_I64: return launch_expand<int64_t>(starts, path, B, Tx, Ty, 1, stream_path, s);
        case ALIGNER_DT_U8:  return launch_expand<uint8_t>(starts, path, B, Tx, Ty, 1, stream_path, s);
        case ALIGNER_DT_F16: return launch_expand<uint16_t>(starts, path, B, Tx, Ty, 0x3C00, stream_path, s);
        case ALIGNER_DT_BF16: return launch_expand<uint16_t>(starts, path, B, Tx, Ty, 0x3F80, stream_path, s);
        default: return fail(ALIGNER_EINVAL, "path dtype %d not supported", path_dtype);
    }
}

}  // namespace aligner

using namespace aligner;

extern "C" {

size_t aligner_maxpath_workspace_bytes(int B, int Tx, int Ty) {
    if (B < 0 || Tx < 1 || Ty < 1) return 0;
    return ws_layout(B, Tx, Ty).total;
}

int aligner_lengths_from_mask(const void *mask, int mask_dtype, int B, int Tx, int Ty,
                              int32_t *t_xs, int32_t *t_ys, void *stream) {
    if (!mask || !t_xs || !t_ys) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (B == 0) return ALIGNER_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (mask_dtype) {
        case ALIGNER_DT_F32:
            hipLaunchKernelGGL(lengths_kernel<float>, dim3(B), dim3(256), 0, s,
                               static_cast<const float *>(mask), Tx, Ty, t_xs, t_ys);
            break;
        case ALIGNER_DT_U8:
            hipLaunchKernelGGL(lengths_kernel<unsigned char>, dim3(B), dim3(256), 0, s,
                               static_cast<const unsigned char *>(mask), Tx, Ty, t_xs, t_ys);
            break;
        case ALIGNER_DT_I32:
            hipLaunchKernelGGL(lengths_kernel<int>, dim3(B), dim3(256), 0, s,
                               static_cast<const int *>(mask), Tx, Ty, t_xs, t_ys);
            break;
        case ALIGNER_DT_BF16:
            hipLaunchKernelGGL((lengths_kernel<unsigned short, VT_BF16>), dim3(B), dim3(256), 0, s,
                               static_cast<const unsigned short *>(mask), Tx, Ty, t_xs, t_ys);
            break;
        case ALIGNER_DT_F16:
            hipLaunchKernelGGL((lengths_kernel<unsigned short, VT_F16>), dim3(B), dim3(256), 0, s,
                               static_cast<const unsigned short *>(mask), Tx, Ty, t_xs, t_ys);
            break;
        default:
            return fail(ALIGNER_EINVAL, "mask dtype %d not supported (use F32, BF16, F16, U8 or I32)", mask_dtype);
    }
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

int aligner_maxpath_forward(const void *value, int value_dtype, const void *mask, int mask_dtype,
                            const int32_t *t_xs, const int32_t *t_ys, int32_t *tok_out,
                            int32_t *dur_out, void *ws, size_t ws_bytes, int B, int Tx, int Ty,
                            float max_neg_val, int flags, void *stream) {
    return forward_impl(value, value_dtype, mask, mask_dtype, t_xs, t_ys, tok_out, dur_out, ws, ws_bytes, B, Tx, Ty,
                        max_neg_val, flags, static_cast<hipStream_t>(stream));
}

int aligner_maxpath_forward_f32(const float *value, const void *mask, int mask_dtype,
                                const int32_t *t_xs, const int32_t *t_ys, int32_t *tok_out,
                                int32_t *dur_out, void *ws, size_t ws_bytes, int B, int Tx, int Ty,
                                float max_neg_val, int flags, void *stream) {
    return forward_impl(value, ALIGNER_DT_F32, mask, mask_dtype, t_xs, t_ys, tok_out, dur_out, ws, ws_bytes, B, Tx, Ty,
                        max_neg_val, flags, static_cast<hipStream_t>(stream));
}

int aligner_maxpath_expand_ex(const void *ws, void *path, int path_dtype, int B, int Tx, int Ty, int flags,
                              void *stream) {
    if (!ws || !path) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (B == 0) return ALIGNER_OK;
    const WsLayout L = ws_layout(B, Tx, Ty);
    const int *starts = reinterpret_cast<const int *>(static_cast<const unsigned char *>(ws) + L.starts_off);
    return expand_impl(starts, path, path_dtype, B, Tx, Ty, (flags & ALIGNER_F_STREAM_PATH) != 0,
                       static_cast<hipStream_t>(stream));
}

int aligner_maxpath_zero_path(void *path, int path_dtype, int B, int Tx, int Ty, int flags, void *stream) {
    if (!path) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    const int es = dtype_size(path_dtype);
    if (es == 0) return fail(ALIGNER_EINVAL, "path dtype %d not supported", path_dtype);
    if (B == 0) return ALIGNER_OK;
    unsigned char *pb = static_cast<unsigned char *>(path);
    const size_t bytes = (size_t)B * Tx * Ty * es;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // 16-byte stores over the aligned middle, byte stores for what is left at either end (< 16 bytes each)
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(pb) & 15)) & 15;
    const size_t h = head < bytes ? head : bytes;
    const size_t n16 = (bytes - h) / 16, tail = (bytes - h) - n16 * 16;
    if (h) {
        hipLaunchKernelGGL(zero_path_kernel<false>, dim3(1), dim3(256), 0, s, reinterpret_cast<uint4 *>(pb), (size_t)0, pb, (int)h);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    size_t blocks = (n16 + 255) / 256;
    const size_t cap = (size_t)device_cu_count() * 16;
    blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
    if (flags & ALIGNER_F_STREAM_PATH)
        hipLaunchKernelGGL(zero_path_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint4 *>(pb + h), n16,
                           pb + h + n16 * 16, (int)tail);
    else
        hipLaunchKernelGGL(zero_path_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint4 *>(pb + h), n16,
                           pb + h + n16 * 16, (int)tail);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

int aligner_maxpath_scatter_path(const void *ws, void *path, int path_dtype, int B, int Tx, int Ty, void *stream) {
    if (!ws || !path) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (B == 0) return ALIGNER_OK;
    if (B > 65535) return fail(ALIGNER_EDOM, "grid too large");
    if ((size_t)(Tx + 1) * sizeof(int) > (size_t)device_lds_limit()) return fail(ALIGNER_EDOM, "Tx=%d too large", Tx);
    const WsLayout L = ws_layout(B, Tx, Ty);
    const int *starts = reinterpret_cast<const int *>(static_cast<const unsigned char *>(ws) + L.starts_off);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (path_dtype) {
        case ALIGNER_DT_F32: return launch_scatter<float>(starts, path, B, Tx, Ty, 1.0f, s);
        case ALIGNER_DT_F64: return launch_scatter<double>(starts, path, B, Tx, Ty, 1.0, s);
        case ALIGNER_DT_I32: return launch_scatter<int32_t>(starts, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_I64: return launch_scatter<int64_t>(starts, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_U8:  return launch_scatter<uint8_t>(starts, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_F16: return launch_scatter<uint16_t>(starts, path, B, Tx, Ty, 0x3C00, s);
        case ALIGNER_DT_BF16: return launch_scatter<uint16_t>(starts, path, B, Tx, Ty, 0x3F80, s);
        default: return fail(ALIGNER_EINVAL, "path dtype %d not supported", path_dtype);
    }
}

int aligner_maxpath_expand(const void *ws, void *path, int path_dtype, int B, int Tx, int Ty,
                           void *stream) {
    return aligner_maxpath_expand_ex(ws, path, path_dtype, B, Tx, Ty, 0, stream);
}

int aligner_maxpath(const void *value, int value_dtype, const void *mask, int mask_dtype, const int32_t *t_xs,
                    const int32_t *t_ys, void *path_out, int path_dtype, int32_t *tok_out,
                    int32_t *dur_out, void *ws, size_t ws_bytes, int B, int Tx, int Ty,
                    float max_neg_val, int flags, void *stream) {
    if (path_out && dtype_size(path_dtype) == 0)
        return fail(ALIGNER_EINVAL, "path dtype %d not supported", path_dtype);
    // ALIGNER_F_PATH_PREZEROED: the search kernel itself writes the ones into the caller's zeros -- no expand launch
    // and without it the pipelined kernel's own zero workgroups do the whole path where they can (path_done)
    const bool prez = path_out && (flags & ALIGNER_F_PATH_PREZEROED);
    bool path_done = false;
    int rc = forward_impl(value, value_dtype, mask, mask_dtype, t_xs, t_ys, tok_out, dur_out, ws, ws_bytes, B, Tx, Ty,
                          max_neg_val, flags, static_cast<hipStream_t>(stream), path_out, path_dtype, prez, &path_done);
    if (rc || !path_out || B == 0 || path_done) return rc;
    return aligner_maxpath_expand_ex(ws, path_out, path_dtype, B, Tx, Ty, flags, stream);
}

int aligner_maxpath_ld(const void *value, int value_dtype, int ld_value, const int32_t *t_xs, const int32_t *t_ys,
                       void *path_out, int path_dtype, int32_t *tok_out, int32_t *dur_out, void *ws, size_t ws_bytes, int B,
                       int Tx, int Ty, float max_neg_val, int flags, void *stream) {
    if (path_out && dtype_size(path_dtype) == 0)
        return fail(ALIGNER_EINVAL, "path dtype %d not supported", path_dtype);
    if (flags & ALIGNER_F_STRICT_MASK) return fail(ALIGNER_EINVAL, "aligner_maxpath_ld takes no mask");
    const bool prez = path_out && (flags & ALIGNER_F_PATH_PREZEROED);
    bool path_done = false;
    int rc = forward_impl(value, value_dtype, nullptr, 0, t_xs, t_ys, tok_out, dur_out, ws, ws_bytes, B, Tx, Ty, max_neg_val,
                          flags, static_cast<hipStream_t>(stream), path_out, path_dtype, prez, &path_done, ld_value);
    if (rc || !path_out || B == 0 || path_done) return rc;
    return aligner_maxpath_expand_ex(ws, path_out, path_dtype, B, Tx, Ty, flags, stream);
}

int aligner_maxpath_f32(const float *value, const void *mask, int mask_dtype, const int32_t *t_xs,
                        const int32_t *t_ys, void *path_out, int path_dtype, int32_t *tok_out,
                        int32_t *dur_out, void *ws, size_t ws_bytes, int B, int Tx, int Ty,
                        float max_neg_val, int flags, void *stream) {
    return aligner_maxpath(value, ALIGNER_DT_F32, mask, mask_dtype, t_xs, t_ys, path_out, path_dtype, tok_out, dur_out,
                           ws, ws_bytes, B, Tx, Ty, max_neg_val, flags, stream);
}

int aligner_maxpath_read_status(void *ws, int32_t *status_host, void *stream) {
    if (!ws || !status_host) return fail(ALIGNER_EINVAL, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ALIGNER_HIP_CHECK(hipMemcpyAsync(status_host, ws, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ALIGNER_HIP_CHECK(hipMemsetAsync(ws, 0, sizeof(int32_t), s));     // status is sticky until read
    ALIGNER_HIP_CHECK(hipStreamSynchronize(s));
    return ALIGNER_OK;
}

int aligner_maxpath_host_f32(int32_t *paths, float *values, const int32_t *t_xs,
                             const int32_t *t_ys, int B, int Tx, int Ty, float max_neg_val, int flags) {
    if (!paths || !values || !t_xs || !t_ys) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (B == 0) return ALIGNER_OK;
    if (!(flags & ALIGNER_F_COMPAT_TXGTTY))
        for (int b = 0; b < B; ++b)
            if (t_xs[b] < 1 || t_xs[b] > t_ys[b] || t_xs[b] > Tx || t_ys[b] > Ty)
                return fail(ALIGNER_EDOM, "utterance %d: t_x=%d t_y=%d outside 1<=t_x<=t_y (Tx=%d,Ty=%d)",
                            b, t_xs[b], t_ys[b], Tx, Ty);
    const size_t n = (size_t)B * Tx * Ty;
    const size_t wsb = aligner_maxpath_workspace_bytes(B, Tx, Ty);
    void *d_val = nullptr, *d_path = nullptr, *d_len = nullptr, *d_ws = nullptr;
    int rc = ALIGNER_OK;
    hipError_t e;
#define HOST_TRY(expr)                                                                          \
    if (rc == ALIGNER_OK && (e = (expr)) != hipSuccess)                                         \
        rc = fail(ALIGNER_EHIP, "%s failed: %s", #expr, hipGetErrorString(e));
    HOST_TRY(hipMalloc(&d_val, n * 4));
    HOST_TRY(hipMalloc(&d_path, n * 4));
    HOST_TRY(hipMalloc(&d_len, (size_t)2 * B * 4));
    HOST_TRY(hipMalloc(&d_ws, wsb));
    HOST_TRY(hipMemset(d_ws, 0, WS_HDR_BYTES));
    HOST_TRY(hipMemcpy(d_val, values, n * 4, hipMemcpyHostToDevice));
    HOST_TRY(hipMemcpy(d_len, t_xs, (size_t)B * 4, hipMemcpyHostToDevice));
    HOST_TRY(hipMemcpy(static_cast<int *>(d_len) + B, t_ys, (size_t)B * 4, hipMemcpyHostToDevice));
    if (rc == ALIGNER_OK)
        rc = aligner_maxpath_f32(static_cast<float *>(d_val), nullptr, 0, static_cast<int *>(d_len),
                                 static_cast<int *>(d_len) + B, d_path, ALIGNER_DT_I32, nullptr, nullptr,
                                 d_ws, wsb, B, Tx, Ty, max_neg_val, flags, nullptr);
    HOST_TRY(hipMemcpy(paths, d_path, n * 4, hipMemcpyDeviceToHost));
    if (flags & ALIGNER_F_WRITE_Q) HOST_TRY(hipMemcpy(values, d_val, n * 4, hipMemcpyDeviceToHost));
    int st = 0;
    HOST_TRY(hipMemcpy(&st, d_ws, sizeof(int), hipMemcpyDeviceToHost));
    if (rc == ALIGNER_OK && (st & ALIGNER_ST_INTERNAL))
        rc = fail(ALIGNER_EHIP, "internal consistency check failed on the device (status word %d): results are not valid", st);
#undef HOST_TRY
    if (d_val) (void)hipFree(d_val);
    if (d_path) (void)hipFree(d_path);
    if (d_len) (void)hipFree(d_len);
    if (d_ws) (void)hipFree(d_ws);
    return rc;
}

}  // extern "C"
