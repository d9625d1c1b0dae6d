// Monotonic alignment search on MI355X (gfx950): forward sweep + backtrack.
//
// Replaces, bit-for-bit on the integer path, the reference's
//   maximum_path_each  (monotonic_align/core.pyx:7-35)
//   maximum_path_c     (monotonic_align/core.pyx:38-45)
// and the marshalling of monotonic_align/__init__.py:11-21, with the score
// tensor resident in HBM.  See DESIGN.md for the derivation; the short form:
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

#include "aligner_amd.h"
#include "common.h"
#include "maxpath_sweep_asm.inc"

namespace aligner {

constexpr int TC       = 32;  // frames per tile == decision bits per word
constexpr int TILE_LD  = 36;  // dwords per LDS tile row: 32 + 4 pad -> 16B-slot stride 9 (odd)
constexpr int RING_T   = 4;   // boundary ring depth in tiles
constexpr int RING_LD  = 64;  // floats per ring slot: 32 used, padded so that all 64 lanes can store unmasked
constexpr int RPW      = 63;  // text rows per compute wave (lane 0 is the ghost lane)
constexpr int WS_HDR_BYTES = 256;

enum Mode { MODE_NORMAL = 0, MODE_EMPTY = 1, MODE_COMPAT = 2 };

// LDS-qualified float: keeps per-lane selected addresses as ds_* instructions
// (a generic pointer would turn them into flat_* accesses).
typedef __attribute__((address_space(3))) float lds_float;
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;    // 16-byte aligned: ds_read_b128

struct MaxpathParams {
    const float *value;
    const void  *mask;      // strict-mask operand (nullable)
    const int   *t_xs;
    const int   *t_ys;
    int         *starts;    // [B,Tx+1] first frame of every token (workspace)
    int         *tok;       // [B,Ty] nullable
    int         *dur;       // [B,Tx] nullable
    unsigned    *bits;      // [B,NT,ROWS] decision words in global memory
    int         *status;
    int B, Tx, Ty, NT, ROWS;
    int WT;                 // tiles per backtrack window when the words live in global memory
    int bits_in_lds;        // pipelined kernel: decision words stay in LDS
    int lds_bits_off;       // byte offset of that LDS region
    int force_exact;        // skip the v_max sweep (non-finite max_neg_val)
    float neg;
    int flags;
    unsigned long long *stamps;   // debug: [B][16 waves][16] shader-clock stamps (nullable)
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

// --------------------------------------------------------------------------
// small device helpers
// --------------------------------------------------------------------------
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

// Token starts -> the caller's outputs.  starts[x] = first frame of token x for
// x < t_x and t_y for t_x <= x <= Tx, so durations are plain differences and a
// frame's token is found by bisection.
__device__ __forceinline__ void store_outputs(const MaxpathParams &p, int b, int tx, int ty, const int *startsL) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    for (int x = tid; x <= p.Tx; x += nthreads) p.starts[(size_t)b * (p.Tx + 1) + x] = startsL[x];
    if (p.dur)
        for (int x = tid; x < p.Tx; x += nthreads) p.dur[(size_t)b * p.Tx + x] = startsL[x + 1] - startsL[x];
    if (p.tok)
        for (int y = tid; y < p.Ty; y += nthreads) {
            int t = -1;
            if (y < ty) {
                int lo = 0, hi = tx - 1;              // last x with starts[x] <= y
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (startsL[mid] <= y) lo = mid; else hi = mid - 1;
                }
                t = lo;
            }
            p.tok[(size_t)b * p.Ty + y] = t;
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
template <int MASKMODE>
__device__ __forceinline__ void write_degenerate(const MaxpathParams &p, int b, int mode, int tx, int ty, int *startsL) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    for (int x = tid; x <= p.Tx; x += nthreads) startsL[x] = (mode == MODE_COMPAT && x >= tx) ? ty : 0;
    __syncthreads();
    if (mode == MODE_COMPAT && tid == 0) {
        const float *val = p.value + (size_t)b * p.Tx * p.Ty;
        const float *msk = (MASKMODE == 1) ? reinterpret_cast<const float *>(p.mask) + (size_t)b * p.Tx * p.Ty : nullptr;
        int index = tx - 1;
        for (int y = ty - 1; y >= 1; --y) {
            if (index == 0) break;
            bool up = (index == y);
            if (!up) {
                float a = val[(size_t)index * p.Ty + (y - 1)], c = val[(size_t)(index - 1) * p.Ty + (y - 1)];
                if (MASKMODE == 1) {
                    a *= msk[(size_t)index * p.Ty + (y - 1)];
                    c *= msk[(size_t)(index - 1) * p.Ty + (y - 1)];
                }
                up = a < c;
            }
            if (up) { startsL[index] = y; index -= 1; }        // row `index` owns frames from y on
        }
        // rows 0..index: `index` owns frames [0, start of index+1), the rows above it nothing
    }
    __syncthreads();
    store_outputs(p, b, tx, mode == MODE_COMPAT ? ty : 0, startsL);      // no frame has a token in the empty modes
}

// --------------------------------------------------------------------------
// Backtrack over decision words (shared by both forward kernels).
//
// Word (tile t, row x) holds the decisions of text row x for frames 32t..32t+31,
// frame 32t+c at bit 31-c.  Wave 0 walks rows from t_x-1 down: row x ending at
// frame e starts at the highest set bit <= e of its own bit string (the move at
// that frame is the reference's `index -= 1`, core.pyx:34-35).  Lane j holds the
// row's word of tile jb+j (one coalesced LDS read per row, prefetched two rows
// ahead); the walk state (x, e) lives in SGPRs.  Common case -- the start lies in
// the word that contains e -- is one v_readlane plus scalar bit ops; otherwise a
// ballot over the row's earlier words.
//
// LDS overlay once the forward sweep is done: startsL at offset 0, then (words in
// global memory only) a window of WT tiles x (ROWS+1) words.
// --------------------------------------------------------------------------
// One row of the single-window backtrack with everything static: wr[K] holds row (64c+K)'s
// words (lane = tile).  The common case (the row starts in the word that contains e) is ten
// scalar/vector issues in one asm statement, no LDS, no loop; `ok == 0` sends the rare long
// token to the ballot search over the row's earlier words.
template <int K, bool CHECK>
__device__ __forceinline__ void walk_row_static(const unsigned (&wr)[64], int xbase, int xtop, int &e, int &startv,
                                                int lane, int *status) {
    const int x = xbase + K;
    if (!CHECK || (x <= xtop && x >= 1)) {                       // uniform
        const unsigned w = wr[K];
        int s, ok, je, t, word;
        asm volatile(
            "s_lshr_b32 %[je], %[e], 5\n\t"                       // tile of frame e
            "s_not_b32 %[t], %[e]\n\t"                            // low 5 bits: 31 - (e & 31)
            "s_or_b32 %[s], %[e], 31\n\t"                         // last frame of that tile
            "v_readlane_b32 %[word], %[w], %[je]\n\t"             // the row's word of that tile
            "s_lshl_b32 %[t], -1, %[t]\n\t"                       // frames <= e  <->  bits >= 31 - (e & 31)
            "s_and_b32 %[word], %[word], %[t]\n\t"                // SCC = some decision bit at or before e
            "s_cselect_b32 %[ok], 1, 0\n\t"
            "s_ff1_i32_b32 %[t], %[word]\n\t"                     // lowest set bit = latest such frame
            "s_sub_i32 %[s], %[s], %[t]\n\t"
            : [je] "=&s"(je), [t] "=&s"(t), [word] "=&s"(word), [s] "=&s"(s), [ok] "=&s"(ok)
            : [e] "s"(__builtin_amdgcn_readfirstlane(e)), [w] "v"(w)
            : "scc");
        if (__builtin_amdgcn_readfirstlane(ok) == 0) {
            const unsigned long long bal = __ballot(lane < __builtin_amdgcn_readfirstlane(je) && w != 0u);
            if (bal != 0ull) {
                const int js = 63 - __builtin_clzll(bal);
                const unsigned wsel = (unsigned)__builtin_amdgcn_readlane((int)w, js);
                s = ((js << 5) | (TC - 1)) - __builtin_ctz(wsel);
            } else {                                             // cannot happen: the diagonal bit is always set
                s = x;
                if (lane == 0) atomicOr(status, ALIGNER_ST_INTERNAL);
            }
        }
        s = __builtin_amdgcn_readfirstlane(s);
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(startv) : "s"(s), "n"(K));   // startv[lane K] = s
        e = __builtin_amdgcn_readfirstlane(s - 1);
    }
}

template <int K, int KMIN, bool CHECK>
struct WalkChunk {
    static __device__ __forceinline__ void run(const unsigned (&wr)[64], int xbase, int xtop, int &e, int &startv,
                                               int lane, int *status) {
        walk_row_static<K, CHECK>(wr, xbase, xtop, e, startv, lane, status);
        WalkChunk<K - 1, KMIN, CHECK>::run(wr, xbase, xtop, e, startv, lane, status);
    }
};
template <int KMIN, bool CHECK>
struct WalkChunk<KMIN, KMIN, CHECK> {
    static __device__ __forceinline__ void run(const unsigned (&wr)[64], int xbase, int xtop, int &e, int &startv,
                                               int lane, int *status) {
        walk_row_static<KMIN, CHECK>(wr, xbase, xtop, e, startv, lane, status);
    }
};

// The same row step for a WINDOW of tiles [jb, jb + ntw) (decision words in the workspace, more than 64
// tiles): lane = window-relative tile.  Rows are still static (wr[K] = row xbase+K), the window is not:
// a row whose frame e lies below the window, or whose start is not in it, sets `stop` and the walk resumes
// at that row in the next (earlier) window; a frame e above the window (a token that spans the boundary)
// means every word of the window qualifies.
template <int K>
__device__ __forceinline__ void walk_row_window(const unsigned (&wr)[64], int xbase, int &x, int &e, int &startv,
                                                int lane, int jb, int ntw, int &stop) {
    const int xr = xbase + K;
    if (stop == 0 && xr <= x && xr >= 1) {                       // uniform; rows above x are done already
        const unsigned w = wr[K];
        int s, ok, jr, jrc, valid, t, word;
        asm volatile(
            "s_lshr_b32 %[jr], %[e], 5\n\t"
            "s_sub_i32 %[jr], %[jr], %[jb]\n\t"                   // window-relative tile of frame e
            "s_not_b32 %[t], %[e]\n\t"
            "s_or_b32 %[s], %[e], 31\n\t"                         // last frame of e's tile
            "s_cmp_lt_u32 %[jr], %[ntw]\n\t"                      // inside the window (unsigned: also jr >= 0)
            "s_cselect_b32 %[jrc], %[jr], 0\n\t"
            "s_cselect_b32 %[valid], -1, 0\n\t"
            "s_lshl_b32 %[t], -1, %[t]\n\t"                       // frames <= e  <->  bits >= 31 - (e & 31)
            "v_readlane_b32 %[word], %[w], %[jrc]\n\t"
            "s_and_b32 %[t], %[t], %[valid]\n\t"
            "s_and_b32 %[word], %[word], %[t]\n\t"                // SCC = a decision bit at or before e in e's word
            "s_cselect_b32 %[ok], 1, 0\n\t"
            "s_ff1_i32_b32 %[t], %[word]\n\t"
            "s_sub_i32 %[s], %[s], %[t]\n\t"
            : [jr] "=&s"(jr), [t] "=&s"(t), [s] "=&s"(s), [jrc] "=&s"(jrc), [valid] "=&s"(valid),
              [word] "=&s"(word), [ok] "=&s"(ok)
            : [e] "s"(__builtin_amdgcn_readfirstlane(e)), [jb] "s"(__builtin_amdgcn_readfirstlane(jb)),
              [ntw] "s"(__builtin_amdgcn_readfirstlane(ntw)), [w] "v"(w)
            : "scc");
        if (__builtin_amdgcn_readfirstlane(ok) == 0) {
            if (jr < 0) {
                stop = 1;                                        // row xr continues in an earlier window
            } else {
                if (jr > ntw) jr = ntw;                          // e lies past this window: every word qualifies
                const unsigned long long bal = __ballot(lane < jr && w != 0u);
                if (bal == 0ull) {
                    stop = 1;                                    // its start lies in an earlier window
                } else {
                    const int js = 63 - __builtin_clzll(bal);
                    const unsigned wsel = (unsigned)__builtin_amdgcn_readlane((int)w, js);
                    s = (((jb + js) << 5) | (TC - 1)) - __builtin_ctz(wsel);
                }
            }
        }
        if (stop == 0) {
            s = __builtin_amdgcn_readfirstlane(s);
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(startv) : "s"(s), "n"(K));   // row xr owns [s, e]
            e = __builtin_amdgcn_readfirstlane(s - 1);
            x = xr - 1;
        }
    }
}

template <int K, int KMIN>
struct WalkChunkW {
    static __device__ __forceinline__ void run(const unsigned (&wr)[64], int xbase, int &x, int &e, int &startv,
                                               int lane, int jb, int ntw, int &stop) {
        walk_row_window<K>(wr, xbase, x, e, startv, lane, jb, ntw, stop);
        WalkChunkW<K - 1, KMIN>::run(wr, xbase, x, e, startv, lane, jb, ntw, stop);
    }
};
template <int KMIN>
struct WalkChunkW<KMIN, KMIN> {
    static __device__ __forceinline__ void run(const unsigned (&wr)[64], int xbase, int &x, int &e, int &startv,
                                               int lane, int jb, int ntw, int &stop) {
        walk_row_window<KMIN>(wr, xbase, x, e, startv, lane, jb, ntw, stop);
    }
};

// Decision words of `ntw` tiles, rows [0, rows_used), from the workspace into the LDS window: 16-byte loads,
// eight in flight per thread (one element at a time this copy was a memory round trip per 4 bytes and a
// quarter of the long-form kernel's time).  rows_used and ROWS are multiples of 64; the window's row
// stride RP is odd (bank spread for the walk), hence the four 4-byte LDS stores.
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
                unsigned *d = win + j * RP + 4 * r4;
                d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
            }
        }
    }
}

__device__ __forceinline__ void backtrack_and_store(const MaxpathParams &p, int b, int tx, int ty, unsigned char *smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthreads = blockDim.x;
    const int RP = p.ROWS + 1;
    int *startsL = reinterpret_cast<int *>(smem);
    const int starts_words = ((p.Tx + 1 + 63) / 64) * 64 + 64;
    unsigned *win = reinterpret_cast<unsigned *>(smem) + starts_words;
    const bool in_lds = p.bits_in_lds != 0;
    if (in_lds) win = reinterpret_cast<unsigned *>(smem + p.lds_bits_off);

    const int ntb = (ty + TC - 1) / TC;            // tiles this utterance uses
    const int rows_used = ((tx + 63) / 64) * 64;
    const unsigned *gbits = p.bits + (size_t)b * p.NT * p.ROWS;
    const int WT = in_lds ? ntb : p.WT;

    // Walk state lives in SGPRs: every update goes through readfirstlane so the loop is
    // scalar control flow (a VGPR-carried loop costs an exec-mask dance per branch).
    const bool walker = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    int x = __builtin_amdgcn_readfirstlane(tx - 1);   // core.pyx:15
    int e = __builtin_amdgcn_readfirstlane(ty - 1);
    int startv = 0;   // starts of rows 64c..64c+63 of the chunk being walked, one per lane

    const bool single = ntb <= WT;                     // every tile of the utterance in one window
    if (single) {
        if (!in_lds) {
            __syncthreads();
            load_window(win, RP, gbits, p.ROWS, rows_used, ntb, tid, nthreads);
            __syncthreads();
        }
        ALIGNER_STAMP(2);
        if (walker) {
            const unsigned *wrow = win + (lane < ntb ? lane : 0) * RP;
            for (int c = x >> 6; c >= 0; --c) {
                unsigned wr[64];
#pragma unroll
                for (int k = 0; k < 64; ++k) wr[k] = wrow[64 * c + k];      // 64 rows x (lane = tile)
                // the top chunk is ragged (rows above t_x-1 do not exist); chunk 0 stops at row 1
                if (64 * c + 63 > x)  WalkChunk<63, 0, true>::run(wr, 64 * c, x, e, startv, lane, p.status);
                else if (c == 0)      WalkChunk<63, 1, false>::run(wr, 0, x, e, startv, lane, p.status);
                else                  WalkChunk<63, 0, false>::run(wr, 64 * c, x, e, startv, lane, p.status);
                if (c == 0) startv = (lane == 0) ? 0 : startv;              // row 0 starts at frame 0
                startsL[64 * c + lane] = startv;
            }
            x = 0;
        }
    }
    for (int jhi = single ? 0 : ntb; jhi > 0; jhi -= WT) {
        const int jb = (jhi - WT > 0) ? (jhi - WT) : 0;
        const int ntw = jhi - jb;
        if (!in_lds) {
            __syncthreads();                           // previous window fully consumed
            load_window(win, RP, gbits + (size_t)jb * p.ROWS, p.ROWS, rows_used, ntw, tid, nthreads);
            __syncthreads();
        }
        ALIGNER_STAMP(2);
        if (walker && x >= 1) {
            // lanes past the window read tile 0's words; they can never be selected (lane < jr <= ntw)
            const unsigned *wrow = win + (lane < ntw ? lane : 0) * RP;
            int stop = 0;
            for (int c = x >> 6; c >= 0 && stop == 0; --c) {
                unsigned wr[64];
#pragma unroll
                for (int k = 0; k < 64; ++k) wr[k] = wrow[64 * c + k];      // 64 rows x (lane = window tile)
                WalkChunkW<63, 0>::run(wr, 64 * c, x, e, startv, lane, jb, ntw, stop);
                // chunk finished (a stopped one is finished from the next window; startv carries over)
                if (stop == 0) startsL[64 * c + lane] = startv;
            }
        }
    }
    ALIGNER_STAMP(3);
    if (walker && !single) {
        // x == 0 here for every valid input (forced diagonal, core.pyx:34): row 0 starts at frame 0.
        if (x != 0 && lane == 0) atomicOr(p.status, ALIGNER_ST_INTERNAL);
        startv = (lane == 0) ? 0 : startv;
        startsL[lane] = startv;
    }
    __syncthreads();
    for (int r = tx + tid; r <= p.Tx; r += nthreads) startsL[r] = ty;
    __syncthreads();
    store_outputs(p, b, tx, ty, startsL);
}

// --------------------------------------------------------------------------
// Generic forward kernel: any Tx <= 256*R, one barrier per frame.  Slow but
// shape-agnostic; also the independent cross-check of the pipelined kernel.
// --------------------------------------------------------------------------
template <int R, int MASKMODE>
__global__ __launch_bounds__(256) void maxpath_generic_kernel(MaxpathParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    int tx, ty;
    const int mode = classify_lengths(p, b, tx, ty);
    if (mode != MODE_NORMAL) { write_degenerate<MASKMODE>(p, b, mode, tx, ty, reinterpret_cast<int *>(smem)); return; }

    float *qcol = reinterpret_cast<float *>(smem);   // [2][256*R + 1], index x+1
    const int QLD = 256 * R + 1;
    const float *val = p.value + (size_t)b * p.Tx * p.Ty;
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
                float v = val[(size_t)x * p.Ty + y];
                if (MASKMODE == 1)
                    v = v * reinterpret_cast<const float *>(p.mask)[((size_t)b * p.Tx + x) * p.Ty + y];
                q[r] = (adv ? up : cur) + v;                       // core.pyx:30
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
    backtrack_and_store(p, b, tx, ty, smem);
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
__device__ __forceinline__ __amdgpu_buffer_rsrc_t utterance_rsrc(const float *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, bytes, 0x00020000);
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
template <int MASKMODE>
__device__ __forceinline__ void exact_fallback_sweep(const MaxpathParams &p, int b, int tx, int ty, unsigned char *smem,
                                                   unsigned *bitsL, int RPB) {
    const int tid = threadIdx.x, nthreads = blockDim.x;
    float *qcol = reinterpret_cast<float *>(smem);            // [2][nthreads + 1], index x+1
    const int QLD = nthreads + 1;
    const float *val = p.value + (size_t)b * p.Tx * p.Ty;
    const int x = tid;                                        // tx <= 63*NW < nthreads
    const int slot = x;
    unsigned *gw = p.bits + (size_t)b * p.NT * p.ROWS + slot;
    float q = p.neg;
    unsigned bits = 0u;
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
            float v = val[(size_t)x * p.Ty + y];
            if (MASKMODE == 1) v *= reinterpret_cast<const float *>(p.mask)[((size_t)b * p.Tx + x) * p.Ty + y];
            q = (adv ? up : cur) + v;                           // core.pyx:30
            bits = (bits << 1) | ((adv || x == y) ? 1u : 0u);   // core.pyx:34
            dst[x + 1] = q;
        }
        if (tid == 0) dst[0] = p.neg;                           // core.pyx:27
        if ((y & (TC - 1)) == TC - 1 || y == ty - 1) {
            const int t = y / TC;
            const unsigned wv = bits << ((TC - 1) - (y & (TC - 1)));
            if (x < tx) {
                if (p.bits_in_lds) bitsL[t * RPB + slot] = wv;
                else               gw[(size_t)t * p.ROWS] = wv;
            }
            bits = 0u;
        }
        __syncthreads();
    }
}

template <int NW, int DEPTH, bool VEC, int MASKMODE>
__global__ __launch_bounds__(NW * 128, NW <= 4 ? 2 : 1) void maxpath_pipelined_kernel(MaxpathParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    int tx, ty;
    const int mode = classify_lengths(p, b, tx, ty);
    if (mode != MODE_NORMAL) { write_degenerate<MASKMODE>(p, b, mode, tx, ty, reinterpret_cast<int *>(smem)); return; }
    ALIGNER_STAMP(0);
    ALIGNER_STAMP(6);

    float *tiles = reinterpret_cast<float *>(smem);              // [NW][2][64][TILE_LD]
    float *ring  = tiles + NW * 2 * 64 * TILE_LD;                // [NW][RING_T][RING_LD]: row 63w-1 for wave w
    int   *flagp = reinterpret_cast<int *>(ring + NW * RING_T * RING_LD);   // [4] non-finite score seen
    unsigned *bitsL = reinterpret_cast<unsigned *>(smem + p.lds_bits_off);   // [NT][ROWS+1] when in LDS
    const int RPB = p.ROWS + 1;
    const int ntb = (ty + TC - 1) / TC;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    const int nw_act = (tx + RPW - 1) / RPW;                    // waves that own at least one real row

    // wave 0's ghost lane replays "row -1": max_neg_val for every frame (core.pyx:27)
    for (int i = tid; i < RING_T * RING_LD; i += NW * 128) ring[i] = p.neg;
    if (tid == 0) flagp[0] = 0;
    __syncthreads();

    if (!p.force_exact) {
        // Tiles a wave needs form one contiguous range [t_lo, t_hi] (band of rows 63w..63w+62,
        // core.pyx:18); every wave still takes part in all ntb + NW phase barriers.
        const int w = (wave < NW) ? wave : wave - NW;
        const bool active = w < nw_act;
        const int t_lo = active ? (RPW * w) / TC : 0;
        int t_hi = active ? (ty - tx + RPW * w + RPW - 1) / TC : -1;
        if (t_hi > ntb - 1) t_hi = ntb - 1;
        const int ntiles = t_hi - t_lo + 1;
        if (wave < NW) {
            // ------------------------------ compute wave ------------------------------
            const int row = RPW * w + lane - 1;                 // lane 0: ghost (row 63w-1)
            const bool publish = w + 1 < nw_act;
            float q = (w == 0 && lane == 0) ? 0.0f : p.neg;     // Q[-1,-1] = 0 (core.pyx:24-25)
            float m = 0.0f;                                     // lane 0 stays 0 through the asm sweep
            unsigned bits = 0u;
            int coll = 0;
            const float *mytiles = tiles + w * 2 * 64 * TILE_LD + lane * TILE_LD;
            const float *myring = ring + w * RING_T * RING_LD;
            float *outring = ring + (publish ? w + 1 : w) * RING_T * RING_LD;
            const int brow = (lane == 0) ? p.ROWS - 1 : row;            // decision word column (ghost lane: padding)
            unsigned *gbits = p.bits + (size_t)b * p.NT * p.ROWS + brow;

            for (int i = 0; i < t_lo + w + 1; ++i) __syncthreads();
            for (int t = t_lo; t <= t_hi; ++t) {
                // one ds_read_b128 per 4 frames; the padded row stride (9 x 16 B) makes the 16-lane
                // groups of a b128 read hit 16 different 16-byte slots: conflict-free
                const lds_f32x4 *src = (const lds_f32x4 *)((lane == 0) ? (myring + (t & (RING_T - 1)) * RING_LD)
                                                                         : (mytiles + (t & 1) * 64 * TILE_LD));
                float4 vv[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 r = src[g];
                    vv[g] = make_float4(r.x, r.y, r.z, r.w);
                }
                const int y0 = t * TC;
                const bool diag = (y0 <= RPW * w + RPW - 1) && (y0 + TC - 1 >= RPW * w);
                const int rrel = row - y0;
                if (diag) {
                    if (publish) sweep_tile_fast<true, true>(q, m, bits, coll, vv, rrel, p.neg);
                    else         sweep_tile_fast<false, true>(q, m, bits, coll, vv, rrel, p.neg);
                } else {
                    if (publish) sweep_tile_fast<true, false>(q, m, bits, coll, vv, rrel, p.neg);
                    else         sweep_tile_fast<false, false>(q, m, bits, coll, vv, rrel, p.neg);
                }
                // unmasked stores: lanes >= 32 hit the slot's padding, the ghost lane a padding column
                if (publish) outring[(t & (RING_T - 1)) * RING_LD + lane] = __builtin_bit_cast(float, coll);
                if (p.bits_in_lds) bitsL[t * RPB + brow] = bits;              // frame 32t+c <-> bit 31-c
                else               gbits[(size_t)t * p.ROWS] = bits;
                bits = 0u;
                __syncthreads();
            }
            for (int i = 0; i < ntb + NW - w - t_hi - 2; ++i) __syncthreads();
        } else {
            // ------------------------------ loader wave -------------------------------
            if (active) {
                const int rr = lane >> 3, cg = lane & 7;
                float *mytiles = tiles + w * 2 * 64 * TILE_LD + rr * TILE_LD + 4 * cg;
                const float *ub = p.value + ubase;
                const float *mb = (MASKMODE == 1) ? reinterpret_cast<const float *>(p.mask) + ubase : nullptr;
                // LDS row slot 8k+rr <-> text row 63w + 8k + rr - 1 (slot 0 is the ghost lane's: never read)
                unsigned rowoff[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // rows past the utterance's own text (padding: often -inf log-probs) are replaced
                    // by its last row: they are never read by the DP and must not trip the finiteness scan
                    int r = RPW * w + 8 * k + rr - 1;
                    r = r < 0 ? 0 : (r > tx - 1 ? tx - 1 : r);
                    rowoff[k] = (unsigned)r * (unsigned)p.Ty;
                }
                float nf = 0.f;                             // maximum |score| seen, NaN-propagating (scan4)
                float4 buf[DEPTH][8];
                // VEC: buffer loads, per-lane byte offsets fixed for the whole sweep + a scalar tile offset
                const unsigned ubytes = (unsigned)p.Tx * (unsigned)p.Ty * 4u;
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
                for (int i = 0; i < t_lo + w; ++i) __syncthreads();
                for (int i0 = 0; i0 < ntiles; i0 += DEPTH) {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        const int t = t_lo + i0 + d;
                        if (t <= t_hi) {
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
                        if (t <= t_hi) __syncthreads();
                    }
                }
                if (absbits(nf) >= 0x7F800000u) flagp[0] = 1;   // benign race: every writer stores 1
                for (int i = 0; i < ntb + NW - w - t_hi - 1; ++i) __syncthreads();
            } else {
                for (int i = 0; i < ntb + NW; ++i) __syncthreads();
            }
        }
        __syncthreads();
    }
    // A NaN or an infinity among the scores (or max_neg_val): v_max no longer equals the
    // reference's select, so redo this utterance with the exact barrier-per-frame sweep.
    if (p.force_exact || flagp[0] != 0) exact_fallback_sweep<MASKMODE>(p, b, tx, ty, smem, bitsL, RPB);
    // decision words: in LDS, or in global memory written by this CU
    __threadfence_block();
    ALIGNER_STAMP(1);
    backtrack_and_store(p, b, tx, ty, smem);
    ALIGNER_STAMP(5);
    ALIGNER_STAMP(7);
}

// --------------------------------------------------------------------------
// Halo-lane forward kernel (Tx <= 256, NT <= 64, everything LDS-resident): the fast path.
//
// Same systolic tile pipeline as above, but a wave owns only 32 text rows (lanes 32..63) and
// its 32 low lanes RECOMPUTE the 32 rows above it instead of receiving the boundary row through
// LDS every frame.  A recomputed row is wrong once the missing row above the halo has propagated
// down to it, i.e. halo lane h is wrong after frame h of a tile -- so lane 31 stays right for the
// whole 32-frame tile, and the halo is resynchronised from the upper wave's saved registers once
// per tile (one ds_write_b32 + one ds_read_b32 per wave and tile; nothing per frame).  The inner
// loop is 5 VALU issues per frame for every wave (ALIGNER_HALO16_*).
//
// The sweep also tracks, per cell, the row its best path occupied just before the tile's first
// frame (B, one v_cndmask_b32_dpp per frame).  At the end of a tile that is the tile's backtrack
// transition  T_t: row at frame 32t+31 -> row at frame 32t-1,  stored as one byte per row.  The
// backtrack then is: (1) a short serial walk through the last (partial) tile, (2) NT-2 dependent
// table look-ups giving the path's row at every tile boundary, (3) all tiles resolved in
// parallel, one lane per tile (each lane walks the <= 32 rows that start inside its tile).
// --------------------------------------------------------------------------
constexpr int HB = 32;          // real rows (and halo rows) per wave
constexpr int HSLOTS = 3;       // LDS slots per band tile / saved-register ring

struct HaloLds {                 // byte offsets inside dynamic LDS
    int tiles, qsave, zero, flag, bits, tbl, total;
};

__host__ __device__ inline HaloLds halo_lds_layout(int nw, int NT, int ROWS) {
    HaloLds L;
    int o = 0;
    L.tiles = o; o += nw * HSLOTS * HB * TILE_LD * 4;
    L.qsave = o; o += nw * HSLOTS * 64 * 4;
    L.zero = o;  o += TILE_LD * 4 + 16;
    L.flag = o;  o += 16;
    L.bits = o;  o += NT * (ROWS + 1) * 4;
    L.tbl = o;   o += ((NT * ROWS + 15) / 16) * 16;
    L.total = o;
    return L;
}

template <bool DIAG>
__device__ __forceinline__ void sweep_tile_halo(float &q, float &m, unsigned &nbits, int &B, const float4 (&vv)[8],
                                                int rrel, float negv) {
    float qb, cur;
    unsigned long long sM0, sM1;
#define ALIGNER_HALO_OPERANDS(G)                                                                         \
    : [qa] "+v"(q), [qb] "=&v"(qb), [m] "+v"(m), [bits] "+v"(nbits), [B] "+v"(B), [cur] "=&v"(cur),      \
      [sM0] "=&s"(sM0), [sM1] "=&s"(sM1)                                                                 \
    : [rrel] "v"(rrel), [neg] "v"(negv),                                                                 \
      [v0] "v"(vv[G].x), [v1] "v"(vv[G].y), [v2] "v"(vv[G].z), [v3] "v"(vv[G].w),                       \
      [v4] "v"(vv[G + 1].x), [v5] "v"(vv[G + 1].y), [v6] "v"(vv[G + 1].z), [v7] "v"(vv[G + 1].w),       \
      [v8] "v"(vv[G + 2].x), [v9] "v"(vv[G + 2].y), [v10] "v"(vv[G + 2].z), [v11] "v"(vv[G + 2].w),     \
      [v12] "v"(vv[G + 3].x), [v13] "v"(vv[G + 3].y), [v14] "v"(vv[G + 3].z), [v15] "v"(vv[G + 3].w)    \
    : "vcc"
    if (DIAG) {
        asm volatile(ALIGNER_HALO16_DIAG_0 ALIGNER_HALO_OPERANDS(0));
        asm volatile(ALIGNER_HALO16_DIAG_16 ALIGNER_HALO_OPERANDS(4));
    } else {
        asm volatile(ALIGNER_HALO16_0 ALIGNER_HALO_OPERANDS(0));
        asm volatile(ALIGNER_HALO16_16 ALIGNER_HALO_OPERANDS(4));
    }
#undef ALIGNER_HALO_OPERANDS
}

template <bool VEC, int MASKMODE>
__global__ __launch_bounds__(1024) void maxpath_halo_kernel(MaxpathParams p, int nw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x;
    const int b = blockIdx.x;
    int tx, ty;
    const int mode = classify_lengths(p, b, tx, ty);
    if (mode != MODE_NORMAL) { write_degenerate<MASKMODE>(p, b, mode, tx, ty, reinterpret_cast<int *>(smem)); return; }
    ALIGNER_STAMP(0);
    ALIGNER_STAMP(6);

    const HaloLds L = halo_lds_layout(nw, p.NT, p.ROWS);
    float *tiles = reinterpret_cast<float *>(smem + L.tiles);        // [nw][HSLOTS][32][TILE_LD]
    float *qsave = reinterpret_cast<float *>(smem + L.qsave);        // [nw][HSLOTS][64]
    float *zrow  = reinterpret_cast<float *>(smem + L.zero);         // 36 zeros (+ wave 0's frame-0 stream)
    int   *flagp = reinterpret_cast<int *>(smem + L.flag);
    unsigned *bitsL = reinterpret_cast<unsigned *>(smem + L.bits);   // [NT][ROWS+1]
    unsigned char *tbl = smem + L.tbl;                               // [NT][ROWS]
    const int RPB = p.ROWS + 1;
    const int ntb = (ty + TC - 1) / TC;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    const int nw_act = (tx + HB - 1) / HB;

    // zrow[0..35] = 0; zrow[36..39]: {max_neg_val, 0, 0, 0} = the first piece of "row -1" in tile 0
    for (int i = tid; i < TILE_LD + 4; i += nthreads) zrow[i] = (i == TILE_LD) ? p.neg : 0.0f;
    if (tid == 0) flagp[0] = 0;
    __syncthreads();

    if (!p.force_exact) {
        const int w = (wave < nw) ? wave : wave - nw;
        const bool active = w < nw_act;
        const int t_lo = active ? w : 0;                             // rows 32w.. start at frame 32w
        int t_hi = active ? (ty - tx + HB * w + HB - 1) / TC : -1;
        if (t_hi > ntb - 1) t_hi = ntb - 1;
        const int ntiles = t_hi - t_lo + 1;
        if (wave < nw) {
            // ------------------------------ compute wave ------------------------------
            const bool real = lane >= HB;
            const int row = HB * w + lane - HB;                      // halo lanes: rows of the wave above
            float q = p.neg;
            float m = (w == 0) ? p.neg : 0.0f;                       // lane 0 is never written by the DPP ops
            const float *band_lo = tiles + (w > 0 ? w - 1 : 0) * HSLOTS * HB * TILE_LD;   // halo rows' scores
            const float *band_hi = tiles + w * HSLOTS * HB * TILE_LD;
            const float *qs_in = qsave + (w > 0 ? w - 1 : 0) * HSLOTS * 64;
            float *qs_out = qsave + w * HSLOTS * 64;

            for (int i = 0; i < t_lo + w + 1; ++i) __syncthreads();
            for (int t = t_lo; t <= t_hi; ++t) {
                const int slot = t % HSLOTS;
                // ---- resynchronise the halo lanes from the upper wave's registers after tile t-1 ----
                if (w == 0) {
                    if (!real) q = (t == 0 && lane == HB - 1) ? 0.0f : p.neg;      // Q[-1,-1] = 0 (core.pyx:24-27)
                } else {
                    const float hq = qs_in[((t + HSLOTS - 1) % HSLOTS) * 64 + (real ? lane : lane + HB)];
                    q = real ? q : hq;
                }
                // ---- scores: one ds_read_b128 per 4 frames (halo lanes read the band above) ----
                const float *base;
                if (real) base = band_hi + (slot * HB + (lane - HB)) * TILE_LD;
                else if (w > 0) base = band_lo + (slot * HB + lane) * TILE_LD;
                else base = zrow;
                const lds_f32x4 *src = (const lds_f32x4 *)base;
                float4 vv[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 r = src[g];
                    vv[g] = make_float4(r.x, r.y, r.z, r.w);
                }
                if (w == 0 && t == 0 && lane == HB - 1) vv[0].x = p.neg;           // Q[-1,0] = max_neg_val
                const int y0 = t * TC;
                unsigned nbits = 0u;
                int B = lane;
                if (t == w) sweep_tile_halo<true>(q, m, nbits, B, vv, row - y0, p.neg);   // the diagonal's tile
                else        sweep_tile_halo<false>(q, m, nbits, B, vv, row - y0, p.neg);
                qs_out[slot * 64 + lane] = q;                                      // for the wave below
                if (real) {
                    bitsL[t * RPB + row] = ~nbits;                                 // frame 32t+c <-> bit 31-c
                    tbl[t * p.ROWS + row] = (unsigned char)(lane - B);             // rows climbed inside tile t
                }
                __syncthreads();
            }
            for (int i = 0; i < ntb + nw - w - t_hi - 2; ++i) __syncthreads();
        } else {
            // ------------------------------ loader wave -------------------------------
            if (active) {
                const int rr = lane >> 3, cg = lane & 7;
                float *mytiles = tiles + w * HSLOTS * HB * TILE_LD + rr * TILE_LD + 4 * cg;
                const float *ub = p.value + ubase;
                const float *mb = (MASKMODE == 1) ? reinterpret_cast<const float *>(p.mask) + ubase : nullptr;
                unsigned rowoff[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int r = HB * w + 8 * k + rr;
                    r = r > tx - 1 ? tx - 1 : r;                    // padding rows: replay the last real row
                    rowoff[k] = (unsigned)r * (unsigned)p.Ty;
                }
                unsigned nf = 0u;
                constexpr int DEPTH = 4;
                float4 buf[DEPTH][4];
                auto issue = [&](float4 (&dst)[4], int t) {
                    const int tc = t < t_hi ? t : t_hi;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        dst[k] = load_tile_piece<VEC, MASKMODE>(ub, mb, rowoff[k], TC * tc + 4 * cg, p.Ty);
                };
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) issue(buf[d], t_lo + d);
                for (int i = 0; i < t_lo + w; ++i) __syncthreads();
                for (int i0 = 0; i0 < ntiles; i0 += DEPTH) {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        const int t = t_lo + i0 + d;
                        if (t <= t_hi) {
                            float *dst = mytiles + (t % HSLOTS) * HB * TILE_LD;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float4 v = buf[d][k];
                                if (t == ntb - 1) {
                                    const int c0 = TC * t + 4 * cg;
                                    v.x = (c0 + 0 < ty) ? v.x : 0.f; v.y = (c0 + 1 < ty) ? v.y : 0.f;
                                    v.z = (c0 + 2 < ty) ? v.z : 0.f; v.w = (c0 + 3 < ty) ? v.w : 0.f;
                                }
                                *reinterpret_cast<float4 *>(dst + 8 * k * TILE_LD) = v;
                                const unsigned a = absbits(v.x), bb = absbits(v.y), c = absbits(v.z), dd = absbits(v.w);
                                const unsigned ab = a > bb ? a : bb, cd = c > dd ? c : dd;
                                const unsigned mx = ab > cd ? ab : cd;
                                nf = nf > mx ? nf : mx;
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        issue(buf[d], t + DEPTH);
                        __builtin_amdgcn_sched_barrier(0);
                        if (t <= t_hi) __syncthreads();
                    }
                }
                if (nf >= 0x7F800000u) flagp[0] = 1;
                for (int i = 0; i < ntb + nw - w - t_hi - 1; ++i) __syncthreads();
            } else {
                for (int i = 0; i < ntb + nw; ++i) __syncthreads();
            }
        }
        __syncthreads();
    }
    const bool exact = p.force_exact || flagp[0] != 0;
    if (exact) exact_fallback_sweep<MASKMODE>(p, b, tx, ty, smem, bitsL, RPB);
    ALIGNER_STAMP(1);
    __syncthreads();

    // ------------------------------ backtrack ------------------------------
    int *startsL = reinterpret_cast<int *>(smem);                    // overlays the (dead) score tiles
    int *Rrow = startsL + ((p.Tx + 1 + 63) / 64) * 64 + 64;          // [NT] path row at each tile's last frame
    if (exact) {
        // no transition tables on this path: the generic row walk
        MaxpathParams pe = p;
        pe.bits_in_lds = 1;
        pe.lds_bits_off = L.bits;
        backtrack_and_store(pe, b, tx, ty, smem);
        return;
    }
    if (wave == 0) {
        // (1) serial walk through the last tile: rows that start at or after frame 32*(ntb-1)
        int x = __builtin_amdgcn_readfirstlane(tx - 1);
        int e = __builtin_amdgcn_readfirstlane(ty - 1);
        const int tl = ntb - 1;
        while (x >= 1) {
            const unsigned word = bitsL[tl * RPB + x];                // uniform address: broadcast read
            const unsigned mw = __builtin_amdgcn_readfirstlane(word & (0xFFFFFFFFu << ((~e) & (TC - 1))));
            if (mw == 0u) break;                                     // row x started in an earlier tile
            const int s = (e | (TC - 1)) - __builtin_ctz(mw);
            if (lane == 0) startsL[x] = s;
            e = s - 1;
            x -= 1;
            if (e < tl * TC) break;
        }
        // (2) the path's row at the last frame of every earlier tile: NT-2 dependent look-ups
        int r = x;
        if (lane == 0 && tl >= 1) Rrow[tl - 1] = r;
        for (int t = tl - 1; t >= 1; --t) {
            const int mv = tbl[t * p.ROWS + r];
            r = __builtin_amdgcn_readfirstlane(r - mv);
            r = r < 0 ? 0 : r;
            if (lane == 0) Rrow[t - 1] = r;
        }
    }
    __syncthreads();
    if (wave == 0 && lane < ntb - 1) {
        // (3) lane = tile: rows (R[t-1], R[t]] start inside tile t; highest set bit at or below `lim`
        const int t = lane;
        int xr = Rrow[t];
        const int xstop = (t >= 1) ? Rrow[t - 1] : 0;
        int lim = TC - 1;
        while (xr > xstop) {
            const unsigned wd = bitsL[t * RPB + xr] & (0xFFFFFFFFu << (TC - 1 - lim));
            if (wd == 0u) { atomicOr(p.status, ALIGNER_ST_INTERNAL); break; }   // cannot happen
            const int c = (TC - 1) - __builtin_ctz(wd);
            startsL[xr] = t * TC + c;
            lim = c - 1;
            xr -= 1;
            if (lim < 0 && xr > xstop) { atomicOr(p.status, ALIGNER_ST_INTERNAL); break; }
        }
    }
    if (tid == 0) startsL[0] = 0;                                    // row 0 starts at frame 0
    __syncthreads();
    for (int r2 = tx + tid; r2 <= p.Tx; r2 += nthreads) startsL[r2] = ty;
    __syncthreads();
    ALIGNER_STAMP(3);
    store_outputs(p, b, tx, ty, startsL);
    ALIGNER_STAMP(5);
    ALIGNER_STAMP(7);
}

// --------------------------------------------------------------------------
// starts -> dense 0/1 path in the caller's dtype (the reference's return value,
// __init__.py:21).  Pure streaming store: path[b,x,y] = starts[x] <= y < starts[x+1].
// --------------------------------------------------------------------------
template <typename T, bool VEC>
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
            *reinterpret_cast<Vec4 *>(dst) = o;
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
template <typename T>
__global__ __launch_bounds__(256) void lengths_kernel(const T *__restrict__ mask, int Tx, int Ty,
                                                       int *__restrict__ t_xs, int *__restrict__ t_ys) {
    __shared__ float red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const T *m = mask + (size_t)b * Tx * Ty;
    float sx = 0.f, sy = 0.f;
    for (int x = tid; x < Tx; x += 256) sx += (float)m[(size_t)x * Ty];   // mask[b, x, 0]
    for (int y = tid; y < Ty; y += 256) sy += (float)m[y];                // mask[b, 0, y]
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

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------

struct WsLayout {
    size_t status_off, len_off, starts_off, bits_off, total;
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
    L.starts_off = align_up(L.len_off + (size_t)2 * B * sizeof(int), 256);
    L.bits_off = align_up(L.starts_off + (size_t)B * (Tx + 1) * sizeof(int), 256);
    L.total = align_up(L.bits_off + (size_t)B * L.NT * L.ROWS * sizeof(unsigned), 256);
    return L;
}

static int lds_limit() { return device_lds_limit(); }

static size_t starts_bytes(int Tx) { return (size_t)(((Tx + 1 + 63) / 64) * 64 + 64) * 4; }

// Backtrack overlay size for a window of WT tiles (decision words in global memory).
static size_t walk_bytes(int WT, int ROWS, int Tx) { return starts_bytes(Tx) + (size_t)WT * (ROWS + 1) * 4; }

static int pick_window(int NT, int ROWS, int Tx, size_t budget) {
    int WT = NT < 64 ? NT : 64;
    while (WT > 1 && walk_bytes(WT, ROWS, Tx) > budget) --WT;
    return walk_bytes(WT, ROWS, Tx) <= budget ? WT : 0;
}

template <typename K>
static int launch_with_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, MaxpathParams p) {
    ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), lds));
    hipLaunchKernelGGL(kernel, grid, block, lds, s, p);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

template <int NW, int DEPTH>
static int launch_pipelined(MaxpathParams p, bool vec, int maskmode, size_t lds, hipStream_t s) {
    dim3 grid(p.B), block(NW * 128);
    if (vec) {
        if (maskmode == 0) return launch_with_lds(maxpath_pipelined_kernel<NW, DEPTH, true, 0>, grid, block, lds, s, p);
        return launch_with_lds(maxpath_pipelined_kernel<NW, DEPTH, true, 1>, grid, block, lds, s, p);
    }
    if (maskmode == 0) return launch_with_lds(maxpath_pipelined_kernel<NW, DEPTH, false, 0>, grid, block, lds, s, p);
    return launch_with_lds(maxpath_pipelined_kernel<NW, DEPTH, false, 1>, grid, block, lds, s, p);
}

template <int R>
static int launch_generic(MaxpathParams p, int maskmode, size_t lds, hipStream_t s) {
    dim3 grid(p.B), block(256);
    if (maskmode == 0) return launch_with_lds(maxpath_generic_kernel<R, 0>, grid, block, lds, s, p);
    return launch_with_lds(maxpath_generic_kernel<R, 1>, grid, block, lds, s, p);
}

static int forward_impl(const float *value, const void *mask, int mask_dtype, const int32_t *t_xs,
                        const int32_t *t_ys, int32_t *tok_out, int32_t *dur_out, void *ws,
                        size_t ws_bytes, int B, int Tx, int Ty, float neg, int flags, hipStream_t s) {
    if (!value || !ws) return fail(ALIGNER_EINVAL, "value/workspace pointer is null");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if ((size_t)Tx * (size_t)Ty >= (1ull << 31))
        return fail(ALIGNER_EDOM, "Tx*Ty=%zu exceeds 2^31", (size_t)Tx * Ty);
    if ((!t_xs || !t_ys) && !mask)
        return fail(ALIGNER_EINVAL, "need either lengths (t_xs,t_ys) or a mask to derive them from");
    if ((flags & ALIGNER_F_STRICT_MASK) && !mask)
        return fail(ALIGNER_EINVAL, "ALIGNER_F_STRICT_MASK needs a mask");
    if ((flags & ALIGNER_F_STRICT_MASK) && mask_dtype != ALIGNER_DT_F32)
        return fail(ALIGNER_EINVAL, "strict mask must be fp32 (dtype %d)", mask_dtype);
    if (B == 0) return ALIGNER_OK;
    const WsLayout L = ws_layout(B, Tx, Ty);
    if (ws_bytes < L.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", ws_bytes, L.total);
    unsigned char *wsb = static_cast<unsigned char *>(ws);

    if (!t_xs || !t_ys) {
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
    p.B = B; p.Tx = Tx; p.Ty = Ty; p.NT = L.NT; p.ROWS = L.ROWS;
    p.neg = neg; p.flags = flags;
    p.bits_in_lds = 0; p.lds_bits_off = 0;
    p.force_exact = !(neg - neg == 0.0f);            // NaN / inf max_neg_val
    p.stamps = g_debug_stamps;
    const int maskmode = (flags & ALIGNER_F_STRICT_MASK) ? 1 : 0;
    const size_t lds_max = (size_t)lds_limit();
    // vec: 16-byte loads; the pipelined kernel's loaders address an utterance with 32-bit byte offsets
    const bool vec = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) & 15) == 0) &&
                     (!maskmode || (reinterpret_cast<uintptr_t>(mask) & 15) == 0) &&
                     (size_t)Tx * (size_t)Ty * 4 < (1ull << 31);

    // halo-lane kernel (opt-in: measured on par with the 63-rows-per-wave kernel at [64,200,1000] --
    // 5 instead of 6 issues per frame and a 3x faster table-driven backtrack, but 7 instead of 4
    // compute waves share the CU's SIMDs and LDS)
    if ((flags & ALIGNER_F_FORCE_HALO) && !(flags & ALIGNER_F_FORCE_GENERIC) && Tx <= 8 * HB && L.NT <= 64) {
        const int nwh = (Tx + HB - 1) / HB;
        const HaloLds HL = halo_lds_layout(nwh, L.NT, L.ROWS);
        const size_t fb = (size_t)2 * (2 * nwh * 64 + 1) * 4;            // exact fallback's column buffers
        if ((size_t)HL.total <= lds_max && starts_bytes(Tx) + (size_t)L.NT * 4 <= (size_t)HL.bits &&
            fb <= (size_t)HL.bits) {
            p.WT = L.NT;
            p.bits_in_lds = 1;
            p.lds_bits_off = HL.bits;
            dim3 grid(B), block(2 * nwh * 64);
            auto launch = [&](auto kern) {
                hipError_t e_ = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (size_t)HL.total);
                if (e_ != hipSuccess) return fail(ALIGNER_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e_));
                hipLaunchKernelGGL(kern, grid, block, (size_t)HL.total, s, p, nwh);
                hipError_t e2 = hipGetLastError();
                if (e2 != hipSuccess) return fail(ALIGNER_EHIP, "launch failed: %s", hipGetErrorString(e2));
                return (int)ALIGNER_OK;
            };
            if (vec) return maskmode ? launch(maxpath_halo_kernel<true, 1>) : launch(maxpath_halo_kernel<true, 0>);
            return maskmode ? launch(maxpath_halo_kernel<false, 1>) : launch(maxpath_halo_kernel<false, 0>);
        }
    }

    const int nw_need = (Tx + RPW - 1) / RPW;
    if (!(flags & ALIGNER_F_FORCE_GENERIC) && nw_need <= 8) {
        const int NW = nw_need <= 1 ? 1 : nw_need <= 2 ? 2 : nw_need <= 4 ? 4 : 8;
        const size_t fwd = align_up((size_t)NW * (2 * 64 * TILE_LD + RING_T * RING_LD) * 4 + 16, 16);
        if (fwd <= lds_max && starts_bytes(Tx) <= fwd) {
            size_t lds = 0;
            const size_t bits_lds = (size_t)L.NT * (L.ROWS + 1) * 4;
            if (L.NT <= 64 && fwd + bits_lds <= lds_max) {
                p.bits_in_lds = 1;
                p.lds_bits_off = (int)fwd;
                p.WT = L.NT;
                lds = fwd + bits_lds;
            } else {
                p.WT = pick_window(L.NT, L.ROWS, Tx, lds_max);
                if (p.WT > 0) {
                    lds = walk_bytes(p.WT, L.ROWS, Tx);
                    if (lds < fwd) lds = fwd;
                }
            }
            if (lds) {
                switch (NW) {
                    case 1: return launch_pipelined<1, 4>(p, vec, maskmode, lds, s);
                    case 2: return launch_pipelined<2, 4>(p, vec, maskmode, lds, s);
                    case 4: return launch_pipelined<4, 2>(p, vec, maskmode, lds, s);
                    default: return launch_pipelined<8, 2>(p, vec, maskmode, lds, s);
                }
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
        case 1: return launch_generic<1>(p, maskmode, lds, s);
        case 2: return launch_generic<2>(p, maskmode, lds, s);
        case 4: return launch_generic<4>(p, maskmode, lds, s);
        default: return launch_generic<8>(p, maskmode, lds, s);
    }
}

template <typename T>
static int launch_expand(const int *starts, void *path, int B, int Tx, int Ty, T one, hipStream_t s) {
    const int rpb = 8;
    dim3 grid((Ty + 1023) / 1024, (Tx + rpb - 1) / rpb, B), block(256);
    const bool vec = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(path) % (sizeof(T) * 4)) == 0);
    if (vec)
        hipLaunchKernelGGL((expand_kernel<T, true>), grid, block, 0, s, starts, static_cast<T *>(path), Tx, Ty, rpb, one);
    else
        hipLaunchKernelGGL((expand_kernel<T, false>), grid, block, 0, s, starts, static_cast<T *>(path), Tx, Ty, rpb, one);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

static int expand_impl(const int *starts, void *path, int path_dtype, int B, int Tx, int Ty, hipStream_t s) {
    if (B > 65535 || (Tx + 7) / 8 > 65535) return fail(ALIGNER_EDOM, "grid too large");
    switch (path_dtype) {
        case ALIGNER_DT_F32: return launch_expand<float>(starts, path, B, Tx, Ty, 1.0f, s);
        case ALIGNER_DT_F64: return launch_expand<double>(starts, path, B, Tx, Ty, 1.0, s);
        case ALIGNER_DT_I32: return launch_expand<int32_t>(starts, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_I64: return launch_expand<int64_t>(starts, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_U8:  return launch_expand<uint8_t>(starts, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_F16: return launch_expand<uint16_t>(starts, path, B, Tx, Ty, 0x3C00, s);
        case ALIGNER_DT_BF16: return launch_expand<uint16_t>(starts, path, B, Tx, Ty, 0x3F80, s);
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
        default:
            return fail(ALIGNER_EINVAL, "mask dtype %d not supported (use F32, U8 or I32)", mask_dtype);
    }
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

int aligner_maxpath_forward_f32(const float *value, const void *mask, int mask_dtype,
                                const int32_t *t_xs, const int32_t *t_ys, int32_t *tok_out,
                                int32_t *dur_out, void *ws, size_t ws_bytes, int B, int Tx, int Ty,
                                float max_neg_val, int flags, void *stream) {
    return forward_impl(value, mask, mask_dtype, t_xs, t_ys, tok_out, dur_out, ws, ws_bytes, B, Tx, Ty,
                        max_neg_val, flags, static_cast<hipStream_t>(stream));
}

int aligner_maxpath_expand(const void *ws, void *path, int path_dtype, int B, int Tx, int Ty,
                           void *stream) {
    if (!ws || !path) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (B == 0) return ALIGNER_OK;
    const WsLayout L = ws_layout(B, Tx, Ty);
    const int *starts = reinterpret_cast<const int *>(static_cast<const unsigned char *>(ws) + L.starts_off);
    return expand_impl(starts, path, path_dtype, B, Tx, Ty, static_cast<hipStream_t>(stream));
}

int aligner_maxpath_f32(const float *value, const void *mask, int mask_dtype, const int32_t *t_xs,
                        const int32_t *t_ys, void *path_out, int path_dtype, int32_t *tok_out,
                        int32_t *dur_out, void *ws, size_t ws_bytes, int B, int Tx, int Ty,
                        float max_neg_val, int flags, void *stream) {
    if (path_out && dtype_size(path_dtype) == 0)
        return fail(ALIGNER_EINVAL, "path dtype %d not supported", path_dtype);
    int rc = forward_impl(value, mask, mask_dtype, t_xs, t_ys, tok_out, dur_out, ws, ws_bytes, B, Tx, Ty,
                          max_neg_val, flags, static_cast<hipStream_t>(stream));
    if (rc || !path_out || B == 0) return rc;
    return aligner_maxpath_expand(ws, path_out, path_dtype, B, Tx, Ty, stream);
}

int aligner_maxpath_read_status(void *ws, int32_t *status_host, void *stream) {
    if (!ws || !status_host) return fail(ALIGNER_EINVAL, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ALIGNER_HIP_CHECK(hipMemcpyAsync(status_host, ws, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    ALIGNER_HIP_CHECK(hipMemsetAsync(ws, 0, sizeof(int32_t), s));     // status is sticky until read
    ALIGNER_HIP_CHECK(hipStreamSynchronize(s));
    return ALIGNER_OK;
}

int aligner_maxpath_host_f32(int32_t *paths, const float *values, const int32_t *t_xs,
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
#undef HOST_TRY
    if (d_val) (void)hipFree(d_val);
    if (d_path) (void)hipFree(d_path);
    if (d_len) (void)hipFree(d_len);
    if (d_ws) (void)hipFree(d_ws);
    return rc;
}

}  // extern "C"
