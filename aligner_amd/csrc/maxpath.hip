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
//    live on lanes (one row per lane, 64 rows per wave), the running column Q
//    stays in a VGPR and the "row above" operand is a DPP wave-shift -- never LDS.
//  * Waves of a workgroup form a systolic pipeline over 32-frame tiles: in phase
//    p wave w sweeps tile p-w and receives the boundary row of wave w-1 for that
//    tile through a small LDS ring; one s_barrier per phase.
//  * The score tensor is row-major with the mel axis contiguous, so a lane-per-
//    row frame read is strided.  Dedicated loader waves (one per compute wave)
//    stream it with coalesced 16-byte loads, DEPTH tiles in flight in registers,
//    and transpose through a padded LDS tile that the compute lanes read back
//    conflict-free with ds_read_b128 (4 frames per read).
//  * Q is never stored.  Each cell leaves one decision bit
//        dec = (x == y) | (Q[x-1,y-1] > Q[x,y-1])
//    which is exactly the reference's backtrack predicate (core.pyx:34); 32
//    frames make one word per lane.  The backtrack walks text rows (not frames):
//    for row x ending at frame e the start is the highest set bit <= e, found by
//    one ballot over the row's words (lane = tile).
//
// Only cells inside the reference's band (core.pyx:18) influence the result;
// cells outside it are either skipped (whole tiles) or computed and ignored --
// they are provably never read by an in-band cell.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

constexpr int TC       = 32;  // frames per tile == decision bits per word
constexpr int TILE_LD  = 36;  // dwords per LDS tile row: 32 + 4 pad -> 16B-slot stride 9 (odd)
constexpr int RING_T   = 4;   // boundary ring depth in tiles
constexpr int DUMMY_W  = 96;  // per-wave scratch words for the off-lane boundary stores
constexpr int WS_HDR_BYTES = 256;

enum Mode { MODE_NORMAL = 0, MODE_EMPTY = 1, MODE_COMPAT = 2 };

// LDS-qualified float: keeps per-lane selected addresses as ds_* instructions
// (a generic pointer would turn them into flat_* accesses).
typedef __attribute__((address_space(3))) float lds_float;

struct MaxpathParams {
    const float *value;
    const void  *mask;      // strict-mask operand (nullable)
    const int   *t_xs;
    const int   *t_ys;
    int         *tok;       // [B,Ty] (never null inside the kernels)
    int         *dur;       // [B,Tx] nullable
    unsigned    *bits;      // [B,NT,ROWS] decision words
    int         *status;
    int B, Tx, Ty, NT, ROWS;
    int WT;                 // tiles per backtrack window (<= 64)
    float neg;
    int flags;
};

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
        mode = MODE_COMPAT;                     // reference: row t_x-1 all ones
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

// Outputs for the degenerate modes (whole block, uniform).
__device__ void write_degenerate(const MaxpathParams &p, int b, int mode, int tx, int ty) {
    for (int y = threadIdx.x; y < p.Ty; y += blockDim.x)
        p.tok[(size_t)b * p.Ty + y] = (mode == MODE_COMPAT && y < ty) ? (tx - 1) : -1;
    if (p.dur)
        for (int x = threadIdx.x; x < p.Tx; x += blockDim.x)
            p.dur[(size_t)b * p.Tx + x] = (mode == MODE_COMPAT && x == tx - 1) ? ty : 0;
}

// --------------------------------------------------------------------------
// Backtrack over decision words (shared by both forward kernels).
//
// LDS overlay (valid once the forward sweep is done):
//   win  [WT][RP]   decision words of the current window, RP = ROWS+1 (odd-ish
//                   stride: lane=tile reads of one row are conflict-free)
//   tokL [Ty]       token index per frame
//   durL [ROWS]     frames per token
// Windows are visited from the last tile to the first; inside a window wave 0
// walks rows downwards.  State (x = current row, e = its last frame) is
// wave-uniform.
// --------------------------------------------------------------------------
__device__ void backtrack_and_store(const MaxpathParams &p, int b, int tx, int ty,
                                    unsigned char *smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int nthreads = blockDim.x;
    const int RP = p.ROWS + 1;
    unsigned *win = reinterpret_cast<unsigned *>(smem);
    int *tokL = reinterpret_cast<int *>(win + (size_t)p.WT * RP);
    int *durL = tokL + p.Ty;

    const int ntb = (ty + TC - 1) / TC;            // tiles this utterance uses
    const int rows_used = ((tx + 63) / 64) * 64;   // rows whose words exist
    const unsigned *gbits = p.bits + (size_t)b * p.NT * p.ROWS;

    int x = tx - 1;   // core.pyx:15
    int e = ty - 1;

    for (int jhi = ntb; jhi > 0; jhi -= p.WT) {
        const int jb = (jhi - p.WT > 0) ? (jhi - p.WT) : 0;
        const int ntw = jhi - jb;
        __syncthreads();                           // previous window fully consumed
        for (int idx = tid; idx < ntw * rows_used; idx += nthreads) {
            const int j = idx / rows_used, r = idx - j * rows_used;
            win[j * RP + r] = gbits[(size_t)(jb + j) * p.ROWS + r];
        }
        __syncthreads();
        if (tid < 64) {
            const int col0 = (jb + lane) * TC;
            unsigned wn = 0;
            if (x >= 1 && lane < ntw) wn = win[lane * RP + x];
            while (x >= 1) {
                const unsigned w = wn;
                if (x >= 2 && lane < ntw) wn = win[lane * RP + (x - 1)];   // prefetch next row
                const int lim = e - col0;          // frames 0..lim of my tile are <= e
                unsigned m;
                if (lim >= TC - 1)      m = w;
                else if (lim < 0)       m = 0u;
                else                    m = w & (0xFFFFFFFFu << (TC - 1 - lim));
                if (lane >= ntw) m = 0u;
                const unsigned long long bal = __ballot(m != 0u);
                if (bal == 0ull) break;            // row x starts in an earlier window
                const int js = 63 - __builtin_clzll(bal);
                const unsigned word = (unsigned)__builtin_amdgcn_readlane((int)m, js);
                const int s = (jb + js) * TC + (TC - 1) - __builtin_ctz(word);
                // row x owns frames [s, e]   (core.pyx:33-35: path[index,y]=1 until the move)
                if (lane == 0) durL[x] = e - s + 1;
                for (int y = s + lane; y <= e; y += 64) tokL[y] = x;
                e = s - 1;
                x -= 1;
            }
        }
    }
    if (tid < 64) {
        // x == 0 here for every valid input (forced diagonal, core.pyx:34): row 0 takes the rest.
        if (x != 0 && lane == 0) atomicOr(p.status, ALIGNER_ST_INTERNAL);
        if (lane == 0) durL[0] = e + 1;
        for (int y = lane; y <= e; y += 64) tokL[y] = 0;
    }
    __syncthreads();
    for (int y = tid; y < p.Ty; y += nthreads)
        p.tok[(size_t)b * p.Ty + y] = (y < ty) ? tokL[y] : -1;
    if (p.dur)
        for (int r = tid; r < p.Tx; r += nthreads)
            p.dur[(size_t)b * p.Tx + r] = (r < tx) ? durL[r] : 0;
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
    if (mode != MODE_NORMAL) { write_degenerate(p, b, mode, tx, ty); return; }

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
    backtrack_and_store(p, b, tx, ty, smem);
}

// --------------------------------------------------------------------------
// Pipelined forward kernel: NW compute waves + NW loader waves.
// --------------------------------------------------------------------------
template <bool VEC, int MASKMODE>
__device__ __forceinline__ float4 load_tile_piece(const MaxpathParams &p, size_t row_off, int col) {
    // 4 consecutive frames of one row; frames >= Ty read as 0 and never matter.
    const float *src = p.value + row_off + col;
    float4 v;
    if (VEC && col + 4 <= p.Ty) {
        v = *reinterpret_cast<const float4 *>(src);
        if (MASKMODE == 1) {
            const float4 m = *reinterpret_cast<const float4 *>(
                reinterpret_cast<const float *>(p.mask) + row_off + col);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
    } else {
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t[i] = 0.0f;
            if (col + i < p.Ty) {
                t[i] = src[i];
                if (MASKMODE == 1)
                    t[i] *= reinterpret_cast<const float *>(p.mask)[row_off + col + i];
            }
        }
        v = make_float4(t[0], t[1], t[2], t[3]);
    }
    return v;
}

template <bool DIAG, bool WAVE0, bool PUBLISH>
__device__ __forceinline__ void sweep_tile(float &q, unsigned &bits, const float4 (&vv)[8],
                                           const float4 (&bc)[8], float bprev, int row, int y0,
                                           float neg, lds_float *pub) {
#pragma unroll
    for (int k = 0; k < TC; ++k) {
        float bnd;
        if (WAVE0) {
            bnd = bprev;                      // 0 at (t=0,k=0), max_neg_val otherwise
            bprev = neg;
        } else {
            bnd = (k == 0) ? bprev : comp(bc[(k - 1) >> 2], (k - 1) & 3);
        }
        const float up = dpp_wave_shr1(bnd, q);                 // Q[x-1, y-1]
        float cur = q;                                          // Q[x,   y-1]
        if (DIAG) cur = (row == y0 + k) ? neg : q;              // core.pyx:19-20
        const bool adv = up > cur;                              // core.c:19384 (NaN -> keep cur)
        q = (adv ? up : cur) + comp(vv[k >> 2], k & 3);          // core.pyx:30
        // decision bit = the backtrack predicate (core.pyx:34): on the diagonal the move is
        // forced whatever the scores are (NaN / scores below max_neg_val included)
        const bool dec = DIAG ? (adv || row == y0 + k) : adv;
        bits = (bits << 1) | (dec ? 1u : 0u);
        if (PUBLISH) pub[k] = q;                                // lane 63 -> ring, others -> scratch
    }
}

template <int NW, int DEPTH, bool VEC, int MASKMODE>
__global__ __launch_bounds__(NW * 128) void maxpath_pipelined_kernel(MaxpathParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    int tx, ty;
    const int mode = classify_lengths(p, b, tx, ty);
    if (mode != MODE_NORMAL) { write_degenerate(p, b, mode, tx, ty); return; }

    float *tiles = reinterpret_cast<float *>(smem);              // [NW][2][64][TILE_LD]
    float *ring  = tiles + NW * 2 * 64 * TILE_LD;                // [NW][RING_T][TC]
    float *dummy = ring + NW * RING_T * TC;                      // [NW][DUMMY_W]
    const int ntb = (ty + TC - 1) / TC;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;

    if (wave < NW) {
        // ------------------------------ compute wave ------------------------------
        const int w = wave;
        const int row = 64 * w + lane;
        const bool active = 64 * w < tx;
        float q = p.neg;
        unsigned bits = 0u;
        float *mytiles = tiles + w * 2 * 64 * TILE_LD;
        const float *myring = ring + w * RING_T * TC;
        float *outring = ring + (w + 1 < NW ? w + 1 : w) * RING_T * TC;
        const bool publish = (w + 1 < NW) && (64 * (w + 1) < tx);
        unsigned *gbits = p.bits + (size_t)b * p.NT * p.ROWS + row;

        for (int i = 0; i < w + 1; ++i) __syncthreads();
        for (int t = 0; t < ntb; ++t) {
            if (active && tile_in_band(t, 64 * w, 64, tx, ty)) {
                const float *tile = mytiles + (t & 1) * 64 * TILE_LD + lane * TILE_LD;
                float4 vv[8], bc[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) vv[g] = *reinterpret_cast<const float4 *>(tile + 4 * g);
                float bprev;
                if (w == 0) {
                    bprev = (t == 0) ? 0.0f : p.neg;                         // core.pyx:24-27
                } else {
                    const float *rs = myring + (t & (RING_T - 1)) * TC;
#pragma unroll
                    for (int g = 0; g < 8; ++g) bc[g] = *reinterpret_cast<const float4 *>(rs + 4 * g);
                    bprev = myring[((t - 1) & (RING_T - 1)) * TC + (TC - 1)];
                }
                lds_float *pub = (lds_float *)((lane == 63) ? (outring + (t & (RING_T - 1)) * TC)
                                                            : (dummy + w * DUMMY_W + lane));
                const int y0 = t * TC;
                const bool diag = (y0 <= 64 * w + 63) && (y0 + TC - 1 >= 64 * w);
                if (w == 0) {
                    if (diag) { if (publish) sweep_tile<true, true, true>(q, bits, vv, bc, bprev, row, y0, p.neg, pub);
                                else         sweep_tile<true, true, false>(q, bits, vv, bc, bprev, row, y0, p.neg, pub); }
                    else      { if (publish) sweep_tile<false, true, true>(q, bits, vv, bc, bprev, row, y0, p.neg, pub);
                                else         sweep_tile<false, true, false>(q, bits, vv, bc, bprev, row, y0, p.neg, pub); }
                } else {
                    if (diag) { if (publish) sweep_tile<true, false, true>(q, bits, vv, bc, bprev, row, y0, p.neg, pub);
                                else         sweep_tile<true, false, false>(q, bits, vv, bc, bprev, row, y0, p.neg, pub); }
                    else      { if (publish) sweep_tile<false, false, true>(q, bits, vv, bc, bprev, row, y0, p.neg, pub);
                                else         sweep_tile<false, false, false>(q, bits, vv, bc, bprev, row, y0, p.neg, pub); }
                }
                gbits[(size_t)t * p.ROWS] = bits;      // frame 32t+c <-> bit 31-c
                bits = 0u;
            }
            __syncthreads();
        }
        for (int i = 0; i < NW - 1 - w; ++i) __syncthreads();
    } else {
        // ------------------------------ loader wave -------------------------------
        const int w = wave - NW;
        const bool active = 64 * w < tx;
        const int rr = lane >> 3, cg = lane & 7;
        float *mytiles = tiles + w * 2 * 64 * TILE_LD;
        size_t row_off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int r = 64 * w + 8 * k + rr;
            if (r > p.Tx - 1) r = p.Tx - 1;            // rows >= Tx: any in-bounds row will do
            row_off[k] = ubase + (size_t)r * p.Ty;
        }
        float4 buf[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (active && d < ntb && tile_in_band(d, 64 * w, 64, tx, ty)) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    buf[d][k] = load_tile_piece<VEC, MASKMODE>(p, row_off[k], TC * d + 4 * cg);
            }
        }
        for (int i = 0; i < w; ++i) __syncthreads();
        for (int t0 = 0; t0 < ntb; t0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int t = t0 + d;
                if (t < ntb) {
                    if (active && tile_in_band(t, 64 * w, 64, tx, ty)) {
                        float *dst = mytiles + (t & 1) * 64 * TILE_LD + rr * TILE_LD + 4 * cg;
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            *reinterpret_cast<float4 *>(dst + 8 * k * TILE_LD) = buf[d][k];
                    }
                    const int tn = t + DEPTH;
                    if (active && tn < ntb && tile_in_band(tn, 64 * w, 64, tx, ty)) {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            buf[d][k] = load_tile_piece<VEC, MASKMODE>(p, row_off[k], TC * tn + 4 * cg);
                    }
                    __syncthreads();
                }
            }
        }
        for (int i = 0; i < NW - w; ++i) __syncthreads();
    }
    // all decision words of this utterance are in global memory (same CU wrote them)
    __threadfence_block();
    backtrack_and_store(p, b, tx, ty, smem);
}

// --------------------------------------------------------------------------
// tok -> dense 0/1 path in the caller's dtype (the reference's return value,
// __init__.py:21).  Pure streaming store.
// --------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void expand_kernel(const int *__restrict__ tok, T *__restrict__ path,
                                                      int Tx, int Ty, int rows_per_block, T one) {
    const int b = blockIdx.z;
    const int x0 = blockIdx.y * rows_per_block;
    const int y0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y0 >= Ty) return;
    int tk[4];
    if (VEC) {
        const int4 v = *reinterpret_cast<const int4 *>(tok + (size_t)b * Ty + y0);
        tk[0] = v.x; tk[1] = v.y; tk[2] = v.z; tk[3] = v.w;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) tk[i] = (y0 + i < Ty) ? tok[(size_t)b * Ty + y0 + i] : -1;
    }
    const int x1 = (x0 + rows_per_block < Tx) ? x0 + rows_per_block : Tx;
    struct alignas(sizeof(T) * 4) Vec4 { T v[4]; };
    for (int x = x0; x < x1; ++x) {
        T *dst = path + ((size_t)b * Tx + x) * Ty + y0;
        Vec4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[i] = (tk[i] == x) ? one : T(0);
        if (VEC) {
            *reinterpret_cast<Vec4 *>(dst) = o;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (y0 + i < Ty) dst[i] = o.v[i];
        }
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
    size_t status_off, len_off, tok_off, bits_off, total;
    int NT, ROWS;
};

static WsLayout ws_layout(int B, int Tx, int Ty) {
    WsLayout L;
    L.NT = (Ty + TC - 1) / TC;
    L.ROWS = (Tx + 63) / 64 * 64;
    L.status_off = 0;
    L.len_off = WS_HDR_BYTES;
    L.tok_off = align_up(L.len_off + (size_t)2 * B * sizeof(int), 256);
    L.bits_off = align_up(L.tok_off + (size_t)B * Ty * sizeof(int), 256);
    L.total = align_up(L.bits_off + (size_t)B * L.NT * L.ROWS * sizeof(unsigned), 256);
    return L;
}

static int lds_limit() {
    // gfx950 lets one workgroup own the CU's whole 160 KiB LDS.
    static int lim = 0;
    if (!lim) {
        int dev = 0, v = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            std::strncmp(prop.gcnArchName, "gfx950", 6) == 0)
            lim = 160 * 1024;
        else if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
            lim = v;
        else
            lim = 64 * 1024;
    }
    return lim;
}

// Backtrack overlay size for a window of WT tiles.
static size_t walk_bytes(int WT, int ROWS, int Ty) {
    return ((size_t)WT * (ROWS + 1) + Ty + ROWS) * 4;
}

static int pick_window(int NT, int ROWS, int Ty, size_t budget) {
    int WT = NT < 64 ? NT : 64;
    while (WT > 1 && walk_bytes(WT, ROWS, Ty) > budget) --WT;
    return walk_bytes(WT, ROWS, Ty) <= budget ? WT : 0;
}

template <typename K>
static int launch_with_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, MaxpathParams p) {
    if (lds > 64 * 1024)
        ALIGNER_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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

    ALIGNER_HIP_CHECK(hipMemsetAsync(wsb + L.status_off, 0, WS_HDR_BYTES, s));
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
    p.tok = tok_out ? tok_out : reinterpret_cast<int *>(wsb + L.tok_off);
    p.dur = dur_out;
    p.bits = reinterpret_cast<unsigned *>(wsb + L.bits_off);
    p.status = reinterpret_cast<int *>(wsb + L.status_off);
    p.B = B; p.Tx = Tx; p.Ty = Ty; p.NT = L.NT; p.ROWS = L.ROWS;
    p.neg = neg; p.flags = flags;
    const int maskmode = (flags & ALIGNER_F_STRICT_MASK) ? 1 : 0;
    const size_t lds_max = (size_t)lds_limit();
    const bool vec = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) & 15) == 0) &&
                     (!maskmode || (reinterpret_cast<uintptr_t>(mask) & 15) == 0);

    const int nw_need = (Tx + 63) / 64;
    if (!(flags & ALIGNER_F_FORCE_GENERIC) && nw_need <= 8) {
        const int NW = nw_need <= 1 ? 1 : nw_need <= 2 ? 2 : nw_need <= 4 ? 4 : 8;
        const size_t fwd = (size_t)NW * (2 * 64 * TILE_LD + RING_T * TC + DUMMY_W) * 4;
        if (fwd <= lds_max) {
            p.WT = pick_window(L.NT, L.ROWS, Ty, lds_max);
            if (p.WT > 0) {
                size_t lds = walk_bytes(p.WT, L.ROWS, Ty);
                if (lds < fwd) lds = fwd;
                switch (NW) {
                    case 1: return launch_pipelined<1, 4>(p, vec, maskmode, lds, s);
                    case 2: return launch_pipelined<2, 4>(p, vec, maskmode, lds, s);
                    case 4: return launch_pipelined<4, 4>(p, vec, maskmode, lds, s);
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
    p.WT = pick_window(L.NT, L.ROWS, Ty, lds_max);
    if (p.WT <= 0) return fail(ALIGNER_EDOM, "Ty=%d too long for the backtrack window", Ty);
    size_t lds = walk_bytes(p.WT, L.ROWS, Ty);
    if (lds < fwd) lds = fwd;
    switch (RR) {
        case 1: return launch_generic<1>(p, maskmode, lds, s);
        case 2: return launch_generic<2>(p, maskmode, lds, s);
        case 4: return launch_generic<4>(p, maskmode, lds, s);
        default: return launch_generic<8>(p, maskmode, lds, s);
    }
}

template <typename T>
static int launch_expand(const int *tok, void *path, int B, int Tx, int Ty, T one, hipStream_t s) {
    const int rpb = 8;
    dim3 grid((Ty + 1023) / 1024, (Tx + rpb - 1) / rpb, B), block(256);
    const bool vec = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(path) % (sizeof(T) * 4)) == 0) &&
                     ((reinterpret_cast<uintptr_t>(tok) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL((expand_kernel<T, true>), grid, block, 0, s, tok, static_cast<T *>(path), Tx, Ty, rpb, one);
    else
        hipLaunchKernelGGL((expand_kernel<T, false>), grid, block, 0, s, tok, static_cast<T *>(path), Tx, Ty, rpb, one);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
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

int aligner_maxpath_expand(const int32_t *tok, void *path, int path_dtype, int B, int Tx, int Ty,
                           void *stream) {
    if (!tok || !path) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (B == 0) return ALIGNER_OK;
    if (B > 65535 || (Tx + 7) / 8 > 65535) return fail(ALIGNER_EDOM, "grid too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (path_dtype) {
        case ALIGNER_DT_F32: return launch_expand<float>(tok, path, B, Tx, Ty, 1.0f, s);
        case ALIGNER_DT_F64: return launch_expand<double>(tok, path, B, Tx, Ty, 1.0, s);
        case ALIGNER_DT_I32: return launch_expand<int32_t>(tok, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_I64: return launch_expand<int64_t>(tok, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_U8:  return launch_expand<uint8_t>(tok, path, B, Tx, Ty, 1, s);
        case ALIGNER_DT_F16: return launch_expand<uint16_t>(tok, path, B, Tx, Ty, 0x3C00, s);
        case ALIGNER_DT_BF16: return launch_expand<uint16_t>(tok, path, B, Tx, Ty, 0x3F80, s);
        default: return fail(ALIGNER_EINVAL, "path dtype %d not supported", path_dtype);
    }
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
    const WsLayout L = ws_layout(B, Tx, Ty);
    const int *tok = tok_out ? tok_out
                             : reinterpret_cast<const int *>(static_cast<unsigned char *>(ws) + L.tok_off);
    return aligner_maxpath_expand(tok, path_out, path_dtype, B, Tx, Ty, stream);
}

int aligner_maxpath_read_status(const void *ws, int32_t *status_host, void *stream) {
    if (!ws || !status_host) return fail(ALIGNER_EINVAL, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ALIGNER_HIP_CHECK(hipMemcpyAsync(status_host, ws, sizeof(int32_t), hipMemcpyDeviceToHost, s));
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
