// MoBoAligner monotonic boundary search on MI355X (gfx950): BASELINE config 5 / SURVEY.md section 8 rows a7, f3.
//
// Build-defined spec (the reference snapshot only names the branch and links the paper, README.md:9-13,49;
// restated in oracle/mobo_oracle.py, parity UNPINNED): tokens i, frames y, energies e[i,y]; a segmentation is a
// boundary sequence 0 = b_-1 < b_0 < ... < b_{I-1} = J with durations 1..D (the maximum-duration window);
//     P(b_i = j | b_{i-1} = k) = exp(e[i,j-1]) / sum_{m in A_i(k)} exp(e[i,m-1]),
//     A_i(k) = (k, k+D] intersected with [lo_i, hi_i]   (the later tokens still fit: see the oracle).
// Outputs: log_alpha[i,j-1] = log P(b_i = j) (sum-product), the MAP boundary sequence (max-product, ties: the
// shortest token) with its log-probability, and -- a row-parallel kernel -- the soft alignment
// gamma[i,y] = P(b_{i-1} <= y < b_i).
//
// Shape of the computation (round 3; round 2 ran everything on one CU per utterance, 5.7 ms at [8,500,4000]):
//     L_i(k)      = logsumexp_{m in A_i(k)} e_i(m)                       (normaliser of the step out of k)
//     la_i(j)     = e_i(j) + logsumexp_{k in [j-D, j)} (la_{i-1}(k) - L_i(k))
//     delta_i(j)  = e_i(j) +    max    _{k in [j-D, j)} (delta_{i-1}(k) - L_i(k))   (+ argmax)
//  1. mobo_norm_kernel: L depends on the energies only, so it is computed for every (utterance, token, position)
//     at once on the whole chip, before the chain starts.
//  2. mobo_chain_kernel: token rows are a dependent chain, positions inside a row are independent, and a row
//     only looks BACK (at most D positions).  So the positions of an utterance are cut into S segments, one
//     workgroup (its own CU) each; segment s needs from segment s-1 only the last D entries of its row, which
//     travel through a ring in the workspace whose words are their own flags (filled with 0xFFFFFFFF before the
//     launch; never a value).  Data flows one way, so segment s simply runs a little behind segment s-1 and
//     nobody waits for a consumer.  Rows in which a segment has no reachable position cost it a few stores.
//  3. mobo_backtrack_kernel: the MAP sequence from the per-(token, position) durations, a batch of rows at a time
//     (a step's position lies within t*D of the batch's start, so the batch's window is fetched at once).
// Windows are summed directly, one position per lane, over an exact representation of every term:
// u = M + log2(s) with an INTEGER M = ceil(u) and s = 2^(u-M) in (0.5, 1]; a window's sum is
// sum_k ldexp(s_k, M_k - Mw) with Mw the largest M in the window -- every scaling is an exact power of two, the
// reference is the window's own maximum, so a window keeps full relative precision however far below the row's
// bulk it lies (the reason round 2 carried fp64 block sums).  All logs are base 2 inside (v_exp_f32 /
// v_log_f32 are exp2 / log2), "log 0" is the finite -1e30.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

constexpr float MB_NEG = -1e30f;                  // log 0
constexpr float MB_DEADF = -1e7f;                 // anything below is "log 0" (|live values| stay far below 2^24)
constexpr int MB_DEADM = -(1 << 24);              // integer part of a dead term
constexpr unsigned MB_FILL = 0xFFFFFFFFu;         // ring filler (a NaN pattern no value ever has)
constexpr float MB_LOG2E = 1.4426950408889634f, MB_LN2 = 0.6931471805599453f;
constexpr int MB_SPIN_LIMIT = 1 << 21;            // polls of one ring row before a segment gives up (~ seconds)

struct MoboParams {
    const void *e;            // [B,Tx,Ty] fp32 / bf16 / fp16
    const int *t_xs, *t_ys;   // [B]
    float *log_alpha;         // nullable [B,Tx,Ty]
    int *boundaries;          // [B,Tx]
    int *durations;           // nullable [B,Tx]
    float *map_score;         // nullable [B]
    float *Lw;                // workspace [B,Tx,Ty]: L_i(k), base 2, k = 0..Ty-1
    unsigned short *back;     // workspace [B,Tx,Ty+1]: duration of token i when it ends at j
    unsigned *ring;           // workspace [B,S-1,Tx,3,D]: (M, s, v) of a segment's last D positions, row by row
    int *failw;               // workspace [B]: 1 = a segment of this utterance gave up waiting
    unsigned *trash;          // workspace [B*S,1024] (+ slack): where lanes with nothing to store store
    int *status;              // workspace: ALIGNER_ST_* bits
    int B, Tx, Ty, D, S, nmax;
    int bstride;              // entries per row of `back`: a multiple of 8 >= Ty + 1 (16-byte pieces in the backtrack)
    int start_lag;            // rows a segment lets the one before it get ahead before it starts (see the chain kernel)
    unsigned long long *stamps;  // development (aligner_debug_set_stamps): per block 16 words of phase cycle totals
    int drop_seg, spin_limit;  // testing (aligner_debug_set_option "mobo_drop_segment"): that segment publishes nothing
};

template <int VT> __device__ __forceinline__ unsigned mb_load_raw(const void *base, size_t idx) {
    if (VT == 0) return static_cast<const unsigned *>(base)[idx];
    return static_cast<const unsigned short *>(base)[idx];
}
template <int VT> __device__ __forceinline__ float mb_value(unsigned raw) {
    if (VT == 0) return __builtin_bit_cast(float, raw);
    if (VT == 1) return __builtin_bit_cast(float, raw << 16);
    return (float)__builtin_bit_cast(_Float16, (unsigned short)raw);
}

// u = M + log2(s), M integer, s in (0.5, 1]; dead: (MB_DEADM, 0)
__device__ __forceinline__ void mb_encode(float u, int &M, float &s) {
    if (u > MB_DEADF) {
        const float c = ceilf(u);
        M = (int)c;
        s = __builtin_amdgcn_exp2f(u - c);
    } else {
        M = MB_DEADM;
        s = 0.f;
    }
}

// One window of D consecutive entries: Mw = max M, acc = sum ldexp(s, M - Mw) (so the sum is 2^Mw * acc, acc in
// [0.5, D] unless every entry is dead), and -- WITH_V -- the largest v with the LARGEST index among equals.
template <bool WITH_V>
__device__ __forceinline__ void mb_window(const int *__restrict__ sM, const float *__restrict__ sS,
                                          const float *__restrict__ sV, int D, int &Mw, float &acc, float &best,
                                          int &qbest) {
    Mw = MB_DEADM;
    acc = 0.f;
    best = MB_NEG;
    qbest = 0;
    int c = 0;
    const int r = D & 7;
    for (; c < r; ++c) {                            // the D % 8 lowest entries one by one, then groups of eight
        const int m = sM[c];
        const int Mn = Mw > m ? Mw : m;
        acc = __builtin_ldexpf(acc, Mw - Mn) + __builtin_ldexpf(sS[c], m - Mn);
        Mw = Mn;
        if (WITH_V) {
            const float v = sV[c];
            if (v >= best) {
                best = v;
                qbest = c;
            }
        }
    }
    int cbest = -1;                                 // the LAST group whose maximum is the running maximum
    for (; c < D; c += 8) {
        int m[8];
        float s[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            m[u] = sM[c + u];
            s[u] = sS[c + u];
            if (WITH_V) v[u] = sV[c + u];
        }
        int cm = m[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) cm = cm > m[u] ? cm : m[u];
        const int Mn = Mw > cm ? Mw : cm;
        acc = __builtin_ldexpf(acc, Mw - Mn);
        Mw = Mn;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += __builtin_ldexpf(s[u], m[u] - Mw);
        if (WITH_V) {                               // (v is never NaN: phase 1 replaces anything not above MB_DEADF)
            float vm = v[0];
#pragma unroll
            for (int u = 1; u < 8; ++u) vm = __builtin_fmaxf(vm, v[u]);
            if (vm >= best) {
                best = vm;
                cbest = c;
            }
        }
    }
    if (WITH_V && cbest >= 0) {                     // ties: the largest index
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (sV[cbest + u] == best) qbest = cbest + u;
    }
}

__device__ __forceinline__ void mb_bounds(int I, int J, int D, int i, int &lo, int &hi) {
    hi = J - (I - 1 - i);
    const long long lo64 = (long long)J - (long long)(I - 1 - i) * D;
    lo = (lo64 > i + 1) ? (int)lo64 : i + 1;
}

// Rows [i0, i1] in which positions [a, bnd) of an utterance hold a reachable boundary (i1 < 0: none).  Both limits of a
// row's band move up with i, so it is an interval, in closed form: a loop over the token rows here was 40 us of
// every chain launch at 500 tokens (every workgroup ran it before its first row).
__device__ __forceinline__ void mb_active_rows(int I, int J, int D, int a, int bnd, int &i0, int &i1) {
    // min(hi_i, (i+1) D) >= a  <=>  i >= a - J + I - 1  and  i >= ceil(a / D) - 1
    int lo_i = a - J + I - 1;
    const int byreach = (a + D - 1) / D - 1;
    lo_i = lo_i > byreach ? lo_i : byreach;
    lo_i = lo_i > 0 ? lo_i : 0;
    // max(i + 1, J - (I-1-i) D) < bnd  <=>  i <= bnd - 2  and  (bnd > J  or  i <= I - 2 - floor((J - bnd) / D))
    int hi_i = I - 1 < bnd - 2 ? I - 1 : bnd - 2;
    if (bnd <= J) {
        const int bylo = I - 2 - (J - bnd) / D;
        hi_i = hi_i < bylo ? hi_i : bylo;
    }
    if (lo_i > hi_i) {
        i0 = I;
        i1 = -1;
    } else {
        i0 = lo_i;
        i1 = hi_i;
    }
}

// Barrier for LDS hand-offs only: __syncthreads() also drains vmcnt, which would put the latency of the next
// row's loads and of this row's result stores on every row.
__device__ __forceinline__ void mb_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- cross-lane helpers of the split form -------------------------------------------------------------------
__device__ __forceinline__ int mb_quad_xor_i(int v, int m) {       // the value of lane (lane ^ m), m = 1 or 2
    return m == 1 ? __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
}
__device__ __forceinline__ float mb_quad_xor_f(float v, int m) {
    return __builtin_bit_cast(float, mb_quad_xor_i(__builtin_bit_cast(int, v), m));
}
__device__ __forceinline__ int mb_wave_max_i32(int v) {            // maximum over the wave's 64 lanes (uniform result)
#define MB_DPP_MAX(ctrl_, rows_)                                                          \
    {                                                                                     \
        const int o_ = __builtin_amdgcn_update_dpp(v, v, ctrl_, rows_, 0xF, false);       \
        v = v > o_ ? v : o_;                                                              \
    }
    MB_DPP_MAX(0x111, 0xF)      // row_shr:1
    MB_DPP_MAX(0x112, 0xF)      // row_shr:2
    MB_DPP_MAX(0x114, 0xF)      // row_shr:4
    MB_DPP_MAX(0x118, 0xF)      // row_shr:8   -> lane 15 of every row holds its row's maximum
    MB_DPP_MAX(0x142, 0xA)      // row_bcast:15 into rows 1 and 3
    MB_DPP_MAX(0x143, 0xC)      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's
#undef MB_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------------------------------------------------
// 1. L_i(k) for every (utterance, token, position); also refills the ring and clears the give-up words.
//    grid (ceil(Ty/256), Tx, B) x 256 threads, one position k per thread, window (k, k+D] read from LDS.
// ---------------------------------------------------------------------------------------------------------
constexpr int MB_NCH = 1024;                      // positions per workgroup of the normaliser kernel (4 per thread)

template <int VT>
__global__ __launch_bounds__(256) void mobo_norm_kernel(MoboParams p, unsigned long long ring_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.z, i = blockIdx.y, k0 = blockIdx.x * MB_NCH;
    {   // housekeeping spread over the whole grid
        const unsigned long long nthr = (unsigned long long)gridDim.x * gridDim.y * gridDim.z * 256ull;
        const unsigned long long gid = (((unsigned long long)b * gridDim.y + i) * gridDim.x + blockIdx.x) * 256ull + tid;
        for (unsigned long long w = gid; w < ring_words; w += nthr) p.ring[w] = MB_FILL;
        if (gid < (unsigned long long)p.B) p.failw[gid] = 0;
    }
    const int D = p.D;
    int I = p.t_xs[b], J = p.t_ys[b];
    I = I > p.Tx ? p.Tx : I;
    J = J > p.Ty ? p.Ty : J;
    const bool ok = I >= 1 && J >= I && (long long)J <= (long long)I * D;
    const size_t rowoff = ((size_t)b * p.Tx + i) * p.Ty;
    if (!ok || i >= I) {                              // nothing to search here: only the padding of log_alpha
        if (p.log_alpha)
            for (int k = k0 + tid; k < k0 + MB_NCH && k < p.Ty; k += 256) p.log_alpha[rowoff + k] = -__builtin_huge_valf();
        return;
    }
    int lo, hi;
    mb_bounds(I, J, D, i, lo, hi);
    // positions the previous row can occupy (the only k whose normaliser is ever used)
    int klo = 0, khi = 0;
    if (i > 0) {
        mb_bounds(I, J, D, i - 1, klo, khi);
        const long long reach = (long long)i * D;
        if (khi > reach) khi = (int)reach;
    }
    if (k0 > khi || k0 + MB_NCH - 1 < klo) {          // the whole block is outside the band
        for (int k = k0 + tid; k < k0 + MB_NCH && k < p.Ty; k += 256) {
            p.Lw[rowoff + k] = MB_NEG;
            if (p.log_alpha && k >= J) p.log_alpha[rowoff + k] = -__builtin_huge_valf();
        }
        return;
    }
    const int NE = MB_NCH + D;                        // staged entries
    const int NEp = (NE + 3) & ~3;                    // (each array 16-byte aligned: the fast sums read float4s)
    int *sM = reinterpret_cast<int *>(smem);
    float *sS = reinterpret_cast<float *>(sM + NEp);
    float *sT = sS + NEp;
    __shared__ int s_blockmax;
    if (tid == 0) s_blockmax = MB_DEADM;
    // entries x = 0 .. NE-1 stand for boundary positions m = k0+1+x, i.e. frames m-1
    int Mmax = MB_DEADM;
    {   // the first 1 280 entries (everything when D <= 256): all of a thread's loads in flight before the first is used
        // -- one load per loop iteration was a chain of HBM round trips, and the kernel's whole time
        unsigned raw[5];
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            int f = k0 + tid + 256 * it;                  // frame m - 1 of entry x = tid + 256 it
            f = f < p.Ty ? f : p.Ty - 1;
            raw[it] = mb_load_raw<VT>(p.e, rowoff + f);
        }
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int x = tid + 256 * it, m = k0 + 1 + x;
            if (x < NE) {
                const float e2 = (m >= lo && m <= hi) ? mb_value<VT>(raw[it]) * MB_LOG2E : MB_NEG;
                int M;
                float s;
                mb_encode(e2, M, s);
                sM[x] = M;
                sS[x] = s;
                Mmax = Mmax > M ? Mmax : M;
            }
        }
    }
    for (int x = tid + 1280; x < NE; x += 256) {
        const int m = k0 + 1 + x;
        float e2 = MB_NEG;
        if (m >= lo && m <= hi) e2 = mb_value<VT>(mb_load_raw<VT>(p.e, rowoff + (m - 1))) * MB_LOG2E;
        int M;
        float s;
        mb_encode(e2, M, s);
        sM[x] = M;
        sS[x] = s;
        Mmax = Mmax > M ? Mmax : M;
    }
    __syncthreads();
    {   // the block's largest M (one LDS atomic per wave)
        const int wm = mb_wave_max_i32(Mmax);
        if ((tid & 63) == 0) atomicMax(&s_blockmax, wm);
    }
    __syncthreads();
    // FAST sums: every term as a plain float against the block's largest M -- exact while every live entry is within
    // 100 of it (the energies of a thousand neighbouring frames of one token: always, on real log-likelihoods); a
    // block with an entry further down takes the exact (M, s) sums
    const int Rb = s_blockmax;
    bool fits = true;
    for (int x = tid; x < NE; x += 256) {
        const int M = sM[x];
        sT[x] = __builtin_ldexpf(sS[x], M - Rb);
        if (M != MB_DEADM && M < Rb - 100) fits = false;
    }
    const int slow = __syncthreads_or(!fits);
    // Four ADJACENT positions per thread: their windows [x, x+D) share all but three entries at either end, so the D + 3
    // entries are read once (aligned 16-byte LDS reads) and the common part is summed once -- a quarter of the LDS
    // traffic and of the additions of four separate windows (the kernel was bound by both: 60 us at [8,500,4000]).
    const int xb = 4 * tid;
    float Ls[4] = {MB_NEG, MB_NEG, MB_NEG, MB_NEG};
    if (k0 + xb < p.Ty) {
        if (!slow && D >= 8) {
            const float *t = sT + xb;
            const float4 h = *reinterpret_cast<const float4 *>(t);              // entries 0..3
            float core = h.w;                                                     // entries 3 .. D-1: in every one of the four
            int c = 4;
            for (; c + 8 <= D; c += 8) {
                const float4 a = *reinterpret_cast<const float4 *>(t + c), bq = *reinterpret_cast<const float4 *>(t + c + 4);
                core += ((a.x + a.y) + (a.z + a.w)) + ((bq.x + bq.y) + (bq.z + bq.w));
            }
            for (; c < D; ++c) core += t[c];
            const float e0 = t[D], e1 = t[D + 1], e2 = t[D + 2];
            const float acc[4] = {core + ((h.x + h.y) + h.z), core + ((h.y + h.z) + e0), core + ((h.z + e0) + e1),
                                  core + ((e0 + e1) + e2)};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (acc[u] > 0.f) Ls[u] = (float)Rb + __builtin_amdgcn_logf(acc[u]);
        } else if (!slow) {
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                float acc = 0.f;
                for (int c = 0; c < D; ++c) acc += sT[xb + u + c];
                if (acc > 0.f) Ls[u] = (float)Rb + __builtin_amdgcn_logf(acc);
            }
        } else {
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                int Mw, qb;
                float acc, best;
                mb_window<false>(sM + xb + u, sS + xb + u, nullptr, D, Mw, acc, best, qb);
                if (acc > 0.f) Ls[u] = (float)Mw + __builtin_amdgcn_logf(acc);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + xb + u;
            if (k < p.Ty) {
                p.Lw[rowoff + k] = (k < J && k >= klo && k <= khi) ? Ls[u] : MB_NEG;
                if (p.log_alpha && k >= J) p.log_alpha[rowoff + k] = -__builtin_huge_valf();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 2. The chain.  Block b*S + s = segment s of utterance b: positions [a, a+n) of its J+1 boundary positions.
//    LDS: two buffers (row parity) of (M, s, v) for the segment's positions, preceded by the D halo entries
//    that belong to the segment before -- one barrier per row.  NP positions per thread (j = a + tid + q*T).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mb_ring_store(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned mb_ring_load(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what the two forms of the chain kernel share: one position of phase 1 / phase 2
struct MbSeg {
    int a, bnd, D, J, W;
    bool has_next;
    int *sM;
    float *sS, *sV;
    unsigned *ring_out;
};
__device__ __forceinline__ void mb_phase1(const MbSeg &g, int i, int bo, int k, float la, float de, float Lraw) {
    const float L = (k < g.J) ? Lraw : MB_NEG;     // the step out of J does not exist
    const bool live = L > MB_DEADF;
    const float u = (live && la > MB_DEADF) ? la - L : MB_NEG;
    float v = (live && de > MB_DEADF) ? de - L : MB_NEG;
    v = (v > MB_DEADF) ? v : MB_NEG;
    int M;
    float s;
    mb_encode(u, M, s);
    const int x = bo + g.D + (k - g.a);
    g.sM[x] = M;
    g.sS[x] = s;
    g.sV[x] = v;
    if (g.has_next && k >= g.bnd - g.D) {
        unsigned *r = g.ring_out + (size_t)i * 3 * g.D + (k - (g.bnd - g.D));
        mb_ring_store(r, __builtin_bit_cast(unsigned, (float)M));
        mb_ring_store(r + g.D, __builtin_bit_cast(unsigned, s));
        mb_ring_store(r + 2 * g.D, __builtin_bit_cast(unsigned, v));
    }
}
__device__ __forceinline__ void mb_phase2(const MbSeg &g, int bo, int j, int lo, int hi, float ev, float &lav, float &dev,
                                          int &dur) {
    lav = MB_NEG;
    dev = MB_NEG;
    dur = 0;
    if (j >= lo && j <= hi) {
        int Mw, qb;
        float acc, best;
        const int x = bo + (j - g.a);
        mb_window<true>(g.sM + x, g.sS + x, g.sV + x, g.D, Mw, acc, best, qb);
        if (acc > 0.f && ev > MB_DEADF) lav = ev + ((float)Mw + __builtin_amdgcn_logf(acc));
        if (best > MB_DEADF && ev > MB_DEADF) {
            dev = ev + best;
            dur = g.D - qb;
        }
        lav = (lav > MB_DEADF) ? lav : MB_NEG;
        dev = (dev > MB_DEADF) ? dev : MB_NEG;
        if (!(dev > MB_DEADF)) dur = 0;
    }
}

// MULTI = false: at most one position per thread -- the form of a split utterance (a segment per CU): the state
//   of a position lives in registers and the next row's operands (energy, normaliser, halo) are in flight while
//   this row is computed.  MULTI = true: any number of positions per thread, state in LDS, operands loaded where
//   they are used -- the form of a large batch, where a CU's many waves hide the latency.
template <int VT, bool MULTI>
__global__ __launch_bounds__(1024) void mobo_chain_kernel(MoboParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int b = blockIdx.x / p.S, sg = blockIdx.x - b * p.S;
    const int D = p.D;
    int I = p.t_xs[b], J = p.t_ys[b];
    I = I > p.Tx ? p.Tx : I;
    J = J > p.Ty ? p.Ty : J;
    const bool ok = I >= 1 && J >= I && (long long)J <= (long long)I * D;
    if (!ok) {                                     // infeasible: no segmentation exists (outputs: the other kernels)
        if (sg == 0 && tid == 0) atomicOr(p.status, ALIGNER_ST_BAD_LENGTHS);
        return;
    }
    // this utterance's segments: every one but the last holds n >= D positions, so a halo has one source
    const int P = J + 1;
    int Sb = P / D;
    Sb = Sb < 1 ? 1 : (Sb > p.S ? p.S : Sb);
    const int n = (P + Sb - 1) / Sb;
    const int a = sg * n;
    if (sg >= Sb || a >= P) return;
    MbSeg g;
    g.a = a;
    g.bnd = (a + n < P) ? a + n : P;
    g.D = D;
    g.J = J;
    g.W = p.nmax + D;                              // entries per buffer
    g.has_next = (sg + 1 < Sb) && (a + n < P) && sg != p.drop_seg;
    g.sM = reinterpret_cast<int *>(smem);
    g.sS = reinterpret_cast<float *>(g.sM + 2 * g.W);
    g.sV = g.sS + 2 * g.W;
    float *sLa = g.sV + 2 * g.W, *sDe = sLa + p.nmax;     // MULTI only
    const int bnd = g.bnd, W = g.W;
    for (int h = tid; h < D; h += T) {             // positions before the utterance's start (segment 0 keeps these)
        g.sM[h] = MB_DEADM;  g.sM[W + h] = MB_DEADM;
        g.sS[h] = 0.f;       g.sS[W + h] = 0.f;
        g.sV[h] = MB_NEG;    g.sV[W + h] = MB_NEG;
    }
    const int j1 = a + tid;                        // !MULTI: this thread's position
    float la = (j1 == 0) ? 0.f : MB_NEG, de = la;  // P(b_-1 = 0) = 1
    if (MULTI)
        for (int j = j1; j < bnd; j += T) {
            sLa[j - a] = (j == 0) ? 0.f : MB_NEG;
            sDe[j - a] = (j == 0) ? 0.f : MB_NEG;
        }
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    unsigned short *backb = p.back + (size_t)b * p.Tx * p.bstride;
    g.ring_out = p.ring + ((size_t)b * (p.S - 1) + (g.has_next ? sg : 0)) * (size_t)p.Tx * 3 * D;
    const unsigned *ring_in = p.ring + ((size_t)b * (p.S - 1) + (sg > 0 ? sg - 1 : 0)) * (size_t)p.Tx * 3 * D;
    const bool polls = sg > 0 && tid < D;

    unsigned e_nx = 0, h_nx[3] = {0, 0, 0};
    float L_nx = MB_NEG;
    auto issue = [&](int i) {                      // row i's operands, in flight while row i-1 is computed
        const size_t ro = ubase + (size_t)i * p.Ty;
        if (!MULTI) {
            const int j = j1 < 1 ? 1 : (j1 > J ? J : j1);
            const int k = j1 > J - 1 ? J - 1 : j1;
            e_nx = mb_load_raw<VT>(p.e, ro + (j - 1));
            L_nx = p.Lw[ro + k];
        }
        if (polls) {
            const unsigned *r = ring_in + (size_t)i * 3 * D + tid;
            h_nx[0] = mb_ring_load(r);
            h_nx[1] = mb_ring_load(r + D);
            h_nx[2] = mb_ring_load(r + 2 * D);
        }
    };
    issue(0);
    bool gave_up = false;
    bool prev_active = (sg == 0);                 // row -1: position 0 holds P(b_-1 = 0) = 1
    for (int i = 0; i < I; ++i) {
        int lo, hi;
        mb_bounds(I, J, D, i, lo, hi);
        const long long reach = (long long)(i + 1) * D;
        const int hi2 = hi < reach ? hi : (int)reach;
        const bool active = lo < bnd && hi2 >= a && lo <= hi2;      // some position of the segment is reachable
        const unsigned e_c = e_nx, h_c0 = h_nx[0], h_c1 = h_nx[1], h_c2 = h_nx[2];
        const float L_c = L_nx;
        if (i + 1 < I) issue(i + 1);
        const size_t ro = ubase + (size_t)i * p.Ty;
        const int bo = (i & 1) * W;
        // phase 1 concerns the PREVIOUS row's states: the segment's last D entries are owed to the next segment as
        // long as row i-1 had a reachable position here, whether or not row i has one
        if (active || prev_active) {
            // ---- phase 1: u = la_{i-1}(k) - L_i(k), v = delta_{i-1}(k) - L_i(k) for the segment's own positions ----
            if (!MULTI) {
                if (j1 < bnd) mb_phase1(g, i, bo, j1, la, de, L_c);
            } else {
#pragma unroll 1
                for (int k = j1; k < bnd; k += T)
                    mb_phase1(g, i, bo, k, sLa[k - a], sDe[k - a], p.Lw[ro + (k > J - 1 ? J - 1 : k)]);
            }
        } else if (g.has_next) {
            for (int h = tid; h < D; h += T) {
                unsigned *r = g.ring_out + (size_t)i * 3 * D + h;
                mb_ring_store(r, __builtin_bit_cast(unsigned, (float)MB_DEADM));
                mb_ring_store(r + D, 0u);
                mb_ring_store(r + 2 * D, __builtin_bit_cast(unsigned, MB_NEG));
            }
        }
        if (active) {
            // ---- the D entries before the segment: the previous segment's row i (its words are their own flags) ----
            if (sg > 0) {
#pragma unroll 1
                for (int h = tid; h < D; h += T) {
                    const unsigned *r = ring_in + (size_t)i * 3 * D + h;
                    unsigned w0, w1, w2;
                    if (h == tid) { w0 = h_c0; w1 = h_c1; w2 = h_c2; }
                    else { w0 = mb_ring_load(r); w1 = mb_ring_load(r + D); w2 = mb_ring_load(r + 2 * D); }
                    int spins = 0;
                    while ((w0 == MB_FILL || w1 == MB_FILL || w2 == MB_FILL) && !gave_up) {
                        __builtin_amdgcn_s_sleep(4);
                        w0 = mb_ring_load(r);
                        w1 = mb_ring_load(r + D);
                        w2 = mb_ring_load(r + 2 * D);
                        if (++spins > p.spin_limit) gave_up = true;
                    }
                    const bool bad = (w0 == MB_FILL || w1 == MB_FILL || w2 == MB_FILL);
                    g.sM[bo + h] = bad ? MB_DEADM : (int)__builtin_bit_cast(float, w0);
                    g.sS[bo + h] = bad ? 0.f : __builtin_bit_cast(float, w1);
                    g.sV[bo + h] = bad ? MB_NEG : __builtin_bit_cast(float, w2);
                }
            }
            mb_lds_barrier();
            // ---- phase 2: the windows [j-D, j) ----
#pragma unroll 1
            for (int j = j1; j < bnd; j += T) {
                float ev;
                if (!MULTI) ev = mb_value<VT>(e_c) * MB_LOG2E;
                else ev = (j >= lo && j <= hi) ? mb_value<VT>(mb_load_raw<VT>(p.e, ro + (j - 1))) * MB_LOG2E : MB_NEG;
                float lav, dev;
                int dur;
                mb_phase2(g, bo, j, lo, hi, ev, lav, dev, dur);
                if (!MULTI) { la = lav; de = dev; }
                else { sLa[j - a] = lav; sDe[j - a] = dev; }
                backb[(size_t)i * p.bstride + j] = (unsigned short)dur;
                if (p.log_alpha && j >= 1)
                    p.log_alpha[ro + (j - 1)] = (lav > MB_DEADF) ? lav * MB_LN2 : -__builtin_huge_valf();
                if (i == I - 1 && j == J && p.map_score)
                    p.map_score[b] = (dev > MB_DEADF) ? dev * MB_LN2 : -__builtin_huge_valf();
                if (!MULTI) break;
            }
        } else {
            // ---- no reachable position in this row: everything is "log 0" ----
            la = MB_NEG;
            de = MB_NEG;
#pragma unroll 1
            for (int j = j1; j < bnd; j += T) {
                if (MULTI) { sLa[j - a] = MB_NEG; sDe[j - a] = MB_NEG; }
                if (p.log_alpha && j >= 1) p.log_alpha[ro + (j - 1)] = -__builtin_huge_valf();
                if (i == I - 1 && j == J && p.map_score) p.map_score[b] = -__builtin_huge_valf();
            }
        }
        prev_active = active;
    }
    if (gave_up) {                                 // the segment before never delivered: say so, loudly
        atomicOr(p.status, ALIGNER_ST_INTERNAL);
        p.failw[b] = 1;
    }
}

// One part of a window for the FAST sum: plain fp32 terms t = 2^(u - R) against the row's common reference R (see
// the kernel), and the same maximum search as mb_window (the position of a group's maximum is found in the registers).
__device__ __forceinline__ void mb_window_fast(const float *__restrict__ sT, const float *__restrict__ sV, int cnt,
                                               float &acc, float &best, int &qbest) {
    acc = 0.f;
    best = MB_NEG;
    qbest = 0;
    int c = 0;
    const int r = cnt & 7;
    for (; c < r; ++c) {
        acc += sT[c];
        const float v = sV[c];
        if (v >= best) {
            best = v;
            qbest = c;
        }
    }
    for (; c < cnt; c += 8) {
        float t[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            t[u] = sT[c + u];
            v[u] = sV[c + u];
        }
        acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        float vm = v[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) vm = __builtin_fmaxf(vm, v[u]);
        int qi = 0;                                     // the last entry of the group that equals its maximum
#pragma unroll
        for (int u = 1; u < 8; ++u) qi = (v[u] == vm) ? u : qi;
        if (vm >= best) {
            best = vm;
            qbest = c + qi;
        }
    }
}

// A wave's report about a row -- (a live exponent or MB_DEADM, "an entry does not fit") -- without atomics: every lane of the
// wave stores the same pair into the wave's own slot (one ds_write_b64, no branch), and behind the row's barrier lane l of
// every wave reads slot l & 15 and two ballots combine them.  Two slot sets by row parity; slots of waves that do not
// exist or do not report stay neutral.
__device__ __forceinline__ void mb_report_put(int *sRep, int par, int wave, int wm, int nofit) {
    int2 v;
    v.x = wm;
    v.y = nofit;
    *reinterpret_cast<int2 *>(sRep + (par * 16 + wave) * 2) = v;
}
__device__ __forceinline__ void mb_report_get(const int *sRep, int par, int lane, int &Rn, int &slow) {
    const int2 v = *reinterpret_cast<const int2 *>(sRep + (par * 16 + (lane & 15)) * 2);
    const unsigned long long live = __builtin_amdgcn_ballot_w64(v.x != MB_DEADM);
    slow = __builtin_amdgcn_ballot_w64(v.y != 0) != 0;
    Rn = live ? __builtin_amdgcn_readlane(v.x, __builtin_ctzll(live)) : MB_DEADM;
}

// The split form: one position per H lanes, everything a row needs in flight two rows ahead.
//
// What paces a row here is first of all `s_waitcnt`: loads, stores and atomics retire through ONE counter in issue
// order, and hipcc can only emit a counted wait (leave the N youngest operations in flight) when it can prove N
// operations follow -- anything behind a branch counts as "maybe" and turns the wait into vmcnt(0), which then also
// waits for the row's result stores to be acknowledged (measured: 4 500 cycles a row instead of 3 300).  So the rows
// in which the segment has a reachable position run as one straight-line body: every lane issues every load and
// store of the row, lanes with nothing to say address a trash area of the workspace, and the first use of the
// prefetched operands waits behind exactly the stores issued after them.  Rows before / after that range only hand
// "log 0" entries on.
//
// Then the windows.  H lanes share a position's window (H = 1, 2, 4: a split utterance leaves SIMDs idle otherwise)
// and meet through two DPP quad exchanges.  And a row normally takes the FAST sum: next to the exact (M, s) pair every
// entry also leaves t = 2^(u - R) as a plain float, R an integer reference common to the whole workgroup (the
// largest M of the row before); while every live M of the row -- halo included -- lies within 100 of R, all terms
// and every window's own largest term are normal floats of full precision, and a window is D additions.  Each wave
// reports whether its entries fit (and their maximum, the next row's R); a row with an entry out of range takes
// the exact sum, which needs no reference.
template <int VT, bool WANT_LA, int H, bool STAMPS>
__global__ __launch_bounds__(1024) void mobo_chain_one_kernel(MoboParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // a launch that splits utterances (S > 1) carries one more wave: the HELPER takes the halo off the others' critical
    // path -- it fetches the previous segment's row, waits for it if it must, and puts it into LDS (550-850 of a row's
    // 3 300 cycles when wave 0 did it beside its own positions)
    const int tid = threadIdx.x;
    const bool has_helper = p.S > 1;
    const int T = (int)blockDim.x - (has_helper ? 64 : 0);
    const bool helper = tid >= T;
    const bool stamping = STAMPS && p.stamps != nullptr && tid == 0;      // (a build of its own: the stamps cost registers)
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0, st_entry = 0, st_rt = 0, st_polls = 0, st_slow = 0;
    if (stamping) { st_entry = __builtin_amdgcn_s_memtime(); st_rt = __builtin_amdgcn_s_memrealtime(); }
#define MB_STAMP(k_) if (stamping) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k_] += t_ - st_t; st_t = t_; }
    const int b = blockIdx.x / p.S, sg = blockIdx.x - b * p.S;
    const int D = p.D;
    int I = p.t_xs[b], J = p.t_ys[b];
    I = I > p.Tx ? p.Tx : I;
    J = J > p.Ty ? p.Ty : J;
    const bool ok = I >= 1 && J >= I && (long long)J <= (long long)I * D;
    if (!ok) {
        if (sg == 0 && tid == 0) atomicOr(p.status, ALIGNER_ST_BAD_LENGTHS);
        return;
    }
    const int P = J + 1;
    int Sb = P / D;
    Sb = Sb < 1 ? 1 : (Sb > p.S ? p.S : Sb);
    const int n = (P + Sb - 1) / Sb;
    const int a = sg * n;
    if (sg >= Sb || a >= P) return;
    const int bnd = (a + n < P) ? a + n : P;
    const bool has_next = (sg + 1 < Sb) && (a + n < P) && sg != p.drop_seg;
    const int W = p.nmax + D;                      // entries per buffer
    int *sM = reinterpret_cast<int *>(smem);
    float *sS = reinterpret_cast<float *>(sM + 2 * W);
    float *sV = sS + 2 * W;
    float *sT = sV + 2 * W;
    int *sRep = reinterpret_cast<int *>(sT + 2 * W);    // [row parity][16 waves][2]: the waves' reports (mb_report_put)
    if (tid < 32) { sRep[2 * tid] = MB_DEADM; sRep[2 * tid + 1] = 0; }
    for (int h = tid; h < D; h += T) {             // positions before the utterance's start (segment 0 keeps these)
        sM[h] = MB_DEADM;  sM[W + h] = MB_DEADM;
        sS[h] = 0.f;       sS[W + h] = 0.f;
        sV[h] = MB_NEG;    sV[W + h] = MB_NEG;
        sT[h] = 0.f;       sT[W + h] = 0.f;
    }
    mb_lds_barrier();
    const int pi = tid / H, sub = tid - pi * H;    // H consecutive lanes per position
    const int j1 = a + pi;
    const bool mine = !helper && j1 < bnd;
    const bool lead = mine && sub == 0;            // the lane of a position that writes its results
    float la = (j1 == 0) ? 0.f : MB_NEG, de = la;  // P(b_-1 = 0) = 1
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    unsigned short *backb = p.back + (size_t)b * p.Tx * p.bstride;
    unsigned *trash = p.trash + (size_t)blockIdx.x * 1024 + tid;
    unsigned *ring_out = p.ring + ((size_t)b * (p.S - 1) + (has_next ? sg : 0)) * (size_t)p.Tx * 3 * D;
    const unsigned *ring_in = p.ring + ((size_t)b * (p.S - 1) + (sg > 0 ? sg - 1 : 0)) * (size_t)p.Tx * 3 * D;
    const int hl = tid - T;                        // helper: its lane
    const bool publishes = has_next && lead && j1 >= bnd - D;
    // this lane's part of a window: entries [w0, w1) of its D
    const int w0 = (sub * D) / H, w1 = ((sub + 1) * D) / H;

    // rows [i0, i1]: the segment has a reachable position
    int i0, i1;
    mb_active_rows(I, J, D, a, bnd, i0, i1);
    auto dead_rows = [&](int from, int to) {      // rows without a reachable position: "log 0" everywhere
        for (int i = from; i < to; ++i) {
            if (has_next && !helper)
                for (int h = tid; h < D; h += T) {
                    unsigned *r = ring_out + (size_t)i * 3 * D + h;
                    mb_ring_store(r, __builtin_bit_cast(unsigned, (float)MB_DEADM));
                    mb_ring_store(r + D, 0u);
                    mb_ring_store(r + 2 * D, __builtin_bit_cast(unsigned, MB_NEG));
                }
            if (WANT_LA && lead && j1 >= 1) p.log_alpha[ubase + (size_t)i * p.Ty + (j1 - 1)] = -__builtin_huge_valf();
        }
    };
    if (i1 < 0) {                                  // never reachable (an utterance much shorter than the batch's Ty)
        dead_rows(0, I);
        if (lead && j1 == J && p.map_score) p.map_score[b] = -__builtin_huge_valf();
        return;
    }
    dead_rows(0, i0);
    bool gave_up = false;
    int R = 0;                                     // the row's reference for the fast sum (uniform: every wave tracks it)
    const int wave = tid >> 6, lane = tid & 63;

    if (helper) {
        // ------------------------------ the helper wave: the halo of rows i0..i1 ------------------------------
        if (sg > 0) {
            if (p.start_lag > 0) {
                // Let the segment before get `start_lag` rows ahead first.  A row's halo is fetched a row early; that
                // only finds it if the producer is more than a row + the visibility latency ahead -- two segments in
                // step pay a round trip to memory per row (measured: 1 200-3 000 cycles of every row).
                int ig = i0 + p.start_lag;
                ig = ig > I - 1 ? I - 1 : ig;
                const unsigned *r = ring_in + (size_t)ig * 3 * D + (hl < D ? hl : 0) + 2 * D;    // the row's last word
                int spins = 0;
                while (mb_ring_load(r) == MB_FILL && !gave_up) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > p.spin_limit) gave_up = true;
                }
            }
            const bool fetches = hl < D;
            unsigned h0_nx, h1_nx, h2_nx;
            auto issue_halo = [&](int i) {
                const unsigned *r = fetches ? ring_in + (size_t)i * 3 * D + hl : trash;
                const int st = fetches ? D : 0;
                h0_nx = mb_ring_load(r);
                h1_nx = mb_ring_load(r + st);
                h2_nx = mb_ring_load(r + 2 * st);
            };
            issue_halo(i0);
#pragma unroll 1
            for (int i = i0; i <= i1; ++i) {
                const unsigned h_c0 = h0_nx, h_c1 = h1_nx, h_c2 = h2_nx;
                issue_halo(i + 1 < I ? i + 1 : I - 1);
                const int bo = (i & 1) * W;
                // this lane's own entry comes from the words fetched a row ago: its test must not share a join with a
                // path that loads (hipcc would then wait vmcnt(0) here -- for the loads issued a moment ago)
                int Mstat = MB_DEADM;
                bool fits = true;
                auto halo_entry = [&](int h, unsigned x0, unsigned x1, unsigned x2) {
                    if (x0 == MB_FILL || x1 == MB_FILL || x2 == MB_FILL) {          // not published yet: poll
                        const unsigned *r = ring_in + (size_t)i * 3 * D + h;
                        int spins = 0;
                        do {
                            __builtin_amdgcn_s_sleep(2);
                            x0 = mb_ring_load(r);
                            x1 = mb_ring_load(r + D);
                            x2 = mb_ring_load(r + 2 * D);
                            if (++spins > p.spin_limit) gave_up = true;
                        } while ((x0 == MB_FILL || x1 == MB_FILL || x2 == MB_FILL) && !gave_up);
                    }
                    const bool bad = (x0 == MB_FILL || x1 == MB_FILL || x2 == MB_FILL);
                    const int M = bad ? MB_DEADM : (int)__builtin_bit_cast(float, x0);
                    const float sv = bad ? 0.f : __builtin_bit_cast(float, x1);
                    sM[bo + h] = M;
                    sS[bo + h] = sv;
                    sV[bo + h] = bad ? MB_NEG : __builtin_bit_cast(float, x2);
                    sT[bo + h] = __builtin_ldexpf(sv, M - R);
                    if (M != MB_DEADM) {
                        Mstat = Mstat > M ? Mstat : M;
                        fits = fits && M >= R - 100 && M <= R + 100;
                    }
                };
                if (fetches) halo_entry(hl, h_c0, h_c1, h_c2);
                if (D > 64) {
#pragma unroll 1
                    for (int h = hl + 64; h < D; h += 64) halo_entry(h, MB_FILL, MB_FILL, MB_FILL);
                }
                {   // the wave's report (mb_report_put: no atomics -- a same-address atomic per lane, or even per wave,
                    // serialises in the LDS pipe and the barrier's lgkmcnt(0) waits for it)
                    const int wm = mb_wave_max_i32(Mstat);
                    const bool wfit = __builtin_amdgcn_ballot_w64(!fits) == 0;
                    mb_report_put(sRep, i & 1, wave, wm, wfit ? 0 : 1);
                }
                mb_lds_barrier();
                int Rn, slow_;
                mb_report_get(sRep, i & 1, lane, Rn, slow_);
                R = (Rn != MB_DEADM) ? Rn : R;
            }
        }                                          // (segment 0 has no halo: the wave ends, a barrier only counts live waves)
        if (gave_up) {
            atomicOr(p.status, ALIGNER_ST_INTERNAL);
            p.failw[b] = 1;
        }
        return;
    }

    // ------------------------------ the compute waves ------------------------------
    // operands of a row, all lanes, clamped addresses; pointers advance by a row
    const int je = (j1 < 1 ? 1 : (j1 > J ? J : j1)) - 1;               // frame of this lane's boundary position
    const int kl = j1 > J - 1 ? J - 1 : j1;                            // the step out of J does not exist (phase 1)
    // The operands of a row are requested TWO rows ahead: they come from HBM (every row a new cache line per plane) and a
    // row is about one HBM round trip -- with one row of lead some wave of the workgroup found its operands late in most
    // rows and the others waited for it at the barrier.  Two register sets take turns (the loop is unrolled by two):
    // shifting a queue with moves would wait for the younger set's loads, a move reads its source.
    unsigned e_n1, e_n2;
    float L_n1, L_n2;
    auto issue = [&](int i, unsigned &e_o, float &L_o) {
        const size_t ro = ubase + (size_t)i * p.Ty;
        e_o = mb_load_raw<VT>(p.e, ro + je);
        L_o = p.Lw[ro + kl];
    };
    // without a helper wave (an unsplit launch) nobody else fetches a halo -- and there is none: S == 1
    // The loop is entered in the state every later row finds: two rows of operand loads, each followed by a row's
    // stores (three ring words, one duration, log_alpha).  hipcc sizes a counted wait for the FEWEST operations that can
    // follow on any path into it: without these the waits inside the loop also waited for the previous row's stores to be
    // acknowledged -- 1 200 cycles of every row.
    issue(i0, e_n1, L_n1);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    if (WANT_LA) mb_ring_store(trash, 0u);
    issue(i0 + 1 < I ? i0 + 1 : I - 1, e_n2, L_n2);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    if (WANT_LA) mb_ring_store(trash, 0u);
    unsigned long long st_loop = 0;
    if (stamping) { st_loop = st_t = __builtin_amdgcn_s_memtime(); }
    auto do_row = [&](const int i, unsigned &e_s, float &L_s) {
        int lo, hi;
        mb_bounds(I, J, D, i, lo, hi);
        const unsigned e_c = e_s;
        const float L_c = L_s;
        issue(i + 2 < I ? i + 2 : I - 1, e_s, L_s);                   // (the set just read takes the row after next)
        const size_t ro = ubase + (size_t)i * p.Ty;
        const int bo = (i & 1) * W;
        if (stamping) { asm volatile("" :: "v"(e_c), "v"(L_c)); }
        MB_STAMP(0)
        // ---- phase 1: u = la_{i-1}(k) - L_i(k), v = delta_{i-1}(k) - L_i(k) ----
        {
            const float L = (j1 < J) ? L_c : MB_NEG;
            const bool live = L > MB_DEADF;
            const float u = (live && la > MB_DEADF) ? la - L : MB_NEG;
            float v = (live && de > MB_DEADF) ? de - L : MB_NEG;
            v = (v > MB_DEADF) ? v : MB_NEG;
            int M;
            float s;
            mb_encode(u, M, s);
            if (lead) {
                const int x = bo + D + pi;
                sM[x] = M;
                sS[x] = s;
                sV[x] = v;
                sT[x] = __builtin_ldexpf(s, M - R);
            }
            {   // the wave's report about the row -- one LDS atomic each: "an entry does not fit", and a candidate for the
                // next row's reference: the M of ANY live lane will do (while the row fits, every live M is within 100 of
                // R and so of one another; a wave-wide maximum costs six DPP steps a row)
                const bool livem = lead && M != MB_DEADM;
                const unsigned long long lm = __builtin_amdgcn_ballot_w64(livem);
                const bool wfit = __builtin_amdgcn_ballot_w64(livem && !(M >= R - 100 && M <= R + 100)) == 0;
                const int wm = lm ? __builtin_amdgcn_readlane(M, __builtin_ctzll(lm)) : MB_DEADM;
                mb_report_put(sRep, i & 1, wave, wm, wfit ? 0 : 1);
            }
            unsigned *r = publishes ? ring_out + (size_t)i * 3 * D + (j1 - (bnd - D)) : trash;
            const int st = publishes ? D : 0;
            mb_ring_store(r, __builtin_bit_cast(unsigned, (float)M));
            mb_ring_store(r + st, __builtin_bit_cast(unsigned, s));
            mb_ring_store(r + 2 * st, __builtin_bit_cast(unsigned, v));
        }
        MB_STAMP(1)
        MB_STAMP(2)
        mb_lds_barrier();
        MB_STAMP(3)
        // ---- phase 2: the windows [j-D, j) ----
        {
            int Rn, slow;
            mb_report_get(sRep, i & 1, lane, Rn, slow);
            const float ev = mb_value<VT>(e_c) * MB_LOG2E;
            const bool feasible = mine && j1 >= lo && j1 <= hi;
            const int x = bo + (mine ? pi : 0) + w0;           // first entry of this lane's part
            float lsum = MB_NEG, best;
            int qb;
            if (!slow) {
                float acc;
                mb_window_fast(sT + x, sV + x, w1 - w0, acc, best, qb);
#pragma unroll
                for (int m = 1; m < H; m <<= 1) acc += mb_quad_xor_f(acc, m);
                if (acc > 0.f) lsum = (float)R + __builtin_amdgcn_logf(acc);
            } else {
                if (stamping) ++st_slow;
                int Mw;
                float acc;
                mb_window<true>(sM + x, sS + x, sV + x, w1 - w0, Mw, acc, best, qb);
#pragma unroll
                for (int m = 1; m < H; m <<= 1) {
                    const int Mo = mb_quad_xor_i(Mw, m);
                    const float ao = mb_quad_xor_f(acc, m);
                    const int Mn = Mw > Mo ? Mw : Mo;
                    acc = __builtin_ldexpf(acc, Mw - Mn) + __builtin_ldexpf(ao, Mo - Mn);
                    Mw = Mn;
                }
                if (acc > 0.f) lsum = (float)Mw + __builtin_amdgcn_logf(acc);
            }
            qb += w0;
#pragma unroll
            for (int m = 1; m < H; m <<= 1) {                  // the largest v; among equals the largest index
                const float bo_ = mb_quad_xor_f(best, m);
                const int qo = mb_quad_xor_i(qb, m);
                const bool take = bo_ > best || (bo_ == best && qo > qb);
                best = take ? bo_ : best;
                qb = take ? qo : qb;
            }
            float lav = MB_NEG, dev = MB_NEG;
            int dur = 0;
            if (feasible && ev > MB_DEADF) {
                if (lsum > MB_DEADF) lav = ev + lsum;
                if (best > MB_DEADF) { dev = ev + best; dur = D - qb; }
                lav = (lav > MB_DEADF) ? lav : MB_NEG;
                dev = (dev > MB_DEADF) ? dev : MB_NEG;
                if (!(dev > MB_DEADF)) dur = 0;
            }
            la = lav;
            de = dev;
            R = (Rn != MB_DEADM) ? Rn : R;
            if (stamping) { asm volatile("" :: "v"(la), "v"(de)); }
            MB_STAMP(4)
            unsigned short *bp = lead ? backb + (size_t)i * p.bstride + j1 : reinterpret_cast<unsigned short *>(trash);
            *bp = (unsigned short)dur;
            if (WANT_LA) {
                float *lp = (lead && j1 >= 1) ? p.log_alpha + ro + (j1 - 1) : reinterpret_cast<float *>(trash);
                *lp = (lav > MB_DEADF) ? lav * MB_LN2 : -__builtin_huge_valf();
            }
        }
        MB_STAMP(5)
    };
#pragma unroll 1
    for (int i = i0; i <= i1; i += 2) {
        do_row(i, e_n1, L_n1);
        if (i + 1 > i1) break;
        do_row(i + 1, e_n2, L_n2);
    }
    if (stamping) {
        unsigned long long *o = p.stamps + (size_t)blockIdx.x * 24;
        o[0] = st_entry; o[1] = st_loop; o[2] = __builtin_amdgcn_s_memtime();
        for (int q = 0; q < 6; ++q) o[3 + q] = st_acc[q];
        o[9] = (unsigned long long)(i1 - i0 + 1); o[10] = (unsigned long long)i0; o[11] = st_rt;
        o[12] = __builtin_amdgcn_s_memrealtime(); o[13] = st_polls; o[14] = st_slow;
        o[15] = st_acc[6]; o[16] = st_acc[7]; o[17] = st_acc[8];
    }
#undef MB_STAMP
    if (lead && j1 == J && i1 == I - 1 && p.map_score)
        p.map_score[b] = (de > MB_DEADF) ? de * MB_LN2 : -__builtin_huge_valf();
    if (i1 + 1 < I) {
        // row i1+1: nothing reachable here any more, but the states of row i1 are still owed to the next segment
        const int i = i1 + 1;
        if (publishes) {
            const float Lr = p.Lw[ubase + (size_t)i * p.Ty + kl];
            const float L = (j1 < J) ? Lr : MB_NEG;
            const bool live = L > MB_DEADF;
            const float u = (live && la > MB_DEADF) ? la - L : MB_NEG;
            float v = (live && de > MB_DEADF) ? de - L : MB_NEG;
            v = (v > MB_DEADF) ? v : MB_NEG;
            int M;
            float s;
            mb_encode(u, M, s);
            unsigned *r = ring_out + (size_t)i * 3 * D + (j1 - (bnd - D));
            mb_ring_store(r, __builtin_bit_cast(unsigned, (float)M));
            mb_ring_store(r + D, __builtin_bit_cast(unsigned, s));
            mb_ring_store(r + 2 * D, __builtin_bit_cast(unsigned, v));
        }
        if (WANT_LA && lead && j1 >= 1) p.log_alpha[ubase + (size_t)i * p.Ty + (j1 - 1)] = -__builtin_huge_valf();
        dead_rows(i1 + 2, I);
        if (lead && j1 == J && p.map_score) p.map_score[b] = -__builtin_huge_valf();
    }
}

// The split form when only the MAP sequence is asked for (no log_alpha): the max-product chain alone.  It never reads the
// sum-product one, so everything that serves the sums goes -- the (M, s) encoding, the fast terms and their reference,
// the waves' reports, the exact sums, two of the three ring words, the window's additions and its logarithm: a row is a
// subtraction, one LDS write, one ring word, a barrier, the window's maximum with its position, an addition.  Same
// devices otherwise (helper wave, operands two rows ahead, counted waits, trash-address stores), same results bit for
// bit (`mobo_full_chain` selects the full kernel for the comparison: tests/test_mobo.py).
template <int VT, int H>
__global__ __launch_bounds__(1024) void mobo_chain_map_kernel(MoboParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const bool has_helper = p.S > 1;
    const int T = (int)blockDim.x - (has_helper ? 64 : 0);
    const bool helper = tid >= T;
    const int b = blockIdx.x / p.S, sg = blockIdx.x - b * p.S;
    const int D = p.D;
    int I = p.t_xs[b], J = p.t_ys[b];
    I = I > p.Tx ? p.Tx : I;
    J = J > p.Ty ? p.Ty : J;
    const bool ok = I >= 1 && J >= I && (long long)J <= (long long)I * D;
    if (!ok) {
        if (sg == 0 && tid == 0) atomicOr(p.status, ALIGNER_ST_BAD_LENGTHS);
        return;
    }
    const int P = J + 1;
    int Sb = P / D;
    Sb = Sb < 1 ? 1 : (Sb > p.S ? p.S : Sb);
    const int n = (P + Sb - 1) / Sb;
    const int a = sg * n;
    if (sg >= Sb || a >= P) return;
    const int bnd = (a + n < P) ? a + n : P;
    const bool has_next = (sg + 1 < Sb) && (a + n < P) && sg != p.drop_seg;
    const int W = p.nmax + D;
    float *sV = reinterpret_cast<float *>(smem);          // [2][W]: D halo entries, then the segment's positions
    float *sSpare = sV + 2 * W;                           // where lanes without a position write (never read)
    for (int h = tid; h < D; h += T) {                    // positions before the utterance's start (segment 0 keeps these)
        sV[h] = MB_NEG;
        sV[W + h] = MB_NEG;
    }
    mb_lds_barrier();
    const int pi = tid / H, sub = tid - pi * H;
    const int j1 = a + pi;
    const bool mine = !helper && j1 < bnd;
    const bool lead = mine && sub == 0;
    float de = (j1 == 0) ? 0.f : MB_NEG;                  // P(b_-1 = 0) = 1
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    unsigned short *backb = p.back + (size_t)b * p.Tx * p.bstride;
    unsigned *trash = p.trash + (size_t)blockIdx.x * 1024 + tid;
    // (the ring keeps the full kernel's layout [row][3][D]; only its third plane, v, is written and awaited)
    unsigned *ring_out = p.ring + ((size_t)b * (p.S - 1) + (has_next ? sg : 0)) * (size_t)p.Tx * 3 * D + 2 * D;
    const unsigned *ring_in = p.ring + ((size_t)b * (p.S - 1) + (sg > 0 ? sg - 1 : 0)) * (size_t)p.Tx * 3 * D + 2 * D;
    const int hl = tid - T;
    const bool publishes = has_next && lead && j1 >= bnd - D;
    const int w0 = (sub * D) / H, w1 = ((sub + 1) * D) / H;
    int i0, i1;
    mb_active_rows(I, J, D, a, bnd, i0, i1);
    auto dead_rows = [&](int from, int to) {
        if (has_next && !helper)
            for (int i = from; i < to; ++i)
                for (int h = tid; h < D; h += T) mb_ring_store(ring_out + (size_t)i * 3 * D + h, __builtin_bit_cast(unsigned, MB_NEG));
    };
    if (i1 < 0) {
        dead_rows(0, I);
        if (lead && j1 == J && p.map_score) p.map_score[b] = -__builtin_huge_valf();
        return;
    }
    dead_rows(0, i0);
    bool gave_up = false;

    if (helper) {
        if (sg > 0) {
            if (p.start_lag > 0) {
                int ig = i0 + p.start_lag;
                ig = ig > I - 1 ? I - 1 : ig;
                const unsigned *r = ring_in + (size_t)ig * 3 * D + (hl < D ? hl : 0);
                int spins = 0;
                while (mb_ring_load(r) == MB_FILL && !gave_up) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > p.spin_limit) gave_up = true;
                }
            }
            const bool fetches = hl < D;
            unsigned h_nx;
            auto issue_halo = [&](int i) { h_nx = mb_ring_load(fetches ? ring_in + (size_t)i * 3 * D + hl : trash); };
            issue_halo(i0);
#pragma unroll 1
            for (int i = i0; i <= i1; ++i) {
                const unsigned h_c = h_nx;
                issue_halo(i + 1 < I ? i + 1 : I - 1);
                const int bo = (i & 1) * W;
                auto halo_entry = [&](int h, unsigned x) {
                    if (x == MB_FILL) {                                            // not published yet: poll
                        const unsigned *r = ring_in + (size_t)i * 3 * D + h;
                        int spins = 0;
                        do {
                            __builtin_amdgcn_s_sleep(2);
                            x = mb_ring_load(r);
                            if (++spins > p.spin_limit) gave_up = true;
                        } while (x == MB_FILL && !gave_up);
                    }
                    sV[bo + h] = (x == MB_FILL) ? MB_NEG : __builtin_bit_cast(float, x);
                };
                if (fetches) halo_entry(hl, h_c);
                if (D > 64) {
#pragma unroll 1
                    for (int h = hl + 64; h < D; h += 64) halo_entry(h, MB_FILL);
                }
                mb_lds_barrier();
            }
        }
        if (gave_up) {
            atomicOr(p.status, ALIGNER_ST_INTERNAL);
            p.failw[b] = 1;
        }
        return;
    }

    const int je = (j1 < 1 ? 1 : (j1 > J ? J : j1)) - 1;
    const int kl = j1 > J - 1 ? J - 1 : j1;
    unsigned e_n1, e_n2;
    float L_n1, L_n2;
    auto issue = [&](int i, unsigned &e_o, float &L_o) {
        const size_t ro = ubase + (size_t)i * p.Ty;
        e_o = mb_load_raw<VT>(p.e, ro + je);
        L_o = p.Lw[ro + kl];
    };
    // (entered in the state every later row finds: two rows of operand loads, each with a row's two stores behind it)
    issue(i0, e_n1, L_n1);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    issue(i0 + 1 < I ? i0 + 1 : I - 1, e_n2, L_n2);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    auto do_row = [&](const int i, unsigned &e_s, float &L_s) {
        int lo, hi;
        mb_bounds(I, J, D, i, lo, hi);
        const unsigned e_c = e_s;
        const float L_c = L_s;
        issue(i + 2 < I ? i + 2 : I - 1, e_s, L_s);
        const int bo = (i & 1) * W;
        {   // phase 1: v = delta_{i-1}(k) - L_i(k)
            const float L = (j1 < J) ? L_c : MB_NEG;
            float v = (L > MB_DEADF && de > MB_DEADF) ? de - L : MB_NEG;
            v = (v > MB_DEADF) ? v : MB_NEG;
            *(mine ? sV + bo + D + pi : sSpare) = v;
            mb_ring_store(publishes ? ring_out + (size_t)i * 3 * D + (j1 - (bnd - D)) : trash, __builtin_bit_cast(unsigned, v));
        }
        mb_lds_barrier();
        {   // phase 2: the window [j-D, j): its largest v, among equals the largest index
            const float ev = mb_value<VT>(e_c) * MB_LOG2E;
            const bool feasible = mine && j1 >= lo && j1 <= hi;
            const float *vv = sV + bo + (mine ? pi : 0) + w0;
            const int cnt = w1 - w0;
            float best = MB_NEG;
            int qb = 0, c = 0;
            for (const int r8 = cnt & 7; c < r8; ++c) {
                const float v = vv[c];
                if (v >= best) { best = v; qb = c; }
            }
            // (shallow trees: a lone wave per SIMD pays dependent latency -- the chunk's maximum by v_max3, then independent
            // "equals the maximum ? index : -1" and an integer v_max3 tree for the LAST entry that has it; chains of
            // 15 maxima and 15 selects measured 5 % slower at 16 entries a lane)
#define MB_MAX3F(a_, b_, c_) __builtin_fmaxf(__builtin_fmaxf(a_, b_), c_)
#define MB_MAX3I(a_, b_, c_) ((a_) > (b_) ? ((a_) > (c_) ? (a_) : (c_)) : ((b_) > (c_) ? (b_) : (c_)))
            if (((cnt - c) & 15) == 0) {
#pragma unroll 1
                for (; c < cnt; c += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = vv[c + u];
                    const float m0 = MB_MAX3F(v[0], v[1], v[2]), m1 = MB_MAX3F(v[3], v[4], v[5]), m2 = MB_MAX3F(v[6], v[7], v[8]);
                    const float m3 = MB_MAX3F(v[9], v[10], v[11]), m4 = MB_MAX3F(v[12], v[13], v[14]);
                    const float vm = __builtin_fmaxf(MB_MAX3F(m0, m1, m2), MB_MAX3F(m3, m4, v[15]));
                    int q[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) q[u] = (v[u] == vm) ? u : -1;
                    const int q0 = MB_MAX3I(q[0], q[1], q[2]), q1 = MB_MAX3I(q[3], q[4], q[5]), q2 = MB_MAX3I(q[6], q[7], q[8]);
                    const int q3 = MB_MAX3I(q[9], q[10], q[11]), q4 = MB_MAX3I(q[12], q[13], q[14]);
                    const int qa = MB_MAX3I(q0, q1, q2), qc = MB_MAX3I(q3, q4, q[15]);
                    const int qi = qa > qc ? qa : qc;
                    const bool take = vm >= best;                               // (ties: the later chunk)
                    best = take ? vm : best;
                    qb = take ? c + qi : qb;
                }
            } else {
#pragma unroll 1
                for (; c < cnt; c += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = vv[c + u];
                    float vm = v[0];                                            // (at eight entries the plain chains are the faster form)
#pragma unroll
                    for (int u = 1; u < 8; ++u) vm = __builtin_fmaxf(vm, v[u]);
                    int qi = 0;
#pragma unroll
                    for (int u = 1; u < 8; ++u) qi = (v[u] == vm) ? u : qi;
                    if (vm >= best) { best = vm; qb = c + qi; }
                }
            }
#undef MB_MAX3F
#undef MB_MAX3I
            qb += w0;
#pragma unroll
            for (int m = 1; m < H; m <<= 1) {
                const float bo_ = mb_quad_xor_f(best, m);
                const int qo = mb_quad_xor_i(qb, m);
                const bool take = bo_ > best || (bo_ == best && qo > qb);
                best = take ? bo_ : best;
                qb = take ? qo : qb;
            }
            float dev = MB_NEG;
            int dur = 0;
            if (feasible && ev > MB_DEADF && best > MB_DEADF) {
                dev = ev + best;
                dur = D - qb;
                if (!(dev > MB_DEADF)) { dev = MB_NEG; dur = 0; }
            }
            de = dev;
            unsigned short *bp = lead ? backb + (size_t)i * p.bstride + j1 : reinterpret_cast<unsigned short *>(trash);
            *bp = (unsigned short)dur;
        }
    };
#pragma unroll 1
    for (int i = i0; i <= i1; i += 2) {
        do_row(i, e_n1, L_n1);
        if (i + 1 > i1) break;
        do_row(i + 1, e_n2, L_n2);
    }
    if (lead && j1 == J && i1 == I - 1 && p.map_score)
        p.map_score[b] = (de > MB_DEADF) ? de * MB_LN2 : -__builtin_huge_valf();
    if (i1 + 1 < I) {
        // row i1+1: nothing reachable here any more, but the states of row i1 are still owed to the next segment
        const int i = i1 + 1;
        if (publishes) {
            const float Lr = p.Lw[ubase + (size_t)i * p.Ty + kl];
            const float L = (j1 < J) ? Lr : MB_NEG;
            float v = (L > MB_DEADF && de > MB_DEADF) ? de - L : MB_NEG;
            v = (v > MB_DEADF) ? v : MB_NEG;
            mb_ring_store(ring_out + (size_t)i * 3 * D + (j1 - (bnd - D)), __builtin_bit_cast(unsigned, v));
        }
        dead_rows(i1 + 2, I);
        if (lead && j1 == J && p.map_score) p.map_score[b] = -__builtin_huge_valf();
    }
}

// ... and the split form when only log_alpha (and gamma) is asked for -- a training step's forward pass: the sum-product
// chain alone.  No v, no maximum and position, no durations plane, one ring word less; the window is its fast terms'
// sum (or the exact (M, s) sum in a row that does not fit).  Bit for bit the full kernel's log_alpha.
template <int VT, int H>
__global__ __launch_bounds__(1024) void mobo_chain_sum_kernel(MoboParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const bool has_helper = p.S > 1;
    const int T = (int)blockDim.x - (has_helper ? 64 : 0);
    const bool helper = tid >= T;
    const int b = blockIdx.x / p.S, sg = blockIdx.x - b * p.S;
    const int D = p.D;
    int I = p.t_xs[b], J = p.t_ys[b];
    I = I > p.Tx ? p.Tx : I;
    J = J > p.Ty ? p.Ty : J;
    const bool ok = I >= 1 && J >= I && (long long)J <= (long long)I * D;
    if (!ok) {
        if (sg == 0 && tid == 0) atomicOr(p.status, ALIGNER_ST_BAD_LENGTHS);
        return;
    }
    const int P = J + 1;
    int Sb = P / D;
    Sb = Sb < 1 ? 1 : (Sb > p.S ? p.S : Sb);
    const int n = (P + Sb - 1) / Sb;
    const int a = sg * n;
    if (sg >= Sb || a >= P) return;
    const int bnd = (a + n < P) ? a + n : P;
    const bool has_next = (sg + 1 < Sb) && (a + n < P) && sg != p.drop_seg;
    const int W = p.nmax + D;
    int *sM = reinterpret_cast<int *>(smem);
    float *sS = reinterpret_cast<float *>(sM + 2 * W);
    float *sT = sS + 2 * W;
    int *sRep = reinterpret_cast<int *>(sT + 2 * W);      // (the full kernel's fourth array is free here: room for both)
    float *sSpare = reinterpret_cast<float *>(sRep + 64);
    if (tid < 32) { sRep[2 * tid] = MB_DEADM; sRep[2 * tid + 1] = 0; }
    for (int h = tid; h < D; h += T) {
        sM[h] = MB_DEADM;  sM[W + h] = MB_DEADM;
        sS[h] = 0.f;       sS[W + h] = 0.f;
        sT[h] = 0.f;       sT[W + h] = 0.f;
    }
    mb_lds_barrier();
    const int pi = tid / H, sub = tid - pi * H;
    const int j1 = a + pi;
    const bool mine = !helper && j1 < bnd;
    const bool lead = mine && sub == 0;
    float la = (j1 == 0) ? 0.f : MB_NEG;
    const size_t ubase = (size_t)b * p.Tx * p.Ty;
    unsigned *trash = p.trash + (size_t)blockIdx.x * 1024 + tid;
    unsigned *ring_out = p.ring + ((size_t)b * (p.S - 1) + (has_next ? sg : 0)) * (size_t)p.Tx * 3 * D;
    const unsigned *ring_in = p.ring + ((size_t)b * (p.S - 1) + (sg > 0 ? sg - 1 : 0)) * (size_t)p.Tx * 3 * D;
    const int hl = tid - T;
    const bool publishes = has_next && lead && j1 >= bnd - D;
    const int w0 = (sub * D) / H, w1 = ((sub + 1) * D) / H;
    const int wave = tid >> 6, lane = tid & 63;
    int i0, i1;
    mb_active_rows(I, J, D, a, bnd, i0, i1);
    auto dead_rows = [&](int from, int to) {
        for (int i = from; i < to; ++i) {
            if (has_next && !helper)
                for (int h = tid; h < D; h += T) {
                    unsigned *r = ring_out + (size_t)i * 3 * D + h;
                    mb_ring_store(r, __builtin_bit_cast(unsigned, (float)MB_DEADM));
                    mb_ring_store(r + D, 0u);
                }
            if (lead && j1 >= 1) p.log_alpha[ubase + (size_t)i * p.Ty + (j1 - 1)] = -__builtin_huge_valf();
        }
    };
    if (i1 < 0) {
        dead_rows(0, I);
        return;
    }
    dead_rows(0, i0);
    bool gave_up = false;
    int R = 0;

    if (helper) {
        if (sg > 0) {
            if (p.start_lag > 0) {
                int ig = i0 + p.start_lag;
                ig = ig > I - 1 ? I - 1 : ig;
                const unsigned *r = ring_in + (size_t)ig * 3 * D + (hl < D ? hl : 0) + D;        // the row's last word
                int spins = 0;
                while (mb_ring_load(r) == MB_FILL && !gave_up) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > p.spin_limit) gave_up = true;
                }
            }
            const bool fetches = hl < D;
            unsigned h0_nx, h1_nx;
            auto issue_halo = [&](int i) {
                const unsigned *r = fetches ? ring_in + (size_t)i * 3 * D + hl : trash;
                h0_nx = mb_ring_load(r);
                h1_nx = mb_ring_load(r + (fetches ? D : 0));
            };
            issue_halo(i0);
#pragma unroll 1
            for (int i = i0; i <= i1; ++i) {
                const unsigned h_c0 = h0_nx, h_c1 = h1_nx;
                issue_halo(i + 1 < I ? i + 1 : I - 1);
                const int bo = (i & 1) * W;
                int Mstat = MB_DEADM;
                bool fits = true;
                auto halo_entry = [&](int h, unsigned x0, unsigned x1) {
                    if (x0 == MB_FILL || x1 == MB_FILL) {
                        const unsigned *r = ring_in + (size_t)i * 3 * D + h;
                        int spins = 0;
                        do {
                            __builtin_amdgcn_s_sleep(2);
                            x0 = mb_ring_load(r);
                            x1 = mb_ring_load(r + D);
                            if (++spins > p.spin_limit) gave_up = true;
                        } while ((x0 == MB_FILL || x1 == MB_FILL) && !gave_up);
                    }
                    const bool bad = (x0 == MB_FILL || x1 == MB_FILL);
                    const int M = bad ? MB_DEADM : (int)__builtin_bit_cast(float, x0);
                    const float sv = bad ? 0.f : __builtin_bit_cast(float, x1);
                    sM[bo + h] = M;
                    sS[bo + h] = sv;
                    sT[bo + h] = __builtin_ldexpf(sv, M - R);
                    if (M != MB_DEADM) {
                        Mstat = Mstat > M ? Mstat : M;
                        fits = fits && M >= R - 100 && M <= R + 100;
                    }
                };
                if (fetches) halo_entry(hl, h_c0, h_c1);
                if (D > 64) {
#pragma unroll 1
                    for (int h = hl + 64; h < D; h += 64) halo_entry(h, MB_FILL, MB_FILL);
                }
                {
                    const int wm = mb_wave_max_i32(Mstat);
                    const bool wfit = __builtin_amdgcn_ballot_w64(!fits) == 0;
                    mb_report_put(sRep, i & 1, wave, wm, wfit ? 0 : 1);
                }
                mb_lds_barrier();
                int Rn, slow_;
                mb_report_get(sRep, i & 1, lane, Rn, slow_);
                R = (Rn != MB_DEADM) ? Rn : R;
            }
        }
        if (gave_up) {
            atomicOr(p.status, ALIGNER_ST_INTERNAL);
            p.failw[b] = 1;
        }
        return;
    }

    const int je = (j1 < 1 ? 1 : (j1 > J ? J : j1)) - 1;
    const int kl = j1 > J - 1 ? J - 1 : j1;
    unsigned e_n1, e_n2;
    float L_n1, L_n2;
    auto issue = [&](int i, unsigned &e_o, float &L_o) {
        const size_t ro = ubase + (size_t)i * p.Ty;
        e_o = mb_load_raw<VT>(p.e, ro + je);
        L_o = p.Lw[ro + kl];
    };
    // (entered in the state every later row finds: two rows of operand loads, each with a row's three stores behind it)
    issue(i0, e_n1, L_n1);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    issue(i0 + 1 < I ? i0 + 1 : I - 1, e_n2, L_n2);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    mb_ring_store(trash, 0u);
    auto do_row = [&](const int i, unsigned &e_s, float &L_s) {
        int lo, hi;
        mb_bounds(I, J, D, i, lo, hi);
        const unsigned e_c = e_s;
        const float L_c = L_s;
        issue(i + 2 < I ? i + 2 : I - 1, e_s, L_s);
        const size_t ro = ubase + (size_t)i * p.Ty;
        const int bo = (i & 1) * W;
        {   // phase 1: u = la_{i-1}(k) - L_i(k)
            const float L = (j1 < J) ? L_c : MB_NEG;
            const float u = (L > MB_DEADF && la > MB_DEADF) ? la - L : MB_NEG;
            int M;
            float sm;
            mb_encode(u, M, sm);
            int *wm_ = mine ? sM + bo + D + pi : reinterpret_cast<int *>(sSpare);
            wm_[0] = M;
            reinterpret_cast<float *>(mine ? wm_ + 2 * W : wm_ + 1)[0] = sm;
            reinterpret_cast<float *>(mine ? wm_ + 4 * W : wm_ + 2)[0] = __builtin_ldexpf(sm, M - R);
            {
                const bool livem = lead && M != MB_DEADM;
                const unsigned long long lm = __builtin_amdgcn_ballot_w64(livem);
                const bool wfit = __builtin_amdgcn_ballot_w64(livem && !(M >= R - 100 && M <= R + 100)) == 0;
                const int wm = lm ? __builtin_amdgcn_readlane(M, __builtin_ctzll(lm)) : MB_DEADM;
                mb_report_put(sRep, i & 1, wave, wm, wfit ? 0 : 1);
            }
            unsigned *r = publishes ? ring_out + (size_t)i * 3 * D + (j1 - (bnd - D)) : trash;
            mb_ring_store(r, __builtin_bit_cast(unsigned, (float)M));
            mb_ring_store(r + (publishes ? D : 0), __builtin_bit_cast(unsigned, sm));
        }
        mb_lds_barrier();
        {   // phase 2: the window [j-D, j)
            int Rn, slow;
            mb_report_get(sRep, i & 1, lane, Rn, slow);
            const float ev = mb_value<VT>(e_c) * MB_LOG2E;
            const bool feasible = mine && j1 >= lo && j1 <= hi;
            const int x = bo + (mine ? pi : 0) + w0;
            const int cnt = w1 - w0;
            float lsum = MB_NEG;
            if (!slow) {
                const float *t = sT + x;
                float acc = 0.f;
                int c = 0;
                for (const int r8 = cnt & 7; c < r8; ++c) acc += t[c];
                for (; c < cnt; c += 8)
                    acc += ((t[c] + t[c + 1]) + (t[c + 2] + t[c + 3])) + ((t[c + 4] + t[c + 5]) + (t[c + 6] + t[c + 7]));
#pragma unroll
                for (int m = 1; m < H; m <<= 1) acc += mb_quad_xor_f(acc, m);
                if (acc > 0.f) lsum = (float)R + __builtin_amdgcn_logf(acc);
            } else {
                int Mw, qb;
                float acc, best;
                mb_window<false>(sM + x, sS + x, nullptr, cnt, Mw, acc, best, qb);
#pragma unroll
                for (int m = 1; m < H; m <<= 1) {
                    const int Mo = mb_quad_xor_i(Mw, m);
                    const float ao = mb_quad_xor_f(acc, m);
                    const int Mn = Mw > Mo ? Mw : Mo;
                    acc = __builtin_ldexpf(acc, Mw - Mn) + __builtin_ldexpf(ao, Mo - Mn);
                    Mw = Mn;
                }
                if (acc > 0.f) lsum = (float)Mw + __builtin_amdgcn_logf(acc);
            }
            float lav = MB_NEG;
            if (feasible && ev > MB_DEADF && lsum > MB_DEADF) lav = ev + lsum;
            lav = (lav > MB_DEADF) ? lav : MB_NEG;
            la = lav;
            R = (Rn != MB_DEADM) ? Rn : R;
            float *lp = (lead && j1 >= 1) ? p.log_alpha + ro + (j1 - 1) : reinterpret_cast<float *>(trash);
            *lp = (lav > MB_DEADF) ? lav * MB_LN2 : -__builtin_huge_valf();
        }
    };
#pragma unroll 1
    for (int i = i0; i <= i1; i += 2) {
        do_row(i, e_n1, L_n1);
        if (i + 1 > i1) break;
        do_row(i + 1, e_n2, L_n2);
    }
    if (i1 + 1 < I) {
        const int i = i1 + 1;
        if (publishes) {
            const float Lr = p.Lw[ubase + (size_t)i * p.Ty + kl];
            const float L = (j1 < J) ? Lr : MB_NEG;
            const float u = (L > MB_DEADF && la > MB_DEADF) ? la - L : MB_NEG;
            int M;
            float sm;
            mb_encode(u, M, sm);
            unsigned *r = ring_out + (size_t)i * 3 * D + (j1 - (bnd - D));
            mb_ring_store(r, __builtin_bit_cast(unsigned, (float)M));
            mb_ring_store(r + D, __builtin_bit_cast(unsigned, sm));
        }
        if (lead && j1 >= 1) p.log_alpha[ubase + (size_t)i * p.Ty + (j1 - 1)] = -__builtin_huge_valf();
        dead_rows(i1 + 2, I);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 3. Backtrack of the MAP sequence: RB rows at a time.  After t steps from position j the walk is within
//    [j - t*D, j - t], so the durations of the batch's rows over those windows are fetched in one go.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mobo_backtrack_kernel(MoboParams p, int RB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_j, s_bad;
    __shared__ int s_b[64], s_d[64];
    typedef unsigned piece_t __attribute__((ext_vector_type(4)));      // 8 durations
    piece_t *win = reinterpret_cast<piece_t *>(smem);
    const int tid = threadIdx.x, b = blockIdx.x, D = p.D;
    int I = p.t_xs[b], J = p.t_ys[b];
    I = I > p.Tx ? p.Tx : I;
    J = J > p.Ty ? p.Ty : J;
    const bool ok = I >= 1 && J >= I && (long long)J <= (long long)I * D;
    int *bo = p.boundaries + (size_t)b * p.Tx;
    int *du = p.durations ? p.durations + (size_t)b * p.Tx : nullptr;
    if (!ok || p.failw[b]) {
        for (int i = tid; i < p.Tx; i += 256) {
            bo[i] = 0;
            if (du) du[i] = 0;
        }
        if (p.map_score && tid == 0) p.map_score[b] = -__builtin_huge_valf();
        return;
    }
    for (int i = I + tid; i < p.Tx; i += 256) {
        bo[i] = J;
        if (du) du[i] = 0;
    }
    const unsigned short *backb = p.back + (size_t)b * p.Tx * p.bstride;
    // Row t of a batch (token i-t) is needed over positions [j - t*D, j - t]; it is fetched in aligned pieces of 8
    // entries, 256/RB threads per row and at most 16 pieces per thread, ALL in flight before the first is used.
    const int tpr = 256 / RB, t_of = tid / tpr, c_of = tid - t_of * tpr;
    const int wp = ((RB - 1) * D + 8) / 8 + 1;      // pieces kept per row
    if (tid == 0) { s_j = J; s_bad = 0; }
    __syncthreads();
    for (int i = I - 1; i >= 0; i -= RB) {
        const int nb = (i + 1 < RB) ? i + 1 : RB;
        const int j = s_j;
        {
            const int t = t_of;
            int first = j - t * D;                                    // first position of the row's window ...
            first = first < 0 ? 0 : first;
            const int p0 = first >> 3, p1 = (j - t) >> 3;             // ... and its pieces p0 .. p1 (none if j < t)
            const piece_t *src = reinterpret_cast<const piece_t *>(backb + (size_t)(i - t) * p.bstride);
            piece_t r[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pc = p0 + c_of + q * tpr;
                r[q] = (t < nb && j >= t && pc <= p1) ? __builtin_nontemporal_load(src + pc) : piece_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pc = c_of + q * tpr;
                if (pc < wp) win[t * wp + pc] = r[q];
            }
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned short *w16 = reinterpret_cast<const unsigned short *>(win);
            int jj = j;
            for (int t = 0; t < nb; ++t) {
                int first = j - t * D;
                first = first < 0 ? 0 : first;
                const int d = w16[(size_t)t * wp * 8 + (jj - (first & ~7))];
                s_b[t] = jj;
                s_d[t] = d;
                // the very first look-up finding no duration: P(b_{I-1} = J) = 0 -- masked (-inf) energies leave no
                // segmentation with positive probability.  Not an internal error: reported like an infeasible length
                if (d < 1 || d > D || d > jj) { s_bad = (i == I - 1 && t == 0 && d == 0) ? 2 : 1; jj = 0; break; }
                jj -= d;
            }
            s_j = jj;
        }
        __syncthreads();
        if (s_bad) break;
        if (tid < nb) {
            bo[i - tid] = s_b[tid];
            if (du) du[i - tid] = s_d[tid];
        }
        __syncthreads();
    }
    if (s_bad) {                                   // no (or no consistent) sequence: all-zero outputs, as for an infeasible length
        for (int i = tid; i < p.Tx; i += 256) {
            bo[i] = 0;
            if (du) du[i] = 0;
        }
        if (p.map_score && tid == 0) p.map_score[b] = -__builtin_huge_valf();
        if (tid == 0) atomicOr(p.status, s_bad == 2 ? ALIGNER_ST_BAD_LENGTHS : ALIGNER_ST_INTERNAL);
    } else if (tid == 0 && s_j != 0) {
        atomicOr(p.status, ALIGNER_ST_INTERNAL);
    }
}

// gamma[i,y] = P(b_{i-1} <= y) - P(b_i <= y) from the forward variables: one workgroup per (utterance, token),
// a block-wide prefix sum of alpha_i and of alpha_{i-1} over the frames.
__global__ __launch_bounds__(256) void mobo_gamma_kernel(const float *__restrict__ log_alpha, const int *__restrict__ t_xs,
                                                          const int *__restrict__ t_ys, float *__restrict__ gamma,
                                                          int Tx, int Ty) {
    __shared__ float wsum[2][4];
    __shared__ float carry[2];
    const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int I = t_xs[b], J = t_ys[b];
    I = I > Tx ? Tx : I;
    J = J > Ty ? Ty : J;
    float *g = gamma + ((size_t)b * Tx + i) * Ty;
    if (i >= I || I < 1 || J < I) {
        for (int y = tid; y < Ty; y += 256) g[y] = 0.f;
        return;
    }
    const float *cur = log_alpha + ((size_t)b * Tx + i) * Ty;
    const float *prv = cur - Ty;
    if (tid == 0) { carry[0] = 0.f; carry[1] = 0.f; }
    __syncthreads();
    // cdf(y) = sum_{j <= y} alpha(j) with alpha(j) stored at index j-1: exclusive prefix over the stored row
    for (int y0 = 0; y0 < Ty; y0 += 256) {
        const int y = y0 + tid;
        float a0 = (y < J) ? __expf(cur[y]) : 0.f;
        float a1 = (i > 0 && y < J) ? __expf(prv[y]) : 0.f;
        float s0 = a0, s1 = a1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float t0 = __shfl_up(s0, off), t1 = __shfl_up(s1, off);
            if (lane >= off) { s0 += t0; s1 += t1; }
        }
        if (lane == 63) { wsum[0][wave] = s0; wsum[1][wave] = s1; }
        __syncthreads();
        float base0 = carry[0], base1 = carry[1];
        for (int w = 0; w < wave; ++w) { base0 += wsum[0][w]; base1 += wsum[1][w]; }
        const float c0 = base0 + s0 - a0, c1 = base1 + s1 - a1;      // exclusive: P(b <= y)
        if (y < Ty) g[y] = (y < J) ? ((i > 0 ? c1 : 1.f) - c0) : 0.f;
        __syncthreads();
        if (tid == 255) { carry[0] = base0 + s0; carry[1] = base1 + s1; }
        __syncthreads();
    }
}

#include "mobo_bwd.inc"

// ---------------------------------------------------------------------------------------------------------
// launch plan: D, segments per utterance, positions per segment, threads, workspace layout
// ---------------------------------------------------------------------------------------------------------
struct MoboPlan {
    int D, S, nmax, NP, T, RB, bstride, H;
    size_t lds, bt_lds, ring_words;
    size_t status_off, fail_off, trash_off, L_off, back_off, ring_off, total;
};

static int mobo_plan(int B, int Tx, int Ty, int max_duration, MoboPlan &pl, bool quiet) {
    const int D = max_duration > Ty ? Ty : max_duration;
    const int P = Ty + 1;
    const int cu = device_cu_count();
    const int lds_limit = device_lds_limit();
    const int smax = P / D < 1 ? 1 : P / D;        // a segment (but the last) holds at least D positions
    int S = cu / (B < 1 ? 1 : B);
    S = S < 1 ? 1 : S;
    S = S > smax ? smax : S;
    const int swave = (P + 63) / 64;               // ... and a wave's worth of them
    S = S > swave ? swave : S;
    int nmax = 0;
    size_t lds = 0;
    for (;; ++S) {
        const int n1 = (P + S - 1) / S, n2 = (2 * D < P) ? 2 * D : P;
        nmax = n1 > n2 ? n1 : n2;
        lds = nmax > 1024 ? (size_t)2 * 3 * 4 * ((size_t)nmax + D) + (size_t)8 * nmax      // several positions per thread
                          : (size_t)2 * 4 * 4 * ((size_t)nmax + D) + 64 * sizeof(int);       // the split form (+ fast terms, wave reports)
        if (lds <= (size_t)lds_limit) break;
        if (S >= smax)
            return quiet ? ALIGNER_EDOM
                         : fail(ALIGNER_EDOM, "Ty=%d with max_duration=%d needs %zu bytes of LDS per segment (limit %d)", Ty,
                                D, lds, lds_limit);
    }
    pl.D = D;
    pl.S = S;
    pl.nmax = nmax;
    pl.NP = nmax <= 1024 ? 1 : 2;                  // 1: a position per thread (registers); 2: "several" (LDS, loops)
    // the split form: H lanes per position while that leaves at most one wave per SIMD of a CU
    pl.H = nmax <= 64 ? 4 : nmax <= 128 ? 2 : 1;
    if (g_opt_mobo_lanes == 1 || g_opt_mobo_lanes == 2 || g_opt_mobo_lanes == 4)       // development: force it
        if ((long long)nmax * g_opt_mobo_lanes <= 1024) pl.H = g_opt_mobo_lanes;
    pl.T = nmax <= 1024 ? (nmax * pl.H + 63) / 64 * 64 : 1024;
    if (nmax <= 1024 && S > 1) {                   // the split form's helper wave (the halo): one more wave, at most 16 in all
        if (pl.T + 64 > 1024) { pl.H = 1; pl.T = (nmax + 63) / 64 * 64; }
        if (pl.T + 64 <= 1024) pl.T += 64;
        else pl.NP = 2;                            // (1 024 positions in a split launch: the several-positions form)
    }
    pl.lds = lds;
    int RB = 32;                                   // rows per backtrack batch: a power of two with RB^2 * D <= 32768
    while (RB > 1 && (long long)RB * RB * D > 32768) RB >>= 1;
    pl.RB = RB;
    pl.bt_lds = (size_t)RB * (((size_t)(RB - 1) * D + 8) / 8 + 1) * 16;
    pl.ring_words = (size_t)B * (S - 1) * Tx * 3 * D;
    pl.status_off = 0;
    pl.fail_off = 256;
    pl.trash_off = align_up(pl.fail_off + (size_t)B * sizeof(int), 256);
    pl.L_off = align_up(pl.trash_off + ((size_t)B * S * 1024 + 64) * sizeof(unsigned), 256);
    pl.back_off = align_up(pl.L_off + (size_t)B * Tx * Ty * sizeof(float), 256);
    pl.bstride = (Ty + 1 + 7) / 8 * 8;
    pl.ring_off = align_up(pl.back_off + (size_t)B * Tx * pl.bstride * sizeof(unsigned short), 256);
    pl.total = align_up(pl.ring_off + pl.ring_words * sizeof(unsigned), 256);
    return ALIGNER_OK;
}

// the gradient's launch plan: the search's segments, its own workspace
struct MoboBwdPlan {
    MoboPlan f;
    int T;
    bool multi;
    size_t lds, ring_words;
    size_t status_off, fail_off, trash_off, L_off, G_off, Y_off, ring_off, total;
};
static int mobo_bwd_plan(int B, int Tx, int Ty, int max_duration, MoboBwdPlan &pl, bool quiet) {
    const int rc = mobo_plan(B, Tx, Ty, max_duration, pl.f, quiet);
    if (rc) return rc;
    const MoboPlan &f = pl.f;
    pl.multi = f.NP != 1;                          // the search's choice: the split form or several positions per thread
    pl.T = f.T;
    pl.lds = f.lds;                                // (Z, q, t and the reports, or Z, q and Y: never more than the search's)
    pl.ring_words = (size_t)B * (f.S - 1) * Tx * 2 * f.D;
    const size_t cells = align_up((size_t)B * Tx * Ty * sizeof(float), 256);
    pl.status_off = 0;
    pl.fail_off = 256;
    pl.trash_off = align_up(pl.fail_off + (size_t)B * sizeof(int), 256);
    pl.L_off = align_up(pl.trash_off + ((size_t)B * f.S * 1024 + 64) * sizeof(unsigned), 256);
    pl.G_off = pl.L_off + cells;
    pl.Y_off = pl.G_off + cells;
    pl.ring_off = pl.Y_off + cells;
    pl.total = align_up(pl.ring_off + pl.ring_words * sizeof(unsigned), 256);
    return ALIGNER_OK;
}

}  // namespace aligner

using namespace aligner;

extern "C" {

size_t aligner_boundary_search_backward_workspace_bytes(int B, int Tx, int Ty, int max_duration) {
    if (B < 0 || Tx < 1 || Ty < 1 || max_duration < 1) return 0;
    MoboBwdPlan pl;
    if (mobo_bwd_plan(B, Tx, Ty, max_duration, pl, true) != ALIGNER_OK) return 0;
    return pl.total;
}

int aligner_boundary_search_backward(const void *energies, int energy_dtype, const int32_t *t_xs, const int32_t *t_ys,
                                     int max_duration, const float *log_alpha, const float *grad_log_alpha,
                                     const float *grad_gamma, float *grad_energies_out, void *workspace,
                                     size_t workspace_bytes, int B, int Tx, int Ty, void *stream) {
    if (!energies || !t_xs || !t_ys || !log_alpha || !grad_energies_out || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (!grad_log_alpha && !grad_gamma) return fail(ALIGNER_EINVAL, "no cotangent: grad_log_alpha and grad_gamma are both null");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (max_duration < 1) return fail(ALIGNER_EINVAL, "max_duration %d < 1", max_duration);
    const int vt = energy_dtype == ALIGNER_DT_F32 ? 0 : energy_dtype == ALIGNER_DT_BF16 ? 1 : energy_dtype == ALIGNER_DT_F16 ? 2 : -1;
    if (vt < 0) return fail(ALIGNER_EINVAL, "energy dtype %d not supported (F32, BF16, F16)", energy_dtype);
    if (B == 0) return ALIGNER_OK;
    if (B > 65535 || Tx > 65535) return fail(ALIGNER_EDOM, "grid too large");
    if ((max_duration > Ty ? Ty : max_duration) > 65535) return fail(ALIGNER_EDOM, "max_duration %d too large", max_duration);
    MoboBwdPlan pl;
    const int prc = mobo_bwd_plan(B, Tx, Ty, max_duration, pl, false);
    if (prc) return prc;
    const MoboPlan &f = pl.f;
    if ((long long)B * f.S > 0x7fffffffLL) return fail(ALIGNER_EDOM, "grid too large");
    if (workspace_bytes < pl.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, pl.total);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nlds = (size_t)(MB_NCH + f.D + 4) * 12;
    if (nlds > (size_t)device_lds_limit()) return fail(ALIGNER_EDOM, "max_duration %d needs %zu bytes of LDS", f.D, nlds);
    {   // 1. normalisers (the search's kernel; it also refills the ring and clears the give-up words)
        MoboParams p{};
        p.e = energies;  p.t_xs = t_xs;  p.t_ys = t_ys;
        p.Lw = reinterpret_cast<float *>(ws + pl.L_off);
        p.ring = reinterpret_cast<unsigned *>(ws + pl.ring_off);
        p.failw = reinterpret_cast<int *>(ws + pl.fail_off);
        p.status = reinterpret_cast<int *>(ws + pl.status_off);
        p.B = B;  p.Tx = Tx;  p.Ty = Ty;  p.D = f.D;  p.S = f.S;  p.nmax = f.nmax;
        const dim3 grid((Ty + MB_NCH - 1) / MB_NCH, Tx, B);
        auto launch = [&](auto kern) -> int {
            ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), nlds));
            hipLaunchKernelGGL(kern, grid, dim3(256), nlds, s, p, (unsigned long long)pl.ring_words);
            ALIGNER_HIP_CHECK(hipGetLastError());
            return ALIGNER_OK;
        };
        const int rc = vt == 0 ? launch(mobo_norm_kernel<0>) : vt == 1 ? launch(mobo_norm_kernel<1>) : launch(mobo_norm_kernel<2>);
        if (rc) return rc;
    }
    float *Gw = reinterpret_cast<float *>(ws + pl.G_off);
    MoboBwdParams q{energies, t_xs, t_ys, log_alpha, grad_log_alpha, grad_gamma, grad_energies_out,
                    reinterpret_cast<float *>(ws + pl.L_off), grad_gamma ? Gw : grad_log_alpha, Gw,
                    reinterpret_cast<float *>(ws + pl.Y_off), reinterpret_cast<unsigned *>(ws + pl.ring_off),
                    reinterpret_cast<int *>(ws + pl.fail_off), reinterpret_cast<unsigned *>(ws + pl.trash_off),
                    reinterpret_cast<int *>(ws + pl.status_off), B, Tx, Ty, f.D, f.S, f.nmax, g_opt_mobo_start_lag,
                    g_opt_mobo_drop_segment, g_opt_mobo_drop_segment >= 0 ? 2048 : MB_SPIN_LIMIT, g_debug_stamps, g_opt_mobo_stamp_wave};
    if (grad_gamma) {   // 2. the direct cotangent of log_alpha with the gamma term folded in
        hipLaunchKernelGGL(mobo_bwd_cotangent_kernel, dim3(Tx, B), dim3(256), 0, s, q);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    {   // 3. the chain over the token rows, last to first
        auto launch = [&](auto kern) -> int {
            ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), pl.lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)B * f.S), dim3(pl.T), pl.lds, s, q);
            ALIGNER_HIP_CHECK(hipGetLastError());
            return ALIGNER_OK;
        };
#define MB_BWD_LAUNCH(VT_)                                                                                             \
    (pl.multi ? launch(mobo_bwd_chain_kernel<VT_, true>)                                                              \
     : g_opt_mobo_bwd_general ? launch(mobo_bwd_chain_kernel<VT_, false>)                                             \
     : g_debug_stamps ? MB_BWD_LAUNCH_H(VT_, true) : MB_BWD_LAUNCH_H(VT_, false))
#define MB_BWD_LAUNCH_H(VT_, ST_)                                                                                     \
    (f.H == 1 ? launch(mobo_bwd_chain_one_kernel<VT_, 1, ST_>) : f.H == 2 ? launch(mobo_bwd_chain_one_kernel<VT_, 2, ST_>) \
                                                                           : launch(mobo_bwd_chain_one_kernel<VT_, 4, ST_>))
        if (g_opt_mobo_bwd_general && !pl.multi) pl.T = (f.nmax + 63) / 64 * 64;      // (no helper wave, one position per thread)
        const int rc = vt == 0 ? MB_BWD_LAUNCH(0) : vt == 1 ? MB_BWD_LAUNCH(1) : MB_BWD_LAUNCH(2);
#undef MB_BWD_LAUNCH
#undef MB_BWD_LAUNCH_H
        if (rc) return rc;
    }
    {   // 4. the gradient, every cell at once
        const dim3 grid((Ty + MB_NCH - 1) / MB_NCH, Tx, B);
        auto launch = [&](auto kern) -> int {
            ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), nlds));
            hipLaunchKernelGGL(kern, grid, dim3(256), nlds, s, q);
            ALIGNER_HIP_CHECK(hipGetLastError());
            return ALIGNER_OK;
        };
        const int rc = vt == 0 ? launch(mobo_bwd_grad_kernel<0>) : vt == 1 ? launch(mobo_bwd_grad_kernel<1>) : launch(mobo_bwd_grad_kernel<2>);
        if (rc) return rc;
    }
    return ALIGNER_OK;
}

size_t aligner_boundary_search_workspace_bytes_ex(int B, int Tx, int Ty, int max_duration) {
    if (B < 0 || Tx < 1 || Ty < 1 || max_duration < 1) return 0;
    MoboPlan pl;
    if (mobo_plan(B, Tx, Ty, max_duration, pl, true) != ALIGNER_OK) return 0;
    return pl.total;
}

size_t aligner_boundary_search_workspace_bytes(int B, int Tx, int Ty) {
    if (B < 0 || Tx < 1 || Ty < 1) return 0;
    // whatever the window: the ring holds (S-1)*D <= Ty positions of 3 words per token row
    const size_t fixed = 256 + align_up((size_t)B * sizeof(int), 256) + align_up((size_t)B * Tx * Ty * sizeof(float), 256) +
                         align_up(((size_t)(B > 1024 ? B : 1024 + B) * 1024 + 64) * sizeof(unsigned), 256) +
                         align_up((size_t)B * Tx * ((size_t)Ty + 8) * sizeof(unsigned short), 256);
    return fixed + align_up((size_t)B * Tx * 3 * ((size_t)Ty + 1) * sizeof(unsigned), 256) + 256;
}

int aligner_boundary_search(const void *energies, int energy_dtype, const int32_t *t_xs, const int32_t *t_ys,
                            int max_duration, int32_t *boundaries_out, int32_t *durations_out, float *map_score_out,
                            float *log_alpha_out, float *gamma_out, void *workspace, size_t workspace_bytes, int B,
                            int Tx, int Ty, void *stream) {
    if (!energies || !t_xs || !t_ys || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (!boundaries_out && !log_alpha_out) return fail(ALIGNER_EINVAL, "nothing asked for: boundaries_out and log_alpha_out are both null");
    if (!boundaries_out && (durations_out || map_score_out))
        return fail(ALIGNER_EINVAL, "durations / map_score need boundaries_out (the MAP sequence is searched or it is not)");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (max_duration < 1) return fail(ALIGNER_EINVAL, "max_duration %d < 1", max_duration);
    const int vt = energy_dtype == ALIGNER_DT_F32 ? 0 : energy_dtype == ALIGNER_DT_BF16 ? 1 : energy_dtype == ALIGNER_DT_F16 ? 2 : -1;
    if (vt < 0) return fail(ALIGNER_EINVAL, "energy dtype %d not supported (F32, BF16, F16)", energy_dtype);
    if (gamma_out && !log_alpha_out) return fail(ALIGNER_EINVAL, "gamma needs the log_alpha buffer as well");
    if (B == 0) return ALIGNER_OK;
    if (B > 65535 || Tx > 65535) return fail(ALIGNER_EDOM, "grid too large");
    if ((max_duration > Ty ? Ty : max_duration) > 65535) return fail(ALIGNER_EDOM, "max_duration %d too large", max_duration);
    MoboPlan pl;
    const int prc = mobo_plan(B, Tx, Ty, max_duration, pl, false);
    if (prc) return prc;
    if ((long long)B * pl.S > 0x7fffffffLL) return fail(ALIGNER_EDOM, "grid too large");
    if (workspace_bytes < pl.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, pl.total);
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    MoboParams p{energies, t_xs, t_ys, log_alpha_out, boundaries_out, durations_out, map_score_out,
                 reinterpret_cast<float *>(ws + pl.L_off), reinterpret_cast<unsigned short *>(ws + pl.back_off),
                 reinterpret_cast<unsigned *>(ws + pl.ring_off), reinterpret_cast<int *>(ws + pl.fail_off),
                 reinterpret_cast<unsigned *>(ws + pl.trash_off),
                 reinterpret_cast<int *>(ws + pl.status_off), B, Tx, Ty, pl.D, pl.S, pl.nmax, pl.bstride, g_opt_mobo_start_lag, g_debug_stamps,
                 g_opt_mobo_drop_segment, g_opt_mobo_drop_segment >= 0 ? 2048 : MB_SPIN_LIMIT};
    {   // 1. normalisers (+ ring refill)
        const size_t nlds = (size_t)(MB_NCH + pl.D + 4) * 12;
        if (nlds > (size_t)device_lds_limit()) return fail(ALIGNER_EDOM, "max_duration %d needs %zu bytes of LDS", pl.D, nlds);
        const dim3 grid((Ty + MB_NCH - 1) / MB_NCH, Tx, B);
        auto launch = [&](auto kern) -> int {
            ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), nlds));
            hipLaunchKernelGGL(kern, grid, dim3(256), nlds, s, p, (unsigned long long)pl.ring_words);
            ALIGNER_HIP_CHECK(hipGetLastError());
            return ALIGNER_OK;
        };
        const int rc = vt == 0 ? launch(mobo_norm_kernel<0>) : vt == 1 ? launch(mobo_norm_kernel<1>) : launch(mobo_norm_kernel<2>);
        if (rc) return rc;
    }
    {   // 2. the chain
        auto launch = [&](auto kern) -> int {
            ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), pl.lds));
            hipLaunchKernelGGL(kern, dim3((unsigned)B * pl.S), dim3(pl.T), pl.lds, s, p);
            ALIGNER_HIP_CHECK(hipGetLastError());
            return ALIGNER_OK;
        };
        int rc;
#define MB_LAUNCH_NP(VT_)                                                                               \
    (pl.NP != 1 ? launch(mobo_chain_kernel<VT_, true>)                                                  \
     : (!boundaries_out && !g_opt_mobo_full_chain && !g_debug_stamps) ? MB_LAUNCH_SUM(VT_)              \
     : log_alpha_out ? MB_LAUNCH_H(VT_, true)                                                           \
     : (g_opt_mobo_full_chain || g_debug_stamps) ? MB_LAUNCH_H(VT_, false) : MB_LAUNCH_MAP(VT_))
#define MB_LAUNCH_SUM(VT_)                                                                              \
    (pl.H == 1 ? launch(mobo_chain_sum_kernel<VT_, 1>) : pl.H == 2 ? launch(mobo_chain_sum_kernel<VT_, 2>) \
                                                                    : launch(mobo_chain_sum_kernel<VT_, 4>))
#define MB_LAUNCH_MAP(VT_)                                                                              \
    (pl.H == 1 ? launch(mobo_chain_map_kernel<VT_, 1>) : pl.H == 2 ? launch(mobo_chain_map_kernel<VT_, 2>) \
                                                                    : launch(mobo_chain_map_kernel<VT_, 4>))
#define MB_LAUNCH_H(VT_, LA_)                                                                            \
    (g_debug_stamps ? MB_LAUNCH_ST(VT_, LA_, true) : MB_LAUNCH_ST(VT_, LA_, false))
#define MB_LAUNCH_ST(VT_, LA_, ST_)                                                                      \
    (pl.H == 1 ? launch(mobo_chain_one_kernel<VT_, LA_, 1, ST_>) : pl.H == 2 ? launch(mobo_chain_one_kernel<VT_, LA_, 2, ST_>) \
                                                                              : launch(mobo_chain_one_kernel<VT_, LA_, 4, ST_>))
        rc = vt == 0 ? MB_LAUNCH_NP(0) : vt == 1 ? MB_LAUNCH_NP(1) : MB_LAUNCH_NP(2);
        if (rc) return rc;
    }
    if (boundaries_out) {   // 3. the MAP sequence
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(mobo_backtrack_kernel), pl.bt_lds));
        hipLaunchKernelGGL(mobo_backtrack_kernel, dim3(B), dim3(256), pl.bt_lds, s, p, pl.RB);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    if (gamma_out) {
        hipLaunchKernelGGL(mobo_gamma_kernel, dim3(Tx, B), dim3(256), 0, s, log_alpha_out, t_xs, t_ys, gamma_out, Tx, Ty);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    return ALIGNER_OK;
}

}  // extern "C"
