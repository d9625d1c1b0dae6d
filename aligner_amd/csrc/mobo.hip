// MoBoAligner monotonic boundary search on MI355X (gfx950): BASELINE config 5 / SURVEY.md section 8 rows a7, f3.
//
// Build-defined spec (the reference snapshot only names the branch and links the paper, README.md:9-13,49;
// restated in oracle/mobo_oracle.py, parity UNPINNED): tokens i, frames y, energies e[i,y]; a segmentation is a
// boundary sequence 0 = b_-1 < b_0 < ... < b_{I-1} = J with durations 1..D (the maximum-duration window);
//     P(b_i = j | b_{i-1} = k) = exp(e[i,j-1]) / sum_{m in A_i(k)} exp(e[i,m-1]),
//     A_i(k) = (k, k+D] intersected with [lo_i, hi_i]   (the later tokens still fit: see the oracle).
// Outputs: log_alpha[i,j-1] = log P(b_i = j) (sum-product), the MAP boundary sequence (max-product, ties: the
// shortest token) with its log-probability, and -- a second, row-parallel kernel -- the soft alignment
// gamma[i,y] = P(b_{i-1} <= y < b_i).
//
// Shape of the computation.  Token rows are a dependent chain (row i needs row i-1); inside a row every boundary
// position is independent, so lanes own positions and one workgroup owns one utterance (grid = batch, like the
// alignment search).  Both steps of a row are sliding-window reductions of width D over the position axis,
//     L_i(k)      = logsumexp_{m in A_i(k)} e_i(m)                       (normaliser of the step out of k)
//     la_i(j)     = e_i(j) + logsumexp_{k in [j-D, j)} (la_{i-1}(k) - L_i(k))
//     delta_i(j)  = e_i(j) +    max    _{k in [j-D, j)} (delta_{i-1}(k) - L_i(k))   (+ argmax)
// and a width-D window over blocks of D aligned positions is (a suffix of one block) + (a prefix of the next):
// two lookups into per-block prefix / suffix sums instead of D terms.  The sums are fp64 (2^(v - block max), a
// chain of plain adds) and the two parts of a window are only ever ADDED, so a window keeps full relative
// precision however far below the row's bulk it lies (a row-wide prefix sum would cancel there, fp32 sums
// normalised by one block maximum underflow).  The exp2s run one position per lane; only the adds of a block's
// scan are serial (one block per thread, J/D threads busy).  All logs are base 2 inside (v_exp_f32 / v_log_f32 are exp2 / log2), "log 0" is the
// finite -1e30, which absorbs every addend: no inf - inf.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

constexpr float MB_NEG = -1e30f;                  // log 0
constexpr float MB_LOG2E = 1.4426950408889634f, MB_LN2 = 0.6931471805599453f;
constexpr int MB_THREADS = 1024;

struct MoboParams {
    const void *e;            // [B,Tx,Ty] fp32 / bf16 / fp16
    int vt;                   // 0 f32, 1 bf16, 2 f16
    const int *t_xs, *t_ys;   // [B]
    float *log_alpha;         // nullable [B,Tx,Ty]
    int *boundaries;          // [B,Tx]
    int *durations;           // nullable [B,Tx]
    float *map_score;         // nullable [B]
    unsigned short *back;     // workspace [B,Tx,Ty+1]: duration of token i when it ends at j
    int *status;              // workspace: ALIGNER_ST_BAD_LENGTHS when an utterance is infeasible
    int B, Tx, Ty, D, P;      // P = padded positions (multiple of D, >= Ty+1)
};

// raw bits of one energy (the load), and their value (the conversion): kept apart so that the loads of the next
// row can be in flight, unconverted, while this row is computed
template <int VT> __device__ __forceinline__ unsigned mb_load_raw(const void *base, size_t idx) {
    if (VT == 0) return static_cast<const unsigned *>(base)[idx];
    return static_cast<const unsigned short *>(base)[idx];
}
template <int VT> __device__ __forceinline__ float mb_value(unsigned raw) {
    if (VT == 0) return __builtin_bit_cast(float, raw);
    if (VT == 1) return __builtin_bit_cast(float, raw << 16);
    return (float)__builtin_bit_cast(_Float16, (unsigned short)raw);
}

// Position arrays in LDS carry one pad word per block of D (index = block * (D+1) + offset): the scan threads
// walk their blocks in step, D words apart -- without the pad every lane of a wave would hit the same bank.
//
// Block scans in three steps, so that the only serial part is a chain of fp64 adds:
//   mb_block_max   one block per thread: M[blk] = max of the block                       (D reads)
//   mb_block_pow   one position per lane: T[x] = 2^(v - M[blk]) as a DOUBLE              (the exp2s, all parallel)
//   mb_block_sums  one block per thread: S[x] = suffix sums of T, then T[x] = prefix sums (in place)
// A double keeps 2^-1000: a prefix or suffix that lies hundreds of bits below the block's largest entry -- a
// window far from the row's bulk is made of exactly such parts -- still has full relative precision (fp32 sums
// normalised by one block maximum underflow there; measured, tests/test_mobo.py rows 57+ of the [64,257] case).
__device__ __forceinline__ void mb_block_max(const float *__restrict__ x, float *__restrict__ bmax, int nblk, int D) {
    for (int blk = threadIdx.x; blk < nblk; blk += blockDim.x) {
        const int o = blk * (D + 1);
        float M = MB_NEG;
#pragma unroll 8
        for (int r = 0; r < D; ++r) {                               // (unrolled: the LDS reads of a batch go out together)
            const float v = x[o + r];
            M = (v > M) ? v : M;          // fmaxf(M, v) for a never-NaN M -- as fmaxf hipcc vectorised this reduction
        }                                 // with a NaN test and an exec-mask exit per element
        bmax[blk] = (M > 0.5f * MB_NEG) ? ceilf(M) : MB_NEG;      // an INTEGER reference: two blocks' sums are brought
                                                                  // to a common scale by an exact ldexp (mb_window)
    }
}
__device__ __forceinline__ double mb_pow2(float f) {            // 2^f for f <= 0 as a double (0 below 2^-1000)
    if (!(f > -1000.f)) return 0.0;
    const float k = floorf(f);
    return __builtin_ldexp((double)__builtin_amdgcn_exp2f(f - k), (int)k);
}
__device__ __forceinline__ float mb_log2(double s) {              // log2 of a double (MB_NEG for 0)
    if (!(s > 0.0)) return MB_NEG;
    int ex;
    const double mant = __builtin_frexp(s, &ex);
    return (float)ex + __builtin_amdgcn_logf((float)mant);
}
__device__ __forceinline__ void mb_block_sums(double *__restrict__ T, double *__restrict__ S, int nblk, int D) {
    for (int blk = threadIdx.x; blk < nblk; blk += blockDim.x) {
        const int o = blk * (D + 1);
        double acc = 0.0;
#pragma unroll 8
        for (int r = D - 1; r >= 0; --r) { acc += T[o + r]; S[o + r] = acc; }
        acc = 0.0;
        int r = 0;
        for (; r + 8 <= D; r += 8) {               // in place: read a batch, then write it
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = T[o + r + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc += t[u]; T[o + r + u] = acc; }
        }
        for (; r < D; ++r) { acc += T[o + r]; T[o + r] = acc; }
    }
}
// log2(2^M1 * S1 + 2^M2 * S2): the two parts of a window, each relative to its block's (integer) reference
__device__ __forceinline__ float mb_window(float M1, double S1, float M2, double S2) {
    const bool a = S1 > 0.0, b = S2 > 0.0;
    if (!a && !b) return MB_NEG;
    const float m = fmaxf(a ? M1 : MB_NEG, b ? M2 : MB_NEG);
    const float d1 = a ? fmaxf(M1 - m, -2000.f) : 0.f, d2 = b ? fmaxf(M2 - m, -2000.f) : 0.f;   // integers <= 0
    const double t = (a ? __builtin_ldexp(S1, (int)d1) : 0.0) + (b ? __builtin_ldexp(S2, (int)d2) : 0.0);
    return m + mb_log2(t);
}

// One block per thread: position of the maximum of every prefix (ties: the LARGEST position) and of every
// suffix (ties: the largest position as well), as offsets inside the block.
// (`first`: the thread that takes block 0 -- the caller runs this scan on other waves than the block maxima that
// share its phase, so the two serial scans of phase D go side by side)
__device__ __forceinline__ void mb_scan_argmax(const float *__restrict__ x, unsigned short *__restrict__ ipre,
                                               unsigned short *__restrict__ isuf, int nblk, int D, int first) {
    const int nthr = (int)blockDim.x, t0 = ((int)threadIdx.x - first + nthr) % nthr;
    for (int blk = t0; blk < nblk; blk += nthr) {
        const int o = blk * (D + 1);
        float m = x[o];
        int im = 0;
        ipre[o] = 0;
#pragma unroll 8
        for (int r = 1; r < D; ++r) {
            if (x[o + r] >= m) { m = x[o + r]; im = r; }
            ipre[o + r] = (unsigned short)im;
        }
        m = x[o + D - 1];
        im = D - 1;
        isuf[o + D - 1] = (unsigned short)im;
#pragma unroll 8
        for (int r = D - 2; r >= 0; --r) {
            if (x[o + r] > m) { m = x[o + r]; im = r; }
            isuf[o + r] = (unsigned short)im;
        }
    }
}

// Barrier for LDS hand-offs only: __syncthreads() also drains vmcnt, which would put the latency of the next
// row's energy loads and of this row's result stores on every phase of the row loop.
__device__ __forceinline__ void mb_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// every boundary position of this thread (j = tid, tid + nthr, ...), as a ROLLED loop: block and offset advance by
// (nthr / D, nthr % D) without a division; the heavy phase bodies are instantiated once (unrolled x5 they spilled)
#define MB_FOR_POS(...)                                                                        \
    {                                                                                          \
        int blk = blk0, r = r0;                                                                \
        _Pragma("unroll 1") for (int j = tid; j < P; j += nthr) {                              \
            const int x = blk * (D + 1) + r;                                                   \
            __VA_ARGS__                                                                        \
            r += rstep;                                                                        \
            blk += bstep;                                                                      \
            if (r >= D) { r -= D; ++blk; }                                                     \
        }                                                                                      \
    }
constexpr int MB_NPOS = 5;     // boundary positions per thread (P <= 5 * 1024)

template <int VT>
__global__ __launch_bounds__(MB_THREADS) void mobo_forward_kernel(MoboParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int b = blockIdx.x;
    const int P = p.P, D = p.D, nblk = P / D, PP = P + nblk;     // PP: padded array length
    double *sT = reinterpret_cast<double *>(smem);  // 2^(v - block max), then the block prefix sums (in place)
    double *sS = sT + PP;                           // block suffix sums
    float *sE = reinterpret_cast<float *>(sS + PP); // e_i at boundary position j (energy of the token's last frame j-1)
    float *sV = sE + PP;                            // delta_{i-1}(k) - L_i(k)
    float *sA = sV + PP;                            // la_{i-1}; la_{i-1}(k) - L_i(k) between steps C and E; then la_i
    float *sDl = sA + PP;                           // delta
    float *sM = sDl + PP;                           // [nblk] block maxima (of e, then of U)
    unsigned short *iP = reinterpret_cast<unsigned short *>(sM + nblk);     // [PP] prefix argmax of V
    unsigned short *iS = iP + PP;                                           // [PP] suffix argmax of V

    int I = p.t_xs[b], J = p.t_ys[b];
    I = I > p.Tx ? p.Tx : I;
    J = J > p.Ty ? p.Ty : J;
    const bool ok = I >= 1 && J >= I && (long long)J <= (long long)I * D;
    if (!ok) {                                     // infeasible: no segmentation exists
        if (tid == 0) atomicOr(p.status, ALIGNER_ST_BAD_LENGTHS);
        for (int i = tid; i < p.Tx; i += nthr) {
            p.boundaries[(size_t)b * p.Tx + i] = 0;
            if (p.durations) p.durations[(size_t)b * p.Tx + i] = 0;
        }
        if (p.log_alpha)
            for (size_t n = tid; n < (size_t)p.Tx * p.Ty; n += nthr) p.log_alpha[(size_t)b * p.Tx * p.Ty + n] = -__builtin_huge_valf();
        if (p.map_score && tid == 0) p.map_score[b] = -__builtin_huge_valf();
        return;
    }
    // this thread's boundary positions (the same for every token row): j, its block, offset and padded index
    int pj[MB_NPOS], pb[MB_NPOS], pr[MB_NPOS];
#pragma unroll
    for (int n = 0; n < MB_NPOS; ++n) {
        const int j = tid + n * nthr;
        pj[n] = j < P ? j : -1;
        pb[n] = j / D;
        pr[n] = j - pb[n] * D;
    }
#pragma unroll
    for (int n = 0; n < MB_NPOS; ++n)
        if (pj[n] >= 0) {
            const int x = pb[n] * (D + 1) + pr[n];
            sA[x] = (pj[n] == 0) ? 0.f : MB_NEG;   // P(b_-1 = 0) = 1
            sDl[x] = (pj[n] == 0) ? 0.f : MB_NEG;
        }
    __syncthreads();
    unsigned short *backb = p.back + (size_t)b * p.Tx * (p.Ty + 1);
    const int blk0 = tid / D, r0 = tid - blk0 * D, bstep = nthr / D, rstep = nthr - bstep * D;
    unsigned enext[MB_NPOS];                       // raw bits of the next row's energies (clamped index: no branch)
#pragma unroll
    for (int n = 0; n < MB_NPOS; ++n) {
        const int jj = pj[n] < 1 ? 1 : (pj[n] > J ? J : pj[n]);
        enext[n] = mb_load_raw<VT>(p.e, (size_t)b * p.Tx * p.Ty + (jj - 1));
    }
    for (int i = 0; i < I; ++i) {
        const int hi = J - (I - 1 - i);
        const long long lo64 = (long long)J - (long long)(I - 1 - i) * D;
        const int lo = (lo64 > i + 1) ? (int)lo64 : i + 1;
        // ---- A: this token's energies at the feasible boundary positions, base-2 (fetched during the previous
        //         row: the loads of row i+1 are in flight while row i is computed) ----
        const size_t rowoff = ((size_t)b * p.Tx + i) * p.Ty;
#pragma unroll
        for (int n = 0; n < MB_NPOS; ++n)
            if (pj[n] >= 0) {
                const int j = pj[n];
                sE[pb[n] * (D + 1) + pr[n]] = (j >= lo && j <= hi) ? mb_value<VT>(enext[n]) * MB_LOG2E : MB_NEG;
            }
        mb_lds_barrier();
        if (i + 1 < I) {
            const size_t nextoff = rowoff + p.Ty;
#pragma unroll
            for (int n = 0; n < MB_NPOS; ++n) {
                const int jj = pj[n] < 1 ? 1 : (pj[n] > J ? J : pj[n]);
                enext[n] = mb_load_raw<VT>(p.e, nextoff + (jj - 1));
            }
        }
        // ---- B: block scans of e ----
        mb_block_max(sE, sM, nblk, D);
        mb_lds_barrier();
        MB_FOR_POS({ sT[x] = mb_pow2(sE[x] - sM[blk]); });
        mb_lds_barrier();
        mb_block_sums(sT, sS, nblk, D);
        mb_lds_barrier();
        // ---- C: normaliser of the step out of k: positions (k, k+D] = rest of k's block + head of the next.
        //         u = la_{i-1}(k) - L replaces la_{i-1}(k) in place (a lane only ever reads its own entry of sA) ----
        MB_FOR_POS({
            const double S1 = (r + 1 < D) ? sS[x + 1] : 0.0;
            const double S2 = (blk + 1 < nblk) ? sT[(blk + 1) * (D + 1) + r] : 0.0;
            const float L = mb_window(sM[blk], S1, (blk + 1 < nblk) ? sM[blk + 1] : MB_NEG, S2);
            const bool live = L > 0.5f * MB_NEG;
            const float a = sA[x], d = sDl[x];
            sA[x] = (live && a > 0.5f * MB_NEG) ? a - L : MB_NEG;
            sV[x] = (live && d > 0.5f * MB_NEG) ? d - L : MB_NEG;
        });
        mb_lds_barrier();
        // ---- D: block scans of U (sums) and V (argmax) ----
        mb_block_max(sA, sM, nblk, D);
        mb_scan_argmax(sV, iP, iS, nblk, D, (nthr >= 2 * nblk) ? (nthr / 2) & ~63 : 0);
        mb_lds_barrier();
        MB_FOR_POS({ sT[x] = mb_pow2(sA[x] - sM[blk]); });
        mb_lds_barrier();
        mb_block_sums(sT, sS, nblk, D);
        mb_lds_barrier();
        // ---- E: window [j-D, j) = tail of the previous block + head of j's block ----
        MB_FOR_POS({
            float la = MB_NEG, de = MB_NEG;
            int dur = 0;
            if (j >= lo && j <= hi) {
                const int xp = (blk - 1) * (D + 1) + r;              // same offset, previous block
                const double S1 = (blk >= 1) ? sS[xp] : 0.0;
                const double S2 = (r >= 1) ? sT[x - 1] : 0.0;
                const float w = mb_window((blk >= 1) ? sM[blk - 1] : MB_NEG, S1, sM[blk], S2);
                if (w > 0.5f * MB_NEG) la = sE[x] + w;
                // max-product twin: best previous boundary, the larger position on a tie
                int kb = -1;
                float vb = MB_NEG;
                if (blk >= 1) {
                    const int rr = iS[xp];
                    kb = (blk - 1) * D + rr;
                    vb = sV[(blk - 1) * (D + 1) + rr];
                }
                if (r >= 1) {
                    const int rr = iP[x - 1];
                    const float v2 = sV[blk * (D + 1) + rr];
                    if (v2 >= vb) { kb = blk * D + rr; vb = v2; }
                }
                if (kb >= 0 && vb > 0.5f * MB_NEG) { de = sE[x] + vb; dur = j - kb; }
            }
            if (j <= p.Ty) backb[(size_t)i * (p.Ty + 1) + j] = (unsigned short)dur;
            if (p.log_alpha && j >= 1 && j <= p.Ty)
                p.log_alpha[rowoff + (j - 1)] = (la > 0.5f * MB_NEG) ? la * MB_LN2 : -__builtin_huge_valf();
            sA[x] = la;                                // (the windows read the scans, not these)
            sDl[x] = de;
        });
        mb_lds_barrier();
    }
    if (p.log_alpha)       // rows past the utterance's own text
        for (size_t n = (size_t)I * p.Ty + tid; n < (size_t)p.Tx * p.Ty; n += nthr)
            p.log_alpha[(size_t)b * p.Tx * p.Ty + n] = -__builtin_huge_valf();
    __threadfence_block();
    __syncthreads();
    // ---- backtrack of the MAP sequence: I dependent look-ups ----
    if (tid == 0) {
        const float sc = sDl[(J / D) * (D + 1) + (J % D)];
        if (p.map_score) p.map_score[b] = (sc > 0.5f * MB_NEG) ? sc * MB_LN2 : -__builtin_huge_valf();
        int j = J;
        for (int i = I - 1; i >= 0; --i) {
            const int d = backb[(size_t)i * (p.Ty + 1) + j];
            p.boundaries[(size_t)b * p.Tx + i] = j;
            if (p.durations) p.durations[(size_t)b * p.Tx + i] = d;
            j -= d;
        }
        if (j != 0) atomicOr(p.status, ALIGNER_ST_INTERNAL);
    }
    for (int i = I + tid; i < p.Tx; i += nthr) {
        p.boundaries[(size_t)b * p.Tx + i] = J;
        if (p.durations) p.durations[(size_t)b * p.Tx + i] = 0;
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

struct MoboWs { size_t status_off, back_off, total; };
static MoboWs mobo_ws(int B, int Tx, int Ty) {
    MoboWs L;
    L.status_off = 0;
    L.back_off = 256;
    L.total = align_up(L.back_off + (size_t)B * Tx * (Ty + 1) * sizeof(unsigned short), 256);
    return L;
}

}  // namespace aligner

using namespace aligner;

extern "C" {

size_t aligner_boundary_search_workspace_bytes(int B, int Tx, int Ty) {
    if (B < 0 || Tx < 1 || Ty < 1) return 0;
    return mobo_ws(B, Tx, Ty).total;
}

int aligner_boundary_search(const void *energies, int energy_dtype, const int32_t *t_xs, const int32_t *t_ys,
                            int max_duration, int32_t *boundaries_out, int32_t *durations_out, float *map_score_out,
                            float *log_alpha_out, float *gamma_out, void *workspace, size_t workspace_bytes, int B,
                            int Tx, int Ty, void *stream) {
    if (!energies || !t_xs || !t_ys || !boundaries_out || !workspace) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape B=%d Tx=%d Ty=%d", B, Tx, Ty);
    if (max_duration < 1) return fail(ALIGNER_EINVAL, "max_duration %d < 1", max_duration);
    const int vt = energy_dtype == ALIGNER_DT_F32 ? 0 : energy_dtype == ALIGNER_DT_BF16 ? 1 : energy_dtype == ALIGNER_DT_F16 ? 2 : -1;
    if (vt < 0) return fail(ALIGNER_EINVAL, "energy dtype %d not supported (F32, BF16, F16)", energy_dtype);
    if (gamma_out && !log_alpha_out) return fail(ALIGNER_EINVAL, "gamma needs the log_alpha buffer as well");
    if (B == 0) return ALIGNER_OK;
    if (B > 65535 || Tx > 65535) return fail(ALIGNER_EDOM, "grid too large");
    const int D = max_duration > Ty ? Ty : max_duration;
    if (D > 65535) return fail(ALIGNER_EDOM, "max_duration %d too large", max_duration);
    const MoboWs L = mobo_ws(B, Tx, Ty);
    if (workspace_bytes < L.total) return fail(ALIGNER_ENOSPC, "workspace %zu < %zu bytes", workspace_bytes, L.total);
    const int P = (Ty + 1 + D - 1) / D * D;
    const int nblk = P / D;
    const size_t PP = (size_t)P + nblk;            // one pad word per block
    const size_t lds = 2 * PP * sizeof(double) + 4 * PP * sizeof(float) + (size_t)nblk * sizeof(float) +
                       2 * PP * sizeof(unsigned short) + 16;
    if (P > MB_NPOS * 1024) return fail(ALIGNER_EDOM, "Ty=%d exceeds %d boundary positions", Ty, MB_NPOS * 1024);
    if (lds > (size_t)device_lds_limit())
        return fail(ALIGNER_EDOM, "Ty=%d with max_duration=%d needs %zu bytes of LDS (limit %d)", Ty, D, lds, device_lds_limit());
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    MoboParams p{energies, vt, t_xs, t_ys, log_alpha_out, boundaries_out, durations_out, map_score_out,
                 reinterpret_cast<unsigned short *>(ws + L.back_off), reinterpret_cast<int *>(ws + L.status_off), B, Tx, Ty,
                 D, P};
    const int threads = P >= 1024 ? 1024 : (P + 63) / 64 * 64;
    auto launch = [&](auto kern) -> int {
        ALIGNER_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
        hipLaunchKernelGGL(kern, dim3(B), dim3(threads), lds, s, p);
        ALIGNER_HIP_CHECK(hipGetLastError());
        return ALIGNER_OK;
    };
    const int rc = vt == 0 ? launch(mobo_forward_kernel<0>) : vt == 1 ? launch(mobo_forward_kernel<1>) : launch(mobo_forward_kernel<2>);
    if (rc) return rc;
    if (gamma_out) {
        hipLaunchKernelGGL(mobo_gamma_kernel, dim3(Tx, B), dim3(256), 0, s, log_alpha_out, t_xs, t_ys, gamma_out, Tx, Ty);
        ALIGNER_HIP_CHECK(hipGetLastError());
    }
    return ALIGNER_OK;
}

}  // extern "C"
