// The two small steps either side of the alignment hot path (not in the reference
// snapshot -- README.md:21-25,50 only points at the OTA paper -- so the spec is build-defined: DESIGN.md 5 / DESIGN_HISTORY.md 7,
// oracle/forward_sum_oracle.py):
//
//   beta-binomial prior   prior[b,x,y] = BetaBinomial(n = t_x, a = s*(y+1), b = s*(t_y-y)).pmf(x)
//                         in the DP's [B, T_text, T_mel] layout: what soft_attention() adds as
//                         log(prior + 1e-8) before the path search
//   length regulator      out[b,c,y] = h[b,c,x(y)], x(y) = the token that owns frame y under the
//                         durations the path search produced (frames past sum(dur): 0)
//
// Both are streaming writes of a [B,*,Ty] tensor with the mel axis contiguous: a thread owns a frame,
// loops over rows / channels, consecutive threads store consecutive frames.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "aligner_amd.h"
#include "common.h"

namespace aligner {

// log pmf(x) = lgamma(n+1) - lgamma(x+1) - lgamma(n-x+1) + lgamma(x+a) + lgamma(n-x+b) - lgamma(n+a+b)
//              - lgamma(a) - lgamma(b) + lgamma(a+b)
// The two terms that depend on both x and the frame are walked with lgamma(z+1) = lgamma(z) + log z:
// three lgamma per frame, two logs and one exp per cell, all in double (fp32 lgamma near 2000 is off
// by 1e-3, i.e. 0.1 % of the pmf).
__global__ __launch_bounds__(256) void prior_kernel(const int *__restrict__ t_xs, const int *__restrict__ t_ys,
                                                    float *__restrict__ prior, int Tx, int Ty, double scaling,
                                                    int ktab) {
    extern __shared__ double logfact[];                      // lgamma(k+1), k = 0..max(n, ktab)
    const int b = blockIdx.y;
    const int y = blockIdx.x * 256 + threadIdx.x;
    int n = t_xs[b], ty = t_ys[b];
    n = n < 0 ? 0 : (n > Tx ? Tx : n);
    ty = ty < 0 ? 0 : (ty > Ty ? Ty : ty);
    for (int k = threadIdx.x; k <= (n > ktab ? n : ktab); k += 256) logfact[k] = lgamma((double)k + 1.0);
    __syncthreads();
    if (y >= Ty) return;
    float *out = prior + (size_t)b * Tx * Ty + y;
    if (y >= ty || n < 1) {
        for (int x = 0; x < Tx; ++x) out[(size_t)x * Ty] = 0.f;
        return;
    }
    const double a = scaling * (double)(y + 1), bb = scaling * (double)(ty - y);
    if (ktab > 0) {
        // integer scaling (the usual 1.0): a and b are integers, every lgamma is a log-factorial out of the
        // LDS table (filled up to n + s*(t_y + 1) by the caller's choice of `ktab`): no log in the cell loop
        const int ia = (int)a, ib = (int)bb;
        const double frame = logfact[n] - logfact[n + ia + ib - 1] - logfact[ia - 1] - logfact[ib - 1] + logfact[ia + ib - 1];
        for (int x = 0; x < n; ++x) {
            const double lp = frame - logfact[x] - logfact[n - x] + logfact[x + ia - 1] + logfact[n - x + ib - 1];
            out[(size_t)x * Ty] = (float)exp(lp);
        }
        for (int x = n; x < Tx; ++x) out[(size_t)x * Ty] = 0.f;
        return;
    }
    const double frame = logfact[n] - lgamma((double)n + a + bb) - lgamma(a) - lgamma(bb) + lgamma(a + bb);
    double l1 = lgamma(a);                                   // lgamma(x + a) at x = 0
    double l2 = lgamma((double)n + bb);                      // lgamma(n - x + b) at x = 0
    for (int x = 0; x < n; ++x) {
        const double lp = frame - logfact[x] - logfact[n - x] + l1 + l2;
        out[(size_t)x * Ty] = (float)exp(lp);
        l1 += log((double)x + a);                            // -> lgamma(x + 1 + a)
        l2 -= log((double)(n - x - 1) + bb);                 // -> lgamma(n - x - 1 + b)
    }
    for (int x = n; x < Tx; ++x) out[(size_t)x * Ty] = 0.f;
}

// durations -> token of every frame (inclusive scan in LDS, bisection per frame) -> gather.
// One workgroup per 256 frames of one utterance; the scan of <= 2048 durations is redone per workgroup.
__global__ __launch_bounds__(256) void regulate_kernel(const float *__restrict__ h, const int *__restrict__ dur,
                                                       float *__restrict__ out, int *__restrict__ tok_out, int C,
                                                       int Tx, int Ty) {
    extern __shared__ int ends[];                            // ends[x] = sum(dur[0..x]); ends[Tx..] scratch
    int *part = ends + Tx;                                   // [256] per-thread partial sums
    const int b = blockIdx.y, tid = threadIdx.x;
    const int per = (Tx + 255) / 256;                        // consecutive tokens per thread
    const int x0 = tid * per;
    int s = 0;
    for (int i = 0; i < per; ++i) {
        const int x = x0 + i;
        int d = (x < Tx) ? dur[(size_t)b * Tx + x] : 0;
        d = d < 0 ? 0 : d;
        s += d;
        if (x < Tx) ends[x] = s;                             // local inclusive sum for now
    }
    part[tid] = s;
    __syncthreads();
    // exclusive scan of the 256 partials (Hillis-Steele in place, 8 rounds)
    for (int o = 1; o < 256; o <<= 1) {
        const int v = (tid >= o) ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const int base = (tid > 0) ? part[tid - 1] : 0;
    for (int i = 0; i < per; ++i)
        if (x0 + i < Tx) ends[x0 + i] += base;
    __syncthreads();
    const int y = blockIdx.x * 256 + tid;
    if (y >= Ty) return;
    const int total = ends[Tx - 1];
    int tok = -1;
    if (y < total) {                                         // first x with ends[x] > y
        int lo = 0, hi = Tx - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ends[mid] > y) hi = mid; else lo = mid + 1;
        }
        tok = lo;
    }
    if (tok_out && blockIdx.z == 0) tok_out[(size_t)b * Ty + y] = tok;
    if (out) {
        // this workgroup's slice of the channels (gridDim.z slices), eight gathers in flight per thread.
        // The loads are unconditional (token clamped) and masked afterwards: a select fed by a load is
        // compiled to "load, wait, select", one memory round trip per element.
        const int cpz = (C + gridDim.z - 1) / gridDim.z;
        const int c0 = blockIdx.z * cpz, c1 = (c0 + cpz < C) ? c0 + cpz : C;
        const unsigned keep = (tok >= 0) ? 0xFFFFFFFFu : 0u;
        const int tc = tok >= 0 ? tok : 0;
        const float *hb = h + (size_t)b * C * Tx + tc;
        float *ob = out + (size_t)b * C * Ty + y;
        int c = c0;
        for (; c + 8 <= c1; c += 8) {
            unsigned v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(unsigned, hb[(size_t)(c + i) * Tx]);
#pragma unroll
            for (int i = 0; i < 8; ++i) ob[(size_t)(c + i) * Ty] = __builtin_bit_cast(float, v[i] & keep);
        }
        for (; c < c1; ++c) ob[(size_t)c * Ty] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, hb[(size_t)c * Tx]) & keep);
    }
}

}  // namespace aligner

using namespace aligner;

extern "C" {

int aligner_beta_binomial_prior_f32(const int32_t *t_xs, const int32_t *t_ys, float *prior_out, int B, int Tx, int Ty,
                                    float scaling, void *stream) {
    if (!t_xs || !t_ys || !prior_out) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || Tx < 1 || Ty < 1 || !(scaling > 0.f)) return fail(ALIGNER_EINVAL, "bad shape / scaling");
    if (B > 65535) return fail(ALIGNER_EDOM, "B=%d too large", B);
    if ((size_t)(Tx + 1) * sizeof(double) > 60 * 1024) return fail(ALIGNER_EDOM, "Tx=%d too large", Tx);
    if (B == 0) return ALIGNER_OK;
    dim3 grid((Ty + 255) / 256, B);
    // integer scaling: log-factorial table up to Tx + s*(Ty + 1) in LDS if it fits, else the lgamma recurrence
    int ktab = 0;
    size_t lds = (size_t)(Tx + 1) * sizeof(double);
    if (scaling == std::floor(scaling) && scaling >= 1.f) {
        const double top = (double)Tx + (double)scaling * ((double)Ty + 1.0) + 1.0;
        if ((top + 1.0) * sizeof(double) <= 60.0 * 1024) {
            ktab = (int)top;
            lds = (size_t)(ktab + 1) * sizeof(double);
        }
    }
    hipLaunchKernelGGL(prior_kernel, grid, dim3(256), lds, static_cast<hipStream_t>(stream),
                       t_xs, t_ys, prior_out, Tx, Ty, (double)scaling, ktab);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

int aligner_regulate_f32(const float *h, const int32_t *durations, float *out, int32_t *tok_out, int B, int C, int Tx,
                         int Ty, void *stream) {
    if (!durations || (!out && !tok_out) || (out && !h)) return fail(ALIGNER_EINVAL, "null pointer");
    if (B < 0 || C < 0 || Tx < 1 || Ty < 1) return fail(ALIGNER_EINVAL, "bad shape");
    if (B > 65535) return fail(ALIGNER_EDOM, "B=%d too large", B);
    if ((size_t)(Tx + 256) * sizeof(int) > 60 * 1024) return fail(ALIGNER_EDOM, "Tx=%d too large", Tx);
    if (B == 0) return ALIGNER_OK;
    // channel slices: enough workgroups to keep every CU's memory queue busy (>= ~8 per CU)
    int nz = 1;
    if (out) {
        const long wgs = (long)((Ty + 255) / 256) * B;
        while (nz < 16 && wgs * nz < 2048 && C / (nz * 2) >= 16) nz *= 2;
    }
    dim3 grid((Ty + 255) / 256, B, nz);
    hipLaunchKernelGGL(regulate_kernel, grid, dim3(256), (size_t)(Tx + 256) * sizeof(int),
                       static_cast<hipStream_t>(stream), h, durations, out, tok_out, C, Tx, Ty);
    ALIGNER_HIP_CHECK(hipGetLastError());
    return ALIGNER_OK;
}

}  // extern "C"
