// Cycles per frame of the forward-sum recurrence on one wave (one row per lane), two forms:
//   LOG: alpha' = logaddexp2(alpha, alpha[lane-1]) + lp                      (sub, exp2, add, log2, max, add + dpp)
//   LIN: (M, s) per lane, s*2^M: align the neighbour by ldexp, add, multiply by the cell's (Mp, sp)
// hipcc --offload-arch=gfx950 -O3 tools/microbench_fwdsum.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float dpp_below(float edge, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int dpp_below_i(int edge, int src) { return __builtin_amdgcn_update_dpp(edge, src, 0x138, 0xf, 0xf, false); }

__device__ __forceinline__ float lae2(float a, float b) {
    float m;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a - b)));
}
__device__ __forceinline__ float lae3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m));
}
// the CTC form's frame: two states a row.  MODE 2: B' = blank + lse(B, T^), T' = x + lse3(T, B, T^) (six transcendentals, two
// independent chains); MODE 3: L = lse(B, T^), B' = blank + L, T' = x + lse(T, L) (four, one longer chain)
template <int MODE>
__global__ void kctc(const float *lp, float *out, long long *cyc, int T) {
    const int lane = threadIdx.x;
    float pT = -1e30f, pB = lane == 0 ? 0.f : -1e30f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int y0 = 0; y0 < T; y0 += 16) {
        float l[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) l[c] = lp[(y0 + c) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float up = dpp_below(-1e30f, pT);
            if (MODE == 2) {
                const float vb = lae2(pB, up) - 1.5f;
                const float vt = lae3(pT, pB, up) + l[c];
                pB = vb; pT = vt;
            } else {
                const float L = lae2(pB, up);
                pT = lae2(pT, L) + l[c];
                pB = L - 1.5f;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = pT + pB;
    if (lane == 0) cyc[0] = t1 - t0;
}

template <int MODE>
__global__ void k(const float *lp, float *out, long long *cyc, int T, int extra) {
    const int lane = threadIdx.x;
    float a = lane == 0 ? 0.f : -1e30f;
    float s = lane == 0 ? 1.f : 0.f;
    int M = lane == 0 ? 0 : -(1 << 28);
    float acc = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int y0 = 0; y0 < T; y0 += 16) {
        float l[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) l[c] = lp[(y0 + c) * 64 + lane];
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float up = dpp_below(-1e30f, a);
                float m;
                asm("v_max_f32_e32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(up));
                a = m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a - up))) + l[c];
                if (extra) acc += a;
            }
        } else {
            float sp[16]; int mp[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float lc = fmaxf(l[c], -1048576.f);
                const float cl = ceilf(lc);
                sp[c] = __builtin_amdgcn_exp2f(lc - cl);
                mp[c] = (int)cl;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float us = dpp_below(0.f, s);
                const int uM = dpp_below_i(-(1 << 28), M);
                const int Mn = M > uM ? M : uM;
                float v = __builtin_ldexpf(s, M - Mn) + __builtin_ldexpf(us, uM - Mn);
                s = v * sp[c];
                int Mv = Mn + mp[c];
                M = Mv > -(1 << 29) ? Mv : -(1 << 29);
                if (extra) acc += (float)M + __builtin_amdgcn_logf(s);      // the stored log form
                if ((c & 7) == 7) { M += __builtin_amdgcn_frexp_expf(s); s = __builtin_amdgcn_frexp_mantf(s); }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[lane] = MODE == 0 ? a + acc : (float)M + __builtin_amdgcn_logf(s) + acc;
    if (lane == 0) cyc[0] = t1 - t0;
}

int main() {
    const int T = 1024;
    std::vector<float> h(T * 64);
    unsigned z = 12345;
    for (auto &x : h) { z = z * 1664525u + 1013904223u; x = -8.f * ((z >> 8) & 0xffff) / 65536.f; }
    float *lp, *out; long long *cyc;
    hipMalloc(&lp, h.size() * 4); hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    hipMemcpy(lp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float o0[64], o1[64]; long long c;
    for (int extra = 0; extra < 2; ++extra) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, lp, out, cyc, T, extra); }
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o0, out, 256, hipMemcpyDeviceToHost);
        printf("LOG form%s: %.1f cycles/frame   out[5] = %f\n", extra ? " + store value" : "", (double)c / T, o0[5]);
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, lp, out, cyc, T, extra); }
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o1, out, 256, hipMemcpyDeviceToHost);
        printf("LIN form%s: %.1f cycles/frame   out[5] = %f\n", extra ? " + store value" : "", (double)c / T, o1[5]);
    }
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kctc<2>, dim3(1), dim3(64), 0, 0, lp, out, cyc, T);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o0, out, 256, hipMemcpyDeviceToHost);
    printf("CTC frame, lse3 + lse2 (6 transcendentals): %.1f cycles/frame   out[5] = %f\n", (double)c / T, o0[5]);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kctc<3>, dim3(1), dim3(64), 0, 0, lp, out, cyc, T);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(o1, out, 256, hipMemcpyDeviceToHost);
    printf("CTC frame, shared lse (4 transcendentals):  %.1f cycles/frame   out[5] = %f\n", (double)c / T, o1[5]);
    return 0;
}
