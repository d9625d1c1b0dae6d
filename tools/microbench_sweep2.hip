// Development microbenchmark (not part of the product): cycles per frame of the in-kernel sweep statements
// (aligner_amd/csrc/maxpath_sweep_asm.inc) against VCC-form variants, by waves per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -Ialigner_amd/csrc tools/microbench_sweep2.hip -o /tmp/mb2 && /tmp/mb2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "maxpath_sweep_asm.inc"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OPERANDS                                                                                        \
    : [qa] "+v"(q), [qb] "=&v"(qb), [m] "+v"(m), [bits] "+v"(bits), [coll] "+v"(coll), [cur] "=&v"(cur), \
      [sA] "=&s"(sA), [sB] "=&s"(sB), [sM0] "=&s"(sM0), [sM1] "=&s"(sM1), [sk0] "=&s"(sk0), [sk1] "=&s"(sk1) \
    : [rrel] "v"(rrel), [neg] "v"(negv),                                                                 \
      [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]),   \
      [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]), [v11] "v"(v[11]), \
      [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15])                             \
    : "vcc"

// VCC forms: cmp -> vcc, addc <- vcc (e32 encodings)
#define F_VCC(V) \
    "v_max_f32_dpp %[m], %[qa], %[qa] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_lg_f32 vcc, %[m], %[qa]\n\t" \
    "v_add_f32_e32 %[qa], %[m], " V "\n\t" \
    "v_addc_co_u32_e32 %[bits], vcc, %[bits], %[bits], vcc\n\t"
// chain only (max_dpp + add + 2 wait states)
#define F_CHAIN(V) \
    "v_max_f32_dpp %[m], %[qa], %[qa] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_e32 %[qa], %[m], " V "\n\t" \
    "s_nop 1\n\t"
// chain with plain VALU fillers instead of nops
#define F_CHAIN_F(V) \
    "v_max_f32_dpp %[m], %[qa], %[qa] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_e32 %[qa], %[m], " V "\n\t" \
    "v_add_u32_e32 %[bits], %[bits], %[bits]\n\t" \
    "v_add_u32_e32 %[cur], %[bits], %[bits]\n\t"
// no DPP at all (plain max): what does the DPP cost?
#define F_NODPP(V) \
    "v_max_f32_e32 %[m], %[qa], %[qa]\n\t" \
    "v_cmp_lg_f32 vcc, %[m], %[qa]\n\t" \
    "v_add_f32_e32 %[qa], %[m], " V "\n\t" \
    "v_addc_co_u32_e32 %[bits], vcc, %[bits], %[bits], vcc\n\t"
// e64 cmp into an SGPR pair + e64 addc from it, same frame (hazard covered by one instruction + nop)
#define F_E64(V) \
    "v_max_f32_dpp %[m], %[qa], %[qa] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "v_cmp_lg_f32_e64 %[sA], %[m], %[qa]\n\t" \
    "v_add_f32_e32 %[qa], %[m], " V "\n\t" \
    "s_nop 0\n\t" \
    "v_addc_co_u32_e64 %[bits], vcc, %[bits], %[bits], %[sA]\n\t"
#define X16(F) F("%[v0]") F("%[v1]") F("%[v2]") F("%[v3]") F("%[v4]") F("%[v5]") F("%[v6]") F("%[v7]") \
               F("%[v8]") F("%[v9]") F("%[v10]") F("%[v11]") F("%[v12]") F("%[v13]") F("%[v14]") F("%[v15]")

template <int MODE>
__global__ void bench(float *out, long long *cyc, int ntiles) {
    const int lane = threadIdx.x & 63;
    float q = -1.0f * lane, m = 0.0f, qb, cur, negv = -1e9f;
    unsigned bits = 0;
    int coll = 0, rrel = lane - 1000;
    unsigned long long sA, sB, sM0, sM1;
    int sk0, sk1;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = -0.5f - lane * 0.01f * i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MODE == 0) asm volatile(ALIGNER_SWEEP16_NOPUB_0 OPERANDS);
            if (MODE == 1) asm volatile(ALIGNER_SWEEP16_PUB_0 OPERANDS);
            if (MODE == 2) asm volatile(X16(F_VCC) OPERANDS);
            if (MODE == 3) asm volatile(X16(F_CHAIN) OPERANDS);
            if (MODE == 4) asm volatile(X16(F_CHAIN_F) OPERANDS);
            if (MODE == 5) asm volatile(X16(F_NODPP) OPERANDS);
            if (MODE == 6) asm volatile(X16(F_E64) OPERANDS);
            if (MODE == 7) asm volatile(ALIGNER_SWEEP16_DIAG_PUB_0 OPERANDS);
        }
        v[0] += 1e-3f;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = q + (float)bits + m + (float)coll;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float *d_out; long long *d_cyc;
    CK(hipMalloc(&d_out, 1 << 22)); CK(hipMalloc(&d_cyc, 1 << 16));
    const int ntiles = 2000;
    const char *names[8] = {"kernel NOPUB (current .inc)   ", "kernel PUB", "vcc forms", "chain only + s_nop 1", "chain + 2 plain fillers",
                            "vcc forms, no DPP", "e64 same frame", "kernel DIAG_PUB"};
    const int wv[4] = {1, 4, 8, 16};
    for (int wi = 0; wi < 4; ++wi) {
        const int waves = wv[wi];
        for (int mode = 0; mode < 8; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
#define L(M) if (mode == M) hipLaunchKernelGGL(bench<M>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
                CK(hipDeviceSynchronize());
            }
            long long c[16];
            CK(hipMemcpy(c, d_cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
            long long mx = 0; for (int i = 0; i < waves; ++i) mx = c[i] > mx ? c[i] : mx;
            printf("waves/block=%2d %-32s cycles/frame = %.2f (slowest wave %.2f)\n", waves, names[mode], (double)c[0] / (ntiles * 32.0), (double)mx / (ntiles * 32.0));
        }
    }
    return 0;
}
