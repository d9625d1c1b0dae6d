// Development microbenchmark (not part of the product): what an LDS instruction costs the wave that issues it,
// interleaved with a block of dependent VALU work (the shape of the DP's tile loop).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_lds.hip -o /tmp/mbl && /tmp/mbl
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VALU32 \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t" \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t" \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t" \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t" \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t" \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t" \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t" \
    "v_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\tv_add_f32 %[a], %[a], %[a]\n\t"
#define R8 \
    "ds_read_b128 %[r0], %[ad]\n\tds_read_b128 %[r1], %[ad] offset:16\n\tds_read_b128 %[r2], %[ad] offset:32\n\tds_read_b128 %[r3], %[ad] offset:48\n\t" \
    "ds_read_b128 %[r4], %[ad] offset:64\n\tds_read_b128 %[r5], %[ad] offset:80\n\tds_read_b128 %[r6], %[ad] offset:96\n\tds_read_b128 %[r7], %[ad] offset:112\n\t"
#define W8_128 \
    "ds_write_b128 %[ad], %[r0]\n\tds_write_b128 %[ad], %[r1] offset:16\n\tds_write_b128 %[ad], %[r2] offset:32\n\tds_write_b128 %[ad], %[r3] offset:48\n\t" \
    "ds_write_b128 %[ad], %[r4] offset:64\n\tds_write_b128 %[ad], %[r5] offset:80\n\tds_write_b128 %[ad], %[r6] offset:96\n\tds_write_b128 %[ad], %[r7] offset:112\n\t"
#define W16_2B32 \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:0 offset1:1\n\tds_write2_b32 %[ad], %[a], %[a] offset0:2 offset1:3\n\t" \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:4 offset1:5\n\tds_write2_b32 %[ad], %[a], %[a] offset0:6 offset1:7\n\t" \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:8 offset1:9\n\tds_write2_b32 %[ad], %[a], %[a] offset0:10 offset1:11\n\t" \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:12 offset1:13\n\tds_write2_b32 %[ad], %[a], %[a] offset0:14 offset1:15\n\t" \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:16 offset1:17\n\tds_write2_b32 %[ad], %[a], %[a] offset0:18 offset1:19\n\t" \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:20 offset1:21\n\tds_write2_b32 %[ad], %[a], %[a] offset0:22 offset1:23\n\t" \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:24 offset1:25\n\tds_write2_b32 %[ad], %[a], %[a] offset0:26 offset1:27\n\t" \
    "ds_write2_b32 %[ad], %[a], %[a] offset0:28 offset1:29\n\tds_write2_b32 %[ad], %[a], %[a] offset0:30 offset1:31\n\t"
#define W8_B32 \
    "ds_write_b32 %[ad], %[a]\n\tds_write_b32 %[ad], %[a] offset:4\n\tds_write_b32 %[ad], %[a] offset:8\n\tds_write_b32 %[ad], %[a] offset:12\n\t" \
    "ds_write_b32 %[ad], %[a] offset:16\n\tds_write_b32 %[ad], %[a] offset:20\n\tds_write_b32 %[ad], %[a] offset:24\n\tds_write_b32 %[ad], %[a] offset:28\n\t"
#define OPS : [a] "+v"(a), [r0] "+v"(r[0]), [r1] "+v"(r[1]), [r2] "+v"(r[2]), [r3] "+v"(r[3]), [r4] "+v"(r[4]), [r5] "+v"(r[5]), [r6] "+v"(r[6]), [r7] "+v"(r[7]) : [ad] "v"(ad) : "memory"

template <int MODE>
__global__ void bench(float *out, long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = 1.0f + lane * 1e-6f;
    f32x4 r[8];
    for (int i = 0; i < 8; ++i) r[i] = (f32x4){a, a, a, a};
    unsigned ad = wave * 64 * 144 + lane * 144;           // the DP's padded tile rows
    if (MODE == 5 || MODE == 6) ad = wave * 1024;         // one row, as the ring
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < iters; ++t) {
        if (MODE == 0) asm volatile(VALU32 VALU32 VALU32 VALU32 OPS);                                   // 128 VALU
        if (MODE == 1) asm volatile(R8 VALU32 VALU32 VALU32 VALU32 "s_waitcnt lgkmcnt(0)\n\t" OPS);     // + 8 reads, waited for after the VALU
        if (MODE == 2) asm volatile(R8 "s_waitcnt lgkmcnt(0)\n\t" VALU32 VALU32 VALU32 VALU32 OPS);     // + 8 reads, waited for at once
        if (MODE == 3) asm volatile(VALU32 VALU32 VALU32 VALU32 W8_128 OPS);                            // + 8 ds_write_b128
        if (MODE == 4) asm volatile(VALU32 VALU32 VALU32 VALU32 W8_B32 OPS);                            // + 8 ds_write_b32
        if (MODE == 5) asm volatile(VALU32 VALU32 VALU32 VALU32 "s_mov_b64 exec, 1\n\t" W16_2B32 "s_mov_b64 exec, -1\n\t" OPS);  // + 16 one-lane ds_write2_b32
        if (MODE == 6) asm volatile(VALU32 VALU32 VALU32 VALU32 "s_mov_b64 exec, 1\n\t" W8_128 "s_mov_b64 exec, -1\n\t" OPS);    // + 8 one-lane ds_write_b128
        if (MODE == 7) asm volatile(VALU32 R8 VALU32 VALU32 VALU32 "s_waitcnt lgkmcnt(0)\n\t" W8_128 OPS);  // reads + writes
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + r[0].x + r[7].w;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

int main() {
    float *d_out; long long *d_cyc;
    CK(hipMalloc(&d_out, 1 << 22)); CK(hipMalloc(&d_cyc, 1 << 16));
    const int iters = 2000;
    const char *names[8] = {"128 VALU", "+8 ds_read_b128 (wait after VALU)", "+8 ds_read_b128 (wait at once)", "+8 ds_write_b128", "+8 ds_write_b32",
                            "+16 one-lane ds_write2_b32", "+8 one-lane ds_write_b128", "+8 reads +8 ds_write_b128"};
    const int wv[3] = {1, 4, 8};
    for (int wi = 0; wi < 3; ++wi) {
        const int waves = wv[wi];
        for (int mode = 0; mode < 8; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
#define L(M) if (mode == M) { CK(hipFuncSetAttribute((const void *)bench<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); hipLaunchKernelGGL(bench<M>, dim3(64), dim3(64 * waves), 80 * 1024, 0, d_out, d_cyc, iters); }
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7)
                CK(hipDeviceSynchronize());
            }
            long long c[16];
            CK(hipMemcpy(c, d_cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
            long long mx = 0; for (int i = 0; i < waves; ++i) mx = c[i] > mx ? c[i] : mx;
            printf("waves/block=%2d %-36s cycles/iteration = %.1f (slowest wave %.1f)\n", waves, names[mode], (double)c[0] / iters, (double)mx / iters);
        }
    }
    return 0;
}
