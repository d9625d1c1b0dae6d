// Development microbenchmark (not part of the product): what one s_barrier costs the waves of a small workgroup on
// gfx950 -- alone, with global loads in flight, with global stores in flight (the boundary search's chain kernels
// cross one barrier per token row with the next row's operand loads and this row's result stores outstanding).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_barrier.hip -o /tmp/mbb && /tmp/mbb
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: VALU work + barrier.  1: + 4 global loads issued before the barrier, consumed after the next one.
// 2: + 3 global stores (sc1) before the barrier.  3: loads and stores.  4: as 3 without the barrier (the control).
template <int MODE>
__global__ void bench(float *buf, long long *cyc, int iters, int stride) {
    __shared__ float lds[1024];
    const int tid = threadIdx.x;
    float a = (float)tid, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    float *p = buf + (size_t)blockIdx.x * 65536 + tid;
    long long t0 = 0, wait = 0;
    lds[tid] = a;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        const float c0 = l0, c1 = l1, c2 = l2, c3 = l3;
        if (MODE == 1 || MODE >= 3) {
            l0 = __builtin_nontemporal_load(p + (size_t)(i & 15) * stride);
            l1 = __builtin_nontemporal_load(p + (size_t)(i & 15) * stride + 1024);
            l2 = __builtin_nontemporal_load(p + (size_t)(i & 15) * stride + 2048);
            l3 = __builtin_nontemporal_load(p + (size_t)(i & 15) * stride + 3072);
        }
#pragma unroll
        for (int k = 0; k < 64; ++k) a = a * 1.0001f + c0;
        a += c1 + c2 + c3;
        lds[tid] = a;
        if (MODE == 2 || MODE >= 3) {
            __hip_atomic_store(reinterpret_cast<unsigned *>(p) + 8192, __builtin_bit_cast(unsigned, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned *>(p) + 9216, __builtin_bit_cast(unsigned, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p[10240] = a;
        }
        const long long tb = __builtin_amdgcn_s_memtime();
        if (MODE != 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait += __builtin_amdgcn_s_memtime() - tb;
        a += lds[(tid + 1) & 255];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0) {
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 2] = (t1 - t0) / iters;
        cyc[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = wait / iters;
    }
    buf[(size_t)blockIdx.x * 65536 + 20000 + tid] = a;
}

int main() {
    float *buf; long long *cyc;
    const int blocks = 64, iters = 2000;
    CK(hipMalloc(&buf, (size_t)blocks * 65536 * 4 + (1 << 20)));
    CK(hipMemset(buf, 0, (size_t)blocks * 65536 * 4 + (1 << 20)));
    CK(hipMalloc(&cyc, blocks * 8 * sizeof(long long)));
    long long h[blocks * 8];
    const char *names[5] = {"VALU + barrier", "+ 4 loads in flight", "+ 3 stores in flight", "+ loads and stores", "loads and stores, NO barrier"};
    for (int m = 0; m < 5; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
                case 0: hipLaunchKernelGGL(bench<0>, dim3(blocks), dim3(256), 0, 0, buf, cyc, iters, 256); break;
                case 1: hipLaunchKernelGGL(bench<1>, dim3(blocks), dim3(256), 0, 0, buf, cyc, iters, 256); break;
                case 2: hipLaunchKernelGGL(bench<2>, dim3(blocks), dim3(256), 0, 0, buf, cyc, iters, 256); break;
                case 3: hipLaunchKernelGGL(bench<3>, dim3(blocks), dim3(256), 0, 0, buf, cyc, iters, 256); break;
                case 4: hipLaunchKernelGGL(bench<4>, dim3(blocks), dim3(256), 0, 0, buf, cyc, iters, 256); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-32s per iteration %5lld cycles, of which around the barrier %5lld   (waves of block 0: %lld %lld %lld %lld)\n", names[m], h[0], h[1],
               h[1], h[3], h[5], h[7]);
    }
    return 0;
}
