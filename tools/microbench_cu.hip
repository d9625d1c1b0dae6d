// Development microbenchmark (not part of the product): how fast can ONE CU pull a [252 rows x Ty] fp32
// band (what one utterance's DP reads), by access pattern and number of loader waves?  64 workgroups
// (one per CU, like the DP kernel at B = 64), each streaming its own 800 KB.
// Build+run:  hipcc --offload-arch=gfx950 -O3 tools/microbench_cu.hip -o tools/bin/mcu && tools/bin/mcu
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int B = 64, TX = 200, TY = 1000;

// pattern 0: wave-instruction = 8 rows x 128 B (the DP loader's); tiles of 32 frames, DEPTH tiles in flight
// pattern 1: wave-instruction = 2 rows x 512 B
// pattern 2: wave-instruction = 1 row  x 1 KB
template <int PAT, int NWV, int DEPTH>
__global__ __launch_bounds__(NWV * 64) void stream(const float *v, float *sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x;
    const float *ub = v + (size_t)b * TX * TY;
    float acc = 0.f;
    constexpr int RW = 252 / NWV;                  // rows per wave (63 for 4 waves)
    if (PAT == 0) {
        const int rr = lane >> 3, cg = lane & 7;
        float4 buf[DEPTH][8];
        auto issue = [&](float4 (&d)[8], int t) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int r = RW * w + 8 * k + rr; r = r > TX - 1 ? TX - 1 : r;
                int c = 32 * t + 4 * cg; c = c > TY - 4 ? TY - 4 : c;
                d[k] = *reinterpret_cast<const float4 *>(ub + (size_t)r * TY + c);
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(buf[d], d);
        for (int t0 = 0; t0 < 32; t0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += buf[d][k].x + buf[d][k].w;
                issue(buf[d], t0 + d + DEPTH < 32 ? t0 + d + DEPTH : 31);
            }
        }
    } else {
        constexpr int RPI = PAT == 1 ? 2 : 1;      // rows per instruction
        constexpr int LPR = 64 / RPI;              // lanes per row -> LPR*16 B contiguous
        const int rsel = lane / LPR, cl = lane % LPR;
        constexpr int NCH = (TY * 4 + LPR * 16 - 1) / (LPR * 16);   // column chunks per row
        constexpr int NIT = (RW / RPI) * NCH;      // instructions per wave
        float4 buf[DEPTH * 8];
        auto addr = [&](int it) {
            const int rg = it / NCH, ch = it % NCH;
            int r = RW * w + rg * RPI + rsel; r = r > TX - 1 ? TX - 1 : r;
            int c = ch * LPR * 4 + cl * 4; c = c > TY - 4 ? TY - 4 : c;
            return ub + (size_t)r * TY + c;
        };
#pragma unroll
        for (int i = 0; i < DEPTH * 8; ++i) buf[i] = *reinterpret_cast<const float4 *>(addr(i));
        for (int it0 = 0; it0 < NIT; it0 += DEPTH * 8) {
#pragma unroll
            for (int i = 0; i < DEPTH * 8; ++i) {
                acc += buf[i].x + buf[i].w;
                const int nx = it0 + i + DEPTH * 8;
                buf[i] = *reinterpret_cast<const float4 *>(addr(nx < NIT ? nx : NIT - 1));
            }
        }
    }
    if (acc == 1234.5f) sink[0] = acc;
}

// pattern 3: LDS-DMA (global_load_lds_dwordx4): 8 rows x 128 B per wave-instruction straight into LDS,
// INFL instructions in flight per wave, no consumer
template <int NWV, int INFL>
__global__ __launch_bounds__(NWV * 64) void stream_dma(const float *v, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x;
    const float *ub = v + (size_t)b * TX * TY;
    constexpr int RW = 252 / NWV;
    const int rr = lane >> 3, cg = lane & 7;
    constexpr int NIT = 32 * 8;                   // 32 tiles x 8 row groups
    int slot = 0;
    for (int it = 0; it < NIT; ++it) {
        const int t = it >> 3, k = it & 7;
        int r = RW * w + 8 * k + rr; r = r > TX - 1 ? TX - 1 : r;
        int c = 32 * t + 4 * cg; c = c > TY - 4 ? TY - 4 : c;
        const float *src = ub + (size_t)r * TY + c;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(smem + (w * INFL + slot) * 1024), 16, 0, 0);
        slot = (slot + 1 == INFL) ? 0 : slot + 1;
        if (it >= INFL - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFL - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<float *>(smem)[threadIdx.x] == 1234.5f) sink[0] = 1.f;
}

template <typename F>
static float time_us(F launch, int iters = 30) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    return ms * 1e3f / iters;
}

int main() {
    float *v, *sink;
    const size_t n = (size_t)B * TX * TY;
    CK(hipMalloc(&v, n * 4 + 4096)); CK(hipMalloc(&sink, 256)); CK(hipMemset(v, 0, n * 4 + 4096));
    const double kb = 200.0 * TY * 4 / 1e3;     // useful KB per CU
#define RUN(PAT, NWV, DEPTH) { float t = time_us([&] { hipLaunchKernelGGL((stream<PAT, NWV, DEPTH>), dim3(B), dim3(NWV * 64), 0, 0, v, sink); }); \
    printf("pattern %d  waves %d  depth %d: %7.2f us  -> %6.1f GB/s per CU\n", PAT, NWV, DEPTH, t, kb / t * 1e-3 * 1e3); }
    RUN(0, 4, 4) RUN(0, 4, 8) RUN(1, 4, 4) RUN(2, 4, 4) RUN(2, 4, 2)
    RUN(0, 6, 4) RUN(2, 6, 4) RUN(2, 12, 2)
#define RUND(NWV, INFL) { hipFuncSetAttribute(reinterpret_cast<const void *>(stream_dma<NWV, INFL>), hipFuncAttributeMaxDynamicSharedMemorySize, NWV * INFL * 1024); \
    float t = time_us([&] { hipLaunchKernelGGL((stream_dma<NWV, INFL>), dim3(B), dim3(NWV * 64), NWV * INFL * 1024, 0, v, sink); }); \
    printf("LDS-DMA  waves %d  in flight %d KB/wave: %7.2f us  -> %6.1f GB/s per CU\n", NWV, INFL, t, kb / t * 1e-3 * 1e3); }
    RUND(4, 8) RUND(4, 16) RUND(4, 32) RUND(8, 16)
    return 0;
}
