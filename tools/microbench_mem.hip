// Development microbenchmark (not part of the product): HBM rate of the access patterns the
// soft-attention kernel can choose between, at the C2 shape [B=64, Tx=200, Ty=1000] fp32.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_mem.hip -o /tmp/mm && /tmp/mm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int B = 64, TX = 200, TY = 1000, C = 80;

// S1: the MFMA C/D-layout store of softattn_kernel: lane = frame, 16 regs x 7 tiles = rows; every
// wave-store writes 2 x 128 B (rows iu and iu+4), 4000 B apart from the next one.
__global__ __launch_bounds__(512) void store_cd_layout(float *out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int b = blockIdx.y, col = blockIdx.x * 256 + wave * 32 + (lane & 31);
    if (col >= TY) return;
    float *o = out + ((size_t)b * TX + 4 * half) * TY + col;
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int iu = 32 * r + (e & 3) + 8 * (e >> 2);
            if (iu + 4 * half < TX) o[(size_t)iu * TY] = v + e;
        }
}

// S2: same workgroup -> same output region, but each wave-store writes one 1-KB row segment
// (256 frames of one text row), 16 B per lane.
__global__ __launch_bounds__(512) void store_row_1k(float *out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, col = blockIdx.x * 256 + 4 * lane;
    if (col >= TY) return;
    for (int row = wave; row < TX; row += 8)
        *reinterpret_cast<float4 *>(out + ((size_t)b * TX + row) * TY + col) = make_float4(v, v + 1, v + 2, v + 3);
}

// S2b: as S2 but 512-B row segments (128 frames, 8 B per lane)
__global__ __launch_bounds__(512) void store_row_512(float *out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    for (int k = wave; k < 2 * TX; k += 8) {
        const int row = k >> 1, col = blockIdx.x * 256 + 128 * (k & 1) + 2 * lane;
        if (col < TY) *reinterpret_cast<float2 *>(out + ((size_t)b * TX + row) * TY + col) = make_float2(v, v + 1);
    }
}

// S3: fully linear streaming store (what expand_kernel does)
__global__ __launch_bounds__(256) void store_linear(float4 *out, float v, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        out[i] = make_float4(v, v, v, v);
}

// L1: the B-fragment load of softattn_kernel: lane = (frame, channel half), 40 strided dwords
__global__ __launch_bounds__(512) void load_frag_layout(const float *q, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int b = blockIdx.y, col = blockIdx.x * 256 + wave * 32 + (lane & 31);
    if (col >= TY) return;
    const float *Qb = q + (size_t)b * C * TY + col;
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) acc += Qb[(size_t)(16 * s + 8 * half + jj) * TY];
    if (acc == 12345.678f) sink[0] = acc;
}

// L2: the same [80 x 256] tile per workgroup read as 1-KB row segments, 16 B per lane
__global__ __launch_bounds__(512) void load_row_1k(const float *q, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, col = blockIdx.x * 256 + 4 * lane;
    if (col >= TY) return;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(q + ((size_t)b * C + wave + 8 * k) * TY + col);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <typename F>
static float time_us(F launch, int iters = 50) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0; hipEventElapsedTime(&ms, s, e);
    return ms * 1e3f / iters;
}

int main() {
    float *out, *q, *sink;
    const size_t n = (size_t)B * TX * TY;
    CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&q, (size_t)B * C * TY * 4)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(out, 0, n * 4)); CK(hipMemset(q, 0, (size_t)B * C * TY * 4));
    const double mb_out = n * 4 / 1e6, mb_q = (double)B * C * TY * 4 / 1e6;
    dim3 g(4, B);
    float t;
    t = time_us([&] { hipLaunchKernelGGL(store_cd_layout, g, dim3(512), 0, 0, out, 1.f); });
    printf("store C/D layout (2x128B per wave-store): %7.2f us  %7.1f GB/s\n", t, mb_out / t * 1e3);
    t = time_us([&] { hipLaunchKernelGGL(store_row_1k, g, dim3(512), 0, 0, out, 1.f); });
    printf("store 1-KB row segments (16 B/lane):      %7.2f us  %7.1f GB/s\n", t, mb_out / t * 1e3);
    t = time_us([&] { hipLaunchKernelGGL(store_row_512, g, dim3(512), 0, 0, out, 1.f); });
    printf("store 512-B row segments (8 B/lane):      %7.2f us  %7.1f GB/s\n", t, mb_out / t * 1e3);
    t = time_us([&] { hipLaunchKernelGGL(store_linear, dim3(2048), dim3(256), 0, 0, (float4 *)out, 1.f, n / 4); });
    printf("store linear:                             %7.2f us  %7.1f GB/s\n", t, mb_out / t * 1e3);
    t = time_us([&] { hipLaunchKernelGGL(load_frag_layout, g, dim3(512), 0, 0, q, sink); });
    printf("load B-fragment layout (40 dwords/lane):  %7.2f us  %7.1f GB/s\n", t, mb_q / t * 1e3);
    t = time_us([&] { hipLaunchKernelGGL(load_row_1k, g, dim3(512), 0, 0, q, sink); });
    printf("load 1-KB row segments (16 B/lane):       %7.2f us  %7.1f GB/s\n", t, mb_q / t * 1e3);
    return 0;
}
