// Development microbenchmark (not part of the product): DPP lane-0 semantics and
// cycles per frame of candidate inner loops for the alignment forward sweep.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_sweep.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---- T1: what do DPP ops do on lane 0 with bound_ctrl off? ----
__global__ void dpp_semantics(float *out) {
    const int lane = threadIdx.x;
    float q = 10.0f + lane;            // lane l holds 10+l
    float m = -777.0f;                 // sentinel: does lane 0 keep it?
    float mx = -555.0f;
    unsigned long long vccv = 0;
    unsigned bits = 0x0;
    asm volatile(
        "s_nop 4\n\t"
        "v_mov_b32_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_cmp_ngt_f32 vcc, %1, %4\n\t"
        "s_nop 4\n\t"
        "s_mov_b64 %2, vcc\n\t"
        "v_cndmask_b32_dpp %0, %4, %4, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %3, vcc\n\t"
        "s_nop 4\n\t"
        : "+v"(m), "+v"(mx), "=s"(vccv), "+v"(bits)
        : "v"(q)
        : "vcc");
    out[lane] = m;
    out[64 + lane] = mx;
    out[128 + lane] = (float)bits;
    if (lane == 0) { out[192] = (float)(unsigned)(vccv & 0xffffffffu); out[193] = (float)(unsigned)(vccv >> 32); }
}

// ---- T2: timing of inner loops.  One "column" = one DP frame for 64 rows. ----
// exact: cmp_dpp / cndmask_dpp / addc / add   (+ optional LDS publish)
#define COL_EXACT(VAL, PUB)                                                                     \
    "v_mov_b32_dpp %[up], %[q] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                       \
    PUB                                                                                         \
    "v_cmp_ngt_f32 vcc, %[up], %[q]\n\t"                                                        \
    "s_nop 1\n\t"                                                                               \
    "v_cndmask_b32_dpp %[m], %[q], %[q], vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"         \
    "v_addc_co_u32_e32 %[bits], vcc, %[bits], %[bits], vcc\n\t"                                 \
    "v_add_f32_e32 %[q], %[m], " VAL "\n\t"                                                     \
    "s_nop 1\n\t"

#define COL_FAST(VAL, PUB)                                                                      \
    "v_max_f32_dpp %[m], %[q], %[q] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_cmp_lg_f32 vcc, %[m], %[q]\n\t"                                                          \
    "v_add_f32_e32 %[q], %[m], " VAL "\n\t"                                                     \
    PUB                                                                                         \
    "v_addc_co_u32_e32 %[bits], vcc, %[bits], %[bits], vcc\n\t"

#define PUB_NONE ""
#define PUB_LDS "ds_write_b32 %[pa], %[q]\n\t"

// publish with EXEC narrowed to lane 63 (SALU exec moves sit in the DPP wait-state slots)
#define PUB_LDS_X "s_mov_b64 exec, %[l63]\n\tds_write_b32 %[pa], %[q]\n\ts_mov_b64 exec, -1\n\t"
// collect lane 63's q into lane k of `coll` (readlane -> writelane), one LDS write per tile
#define PUB_RL(K) "v_readlane_b32 %[sk], %[q], 63\n\t" "s_nop 1\n\t" "v_writelane_b32 %[coll], %[sk], " #K "\n\t"
#define COL_FAST_RL(VAL, K)                                                                     \
    "v_max_f32_dpp %[m], %[q], %[q] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_cmp_lg_f32 vcc, %[m], %[q]\n\t"                                                          \
    "v_add_f32_e32 %[q], %[m], " VAL "\n\t"                                                     \
    "v_addc_co_u32_e32 %[bits], vcc, %[bits], %[bits], vcc\n\t"                                 \
    "v_readlane_b32 %[sk], %[q], 63\n\t"                                                        \
    "v_writelane_b32 %[coll], %[sk], " #K "\n\t"
// copy q aside, one ds_write_b128 per 4 frames
#define COL_FAST_MV(VAL, TMP)                                                                   \
    "v_max_f32_dpp %[m], %[q], %[q] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"                  \
    "v_cmp_lg_f32 vcc, %[m], %[q]\n\t"                                                          \
    "v_add_f32_e32 %[q], %[m], " VAL "\n\t"                                                     \
    "v_addc_co_u32_e32 %[bits], vcc, %[bits], %[bits], vcc\n\t"                                 \
    "v_mov_b32_e32 " TMP ", %[q]\n\t"

#define FOUR(M, PUB) M("%[v0]", PUB) M("%[v1]", PUB) M("%[v2]", PUB) M("%[v3]", PUB)

template <int MODE>
__global__ void sweep_bench(float *out, long long *cyc, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 8 * 4 + 256];
    const unsigned long long l63 = 1ull << 63;
    float coll = 0.f;
    const int lane = threadIdx.x & 63;
    float q = -1.0f * lane, m = 0.0f, up = 0.0f;
    unsigned bits = 0;
    float v0 = -0.5f - lane * 0.01f, v1 = -0.25f, v2 = -0.75f + lane * 0.02f, v3 = -0.125f;
    unsigned pa = (unsigned)(size_t)(&lds[0]) + (threadIdx.x) * 4;   // per-thread LDS address (generic->lds low bits)
    pa = (threadIdx.x) * 4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == 0)
                asm volatile(FOUR(COL_EXACT, PUB_NONE) : [q] "+v"(q), [m] "+v"(m), [bits] "+v"(bits), [up] "+v"(up)
                             : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [pa] "v"(pa) : "vcc");
            else if (MODE == 1)
                asm volatile(FOUR(COL_EXACT, PUB_LDS) : [q] "+v"(q), [m] "+v"(m), [bits] "+v"(bits), [up] "+v"(up)
                             : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [pa] "v"(pa) : "vcc", "memory");
            else if (MODE == 2)
                asm volatile(FOUR(COL_FAST, PUB_NONE) : [q] "+v"(q), [m] "+v"(m), [bits] "+v"(bits), [up] "+v"(up)
                             : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [pa] "v"(pa) : "vcc");
            else if (MODE == 4)
                asm volatile(FOUR(COL_FAST, PUB_LDS_X) : [q] "+v"(q), [m] "+v"(m), [bits] "+v"(bits), [up] "+v"(up)
                             : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [pa] "v"(pa), [l63] "s"(l63) : "vcc", "memory");
            else if (MODE == 5) {
                unsigned sk;
                asm volatile(COL_FAST_RL("%[v0]", 0) COL_FAST_RL("%[v1]", 1) COL_FAST_RL("%[v2]", 2) COL_FAST_RL("%[v3]", 3)
                             : [q] "+v"(q), [m] "+v"(m), [bits] "+v"(bits), [coll] "+v"(coll), [sk] "=&s"(sk)
                             : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3) : "vcc");
                if (g == 7) asm volatile("ds_write_b32 %0, %1" :: "v"(pa), "v"(coll) : "memory");
            } else if (MODE == 6) {
                float4 tq;
                asm volatile(COL_FAST_MV("%[v0]", "%[t0]") COL_FAST_MV("%[v1]", "%[t1]") COL_FAST_MV("%[v2]", "%[t2]") COL_FAST_MV("%[v3]", "%[t3]")
                             : [q] "+v"(q), [m] "+v"(m), [bits] "+v"(bits), [t0] "=&v"(tq.x), [t1] "=&v"(tq.y), [t2] "=&v"(tq.z), [t3] "=&v"(tq.w)
                             : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3) : "vcc");
                *reinterpret_cast<float4 *>(&lds[threadIdx.x * 4]) = tq;
            }
            else if (MODE == 3)
                asm volatile(FOUR(COL_FAST, PUB_LDS) : [q] "+v"(q), [m] "+v"(m), [bits] "+v"(bits), [up] "+v"(up)
                             : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [pa] "v"(pa) : "vcc", "memory");
        }
        v0 += 1e-3f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = q + (float)bits + m + up + coll + lds[lane];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float *d_out; long long *d_cyc;
    CK(hipMalloc(&d_out, 1 << 20)); CK(hipMalloc(&d_cyc, 1 << 16));
    // T1
    hipLaunchKernelGGL(dpp_semantics, dim3(1), dim3(64), 0, 0, d_out);
    CK(hipDeviceSynchronize());
    std::vector<float> h(256);
    CK(hipMemcpy(h.data(), d_out, 256 * 4, hipMemcpyDeviceToHost));
    printf("T1 q[l]=10+l.  cndmask_dpp m: lane0=%g lane1=%g lane2=%g lane63=%g  (sentinel -777 kept on lane0?)\n", h[0], h[1], h[2], h[63]);
    printf("T1 max_dpp   mx: lane0=%g lane1=%g lane63=%g (sentinel -555)\n", h[64], h[65], h[127]);
    printf("T1 bits after addc: lane0=%g lane1=%g ; vcc lo=0x%08x hi=0x%08x (bit=!(q[l-1]>q[l]) -> expect all ones; lane0?)\n",
           h[128], h[129], (unsigned)h[192], (unsigned)h[193]);
    // T2
    const int ntiles = 2000;
    const char *names[7] = {"exact", "exact+lds_publish", "fast(max)", "fast(max)+lds_publish", "fast+publish exec=lane63", "fast+readlane/writelane", "fast+mov+ds_write_b128/4"};
    for (int waves = 1; waves <= 8; waves *= 8) {
        for (int mode = 0; mode < 7; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(sweep_bench<0>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                if (mode == 1) hipLaunchKernelGGL(sweep_bench<1>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                if (mode == 2) hipLaunchKernelGGL(sweep_bench<2>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                if (mode == 3) hipLaunchKernelGGL(sweep_bench<3>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                if (mode == 4) hipLaunchKernelGGL(sweep_bench<4>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                if (mode == 5) hipLaunchKernelGGL(sweep_bench<5>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                if (mode == 6) hipLaunchKernelGGL(sweep_bench<6>, dim3(64), dim3(64 * waves), 0, 0, d_out, d_cyc, ntiles);
                CK(hipDeviceSynchronize());
            }
            long long c[8];
            CK(hipMemcpy(c, d_cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost));
            printf("T2 waves/block=%d %-24s cycles/column = %.2f\n", waves, names[mode], (double)c[0] / (ntiles * 32.0));
        }
    }
    return 0;
}
