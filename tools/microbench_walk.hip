// Development microbenchmark (not part of the product): cycles per step of candidate dependent chains for
// the row walk of the backtrack (one wave, SGPR-carried state).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_walk.hip -o /tmp/mbw && /tmp/mbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R64(X) R4(R16(X))

// every variant: 64 steps per asm statement, `iters` statements; state e in an SGPR
#define STEP_SALU6 \
    "s_lshr_b32 %[a], %[e], 5\n\t" "s_not_b32 %[b], %[a]\n\t" "s_lshr_b32 %[c], %[b], 3\n\t"        \
    "s_ff1_i32_b32 %[d], %[c]\n\t" "s_and_b32 %[d], %[d], 7\n\t" "s_sub_u32 %[e], %[e], %[d]\n\t"
#define STEP_SALU3 \
    "s_lshr_b32 %[a], %[e], 5\n\t" "s_and_b32 %[d], %[a], 7\n\t" "s_sub_u32 %[e], %[e], %[d]\n\t"
#define STEP_CHAIN \
    "s_lshr_b32 %[a], %[e], 5\n\t" "s_not_b32 %[b], %[e]\n\t" "v_readlane_b32 %[c], %[w], %[a]\n\t"   \
    "s_lshr_b32 %[d], %[c], %[b]\n\t" "s_ff1_i32_b32 %[d], %[d]\n\t" "s_subb_u32 %[e], %[e], %[d]\n\t"
#define STEP_CHAIN_BR \
    "s_lshr_b32 %[a], %[e], 5\n\t" "s_not_b32 %[b], %[e]\n\t" "v_readlane_b32 %[c], %[w], %[a]\n\t"   \
    "s_lshr_b32 %[d], %[c], %[b]\n\t" "s_ff1_i32_b32 %[d], %[d]\n\t" "s_subb_u32 %[a], %[e], %[d]\n\t" \
    "s_cmp_le_i32 %[a], %[e]\n\t" "s_mov_b32 %[e], %[a]\n\t" "s_cbranch_scc0 9f\n\t"
#define STEP_CHAIN_BR_WL \
    STEP_CHAIN_BR "v_writelane_b32 %[sv], %[e], 7\n\t"
// readlane with a constant lane (no SALU -> lane-select dependency), result feeds the chain
#define STEP_RL_CONST \
    "s_not_b32 %[b], %[e]\n\t" "v_readlane_b32 %[c], %[w], 5\n\t"                                     \
    "s_lshr_b32 %[d], %[c], %[b]\n\t" "s_ff1_i32_b32 %[d], %[d]\n\t" "s_subb_u32 %[e], %[e], %[d]\n\t"
// lane select from the chain, but the result is NOT on the chain
#define STEP_RL_SINK \
    "s_lshr_b32 %[a], %[e], 5\n\t" "v_readlane_b32 %[c], %[w], %[a]\n\t" "s_sub_u32 %[e], %[e], 3\n\t"
// VALU-only chain in all lanes with a ds_bpermute for the word
#define STEP_BPERM \
    "v_lshrrev_b32 %[va], 3, %[ve]\n\t" "v_and_b32 %[va], 0xfc, %[va]\n\t" "ds_bpermute_b32 %[vc], %[va], %[w]\n\t"  \
    "v_not_b32 %[vb], %[ve]\n\t" "s_waitcnt lgkmcnt(0)\n\t" "v_lshrrev_b32 %[vc], %[vb], %[vc]\n\t"       \
    "v_ffbl_b32 %[vc], %[vc]\n\t" "v_sub_u32 %[ve], %[ve], %[vc]\n\t" "v_subrev_u32 %[ve], 1, %[ve]\n\t"
// movrels: word from an SGPR window indexed by M0
#define STEP_MOVRELS \
    "s_lshr_b32 m0, %[e], 5\n\t" "s_not_b32 %[b], %[e]\n\t" "s_movrels_b32 %[c], s40\n\t"            \
    "s_lshr_b32 %[d], %[c], %[b]\n\t" "s_ff1_i32_b32 %[d], %[d]\n\t" "s_subb_u32 %[e], %[e], %[d]\n\t"

// two words (e's tile and the one before) as one 64-bit window: covers tokens that start in the previous tile
#define STEP_PAIR \
    "s_lshr_b32 %[a], %[e], 5\n\t" "s_not_b32 %[b], %[e]\n\t" "s_add_u32 %[c], %[a], -1\n\t"       \
    "v_readlane_b32 s62, %[w], %[a]\n\t" "v_readlane_b32 s63, %[w], %[c]\n\t"                      \
    "s_lshr_b64 s[62:63], s[62:63], %[b]\n\t" "s_ff1_i32_b64 %[d], s[62:63]\n\t" "s_subb_u32 %[e], %[e], %[d]\n\t"
#define STEP_PAIR_FULL \
    STEP_PAIR "s_or_b32 %[f], %[f], %[d]\n\t" "v_writelane_b32 %[sv], %[e], 7\n\t"
#define STEP_ONE_FULL \
    STEP_CHAIN "s_or_b32 %[f], %[f], %[d]\n\t" "v_writelane_b32 %[sv], %[e], 7\n\t"

#define STEP_PAIR_VCC \
    "s_lshr_b32 %[a], %[e], 5\n\t" "s_add_u32 %[c], %[a], -1\n\t" "s_andn2_b32 %[b], 31, %[e]\n\t"   \
    "v_readlane_b32 vcc_lo, %[w], %[a]\n\t" "v_readlane_b32 vcc_hi, %[w], %[c]\n\t"                  \
    "s_lshr_b64 vcc, vcc, %[b]\n\t" "s_ff1_i32_b64 %[d], vcc\n\t" "s_subb_u32 %[e], %[e], %[d]\n\t" \
    "s_or_b32 %[f], %[f], %[d]\n\t" "v_writelane_b32 %[sv], %[e], 7\n\t" "s_nop 0\n\t"

// straight-line code executed ONCE per launch after the instruction cache was flushed by 128 KB of s_nops:
// what a long unrolled walk pays for fetching its own instructions
__global__ void walk_cold(int *out, long long *cyc, int thrash) {
    const int lane = threadIdx.x;
    unsigned w = 0xffffffffu;
    int e = 64 * 31 + 20, a, b, c, d, sv = 0, fl = 0;
    for (int i = 0; i < thrash; ++i) asm volatile(R64(R64("s_nop 0\n\t")) R64(R64("s_nop 0\n\t")) ::: "memory");
    long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile(R64(STEP_PAIR_VCC) R64(STEP_PAIR_VCC) R64(STEP_PAIR_VCC)
                 : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d), [f] "+s"(fl), [sv] "+v"(sv) : [w] "v"(w) : "scc", "vcc");
    long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile(R64(STEP_PAIR_VCC) R64(STEP_PAIR_VCC) R64(STEP_PAIR_VCC)
                 : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d), [f] "+s"(fl), [sv] "+v"(sv) : [w] "v"(w) : "scc", "vcc");
    long long t2 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = e + sv + fl;
    if (lane == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = t2 - t1; }
}

template <int MODE>
__global__ void walk_bench(int *out, long long *cyc, int iters, unsigned seedw) {
    const int lane = threadIdx.x;
    unsigned w = 0xffffffffu;                 // every frame a decision bit: each step moves one frame back
    int e = 64 * 31 + 20 + (int)(seedw & 1);
    int a, b, c, d, sv = 0, fl = 0;
    int ve = e, va = 0, vb = 0, vc = 0;
    if (MODE == 8) {
        asm volatile(
            "s_mov_b32 s40, -1\n\ts_mov_b32 s41, -1\n\ts_mov_b32 s42, -1\n\ts_mov_b32 s43, -1\n\t"
            "s_mov_b32 s44, -1\n\ts_mov_b32 s45, -1\n\ts_mov_b32 s46, -1\n\ts_mov_b32 s47, -1\n\t"
            ::: "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
        e = 32 * 7 + 20;
    }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 8) e = 32 * 7 + 20 + (it & 1);
        if (MODE == 0) asm volatile(R64(STEP_SALU6) : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d) :: "scc");
        if (MODE == 1) asm volatile(R64(STEP_SALU3) : [e] "+s"(e), [a] "=&s"(a), [d] "=&s"(d) :: "scc");
        if (MODE == 2) asm volatile(R16(STEP_CHAIN) : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d) : [w] "v"(w) : "scc");
        if (MODE == 3) asm volatile(R16(STEP_CHAIN_BR) "9:\n\t" : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d) : [w] "v"(w) : "scc");
        if (MODE == 4) asm volatile(R16(STEP_CHAIN_BR_WL) "9:\n\t" : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d), [sv] "+v"(sv) : [w] "v"(w) : "scc");
        if (MODE == 5) asm volatile(R16(STEP_RL_CONST) : [e] "+s"(e), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d) : [w] "v"(w) : "scc");
        if (MODE == 6) asm volatile(R16(STEP_RL_SINK) : [e] "+s"(e), [a] "=&s"(a), [c] "=&s"(c) : [w] "v"(w) : "scc");
        if (MODE == 7) asm volatile(R16(STEP_BPERM) : [ve] "+v"(ve), [va] "+v"(va), [vb] "+v"(vb), [vc] "+v"(vc) : [w] "v"(w) : "memory");
        if (MODE == 8) asm volatile(R16(STEP_MOVRELS) : [e] "+s"(e), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d) :: "scc", "m0", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
        if (MODE == 9) asm volatile(R16(STEP_PAIR) : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d) : [w] "v"(w) : "scc", "s62", "s63");
        if (MODE == 10) asm volatile(R16(STEP_PAIR_FULL) : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d), [f] "+s"(fl), [sv] "+v"(sv) : [w] "v"(w) : "scc", "s62", "s63");
        if (MODE == 11) asm volatile(R16(STEP_ONE_FULL) : [e] "+s"(e), [a] "=&s"(a), [b] "=&s"(b), [c] "=&s"(c), [d] "=&s"(d), [f] "+s"(fl), [sv] "+v"(sv) : [w] "v"(w) : "scc");
        if (MODE != 0 && MODE != 1 && MODE != 7 && MODE != 8 && e < 64) e += 64 * 31;   // stay inside the 64 lanes' words
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = e + sv + ve + fl;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    int *d_out; long long *d_cyc;
    CK(hipMalloc(&d_out, 1 << 16)); CK(hipMalloc(&d_cyc, 1 << 12));
    const int iters = 400;
    const char *names[12] = {"6 dependent SALU", "3 dependent SALU", "chain (lshr,not,readlane,lshr,ff1,subb)", "chain + cmp/mov/cbranch(not taken)",
                            "chain + branch + v_writelane", "chain, readlane lane = const", "readlane off-chain (lshr,readlane,sub)",
                            "VALU chain with ds_bpermute", "chain with s_movrels instead of readlane",
                            "64-bit window: 2 readlane + lshr_b64/ff1_b64/subb", "64-bit window + flag + v_writelane", "32-bit chain + flag + v_writelane"};
    const int steps[12] = {64, 64, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    for (int mode = 0; mode < 12; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
#define L(M) if (mode == M) hipLaunchKernelGGL(walk_bench<M>, dim3(64), dim3(64), 0, 0, d_out, d_cyc, iters, 0u)
            L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); L(9); L(10); L(11);
            CK(hipDeviceSynchronize());
        }
        long long c;
        int e0;
        CK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&e0, d_out, 4, hipMemcpyDeviceToHost));
        printf("%-44s cycles/step = %7.2f   (state %d)\n", names[mode], (double)c / (iters * (double)steps[mode]), e0);
    }
    for (int thrash = 0; thrash <= 4; thrash += 4) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(walk_cold, dim3(64), dim3(64), 0, 0, d_out, d_cyc, thrash);
            CK(hipDeviceSynchronize());
            long long c2[4];
            CK(hipMemcpy(c2, d_cyc, 32, hipMemcpyDeviceToHost));
            printf("straight-line 192 vcc steps once, %3d KB of s_nops before: first copy %.1f, second copy %.1f cycles/step (block 0; block 1: %.1f, %.1f)\n",
                   thrash * 32, c2[0] / 192.0, c2[1] / 192.0, c2[2] / 192.0, c2[3] / 192.0);
        }
    }
    return 0;
}
