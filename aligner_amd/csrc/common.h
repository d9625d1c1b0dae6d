// Shared host-side helpers for libaligner_amd.so (error reporting, HIP checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "aligner_amd.h"

namespace aligner {

// thread-local message behind aligner_last_error()
char *error_buffer();
int   fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define ALIGNER_HIP_CHECK(expr)                                                        \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess)                                                          \
            return ::aligner::fail(ALIGNER_EHIP, "%s failed: %s (%s:%d)", #expr,       \
                                   hipGetErrorString(_e), __FILE__, __LINE__);         \
    } while (0)

inline int dtype_size(int dt) {
    switch (dt) {
        case ALIGNER_DT_F32: return 4;
        case ALIGNER_DT_F16: return 2;
        case ALIGNER_DT_BF16: return 2;
        case ALIGNER_DT_F64: return 8;
        case ALIGNER_DT_I32: return 4;
        case ALIGNER_DT_U8: return 1;
        case ALIGNER_DT_I64: return 8;
        default: return 0;
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver call: do it once per kernel and size,
// not on every launch (returns hipSuccess when nothing had to be done)
hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes);   // cached per (device, kernel)

// LDS bytes one workgroup may own on the current device (160 KiB on gfx950), cached per device
int device_lds_limit();
int device_cu_count();                // compute units of the current device (cached per device)

// development aid shared by the kernel files (aligner_debug_set_stamps)
extern unsigned long long *g_debug_stamps;
extern int g_opt_softattn_exact;       // "softattn_exact": always the exact-product similarity kernel
extern int g_opt_mobo_start_lag;       // "mobo_start_lag": rows a position segment lets its predecessor get ahead before it starts (default 0)
extern int g_opt_mobo_stamp_wave;      // "mobo_stamp_wave": development, the wave whose first lane takes the gradient chain's stamps
extern int g_opt_mobo_full_chain;      // "mobo_full_chain": testing, the search without log_alpha on the full (sum + max) chain kernel
extern int g_opt_conv_narrow_ft;      // "conv_narrow_ft": testing, frame tiles (8, 4, 2) per workgroup of the narrow conv kernel (0: the plan's choice)
extern int g_opt_conv_no_ring;        // "conv_no_ring": A-B / testing, a k = 1 layer with many input chunks stages them all at once (conv_narrow_kernel)
extern int g_opt_maxpath_no_optimistic_mask;   // "maxpath_no_optimistic_mask": A-B / testing, the strict mask is verified by a pass in front of the search, not by the zero workgroups beside it
extern int g_opt_maxpath_no_mask_verify;   // "maxpath_no_mask_verify": A-B / testing, a strict mask is always multiplied in (no verification pass)
extern int g_opt_maxpath_zero_blocks;      // "maxpath_zero_blocks": development, zero workgroups of the two-workgroup search launch (0: all idle CUs)
extern int g_opt_maxpath_no_split_walk;   // "maxpath_no_split_walk": A-B / testing, the two-workgroup search with one walker (the second half) for all rows
extern int g_opt_conv_no_fuse;        // "conv_no_fuse": testing / A-B, a stack's trailing narrow layers as separate kernels instead of conv_narrow_fused_kernel
extern int g_opt_conv_split_always;    // "conv_split_always": narrow layers never stage fp32 themselves (A/B, tests)
extern int g_opt_softattn_split;       // "softattn_split": development, waves per strip of the row-group form (2, 4; 1: at most 2; 0: the launch's choice)
extern int g_opt_softattn_strips;      // "softattn_strips": A/B / testing, the one-row-group similarity kernel in its strip-per-wave form (softattn_kernel) instead of the row-tile form
extern int g_opt_softattn_rt_drop_merge;   // "softattn_rt_drop_merge": testing, the row-tile similarity kernel's loader never publishes a strip's normaliser
extern int g_opt_softattn_no_pair;     // "softattn_no_pair": testing, the similarity kernel's row-group form with one wave per strip only
extern int g_opt_mobo_bwd_general;     // "mobo_bwd_general": testing, the gradient's chain in its general (one exp2 per term) form
extern int g_opt_mobo_lanes;           // "mobo_lanes": development, lanes per position in the split form (0: the plan's choice)
extern int g_opt_mobo_drop_segment;    // "mobo_drop_segment": testing, that position segment never publishes (-1: off)
extern int g_opt_fwdsum_no_grad_stager; // "fwdsum_no_grad_stager": A-B / testing, the gradient-making backward kernel with its compiler-scheduled stager
extern int g_opt_fwdsum_serial;        // "fwdsum_serial": forward then backward sweep, never side by side (A/B, tests)
extern int g_opt_fwdsum_one_wave;      // aligner_debug_set_option("fwdsum_one_wave", ...); default: env, read once

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace aligner
