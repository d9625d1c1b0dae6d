// Entry points that are not tied to one kernel family: versioning, error text,
// device probing.  Part of libaligner_amd.so (see include/aligner_amd.h).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include <mutex>
#include <cstring>
#include <map>
#include <utility>

#include "common.h"

namespace aligner {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

unsigned long long *g_debug_stamps = nullptr;

// Development switches (aligner_debug_set_option): default from the environment, read once.
static int env_flag(const char *name) { const char *e = getenv(name); return e && e[0] && e[0] != '0'; }
int g_opt_fwdsum_one_wave = env_flag("ALIGNER_FWDSUM_ONE_WAVE");
int g_opt_fwdsum_serial = env_flag("ALIGNER_FWDSUM_SERIAL");
int g_opt_fwdsum_no_grad_stager = 0;
int g_opt_softattn_exact = env_flag("ALIGNER_SOFTATTN_EXACT");
int g_opt_mobo_drop_segment = -1;
int g_opt_mobo_start_lag = 0;
int g_opt_mobo_lanes = 0;
int g_opt_mobo_bwd_general = 0;
int g_opt_softattn_no_pair = 0;
int g_opt_softattn_rt_drop_merge = 0;
int g_opt_softattn_strips = env_flag("ALIGNER_SOFTATTN_STRIPS");
int g_opt_softattn_split = 0;
int g_opt_conv_narrow_ft = 0;
int g_opt_conv_no_ring = 0;
int g_opt_conv_no_fuse = 0;
int g_opt_conv_split_always = env_flag("ALIGNER_CONV_SPLIT_ALWAYS");
int g_opt_maxpath_no_split_walk = 0;
int g_opt_maxpath_zero_blocks = 0;
int g_opt_maxpath_no_mask_verify = 0;
int g_opt_maxpath_no_optimistic_mask = 0;
int g_opt_mobo_full_chain = 0;
int g_opt_mobo_stamp_wave = 0;

hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    // the attribute is a property of (device, kernel): a grant on one device says nothing about another
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> granted;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, kernel);
    auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted[key] = bytes;
    return e;
}

int device_cu_count() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    static std::mutex mu;
    static std::map<int, int> counts;
    std::lock_guard<std::mutex> lock(mu);
    auto it = counts.find(dev);
    if (it != counts.end()) return it->second;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) v = 1;
    counts[dev] = v;
    return v;
}

int device_lds_limit() {
    // gfx950 lets one workgroup own the CU's whole 160 KiB LDS; cached per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 64 * 1024;
    static std::mutex mu;
    static std::map<int, int> limits;
    std::lock_guard<std::mutex> lock(mu);
    auto it = limits.find(dev);
    if (it != limits.end()) return it->second;
    int lim = 64 * 1024, v = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0)
        lim = 160 * 1024;
    else if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0)
        lim = v;
    limits[dev] = lim;
    return lim;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace aligner

extern "C" {

int aligner_abi_version(void) { return ALIGNER_ABI_VERSION; }

const char *aligner_last_error(void) { return aligner::error_buffer(); }

void aligner_debug_set_stamps(void *stamps_dev) {
    aligner::g_debug_stamps = static_cast<unsigned long long *>(stamps_dev);
}

int aligner_debug_set_option(const char *name, int value) {
    if (!name) return aligner::fail(ALIGNER_EINVAL, "null option name");
    if (std::strcmp(name, "fwdsum_one_wave") == 0) { aligner::g_opt_fwdsum_one_wave = value; return ALIGNER_OK; }
    if (std::strcmp(name, "fwdsum_no_grad_stager") == 0) { aligner::g_opt_fwdsum_no_grad_stager = value; return ALIGNER_OK; }
    if (std::strcmp(name, "fwdsum_serial") == 0) { aligner::g_opt_fwdsum_serial = value; return ALIGNER_OK; }
    if (std::strcmp(name, "softattn_exact") == 0) { aligner::g_opt_softattn_exact = value; return ALIGNER_OK; }
    if (std::strcmp(name, "mobo_start_lag") == 0) { aligner::g_opt_mobo_start_lag = value; return ALIGNER_OK; }
    if (std::strcmp(name, "mobo_lanes") == 0) { aligner::g_opt_mobo_lanes = value; return ALIGNER_OK; }
    if (std::strcmp(name, "mobo_stamp_wave") == 0) { aligner::g_opt_mobo_stamp_wave = value; return ALIGNER_OK; }
    if (std::strcmp(name, "mobo_full_chain") == 0) { aligner::g_opt_mobo_full_chain = value; return ALIGNER_OK; }
    if (std::strcmp(name, "softattn_split") == 0) { aligner::g_opt_softattn_split = value; return ALIGNER_OK; }
    if (std::strcmp(name, "conv_narrow_ft") == 0) { aligner::g_opt_conv_narrow_ft = value; return ALIGNER_OK; }
    if (std::strcmp(name, "conv_no_ring") == 0) { aligner::g_opt_conv_no_ring = value; return ALIGNER_OK; }
    if (std::strcmp(name, "conv_no_fuse") == 0) { aligner::g_opt_conv_no_fuse = value; return ALIGNER_OK; }
    if (std::strcmp(name, "conv_split_always") == 0) { aligner::g_opt_conv_split_always = value; return ALIGNER_OK; }
    if (std::strcmp(name, "maxpath_no_optimistic_mask") == 0) { aligner::g_opt_maxpath_no_optimistic_mask = value; return ALIGNER_OK; }
    if (std::strcmp(name, "maxpath_no_mask_verify") == 0) { aligner::g_opt_maxpath_no_mask_verify = value; return ALIGNER_OK; }
    if (std::strcmp(name, "maxpath_zero_blocks") == 0) { aligner::g_opt_maxpath_zero_blocks = value; return ALIGNER_OK; }
    if (std::strcmp(name, "maxpath_no_split_walk") == 0) { aligner::g_opt_maxpath_no_split_walk = value; return ALIGNER_OK; }
    if (std::strcmp(name, "softattn_strips") == 0) { aligner::g_opt_softattn_strips = value; return ALIGNER_OK; }
    if (std::strcmp(name, "softattn_rt_drop_merge") == 0) { aligner::g_opt_softattn_rt_drop_merge = value; return ALIGNER_OK; }
    if (std::strcmp(name, "softattn_no_pair") == 0) { aligner::g_opt_softattn_no_pair = value; return ALIGNER_OK; }
    if (std::strcmp(name, "mobo_bwd_general") == 0) { aligner::g_opt_mobo_bwd_general = value; return ALIGNER_OK; }
    if (std::strcmp(name, "mobo_drop_segment") == 0) { aligner::g_opt_mobo_drop_segment = value; return ALIGNER_OK; }
    return aligner::fail(ALIGNER_EINVAL, "unknown option '%s'", name);
}

int aligner_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
