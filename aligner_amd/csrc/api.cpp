// Entry points that are not tied to one kernel family: versioning, error text,
// device probing.  Part of libaligner_amd.so (see include/aligner_amd.h).
#include <cstdarg>
#include <cstdio>

#include <mutex>
#include <unordered_map>

#include "common.h"

namespace aligner {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

unsigned long long *g_debug_stamps = nullptr;

hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::unordered_map<const void *, size_t> granted;      // per device would be stricter; one GPU per process
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(kernel);
    if (it != granted.end() && it->second >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted[kernel] = bytes;
    return e;
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

int aligner_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
