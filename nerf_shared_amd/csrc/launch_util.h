// launch_util.h -- per-device launch state shared by the kernel launchers.
//
// The C ABI is re-entrant (include/nerf_amd.h): any host thread may launch on any device.  What a
// launcher remembers between calls is therefore kept per device and updated atomically:
//   * whether a kernel's dynamic-LDS limit has been raised on a device (hipFuncSetAttribute is a
//     per-device property of the loaded code object),
//   * a device's CU count.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace na {

constexpr int MAX_TRACKED_DEVICES = 64;

inline int current_device() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}

// CU count of the current device (cached per device; 256 if the query fails).
inline int device_cu_count() {
    static std::atomic<int> cache[MAX_TRACKED_DEVICES];      // zero-initialised
    const int dev = current_device();
    const bool tracked = dev >= 0 && dev < MAX_TRACKED_DEVICES;
    if (tracked) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int n = 0;
    if (dev < 0 || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (tracked) cache[dev].store(n, std::memory_order_relaxed);
    return n;
}

// One object per kernel instantiation (a function-local static of the launcher template): the set of
// devices on which the kernel's dynamic-LDS limit has been raised.  Two threads racing on the same
// device both call hipFuncSetAttribute with the same value, which is harmless.
struct DynamicLdsOptIn {
    std::atomic<uint64_t> done{0};
    hipError_t ensure(const void *kernel, size_t bytes) {
        const int dev = current_device();
        const bool tracked = dev >= 0 && dev < MAX_TRACKED_DEVICES;
        const uint64_t bit = tracked ? (uint64_t)1 << dev : 0;
        if (tracked && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess && tracked) done.fetch_or(bit, std::memory_order_release);
        return e;
    }
};

}  // namespace na
