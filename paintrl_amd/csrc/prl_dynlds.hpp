// prl_dynlds.hpp -- host side: dynamic-LDS opt-in of a kernel, remembered per (device, kernel).
//
// A launch with more dynamic LDS than the default limit needs hipFuncSetAttribute(MaxDynamicSharedMemorySize) first.  The
// attribute belongs to the function object of the CURRENT device (the C ABI allows parts on several devices in one process,
// prl_part_create(..., device, ...)), and the call is a host round trip: the rollout hot loop must not repeat it per launch.
// prl_grant_dyn_lds asks once per (device, kernel, larger size); the table is read and written under its lock only.
// Each translation unit has its own table (anonymous namespace): kernel pointers are per unit anyway.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <map>
#include <mutex>
#include <utility>

namespace {

struct DynLdsInfo {
    size_t static_lds = 0, dyn_granted = 0;
    bool known = false;
};

struct DynLdsTable {
    std::mutex mu;
    std::map<std::pair<int, const void *>, DynLdsInfo> table;
    DynLdsInfo &entry(const void *kernel, size_t default_limit) {          // (call with `mu` held)
        int dev = 0;
        (void)hipGetDevice(&dev);
        DynLdsInfo &info = table[std::make_pair(dev, kernel)];
        if (!info.known) {
            hipFuncAttributes attr;
            info.static_lds = hipFuncGetAttributes(&attr, kernel) == hipSuccess ? attr.sharedSizeBytes : 28 * 1024;
            info.dyn_granted = default_limit;
            info.known = true;
        }
        return info;
    }
};

inline DynLdsTable &dyn_lds_table() {
    static DynLdsTable t;
    return t;
}

// static LDS of `kernel` (bytes), as the runtime reports it
inline size_t prl_static_lds(const void *kernel) {
    DynLdsTable &t = dyn_lds_table();
    std::lock_guard<std::mutex> lock(t.mu);
    return t.entry(kernel, 32 * 1024).static_lds;
}

// Makes sure `kernel` may be launched with `lds` bytes of dynamic LDS on the calling thread's current device.
// `default_limit`: what a launch may use without the opt-in (the attribute is only set beyond it).
inline hipError_t prl_grant_dyn_lds(const void *kernel, size_t lds, size_t default_limit = 32 * 1024) {
    DynLdsTable &t = dyn_lds_table();
    std::lock_guard<std::mutex> lock(t.mu);
    DynLdsInfo &info = t.entry(kernel, default_limit);
    if (lds <= info.dyn_granted) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) info.dyn_granted = lds;
    return e;
}

}  // namespace
