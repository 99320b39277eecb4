// common.h -- shared host-side helpers of libgaussctrl_hip.so (error channel, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/gaussctrl_hip.h"

namespace gc {
void set_error(const char *fmt, ...);
inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return GC_ELAUNCH;
    }
    return GC_OK;
}
inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), safe for several host threads and several devices
// in one process: a bit per device ordinal; racing first calls both set the (idempotent) attribute.
struct AttrOnce { std::atomic<uint64_t> done{0}; };
inline void ensure_dynamic_lds(AttrOnce &once, const void *func, int bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (once.done.load(std::memory_order_acquire) & bit) return;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess)
        once.done.fetch_or(bit, std::memory_order_release);
}
inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }
}  // namespace gc

#define GC_REQUIRE(cond, msg)                \
    do {                                     \
        if (!(cond)) {                       \
            gc::set_error("%s: %s", __func__, msg); \
            return GC_EINVAL;                \
        }                                    \
    } while (0)
