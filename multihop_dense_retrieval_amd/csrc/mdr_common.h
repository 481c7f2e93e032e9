// csrc/mdr_common.h -- error plumbing shared by the translation units of libmdrhip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/mdr_hip.h"
#include "../../include/mdr_hip_measure.h"

namespace mdr {

// thread-local message behind mdr_last_error()
char* last_error_buf();
int set_error(int code, const char* fmt, ...);

#define MDR_HIP_TRY(expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return ::mdr::set_error(MDR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                                    __LINE__);                                                              \
    } while (0)

#define MDR_REQUIRE(cond, ...)                                          \
    do {                                                                \
        if (!(cond)) return ::mdr::set_error(MDR_E_INVALID, __VA_ARGS__); \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: remember which (kernel, current device) pairs have
// been opted in, so a second device in one process (mdr_index_create(device) allows it) is opted in as well.
int ensure_dynamic_lds(const void* kernel, int bytes);

// RAII device switch: every entry point runs on its handle's device and restores the caller's.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        cur = dev;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != cur) (void)hipSetDevice(prev);
    }
    int cur = -1;
};

}  // namespace mdr
