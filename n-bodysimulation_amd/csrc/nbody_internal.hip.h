// nbody_internal.hip.h — small helpers shared by the translation units of libnbody_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "nbody.h"

int nbody_fail(int code, const char* fmt, ...);  // nbody_context.hip: sets nbody_last_error(), returns code

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return nbody_fail(NBODY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                                 \
    } while (0)

namespace {

// Makes `device` current for the scope and restores the caller's device afterwards.
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            changed = (err == hipSuccess);
        }
    }
    ~DeviceScope()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

}  // namespace
