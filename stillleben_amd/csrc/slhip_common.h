// slhip_common.h -- shared host/device helpers of libslhip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "slhip.h"

namespace slhip {

void set_error(const char* fmt, ...);

#define SLHIP_CHECK(expr)                                                                     \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::slhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                               __LINE__);                                                     \
            return -2;                                                                        \
        }                                                                                     \
    } while (0)

#define SLHIP_LAUNCH_CHECK()                                                                  \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess) {                                                               \
            ::slhip::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),     \
                               __FILE__, __LINE__);                                           \
            return -3;                                                                        \
        }                                                                                     \
    } while (0)

}  // namespace slhip
