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

// (float)x / 255.0f for a byte, without the 11-instruction IEEE division: one multiply by the rounded
// reciprocal plus one fma residual correction.  Equal to the correctly rounded quotient for all 256
// inputs (tests/test_oracle_render.py::test_unorm8_shortcut_is_exact checks them exhaustively).
#ifdef __HIPCC__
__device__ __forceinline__ float unorm8(unsigned char x)
{
    const float xf = (float)x, c = 1.0f / 255.0f;
    const float q = xf * c;
    return fmaf(fmaf(-q, 255.0f, xf), c, q);
}
#endif

}  // namespace slhip
