// slhip_api.cpp -- library-level entry points of libslhip.so (error reporting, device init).
#include <stdarg.h>
#include <string.h>

#include "slhip_common.h"

namespace slhip {

static thread_local char g_error[1024] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace slhip

extern "C" int slhip_abi_version(void) { return SLHIP_ABI_VERSION; }

extern "C" const char* slhip_last_error(void) { return slhip::g_error; }

extern "C" int slhip_device_init(int device_index)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        slhip::set_error("no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return -2;
    }
    if (device_index < 0 || device_index >= n) {
        slhip::set_error("device index %d out of range (have %d)", device_index, n);
        return -1;
    }
    SLHIP_CHECK(hipSetDevice(device_index));
    hipDeviceProp_t prop;
    SLHIP_CHECK(hipGetDeviceProperties(&prop, device_index));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        slhip::set_error("libslhip.so is built for gfx950 only, device %d is %s", device_index, prop.gcnArchName);
        return -4;
    }
    return 0;
}
