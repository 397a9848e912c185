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

// A stream whose kernels run only on CUs [first_cu, first_cu + n_cus) of the current device's CU mask
// order (the driver deals consecutive mask bits round-robin over the XCDs and shader engines, so a
// contiguous range takes the same share of every XCD).
extern "C" int slhip_stream_create_cu_range(uint32_t first_cu, uint32_t n_cus, void** stream_out)
{
    if (!stream_out || n_cus == 0) {
        slhip::set_error("slhip_stream_create_cu_range: null output or empty range");
        return -1;
    }
    int dev = 0;
    SLHIP_CHECK(hipGetDevice(&dev));
    int total = 0;
    SLHIP_CHECK(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev));
    if (first_cu + n_cus > (uint32_t)total) {
        slhip::set_error("slhip_stream_create_cu_range: CUs [%u, %u) exceed the device's %d", first_cu, first_cu + n_cus, total);
        return -1;
    }
    uint32_t mask[32] = {0};
    const uint32_t words = ((uint32_t)total + 31u) / 32u;
    if (words > 32u) {
        slhip::set_error("slhip_stream_create_cu_range: %d CUs exceed the mask buffer", total);
        return -1;
    }
    for (uint32_t c = first_cu; c < first_cu + n_cus; ++c) mask[c / 32u] |= 1u << (c % 32u);
    hipStream_t st = nullptr;
    SLHIP_CHECK(hipExtStreamCreateWithCUMask(&st, words, mask));
    *stream_out = (void*)st;
    return 0;
}

extern "C" int slhip_stream_destroy(void* stream)
{
    if (stream) SLHIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
