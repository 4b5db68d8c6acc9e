// runtime.hip -- error state and the small device-runtime part of the C ABI.
#include "common.hpp"
#include <chrono>

namespace gemhip {

std::string &last_error_ref()
{
    static thread_local std::string s;
    return s;
}

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

double *phase_acc()
{
    static thread_local double acc[PH_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
    return acc;
}

double phase_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace gemhip

using namespace gemhip;

// {total_seconds, host_prepare_seconds, h2d_seconds, kernel_seconds, d2h_seconds, 0, 0, 0} of the LAST one-shot call on this thread
extern "C" int gemhip_last_call_phases(double *out)
{
    GEMHIP_REQUIRE(out != nullptr, "last_call_phases: NULL");
    for (int k = 0; k < PH_COUNT; ++k) out[k] = phase_acc()[k];
    return GEMHIP_OK;
}

extern "C" int gemhip_version(void) { return GEMHIP_VERSION; }

extern "C" const char *gemhip_last_error(void) { return last_error_ref().c_str(); }

extern "C" int gemhip_device_count(int *n)
{
    GEMHIP_REQUIRE(n != nullptr, "device_count: NULL");
    *n = 0;
    int c = 0;
    const hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) return fail(GEMHIP_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    *n = c;
    return GEMHIP_OK;
}

extern "C" int gemhip_set_device(int device)
{
    GEMHIP_CHECK(hipSetDevice(device));
    return GEMHIP_OK;
}

extern "C" int gemhip_malloc(void **dptr, int64_t bytes)
{
    GEMHIP_REQUIRE(dptr && bytes >= 0, "malloc: bad arguments");
    GEMHIP_CHECK(hipMalloc(dptr, bytes ? (size_t)bytes : 16));
    return GEMHIP_OK;
}

extern "C" int gemhip_free(void *dptr)
{
    if (dptr) GEMHIP_CHECK(hipFree(dptr));
    return GEMHIP_OK;
}

extern "C" int gemhip_memcpy_h2d(void *dst_dev, const void *src_host, int64_t bytes)
{
    GEMHIP_REQUIRE(dst_dev && src_host && bytes >= 0, "memcpy_h2d: bad arguments");
    GEMHIP_CHECK(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
    return GEMHIP_OK;
}

extern "C" int gemhip_memcpy_d2h(void *dst_host, const void *src_dev, int64_t bytes)
{
    GEMHIP_REQUIRE(dst_host && src_dev && bytes >= 0, "memcpy_d2h: bad arguments");
    GEMHIP_CHECK(hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return GEMHIP_OK;
}

extern "C" int gemhip_synchronize(void *stream)
{
    if (stream) GEMHIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    else GEMHIP_CHECK(hipDeviceSynchronize());
    return GEMHIP_OK;
}
