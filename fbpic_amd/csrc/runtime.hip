// Runtime glue of libfbpic_amd.so: error reporting, device binding, stream sync.
#include "fb_common.h"
#include <stdio.h>
#include <string.h>

namespace fb {

static thread_local char g_err[512] = "";

void set_error(const char *where, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", where, what);
}

int check(hipError_t e, const char *where)
{
    if (e == hipSuccess) return 0;
    set_error(where, hipGetErrorString(e));
    return (int)e;
}

}  // namespace fb

extern "C" int fb_abi_version(void) { return FB_ABI_VERSION; }
extern "C" const char *fb_last_error(void) { return fb::g_err; }
#ifndef FB_BUILD_INFO
#define FB_BUILD_INFO "unknown (built without the Makefile)"
#endif
extern "C" const char *fb_build_info(void) { return FB_BUILD_INFO; }
extern "C" const char *fb_last_error_string(void) { return fb::g_err; }
extern "C" int fb_malloc(size_t nbytes, void **device_ptr)
{
    if (!device_ptr) { fb::set_error("fb_malloc", "device_ptr is NULL"); return -1; }
    *device_ptr = nullptr;
    if (nbytes == 0) return 0;
    return fb::check(hipMalloc(device_ptr, nbytes), "fb_malloc");
}
extern "C" int fb_free(void *device_ptr) { return device_ptr ? fb::check(hipFree(device_ptr), "fb_free") : 0; }
extern "C" int fb_h2d(void *device_dst, const void *host_src, size_t nbytes, void *stream)
{
    if (nbytes == 0) return 0;
    return fb::check(hipMemcpyAsync(device_dst, host_src, nbytes, hipMemcpyHostToDevice, (hipStream_t)stream), "fb_h2d");
}
extern "C" int fb_d2h(void *host_dst, const void *device_src, size_t nbytes, void *stream)
{
    if (nbytes == 0) return 0;
    return fb::check(hipMemcpyAsync(host_dst, device_src, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "fb_d2h");
}
extern "C" int fb_set_device(int device) { return fb::check(hipSetDevice(device), "fb_set_device"); }
extern "C" int fb_sync(void *stream)
{
    return fb::check(hipStreamSynchronize((hipStream_t)stream), "fb_sync");
}
