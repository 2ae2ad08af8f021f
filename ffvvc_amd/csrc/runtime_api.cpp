// Runtime helper entries of the C ABI (device memory, streams) — see include/vvc_mi355.h.
#include "runtime.hpp"
#include "../../include/vvc_mi355.h"

extern "C" {

int vvc355_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}
void vvc355_set_device(int ordinal) { HIP_CHECK(hipSetDevice(ordinal)); }
void *vvc355_malloc(size_t bytes)
{
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
    return p;
}
void vvc355_free(void *dev) { HIP_CHECK(hipFree(dev)); }
void vvc355_upload(void *dev, const void *host, size_t bytes) { HIP_CHECK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice)); }
void vvc355_download(void *host, const void *dev, size_t bytes) { HIP_CHECK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost)); }
void *vvc355_stream_create(void)
{
    hipStream_t s;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return (void *)s;
}
void vvc355_stream_destroy(void *stream) { HIP_CHECK(hipStreamDestroy((hipStream_t)stream)); }
void vvc355_stream_sync(void *stream) { HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }
const char *vvc355_version(void) { return "vvc_mi355 0.1 (gfx950)"; }

} // extern "C"
