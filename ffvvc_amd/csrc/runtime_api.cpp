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
void vvc355_set_error_policy(int record_instead_of_abort) { vvc355::g_error_policy.store(record_instead_of_abort ? 1 : 0); }
int vvc355_last_error(void) { return vvc355::g_last_error.load(); }
const char *vvc355_last_error_string(void) { return vvc355::g_last_error.load() ? vvc355::last_error_text() : ""; }
void vvc355_clear_error(void) { (void)hipGetLastError(); vvc355::g_last_error.store(0); }

void vvc355_set_device(int ordinal)
{
    HIP_CHECK(hipSetDevice(ordinal));
    vvc355::g_device.store(ordinal);        // worker threads that first call a slot later start on this device
}
void *vvc355_malloc(size_t bytes)
{
    void *p = nullptr;
    HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
    return p;
}
void vvc355_free(void *dev) { HIP_CHECK(hipFree(dev)); }
void vvc355_upload(void *dev, const void *host, size_t bytes) { HIP_CHECK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice)); }
void vvc355_download(void *host, const void *dev, size_t bytes) { HIP_CHECK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost)); }
void vvc355_copy_async(void *stream, void *dst_dev, const void *src_dev, size_t bytes)
{
    HIP_CHECK(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
}
void *vvc355_stream_create(void)
{
    hipStream_t s;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return (void *)s;
}
void vvc355_stream_destroy(void *stream) { HIP_CHECK(hipStreamDestroy((hipStream_t)stream)); }
void vvc355_stream_sync(void *stream) { HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }
void vvc355_graph_begin(void *stream) { HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal)); }
void *vvc355_graph_end(void *stream)
{
    hipGraph_t g;
    hipGraphExec_t e;
    HIP_CHECK(hipStreamEndCapture((hipStream_t)stream, &g));
    HIP_CHECK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphDestroy(g));
    return (void *)e;
}
void vvc355_graph_launch(void *graph_exec, void *stream) { HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream)); }
void vvc355_graph_destroy(void *graph_exec) { HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph_exec)); }
const char *vvc355_version(void) { return "vvc_mi355 0.1 (gfx950)"; }

} // extern "C"
