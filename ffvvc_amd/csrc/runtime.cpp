#include "runtime.hpp"
#include <mutex>

namespace vvc355 {

static constexpr size_t kArenaBytes = 48u << 20;   // far above any single slot call (largest: 128x135 int16 planes)

// The device ordinal the process decodes on (vvc355_set_device): HIP's current device is per-thread state, and the reference calls
// the slots from its own worker threads (libavutil/executor.c:92-110, vvc_thread.c:647-654), which never called hipSetDevice.
std::atomic<int> g_device{ 0 };

// error policy of HIP_CHECK: 0 = print and abort (default: the void slots have no other channel), 1 = print, record the first failure
// and go on (for hosts that call the batched / frame entries and check vvc355_last_error())
std::atomic<int> g_error_policy{ 0 }, g_last_error{ 0 };
static char g_last_error_text[256];

const char *last_error_text() { return g_last_error_text; }

void hip_fail(const char *expr, int err, const char *what, const char *file, int line)
{
    fprintf(stderr, "vvc_mi355: %s failed: %s (%s:%d)\n", expr, what, file, line);
    if (g_error_policy.load() == 0)
        abort();
    // the first failure is kept: the text is complete before the code becomes visible to vvc355_last_error()
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (g_last_error.load() == 0) {
        snprintf(g_last_error_text, sizeof(g_last_error_text), "%s failed: %s (%s:%d)", expr, what, file, line);
        g_last_error.store(err ? err : -1);
    }
}

ThreadCtx::ThreadCtx()
{
    device = g_device.load();
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_CHECK(hipMalloc((void **)&dev, kArenaBytes));
    cap = kArenaBytes;
}

ThreadCtx::~ThreadCtx()
{
    // Process teardown may already have destroyed the HIP runtime; ignore errors here.
    if (dev) (void)hipFree(dev);
    if (stream) (void)hipStreamDestroy(stream);
}

ThreadCtx &thread_ctx()
{
    static thread_local ThreadCtx ctx;
    return ctx;
}

SlotCall::SlotCall() : ctx_(thread_ctx())
{
    if (ctx_.device != g_device.load()) {
        fprintf(stderr, "vvc_mi355: vvc355_set_device(%d) after this thread started issuing slot calls on device %d\n", g_device.load(), ctx_.device);
        abort();
    }
}

SlotCall::~SlotCall()
{
    for (const Out &o : outs_)
        HIP_CHECK(hipMemcpy2DAsync(o.host, o.hstride, o.dev, o.dpitch, o.width, o.rows, hipMemcpyDeviceToHost, ctx_.stream));
    HIP_CHECK(hipStreamSynchronize(ctx_.stream));
}

uint8_t *SlotCall::bump(size_t bytes)
{
    const size_t at = (used_ + 255) & ~(size_t)255;
    if (!ctx_.dev || !ctx_.stream) {
        // under error policy 1 a failed stream / arena creation was only recorded; a slot call cannot go on without them
        fprintf(stderr, "vvc_mi355: this thread has no staging arena (its creation failed: %s)\n", last_error_text());
        abort();
    }
    if (at + bytes > ctx_.cap) {
        fprintf(stderr, "vvc_mi355: slot staging arena exhausted (%zu + %zu > %zu)\n", at, bytes, ctx_.cap);
        abort();
    }
    used_ = at + bytes;
    return ctx_.dev + at;
}

Staged SlotCall::rect(const void *host, ptrdiff_t stride, ptrdiff_t x_lo, ptrdiff_t x_hi, int y_lo, int y_hi,
                      bool upload, bool download)
{
    Staged s;
    const size_t width = (size_t)(x_hi - x_lo);
    const int rows = y_hi - y_lo;
    // the caller's origin lands 64-byte aligned on the device, with a 64-byte-multiple pitch, so the
    // kernels' vector paths apply whatever the host alignment was
    const size_t lead = ((size_t)(-x_lo) + 63) & ~(size_t)63;       // x_lo <= 0 for every caller
    s.pitch = (ptrdiff_t)((lead + (size_t)x_hi + 63) & ~(size_t)63);
    uint8_t *blk = bump((size_t)s.pitch * (size_t)rows + 64);
    s.dev = blk + lead - (ptrdiff_t)y_lo * s.pitch;
    uint8_t *h0 = (uint8_t *)const_cast<void *>(host) + (ptrdiff_t)y_lo * stride + x_lo;
    uint8_t *d0 = s.dev + (ptrdiff_t)y_lo * s.pitch + x_lo;
    if (width == 0 || rows <= 0)
        return s;
    if (upload)
        HIP_CHECK(hipMemcpy2DAsync(d0, s.pitch, h0, stride, width, rows, hipMemcpyHostToDevice, ctx_.stream));
    if (download)
        outs_.push_back({ h0, stride, d0, s.pitch, width, rows });
    return s;
}

void *SlotCall::linear(const void *host, size_t bytes, bool upload, bool download)
{
    uint8_t *d = bump(bytes ? bytes : 1);
    if (bytes && upload)
        HIP_CHECK(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx_.stream));
    if (bytes && download)
        outs_.push_back({ const_cast<void *>(host), (ptrdiff_t)bytes, d, (ptrdiff_t)bytes, bytes, 1 });
    return d;
}

void *SlotCall::scratch(size_t bytes) { return bump(bytes ? bytes : 1); }

} // namespace vvc355
