// Host-side plumbing shared by the C-ABI entry points: a per-thread stream + device scratch arena,
// and staging of caller-owned HOST rectangles for the synchronous per-slot ("parity") entries.
//
// The reference's DSP slots are called synchronously with host pointers, concurrently from several
// decoder worker threads, and keep no state between calls (libavcodec/vvc/vvc_thread.c:647).
// Each host thread therefore owns its own stream and scratch arena; entries are re-entrant.
#pragma once
#include "common.hpp"
#include <atomic>
#include <vector>

namespace vvc355 {

struct ThreadCtx {
    hipStream_t stream = nullptr;
    uint8_t *dev = nullptr;      // fixed-size device scratch arena, allocated on first use
    size_t cap = 0;
    int device = 0;              // the ordinal this context's stream and arena live on
    ThreadCtx();
    ~ThreadCtx();
};
ThreadCtx &thread_ctx();
extern std::atomic<int> g_device;
extern std::atomic<int> g_error_policy, g_last_error;
const char *last_error_text();

// A staged host rectangle: `dev` is the device address that corresponds to the caller's host pointer.
struct Staged {
    uint8_t *dev = nullptr;
    ptrdiff_t pitch = 0;         // device row pitch in bytes
};

// One synchronous slot call: stage inputs, launch on the thread's stream, copy results back, wait.
class SlotCall {
public:
    SlotCall();
    ~SlotCall();   // copies registered outputs back and synchronises the stream
    hipStream_t stream() const { return ctx_.stream; }

    // Rows [y_lo, y_hi) and byte columns [x_lo, x_hi) relative to `host` are made available on the device.
    // upload: copy host -> device now; download: copy device -> host when the call object is destroyed.
    Staged rect(const void *host, ptrdiff_t stride, ptrdiff_t x_lo, ptrdiff_t x_hi, int y_lo, int y_hi,
                bool upload, bool download);
    // copy back only `rows` rows of `width` bytes at `host` (whose staged twin is `dev`) when the call object is destroyed: for slots
    // that read a window but own only a part of it (other decoder threads may be writing the rest of the window meanwhile)
    void download(void *host, ptrdiff_t hstride, uint8_t *dev, ptrdiff_t dpitch, size_t width, int rows)
    {
        if (width && rows > 0) outs_.push_back({ host, hstride, dev, dpitch, width, rows });
    }
    // linear byte range [0, bytes)
    void *linear(const void *host, size_t bytes, bool upload, bool download);
    void *scratch(size_t bytes);
    template <typename T> T *upload(const T *host, size_t n) { return (T *)linear(host, n * sizeof(T), true, false); }

private:
    struct Out { void *host; ptrdiff_t hstride; uint8_t *dev; ptrdiff_t dpitch; size_t width; int rows; };
    ThreadCtx &ctx_;
    size_t used_ = 0;
    std::vector<Out> outs_;
    uint8_t *bump(size_t bytes);
};

} // namespace vvc355
