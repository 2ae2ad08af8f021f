// Runtime helper entries of the C ABI (device memory, streams) — see include/vvc_mi355.h.
#include <algorithm>
#include <queue>
#include <utility>
#include <vector>

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
// ---- vvc355_recon_order: critical-path-first ticket order of the in-order pass (host only; see include/vvc_mi355.h)
namespace {
// the CTUs `rs` waits for, as recon_one_ctu / recon_light_ctu (intra.hip) do: neighbours that have commands
inline int recon_deps(const vvc355_recon_ctu *ctus, int ncx, int rs, int out[4])
{
    const int ry = rs / ncx, rx = rs - ry * ncx;
    int n = 0;
    auto take = [&](int d) { if (ctus[d].n_cmd) out[n++] = d; };
    if (ctus[rs].flags & VVC355_RECON_CTU_LIGHT) {
        if ((ctus[rs].flags & VVC355_RECON_CTU_LUMA_LEFT) && rx > 0) take(rs - 1);
        if ((ctus[rs].flags & VVC355_RECON_CTU_LUMA_UP) && ry > 0) take(rs - ncx);
        return n;
    }
    if (rx > 0) take(rs - 1);
    if (rx > 0 && ry > 0) take(rs - ncx - 1);
    if (ry > 0) take(rs - ncx);
    if (ry > 0 && rx + 1 < ncx) take(rs - ncx + 1);
    return n;
}
} // namespace

int vvc355_recon_order(const vvc355_recon_ctu *ctus, int ncx, int ncy, int32_t *order)
{
    const int n = ncx * ncy;
    if (n <= 0)
        return 0;
    std::vector<int64_t> tail(n, 0);       // weight of the heaviest chain that starts at the CTU (itself included)
    std::vector<int> indeg(n, 0);
    int dep[4];
    std::vector<int64_t> below(n, 0);      // ... of the CTUs waiting for it
    for (int rs = n - 1; rs >= 0; rs--) {  // a CTU's successors have larger raster indices: below[rs] is final here
        if (!ctus[rs].n_cmd)
            continue;
        tail[rs] = ((ctus[rs].flags & VVC355_RECON_CTU_LIGHT) ? (ctus[rs].n_cmd + 3) / 4 : ctus[rs].n_cmd) + below[rs];
        const int nd = recon_deps(ctus, ncx, rs, dep);
        indeg[rs] = nd;
        for (int i = 0; i < nd; i++)
            below[dep[i]] = std::max(below[dep[i]], tail[rs]);
    }
    using Item = std::pair<int64_t, int>;   // (tail, -rs): heaviest first, raster order among equals
    std::priority_queue<Item> ready;
    for (int rs = 0; rs < n; rs++)
        if (ctus[rs].n_cmd && !indeg[rs])
            ready.push({ tail[rs], -rs });
    int n_work = 0;
    while (!ready.empty()) {
        const int rs = -ready.top().second;
        ready.pop();
        order[n_work++] = rs;
        const int ry = rs / ncx, rx = rs - ry * ncx;
        const int succ[4] = { rx + 1 < ncx ? rs + 1 : -1, (ry + 1 < ncy && rx > 0) ? rs + ncx - 1 : -1, ry + 1 < ncy ? rs + ncx : -1,
                              (ry + 1 < ncy && rx + 1 < ncx) ? rs + ncx + 1 : -1 };
        for (int s : succ) {
            if (s < 0 || !ctus[s].n_cmd)
                continue;
            const int nd = recon_deps(ctus, ncx, s, dep);
            for (int i = 0; i < nd; i++)
                if (dep[i] == rs && --indeg[s] == 0)
                    ready.push({ tail[s], -s });
        }
    }
    return n_work;
}

const char *vvc355_version(void) { return "vvc_mi355 0.1 (gfx950)"; }

} // extern "C"
