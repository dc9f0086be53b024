// hipGraph capture helpers of the C ABI (a UNet pass = one graph launch).
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>

#include "common.h"

using namespace leco;

extern "C" int leco_graph_begin_capture(leco_stream_t stream) {
    // LECO_CAPTURE_MODE=global|relaxed: experiment switch for the ROCm 7.2 capture crash; default
    // thread-local (another thread -- torch's autograd worker -- may keep making runtime calls while this thread captures)
    static const hipStreamCaptureMode mode = [] {
        const char* e = getenv("LECO_CAPTURE_MODE");
        if (e && !strcmp(e, "global")) return hipStreamCaptureModeGlobal;
        if (e && !strcmp(e, "relaxed")) return hipStreamCaptureModeRelaxed;
        return hipStreamCaptureModeThreadLocal;
    }();
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, mode);
    if (e != hipSuccess) return fail(-EIO, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int leco_graph_end_capture(leco_stream_t stream, leco_graph_t* out) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess || !g) return fail(-EIO, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(-EIO, "hipGraphInstantiate: %s", hipGetErrorString(e));
    *out = (leco_graph_t)ex;
    return 0;
}
extern "C" int leco_graph_launch(leco_graph_t graph, leco_stream_t stream) {
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)stream);
    if (e != hipSuccess) return fail(-EIO, "hipGraphLaunch: %s", hipGetErrorString(e));
    return 0;
}
extern "C" int leco_graph_destroy(leco_graph_t graph) {
    if (graph) (void)hipGraphExecDestroy((hipGraphExec_t)graph);
    return 0;
}

// ---- two-stream sections (leco_hip.h): one side stream per device + a ring of timing-less events
namespace {
constexpr int MAX_DEV = 64, NEV = 256;
hipStream_t g_side[MAX_DEV] = {};
hipEvent_t g_ev[MAX_DEV][NEV] = {};
int g_ev_next[MAX_DEV] = {};
int cur_dev() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d < 0 || d >= MAX_DEV ? 0 : d;
}
// stream + event ring are created together on first use (leco_side_stream() before a capture begins: resource creation
// is not a capturable operation)
hipStream_t side_of(int d) {
    if (!g_side[d]) {
        if (hipStreamCreateWithFlags(&g_side[d], hipStreamNonBlocking) != hipSuccess) { g_side[d] = nullptr; return nullptr; }
        for (int i = 0; i < NEV; ++i)
            if (hipEventCreateWithFlags(&g_ev[d][i], hipEventDisableTiming) != hipSuccess) g_ev[d][i] = nullptr;
    }
    return g_side[d];
}
// an event of the ring: a record / wait pair is enqueued back to back, so re-use after NEV pairs is safe
hipEvent_t next_event(int d) {
    const int i = g_ev_next[d];
    g_ev_next[d] = (i + 1) % NEV;
    return g_ev[d][i];
}
int edge(hipStream_t from, hipStream_t to, const char* what) {
    const int d = cur_dev();
    (void)side_of(d);
    hipEvent_t ev = next_event(d);
    if (!ev) return fail(-EIO, "%s: hipEventCreate failed", what);
    hipError_t e = hipEventRecord(ev, from);
    if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
    if (e != hipSuccess) return fail(-EIO, "%s: %s", what, hipGetErrorString(e));
    return 0;
}
}  // namespace

extern "C" leco_stream_t leco_side_stream(void) { return (leco_stream_t)side_of(cur_dev()); }
extern "C" int leco_fork(leco_stream_t stream) {
    hipStream_t side = side_of(cur_dev());
    if (!side) return fail(-EIO, "leco_fork: cannot create the side stream");
    return edge((hipStream_t)stream, side, "leco_fork");
}
extern "C" int leco_join(leco_stream_t stream) {
    hipStream_t side = side_of(cur_dev());
    if (!side) return fail(-EIO, "leco_join: no side stream");
    return edge(side, (hipStream_t)stream, "leco_join");
}
