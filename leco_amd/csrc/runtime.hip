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
