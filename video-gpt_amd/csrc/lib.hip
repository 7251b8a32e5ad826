// Library-level pieces of libvgpt_hip.so: thread-local error string, ABI version and the hipGraph
// helpers used to capture the sampler loop (LVM/scheduler.py:168-204) into one replayable graph.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vgpt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

VGPT_EXPORT const char* vgpt_last_error(void) { return g_err; }
VGPT_EXPORT int vgpt_abi_version(void) { return VGPT_ABI_VERSION; }

VGPT_EXPORT int vgpt_graph_begin_capture(void* stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        vgpt_set_error("vgpt_graph_begin_capture: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_graph_end_capture(void* stream, void** graph_exec_out) {
    VGPT_REQUIRE(graph_exec_out, VGPT_ERR_INVALID, "vgpt_graph_end_capture: null pointer");
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess || !graph) {
        vgpt_set_error("vgpt_graph_end_capture: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);  // the executable graph keeps what it needs
    if (e != hipSuccess) {
        vgpt_set_error("vgpt_graph_end_capture: instantiate: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    *graph_exec_out = (void*)exec;
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_graph_launch(void* graph_exec, void* stream) {
    VGPT_REQUIRE(graph_exec, VGPT_ERR_INVALID, "vgpt_graph_launch: null graph");
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        vgpt_set_error("vgpt_graph_launch: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    return VGPT_OK;
}

VGPT_EXPORT int vgpt_graph_destroy(void* graph_exec) {
    if (!graph_exec) return VGPT_OK;
    hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    if (e != hipSuccess) {
        vgpt_set_error("vgpt_graph_destroy: %s", hipGetErrorString(e));
        return VGPT_ERR_HIP;
    }
    return VGPT_OK;
}
