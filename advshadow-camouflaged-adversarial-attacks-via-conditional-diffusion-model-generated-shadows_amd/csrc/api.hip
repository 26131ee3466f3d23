// Library plumbing of the C ABI: error text, zero page, stream capture, events.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void advs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static void* g_zero[64] = {0};   // one zero page per device

const void* advs_zero_page() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return g_zero[dev];
}

extern "C" const char* advs_last_error(void) { return g_err; }
extern "C" int advs_abi_version(void) { return 1; }

extern "C" int advs_init(void) {
    int dev = 0;
    ADVS_HIP(hipGetDevice(&dev));
    ADVS_REQUIRE(dev >= 0 && dev < 64, "advs_init: device index %d out of range", dev);
    if (!g_zero[dev]) {
        void* p = nullptr;
        ADVS_HIP(hipMalloc(&p, 4096));
        ADVS_HIP(hipMemset(p, 0, 4096));
        ADVS_HIP(hipDeviceSynchronize());
        g_zero[dev] = p;
    }
    return ADVS_OK;
}

// ---- stream capture ---------------------------------------------------------------------
extern "C" int advs_graph_begin(void* stream) {
    ADVS_REQUIRE(stream, "graph_begin: capture needs an explicit (non-default) stream");
    ADVS_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return ADVS_OK;
}
extern "C" int advs_graph_end(void* stream, void** graph_exec_out) {
    ADVS_REQUIRE(stream && graph_exec_out, "graph_end: bad args");
    hipGraph_t g = nullptr;
    ADVS_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) ADVS_FAIL(ADVS_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    *graph_exec_out = (void*)ge;
    return ADVS_OK;
}
extern "C" int advs_graph_launch(void* graph_exec, void* stream) {
    ADVS_REQUIRE(graph_exec, "graph_launch: null graph");
    ADVS_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return ADVS_OK;
}
extern "C" int advs_graph_destroy(void* graph_exec) {
    if (graph_exec) ADVS_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return ADVS_OK;
}

// ---- events -------------------------------------------------------------------------------
extern "C" int advs_event_create(void** ev) {
    ADVS_REQUIRE(ev, "event_create: null");
    hipEvent_t e;
    ADVS_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return ADVS_OK;
}
extern "C" int advs_event_record(void* ev, void* stream) {
    ADVS_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return ADVS_OK;
}
extern "C" int advs_event_elapsed_ms(void* start, void* stop, float* ms) {
    ADVS_REQUIRE(ms, "event_elapsed_ms: null");
    ADVS_HIP(hipEventSynchronize((hipEvent_t)stop));
    ADVS_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ADVS_OK;
}
extern "C" int advs_event_destroy(void* ev) {
    if (ev) ADVS_HIP(hipEventDestroy((hipEvent_t)ev));
    return ADVS_OK;
}
extern "C" int advs_stream_sync(void* stream) {
    ADVS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return ADVS_OK;
}
