// Host-only HIP stand-in + recording kernel launchers (test infrastructure: CPU sanitizer build of scg_capi.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../safe_control_gym_amd/csrc/scg_device.h"
#include "hip_host_shim.h"

ShimState g_shim;
void shim_reset(void) {          // everything but the allocation balance, which spans the life of a handle
    const int64_t live = g_shim.live_allocations;
    const int dev = g_shim.current_device;
    void* exec = g_shim.last_exec;
    std::memset(&g_shim, 0, sizeof(g_shim));
    g_shim.last_exec = exec;
    g_shim.live_allocations = live; g_shim.current_device = dev; g_shim.fail_launch_after = -1; g_shim.fail_event_record_after = -1;
}

static int g_next_id = 1;
static hipError_t g_last = hipSuccess;

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorLaunchFailure ? "launch failure (injected)" : "error"; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
hipError_t hipGetDevice(int* dev) { *dev = g_shim.current_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; p->sharedMemPerBlock = 65536; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t, int) { *value = 256; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t bytes) { *p = std::malloc(bytes ? bytes : 1); if (!*p) return hipErrorOutOfMemory; ++g_shim.live_allocations; return hipSuccess; }
hipError_t hipFree(void* p) { if (p) { std::free(p); --g_shim.live_allocations; } return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { std::memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipMemset(void* dst, int value, size_t bytes) { std::memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) {
    ++g_shim.async_copies; g_shim.last_async_copy_bytes = (int64_t)bytes;
    std::memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new ihipStream_t{g_next_id++, 0, 0, 0}; return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* status) {
    *status = (s ? s->capturing : g_shim.caller_stream_capturing) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
    if (!s || g_shim.fail_begin_capture) return hipErrorInvalidValue;      // the legacy default stream cannot be captured
    s->capturing = 1; s->captured_nodes = 0; ++g_shim.captures_open;
    return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* graph) {
    *graph = nullptr;
    if (!s || !s->capturing) return hipErrorInvalidValue;
    s->capturing = 0; --g_shim.captures_open;
    *graph = new ihipGraph_t{s->captured_nodes}; ++g_shim.graphs_live;
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, void*, char*, size_t) {
    *exec = nullptr;
    if (g_shim.fail_instantiate) return hipErrorOutOfMemory;
    *exec = new ihipGraphExec_t{graph->nodes}; ++g_shim.execs_live; g_shim.last_exec = *exec;
    return hipSuccess;
}
hipError_t hipGraphDestroy(hipGraph_t graph) { delete graph; --g_shim.graphs_live; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t exec) { delete exec; --g_shim.execs_live; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s) {
    ++g_shim.graph_launches; g_shim.graph_nodes_replayed += exec->nodes;
    if (s && !s->pending) { s->pending = 1; ++g_shim.aux_streams_open; }
    return hipSuccess;
}
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t s) {
    for (size_t q = 0; q < count; ++q) static_cast<int*>(dst)[q] = value;
    ++g_shim.serial_memsets; g_shim.last_serial = (uint32_t)value;
    if (s && !s->pending) { s->pending = 1; ++g_shim.aux_streams_open; }
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { ++g_shim.stream_syncs; if (s && s->pending) { s->pending = 0; --g_shim.aux_streams_open; } return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new ihipEvent_t{g_next_id++, nullptr}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    if (g_shim.fail_event_record_after >= 0 && g_shim.event_records >= g_shim.fail_event_record_after) { ++g_shim.event_records; return hipErrorInvalidValue; }
    ++g_shim.event_records;
    e->recorded_on = s;
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    ++g_shim.event_syncs;
    if (g_shim.fail_event_sync) return hipErrorInvalidValue;
    (void)e;
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t waiter, hipEvent_t e, unsigned) {
    ++g_shim.stream_waits;
    // the caller's stream (null here) waiting on an event recorded on an aux stream = that stream is joined
    if (waiter == nullptr && e->recorded_on && e->recorded_on->pending) { e->recorded_on->pending = 0; --g_shim.aux_streams_open; }
    return hipSuccess;
}

namespace scg {
LaunchGeom step_geometry(const KParams& p, int, int64_t n, int num_cus) {
    LaunchGeom g; g.block = 256; int64_t b = (n + 255) / 256;
    if (p.grid_cap_per_cu > 0 && b > (int64_t)num_cus * p.grid_cap_per_cu) b = (int64_t)num_cus * p.grid_cap_per_cu;
    g.grid = (int32_t)(b < 1 ? 1 : b); g.lds_bytes = 0; return g;
}
hipError_t launch_step(int, const KParams& p, const scg_step_io& io, const KTables*, const ConTable*, const float*, int64_t n, int, hipStream_t s) {
    g_shim.last_io = io; g_shim.last_serial_base = (p.flags & scg::F_SERIAL_WORD) ? p.serial_base : nullptr;
    if (g_shim.fail_launch_after >= 0 && g_shim.step_launches_main + g_shim.step_launches_aux + g_shim.captured_nodes >= g_shim.fail_launch_after) return hipErrorLaunchFailure;
    if (s && s->capturing) { ++s->captured_nodes; ++g_shim.captured_nodes; g_shim.last_n = n; return hipSuccess; }   // recorded, not executed
    ++g_shim.launches; g_shim.last_n = n; g_shim.last_env_base = p.env_base;
    if (s) { ++g_shim.step_launches_aux; if (!s->pending) { s->pending = 1; ++g_shim.aux_streams_open; } } else ++g_shim.step_launches_main;
    return hipSuccess;
}
bool fused_rollout_ok(const KParams&, const scg_step_io& io) { return !io.obs && !io.state_lo; }
hipError_t launch_rollout_fused(int, const KParams&, const scg_step_io&, const KTables*, const ConTable*, const float*, int64_t, const RolloutArgs& ro, int, hipStream_t) {
    ++g_shim.launches; ++g_shim.fused_launches; g_shim.last_fused_sink_obs = ro.sink_obs;
    return hipSuccess;
}
hipError_t launch_reset(int, const KParams&, const KTables*, float*, float*, int32_t*, float*, int32_t*, float*, const uint8_t*, const float*, float*, uint32_t*, const float*, uint32_t, int64_t, int64_t, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_gae(const float*, const float*, const float*, const uint8_t*, const float*, const float*, float*, float*, float*, double*, int, int64_t, int64_t, float, float, int, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_normalize(float*, const double*, int, int64_t, int64_t, float, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_copy_segments(const scg_copy_segment*, int, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_model_eval(int, const KParams&, const KTables*, const float*, const float*, const float*, float*, float*, int64_t, int64_t, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_model_flow(int, const PhysT<double>&, const double*, const double*, const double*, double, double, double, int, double*, double*, double*, int32_t*, int64_t, int64_t, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_linearize(int, const PhysT<float>&, const float*, const float*, const float*, float*, float*, float, int64_t, int64_t, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_fc_linear(int, const PhysT<float>&, const float*, const float*, const float*, const float*, const float*, float*, int64_t, int64_t, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_discretize_exact(const float*, const float*, int, int, float, float*, float*, const float*, const float*, float*, int64_t, int, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_loss(int, const double*, const double*, const float*, const float*, const float*, int64_t, const float*, int64_t, float*, float*, float*, int64_t, int64_t, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_mean_std_normalize(const float*, float*, int, int64_t, int64_t, int64_t, const double*, double*, double*, float, double, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_reward_normalize(const float*, const uint8_t*, float*, float*, int64_t, const double*, double*, double*, float, float, double, int, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_mean_std_moments(const float*, int, int64_t, int64_t, const double*, double*, double*, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_mean_std_apply_sums(const float*, float*, int, int64_t, int64_t, int64_t, const double*, double*, const double*, double, float, double, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_reward_moments(const float*, const uint8_t*, float*, int64_t, const double*, double*, double*, float, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_reward_apply_sums(const float*, float*, int64_t, const double*, double*, const double*, double, float, double, hipStream_t) { ++g_shim.launches; return hipSuccess; }
hipError_t launch_eval_constraints(int, const ConTable*, const float*, const float*, float*, uint8_t*, int64_t, int64_t, hipStream_t) { ++g_shim.launches; return hipSuccess; }
}  // namespace scg
