// Bookkeeping of the host-only HIP stand-in (see hip/hip_runtime.h in this directory).
#pragma once
#include <cstdint>
#include "../../include/scg_rollout.h"
struct ShimState {
    int64_t live_allocations;       // hipMalloc - hipFree
    int64_t launches;               // launch_* calls recorded
    int64_t step_launches_main, step_launches_aux;   // launch_step per stream kind (caller's stream = null here)
    int fail_launch_after;          // >= 0: the (fail_launch_after + 1)-th launch_step returns hipErrorLaunchFailure
    int fail_event_record_after;    // >= 0: the n-th hipEventRecord fails
    int64_t event_records, stream_waits, stream_syncs;
    int aux_streams_open;           // work recorded on an aux stream that has not been joined (waited on / synchronised)
    int64_t last_n;                 // n of the last launch_step
    int64_t last_env_base;
    // stream capture / graphs
    int captures_open;              // hipStreamBeginCapture - hipStreamEndCapture: must be 0 whenever the layer returns
    int graphs_live, execs_live;    // hipGraph_t / hipGraphExec_t objects alive
    int64_t captured_nodes;         // launch_step calls recorded into a capture (not executed)
    int64_t graph_launches, graph_nodes_replayed, serial_memsets;
    int64_t last_serial;            // value of the last hipMemsetD32Async
    int fail_begin_capture, fail_instantiate;       // != 0: that call fails
    int caller_stream_capturing;    // != 0: hipStreamIsCapturing(null stream) reports an active capture
    int current_device;             // what hipGetDevice reports (survives shim_reset: the "current device" of the process)
    int64_t async_copies, last_async_copy_bytes;    // hipMemcpyAsync
    int64_t event_syncs;            // hipEventSynchronize
    int fail_event_sync;            // != 0: hipEventSynchronize fails (an exec must then be leaked, not destroyed)
    int64_t fused_launches;
    const float* last_fused_sink_obs;
    const uint32_t* last_serial_base;   // KParams.serial_base of the last launch_step when F_SERIAL_WORD is set, else null
    scg_step_io last_io;            // the scg_step_io of the last launch_step
    void* last_exec;                // the last hipGraphExec_t instantiated (so that a test can free one the layer leaked on purpose)
};
extern ShimState g_shim;
void shim_reset(void);
