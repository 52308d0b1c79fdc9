// Bookkeeping of the host-only HIP stand-in (see hip/hip_runtime.h in this directory).
#pragma once
#include <cstdint>
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
};
extern ShimState g_shim;
void shim_reset(void);
