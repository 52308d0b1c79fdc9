// HOST-ONLY stand-in for <hip/hip_runtime.h>, used by ONE thing: the CPU sanitizer build of the C-ABI validation layer
// (tests/hostshim/build.sh compiles safe_control_gym_amd/csrc/scg_capi.hip as plain C++ with -fsanitize=address,undefined).
// "Device" memory is host heap memory, so AddressSanitizer checks every table the layer builds and copies; streams and
// events are counters, so the driver can assert that scg_rollout joins its second launch queue on every error path.
// The kernels themselves are NOT here (GPU sanitizers are unavailable on this pool): launches are recorded by
// tests/hostshim/hip_host_shim.cc and may be told to fail.  Never part of the product build.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorLaunchFailure = 719, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };

struct ihipStream_t { int id; int pending; int capturing; int captured_nodes; };
struct ihipEvent_t { int id; ihipStream_t* recorded_on; };
typedef ihipStream_t* hipStream_t;
typedef ihipEvent_t* hipEvent_t;

struct hipDeviceProp_t { int multiProcessorCount; size_t sharedMemPerBlock; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDevice(int* dev);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t attr, int dev);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize(void);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);

// stream capture / graphs (scg_rollout's replay of small-batch rollouts): a graph is a node count
struct ihipGraph_t { int nodes; };
struct ihipGraphExec_t { int nodes; };
typedef ihipGraph_t* hipGraph_t;
typedef ihipGraphExec_t* hipGraphExec_t;
typedef void* hipDeviceptr_t;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* status);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* graph);
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, void* error_node, char* log, size_t log_size);
hipError_t hipGraphDestroy(hipGraph_t graph);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s);
hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t s);

template <class T> hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
