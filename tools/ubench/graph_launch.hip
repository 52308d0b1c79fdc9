// Micro-benchmark (round 3): what does a hipGraph replay of a linear chain of short kernels cost per node on this stack, and which
// way of launching it matters?  K nodes of a ~3 us kernel (256 workgroups).  Variants:
//   plain      K launches on a non-blocking stream
//   graph      the same K launches captured from that stream and replayed on it
//   graph_fj   replayed on the side stream, forked from / joined into the legacy default stream by events (scg_rollout's way)
//   graph_ms   graph_fj + a hipMemsetD32Async on the side stream before the launch
//   graph_arg  graph whose kernels dereference a device word at their start (scalar load ahead of everything)
//   plain_null / graph_null / graph_null_ms   the chain, the graph, and memset + word-reading graph, all on the legacy default stream
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/graph_launch tools/ubench/graph_launch.hip ; run: tools/ubench/graph_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void spin(float* out, const unsigned* word, int iters) {
    unsigned add = word ? *word : 0u;
    float x = threadIdx.x + add;
    for (int i = 0; i < iters; ++i) x = fmaf(x, 0.999f, 0.001f);
    if (x == 12345.678f) out[0] = x;
}

int main() {
    const int K = 300, iters = 150;
    float* out; unsigned* word;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&word, 4)); CK(hipMemset(word, 0, 4));
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    auto chain = [&](hipStream_t st, const unsigned* w) { for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, out, w, iters); };
    hipGraphExec_t ex[2];
    for (int v = 0; v < 2; ++v) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        chain(s, v ? word : nullptr);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ex[v], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    for (int rep = 0; rep < 3; ++rep) {
        for (int variant = 0; variant < 10; ++variant) {
            const char* names[] = {"plain", "graph", "graph_fj", "graph_ms", "graph_arg", "plain_null", "graph_null", "graph_null_ms", "plain_fj", "plain_fj2"};
            CK(hipDeviceSynchronize());
            float ms = 0;
            const int R = 5;
            if (variant == 0) { CK(hipEventRecord(e0, s)); for (int r = 0; r < R; ++r) chain(s, nullptr); CK(hipEventRecord(e1, s)); }
            if (variant == 1) { CK(hipEventRecord(e0, s)); for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ex[0], s)); CK(hipEventRecord(e1, s)); }
            if (variant == 4) { CK(hipEventRecord(e0, s)); for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ex[1], s)); CK(hipEventRecord(e1, s)); }
            if (variant == 2 || variant == 3) {
                CK(hipEventRecord(e0, nullptr));
                for (int r = 0; r < R; ++r) {
                    CK(hipEventRecord(ef, nullptr)); CK(hipStreamWaitEvent(s, ef, 0));
                    if (variant == 3) CK(hipMemsetD32Async((hipDeviceptr_t)word, r, 1, s));
                    CK(hipGraphLaunch(ex[0], s));
                    CK(hipEventRecord(ej, s)); CK(hipStreamWaitEvent(nullptr, ej, 0));
                }
                CK(hipEventRecord(e1, nullptr));
            }
            if (variant == 5) { CK(hipEventRecord(e0, nullptr)); for (int r = 0; r < R; ++r) chain(nullptr, nullptr); CK(hipEventRecord(e1, nullptr)); }
            if (variant == 6 || variant == 7) {
                CK(hipEventRecord(e0, nullptr));
                for (int r = 0; r < R; ++r) {
                    if (variant == 7) CK(hipMemsetD32Async((hipDeviceptr_t)word, r, 1, nullptr));
                    CK(hipGraphLaunch(ex[variant == 7 ? 1 : 0], nullptr));
                }
                CK(hipEventRecord(e1, nullptr));
            }
            if (variant == 8 || variant == 9) {      // plain launches on the side stream, forked from / joined into the null (8) or another non-blocking (9) stream
                hipStream_t origin = variant == 8 ? nullptr : s2;
                CK(hipEventRecord(e0, origin));
                for (int r = 0; r < R; ++r) {
                    CK(hipEventRecord(ef, origin)); CK(hipStreamWaitEvent(s, ef, 0));
                    chain(s, nullptr);
                    CK(hipEventRecord(ej, s)); CK(hipStreamWaitEvent(origin, ej, 0));
                }
                CK(hipEventRecord(e1, origin));
            }
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-10s %.2f us per node\n", names[variant], ms * 1e3 / (R * K));
        }
        printf("--\n");
    }
    return 0;
}
