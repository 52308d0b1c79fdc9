// Micro-benchmark: is v_pk_fma_f32 (2 fp32 FMAs per lane per instruction) faster per FLOP than
// v_fma_f32 on gfx950?  Decides whether a 2-envs-per-lane packed formulation of the step kernel pays.
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_fma tools/ubench/pk_fma.hip ; run: ./pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int ITERS>
__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int ITERS>
__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b) {
    float2v A = {a, a}, B = {b, b};
    float2v x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int i = 0; i < ITERS; ++i) {
        x0 = __builtin_elementwise_fma(x0, A, B); x1 = __builtin_elementwise_fma(x1, A, B);
        x2 = __builtin_elementwise_fma(x2, A, B); x3 = __builtin_elementwise_fma(x3, A, B);
    }
    float2v s = x0 + x1 + x2 + x3;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
int main() {
    constexpr int ITERS = 4096;
    const int blocks = 256 * 8;
    float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(e0); hipLaunchKernelGGL(k_scalar<ITERS>, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        double fl = 2.0 * 8 * ITERS * blocks * 256;
        printf("scalar v_fma_f32   : %.3f ms  %.1f TFLOP/s\n", ms, fl / ms * 1e-9);
        hipEventRecord(e0); hipLaunchKernelGGL(k_packed<ITERS>, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("packed v_pk_fma_f32: %.3f ms  %.1f TFLOP/s\n", ms, fl / ms * 1e-9);
    }
    return 0;
}
