// Micro-benchmark: memory-side floor of the step kernel's access pattern under different plane widths.
//   A: struct-of-arrays, 4 B per lane and plane -- 17 read streams (12 state + 4 action + counter),
//      15 write streams (12 state + reward + done u8 + counter)         [the shipped layout]
//   B: array-of-structs-of-arrays, 16 B per lane -- state as 3 float4 planes, action as 1 float4 plane,
//      scalars unchanged: 5 read streams, 6 write streams
//   C: like A but every lane owns 4 consecutive envs and moves float4 per plane (SoA in memory, 16 B per lane)
// All move 125 B per env.  build: hipcc --offload-arch=gfx950 -O3 -o streams tools/ubench/streams.hip ; run: ./streams
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_soa(const float* __restrict__ st, const float* __restrict__ ac,
                                             int* __restrict__ t, float* __restrict__ so, float* __restrict__ rew,
                                             uint8_t* __restrict__ done, int64_t ld, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float x[12], a[4];
#pragma unroll
    for (int j = 0; j < 12; ++j) x[j] = __builtin_nontemporal_load(st + j * ld + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = __builtin_nontemporal_load(ac + j * ld + i);
    const int tt = __builtin_nontemporal_load(t + i);
    const float s = (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int j = 0; j < 12; ++j) __builtin_nontemporal_store(fmaf(x[j], 0.999f, s), so + j * ld + i);
    __builtin_nontemporal_store(s, rew + i);
    __builtin_nontemporal_store((uint8_t)(tt > 299), done + i);
    __builtin_nontemporal_store(tt + 1, t + i);
}

__global__ __launch_bounds__(256) void k_aosoa(const f4* __restrict__ st, const f4* __restrict__ ac,
                                               int* __restrict__ t, f4* __restrict__ so, float* __restrict__ rew,
                                               uint8_t* __restrict__ done, int64_t ld, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f4 x[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) x[j] = __builtin_nontemporal_load(st + j * ld + i);
    const f4 a = __builtin_nontemporal_load(ac + i);
    const int tt = __builtin_nontemporal_load(t + i);
    const float s = (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int j = 0; j < 3; ++j) __builtin_nontemporal_store(x[j] * 0.999f + s, so + j * ld + i);
    __builtin_nontemporal_store(s, rew + i);
    __builtin_nontemporal_store((uint8_t)(tt > 299), done + i);
    __builtin_nontemporal_store(tt + 1, t + i);
}

// SoA in memory, 4 consecutive envs per lane
__global__ __launch_bounds__(256) void k_soa4(const float* __restrict__ st, const float* __restrict__ ac,
                                              int* __restrict__ t, float* __restrict__ so, float* __restrict__ rew,
                                              uint8_t* __restrict__ done, int64_t ld, uint32_t n) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;          // group of 4 envs
    if (q * 4u >= n) return;
    f4 x[12], a[4];
#pragma unroll
    for (int j = 0; j < 12; ++j) x[j] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(st + j * ld) + q);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(ac + j * ld) + q);
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4 tt = __builtin_nontemporal_load(reinterpret_cast<const i4*>(t) + q);
    const f4 s = (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int j = 0; j < 12; ++j) __builtin_nontemporal_store(x[j] * 0.999f + s, reinterpret_cast<f4*>(so + j * ld) + q);
    __builtin_nontemporal_store(s, reinterpret_cast<f4*>(rew) + q);
    uint32_t d = (tt[0] > 299) | ((tt[1] > 299) << 8) | ((tt[2] > 299) << 16) | ((tt[3] > 299) << 24);
    __builtin_nontemporal_store(d, reinterpret_cast<uint32_t*>(done) + q);
    __builtin_nontemporal_store(tt + 1, reinterpret_cast<i4*>(t) + q);
}

// D: SoA in memory, one env per lane for the arithmetic, but every global access is 16 B per lane: wave w of the workgroup
// moves whole planes' 1-KB chunks (256 envs x 4 B) with one float4 per lane and the lanes meet their values in LDS.
__global__ __launch_bounds__(256) void k_lds(const float* __restrict__ st, const float* __restrict__ ac,
                                             int* __restrict__ t, float* __restrict__ so, float* __restrict__ rew,
                                             uint8_t* __restrict__ done, int64_t ld, uint32_t n) {
    __shared__ f4 in_[17][64];
    __shared__ f4 out_[14][64];
    const uint32_t tid = threadIdx.x, w = tid >> 6, l = tid & 63u;
    const uint32_t base = blockIdx.x * 256u;                     // n is a multiple of 256 here
    for (uint32_t j = w; j < 17; j += 4) {
        const float* src = j < 12 ? st + j * ld : (j < 16 ? ac + (j - 12) * ld : reinterpret_cast<const float*>(t));
        in_[j][l] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + base) + l);
    }
    __syncthreads();
    const float* inf = reinterpret_cast<const float*>(in_);
    float x[12], a[4];
#pragma unroll
    for (int j = 0; j < 12; ++j) x[j] = inf[j * 256 + tid];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = inf[(12 + j) * 256 + tid];
    const int tt = __float_as_int(inf[16 * 256 + tid]);
    const float s = (a[0] + a[1]) + (a[2] + a[3]);
    float* outf = reinterpret_cast<float*>(out_);
#pragma unroll
    for (int j = 0; j < 12; ++j) outf[j * 256 + tid] = fmaf(x[j], 0.999f, s);
    outf[12 * 256 + tid] = s;
    outf[13 * 256 + tid] = __int_as_float(tt + 1);
    __builtin_nontemporal_store((uint8_t)(tt > 299), done + base + tid);
    __syncthreads();
    for (uint32_t j = w; j < 14; j += 4) {
        float* dst = j < 12 ? so + j * ld : (j == 12 ? rew : reinterpret_cast<float*>(t));
        __builtin_nontemporal_store(out_[j][l], reinterpret_cast<f4*>(dst + base) + l);
    }
}

int main() {
    const uint32_t n = 1u << 20;
    const int64_t ld = n + 17 * 64;
    float *st, *so, *ac, *rew; int* t; uint8_t* done;
    hipMalloc(&st, 12 * ld * 4); hipMalloc(&so, 12 * ld * 4); hipMalloc(&ac, 4 * ld * 4); hipMalloc(&rew, ld * 4);
    hipMalloc(&t, ld * 4); hipMalloc(&done, ld);
    hipMemset(st, 0, 12 * ld * 4); hipMemset(ac, 0, 4 * ld * 4); hipMemset(t, 0, ld * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    const double bytes = 125.0 * n;
    for (int round = 0; round < 3; ++round) {
        float ms;
        for (int variant = 0; variant < 5; ++variant) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                // ping-pong state buffers so that reads and writes hit different addresses, as in-place would not matter here
                float* a_in = (r & 1) ? so : st; float* a_out = (r & 1) ? st : so;
                if (variant == 0) hipLaunchKernelGGL(k_soa, dim3(n / 256), dim3(256), 0, 0, a_in, ac, t, a_out, rew, done, ld, n);
                else if (variant == 1) hipLaunchKernelGGL(k_soa, dim3(n / 256), dim3(256), 0, 0, a_in, ac, t, a_in, rew, done, ld, n);
                else if (variant == 2) hipLaunchKernelGGL(k_aosoa, dim3(n / 256), dim3(256), 0, 0, (const f4*)a_in, (const f4*)ac, t, (f4*)a_in, rew, done, ld, n);
                else if (variant == 4) hipLaunchKernelGGL(k_lds, dim3(n / 256), dim3(256), 0, 0, a_in, ac, t, a_in, rew, done, ld, n);
                else hipLaunchKernelGGL(k_soa4, dim3(n / 1024), dim3(256), 0, 0, a_in, ac, t, a_in, rew, done, ld, n);
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const char* names[5] = {"A  SoA 4B/lane, out-of-place", "A' SoA 4B/lane, in-place    ", "B  AoSoA float4 planes      ", "C  SoA, 4 envs per lane     ", "D  SoA, float4 via LDS      "};
            printf("%s : %.2f us/launch  %.2f TB/s\n", names[variant], ms / reps * 1e3, bytes / (ms / reps * 1e-3) * 1e-12);
        }
    }
    return 0;
}
