// Micro-benchmark (round 3): what does the OUTPUT PATTERN of scg_linearize (3-D: per env a 576-byte dfdx record and a 192-byte dfdu
// record, row-major batches; a wave owns 64 consecutive envs = 36 KiB + 12 KiB) reach as a pure write stream, against a linear fill of
// the same 805 MB?  Separates "the pattern" from "the staging" in the kernel's 4.9 TB/s.
//   linear / linear_nt      grid-stride float4 stores over one 805 MB buffer (plain / nontemporal)
//   records / records_nt    every wave writes ITS 36 KiB + 12 KiB chunks as 36 + 12 float4 store instructions of 1 KiB each
//   staged_nt               records_nt with the kernel's LDS round trip in front of every pass (16 lanes write a 9 KiB region, wave
//                           barrier, 64 lanes read it back and store it), data made up in registers
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/write_pattern tools/ubench/write_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ void st(v4 v, v4* p) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT>
__global__ __launch_bounds__(256) void linear(v4* out, size_t units) {
    const v4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < units; q += (size_t)gridDim.x * 256) st<NT>(v, out + q);
}

// a workgroup writes ONE contiguous tile of 256 * K float4 and exits (the shape of torch's vectorized elementwise kernels)
template <int K, bool NT>
__global__ __launch_bounds__(256) void tiles(v4* out, size_t units) {
    const v4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    const size_t base = (size_t)blockIdx.x * (256 * K) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (base + k * 256 < units) st<NT>(v, out + base + k * 256);
}
// each thread writes K CONSECUTIVE float4 (64 B per thread for K = 4)
template <int K, bool NT>
__global__ __launch_bounds__(256) void thread_runs(v4* out, size_t units) {
    const v4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * K;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (base + k < units) st<NT>(v, out + base + k);
}

// every wave writes ONE contiguous chunk of K KiB (K stores of 1 KiB) and exits: how many stores per wave can a chunked layout afford?
template <int K>
__global__ __launch_bounds__(256) void chunks(v4* out, size_t units) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const v4 v = {1.f, 2.f, 3.f, (float)lane};
    v4* p = out + w * (64 * K);
    if ((w + 1) * (64 * K) > units) return;
#pragma unroll
    for (int t = 0; t < K; ++t) *(p + t * 64 + lane) = v;
}

// one 1 KiB store per wave, one 4 KiB block per workgroup -- but WHICH 4 KiB block: does it matter that workgroup b (dispatched to
// XCD b mod 8) writes block b?   MODE 0: block b;  1: block b + 1 (every workgroup off by one);  2: b's low three bits permuted
// (x -> 3x mod 8);  3: XCD x writes the x-th contiguous eighth of the buffer;  4: block b ^ 1;  5: blocks of 8 KiB granularity
// (index (b / 16) * 16 + (b % 8) * 2 + ((b / 8) % 2)): residue mod 8 of the 4-KiB index = 2 (b % 8) + ... , i.e. XCD x writes 8-KiB unit x
template <int MODE>
__global__ __launch_bounds__(256) void placed(v4* out, size_t units) {
    const size_t nb = units / 256;
    size_t b = blockIdx.x;
    if (MODE == 1) b = (b + 1) % nb;
    if (MODE == 2) b = (b & ~(size_t)7) | ((b * 3) & 7);
    if (MODE == 3) b = (b >> 3) + (b & 7) * (nb >> 3);
    if (MODE == 4) b = b ^ 1;
    if (MODE == 5) b = (b / 16) * 16 + (b % 8) * 2 + ((b / 8) % 2);
    if (b >= nb) return;
    const v4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    out[b * 256 + threadIdx.x] = v;
}

// the records layout written COOPERATIVELY: a workgroup owns 256 envs (144 KiB of dfdx + 48 KiB of dfdu); in each of 16 passes its four
// waves together write the next 9 KiB + 3 KiB (wave j the KiB j, j + 4, j + 8 of the pass), a workgroup barrier between passes (where
// the kernel would restage LDS).  SYNC = false drops the barriers.
template <bool SYNC>
__global__ __launch_bounds__(256) void coop_passes(v4* a, v4* b, uint32_t blocks) {
    if (blockIdx.x >= blocks) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const v4 v = {1.f, 2.f, 3.f, (float)lane};
    v4* pa = a + (size_t)blockIdx.x * (256 * 144 / 4);
    v4* pb = b + (size_t)blockIdx.x * (256 * 48 / 4);
#pragma unroll 1
    for (int p = 0; p < 16; ++p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kib = (int)wave + 4 * k;
            if (kib < 9) pa[(p * 9 + kib) * 64 + lane] = v;
        }
        if (wave < 3) pb[(p * 3 + wave) * 64 + lane] = v;
        if (SYNC) __syncthreads();
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void records(v4* a, v4* b, uint32_t waves) {
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= waves) return;
    const v4 v = {1.f, 2.f, 3.f, (float)lane};
    v4* pa = a + (size_t)w * (64 * 144 / 4);
    v4* pb = b + (size_t)w * (64 * 48 / 4);
#pragma unroll
    for (int t = 0; t < 36; ++t) st<NT>(v, pa + t * 64 + lane);
#pragma unroll
    for (int t = 0; t < 12; ++t) st<NT>(v, pb + t * 64 + lane);
}

__global__ __launch_bounds__(256) void staged_nt(v4* a, v4* b, uint32_t waves, float seed) {
    extern __shared__ float lds_all[];
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= waves) return;
    float* lds = lds_all + (threadIdx.x >> 6) * (16 * 144);
    v4* pa = a + (size_t)w * (64 * 144 / 4);
    v4* pb = b + (size_t)w * (64 * 48 / 4);
    float vals[144];
#pragma unroll
    for (int k = 0; k < 144; ++k) vals[k] = seed * (float)(k + 1) + (float)lane;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if ((lane >> 4) == h) {
            float* mine = lds + (lane & 15) * 144;
#pragma unroll
            for (int k = 0; k < 144; k += 4) *reinterpret_cast<v4*>(mine + k) = v4{vals[k], vals[k + 1], vals[k + 2], vals[k + 3]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < 9; ++t) st<true>(*reinterpret_cast<const v4*>(lds + (t * 64 + lane) * 4), pa + h * 576 + t * 64 + lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
            float* mine = lds + (lane & 31) * 48;
#pragma unroll
            for (int k = 0; k < 48; k += 4) *reinterpret_cast<v4*>(mine + k) = v4{vals[k], vals[k + 1], vals[k + 2], vals[k + 3]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < 6; ++t) st<true>(*reinterpret_cast<const v4*>(lds + (t * 64 + lane) * 4), pb + h * 384 + t * 64 + lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
}

int main() {
    const uint32_t n = 1u << 20, waves = n / 64;
    const size_t bytes_a = (size_t)n * 576, bytes_b = (size_t)n * 192;
    char* buf; CK(hipMalloc(&buf, bytes_a + bytes_b));
    v4* a = reinterpret_cast<v4*>(buf); v4* b = reinterpret_cast<v4*>(buf + bytes_a);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t units = (bytes_a + bytes_b) / 16;
    const char* names[] = {"linear", "linear_nt", "records", "records_nt", "staged_nt", "tiles4", "tiles4_nt", "tiles1", "tiles16", "runs4", "linear_g2048", "linear_g65536", "hipMemsetD32", "chunks48", "chunks24", "chunks12", "chunks6", "chunks3", "chunks1", "placed_b", "placed_b+1", "placed_perm3", "placed_eighths", "placed_b^1", "placed_8KiB", "coop_passes", "coop_nosync"};
    for (int rep = 0; rep < 3; ++rep) {
        for (int v = 0; v < 27; ++v) {
            const int R = 20;
            for (int r = -2; r < R; ++r) {
                if (r == 0) CK(hipEventRecord(e0, nullptr));
                if (v == 0) hipLaunchKernelGGL(linear<false>, dim3(8192), dim3(256), 0, nullptr, a, units);
                if (v == 1) hipLaunchKernelGGL(linear<true>, dim3(8192), dim3(256), 0, nullptr, a, units);
                if (v == 2) hipLaunchKernelGGL(records<false>, dim3(waves / 4), dim3(256), 0, nullptr, a, b, waves);
                if (v == 3) hipLaunchKernelGGL(records<true>, dim3(waves / 4), dim3(256), 0, nullptr, a, b, waves);
                if (v == 5) hipLaunchKernelGGL((tiles<4, false>), dim3((unsigned)((units + 1023) / 1024)), dim3(256), 0, nullptr, a, units);
                if (v == 6) hipLaunchKernelGGL((tiles<4, true>), dim3((unsigned)((units + 1023) / 1024)), dim3(256), 0, nullptr, a, units);
                if (v == 7) hipLaunchKernelGGL((tiles<1, false>), dim3((unsigned)((units + 255) / 256)), dim3(256), 0, nullptr, a, units);
                if (v == 8) hipLaunchKernelGGL((tiles<16, false>), dim3((unsigned)((units + 4095) / 4096)), dim3(256), 0, nullptr, a, units);
                if (v == 9) hipLaunchKernelGGL((thread_runs<4, false>), dim3((unsigned)((units + 1023) / 1024)), dim3(256), 0, nullptr, a, units);
                if (v == 10) hipLaunchKernelGGL(linear<false>, dim3(2048), dim3(256), 0, nullptr, a, units);
                if (v == 11) hipLaunchKernelGGL(linear<false>, dim3(65536), dim3(256), 0, nullptr, a, units);
                if (v == 12) CK(hipMemsetD32Async((hipDeviceptr_t)a, 0x3fc00000, units * 4, nullptr));
                if (v == 13) hipLaunchKernelGGL((chunks<48>), dim3((unsigned)(units / (64 * 48) / 4)), dim3(256), 0, nullptr, a, units);
                if (v == 14) hipLaunchKernelGGL((chunks<24>), dim3((unsigned)(units / (64 * 24) / 4)), dim3(256), 0, nullptr, a, units);
                if (v == 15) hipLaunchKernelGGL((chunks<12>), dim3((unsigned)(units / (64 * 12) / 4)), dim3(256), 0, nullptr, a, units);
                if (v == 16) hipLaunchKernelGGL((chunks<6>), dim3((unsigned)(units / (64 * 6) / 4)), dim3(256), 0, nullptr, a, units);
                if (v == 17) hipLaunchKernelGGL((chunks<3>), dim3((unsigned)(units / (64 * 3) / 4)), dim3(256), 0, nullptr, a, units);
                if (v == 18) hipLaunchKernelGGL((chunks<1>), dim3((unsigned)(units / 64 / 4)), dim3(256), 0, nullptr, a, units);
                if (v == 19) hipLaunchKernelGGL((placed<0>), dim3((unsigned)(units / 256)), dim3(256), 0, nullptr, a, units);
                if (v == 20) hipLaunchKernelGGL((placed<1>), dim3((unsigned)(units / 256)), dim3(256), 0, nullptr, a, units);
                if (v == 21) hipLaunchKernelGGL((placed<2>), dim3((unsigned)(units / 256)), dim3(256), 0, nullptr, a, units);
                if (v == 22) hipLaunchKernelGGL((placed<3>), dim3((unsigned)(units / 256)), dim3(256), 0, nullptr, a, units);
                if (v == 23) hipLaunchKernelGGL((placed<4>), dim3((unsigned)(units / 256)), dim3(256), 0, nullptr, a, units);
                if (v == 24) hipLaunchKernelGGL((placed<5>), dim3((unsigned)(units / 256)), dim3(256), 0, nullptr, a, units);
                if (v == 25) hipLaunchKernelGGL((coop_passes<true>), dim3(n / 256), dim3(256), 0, nullptr, a, b, n / 256);
                if (v == 26) hipLaunchKernelGGL((coop_passes<false>), dim3(n / 256), dim3(256), 0, nullptr, a, b, n / 256);
                if (v == 4) hipLaunchKernelGGL(staged_nt, dim3(waves / 4), dim3(256), 4 * 16 * 144 * 4, nullptr, a, b, waves, 0.5f);
            }
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("rep %d %-12s %8.2f us  %7.1f GB/s\n", rep, names[v], ms / R * 1e3, (bytes_a + bytes_b) / (ms / R * 1e-3) / 1e9);
        }
    }
    return 0;
}
