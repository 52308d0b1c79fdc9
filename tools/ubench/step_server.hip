// Probe for VERDICT r1 item 6: can a PERSISTENT "step server" beat one kernel launch per step for small batches?
//
// Workload: the C2 shape -- CartPole, N envs (default 65536), one RK4 step per env and step, 4 state planes + action + counter
// in, 4 state planes + reward + done + counter out (49 B per env-step), the same f_eval / integrate device code as the engine.
//   A  launches        one kernel per step, all queued on one stream (what scg_rollout / env.step() do today)
//   A2 launch + sync   one kernel per step, hipStreamSynchronize after each (a host policy in the loop)
//   B  server, device  ONE resident grid (one workgroup per 256 envs); per step every workgroup waits for a doorbell word, steps
//                      its envs, arrives on one of 8 sharded counters; a one-wave "ringer" kernel on a second stream plays the
//                      caller: ring doorbell k, wait for all arrivals, ring k + 1 ... (a device-side policy in the loop)
//   C  server, host    the same server; the HOST rings the doorbell (pinned, mapped host memory) and waits for a completion
//                      word the last-arriving workgroup writes to pinned host memory (a host policy in the loop)
// Hand-off rules: cdna_hip_programming.md Guideline 16 (relaxed agent / system scope polls by ONE lane, s_sleep between
// polls, every storing wave drains, one lane releases and arrives; every spin is bounded and sets a timeout word).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o step_server tools/ubench/step_server.hip ; run: ./step_server [N] [steps]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../safe_control_gym_amd/csrc/scg_dynamics.h"

using namespace scg;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

struct Ctl {                       // device words, 128 B apart
    unsigned doorbell; unsigned pad0[31];
    unsigned done[8][32];          // 8 shards, one line each
    unsigned timeout; unsigned pad1[31];
    unsigned last; unsigned pad2[31];
};

__device__ int* g_count;        // per-env count of executed steps (protocol check: every env steps exactly `steps` times)
__device__ __forceinline__ void step_env(const PhysT<float>& ph, float* st, const float* act, int* t, float* rew, uint8_t* done,
                                         int64_t ld, uint32_t i) {
    g_count[i] += 1;
    float x[4], inc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = st[j * ld + i];
    const float a = fminf(fmaxf(act[i], -10.f), 10.f);
    const int tt = t[i];
    Inputs<SCG_CARTPOLE, float> in;
    const float u[1] = {a};
    make_inputs<SCG_CARTPOLE, float>(ph, u, 0.f, 0.f, 0.f, in);
    integrate<SCG_CARTPOLE, float>(ph, 0u, in, x, inc);
    float l = 1e-4f * a * a;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] += inc[j]; l = fmaf(x[j], x[j], l); }
    const bool d = fabsf(x[0]) > 2.4f || fabsf(x[2]) > 1.57f || tt + 1 >= 250;
#pragma unroll
    for (int j = 0; j < 4; ++j) st[j * ld + i] = d ? 0.01f * (float)(i & 7) : x[j];
    rew[i] = -0.5f * l;
    done[i] = d;
    t[i] = d ? 0 : tt + 1;
}

__global__ __launch_bounds__(256) void k_step(PhysT<float> ph, float* st, const float* act, int* t, float* rew, uint8_t* done,
                                              int64_t ld, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) step_env(ph, st, act, t, rew, done, ld, i);
}

constexpr unsigned kSpinLimit = 1u << 22;

// one lane polls a word until it reaches `want`; false on timeout (and the timeout word is set)
template <int SCOPE>
__device__ __forceinline__ bool wait_ge(const unsigned* w, unsigned want, unsigned* tmo) {
    for (unsigned spins = 0; spins < kSpinLimit; ++spins) {
        if (__hip_atomic_load(w, __ATOMIC_RELAXED, SCOPE) >= want) return true;
        if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// HOSTBELL: doorbell in pinned host memory (system scope), completion written to pinned host memory by the last arriver
template <bool HOSTBELL>
__global__ __launch_bounds__(256) void k_server(PhysT<float> ph, float* st, const float* act, int64_t act_stride, int period, int* t,
                                                float* rew, uint8_t* done, int64_t ld, uint32_t n, int steps, Ctl* ctl,
                                                const unsigned* bell, unsigned* host_done) {
    __shared__ int s_ok;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const unsigned shard = blockIdx.x & 7u;
    for (int k = 1; k <= steps; ++k) {
        if (threadIdx.x == 0)
            s_ok = HOSTBELL ? wait_ge<__HIP_MEMORY_SCOPE_SYSTEM>(bell, (unsigned)k, &ctl->timeout)
                            : wait_ge<__HIP_MEMORY_SCOPE_AGENT>(bell, (unsigned)k, &ctl->timeout);
        __syncthreads();
        if (!s_ok) return;                                                  // block-uniform: a timeout ends every workgroup
        if (i < n) step_env(ph, st, act + (int64_t)((k - 1) % period) * act_stride, t, rew, done, ld, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every storing wave drains
        __syncthreads();
        if (threadIdx.x == 0) {
            if (HOSTBELL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // outputs visible before the arrival ("" = system scope)
            else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (HOSTBELL) {
                const unsigned prev = __hip_atomic_fetch_add(&ctl->last, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (prev + 1u == (unsigned)k * gridDim.x) __hip_atomic_store(host_done, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                __hip_atomic_fetch_add(&ctl->done[shard][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// the caller's stand-in on the device: ring, wait for all arrivals, repeat
__global__ __launch_bounds__(64) void k_ringer(Ctl* ctl, unsigned n_wg, int steps) {
    const int lane = threadIdx.x;
    for (int k = 1; k <= steps; ++k) {
        if (lane == 0) __hip_atomic_store(&ctl->doorbell, (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = false;
        for (unsigned spins = 0; spins < kSpinLimit; ++spins) {
            unsigned v = lane < 8 ? __hip_atomic_load(&ctl->done[lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) v += __shfl_xor(v, o, 64);
            if (__builtin_amdgcn_readfirstlane(v) >= (unsigned)k * n_wg) { ok = true; break; }
            if (__hip_atomic_load(&ctl->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { if (lane == 0) __hip_atomic_store(&ctl->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    }
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)std::atol(argv[1]) : 65536u;
    const int steps = argc > 2 ? std::atoi(argv[2]) : 2000;
    const int period = 16;
    const int64_t ld = n;
    const unsigned n_wg = (n + 255) / 256;
    PhysT<float> ph;
    std::memset(&ph, 0, sizeof(ph));
    ph.h = 0.02f; ph.cp_l = 0.5f; ph.cp_m = 0.1f; ph.cp_inv_Mm = 1.f / 1.1f; ph.cp_ml = 0.05f; ph.cp_g = 9.8f;
    float *st, *act, *rew; int* t; uint8_t* done; Ctl* ctl;
    CK(hipMalloc(&st, 4 * ld * sizeof(float))); CK(hipMalloc(&act, (size_t)period * ld * sizeof(float))); CK(hipMalloc(&rew, ld * sizeof(float)));
    CK(hipMalloc(&t, ld * sizeof(int))); CK(hipMalloc(&done, ld)); CK(hipMalloc(&ctl, sizeof(Ctl)));
    std::vector<float> h((size_t)period * ld);
    for (size_t k = 0; k < h.size(); ++k) h[k] = 20.f * ((float)((k * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f);
    CK(hipMemcpy(act, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    int* cnt; CK(hipMalloc(&cnt, ld * sizeof(int))); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_count), &cnt, sizeof(cnt)));
    auto reset = [&] { CK(hipMemset(st, 0, 4 * ld * sizeof(float))); CK(hipMemset(t, 0, ld * sizeof(int))); CK(hipMemset(cnt, 0, ld * sizeof(int))); CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipDeviceSynchronize()); };
    auto all_stepped = [&](int want) { std::vector<int> c(ld); CK(hipMemcpy(c.data(), cnt, ld * sizeof(int), hipMemcpyDeviceToHost)); for (int v : c) if (v != want) return false; return true; };
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned *h_bell, *h_done;
    CK(hipHostMalloc(&h_bell, 64, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostMalloc(&h_done, 64, hipHostMallocMapped | hipHostMallocCoherent));
    float ms = 0.f;
    std::printf("{\"probe\": \"step_server\", \"envs\": %u, \"workgroups\": %u, \"steps\": %d, \"bytes_per_env_step\": 49", n, n_wg, steps);

    // ---- A: one launch per step, queued
    reset();
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(k_step, dim3(n_wg), dim3(256), 0, s0, ph, st, act + (k % period) * ld, t, rew, done, ld, n);
    CK(hipStreamSynchronize(s0));
    reset();                                                               // the timed run starts from the same state as B and C
    CK(hipEventRecord(e0, s0));
    for (int k = 0; k < steps; ++k) hipLaunchKernelGGL(k_step, dim3(n_wg), dim3(256), 0, s0, ph, st, act + (k % period) * ld, t, rew, done, ld, n);
    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0)); CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf(", \"A_launch_per_step_us\": %.3f", ms * 1e3 / steps);
    std::printf(", \"A_every_env_stepped\": %s", all_stepped(steps) ? "true" : "false");

    // ---- A2: launch + host sync per step
    reset();
    { auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < steps; ++k) { hipLaunchKernelGGL(k_step, dim3(n_wg), dim3(256), 0, s0, ph, st, act + (k % period) * ld, t, rew, done, ld, n); CK(hipStreamSynchronize(s0)); }
      std::printf(", \"A2_launch_and_host_sync_per_step_us\": %.3f", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps); }

    // ---- B: resident server + device ringer (two streams; both grids are resident together: n_wg <= 8 * CUs)
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    if (n_wg + 1 > (unsigned)prop.multiProcessorCount * 4u) { std::printf(", \"error\": \"grid not co-resident\"}\n"); return 1; }
    for (int rep = 0; rep < 2; ++rep) {                                    // rep 0 = warm-up
        reset();
        CK(hipEventRecord(e0, s1));
        hipLaunchKernelGGL(k_server<false>, dim3(n_wg), dim3(256), 0, s0, ph, st, act, ld, period, t, rew, done, ld, n, steps, ctl, &ctl->doorbell, (unsigned*)nullptr);
        hipLaunchKernelGGL(k_ringer, dim3(1), dim3(64), 0, s1, ctl, n_wg, steps);
        CK(hipEventRecord(e1, s1));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s0)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned tmo = 0; CK(hipMemcpy(&tmo, &ctl->timeout, 4, hipMemcpyDeviceToHost));
    std::printf(", \"B_server_device_ringer_us\": %.3f, \"B_timeout\": %u, \"B_every_env_stepped\": %s", ms * 1e3 / steps, tmo, all_stepped(steps) ? "true" : "false");

    // ---- C: resident server, host rings (pinned host doorbell, completion word in pinned host memory)
    if (!tmo) {
        reset();
        *(volatile unsigned*)h_bell = 0; *(volatile unsigned*)h_done = 0;
        unsigned *d_bell, *d_done; CK(hipHostGetDevicePointer((void**)&d_bell, h_bell, 0)); CK(hipHostGetDevicePointer((void**)&d_done, h_done, 0));
        const int hsteps = steps < 500 ? steps : 500;
        hipLaunchKernelGGL(k_server<true>, dim3(n_wg), dim3(256), 0, s0, ph, st, act, ld, period, t, rew, done, ld, n, hsteps, ctl, d_bell, d_done);
        auto t0 = std::chrono::steady_clock::now();
        bool ok = true;
        for (int k = 1; k <= hsteps && ok; ++k) {
            __atomic_store_n(h_bell, (unsigned)k, __ATOMIC_RELEASE);
            const auto dl = std::chrono::steady_clock::now() + std::chrono::seconds(5);
            while (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) < (unsigned)k)
                if (std::chrono::steady_clock::now() > dl) { ok = false; break; }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / hsteps;
        if (!ok) { unsigned one = 1; CK(hipMemcpy(&ctl->timeout, &one, 4, hipMemcpyHostToDevice)); __atomic_store_n(h_bell, 0xffffffffu, __ATOMIC_RELEASE); }
        CK(hipStreamSynchronize(s0));
        std::printf(", \"C_server_host_ringer_us\": %.3f, \"C_ok\": %s, \"C_every_env_stepped\": %s", us, ok ? "true" : "false", all_stepped(hsteps) ? "true" : "false");
    }
    std::printf("}\n");
    return 0;
}
