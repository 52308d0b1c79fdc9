// Returns / GAE advantages over a device-resident rollout.
//
// Replaces compute_returns_and_advantages (controllers/ppo/ppo_utils.py:428-456) -- a Python loop over T with
// (N,1) NumPy temporaries on host copies of the rollout -- for rollouts that never leave the GPU.  Layout is
// the PPOBuffer scheme's (T, N, 1) arrays (ppo_utils.py:308-336) stored time-major: plane t holds the N envs of
// step t, so one lane per env walks t = T-1 .. 0 with fully coalesced 256-byte wave transactions.
// HBM-bound: 4 reads + 2 writes of 4 bytes per (t, env) = 24 B (28 B when the time-limit-compensated rewards are
// written back, as the reference's in-place `rews += gamma * terminal_vals` does, :443).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scg_device.h"

namespace scg {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#ifndef SCG_GAE_DEPTH
#define SCG_GAE_DEPTH 4        // time steps whose loads are in flight together
#endif
constexpr int kGaeDepth = SCG_GAE_DEPTH;

// rew and rew_out carry NO __restrict__: the reference updates the rewards in place (ppo_utils.py:443) and callers pass the
// same buffer for both; each element is loaded before the same lane stores it.
__global__ __launch_bounds__(256) void scg_gae_kernel(const float* rew, const float* __restrict__ val,
                                                      const float* __restrict__ mask, const float* __restrict__ term_val,
                                                      const float* __restrict__ last_val, float* __restrict__ ret_out,
                                                      float* __restrict__ adv_out, float* rew_out,
                                                      double* __restrict__ stats, int T, uint32_t n, int64_t ld,
                                                      float gamma, float lam, int use_gae) {
    double s1 = 0.0, s2 = 0.0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float ret = last_val[i];                     // ppo_utils.py:440: ret, adv = last_val, 0
        float v_next = ret;                          // vals[T] = last_val (:441)
        float adv = 0.f;
        // The scan over t is sequential per lane, so bandwidth comes from bytes in flight: the loads of kGaeDepth steps
        // are issued together (4 steps x 3-4 planes x 256 B per wave) before the dependent arithmetic of those steps.
        // Measured at T = 128 x 2^20: depth 1 / 2 / 4 / 8 = 0.568 / 0.563 / 0.515-0.564 / 0.532 ms -- within the box-to-box
        // noise of each other (~5 TB/s either way); 4 is kept.
        constexpr int D = kGaeDepth;
        int t = T - 1;
        for (; t >= D - 1; t -= D) {
            float r[D], v[D], m[D], tv[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int64_t o = (int64_t)(t - k) * ld + i;
                r[k] = __builtin_nontemporal_load(rew + o);
                v[k] = __builtin_nontemporal_load(val + o);
                m[k] = __builtin_nontemporal_load(mask + o);
                tv[k] = term_val ? __builtin_nontemporal_load(term_val + o) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const float rr = fmaf(gamma, tv[k], r[k]);                    // :443 rews += gamma * terminal_vals
                const float gm = gamma * m[k];
                ret = fmaf(gm, ret, rr);                                      // :446
                if (use_gae) {
                    const float td = fmaf(gm, v_next, rr) - v[k];             // :450
                    adv = fmaf(adv * lam, gm, td);                            // :451
                } else {
                    adv = ret - v[k];                                         // :448
                }
                const int64_t o = (int64_t)(t - k) * ld + i;
                __builtin_nontemporal_store(ret, ret_out + o);
                __builtin_nontemporal_store(adv, adv_out + o);
                if (rew_out) __builtin_nontemporal_store(rr, rew_out + o);
                s1 += (double)adv;
                s2 += (double)adv * (double)adv;
                v_next = v[k];
            }
        }
        for (; t >= 0; --t) {                                                 // the T mod D oldest steps
            const int64_t o = (int64_t)t * ld + i;
            const float r = rew[o], v = val[o], m = mask[o], tv = term_val ? term_val[o] : 0.f;
            const float rr = fmaf(gamma, tv, r);
            const float gm = gamma * m;
            ret = fmaf(gm, ret, rr);
            if (use_gae) {
                const float td = fmaf(gm, v_next, rr) - v;
                adv = fmaf(adv * lam, gm, td);
            } else {
                adv = ret - v;
            }
            ret_out[o] = ret;
            adv_out[o] = adv;
            if (rew_out) rew_out[o] = rr;
            s1 += (double)adv;
            s2 += (double)adv * (double)adv;
            v_next = v;
        }
    }
    if (stats) {
        // workgroup reduction first: same-address f64 atomics serialise, so one pair per workgroup, not per wave
        __shared__ double red[8];
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = s1; red[(threadIdx.x >> 6) * 2 + 1] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&stats[0], (red[0] + red[2]) + (red[4] + red[6]));
            atomicAdd(&stats[1], (red[1] + red[3]) + (red[5] + red[7]));
        }
    }
}

// adv <- (adv - mean) / (std + eps) with the population statistics accumulated by scg_gae_kernel (ppo.py:288).
// One grid row per time step (plane t = adv + t ld): no per-element index arithmetic; 16-byte accesses when the planes allow.
template <bool VEC>
__global__ __launch_bounds__(256) void scg_normalize_kernel(float* __restrict__ adv, const double* __restrict__ stats,
                                                            int T, uint32_t n, int64_t ld, float eps) {
    const double count = (double)T * (double)n;
    const double mean = stats[0] / count;
    double var = stats[1] / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float mu = (float)mean, inv = (float)(1.0 / (sqrt(var) + (double)eps));
    float* plane = adv + (int64_t)blockIdx.y * ld;
    if constexpr (VEC) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        const uint32_t n4 = n >> 2, stride = gridDim.x * 256u;
        for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < n4; q += 2u * stride) {
            v4 a = reinterpret_cast<v4*>(plane)[q], b = a;
            const bool two = q + stride < n4;
            if (two) b = reinterpret_cast<v4*>(plane)[q + stride];
            __builtin_nontemporal_store((a - mu) * inv, reinterpret_cast<v4*>(plane) + q);
            if (two) __builtin_nontemporal_store((b - mu) * inv, reinterpret_cast<v4*>(plane) + q + stride);
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
            const uint32_t i = (n & ~3u) + threadIdx.x;
            plane[i] = (plane[i] - mu) * inv;
        }
    } else {
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) plane[i] = (plane[i] - mu) * inv;
    }
}

hipError_t launch_gae(const float* rew, const float* val, const float* mask, const float* term_val,
                      const float* last_val, float* ret, float* adv, float* rew_out, double* stats, int T, int64_t n,
                      int64_t ld, float gamma, float lam, int use_gae, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(scg_gae_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rew, val, mask, term_val, last_val, ret,
                       adv, rew_out, stats, T, (uint32_t)n, ld, gamma, lam, use_gae);
    return hipGetLastError();
}

hipError_t launch_normalize(float* adv, const double* stats, int T, int64_t n, int64_t ld, float eps, hipStream_t s) {
    if (T <= 0 || n <= 0) return hipSuccess;
    const bool vec = ((reinterpret_cast<uintptr_t>(adv) | (uintptr_t)(ld * 4)) & 15u) == 0;
    int64_t per_row = ((vec ? (n + 3) / 4 : n) + 255) / 256;                 // workgroups that cover one plane
    const int64_t cap = T >= 64 ? 64 : 4096 / T;                              // ~4096 workgroups in all
    if (per_row > cap) per_row = cap;
    if (per_row < 1) per_row = 1;
    for (int t0 = 0; t0 < T; t0 += 65535) {                                   // (grid.y holds at most 65535 rows)
        const int rows = T - t0 < 65535 ? T - t0 : 65535;
        const dim3 grid((unsigned)per_row, (unsigned)rows), block(256);
        float* base = adv + (int64_t)t0 * ld;
        if (vec) hipLaunchKernelGGL(scg_normalize_kernel<true>, grid, block, 0, s, base, stats, T, (uint32_t)n, ld, eps);
        else hipLaunchKernelGGL(scg_normalize_kernel<false>, grid, block, 0, s, base, stats, T, (uint32_t)n, ld, eps);
    }
    return hipGetLastError();
}

}  // namespace scg
