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

#ifndef SCG_GAE_W
#define SCG_GAE_W 2            // envs per lane when the planes allow (n, ld multiples of W, W * 4-byte aligned bases).  Measured at
#endif                         // T = 128 x 2^20, interleaved on one box (profiles/r03/gae_envs_per_lane_ab.txt): W = 1 / 2 / 4 at depth 4 =
                               // 0.524-0.540 / 0.502-0.509 / 0.542-0.543 ms (W = 4: 148 VGPRs, 3 waves per SIMD; at depth 2 / 1: 0.527-0.531)
#ifndef SCG_GAE_NT
#define SCG_GAE_NT 1           // nontemporal hint on the streaming loads
#endif

template <int W> struct GaeVec;
template <> struct GaeVec<1> { typedef float type; };
template <> struct GaeVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct GaeVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int W>
__device__ __forceinline__ void gae_load(const float* p, float (&out)[W]) {
    typedef typename GaeVec<W>::type VT;
#if SCG_GAE_NT
    const VT v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(p));
#else
    const VT v = *reinterpret_cast<const VT*>(p);
#endif
    if constexpr (W == 1) out[0] = v;
    else {
#pragma unroll
        for (int c = 0; c < W; ++c) out[c] = v[c];
    }
}
template <int W>
__device__ __forceinline__ void gae_store(const float (&in)[W], float* p) {
    typedef typename GaeVec<W>::type VT;
    VT v;
    if constexpr (W == 1) v = in[0];
    else {
#pragma unroll
        for (int c = 0; c < W; ++c) v[c] = in[c];
    }
    __builtin_nontemporal_store(v, reinterpret_cast<VT*>(p));
}

// mask = 1 - done from the step kernel's byte planes (W consecutive bytes per lane), or the float mask planes
template <int W>
__device__ __forceinline__ void gae_load_mask(const float* mask, const uint8_t* done, int64_t o, float (&m)[W]) {
    if (done) {
        if constexpr (W == 1) m[0] = 1.0f - (float)__builtin_nontemporal_load(done + o);
        else if constexpr (W == 2) {
            const uint16_t d = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(done + o));
            m[0] = 1.0f - (float)(d & 0xffu); m[1] = 1.0f - (float)(d >> 8);
        } else {
            const uint32_t d = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(done + o));
#pragma unroll
            for (int c = 0; c < W; ++c) m[c] = 1.0f - (float)((d >> (8 * c)) & 0xffu);
        }
    } else {
        gae_load<W>(mask + o, m);
    }
}

// rew and rew_out carry NO __restrict__: the reference updates the rewards in place (ppo_utils.py:443) and callers pass the
// same buffer for both; each element is loaded before the same lane stores it.
// W envs per lane (adjacent columns of every plane): W = 2 makes every access an 8-byte one -- half the memory instructions
// for the same bytes, twice the bytes in flight per lane of the sequential scan.
template <int W>
__global__ __launch_bounds__(256) void scg_gae_kernel(const float* rew, const float* __restrict__ val,
                                                      const float* __restrict__ mask, const uint8_t* __restrict__ done,
                                                      const float* __restrict__ term_val,
                                                      const float* __restrict__ last_val, float* __restrict__ ret_out,
                                                      float* __restrict__ adv_out, float* rew_out,
                                                      double* __restrict__ stats, int T, uint32_t n, int64_t ld,
                                                      float gamma, float lam, int use_gae) {
    double s1 = 0.0, s2 = 0.0;
    const uint32_t groups = n / W;                   // the launcher picks W = 1 unless n is a multiple of W
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < groups; q += gridDim.x * 256) {
        const uint32_t i = q * W;
        float ret[W], v_next[W], adv[W];
        gae_load<W>(last_val + i, ret);              // ppo_utils.py:440: ret, adv = last_val, 0
#pragma unroll
        for (int c = 0; c < W; ++c) { v_next[c] = ret[c]; adv[c] = 0.f; }   // vals[T] = last_val (:441)
        // The scan over t is sequential per lane, so bandwidth comes from bytes in flight: the loads of kGaeDepth steps
        // are issued together (4 steps x 3-4 planes) before the dependent arithmetic of those steps.
        constexpr int D = kGaeDepth;
        int t = T - 1;
        for (; t >= D - 1; t -= D) {
            float r[D][W], v[D][W], m[D][W], tv[D][W];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int64_t o = (int64_t)(t - k) * ld + i;
                gae_load<W>(rew + o, r[k]);
                gae_load<W>(val + o, v[k]);
                gae_load_mask<W>(mask, done, o, m[k]);
                if (term_val) gae_load<W>(term_val + o, tv[k]);
                else {
#pragma unroll
                    for (int c = 0; c < W; ++c) tv[k][c] = 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
                float rr[W];
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    rr[c] = fmaf(gamma, tv[k][c], r[k][c]);                   // :443 rews += gamma * terminal_vals
                    const float gm = gamma * m[k][c];
                    ret[c] = fmaf(gm, ret[c], rr[c]);                         // :446
                    if (use_gae) {
                        const float td = fmaf(gm, v_next[c], rr[c]) - v[k][c];   // :450
                        adv[c] = fmaf(adv[c] * lam, gm, td);                  // :451
                    } else {
                        adv[c] = ret[c] - v[k][c];                            // :448
                    }
                    s1 += (double)adv[c];
                    s2 += (double)adv[c] * (double)adv[c];
                    v_next[c] = v[k][c];
                }
                const int64_t o = (int64_t)(t - k) * ld + i;
                gae_store<W>(ret, ret_out + o);
                gae_store<W>(adv, adv_out + o);
                if (rew_out) gae_store<W>(rr, rew_out + o);
            }
        }
        for (; t >= 0; --t) {                                                 // the T mod D oldest steps
            const int64_t o = (int64_t)t * ld + i;
            float r[W], v[W], m[W], tv[W], rr[W];
            gae_load<W>(rew + o, r);
            gae_load<W>(val + o, v);
            gae_load_mask<W>(mask, done, o, m);
            if (term_val) gae_load<W>(term_val + o, tv);
#pragma unroll
            for (int c = 0; c < W; ++c) {
                rr[c] = fmaf(gamma, term_val ? tv[c] : 0.f, r[c]);
                const float gm = gamma * m[c];
                ret[c] = fmaf(gm, ret[c], rr[c]);
                if (use_gae) {
                    const float td = fmaf(gm, v_next[c], rr[c]) - v[c];
                    adv[c] = fmaf(adv[c] * lam, gm, td);
                } else {
                    adv[c] = ret[c] - v[c];
                }
                s1 += (double)adv[c];
                s2 += (double)adv[c] * (double)adv[c];
                v_next[c] = v[c];
            }
            gae_store<W>(ret, ret_out + o);
            gae_store<W>(adv, adv_out + o);
            if (rew_out) gae_store<W>(rr, rew_out + o);
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

hipError_t launch_gae(const float* rew, const float* val, const float* mask, const uint8_t* done, const float* term_val,
                      const float* last_val, float* ret, float* adv, float* rew_out, double* stats, int T, int64_t n,
                      int64_t ld, float gamma, float lam, int use_gae, hipStream_t s) {
    // W envs per lane when every plane row starts on a W * 4-byte boundary and n is a multiple of W; one otherwise
    constexpr int W = SCG_GAE_W;
    uintptr_t bits = (uintptr_t)(ld * 4) | (uintptr_t)(n * 4);
    for (const void* ptr : {(const void*)rew, (const void*)val, (const void*)mask, (const void*)term_val, (const void*)last_val,
                            (const void*)ret, (const void*)adv, (const void*)rew_out})
        bits |= reinterpret_cast<uintptr_t>(ptr);
    const bool wide = W > 1 && (bits & (uintptr_t)(W * 4 - 1)) == 0 && (reinterpret_cast<uintptr_t>(done) & (uintptr_t)(W - 1)) == 0;
    int64_t blocks = ((wide ? n / W : n) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    if (wide) hipLaunchKernelGGL(scg_gae_kernel<W>, dim3((unsigned)blocks), dim3(256), 0, s, rew, val, mask, done, term_val, last_val, ret,
                                 adv, rew_out, stats, T, (uint32_t)n, ld, gamma, lam, use_gae);
    else hipLaunchKernelGGL(scg_gae_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, rew, val, mask, done, term_val, last_val, ret,
                            adv, rew_out, stats, T, (uint32_t)n, ld, gamma, lam, use_gae);
    return hipGetLastError();
}

hipError_t launch_normalize(float* adv, const double* stats, int T, int64_t n, int64_t ld, float eps, hipStream_t s) {
    if (T <= 0 || n <= 0) return hipSuccess;
    const bool vec = ((reinterpret_cast<uintptr_t>(adv) | (uintptr_t)(ld * 4)) & 15u) == 0;
    // One 16-byte unit per thread and NO loop: short-lived waves over a large grid that is dispatched in address order.  A streaming
    // pass is faster the closer the order in which its accesses ARRIVE follows the addresses (tools/ubench/write_pattern.hip: one
    // store per wave 6.9 TB/s, 3-48 stores per wave 5.4, a 98-iteration grid-stride loop 4.3).  Measured here at 128 x 2^20, interleaved
    // on one box (profiles/r03/advnorm_grid_ab.txt): 64 / 128 / 256 / 512 workgroups per row = 240 / 211 / 185 / 178 us; 4 / 2 / 1 units
    // per thread without a cap = 185 / 178 / 176 us.
#ifndef SCG_ADVNORM_UNITS
#define SCG_ADVNORM_UNITS 1
#endif
    int64_t per_row = ((vec ? (n + 3) / 4 : n) + 256 * SCG_ADVNORM_UNITS - 1) / (256 * SCG_ADVNORM_UNITS);   // workgroups that cover one plane
    if (per_row > 65535) per_row = 65535;
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

// ---- PPOBuffer.push in one launch (scg_copy_segments): up to 8 strided rows x cols float32 copies ------------------------------
// The segments travel by value (kernel arguments): no table in device memory, nothing to keep alive.  grid.y = segment; a segment
// that is dense on both sides (the usual pushed key: a contiguous policy output into a contiguous buffer row) moves as 16-byte
// units, anything else -- the [N, d] view of SoA planes that env.step() returns, row-major actions into action planes -- element by
// element, consecutive threads along the SOURCE's fastest dimension so that the reads coalesce.
struct CopySegments {
    scg_copy_segment seg[SCG_MAX_COPY_SEGMENTS];
};

__global__ __launch_bounds__(256) void scg_copy_segments_kernel(const CopySegments cs) {
    const scg_copy_segment& g = cs.seg[blockIdx.y];
    const int64_t total = g.rows * g.cols;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    const bool dense = g.src_col_stride == 1 && g.dst_col_stride == 1 && g.src_row_stride == g.cols && g.dst_row_stride == g.cols;
    if (dense && ((reinterpret_cast<uintptr_t>(g.src) | reinterpret_cast<uintptr_t>(g.dst)) & 15u) == 0) {
        typedef float v4 __attribute__((ext_vector_type(4)));
        const int64_t units = total >> 2;
        for (int64_t q = tid; q < units; q += step) reinterpret_cast<v4*>(g.dst)[q] = reinterpret_cast<const v4*>(g.src)[q];
        for (int64_t e = (units << 2) + tid; e < total; e += step) g.dst[e] = g.src[e];
        return;
    }
    if (g.src_col_stride <= g.src_row_stride || g.rows == 1) {
        // columns are the source's fast dimension (row-major rows)
        for (int64_t e = tid; e < total; e += step) {
            const int64_t i = e / g.cols, k = e - i * g.cols;
            g.dst[i * g.dst_row_stride + k * g.dst_col_stride] = g.src[i * g.src_row_stride + k * g.src_col_stride];
        }
    } else {
        // rows are the source's fast dimension (an [N, d] view of planes): walk plane by plane
        for (int64_t e = tid; e < total; e += step) {
            const int64_t k = e / g.rows, i = e - k * g.rows;
            g.dst[i * g.dst_row_stride + k * g.dst_col_stride] = g.src[i * g.src_row_stride + k * g.src_col_stride];
        }
    }
}

hipError_t launch_copy_segments(const scg_copy_segment* seg, int count, hipStream_t s) {
    CopySegments cs{};
    int64_t most = 0;
    for (int j = 0; j < count; ++j) {
        cs.seg[j] = seg[j];
        const int64_t total = seg[j].rows * seg[j].cols;
        most = total > most ? total : most;
    }
    if (most == 0) return hipSuccess;
    int64_t blocks = (most + 1023) / 1024;                   // ~4 elements per thread, grid-stride beyond 2048 workgroups per segment
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scg_copy_segments_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, s, cs);
    return hipGetLastError();
}

}  // namespace scg
