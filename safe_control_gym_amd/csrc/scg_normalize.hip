// Running mean / std normalisers on the device.
//
// Replaces math_and_models/normalization.py: RunningMeanStd (:17-62), MeanStdNormalizer (:107-147) and
// RewardStdNormalizer (:150-190), which PPO calls on every rollout step on host copies of the observations
// and rewards (controllers/ppo/ppo.py:260-261).  Here the batch stays where the step kernel left it.
//
// One call = two launches (batches above 2^14 envs; smaller ones take the one-launch form further down):
//   1. moments: per feature j, sum (x - c_j) and sum (x - c_j)^2 over the batch in double, c_j = the current
//      running mean (a shift that removes the cancellation of the one-pass variance).  Each of the <= 64
//      workgroups of a feature reduces through wave shuffles + LDS and writes ONE partial pair -- no atomics:
//      f64 atomics on one address serialise at ~40 ns each on this part (measured: 2048 of them = 86 us), and the
//      fixed summation order makes the tracker bit-reproducible.  Planes are SoA ([d][ld]), so every load is a
//      coalesced wave transaction (float4 per lane when the planes are 16-byte aligned).
//   2. normalise: every wave sums the feature's partials (one per lane), recomputes the tiny update_from_moments
//      (:41-62) from the CURRENT state (the state is double-buffered: workgroup (0, j) writes the NEXT state,
//      nobody reads it in this launch), then streams y = clip((x - mean) / sqrt(var + eps)).
// The second pass re-reads x out of the 256 MB Infinity Cache for the batch sizes of interest (48 MB at 2^20 x 12).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scg_device.h"
#include "scg_math.h"

#ifndef SCG_NORMALIZE_SMALL_BATCH
#define SCG_NORMALIZE_SMALL_BATCH 16384   // envs up to which a normaliser call is ONE launch (one workgroup per feature, the plane in registers); 0: never
#endif

namespace scg {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// state record (doubles): mean[d], var[d], count
__device__ __forceinline__ const double* st_mean(const double* s) { return s; }
__device__ __forceinline__ const double* st_var(const double* s, int d) { return s + d; }

// ---- pass 1 ---------------------------------------------------------------------------------------
// workgroups (= partial sums) per feature: enough of them to occupy the chip when there are few features
// (the reward tracker has one), few enough that every wave of pass 2 can re-sum them for pennies
__host__ __device__ constexpr int max_partials(int d) { return d >= 16 ? 64 : 1024 / d; }
typedef float nf4 __attribute__((ext_vector_type(4)));

// block-wide sum of up to 3 doubles; result valid in thread 0
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* lds) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wsum(v[k]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = lds[k] + lds[K + k] + lds[2 * K + k] + lds[3 * K + k];
    }
}

// partials: [d][max_partials(d)][2]
template <bool VEC>
__global__ __launch_bounds__(256) void scg_moments_kernel(const float* __restrict__ x, int64_t ld, uint32_t n,
                                                          const double* __restrict__ state, int d,
                                                          double* __restrict__ partials) {
    __shared__ double lds[8];
    const int j = blockIdx.y;
    const float* plane = uptr(x + (int64_t)j * ld);
    const double c = state[j];
    double acc[2] = {0.0, 0.0};
    if constexpr (VEC) {
        // four independent 16-byte loads in flight per thread before the first use (round 3): with one, the dozen iterations of a
        // thread were a dozen exposed memory latencies -- this pass ran at 2.3 TB/s while the apply pass behind it streams at 10
        const uint32_t n4 = n >> 2, stride = gridDim.x * 256u;
        for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < n4; q += 4u * stride) {
            nf4 v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u * stride < n4) v4[u] = reinterpret_cast<const nf4*>(plane)[q + u * stride];     // plain load: the apply pass re-reads x from the cache
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q + u * stride < n4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const double v = (double)v4[u][k] - c;
                        acc[0] += v;
                        acc[1] = fma(v, v, acc[1]);
                    }
                }
            }
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {                // ragged tail: at most 3 elements
            const double v = (double)plane[(n & ~3u) + threadIdx.x] - c;
            acc[0] += v;
            acc[1] = fma(v, v, acc[1]);
        }
    } else {
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
            const double v = (double)ldp(plane, i * 4u) - c;
            acc[0] += v;
            acc[1] = fma(v, v, acc[1]);
        }
    }
    block_sum<2>(acc, lds);
    if (threadIdx.x == 0) {
        double* out = partials + ((int64_t)j * max_partials(d) + blockIdx.x) * 2;
        out[0] = acc[0];
        out[1] = acc[1];
    }
}

// RewardStdNormalizer.__call__ (:172-189), first half: ret = ret * gamma + rew; moments of ret; then the
// "information leak" reset.  legacy == 0: ret[i] = 0 where done[i] (what the comment at :187 intends).
// legacy != 0 reproduces the reference statement literally -- `self.ret[dones.astype(np.long)] = 0` indexes with
// the 0 / 1 VALUES of dones, i.e. clears ret[0] when some env is not done and ret[1] when some env is done --
// which the normalise pass applies once the done count is known.  partials: [<= kRewardPartials][3].
// 13 B per env in one pass: <= 256 workgroups, every thread with four independent 16-byte groups (ret, rew, 4 done bytes)
// in flight before the first use (round 4: the one-element-per-iteration form spent its time on exposed latencies).
constexpr int kRewardPartials = 256;

template <bool VEC>
__global__ __launch_bounds__(256) void scg_reward_moments_kernel(const float* __restrict__ rew,
                                                                 const uint8_t* __restrict__ done,
                                                                 float* __restrict__ ret, uint32_t n, float gamma,
                                                                 const double* __restrict__ state, int legacy,
                                                                 double* __restrict__ partials) {
    __shared__ double lds[12];
    const double cd = state[0];
    double acc[3] = {0.0, 0.0, 0.0};
    if constexpr (VEC) {
        const uint32_t n4 = n >> 2, stride = gridDim.x * 256u;
        for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < n4; q += 4u * stride) {
            nf4 a4[4], r4[4];
            uint32_t d4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u * stride < n4) {
                    a4[u] = reinterpret_cast<const nf4*>(ret)[q + u * stride];
                    r4[u] = reinterpret_cast<const nf4*>(rew)[q + u * stride];
                    d4[u] = reinterpret_cast<const uint32_t*>(done)[q + u * stride];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u * stride < n4) {
                    nf4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float r = fmaf(a4[u][k], gamma, r4[u][k]);
                        const bool dn = ((d4[u] >> (8 * k)) & 0xffu) != 0;
                        const double v = (double)r - cd;
                        acc[0] += v;
                        acc[1] = fma(v, v, acc[1]);
                        acc[2] += dn ? 1.0 : 0.0;
                        o[k] = (!legacy && dn) ? 0.f : r;
                    }
                    reinterpret_cast<nf4*>(ret)[q + u * stride] = o;
                }
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {                // ragged tail: at most 3 elements
            const uint32_t i = (n & ~3u) + threadIdx.x;
            const float r = fmaf(ret[i], gamma, rew[i]);
            const bool dn = done[i] != 0;
            const double v = (double)r - cd;
            acc[0] += v;
            acc[1] = fma(v, v, acc[1]);
            acc[2] += dn ? 1.0 : 0.0;
            ret[i] = (!legacy && dn) ? 0.f : r;
        }
    } else {
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
            const float r = fmaf(ret[i], gamma, rew[i]);
            const double v = (double)r - cd;
            acc[0] += v;
            acc[1] = fma(v, v, acc[1]);
            const bool dn = done[i] != 0;
            acc[2] += dn ? 1.0 : 0.0;
            ret[i] = (!legacy && dn) ? 0.f : r;
        }
    }
    block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 3 + 0] = acc[0];
        partials[blockIdx.x * 3 + 1] = acc[1];
        partials[blockIdx.x * 3 + 2] = acc[2];
    }
}

// every lane of the calling wave gets the sum over the feature's `count` partials (fixed order: bit-reproducible)
template <int K>
__device__ __forceinline__ void sum_partials(const double* __restrict__ p, int count, double (&tot)[K]) {
    const int lane = threadIdx.x & 63;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (int q = lane; q < count; q += 64) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] += p[q * K + k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) tot[k] = wsum(acc[k]);
}

// ---- RunningMeanStd.update_from_moments (:41-62) for one feature --------------------------------------
struct Updated { double mean, var, count; };
__device__ __forceinline__ Updated update_from_moments(double mean, double var, double count, double c, double s1,
                                                       double s2, double n) {
    const double m1 = s1 / n;
    const double batch_mean = c + m1;
    const double batch_var = fmax(s2 / n - m1 * m1, 0.0);      // np.var (population variance, :36)
    const double delta = batch_mean - mean;
    const double tot = count + n;
    Updated u;
    u.mean = mean + delta * n / tot;
    const double m2 = var * count + batch_var * n + delta * delta * count * n / tot;
    u.var = m2 / tot;
    u.count = tot;
    return u;
}

// ---- pass 2 ---------------------------------------------------------------------------------------
// state_out may be null (read-only normaliser: no update, partials unused).
template <bool VEC>
__global__ __launch_bounds__(256) void scg_mean_std_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int64_t ld_x, int64_t ld_y, uint32_t n, int d,
                                                                 const double* __restrict__ state_in,
                                                                 double* __restrict__ state_out,
                                                                 const double* __restrict__ partials, int n_partials,
                                                                 int partials_stride, double batch_n, float clip, double epsilon) {
    // batch_n: rows behind the sums -- n, or the rows of ALL shards when the sums were added up across devices (partials_stride 1)
    const int j = blockIdx.y;
    double mean = state_in[j], var = state_in[d + j];
    if (state_out) {
        double tot[2];
        sum_partials<2>(partials + (int64_t)j * partials_stride * 2, n_partials, tot);
        const Updated u = update_from_moments(mean, var, state_in[2 * d], mean, tot[0], tot[1], batch_n);
        mean = u.mean; var = u.var;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            state_out[j] = u.mean;
            state_out[d + j] = u.var;
            if (j == 0) state_out[2 * d] = u.count;
        }
    }
    if (!y) return;                                             // RunningMeanStd.update(): the tracker moves, nothing is normalised
    const double inv = 1.0 / sqrt(var + epsilon);               // :131-133
    const float* src = uptr(x + (int64_t)j * ld_x);
    float* dst = uptr(y + (int64_t)j * ld_y);
    const double lo = -(double)clip, hi = (double)clip;
    if constexpr (VEC) {
        const uint32_t n4 = n >> 2, stride = gridDim.x * 256u;
        // four independent 16-byte loads in flight per thread before the first use: at one per thread the 12 resident workgroups
        // of a CU hold 49 KB in flight, less than latency x bandwidth asks for (62 KB per CU)
        for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < n4; q += 4u * stride) {
            nf4 v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u * stride < n4) v4[u] = reinterpret_cast<const nf4*>(src)[q + u * stride];   // plain load: this pass may hit the cache
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q + u * stride < n4) {
                    nf4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (float)fmin(fmax(((double)v4[u][k] - mean) * inv, lo), hi);
                    __builtin_nontemporal_store(o, reinterpret_cast<nf4*>(dst) + q + u * stride);
                }
            }
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
            const uint32_t i = (n & ~3u) + threadIdx.x;
            dst[i] = (float)fmin(fmax(((double)src[i] - mean) * inv, lo), hi);
        }
    } else {
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
            const double v = ((double)src[i] - mean) * inv;
            stp(dst, i * 4u, (float)fmin(fmax(v, lo), hi));
        }
    }
}

// RewardStdNormalizer, second half (:189): out = clip(rew / sqrt(var + eps)) -- scaling only.
template <bool VEC>
__global__ __launch_bounds__(256) void scg_reward_apply_kernel(const float* __restrict__ rew, float* __restrict__ out,
                                                               float* __restrict__ ret, uint32_t n,
                                                               const double* __restrict__ state_in,
                                                               double* __restrict__ state_out,
                                                               const double* __restrict__ partials, int n_partials,
                                                               double batch_n, int legacy, float clip, double epsilon) {
    // this thread's rewards first: their latency covers the partial sums below
    const uint32_t n4 = n >> 2, stride = gridDim.x * 256u, q0 = blockIdx.x * 256u + threadIdx.x;
    nf4 r4[4];
    if constexpr (VEC) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (q0 + u * stride < n4) r4[u] = reinterpret_cast<const nf4*>(rew)[q0 + u * stride];
    }
    double var = state_in[1];
    if (state_out) {
        double tot[3];
        sum_partials<3>(partials, n_partials, tot);
        const Updated u = update_from_moments(state_in[0], var, state_in[2], state_in[0], tot[0], tot[1], batch_n);
        var = u.var;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            state_out[0] = u.mean; state_out[1] = u.var; state_out[2] = u.count;
            if (legacy) {
                const double nd = tot[2];
                if (nd < (double)n) ret[0] = 0.f;               // some dones == 0 -> index 0
                if (nd > 0.0 && n > 1) ret[1] = 0.f;            // some dones == 1 -> index 1
            }
        }
    }
    const double inv = 1.0 / sqrt(var + epsilon);
    const double lo = -(double)clip, hi = (double)clip;
    if constexpr (VEC) {
        for (uint32_t q = q0; q < n4; q += 4u * stride) {
            if (q != q0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (q + u * stride < n4) r4[u] = reinterpret_cast<const nf4*>(rew)[q + u * stride];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u * stride < n4) {
                    nf4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (float)fmin(fmax((double)r4[u][k] * inv, lo), hi);
                    reinterpret_cast<nf4*>(out)[q + u * stride] = o;
                }
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
            const uint32_t i = (n & ~3u) + threadIdx.x;
            out[i] = (float)fmin(fmax((double)rew[i] * inv, lo), hi);
        }
    } else {
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
            out[i] = (float)fmin(fmax((double)rew[i] * inv, lo), hi);
    }
}

// ---- statistics shared by several shards (one tracker, the batch spread over devices) -------------------------------------
// sums[j][k] = the feature's partials added up: what every shard contributes to the all-reduce between the two passes.
template <int K>
__global__ __launch_bounds__(64) void scg_sum_partials_kernel(const double* __restrict__ partials, int n_partials, int partials_stride,
                                                              double* __restrict__ sums) {
    double tot[K];
    sum_partials<K>(partials + (int64_t)blockIdx.x * partials_stride * K, n_partials, tot);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sums[blockIdx.x * K + k] = tot[k];
    }
}

// ---- small batches: ONE launch -------------------------------------------------------------------------
// A feature's normalisation depends on that feature's moments only, so ONE workgroup per feature can do the whole call: read its
// plane, block reduction, update_from_moments, normalise -- no second launch and no partials.  Up to kSmallBatch envs the second
// launch (~3 us on the host, ~3 us of dependent-launch gap on the device) costs more than one workgroup's pass over its plane;
// above, one workgroup per feature is too little parallelism and the two-launch form takes over.
constexpr int kSmallBlock = 1024;
constexpr int64_t kSmallBatch = SCG_NORMALIZE_SMALL_BATCH < 16 * kSmallBlock ? SCG_NORMALIZE_SMALL_BATCH : 16 * kSmallBlock;

template <int K>
__device__ __forceinline__ void block_sum_all(double (&v)[K], double* lds) {          // every thread gets the sums; kSmallBlock threads
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wsum(v[k]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kSmallBlock / 64; ++w) t += lds[w * K + k];               // fixed order: bit-reproducible
        v[k] = t;
    }
}

// Up to 4 x 16 bytes per thread and plane: kSmallBatch = 4 * 4 * kSmallBlock envs, read ONCE -- the values wait in registers for the
// statistics (16-byte aligned planes; anything else takes the two-launch form).
__global__ __launch_bounds__(kSmallBlock) void scg_mean_std_small_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                          int64_t ld_x, int64_t ld_y, uint32_t n, int d,
                                                                          const double* __restrict__ state_in,
                                                                          double* __restrict__ state_out, float clip, double epsilon) {
    __shared__ double lds[2 * (kSmallBlock / 64)];
    const int j = blockIdx.x;
    const float* src = uptr(x + (int64_t)j * ld_x);
    const uint32_t n4 = n >> 2, tail = (n & ~3u) + threadIdx.x;
    const bool has_tail = threadIdx.x < (n & 3u);
    nf4 v4[4];
    float vt = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (threadIdx.x + u * kSmallBlock < n4) v4[u] = reinterpret_cast<const nf4*>(src)[threadIdx.x + u * kSmallBlock];
    if (has_tail) vt = src[tail];
    double mean = state_in[j], var = state_in[d + j];
    if (state_out) {
        const double c = mean;
        double acc[2] = {0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (threadIdx.x + u * kSmallBlock < n4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double v = (double)v4[u][k] - c;
                    acc[0] += v;
                    acc[1] = fma(v, v, acc[1]);
                }
            }
        if (has_tail) {
            const double v = (double)vt - c;
            acc[0] += v;
            acc[1] = fma(v, v, acc[1]);
        }
        block_sum_all<2>(acc, lds);
        const Updated u = update_from_moments(mean, var, state_in[2 * d], c, acc[0], acc[1], (double)n);
        mean = u.mean; var = u.var;
        if (threadIdx.x == 0) {
            state_out[j] = u.mean;
            state_out[d + j] = u.var;
            if (j == 0) state_out[2 * d] = u.count;
        }
    }
    const double inv = 1.0 / sqrt(var + epsilon);
    const double lo = -(double)clip, hi = (double)clip;
    float* dst = uptr(y + (int64_t)j * ld_y);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (threadIdx.x + u * kSmallBlock < n4) {
            nf4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (float)fmin(fmax(((double)v4[u][k] - mean) * inv, lo), hi);
            reinterpret_cast<nf4*>(dst)[threadIdx.x + u * kSmallBlock] = o;
        }
    if (has_tail) dst[tail] = (float)fmin(fmax(((double)vt - mean) * inv, lo), hi);
}

__global__ __launch_bounds__(kSmallBlock) void scg_reward_small_kernel(const float* __restrict__ rew, const uint8_t* __restrict__ done,
                                                                        float* __restrict__ ret, float* __restrict__ out, uint32_t n,
                                                                        float gamma, const double* __restrict__ state_in,
                                                                        double* __restrict__ state_out, int legacy, float clip,
                                                                        double epsilon) {
    __shared__ double lds[3 * (kSmallBlock / 64)];
    const uint32_t n4 = n >> 2, tail = (n & ~3u) + threadIdx.x;
    const bool has_tail = threadIdx.x < (n & 3u);
    nf4 r4[4], a4[4];
    uint32_t d4[4];
    float rt = 0.f, at = 0.f;
    bool dt = false;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (threadIdx.x + u * kSmallBlock < n4) {
            const uint32_t q = threadIdx.x + u * kSmallBlock;
            r4[u] = reinterpret_cast<const nf4*>(rew)[q];
            a4[u] = reinterpret_cast<const nf4*>(ret)[q];
            d4[u] = reinterpret_cast<const uint32_t*>(done)[q];
        }
    if (has_tail) { rt = rew[tail]; at = ret[tail]; dt = done[tail] != 0; }
    const double c = state_in[0];
    double var = state_in[1];
    double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (threadIdx.x + u * kSmallBlock < n4) {
            nf4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float r = fmaf(a4[u][k], gamma, r4[u][k]);
                const bool dn = ((d4[u] >> (8 * k)) & 0xffu) != 0;
                const double v = (double)r - c;
                acc[0] += v;
                acc[1] = fma(v, v, acc[1]);
                acc[2] += dn ? 1.0 : 0.0;
                o[k] = (!legacy && dn) ? 0.f : r;
            }
            reinterpret_cast<nf4*>(ret)[threadIdx.x + u * kSmallBlock] = o;
        }
    if (has_tail) {
        const float r = fmaf(at, gamma, rt);
        const double v = (double)r - c;
        acc[0] += v;
        acc[1] = fma(v, v, acc[1]);
        acc[2] += dt ? 1.0 : 0.0;
        ret[tail] = (!legacy && dt) ? 0.f : r;
    }
    block_sum_all<3>(acc, lds);                                    // (its barrier orders every ret store above before the two below)
    const Updated u = update_from_moments(c, var, state_in[2], c, acc[0], acc[1], (double)n);
    var = u.var;
    if (threadIdx.x == 0) {
        state_out[0] = u.mean; state_out[1] = u.var; state_out[2] = u.count;
        if (legacy) {
            if (acc[2] < (double)n) ret[0] = 0.f;                   // some dones == 0 -> index 0
            if (acc[2] > 0.0 && n > 1) ret[1] = 0.f;                // some dones == 1 -> index 1
        }
    }
    const double inv = 1.0 / sqrt(var + epsilon);
    const double lo = -(double)clip, hi = (double)clip;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (threadIdx.x + u * kSmallBlock < n4) {
            nf4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (float)fmin(fmax((double)r4[u][k] * inv, lo), hi);
            reinterpret_cast<nf4*>(out)[threadIdx.x + u * kSmallBlock] = o;
        }
    if (has_tail) out[tail] = (float)fmin(fmax((double)rt * inv, lo), hi);
}

// ---- launchers ------------------------------------------------------------------------------------
static unsigned partial_blocks(int64_t n, int d) {
    int64_t b = (n + 1023) / 1024;                              // >= 4 elements per thread
    if (b > max_partials(d)) b = max_partials(d);
    return (unsigned)(b < 1 ? 1 : b);
}
static unsigned apply_blocks(int64_t n, int d) {
    int64_t b = (n + 4095) / 4096;
    const int64_t cap = d >= 8 ? 256 : 1024;
    if (b > cap) b = cap;
    return (unsigned)(b < 1 ? 1 : b);
}

// scratch: 2 d max_partials(d) doubles (mean / std normaliser), 3 kRewardPartials = 768 doubles (reward normaliser)
hipError_t launch_mean_std_normalize(const float* x, float* y, int d, int64_t ld_x, int64_t ld_y, int64_t n,
                                     const double* state_in, double* state_out, double* partials, float clip,
                                     double epsilon, hipStream_t s) {
    auto addr = [](const void* p) { return reinterpret_cast<uintptr_t>(p); };
    if (n <= kSmallBatch && state_out && y && ((addr(x) | addr(y) | (uintptr_t)(ld_x * 4) | (uintptr_t)(ld_y * 4)) & 15u) == 0) {
        hipLaunchKernelGGL(scg_mean_std_small_kernel, dim3((unsigned)d), dim3(kSmallBlock), 0, s, x, y, ld_x, ld_y, (uint32_t)n, d, state_in,
                           state_out, clip, epsilon);
        return hipGetLastError();
    }
    const unsigned pb = partial_blocks(n, d);
    if (state_out) {
        const dim3 grid(pb, (unsigned)d), block(256);
        const bool vec = ((reinterpret_cast<uintptr_t>(x) | (uintptr_t)(ld_x * 4)) & 15u) == 0;
        if (vec) hipLaunchKernelGGL(scg_moments_kernel<true>, grid, block, 0, s, x, ld_x, (uint32_t)n, state_in, d, partials);
        else hipLaunchKernelGGL(scg_moments_kernel<false>, grid, block, 0, s, x, ld_x, (uint32_t)n, state_in, d, partials);
    }
    const dim3 grid(y ? apply_blocks(n, d) : 1u, (unsigned)d), block(256);      // y == null: one workgroup per dim writes state_out
    const bool vec2 = ((reinterpret_cast<uintptr_t>(x) | (uintptr_t)(ld_x * 4) | reinterpret_cast<uintptr_t>(y) |
                        (uintptr_t)(ld_y * 4)) & 15u) == 0;
    if (vec2) hipLaunchKernelGGL(scg_mean_std_apply_kernel<true>, grid, block, 0, s, x, y, ld_x, ld_y, (uint32_t)n, d,
                                 state_in, state_out, partials, (int)pb, max_partials(d), (double)n, clip, epsilon);
    else hipLaunchKernelGGL(scg_mean_std_apply_kernel<false>, grid, block, 0, s, x, y, ld_x, ld_y, (uint32_t)n, d,
                            state_in, state_out, partials, (int)pb, max_partials(d), (double)n, clip, epsilon);
    return hipGetLastError();
}

// The two passes apart, for a tracker whose batch is spread over several devices: pass 1 leaves sums[d][2] (this shard's
// sum (x - c), sum (x - c)^2 with c = the tracker's current mean -- the same on every shard), the caller adds the shards' sums
// (an all-reduce of 2 d doubles), pass 2 takes the added sums and the total row count.
hipError_t launch_mean_std_moments(const float* x, int d, int64_t ld_x, int64_t n, const double* state_in, double* partials,
                                   double* sums, hipStream_t s) {
    const unsigned pb = partial_blocks(n, d);
    const dim3 grid(pb, (unsigned)d), block(256);
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | (uintptr_t)(ld_x * 4)) & 15u) == 0;
    if (vec) hipLaunchKernelGGL(scg_moments_kernel<true>, grid, block, 0, s, x, ld_x, (uint32_t)n, state_in, d, partials);
    else hipLaunchKernelGGL(scg_moments_kernel<false>, grid, block, 0, s, x, ld_x, (uint32_t)n, state_in, d, partials);
    hipLaunchKernelGGL(scg_sum_partials_kernel<2>, dim3((unsigned)d), dim3(64), 0, s, partials, (int)pb, max_partials(d), sums);
    return hipGetLastError();
}

hipError_t launch_mean_std_apply_sums(const float* x, float* y, int d, int64_t ld_x, int64_t ld_y, int64_t n, const double* state_in,
                                      double* state_out, const double* sums, double batch_n, float clip, double epsilon, hipStream_t s) {
    const dim3 grid(y ? apply_blocks(n, d) : 1u, (unsigned)d), block(256);
    const bool vec2 = ((reinterpret_cast<uintptr_t>(x) | (uintptr_t)(ld_x * 4) | reinterpret_cast<uintptr_t>(y) |
                        (uintptr_t)(ld_y * 4)) & 15u) == 0;
    if (vec2) hipLaunchKernelGGL(scg_mean_std_apply_kernel<true>, grid, block, 0, s, x, y, ld_x, ld_y, (uint32_t)n, d,
                                 state_in, state_out, sums, 1, 1, batch_n, clip, epsilon);
    else hipLaunchKernelGGL(scg_mean_std_apply_kernel<false>, grid, block, 0, s, x, y, ld_x, ld_y, (uint32_t)n, d,
                            state_in, state_out, sums, 1, 1, batch_n, clip, epsilon);
    return hipGetLastError();
}

static unsigned reward_blocks(int64_t n) {
    int64_t b = (n + 4095) / 4096;                              // 16 elements per thread: four 16-byte groups in flight
    if (b > kRewardPartials) b = kRewardPartials;
    return (unsigned)(b < 1 ? 1 : b);
}

hipError_t launch_reward_normalize(const float* rew, const uint8_t* done, float* ret, float* out, int64_t n,
                                   const double* state_in, double* state_out, double* partials, float gamma,
                                   float clip, double epsilon, int legacy, hipStream_t s) {
    auto addr = [](const void* p) { return reinterpret_cast<uintptr_t>(p); };
    if (n <= kSmallBatch && state_out && ((addr(rew) | addr(ret) | addr(out)) & 15u) == 0 && (addr(done) & 3u) == 0) {
        hipLaunchKernelGGL(scg_reward_small_kernel, dim3(1), dim3(kSmallBlock), 0, s, rew, done, ret, out, (uint32_t)n, gamma, state_in,
                           state_out, legacy, clip, epsilon);
        return hipGetLastError();
    }
    const unsigned pb = reward_blocks(n);
    if (state_out) {
        const bool vec = ((addr(rew) | addr(ret)) & 15u) == 0 && (addr(done) & 3u) == 0;
        if (vec) hipLaunchKernelGGL(scg_reward_moments_kernel<true>, dim3(pb), dim3(256), 0, s, rew, done, ret, (uint32_t)n, gamma,
                                    state_in, legacy, partials);
        else hipLaunchKernelGGL(scg_reward_moments_kernel<false>, dim3(pb), dim3(256), 0, s, rew, done, ret, (uint32_t)n, gamma,
                                state_in, legacy, partials);
    }
    const bool vec2 = ((addr(rew) | addr(out)) & 15u) == 0;
    if (vec2) hipLaunchKernelGGL(scg_reward_apply_kernel<true>, dim3(pb), dim3(256), 0, s, rew, out, ret, (uint32_t)n,
                                 state_in, state_out, partials, (int)pb, (double)n, legacy, clip, epsilon);
    else hipLaunchKernelGGL(scg_reward_apply_kernel<false>, dim3(pb), dim3(256), 0, s, rew, out, ret, (uint32_t)n,
                            state_in, state_out, partials, (int)pb, (double)n, legacy, clip, epsilon);
    return hipGetLastError();
}

// The reward normaliser's two passes apart (statistics shared by several shards; the finished envs' returns are cleared, i.e. the
// non-legacy reset): sums[3] = this shard's sum (ret - c), sum (ret - c)^2, number of finished envs.
hipError_t launch_reward_moments(const float* rew, const uint8_t* done, float* ret, int64_t n, const double* state_in, double* partials,
                                 double* sums, float gamma, hipStream_t s) {
    const unsigned pb = reward_blocks(n);
    auto addr = [](const void* p) { return reinterpret_cast<uintptr_t>(p); };
    const bool vec = ((addr(rew) | addr(ret)) & 15u) == 0 && (addr(done) & 3u) == 0;
    if (vec) hipLaunchKernelGGL(scg_reward_moments_kernel<true>, dim3(pb), dim3(256), 0, s, rew, done, ret, (uint32_t)n, gamma,
                                state_in, 0, partials);
    else hipLaunchKernelGGL(scg_reward_moments_kernel<false>, dim3(pb), dim3(256), 0, s, rew, done, ret, (uint32_t)n, gamma,
                            state_in, 0, partials);
    hipLaunchKernelGGL(scg_sum_partials_kernel<3>, dim3(1), dim3(64), 0, s, partials, (int)pb, 0, sums);
    return hipGetLastError();
}

hipError_t launch_reward_apply_sums(const float* rew, float* out, int64_t n, const double* state_in, double* state_out,
                                    const double* sums, double batch_n, float clip, double epsilon, hipStream_t s) {
    const unsigned pb = reward_blocks(n);
    auto addr = [](const void* p) { return reinterpret_cast<uintptr_t>(p); };
    const bool vec2 = ((addr(rew) | addr(out)) & 15u) == 0;
    if (vec2) hipLaunchKernelGGL(scg_reward_apply_kernel<true>, dim3(pb), dim3(256), 0, s, rew, out, (float*)nullptr, (uint32_t)n,
                                 state_in, state_out, sums, 1, batch_n, 0, clip, epsilon);
    else hipLaunchKernelGGL(scg_reward_apply_kernel<false>, dim3(pb), dim3(256), 0, s, rew, out, (float*)nullptr, (uint32_t)n,
                            state_in, state_out, sums, 1, batch_n, 0, clip, epsilon);
    return hipGetLastError();
}

}  // namespace scg
