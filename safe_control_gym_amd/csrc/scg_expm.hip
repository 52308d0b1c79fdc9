// Batched exact discretisation of linear systems on the matrix cores of gfx950.
//
// Reference: discretize_linear_system(A, B, dt, exact=True) (controllers/mpc/mpc_utils.py:24-54):
//     M = [[A, B], [0, 0]],  Md = scipy.linalg.expm(M dt),  Ad = Md[:nx, :nx],  Bd = Md[:nx, nx:]
// called per state by the model-based controllers (lqr_utils.py:39-60, mpc.py:170-175); here for n (A, B) pairs in
// one launch.  This IS GEMM-shaped work (a dozen 16 x 16 products per matrix), so it runs on MFMA:
// v_mfma_f32_16x16x4_f32 -- f32 in, f32 accumulate, bit-for-bit an fmaf chain, at the vector FMA rate but one
// instruction per 1024 FMAs (MI355X_MICROARCH.md, Matrix cores).
//
// One wave owns one 16 x 16 matrix W = M dt / 2^s (nx + m <= 16; systems with nx + m <= 8 / <= 4 pack 2 / 4 envs per
// wave as diagonal blocks -- block-diagonal matrices are closed under products).  Taylor series by Horner,
//     T_j-1 = I + (W / j) T_j,   j = 9 .. 1,   ||W||_inf <= 1/2  ->  truncation < 3e-10,
// followed by s squarings.  Register layouts (cdna_hip_programming.md section 3): the running T lives in the C/D
// accumulator layout (lane l, register i: T[4 (l >> 4) + i][l & 15]); used as the B operand of the next product, register i
// supplies rows k = 4 g + i (g = l >> 4) -- a permutation of the k index, which a sum over k does not care about as long as
// the A operand of that k-step supplies the matching columns: lane l, step i: W[l & 15][4 (l >> 4) + i].  That is four
// CONSECUTIVE floats of one row of W per lane: one 16-byte load straight from the row-major input, no LDS, no shuffles.
// Only a squaring needs the transposed layout of T itself and goes through a 16 x 17 LDS tile (rare: s > 0 means
// ||A|| dt > 1/2).
//
// Epilogue: the rows [0, nx) of E = exp(M dt) are Ad | Bd; optionally z1 = Ad z0 + Bd w per env (the one-step
// prediction of the linear model; with B := c, w := 1 the exact flow of the affine ODE z' = A z + c that
// SymbolicModel.fd_linear_func integrates with CVODES, symbolic_systems.py:102-107).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scg_device.h"

namespace scg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kExpmDegree = 9;
constexpr int kExpmBlock = 256;          // 4 waves = 4 matrices per workgroup

// STRIDE: rows / columns one env occupies in the 16 x 16 tile (16, 8 or 4); PER = 16 / STRIDE envs per wave.
template <int STRIDE>
__global__ __launch_bounds__(kExpmBlock) void scg_expm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              int nx, int m, float dt, float* __restrict__ Ad,
                                                              float* __restrict__ Bd, const float* __restrict__ z0,
                                                              const float* __restrict__ w, float* __restrict__ z1,
                                                              uint32_t n) {
    constexpr int PER = 16 / STRIDE;
    __shared__ float s_t[kExpmBlock / 64][16 * 17];
    const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;                     // A-operand view: row r, columns 4 g .. 4 g + 3
    const uint32_t waves = gridDim.x * (kExpmBlock / 64);
    const uint32_t groups = (n + PER - 1) / PER;
    for (uint32_t grp = blockIdx.x * (kExpmBlock / 64) + wv; grp < groups; grp += waves) {   // wave-uniform loop
        // ---- W = M dt in the A-operand layout: lane (r, g) holds W[r][4 g + i], i = 0..3
        const int e_r = r / STRIDE, lr = r % STRIDE;            // env slot and local row of this lane's row
        const uint32_t env_r = grp * PER + (uint32_t)e_r;
        float a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cc = 4 * g + i, e_c = cc / STRIDE, lc = cc % STRIDE;
            float v = 0.0f;
            if (e_c == e_r && env_r < n && lr < nx) {
                if (lc < nx) v = A[((size_t)env_r * nx + lr) * nx + lc];
                else if (lc < nx + m) v = B[((size_t)env_r * nx + lr) * m + (lc - nx)];
            }
            a[i] = v * dt;
        }
        // ---- scaling: ||A dt||_inf over the wave's envs (the B block enters the series linearly and needs none)
        float rs = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) rs += ((4 * g + i) % STRIDE < nx) ? fabsf(a[i]) : 0.0f;
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);                           // row sum (4 lane groups of one row)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) rs = fmaxf(rs, __shfl_xor(rs, o, 64));   // max over the 16 rows
        const float theta = __builtin_amdgcn_readfirstlane(rs);
        int sq = 0;
        if (theta > 0.5f) {
            int ex;
            (void)frexpf(theta, &ex);                           // theta = f 2^ex, f in [0.5, 1)
            sq = ex + 1;                                        // theta / 2^sq < 0.5
            if (sq > 40) sq = 40;                               // non-finite / absurd input: the result is inf / nan anyway
        }
        const float h = ldexpf(1.0f, -sq);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] *= h;
        // ---- identity in the C/D layout: register i of lane (c = r, g) is element [4 g + i][c]
        f32x4 eye;
#pragma unroll
        for (int i = 0; i < 4; ++i) eye[i] = (4 * g + i == r) ? 1.0f : 0.0f;
        f32x4 t = eye;
        // ---- Horner: T <- I + (W / j) T
#pragma unroll
        for (int j = kExpmDegree; j >= 1; --j) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], t[i], acc, 0, 0, 0);
            const float rj = 1.0f / (float)j;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = fmaf(acc[i], rj, eye[i]);
        }
        // ---- s squarings: T <- T T; the left factor needs T in the A-operand layout = a transposed read through LDS
        for (int q = 0; q < sq; ++q) {                          // wave-uniform trip count
            float* tile = s_t[wv];
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[(4 * g + i) * 17 + r] = t[i];            // element [row 4 g + i][col r]
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            float ta[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ta[i] = tile[r * 17 + 4 * g + i];              // T[r][4 g + i]
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[i], t[i], acc, 0, 0, 0);
            t = acc;
        }
        // ---- epilogue.  C/D layout: lane (c = r, g), register i: E[row 4 g + i][col c]
        const int e_c = r / STRIDE, lc = r % STRIDE;
        const uint32_t env_c = grp * PER + (uint32_t)e_c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = 4 * g + i, e_row = rr / STRIDE, lrow = rr % STRIDE;
            if (e_row == e_c && env_c < n && lrow < nx) {
                if (lc < nx) { if (Ad) Ad[((size_t)env_c * nx + lrow) * nx + lc] = t[i]; }
                else if (lc < nx + m) { if (Bd) Bd[((size_t)env_c * nx + lrow) * m + (lc - nx)] = t[i]; }
            }
        }
        if (z1) {
            // z1 = (E [z0; w])[0 : nx]: each lane multiplies its column's entry, the 16 lanes of a group add up
            float v = 0.0f;
            if (env_c < n) {
                if (lc < nx) v = z0[(size_t)env_c * nx + lc];
                else if (lc < nx + m) v = w ? w[(size_t)env_c * m + (lc - nx)] : 1.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = 4 * g + i, e_row = rr / STRIDE, lrow = rr % STRIDE;
                float pv = (e_row == e_c) ? t[i] * v : 0.0f;    // only this env's diagonal block contributes
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) pv += __shfl_xor(pv, o, 64);
                const uint32_t env_row = grp * PER + (uint32_t)e_row;
                if (r == 0 && env_row < n && lrow < nx) z1[(size_t)env_row * nx + lrow] = pv;
            }
        }
    }
}

hipError_t launch_discretize_exact(const float* A, const float* B, int nx, int m, float dt, float* Ad, float* Bd,
                                   const float* z0, const float* w, float* z1, int64_t n, int num_cus, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int nv = nx + m;
    const int stride = nv <= 4 ? 4 : (nv <= 8 ? 8 : 16);
    const int64_t groups = (n + (16 / stride) - 1) / (16 / stride);
    int64_t blocks = (groups + (kExpmBlock / 64) - 1) / (kExpmBlock / 64);
    const int64_t cap = (int64_t)(num_cus > 0 ? num_cus : 256) * 16;           // grid-stride beyond 16 workgroups per CU
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks), block(kExpmBlock);
    const uint32_t nn = (uint32_t)n;
    if (stride == 4) hipLaunchKernelGGL(scg_expm_kernel<4>, grid, block, 0, s, A, B, nx, m, dt, Ad, Bd, z0, w, z1, nn);
    else if (stride == 8) hipLaunchKernelGGL(scg_expm_kernel<8>, grid, block, 0, s, A, B, nx, m, dt, Ad, Bd, z0, w, z1, nn);
    else hipLaunchKernelGGL(scg_expm_kernel<16>, grid, block, 0, s, A, B, nx, m, dt, Ad, Bd, z0, w, z1, nn);
    return hipGetLastError();
}

}  // namespace scg
