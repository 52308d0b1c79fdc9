// Batched model queries for gfx950: df_func (Jacobians of the continuous dynamics) and the quadratic
// stage cost with its gradients, for N (state, input) pairs in one launch.
//
// Reference: SymbolicModel.setup_linearization (math_and_models/symbolic_systems.py:77-123) builds
// dfdx = jacobian(x_dot, x), dfdu = jacobian(x_dot, u) with CasADi and hands them out as `df_func`;
// `loss` / l_x / l_u / l_xx / l_uu come from the same place (:94-123).  The callers linearise one state at
// a time (controllers/lqr/lqr.py:186, controllers/mpc/mpc.py:170, controllers/lqr/ilqr.py:412-425) and
// discretise with discretize_linear_system (controllers/mpc/mpc_utils.py:24-55; exact=False: Ad = I + A dt,
// Bd = B dt).  Here the Jacobians are the closed forms of the equations in quadrotor.py:594-676 (and the
// textbook cart-pole of DESIGN.md row A16), one wavefront lane per (x, u) pair.
//
// Layout.  Inputs are the engine's SoA planes (state[nx][ld], input[nu][ld]): coalesced 256-byte loads.
// Outputs are the [n][nx][nx] / [n][nx][nu] row-major batches that batched linear algebra (torch.bmm,
// torch.linalg.*) wants.  A lane that stored its own matrix directly would scatter 16 bytes per lane at a
// 576-byte stride; instead every wave stages a few matrix rows of its 64 envs in LDS and copies the staged
// chunk out with lane-contiguous float4 stores, so each store instruction writes whole 64-byte segments.
// The kernel is HBM-write bound: 3-D moves 64 B in and 768 B out per env.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scg_device.h"
#include "scg_math.h"
#include "scg_dynamics.h"

namespace scg {

// ------------------------------------------------------------------------------------------
// closed-form Jacobians of f(x, u)
// ------------------------------------------------------------------------------------------
template <int SYS>
__device__ __forceinline__ void jacobian(const PhysT<float>& c, const float (&x)[Dim<SYS>::NX],
                                         const float (&u)[Dim<SYS>::NU], float (&A)[Dim<SYS>::NX][Dim<SYS>::NX],
                                         float (&B)[Dim<SYS>::NX][Dim<SYS>::NU]) {
    constexpr int NX = Dim<SYS>::NX, NU = Dim<SYS>::NU;
#pragma unroll
    for (int a = 0; a < NX; ++a) {
#pragma unroll
        for (int b = 0; b < NX; ++b) A[a][b] = 0.0f;
#pragma unroll
        for (int b = 0; b < NU; ++b) B[a][b] = 0.0f;
    }
    if constexpr (SYS == SCG_QUAD_1D) {
        // f = [z_dot, T/m - g]                                             quadrotor.py:602
        A[0][1] = 1.0f;
        B[1][0] = c.inv_m;
    } else if constexpr (SYS == SCG_QUAD_2D) {
        // f = [x_dot, sin(th) T/m, z_dot, cos(th) T/m - g, th_dot, l (T2 - T1) / Iyy / sqrt 2]   :618-621
        float s, co;
        sincos_f(x[4], s, co);
        const float Tm = (u[0] + u[1]) * c.inv_m;
        A[0][1] = 1.0f;
        A[1][4] = co * Tm;
        A[2][3] = 1.0f;
        A[3][4] = -s * Tm;
        A[4][5] = 1.0f;
        B[1][0] = B[1][1] = s * c.inv_m;
        B[3][0] = B[3][1] = co * c.inv_m;
        B[5][0] = -c.l_s2_iyy;
        B[5][1] = c.l_s2_iyy;
    } else if constexpr (SYS == SCG_QUAD_3D) {
        // state (x, xd, y, yd, z, zd, phi, theta, psi, p, q, r), input (f1..f4)               :649-674
        float sph, cph, sth, cth, sps, cps;
        sincos_f(x[6], sph, cph);
        sincos_f(x[7], sth, cth);
        sincos_f(x[8], sps, cps);
        const float P = x[9], Q = x[10], R = x[11];
        const float Tm = ((u[0] + u[1]) + (u[2] + u[3])) * c.inv_m;
        const float rcth = rcp_f(cth), tth = sth * rcth;
        A[0][1] = 1.0f;
        A[2][3] = 1.0f;
        A[4][5] = 1.0f;
        // third column of R = Rz(psi) Ry(theta) Rx(phi) and its angle derivatives
        const float rx = fmaf(cps, sth * cph, sps * sph);
        const float ry = fmaf(sps, sth * cph, -(cps * sph));
        const float rz = cth * cph;
        A[1][6] = fmaf(-cps, sth * sph, sps * cph) * Tm;
        A[1][7] = cps * cth * cph * Tm;
        A[1][8] = -ry * Tm;                                    // d rx / d psi = -(sps sth cph - cps sph)
        A[3][6] = -fmaf(sps, sth * sph, cps * cph) * Tm;
        A[3][7] = sps * cth * cph * Tm;
        A[3][8] = rx * Tm;                                     // d ry / d psi = cps sth cph + sps sph
        A[5][6] = -cth * sph * Tm;
        A[5][7] = -sth * cph * Tm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            B[1][j] = rx * c.inv_m;
            B[3][j] = ry * c.inv_m;
            B[5][j] = rz * c.inv_m;
        }
        // Euler-angle rates (quadrotor.py:671-673)
        const float w = fmaf(sph, Q, cph * R);                 // sin(phi) q + cos(phi) r
        const float v = fmaf(cph, Q, -(sph * R));              // cos(phi) q - sin(phi) r
        A[6][6] = tth * v;
        A[6][7] = w * rcth * rcth;
        A[6][9] = 1.0f;
        A[6][10] = sph * tth;
        A[6][11] = cph * tth;
        A[7][6] = -w;
        A[7][10] = cph;
        A[7][11] = -sph;
        A[8][6] = v * rcth;
        A[8][7] = w * tth * rcth;
        A[8][10] = sph * rcth;
        A[8][11] = cph * rcth;
        // body rates: Jinv (Mb - w x J w)  (quadrotor.py:667-670)
        A[9][10] = -c.cxx * R;
        A[9][11] = -c.cxx * Q;
        A[10][9] = -c.cyy * R;
        A[10][11] = -c.cyy * P;
        A[11][9] = -c.czz * Q;
        A[11][10] = -c.czz * P;
        const float kx = c.arm_s2 * c.inv_ixx, ky = c.arm_s2 * c.inv_iyy, kz = c.gamma * c.inv_izz;
        B[9][0] = kx;   B[9][1] = kx;   B[9][2] = -kx;  B[9][3] = -kx;
        B[10][0] = -ky; B[10][1] = ky;  B[10][2] = ky;  B[10][3] = -ky;
        B[11][0] = kz;  B[11][1] = -kz; B[11][2] = kz;  B[11][3] = -kz;
    } else {
        // cart-pole, state (x, x_dot, theta, theta_dot), input F
        float s, co;
        sincos_f(x[2], s, co);
        const float thd = x[3];
        const float iMm = c.cp_inv_Mm;
        const float tmp = fmaf(c.cp_ml * thd * thd, s, u[0]) * iMm;
        const float den = c.cp_l * fmaf(-c.cp_m * iMm, co * co, 4.0f / 3.0f);
        const float rden = rcp_f(den);
        const float thdd = fmaf(c.cp_g, s, -(co * tmp)) * rden;
        const float tmp_th = c.cp_ml * thd * thd * co * iMm;
        const float tmp_thd = 2.0f * c.cp_ml * thd * s * iMm;
        const float den_th = c.cp_l * 2.0f * c.cp_m * iMm * co * s;
        const float thdd_th = (fmaf(c.cp_g, co, s * tmp) - co * tmp_th - thdd * den_th) * rden;
        const float thdd_thd = -co * tmp_thd * rden;
        const float thdd_F = -co * iMm * rden;
        const float k = c.cp_ml * iMm;
        A[0][1] = 1.0f;
        A[1][2] = tmp_th - k * (thdd_th * co - thdd * s);
        A[1][3] = tmp_thd - k * thdd_thd * co;
        A[2][3] = 1.0f;
        A[3][2] = thdd_th;
        A[3][3] = thdd_thd;
        B[1][0] = iMm - k * thdd_F * co;
        B[3][0] = thdd_F;
    }
}

// per-env constants when the inertial properties are randomised (quadrotor.py:455-467)
template <int SYS>
__device__ __forceinline__ PhysT<float> phys_with(const PhysT<float>& shared, const float (&prop)[4]) {
    PhysT<float> c = shared;
    if constexpr (SYS == SCG_CARTPOLE) {
        c.cp_l = prop[0]; c.cp_m = prop[1]; c.cp_inv_Mm = 1.0f / (prop[1] + prop[2]); c.cp_ml = prop[1] * prop[0];
    } else {
        c.inv_m = 1.0f / prop[0];
        c.inv_ixx = 1.0f / prop[1]; c.inv_iyy = 1.0f / prop[2]; c.inv_izz = 1.0f / prop[3];
        c.cxx = (prop[3] - prop[2]) * c.inv_ixx; c.cyy = (prop[1] - prop[3]) * c.inv_iyy; c.czz = (prop[2] - prop[1]) * c.inv_izz;
        c.l_s2_iyy = c.arm_s2 * c.inv_iyy;
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// wave-staged store of a [64 envs][ROWS x COLS] chunk of an [n][TOTAL_ROWS][COLS] batch
// ------------------------------------------------------------------------------------------
#ifndef SCG_LIN_BLOCK
#define SCG_LIN_BLOCK 256      // threads per workgroup
#endif
#ifndef SCG_LIN_EP3
#define SCG_LIN_EP3 16         // 3-D: envs staged per pass (576-B records): LDS per wave = EP * 576 B
#endif
#ifndef SCG_LIN_NT
#define SCG_LIN_NT 1           // nontemporal hint on the output stores
#endif
#ifndef SCG_LIN_MIN_WAVES
#define SCG_LIN_MIN_WAVES 1    // waves per SIMD the register allocator must allow
#endif
constexpr int kLinBlock = SCG_LIN_BLOCK;
template <class V>
__device__ __forceinline__ void st_out(V v, V* dst) {
#if SCG_LIN_NT
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;
#endif
}
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

// Copy the REC-float records of this wave's 64 envs out of registers into global memory through LDS, EP envs per
// pass: the lanes of env group h write their records into the wave's staging region ([EP][REC], the layout of
// the output batch itself), then all 64 lanes copy the region out as one contiguous run of EP * REC floats.
// Only whole records keep every store on full 128-byte lines: staging row groups instead (segments of 96-288 B at
// a 576 B stride) measured 1.3-3x slower.
template <int REC, int EP>
__device__ __forceinline__ void stage_and_store(const float (&vals)[REC], float* lds, float* __restrict__ out,
                                                uint32_t env0, uint32_t n, int lane) {
    constexpr int VW = (REC % 4 == 0) ? 4 : ((REC % 2 == 0) ? 2 : 1);
    constexpr int UNITS = EP * REC / VW;              // vector units per pass
    static_assert(64 % EP == 0, "EP must divide the wave");
#pragma unroll
    for (int h = 0; h < 64 / EP; ++h) {
        if (EP == 64 || (lane / EP) == h) {
            float* mine = lds + (lane % EP) * REC;
#pragma unroll
            for (int k = 0; k < REC; k += VW) {
                if constexpr (VW == 4) *reinterpret_cast<vf4*>(mine + k) = vf4{vals[k], vals[k + 1], vals[k + 2], vals[k + 3]};
                else if constexpr (VW == 2) *reinterpret_cast<vf2*>(mine + k) = vf2{vals[k], vals[k + 1]};
                else mine[k] = vals[k];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t first = env0 + (uint32_t)(h * EP);                         // first env of this pass
        const uint32_t live = first < n ? (n - first < (uint32_t)EP ? n - first : (uint32_t)EP) : 0u;
        const uint32_t limit = live * (uint32_t)(REC / VW);                       // units that belong to real envs
        float* dst0 = out + (size_t)first * REC;
#pragma unroll
        for (int t = 0; t < (UNITS + 63) / 64; ++t) {
            const uint32_t q = (uint32_t)(t * 64 + lane);
            if (q < limit) {
                if constexpr (VW == 4) st_out(*reinterpret_cast<const vf4*>(lds + q * 4), reinterpret_cast<vf4*>(dst0 + q * 4));
                else if constexpr (VW == 2) st_out(*reinterpret_cast<const vf2*>(lds + q * 2), reinterpret_cast<vf2*>(dst0 + q * 2));
                else st_out(lds[q], dst0 + q);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();              // the region is reused by the next pass
    }
}

// envs staged per pass: the largest power of two whose records fit the wave's LDS budget
constexpr int lin_ep(int rec, int budget_floats) {
    int ep = 64;
    while (ep > 1 && ep * rec > budget_floats) ep /= 2;
    return ep;
}
template <int SYS>
constexpr int lin_lds_floats_per_wave() {
    constexpr int NX = Dim<SYS>::NX;
    // 3-D: SCG_LIN_EP3 records of 144 floats; the small systems stage all 64 records at once
    return (SYS == SCG_QUAD_3D ? SCG_LIN_EP3 : 64) * NX * NX;
}

template <int SYS, bool DIRECT>
__global__ __launch_bounds__(kLinBlock, SCG_LIN_MIN_WAVES) void scg_linearize_kernel(PhysT<float> shared, const float* __restrict__ state,
                                                                   const float* __restrict__ input,
                                                                   const float* __restrict__ inertial,
                                                                   float* __restrict__ dfdx, float* __restrict__ dfdu,
                                                                   float disc_dt, int64_t ld, uint32_t n) {
    constexpr int NX = Dim<SYS>::NX, NU = Dim<SYS>::NU;
    extern __shared__ float lds_all[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* lds = lds_all + wave * lin_lds_floats_per_wave<SYS>();
    const uint32_t env0 = (blockIdx.x * (uint32_t)kLinBlock) + (uint32_t)(wave * 64);
    if (env0 >= n) return;                                           // whole wave past the end (wave-uniform)
    const uint32_t i = env0 + lane < n ? env0 + lane : n - 1;        // tail lanes recompute the last env, store nothing
    float x[NX], u[NU];
#pragma unroll
    for (int j = 0; j < NX; ++j) x[j] = ldp(uptr(state + j * ld), i * 4u);
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = ldp(uptr(input + j * ld), i * 4u);
    PhysT<float> c = shared;
    if (inertial) {
        float prop[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) prop[q] = ldp(uptr(inertial + q * ld), i * 4u);
        c = phys_with<SYS>(shared, prop);
    }
    float A[NX][NX], B[NX][NU];
    jacobian<SYS>(c, x, u, A, B);
    if (disc_dt > 0.0f) {
        // discretize_linear_system(exact=False): Ad = I + A dt, Bd = B dt  (mpc_utils.py:51-54)
#pragma unroll
        for (int a = 0; a < NX; ++a) {
#pragma unroll
            for (int b = 0; b < NX; ++b) A[a][b] = fmaf(A[a][b], disc_dt, a == b ? 1.0f : 0.0f);
#pragma unroll
            for (int b = 0; b < NU; ++b) B[a][b] *= disc_dt;
        }
    }
    if constexpr (DIRECT) {
        // reference variant for the tuning log: every lane streams its own matrices
        if (env0 + lane < n) {
            float* pa = dfdx + (size_t)i * NX * NX;
            float* pb = dfdu + (size_t)i * NX * NU;
#pragma unroll
            for (int a = 0; a < NX; ++a) {
#pragma unroll
                for (int b = 0; b < NX; ++b) __builtin_nontemporal_store(A[a][b], pa + a * NX + b);
#pragma unroll
                for (int b = 0; b < NU; ++b) __builtin_nontemporal_store(B[a][b], pb + a * NU + b);
            }
        }
    } else {
        constexpr int LW = lin_lds_floats_per_wave<SYS>();
        float va[NX * NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
#pragma unroll
            for (int b = 0; b < NX; ++b) va[a * NX + b] = A[a][b];
        }
        stage_and_store<NX * NX, lin_ep(NX * NX, LW)>(va, lds, dfdx, env0, n, lane);
        float vb[NX * NU];
#pragma unroll
        for (int a = 0; a < NX; ++a) {
#pragma unroll
            for (int b = 0; b < NU; ++b) vb[a * NU + b] = B[a][b];
        }
        stage_and_store<NX * NU, lin_ep(NX * NU, LW)>(vb, lds, dfdu, env0, n, lane);
    }
}

// ------------------------------------------------------------------------------------------
// quadratic stage cost  l = 1/2 (x - xr)' Q (x - xr) + 1/2 (u - ur)' R (u - ur)   (quadrotor.py:682)
// with l_x = 1/2 (Q + Q') ex, l_u = 1/2 (R + R') eu  (symbolic_systems.py:94-103)
// ------------------------------------------------------------------------------------------
struct LossWeights {
    float Q[SCG_MAX_NX * SCG_MAX_NX];     // as given (row-major nx x nx)
    float R[SCG_MAX_NU * SCG_MAX_NU];
};

template <int NX, int NU>
__global__ __launch_bounds__(256) void scg_loss_kernel(LossWeights w, const float* __restrict__ state,
                                                       const float* __restrict__ input, const float* __restrict__ xr,
                                                       int64_t xr_ld, const float* __restrict__ ur, int64_t ur_ld,
                                                       float* __restrict__ l, float* __restrict__ l_x,
                                                       float* __restrict__ l_u, int64_t ld, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float ex[NX], eu[NU];
#pragma unroll
    for (int j = 0; j < NX; ++j)
        ex[j] = ldp(uptr(state + j * ld), i * 4u) - (xr_ld ? ldp(uptr(xr + j * xr_ld), i * 4u) : xr[j]);
#pragma unroll
    for (int j = 0; j < NU; ++j)
        eu[j] = ldp(uptr(input + j * ld), i * 4u) - (ur_ld ? ldp(uptr(ur + j * ur_ld), i * 4u) : ur[j]);
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < NX; ++a) {
        float row = 0.0f, sym = 0.0f;
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            row = fmaf(w.Q[a * NX + b], ex[b], row);
            sym = fmaf(0.5f * (w.Q[a * NX + b] + w.Q[b * NX + a]), ex[b], sym);
        }
        acc = fmaf(ex[a], row, acc);
        if (l_x) stp(uptr(l_x + a * ld), i * 4u, sym);
    }
#pragma unroll
    for (int a = 0; a < NU; ++a) {
        float row = 0.0f, sym = 0.0f;
#pragma unroll
        for (int b = 0; b < NU; ++b) {
            row = fmaf(w.R[a * NU + b], eu[b], row);
            sym = fmaf(0.5f * (w.R[a * NU + b] + w.R[b * NU + a]), eu[b], sym);
        }
        acc = fmaf(eu[a], row, acc);
        if (l_u) stp(uptr(l_u + a * ld), i * 4u, sym);
    }
    if (l) stp(l, i * 4u, 0.5f * acc);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// linearised model: SymbolicModel.fc_linear_func (math_and_models/symbolic_systems.py:93-101)
// ------------------------------------------------------------------------------------------
// x_dot_linear(x_eval, u_eval; x, u) = f(x, u) + dfdx(x, u) (x_eval - x) + dfdu(x, u) (u_eval - u), one lane per env; all
// four arguments and the result are SoA planes.  x_eval == nullptr evaluates at x_eval = 0, which yields the constant
// term c of the affine ODE  z' = A z + c  that fd_linear_func (:102-107) integrates (see scg_discretize_exact).
template <int SYS>
__global__ __launch_bounds__(256) void scg_fc_linear_kernel(PhysT<float> shared, const float* __restrict__ x_eval,
                                                            const float* __restrict__ u_eval,
                                                            const float* __restrict__ state,
                                                            const float* __restrict__ input,
                                                            const float* __restrict__ inertial, float* __restrict__ out,
                                                            int64_t ld, uint32_t n) {
    constexpr int NX = Dim<SYS>::NX, NU = Dim<SYS>::NU;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float x[NX], u[NU], dx[NX], du[NU];
#pragma unroll
        for (int j = 0; j < NX; ++j) { x[j] = state[j * ld + i]; dx[j] = (x_eval ? x_eval[j * ld + i] : 0.0f) - x[j]; }
#pragma unroll
        for (int j = 0; j < NU; ++j) { u[j] = input[j * ld + i]; du[j] = u_eval[j * ld + i] - u[j]; }
        PhysT<float> c = shared;
        if (inertial) {
            float prop[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) prop[q] = inertial[q * ld + i];
            c = phys_with<SYS>(shared, prop);
        }
        float A[NX][NX], B[NX][NU], f[NX];
        jacobian<SYS>(c, x, u, A, B);
        Inputs<SYS, float> in;
        make_inputs<SYS, float>(c, u, 0.f, 0.f, 0.f, in);
        f_eval<SYS, float>(c, in, x, f);
#pragma unroll
        for (int a = 0; a < NX; ++a) {
            float v = f[a];
#pragma unroll
            for (int b = 0; b < NX; ++b) v = fmaf(A[a][b], dx[b], v);
#pragma unroll
            for (int b = 0; b < NU; ++b) v = fmaf(B[a][b], du[b], v);
            out[a * ld + i] = v;
        }
    }
}

hipError_t launch_fc_linear(int system, const PhysT<float>& ph, const float* x_eval, const float* u_eval, const float* state,
                            const float* input, const float* inertial, float* out, int64_t ld, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const dim3 grid((unsigned)blocks), block(256);
    const uint32_t nn = (uint32_t)n;
    switch (system) {
        case SCG_QUAD_1D: hipLaunchKernelGGL(scg_fc_linear_kernel<SCG_QUAD_1D>, grid, block, 0, s, ph, x_eval, u_eval, state, input, inertial, out, ld, nn); break;
        case SCG_QUAD_2D: hipLaunchKernelGGL(scg_fc_linear_kernel<SCG_QUAD_2D>, grid, block, 0, s, ph, x_eval, u_eval, state, input, inertial, out, ld, nn); break;
        case SCG_QUAD_3D: hipLaunchKernelGGL(scg_fc_linear_kernel<SCG_QUAD_3D>, grid, block, 0, s, ph, x_eval, u_eval, state, input, inertial, out, ld, nn); break;
        case SCG_CARTPOLE: hipLaunchKernelGGL(scg_fc_linear_kernel<SCG_CARTPOLE>, grid, block, 0, s, ph, x_eval, u_eval, state, input, inertial, out, ld, nn); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int SYS>
static hipError_t launch_lin_sys(const PhysT<float>& ph, const float* state, const float* input, const float* inertial,
                                 float* dfdx, float* dfdu, float disc_dt, int64_t ld, uint32_t n, bool direct,
                                 hipStream_t s) {
    const dim3 grid((n + kLinBlock - 1) / kLinBlock), block(kLinBlock);
    if (direct) {
        hipLaunchKernelGGL((scg_linearize_kernel<SYS, true>), grid, block, 0, s, ph, state, input, inertial, dfdx, dfdu,
                           disc_dt, ld, n);
    } else {
        const size_t lds = (size_t)(kLinBlock / 64) * lin_lds_floats_per_wave<SYS>() * sizeof(float);
        hipLaunchKernelGGL((scg_linearize_kernel<SYS, false>), grid, block, lds, s, ph, state, input, inertial, dfdx,
                           dfdu, disc_dt, ld, n);
    }
    return hipGetLastError();
}

hipError_t launch_linearize(int system, const PhysT<float>& ph, const float* state, const float* input,
                            const float* inertial, float* dfdx, float* dfdu, float disc_dt, int64_t ld, int64_t n,
                            hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const bool direct = false;     // true = every lane streams its own record without LDS staging (16x slower)
    switch (system) {
        case SCG_QUAD_1D: return launch_lin_sys<SCG_QUAD_1D>(ph, state, input, inertial, dfdx, dfdu, disc_dt, ld, (uint32_t)n, direct, s);
        case SCG_QUAD_2D: return launch_lin_sys<SCG_QUAD_2D>(ph, state, input, inertial, dfdx, dfdu, disc_dt, ld, (uint32_t)n, direct, s);
        case SCG_QUAD_3D: return launch_lin_sys<SCG_QUAD_3D>(ph, state, input, inertial, dfdx, dfdu, disc_dt, ld, (uint32_t)n, direct, s);
        case SCG_CARTPOLE: return launch_lin_sys<SCG_CARTPOLE>(ph, state, input, inertial, dfdx, dfdu, disc_dt, ld, (uint32_t)n, direct, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_loss(int system, const double* Q, const double* R, const float* state, const float* input,
                       const float* xr, int64_t xr_ld, const float* ur, int64_t ur_ld, float* l, float* l_x, float* l_u,
                       int64_t ld, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int nx = 0, nu = 0;
    switch (system) {
        case SCG_QUAD_1D: nx = 2; nu = 1; break;
        case SCG_QUAD_2D: nx = 6; nu = 2; break;
        case SCG_QUAD_3D: nx = 12; nu = 4; break;
        case SCG_CARTPOLE: nx = 4; nu = 1; break;
        default: return hipErrorInvalidValue;
    }
    LossWeights w;
    for (int a = 0; a < nx * nx; ++a) w.Q[a] = (float)Q[a];
    for (int a = 0; a < nu * nu; ++a) w.R[a] = (float)R[a];
    const dim3 grid(((uint32_t)n + 255u) / 256u), block(256);
    const uint32_t nn = (uint32_t)n;
    switch (system) {
        case SCG_QUAD_1D: hipLaunchKernelGGL((scg_loss_kernel<2, 1>), grid, block, 0, s, w, state, input, xr, xr_ld, ur, ur_ld, l, l_x, l_u, ld, nn); break;
        case SCG_QUAD_2D: hipLaunchKernelGGL((scg_loss_kernel<6, 2>), grid, block, 0, s, w, state, input, xr, xr_ld, ur, ur_ld, l, l_x, l_u, ld, nn); break;
        case SCG_QUAD_3D: hipLaunchKernelGGL((scg_loss_kernel<12, 4>), grid, block, 0, s, w, state, input, xr, xr_ld, ur, ur_ld, l, l_x, l_u, ld, nn); break;
        case SCG_CARTPOLE: hipLaunchKernelGGL((scg_loss_kernel<4, 1>), grid, block, 0, s, w, state, input, xr, xr_ld, ur, ur_ld, l, l_x, l_u, ld, nn); break;
    }
    return hipGetLastError();
}

}  // namespace scg
