// SymbolicModel.fd_func as the reference defines it, for N (x0, p) pairs in one launch.
//
// Reference: math_and_models/symbolic_systems.py:70-73 builds `fd_func = cs.integrator('fd', 'cvodes', {x, p, ode}, {'tf': dt})`:
// the FLOW of x' = f(x, p) over [0, dt] under a constant input, integrated by CVODES to a requested tolerance (CasADi's
// defaults: reltol 1e-6, abstol 1e-8), in float64, with forward sensitivities d xf / d x0 and d xf / d p on request
// (the integrator's `jac_xf_x0` / `jac_xf_p`).  The engine's fixed-step RK4 (`scg_model_eval`, the reference's rk_discrete)
// answers the same question to ~5e-8 per control step; this entry point answers it to the tolerance the caller states.
//
// Method (one lane = one flow, all in fp64 -- the model's double constants, the fdlibm-grade sincos of scg_dynamics.h):
//   * error control by step doubling: m and 2m RK4 steps of the same interval; |y_2m - y_m| / 15 estimates the error of
//     y_2m (Richardson, order 4); accept when it is below atol + rtol |y| in every component, else double m (at most
//     2^max_doublings); the value handed back is the extrapolated y_2m + (y_2m - y_m) / 15 (order 5).  These dynamics are
//     smooth and dt is a control period: m = 1 is accepted at the CVODES default tolerances almost everywhere, so the
//     lanes of a wave rarely diverge.
//   * sensitivities by internal numerical differentiation (Bock): central differences of THE SAME discretisation (the m
//     frozen at the nominal's value) with an exactly representable step of eps^(1/3) max(|v|, 1): the difference quotient
//     of a smooth map, 1e-9-1e-10 relative -- four orders below the integrator tolerance it replaces.  One lane per
//     (env, column): the nx + nu lanes of an env perturb one coordinate each, so a store instruction writes runs of
//     nx (nu) consecutive doubles of the [n][nx][nx] ([n][nx][nu]) row-major batches, and a batch of a few trajectories
//     (an MPC horizon) still fills waves.  The lanes of an env repeat the nominal flow (identical arithmetic, same m): no
//     cross-lane traffic, no LDS.
// Compute-bound on the fp64 VALU; not a bandwidth kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scg_device.h"
#include "scg_math.h"
#include "scg_dynamics.h"

namespace scg {

constexpr int kFlowBlock = 256;

template <int SYS>
__device__ __forceinline__ PhysT<double> flow_phys(const PhysT<double>& shared, const double (&prop)[4]) {
    // per-env (M, Ixx, Iyy, Izz) / (pole_length, pole_mass, cart_mass, -): quadrotor.py:455-467
    PhysT<double> c = shared;
    if constexpr (SYS == SCG_CARTPOLE) {
        c.cp_l = prop[0]; c.cp_m = prop[1]; c.cp_inv_Mm = 1.0 / (prop[1] + prop[2]); c.cp_ml = prop[1] * prop[0];
    } else {
        c.inv_m = 1.0 / prop[0];
        c.inv_ixx = 1.0 / prop[1]; c.inv_iyy = 1.0 / prop[2]; c.inv_izz = 1.0 / prop[3];
        c.cxx = (prop[3] - prop[2]) * c.inv_ixx; c.cyy = (prop[1] - prop[3]) * c.inv_iyy; c.czz = (prop[2] - prop[1]) * c.inv_izz;
        c.l_s2_iyy = c.arm_s2 * c.inv_iyy;
    }
    return c;
}

// y = m RK4 steps of dt / m from x0 (zero-order-hold input)
template <int SYS>
__device__ __forceinline__ void rk4_chain(PhysT<double> ph, const Inputs<SYS, double>& in, const double (&x0)[Dim<SYS>::NX],
                                          double dt, int m, double (&y)[Dim<SYS>::NX]) {
    constexpr int NX = Dim<SYS>::NX;
    ph.h = dt / (double)m;
#pragma unroll
    for (int j = 0; j < NX; ++j) y[j] = x0[j];
#pragma unroll 1
    for (int s = 0; s < m; ++s) {
        double inc[NX];
        integrate<SYS, double>(ph, 0u, in, y, inc);
#pragma unroll
        for (int j = 0; j < NX; ++j) y[j] += inc[j];
    }
}

// the order-5 value of the (m, 2m) pair
template <int SYS>
__device__ __forceinline__ void extrapolated_flow(const PhysT<double>& ph, const Inputs<SYS, double>& in,
                                                  const double (&x0)[Dim<SYS>::NX], double dt, int m, double (&y)[Dim<SYS>::NX]) {
    constexpr int NX = Dim<SYS>::NX;
    double coarse[NX];
    rk4_chain<SYS>(ph, in, x0, dt, m, coarse);
    rk4_chain<SYS>(ph, in, x0, dt, 2 * m, y);
#pragma unroll
    for (int j = 0; j < NX; ++j) y[j] += (y[j] - coarse[j]) * (1.0 / 15.0);
}

template <int SYS, bool SENS>
__global__ __launch_bounds__(kFlowBlock) void scg_flow_kernel(const PhysT<double> shared, const double* __restrict__ state,
                                                              const double* __restrict__ input,
                                                              const double* __restrict__ inertial, double dt, double rtol,
                                                              double atol, int max_doublings, double* __restrict__ xf,
                                                              double* __restrict__ dxf_dx0, double* __restrict__ dxf_dp,
                                                              int32_t* __restrict__ intervals, int64_t ld, int64_t n) {
    constexpr int NX = Dim<SYS>::NX, NU = Dim<SYS>::NU, NC = NX + NU;
    const int64_t t = (int64_t)blockIdx.x * kFlowBlock + threadIdx.x;
    const int64_t e = SENS ? t / NC : t;
    const int col = SENS ? (int)(t - e * NC) : 0;
    if (e >= n) return;
    double x0[NX], u[NU];
#pragma unroll
    for (int j = 0; j < NX; ++j) x0[j] = state[(int64_t)j * ld + e];
#pragma unroll
    for (int j = 0; j < NU; ++j) u[j] = input[(int64_t)j * ld + e];
    PhysT<double> ph = shared;
    if (inertial) {
        double prop[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) prop[q] = inertial[(int64_t)q * ld + e];
        ph = flow_phys<SYS>(shared, prop);
    }
    Inputs<SYS, double> in;
    make_inputs<SYS, double>(ph, u, 0.0, 0.0, 0.0, in);

    // ---- the nominal flow, error-controlled
    double y1[NX], y2[NX];
    int m = 1, lvl = 0;
    bool ok;
    rk4_chain<SYS>(ph, in, x0, dt, m, y1);
    for (;;) {
        rk4_chain<SYS>(ph, in, x0, dt, 2 * m, y2);
        ok = true;
#pragma unroll
        for (int j = 0; j < NX; ++j) ok = ok && (fabs(y2[j] - y1[j]) <= 15.0 * fma(rtol, fabs(y2[j]), atol));   // false for a NaN
        if (ok || lvl >= max_doublings) break;
#pragma unroll
        for (int j = 0; j < NX; ++j) y1[j] = y2[j];
        m *= 2;
        ++lvl;
    }
    if (!SENS || col == 0) {
        if (xf) {
#pragma unroll
            for (int j = 0; j < NX; ++j) xf[(int64_t)j * ld + e] = y2[j] + (y2[j] - y1[j]) * (1.0 / 15.0);
        }
        if (intervals) intervals[e] = ok ? 2 * m : -2 * m;       // RK4 steps of the accepted solution; negative: tolerance not met
    }
    if constexpr (SENS) {
        const bool is_x = col < NX;
        if ((is_x && !dxf_dx0) || (!is_x && !dxf_dp)) return;
        double base = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) base = (col == j) ? x0[j] : base;
#pragma unroll
        for (int j = 0; j < NU; ++j) base = (col == NX + j) ? u[j] : base;
        // step eps^(1/3) max(|v|, 1), made exactly representable: (base + h) - base
        double h = 6.0554544523933395e-6 * fmax(fabs(base), 1.0);
        {
            double moved = base + h;
            asm volatile("" : "+v"(moved));                  // keep the sum a rounded double (no algebraic folding of moved - base)
            h = moved - base;
        }
        double d[NX];
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const double delta = side == 0 ? h : -h;
            double xp[NX], up[NU], y[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) xp[j] = (col == j) ? x0[j] + delta : x0[j];
#pragma unroll
            for (int j = 0; j < NU; ++j) up[j] = (col == NX + j) ? u[j] + delta : u[j];
            Inputs<SYS, double> inp;
            make_inputs<SYS, double>(ph, up, 0.0, 0.0, 0.0, inp);
            extrapolated_flow<SYS>(ph, inp, xp, dt, m, y);
#pragma unroll
            for (int j = 0; j < NX; ++j) d[j] = side == 0 ? y[j] : d[j] - y[j];
        }
        const double inv = 0.5 / h;
        if (is_x) {
#pragma unroll
            for (int i = 0; i < NX; ++i) dxf_dx0[(e * NX + i) * NX + col] = d[i] * inv;
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) dxf_dp[(e * NX + i) * NU + (col - NX)] = d[i] * inv;
        }
    }
}

template <int SYS>
static hipError_t launch_flow_sys(const PhysT<double>& ph, const double* state, const double* input, const double* inertial,
                                  double dt, double rtol, double atol, int max_doublings, double* xf, double* dxf_dx0,
                                  double* dxf_dp, int32_t* intervals, int64_t ld, int64_t n, hipStream_t s) {
    constexpr int NC = Dim<SYS>::NX + Dim<SYS>::NU;
    const bool sens = dxf_dx0 || dxf_dp;
    const int64_t threads = sens ? n * NC : n;
    const dim3 grid((uint32_t)((threads + kFlowBlock - 1) / kFlowBlock)), block(kFlowBlock);
    if (sens) {
        hipLaunchKernelGGL((scg_flow_kernel<SYS, true>), grid, block, 0, s, ph, state, input, inertial, dt, rtol, atol,
                           max_doublings, xf, dxf_dx0, dxf_dp, intervals, ld, n);
    } else {
        hipLaunchKernelGGL((scg_flow_kernel<SYS, false>), grid, block, 0, s, ph, state, input, inertial, dt, rtol, atol,
                           max_doublings, xf, dxf_dx0, dxf_dp, intervals, ld, n);
    }
    return hipGetLastError();
}

hipError_t launch_model_flow(int system, const PhysT<double>& ph, const double* state, const double* input,
                             const double* inertial, double dt, double rtol, double atol, int max_doublings, double* xf,
                             double* dxf_dx0, double* dxf_dp, int32_t* intervals, int64_t ld, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    switch (system) {
        case SCG_QUAD_1D: return launch_flow_sys<SCG_QUAD_1D>(ph, state, input, inertial, dt, rtol, atol, max_doublings, xf, dxf_dx0, dxf_dp, intervals, ld, n, s);
        case SCG_QUAD_2D: return launch_flow_sys<SCG_QUAD_2D>(ph, state, input, inertial, dt, rtol, atol, max_doublings, xf, dxf_dx0, dxf_dp, intervals, ld, n, s);
        case SCG_QUAD_3D: return launch_flow_sys<SCG_QUAD_3D>(ph, state, input, inertial, dt, rtol, atol, max_doublings, xf, dxf_dx0, dxf_dp, intervals, ld, n, s);
        case SCG_CARTPOLE: return launch_flow_sys<SCG_CARTPOLE>(ph, state, input, inertial, dt, rtol, atol, max_doublings, xf, dxf_dx0, dxf_dp, intervals, ld, n, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace scg
