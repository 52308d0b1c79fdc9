// The prior model's continuous dynamics x_dot = f(x, u) and the fixed-step integrators, shared by the step kernel
// (scg_kernels.hip), the stand-alone model queries and the linearised-model kernels (scg_linearize.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "scg_device.h"
#include "scg_math.h"

namespace scg {

// ------------------------------------------------------------------------------------------
// continuous dynamics  x_dot = f(x, u)   (quadrotor.py:594-676; cartpole: DESIGN.md A16)
// ------------------------------------------------------------------------------------------
// All of it is generic over the real type T: float for the default path, double for the precise mode
// (compensated_state), where the state travels as hi + lo float32 planes and the ODE is integrated in fp64
// (fp64 FMA issues at the same 4-cycle rate as unpacked fp32 on this chip).
template <class T> struct Real;
template <> struct Real<float> {
    static __device__ __forceinline__ float rcp(float x) { return rcp_f(x); }
    static __device__ __forceinline__ void sincos(float x, float& s, float& c) { sincos_f(x, s, c); }
};
template <> struct Real<double> {
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    // fdlibm-style kernels on [-pi/4, pi/4] after a two-piece FMA Cody-Waite reduction; |err| < 4e-15 for |x| < 50
    static __device__ __forceinline__ void sincos(double x, double& s, double& c) {
        const double kf = rint(x * 0.63661977236758134308);
        const int k = (int)kf;
        double r = fma(kf, -1.5707963267948966, x);
        r = fma(kf, -6.123233995736766e-17, r);
        const double z = r * r;
        double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
        ps = fma(ps, z, 2.75573137070700676789e-06);
        ps = fma(ps, z, -1.98412698298579493134e-04);
        ps = fma(ps, z, 8.33333333332248946124e-03);
        ps = fma(ps, z, -1.66666666666666324348e-01);
        const double sr = fma(r * z, ps, r);
        double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
        pc = fma(pc, z, -2.75573143513906633035e-07);
        pc = fma(pc, z, 2.48015872894767294178e-05);
        pc = fma(pc, z, -1.38888888888741095749e-03);
        pc = fma(pc, z, 4.16666666666666019037e-02);
        const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
        const double ss = (k & 1) ? cr : sr, cc = (k & 1) ? sr : cr;
        s = (k & 2) ? -ss : ss;
        c = ((k + 1) & 2) ? -cc : cc;
    }
};
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }

template <int SYS, class T> struct Inputs;
template <class T> struct Inputs<SCG_QUAD_1D, T> { T acc; };                       // T/m - g (+ Fz/m)
template <class T> struct Inputs<SCG_QUAD_2D, T> { T Tm, ax, az, thdd; };
template <class T> struct Inputs<SCG_QUAD_3D, T> { T Tm, ax, ay, az, mx, my, mz; };
template <class T> struct Inputs<SCG_CARTPOLE, T> { T F; };

// ZOH inputs of one step.  In the precise mode the physical constants are the float-rounded kernarg values
// widened to double -- the same constants the float path uses, so the two modes integrate the same model.
// FORCE = false: no external force can act in this instantiation (lean / info tiers): ax = ay = 0 and az = -g are the values
// 0 * inv_m and fma(0, inv_m, -g) take -- as literals / one SGPR instead of three VGPRs held across the whole integration.
template <int SYS, class T, bool FORCE = true>
__device__ __forceinline__ void make_inputs(const PhysT<T>& c, const T* u, T fx, T fy, T fz,
                                            Inputs<SYS, T>& in) {
    const T inv_m = c.inv_m, g = c.g;
    if constexpr (SYS == SCG_QUAD_1D) {
        in.acc = fma_t(u[0] + (T)fz, inv_m, -g);                                // quadrotor.py:602
    } else if constexpr (SYS == SCG_QUAD_2D) {
        in.Tm = (u[0] + u[1]) * inv_m;                                          // quadrotor.py:619-620
        in.ax = FORCE ? (T)fx * inv_m : (T)0;
        in.az = FORCE ? fma_t((T)fz, inv_m, -g) : -g;
        in.thdd = c.l_s2_iyy * (u[1] - u[0]);                                // quadrotor.py:621
    } else if constexpr (SYS == SCG_QUAD_3D) {
        in.Tm = ((u[0] + u[1]) + (u[2] + u[3])) * inv_m;                        // quadrotor.py:664
        in.ax = FORCE ? (T)fx * inv_m : (T)0;
        in.ay = FORCE ? (T)fy * inv_m : (T)0;
        in.az = FORCE ? fma_t((T)fz, inv_m, -g) : -g;
        in.mx = c.arm_s2 * ((u[0] + u[1]) - (u[2] + u[3])) * c.inv_ixx;   // quadrotor.py:667
        in.my = c.arm_s2 * ((u[1] + u[2]) - (u[0] + u[3])) * c.inv_iyy;   // quadrotor.py:668
        in.mz = c.gamma * ((u[0] + u[2]) - (u[1] + u[3])) * c.inv_izz;    // quadrotor.py:669
    } else {
        in.F = u[0] + (T)fx;
    }
}

template <int SYS, class T>
__device__ __forceinline__ void f_eval(const PhysT<T>& c, const Inputs<SYS, T>& in, const T (&x)[Dim<SYS>::NX],
                                       T (&d)[Dim<SYS>::NX]) {
    if constexpr (SYS == SCG_QUAD_1D) {
        d[0] = x[1];
        d[1] = in.acc;
    } else if constexpr (SYS == SCG_QUAD_2D) {
        T s, co;
        Real<T>::sincos(x[4], s, co);
        d[0] = x[1];
        d[1] = fma_t(s, in.Tm, in.ax);
        d[2] = x[3];
        d[3] = fma_t(co, in.Tm, in.az);
        d[4] = x[5];
        d[5] = in.thdd;
    } else if constexpr (SYS == SCG_QUAD_3D) {
        T sph, cph, sth, cth, sps, cps;
        Real<T>::sincos(x[6], sph, cph);
        Real<T>::sincos(x[7], sth, cth);
        Real<T>::sincos(x[8], sps, cps);
        const T P = x[9], Q = x[10], R = x[11];
        // third column of R = Rz(psi) Ry(theta) Rx(phi)  (transformations.py:125-182)
        const T sthcph = sth * cph;
        d[0] = x[1];
        d[1] = fma_t(fma_t(cps, sthcph, sps * sph), in.Tm, in.ax);
        d[2] = x[3];
        d[3] = fma_t(fma_t(sps, sthcph, -(cps * sph)), in.Tm, in.ay);
        d[4] = x[5];
        d[5] = fma_t(cth * cph, in.Tm, in.az);
        // Euler-angle rates (quadrotor.py:671-673)
        const T rcth = Real<T>::rcp(cth);
        const T w = fma_t(sph, Q, cph * R);               // sin(phi) q + cos(phi) r
        d[6] = fma_t(sth * rcth, w, P);
        d[7] = fma_t(cph, Q, -(sph * R));
        d[8] = w * rcth;
        // body rates: Jinv (Mb - w x J w)  (quadrotor.py:670)
        d[9] = fma_t(-c.cxx, Q * R, in.mx);
        d[10] = fma_t(-c.cyy, R * P, in.my);
        d[11] = fma_t(-c.czz, P * Q, in.mz);
    } else {
        T s, co;
        Real<T>::sincos(x[2], s, co);
        const T thd = x[3];
        const T tmp = fma_t(c.cp_ml * thd * thd, s, in.F) * c.cp_inv_Mm;
        const T den = c.cp_l * fma_t(-c.cp_m * c.cp_inv_Mm, co * co, (T)(4.0 / 3.0));
        const T thdd = fma_t(c.cp_g, s, -(co * tmp)) * Real<T>::rcp(den);
        d[0] = x[1];
        d[1] = fma_t(-c.cp_ml * c.cp_inv_Mm, thdd * co, tmp);
        d[2] = thd;
        d[3] = thdd;
    }
}

// One fixed step; returns the increment (so the float path can add it with its own rounding).
template <int SYS, class T>
__device__ __forceinline__ void integrate(const PhysT<T>& c, uint32_t flags, const Inputs<SYS, T>& in,
                                          const T (&x)[Dim<SYS>::NX], T (&inc)[Dim<SYS>::NX]) {
    constexpr int NX = Dim<SYS>::NX;
    const T h = c.h;
    T k[NX];
    f_eval<SYS, T>(c, in, x, k);
    if (flags & F_EULER) {
#pragma unroll
        for (int j = 0; j < NX; ++j) inc[j] = h * k[j];
        return;
    }
    // RK4, zero-order-hold input (controllers/mpc/mpc_utils.py:93-97)
    T xt[NX], acc[NX];
    const T hh = (T)0.5 * h;
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] = k[j]; xt[j] = fma_t(hh, k[j], x[j]); }
    f_eval<SYS, T>(c, in, xt, k);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] = fma_t((T)2, k[j], acc[j]); xt[j] = fma_t(hh, k[j], x[j]); }
    f_eval<SYS, T>(c, in, xt, k);
#pragma unroll
    for (int j = 0; j < NX; ++j) { acc[j] = fma_t((T)2, k[j], acc[j]); xt[j] = fma_t(h, k[j], x[j]); }
    f_eval<SYS, T>(c, in, xt, k);
    const T h6 = h * (T)(1.0 / 6.0);
#pragma unroll
    for (int j = 0; j < NX; ++j) inc[j] = h6 * (acc[j] + k[j]);
}

// ---- mixed-precision step of the 3-D quadrotor ("attitude" precise mode) ---------------------------------------------
// The open-loop drift of the float32 step is born in the rotational chain (torque -> rate -> attitude) and amplified by the
// thrust direction; the translational chain only integrates it (tests/studies/precise_mode_study.py: float32 positions and
// velocities + float64 attitude and rates hold 2e-6 over 1000 open-loop steps, float32 everything 2e-1).  So: attitude and
// rates (states 6..11) ACCUMULATE in double -- hi + lo float32 planes in memory, RK4 combination in double --, positions and
// velocities (states 0..5) in float.
//
// Which arithmetic has to be double was measured too (tests/studies/precise_mode_study.py, "arithmetic" rows, on the hover and on the
// circle-tracking scenario): what integrates into the drift is a BIASED error -- the cancellation of the thrusts in the torques
// (u0 + u1 - u2 - u3 ...), the products of the rate equations, and the vertical acceleration  c_theta c_phi T/m - g  with T/m and
// g ROUNDED to float32 (3.2e-5 at 1000 steps of circle tracking) --, while the rounding of sin / cos themselves, of their
// products and of the Euler-rate kinematics to float32 averages out (3.3e-6 against 2.9e-6 with everything in double).  Hence
// (default, SCG_ATT_FP64_TRIG = 0): inputs (T/m, torques) and the body-rate derivatives in double; sin / cos of the
// float32-rounded stage angles, the Euler-rate kinematics and the acceleration in float32 (sincos_f of scg_math.h: 22 instead
// of ~40 instructions per pair), with T/m and Fz/m - g entering the vertical channel as double-float pairs (MixedInputs).  SCG_ATT_FP64_TRIG = 1 is round 2's form (12 float64 sin/cos pairs per step).
#ifndef SCG_ATT_FP64_TRIG
#define SCG_ATT_FP64_TRIG 0
#endif
// float32 images of the double inputs the float32 part of f_mixed3d needs, formed once per (sub-)step: the specific thrust and
// the vertical offset  Fz/m - g  as double-float pairs (hi + lo) -- the vertical channel is where a constant rounding error would
// be born (c_theta c_phi T/m - g with T/m ~ g) --, the horizontal offsets as plain floats.
struct MixedInputs {
    float Tm_hi, Tm_lo, az_hi, az_lo, ax, ay;
};
__device__ __forceinline__ MixedInputs mixed_inputs(const Inputs<SCG_QUAD_3D, double>& in) {
    MixedInputs m;
    m.Tm_hi = (float)in.Tm; m.Tm_lo = (float)(in.Tm - (double)m.Tm_hi);
    m.az_hi = (float)in.az; m.az_lo = (float)(in.az - (double)m.az_hi);
    m.ax = (float)in.ax; m.ay = (float)in.ay;
    return m;
}

__device__ __forceinline__ void f_mixed3d(const PhysT<double>& c, const Inputs<SCG_QUAD_3D, double>& in, const MixedInputs& mi,
                                          const float (&xt)[6], const double (&xa)[6], float (&dt)[6], double (&da)[6]) {
    const double P = xa[3], Q = xa[4], R = xa[5];
#if SCG_ATT_FP64_TRIG
    double sph, cph, sth, cth, sps, cps;
    Real<double>::sincos(xa[0], sph, cph);
    Real<double>::sincos(xa[1], sth, cth);
    Real<double>::sincos(xa[2], sps, cps);
    const double sthcph = sth * cph;
    dt[0] = xt[1];
    dt[1] = (float)fma(fma(cps, sthcph, sps * sph), in.Tm, in.ax);
    dt[2] = xt[3];
    dt[3] = (float)fma(fma(sps, sthcph, -(cps * sph)), in.Tm, in.ay);
    dt[4] = xt[5];
    dt[5] = (float)fma(cth * cph, in.Tm, in.az);
    const double rcth = 1.0 / cth;
    const double w = fma(sph, Q, cph * R);
    da[0] = fma(sth * rcth, w, P);
    da[1] = fma(cph, Q, -(sph * R));
    da[2] = w * rcth;
#else
    float sph, cph, sth, cth, sps, cps;
    sincos_f((float)xa[0], sph, cph);
    sincos_f((float)xa[1], sth, cth);
    sincos_f((float)xa[2], sps, cps);
    const float Qf = (float)Q, Rf = (float)R;
    const float sthcph = sth * cph, cthcph = cth * cph;
    dt[0] = xt[1];
    dt[1] = fmaf(fmaf(cps, sthcph, sps * sph), mi.Tm_hi, mi.ax);
    dt[2] = xt[3];
    dt[3] = fmaf(fmaf(sps, sthcph, -(cps * sph)), mi.Tm_hi, mi.ay);
    dt[4] = xt[5];
    // the vertical channel:  c_theta c_phi (Tm_hi + Tm_lo) + (az_hi + az_lo), each FMA rounding a SMALL result once
    dt[5] = fmaf(cthcph, mi.Tm_hi, mi.az_hi) + fmaf(cthcph, mi.Tm_lo, mi.az_lo);
    const float rcth = rcp_f(cth);
    const float w = fmaf(sph, Qf, cph * Rf);
    da[0] = (double)(sth * rcth * w) + P;                 // the rate itself enters in double
    da[1] = (double)fmaf(cph, Qf, -(sph * Rf));
    da[2] = (double)(w * rcth);
#endif
    da[3] = fma(-c.cxx, Q * R, in.mx);
    da[4] = fma(-c.cyy, R * P, in.my);
    da[5] = fma(-c.czz, P * Q, in.mz);
}

__device__ __forceinline__ void integrate_mixed3d(const PhysT<double>& c, uint32_t flags, const Inputs<SCG_QUAD_3D, double>& in,
                                                  float (&xt)[6], double (&xa)[6]) {
    const double h = c.h;
    const float hf = (float)h;
    const MixedInputs mi = mixed_inputs(in);
    float kt[6];
    double ka[6];
    f_mixed3d(c, in, mi, xt, xa, kt, ka);
    if (flags & F_EULER) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { xt[j] = fmaf(hf, kt[j], xt[j]); xa[j] = fma(h, ka[j], xa[j]); }
        return;
    }
    float st[6], at[6];
    double sa[6], aa[6];
    const double hh = 0.5 * h;
    const float hhf = 0.5f * hf;
#pragma unroll
    for (int j = 0; j < 6; ++j) { at[j] = kt[j]; aa[j] = ka[j]; st[j] = fmaf(hhf, kt[j], xt[j]); sa[j] = fma(hh, ka[j], xa[j]); }
    f_mixed3d(c, in, mi, st, sa, kt, ka);
#pragma unroll
    for (int j = 0; j < 6; ++j) { at[j] = fmaf(2.0f, kt[j], at[j]); aa[j] = fma(2.0, ka[j], aa[j]); st[j] = fmaf(hhf, kt[j], xt[j]); sa[j] = fma(hh, ka[j], xa[j]); }
    f_mixed3d(c, in, mi, st, sa, kt, ka);
#pragma unroll
    for (int j = 0; j < 6; ++j) { at[j] = fmaf(2.0f, kt[j], at[j]); aa[j] = fma(2.0, ka[j], aa[j]); st[j] = fmaf(hf, kt[j], xt[j]); sa[j] = fma(h, ka[j], xa[j]); }
    f_mixed3d(c, in, mi, st, sa, kt, ka);
    const double h6 = h * (1.0 / 6.0);
    const float h6f = (float)h6;
#pragma unroll
    for (int j = 0; j < 6; ++j) { xt[j] = fmaf(h6f, at[j] + kt[j], xt[j]); xa[j] = fma(h6, aa[j] + ka[j], xa[j]); }
}

}  // namespace scg
