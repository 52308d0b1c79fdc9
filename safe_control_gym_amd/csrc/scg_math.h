// Device-side helpers shared by the kernels of the rollout engine: system dimensions, 1-ulp hardware
// reciprocal / sqrt, SGPR-based plane addressing and the fp32 sin/cos pair.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "scg_device.h"

namespace scg {

#ifndef SCG_NT
#define SCG_NT 1               // nontemporal hint on the streamed planes: 0 none, 1 loads+stores (measured -5%), 2 stores only, 3 loads only
#endif

template <int SYS> struct Dim;
template <> struct Dim<SCG_QUAD_1D> { static constexpr int NX = 2, NU = 1; };
template <> struct Dim<SCG_QUAD_2D> { static constexpr int NX = 6, NU = 2; };
template <> struct Dim<SCG_QUAD_3D> { static constexpr int NX = 12, NU = 4; };
template <> struct Dim<SCG_CARTPOLE> { static constexpr int NX = 4, NU = 1; };

// ------------------------------------------------------------------------------------------
// math helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32, 1 ulp
__device__ __forceinline__ float sqrt_f(float x) { return __builtin_amdgcn_sqrtf(x); }    // v_sqrt_f32, 1 ulp
// clamp as ONE v_med3_f32 (fminf(fmaxf()) costs two extra canonicalising v_max_f32 per call)
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// make a wave-uniform 64-bit value opaque to CSE: the expression that uses it is re-evaluated on the
// scalar unit at each use instead of being kept alive (and spilled) across the ~1200-instruction body
__device__ __forceinline__ int64_t opaque_s(int64_t v) { asm volatile("" : "+s"(v)); return v; }

// Plane access with a wave-uniform base and a 32-bit per-lane BYTE offset: lowers to the
// `global_load_dword v, v_off, s[base:base+1]` (saddr) form -- one VGPR of address for every plane
// instead of a 64-bit VGPR pair per plane.  n <= 2^30 (checked on the host) keeps 4*i in 32 bits.
template <class T>
__device__ __forceinline__ T* uptr(T* p) {       // pin a wave-uniform pointer into SGPRs
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    typedef __attribute__((address_space(1))) T* global_ptr;   // keep it a global (not flat) pointer
    return (T*)(global_ptr)(((uint64_t)hi << 32) | lo);
}
template <class T>
__device__ __forceinline__ T ldp(const T* base, uint32_t boff) {
#if SCG_NT == 1 || SCG_NT == 3
    return __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + boff));
#else
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + boff);
#endif
}
template <class T>
__device__ __forceinline__ void stp(T* base, uint32_t boff, T v) {
#if SCG_NT == 1 || SCG_NT == 2
    __builtin_nontemporal_store(v, reinterpret_cast<T*>(reinterpret_cast<char*>(base) + boff));
#else
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + boff) = v;
#endif
}

// sin and cos together.  Cody-Waite reduction by pi/2 (two fp32 pieces, each applied with one
// exact-product FMA; the neglected third piece is 1.7e-15 per quadrant) followed by minimax
// polynomials on [-pi/4, pi/4] fitted for this kernel (max abs error 9.3e-8 = 0.8 ulp(1) for
// |x| <= 1000, measured against float64 sin/cos over 2e6 points: DESIGN.md section 5).
//
// 22 VALU instructions per pair (round 2: 28; the 12 pairs of an RK4 step were 336 of the kernel's 858 per env-step):
//   * the quadrant k = rint(x 2/pi) comes from the magic-number form t = fma(x, 2/pi, 1.5 2^23): kf = t - 1.5 2^23, and the
//     low mantissa bits of t ARE k mod 4 (|x| < 6.5e6) -- no v_rndne / v_cvt;
//   * the quadrant fix-up is integer work on those bits with v_bitop3_b32 (gfx950: any three-input bitwise function, here
//     0xCA = a ? b : c and 0x78 = a ^ (b & c)): the sin / cos swap under a v_bfe_i32 mask, the two signs from shifted bits --
//     7 instructions instead of 12.
// A two-angles-per-instruction form on v_pk_fma_f32 (and one with the sine and cosine chains of one angle packed) was built and
// measured: 653 instead of 716 instructions per env-step, but 9-11 more VGPRs and no faster anywhere.
#ifndef SCG_FAST_TRIG
#define SCG_FAST_TRIG 0        // 1: hardware v_sin_f32 / v_cos_f32 (tuning experiment; ~1e-6 abs error)
#endif
constexpr float kTwoOverPi = 0.636619772367581343f, kPio2Hi = -1.5707963705062866f, kPio2Lo = 4.371138828673793e-08f;
constexpr float kRintMagic = 12582912.0f;                       // 1.5 * 2^23
constexpr float kS3 = 2.716089966270374e-06f, kS2 = -0.0001983909896807745f, kS1 = 0.008333329111337662f, kS0 = -0.1666666716337204f;
constexpr float kC2 = 2.4417491658823565e-05f, kC1 = -0.0013887188397347927f, kC0 = 0.04166664183139801f;

// (sin r, cos r) of the reduced argument -> (sin x, cos x) for quadrant bits kb (k mod 4 in the two low bits)
__device__ __forceinline__ void quadrant_fix(uint32_t kb, float sr, float cr, float& s, float& c) {
    const uint32_t usr = __float_as_uint(sr), ucr = __float_as_uint(cr);
    const uint32_t odd = (uint32_t)__builtin_amdgcn_sbfe((int32_t)kb, 0u, 1u);          // all ones when k is odd
    const uint32_t ss = __builtin_amdgcn_bitop3_b32(odd, ucr, usr, 0xCA);               // swap sin / cos in odd quadrants
    const uint32_t cc = __builtin_amdgcn_bitop3_b32(odd, usr, ucr, 0xCA);
    s = __uint_as_float(__builtin_amdgcn_bitop3_b32(ss, kb << 30, 0x80000000u, 0x78));          // sin < 0 in quadrants 2, 3
    c = __uint_as_float(__builtin_amdgcn_bitop3_b32(cc, (kb + 1u) << 30, 0x80000000u, 0x78));   // cos < 0 in quadrants 1, 2
}

#ifndef SCG_TRIG_R02
#define SCG_TRIG_R02 0         // 1 (bisect arm of profiles/r04): round 2's quadrant logic (rintf + v_cvt + compares / selects), same polynomials
#endif
#if defined(__FAST_MATH__) || defined(__ASSOCIATIVE_MATH__)
#error "scg_math.h: sincos_f reads the quadrant off the bits of t = fma(x, 2/pi, 1.5 2^23) and forms k as t - 1.5 2^23: value-unsafe math flags (-ffast-math, -fassociative-math) fold that pair and every trig value comes out wrong"
#endif
// Valid for |x| < 2^22 / (2/pi) ~ 6.5e6 (beyond that 1.5 2^23 + x 2/pi leaves the binade whose ulp is 1): angles, rates times a
// time step and 2 pi u stay far inside; a yaw that winds up without bound (done_on_out_of_bound=False) reaches it after ~1e6 turns.
__device__ __forceinline__ void sincos_f(float x, float& s, float& c) {
#if SCG_FAST_TRIG
    const float rev = x * 0.15915494309189535f;                 // revolutions
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
#elif SCG_TRIG_R02
    const float kf = rintf(x * kTwoOverPi);
    const int k = (int)kf;
    float r = fmaf(kf, kPio2Hi, x);
    r = fmaf(kf, kPio2Lo, r);
    const float r2 = r * r;
    float ps = fmaf(r2, kS3, kS2);
    ps = fmaf(ps, r2, kS1);
    ps = fmaf(ps, r2, kS0);
    const float sr = fmaf(r * r2, ps, r);
    float pc = fmaf(r2, kC2, kC1);
    pc = fmaf(pc, r2, kC0);
    const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    const float ss = (k & 1) ? cr : sr;
    const float cc = (k & 1) ? sr : cr;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
#else
    float t = fmaf(x, kTwoOverPi, kRintMagic);
    asm("" : "+v"(t));                                          // opaque to the optimiser (not volatile: free to schedule): the add here and the
                                                                // subtract below stay two roundings, whatever the flags
    const float kf = t - kRintMagic;
    float r = fmaf(kf, kPio2Hi, x);                             // fl32(pi/2)
    r = fmaf(kf, kPio2Lo, r);                                   // pi/2 - fl32(pi/2)
    const float r2 = r * r;
    // sin(r) = r + r^3 S(r2)
    float ps = fmaf(r2, kS3, kS2);
    ps = fmaf(ps, r2, kS1);
    ps = fmaf(ps, r2, kS0);
    const float sr = fmaf(r * r2, ps, r);
    // cos(r) = 1 - r2/2 + r2^2 C(r2)
    float pc = fmaf(r2, kC2, kC1);
    pc = fmaf(pc, r2, kC0);
    const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    quadrant_fix(__float_as_uint(t), sr, cr, s, c);
#endif
}

}  // namespace scg
