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
// Instruction budget (the 12 pairs of an RK4 step were 336 of the kernel's 858 VALU instructions per env-step):
//   * the quadrant k = rint(x 2/pi) comes from the magic-number form t = fma(x, 2/pi, 1.5 2^23): kf = t - 1.5 2^23 and
//     the low mantissa bits of t ARE k mod 4 (|x| < 6.5e6) -- no v_rndne / v_cvt;
//   * the quadrant fix-up is integer work on those bits: the sin/cos swap as two v_bfi_b32 under a v_bfe_i32 mask, the
//     signs as XORs of shifted bits ((k + 1) & 2 = (k & 2) ^ ((k & 1) << 1)): 9 instructions instead of 12;
//   * sincos2_f evaluates TWO angles on v_pk_mul_f32 / v_pk_fma_f32 (each half of a packed FMA is an IEEE fma: the
//     results equal sincos_f's bit for bit); sincos_f packs its own sine chain with its cosine chain (same r^2).
#ifndef SCG_FAST_TRIG
#define SCG_FAST_TRIG 0        // 1: hardware v_sin_f32 / v_cos_f32 (tuning experiment; ~1e-6 abs error)
#endif
#ifndef SCG_PACKED_TRIG
#define SCG_PACKED_TRIG 1      // 0: the scalar form of round 2 (A/B arm)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

constexpr float kTwoOverPi = 0.636619772367581343f, kPio2Hi = -1.5707963705062866f, kPio2Lo = 4.371138828673793e-08f;
constexpr float kRintMagic = 12582912.0f;                       // 1.5 * 2^23
constexpr float kS3 = 2.716089966270374e-06f, kS2 = -0.0001983909896807745f, kS1 = 0.008333329111337662f, kS0 = -0.1666666716337204f;
constexpr float kC2 = 2.4417491658823565e-05f, kC1 = -0.0013887188397347927f, kC0 = 0.04166664183139801f;

// (sin r, cos r) of the reduced argument -> (sin x, cos x) for quadrant bits kb (k mod 4 in the two low bits).
// v_bitop3_b32 (gfx950) evaluates any three-input bitwise function: 0xCA = a ? b : c, 0x78 = a ^ (b & c).
__device__ __forceinline__ void quadrant_fix(uint32_t kb, float sr, float cr, float& s, float& c) {
    const uint32_t usr = __float_as_uint(sr), ucr = __float_as_uint(cr);
    const uint32_t odd = (uint32_t)__builtin_amdgcn_sbfe((int32_t)kb, 0u, 1u);          // all ones when k is odd
    const uint32_t ss = __builtin_amdgcn_bitop3_b32(odd, ucr, usr, 0xCA);               // swap sin / cos in odd quadrants
    const uint32_t cc = __builtin_amdgcn_bitop3_b32(odd, usr, ucr, 0xCA);
    s = __uint_as_float(__builtin_amdgcn_bitop3_b32(ss, kb << 30, 0x80000000u, 0x78));          // sin < 0 in quadrants 2, 3
    c = __uint_as_float(__builtin_amdgcn_bitop3_b32(cc, (kb + 1u) << 30, 0x80000000u, 0x78));   // cos < 0 in quadrants 1, 2
}

__device__ __forceinline__ void sincos_f(float x, float& s, float& c) {
#if SCG_FAST_TRIG
    const float rev = x * 0.15915494309189535f;                 // revolutions
    s = __builtin_amdgcn_sinf(rev);
    c = __builtin_amdgcn_cosf(rev);
#elif SCG_PACKED_TRIG == 2
    // scalar arithmetic, integer quadrant logic (A/B arm: no register pairs)
    const float t = fmaf(x, kTwoOverPi, kRintMagic);
    const float kf = t - kRintMagic;
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
    quadrant_fix(__float_as_uint(t), sr, cr, s, c);
#elif SCG_PACKED_TRIG
    const float t = fmaf(x, kTwoOverPi, kRintMagic);
    const float kf = t - kRintMagic;
    float r = fmaf(kf, kPio2Hi, x);
    r = fmaf(kf, kPio2Lo, r);
    const float r2 = r * r;
    // the two Horner chains share r2: (sine, cosine) halves of one packed FMA per step
    f32x2 p = fma2(splat2(r2), f32x2{kS3, kC2}, f32x2{kS2, kC1});
    p = fma2(p, splat2(r2), f32x2{kS1, kC0});
    const float ps = fmaf(p.x, r2, kS0), pc = p.y;
    const f32x2 hi = f32x2{r, r2} * splat2(r2);                 // (r^3, r^4)
    const float h = fmaf(r2, -0.5f, 1.0f);
    const f32x2 v = fma2(hi, f32x2{ps, pc}, f32x2{r, h});       // sin(r) = r + r^3 S(r2);  cos(r) = 1 - r2/2 + r2^2 C(r2)
    quadrant_fix(__float_as_uint(t), v.x, v.y, s, c);
#else
    const float kf = rintf(x * 0.636619772367581343f);          // x * 2/pi
    const int k = (int)kf;
    float r = fmaf(kf, -1.5707963705062866f, x);                // fl32(pi/2)
    r = fmaf(kf, 4.371138828673793e-08f, r);                    // pi/2 - fl32(pi/2)
    const float r2 = r * r;
    // sin(r) = r + r^3 S(r2)
    float ps = fmaf(r2, 2.716089966270374e-06f, -0.0001983909896807745f);
    ps = fmaf(ps, r2, 0.008333329111337662f);
    ps = fmaf(ps, r2, -0.1666666716337204f);
    const float sr = fmaf(r * r2, ps, r);
    // cos(r) = 1 - r2/2 + r2^2 C(r2)
    float pc = fmaf(r2, 2.4417491658823565e-05f, -0.0013887188397347927f);
    pc = fmaf(pc, r2, 0.04166664183139801f);
    const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    const float ss = (k & 1) ? cr : sr;
    const float cc = (k & 1) ? sr : cr;
    s = (k & 2) ? -ss : ss;
    c = ((k + 1) & 2) ? -cc : cc;
#endif
}

// Two independent angles at once (roll with pitch in the 3-D model): every multiply / FMA is packed.
__device__ __forceinline__ void sincos2_f(float xa, float xb, float& sa, float& ca, float& sb, float& cb) {
#if SCG_FAST_TRIG || SCG_PACKED_TRIG != 1
    sincos_f(xa, sa, ca);
    sincos_f(xb, sb, cb);
#else
    const f32x2 x = {xa, xb};
    const f32x2 t = fma2(x, splat2(kTwoOverPi), splat2(kRintMagic));
    const f32x2 kf = t - splat2(kRintMagic);
    f32x2 r = fma2(kf, splat2(kPio2Hi), x);
    r = fma2(kf, splat2(kPio2Lo), r);
    const f32x2 r2 = r * r;
    f32x2 ps = fma2(r2, splat2(kS3), splat2(kS2));
    ps = fma2(ps, r2, splat2(kS1));
    ps = fma2(ps, r2, splat2(kS0));
    const f32x2 sr = fma2(r * r2, ps, r);
    f32x2 pc = fma2(r2, splat2(kC2), splat2(kC1));
    pc = fma2(pc, r2, splat2(kC0));
    const f32x2 cr = fma2(r2 * r2, pc, fma2(r2, splat2(-0.5f), splat2(1.0f)));
    quadrant_fix(__float_as_uint(t.x), sr.x, cr.x, sa, ca);
    quadrant_fix(__float_as_uint(t.y), sr.y, cr.y, sb, cb);
#endif
}

}  // namespace scg
