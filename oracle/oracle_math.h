// oracle_math.h — TEST INFRASTRUCTURE (parity oracle), not product code.
//
// Scalar f32 helpers of the CPU restatement.  Everything here is a fixed
// sequence of IEEE-754 binary32 operations (+ - * / sqrt, compares, integer
// bit moves): compiled with -ffp-contract=off and no fast-math it gives the
// same bits on any conforming machine, which is what lets the GPU parity
// tests demand bit-equal geometry.
//
// Transcendentals: the reference calls Rust's f32::{sin,cos,acos,atan2}
// (reference src/raytracing.rs:610,619,620,862; src/shape/sphere.rs:92,95),
// i.e. the platform libm, whose last-bit behaviour is not specified.  The
// oracle pins them to the classic single-precision Cephes algorithms
// (Moshier, public domain: sinf.c, asinf.c, atanf.c), restated below.  They
// are within 2 ulp of the correctly rounded result on the ranges the trace
// loop uses (tests/test_oracle_math.py measures this against float64).
// powf (specular exponent, gamma) only scales colours and stays libm.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rro {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Rust f32::max / f32::min return the non-NaN operand.
static inline float rs_max(float a, float b) { return (a > b || b != b) ? a : b; }
static inline float rs_min(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float rs_abs(float a) { return u2f(f2u(a) & 0x7fffffffu); }

// Rust `as i32` / `as u32` / `as u8` from f32: truncate toward zero,
// saturate, NaN -> 0.
static inline int32_t as_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
static inline uint32_t as_u32(float f) {
    if (f != f) return 0;
    if (f >= 4294967296.0f) return UINT32_MAX;
    if (f <= 0.0f) return 0;
    return (uint32_t)f;
}
static inline uint8_t as_u8(float f) {
    if (f != f) return 0;
    if (f >= 255.0f) return 255;
    if (f <= 0.0f) return 0;
    return (uint8_t)f;
}

static const float RR_PI = 3.14159265358979323846f; // std::f32::consts::PI

// ---- sin / cos: Cephes sinf.c / cosf.c -----------------------------------
// Octant reduction by pi/4 with a three-part Cody-Waite constant, then the
// degree-7 / degree-8 minimax polynomials.  Valid (|err| <= 2 ulp) for
// |x| <= 8192; beyond that the reduction loses bits but stays deterministic.
static inline void sincos_f32(float xin, float* s_out, float* c_out) {
    const float FOPI = 1.27323954473516f; // 4/pi
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    float x = rs_abs(xin);
    int sign_s = (f2u(xin) >> 31) ? -1 : 1;
    int sign_c = 1;
    uint32_t j = (uint32_t)as_i32(FOPI * x); // integer part of x/(pi/4)
    float y = (float)j;
    if (j & 1u) { j += 1u; y += 1.0f; }     // map zeros to origin
    j &= 7u;
    if (j > 3u) { sign_s = -sign_s; sign_c = -sign_c; j -= 4u; }
    if (j > 1u) sign_c = -sign_c;
    // extended precision modular arithmetic
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    // sin polynomial on [-pi/4, pi/4]
    float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
    // cos polynomial on [-pi/4, pi/4]
    float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z
               - 0.5f * z + 1.0f;
    float s, c;
    if (j == 1u || j == 2u) { s = pc; c = ps; } else { s = ps; c = pc; }
    *s_out = (sign_s < 0) ? -s : s;
    *c_out = (sign_c < 0) ? -c : c;
}
static inline float sin_f32(float x) { float s, c; sincos_f32(x, &s, &c); return s; }
static inline float cos_f32(float x) { float s, c; sincos_f32(x, &s, &c); return c; }

// ---- asin / acos: Cephes asinf.c ------------------------------------------
static inline float asin_f32(float xx) {
    float x = xx;
    int neg = 0;
    if (x < 0.0f) { neg = 1; x = -x; }
    if (x > 1.0f || x != x) return u2f(0x7fc00000u); // NaN, as f32::asin
    float z;
    int flag = 0;
    if (x > 0.5f) {
        z = 0.5f * (1.0f - x);
        x = std::sqrt(z);
        flag = 1;
    } else {
        z = x * x;
    }
    float p = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z
                + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * x + x;
    if (flag) { p = p + p; p = 1.5707963267948966192f - p; }
    return neg ? -p : p;
}
static inline float acos_f32(float x) {
    if (x != x || x > 1.0f || x < -1.0f) return u2f(0x7fc00000u);
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin_f32(std::sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_f32(std::sqrt(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - asin_f32(x);
}

// ---- atan / atan2: Cephes atanf.c -----------------------------------------
static inline float atan_f32(float xx) {
    float x = xx;
    int neg = 0;
    if (x < 0.0f) { neg = 1; x = -x; }
    float y;
    if (x > 2.414213562373095f) { // tan(3pi/8)
        y = 1.5707963267948966192f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { // tan(pi/8)
        y = 0.7853981633974483096f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z
          - 3.33329491539E-1f) * z * x + x;
    return neg ? -y : y;
}
// f32::atan2(y, x) quadrant rules (IEEE / C99 atan2f special cases for 0 and NaN;
// infinities do not occur on the trace path and return the finite-limit values).
static inline float atan2_f32(float y, float x) {
    const float PI = 3.14159265358979323846f;
    const float PIO2 = 1.5707963267948966192f;
    if (x != x || y != y) return u2f(0x7fc00000u);
    bool yneg = (f2u(y) >> 31) != 0;
    bool xneg = (f2u(x) >> 31) != 0;
    if (y == 0.0f) {
        if (!xneg) return y;               // +-0 for x >= +0
        return yneg ? -PI : PI;            // +-pi for x <= -0
    }
    if (x == 0.0f) return yneg ? -PIO2 : PIO2;
    float z = atan_f32(y / x);
    if (xneg) return yneg ? (z - PI) : (z + PI);
    return z;
}

} // namespace rro
