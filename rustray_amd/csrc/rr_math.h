// rr_math.h — f32 arithmetic of the device trace loop (gfx950).
//
// Numeric contract of this library (DESIGN.md "Numerics"): everything that
// decides WHICH surface a ray hits — ray generation, inverse-ray transform,
// box / triangle / ball tests, hit points, normals, jitter directions,
// reflection / refraction rays — is evaluated as the plain IEEE-754 binary32
// sequence the reference's Rust code performs: no FMA contraction
// (-ffp-contract=off), correctly rounded divide and sqrt (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), no fast-math, denormals kept.
// Evaluation order follows nalgebra 0.32: dot = (x*x' + y*y') + z*z',
// normalize = component / norm, mat*vec = column axpy.
//
// sin / cos / acos / atan2 are fixed polynomial sequences (single-precision
// Cephes algorithms) so that a host with any libm reproduces them bit for bit;
// powf (specular exponent, gamma) only scales colours and uses the device libm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RR_DEV __device__ __forceinline__
// also callable from the host side of rr_api.hip (per-material constants that the kernels would otherwise re-derive per hit);
// host and device evaluate the same IEEE binary32 sequence (-ffp-contract=off on both sides): tests/test_gpu_math.py compares them bit for bit
#define RR_HD __host__ __device__ __forceinline__
#define RR_FLT_MAX 3.40282347e+38f
#define RR_PI_F 3.14159265358979323846f

struct f3 { float x, y, z; };
struct f2 { float x, y; };

RR_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RR_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RR_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RR_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RR_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RR_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RR_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RR_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RR_DEV f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RR_DEV float norm3(f3 a) { return sqrtf(dot3(a, a)); }
RR_DEV f3 normalize3(f3 a) { return a / norm3(a); }

RR_HD uint32_t f_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
RR_HD float bits_f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
RR_HD float rr_abs(float a) { return bits_f(f_bits(a) & 0x7fffffffu); }
// Rust f32::max / min: the non-NaN operand wins
RR_DEV float rs_max(float a, float b) { return (a > b || b != b) ? a : b; }
RR_DEV float rs_min(float a, float b) { return (a < b || b != b) ? a : b; }

// Rust `as` casts from f32: truncate, saturate, NaN -> 0
RR_HD int32_t as_i32(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)f;
}
RR_DEV uint32_t as_u32(float f) {
    if (f != f) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    if (f <= 0.0f) return 0u;
    return (uint32_t)f;
}
RR_DEV uint32_t as_u8(float f) {
    if (f != f) return 0u;
    if (f >= 255.0f) return 255u;
    if (f <= 0.0f) return 0u;
    return (uint32_t)f;
}

// helper::approx_equal, reference src/helper.rs:11-20
RR_DEV bool approx_equal(float a, float b) {
    const float factor = 1000000.0f;
    return truncf(a * factor) == truncf(b * factor);
}

// One row of a column-major 4x4 times (x, y, z, w): ((m0*x + m1*y) + m2*z) + m3*w
RR_DEV float row4(float4 r, float x, float y, float z, float w) {
    return ((r.x * x + r.y * y) + r.z * z) + r.w * w;
}

// ---- sin / cos (Cephes sinf.c / cosf.c), |x| <= 8192 within 2 ulp -----------------
RR_HD void rr_sincos(float xin, float* s_out, float* c_out) {
    const float FOPI = 1.27323954473516f;
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;
    float x = rr_abs(xin);
    bool neg_s = (f_bits(xin) >> 31) != 0u;
    bool neg_c = false;
    uint32_t j = (uint32_t)as_i32(FOPI * x);
    float y = (float)j;
    if (j & 1u) { j += 1u; y += 1.0f; }
    j &= 7u;
    if (j > 3u) { neg_s = !neg_s; neg_c = !neg_c; j -= 4u; }
    if (j > 1u) neg_c = !neg_c;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
    float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z
               - 0.5f * z + 1.0f;
    bool swap = (j == 1u || j == 2u);
    float s = swap ? pc : ps;
    float c = swap ? ps : pc;
    *s_out = neg_s ? -s : s;
    *c_out = neg_c ? -c : c;
}
RR_HD float rr_cos(float x) { float s, c; rr_sincos(x, &s, &c); return c; }

// ---- asin / acos (Cephes asinf.c) ------------------------------------------------------
RR_DEV float rr_asin(float xx) {
    float x = xx;
    bool neg = false;
    if (x < 0.0f) { neg = true; x = -x; }
    if (x > 1.0f || x != x) return bits_f(0x7fc00000u);
    float z;
    bool flag = false;
    if (x > 0.5f) {
        z = 0.5f * (1.0f - x);
        x = sqrtf(z);
        flag = true;
    } else {
        z = x * x;
    }
    float p = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z
                + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * x + x;
    if (flag) { p = p + p; p = 1.5707963267948966192f - p; }
    return neg ? -p : p;
}
RR_DEV float rr_acos(float x) {
    if (x != x || x > 1.0f || x < -1.0f) return bits_f(0x7fc00000u);
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * rr_asin(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * rr_asin(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - rr_asin(x);
}

// ---- atan / atan2 (Cephes atanf.c) --------------------------------------------------------
RR_DEV float rr_atan(float xx) {
    float x = xx;
    bool neg = false;
    if (x < 0.0f) { neg = true; x = -x; }
    float y;
    if (x > 2.414213562373095f) {
        y = 1.5707963267948966192f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
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
RR_DEV float rr_atan2(float y, float x) {
    const float PI = 3.14159265358979323846f;
    const float PIO2 = 1.5707963267948966192f;
    if (x != x || y != y) return bits_f(0x7fc00000u);
    bool yneg = (f_bits(y) >> 31) != 0u;
    bool xneg = (f_bits(x) >> 31) != 0u;
    if (y == 0.0f) {
        if (!xneg) return y;
        return yneg ? -PI : PI;
    }
    if (x == 0.0f) return yneg ? -PIO2 : PIO2;
    float z = rr_atan(y / x);
    if (xneg) return yneg ? (z - PI) : (z + PI);
    return z;
}

// ---- Philox4x32-10: the counter-based generator behind jitter() ----------------------------
RR_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t* r0, uint32_t* r1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    *r0 = c0; *r1 = c1;
}

// rand 0.8 UniformFloat<f32>::sample_single mapping (23 mantissa bits), clamped below `high`
RR_DEV float uniform_f32(uint32_t bits, float low, float high) {
    float scale = high - low;
    float value0_1 = bits_f((bits >> 9) | 0x3f800000u) - 1.0f;
    float res = value0_1 * scale + low;
    if (!(res < high)) res = bits_f(f_bits(high) - 1u);
    return res;
}
