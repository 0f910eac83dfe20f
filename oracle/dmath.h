// TEST INFRASTRUCTURE — CPU oracle copy of the deterministic f32 math contract.
//
// The reference shader calls WGSL builtins (sin, cos, atan2, asin, acos, log, pow) whose results
// are only bounded in ULPs by the WGSL spec, so "the reference's value" is not a single bit
// pattern.  This contract fixes ONE conforming implementation of each builtin, built only from
// IEEE-754 binary32 add/sub/mul/div/sqrt/fma, round-to-nearest-even conversions and bit
// manipulation.  Every one of those operations is exactly defined, so the same source gives the
// same bits on x86-64 and on gfx950 as long as the compiler neither contracts (-ffp-contract=off)
// nor reassociates (no fast-math).  Polynomial coefficients are the public Cephes single-precision
// ones (Moshier, netlib cephes/single: sinf.c, atanf.c, asinf.c, logf.c, exp2f.c).
//
// Reference call sites these replace:
//   sin/cos   src/shaders/wgsl/utils.wgsl:22, materials.wgsl:46, lighting.wgsl:14-15,
//             figures/mandelbulb.wgsl:19, figures/mebius.wgsl:13-22
//   atan2     figures/sphere.wgsl:35, cylinder.wgsl:41, cone.wgsl:55, torus.wgsl:64-66,
//             mandelbulb.wgsl:17, mebius.wgsl:10,75, skybox.wgsl:15, ellipsoid.wgsl:47 ...
//   asin      figures/sphere.wgsl:36, ellipsoid.wgsl:48, skybox.wgsl:16
//   acos      figures/mandelbulb.wgsl:16
//   log       figures/mandelbulb.wgsl:29, julia.wgsl:31
//   pow       figures/mandelbulb.wgsl:12,18, tonemap.wgsl:21
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace dm {

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// IEEE minNum/maxNum with -0 < +0 (what v_min_f32 / v_max_f32 return on gfx950).
static inline float fmin_(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? a : b;   // prefers -0
    return a < b ? a : b;
}
static inline float fmax_(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return (f2u(a) & 0x80000000u) ? b : a;   // prefers +0
    return a > b ? a : b;
}

// Division.  WGSL specifies x / y to 2.5 ULP for |y| in [2^-126, 2^126] and lets implementations flush
// subnormal values; the contract is
//   rcp_(y)    = the correctly rounded 1/y, with a subnormal y treated as a zero of its sign (result: infinity)
//                and a subnormal result replaced by a zero of its sign;
//   div_(x, y) = x * rcp_(y)   (at most 1.5 ULP; a literal divisor C is written x * (1.0f / C), the same value).
// On gfx950 rcp_ is the hardware estimate plus one fused Newton step (checked there against this definition
// over all 2^32 inputs); sqrt stays the correctly rounded IEEE square root.
static inline float rcp_(float y) {
    if (y != y) return y + y;
    uint32_t sign = f2u(y) & 0x80000000u;
    if (std::fabs(y) < 1.17549435e-38f) return u2f(sign | 0x7f800000u);
    float r = 1.0f / y;
    if (std::fabs(r) < 1.17549435e-38f) return u2f(sign);
    return r;
}
static inline float div_(float x, float y) { return x * rcp_(y); }

static const float PIO2 = 1.57079632679489661923f;
static const float PI_F = 3.14159265358979323846f;
static const float PIO4 = 0.78539816339744830962f;

// sin and cos together.  Cody-Waite reduction by pi/2 in three parts, Cephes minimax polynomials
// on [-pi/4, pi/4].  Intended domain |x| <= 1e4.
static inline void sincos_(float x, float* s_out, float* c_out) {
    float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    int q = (int)k & 3;
    float z = r * r;
    float sp = fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
    float s = fmaf(r * z, sp, r);
    float cp = fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
    float c = fmaf(z * z, cp, fmaf(-0.5f, z, 1.0f));
    float so = (q & 1) ? c : s;
    float co = (q & 1) ? s : c;
    if (q & 2) so = -so;
    if ((q + 1) & 2) co = -co;
    *s_out = so;
    *c_out = co;
}
static inline float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
static inline float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }

// atan2(y, x); atan2(0, 0) = 0.
static inline float atan2_(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmax_(ax, ay), mn = fmin_(ax, ay);
    if (mx == 0.0f) return 0.0f;
    bool big = mn > 0.4142135623730950f * mx;
    float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    float t = div_(num, den);
    float z = t * t;
    float p = fmaf(z, fmaf(z, fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f), 1.99777106478e-1f), -3.33329491539e-1f);
    float r = fmaf(p * z, t, t);
    if (big) r = PIO4 + r;
    if (ay > ax) r = PIO2 - r;
    if (x < 0.0f) r = PI_F - r;
    if (y < 0.0f) r = -r;
    return r;
}

static inline float asin_poly(float z) {
    return fmaf(z, fmaf(z, fmaf(z, fmaf(z, 4.2163199048e-2f, 2.4181311049e-2f), 4.5470025998e-2f), 7.4953002686e-2f), 1.6666752422e-1f);
}
static inline float asin_(float x) {
    float a = fabsf(x);
    bool big = a > 0.5f;
    float z = big ? 0.5f * (1.0f - a) : a * a;
    float s = big ? sqrtf(z) : a;
    float r = fmaf(s * z, asin_poly(z), s);
    if (big) r = fmaf(-2.0f, r, PIO2);
    return x < 0.0f ? -r : r;
}
static inline float acos_(float x) {
    float a = fabsf(x);
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = sqrtf(z);
        float r = fmaf(s * z, asin_poly(z), s);
        r = r + r;
        return x < 0.0f ? PI_F - r : r;
    }
    float z = x * x;
    float r = fmaf(x * z, asin_poly(z), x);
    return PIO2 - r;
}

// Shared core of log/log2: x > 0 finite.  Returns e (as float) and ln(m'), x = 2^e * m',
// m' in [sqrt(1/2), sqrt(2)).
static inline void log_core(float x, float* fe_out, float* lnm_out) {
    int eadj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; eadj = -23; }
    uint32_t b = f2u(x);
    int e = (int)((b >> 23) & 0xffu) - 126 + eadj;
    float m = u2f((b & 0x007fffffu) | 0x3f000000u);      // [0.5, 1)
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = fmaf(m, 7.0376836292e-2f, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    y = fmaf(-0.5f, z, y);
    *fe_out = (float)e;
    *lnm_out = m + y;
}
static inline float log_(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -INFINITY : NAN;
    if (x == INFINITY) return x;
    float fe, lnm;
    log_core(x, &fe, &lnm);
    float r = fmaf(fe, -2.12194440e-4f, lnm);
    return fmaf(fe, 0.693359375f, r);
}
static inline float log2_(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -INFINITY : NAN;
    if (x == INFINITY) return x;
    float fe, lnm;
    log_core(x, &fe, &lnm);
    return fmaf(lnm, 1.44269504088896341f, fe);
}
static inline float exp2_(float x) {
    if (x != x) return x;
    if (x > 128.0f) return INFINITY;
    if (x < -150.0f) return 0.0f;
    float k = rintf(x);
    float f = x - k;
    float p = fmaf(f, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = fmaf(p, f, 9.618437357674640e-3f);
    p = fmaf(p, f, 5.550332471162809e-2f);
    p = fmaf(p, f, 2.402264791363012e-1f);
    p = fmaf(p, f, 6.931472028550421e-1f);
    p = fmaf(p, f, 1.0f);
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;
    float s1 = u2f((uint32_t)(k1 + 127) << 23);
    float s2 = u2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}
static inline float pow_(float x, float y) { return exp2_(y * log2_(x)); }

}  // namespace dm
