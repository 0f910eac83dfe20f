// Deterministic binary32 math for the gfx950 kernels.
//
// The reference shader's sin/cos/atan2/asin/acos/log/pow are WGSL builtins with ULP-bounded,
// implementation-defined results.  The kernels use one fixed conforming implementation built only from
// IEEE add/sub/mul/div/sqrt/fma, round-to-nearest conversions and bit operations, so that results are
// reproducible bit for bit on any IEEE machine (Cephes single-precision polynomials: sinf.c, atanf.c,
// asinf.c, logf.c, exp2f.c).  Compile with -ffp-contract=off: the only fused operations are the
// explicit __builtin_fmaf calls.  hipcc's default correctly-rounded f32 divide/sqrt is required.
//
// Reference call sites: src/shaders/wgsl/utils.wgsl:22, materials.wgsl:46, lighting.wgsl:14-15,
// figures/{sphere,cylinder,cone,torus,ellipsoid,paraboloid,hyperboloid,mebius,mandelbulb,julia,skybox}.wgsl,
// tonemap.wgsl:21.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV static __device__ __forceinline__

// Diagnostic builds (-DPTX_SITE_STATS): per code site, how often a wave ran it and with how many lanes.
#ifdef PTX_SITE_STATS
#define PT_NSITES 96
__device__ unsigned long long pt_sites[PT_NSITES * 2];
static __device__ __noinline__ void pt_site(int i) {
    unsigned long long m = __ballot(1);
    if (i < 0 || i >= PT_NSITES) return;
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) {
        atomicAdd(&pt_sites[2 * i], 1ull);
        atomicAdd(&pt_sites[2 * i + 1], (unsigned long long)__popcll(m));
    }
}
static __device__ __noinline__ void pt_site_typed(int base, uint32_t type) {
    if (base < 0 || base + 18 > PT_NSITES) return;
    for (uint32_t T = 0; T < 18u; T++) {
        unsigned long long m = __ballot((type > 16u ? 17u : type) == T);
        if (m != 0ull && (int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) {
            atomicAdd(&pt_sites[2 * (base + T)], 1ull);
            atomicAdd(&pt_sites[2 * (base + T) + 1], (unsigned long long)__popcll(m));
        }
    }
}
static __device__ __noinline__ void pt_site_hist(int base, uint32_t n) {  // histogram of a wave-uniform count 0..64, 8 buckets (0 counts with 1-8)
    unsigned long long m = __ballot(1);
    if (base < 0 || base + 8 > PT_NSITES) return;
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) {
        uint32_t k = n == 0u ? 0u : (n - 1u) >> 3;
        int b = base + (int)(k > 7u ? 7u : k);
        atomicAdd(&pt_sites[2 * b], 1ull);
        atomicAdd(&pt_sites[2 * b + 1], (unsigned long long)n);
    }
}
static __device__ __noinline__ void pt_site_add(int i, uint32_t n) {  // mean of a wave-uniform count
    unsigned long long m = __ballot(1);
    if (i < 0 || i >= PT_NSITES) return;
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) {
        atomicAdd(&pt_sites[2 * i], 1ull);
        atomicAdd(&pt_sites[2 * i + 1], (unsigned long long)n);
    }
}
#define PT_SITE(i) pt_site(i)
#define PT_SITE_HIST(b, n) pt_site_hist(b, n)
#define PT_SITE_ADD(i, n) pt_site_add(i, n)
#define PT_SITE_TYPED(b, t) pt_site_typed(b, t)
#else
#define PT_SITE(i)
#define PT_SITE_HIST(b, n)
#define PT_SITE_ADD(i, n)
#define PT_SITE_TYPED(b, t)
#endif

namespace dm {

PT_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
PT_DEV float u2f(uint32_t u) { return __uint_as_float(u); }
PT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// v_min_f32 / v_max_f32: IEEE minNum/maxNum, -0 < +0.
PT_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
PT_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
// Division contract: x / y = x * rcp_(y), where rcp_ is the correctly rounded reciprocal that treats a
// subnormal y as a zero of its sign and replaces a subnormal result by a zero of its sign (WGSL bounds x / y
// to 2.5 ULP for |y| in [2^-126, 2^126] and allows flushing).  That is exactly what the hardware estimate
// (1 ulp, flushing) plus one fused Newton step plus v_div_fixup gives: 4 instructions instead of the 11 of
// an IEEE divide.  ptx_kat_exhaustive (ptx355.hip) checks it against the definition over all 2^32 inputs.
PT_DEV float rcp_(float y) {
#ifdef PT_HW_APPROX  // experiment only: the hardware's 1-ulp estimates, as a shader compiler emits them for WGSL's division and sqrt
    return __builtin_amdgcn_rcpf(y);
#endif
    float r = __builtin_amdgcn_rcpf(y);
    float e = __builtin_fmaf(-y, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return __builtin_amdgcn_div_fixupf(r, y, 1.0f);
}
PT_DEV float div_(float x, float y) { return x * rcp_(y); }
// Correctly rounded IEEE square root.  For 2^-102 <= x < inf the reciprocal-square-root estimate plus one
// fused residual step is correctly rounded (same exhaustive check); the rest (zeros, subnormal and tiny
// inputs whose residual would underflow, infinity, NaN, negative inputs) takes the compiler's long sequence.
PT_DEV float sqrt_(float x) {
#ifdef PT_HW_APPROX
    return __builtin_amdgcn_sqrtf(x);
#endif
    if (__builtin_expect((__float_as_uint(x) - 0x0C800000u) >= (0x7F800000u - 0x0C800000u), 0)) return __builtin_sqrtf(x);
    float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
PT_DEV float abs_(float a) { return __builtin_fabsf(a); }

#define DM_PIO2 1.57079632679489661923f
#define DM_PI 3.14159265358979323846f
#define DM_PIO4 0.78539816339744830962f

PT_DEV void sincos_(float x, float* s_out, float* c_out) {
    float k = __builtin_rintf(x * 0.63661977236758134f);
    float r = fma_(-k, 1.5703125f, x);
    r = fma_(-k, 4.837512969970703125e-4f, r);
    r = fma_(-k, 7.54978995489188216e-8f, r);
    int q = (int)k & 3;
    float z = r * r;
    float sp = fma_(z, fma_(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
    float s = fma_(r * z, sp, r);
    float cp = fma_(z, fma_(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
    float c = fma_(z * z, cp, fma_(-0.5f, z, 1.0f));
    float so = (q & 1) ? c : s;
    float co = (q & 1) ? s : c;
    if (q & 2) so = -so;
    if ((q + 1) & 2) co = -co;
    *s_out = so;
    *c_out = co;
}
PT_DEV float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
PT_DEV float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }

PT_DEV float atan2_(float y, float x) {
    float ax = abs_(x), ay = abs_(y);
    float mx = fmax_(ax, ay), mn = fmin_(ax, ay);
    if (mx == 0.0f) return 0.0f;
    bool big = mn > 0.4142135623730950f * mx;
    float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    float t = div_(num, den);
    float z = t * t;
    float p = fma_(z, fma_(z, fma_(z, 8.05374449538e-2f, -1.38776856032e-1f), 1.99777106478e-1f), -3.33329491539e-1f);
    float r = fma_(p * z, t, t);
    if (big) r = DM_PIO4 + r;
    if (ay > ax) r = DM_PIO2 - r;
    if (x < 0.0f) r = DM_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

PT_DEV float asin_poly(float z) {
    return fma_(z, fma_(z, fma_(z, fma_(z, 4.2163199048e-2f, 2.4181311049e-2f), 4.5470025998e-2f), 7.4953002686e-2f), 1.6666752422e-1f);
}
PT_DEV float asin_(float x) {
    float a = abs_(x);
    bool big = a > 0.5f;
    float z = big ? 0.5f * (1.0f - a) : a * a;
    float s = big ? sqrt_(z) : a;
    float r = fma_(s * z, asin_poly(z), s);
    if (big) r = fma_(-2.0f, r, DM_PIO2);
    return x < 0.0f ? -r : r;
}
PT_DEV float acos_(float x) {
    float a = abs_(x);
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = sqrt_(z);
        float r = fma_(s * z, asin_poly(z), s);
        r = r + r;
        return x < 0.0f ? DM_PI - r : r;
    }
    float z = x * x;
    float r = fma_(x * z, asin_poly(z), x);
    return DM_PIO2 - r;
}

PT_DEV void log_core(float x, float* fe_out, float* lnm_out) {
    int eadj = 0;
    if (x < 1.17549435e-38f) { x = x * 8388608.0f; eadj = -23; }
    uint32_t b = f2u(x);
    int e = (int)((b >> 23) & 0xffu) - 126 + eadj;
    float m = u2f((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = fma_(m, 7.0376836292e-2f, -1.1514610310e-1f);
    p = fma_(p, m, 1.1676998740e-1f);
    p = fma_(p, m, -1.2420140846e-1f);
    p = fma_(p, m, 1.4249322787e-1f);
    p = fma_(p, m, -1.6668057665e-1f);
    p = fma_(p, m, 2.0000714765e-1f);
    p = fma_(p, m, -2.4999993993e-1f);
    p = fma_(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    y = fma_(-0.5f, z, y);
    *fe_out = (float)e;
    *lnm_out = m + y;
}
PT_DEV float log_(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    float fe, lnm;
    log_core(x, &fe, &lnm);
    float r = fma_(fe, -2.12194440e-4f, lnm);
    return fma_(fe, 0.693359375f, r);
}
PT_DEV float log2_(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    float fe, lnm;
    log_core(x, &fe, &lnm);
    return fma_(lnm, 1.44269504088896341f, fe);
}
PT_DEV float exp2_(float x) {
    if (x != x) return x;
    if (x > 128.0f) return __builtin_inff();
    if (x < -150.0f) return 0.0f;
    float k = __builtin_rintf(x);
    float f = x - k;
    float p = fma_(f, 1.535336188319500e-4f, 1.339887440266574e-3f);
    p = fma_(p, f, 9.618437357674640e-3f);
    p = fma_(p, f, 5.550332471162809e-2f);
    p = fma_(p, f, 2.402264791363012e-1f);
    p = fma_(p, f, 6.931472028550421e-1f);
    p = fma_(p, f, 1.0f);
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;
    float s1 = u2f((uint32_t)(k1 + 127) << 23);
    float s2 = u2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}
PT_DEV float pow_(float x, float y) { return exp2_(y * log2_(x)); }

}  // namespace dm


// ---- two evaluations side by side --------------------------------------------------------------------------------
// gfx950 issues v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- two independent IEEE binary32 operations on a register
// pair -- at the rate of one unpacked operation, so two evaluations of one polynomial cost little more than one.  The
// functions below are dm::sincos_, dm::log2_, dm::exp2_ and dm::pow_ on TWO arguments at once: element k of every result
// is, operation for operation, what the scalar function computes for element k (the same constants, the same fma
// chains, the same roundings, selects where the scalar code branches), so results are bit-identical; the distance
// estimators' inner iterations (figures_dev.h) use them for their pairs of calls.  The compiler's own vectoriser does not
// form these pairs (forced with -slp-threshold it loses more in moves than it gains).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
namespace dm2 {
PT_DEV f2 mk(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
PT_DEV f2 splat(float v) { return mk(v, v); }
PT_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
PT_DEV f2 sel(i2 cond, f2 a, f2 b) { return mk(cond.x ? a.x : b.x, cond.y ? a.y : b.y); }  // cond: 0 / non-zero per element
PT_DEV f2 u2f(u2 u) { return mk(__uint_as_float(u.x), __uint_as_float(u.y)); }
PT_DEV u2 f2u(f2 f) { u2 r; r.x = __float_as_uint(f.x); r.y = __float_as_uint(f.y); return r; }

PT_DEV void sincos2(f2 x, f2* s_out, f2* c_out) {  // dm::sincos_
    f2 k = __builtin_elementwise_rint(x * 0.63661977236758134f);
    f2 r = fma2(-k, splat(1.5703125f), x);
    r = fma2(-k, splat(4.837512969970703125e-4f), r);
    r = fma2(-k, splat(7.54978995489188216e-8f), r);
    const i2 q = __builtin_convertvector(k, i2) & 3;
    f2 z = r * r;
    f2 sp = fma2(z, fma2(z, splat(-1.9515295891e-4f), splat(8.3321608736e-3f)), splat(-1.6666654611e-1f));
    f2 s = fma2(r * z, sp, r);
    f2 cp = fma2(z, fma2(z, splat(2.443315711809948e-5f), splat(-1.388731625493765e-3f)), splat(4.166664568298827e-2f));
    f2 c = fma2(z * z, cp, fma2(splat(-0.5f), z, splat(1.0f)));
    f2 so = sel(q & 1, c, s), co = sel(q & 1, s, c);
    so = sel(q & 2, -so, so);
    co = sel((q + 1) & 2, -co, co);
    *s_out = so;
    *c_out = co;
}

// dm::log2_ for x > 0 finite; the other inputs (zero, negative, infinity, NaN) are patched in by selects from the scalar code's answers
PT_DEV f2 log2_2(f2 x0) {
    const i2 tiny = x0 < 1.17549435e-38f;
    f2 x = sel(tiny, x0 * 8388608.0f, x0);
    const u2 b = f2u(x);
    i2 e = __builtin_convertvector((b >> 23) & 0xffu, i2) - 126 + (tiny & -23);
    f2 m = u2f((b & 0x007fffffu) | 0x3f000000u);
    const i2 low = m < 0.70710678118654752440f;
    e = e + low;  // a true comparison is -1 per element
    m = sel(low, m + m - 1.0f, m - 1.0f);
    f2 z = m * m;
    f2 p = fma2(m, splat(7.0376836292e-2f), splat(-1.1514610310e-1f));
    p = fma2(p, m, splat(1.1676998740e-1f));
    p = fma2(p, m, splat(-1.2420140846e-1f));
    p = fma2(p, m, splat(1.4249322787e-1f));
    p = fma2(p, m, splat(-1.6668057665e-1f));
    p = fma2(p, m, splat(2.0000714765e-1f));
    p = fma2(p, m, splat(-2.4999993993e-1f));
    p = fma2(p, m, splat(3.3333331174e-1f));
    f2 y = (p * m) * z;
    y = fma2(splat(-0.5f), z, y);
    const f2 fe = __builtin_convertvector(e, f2);
    const f2 lnm = m + y;
    f2 r = fma2(lnm, splat(1.44269504088896341f), fe);
    // dm::log2_'s early returns
    r = sel(x0 == __builtin_inff(), x0, r);
    r = sel(x0 > 0.0f, r, sel(x0 == 0.0f, splat(-__builtin_inff()), splat(__builtin_nanf(""))));
    return r;
}

PT_DEV f2 exp2_2(f2 x) {  // dm::exp2_
    f2 k = __builtin_elementwise_rint(x);
    f2 f = x - k;
    f2 p = fma2(f, splat(1.535336188319500e-4f), splat(1.339887440266574e-3f));
    p = fma2(p, f, splat(9.618437357674640e-3f));
    p = fma2(p, f, splat(5.550332471162809e-2f));
    p = fma2(p, f, splat(2.402264791363012e-1f));
    p = fma2(p, f, splat(6.931472028550421e-1f));
    p = fma2(p, f, splat(1.0f));
    // the scalar code converts k only for -150 <= x <= 128; outside that range the result is replaced below, so k is clamped
    // first to keep the conversion defined
    const f2 kc = __builtin_elementwise_max(__builtin_elementwise_min(k, splat(200.0f)), splat(-200.0f));
    const i2 ki = __builtin_convertvector(kc, i2);
    const i2 k1 = ki / 2, k2 = ki - k1;
    const f2 s1 = u2f(__builtin_convertvector(k1 + 127, u2) << 23);
    const f2 s2 = u2f(__builtin_convertvector(k2 + 127, u2) << 23);
    f2 r = (p * s1) * s2;
    r = sel(x < -150.0f, splat(0.0f), r);
    r = sel(x > 128.0f, splat(__builtin_inff()), r);
    r = sel(x != x, x, r);
    return r;
}
PT_DEV f2 pow2(f2 x, f2 y) { return exp2_2(y * log2_2(x)); }  // dm::pow_

}  // namespace dm2

// ---- vector helpers: fixed expression trees ------------------------------------------------------
// dot = fma chain from the low index; cross = fma(a,b,-(c*d)); madd = fma(d,t,o); vector / scalar is
// vector * (1 / scalar); everything else is unfused and left to right.
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
PT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV V3 v3p(const float* p) { return v3(p[0], p[1], p[2]); }
PT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_DEV V3 operator/(V3 a, V3 b) { return v3(dm::div_(a.x, b.x), dm::div_(a.y, b.y), dm::div_(a.z, b.z)); }
PT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_DEV float dot(V3 a, V3 b) { return dm::fma_(a.z, b.z, dm::fma_(a.y, b.y, a.x * b.x)); }
PT_DEV float dot2(float ax, float ay, float bx, float by) { return dm::fma_(ay, by, ax * bx); }
PT_DEV float dot4(V4 a, V4 b) { return dm::fma_(a.w, b.w, dm::fma_(a.z, b.z, dm::fma_(a.y, b.y, a.x * b.x))); }
PT_DEV V3 cross(V3 a, V3 b) {
    return v3(dm::fma_(a.y, b.z, -(a.z * b.y)), dm::fma_(a.z, b.x, -(a.x * b.z)), dm::fma_(a.x, b.y, -(a.y * b.x)));
}
PT_DEV V3 madd(V3 o, V3 d, float t) { return v3(dm::fma_(d.x, t, o.x), dm::fma_(d.y, t, o.y), dm::fma_(d.z, t, o.z)); }
PT_DEV float length(V3 a) { return dm::sqrt_(dot(a, a)); }
#ifdef PT_NORMALIZE_RSQ  // experiment only: the hardware's reciprocal square root (1 ulp), as a shader compiler may emit for normalize()
PT_DEV V3 normalize(V3 a) { float inv = __builtin_amdgcn_rsqf(dot(a, a)); return a * inv; }
#else
PT_DEV V3 normalize(V3 a) { float inv = dm::rcp_(dm::sqrt_(dot(a, a))); return a * inv; }
#endif
// vector / scalar = vector * (1 / scalar): one correctly rounded reciprocal, then three multiplies.
PT_DEV V3 vdivs(V3 a, float s) { float inv = dm::rcp_(s); return a * inv; }
PT_DEV V3 vabs(V3 a) { return v3(dm::abs_(a.x), dm::abs_(a.y), dm::abs_(a.z)); }
PT_DEV V3 vmin(V3 a, V3 b) { return v3(dm::fmin_(a.x, b.x), dm::fmin_(a.y, b.y), dm::fmin_(a.z, b.z)); }
PT_DEV V3 vmax(V3 a, V3 b) { return v3(dm::fmax_(a.x, b.x), dm::fmax_(a.y, b.y), dm::fmax_(a.z, b.z)); }
PT_DEV V3 vrcp(V3 a) { return v3(dm::rcp_(a.x), dm::rcp_(a.y), dm::rcp_(a.z)); }
PT_DEV float clampf(float x, float lo, float hi) { return dm::fmin_(dm::fmax_(x, lo), hi); }
PT_DEV float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
PT_DEV float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

#define PT_PI 3.14159265359f
#define PT_TWO_PI 6.28318530718f
#define PT_INV_PI 0.31830988618f
#define PT_EPS 0.0001f
#define PT_MAX_T 1e20f
