// TEST INFRASTRUCTURE — CPU oracle for the per-pixel path-tracing compute pass.
//
// This file is a checker, not a product: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  It restates, in scalar strict-binary32 C++, what the reference's
// WGSL megakernel computes for one pixel-sample, following the reference file by file (citations
// are relative to the reference checkout).  The reference's own kernel cannot run in this
// environment (WGSL through wgpu/naga on a Vulkan device; no Rust toolchain, no Vulkan ICD) and
// the reference ships no golden vectors for this path, so PARITY IS UNPINNED BY THE REFERENCE at the
// level of bits: this restatement plus the committed fixtures generated from it are the pins there.
// One output of the reference's own pipeline does exist -- resources/cornell_box.png, the picture in
// its README -- and pins this file statistically: rendered with it, the scene that picture shows
// comes out within 0.5 % of the picture's radiance in every region and channel
// (tests/test_reference_image.py; the HIP kernels, block for block and down to the picture's own
// noise -- it is the mean of 3272 frames of the same random sequences: tests/test_gpu_reference_image.py).
//
// Numerical contract.  WGSL leaves fusing, the association of dot/cross sums and the accuracy of
// transcendental builtins implementation-defined.  The oracle fixes one conforming choice:
//   dot(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))          (dot2/dot4 likewise, low index first)
//   cross(a,b).x  = fma(a.y,b.z, -(a.z*b.y))   (cyclic)
//   o + d*t       = fma(d,t,o) per component  ("madd")
//   length(v)     = sqrt(dot(v,v));  normalize(v) = v * (1/length(v));  vector / scalar = vector * (1/scalar)
//   mix(a,b,t)    = a*(1-t) + b*t  (unfused);  every other vector expression is unfused, left to right
//   min/max       = IEEE minNum/maxNum with -0 < +0
//   x / y         = x * rcp(y), rcp = correctly rounded reciprocal that flushes subnormals (oracle/dmath.h);
//                   sqrt = correctly rounded IEEE
//   sin, cos, atan2, asin, acos, log, pow = oracle/dmath.h
// Build with -O2 -ffp-contract=off and no fast-math so the compiler adds no fusing of its own.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "dmath.h"

typedef uint32_t u32;
typedef int32_t i32;

// ---- reference ABI records -----------------------------------------------------------------
// src/scene/shape.rs:223-248 (WGSL Figure, types.wgsl:24-41)
struct GpuShape {
    u32 shape_type, material_idx; float radius, radius2;
    float position[3], height;
    float normal[3]; u32 csg_op;
    float rotation[3], texture_scale;
    float v0[3], pad2;
    float v1[3], pad3;
    float v2[3], pad4;
};
// src/scene/material.rs:108-119 (types.wgsl:43-52)
struct GpuMaterial {
    float base_color[3], metallic;
    float emission[3], roughness;
    float emission_strength, ior, transmission; i32 texture_id;
};
// src/accel/bvh.rs:12-19 (types.wgsl:54-59)
struct GpuBvhNode { float aabb_min[3]; u32 left_or_prim; float aabb_max[3]; u32 prim_count; };
// src/io/texture_atlas.rs:10-18
struct TextureInfo { u32 width, height, offset, pad; };
// src/camera/camera.rs:150-171 (types.wgsl:3-22)
struct GpuCamera {
    float position[3], focal_length;
    float right[3], aspect;
    float up[3], exposure;
    float forward[3]; u32 frame_index;
    u32 width, height, sample_count, max_bounces;
    u32 tone_mapper, fractal_march_steps; float firefly_clamp, skybox_brightness;
    float skybox_color[3], pad2;
};
static_assert(sizeof(GpuShape) == 112 && sizeof(GpuMaterial) == 48 && sizeof(GpuBvhNode) == 32 &&
              sizeof(TextureInfo) == 16 && sizeof(GpuCamera) == 112, "reference ABI sizes");

extern "C" {
struct PtoScene {
    const GpuShape* figs; u32 nfig;
    const GpuMaterial* mats; u32 nmat;
    const GpuBvhNode* nodes; u32 nnode;
    const u32* prims; u32 nprim;
    const u32* lights; u32 nlight;
    const u32* infinite; u32 ninf;
    const u32* tex_pixels; u32 npix;
    const TextureInfo* tex_infos; u32 ntex;
};
// Work tallies for the algorithmic flop/byte figures.
struct PtoCounters {
    uint64_t paths, bounces, shadow_rays, node_visits, aabb_tests;
    uint64_t prim_tests[17];
    uint64_t sdf_evals[17];     // indexed by figure type (5 torus, 9 mandelbulb, 10 julia, 14 mebius)
    uint64_t fractal_inner;     // inner escape-loop iterations (mandelbulb + julia)
    uint64_t eval_brdf, sample_brdf, sample_glass, nee, plane_tests_infinite;
};
}

namespace {

const float PI = 3.14159265359f;
const float TWO_PI = 6.28318530718f;
const float INV_PI = 0.31830988618f;
const float EPSILON = 0.0001f;
const float MAX_T = 1e20f;

using dm::fmin_;
using dm::fmax_;

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, V3 b) { return {dm::div_(a.x, b.x), dm::div_(a.y, b.y), dm::div_(a.z, b.z)}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline float dot2(float ax, float ay, float bx, float by) { return fmaf(ay, by, ax * bx); }
inline float dot4(V4 a, V4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }
inline V3 cross(V3 a, V3 b) {
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
inline V3 madd(V3 o, V3 d, float t) { return {fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z)}; }
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { float inv = dm::rcp_(sqrtf(dot(a, a))); return a * inv; }
// vector / scalar = vector * (1 / scalar): one correctly rounded reciprocal, then three multiplies.
inline V3 vdivs(V3 a, float s) { float inv = dm::rcp_(s); return a * inv; }
inline V3 vabs(V3 a) { return {fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
inline V3 vmin(V3 a, V3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline float clampf(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
inline float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

struct Ray { V3 origin, direction; };
// types.wgsl:61-68; WGSL zero-initialises `var hit: HitRecord` and `HitRecord()`.
struct Hit {
    float t = 0.0f; V3 position{0, 0, 0}; V3 normal{0, 0, 0}; V2 uv{0, 0}; u32 figure_idx = 0; bool hit = false;
};
inline Hit miss_hit() { Hit h; h.hit = false; h.t = MAX_T; return h; }

struct Ctx {
    const PtoScene* sc;
    const GpuCamera* cam;
    u32 rng;
    PtoCounters* cnt;   // may be null
    std::vector<u32> negatives;  // indices with csg_op == 1, ascending (bvh.wgsl:132-140 scans all figures)
    i32 sky_fig = -1;            // first FIG_SKYBOX whose material has texture_id >= 0 (skybox.wgsl:9-20)
};

// ---- random.wgsl:6-23 ------------------------------------------------------------------------
inline u32 pcg_hash(u32 input) {
    u32 state = input * 747796405u + 2891336453u;
    u32 word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
inline void init_rng(Ctx& c, u32 px, u32 py, u32 frame) { c.rng = pcg_hash(px + py * 65536u + frame * 16777259u); }
// The literal 4294967295.0 rounds to 2^32 in f32 and f32(u32) rounds to nearest-even, so the value
// is f32(state) * 2^-32 exactly (and can be 1.0).
inline float rand_f32(Ctx& c) { c.rng = pcg_hash(c.rng); return (float)c.rng * 2.3283064365386963e-10f; }

// ---- utils.wgsl ------------------------------------------------------------------------------
struct Onb { V3 u, v, n; };
inline Onb build_onb(V3 n) {  // utils.wgsl:6-13
    float s = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = dm::div_(-1.0f, s + n.z);
    float b = n.x * n.y * a;
    Onb o;
    o.u = v3(1.0f + s * n.x * n.x * a, s * b, -s * n.x);
    o.v = v3(b, s + n.y * n.y * a, -n.y);
    o.n = n;
    return o;
}
inline V3 onb_mul(const Onb& o, V3 l) {  // mat3x3 * vec3 = u*l.x + v*l.y + n*l.z
    return {fmaf(o.n.x, l.z, fmaf(o.v.x, l.y, o.u.x * l.x)), fmaf(o.n.y, l.z, fmaf(o.v.y, l.y, o.u.y * l.x)),
            fmaf(o.n.z, l.z, fmaf(o.v.z, l.y, o.u.z * l.x))};
}
inline V3 sample_cosine_hemisphere(Ctx& c, V3 n) {  // utils.wgsl:16-25
    float rx = rand_f32(c), ry = rand_f32(c);
    float phi = TWO_PI * rx;
    float cos_theta = sqrtf(ry);
    float sin_theta = sqrtf(1.0f - ry);
    float sp, cp;
    dm::sincos_(phi, &sp, &cp);
    V3 local = v3(cp * sin_theta, sp * sin_theta, cos_theta);
    return normalize(onb_mul(build_onb(n), local));
}
inline V3 reflect_vec(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }  // utils.wgsl:28-30
inline V3 refract_vec(V3 v, V3 n, float eta) {                               // utils.wgsl:33-41
    float cos_i = dot(-v, n);
    float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
    if (sin2_t > 1.0f) return v3(0, 0, 0);
    float cos_t = sqrtf(1.0f - sin2_t);
    return v * eta + n * (eta * cos_i - cos_t);
}
inline V3 fresnel_schlick(float cos_theta, V3 f0) {  // utils.wgsl:44-48
    float t = 1.0f - cos_theta;
    float t2 = t * t;
    float w = t2 * t2 * t;
    return f0 + (v3(1, 1, 1) - f0) * w;
}
inline float fresnel_schlick_scalar(float cos_theta, float ior) {  // utils.wgsl:51-57
    float r0 = dm::div_(1.0f - ior, 1.0f + ior);
    r0 = r0 * r0;
    float t = 1.0f - cos_theta;
    float t2 = t * t;
    return r0 + (1.0f - r0) * (t2 * t2 * t);
}
inline float luminance(V3 c) { return dot(c, v3(0.2126f, 0.7152f, 0.0722f)); }  // utils.wgsl:60-62

// ---- textures.wgsl:13-29 -----------------------------------------------------------------------
inline V3 sample_texture(const Ctx& c, i32 texture_id, V2 uv) {
    if (texture_id < 0) return v3(1, 1, 1);
    if ((u32)texture_id >= c.sc->ntex) return v3(0, 0, 0);
    const TextureInfo& info = c.sc->tex_infos[(u32)texture_id];
    float wx = uv.x - floorf(uv.x), wy = uv.y - floorf(uv.y);
    // u32(f32) saturates in WGSL; the products are >= 0 here unless NaN (-> 0).
    float fx = wx * (float)info.width, fy = wy * (float)info.height;
    u32 px = fx > 0.0f ? (u32)fx : 0u, py = fy > 0.0f ? (u32)fy : 0u;
    px = std::min(px, info.width - 1u);
    py = std::min(py, info.height - 1u);
    u32 idx = info.offset + py * info.width + px;
    u32 packed = idx < c.sc->npix ? c.sc->tex_pixels[idx] : 0u;
    return v3((float)(packed & 0xFFu) * (1.0f / 255.0f), (float)((packed >> 8) & 0xFFu) * (1.0f / 255.0f),
              (float)((packed >> 16) & 0xFFu) * (1.0f / 255.0f));
}
inline V2 sphere_uv(V3 d) {  // sphere.wgsl:34-37, skybox.wgsl:14-17, ellipsoid.wgsl:46-49
    return {0.5f + dm::atan2_(d.z, d.x) * (1.0f / TWO_PI), 0.5f - dm::asin_(clampf(d.y, -1.0f, 1.0f)) * (1.0f / PI)};
}

// ---- figures ---------------------------------------------------------------------------------
Hit intersect_sphere(const Ray& ray, const GpuShape& f) {  // sphere.wgsl:4-40
    Hit hit = miss_hit();
    V3 oc = ray.origin - v3(f.position);
    float half_b = dot(oc, ray.direction);
    float c = dot(oc, oc) - f.radius * f.radius;
    float disc = fmaf(half_b, half_b, -c);
    if (disc < 0.0f) return hit;
    float sqrtd = sqrtf(disc);
    float t = -half_b - sqrtd;
    if (t < EPSILON) {
        t = -half_b + sqrtd;
        if (t < EPSILON) return hit;
    }
    hit.hit = true;
    hit.t = t;
    hit.position = madd(ray.origin, ray.direction, t);
    hit.normal = normalize(hit.position - v3(f.position));
    hit.uv = sphere_uv(hit.normal);
    return hit;
}
Hit intersect_plane(const Ray& ray, const GpuShape& f) {  // plane.wgsl:4-30
    Hit hit = miss_hit();
    V3 n = v3(f.normal);
    float denom = dot(n, ray.direction);
    if (fabsf(denom) < EPSILON) return hit;
    float t = dm::div_(dot(v3(f.position) - ray.origin, n), denom);
    if (t < EPSILON) return hit;
    hit.hit = true;
    hit.t = t;
    hit.position = madd(ray.origin, ray.direction, t);
    hit.normal = denom < 0.0f ? n : -n;
    Onb onb = build_onb(hit.normal);
    V3 local = hit.position - v3(f.position);
    hit.uv = {dot(local, onb.u) * 0.25f, dot(local, onb.v) * 0.25f};
    return hit;
}
Hit intersect_cube(const Ray& ray, const GpuShape& f) {  // cube.wgsl:4-60
    Hit hit = miss_hit();
    V3 half = v3(f.radius, f.radius, f.radius);
    V3 box_min = v3(f.position) - half, box_max = v3(f.position) + half;
    V3 inv_dir = v3(dm::rcp_(ray.direction.x), dm::rcp_(ray.direction.y), dm::rcp_(ray.direction.z));
    V3 t1 = (box_min - ray.origin) * inv_dir, t2 = (box_max - ray.origin) * inv_dir;
    V3 tmin = vmin(t1, t2), tmax = vmax(t1, t2);
    float t_near = fmax_(fmax_(tmin.x, tmin.y), tmin.z);
    float t_far = fmin_(fmin_(tmax.x, tmax.y), tmax.z);
    if (t_near > t_far || t_far < EPSILON) return hit;
    float t = t_near;
    if (t < EPSILON) {
        t = t_far;
        if (t < EPSILON) return hit;
    }
    hit.hit = true;
    hit.t = t;
    hit.position = madd(ray.origin, ray.direction, t);
    V3 p = vdivs(hit.position - v3(f.position), f.radius);
    V3 ap = vabs(p);
    if (ap.x > ap.y && ap.x > ap.z) hit.normal = v3(signf(p.x), 0, 0);
    else if (ap.y > ap.z) hit.normal = v3(0, signf(p.y), 0);
    else hit.normal = v3(0, 0, signf(p.z));
    if (fabsf(hit.normal.x) > 0.5f) hit.uv = {(p.y + 1.0f) * 0.5f, (p.z + 1.0f) * 0.5f};
    else if (fabsf(hit.normal.y) > 0.5f) hit.uv = {(p.x + 1.0f) * 0.5f, (p.z + 1.0f) * 0.5f};
    else hit.uv = {(p.x + 1.0f) * 0.5f, (p.y + 1.0f) * 0.5f};
    return hit;
}
Hit intersect_cylinder(const Ray& ray, const GpuShape& f) {  // cylinder.wgsl:4-69
    Hit hit = miss_hit();
    V3 axis = v3(f.normal), pos = v3(f.position);
    float half_h = f.height * 0.5f;
    V3 oc = ray.origin - pos;
    V3 d_perp = ray.direction - axis * dot(ray.direction, axis);
    V3 oc_perp = oc - axis * dot(oc, axis);
    float a = dot(d_perp, d_perp);
    float half_b = dot(d_perp, oc_perp);
    float c = dot(oc_perp, oc_perp) - f.radius * f.radius;
    float disc = fmaf(half_b, half_b, -(a * c));
    if (disc < 0.0f) return hit;
    float sqrtd = sqrtf(disc);
    float inv_a = dm::rcp_(a);
    for (int i = 0; i < 2; i++) {
        float t = i == 0 ? (-half_b - sqrtd) * inv_a : (-half_b + sqrtd) * inv_a;
        if (t < EPSILON) continue;
        V3 p = madd(ray.origin, ray.direction, t);
        float proj = dot(p - pos, axis);
        if (fabsf(proj) <= half_h) {
            hit.hit = true;
            hit.t = t;
            hit.position = p;
            hit.normal = normalize((p - pos) - axis * proj);
            float angle = dm::atan2_(hit.normal.z, hit.normal.x);
            hit.uv = {angle * (1.0f / TWO_PI) + 0.5f, dm::div_(proj + half_h, f.height)};
            return hit;
        }
    }
    float d_axis = dot(ray.direction, axis);
    if (fabsf(d_axis) > EPSILON) {
        for (int sign_i = 0; sign_i < 2; sign_i++) {
            float cap_sign = sign_i == 1 ? -1.0f : 1.0f;
            V3 cap_center = pos + axis * half_h * cap_sign;
            float t = dm::div_(dot(cap_center - ray.origin, axis * cap_sign), d_axis * cap_sign);
            if (t > EPSILON && t < hit.t) {
                V3 p = madd(ray.origin, ray.direction, t);
                V3 offset = p - cap_center;
                if (dot(offset, offset) <= f.radius * f.radius) {
                    hit.hit = true;
                    hit.t = t;
                    hit.position = p;
                    hit.normal = axis * cap_sign;
                    hit.uv = {(dm::div_(offset.x, f.radius) + 1.0f) * 0.5f, (dm::div_(offset.z, f.radius) + 1.0f) * 0.5f};
                }
            }
        }
    }
    return hit;
}
Hit intersect_cone(const Ray& ray, const GpuShape& f) {  // cone.wgsl:4-80 (lines 17-20 are dead code)
    Hit hit = miss_hit();
    V3 axis = v3(f.normal), pos = v3(f.position);
    V3 apex = pos + axis * f.height;
    float tan2 = f.radius2;  // raw radius2: the host never converts it (shape.rs:264)
    V3 oc = ray.origin - apex;
    float d_dot_v = dot(ray.direction, axis);
    float oc_dot_v = dot(oc, axis);
    float cos2 = dm::rcp_(1.0f + tan2);
    float a_c = fmaf(d_dot_v, d_dot_v, -(cos2 * dot(ray.direction, ray.direction)));
    float b_c = fmaf(d_dot_v, oc_dot_v, -(cos2 * dot(ray.direction, oc)));
    float c_c = fmaf(oc_dot_v, oc_dot_v, -(cos2 * dot(oc, oc)));
    float disc = fmaf(b_c, b_c, -(a_c * c_c));
    if (disc < 0.0f) return hit;
    float sqrtd = sqrtf(disc);
    float inv_a = dm::rcp_(a_c);
    for (int i = 0; i < 2; i++) {
        float t = i == 0 ? (-b_c - sqrtd) * inv_a : (-b_c + sqrtd) * inv_a;
        if (t < EPSILON) continue;
        V3 p = madd(ray.origin, ray.direction, t);
        float proj = dot(p - apex, axis);
        if (proj >= -f.height && proj <= 0.0f) {
            hit.hit = true;
            hit.t = t;
            hit.position = p;
            V3 to_p = normalize(p - apex);
            float n_proj = dot(to_p, axis);
            hit.normal = normalize(to_p - axis * n_proj * (1.0f + tan2));
            float angle = dm::atan2_(hit.normal.z, hit.normal.x);
            hit.uv = {angle * (1.0f / TWO_PI) + 0.5f, dm::div_(-proj, f.height)};
            return hit;
        }
    }
    float d_axis2 = dot(ray.direction, axis);
    if (fabsf(d_axis2) > EPSILON) {
        float t = dm::div_(dot(pos - ray.origin, axis), d_axis2);
        if (t > EPSILON && t < hit.t) {
            V3 p = madd(ray.origin, ray.direction, t);
            V3 offset = p - pos;
            if (dot(offset, offset) <= f.radius * f.radius) {
                hit.hit = true;
                hit.t = t;
                hit.position = p;
                hit.normal = -axis;
                hit.uv = {(dm::div_(offset.x, f.radius) + 1.0f) * 0.5f, (dm::div_(offset.z, f.radius) + 1.0f) * 0.5f};
            }
        }
    }
    return hit;
}
Hit intersect_disc(const Ray& ray, const GpuShape& f) {  // disc.wgsl:4-35
    Hit hit = miss_hit();
    V3 n = v3(f.normal), pos = v3(f.position);
    float denom = dot(n, ray.direction);
    if (fabsf(denom) < EPSILON) return hit;
    float t = dm::div_(dot(pos - ray.origin, n), denom);
    if (t < EPSILON) return hit;
    V3 p = madd(ray.origin, ray.direction, t);
    V3 offset = p - pos;
    float dist_sq = dot(offset, offset);
    float r_sq = f.radius * f.radius;
    if (dist_sq > r_sq) return hit;
    hit.hit = true;
    hit.t = t;
    hit.position = p;
    hit.normal = denom < 0.0f ? n : -n;
    hit.uv = {(dm::div_(offset.x, f.radius) + 1.0f) * 0.5f, (dm::div_(offset.z, f.radius) + 1.0f) * 0.5f};
    return hit;
}
// WGSL unpack2x16float: exact binary16 -> binary32.
inline float half_to_float(u32 h) {
    u32 sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return dm::u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f;  // m * 2^-24, exact
        return sign ? -v : v;
    }
    if (e == 31) return dm::u2f(sign | 0x7F800000u | (m << 13));
    return dm::u2f(sign | ((e + 112u) << 23) | (m << 13));
}
inline V2 unpack2x16float(float packed) {
    u32 b = dm::f2u(packed);
    return {half_to_float(b & 0xFFFFu), half_to_float(b >> 16)};
}
// Moller-Trumbore core shared by triangle.wgsl:9-36, pyramid.wgsl:38-61, tetrahedron.wgsl:36-59.
inline bool mt_core(const Ray& ray, V3 v0, V3 e1, V3 e2, float* t_out, float* u_out, float* v_out) {
    V3 h = cross(ray.direction, e2);
    float a = dot(e1, h);
    if (fabsf(a) < EPSILON) return false;
    float fi = dm::rcp_(a);
    V3 s = ray.origin - v0;
    float u = fi * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, e1);
    float v = fi * dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    *t_out = fi * dot(e2, q);
    *u_out = u;
    *v_out = v;
    return true;
}
Hit intersect_triangle(const Ray& ray, const GpuShape& f) {  // triangle.wgsl:4-53
    Hit hit = miss_hit();
    V3 v0 = v3(f.v0);
    V3 e1 = v3(f.v1) - v0, e2 = v3(f.v2) - v0;
    float t, u, v;
    if (!mt_core(ray, v0, e1, e2, &t, &u, &v)) return hit;
    if (t < EPSILON) return hit;
    hit.hit = true;
    hit.t = t;
    hit.position = madd(ray.origin, ray.direction, t);
    hit.normal = normalize(cross(e1, e2));
    if (dot(hit.normal, ray.direction) > 0.0f) hit.normal = -hit.normal;
    V2 a = unpack2x16float(f.pad2), b = unpack2x16float(f.pad3), c = unpack2x16float(f.pad4);
    float w0 = 1.0f - u - v;
    hit.uv = {fmaf(v, c.x, fmaf(u, b.x, w0 * a.x)), fmaf(v, c.y, fmaf(u, b.y, w0 * a.y))};
    return hit;
}
// pyramid.wgsl:35-70 / tetrahedron.wgsl:33-68: keep-closest variant, normal not ray-facing, uv = (u, v).
inline void tri_test(const Ray& ray, V3 v0, V3 v1, V3 v2, Hit& hit) {
    V3 e1 = v1 - v0, e2 = v2 - v0;
    float t, u, v;
    if (!mt_core(ray, v0, e1, e2, &t, &u, &v)) return;
    if (t > EPSILON && t < hit.t) {
        hit.hit = true;
        hit.t = t;
        hit.position = madd(ray.origin, ray.direction, t);
        hit.normal = normalize(cross(e1, e2));
        hit.uv = {u, v};
    }
}
Hit intersect_pyramid(const Ray& ray, const GpuShape& f) {  // pyramid.wgsl:5-32
    Hit hit = miss_hit();
    V3 base = v3(f.position);
    float r = f.radius, h = f.height;
    V3 apex = base + v3(0, h, 0);
    V3 b0 = base + v3(-r, 0, -r), b1 = base + v3(r, 0, -r), b2 = base + v3(r, 0, r), b3 = base + v3(-r, 0, r);
    tri_test(ray, apex, b0, b1, hit);
    tri_test(ray, apex, b1, b2, hit);
    tri_test(ray, apex, b2, b3, hit);
    tri_test(ray, apex, b3, b0, hit);
    tri_test(ray, b0, b2, b1, hit);
    tri_test(ray, b0, b3, b2, hit);
    return hit;
}
Hit intersect_tetrahedron(const Ray& ray, const GpuShape& f) {  // tetrahedron.wgsl:4-31
    Hit hit = miss_hit();
    float r = f.radius;
    V3 pos = v3(f.position);
    float a = r * 0.942809f, b = r * 0.333333f, c = r * 0.816497f;
    V3 v0 = pos + v3(0, r, 0);
    V3 v1 = pos + v3(a, -b, -c * 0.5f);
    V3 v2 = pos + v3(-a, -b, -c * 0.5f);
    V3 v3_ = pos + v3(0, -b, c);
    tri_test(ray, v0, v1, v2, hit);
    tri_test(ray, v0, v2, v3_, hit);
    tri_test(ray, v0, v3_, v1, hit);
    tri_test(ray, v1, v3_, v2, hit);
    return hit;
}
Hit intersect_ellipsoid(const Ray& ray, const GpuShape& f) {  // ellipsoid.wgsl:5-52
    Hit hit = miss_hit();
    V3 radii = v3(f.radius, fmax_(f.height, f.radius), fmax_(f.radius2, f.radius));
    V3 inv_radii = v3(dm::rcp_(radii.x), dm::rcp_(radii.y), dm::rcp_(radii.z));
    V3 pos = v3(f.position);
    V3 oc = (ray.origin - pos) * inv_radii;
    V3 dir = ray.direction * inv_radii;
    float a = dot(dir, dir);
    float half_b = dot(oc, dir);
    float c = dot(oc, oc) - 1.0f;
    float disc = fmaf(half_b, half_b, -(a * c));
    if (disc < 0.0f) return hit;
    float sqrtd = sqrtf(disc);
    float inv_a = dm::rcp_(a);
    float t = (-half_b - sqrtd) * inv_a;
    if (t < EPSILON) {
        t = (-half_b + sqrtd) * inv_a;
        if (t < EPSILON) return hit;
    }
    hit.hit = true;
    hit.t = t;
    hit.position = madd(ray.origin, ray.direction, t);
    V3 local = hit.position - pos;
    hit.normal = normalize(local * inv_radii * inv_radii);
    hit.uv = sphere_uv(normalize(local * inv_radii));
    return hit;
}
Hit intersect_paraboloid(const Ray& ray, const GpuShape& f) {  // paraboloid.wgsl:4-67
    Hit hit = miss_hit();
    V3 pos = v3(f.position);
    V3 oc = ray.origin - pos, d = ray.direction;
    float r = f.radius, h = f.height;
    float a = dot2(d.x, d.z, d.x, d.z);
    float b = 2.0f * dot2(oc.x, oc.z, d.x, d.z) - r * d.y;
    float c = dot2(oc.x, oc.z, oc.x, oc.z) - r * oc.y;
    float disc = fmaf(b, b, -(4.0f * a * c));
    if (disc < 0.0f) return hit;
    float sqrtd = sqrtf(disc);
    float inv_2a = dm::div_(0.5f, a);
    for (int i = 0; i < 2; i++) {
        float t = i == 0 ? (-b - sqrtd) * inv_2a : (-b + sqrtd) * inv_2a;
        if (t < EPSILON) continue;
        V3 p = madd(ray.origin, d, t);
        float local_y = p.y - pos.y;
        if (local_y >= 0.0f && local_y <= h) {
            hit.hit = true;
            hit.t = t;
            hit.position = p;
            V3 local = p - pos;
            hit.normal = normalize(v3(2.0f * local.x, -r, 2.0f * local.z));
            float angle = dm::atan2_(local.z, local.x);
            hit.uv = {angle * (1.0f / TWO_PI) + 0.5f, dm::div_(local_y, h)};
            return hit;
        }
    }
    float d_y = d.y;
    if (fabsf(d_y) > EPSILON) {
        float cap_y = pos.y + h;
        float t = dm::div_(cap_y - ray.origin.y, d_y);
        if (t > EPSILON) {
            V3 p = madd(ray.origin, d, t);
            V3 local = p - pos;
            float dist2 = dot2(local.x, local.z, local.x, local.z);
            if (dist2 <= r * h) {
                hit.hit = true;
                hit.t = t;
                hit.position = p;
                hit.normal = v3(0, 1, 0);
                float s = sqrtf(r * h);
                hit.uv = {dm::div_(local.x, s) * 0.5f + 0.5f, dm::div_(local.z, s) * 0.5f + 0.5f};
            }
        }
    }
    return hit;
}
Hit intersect_hyperboloid(const Ray& ray, const GpuShape& f) {  // hyperboloid.wgsl:4-68
    Hit hit = miss_hit();
    V3 pos = v3(f.position);
    V3 oc = ray.origin - pos, d = ray.direction;
    float r = f.radius;
    float half_h = f.height * 0.5f;
    float inv_r2 = dm::rcp_(r * r);
    float a = (dot2(d.x, d.z, d.x, d.z) - d.y * d.y) * inv_r2;
    float b = 2.0f * (dot2(oc.x, oc.z, d.x, d.z) - oc.y * d.y) * inv_r2;
    float c = (dot2(oc.x, oc.z, oc.x, oc.z) - oc.y * oc.y) * inv_r2 - 1.0f;
    float disc = fmaf(b, b, -(4.0f * a * c));
    if (disc >= 0.0f) {
        float sqrtd = sqrtf(disc);
        float inv_2a = dm::div_(0.5f, a);
        for (int i = 0; i < 2; i++) {
            float t = i == 0 ? (-b - sqrtd) * inv_2a : (-b + sqrtd) * inv_2a;
            if (t < EPSILON || t >= hit.t) continue;
            V3 p = madd(ray.origin, d, t);
            float local_y = p.y - pos.y;
            if (fabsf(local_y) <= half_h) {
                hit.hit = true;
                hit.t = t;
                hit.position = p;
                V3 local = p - pos;
                hit.normal = normalize(v3(2.0f * local.x, -2.0f * local.y, 2.0f * local.z) * inv_r2);
                float angle = dm::atan2_(local.z, local.x);
                hit.uv = {angle * (1.0f / TWO_PI) + 0.5f, dm::div_(local_y + half_h, f.height)};
            }
        }
    }
    for (int cap = 0; cap < 2; cap++) {
        float cap_y = pos.y + (cap == 1 ? half_h : -half_h);
        V3 cap_normal = cap == 1 ? v3(0, 1, 0) : v3(0, -1, 0);
        if (fabsf(d.y) > EPSILON) {
            float t = dm::div_(cap_y - ray.origin.y, d.y);
            if (t > EPSILON && t < hit.t) {
                V3 p = madd(ray.origin, d, t);
                V3 local = p - pos;
                float dist2 = dot2(local.x, local.z, local.x, local.z);
                float cap_r2 = r * r * (1.0f + (half_h * half_h) * inv_r2);
                if (dist2 <= cap_r2) {
                    hit.hit = true;
                    hit.t = t;
                    hit.position = p;
                    hit.normal = cap_normal;
                    float s = sqrtf(cap_r2);
                    hit.uv = {dm::div_(local.x, s) * 0.5f + 0.5f, dm::div_(local.z, s) * 0.5f + 0.5f};
                }
            }
        }
    }
    return hit;
}

// --- ray-marched figures -------------------------------------------------------------------------
inline float sdf_torus(V3 p, float major_r, float minor_r) {  // torus.wgsl:5-8
    float qx = sqrtf(dot2(p.x, p.z, p.x, p.z)) - major_r;
    return sqrtf(dot2(qx, p.y, qx, p.y)) - minor_r;
}
// Tetrahedron-normal weights e.xyy, e.yyx, e.yxy, e.xxx with e = (k, -k).
template <class F>
inline V3 tet_normal(V3 local, float k, F sdf) {
    V3 o0 = v3(k, -k, -k), o1 = v3(-k, -k, k), o2 = v3(-k, k, -k), o3 = v3(k, k, k);
    float d0 = sdf(local + o0), d1 = sdf(local + o1), d2 = sdf(local + o2), d3 = sdf(local + o3);
    return normalize(o0 * d0 + o1 * d1 + o2 * d2 + o3 * d3);
}
Hit intersect_torus(Ctx& c, const Ray& ray, const GpuShape& f) {  // torus.wgsl:10-81
    Hit hit = miss_hit();
    float major_r = f.radius, minor_r = f.radius2;
    float bound_r = major_r + minor_r;
    V3 pos = v3(f.position);
    V3 oc = ray.origin - pos;
    float b = dot(oc, ray.direction);
    float br = bound_r + minor_r;
    float cc = dot(oc, oc) - br * br;
    float disc = fmaf(b, b, -cc);
    if (disc < 0.0f) return hit;
    const float omega = 1.4f;
    float sq = sqrtf(fmax_(disc, 0.0f));
    float t = fmax_(-b - sq, EPSILON);
    float max_t = fmin_(-b + sq, MAX_T);
    float prev_d = 0.0f;
    auto sdf = [&](V3 p) { if (c.cnt) c.cnt->sdf_evals[5]++; return sdf_torus(p, major_r, minor_r); };
    for (int i = 0; i < 128; i++) {
        V3 p = madd(ray.origin, ray.direction, t) - pos;
        float d = sdf(p);
        if (d < 0.0f && prev_d > 0.0f) {
            t -= prev_d * (omega - 1.0f);
            V3 p2 = madd(ray.origin, ray.direction, t) - pos;
            float d2 = sdf(p2);
            t += d2;
            prev_d = d2;
            continue;
        }
        if (fabsf(d) < EPSILON * t) {
            hit.hit = true;
            hit.t = t;
            hit.position = madd(ray.origin, ray.direction, t);
            V3 local = hit.position - pos;
            hit.normal = tet_normal(local, 0.5773f * EPSILON, sdf);
            float angle_major = dm::atan2_(local.z, local.x);
            float px = sqrtf(dot2(local.x, local.z, local.x, local.z)) - major_r;
            float angle_minor = dm::atan2_(local.y, px);
            hit.uv = {angle_major * (1.0f / TWO_PI) + 0.5f, angle_minor * (1.0f / TWO_PI) + 0.5f};
            return hit;
        }
        t += d * omega;
        prev_d = d;
        if (t > max_t) break;
    }
    return hit;
}
inline float sdf_mebius(V3 p, float R) {  // mebius.wgsl:5-36
    float w = R * 0.3f;
    float angle = dm::atan2_(p.z, p.x);
    float sa, ca;
    dm::sincos_(angle, &sa, &ca);
    V3 center = v3(R * ca, 0.0f, R * sa);
    V3 radial = normalize(v3(ca, 0.0f, sa));
    float half_angle = angle * 0.5f;
    float sh, ch;
    dm::sincos_(half_angle, &sh, &ch);
    V3 strip_up = v3(0, 1, 0) * ch + radial * sh;
    V3 dp = p - center;
    float v = dot(dp, strip_up);
    float u = length(dp - strip_up * v);
    float du = fabsf(u) - w * 0.15f;
    float dv = fabsf(v) - w;
    float mx = fmax_(du, 0.0f), my = fmax_(dv, 0.0f);
    return sqrtf(dot2(mx, my, mx, my)) + fmin_(fmax_(du, dv), 0.0f);
}
Hit intersect_mebius(Ctx& c, const Ray& ray, const GpuShape& f) {  // mebius.wgsl:38-87
    Hit hit = miss_hit();
    V3 pos = v3(f.position);
    V3 oc = ray.origin - pos;
    float b = dot(oc, ray.direction);
    float bound_r = f.radius * 1.5f;
    float cc = dot(oc, oc) - bound_r * bound_r;
    float disc = fmaf(b, b, -cc);
    if (disc < 0.0f) return hit;
    float sq = sqrtf(disc);
    float t = fmax_(-b - sq, EPSILON);
    float max_t = -b + sq;
    auto sdf = [&](V3 p) { if (c.cnt) c.cnt->sdf_evals[14]++; return sdf_mebius(p, f.radius); };
    for (int i = 0; i < 256; i++) {
        V3 p = madd(ray.origin, ray.direction, t) - pos;
        float d = sdf(p);
        if (fabsf(d) < EPSILON * t * 0.5f) {
            hit.hit = true;
            hit.t = t;
            hit.position = madd(ray.origin, ray.direction, t);
            V3 local = hit.position - pos;
            hit.normal = tet_normal(local, 0.5773f * EPSILON * 2.0f, sdf);
            float angle = dm::atan2_(local.z, local.x);
            hit.uv = {angle * (1.0f / TWO_PI) + 0.5f, 0.5f};
            return hit;
        }
        t += d;
        if (t > max_t) break;
    }
    return hit;
}
inline float sdf_mandelbulb(Ctx& c, V3 p, float power, int max_iter) {  // mandelbulb.wgsl:5-30
    V3 w = p;
    float m = dot(w, w);
    float dz = 1.0f;
    for (int i = 0; i < max_iter; i++) {
        if (c.cnt) c.cnt->fractal_inner++;
        dz = fmaf(power * dm::pow_(m, (power - 1.0f) * 0.5f), dz, 1.0f);
        float r = sqrtf(m);
        float b = power * dm::acos_(clampf(dm::div_(w.y, r), -1.0f, 1.0f));
        float a = power * dm::atan2_(w.x, w.z);
        float rp = dm::pow_(r, power);
        float sb, cb, sa, ca;
        dm::sincos_(b, &sb, &cb);
        dm::sincos_(a, &sa, &ca);
        w = p + v3(sb * sa, cb, sb * ca) * rp;
        m = dot(w, w);
        if (m > 256.0f) break;
    }
    float r = sqrtf(m);
    return dm::div_(0.25f * dm::log_(m) * r, dz);
}
inline V4 quat_mult(V4 a, V4 b) {  // julia.wgsl:4-11 (sums left to right, unfused)
    return {a.x * b.x - a.y * b.y - a.z * b.z - a.w * b.w, a.x * b.y + a.y * b.x + a.z * b.w - a.w * b.z,
            a.x * b.z - a.y * b.w + a.z * b.x + a.w * b.y, a.x * b.w + a.y * b.z - a.z * b.y + a.w * b.x};
}
inline float sdf_julia(Ctx& c, V3 p, V4 cc, int max_iter) {  // julia.wgsl:13-32
    V4 z = {p.x, p.y, p.z, 0.0f};
    V4 dz = {1.0f, 0.0f, 0.0f, 0.0f};
    float r2 = dot4(z, z);
    for (int i = 0; i < max_iter; i++) {
        if (r2 > 16.0f) break;
        if (c.cnt) c.cnt->fractal_inner++;
        V4 q = quat_mult(z, dz);
        dz = {2.0f * q.x, 2.0f * q.y, 2.0f * q.z, 2.0f * q.w};
        V4 zz = quat_mult(z, z);
        z = {zz.x + cc.x, zz.y + cc.y, zz.z + cc.z, zz.w + cc.w};
        r2 = dot4(z, z);
    }
    float r = sqrtf(r2);
    float dr = sqrtf(dot4(dz, dz));
    return dm::div_(0.5f * r * dm::log_(r), dr);
}
// mandelbulb.wgsl:32-101 and julia.wgsl:34-101 share the marcher.
template <class F>
Hit march_fractal(Ctx& c, const Ray& ray, const GpuShape& f, F sdf_unit) {
    Hit hit = miss_hit();
    V3 pos = v3(f.position);
    V3 oc = ray.origin - pos;
    float b = dot(oc, ray.direction);
    float bound_r = f.radius * 1.5f;
    float cs = dot(oc, oc) - bound_r * bound_r;
    float disc = fmaf(b, b, -cs);
    if (disc < 0.0f) return hit;
    const float omega = 1.3f;
    const float inv_radius = dm::rcp_(f.radius);  // p / radius = p * (1 / radius)
    float sq = sqrtf(disc);
    float t = fmax_(-b - sq, EPSILON);
    float max_t = -b + sq;
    float prev_d = 0.0f;
    for (u32 i = 0; i < c.cam->fractal_march_steps; i++) {
        V3 p = madd(ray.origin, ray.direction, t) - pos;
        float d = sdf_unit(p * inv_radius) * f.radius;
        if (d < 0.0f && prev_d > 0.0f) {
            t -= prev_d * (omega - 1.0f);
            V3 p2 = madd(ray.origin, ray.direction, t) - pos;
            float d2 = sdf_unit(p2 * inv_radius) * f.radius;
            t += d2;
            prev_d = d2;
            continue;
        }
        if (fabsf(d) < EPSILON * t * 0.5f) {
            hit.hit = true;
            hit.t = t;
            hit.position = madd(ray.origin, ray.direction, t);
            V3 local = (hit.position - pos) * inv_radius;
            hit.normal = tet_normal(local, 0.5773f * EPSILON * 2.0f, sdf_unit);
            hit.uv = {0.0f, 0.0f};
            return hit;
        }
        t += d * omega;
        prev_d = d;
        if (t > max_t) break;
    }
    return hit;
}
Hit intersect_mandelbulb(Ctx& c, const Ray& ray, const GpuShape& f) {
    float power = f.v0[0];
    int max_iter = (int)f.v0[1];
    return march_fractal(c, ray, f, [&](V3 p) { if (c.cnt) c.cnt->sdf_evals[9]++; return sdf_mandelbulb(c, p, power, max_iter); });
}
Hit intersect_julia(Ctx& c, const Ray& ray, const GpuShape& f) {
    V4 cc = {f.rotation[0], f.rotation[1], f.rotation[2], f.radius2};
    int max_iter = (int)f.v0[1];
    return march_fractal(c, ray, f, [&](V3 p) { if (c.cnt) c.cnt->sdf_evals[10]++; return sdf_julia(c, p, cc, max_iter); });
}
Hit intersect_skybox(const Ray& ray, const GpuShape& f) {  // skybox.wgsl:30-62 (unreachable via the BVH: state.rs:550)
    Hit hit = miss_hit();
    V3 pos = v3(f.position);
    V3 oc = ray.origin - pos;
    float half_b = dot(oc, ray.direction);
    float c = dot(oc, oc) - f.radius * f.radius;
    float disc = fmaf(half_b, half_b, -c);
    if (disc < 0.0f) return hit;
    float t = -half_b + sqrtf(disc);
    if (t < EPSILON) return hit;
    hit.hit = true;
    hit.t = t;
    hit.position = madd(ray.origin, ray.direction, t);
    V3 d = normalize(hit.position - pos);
    hit.normal = -d;
    hit.uv = sphere_uv(d);
    return hit;
}

Hit intersect_figure(Ctx& c, const Ray& ray, u32 idx) {  // figures/dispatch.wgsl:22-89
    const GpuShape& f = c.sc->figs[idx];
    if (c.cnt && f.shape_type < 17) c.cnt->prim_tests[f.shape_type]++;
    Hit hit;
    switch (f.shape_type) {
        case 0: hit = intersect_sphere(ray, f); break;
        case 1: hit = intersect_plane(ray, f); break;
        case 2: hit = intersect_cube(ray, f); break;
        case 3: hit = intersect_cylinder(ray, f); break;
        case 4: hit = intersect_cone(ray, f); break;
        case 5: hit = intersect_torus(c, ray, f); break;
        case 6: hit = intersect_disc(ray, f); break;
        case 7: hit = intersect_triangle(ray, f); break;
        case 8: hit = intersect_skybox(ray, f); break;
        case 9: hit = intersect_mandelbulb(c, ray, f); break;
        case 10: hit = intersect_julia(c, ray, f); break;
        case 11: hit = intersect_ellipsoid(ray, f); break;
        case 12: hit = intersect_paraboloid(ray, f); break;
        case 13: hit = intersect_hyperboloid(ray, f); break;
        case 14: hit = intersect_mebius(c, ray, f); break;
        case 15: hit = intersect_pyramid(ray, f); break;
        case 16: hit = intersect_tetrahedron(ray, f); break;
        default: hit.hit = false; hit.t = MAX_T; break;
    }
    if (hit.hit) hit.figure_idx = idx;
    return hit;
}

// ---- bvh.wgsl --------------------------------------------------------------------------------
inline float intersect_aabb(Ctx& c, V3 origin, V3 inv_dir, const float* bmin, const float* bmax) {  // bvh.wgsl:4-15
    if (c.cnt) c.cnt->aabb_tests++;
    V3 t1 = (v3(bmin) - origin) * inv_dir, t2 = (v3(bmax) - origin) * inv_dir;
    V3 tmin = vmin(t1, t2), tmax = vmax(t1, t2);
    float t_near = fmax_(fmax_(tmin.x, tmin.y), tmin.z);
    float t_far = fmin_(fmin_(tmax.x, tmax.y), tmax.z);
    if (t_near > t_far || t_far < 0.0f) return MAX_T;
    return fmax_(t_near, 0.0f);
}
bool is_inside_figure(const Ctx& c, V3 p, u32 idx) {  // bvh.wgsl:18-64
    const GpuShape& f = c.sc->figs[idx];
    V3 pos = v3(f.position), n = v3(f.normal);
    switch (f.shape_type) {
        case 0: return length(p - pos) < f.radius;
        case 2: { V3 l = vabs(p - pos); float h = f.radius; return l.x < h && l.y < h && l.z < h; }
        case 3: {
            V3 oc = p - pos;
            float proj = dot(oc, n);
            V3 perp = oc - n * proj;
            float half_h = f.height * 0.5f;
            return fabsf(proj) < half_h && dot(perp, perp) < f.radius * f.radius;
        }
        case 4: {
            V3 apex = pos + n * f.height;
            V3 oc = p - apex;
            float proj = dot(oc, n);
            if (proj > 0.0f || proj < -f.height) return false;
            float cos2 = dm::rcp_(1.0f + f.radius2);
            float r2 = dot(oc, oc);
            return proj * proj > cos2 * r2;
        }
        case 11: {
            V3 l = p - pos;
            V3 v = l / v3(fmax_(f.radius, 0.001f), fmax_(f.radius2, 0.001f), fmax_(f.height, 0.001f));
            return dot(v, v) < 1.0f;
        }
        default: return false;
    }
}
float find_exit_t(const Ctx& c, const Ray& ray, u32 idx) {  // bvh.wgsl:67-129
    const GpuShape& f = c.sc->figs[idx];
    V3 pos = v3(f.position), axis = v3(f.normal);
    switch (f.shape_type) {
        case 0: {
            V3 oc = ray.origin - pos;
            float half_b = dot(oc, ray.direction);
            float cc = dot(oc, oc) - f.radius * f.radius;
            float disc = fmaf(half_b, half_b, -cc);
            if (disc < 0.0f) return MAX_T;
            return -half_b + sqrtf(disc);
        }
        case 2: {
            V3 half = v3(f.radius, f.radius, f.radius);
            V3 inv_dir = v3(dm::rcp_(ray.direction.x), dm::rcp_(ray.direction.y), dm::rcp_(ray.direction.z));
            V3 t1 = ((pos - half) - ray.origin) * inv_dir, t2 = ((pos + half) - ray.origin) * inv_dir;
            V3 tmax = vmax(t1, t2);
            return fmin_(fmin_(tmax.x, tmax.y), tmax.z);
        }
        case 3: {
            float half_h = f.height * 0.5f;
            V3 oc = ray.origin - pos;
            V3 d_perp = ray.direction - axis * dot(ray.direction, axis);
            V3 oc_perp = oc - axis * dot(oc, axis);
            float a = dot(d_perp, d_perp);
            float half_b = dot(d_perp, oc_perp);
            float cc = dot(oc_perp, oc_perp) - f.radius * f.radius;
            float disc = fmaf(half_b, half_b, -(a * cc));
            if (disc < 0.0f) return MAX_T;
            float t_far = dm::div_(-half_b + sqrtf(disc), a);
            float d_axis = dot(ray.direction, axis);
            if (fabsf(d_axis) > EPSILON) {
                float t_top = dm::div_(half_h - dot(oc, axis), d_axis);
                float t_bot = dm::div_(-half_h - dot(oc, axis), d_axis);
                return fmin_(t_far, fmax_(t_top, t_bot));
            }
            return t_far;
        }
        case 4: {
            V3 apex = pos + axis * f.height;
            V3 oc = ray.origin - apex;
            float cos2 = dm::rcp_(1.0f + f.radius2);
            float d_dot_v = dot(ray.direction, axis);
            float oc_dot_v = dot(oc, axis);
            float a_c = fmaf(d_dot_v, d_dot_v, -(cos2 * dot(ray.direction, ray.direction)));
            float b_c = fmaf(d_dot_v, oc_dot_v, -(cos2 * dot(ray.direction, oc)));
            float c_c = fmaf(oc_dot_v, oc_dot_v, -(cos2 * dot(oc, oc)));
            float disc = fmaf(b_c, b_c, -(a_c * c_c));
            if (disc < 0.0f) return MAX_T;
            return dm::div_(-b_c + sqrtf(disc), a_c);
        }
        default: return MAX_T;
    }
}
bool is_inside_any_negative(const Ctx& c, V3 p) {  // bvh.wgsl:132-140
    for (u32 i : c.negatives)
        if (is_inside_figure(c, p, i)) return true;
    return false;
}
void test_infinite_shapes(Ctx& c, const Ray& ray, Hit& closest) {  // bvh.wgsl:192-209
    for (u32 i = 0; i < c.sc->ninf; i++) {
        u32 prim = c.sc->infinite[i];
        if (prim == 0xFFFFFFFFu) continue;
        if (c.sc->figs[prim].csg_op == 1u) continue;
        if (c.cnt) c.cnt->plane_tests_infinite++;
        Hit h = intersect_figure(c, ray, prim);
        if (h.hit && h.t > EPSILON && h.t < closest.t) closest = h;
    }
}
bool test_infinite_shapes_shadow(Ctx& c, const Ray& ray, float max_t) {  // bvh.wgsl:212-232
    for (u32 i = 0; i < c.sc->ninf; i++) {
        u32 prim = c.sc->infinite[i];
        if (prim == 0xFFFFFFFFu) continue;
        if (c.sc->figs[prim].csg_op == 1u) continue;
        if (c.cnt) c.cnt->plane_tests_infinite++;
        Hit h = intersect_figure(c, ray, prim);
        if (h.hit && h.t > EPSILON && h.t < max_t - EPSILON)
            if (!is_inside_any_negative(c, h.position)) return true;
    }
    return false;
}
Hit trace_bvh_positive(Ctx& c, const Ray& ray) {  // bvh.wgsl:236-305
    Hit closest;
    closest.t = MAX_T;
    closest.hit = false;
    V3 inv_dir = v3(dm::rcp_(ray.direction.x), dm::rcp_(ray.direction.y), dm::rcp_(ray.direction.z));
    const GpuBvhNode* nodes = c.sc->nodes;
    // An empty BVH is uploaded by the reference as one zeroed node (bvh.rs:38-43); it can never
    // produce a hit, so it is skipped here.
    if (c.sc->nprim > 0 && c.sc->nnode > 0) {
        u32 stack[32];
        int sp = 0;
        stack[0] = 0u;
        sp = 1;
        while (sp > 0) {
            sp -= 1;
            u32 ni = stack[sp];
            const GpuBvhNode& node = nodes[ni];
            if (c.cnt) c.cnt->node_visits++;
            float t_box = intersect_aabb(c, ray.origin, inv_dir, node.aabb_min, node.aabb_max);
            if (t_box >= closest.t) continue;
            if (node.prim_count > 0u) {
                for (u32 i = 0; i < node.prim_count; i++) {
                    u32 prim = c.sc->prims[node.left_or_prim + i];
                    if (c.sc->figs[prim].csg_op == 1u) continue;
                    Hit h = intersect_figure(c, ray, prim);
                    if (h.hit && h.t < closest.t && h.t > EPSILON) closest = h;
                }
            } else {
                u32 li = ni + 1u, ri = node.left_or_prim;
                float t_left = intersect_aabb(c, ray.origin, inv_dir, nodes[li].aabb_min, nodes[li].aabb_max);
                float t_right = intersect_aabb(c, ray.origin, inv_dir, nodes[ri].aabb_min, nodes[ri].aabb_max);
                if (t_left < t_right) {
                    if (t_right < closest.t && sp < 31) stack[sp++] = ri;
                    if (t_left < closest.t && sp < 31) stack[sp++] = li;
                } else {
                    if (t_left < closest.t && sp < 31) stack[sp++] = li;
                    if (t_right < closest.t && sp < 31) stack[sp++] = ri;
                }
            }
        }
    }
    test_infinite_shapes(c, ray, closest);
    return closest;
}
Hit trace_bvh(Ctx& c, const Ray& ray) {  // bvh.wgsl:143-188
    Hit closest = trace_bvh_positive(c, ray);
    if (closest.hit && !c.negatives.empty()) {
        Ray cur = ray;
        float total = 0.0f;
        for (u32 attempt = 0; attempt < 8u; attempt++) {
            if (!is_inside_any_negative(c, closest.position)) break;
            float min_exit = MAX_T;
            for (u32 i : c.negatives)
                if (is_inside_figure(c, closest.position, i)) min_exit = fmin_(min_exit, find_exit_t(c, cur, i));
            if (min_exit >= MAX_T) break;
            float advance = min_exit + EPSILON * 2.0f;
            total += advance;
            cur = Ray{madd(cur.origin, cur.direction, advance), cur.direction};
            closest = trace_bvh_positive(c, cur);
            if (closest.hit) closest.t += total;
            else break;
        }
    }
    return closest;
}
bool trace_shadow(Ctx& c, const Ray& ray, float max_t) {  // bvh.wgsl:308-362
    if (c.cnt) c.cnt->shadow_rays++;
    V3 inv_dir = v3(dm::rcp_(ray.direction.x), dm::rcp_(ray.direction.y), dm::rcp_(ray.direction.z));
    const GpuBvhNode* nodes = c.sc->nodes;
    if (c.sc->nprim > 0 && c.sc->nnode > 0) {
        u32 stack[32];
        int sp = 1;
        stack[0] = 0u;
        while (sp > 0) {
            sp -= 1;
            u32 ni = stack[sp];
            const GpuBvhNode& node = nodes[ni];
            if (c.cnt) c.cnt->node_visits++;
            float t_box = intersect_aabb(c, ray.origin, inv_dir, node.aabb_min, node.aabb_max);
            if (t_box >= max_t) continue;
            if (node.prim_count > 0u) {
                for (u32 i = 0; i < node.prim_count; i++) {
                    u32 prim = c.sc->prims[node.left_or_prim + i];
                    if (c.sc->figs[prim].csg_op == 1u) continue;
                    Hit h = intersect_figure(c, ray, prim);
                    if (h.hit && h.t > EPSILON && h.t < max_t - EPSILON)
                        if (!is_inside_any_negative(c, h.position)) return true;
                }
            } else {
                if (sp < 31) stack[sp++] = ni + 1u;
                if (sp < 31) stack[sp++] = node.left_or_prim;
            }
        }
    }
    return test_infinite_shapes_shadow(c, ray, max_t);
}

// ---- skybox.wgsl:6-27 ----------------------------------------------------------------------
V3 sample_skybox(const Ctx& c, V3 direction) {
    if (c.sky_fig >= 0) {
        const GpuMaterial& m = c.sc->mats[c.sc->figs[c.sky_fig].material_idx];
        V3 d = normalize(direction);
        return sample_texture(c, m.texture_id, sphere_uv(d)) * v3(m.base_color);
    }
    float t = 0.5f * (direction.y + 1.0f);
    V3 top = v3(c.cam->skybox_color);
    V3 col = v3(mixf(1.0f, top.x, t), mixf(1.0f, top.y, t), mixf(1.0f, top.z, t));
    return col * c.cam->skybox_brightness;
}

// ---- materials.wgsl ---------------------------------------------------------------------------
struct Mat { V3 base_color; float metallic; V3 emission; float roughness, emission_strength, ior, transmission; i32 texture_id; };
inline float ggx_ndf(float n_dot_h, float alpha) {  // materials.wgsl:17-21
    float a2 = alpha * alpha;
    float denom = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
    return dm::div_(a2, PI * denom * denom);
}
inline float ggx_g1(float n_dot_v, float alpha) {  // materials.wgsl:24-28
    float a2 = alpha * alpha;
    float denom = n_dot_v + sqrtf(a2 + (1.0f - a2) * n_dot_v * n_dot_v);
    return dm::div_(2.0f * n_dot_v, denom);
}
inline float ggx_pdf(float n_dot_h, float v_dot_h, float alpha) {  // materials.wgsl:52-55
    return dm::div_(ggx_ndf(n_dot_h, alpha) * n_dot_h, 4.0f * v_dot_h);
}
V3 eval_brdf(Ctx& c, V3 wo, V3 wi, V3 n, const Mat& mat) {  // materials.wgsl:65-90
    if (c.cnt) c.cnt->eval_brdf++;
    float n_dot_l = fmax_(dot(n, wi), 0.0f);
    float n_dot_v = fmax_(dot(n, wo), 0.0f);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return v3(0, 0, 0);
    V3 h = normalize(wo + wi);
    float n_dot_h = fmax_(dot(n, h), 0.0f);
    float v_dot_h = fmax_(dot(wo, h), 0.0f);
    float alpha = mat.roughness * mat.roughness;
    V3 f0 = v3(mixf(0.04f, mat.base_color.x, mat.metallic), mixf(0.04f, mat.base_color.y, mat.metallic),
               mixf(0.04f, mat.base_color.z, mat.metallic));
    V3 f = fresnel_schlick(v_dot_h, f0);
    float d = ggx_ndf(n_dot_h, alpha);
    float g = ggx_g1(n_dot_l, alpha) * ggx_g1(n_dot_v, alpha);
    V3 specular = vdivs(f * (d * g), fmax_(4.0f * n_dot_l * n_dot_v, EPSILON));
    V3 kd = (v3(1, 1, 1) - f) * (1.0f - mat.metallic);
    V3 diffuse = kd * mat.base_color * INV_PI;
    return diffuse + specular;
}
struct BrdfSample { V3 direction{0, 0, 0}; V3 brdf_cos{0, 0, 0}; float pdf = 0.0f; };
inline V3 sample_ggx_half(Ctx& c, V3 n, float alpha) {  // materials.wgsl:38-49
    float rx = rand_f32(c), ry = rand_f32(c);
    float a2 = alpha * alpha;
    float cos_theta = sqrtf(dm::div_(1.0f - ry, 1.0f + (a2 - 1.0f) * ry));
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float phi = TWO_PI * rx;
    float sp, cp;
    dm::sincos_(phi, &sp, &cp);
    V3 local = v3(sin_theta * cp, sin_theta * sp, cos_theta);
    return normalize(onb_mul(build_onb(n), local));
}
BrdfSample sample_brdf(Ctx& c, V3 wo, V3 n, const Mat& mat) {  // materials.wgsl:95-160
    if (c.cnt) c.cnt->sample_brdf++;
    float alpha = mat.roughness * mat.roughness;
    float spec_weight = mixf(0.04f, 1.0f, mat.metallic);
    float spec_prob = fmax_(spec_weight, 0.25f);
    BrdfSample res;
    V3 wi, h;
    float n_dot_l;
    if (rand_f32(c) < spec_prob) {
        h = sample_ggx_half(c, n, alpha);
        wi = reflect_vec(-wo, h);
        n_dot_l = dot(n, wi);
        if (n_dot_l <= 0.0f) { res.pdf = 1.0f; return res; }
    } else {
        wi = sample_cosine_hemisphere(c, n);
        n_dot_l = fmax_(dot(n, wi), 0.0f);
        if (n_dot_l <= 0.0f) { res.pdf = 1.0f; return res; }
        h = normalize(wo + wi);
    }
    float n_dot_h = fmax_(dot(n, h), 0.0f);
    float v_dot_h = fmax_(dot(wo, h), 0.0f);
    V3 brdf = eval_brdf(c, wo, wi, n, mat);
    float spec_pdf = ggx_pdf(n_dot_h, v_dot_h, alpha);
    float diff_pdf = n_dot_l * INV_PI;
    float pdf = spec_prob * spec_pdf + (1.0f - spec_prob) * diff_pdf;
    res.direction = wi;
    res.brdf_cos = brdf * n_dot_l;
    res.pdf = fmax_(pdf, EPSILON);
    return res;
}
BrdfSample sample_glass(Ctx& c, V3 wo, V3 n, const Mat& mat) {  // materials.wgsl:163-195
    if (c.cnt) c.cnt->sample_glass++;
    BrdfSample res;
    bool entering = dot(wo, n) > 0.0f;
    V3 face_n = entering ? n : -n;
    float eta = entering ? dm::rcp_(mat.ior) : mat.ior;  // ior / 1.0
    float cos_i = fabsf(dot(wo, face_n));
    float fresnel = fresnel_schlick_scalar(cos_i, mat.ior);
    res.brdf_cos = mat.base_color;
    if (rand_f32(c) < fresnel) {
        res.direction = reflect_vec(-wo, face_n);
        res.pdf = fresnel;
    } else {
        V3 refracted = refract_vec(-wo, face_n, eta);
        if (length(refracted) < 0.001f) {
            res.direction = reflect_vec(-wo, face_n);
            res.pdf = 1.0f;
        } else {
            res.direction = normalize(refracted);
            res.pdf = 1.0f - fresnel;
        }
    }
    return res;
}

// ---- path_trace.wgsl:56-182 ---------------------------------------------------------------------
struct PathOut { V3 radiance; u32 primary_id; float primary_t; u32 bounces; };
PathOut trace_path(Ctx& c, Ray ray) {
    V3 throughput = v3(1, 1, 1), radiance = v3(0, 0, 0);
    PathOut out;
    out.primary_id = 0xFFFFFFFFu;
    out.primary_t = MAX_T;
    out.bounces = 0;
    const PtoScene& sc = *c.sc;
    // An empty light list is uploaded as the one-element sentinel [0xFFFFFFFF] (state.rs:381-387), so
    // arrayLength() is 1 and the NEE branch still runs: it draws its three random numbers, reads an
    // out-of-bounds figure (zero radius under either robust-access policy) and adds nothing.
    const u32 num_lights = sc.nlight > 0 ? sc.nlight : 1u;
    for (u32 bounce = 0; bounce < c.cam->max_bounces; bounce++) {
        if (c.cnt) c.cnt->bounces++;
        out.bounces++;
        Hit hit = trace_bvh(c, ray);
        if (bounce == 0 && hit.hit) { out.primary_id = hit.figure_idx; out.primary_t = hit.t; }
        if (!hit.hit) {
            radiance = radiance + throughput * sample_skybox(c, ray.direction);
            break;
        }
        const GpuShape& fig = sc.figs[hit.figure_idx];
        const GpuMaterial& gm = sc.mats[fig.material_idx];
        Mat mat{v3(gm.base_color), gm.metallic, v3(gm.emission), gm.roughness, gm.emission_strength, gm.ior, gm.transmission, gm.texture_id};
        V2 tex_uv = {hit.uv.x * fig.texture_scale, hit.uv.y * fig.texture_scale};
        mat.base_color = mat.base_color * sample_texture(c, mat.texture_id, tex_uv);
        if (mat.emission_strength > 0.0f) {
            V3 le = mat.emission * mat.emission_strength;
            radiance = radiance + throughput * le;
            break;
        }
        V3 wo = -ray.direction;
        V3 n = hit.normal;
        if (dot(n, wo) < 0.0f) n = -n;
        if (mat.transmission > 0.5f) {
            BrdfSample gs = sample_glass(c, wo, n, mat);
            if (length(gs.direction) < 0.001f) break;
            throughput = throughput * gs.brdf_cos;
            ray = Ray{madd(hit.position, gs.direction, EPSILON * 2.0f), gs.direction};
            continue;
        }
        if (mat.roughness > 0.04f) {
            if (c.cnt) c.cnt->nee++;
            float pick = rand_f32(c);
            u32 light_pick = (u32)(pick * (float)num_lights);
            u32 light_idx_safe = std::min(light_pick, num_lights - 1u);
            float r1 = rand_f32(c), r2 = rand_f32(c);  // sample_sphere_light's rand_vec2, drawn unconditionally
            if (sc.nlight > 0) {
                u32 light_fig_idx = sc.lights[light_idx_safe];
                const GpuShape& lf = sc.figs[light_fig_idx];
                const GpuMaterial& lm = sc.mats[lf.material_idx];
                // lighting.wgsl:6-20
                float cos_theta = 1.0f - 2.0f * r1;
                float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                float phi = TWO_PI * r2;
                float sp, cp;
                dm::sincos_(phi, &sp, &cp);
                V3 local_point = v3(sin_theta * cp, sin_theta * sp, cos_theta) * lf.radius;
                V3 light_point = v3(lf.position) + local_point;
                V3 to_light = light_point - hit.position;
                float light_dist = length(to_light);
                V3 light_dir = vdivs(to_light, light_dist);
                float n_dot_l = dot(n, light_dir);
                if (n_dot_l > 0.0f) {
                    Ray shadow{madd(hit.position, n, EPSILON * 2.0f), light_dir};
                    bool occluded = trace_shadow(c, shadow, light_dist);
                    if (!occluded) {
                        V3 le = v3(lm.emission) * lm.emission_strength;
                        V3 light_normal = normalize(light_point - v3(lf.position));
                        float cos_light = fabsf(dot(-light_dir, light_normal));
                        float area = 4.0f * PI * lf.radius * lf.radius;  // lighting.wgsl:23-26
                        float light_area_pdf = dm::rcp_(area);
                        float light_solid_pdf = cos_light <= 0.0f ? 0.0f : dm::div_(light_area_pdf * (light_dist * light_dist), cos_light);
                        V3 brdf = eval_brdf(c, wo, light_dir, n, mat);
                        float brdf_pdf_val = n_dot_l * INV_PI;
                        float pa = light_solid_pdf * (float)num_lights;
                        float a2 = pa * pa, b2 = brdf_pdf_val * brdf_pdf_val;  // mis.wgsl:3-7
                        float w = dm::div_(a2, fmax_(a2 + b2, 1e-10f));
                        if (light_solid_pdf > 0.0f)
                            radiance = radiance + vdivs(throughput * le * brdf * n_dot_l * w, pa);
                    }
                }
            }
        }
        BrdfSample bs = sample_brdf(c, wo, n, mat);
        if (length(bs.direction) < 0.001f || bs.pdf < EPSILON) break;
        throughput = throughput * vdivs(bs.brdf_cos, bs.pdf);
        ray = Ray{madd(hit.position, bs.direction, EPSILON * 2.0f), bs.direction};
        if (bounce >= 3u) {
            float survival = fmin_(fmax_(throughput.x, fmax_(throughput.y, throughput.z)), 0.95f);
            if (rand_f32(c) > survival) break;
            throughput = vdivs(throughput, survival);
        }
        float lum = luminance(throughput);
        if (lum > c.cam->firefly_clamp) throughput = throughput * (dm::div_(c.cam->firefly_clamp, lum));
    }
    out.radiance = v3(fmax_(radiance.x, 0.0f), fmax_(radiance.y, 0.0f), fmax_(radiance.z, 0.0f));
    return out;
}

// ---- camera.wgsl:6-19 --------------------------------------------------------------------------
Ray generate_ray(Ctx& c, float px_, float py_) {
    const GpuCamera& cam = *c.cam;
    float jx = rand_f32(c) - 0.5f, jy = rand_f32(c) - 0.5f;
    float px = px_ + jx, py = py_ + jy;
    float ndc_x = (dm::div_(2.0f * px, (float)cam.width) - 1.0f) * cam.aspect;
    float ndc_y = 1.0f - dm::div_(2.0f * py, (float)cam.height);
    V3 dir = normalize(v3(cam.right) * ndc_x + v3(cam.up) * ndc_y + v3(cam.forward) * cam.focal_length);
    return Ray{v3(cam.position), dir};
}

// ---- tonemap.wgsl -------------------------------------------------------------------------------
inline float aces(float x) {  // tonemap.wgsl:4-11
    return clampf(dm::div_(x * (2.51f * x + 0.03f), x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
}
inline float linear_to_srgb(float c) {  // tonemap.wgsl:19-23
    float lo = c * 12.92f;
    float hi = 1.055f * dm::pow_(c, (1.0f / 2.4f)) - 0.055f;
    return c <= 0.0031308f ? lo : hi;
}
inline V3 apply_tonemap(const GpuCamera& cam, V3 color) {  // tonemap.wgsl:27-42
    V3 e = color * cam.exposure;
    V3 m;
    switch (cam.tone_mapper) {
        case 1u: m = v3(dm::div_(e.x, 1.0f + e.x), dm::div_(e.y, 1.0f + e.y), dm::div_(e.z, 1.0f + e.z)); break;
        case 2u: m = v3(clampf(e.x, 0, 1), clampf(e.y, 0, 1), clampf(e.z, 0, 1)); break;
        default: m = v3(aces(e.x), aces(e.y), aces(e.z)); break;
    }
    return v3(linear_to_srgb(m.x), linear_to_srgb(m.y), linear_to_srgb(m.z));
}
// rgba8unorm store: clamp to [0,1], scale by 255, round to nearest (ties to even).
inline uint8_t unorm8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 1.0f) return 255;
    return (uint8_t)rintf(v * 255.0f);
}

void init_ctx(Ctx& c, const PtoScene* sc, const GpuCamera* cam, PtoCounters* cnt) {
    c.sc = sc;
    c.cam = cam;
    c.cnt = cnt;
    c.rng = 0;
    for (u32 i = 0; i < sc->nfig; i++) {
        if (sc->figs[i].csg_op == 1u) c.negatives.push_back(i);
        if (c.sky_fig < 0 && sc->figs[i].shape_type == 8u && sc->mats[sc->figs[i].material_idx].texture_id >= 0) c.sky_fig = (i32)i;
    }
}

}  // namespace

extern "C" {

// Renders samples [first_sample, first_sample+n) of every pixel of the tile (x0,y0,tw,th) of a
// cam->width x cam->height frame, applying path_trace.wgsl:29-54 once per sample in order:
// sample k uses sample_count = first_sample + k and frame_index = first_frame + k.
//   accum      in/out, full frame, w*h*4 floats (rgb running mean, a = 1)
//   rgba8      out, full frame, w*h*4 bytes (may be null)
//   primary_id out, full frame u32 (figure hit by the LAST sample's camera ray, 0xFFFFFFFF = miss; may be null)
//   primary_t  out, full frame f32 (may be null)
//   radiance   out, n * tw * th * 3 floats, per-sample radiance in tile order (may be null)
//   m2         in/out, full frame w*h*3 floats: Welford sum of squared deviations per channel (may be null)
//   bounce_cnt out, full-frame u32, bounces summed over the n samples (may be null)
int pto_render(const PtoScene* sc, const GpuCamera* cam_in, u32 first_frame, u32 first_sample, u32 n,
               u32 x0, u32 y0, u32 tw, u32 th, float* accum, uint8_t* rgba8, u32* primary_id, float* primary_t,
               float* radiance, float* m2, u32* bounce_cnt, PtoCounters* counters, int threads) {
    if (!sc || !cam_in || !accum) return -1;
    const u32 W = cam_in->width, H = cam_in->height;
    if (x0 + tw > W || y0 + th > H) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    std::vector<PtoCounters> tl;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
#endif
    tl.resize(nthreads);
    for (auto& t : tl) std::memset(&t, 0, sizeof(t));
#pragma omp parallel for schedule(dynamic, 1)
    for (int ty = 0; ty < (int)th; ty++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        Ctx c;
        GpuCamera cam = *cam_in;
        init_ctx(c, sc, &cam, counters ? &tl[tid] : nullptr);
        u32 py = y0 + (u32)ty;
        for (u32 tx = 0; tx < tw; tx++) {
            u32 px = x0 + tx;
            size_t idx = (size_t)py * W + px;
            V3 mean = v3(accum[idx * 4 + 0], accum[idx * 4 + 1], accum[idx * 4 + 2]);
            u32 bsum = 0;
            PathOut po{};
            for (u32 k = 0; k < n; k++) {
                cam.frame_index = first_frame + k;
                cam.sample_count = first_sample + k;
                init_rng(c, px, py, cam.frame_index);
                Ray ray = generate_ray(c, (float)px, (float)py);
                po = trace_path(c, ray);
                if (c.cnt) c.cnt->paths++;
                bsum += po.bounces;
                float nn = fmax_((float)cam.sample_count, 1.0f);
                V3 delta = po.radiance - mean;
                V3 nm = mean + vdivs(delta, nn);  // path_trace.wgsl:45-49
                if (m2) {
                    V3 d2 = po.radiance - nm;
                    m2[idx * 3 + 0] = fmaf(delta.x, d2.x, m2[idx * 3 + 0]);
                    m2[idx * 3 + 1] = fmaf(delta.y, d2.y, m2[idx * 3 + 1]);
                    m2[idx * 3 + 2] = fmaf(delta.z, d2.z, m2[idx * 3 + 2]);
                }
                mean = nm;
                if (radiance) {
                    size_t r = ((size_t)k * th * tw + (size_t)ty * tw + tx) * 3;
                    radiance[r] = po.radiance.x; radiance[r + 1] = po.radiance.y; radiance[r + 2] = po.radiance.z;
                }
            }
            if (n > 0) {
                accum[idx * 4 + 0] = mean.x; accum[idx * 4 + 1] = mean.y; accum[idx * 4 + 2] = mean.z; accum[idx * 4 + 3] = 1.0f;
                if (rgba8) {
                    V3 col = apply_tonemap(cam, mean);
                    rgba8[idx * 4 + 0] = unorm8(col.x); rgba8[idx * 4 + 1] = unorm8(col.y);
                    rgba8[idx * 4 + 2] = unorm8(col.z); rgba8[idx * 4 + 3] = 255;
                }
                if (primary_id) primary_id[idx] = po.primary_id;
                if (primary_t) primary_t[idx] = po.primary_t;
                if (bounce_cnt) bounce_cnt[idx] = bsum;
            }
        }
    }
    if (counters) {
        uint64_t* dst = (uint64_t*)counters;
        for (auto& t : tl) {
            const uint64_t* src = (const uint64_t*)&t;
            for (size_t i = 0; i < sizeof(PtoCounters) / 8; i++) dst[i] += src[i];
        }
    }
    return 0;
}

// ---- known-answer hooks ----------------------------------------------------------------------
u32 pto_pcg_hash(u32 x) { return pcg_hash(x); }
// First `n` rand_f32() values of pixel (px,py) at `frame`.
void pto_rand_sequence(u32 px, u32 py, u32 frame, u32 n, float* out) {
    Ctx c;
    c.rng = 0;
    init_rng(c, px, py, frame);
    for (u32 i = 0; i < n; i++) out[i] = rand_f32(c);
}
// op: 0 sin, 1 cos, 2 asin, 3 acos, 4 log, 5 log2, 6 exp2, 7 atan2(a,b), 8 pow(a,b), 9 min, 10 max,
// 11 sqrt, 12 a/b, 13 fma(a,b,a)
void pto_math(int op, u32 n, const float* a, const float* b, float* out) {
    for (u32 i = 0; i < n; i++) {
        switch (op) {
            case 0: out[i] = dm::sin_(a[i]); break;
            case 1: out[i] = dm::cos_(a[i]); break;
            case 2: out[i] = dm::asin_(a[i]); break;
            case 3: out[i] = dm::acos_(a[i]); break;
            case 4: out[i] = dm::log_(a[i]); break;
            case 5: out[i] = dm::log2_(a[i]); break;
            case 6: out[i] = dm::exp2_(a[i]); break;
            case 7: out[i] = dm::atan2_(a[i], b[i]); break;
            case 8: out[i] = dm::pow_(a[i], b[i]); break;
            case 9: out[i] = dm::fmin_(a[i], b[i]); break;
            case 10: out[i] = dm::fmax_(a[i], b[i]); break;
            case 11: out[i] = sqrtf(a[i]); break;
            case 12: out[i] = dm::div_(a[i], b[i]); break;
            case 13: out[i] = fmaf(a[i], b[i], a[i]); break;
            case 14: out[i] = dm::rcp_(a[i]); break;
            default: out[i] = 0.0f;
        }
    }
}
void pto_unpack2x16float(u32 n, const float* packed, float* out_xy) {
    for (u32 i = 0; i < n; i++) { V2 v = unpack2x16float(packed[i]); out_xy[2 * i] = v.x; out_xy[2 * i + 1] = v.y; }
}
// One intersect_figure call per ray.  rays: n x 6 (origin, direction).  out: n x 10 floats
// (hit flag, t, position xyz, normal xyz, uv xy).
void pto_intersect(const PtoScene* sc, const GpuCamera* cam, u32 fig_idx, u32 n, const float* rays, float* out) {
    Ctx c;
    init_ctx(c, sc, cam, nullptr);
    for (u32 i = 0; i < n; i++) {
        Ray r{v3(rays + 6 * i), v3(rays + 6 * i + 3)};
        Hit h = intersect_figure(c, r, fig_idx);
        float* o = out + 10 * i;
        o[0] = h.hit ? 1.0f : 0.0f; o[1] = h.t;
        o[2] = h.position.x; o[3] = h.position.y; o[4] = h.position.z;
        o[5] = h.normal.x; o[6] = h.normal.y; o[7] = h.normal.z;
        o[8] = h.uv.x; o[9] = h.uv.y;
    }
}
// Closest hit / shadow query per ray through trace_bvh / trace_shadow.
// out: n x 3 (figure id as float bits via u32 array would lose nothing; ids and flags are separate).
void pto_trace(const PtoScene* sc, const GpuCamera* cam, u32 n, const float* rays, const float* max_t,
               u32* hit_id, float* hit_t, u32* shadow) {
    Ctx c;
    init_ctx(c, sc, cam, nullptr);
    for (u32 i = 0; i < n; i++) {
        Ray r{v3(rays + 6 * i), v3(rays + 6 * i + 3)};
        Hit h = trace_bvh(c, r);
        hit_id[i] = h.hit ? h.figure_idx : 0xFFFFFFFFu;
        hit_t[i] = h.hit ? h.t : MAX_T;
        if (shadow && max_t) shadow[i] = trace_shadow(c, r, max_t[i]) ? 1u : 0u;
    }
}
// Tonemap + unorm8 for n rgb triples.
void pto_tonemap(const GpuCamera* cam, u32 n, const float* rgb, uint8_t* out_rgba8) {
    for (u32 i = 0; i < n; i++) {
        V3 col = apply_tonemap(*cam, v3(rgb + 3 * i));
        out_rgba8[4 * i] = unorm8(col.x); out_rgba8[4 * i + 1] = unorm8(col.y); out_rgba8[4 * i + 2] = unorm8(col.z); out_rgba8[4 * i + 3] = 255;
    }
}
// eval_brdf for n (wo, wi, n) triples with one material: in n x 9, out n x 3.
void pto_eval_brdf(const GpuMaterial* gm, u32 n, const float* in, float* out) {
    PtoScene sc{};
    GpuCamera cam{};
    Ctx c;
    c.sc = &sc; c.cam = &cam; c.cnt = nullptr; c.rng = 0;
    Mat mat{v3(gm->base_color), gm->metallic, v3(gm->emission), gm->roughness, gm->emission_strength, gm->ior, gm->transmission, gm->texture_id};
    for (u32 i = 0; i < n; i++) {
        V3 r = eval_brdf(c, v3(in + 9 * i), v3(in + 9 * i + 3), v3(in + 9 * i + 6), mat);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
int pto_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// ---- post-process pass (src/shaders/wgsl/post_process.wgsl) ------------------------------------------
// params = the reference's 16-word uniform (post_process.wgsl:5-16, app/state.rs:448-466): width, height,
// effect_count, oil_radius, effects[8], comic_levels, 3 pad words.  accum = w*h rgba f32 (the accumulation
// buffer, read untonemapped exactly as the reference's pass does); out = w*h rgba8 (the output texture).
// Same numerical contract as the trace: dot = fma chain from the low index, everything else unfused and left
// to right, vector / scalar = vector * rcp(scalar), a literal divisor C as * (1.0f / C), length = sqrt(dot).
}  // extern "C"
namespace {
struct PostImg { const float* a; i32 w, h; };
inline V3 pp_read(const PostImg& im, i32 x, i32 y) { return v3(im.a + 4 * ((size_t)y * im.w + x)); }  // :32-35
inline V3 pp_read_clamped(const PostImg& im, i32 x, i32 y) {                                          // :37-41
    i32 cx = x < 0 ? 0 : (x > im.w - 1 ? im.w - 1 : x), cy = y < 0 ? 0 : (y > im.h - 1 ? im.h - 1 : y);
    return pp_read(im, cx, cy);
}
inline V3 pp_fxaa(const PostImg& im, i32 x, i32 y) {  // :105-135
    const V3 lw = v3(0.299f, 0.587f, 0.114f);
    V3 center = pp_read(im, x, y);
    float luma_c = dot(center, lw);
    V3 cn = pp_read_clamped(im, x, y - 1), cw = pp_read_clamped(im, x - 1, y), ce = pp_read_clamped(im, x + 1, y),
       cs = pp_read_clamped(im, x, y + 1);
    float luma_n = dot(cn, lw), luma_w = dot(cw, lw), luma_e = dot(ce, lw), luma_s = dot(cs, lw);
    float range = fmax_(fmax_(luma_n, luma_s), fmax_(luma_w, fmax_(luma_e, luma_c))) -
                  fmin_(fmin_(luma_n, luma_s), fmin_(luma_w, fmin_(luma_e, luma_c)));
    if (range < fmax_(0.0312f, luma_c * 0.125f)) return center;
    float horizontal = fabsf(luma_n + luma_s - 2.0f * luma_c);
    float vertical = fabsf(luma_w + luma_e - 2.0f * luma_c);
    if (horizontal > vertical) return (cn + cs + center) * (1.0f / 3.0f);
    return (cw + ce + center) * (1.0f / 3.0f);
}
inline V3 pp_oil(const PostImg& im, i32 x, i32 y, i32 radius) {  // :138-158
    V3 total = v3(0.0f, 0.0f, 0.0f);
    i32 count = 0;
    for (i32 dx = -radius; dx <= radius; dx++)
        for (i32 dy = -radius; dy <= radius; dy++) {
            i32 sx = x + dx, sy = y + dy;
            if (sx >= 0 && sx < im.w && sy >= 0 && sy < im.h) {
                total = total + pp_read(im, sx, sy);
                count += 1;
            }
        }
    if (count > 0) return vdivs(total, (float)count);
    return pp_read(im, x, y);
}
inline V3 pp_comic(const PostImg& im, i32 x, i32 y, V3 color, u32 comic_levels) {  // :161-186
    float levels = (float)comic_levels;
    V3 q = color * levels + v3(0.5f, 0.5f, 0.5f);
    V3 quantized = vdivs(v3(floorf(q.x), floorf(q.y), floorf(q.z)), levels);
    static const float sxk[9] = {-1.0f, 0.0f, 1.0f, -2.0f, 0.0f, 2.0f, -1.0f, 0.0f, 1.0f};
    static const float syk[9] = {-1.0f, -2.0f, -1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 2.0f, 1.0f};
    V3 gx = v3(0.0f, 0.0f, 0.0f), gy = v3(0.0f, 0.0f, 0.0f);
    int ki = 0;
    for (i32 dy = -1; dy <= 1; dy++)
        for (i32 dx = -1; dx <= 1; dx++) {
            V3 sc = pp_read_clamped(im, x + dx, y + dy);
            gx = gx + sc * sxk[ki];
            gy = gy + sc * syk[ki];
            ki++;
        }
    float edge = length(gx) + length(gy);
    float edge_mask = edge > 0.3f ? 0.0f : 1.0f;
    return quantized * edge_mask;
}
inline V3 pp_casting(const PostImg& im, i32 x, i32 y) {  // :189-211
    static const float k[9] = {-2.0f, -1.0f, 0.0f, -1.0f, 1.0f, 1.0f, 0.0f, 1.0f, 2.0f};
    V3 r = v3(0.0f, 0.0f, 0.0f);
    int ki = 0;
    for (i32 dy = -1; dy <= 1; dy++)
        for (i32 dx = -1; dx <= 1; dx++) {
            r = r + pp_read_clamped(im, x + dx, y + dy) * k[ki];
            ki++;
        }
    r = r * 0.5f + v3(0.5f, 0.5f, 0.5f);
    return v3(clampf(r.x, 0.0f, 1.0f), clampf(r.y, 0.0f, 1.0f), clampf(r.z, 0.0f, 1.0f));
}
inline V3 pp_effect(const PostImg& im, V3 color, i32 x, i32 y, u32 effect, const u32* params) {  // :50-86
    switch (effect) {
        case 1u: return v3(1.0f, 1.0f, 1.0f) - color;
        case 2u: {
            float r = dot(color, v3(0.393f, 0.769f, 0.189f)), g = dot(color, v3(0.349f, 0.686f, 0.168f)),
                  b = dot(color, v3(0.272f, 0.534f, 0.131f));
            return v3(clampf(r, 0.0f, 1.0f), clampf(g, 0.0f, 1.0f), clampf(b, 0.0f, 1.0f));
        }
        case 3u: { float lum = dot(color, v3(0.2126f, 0.7152f, 0.0722f)); return v3(lum, lum, lum); }
        case 4u: return pp_fxaa(im, x, y);
        case 5u: return pp_oil(im, x, y, (i32)params[3]);
        case 6u: { float lum = dot(color, v3(0.2126f, 0.7152f, 0.0722f)); return lum > 0.5f ? v3(1.0f, 1.0f, 1.0f) : v3(0.0f, 0.0f, 0.0f); }
        case 7u: return pp_comic(im, x, y, color, params[12]);
        case 8u: return pp_casting(im, x, y);
        default: return color;
    }
}
}  // namespace
extern "C" {
// Returns 0, or -1 when params are not usable (size mismatch, more than 8 effects).
int pto_post_process(const float* accum, u32 w, u32 h, const u32* params, uint8_t* out_rgba8, int threads) {
    if (!accum || !params || !out_rgba8 || params[0] != w || params[1] != h || params[2] > 8u) return -1;
    PostImg im{accum, (i32)w, (i32)h};
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for (i32 y = 0; y < (i32)h; y++)
        for (i32 x = 0; x < (i32)w; x++) {  // :88-102
            V3 result = pp_read(im, x, y);
            for (u32 i = 0; i < params[2]; i++) result = pp_effect(im, result, x, y, params[4 + i], params);
            size_t o = 4 * ((size_t)y * w + x);
            out_rgba8[o] = unorm8(result.x); out_rgba8[o + 1] = unorm8(result.y); out_rgba8[o + 2] = unorm8(result.z); out_rgba8[o + 3] = 255;
        }
    return 0;
}
}
