// Ray / primitive intersection for the 17 figure types (reference: src/shaders/wgsl/figures/*.wgsl).
//
// Two passes instead of the reference's one: during traversal only the hit distance of each candidate
// is computed (FULL = false); the winner alone is then resolved to position, normal and UV
// (FULL = true, or finish_* for ray-marched figures, which reuse the marched distance).  The distance
// is computed by the same instructions in both passes, so the split changes no result.
#pragma once
#include "dmath_dev.h"
#include "dscene.h"

struct Ray { V3 o, d; };
struct HitFull { V3 pos, normal; V2 uv; };

PT_DEV V2 sphere_uv(V3 d) {  // sphere.wgsl:34-37
    V2 r;
    r.x = 0.5f + dm::atan2_(d.z, d.x) * (1.0f / PT_TWO_PI);
    r.y = 0.5f - dm::asin_(clampf(d.y, -1.0f, 1.0f)) * (1.0f / PT_PI);
    return r;
}
struct Onb { V3 u, v, n; };
PT_DEV Onb build_onb(V3 n) {  // utils.wgsl:6-13
    float s = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = dm::div_(-1.0f, s + n.z);
    float b = n.x * n.y * a;
    Onb o;
    o.u = v3(1.0f + s * n.x * n.x * a, s * b, -s * n.x);
    o.v = v3(b, s + n.y * n.y * a, -n.y);
    o.n = n;
    return o;
}
PT_DEV V3 onb_mul(const Onb& o, V3 l) {
    return v3(dm::fma_(o.n.x, l.z, dm::fma_(o.v.x, l.y, o.u.x * l.x)), dm::fma_(o.n.y, l.z, dm::fma_(o.v.y, l.y, o.u.y * l.x)),
              dm::fma_(o.n.z, l.z, dm::fma_(o.v.z, l.y, o.u.z * l.x)));
}

template <bool FULL>
PT_DEV bool isect_sphere(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // sphere.wgsl:4-40
    V3 pos = v3p(g.p);
    V3 oc = ray.o - pos;
    float half_b = dot(oc, ray.d);
    float c = dot(oc, oc) - g.p[4];  // radius * radius, multiplied at upload
    float disc = dm::fma_(half_b, half_b, -c);
    if (disc < 0.0f) return false;
    float sqrtd = dm::sqrt_(disc);
    float t = -half_b - sqrtd;
    if (t < PT_EPS) {
        t = -half_b + sqrtd;
        if (t < PT_EPS) return false;
    }
    t_out = t;
    if (FULL) {
        h->pos = madd(ray.o, ray.d, t);
        h->normal = normalize(h->pos - pos);
        if (need_uv) h->uv = sphere_uv(h->normal);
    }
    return true;
}
template <bool FULL>
PT_DEV bool isect_plane(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // plane.wgsl:4-30
    V3 pos = v3p(g.p), n = v3p(g.p + 3);
    float denom = dot(n, ray.d);
    if (dm::abs_(denom) < PT_EPS) return false;
    float t = dm::div_(dot(pos - ray.o, n), denom);
    if (t < PT_EPS) return false;
    t_out = t;
    if (FULL) {
        h->pos = madd(ray.o, ray.d, t);
        h->normal = denom < 0.0f ? n : -n;
        if (need_uv) {
            Onb onb = build_onb(h->normal);
            V3 local = h->pos - pos;
            h->uv.x = dot(local, onb.u) * 0.25f;
            h->uv.y = dot(local, onb.v) * 0.25f;
        }
    }
    return true;
}
template <bool FULL>
PT_DEV bool isect_cube(const Ray& ray, V3 inv_dir, const DPrim& g, float& t_out, HitFull* h) {  // cube.wgsl:4-60
    V3 pos = v3p(g.p);
    float r = g.p[3];
    V3 bmin = v3p(g.p + 4), bmax = v3p(g.p + 7);  // pos -+ (r, r, r), computed at upload
    V3 t1 = (bmin - ray.o) * inv_dir, t2 = (bmax - ray.o) * inv_dir;
    V3 tmn = vmin(t1, t2), tmx = vmax(t1, t2);
    float t_near = dm::fmax_(dm::fmax_(tmn.x, tmn.y), tmn.z);
    float t_far = dm::fmin_(dm::fmin_(tmx.x, tmx.y), tmx.z);
    if (t_near > t_far || t_far < PT_EPS) return false;
    float t = t_near;
    if (t < PT_EPS) {
        t = t_far;
        if (t < PT_EPS) return false;
    }
    t_out = t;
    if (FULL) {
        h->pos = madd(ray.o, ray.d, t);
        V3 p = vdivs(h->pos - pos, r);
        V3 ap = vabs(p);
        if (ap.x > ap.y && ap.x > ap.z) h->normal = v3(signf(p.x), 0.0f, 0.0f);
        else if (ap.y > ap.z) h->normal = v3(0.0f, signf(p.y), 0.0f);
        else h->normal = v3(0.0f, 0.0f, signf(p.z));
        if (dm::abs_(h->normal.x) > 0.5f) { h->uv.x = (p.y + 1.0f) * 0.5f; h->uv.y = (p.z + 1.0f) * 0.5f; }
        else if (dm::abs_(h->normal.y) > 0.5f) { h->uv.x = (p.x + 1.0f) * 0.5f; h->uv.y = (p.z + 1.0f) * 0.5f; }
        else { h->uv.x = (p.x + 1.0f) * 0.5f; h->uv.y = (p.y + 1.0f) * 0.5f; }
    }
    return true;
}
template <bool FULL>
PT_DEV bool isect_cylinder(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // cylinder.wgsl:4-69
    V3 pos = v3p(g.p), axis = v3p(g.p + 3);
    float radius = g.p[6], height = g.p[7];
    float half_h = height * 0.5f;
    V3 oc = ray.o - pos;
    V3 d_perp = ray.d - axis * dot(ray.d, axis);
    V3 oc_perp = oc - axis * dot(oc, axis);
    float a = dot(d_perp, d_perp);
    float half_b = dot(d_perp, oc_perp);
    float c = dot(oc_perp, oc_perp) - radius * radius;
    float disc = dm::fma_(half_b, half_b, -(a * c));
    if (disc < 0.0f) return false;
    float sqrtd = dm::sqrt_(disc);
    float inv_a = dm::rcp_(a);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float t = i == 0 ? (-half_b - sqrtd) * inv_a : (-half_b + sqrtd) * inv_a;
        if (t < PT_EPS) continue;
        V3 p = madd(ray.o, ray.d, t);
        float proj = dot(p - pos, axis);
        if (dm::abs_(proj) <= half_h) {
            t_out = t;
            if (FULL) {
                h->pos = p;
                h->normal = normalize((p - pos) - axis * proj);
                if (need_uv) {
                    float angle = dm::atan2_(h->normal.z, h->normal.x);
                    h->uv.x = angle * (1.0f / PT_TWO_PI) + 0.5f;
                    h->uv.y = dm::div_(proj + half_h, height);
                }
            }
            return true;
        }
    }
    bool hit = false;
    float best = PT_MAX_T;
    float d_axis = dot(ray.d, axis);
    if (dm::abs_(d_axis) > PT_EPS) {
#pragma unroll
        for (int si = 0; si < 2; si++) {
            float cap_sign = si == 1 ? -1.0f : 1.0f;
            V3 cap_center = pos + axis * half_h * cap_sign;
            float t = dm::div_(dot(cap_center - ray.o, axis * cap_sign), d_axis * cap_sign);
            if (t > PT_EPS && t < best) {
                V3 p = madd(ray.o, ray.d, t);
                V3 off = p - cap_center;
                if (dot(off, off) <= radius * radius) {
                    hit = true;
                    best = t;
                    if (FULL) {
                        h->pos = p;
                        h->normal = axis * cap_sign;
                        h->uv.x = (dm::div_(off.x, radius) + 1.0f) * 0.5f;
                        h->uv.y = (dm::div_(off.z, radius) + 1.0f) * 0.5f;
                    }
                }
            }
        }
    }
    t_out = best;
    return hit;
}
template <bool FULL>
PT_DEV bool isect_cone(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // cone.wgsl:4-80
    V3 pos = v3p(g.p), axis = v3p(g.p + 3);
    float radius = g.p[6], height = g.p[7], tan2 = g.p[8];
    V3 apex = pos + axis * height;
    V3 oc = ray.o - apex;
    float d_dot_v = dot(ray.d, axis);
    float oc_dot_v = dot(oc, axis);
    float cos2 = dm::rcp_(1.0f + tan2);
    float a_c = dm::fma_(d_dot_v, d_dot_v, -(cos2 * dot(ray.d, ray.d)));
    float b_c = dm::fma_(d_dot_v, oc_dot_v, -(cos2 * dot(ray.d, oc)));
    float c_c = dm::fma_(oc_dot_v, oc_dot_v, -(cos2 * dot(oc, oc)));
    float disc = dm::fma_(b_c, b_c, -(a_c * c_c));
    if (disc < 0.0f) return false;
    float sqrtd = dm::sqrt_(disc);
    float inv_a = dm::rcp_(a_c);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float t = i == 0 ? (-b_c - sqrtd) * inv_a : (-b_c + sqrtd) * inv_a;
        if (t < PT_EPS) continue;
        V3 p = madd(ray.o, ray.d, t);
        float proj = dot(p - apex, axis);
        if (proj >= -height && proj <= 0.0f) {
            t_out = t;
            if (FULL) {
                h->pos = p;
                V3 to_p = normalize(p - apex);
                float n_proj = dot(to_p, axis);
                h->normal = normalize(to_p - axis * n_proj * (1.0f + tan2));
                if (need_uv) {
                    float angle = dm::atan2_(h->normal.z, h->normal.x);
                    h->uv.x = angle * (1.0f / PT_TWO_PI) + 0.5f;
                    h->uv.y = dm::div_(-proj, height);
                }
            }
            return true;
        }
    }
    float d_axis2 = dot(ray.d, axis);
    if (dm::abs_(d_axis2) > PT_EPS) {
        float t = dm::div_(dot(pos - ray.o, axis), d_axis2);
        if (t > PT_EPS && t < PT_MAX_T) {
            V3 p = madd(ray.o, ray.d, t);
            V3 off = p - pos;
            if (dot(off, off) <= radius * radius) {
                t_out = t;
                if (FULL) {
                    h->pos = p;
                    h->normal = -axis;
                    h->uv.x = (dm::div_(off.x, radius) + 1.0f) * 0.5f;
                    h->uv.y = (dm::div_(off.z, radius) + 1.0f) * 0.5f;
                }
                return true;
            }
        }
    }
    return false;
}
template <bool FULL>
PT_DEV bool isect_disc(const Ray& ray, const DPrim& g, float& t_out, HitFull* h) {  // disc.wgsl:4-35
    V3 pos = v3p(g.p), n = v3p(g.p + 3);
    float radius = g.p[6];
    float denom = dot(n, ray.d);
    if (dm::abs_(denom) < PT_EPS) return false;
    float t = dm::div_(dot(pos - ray.o, n), denom);
    if (t < PT_EPS) return false;
    V3 p = madd(ray.o, ray.d, t);
    V3 off = p - pos;
    if (dot(off, off) > g.p[7]) return false;  // radius * radius, multiplied at upload
    t_out = t;
    if (FULL) {
        h->pos = p;
        h->normal = denom < 0.0f ? n : -n;
        h->uv.x = (dm::div_(off.x, radius) + 1.0f) * 0.5f;
        h->uv.y = (dm::div_(off.z, radius) + 1.0f) * 0.5f;
    }
    return true;
}
// Moller-Trumbore core of triangle.wgsl:9-36, pyramid.wgsl:38-61, tetrahedron.wgsl:36-59.
PT_DEV bool mt_core(const Ray& ray, V3 v0, V3 e1, V3 e2, float& t, float& u, float& v) {
    V3 hh = cross(ray.d, e2);
    float a = dot(e1, hh);
    if (dm::abs_(a) < PT_EPS) return false;
    float fi = dm::rcp_(a);
    V3 s = ray.o - v0;
    u = fi * dot(s, hh);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, e1);
    v = fi * dot(ray.d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = fi * dot(e2, q);
    return true;
}
PT_DEV float half_to_float(uint32_t hb) {  // WGSL unpack2x16float, exact
    uint32_t sign = (hb & 0x8000u) << 16, e = (hb >> 10) & 0x1Fu, m = hb & 0x3FFu;
    if (e == 0u) {
        if (m == 0u) return dm::u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f;
        return sign ? -v : v;
    }
    if (e == 31u) return dm::u2f(sign | 0x7F800000u | (m << 13));
    return dm::u2f(sign | ((e + 112u) << 23) | (m << 13));
}
template <bool FULL>
PT_DEV bool isect_triangle(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv, const DShade* sh) {  // triangle.wgsl:4-53
    V3 v0 = v3p(g.p), e1 = v3p(g.p + 3), e2 = v3p(g.p + 6);
    float t, u, v;
    if (!mt_core(ray, v0, e1, e2, t, u, v)) return false;
    if (t < PT_EPS) return false;
    t_out = t;
    if (FULL) {
        h->pos = madd(ray.o, ray.d, t);
        h->normal = normalize(cross(e1, e2));
        if (dot(h->normal, ray.d) > 0.0f) h->normal = -h->normal;
        if (need_uv) {
            uint32_t a = sh->uv0, b = sh->uv1, c = sh->uv2;
            float w0 = 1.0f - u - v;
            h->uv.x = dm::fma_(v, half_to_float(c & 0xFFFFu), dm::fma_(u, half_to_float(b & 0xFFFFu), w0 * half_to_float(a & 0xFFFFu)));
            h->uv.y = dm::fma_(v, half_to_float(c >> 16), dm::fma_(u, half_to_float(b >> 16), w0 * half_to_float(a >> 16)));
        }
    }
    return true;
}
// Keep-closest face test of pyramid.wgsl:35-70 / tetrahedron.wgsl:33-68 (normal not ray-facing, uv = (u,v)).
template <bool FULL>
PT_DEV void face_test(const Ray& ray, V3 v0, V3 v1, V3 v2, bool& hit, float& best, HitFull* h, uint32_t face, uint32_t& sub) {
    V3 e1 = v1 - v0, e2 = v2 - v0;
    float t, u, v;
    if (!mt_core(ray, v0, e1, e2, t, u, v)) return;
    if (t > PT_EPS && t < best) {
        hit = true;
        best = t;
        sub = face;
        if (FULL) {
            h->pos = madd(ray.o, ray.d, t);
            h->normal = normalize(cross(e1, e2));
            h->uv.x = u;
            h->uv.y = v;
        }
    }
}
// Vertices of face `f` of the pyramid (pyramid.wgsl:13-29): 0..3 sides, 4..5 base.
PT_DEV void pyramid_face(const DPrim& g, uint32_t f, V3& a, V3& b, V3& c) {
    V3 base = v3p(g.p);
    float r = g.p[3], hh = g.p[4];
    V3 apex = base + v3(0.0f, hh, 0.0f);
    V3 b0 = base + v3(-r, 0.0f, -r), b1 = base + v3(r, 0.0f, -r), b2 = base + v3(r, 0.0f, r), b3 = base + v3(-r, 0.0f, r);
    switch (f) {
        case 0: a = apex; b = b0; c = b1; break;
        case 1: a = apex; b = b1; c = b2; break;
        case 2: a = apex; b = b2; c = b3; break;
        case 3: a = apex; b = b3; c = b0; break;
        case 4: a = b0; b = b2; c = b1; break;
        default: a = b0; b = b3; c = b2; break;
    }
}
template <bool FULL>
PT_DEV bool isect_pyramid(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, uint32_t& sub) {  // pyramid.wgsl:5-32
    bool hit = false;
    float best = PT_MAX_T;
    for (uint32_t f = 0; f < 6u; f++) {
        V3 a, b, c;
        pyramid_face(g, f, a, b, c);
        face_test<FULL>(ray, a, b, c, hit, best, h, f, sub);
    }
    t_out = best;
    return hit;
}
// Vertices of face `f` of the tetrahedron (tetrahedron.wgsl:14-28).
PT_DEV void tetra_face(const DPrim& g, uint32_t f, V3& fa, V3& fb, V3& fc) {
    V3 pos = v3p(g.p);
    float r = g.p[3];
    float a = r * 0.942809f, b = r * 0.333333f, c = r * 0.816497f;
    V3 v0 = pos + v3(0.0f, r, 0.0f);
    V3 v1 = pos + v3(a, -b, -c * 0.5f);
    V3 v2 = pos + v3(-a, -b, -c * 0.5f);
    V3 v3_ = pos + v3(0.0f, -b, c);
    switch (f) {
        case 0: fa = v0; fb = v1; fc = v2; break;
        case 1: fa = v0; fb = v2; fc = v3_; break;
        case 2: fa = v0; fb = v3_; fc = v1; break;
        default: fa = v1; fb = v3_; fc = v2; break;
    }
}
template <bool FULL>
PT_DEV bool isect_tetrahedron(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, uint32_t& sub) {  // tetrahedron.wgsl:4-31
    bool hit = false;
    float best = PT_MAX_T;
    for (uint32_t f = 0; f < 4u; f++) {
        V3 a, b, c;
        tetra_face(g, f, a, b, c);
        face_test<FULL>(ray, a, b, c, hit, best, h, f, sub);
    }
    t_out = best;
    return hit;
}
template <bool FULL>
PT_DEV bool isect_ellipsoid(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // ellipsoid.wgsl:5-52
    V3 pos = v3p(g.p);
    V3 inv_radii = v3p(g.p + 6);  // 1 / (radius, max(height, radius), max(radius2, radius)), computed at upload
    V3 oc = (ray.o - pos) * inv_radii;
    V3 dir = ray.d * inv_radii;
    float a = dot(dir, dir);
    float half_b = dot(oc, dir);
    float c = dot(oc, oc) - 1.0f;
    float disc = dm::fma_(half_b, half_b, -(a * c));
    if (disc < 0.0f) return false;
    float sqrtd = dm::sqrt_(disc);
    float inv_a = dm::rcp_(a);
    float t = (-half_b - sqrtd) * inv_a;
    if (t < PT_EPS) {
        t = (-half_b + sqrtd) * inv_a;
        if (t < PT_EPS) return false;
    }
    t_out = t;
    if (FULL) {
        h->pos = madd(ray.o, ray.d, t);
        V3 local = h->pos - pos;
        h->normal = normalize(local * inv_radii * inv_radii);
        if (need_uv) h->uv = sphere_uv(normalize(local * inv_radii));
    }
    return true;
}
template <bool FULL>
PT_DEV bool isect_paraboloid(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // paraboloid.wgsl:4-67
    V3 pos = v3p(g.p);
    float r = g.p[3], hh = g.p[4];
    V3 oc = ray.o - pos, d = ray.d;
    float a = dot2(d.x, d.z, d.x, d.z);
    float b = 2.0f * dot2(oc.x, oc.z, d.x, d.z) - r * d.y;
    float c = dot2(oc.x, oc.z, oc.x, oc.z) - r * oc.y;
    float disc = dm::fma_(b, b, -(4.0f * a * c));
    if (disc < 0.0f) return false;
    float sqrtd = dm::sqrt_(disc);
    float inv_2a = dm::div_(0.5f, a);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float t = i == 0 ? (-b - sqrtd) * inv_2a : (-b + sqrtd) * inv_2a;
        if (t < PT_EPS) continue;
        V3 p = madd(ray.o, d, t);
        float local_y = p.y - pos.y;
        if (local_y >= 0.0f && local_y <= hh) {
            t_out = t;
            if (FULL) {
                h->pos = p;
                V3 local = p - pos;
                h->normal = normalize(v3(2.0f * local.x, -r, 2.0f * local.z));
                if (need_uv) {
                    float angle = dm::atan2_(local.z, local.x);
                    h->uv.x = angle * (1.0f / PT_TWO_PI) + 0.5f;
                    h->uv.y = dm::div_(local_y, hh);
                }
            }
            return true;
        }
    }
    float d_y = d.y;
    if (dm::abs_(d_y) > PT_EPS) {
        float cap_y = pos.y + hh;
        float t = dm::div_(cap_y - ray.o.y, d_y);
        if (t > PT_EPS) {
            V3 p = madd(ray.o, d, t);
            V3 local = p - pos;
            float dist2 = dot2(local.x, local.z, local.x, local.z);
            if (dist2 <= r * hh) {
                t_out = t;
                if (FULL) {
                    h->pos = p;
                    h->normal = v3(0.0f, 1.0f, 0.0f);
                    float s = dm::sqrt_(r * hh);
                    h->uv.x = dm::div_(local.x, s) * 0.5f + 0.5f;
                    h->uv.y = dm::div_(local.z, s) * 0.5f + 0.5f;
                }
                return true;
            }
        }
    }
    return false;
}
template <bool FULL>
PT_DEV bool isect_hyperboloid(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // hyperboloid.wgsl:4-68
    V3 pos = v3p(g.p);
    float r = g.p[3], height = g.p[4];
    V3 oc = ray.o - pos, d = ray.d;
    float half_h = height * 0.5f;
    float inv_r2 = dm::rcp_(r * r);
    float a = (dot2(d.x, d.z, d.x, d.z) - d.y * d.y) * inv_r2;
    float b = 2.0f * (dot2(oc.x, oc.z, d.x, d.z) - oc.y * d.y) * inv_r2;
    float c = (dot2(oc.x, oc.z, oc.x, oc.z) - oc.y * oc.y) * inv_r2 - 1.0f;
    float disc = dm::fma_(b, b, -(4.0f * a * c));
    bool hit = false;
    float best = PT_MAX_T;
    if (disc >= 0.0f) {
        float sqrtd = dm::sqrt_(disc);
        float inv_2a = dm::div_(0.5f, a);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            float t = i == 0 ? (-b - sqrtd) * inv_2a : (-b + sqrtd) * inv_2a;
            if (t < PT_EPS || t >= best) continue;
            V3 p = madd(ray.o, d, t);
            float local_y = p.y - pos.y;
            if (dm::abs_(local_y) <= half_h) {
                hit = true;
                best = t;
                if (FULL) {
                    h->pos = p;
                    V3 local = p - pos;
                    h->normal = normalize(v3(2.0f * local.x, -2.0f * local.y, 2.0f * local.z) * inv_r2);
                    if (need_uv) {
                        float angle = dm::atan2_(local.z, local.x);
                        h->uv.x = angle * (1.0f / PT_TWO_PI) + 0.5f;
                        h->uv.y = dm::div_(local_y + half_h, height);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int cap = 0; cap < 2; cap++) {
        float cap_y = pos.y + (cap == 1 ? half_h : -half_h);
        if (dm::abs_(d.y) > PT_EPS) {
            float t = dm::div_(cap_y - ray.o.y, d.y);
            if (t > PT_EPS && t < best) {
                V3 p = madd(ray.o, d, t);
                V3 local = p - pos;
                float dist2 = dot2(local.x, local.z, local.x, local.z);
                float cap_r2 = r * r * (1.0f + (half_h * half_h) * inv_r2);
                if (dist2 <= cap_r2) {
                    hit = true;
                    best = t;
                    if (FULL) {
                        h->pos = p;
                        h->normal = cap == 1 ? v3(0.0f, 1.0f, 0.0f) : v3(0.0f, -1.0f, 0.0f);
                        float s = dm::sqrt_(cap_r2);
                        h->uv.x = dm::div_(local.x, s) * 0.5f + 0.5f;
                        h->uv.y = dm::div_(local.z, s) * 0.5f + 0.5f;
                    }
                }
            }
        }
    }
    t_out = best;
    return hit;
}
template <bool FULL>
PT_DEV bool isect_skybox(const Ray& ray, const DPrim& g, float& t_out, HitFull* h, bool need_uv) {  // skybox.wgsl:30-62
    V3 pos = v3p(g.p);
    float radius = g.p[3];
    V3 oc = ray.o - pos;
    float half_b = dot(oc, ray.d);
    float c = dot(oc, oc) - radius * radius;
    float disc = dm::fma_(half_b, half_b, -c);
    if (disc < 0.0f) return false;
    float t = -half_b + dm::sqrt_(disc);
    if (t < PT_EPS) return false;
    t_out = t;
    if (FULL) {
        h->pos = madd(ray.o, ray.d, t);
        V3 d = normalize(h->pos - pos);
        h->normal = -d;
        if (need_uv) h->uv = sphere_uv(d);
    }
    return true;
}

// ---- ray-marched figures -------------------------------------------------------------------------
PT_DEV float sdf_torus(V3 p, float major_r, float minor_r) {  // torus.wgsl:5-8
    float qx = dm::sqrt_(dot2(p.x, p.z, p.x, p.z)) - major_r;
    return dm::sqrt_(dot2(qx, p.y, qx, p.y)) - minor_r;
}
PT_DEV float sdf_mebius(V3 p, float R) {  // mebius.wgsl:5-36
    float w = R * 0.3f;
    float angle = dm::atan2_(p.z, p.x);
    float sa, ca;
    dm::sincos_(angle, &sa, &ca);
    V3 center = v3(R * ca, 0.0f, R * sa);
    V3 radial = normalize(v3(ca, 0.0f, sa));
    float sh, ch;
    dm::sincos_(angle * 0.5f, &sh, &ch);
    V3 strip_up = v3(0.0f, 1.0f, 0.0f) * ch + radial * sh;
    V3 dp = p - center;
    float v = dot(dp, strip_up);
    float u = length(dp - strip_up * v);
    float du = dm::abs_(u) - w * 0.15f;
    float dv = dm::abs_(v) - w;
    float mx = dm::fmax_(du, 0.0f), my = dm::fmax_(dv, 0.0f);
    return dm::sqrt_(dot2(mx, my, mx, my)) + dm::fmin_(dm::fmax_(du, dv), 0.0f);
}
// One iteration of the Mandelbulb's w -> w^power + p with its running derivative (mandelbulb.wgsl:11-23): m = |w|^2 on
// entry, dz is updated.  The two pow calls and the two sincos calls are evaluated side by side (dm2: element for element
// the scalar functions' operations, so the same bits).
PT_DEV V3 mandelbulb_iterate(V3 w, V3 p, float m, float power, float& dz) {
    const float r = dm::sqrt_(m);
    const f2 pw = dm2::pow2(dm2::mk(m, r), dm2::mk((power - 1.0f) * 0.5f, power));  // pow(m, (power - 1) / 2), pow(r, power)
    dz = dm::fma_(power * pw.x, dz, 1.0f);
    const float b = power * dm::acos_(clampf(dm::div_(w.y, r), -1.0f, 1.0f));
    const float a = power * dm::atan2_(w.x, w.z);
    f2 sn, cs;
    dm2::sincos2(dm2::mk(b, a), &sn, &cs);  // sin / cos of b in .x, of a in .y
    return p + v3(sn.x * sn.y, cs.x, sn.x * cs.y) * pw.y;
}
// The same iteration in the tolerance mode (ptx_set_option "arith_mode" 1; never for camera rays): the builtins' hardware
// forms -- v_sqrt / v_rcp (1 ulp), v_log / v_exp (log2 / exp2, 1 ulp), v_sin / v_cos (argument in revolutions, about 1e-6
// absolute here: |angle| <= power * pi) -- which is what WGSL's accuracy bounds for these builtins are written for, instead of
// the deterministic polynomials: about 80 issue slots instead of 270.  The two angles keep their polynomials (no hardware
// form), with the hardware reciprocal and square root inside.
namespace fastm {
PT_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
PT_DEV float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PT_DEV float log2(float x) { return __builtin_amdgcn_logf(x); }
PT_DEV float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
PT_DEV float acos(float x) {
    const float a = dm::abs_(x);
    const bool big = a > 0.5f;
    const float z = big ? 0.5f * (1.0f - a) : x * x;
    const float s = big ? sqrt(z) : x;
    float r = dm::fma_(s * z, dm::asin_poly(z), s);
    if (big) { r = r + r; return x < 0.0f ? DM_PI - r : r; }
    return DM_PIO2 - r;
}
PT_DEV float atan2(float y, float x) {
    const float ax = dm::abs_(x), ay = dm::abs_(y);
    const float mx = dm::fmax_(ax, ay), mn = dm::fmin_(ax, ay);
    if (mx == 0.0f) return 0.0f;
    const bool big = mn > 0.4142135623730950f * mx;
    const float t = (big ? mn - mx : mn) * rcp(big ? mn + mx : mx);
    const float z = t * t;
    const float p = dm::fma_(z, dm::fma_(z, dm::fma_(z, 8.05374449538e-2f, -1.38776856032e-1f), 1.99777106478e-1f), -3.33329491539e-1f);
    float r = dm::fma_(p * z, t, t);
    if (big) r = DM_PIO4 + r;
    if (ay > ax) r = DM_PIO2 - r;
    if (x < 0.0f) r = DM_PI - r;
    return y < 0.0f ? -r : r;
}
}  // namespace fastm
PT_DEV V3 mandelbulb_iterate_fast(V3 w, V3 p, float m, float power, float& dz) {
    const float r = fastm::sqrt(m);
    const float l = fastm::log2(r);
    const float rp = fastm::exp2(power * l);  // r^power; m^((power - 1) / 2) = r^(power - 1)
    dz = dm::fma_(power * fastm::exp2((power - 1.0f) * l), dz, 1.0f);
    const float k = power * 0.15915494309189535f;  // angles in revolutions
    const float b = k * fastm::acos(clampf(w.y * fastm::rcp(r), -1.0f, 1.0f));
    const float a = k * fastm::atan2(w.x, w.z);
    const float sb = __builtin_amdgcn_sinf(b), cb = __builtin_amdgcn_cosf(b), sa = __builtin_amdgcn_sinf(a), ca = __builtin_amdgcn_cosf(a);
    return p + v3(sb * sa, cb, sb * ca) * rp;
}
PT_DEV float sdf_mandelbulb(V3 p, float power, int max_iter) {  // mandelbulb.wgsl:5-30
    V3 w = p;
    float m = dot(w, w);
    float dz = 1.0f;
    for (int i = 0; i < max_iter; i++) {
        PT_SITE(44);
        w = mandelbulb_iterate(w, p, m, power, dz);
        m = dot(w, w);
        if (m > 256.0f) break;
    }
    float r = dm::sqrt_(m);
    return dm::div_(0.25f * dm::log_(m) * r, dz);
}
// julia.wgsl:4-11.  Every component is four products summed left to right, each product rounded on its own:
//   x = ax bx - ay by - az bz - aw bw      y = ax by + ay bx + az bw - aw bz
//   z = ax bz - ay bw + az bx + aw by      w = ax bw + ay bz - az by + aw bx
// evaluated two components at a time on register pairs; a subtraction is the addition of the product with one factor negated
// (exactly the same value), so that both components of a pair add.
PT_DEV V4 quat_mult(V4 a, V4 b) {
    const f2 bxy = dm2::mk(b.x, b.y), byx = dm2::mk(b.y, b.x), bzw = dm2::mk(b.z, b.w), bwz = dm2::mk(b.w, b.z);
    const f2 ax = dm2::splat(a.x), ay_mp = dm2::mk(-a.y, a.y), az_mp = dm2::mk(-a.z, a.z), az_pm = dm2::mk(a.z, -a.z);
    const f2 aw_mm = dm2::splat(-a.w), aw = dm2::splat(a.w);
    const f2 xy = ((ax * bxy + ay_mp * byx) + az_mp * bzw) + aw_mm * bwz;
    const f2 zw = ((ax * bzw + ay_mp * bwz) + az_pm * bxy) + aw * byx;
    V4 r;
    r.x = xy.x; r.y = xy.y; r.z = zw.x; r.w = zw.y;
    return r;
}
PT_DEV float sdf_julia(V3 p, V4 cc, int max_iter) {  // julia.wgsl:13-32
    V4 z; z.x = p.x; z.y = p.y; z.z = p.z; z.w = 0.0f;
    V4 dz; dz.x = 1.0f; dz.y = 0.0f; dz.z = 0.0f; dz.w = 0.0f;
    float r2 = dot4(z, z);
    for (int i = 0; i < max_iter; i++) {
        if (r2 > 16.0f) break;
        PT_SITE(45);
        V4 q = quat_mult(z, dz);
        dz.x = 2.0f * q.x; dz.y = 2.0f * q.y; dz.z = 2.0f * q.z; dz.w = 2.0f * q.w;
        V4 zz = quat_mult(z, z);
        z.x = zz.x + cc.x; z.y = zz.y + cc.y; z.z = zz.z + cc.z; z.w = zz.w + cc.w;
        r2 = dot4(z, z);
    }
    float r = dm::sqrt_(r2);
    float dr = dm::sqrt_(dot4(dz, dz));
    return dm::div_(0.5f * r * dm::log_(r), dr);
}

// Evaluates the figure's distance field at a point given in the space its marcher steps in.
// FRAC: the kernel variant is compiled with the Mandelbulb / Julia marchers.  Scenes without fractal
// figures run a variant without them: their register footprint alone costs a wave of occupancy.
template <bool FRAC>
PT_DEV float sdf_eval(const DPrim& g, V3 p) {
    if (g.type == FIG_TORUS) return sdf_torus(p, g.p[3], g.p[4]);
    if (g.type == FIG_MEBIUS) return sdf_mebius(p, g.p[3]);
    if (FRAC) {
        if (g.type == FIG_MANDELBULB) return sdf_mandelbulb(p, g.p[4], (int)g.p[5]);
        V4 c; c.x = g.p[5]; c.y = g.p[6]; c.z = g.p[7]; c.w = g.p[8];
        return sdf_julia(p, c, (int)g.p[4]);
    }
    return 0.0f;
}
// Tetrahedron normal: offsets e.xyy, e.yyx, e.yxy, e.xxx with e = (k, -k)  (torus.wgsl:54-61 etc.)
template <bool FRAC>
PT_DEV V3 tet_normal(const DPrim& g, V3 local, float k) {
    V3 o0 = v3(k, -k, -k), o1 = v3(-k, -k, k), o2 = v3(-k, k, -k), o3 = v3(k, k, k);
    float d0 = sdf_eval<FRAC>(g, local + o0), d1 = sdf_eval<FRAC>(g, local + o1), d2 = sdf_eval<FRAC>(g, local + o2),
          d3 = sdf_eval<FRAC>(g, local + o3);
    return normalize(o0 * d0 + o1 * d1 + o2 * d2 + o3 * d3);
}
// torus.wgsl:10-81
PT_DEV bool march_torus(const Ray& ray, const DPrim& g, float& t_out) {
    float major_r = g.p[3], minor_r = g.p[4];
    float bound_r = major_r + minor_r;
    V3 pos = v3p(g.p);
    V3 oc = ray.o - pos;
    float b = dot(oc, ray.d);
    float cc = dot(oc, oc) - g.p[5];  // (major + 2 minor)^2, computed at upload
    float disc = dm::fma_(b, b, -cc);
    if (disc < 0.0f) return false;
    const float omega = 1.4f;
    float sq = dm::sqrt_(dm::fmax_(disc, 0.0f));
    float t = dm::fmax_(-b - sq, PT_EPS);
    float max_t = dm::fmin_(-b + sq, PT_MAX_T);
    float prev_d = 0.0f;
    for (int i = 0; i < 128; i++) {
        PT_SITE(40);
        V3 p = madd(ray.o, ray.d, t) - pos;
        float d = sdf_torus(p, major_r, minor_r);
        if (d < 0.0f && prev_d > 0.0f) {
            t -= prev_d * (omega - 1.0f);
            V3 p2 = madd(ray.o, ray.d, t) - pos;
            float d2 = sdf_torus(p2, major_r, minor_r);
            t += d2;
            prev_d = d2;
            continue;
        }
        if (dm::abs_(d) < PT_EPS * t) { t_out = t; return true; }
        t += d * omega;
        prev_d = d;
        if (t > max_t) break;
    }
    return false;
}
PT_DEV void finish_torus(const Ray& ray, const DPrim& g, float t, HitFull* h, bool need_uv) {
    V3 pos = v3p(g.p);
    h->pos = madd(ray.o, ray.d, t);
    V3 local = h->pos - pos;
    h->normal = tet_normal<false>(g, local, 0.5773f * PT_EPS);
    if (need_uv) {
        float angle_major = dm::atan2_(local.z, local.x);
        float px = dm::sqrt_(dot2(local.x, local.z, local.x, local.z)) - g.p[3];
        float angle_minor = dm::atan2_(local.y, px);
        h->uv.x = angle_major * (1.0f / PT_TWO_PI) + 0.5f;
        h->uv.y = angle_minor * (1.0f / PT_TWO_PI) + 0.5f;
    }
}
// mebius.wgsl:38-87
PT_DEV bool march_mebius(const Ray& ray, const DPrim& g, float& t_out) {
    V3 pos = v3p(g.p);
    float R = g.p[3];
    V3 oc = ray.o - pos;
    float b = dot(oc, ray.d);
    float cc = dot(oc, oc) - g.p[4];  // (1.5 R)^2, computed at upload
    float disc = dm::fma_(b, b, -cc);
    if (disc < 0.0f) return false;
    float sq = dm::sqrt_(disc);
    float t = dm::fmax_(-b - sq, PT_EPS);
    float max_t = -b + sq;
    for (int i = 0; i < 256; i++) {
        PT_SITE(41);
        V3 p = madd(ray.o, ray.d, t) - pos;
        float d = sdf_mebius(p, R);
        if (dm::abs_(d) < PT_EPS * t * 0.5f) { t_out = t; return true; }
        t += d;
        if (t > max_t) break;
    }
    return false;
}
PT_DEV void finish_mebius(const Ray& ray, const DPrim& g, float t, HitFull* h, bool need_uv) {
    V3 pos = v3p(g.p);
    h->pos = madd(ray.o, ray.d, t);
    V3 local = h->pos - pos;
    h->normal = tet_normal<false>(g, local, 0.5773f * PT_EPS * 2.0f);
    if (need_uv) {
        float angle = dm::atan2_(local.z, local.x);
        h->uv.x = angle * (1.0f / PT_TWO_PI) + 0.5f;
        h->uv.y = 0.5f;
    }
}
// mandelbulb.wgsl:32-101 / julia.wgsl:34-101
PT_DEV bool march_fractal(const Ray& ray, const DPrim& g, uint32_t march_steps, float& t_out) {
    V3 pos = v3p(g.p);
    float radius = g.p[3];
    V3 oc = ray.o - pos;
    float b = dot(oc, ray.d);
    float cs = dot(oc, oc) - g.p[9];  // (1.5 radius)^2, computed at upload
    float disc = dm::fma_(b, b, -cs);
    if (disc < 0.0f) return false;
    const float omega = 1.3f;
    const float inv_radius = dm::rcp_(radius);
    float sq = dm::sqrt_(disc);
    float t = dm::fmax_(-b - sq, PT_EPS);
    float max_t = -b + sq;
    float prev_d = 0.0f;
    for (uint32_t i = 0; i < march_steps; i++) {
        PT_SITE(42);
        V3 p = madd(ray.o, ray.d, t) - pos;
        float d = sdf_eval<true>(g, p * inv_radius) * radius;
        if (d < 0.0f && prev_d > 0.0f) {
            t -= prev_d * (omega - 1.0f);
            V3 p2 = madd(ray.o, ray.d, t) - pos;
            float d2 = sdf_eval<true>(g, p2 * inv_radius) * radius;
            t += d2;
            prev_d = d2;
            continue;
        }
        if (dm::abs_(d) < PT_EPS * t * 0.5f) { t_out = t; return true; }
        t += d * omega;
        prev_d = d;
        if (t > max_t) break;
    }
    return false;
}
PT_DEV void finish_fractal(const Ray& ray, const DPrim& g, float t, HitFull* h) {
    V3 pos = v3p(g.p);
    h->pos = madd(ray.o, ray.d, t);
    V3 local = (h->pos - pos) * (dm::rcp_(g.p[3]));
    h->normal = tet_normal<true>(g, local, 0.5773f * PT_EPS * 2.0f);
    h->uv.x = 0.0f;
    h->uv.y = 0.0f;
}

// ---- the ray marchers as resumable steps ------------------------------------------------------------------
// The scheduled kernels (sched_dev.h) run a marcher one loop iteration at a time, so that lanes that march can
// be served together however long each of them marches.  march_begin_* is the code before the loop (false =
// the ray misses the bounding sphere), march_step_* one iteration: MARCH_GO = iterate again, MARCH_HIT = hit at
// m.t, MARCH_MISS.  The arithmetic, and so every result, is that of march_torus / march_mebius / march_fractal.
struct MarchState { float t, prev_d, max_t; uint32_t i; };
enum : int { MARCH_GO = 0, MARCH_HIT = 1, MARCH_MISS = 2 };
PT_DEV bool march_begin_torus(const Ray& ray, const DPrim& g, MarchState& m) {
    float major_r = g.p[3], minor_r = g.p[4];
    float bound_r = major_r + minor_r;
    V3 oc = ray.o - v3p(g.p);
    float b = dot(oc, ray.d);
    float cc = dot(oc, oc) - g.p[5];  // (major + 2 minor)^2, computed at upload
    float disc = dm::fma_(b, b, -cc);
    if (disc < 0.0f) return false;
    float sq = dm::sqrt_(dm::fmax_(disc, 0.0f));
    m.t = dm::fmax_(-b - sq, PT_EPS);
    m.max_t = dm::fmin_(-b + sq, PT_MAX_T);
    m.prev_d = 0.0f;
    m.i = 0u;
    return true;
}
PT_DEV int march_step_torus(const Ray& ray, const DPrim& g, MarchState& m) {
    const float omega = 1.4f;
    float major_r = g.p[3], minor_r = g.p[4];
    V3 pos = v3p(g.p);
    V3 p = madd(ray.o, ray.d, m.t) - pos;
    float d = sdf_torus(p, major_r, minor_r);
    if (d < 0.0f && m.prev_d > 0.0f) {
        m.t -= m.prev_d * (omega - 1.0f);
        V3 p2 = madd(ray.o, ray.d, m.t) - pos;
        float d2 = sdf_torus(p2, major_r, minor_r);
        m.t += d2;
        m.prev_d = d2;
    } else {
        if (dm::abs_(d) < PT_EPS * m.t) return MARCH_HIT;
        m.t += d * omega;
        m.prev_d = d;
        if (m.t > m.max_t) return MARCH_MISS;
    }
    m.i += 1u;
    return m.i < 128u ? MARCH_GO : MARCH_MISS;
}
PT_DEV bool march_begin_mebius(const Ray& ray, const DPrim& g, MarchState& m) {
    float R = g.p[3];
    V3 oc = ray.o - v3p(g.p);
    float b = dot(oc, ray.d);
    float cc = dot(oc, oc) - g.p[4];  // (1.5 R)^2, computed at upload
    float disc = dm::fma_(b, b, -cc);
    if (disc < 0.0f) return false;
    float sq = dm::sqrt_(disc);
    m.t = dm::fmax_(-b - sq, PT_EPS);
    m.max_t = -b + sq;
    m.prev_d = 0.0f;
    m.i = 0u;
    return true;
}
PT_DEV int march_step_mebius(const Ray& ray, const DPrim& g, MarchState& m) {
    V3 p = madd(ray.o, ray.d, m.t) - v3p(g.p);
    float d = sdf_mebius(p, g.p[3]);
    if (dm::abs_(d) < PT_EPS * m.t * 0.5f) return MARCH_HIT;
    m.t += d;
    if (m.t > m.max_t) return MARCH_MISS;
    m.i += 1u;
    return m.i < 256u ? MARCH_GO : MARCH_MISS;
}
PT_DEV bool march_begin_fractal(const Ray& ray, const DPrim& g, uint32_t march_steps, MarchState& m) {
    float radius = g.p[3];
    V3 oc = ray.o - v3p(g.p);
    float b = dot(oc, ray.d);
    float cs = dot(oc, oc) - g.p[9];  // (1.5 radius)^2, computed at upload
    float disc = dm::fma_(b, b, -cs);
    if (disc < 0.0f) return false;
    float sq = dm::sqrt_(disc);
    m.t = dm::fmax_(-b - sq, PT_EPS);
    m.max_t = -b + sq;
    m.prev_d = 0.0f;
    m.i = 0u;
    return march_steps > 0u;
}
PT_DEV int march_step_fractal(const Ray& ray, const DPrim& g, uint32_t march_steps, MarchState& m) {
    const float omega = 1.3f;
    V3 pos = v3p(g.p);
    float radius = g.p[3];
    const float inv_radius = dm::rcp_(radius);
    V3 p = madd(ray.o, ray.d, m.t) - pos;
    float d = sdf_eval<true>(g, p * inv_radius) * radius;
    if (d < 0.0f && m.prev_d > 0.0f) {
        m.t -= m.prev_d * (omega - 1.0f);
        V3 p2 = madd(ray.o, ray.d, m.t) - pos;
        float d2 = sdf_eval<true>(g, p2 * inv_radius) * radius;
        m.t += d2;
        m.prev_d = d2;
    } else {
        if (dm::abs_(d) < PT_EPS * m.t * 0.5f) return MARCH_HIT;
        m.t += d * omega;
        m.prev_d = d;
        if (m.t > m.max_t) return MARCH_MISS;
    }
    m.i += 1u;
    return m.i < march_steps ? MARCH_GO : MARCH_MISS;
}

// ---- the fractal distance estimators as resumable iterations ---------------------------------------------
// A Mandelbulb iteration is ~170 instructions (two pow, acos, atan2, two sincos) and points escape after anything
// from 1 to max_iter of them, so in a wave that evaluates the estimator for many lanes at once most iterations
// run for a handful of lanes (measured: 6 of 64).  The scheduled kernels therefore run ONE iteration per step for
// all lanes that are inside either estimator, whatever it is evaluated for (a marching step, the second
// evaluation of an over-relaxed step, one of the four evaluations of the normal).  frac_eval_begin is the code
// before the loop of sdf_mandelbulb / sdf_julia, frac_eval_iter one loop iteration and, when the loop is over,
// the code after it; same operations, same order.
struct FracIter {
    float s[9];   // Mandelbulb: w xyz, m, dz, p xyz.  Julia: z xyzw, dz xyzw, r2.
    uint32_t i;   // iterations done
};
PT_DEV void frac_eval_begin(const DPrim& g, V3 p, FracIter& f) {
    f.i = 0u;
    if (g.type == FIG_MANDELBULB) {
        f.s[0] = p.x; f.s[1] = p.y; f.s[2] = p.z;
        f.s[3] = dot(p, p);
        f.s[4] = 1.0f;
        f.s[5] = p.x; f.s[6] = p.y; f.s[7] = p.z;
        f.s[8] = 0.0f;
    } else {
        V4 z; z.x = p.x; z.y = p.y; z.z = p.z; z.w = 0.0f;
        f.s[0] = z.x; f.s[1] = z.y; f.s[2] = z.z; f.s[3] = z.w;
        f.s[4] = 1.0f; f.s[5] = 0.0f; f.s[6] = 0.0f; f.s[7] = 0.0f;
        f.s[8] = dot4(z, z);
    }
}
// Returns true, with the estimator's value, when the evaluation is complete.
PT_DEV bool frac_eval_iter(const DPrim& g, FracIter& f, float& value, bool fast = false) {
    if (g.type == FIG_MANDELBULB) {  // mandelbulb.wgsl:5-30
        const float power = g.p[4];
        const uint32_t max_iter = (uint32_t)(int)g.p[5];
        float m = f.s[3], dz = f.s[4];
        bool done = (int)g.p[5] <= 0 || f.i >= max_iter;
        if (!done) {
            PT_SITE(44);
            V3 w = v3(f.s[0], f.s[1], f.s[2]), p = v3(f.s[5], f.s[6], f.s[7]);
            w = fast ? mandelbulb_iterate_fast(w, p, m, power, dz) : mandelbulb_iterate(w, p, m, power, dz);
            m = dot(w, w);
            f.s[0] = w.x; f.s[1] = w.y; f.s[2] = w.z;
            f.s[3] = m;
            f.s[4] = dz;
            f.i += 1u;
            done = m > 256.0f || f.i >= max_iter;
        }
        if (!done) return false;
        if (fast) {  // ln m = log2 m * ln 2
            value = 0.25f * (fastm::log2(m) * 0.69314718055994531f) * fastm::sqrt(m) * fastm::rcp(dz);
            return true;
        }
        float r = dm::sqrt_(m);
        value = dm::div_(0.25f * dm::log_(m) * r, dz);
        return true;
    }
    // julia.wgsl:13-32
    const uint32_t max_iter = (uint32_t)(int)g.p[4];
    V4 z; z.x = f.s[0]; z.y = f.s[1]; z.z = f.s[2]; z.w = f.s[3];
    V4 dz; dz.x = f.s[4]; dz.y = f.s[5]; dz.z = f.s[6]; dz.w = f.s[7];
    float r2 = f.s[8];
    bool done = (int)g.p[4] <= 0 || f.i >= max_iter || r2 > 16.0f;
    if (!done) {
        PT_SITE(45);
        V4 cc; cc.x = g.p[5]; cc.y = g.p[6]; cc.z = g.p[7]; cc.w = g.p[8];
        V4 q = quat_mult(z, dz);
        dz.x = 2.0f * q.x; dz.y = 2.0f * q.y; dz.z = 2.0f * q.z; dz.w = 2.0f * q.w;
        V4 zz = quat_mult(z, z);
        z.x = zz.x + cc.x; z.y = zz.y + cc.y; z.z = zz.z + cc.z; z.w = zz.w + cc.w;
        r2 = dot4(z, z);
        f.s[0] = z.x; f.s[1] = z.y; f.s[2] = z.z; f.s[3] = z.w;
        f.s[4] = dz.x; f.s[5] = dz.y; f.s[6] = dz.z; f.s[7] = dz.w;
        f.s[8] = r2;
        f.i += 1u;
        done = f.i >= max_iter || r2 > 16.0f;
    }
    if (!done) return false;
    if (fast) {  // the iteration is plain arithmetic in both modes; only the closing expression has builtins
        const float rr = fastm::sqrt(r2);
        value = 0.5f * rr * (fastm::log2(rr) * 0.69314718055994531f) * __builtin_amdgcn_rsqf(dot4(dz, dz));
        return true;
    }
    float r = dm::sqrt_(r2);
    float dr = dm::sqrt_(dot4(dz, dz));
    value = dm::div_(0.5f * r * dm::log_(r), dr);
    return true;
}

// Candidate test: hit distance only (figures/dispatch.wgsl:22-89).
// `sub` receives the winning face of a pyramid / tetrahedron (0 otherwise) for resolve_hit.
template <bool FRAC>
PT_DEV bool intersect_t(const Ray& ray, V3 inv_dir, const DPrim& g, uint32_t march_steps, float& t, uint32_t& sub) {
    sub = 0u;
    PT_SITE_TYPED(10, g.type);
    switch (g.type) {
        case FIG_SPHERE: return isect_sphere<false>(ray, g, t, nullptr, false);
        case FIG_PLANE: return isect_plane<false>(ray, g, t, nullptr, false);
        case FIG_CUBE: return isect_cube<false>(ray, inv_dir, g, t, nullptr);
        case FIG_CYLINDER: return isect_cylinder<false>(ray, g, t, nullptr, false);
        case FIG_CONE: return isect_cone<false>(ray, g, t, nullptr, false);
        case FIG_TORUS: return march_torus(ray, g, t);
        case FIG_DISC: return isect_disc<false>(ray, g, t, nullptr);
        case FIG_TRIANGLE: return isect_triangle<false>(ray, g, t, nullptr, false, nullptr);
        case FIG_SKYBOX: return isect_skybox<false>(ray, g, t, nullptr, false);
        case FIG_MANDELBULB:
        case FIG_JULIA: if (FRAC) return march_fractal(ray, g, march_steps, t); return false;
        case FIG_ELLIPSOID: return isect_ellipsoid<false>(ray, g, t, nullptr, false);
        case FIG_PARABOLOID: return isect_paraboloid<false>(ray, g, t, nullptr, false);
        case FIG_HYPERBOLOID: return isect_hyperboloid<false>(ray, g, t, nullptr, false);
        case FIG_MEBIUS: return march_mebius(ray, g, t);
        case FIG_PYRAMID: return isect_pyramid<false>(ray, g, t, nullptr, sub);
        case FIG_TETRAHEDRON: return isect_tetrahedron<false>(ray, g, t, nullptr, sub);
        default: return false;
    }
}
// The plane list after the BVH: every lane tests the same figure, so its type is wave-uniform; a plane (the
// only infinite figure the reference's loaders produce) skips the 17-way switch.
template <bool FRAC>
PT_DEV bool intersect_listed(const Ray& ray, V3 inv_dir, const DPrim& g, uint32_t march_steps, float& t, uint32_t& sub) {
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)g.type) == FIG_PLANE) {
        sub = 0u;
        PT_SITE_TYPED(10, FIG_PLANE);
        return isect_plane<false>(ray, g, t, nullptr, false);
    }
    return intersect_t<FRAC>(ray, inv_dir, g, march_steps, t, sub);
}
// The closed-form types only (the scheduled kernels run the marchers as separate steps).
PT_DEV bool intersect_closed(const Ray& ray, V3 inv_dir, const DPrim& g, float& t, uint32_t& sub) {
    sub = 0u;
    PT_SITE_TYPED(10, g.type);
    switch (g.type) {
        case FIG_SPHERE: return isect_sphere<false>(ray, g, t, nullptr, false);
        case FIG_PLANE: return isect_plane<false>(ray, g, t, nullptr, false);
        case FIG_CUBE: return isect_cube<false>(ray, inv_dir, g, t, nullptr);
        case FIG_CYLINDER: return isect_cylinder<false>(ray, g, t, nullptr, false);
        case FIG_CONE: return isect_cone<false>(ray, g, t, nullptr, false);
        case FIG_DISC: return isect_disc<false>(ray, g, t, nullptr);
        case FIG_TRIANGLE: return isect_triangle<false>(ray, g, t, nullptr, false, nullptr);
        case FIG_SKYBOX: return isect_skybox<false>(ray, g, t, nullptr, false);
        case FIG_ELLIPSOID: return isect_ellipsoid<false>(ray, g, t, nullptr, false);
        case FIG_PARABOLOID: return isect_paraboloid<false>(ray, g, t, nullptr, false);
        case FIG_HYPERBOLOID: return isect_hyperboloid<false>(ray, g, t, nullptr, false);
        case FIG_PYRAMID: return isect_pyramid<false>(ray, g, t, nullptr, sub);
        case FIG_TETRAHEDRON: return isect_tetrahedron<false>(ray, g, t, nullptr, sub);
        default: return false;
    }
}
// Winner resolution: position, normal and (when the material is textured) UV of the hit at distance t.
// The common closed-form figures are finished directly from t (the same instructions the reference
// runs after it has found t); a pyramid / tetrahedron from its winning face `sub`; the rest re-run
// their intersector in FULL mode.  The distance is never recomputed differently, so nothing changes.
PT_DEV void finish_face(const Ray& ray, V3 a, V3 b, V3 c, float t, bool need_uv, HitFull* h) {
    V3 e1 = b - a, e2 = c - a;
    h->pos = madd(ray.o, ray.d, t);
    h->normal = normalize(cross(e1, e2));
    if (need_uv) {
        float tt, u, v;
        mt_core(ray, a, e1, e2, tt, u, v);
        h->uv.x = u;
        h->uv.y = v;
    }
}
template <bool FRAC>
PT_DEV void resolve_hit(const Ray& ray, V3 inv_dir, const DPrim& g, float t, uint32_t sub, bool need_uv, const DShade* sh, HitFull* h) {
    float tt;
    uint32_t s2;
    h->uv.x = 0.0f;
    h->uv.y = 0.0f;
    h->pos = madd(ray.o, ray.d, t);  // every figure's hit position; written once ahead of the divergent switch
    switch (g.type) {
        case FIG_SPHERE: {
            V3 pos = v3p(g.p);
            h->pos = madd(ray.o, ray.d, t);
            h->normal = normalize(h->pos - pos);
            if (need_uv) h->uv = sphere_uv(h->normal);
            break;
        }
        case FIG_PLANE: {
            V3 pos = v3p(g.p), n = v3p(g.p + 3);
            float denom = dot(n, ray.d);
            h->pos = madd(ray.o, ray.d, t);
            h->normal = denom < 0.0f ? n : -n;
            if (need_uv) {
                Onb onb = build_onb(h->normal);
                V3 local = h->pos - pos;
                h->uv.x = dot(local, onb.u) * 0.25f;
                h->uv.y = dot(local, onb.v) * 0.25f;
            }
            break;
        }
        case FIG_CUBE: {  // cube.wgsl:28-58 from the known distance
            V3 pos = v3p(g.p);
            V3 p = vdivs(h->pos - pos, g.p[3]);
            V3 ap = vabs(p);
            if (ap.x > ap.y && ap.x > ap.z) h->normal = v3(signf(p.x), 0.0f, 0.0f);
            else if (ap.y > ap.z) h->normal = v3(0.0f, signf(p.y), 0.0f);
            else h->normal = v3(0.0f, 0.0f, signf(p.z));
            if (need_uv) {
                if (dm::abs_(h->normal.x) > 0.5f) { h->uv.x = (p.y + 1.0f) * 0.5f; h->uv.y = (p.z + 1.0f) * 0.5f; }
                else if (dm::abs_(h->normal.y) > 0.5f) { h->uv.x = (p.x + 1.0f) * 0.5f; h->uv.y = (p.z + 1.0f) * 0.5f; }
                else { h->uv.x = (p.x + 1.0f) * 0.5f; h->uv.y = (p.y + 1.0f) * 0.5f; }
            }
            break;
        }
        case FIG_CYLINDER: isect_cylinder<true>(ray, g, tt, h, need_uv); break;
        case FIG_CONE: isect_cone<true>(ray, g, tt, h, need_uv); break;
        case FIG_TORUS: finish_torus(ray, g, t, h, need_uv); break;
        case FIG_DISC: {
            V3 pos = v3p(g.p), n = v3p(g.p + 3);
            float radius = g.p[6];
            float denom = dot(n, ray.d);
            h->pos = madd(ray.o, ray.d, t);
            h->normal = denom < 0.0f ? n : -n;
            if (need_uv) {
                V3 off = h->pos - pos;
                h->uv.x = (dm::div_(off.x, radius) + 1.0f) * 0.5f;
                h->uv.y = (dm::div_(off.z, radius) + 1.0f) * 0.5f;
            }
            break;
        }
        case FIG_TRIANGLE: {
            V3 e1 = v3p(g.p + 3), e2 = v3p(g.p + 6);
            h->pos = madd(ray.o, ray.d, t);
            h->normal = normalize(cross(e1, e2));
            if (dot(h->normal, ray.d) > 0.0f) h->normal = -h->normal;
            if (need_uv) isect_triangle<true>(ray, g, tt, h, true, sh);
            break;
        }
        case FIG_SKYBOX: isect_skybox<true>(ray, g, tt, h, need_uv); break;
        case FIG_MANDELBULB:
        case FIG_JULIA: if (FRAC) finish_fractal(ray, g, t, h); break;
        case FIG_ELLIPSOID: {
            V3 pos = v3p(g.p);
            V3 inv_radii = v3p(g.p + 6);
            h->pos = madd(ray.o, ray.d, t);
            V3 local = h->pos - pos;
            h->normal = normalize(local * inv_radii * inv_radii);
            if (need_uv) h->uv = sphere_uv(normalize(local * inv_radii));
            break;
        }
        case FIG_PARABOLOID: isect_paraboloid<true>(ray, g, tt, h, need_uv); break;
        case FIG_HYPERBOLOID: isect_hyperboloid<true>(ray, g, tt, h, need_uv); break;
        case FIG_MEBIUS: finish_mebius(ray, g, t, h, need_uv); break;
        case FIG_PYRAMID: { V3 a, b, c; pyramid_face(g, sub, a, b, c); finish_face(ray, a, b, c, t, need_uv, h); break; }
        case FIG_TETRAHEDRON: { V3 a, b, c; tetra_face(g, sub, a, b, c); finish_face(ray, a, b, c, t, need_uv, h); break; }
        default: h->pos = madd(ray.o, ray.d, t); h->normal = v3(0.0f, 0.0f, 0.0f); break;
    }
    (void)s2;
}
