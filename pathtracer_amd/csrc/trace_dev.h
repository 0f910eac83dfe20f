// BVH traversal, CSG subtraction, shading and the per-lane path state machine (device side).
// Reference: src/shaders/wgsl/{bvh,path_trace,materials,lighting,mis,textures,tonemap,camera,random}.wgsl.
#pragma once
#include "figures_dev.h"

// ---- random.wgsl:6-19 ---------------------------------------------------------------------------
PT_DEV uint32_t pcg_hash(uint32_t input) {
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// f32(state) / 4294967295.0: the divisor rounds to 2^32 in f32, so this is an exact scaling.
PT_DEV float rand_f32(uint32_t& rng) { rng = pcg_hash(rng); return (float)rng * 2.3283064365386963e-10f; }

// ---- bvh.wgsl:4-15 --------------------------------------------------------------------------------
PT_DEV float intersect_aabb(V3 o, V3 inv_dir, const float* bmin, const float* bmax) {
    V3 t1 = (v3p(bmin) - o) * inv_dir, t2 = (v3p(bmax) - o) * inv_dir;
    V3 tmn = vmin(t1, t2), tmx = vmax(t1, t2);
    float t_near = dm::fmax_(dm::fmax_(tmn.x, tmn.y), tmn.z);
    float t_far = dm::fmin_(dm::fmin_(tmx.x, tmx.y), tmx.z);
    if (t_near > t_far || t_far < 0.0f) return PT_MAX_T;
    return dm::fmax_(t_near, 0.0f);
}

// Per-lane traversal stack in LDS, entry e of this lane at st[e * 64]: {ref | cnt << 29, t_box bits}.
// Storing the box distance computed at push time replaces the reference's re-test of the popped node's
// own box (bvh.wgsl:251-256): the value is a pure function of ray and box, so the comparison against
// the current closest distance gives the same outcome.
typedef uint2 StackEntry;
// The walk variants (scenes read from global memory, possibly deep trees) keep only the first
// PT_LDS_STACK entries of a lane's stack in LDS; deeper entries live in a per-lane global-memory area with
// the same lane-interleaved layout.  Deep entries are rarely touched; a 31-entry LDS stack (16 KB per
// wave) would leave room for one wave per SIMD, 4 entries leave room for four (measured: 8 -> 4 entries is
// +22 % on the teapot scene, +14 % on a 10 M-triangle mesh; 2 is slower again).  Push/pop semantics are unchanged.
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 4u
#endif
struct WalkStack {
    StackEntry* lds;  // entry e < nl of this lane at lds[e * 64]
    StackEntry* glb;  // entry e >= nl at glb[(e - nl) * 64]
    uint32_t nl;      // PT_LDS_STACK, or the whole stack depth for the LDS-staged kernels (no global part)
};
// This lane's global-memory stack area inside the launch's spill buffer (one area of (31 - PT_LDS_STACK) x 64 entries per wave).
PT_DEV StackEntry* walk_stack_glb(StackEntry* spill, uint32_t wave_of_grid, uint32_t lane) {
    if (!spill) return nullptr;
    StackEntry* area = spill + (size_t)wave_of_grid * (31u - PT_LDS_STACK) * 64u;
    return area + lane;  // lane-interleaved like the LDS part (a lane-major layout measured the same on the teapot and the 10 M-triangle scene)
}
PT_DEV StackEntry stack_get(const WalkStack& w, int sp) {
    if ((uint32_t)sp < w.nl) return w.lds[sp * 64];
    PT_SITE(47);
    return w.glb[(sp - (int)w.nl) * 64];
}
PT_DEV void stack_put(const WalkStack& w, int sp, StackEntry e) {
    if ((uint32_t)sp < w.nl) w.lds[sp * 64] = e;
    else { PT_SITE(46); w.glb[(sp - (int)w.nl) * 64] = e; }
}

// Two loop shapes run the same per-lane sequence of box and primitive tests (the reference's):
//   WALK = false (*_loop): one loop whose iteration is "pop, then either test the two child boxes or test every
//          primitive of the leaf".  Best for scenes of a few mixed-type primitives (the LDS-staged kernels).
//   WALK = true (*_walk): each round first walks inner nodes until the lane stands at a leaf primitive, then all
//          lanes test one primitive each ("while-while").  Best for meshes, where the tests are uniform
//          code and the box walk is the divergent part (+11 % on the teapot scene).
// Serving one primitive type at a time to all lanes waiting for it (cheapest first, or costly types only)
// was measured slower on every scene: at most 3-5 lanes of a wave want a ray marcher at any time.

// Closest hit over non-subtracted figures (bvh.wgsl:236-305): distance and primitive slot.
template <bool FRAC>
PT_DEV void trace_positive_loop(const DScene& sc, const Ray& ray, V3 inv_dir, uint32_t march_steps, StackEntry* st,
                           float& t_best, uint32_t& slot_best, uint32_t& sub_best) {
    t_best = PT_MAX_T;
    slot_best = PT_SLOT_NONE;
    sub_best = 0u;
    if (sc.nprim > 0u) {
        int sp = 0;
        uint32_t cur = sc.root_ref;
        float cur_t = intersect_aabb(ray.o, inv_dir, sc.root_min, sc.root_max);
        bool have = true;
        for (;;) {
            if (!have) {
                if (sp == 0) break;
                sp -= 1;
                StackEntry e = st[sp * 64];
                cur = e.x;
                cur_t = __uint_as_float(e.y);
            }
            have = false;
            if (cur_t >= t_best) continue;
            uint32_t cnt = cur >> PT_REF_BITS, ref = cur & PT_REF_MASK;
            if (cnt > 0u) {
                for (uint32_t i = 0; i < cnt; i++) {
                    const DPrim& g = sc.prims[ref + i];
                    float ht;
                    uint32_t sub;
                    if (g.type != FIG_SKIP && intersect_t<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht < t_best && ht > PT_EPS) {
                        t_best = ht;
                        slot_best = ref + i;
                        sub_best = sub;
                    }
                }
            } else {
                PT_SITE(2);
                const DNodePair& nd = sc.pairs[ref];
                float tl = intersect_aabb(ray.o, inv_dir, nd.lmin, nd.lmax);
                float tr = intersect_aabb(ray.o, inv_dir, nd.rmin, nd.rmax);
                uint32_t le = nd.lref | (nd.lcnt << PT_REF_BITS), re = nd.rref | (nd.rcnt << PT_REF_BITS);
                // Far child is pushed first, the near child second; the near child would be popped
                // straight away, so it stays in registers (same stack-pointer arithmetic, incl. the
                // reference's `stack_ptr < 31` guards).
                bool left_near = tl < tr;
                uint32_t far_e = left_near ? re : le, near_e = left_near ? le : re;
                float far_t = left_near ? tr : tl, near_t = left_near ? tl : tr;
                if (far_t < t_best && sp < 31) {
                    st[sp * 64] = make_uint2(far_e, __float_as_uint(far_t));
                    sp += 1;
                }
                if (near_t < t_best && sp < 31) {
                    cur = near_e;
                    cur_t = near_t;
                    have = true;
                }
            }
        }
    }
    // bvh.wgsl:192-209: planes, tested linearly after the BVH.
    for (uint32_t i = 0; i < sc.ninf; i++) {
        PT_SITE(30);
        const DPrim& g = sc.inf[i];
        float ht;
        uint32_t sub;
        if (intersect_listed<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht > PT_EPS && ht < t_best) {
            t_best = ht;
            slot_best = PT_SLOT_INF | i;
            sub_best = sub;
        }
    }
}

template <bool FRAC>
PT_DEV void trace_positive_walk(const DScene& sc, const Ray& ray, V3 inv_dir, uint32_t march_steps, const WalkStack& st,
                           float& t_best, uint32_t& slot_best, uint32_t& sub_best) {
    t_best = PT_MAX_T;
    slot_best = PT_SLOT_NONE;
    sub_best = 0u;
    if (sc.nprim > 0u) {
        int sp = 0;
        uint32_t cur = sc.root_ref;
        float cur_t = intersect_aabb(ray.o, inv_dir, sc.root_min, sc.root_max);
        bool have = true;
        uint32_t pi = 0u, pend = 0u;  // WALK: primitives [pi, pend) of the current leaf are still to be tested
        for (;;) {
            bool finished = false;
            while (pi >= pend) {
                PT_SITE(1);
                if (!have) {
                    if (sp == 0) { finished = true; break; }
                    sp -= 1;
                    StackEntry e = stack_get(st, sp);
                    cur = e.x;
                    cur_t = __uint_as_float(e.y);
                }
                have = false;
                if (cur_t >= t_best) continue;
                uint32_t cnt = cur >> PT_REF_BITS, ref = cur & PT_REF_MASK;
                if (cnt > 0u) {
                    pi = ref;
                    pend = ref + cnt;
                } else {
                    PT_SITE(2);
                    const DNodePair& nd = sc.pairs[ref];
                    float tl = intersect_aabb(ray.o, inv_dir, nd.lmin, nd.lmax);
                    float tr = intersect_aabb(ray.o, inv_dir, nd.rmin, nd.rmax);
                    uint32_t le = nd.lref | (nd.lcnt << PT_REF_BITS), re = nd.rref | (nd.rcnt << PT_REF_BITS);
                    // Far child is pushed first, the near child second; the near child would be popped
                    // straight away, so it stays in registers (same stack-pointer arithmetic, incl. the
                    // reference's `stack_ptr < 31` guards).
                    bool left_near = tl < tr;
                    uint32_t far_e = left_near ? re : le, near_e = left_near ? le : re;
                    float far_t = left_near ? tr : tl, near_t = left_near ? tl : tr;
                    if (far_t < t_best && sp < 31) {
                        stack_put(st, sp, make_uint2(far_e, __float_as_uint(far_t)));
                        sp += 1;
                    }
                    if (near_t < t_best && sp < 31) {
                        cur = near_e;
                        cur_t = near_t;
                        have = true;
                    }
                }
            }
            if (finished) break;
            const DPrim& g = sc.prims[pi];
            float ht;
            uint32_t sub;
            if (g.type != FIG_SKIP && intersect_t<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht < t_best && ht > PT_EPS) {
                t_best = ht;
                slot_best = pi;
                sub_best = sub;
            }
            pi += 1u;
        }
    }
    // bvh.wgsl:192-209: planes, tested linearly after the BVH.
    for (uint32_t i = 0; i < sc.ninf; i++) {
        PT_SITE(30);
        const DPrim& g = sc.inf[i];
        float ht;
        uint32_t sub;
        if (intersect_listed<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht > PT_EPS && ht < t_best) {
            t_best = ht;
            slot_best = PT_SLOT_INF | i;
            sub_best = sub;
        }
    }
}

template <bool FRAC, bool WALK>
PT_DEV void trace_positive(const DScene& sc, const Ray& ray, V3 inv_dir, uint32_t march_steps, const WalkStack& st,
                           float& t_best, uint32_t& slot_best, uint32_t& sub_best) {
    if (WALK) trace_positive_walk<FRAC>(sc, ray, inv_dir, march_steps, st, t_best, slot_best, sub_best);
    else trace_positive_loop<FRAC>(sc, ray, inv_dir, march_steps, st.lds, t_best, slot_best, sub_best);
}

// ---- CSG subtraction (bvh.wgsl:18-188) ----------------------------------------------------------------
PT_DEV bool is_inside_neg(const DNeg& f, V3 p) {
    V3 pos = v3p(f.pos), n = v3p(f.normal);
    switch (f.type) {
        case FIG_SPHERE: return length(p - pos) < f.radius;
        case FIG_CUBE: { V3 l = vabs(p - pos); float hh = f.radius; return l.x < hh && l.y < hh && l.z < hh; }
        case FIG_CYLINDER: {
            V3 oc = p - pos;
            float proj = dot(oc, n);
            V3 perp = oc - n * proj;
            float half_h = f.height * 0.5f;
            return dm::abs_(proj) < half_h && dot(perp, perp) < f.radius * f.radius;
        }
        case FIG_CONE: {
            V3 apex = pos + n * f.height;
            V3 oc = p - apex;
            float proj = dot(oc, n);
            if (proj > 0.0f || proj < -f.height) return false;
            float cos2 = dm::rcp_(1.0f + f.radius2);
            float r2 = dot(oc, oc);
            return proj * proj > cos2 * r2;
        }
        case FIG_ELLIPSOID: {
            V3 l = p - pos;
            V3 v = l / v3(dm::fmax_(f.radius, 0.001f), dm::fmax_(f.radius2, 0.001f), dm::fmax_(f.height, 0.001f));
            return dot(v, v) < 1.0f;
        }
        default: return false;
    }
}
PT_DEV float find_exit_t(const DNeg& f, const Ray& ray) {
    V3 pos = v3p(f.pos), axis = v3p(f.normal);
    switch (f.type) {
        case FIG_SPHERE: {
            V3 oc = ray.o - pos;
            float half_b = dot(oc, ray.d);
            float cc = dot(oc, oc) - f.radius * f.radius;
            float disc = dm::fma_(half_b, half_b, -cc);
            if (disc < 0.0f) return PT_MAX_T;
            return -half_b + dm::sqrt_(disc);
        }
        case FIG_CUBE: {
            V3 half = v3(f.radius, f.radius, f.radius);
            V3 inv_dir = vrcp(ray.d);
            V3 t1 = ((pos - half) - ray.o) * inv_dir, t2 = ((pos + half) - ray.o) * inv_dir;
            V3 tmx = vmax(t1, t2);
            return dm::fmin_(dm::fmin_(tmx.x, tmx.y), tmx.z);
        }
        case FIG_CYLINDER: {
            float half_h = f.height * 0.5f;
            V3 oc = ray.o - pos;
            V3 d_perp = ray.d - axis * dot(ray.d, axis);
            V3 oc_perp = oc - axis * dot(oc, axis);
            float a = dot(d_perp, d_perp);
            float half_b = dot(d_perp, oc_perp);
            float cc = dot(oc_perp, oc_perp) - f.radius * f.radius;
            float disc = dm::fma_(half_b, half_b, -(a * cc));
            if (disc < 0.0f) return PT_MAX_T;
            float t_far = dm::div_(-half_b + dm::sqrt_(disc), a);
            float d_axis = dot(ray.d, axis);
            if (dm::abs_(d_axis) > PT_EPS) {
                float t_top = dm::div_(half_h - dot(oc, axis), d_axis);
                float t_bot = dm::div_(-half_h - dot(oc, axis), d_axis);
                return dm::fmin_(t_far, dm::fmax_(t_top, t_bot));
            }
            return t_far;
        }
        case FIG_CONE: {
            V3 apex = pos + axis * f.height;
            V3 oc = ray.o - apex;
            float cos2 = dm::rcp_(1.0f + f.radius2);
            float d_dot_v = dot(ray.d, axis);
            float oc_dot_v = dot(oc, axis);
            float a_c = dm::fma_(d_dot_v, d_dot_v, -(cos2 * dot(ray.d, ray.d)));
            float b_c = dm::fma_(d_dot_v, oc_dot_v, -(cos2 * dot(ray.d, oc)));
            float c_c = dm::fma_(oc_dot_v, oc_dot_v, -(cos2 * dot(oc, oc)));
            float disc = dm::fma_(b_c, b_c, -(a_c * c_c));
            if (disc < 0.0f) return PT_MAX_T;
            return dm::div_(-b_c + dm::sqrt_(disc), a_c);
        }
        default: return PT_MAX_T;
    }
}
PT_DEV bool inside_any_negative(const DScene& sc, V3 p) {  // bvh.wgsl:132-140 over the precomputed list
    for (uint32_t i = 0; i < sc.nneg; i++)
        if (is_inside_neg(sc.negs[i], p)) return true;
    return false;
}
PT_DEV const DPrim& slot_prim(const DScene& sc, uint32_t slot) {
    return (slot & PT_SLOT_INF) ? sc.inf[slot & ~PT_SLOT_INF] : sc.prims[slot];
}
// bvh.wgsl:143-188.  `ray` is updated to the ray that produced the final hit (its local distance t).
template <bool FRAC, bool WALK>
PT_DEV void trace_closest(const DScene& sc, Ray& ray, V3 inv_dir, uint32_t march_steps, const WalkStack& st, float& t, uint32_t& slot,
                          uint32_t& sub) {
    trace_positive<FRAC, WALK>(sc, ray, inv_dir, march_steps, st, t, slot, sub);
    if (sc.nneg > 0u && slot != PT_SLOT_NONE) {
        for (uint32_t attempt = 0; attempt < 8u; attempt++) {
            V3 hp = madd(ray.o, ray.d, t);
            if (!inside_any_negative(sc, hp)) break;
            float min_exit = PT_MAX_T;
            for (uint32_t i = 0; i < sc.nneg; i++)
                if (is_inside_neg(sc.negs[i], hp)) min_exit = dm::fmin_(min_exit, find_exit_t(sc.negs[i], ray));
            if (min_exit >= PT_MAX_T) break;
            float advance = min_exit + PT_EPS * 2.0f;
            ray.o = madd(ray.o, ray.d, advance);
            trace_positive<FRAC, WALK>(sc, ray, inv_dir, march_steps, st, t, slot, sub);
            if (slot == PT_SLOT_NONE) break;
        }
    }
}

// Any-hit occlusion (bvh.wgsl:308-362).
// Any-hit occlusion (bvh.wgsl:308-362).
template <bool FRAC>
PT_DEV bool trace_shadow_loop(const DScene& sc, const Ray& ray, float max_t, uint32_t march_steps, StackEntry* st) {
    V3 inv_dir = vrcp(ray.d);
    const float t_lim = max_t - PT_EPS;
    if (sc.nprim > 0u) {
        int sp = 0;
        uint32_t cur = sc.root_ref;
        float cur_t = intersect_aabb(ray.o, inv_dir, sc.root_min, sc.root_max);
        for (;;) {
            if (cur_t < max_t) {
                uint32_t cnt = cur >> PT_REF_BITS, ref = cur & PT_REF_MASK;
                if (cnt > 0u) {
                    for (uint32_t i = 0; i < cnt; i++) {
                        const DPrim& g = sc.prims[ref + i];
                        float ht;
                        uint32_t sub;
                        if (g.type != FIG_SKIP && intersect_t<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht > PT_EPS && ht < t_lim)
                            if (sc.nneg == 0u || !inside_any_negative(sc, madd(ray.o, ray.d, ht))) return true;
                    }
                } else {
                    // Left is pushed, then right (bvh.wgsl:346-353); right would be popped next.
                    PT_SITE(2);
                    const DNodePair& nd = sc.pairs[ref];
                    float tl = intersect_aabb(ray.o, inv_dir, nd.lmin, nd.lmax);
                    float tr = intersect_aabb(ray.o, inv_dir, nd.rmin, nd.rmax);
                    if (sp < 31) {
                        st[sp * 64] = make_uint2(nd.lref | (nd.lcnt << PT_REF_BITS), __float_as_uint(tl));
                        sp += 1;
                    }
                    if (sp < 31) {
                        cur = nd.rref | (nd.rcnt << PT_REF_BITS);
                        cur_t = tr;
                        continue;
                    }
                }
            }
            if (sp == 0) break;
            sp -= 1;
            StackEntry e = st[sp * 64];
            cur = e.x;
            cur_t = __uint_as_float(e.y);
        }
    }
    for (uint32_t i = 0; i < sc.ninf; i++) {  // bvh.wgsl:212-232
        PT_SITE(30);
        const DPrim& g = sc.inf[i];
        float ht;
        uint32_t sub;
        if (intersect_listed<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht > PT_EPS && ht < t_lim)
            if (sc.nneg == 0u || !inside_any_negative(sc, madd(ray.o, ray.d, ht))) return true;
    }
    return false;
}

template <bool FRAC>
PT_DEV bool trace_shadow_walk(const DScene& sc, const Ray& ray, float max_t, uint32_t march_steps, const WalkStack& st) {
    V3 inv_dir = vrcp(ray.d);
    const float t_lim = max_t - PT_EPS;
    bool occluded = false;
    if (sc.nprim > 0u) {
        int sp = 0;
        uint32_t cur = sc.root_ref;
        float cur_t = intersect_aabb(ray.o, inv_dir, sc.root_min, sc.root_max);
        bool have = true;
        uint32_t pi = 0u, pend = 0u;
        for (;;) {
            bool finished = false;
            while (pi >= pend) {
                PT_SITE(1);
                if (!have) {
                    if (sp == 0) { finished = true; break; }
                    sp -= 1;
                    StackEntry e = stack_get(st, sp);
                    cur = e.x;
                    cur_t = __uint_as_float(e.y);
                }
                have = false;
                if (!(cur_t < max_t)) continue;
                uint32_t cnt = cur >> PT_REF_BITS, ref = cur & PT_REF_MASK;
                if (cnt > 0u) {
                    pi = ref;
                    pend = ref + cnt;
                } else {
                    // Left is pushed, then right (bvh.wgsl:346-353); right would be popped next.
                    PT_SITE(2);
                    const DNodePair& nd = sc.pairs[ref];
                    float tl = intersect_aabb(ray.o, inv_dir, nd.lmin, nd.lmax);
                    float tr = intersect_aabb(ray.o, inv_dir, nd.rmin, nd.rmax);
                    if (sp < 31) {
                        stack_put(st, sp, make_uint2(nd.lref | (nd.lcnt << PT_REF_BITS), __float_as_uint(tl)));
                        sp += 1;
                    }
                    if (sp < 31) {
                        cur = nd.rref | (nd.rcnt << PT_REF_BITS);
                        cur_t = tr;
                        have = true;
                    }
                }
            }
            if (finished) break;
            const DPrim& g = sc.prims[pi];
            float ht;
            uint32_t sub;
            if (g.type != FIG_SKIP && intersect_t<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht > PT_EPS && ht < t_lim)
                if (sc.nneg == 0u || !inside_any_negative(sc, madd(ray.o, ray.d, ht))) { occluded = true; break; }
            pi += 1u;
        }
    }
    if (occluded) return true;
    for (uint32_t i = 0; i < sc.ninf; i++) {  // bvh.wgsl:212-232
        PT_SITE(30);
        const DPrim& g = sc.inf[i];
        float ht;
        uint32_t sub;
        if (intersect_listed<FRAC>(ray, inv_dir, g, march_steps, ht, sub) && ht > PT_EPS && ht < t_lim)
            if (sc.nneg == 0u || !inside_any_negative(sc, madd(ray.o, ray.d, ht))) return true;
    }
    return false;
}

template <bool FRAC, bool WALK>
PT_DEV bool trace_shadow(const DScene& sc, const Ray& ray, float max_t, uint32_t march_steps, const WalkStack& st) {
    return WALK ? trace_shadow_walk<FRAC>(sc, ray, max_t, march_steps, st) : trace_shadow_loop<FRAC>(sc, ray, max_t, march_steps, st.lds);
}

// ---- textures.wgsl:13-29 ---------------------------------------------------------------------------
PT_DEV V3 sample_texture(const DScene& sc, int32_t texture_id, V2 uv) {
    if (texture_id < 0) return v3(1.0f, 1.0f, 1.0f);
    if ((uint32_t)texture_id >= sc.ntex) return v3(0.0f, 0.0f, 0.0f);
    ptx_texture_info info = sc.tex_infos[(uint32_t)texture_id];
    float wx = uv.x - __builtin_floorf(uv.x), wy = uv.y - __builtin_floorf(uv.y);
    float fx = wx * (float)info.width, fy = wy * (float)info.height;
    uint32_t px = fx > 0.0f ? (uint32_t)fx : 0u, py = fy > 0.0f ? (uint32_t)fy : 0u;
    px = px < info.width - 1u ? px : info.width - 1u;
    py = py < info.height - 1u ? py : info.height - 1u;
    uint32_t idx = info.offset + py * info.width + px;
    uint32_t packed = idx < sc.npix ? sc.tex_pixels[idx] : 0u;
    return v3((float)(packed & 0xFFu) * (1.0f / 255.0f), (float)((packed >> 8) & 0xFFu) * (1.0f / 255.0f), (float)((packed >> 16) & 0xFFu) * (1.0f / 255.0f));
}
// skybox.wgsl:6-27
PT_DEV V3 sample_skybox(const DScene& sc, const ptx_camera& cam, V3 dir) {
    if (sc.sky_tex >= 0) {
        V3 d = normalize(dir);
        return sample_texture(sc, sc.sky_tex, sphere_uv(d)) * v3p(sc.sky_color);
    }
    float t = 0.5f * (dir.y + 1.0f);
    V3 col = v3(mixf(1.0f, cam.skybox_color[0], t), mixf(1.0f, cam.skybox_color[1], t), mixf(1.0f, cam.skybox_color[2], t));
    return col * cam.skybox_brightness;
}

// ---- utils.wgsl / materials.wgsl ---------------------------------------------------------------------
PT_DEV V3 reflect_vec(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }
PT_DEV V3 refract_vec(V3 v, V3 n, float eta) {
    float cos_i = dot(-v, n);
    float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
    if (sin2_t > 1.0f) return v3(0.0f, 0.0f, 0.0f);
    float cos_t = dm::sqrt_(1.0f - sin2_t);
    return v * eta + n * (eta * cos_i - cos_t);
}
PT_DEV float ggx_ndf(float n_dot_h, float alpha) {
    float a2 = alpha * alpha;
    float denom = n_dot_h * n_dot_h * (a2 - 1.0f) + 1.0f;
    return dm::div_(a2, PT_PI * denom * denom);
}
PT_DEV float ggx_g1(float n_dot_v, float alpha) {
    float a2 = alpha * alpha;
    float denom = n_dot_v + dm::sqrt_(a2 + (1.0f - a2) * n_dot_v * n_dot_v);
    return dm::div_(2.0f * n_dot_v, denom);
}
struct MatD { V3 base_color; float metallic, roughness, ior; };
PT_DEV V3 eval_brdf(V3 wo, V3 wi, V3 n, const MatD& mat) {  // materials.wgsl:65-90
    float n_dot_l = dm::fmax_(dot(n, wi), 0.0f);
    float n_dot_v = dm::fmax_(dot(n, wo), 0.0f);
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return v3(0.0f, 0.0f, 0.0f);
    V3 h = normalize(wo + wi);
    float n_dot_h = dm::fmax_(dot(n, h), 0.0f);
    float v_dot_h = dm::fmax_(dot(wo, h), 0.0f);
    float alpha = mat.roughness * mat.roughness;
    V3 f0 = v3(mixf(0.04f, mat.base_color.x, mat.metallic), mixf(0.04f, mat.base_color.y, mat.metallic),
               mixf(0.04f, mat.base_color.z, mat.metallic));
    float tt = 1.0f - v_dot_h;
    float t2 = tt * tt;
    float w5 = t2 * t2 * tt;
    V3 f = f0 + (v3(1.0f, 1.0f, 1.0f) - f0) * w5;  // utils.wgsl:44-48
    float d = ggx_ndf(n_dot_h, alpha);
    float g = ggx_g1(n_dot_l, alpha) * ggx_g1(n_dot_v, alpha);
    V3 specular = vdivs(f * (d * g), dm::fmax_(4.0f * n_dot_l * n_dot_v, PT_EPS));
    V3 kd = (v3(1.0f, 1.0f, 1.0f) - f) * (1.0f - mat.metallic);
    V3 diffuse = kd * mat.base_color * PT_INV_PI;
    return diffuse + specular;
}
// materials.wgsl:95-160.  Returns false when the path ends (zero direction or pdf < EPSILON,
// path_trace.wgsl:158-160); otherwise direction and brdf_cos / pdf.
PT_DEV bool sample_brdf(uint32_t& rng, V3 wo, V3 n, const MatD& mat, V3& wi_out, V3& weight) {
    float alpha = mat.roughness * mat.roughness;
    float spec_weight = mixf(0.04f, 1.0f, mat.metallic);
    float spec_prob = dm::fmax_(spec_weight, 0.25f);
    V3 wi, h;
    float n_dot_l;
    float pick = rand_f32(rng);
    float rx = rand_f32(rng), ry = rand_f32(rng);  // both lobes draw exactly two more numbers
    float phi = PT_TWO_PI * rx;
    float sp, cp;
    dm::sincos_(phi, &sp, &cp);
    Onb onb = build_onb(n);
    // Both lobes (materials.wgsl:38-49 GGX half vector, utils.wgsl:16-25 cosine hemisphere) build a unit vector
    // from (sin t cos p, sin t sin p, cos t) in the frame of n; only cos t and sin t differ.  A wave usually holds
    // lanes of both lobes, so the shared part is written once instead of once per branch (same operations on the
    // same operands per lane).
    const bool spec = pick < spec_prob;
    float cos_arg = ry, sin_arg = 1.0f - ry;
    if (spec) {
        float a2 = alpha * alpha;
        cos_arg = dm::div_(1.0f - ry, 1.0f + (a2 - 1.0f) * ry);
    }
    float cos_theta = dm::sqrt_(cos_arg);
    if (spec) sin_arg = 1.0f - cos_theta * cos_theta;
    float sin_theta = dm::sqrt_(sin_arg);
    V3 dir = normalize(onb_mul(onb, v3(sin_theta * cp, sin_theta * sp, cos_theta)));
    if (spec) {
        h = dir;
        wi = reflect_vec(-wo, h);
        n_dot_l = dot(n, wi);
        if (n_dot_l <= 0.0f) return false;
    } else {
        wi = dir;
        n_dot_l = dm::fmax_(dot(n, wi), 0.0f);
        if (n_dot_l <= 0.0f) return false;
        h = normalize(wo + wi);
    }
    float n_dot_h = dm::fmax_(dot(n, h), 0.0f);
    float v_dot_h = dm::fmax_(dot(wo, h), 0.0f);
    V3 brdf = eval_brdf(wo, wi, n, mat);
    float spec_pdf = dm::div_(ggx_ndf(n_dot_h, alpha) * n_dot_h, 4.0f * v_dot_h);
    float diff_pdf = n_dot_l * PT_INV_PI;
    float pdf = spec_prob * spec_pdf + (1.0f - spec_prob) * diff_pdf;
    pdf = dm::fmax_(pdf, PT_EPS);
    // path_trace.wgsl:158: length(direction) < 0.001 || pdf < EPSILON
    if (length(wi) < 0.001f || pdf < PT_EPS) return false;
    wi_out = wi;
    weight = vdivs(brdf * n_dot_l, pdf);
    return true;
}
// materials.wgsl:163-195.  Returns false when the sampled direction is (near) zero (path_trace.wgsl:99-101).
PT_DEV bool sample_glass(uint32_t& rng, V3 wo, V3 n, float ior, V3& dir_out) {
    bool entering = dot(wo, n) > 0.0f;
    V3 face_n = entering ? n : -n;
    float eta = entering ? dm::rcp_(ior) : ior;  // ior / 1.0
    float cos_i = dm::abs_(dot(wo, face_n));
    float r0 = dm::div_(1.0f - ior, 1.0f + ior);  // utils.wgsl:51-57
    r0 = r0 * r0;
    float tt = 1.0f - cos_i;
    float t2 = tt * tt;
    float fresnel = r0 + (1.0f - r0) * (t2 * t2 * tt);
    V3 dir;
    if (rand_f32(rng) < fresnel) {
        dir = reflect_vec(-wo, face_n);
    } else {
        V3 refr = refract_vec(-wo, face_n, eta);
        if (length(refr) < 0.001f) dir = reflect_vec(-wo, face_n);
        else dir = normalize(refr);
    }
    if (length(dir) < 0.001f) return false;
    dir_out = dir;
    return true;
}

// ---- tonemap.wgsl --------------------------------------------------------------------------------
PT_DEV float aces_c(float x) { return clampf(dm::div_(x * (2.51f * x + 0.03f), x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f); }
PT_DEV float linear_to_srgb(float c) {
    float lo = c * 12.92f;
    float hi = 1.055f * dm::pow_(c, (1.0f / 2.4f)) - 0.055f;
    return c <= 0.0031308f ? lo : hi;
}
PT_DEV uint32_t unorm8(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 1.0f) return 255u;
    return (uint32_t)__builtin_rintf(v * 255.0f);
}
PT_DEV uint32_t tonemap_rgba8(const ptx_camera& cam, V3 color) {  // tonemap.wgsl:27-42 + rgba8unorm store
    V3 e = color * cam.exposure;
    V3 m;
    if (cam.tone_mapper == 1u) m = v3(dm::div_(e.x, 1.0f + e.x), dm::div_(e.y, 1.0f + e.y), dm::div_(e.z, 1.0f + e.z));
    else if (cam.tone_mapper == 2u) m = v3(clampf(e.x, 0.0f, 1.0f), clampf(e.y, 0.0f, 1.0f), clampf(e.z, 0.0f, 1.0f));
    else m = v3(aces_c(e.x), aces_c(e.y), aces_c(e.z));
    return unorm8(linear_to_srgb(m.x)) | (unorm8(linear_to_srgb(m.y)) << 8) | (unorm8(linear_to_srgb(m.z)) << 16) | 0xFF000000u;
}

// ---- camera.wgsl:6-19 -------------------------------------------------------------------------------
PT_DEV Ray generate_ray(const ptx_camera& cam, uint32_t& rng, float fx, float fy) {
    float jx = rand_f32(rng) - 0.5f, jy = rand_f32(rng) - 0.5f;
    float px = fx + jx, py = fy + jy;
    float ndc_x = (dm::div_(2.0f * px, (float)cam.width) - 1.0f) * cam.aspect;
    float ndc_y = 1.0f - dm::div_(2.0f * py, (float)cam.height);
    Ray r;
    r.d = normalize(v3p(cam.right) * ndc_x + v3p(cam.up) * ndc_y + v3p(cam.forward) * cam.focal_length);
    r.o = v3p(cam.position);
    return r;
}

// Optional per-phase cycle attribution (diagnostic builds only: -DPTX_PHASE_TIMING).
#define PT_NTICKS 11  // 0 closest hit, 1 shading up to the occlusion query, 2 occlusion query, 3 shading after it, 4 loop rest, 5 sample hand-over;
                      // pools: 6 top-up, 7 sort, 8 record load, 9 record store + key, 10 waiting at workgroup barriers
#ifdef PTX_PHASE_TIMING
static __device__ __forceinline__ void pt_tick_impl(unsigned long long* acc, unsigned long long& last, int i) {
    unsigned long long now = clock64();
    acc[i] += now - last;
    last = now;
}
#define PT_TICK(i) pt_tick_impl(ps.tick_acc, ps.tick_last, i)
#else
#define PT_TICK(i)
#endif

// One iteration of the bounce loop of path_trace.wgsl:63-179 for this lane.  Returns true when the
// path has ended (radiance final).  `primary` receives the figure index of the closest hit.
struct PathState {
    Ray ray;
    V3 throughput, radiance;
    uint32_t rng, bounce;
#ifdef PTX_PHASE_TIMING
    unsigned long long tick_acc[PT_NTICKS], tick_last;
#endif
};
#define PT_PARK_FIELDS 23u  // per-lane LDS slots for values parked across the occlusion query
template <bool FRAC, bool WALK>
PT_DEV bool bounce_step(const DScene& sc, const ptx_camera& cam, PathState& ps, const WalkStack& st, float* park, uint32_t& hit_fig,
                        uint32_t& n_shadow) {
    PT_SITE(0);
    Ray ray = ps.ray;
    V3 inv_dir = vrcp(ray.d);
    float t;
    uint32_t slot, sub;
    // Path state is parked in LDS while the traversal runs (it is not read there).
    park[14 * 64] = ps.throughput.x; park[15 * 64] = ps.throughput.y; park[16 * 64] = ps.throughput.z;
    park[17 * 64] = ps.radiance.x; park[18 * 64] = ps.radiance.y; park[19 * 64] = ps.radiance.z;
    park[21 * 64] = __uint_as_float(ps.rng);
    park[22 * 64] = __uint_as_float(ps.bounce);
    trace_closest<FRAC, WALK>(sc, ray, inv_dir, cam.fractal_march_steps, st, t, slot, sub);
    ps.throughput = v3(park[14 * 64], park[15 * 64], park[16 * 64]);
    ps.radiance = v3(park[17 * 64], park[18 * 64], park[19 * 64]);
    ps.rng = __float_as_uint(park[21 * 64]);
    ps.bounce = __float_as_uint(park[22 * 64]);
    PT_TICK(0);
    if (slot == PT_SLOT_NONE) {
        hit_fig = 0xFFFFFFFFu;
        ps.radiance = ps.radiance + ps.throughput * sample_skybox(sc, cam, ps.ray.d);
        return true;
    }
    PT_SITE(32);
    const DPrim& g = slot_prim(sc, slot);
    hit_fig = g.fig;
    const DShade& sh = sc.shade[g.fig];
    const ptx_material mt = sh.mat;
    bool need_uv = mt.texture_id >= 0;
    HitFull h;
    resolve_hit<FRAC>(ray, inv_dir, g, t, sub, need_uv, &sh, &h);
    V3 base_color = v3p(mt.base_color);
    if (need_uv) {
        V2 tuv;
        tuv.x = h.uv.x * sh.texture_scale;
        tuv.y = h.uv.y * sh.texture_scale;
        base_color = base_color * sample_texture(sc, mt.texture_id, tuv);
    }
    if (mt.emission_strength > 0.0f) {  // path_trace.wgsl:83-87
        V3 le = v3p(mt.emission) * mt.emission_strength;
        ps.radiance = ps.radiance + ps.throughput * le;
        return true;
    }
    V3 wo = -ps.ray.d;
    V3 n = h.normal;
    if (dot(n, wo) < 0.0f) n = -n;
    if (mt.transmission > 0.5f) {  // path_trace.wgsl:97-105
        PT_SITE(37);
        V3 dir;
        if (!sample_glass(ps.rng, wo, n, mt.ior, dir)) return true;
        ps.throughput = ps.throughput * base_color;
        ps.ray.o = madd(h.pos, dir, PT_EPS * 2.0f);
        ps.ray.d = dir;
        ps.bounce += 1u;
        return false;
    }
    MatD mat;
    mat.base_color = base_color;
    mat.metallic = mt.metallic;
    mat.roughness = mt.roughness;
    mat.ior = mt.ior;
    V3 hpos = h.pos;
    if (mt.roughness > 0.04f) {  // path_trace.wgsl:108-154
        const uint32_t num_lights = sc.nlight > 0u ? sc.nlight : 1u;  // empty list = one-element placeholder
        PT_SITE(33);
        float pick = rand_f32(ps.rng);
        uint32_t light_pick = (uint32_t)(pick * (float)num_lights);
        uint32_t li = light_pick < num_lights - 1u ? light_pick : num_lights - 1u;
        float r1 = rand_f32(ps.rng), r2 = rand_f32(ps.rng);
        if (sc.nlight > 0u) {
            const DLight L = sc.lights[li];
            float cos_theta = 1.0f - 2.0f * r1;  // lighting.wgsl:6-20
            float sin_theta = dm::sqrt_(1.0f - cos_theta * cos_theta);
            float sp, cp;
            dm::sincos_(PT_TWO_PI * r2, &sp, &cp);
            V3 local_point = v3(sin_theta * cp, sin_theta * sp, cos_theta) * L.radius;
            V3 lpos = v3p(L.pos);
            V3 light_point = lpos + local_point;
            V3 to_light = light_point - hpos;
            float light_dist = length(to_light);
            V3 light_dir = vdivs(to_light, light_dist);
            float n_dot_l = dot(n, light_dir);
            if (n_dot_l > 0.0f) {
                // The unoccluded contribution is a pure function of values known here; it is evaluated
                // before the occlusion query so that only it and the BSDF inputs (parked in LDS) stay
                // live across the traversal.
                PT_SITE(34);
                V3 light_normal = normalize(light_point - lpos);
                float cos_light = dm::abs_(dot(-light_dir, light_normal));
                float light_solid_pdf = cos_light <= 0.0f ? 0.0f : dm::div_(L.area_pdf * (light_dist * light_dist), cos_light);
                V3 brdf = eval_brdf(wo, light_dir, n, mat);
                float brdf_pdf_val = n_dot_l * PT_INV_PI;
                float pa = light_solid_pdf * (float)num_lights;
                float a2 = pa * pa, b2 = brdf_pdf_val * brdf_pdf_val;  // mis.wgsl:3-7
                float w = dm::div_(a2, dm::fmax_(a2 + b2, 1e-10f));
                V3 contrib = vdivs(ps.throughput * v3p(L.le) * brdf * n_dot_l * w, pa);
                const bool lit = light_solid_pdf > 0.0f;
                Ray sray;
                sray.o = madd(hpos, n, PT_EPS * 2.0f);
                sray.d = light_dir;
                n_shadow += 1u;
                park[0 * 64] = hpos.x; park[1 * 64] = hpos.y; park[2 * 64] = hpos.z;
                park[3 * 64] = n.x; park[4 * 64] = n.y; park[5 * 64] = n.z;
                park[6 * 64] = mat.base_color.x; park[7 * 64] = mat.base_color.y; park[8 * 64] = mat.base_color.z;
                park[9 * 64] = mat.metallic; park[10 * 64] = mat.roughness;
                park[11 * 64] = contrib.x; park[12 * 64] = contrib.y; park[13 * 64] = contrib.z;
                park[20 * 64] = lit ? 1.0f : 0.0f;
                park[21 * 64] = __uint_as_float(ps.rng);
                park[22 * 64] = __uint_as_float(ps.bounce);
                park[14 * 64] = ps.throughput.x; park[15 * 64] = ps.throughput.y; park[16 * 64] = ps.throughput.z;
                park[17 * 64] = ps.radiance.x; park[18 * 64] = ps.radiance.y; park[19 * 64] = ps.radiance.z;
                PT_TICK(1);
                bool occluded = trace_shadow<FRAC, WALK>(sc, sray, light_dist, cam.fractal_march_steps, st);
                PT_TICK(2);
                hpos = v3(park[0 * 64], park[1 * 64], park[2 * 64]);
                n = v3(park[3 * 64], park[4 * 64], park[5 * 64]);
                mat.base_color = v3(park[6 * 64], park[7 * 64], park[8 * 64]);
                mat.metallic = park[9 * 64];
                mat.roughness = park[10 * 64];
                ps.throughput = v3(park[14 * 64], park[15 * 64], park[16 * 64]);
                ps.radiance = v3(park[17 * 64], park[18 * 64], park[19 * 64]);
                ps.rng = __float_as_uint(park[21 * 64]);
                ps.bounce = __float_as_uint(park[22 * 64]);
                wo = -ps.ray.d;
                if (!occluded && park[20 * 64] != 0.0f)  // path_trace.wgsl:148: only when light_solid_pdf > 0
                    ps.radiance = ps.radiance + v3(park[11 * 64], park[12 * 64], park[13 * 64]);
            }
        }
    }
    PT_SITE(35);
    V3 wi, weight;
    if (!sample_brdf(ps.rng, wo, n, mat, wi, weight)) return true;
    ps.throughput = ps.throughput * weight;
    ps.ray.o = madd(hpos, wi, PT_EPS * 2.0f);
    ps.ray.d = wi;
    if (ps.bounce >= 3u) {  // path_trace.wgsl:166-172
        float survival = dm::fmin_(dm::fmax_(ps.throughput.x, dm::fmax_(ps.throughput.y, ps.throughput.z)), 0.95f);
        if (rand_f32(ps.rng) > survival) return true;
        ps.throughput = vdivs(ps.throughput, survival);
    }
    float lum = dot(ps.throughput, v3(0.2126f, 0.7152f, 0.0722f));
    if (lum > cam.firefly_clamp) ps.throughput = ps.throughput * (dm::div_(cam.firefly_clamp, lum));
    ps.bounce += 1u;
    return false;
}
