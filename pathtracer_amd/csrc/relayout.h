// Scene re-layout on the device: the reference's ingest records (112-byte figures, 48-byte materials, 32-byte nodes,
// the bvh_prims indirection; src/app/state.rs:389-446) into the records the kernels read (dscene.h), one lane per
// record, with the checks ptx_upload_scene makes on the way.  Large scenes take this path: their host-side loops
// over ten million figures cost more than copying the raw arrays to the device.  HBM-bound: per figure 112 B + 48 B
// read, 64 B written; per primitive slot 4 B + 112 B (gathered) read, 64 B written; per inner node 96 B read, 64 B written.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "dscene.h"
#include "dmath_dev.h"

namespace relay {

// Words of the result block the kernels fill.
enum : uint32_t {
    R_ERRORS = 0,      // bits below
    R_NEG_COUNT = 1,   // subtraction shapes among the figures
    R_SKY_INDEX = 2,   // lowest index of a textured skybox figure, or 0xFFFFFFFF
    R_TYPE_MASK = 3,   // 1 << type over the BVH primitives that can be hit
    R_LEVEL0 = 8,      // depth pass: word R_LEVEL0 + l is set when level l of the tree holds an inner node
    R_LEVELS = 32,
    R_WORDS = R_LEVEL0 + R_LEVELS
};
enum : uint32_t {
    E_MATERIAL = 1u,   // figure material_idx out of range
    E_PRIM = 2u,       // bvh_prims entry out of range
    E_CHILD = 4u,      // inner BVH node with child index out of order or range
    E_LEAF_BIG = 8u,   // BVH leaf with more than 7 primitives
    E_LEAF_RANGE = 16u // BVH leaf range outside bvh_prims
};

}  // namespace relay

// ---- one figure -> one intersector record (host and device: the same single f32 operations) ---------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef float pt_onef;  // -ffp-contract=off: every operation below is one rounded f32 operation
#define PT_HD_RCP(y) dm::rcp_(y)
#define PT_HD_FMAX(a, b) dm::fmax_(a, b)
#else
typedef volatile float pt_onef;  // host: volatile keeps the host compiler from fusing or reassociating as well
#define PT_HD_RCP(y) pt_host_rcp(y)
#define PT_HD_FMAX(a, b) pt_host_fmax(a, b)
#endif

// The division contract's reciprocal (dmath_dev.h rcp_) for values precomputed on the host.
inline float pt_host_rcp(float y) {
    if (y != y) return y;
    uint32_t b;
    memcpy(&b, &y, 4);
    const uint32_t sign = b & 0x80000000u, inf = sign | 0x7f800000u;
    float r;
    if ((b & 0x7f800000u) == 0u) { memcpy(&r, &inf, 4); return r; }
    volatile float q = 1.0f / y;
    r = q;
    memcpy(&b, &r, 4);
    if ((b & 0x7f800000u) == 0u) memcpy(&r, &sign, 4);
    return r;
}
// IEEE maxNum with +0 above -0, as v_max_f32 and the oracle's fmax_.
inline float pt_host_fmax(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) { uint32_t u; memcpy(&u, &a, 4); return (u & 0x80000000u) ? b : a; }
    return a > b ? a : b;
}

// Re-layout of one reference figure (src/scene/shape.rs:223-248) into the 64-byte intersector record.
// Ray-independent subexpressions are evaluated here once, with the single f32 operations the intersectors would
// otherwise repeat per ray.
__host__ __device__ inline DPrim make_prim(const ptx_shape& f, uint32_t fig) {
    DPrim g;
    g.type = f.csg_op == 1u ? (uint32_t)FIG_SKIP : f.shape_type;
    g.fig = fig;
    float* p = g.p;
    for (int i = 0; i < 14; i++) p[i] = 0.0f;
    p[0] = f.position[0]; p[1] = f.position[1]; p[2] = f.position[2];
    pt_onef a, b;
    switch (f.shape_type) {
        case FIG_SPHERE: p[3] = f.radius; a = f.radius * f.radius; p[4] = a; break;
        case FIG_CUBE:
            p[3] = f.radius;
            for (int i = 0; i < 3; i++) { a = f.position[i] - f.radius; b = f.position[i] + f.radius; p[4 + i] = a; p[7 + i] = b; }
            break;
        case FIG_TETRAHEDRON: case FIG_SKYBOX: p[3] = f.radius; break;
        case FIG_MEBIUS: p[3] = f.radius; a = f.radius * 1.5f; a = a * a; p[4] = a; break;
        case FIG_PLANE: p[3] = f.normal[0]; p[4] = f.normal[1]; p[5] = f.normal[2]; break;
        case FIG_DISC: p[3] = f.normal[0]; p[4] = f.normal[1]; p[5] = f.normal[2]; p[6] = f.radius; a = f.radius * f.radius; p[7] = a; break;
        case FIG_CYLINDER: p[3] = f.normal[0]; p[4] = f.normal[1]; p[5] = f.normal[2]; p[6] = f.radius; p[7] = f.height; break;
        case FIG_CONE: p[3] = f.normal[0]; p[4] = f.normal[1]; p[5] = f.normal[2]; p[6] = f.radius; p[7] = f.height; p[8] = f.radius2; break;
        case FIG_TORUS: p[3] = f.radius; p[4] = f.radius2; a = f.radius + f.radius2; a = a + f.radius2; a = a * a; p[5] = a; break;
        case FIG_TRIANGLE:
            // e1 = v1 - v0, e2 = v2 - v0: the same single f32 subtractions triangle.wgsl:9-10 performs per ray.
            p[0] = f.v0[0]; p[1] = f.v0[1]; p[2] = f.v0[2];
            for (int i = 0; i < 3; i++) { a = f.v1[i] - f.v0[i]; b = f.v2[i] - f.v0[i]; p[3 + i] = a; p[6 + i] = b; }
            break;
        case FIG_MANDELBULB: p[3] = f.radius; p[4] = f.v0[0]; p[5] = f.v0[1]; a = f.radius * 1.5f; a = a * a; p[9] = a; break;
        case FIG_JULIA:
            p[3] = f.radius; p[4] = f.v0[1]; p[5] = f.rotation[0]; p[6] = f.rotation[1]; p[7] = f.rotation[2]; p[8] = f.radius2;
            a = f.radius * 1.5f; a = a * a; p[9] = a;
            break;
        case FIG_ELLIPSOID:
            p[3] = f.radius; p[4] = f.height; p[5] = f.radius2;
            p[6] = PT_HD_RCP(f.radius); p[7] = PT_HD_RCP(PT_HD_FMAX(f.height, f.radius)); p[8] = PT_HD_RCP(PT_HD_FMAX(f.radius2, f.radius));
            break;
        case FIG_PARABOLOID: case FIG_HYPERBOLOID: case FIG_PYRAMID: p[3] = f.radius; p[4] = f.height; break;
        default: g.type = FIG_SKIP; break;  // unknown type: dispatch.wgsl:78-81 reports no hit
    }
    return g;
}

namespace relay {

__device__ inline uint32_t wave_or(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64); return v; }

// Per figure: its shading record; counts subtraction shapes, finds the first textured skybox.
__global__ void __launch_bounds__(256) shade_kernel(const ptx_shape* figs, uint32_t nfig, const ptx_material* mats, uint32_t nmat,
                                                    DShade* shade, uint32_t* res) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nfig) return;
    const ptx_shape& f = figs[i];
    if (f.material_idx >= nmat) { atomicOr(res + R_ERRORS, E_MATERIAL); return; }
    DShade s;
    s.mat = mats[f.material_idx];
    s.texture_scale = f.texture_scale;
    s.uv0 = f._pad2; s.uv1 = f._pad3; s.uv2 = f._pad4;
    shade[i] = s;
    if (f.csg_op == 1u) atomicAdd(res + R_NEG_COUNT, 1u);
    if (f.shape_type == FIG_SKYBOX && s.mat.texture_id >= 0) atomicMin(res + R_SKY_INDEX, i);
}

// Per BVH primitive slot: the intersector record of the figure it refers to (the bvh_prims indirection applied).
__global__ void __launch_bounds__(256) prim_kernel(const ptx_shape* figs, uint32_t nfig, const uint32_t* prims, uint32_t nprim,
                                                   DPrim* out, uint32_t* res) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    uint32_t bit = 0u;
    bool bad = false;
    if (s < nprim) {
        const uint32_t fi = prims[s];
        if (fi >= nfig) bad = true;
        else {
            const DPrim g = make_prim(figs[fi], fi);
            out[s] = g;
            if (g.type < 32u) bit = 1u << g.type;
        }
    }
    const uint32_t mask = wave_or(bit);
    if ((threadIdx.x & 63u) == 0u && (mask & ~__hip_atomic_load(res + R_TYPE_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u)
        atomicOr(res + R_TYPE_MASK, mask);
    if (bad) atomicOr(res + R_ERRORS, E_PRIM);
}

// Per node: the checks of the flat tree; inner[i] = 1 for an inner node (its pair record's index is the prefix sum).
__global__ void __launch_bounds__(256) node_kernel(const ptx_bvh_node* nodes, uint32_t nnode, uint32_t nprim, uint32_t* inner, uint32_t* res) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nnode) return;
    const ptx_bvh_node& nd = nodes[i];
    uint32_t err = 0u;
    if (nd.prim_count == 0u) {
        // children must have larger indices than their parent (true of the reference's pre-order flatten), which also rules out cycles
        if (i + 1u >= nnode || nd.left_or_prim >= nnode || nd.left_or_prim <= i) err = E_CHILD;
    } else {
        if (nd.prim_count > 7u) err = E_LEAF_BIG;
        else if ((uint64_t)nd.left_or_prim + nd.prim_count > nprim) err = E_LEAF_RANGE;
    }
    inner[i] = nd.prim_count == 0u ? 1u : 0u;
    if (err) atomicOr(res + R_ERRORS, err);
}

__device__ inline void child_entry(const ptx_bvh_node* nodes, const uint32_t* pair_of, uint32_t ci, float* mn, float* mx, uint32_t* ref, uint32_t* cnt) {
    const ptx_bvh_node& ch = nodes[ci];
    for (int k = 0; k < 3; k++) { mn[k] = ch.aabb_min[k]; mx[k] = ch.aabb_max[k]; }
    if (ch.prim_count > 0u) { *ref = ch.left_or_prim; *cnt = ch.prim_count; }
    else { *ref = pair_of[ci]; *cnt = 0u; }
}
// Per inner node: both children in one 64-byte record.
__global__ void __launch_bounds__(256) pair_kernel(const ptx_bvh_node* nodes, uint32_t nnode, const uint32_t* pair_of, DNodePair* pairs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nnode || nodes[i].prim_count != 0u) return;
    DNodePair pr;
    child_entry(nodes, pair_of, i + 1u, pr.lmin, pr.lmax, &pr.lref, &pr.lcnt);
    child_entry(nodes, pair_of, nodes[i].left_or_prim, pr.rmin, pr.rmax, &pr.rref, &pr.rcnt);
    pairs[pair_of[i]] = pr;
}

// Depth of every inner node, one level per pass: nodes at `level` hand level + 1 to their inner children.
__global__ void __launch_bounds__(256) depth_init_kernel(uint32_t* depth, uint32_t nnode) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nnode) depth[i] = i == 0u ? 0u : 0xFFFFFFFFu;
}
__global__ void __launch_bounds__(256) depth_pass_kernel(const ptx_bvh_node* nodes, uint32_t nnode, uint32_t* depth, uint32_t level, uint32_t* res) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nnode || depth[i] != level || nodes[i].prim_count != 0u) return;
    depth[i + 1u] = level + 1u;
    depth[nodes[i].left_or_prim] = level + 1u;
    res[R_LEVEL0 + level] = 1u;  // every writer stores the same value
}

}  // namespace relay
