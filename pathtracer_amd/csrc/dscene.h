// Device-side scene layout.  Built by ptx_upload_scene from the reference's ingest records
// (include/ptx355.h); sized for 64-byte fetches.
#pragma once
#include <stdint.h>
#include "../../include/ptx355.h"

enum : uint32_t {
    FIG_SPHERE = 0, FIG_PLANE = 1, FIG_CUBE = 2, FIG_CYLINDER = 3, FIG_CONE = 4, FIG_TORUS = 5, FIG_DISC = 6,
    FIG_TRIANGLE = 7, FIG_SKYBOX = 8, FIG_MANDELBULB = 9, FIG_JULIA = 10, FIG_ELLIPSOID = 11,
    FIG_PARABOLOID = 12, FIG_HYPERBOLOID = 13, FIG_MEBIUS = 14, FIG_PYRAMID = 15, FIG_TETRAHEDRON = 16,
    FIG_SKIP = 0xFFu  // subtraction shapes never produce hits (bvh.wgsl:263-265)
};

// One primitive as the intersectors read it: 64 bytes = four 16-byte loads.  BVH primitives are stored
// in leaf order (the reference's bvh_prims indirection is applied at upload), so a leaf's primitives
// are contiguous.  Parameter slots p[] by type (p[0..2] is always the position, v0 for triangles):
//   sphere/cube/tetrahedron/skybox/mebius: p3 radius      plane: p3..5 normal     disc: p3..5 normal, p6 radius
//   cylinder: p3..5 axis, p6 radius, p7 height             cone: + p8 radius2 (raw tan^2 slot)
//   torus: p3 major, p4 minor       triangle: p3..5 e1 = v1-v0, p6..8 e2 = v2-v0
//   mandelbulb: p3 radius, p4 power, p5 max_iter           julia: p3 radius, p4 max_iter, p5..8 c
//   ellipsoid: p3 radius, p4 height, p5 radius2            paraboloid/hyperboloid/pyramid: p3 radius, p4 height
struct alignas(16) DPrim {
    uint32_t type;
    uint32_t fig;  // index into the caller's figure list
    float p[14];
};
static_assert(sizeof(DPrim) == 64, "DPrim");

// Shading data per figure: the material plus what the hit needs for texturing.  64 bytes.
struct alignas(16) DShade {
    ptx_material mat;
    float texture_scale;
    uint32_t uv0, uv1, uv2;  // triangle vertex UVs, two binary16 each
};
static_assert(sizeof(DShade) == 64, "DShade");

// Both children of an inner node in one 64-byte record.  ref: for an inner child the index of its own
// pair record, for a leaf child the first primitive slot; cnt: 0 = inner, else leaf primitive count.
struct alignas(16) DNodePair {
    float lmin[3]; uint32_t lref;
    float lmax[3]; uint32_t lcnt;
    float rmin[3]; uint32_t rref;
    float rmax[3]; uint32_t rcnt;
};
static_assert(sizeof(DNodePair) == 64, "DNodePair");

// Emissive figure as NEE samples it (lighting.wgsl:6-26): always a sphere of the figure's radius.
struct alignas(16) DLight {
    float pos[3]; float radius;
    float le[3];  float area_pdf;  // emission * strength; 1 / (4 pi r^2)
};

// Subtraction shape for the CSG point tests (bvh.wgsl:18-129), ascending figure order.
struct alignas(16) DNeg {
    uint32_t type; float radius, radius2, height;
    float pos[3]; float pad0;
    float normal[3]; float pad1;
};

#define PT_REF_BITS 29u
#define PT_REF_MASK 0x1FFFFFFFu
#define PT_SLOT_NONE 0xFFFFFFFFu
#define PT_SLOT_INF 0x80000000u
#ifndef PT_LDS_SCENE_MAX
#define PT_LDS_SCENE_MAX 16384u  // bytes of scene records staged in LDS per workgroup
#endif

struct DScene {
    const DPrim* prims;       // BVH primitives, leaf order
    const DPrim* inf;         // infinite list (planes), ascending figure order, subtraction shapes removed
    const DShade* shade;      // per figure
    const DNodePair* pairs;
    const DLight* lights;
    const DNeg* negs;
    const uint32_t* tex_pixels;
    const ptx_texture_info* tex_infos;
    uint32_t nprim, ninf, nlight, nneg, ntex, npix, nfig;
    uint32_t root_ref;        // ref | cnt << 29 of the root seen as a child entry
    float root_min[3], root_max[3];
    int32_t sky_tex;          // texture id of the first textured skybox figure, or -1 (skybox.wgsl:9-20)
    float sky_color[3];       // its material base_color
    uint32_t stack_entries;   // traversal stack slots a lane can need (<= 31)
    uint32_t lds_stack_entries;  // of those, slots held in LDS (all of them for the LDS-staged kernels)
    uint32_t npair;
    uint32_t has_fractals;    // any Mandelbulb / Julia figure in the BVH: selects the kernel variant with their marchers
    uint32_t lds_scene_bytes; // > 0: the scene records fit in LDS and the LDS-staged kernel is used
};

