/* pth_host — C ABI of the host-side mirror of the reference's scene ingest and frame bookkeeping.
 *
 * The reference keeps these steps in Rust (scene/YAML loader, OBJ import, GPU record packing, SAH BVH
 * build, camera uniform, accumulator); a Rust host would keep using its own code and call only
 * include/ptx355.h.  This library restates them in C++ for hosts without the reference's toolchain, so
 * that the buffers handed to ptx_upload_scene are bit-identical to the reference's.  Citations are
 * relative to the reference checkout.  No GPU code here.
 */
#ifndef PTH_HOST_H
#define PTH_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "ptx355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

/* scene::loader::load_scene (src/scene/loader.rs:12-41) + model expansion (src/app/state.rs:122-140,
 * src/model/obj_loader.rs:48-175).  `resource_root` stands for the working directory the reference is
 * started from (scene files name assets as `resources/...`); may be NULL.  Returns NULL on failure with
 * the message in err. */
pth_scene* pth_scene_load(const char* yaml_path, const char* resource_root, char* err, size_t err_len);
/* An empty scene with the default camera (Scene::empty, src/scene/scene.rs:176-180). */
pth_scene* pth_scene_new(void);
void pth_scene_free(pth_scene* s);
const char* pth_last_error(const pth_scene* s);

/* Appends shapes as AUTHORED values carried in figure + material records (one material per shape, as
 * build_gpu_data lays them out, src/app/state.rs:350-377): `normal` as written in the scene file (it is
 * normalised at build as GpuShape::from_shape does, src/scene/shape.rs:252), roughness unclamped (clamped
 * at build as GpuMaterial::from does), fractal power / max_iterations in v0[0] / v0[1], triangle UVs as
 * packed binary16 pairs in _pad2.._pad4; material_idx is ignored. */
int pth_scene_append_shapes(pth_scene* s, const ptx_shape* figs, const ptx_material* mats, uint32_t n);

/* Names a texture for shape `shape_index` (what a `texture:` key does in a scene file). */
int pth_scene_set_shape_texture(pth_scene* s, uint32_t shape_index, const char* texture_path);
uint32_t pth_scene_shape_count(const pth_scene* s);

/* Camera block of a scene file (src/scene/scene.rs:72-124): position, rotation (pitch, yaw, roll in
 * degrees), fov in degrees. */
int pth_scene_set_camera(pth_scene* s, const float position[3], const float rotation[3], float fov);
int pth_scene_set_sky(pth_scene* s, const float skybox_color[3], float skybox_brightness);

/* Synthetic height-field mesh for the large-BVH configuration: a (quads x quads) grid over
 * x,z in [-extent, extent], y = y0 + amp * fbm(x, z) from an integer-hash value noise seeded by `seed`;
 * two triangles per quad, appended as `type: triangle` shapes.  Every `glass_every`-th triangle (0 = none)
 * gets transmission 1 / ior 1.5, the rest roughness 0.8 diffuse.  Deterministic for a given argument list. */
int pth_scene_append_heightfield(pth_scene* s, uint32_t quads, float extent, float y0, float amp, uint32_t seed,
                                 uint32_t glass_every);

/* Textures: paths the scene references in first-seen order (src/app/state.rs:328-348).  The caller decodes
 * each to RGBA8 and provides it; unprovided textures behave like the reference's load failure (the
 * material keeps texture_id as authored). */
uint32_t pth_scene_texture_count(const pth_scene* s);
const char* pth_scene_texture_path(const pth_scene* s, uint32_t i);
int pth_scene_provide_texture(pth_scene* s, uint32_t i, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* build_texture_atlas + build_gpu_data + build_bvh (src/app/state.rs:328-377, 543-567; src/accel/bvh.rs,
 * src/accel/aabb.rs; src/io/texture_atlas.rs).  Must be called before the getters below. */
int pth_scene_build(pth_scene* s);

/* AppState::save_scene + scene::exporter::save_scene (src/app/scene_ops.rs:115-124, src/scene/exporter.rs:11-70): writes the
 * camera block and every shape (model triangles included: the reference saves its expanded list with `models: []`) as
 * YAML, fields at their serde default left out, numeric arrays in flow style.  Per-vertex UVs are not written
 * (shape.rs:169-175).  pth_scene_load reads the file back to the same scene.  0, or -2 when the file cannot be written. */
int pth_scene_save(const pth_scene* s, const char* yaml_path);

/* A BVH builder over n primitive boxes (n x {min xyz, max xyz}): fills max(1, 2n) nodes / n primitive positions
 * and returns 0.  pth_bvh_build is Bvh::build (src/accel/bvh.rs:37-213) on the host; the device library's
 * ptx_bvh_build has the same signature with its context as `user` and gives the same arrays. */
typedef int (*pth_bvh_fn)(void* user, const float* aabbs, uint32_t n, ptx_bvh_node* nodes_out, uint32_t* n_nodes_out,
                          uint32_t* prim_indices_out);
int pth_bvh_build(void* user, const float* aabbs, uint32_t n, ptx_bvh_node* nodes_out, uint32_t* n_nodes_out,
                  uint32_t* prim_indices_out);
/* pth_scene_build with the tree built by `bvh` (user passed through). */
int pth_scene_build_with(pth_scene* s, pth_bvh_fn bvh, void* user);

const ptx_shape* pth_scene_figures(const pth_scene* s, uint32_t* n);
const ptx_material* pth_scene_materials(const pth_scene* s, uint32_t* n);
const ptx_bvh_node* pth_scene_bvh_nodes(const pth_scene* s, uint32_t* n);
const uint32_t* pth_scene_bvh_prims(const pth_scene* s, uint32_t* n);
const uint32_t* pth_scene_lights(const pth_scene* s, uint32_t* n);
const uint32_t* pth_scene_infinite(const pth_scene* s, uint32_t* n);
const uint32_t* pth_scene_tex_pixels(const pth_scene* s, uint32_t* n);
const ptx_texture_info* pth_scene_tex_infos(const pth_scene* s, uint32_t* n);

/* Camera::from_config + Camera::to_gpu (src/camera/camera.rs:47-51, 97-128). */
int pth_scene_camera(const pth_scene* s, uint32_t width, uint32_t height, uint32_t frame_index, uint32_t sample_count,
                     ptx_camera* out);
/* Overrides of the scene's camera block (src/scene/scene.rs:72-124); NaN / 0xFFFFFFFF leave a field alone. */
int pth_scene_set_render_settings(pth_scene* s, uint32_t max_bounces, uint32_t fractal_march_steps, uint32_t tone_mapper,
                                  float firefly_clamp, float exposure);

/* render::accumulator::Accumulator (src/render/accumulator.rs:6-41) plus the frame loop's free-running
 * frame_index (src/app/rendering.rs:72-82, src/app/state.rs:323). */
typedef struct pth_accumulator { uint32_t sample_count; uint32_t dirty; uint32_t frame_index; } pth_accumulator;
void pth_accumulator_init(pth_accumulator* a);
void pth_accumulator_reset(pth_accumulator* a);
/* Advances to the next sample; returns 1 if the accumulation buffer must be cleared first.  On return
 * *frame_for_sample is the frame_index this sample is traced with. */
int pth_accumulator_advance(pth_accumulator* a, uint32_t* frame_for_sample);

/* buffers::dispatch_size (src/gpu/buffers.rs:64-66): ceil-div used for the reference's 8x8 workgroups. */
uint32_t pth_dispatch_size(uint32_t extent, uint32_t workgroup);

/* AppState::build_post_params (src/app/state.rs:448-466): the post-process pass's 16-word uniform (the layout of
 * ptx_post_params): at most 8 effect ids are taken (POST_PARAMS_MAX_EFFECTS, src/constants.rs:64). */
void pth_build_post_params(uint32_t width, uint32_t height, const uint32_t* effects, uint32_t n_effects, uint32_t oil_radius,
                           uint32_t comic_levels, uint32_t out_params[16]);

/* f32 -> binary16 bits as the reference packs triangle UVs (src/scene/shape.rs:290-322): truncating. */
uint16_t pth_f32_to_f16_bits(float v);

#ifdef __cplusplus
}
#endif
#endif
