/* ptx355 — C ABI of the MI355X path-tracing compute pass.
 *
 * This is the drop-in boundary for the reference's frame-dispatch / accumulator seam.  The reference
 * (a Rust/wgpu application) has no FFI of its own; the seam these entry points replace is the set of
 * wgpu calls its frame loop makes around the `path_trace` compute pipeline.  Each declaration cites
 * the reference code it stands in for (paths relative to the reference checkout).
 *
 * Conventions: plain pointers and sizes, no exceptions; every call returns 0 on success or a negative
 * ptx_status; ptx_last_error() gives the message.  One context = one HIP device + one stream, driven
 * by one thread at a time (as the reference drives everything from its event-loop thread).  Caller
 * owns every array it passes; the library copies during the call.  ptx_trace* enqueue asynchronously;
 * ptx_read_* and ptx_sync synchronise.
 */
#ifndef PTX355_H
#define PTX355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- records, byte-identical to the reference's #[repr(C)] Pod structs ------------------------ */

/* src/camera/camera.rs:150-171 (WGSL Camera, src/shaders/wgsl/types.wgsl:3-22).  112 bytes. */
typedef struct ptx_camera {
    float position[3]; float focal_length;
    float right[3];    float aspect;
    float up[3];       float exposure;
    float forward[3];  uint32_t frame_index;
    uint32_t width, height, sample_count, max_bounces;
    uint32_t tone_mapper, fractal_march_steps; float firefly_clamp, skybox_brightness;
    float skybox_color[3]; float _pad2;
} ptx_camera;

/* src/scene/shape.rs:223-248 (WGSL Figure, types.wgsl:24-41).  112 bytes.  _pad2.._pad4 carry the
 * per-vertex UVs of triangles as two binary16 values each (shape.rs:272-276). */
typedef struct ptx_shape {
    uint32_t shape_type, material_idx; float radius, radius2;
    float position[3]; float height;
    float normal[3];   uint32_t csg_op;
    float rotation[3]; float texture_scale;
    float v0[3]; uint32_t _pad2;
    float v1[3]; uint32_t _pad3;
    float v2[3]; uint32_t _pad4;
} ptx_shape;

/* src/scene/material.rs:108-119 (WGSL Material, types.wgsl:43-52).  48 bytes. */
typedef struct ptx_material {
    float base_color[3]; float metallic;
    float emission[3];   float roughness;
    float emission_strength, ior, transmission; int32_t texture_id;
} ptx_material;

/* src/accel/bvh.rs:12-19 (WGSL BvhNode, types.wgsl:54-59).  32 bytes.  Left child = index + 1. */
typedef struct ptx_bvh_node {
    float aabb_min[3]; uint32_t left_or_prim;
    float aabb_max[3]; uint32_t prim_count;
} ptx_bvh_node;

/* src/io/texture_atlas.rs:10-18.  16 bytes; `offset` counts pixels. */
typedef struct ptx_texture_info { uint32_t width, height, offset, _pad; } ptx_texture_info;

/* PostParams uniform of the post-process pass, 64 bytes (post_process.wgsl:5-16; built by
 * AppState::build_post_params, src/app/state.rs:448-466).  effects[i]: 0 none, 1 negative, 2 sepia, 3 grayscale,
 * 4 fxaa, 5 oil painting, 6 black & white, 7 comic, 8 casting (src/render/post_process.rs:17-30). */
typedef struct ptx_post_params {
    uint32_t width, height, effect_count, oil_radius;
    uint32_t effects[8];
    uint32_t comic_levels, _pad2, _pad3, _pad4;
} ptx_post_params;

typedef enum ptx_status {
    PTX_OK = 0,
    PTX_ERR_BAD_ARG = -1,
    PTX_ERR_HIP = -2,
    PTX_ERR_OOM = -3,
    PTX_ERR_STATE = -4,       /* call order: e.g. trace before upload/resize */
    PTX_ERR_UNSUPPORTED = -5  /* scene outside what the kernels accept (see ptx_upload_scene) */
} ptx_status;

typedef struct ptx_ctx ptx_ctx;

/* Work tallies since the last ptx_clear: paths = pixel-samples finished, bounces = closest-hit traces
 * issued by the bounce loop (one per path·bounce), shadow_rays = NEE occlusion queries. */
typedef struct ptx_stats { uint64_t paths, bounces, shadow_rays; } ptx_stats;

/* Device and stream set-up: replaces GpuContext::new (src/gpu/context.rs:17-82) minus window/surface. */
int ptx_create(int device, ptx_ctx** out);
void ptx_destroy(ptx_ctx* ctx);
const char* ptx_last_error(const ptx_ctx* ctx);

/* Explicit overrides of the choices the library makes per scene (no reference counterpart; for tests and
 * measurement runs: every setting but arith_mode renders the same frames).  Takes effect at once for the uploaded scene, except
 * upload_path, which the next ptx_upload_scene reads.  Keys and values:
 *   arith_mode      0 (default): every result is bit for bit the oracle's (the numerical contract of the kernels: IEEE operations
 *                   in a fixed order, deterministic polynomial sin / cos / atan2 / acos / log / pow).  1, 2: tolerance modes of the
 *                   fractal distance estimators -- the one option that does NOT render the same frames: Mandelbulb / Julia marches
 *                   use the hardware sqrt / rcp / log2 / exp2 / sin / cos (what WGSL's accuracy bounds allow); 1: marches of rays other
 *                   than camera rays, so that primary-hit ids stay identical; 2: every march (ids differ at a few silhouette pixels:
 *                   3 of 2 073 600 measured).  Both are unbiased estimators of the same image: no shift of the mean, per-pixel
 *                   differences within four standard deviations of two independent renders, <= 1 % relative RMSE once converged
 *                   (tests/test_gpu_tolerance.py).  fractals.yaml 1080p: 329 (exact) / 361 / 487 Mpaths/s.  Wave-scheduled fractal
 *                   kernels only; scenes without fractals are unaffected.
 *   kernel_family  -1 per scene (default), 0 lockstep kernels, 1 wave-scheduled kernels
 *   upload_path    -1 by scene size, 0 record re-layout on the host, 1 on the device
 *   claim_tile      0 per scene and launch; pixel jobs a wave reserves per claim: 1, or a power of two from 16 to 1024
 *   claim_guided   -1 per scene (default), 0 every reservation is claim_tile jobs, k = 4..64: a reservation is 1/(k * waves of the
 *                   launch) of the jobs that were left at the wave's previous one, at least claim_tile (wave-scheduled kernels
 *                   without fractals)
 *   claim_ranges    0 per scene (default: sixteen for scenes staged in LDS that have no lights, else one), else 1..16: job counters
 *                   the lockstep kernel's reservations are spread over (scenes staged in LDS without fractals; the other kernels use one)
 *   claim_paced    -1 per scene (default), 0 off, m = 1..64: a wave doubles or halves its reservations (claim_tile to 512 jobs) so that they
 *                   come about m * waves * 10 ns apart (lockstep kernels)
 *   claim_cap      with paced reservations: -1 = 8 (default), 0 none, k: at most 1/(k * waves) of the jobs left per reservation
 *   split_samples   1 (default) / 0: sample-granular jobs + fold, launches that overlap (see ptx_trace_n)
 *   split_small     -1 every launch (default), else: launches of up to this many samples use them, longer ones fold in the
 *                   lane (fractal scenes on the wave-scheduled kernels always use them)
 *   host_waits_per_frame  -1 (default) by the call sequence, 0, 1: a one-sample launch of a host that waits for every frame (reads each
 *                   frame back before it asks for the next, as the reference presents every frame) has no neighbour to overlap
 *                   with and runs whole-pixel jobs: no staging, no fold kernel (lockstep kernels, meshes without lights).  By
 *                   default a launch counts as such when the host has made a synchronising call (ptx_sync, ptx_read_*,
 *                   ptx_kernel_time*) since the launch before it, and before that one as well; 1 declares every one-sample
 *                   launch to be one, 0 none.  Never a function of host timing.  (idle_whole 0 / 1 = 0 / -1: the earlier name.)
 *   split_chunk     0 = bounded by the staging buffer, else samples per chunk
 *   share_walk, share_post, share_march, share_leaf   0 per scene, else 1..64: a state is served from 1/share of the live lanes
 *   walk_mode      -1 per scene, 0 a WALK step walks to a leaf primitive and tests it, 1 one node visit per step with
 *                  lanes at leaf primitives a state of their own, 2 walks to a leaf, tests wait as their own state
 *   sorted_pool    -1 per scene and launch (default: scenes staged in LDS without fractals trace launches from sorted path pools,
 *                   shallow meshes with lights too, shallow meshes without lights their one- and two-sample launches of up to
 *                   4.6 M jobs; pool_over has the lengths, ptx_launch_plan shows the choice for a length), 0 never, 1 also scenes
 *                   in global memory that run the lockstep kernels (kernel_family 0)
 *   pool_over      -1 per scene (default: scenes staged in LDS and shallow meshes with lights: every length -- except that
 *                   one- and two-sample launches of a lockstep scene with several ray-marched figures, whose pixels differ a
 *                   hundredfold in cost, and three- and four-sample launches of one with fewer than four primitives in the tree keep
 *                   lane-owned paths; other scenes, when sorted_pool asks for pools at all: more than four samples), else launches
 *                   of more samples than this trace from pools
 *   pool_paths      0 = by launch length (2048 / 4096 / 8192 for up to 4 / under 64 / more samples, 1024 for one-sample launches of a
 *                   scene in global memory; never more than the workgroup's share of the launch's jobs), else paths per workgroup pool: a power of two from 256 to 16384
 *   pool_finish    -1 = 256; with no job left to reserve, a pool of up to this many paths is traced to the end without further rounds
 *   pool_guided    -1 = 0: a top-up reserves what the pool lacks; k = 1..64: at most 1/(k * workgroups) of the jobs left
 *   pool_spread    -1 one-sample launches, and two-sample launches of scenes in global memory (default), 1 / 0: a pool's reservation samples the frame at a stride of sixteen pixels
 *                   instead of taking neighbouring 16x16 blocks
 *   grid_per_cu     0 = every resident workgroup, else workgroups per compute unit
 *   flights         launches of sample-granular jobs that may overlap: 0 = per launch (fractal scenes and lane-owned launches of up to
 *                   four samples: six, or as many as fit 8 GiB of staging buffers, at least two; launches of up to four samples from
 *                   pools: four, each with a pool set of its own; longer launches of other scenes one after the other); 1 = one after
 *                   the other; up to 8
 *   verbose, bvh_timing   0 / 1: notes on stderr
 * PTX_ERR_BAD_ARG for an unknown key or a value outside its range. */
int ptx_set_option(ptx_ctx* ctx, const char* key, int64_t value);

/* Scene upload: replaces AppState::create_geometry_buffers and the two texture buffers
 * (src/app/state.rs:389-446, 179-182) and bind group 1 (state.rs:717-741, 833-883).  Counts are
 * explicit: pass the real lists, NOT the reference's one-element placeholder buffers
 * (state.rs:381-387, 404-422); an empty light list still makes the NEE branch consume its three random
 * numbers, as the reference's placeholder does.  Figures are re-laid-out on the device.
 * PTX_ERR_UNSUPPORTED: a leaf with more than 7 primitives, more than 2^29 nodes or primitives, or an
 * index out of range. */
int ptx_upload_scene(ptx_ctx* ctx,
                     const ptx_shape* figures, uint32_t n_figures,
                     const ptx_material* materials, uint32_t n_materials,
                     const ptx_bvh_node* bvh_nodes, uint32_t n_nodes,
                     const uint32_t* bvh_prims, uint32_t n_prims,
                     const uint32_t* light_indices, uint32_t n_lights,
                     const uint32_t* infinite_indices, uint32_t n_infinite,
                     const uint32_t* tex_pixels, uint32_t n_pixels,
                     const ptx_texture_info* tex_infos, uint32_t n_textures);

/* Frame size: replaces recreate_size_dependent_resources (src/app/state.rs:492-535).  Allocates the
 * accumulation (w*h*16 B), output (w*h*4 B), primary-id and moment buffers, zeroed; resets the tile to
 * the full frame. */
int ptx_resize(ptx_ctx* ctx, uint32_t width, uint32_t height);

/* Restricts ptx_trace* to the pixel rectangle [x0, x0+w) x [y0, y0+h).  No reference counterpart (the
 * reference always dispatches the full frame, src/render/frame.rs:57-61); used to partition a frame
 * across GPUs.  Pixel values do not depend on the tiling. */
int ptx_set_tile(ptx_ctx* ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h);

/* Interleaved ownership for multi-GPU tile partitioning: ptx_trace* then only processes the 16x16-pixel
 * blocks of the tile whose row-major index is congruent to `rank` modulo `world` (default 0 of 1 = all).
 * No reference counterpart.  Untouched pixels keep their buffer contents (zero after ptx_clear), so the
 * ranks' accumulation buffers sum to the single-GPU frame exactly. */
int ptx_set_partition(ptx_ctx* ctx, uint32_t rank, uint32_t world);

/* encoder.clear_buffer(accumulation) (src/app/rendering.rs:110-112); also zeroes moments and stats. */
int ptx_clear(ptx_ctx* ctx);

/* One sample per pixel: the camera-uniform write plus dispatch_path_trace
 * (src/app/rendering.rs:75-81, 114-120; src/render/frame.rs:7-22).  Uses cam->frame_index and
 * cam->sample_count exactly as path_trace.wgsl:36,47 do; cam->width/height must equal the frame size. */
int ptx_trace(ptx_ctx* ctx, const ptx_camera* cam);

/* n samples per pixel in one launch; bit-identical to n calls of ptx_trace with
 * frame_index = first_frame + k and sample_count = first_sample + k (k = 0..n-1), which is what n
 * iterations of the reference's frame loop do (src/app/rendering.rs:72-82, src/render/accumulator.rs:31-36).
 * The output image is tone-mapped from the final mean, i.e. it equals the last frame's. */
int ptx_trace_n(ptx_ctx* ctx, const ptx_camera* cam, uint32_t first_frame, uint32_t first_sample, uint32_t n);

/* ptx_trace_n for callers that cannot allocate device memory (code under HIP graph capture, serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same frames, bit for bit; the call
 * itself allocates nothing and creates no events: the launch's scratch (pixel-claim counter, per-wave claim state,
 * traversal-stack entries beyond the LDS part) lives in one array of 56.8 MB (PTX_STATIC_BYTES in the source) that
 * every GPU the library runs on holds from load to exit, sized for 4096 waves: what a 256-CU device keeps resident;
 * a larger claiming grid is cut down to that, a launch whose grid follows the frame size (kernel_family 0 on a mesh
 * scene) and runs more waves fails with PTX_ERR_UNSUPPORTED.  Fractal scenes run whole-pixel jobs here (no staging buffer),
 * which is slower than ptx_trace_n; ptx_kernel_time does not see these launches.  All calls on one GPU share the
 * array: two of them in flight at once on one GPU (two contexts, two streams) race.  Scene upload, ptx_resize and
 * the options are set up beforehand with the ordinary calls. */
int ptx_trace_n_static(ptx_ctx* ctx, const ptx_camera* cam, uint32_t first_frame, uint32_t first_sample, uint32_t n);

int ptx_sync(ptx_ctx* ctx);

/* Read-back: replaces copy_texture_to_buffer + map_async (src/app/rendering.rs:314-386).
 * accum: w*h*4 floats (rgb mean, a = 1 once written); output: w*h*4 bytes rgba8unorm. */
int ptx_read_accum(ptx_ctx* ctx, float* rgba_f32);
int ptx_read_output(ptx_ctx* ctx, uint8_t* rgba8);

/* Bvh::build (src/accel/bvh.rs:37-213) on the GPU: the binned-SAH tree over n primitive boxes (aabbs: n x {min xyz,
 * max xyz}, as shape_aabb gives them, src/accel/aabb.rs:105-174), with the flat node array and the primitive order
 * of the reference's sequential builder reproduced exactly.  nodes_out has room for max(1, 2n) nodes,
 * prim_indices_out for n indices (positions in aabbs; the caller remaps them to figure indices as
 * src/app/state.rs:543-567 does).  Synchronous.  The scene is not touched: pass the result to ptx_upload_scene.
 * A bound that is a zero is returned as +0.0 whatever the sign of the zeros it was taken over; a box holding a NaN
 * is refused (PTX_ERR_BAD_ARG).  Device memory while it runs: about 270 bytes per primitive. */
int ptx_bvh_build(ptx_ctx* ctx, const float* aabbs, uint32_t n, ptx_bvh_node* nodes_out, uint32_t* n_nodes_out,
                  uint32_t* prim_indices_out);

/* The reverse of ptx_read_accum / ptx_read_moments (no reference counterpart; its accumulation buffer only ever
 * lives on the device): loads w*h*4 floats into the accumulation buffer and, when m2 is not NULL and moments are
 * enabled, w*h*3 floats into the M2 buffer, to resume a progressive render (continue with first_sample =
 * samples already in the buffer + 1) or to post-process an image rendered elsewhere. */
int ptx_write_accum(ptx_ctx* ctx, const float* rgba_f32, const float* m2);

/* Parity hooks (no reference counterpart): figure index hit by the camera ray of the most recent
 * sample of each pixel (0xFFFFFFFF = miss), w*h u32; and per-sample radiance capture for the next
 * traces: ptx_debug_capture(ctx, max_samples) allocates max_samples * tile_w * tile_h * 3 floats
 * (0 disables), ptx_read_debug_radiance copies the first n captured samples in tile order. */
int ptx_read_primary_ids(ptx_ctx* ctx, uint32_t* ids);
int ptx_debug_capture(ptx_ctx* ctx, uint32_t max_samples);
int ptx_read_debug_radiance(ptx_ctx* ctx, float* out, uint32_t n);

/* Welford second moments (no reference counterpart; the reference keeps only the mean,
 * path_trace.wgsl:44-49).  When enabled, each pixel also accumulates M2 per channel so that per-GPU
 * partial renders can be merged with Chan's formula.  m2: w*h*3 floats. */
int ptx_enable_moments(ptx_ctx* ctx, int enable);
int ptx_read_moments(ptx_ctx* ctx, float* m2);

/* Device-side access for collectives (RCCL via any host framework): raw device pointers to the
 * accumulation (w*h*4 f32) and M2 (w*h*3 f32) buffers of this context. */
void* ptx_device_accum(ptx_ctx* ctx);
void* ptx_device_moments(ptx_ctx* ctx);

/* Chan's parallel merge of another partial render into this context's (mean, M2), per pixel, over
 * the pixel range [first_pixel, first_pixel + n_pixels): n_a samples here, n_b samples in the other.
 * other_mean4 / other_m2 are DEVICE pointers to n_pixels*4 / n_pixels*3 floats.  Then re-tonemaps
 * those pixels into the output image. */
int ptx_merge_moments(ptx_ctx* ctx, uint32_t first_pixel, uint32_t n_pixels, uint32_t n_a,
                      const void* other_mean4, const void* other_m2, uint32_t n_b, const ptx_camera* cam);

/* The multi-GPU exchange inside the boundary, over RCCL (no reference counterpart: the reference runs on one GPU).
 * One process (or thread + context) per GPU.  The library loads RCCL at run time on the first of these calls and
 * reuses a copy the process has already loaded; PTX_ERR_UNSUPPORTED when there is none.
 *   ptx_comm_unique_id: 128 opaque bytes (an ncclUniqueId) made by ONE rank; the host carries them to the others
 *     by whatever channel it has (a file, MPI, a torch.distributed broadcast).
 *   ptx_comm_init: every rank, with the same bytes: creates this context's communicator (collective: returns when all
 *     `world` ranks have called).  ptx_comm_adopt: uses an ncclComm_t the host already has (not destroyed with the
 *     context); NULL detaches.  ptx_comm_free: destroys / detaches now instead of at ptx_destroy.
 *   ptx_merge_sample_split: the epilogue of a sample split.  Every rank has traced the full frame for its own frame
 *     indices with ptx_enable_moments on and first_sample = 1; counts[r] = samples rank r holds.  Slice j of every
 *     rank's (mean, M2) goes to rank j (all peer links at once), rank j folds them with Chan's update in rank order
 *     0, 1, 2, ... (the arithmetic of ptx_merge_moments: deterministic, equal to the sequential mean to rounding,
 *     <= 1e-5 relative), and the merged slices go back to every rank; then the rgba8 image of the whole frame is
 *     re-derived.  On return (asynchronous on the context's stream) every rank holds the merged mean, M2 and image.
 *     A rank with counts[r] == 0 contributes nothing, whatever its buffers hold.
 *   ptx_assemble_tiles: the epilogue of a tile partition.  When the context's partition (ptx_set_partition) is the
 *     communicator's rank and size, every rank sends the 16x16 blocks it traced to every peer and receives theirs (no
 *     arithmetic: exact); for any other disjoint split one SUM all-reduce of the accumulation buffers (exact: untouched
 *     pixels are zero).  Then the rgba8 image of the whole frame. */
int ptx_comm_unique_id(void* id128);
int ptx_comm_init(ptx_ctx* ctx, const void* id128, uint32_t rank, uint32_t world);
int ptx_comm_adopt(ptx_ctx* ctx, void* nccl_comm);
int ptx_comm_free(ptx_ctx* ctx);
int ptx_merge_sample_split(ptx_ctx* ctx, const uint32_t* counts, const ptx_camera* cam);
int ptx_assemble_tiles(ptx_ctx* ctx, const ptx_camera* cam);
/* The slice of the frame's n_pixels that `rank` of `world` merges in ptx_merge_sample_split: contiguous pixel ranges whose
 * sizes differ by at most one (the first n_pixels % world are the longer ones; empty when there are more ranks than
 * pixels).  A pure function (no context, no GPU): hosts and tests that stage the exchange themselves use the same plan. */
void ptx_slice_of(uint64_t n_pixels, uint32_t world, uint32_t rank, uint64_t* first, uint64_t* count);

/* Re-derives the rgba8 output of pixels [first_pixel, first_pixel + n_pixels) from the accumulation buffer
 * (apply_tonemap + rgba8unorm store, tonemap.wgsl:27-42, path_trace.wgsl:52-53): for accumulation data
 * that arrived from another GPU. */
int ptx_tonemap(ptx_ctx* ctx, uint32_t first_pixel, uint32_t n_pixels, const ptx_camera* cam);

/* The reference's post-process pass (dispatch_post_process, src/render/frame.rs:24-39; post_process.wgsl:88-102):
 * applies params->effects[0 .. effect_count) in order to every pixel of the accumulation buffer (read as it is,
 * without tone mapping; the spatial effects read the original buffer) and overwrites the rgba8 output, as the
 * reference's pass overwrites the output texture after the trace of the same frame.  Asynchronous on the
 * context's stream.  width / height must equal the frame size, effect_count <= 8. */
int ptx_post_process(ptx_ctx* ctx, const ptx_post_params* params);

int ptx_read_stats(ptx_ctx* ctx, ptx_stats* out);

/* Name of the kernel variant ptx_trace* launches for the uploaded scene (scene records staged in LDS or
 * read from global memory; with or without the fractal marchers); "" before ptx_upload_scene.  For
 * matching profiler output. */
const char* ptx_trace_kernel_name(const ptx_ctx* ctx);
/* What the launch policy chose for the latest ptx_trace* call (no reference counterpart; for profiles, bench lines and the
 * policy tests): "<kernel> jobs=pixels|samples|pools claim_tile=.. counters=.. guided=.. paced=.. pool=.. flights=.. chunk=..
 * lds_stack=..; grid=..".  The part before ';' depends on the scene, the options and the launch length only, not on the
 * device's size or on host timing.  "" before the first launch on the uploaded scene. */
const char* ptx_last_launch_info(const ptx_ctx* ctx);
/* The same description (the part before ';') of what a ptx_trace_n of n samples would run now, without launching: the row of
 * the launch policy's table for that length, resolved for the current frame size, tile, partition and call sequence.
 * "" before ptx_upload_scene / ptx_resize. */
const char* ptx_launch_plan(ptx_ctx* ctx, uint32_t n);

/* Sum of the trace kernels' durations (HIP events on the context's stream) and their count since the
 * last call; synchronises. */
int ptx_kernel_time(ptx_ctx* ctx, float* total_ms, uint32_t* launches);
/* The same durations one by one, in call order, since the last call of this function (at most 65 536 are kept):
 * up to `capacity` of them go to ms_out, *count receives how many there were; synchronises. */
int ptx_kernel_times(ptx_ctx* ctx, float* ms_out, uint32_t capacity, uint32_t* count);

/* Unit hooks for known-answer tests, evaluated on the device.
 * ptx_kat_math: op 0 sin, 1 cos, 2 asin, 3 acos, 4 log, 5 log2, 6 exp2, 7 atan2(a,b), 8 pow(a,b),
 *   9 min(a,b), 10 max(a,b), 11 sqrt, 12 a / b (the kernels' division: a * rcp(b)), 13 fma(a,b,a), 14 rcp(a);
 *   host arrays of n floats.
 * ptx_kat_exhaustive: runs the kernels' reciprocal (op 0) or square root (op 1) on all 2^32 binary32 inputs and
 *   compares with the definition (correctly rounded, subnormals flushed for the reciprocal; IEEE for the
 *   square root) evaluated on the device with the compiler's IEEE sequences; returns the number of inputs that
 *   differ and one of them.
 * ptx_kat_trace: closest-hit figure index / t and (if max_t != NULL) shadow occlusion for n host rays
 *   (6 floats each: origin, direction) against the uploaded scene; cam supplies fractal_march_steps. */
int ptx_kat_math(ptx_ctx* ctx, int op, uint32_t n, const float* a, const float* b, float* out);
int ptx_kat_exhaustive(ptx_ctx* ctx, int op, uint64_t* n_bad, uint32_t* first_bad);
int ptx_kat_trace(ptx_ctx* ctx, const ptx_camera* cam, uint32_t n, const float* rays, const float* max_t,
                  uint32_t* hit_id, float* hit_t, uint32_t* shadow);

#ifdef __cplusplus
}
#endif
#endif
