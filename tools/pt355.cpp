// pt355: a headless host for the path-tracing compute pass, in C++ over the two C ABIs (include/ptx355.h,
// include/pth_host.h).  It plays the render half of the reference's frame loop
// (AppState::update_and_render, src/app/rendering.rs:72-120) without a window: per frame Accumulator::advance, the
// camera uniform with (frame_index, sample_count), the accumulation clear when the accumulator asks for it, one
// dispatch; then the read-back the reference's screenshot path does (src/app/rendering.rs:314-386).
//
//   pt355 render <scene.yaml> <width> <height> <frames> <out.rgba8> [--batched] [--static] [--accum <out.f32>]
//                [--device <n>] [--resources <dir>]
//
// <out.rgba8>: width*height*4 bytes, rows top to bottom; --accum: width*height*4 floats.  --batched renders the
// frames as one ptx_trace_n call (identical pixels), --static through ptx_trace_n_static.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../include/pth_host.h"
#include "../include/ptx355.h"

static int usage() {
    fprintf(stderr, "usage: pt355 render <scene.yaml> <width> <height> <frames> <out.rgba8> [--batched] [--static] [--accum <out.f32>] "
                    "[--device <n>] [--resources <dir>]\n");
    return 2;
}

static bool write_file(const char* path, const void* data, size_t bytes) {
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, bytes, f) == bytes;
    return fclose(f) == 0 && ok;
}

int main(int argc, char** argv) {
    if (argc < 7 || strcmp(argv[1], "render") != 0) return usage();
    const char* scene_path = argv[2];
    const uint32_t width = (uint32_t)atoi(argv[3]), height = (uint32_t)atoi(argv[4]), frames = (uint32_t)atoi(argv[5]);
    const char* out_path = argv[6];
    const char *accum_path = nullptr, *resources = nullptr;
    bool batched = false, fixed = false;
    int device = 0;
    for (int i = 7; i < argc; i++) {
        if (!strcmp(argv[i], "--batched")) batched = true;
        else if (!strcmp(argv[i], "--static")) fixed = batched = true;
        else if (!strcmp(argv[i], "--accum") && i + 1 < argc) accum_path = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--resources") && i + 1 < argc) resources = argv[++i];
        else return usage();
    }
    if (width == 0 || height == 0) return usage();

    char err[512] = {0};
    pth_scene* scene = pth_scene_load(scene_path, resources, err, sizeof(err));
    if (!scene) { fprintf(stderr, "pt355: cannot load %s: %s\n", scene_path, err); return 1; }
    if (pth_scene_build(scene) != 0) { fprintf(stderr, "pt355: scene build failed: %s\n", pth_last_error(scene)); return 1; }

    ptx_ctx* ctx = nullptr;
    if (ptx_create(device, &ctx) != PTX_OK) { fprintf(stderr, "pt355: no usable HIP device %d\n", device); return 1; }
#define CHECK(call)                                                                        \
    do {                                                                                   \
        const int rc_ = (call);                                                            \
        if (rc_ != PTX_OK) {                                                               \
            fprintf(stderr, "pt355: %s failed (%d): %s\n", #call, rc_, ptx_last_error(ctx)); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)
    uint32_t nf, nm, nn, np, nl, ni, npx, nt;
    const ptx_shape* figs = pth_scene_figures(scene, &nf);
    const ptx_material* mats = pth_scene_materials(scene, &nm);
    const ptx_bvh_node* nodes = pth_scene_bvh_nodes(scene, &nn);
    const uint32_t* prims = pth_scene_bvh_prims(scene, &np);
    const uint32_t* lights = pth_scene_lights(scene, &nl);
    const uint32_t* inf = pth_scene_infinite(scene, &ni);
    const uint32_t* pix = pth_scene_tex_pixels(scene, &npx);
    const ptx_texture_info* infos = pth_scene_tex_infos(scene, &nt);
    CHECK(ptx_upload_scene(ctx, figs, nf, mats, nm, nodes, nn, prims, np, lights, nl, inf, ni, pix, npx, infos, nt));
    CHECK(ptx_resize(ctx, width, height));

    // the frame loop: src/app/rendering.rs:72-82, 110-120
    pth_accumulator acc;
    pth_accumulator_init(&acc);
    ptx_camera cam;
    if (batched && frames > 0) {
        uint32_t first_frame = 0;
        const int needs_clear = pth_accumulator_advance(&acc, &first_frame);
        const uint32_t first_sample = acc.sample_count;
        for (uint32_t k = 1; k < frames; k++) { uint32_t f; (void)pth_accumulator_advance(&acc, &f); }
        if (pth_scene_camera(scene, width, height, first_frame, first_sample, &cam) != 0) { fprintf(stderr, "pt355: camera\n"); return 1; }
        if (needs_clear) CHECK(ptx_clear(ctx));
        if (fixed) CHECK(ptx_trace_n_static(ctx, &cam, first_frame, first_sample, frames));
        else CHECK(ptx_trace_n(ctx, &cam, first_frame, first_sample, frames));
    } else {
        for (uint32_t k = 0; k < frames; k++) {
            uint32_t frame = 0;
            const int needs_clear = pth_accumulator_advance(&acc, &frame);
            if (pth_scene_camera(scene, width, height, frame, acc.sample_count, &cam) != 0) { fprintf(stderr, "pt355: camera\n"); return 1; }
            if (needs_clear) CHECK(ptx_clear(ctx));
            CHECK(ptx_trace(ctx, &cam));
        }
    }
    std::vector<uint8_t> rgba8((size_t)width * height * 4);
    CHECK(ptx_read_output(ctx, rgba8.data()));
    if (!write_file(out_path, rgba8.data(), rgba8.size())) { fprintf(stderr, "pt355: cannot write %s\n", out_path); return 1; }
    if (accum_path) {
        std::vector<float> accum((size_t)width * height * 4);
        CHECK(ptx_read_accum(ctx, accum.data()));
        if (!write_file(accum_path, accum.data(), accum.size() * sizeof(float))) { fprintf(stderr, "pt355: cannot write %s\n", accum_path); return 1; }
    }
    ptx_stats st;
    CHECK(ptx_read_stats(ctx, &st));
    float ms = 0.0f;
    uint32_t launches = 0;
    CHECK(ptx_kernel_time(ctx, &ms, &launches));
    printf("pt355: %s %ux%u, %u frames (%s), kernel %s: %llu paths, %llu path-bounces, %.3f ms in %u launches\n", scene_path, width, height, frames,
           fixed ? "ptx_trace_n_static" : (batched ? "ptx_trace_n" : "ptx_trace per frame"), ptx_trace_kernel_name(ctx),
           (unsigned long long)st.paths, (unsigned long long)st.bounces, ms, launches);
    ptx_destroy(ctx);
    pth_scene_free(scene);
    return 0;
}
