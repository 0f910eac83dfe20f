// Post-process pass (src/shaders/wgsl/post_process.wgsl): up to 8 chained effects over the accumulation buffer,
// written as rgba8.  HBM-bound: 16 B read + 4 B written per pixel; the stencil effects (FXAA cross, 3x3 Sobel
// and emboss, (2r+1)^2 box of the oil painting) read their neighbourhood from an LDS tile, loaded once per
// 16x16-pixel workgroup with its halo already clamped to the image edge the way read_pixel_clamped does.
// Same expression trees as the trace kernels' contract (dmath_dev.h).
#pragma once
#include "dmath_dev.h"
#include "../../include/ptx355.h"

#define PP_TILE 16     // a workgroup's pixels: PP_TILE columns x PP_TILE * ROWS rows, 256 threads of ROWS vertically adjacent pixels each
#define PP_ROWS 4      // ROWS of the kernel that chains with an oil painting run (the others: 1)
#define PP_MAX_HALO 8  // oil radii beyond this read global memory instead of the tile

struct PostImage {
    const float4* accum;  // w * h
    int w, h, tx0, ty0, halo;
};
extern __shared__ float4 pp_tile[];  // TILE: (PP_TILE + 2 halo) x (PP_TILE * ROWS + 2 halo) texels around the workgroup's pixels
// Texel (x, y), -halo <= x - tx0 < PP_TILE + halo, -halo <= y - ty0 < PP_TILE * ROWS + halo when the tile is used;
// outside the image the nearest edge texel.
template <bool TILE>
PT_DEV V3 pp_read_clamped(const PostImage& im, int x, int y) {  // post_process.wgsl:37-41
    float4 a;
    if (TILE) {
        a = pp_tile[(y - im.ty0 + im.halo) * (PP_TILE + 2 * im.halo) + (x - im.tx0 + im.halo)];
    } else {
        int cx = x < 0 ? 0 : (x > im.w - 1 ? im.w - 1 : x), cy = y < 0 ? 0 : (y > im.h - 1 ? im.h - 1 : y);
        a = im.accum[(size_t)cy * im.w + cx];
    }
    return v3(a.x, a.y, a.z);
}
template <bool TILE>
PT_DEV V3 pp_fxaa(const PostImage& im, int x, int y) {  // post_process.wgsl:105-135
    const V3 lw = v3(0.299f, 0.587f, 0.114f);
    V3 center = pp_read_clamped<TILE>(im, x, y);
    float luma_c = dot(center, lw);
    V3 cn = pp_read_clamped<TILE>(im, x, y - 1), cw = pp_read_clamped<TILE>(im, x - 1, y), ce = pp_read_clamped<TILE>(im, x + 1, y),
       cs = pp_read_clamped<TILE>(im, x, y + 1);
    float luma_n = dot(cn, lw), luma_w = dot(cw, lw), luma_e = dot(ce, lw), luma_s = dot(cs, lw);
    float range = dm::fmax_(dm::fmax_(luma_n, luma_s), dm::fmax_(luma_w, dm::fmax_(luma_e, luma_c))) -
                  dm::fmin_(dm::fmin_(luma_n, luma_s), dm::fmin_(luma_w, dm::fmin_(luma_e, luma_c)));
    if (range < dm::fmax_(0.0312f, luma_c * 0.125f)) return center;
    float horizontal = dm::abs_(luma_n + luma_s - 2.0f * luma_c);
    float vertical = dm::abs_(luma_w + luma_e - 2.0f * luma_c);
    if (horizontal > vertical) return (cn + cs + center) * (1.0f / 3.0f);
    return (cw + ce + center) * (1.0f / 3.0f);
}
template <bool TILE>
PT_DEV V3 pp_oil(const PostImage& im, int x, int y, int radius) {  // post_process.wgsl:138-158
    // The in-image part of the window is a rectangle: loop over it directly (same taps in the same order, dx
    // outer and dy inner, as the per-tap bounds test of the shader selects).
    const int dx0 = -radius > -x ? -radius : -x, dx1 = radius < im.w - 1 - x ? radius : im.w - 1 - x;
    const int dy0 = -radius > -y ? -radius : -y, dy1 = radius < im.h - 1 - y ? radius : im.h - 1 - y;
    V3 total = v3(0.0f, 0.0f, 0.0f);
    for (int dx = dx0; dx <= dx1; dx++)
        for (int dy = dy0; dy <= dy1; dy++) total = total + pp_read_clamped<TILE>(im, x + dx, y + dy);
    const int count = (dx1 - dx0 + 1) * (dy1 - dy0 + 1);  // > 0: the pixel itself is inside
    if (count > 0) return vdivs(total, (float)count);
    return pp_read_clamped<TILE>(im, x, y);
}
// The oil painting of PP_ROWS vertically adjacent pixels (x, y0 .. y0 + PP_ROWS - 1) from the tile.  Every pixel's sum
// keeps its own order (dx outer, dy inner, ascending: float addition does not associate), but the pixels share
// their columns: one pass down a column of PP_ROWS + 2 radius texels feeds each texel to the pixels whose window
// holds it, 2.8 times fewer LDS reads at radius 3 than one pixel at a time.
template <int ROWS>
PT_DEV void pp_oil_rows(const PostImage& im, int x, int y0, int radius, V3* out) {
    const int dx0 = -radius > -x ? -radius : -x, dx1 = radius < im.w - 1 - x ? radius : im.w - 1 - x;
    int lo[ROWS], hi[ROWS];  // the in-image rows of each pixel's window
    V3 total[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int y = y0 + k;
        lo[k] = y - radius > 0 ? y - radius : 0;
        hi[k] = y + radius < im.h - 1 ? y + radius : im.h - 1;
        total[k] = v3(0.0f, 0.0f, 0.0f);
    }
    const int stride = PP_TILE + 2 * im.halo;
    // Away from the image edge (nearly every wave) a texel's pixels depend on the row alone: scalar tests, no lane masks.
    const bool interior = __all(x - radius >= 0 && x + radius <= im.w - 1 && y0 - radius >= 0 && y0 + ROWS - 1 + radius <= im.h - 1) != 0;
    for (int dx = dx0; dx <= dx1; dx++) {
        const float4* col = pp_tile + (y0 - radius - im.ty0 + im.halo) * stride + (x + dx - im.tx0 + im.halo);
        if (interior) {
            for (int j = 0; j < ROWS + 2 * radius; j++) {
                const float4 a = col[j * stride];
#pragma unroll
                for (int k = 0; k < ROWS; k++)
                    if (j >= k && j <= k + 2 * radius) total[k] = total[k] + v3(a.x, a.y, a.z);
            }
        } else {
            for (int j = 0; j < ROWS + 2 * radius; j++) {
                const int yy = y0 - radius + j;
                const float4 a = col[j * stride];
#pragma unroll
                for (int k = 0; k < ROWS; k++)
                    if (yy >= lo[k] && yy <= hi[k]) total[k] = total[k] + v3(a.x, a.y, a.z);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
        const int count = (dx1 - dx0 + 1) * (hi[k] - lo[k] + 1);
        out[k] = count > 0 ? vdivs(total[k], (float)count) : pp_read_clamped<true>(im, x, y0 + k);
    }
}
template <bool TILE>
PT_DEV V3 pp_comic(const PostImage& im, int x, int y, V3 color, uint32_t comic_levels) {  // post_process.wgsl:161-186
    float levels = (float)comic_levels;
    V3 q = color * levels + v3(0.5f, 0.5f, 0.5f);
    V3 quantized = vdivs(v3(__builtin_floorf(q.x), __builtin_floorf(q.y), __builtin_floorf(q.z)), levels);
    const float sxk[9] = {-1.0f, 0.0f, 1.0f, -2.0f, 0.0f, 2.0f, -1.0f, 0.0f, 1.0f};
    const float syk[9] = {-1.0f, -2.0f, -1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 2.0f, 1.0f};
    V3 gx = v3(0.0f, 0.0f, 0.0f), gy = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int ki = 0; ki < 9; ki++) {
        V3 sc = pp_read_clamped<TILE>(im, x + ki % 3 - 1, y + ki / 3 - 1);
        gx = gx + sc * sxk[ki];
        gy = gy + sc * syk[ki];
    }
    float edge = length(gx) + length(gy);
    float edge_mask = edge > 0.3f ? 0.0f : 1.0f;
    return quantized * edge_mask;
}
template <bool TILE>
PT_DEV V3 pp_casting(const PostImage& im, int x, int y) {  // post_process.wgsl:189-211
    const float k[9] = {-2.0f, -1.0f, 0.0f, -1.0f, 1.0f, 1.0f, 0.0f, 1.0f, 2.0f};
    V3 r = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int ki = 0; ki < 9; ki++) r = r + pp_read_clamped<TILE>(im, x + ki % 3 - 1, y + ki / 3 - 1) * k[ki];
    r = r * 0.5f + v3(0.5f, 0.5f, 0.5f);
    return v3(clampf(r.x, 0.0f, 1.0f), clampf(r.y, 0.0f, 1.0f), clampf(r.z, 0.0f, 1.0f));
}
template <bool TILE>
PT_DEV V3 pp_effect(const PostImage& im, V3 color, int x, int y, uint32_t effect, const ptx_post_params& pp) {  // :50-86
    switch (effect) {
        case 1u: return v3(1.0f, 1.0f, 1.0f) - color;
        case 2u: {
            float r = dot(color, v3(0.393f, 0.769f, 0.189f)), g = dot(color, v3(0.349f, 0.686f, 0.168f)),
                  b = dot(color, v3(0.272f, 0.534f, 0.131f));
            return v3(clampf(r, 0.0f, 1.0f), clampf(g, 0.0f, 1.0f), clampf(b, 0.0f, 1.0f));
        }
        case 3u: { float lum = dot(color, v3(0.2126f, 0.7152f, 0.0722f)); return v3(lum, lum, lum); }
        case 4u: return pp_fxaa<TILE>(im, x, y);
        case 5u: return pp_oil<TILE>(im, x, y, (int)pp.oil_radius);
        case 6u: { float lum = dot(color, v3(0.2126f, 0.7152f, 0.0722f)); return lum > 0.5f ? v3(1.0f, 1.0f, 1.0f) : v3(0.0f, 0.0f, 0.0f); }
        case 7u: return pp_comic<TILE>(im, x, y, color, pp.comic_levels);
        case 8u: return pp_casting<TILE>(im, x, y);
        default: return color;
    }
}
