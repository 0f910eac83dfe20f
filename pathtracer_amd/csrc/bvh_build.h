// The reference's top-down binned-SAH BVH build (src/accel/bvh.rs:37-213) on the GPU, level by level, with the
// node array and the primitive order of the sequential builder reproduced bit for bit.
//
// What makes that possible:
//  * a node's bounds, its 12 bins per axis and their counts are min / max / integer sums over the node's
//    primitives, so they do not depend on the order the primitives are visited in (atomics are exact here);
//  * the split search (bvh.rs:111-172) is a fixed sequence of f32 operations on those bins, run by one lane;
//  * the in-place partition (bvh.rs:174-186: a failing front element is swapped with the last unexamined one)
//    has a closed form.  With m = number of passing elements of the range [s, e): passing elements below s+m
//    keep their place; the k-th failing element below s+m (ascending) trades with the k-th passing element
//    from the end (descending, at g_k), and lands at g_{k-1} - 1 (g_0 = e); every other failing element moves
//    down by one, except the one at s+m, which lands below the lowest passing element of the right part.
//    Two prefix sums give every element its destination;
//  * the flat array is the pre-order of the tree (bvh.rs:188-213): subtree sizes bottom-up, indices top-down.
//
// Work per level: ranges of at most 8192 primitives are handled by one wave each (bounds, bins in LDS, split
// search with one lane per candidate); larger ranges by workgroups that walk contiguous chunks of positions and
// keep the bounds / bins of the range they are in private until the range changes.  All of it is HBM-bound integer / min-max work: per level ~10 passes over 4-36 B per primitive.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace bvhb {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int NB = 12;                 // constants.rs:10 BVH_NUM_BINS
constexpr uint32_t LEAF_MAX = 4u;      // constants.rs:11 BVH_LEAF_MAX_PRIMS
// Ranges up to this size are handled by one wave.  Measured on 10 M boxes (all levels of a build): 256 -> 41 ms, 2048 -> 26.5 ms,
// 8192 -> 25.6 ms: below it the chunked kernels meet a range boundary in most tiles and fall back to atomics on the records.
constexpr uint32_t SMALL_MAX = 8192u;
constexpr uint32_t BIN_WORDS = 3u * NB * 6u + 3u * NB;  // per axis and bin: min xyz, max xyz; then the counts
constexpr uint32_t POS_INF = 0x7F800000u, NEG_INF = 0xFF800000u;

// One node, in creation (level) order.
struct Rec {
    float mn[3]; uint32_t start;
    float mx[3]; uint32_t end;
    uint32_t left;      // record of the left child (the right one follows it); NONE for a leaf
    uint32_t axis;
    float split;
    uint32_t mid;
    uint32_t size, pre; // subtree size; pre-order index
    uint32_t bin_slot;  // ranges above SMALL_MAX: their bins in global memory
    uint32_t pad;
};
static_assert(sizeof(Rec) == 64, "Rec is one 64-byte line");

// min / max of non-NaN floats through integer atomics on the float's bits (-0.0 is folded into +0.0 first).
__device__ inline void amin_f(float* a, float v) {
    v = v + 0.0f;
    if (v >= 0.0f) atomicMin((int*)a, __float_as_int(v)); else atomicMax((unsigned int*)a, __float_as_uint(v));
}
__device__ inline void amax_f(float* a, float v) {
    v = v + 0.0f;
    if (v >= 0.0f) atomicMax((int*)a, __float_as_int(v)); else atomicMin((unsigned int*)a, __float_as_uint(v));
}
// Waits until this wave's LDS operations have completed (s_waitcnt lgkmcnt(0)).  The compiler puts no such wait
// between LDS atomics that return nothing and a following workgroup barrier; measured on gfx950, a read after the
// barrier can then overtake an atomic that many lanes aim at one address (counts came out short), so every
// "atomics, then barrier, then read" below drains first.
__device__ inline void lds_drain() { __builtin_amdgcn_s_waitcnt(0xC07F); }
// Orders the LDS operations of one wave's lanes against each other.
__device__ inline void wave_sync() {
    lds_drain();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ inline float wave_min_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64)); return v; }
__device__ inline float wave_max_f(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }

__device__ inline float surface_area(const float* mn, const float* mx) {  // aabb.rs:46-49
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
// The bin of a centroid (bvh.rs:129-131: a saturating float -> usize cast, then min with the last bin).
__device__ inline int bin_of(float c, float mn, float inv_extent) {
    float v = (c - mn) * inv_extent;
    return !(v > 0.0f) ? 0 : (v >= (float)NB ? NB - 1 : (int)v);
}
// Adds one primitive to the bins of every axis that has an extent (bvh.rs:116-134).
__device__ inline void bin_add(uint32_t* bins, const float* pmn, const float* pmx, const float* box, const float* cen) {
    for (int axis = 0; axis < 3; axis++) {
        float extent = pmx[axis] - pmn[axis];
        if (fabsf(extent) < 1e-8f) continue;
        float inv_extent = (float)NB / extent;
        int b = bin_of(cen[axis], pmn[axis], inv_extent);
        float* f = (float*)(bins + (axis * NB + b) * 6);
        amin_f(f + 0, box[0]); amin_f(f + 1, box[1]); amin_f(f + 2, box[2]);
        amax_f(f + 3, box[3]); amax_f(f + 4, box[4]); amax_f(f + 5, box[5]);
        atomicAdd(bins + 3 * NB * 6 + axis * NB + b, 1u);
    }
}
// The same for a whole wave whose lanes lie in one range (`valid` lanes hold a primitive; every lane calls).  Where
// all the wave's primitives fall into one bin of an axis - the rule near the top of a tree over primitives that
// arrive in spatial order - the wave combines them first and touches the bin once instead of 64 times.
__device__ inline void bin_add_wave(uint32_t* bins, const float* pmn, const float* pmx, bool valid, const float* box, const float* cen) {
    const float INF = __uint_as_float(POS_INF);
    const unsigned long long live = __ballot(valid);
    if (live == 0ull) return;
    const int leader = __ffsll((long long)live) - 1;
    float bx[6];
    for (int k = 0; k < 6; k++) bx[k] = valid ? box[k] : (k < 3 ? INF : -INF);
    for (int axis = 0; axis < 3; axis++) {
        float extent = pmx[axis] - pmn[axis];
        if (fabsf(extent) < 1e-8f) continue;  // the same for every lane
        float inv_extent = (float)NB / extent;
        const int b = valid ? bin_of(cen[axis], pmn[axis], inv_extent) : -1;
        const int b0 = __shfl(b, leader, 64);
        if (__ballot(valid && b != b0) == 0ull) {
            float r[6];
            for (int k = 0; k < 3; k++) { r[k] = wave_min_f(bx[k]); r[3 + k] = wave_max_f(bx[3 + k]); }
            if ((int)(threadIdx.x & 63u) == leader) {
                float* f = (float*)(bins + (axis * NB + b0) * 6);
                amin_f(f + 0, r[0]); amin_f(f + 1, r[1]); amin_f(f + 2, r[2]);
                amax_f(f + 3, r[3]); amax_f(f + 4, r[4]); amax_f(f + 5, r[5]);
                atomicAdd(bins + 3 * NB * 6 + axis * NB + b0, (uint32_t)__popcll(live));
            }
        } else if (valid) {
            float* f = (float*)(bins + (axis * NB + b) * 6);
            amin_f(f + 0, box[0]); amin_f(f + 1, box[1]); amin_f(f + 2, box[2]);
            amax_f(f + 3, box[3]); amax_f(f + 4, box[4]); amax_f(f + 5, box[5]);
            atomicAdd(bins + 3 * NB * 6 + axis * NB + b, 1u);
        }
    }
}
__device__ inline void bins_clear(uint32_t* bins, uint32_t lane, uint32_t stride) {
    for (uint32_t w = lane; w < BIN_WORDS; w += stride)
        bins[w] = w >= 3u * NB * 6u ? 0u : ((w % 6u) < 3u ? POS_INF : NEG_INF);
}
// bvh.rs:111-172 after the binning: both sweeps and the cost comparison, in the reference's order.
__device__ inline void find_split(const uint32_t* bins, const float* pmn, const float* pmx, uint32_t& best_axis, float& best_split) {
    const float INF = __uint_as_float(POS_INF);
    float best_cost = INF;
    best_axis = 0u;
    best_split = 0.0f;
    for (int axis = 0; axis < 3; axis++) {
        float mn = pmn[axis], mx = pmx[axis];
        float extent = mx - mn;
        if (fabsf(extent) < 1e-8f) continue;
        const float* bb = (const float*)(bins + axis * NB * 6);
        const uint32_t* bc = bins + 3 * NB * 6 + axis * NB;
        float right_area[NB - 1];
        uint32_t right_count[NB - 1];
        float rmn[3] = {INF, INF, INF}, rmx[3] = {-INF, -INF, -INF};
        uint32_t rc = 0;
        for (int i = NB - 1; i >= 1; i--) {
            for (int k = 0; k < 3; k++) { rmn[k] = fminf(rmn[k], bb[i * 6 + k]); rmx[k] = fmaxf(rmx[k], bb[i * 6 + 3 + k]); }
            rc += bc[i];
            right_area[i - 1] = surface_area(rmn, rmx);
            right_count[i - 1] = rc;
        }
        float lmn[3] = {INF, INF, INF}, lmx[3] = {-INF, -INF, -INF};
        uint32_t lc = 0;
        float bin_width = extent / (float)NB;
        for (int i = 0; i < NB - 1; i++) {
            for (int k = 0; k < 3; k++) { lmn[k] = fminf(lmn[k], bb[i * 6 + k]); lmx[k] = fmaxf(lmx[k], bb[i * 6 + 3 + k]); }
            lc += bc[i];
            if (lc == 0u || right_count[i] == 0u) continue;
            float cost = (float)lc * surface_area(lmn, lmx) + (float)right_count[i] * right_area[i];
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = (uint32_t)axis;
                best_split = mn + (float)(i + 1) * bin_width;
            }
        }
    }
}

// find_split with one lane per candidate (axis, boundary): lane = axis * (NB - 1) + i evaluates the split after
// bin i.  min / max and the integer counts do not depend on the order of the sweep, the cost expression is the
// reference's, and the sequential search keeps the FIRST candidate that reaches the lowest cost, which is the
// lowest lane among equal costs.  A cost that is not below +inf (or is a NaN) is never taken, as in the loop.
__device__ inline void find_split_wave(const uint32_t* bins, const float* pmn, const float* pmx, uint32_t lane, uint32_t& best_axis, float& best_split) {
    const float INF = __uint_as_float(POS_INF);
    float cost = INF, split = 0.0f;
    if (lane < 3u * (NB - 1)) {
        const int axis = (int)lane / (NB - 1), i = (int)lane % (NB - 1);
        const float mn = pmn[axis], extent = pmx[axis] - mn;
        if (!(fabsf(extent) < 1e-8f)) {
            const float* bb = (const float*)(bins + axis * NB * 6);
            const uint32_t* bc = bins + 3 * NB * 6 + axis * NB;
            float lmn[3] = {INF, INF, INF}, lmx[3] = {-INF, -INF, -INF}, rmn[3] = {INF, INF, INF}, rmx[3] = {-INF, -INF, -INF};
            uint32_t lc = 0u, rc = 0u;
            for (int b = 0; b < NB; b++) {
                const bool left = b <= i;
                for (int k = 0; k < 3; k++) {
                    const float lo = bb[b * 6 + k], hi = bb[b * 6 + 3 + k];
                    if (left) { lmn[k] = fminf(lmn[k], lo); lmx[k] = fmaxf(lmx[k], hi); }
                    else { rmn[k] = fminf(rmn[k], lo); rmx[k] = fmaxf(rmx[k], hi); }
                }
                if (left) lc += bc[b]; else rc += bc[b];
            }
            if (lc != 0u && rc != 0u) {
                const float c = (float)lc * surface_area(lmn, lmx) + (float)rc * surface_area(rmn, rmx);
                if (c < INF) cost = c;
                const float bin_width = extent / (float)NB;
                split = mn + (float)(i + 1) * bin_width;
            }
        }
    }
    uint32_t who = cost < INF ? lane : 64u;
    for (int o = 32; o > 0; o >>= 1) {
        const float oc = __shfl_xor(cost, o, 64);
        const uint32_t ow = (uint32_t)__shfl_xor((int)who, o, 64);
        if (oc < cost || (oc == cost && ow < who)) { cost = oc; who = ow; }
    }
    if (who >= 64u) { best_axis = 0u; best_split = 0.0f; return; }
    best_axis = who / (uint32_t)(NB - 1);
    best_split = __shfl(split, (int)who, 64);
}

// ---- per level, ranges of at most SMALL_MAX primitives: one wave each ----------------------------------------------
__global__ void __launch_bounds__(256) seg_small_kernel(Rec* recs, uint32_t base, uint32_t nseg, const uint32_t* idx,
                                                        const float* aabb, const float* cent) {
    __shared__ uint32_t lds_bins[4][BIN_WORDS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t s_i = blockIdx.x * 4u + wave;
    const bool valid = s_i < nseg;
    Rec* r = recs + base + (valid ? s_i : 0u);
    const uint32_t start = r->start, end = r->end, count = end - start;
    const bool mine = valid && count <= SMALL_MAX;
    const float INF = __uint_as_float(POS_INF);
    float mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
    if (mine)
        for (uint32_t k = start + lane; k < end; k += 64u) {
            const float* b = aabb + (size_t)idx[k] * 6;
            for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], b[a]); mx[a] = fmaxf(mx[a], b[3 + a]); }
        }
    for (int a = 0; a < 3; a++) { mn[a] = wave_min_f(mn[a]); mx[a] = wave_max_f(mx[a]); }
    const bool inner = mine && count > LEAF_MAX;
    uint32_t* bins = lds_bins[wave];
    // a wave owns its bins: its LDS operations complete in order, no workgroup barrier is needed
    uint32_t axis = 0u;
    float split = 0.0f;
    if (inner) {
        bins_clear(bins, lane, 64u);
        wave_sync();
        for (uint32_t k = start + lane; k < end; k += 64u) {
            const uint32_t p = idx[k];
            bin_add(bins, mn, mx, aabb + (size_t)p * 6, cent + (size_t)p * 3);
        }
        wave_sync();
        find_split_wave(bins, mn, mx, lane, axis, split);
    }
    if (mine && lane == 0u) {
        for (int a = 0; a < 3; a++) { r->mn[a] = mn[a]; r->mx[a] = mx[a]; }
        if (inner) {
            r->axis = axis;
            r->split = split;
        }
    }
}

// ---- per level, larger ranges ----------------------------------------------------------------------------------------
// Every workgroup walks a contiguous chunk of positions in tiles of 256.  Ranges are contiguous, so a tile whose
// first and last position lie in one range lies in it entirely; while consecutive tiles stay in one range the
// workgroup accumulates privately (registers for the bounds, LDS for the bins) and touches the range's record
// once, when the range changes or the chunk ends.  Tiles across a range boundary go to the record directly.
__device__ inline bool is_big(const Rec& r) { return r.end - r.start > SMALL_MAX; }

__global__ void __launch_bounds__(256) bounds_big_kernel(uint32_t n, uint32_t chunk, const uint32_t* seg_of, Rec* recs, const uint32_t* idx, const float* aabb) {
    __shared__ float red[6];
    const float INF = __uint_as_float(POS_INF);
    const uint32_t c0 = blockIdx.x * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
    uint32_t cur = NONE;
    float mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
    for (uint32_t t0 = c0;; t0 += 256u) {
        const bool done = t0 >= c1;
        uint32_t sa = NONE;
        bool uni = false;
        if (!done) {
            const uint32_t last = t0 + 255u < c1 ? t0 + 255u : c1 - 1u;
            sa = seg_of[t0];
            uni = sa != NONE && sa == seg_of[last] && is_big(recs[sa]);
        }
        if (cur != NONE && !(uni && sa == cur)) {  // the private bounds go to the record (every lane takes this branch together)
            if (threadIdx.x < 6u) red[threadIdx.x] = threadIdx.x < 3u ? INF : -INF;
            lds_drain();
            __syncthreads();
            for (int a = 0; a < 3; a++) { mn[a] = wave_min_f(mn[a]); mx[a] = wave_max_f(mx[a]); }
            if ((threadIdx.x & 63u) == 0u)
                for (int a = 0; a < 3; a++) { amin_f(&red[a], mn[a]); amax_f(&red[3 + a], mx[a]); }
            lds_drain();
            __syncthreads();
            if (threadIdx.x < 3u) amin_f(&recs[cur].mn[threadIdx.x], red[threadIdx.x]);
            else if (threadIdx.x < 6u) amax_f(&recs[cur].mx[threadIdx.x - 3u], red[threadIdx.x]);
            lds_drain();
            __syncthreads();
            for (int a = 0; a < 3; a++) { mn[a] = INF; mx[a] = -INF; }
            cur = NONE;
        }
        if (done) break;
        const uint32_t i = t0 + threadIdx.x;
        if (uni) {
            cur = sa;
            if (i < c1) {
                const float* b = aabb + (size_t)idx[i] * 6;
                for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], b[a]); mx[a] = fmaxf(mx[a], b[3 + a]); }
            }
        } else {
            uint32_t seg = i < c1 ? seg_of[i] : NONE;
            if (seg != NONE && !is_big(recs[seg])) seg = NONE;
            float bn[3] = {INF, INF, INF}, bx[3] = {-INF, -INF, -INF};
            if (seg != NONE) {
                const float* b = aabb + (size_t)idx[i] * 6;
                for (int a = 0; a < 3; a++) { bn[a] = b[a]; bx[a] = b[3 + a]; }
            }
            const uint32_t first = (uint32_t)__shfl((int)seg, 0, 64);
            if (__all(seg == first)) {
                if (first != NONE) {
                    for (int a = 0; a < 3; a++) { bn[a] = wave_min_f(bn[a]); bx[a] = wave_max_f(bx[a]); }
                    if ((threadIdx.x & 63u) == 0u)
                        for (int a = 0; a < 3; a++) { amin_f(&recs[seg].mn[a], bn[a]); amax_f(&recs[seg].mx[a], bx[a]); }
                }
            } else if (seg != NONE) {
                for (int a = 0; a < 3; a++) { amin_f(&recs[seg].mn[a], bn[a]); amax_f(&recs[seg].mx[a], bx[a]); }
            }
        }
    }
}
__global__ void __launch_bounds__(256) bins_clear_kernel(uint32_t* bins, uint32_t nslots) {
    const uint32_t slot = blockIdx.x;
    if (slot < nslots) bins_clear(bins + (size_t)slot * BIN_WORDS, threadIdx.x, 256u);
}
__global__ void __launch_bounds__(256) bins_big_kernel(uint32_t n, uint32_t chunk, const uint32_t* seg_of, const Rec* recs, const uint32_t* idx,
                                                       const float* aabb, const float* cent, uint32_t* bins) {
    __shared__ uint32_t lds_bins[BIN_WORDS];
    const uint32_t c0 = blockIdx.x * chunk, c1 = c0 + chunk < n ? c0 + chunk : n;
    uint32_t cur = NONE;
    for (uint32_t t0 = c0;; t0 += 256u) {
        const bool done = t0 >= c1;
        uint32_t sa = NONE;
        bool uni = false;
        if (!done) {
            const uint32_t last = t0 + 255u < c1 ? t0 + 255u : c1 - 1u;
            sa = seg_of[t0];
            uni = sa != NONE && sa == seg_of[last] && is_big(recs[sa]);
        }
        if (cur != NONE && !(uni && sa == cur)) {  // the private bins go to the range's bins
            lds_drain();
            __syncthreads();
            uint32_t* g = bins + (size_t)recs[cur].bin_slot * BIN_WORDS;
            for (uint32_t w = threadIdx.x; w < 3u * NB; w += 256u) {  // one lane per (axis, bin)
                const uint32_t c = lds_bins[3 * NB * 6 + w];
                if (c == 0u) continue;
                float* f = (float*)(g + w * 6u);
                const float* l = (const float*)(lds_bins + w * 6u);
                amin_f(f + 0, l[0]); amin_f(f + 1, l[1]); amin_f(f + 2, l[2]);
                amax_f(f + 3, l[3]); amax_f(f + 4, l[4]); amax_f(f + 5, l[5]);
                atomicAdd(g + 3 * NB * 6 + w, c);
            }
            lds_drain();
            __syncthreads();
            cur = NONE;
        }
        if (done) break;
        const uint32_t i = t0 + threadIdx.x;
        if (uni) {
            if (cur == NONE) {
                bins_clear(lds_bins, threadIdx.x, 256u);
                lds_drain();
                __syncthreads();
                cur = sa;
            }
            const Rec& r = recs[sa];
            const uint32_t p = i < c1 ? idx[i] : 0u;
            bin_add_wave(lds_bins, r.mn, r.mx, i < c1, aabb + (size_t)p * 6, cent + (size_t)p * 3);
        } else {
            uint32_t seg = i < c1 ? seg_of[i] : NONE;
            if (seg != NONE && !is_big(recs[seg])) seg = NONE;
            if (seg != NONE) {
                const Rec& r = recs[seg];
                const uint32_t p = idx[i];
                bin_add(bins + (size_t)r.bin_slot * BIN_WORDS, r.mn, r.mx, aabb + (size_t)p * 6, cent + (size_t)p * 3);
            }
        }
    }
}
__global__ void __launch_bounds__(256) split_big_kernel(Rec* recs, uint32_t base, uint32_t nseg, const uint32_t* bins) {
    const uint32_t s_i = blockIdx.x * 256u + threadIdx.x;
    if (s_i >= nseg) return;
    Rec* r = recs + base + s_i;
    if (!is_big(*r)) return;
    uint32_t axis;
    float split;
    find_split(bins + (size_t)r->bin_slot * BIN_WORDS, r->mn, r->mx, axis, split);
    r->axis = axis;
    r->split = split;
}

// ---- the partition ----------------------------------------------------------------------------------------------------
__device__ inline bool is_inner(const Rec& r) { return r.end - r.start > LEAF_MAX; }
__global__ void __launch_bounds__(256) flag_kernel(uint32_t n, const uint32_t* seg_of, const Rec* recs, const uint32_t* idx,
                                                   const float* cent, uint32_t* pass) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t seg = seg_of[i];
    uint32_t p = 0u;
    if (seg != NONE && is_inner(recs[seg])) p = cent[(size_t)idx[i] * 3 + recs[seg].axis] < recs[seg].split ? 1u : 0u;
    pass[i] = p;
}
// P = exclusive prefix sum of `pass` (P[n] = total).  Records, per range, where its k-th exchanged elements sit.
__global__ void __launch_bounds__(256) pair_kernel(uint32_t n, const uint32_t* seg_of, const Rec* recs, const uint32_t* pass,
                                                   const uint32_t* P, uint32_t* tmpX, uint32_t* tmpY) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t seg = seg_of[i];
    if (seg == NONE || !is_inner(recs[seg])) return;
    const uint32_t s = recs[seg].start, e = recs[seg].end, m = P[e] - P[s];
    if (i < s + m) {
        if (!pass[i]) tmpX[s + (i - s) - (P[i] - P[s])] = i;          // k-th failing element of the left part, ascending
    } else if (pass[i]) {
        tmpY[s + (P[e] - P[i + 1])] = i;                              // k-th passing element of the right part, descending
    }
}
__global__ void __launch_bounds__(256) move_kernel(uint32_t n, const uint32_t* seg_of, const Rec* recs, const uint32_t* pass,
                                                   const uint32_t* P, const uint32_t* tmpX, const uint32_t* tmpY,
                                                   const uint32_t* idx_in, uint32_t* idx_out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t seg = seg_of[i];
    uint32_t dest = i;
    if (seg != NONE && is_inner(recs[seg])) {
        const uint32_t s = recs[seg].start, e = recs[seg].end, m = P[e] - P[s];
        const uint32_t K = m - (P[s + m] - P[s]);  // failing elements of the left part = passing elements of the right part
        if (i < s + m) {
            if (!pass[i]) {
                const uint32_t k = (i - s) - (P[i] - P[s]);
                dest = (k == 0u ? e : tmpY[s + k - 1u]) - 1u;
            }
        } else if (pass[i]) {
            dest = tmpX[s + (P[e] - P[i + 1])];
        } else if (i == s + m) {
            dest = (K > 0u ? tmpY[s + K - 1u] : e) - 1u;
        } else {
            dest = i - 1u;
        }
    }
    idx_out[dest] = idx_in[i];
}

// ---- next level --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) inner_flag_kernel(const Rec* recs, uint32_t base, uint32_t nseg, uint32_t* flag) {
    const uint32_t s_i = blockIdx.x * 256u + threadIdx.x;
    if (s_i < nseg) flag[s_i] = is_inner(recs[base + s_i]) ? 1u : 0u;
}
__device__ inline void rec_init(Rec* c, uint32_t start, uint32_t end, uint32_t* big_counter) {
    c->mn[0] = c->mn[1] = c->mn[2] = __uint_as_float(POS_INF);
    c->mx[0] = c->mx[1] = c->mx[2] = __uint_as_float(NEG_INF);
    c->start = start; c->end = end;
    c->left = NONE; c->axis = 0u; c->split = 0.0f; c->mid = 0u; c->size = 1u; c->pre = 0u; c->pad = 0u;
    c->bin_slot = end - start > SMALL_MAX ? atomicAdd(big_counter, 1u) : 0u;
}
// rank = exclusive prefix sum of the inner flags: the children of the r-th inner range are records next_base + 2r, + 2r + 1.
__global__ void __launch_bounds__(256) children_kernel(Rec* recs, uint32_t base, uint32_t nseg, uint32_t next_base, const uint32_t* rank,
                                                       const uint32_t* P, uint32_t* big_counter) {
    const uint32_t s_i = blockIdx.x * 256u + threadIdx.x;
    if (s_i >= nseg) return;
    Rec* r = recs + base + s_i;
    if (!is_inner(*r)) { r->left = NONE; return; }
    const uint32_t s = r->start, e = r->end, raw_mid = s + (P[e] - P[s]);
    const uint32_t mid = (raw_mid == s || raw_mid == e) ? s + (e - s) / 2u : raw_mid;  // bvh.rs:88-92: (start + end) / 2
    const uint32_t left = next_base + 2u * rank[s_i];
    r->left = left;
    r->mid = mid;
    rec_init(recs + left, s, mid, big_counter);
    rec_init(recs + left + 1u, mid, e, big_counter);
}
__global__ void __launch_bounds__(256) mark_kernel(uint32_t n, uint32_t* seg_of, const Rec* recs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t seg = seg_of[i];
    if (seg == NONE) return;
    const Rec& r = recs[seg];
    seg_of[i] = r.left == NONE ? NONE : r.left + (i >= r.mid ? 1u : 0u);
}

// ---- flat array --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) size_kernel(Rec* recs, uint32_t base, uint32_t nseg) {
    const uint32_t s_i = blockIdx.x * 256u + threadIdx.x;
    if (s_i >= nseg) return;
    Rec* r = recs + base + s_i;
    r->size = r->left == NONE ? 1u : 1u + recs[r->left].size + recs[r->left + 1u].size;
}
__global__ void __launch_bounds__(256) preorder_kernel(Rec* recs, uint32_t base, uint32_t nseg) {
    const uint32_t s_i = blockIdx.x * 256u + threadIdx.x;
    if (s_i >= nseg) return;
    const Rec* r = recs + base + s_i;
    if (r->left == NONE) return;
    recs[r->left].pre = r->pre + 1u;
    recs[r->left + 1u].pre = r->pre + 1u + recs[r->left].size;
}
struct FlatNode { float mn[3]; uint32_t left_or_prim; float mx[3]; uint32_t prim_count; };  // GpuBvhNode, bvh.rs:12-19
__global__ void __launch_bounds__(256) emit_kernel(const Rec* recs, uint32_t nrec, FlatNode* out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= nrec) return;
    const Rec& r = recs[k];
    FlatNode o;
    for (int a = 0; a < 3; a++) { o.mn[a] = r.mn[a] + 0.0f; o.mx[a] = r.mx[a] + 0.0f; }  // a zero bound leaves as +0.0
    if (r.left == NONE) { o.left_or_prim = r.start; o.prim_count = r.end - r.start; }
    else { o.left_or_prim = recs[r.left + 1u].pre; o.prim_count = 0u; }
    out[r.pre] = o;
}

// ---- helpers -----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prep_kernel(uint32_t n, const float* aabb, float* cent, uint32_t* idx, uint32_t* seg_of, uint32_t* has_nan) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float* b = aabb + (size_t)i * 6;
    bool bad = false;
    for (int a = 0; a < 6; a++) bad = bad || b[a] != b[a];
    if (bad) atomicOr(has_nan, 1u);  // the integer min / max above order every value except NaN
    for (int a = 0; a < 3; a++) cent[(size_t)i * 3 + a] = (b[a] + b[3 + a]) * 0.5f;  // aabb.rs:51-53
    idx[i] = i;
    seg_of[i] = 0u;
}
__global__ void root_kernel(Rec* recs, uint32_t n, uint32_t* big_counter) { rec_init(recs, 0u, n, big_counter); }

// Exclusive prefix sum of n 32-bit counts into out[0..n] (out[n] = total): 2048 items per workgroup, the
// workgroup totals scanned by one workgroup.
constexpr uint32_t SCAN_ITEMS = 8u, SCAN_TILE = 256u * SCAN_ITEMS;
__device__ inline uint32_t block_exclusive(uint32_t v, uint32_t* total) {
    __shared__ uint32_t sh[256];
    const uint32_t t = threadIdx.x;
    sh[t] = v;
    lds_drain();
    __syncthreads();
    for (uint32_t o = 1u; o < 256u; o <<= 1) {
        uint32_t add = t >= o ? sh[t - o] : 0u;
        lds_drain();
        __syncthreads();
        sh[t] += add;
        lds_drain();
        __syncthreads();
    }
    const uint32_t incl = sh[t];
    if (total) *total = sh[255];
    lds_drain();
    __syncthreads();
    return incl - v;
}
__global__ void __launch_bounds__(256) scan_sums_kernel(const uint32_t* in, uint32_t n, uint32_t* sums) {
    const uint32_t b0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v = 0u;
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) if (b0 + k < n) v += in[b0 + k];
    uint32_t total;
    (void)block_exclusive(v, &total);
    if (threadIdx.x == 0u) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256) scan_top_kernel(uint32_t* sums, uint32_t nblocks, uint32_t* grand) {
    uint32_t carry = 0u;
    for (uint32_t b = 0; b < nblocks; b += 256u) {
        const uint32_t k = b + threadIdx.x;
        const uint32_t v = k < nblocks ? sums[k] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive(v, &total);
        if (k < nblocks) sums[k] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0u) *grand = carry;
}
__global__ void __launch_bounds__(256) scan_final_kernel(const uint32_t* in, uint32_t n, const uint32_t* sums, uint32_t* out) {
    const uint32_t b0 = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t item[SCAN_ITEMS], v = 0u;
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) { item[k] = b0 + k < n ? in[b0 + k] : 0u; v += item[k]; }
    uint32_t run = sums[blockIdx.x] + block_exclusive(v, nullptr);
    for (uint32_t k = 0; k < SCAN_ITEMS; k++) {
        if (b0 + k < n) out[b0 + k] = run;
        run += item[k];
    }
}

}  // namespace bvhb
