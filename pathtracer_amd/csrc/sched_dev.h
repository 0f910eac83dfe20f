// Wave-scheduled form of the path state machine.
//
// In trace_body (ptx355.hip) every lane of a wave runs the same phase at the same time: closest-hit
// traversal, shading, occlusion traversal, shading.  A phase lasts as long as its slowest lane, so a wave
// whose lanes need 5, 40 and 300 traversal steps (a mesh) or 0, 10 and 256 marching steps (a fractal) runs
// most instructions for a handful of lanes.  Here each lane carries its own state,
//     WALK   walking the BVH (at an inner node, or at a closed-form leaf primitive),
//     MARCH  inside one of the ray marchers (torus, Moebius, Mandelbulb, Julia), one loop iteration at a time,
//     POST   traversal finished: plane list, CSG retry, shading, next ray or next sample,
// and the wave repeatedly picks, by ballot, the state most of its lanes are in and runs ONE step of it for
// those lanes.  Lanes that finish a long phase early move on and come back with new rays, so the lanes pile up
// in front of the expensive steps instead of idling behind the slowest of them.  The arithmetic a lane performs,
// and the order it performs it in, are exactly those of trace_body (same functions, same expression trees), so
// results are bit-identical; only the interleaving between lanes changes.
//
// Reference: the per-invocation control flow of src/shaders/wgsl/path_trace.wgsl:56-183 and bvh.wgsl:143-362.
#pragma once
#include "trace_dev.h"

#define PTX_BLOCK 256            // lanes per workgroup of the trace kernels
#define PTX_WAVES_PER_BLOCK 4

// Kernel arguments of the trace kernels (both families).
struct TraceParams {
    DScene sc;
    ptx_camera cam;
    uint32_t first_frame, first_sample, n;
    uint32_t tile_x0, tile_y0, tile_w, tile_h;
    uint32_t blocks_x, n_blocks;        // 16x16-pixel blocks across the tile; blocks this launch runs
    uint32_t part_rank, part_world;      // this launch owns tile blocks part_rank, part_rank + part_world, ...
    float4* accum;
    uint32_t* out_rgba8;
    uint32_t* primary_ids;
    float* m2;            // null unless moments are enabled
    StackEntry* stack_spill;  // per-lane stack entries beyond the LDS part (null when none are needed)
    struct WaveState* wave_state;  // per wave of the grid: its reserved jobs and its path tally (written by the kernel only)
    float* dbg_radiance;  // null unless capture is enabled
    uint32_t dbg_capacity;
    unsigned long long* stats;  // paths, bounces, shadow rays
    uint32_t* job_counter;  // next unclaimed 64-pixel tile of jobs (launch-wide)
    uint32_t claim_tile;    // jobs a wave reserves from job_counter at a time: 64, or 1 (just what its lanes need at that moment)
    uint32_t total_jobs;    // n_blocks * 256
    // The jobs behind the lanes' first ones (job_first) are reserved from claim_ranges counters, 256 bytes apart: counter r hands
    // out jobs job_first + r * range_len .. + range_len (one address serves 88 M atomics a second, sixteen serve sixteen times as many)
    uint32_t job_first, claim_ranges, range_len;
    uint32_t claim_shift;   // 0, or: a reservation is the jobs that were left at the wave's previous one >> claim_shift (at least claim_tile)
    uint32_t claim_period;  // 0, or: a wave sizes its reservations (64 to 512 jobs) so that they come about this many 10 ns ticks apart
    uint32_t claim_cap_shift; // with claim_period: 0, or a reservation is at most (jobs left >> claim_cap_shift) * 64 (at least 64)
    uint32_t share_walk, share_post, share_march;  // scheduled kernels: a state is served from 1/share of the live lanes
    uint32_t walk_single;   // scheduled kernels, WALK mode: 0 a step walks to a leaf primitive and tests it; 1 a step is one node visit and
                            // lanes at leaf primitives wait in LS_LEAF (deep trees); 2 a step walks to a leaf primitive, tests wait in LS_LEAF
    uint32_t share_leaf;    // walk_single: lanes at a leaf primitive are served together from 1/share of the live lanes
    // Sample-granular jobs (scheduled fractal kernels, split_n > 0): job j is sample j % split_n of pixel job
    // j / split_n -- the 64 consecutive jobs a wave reserves at a time are samples of one pixel (or of neighbours), whose
    // marches are alike: fractals 1080p, 128 spp per launch, 281 -> 302 Mpaths/s against single claims; the same reserve
    // spread over the frame measured 278 -- a finished sample's radiance goes to staging[sample * frame_pixels + pixel] and the fold
    // kernel folds the samples of each pixel into its running mean in sample order afterwards.
    uint32_t split_n, frame_pixels;
    float4* staging;
    // Tolerance mode of the fractal kernels (the arith_mode option): marches of rays other than camera rays evaluate the distance
    // estimators with the hardware transcendentals (mandelbulb_iterate_fast).  0: every march is exact.
    uint32_t frac_fast;
    // Sorted path pools (ptx_trace_kernel_lds_sorted): per workgroup two buffers of pool_m path records, their keys, and the
    // order the current buffer is traced in.
    struct PoolPath* pool;
    uint8_t* pool_keys;     // < 128
    uint16_t* pool_order;   // pool_m <= 16384
    uint32_t pool_m;
    uint32_t pool_guided;   // 0, or: a top-up reserves at most 1/(pool_guided * workgroups) of the jobs left (at least 256)
    uint32_t job_transposed; // sample jobs of the pools: consecutive pixel jobs are one pixel position of consecutive blocks (job_pixel)
    uint32_t pool_finish;   // with no job left to reserve, a pool of up to this many paths is traced to the end without further rounds
};
// A path between two bounces: 64 bytes = four 16-byte accesses.
struct alignas(16) PoolPath {
    float o[3]; uint32_t rng;
    float d[3]; uint32_t bounce;
    float thr[3]; uint32_t job;
    float rad[3]; uint32_t primary;
};
static_assert(sizeof(PoolPath) == 64, "PoolPath");

// LS_LEAF (one-visit-per-step mode only): the lane stands at a leaf primitive it has yet to test.
enum : uint32_t { LS_IDLE = 0, LS_WALK = 1, LS_POST = 2, LS_TORUS = 3, LS_MEBIUS = 4, LS_MANDEL = 5, LS_JULIA = 6, LS_LEAF = 7, LS_COUNT = 8 };

// Per-lane LDS slots (lane-interleaved: slot f of this lane at base[f * 64]).
#define SC_PIXEL_FIELDS 8u   // mean xyz, M2 xyz, samples finished, path-bounce tally
#define SC_SAMPLE_JOB_FREE 4u  // of those, slots 0-3 that sample jobs of the non-fractal kernels do not have (sched_body)
#define SC_PARK_FIELDS 24u   // below
enum : uint32_t {
    PK_HPOS = 0, PK_N = 3, PK_BASE = 6, PK_FIG = 9, PK_CONTRIB = 10,  // inputs of the BSDF step, parked across the occlusion query
    PK_THR = 13, PK_RAD = 16, PK_RNG = 19, PK_MISC = 20,              // path state; MISC = bounce | csg attempt << 24 | normal ready << 28 | lit << 31
    PK_WO = 21
};
// Kernels with the fractal code: a fractal MARCH is a task whose whole state follows the parked fields in LDS: the ray,
// the marcher's state, the distance estimator's iteration state.  A march is a pure function of that state and the
// figure, so ANY lane of the workgroup may advance it: evaluation after evaluation, marching step after marching step,
// until it hits, misses or (for the four evaluations of a normal) is complete.  A wave that serves an estimator has
// most of its lanes waiting in other states; those lanes take marches of the workgroup's other waves.  Whoever gets to
// a march first claims it (READY -> BUSY), carries it in registers and hands it back finished (DONE*, result in the
// slot) or, when its wave turns to other work, unfinished (READY); only the owning lane acts on the result.  The
// arithmetic, and its order, are those of march_fractal / finish_fractal whoever runs them.
// The slot holds what cannot be recomputed: 19 words per lane (with 8 pixel and 24 parked words that leaves room for three
// workgroups per compute unit on a scene whose tree is one leaf; the fourth wave per SIMD the estimators' registers forbid
// anyway).  Recomputed when a march is claimed, by the very functions that made them: the Julia set's r2 = dot4(z, z)
// (FracIter::s[8]) and the marcher's max_t (march_begin_fractal).
#define SC_FRAC_FIELDS 19u
enum : uint32_t {
    TK_S = 0,        // FracIter::s[0..7]
    TK_WORD = 8,     // estimator iterations done | tolerance mode << 15 | phase << 16 | status << 20: one word, so that a claim is one compare-and-swap
    TK_REF = 9,      // the figure's primitive slot (slot_prim)
    TK_O = 10, TK_D = 13,                  // the ray
    TK_T = 16, TK_PREV = 17, TK_STEP = 18  // MarchState: t (the hit distance when done), prev_d, marching steps done
};
#define TK_STATUS_SHIFT 20u
enum : uint32_t { FS_IDLE = 0, FS_READY = 1, FS_BUSY = 2, FS_DONE_NORMAL = 3, FS_DONE_HIT = 4, FS_DONE_MISS = 5 };
#ifndef FRAC_ITERS_PER_PASS
#define FRAC_ITERS_PER_PASS 24  // estimator iterations a serving wave runs between two rounds of claims
#endif
#ifndef FRAC_GLUE_MIN
#define FRAC_GLUE_MIN 12        // lanes whose evaluation has ended before the code that follows an evaluation is run for them
#endif                          // (fractals 1080p, 128 spp/launch: 6 iterations + once per pass 262 Mpaths/s; 16/16 277, 12/24 281, 8/24 279, 4/16 264)
#ifndef FRAC_ADOPT_MIN
#define FRAC_ADOPT_MIN 8        // lanes of a serving wave without a march before they look for other waves' (the search costs all lanes)
#endif
// What a fractal evaluation is for: a marching step, the second evaluation of an over-relaxed step, or one of
// the four evaluations of the winner's normal (torus.wgsl:54-61 pattern in mandelbulb.wgsl:83-93 / julia.wgsl:83-93).
enum : uint32_t { FP_D = 0, FP_D2 = 1, FP_N0 = 2, FP_N1 = 3, FP_N2 = 4, FP_N3 = 5 };
#define PK_MISC_NORMAL_READY 0x10000000u

struct Trav {
    Ray ray;
    V3 inv_dir;
    float bound;          // closest: nearest hit so far; occlusion: distance to the light
    uint32_t slot, sub;   // closest: best primitive slot / face; occlusion: slot != PT_SLOT_NONE means occluded
    int sp;
    uint32_t cur;
    float cur_t;
    uint32_t pi, pend;    // primitives [pi, pend) of the current leaf are still to be tested
    bool have, shadow;
    MarchState m;
};

// Starts a traversal of tv.ray (inv_dir set by the caller).  bvh.wgsl:236-249 / 308-322.
PT_DEV void trav_begin(const DScene& sc, Trav& tv, bool shadow, float bound) {
    tv.shadow = shadow;
    tv.bound = bound;
    tv.slot = PT_SLOT_NONE;
    tv.sub = 0u;
    tv.sp = 0;
    tv.cur = sc.root_ref;
    tv.cur_t = sc.nprim > 0u ? intersect_aabb(tv.ray.o, tv.inv_dir, sc.root_min, sc.root_max) : PT_MAX_T;
    tv.have = sc.nprim > 0u;
    tv.pi = 0u;
    tv.pend = 0u;
}

// A candidate's result: closest keeps the nearest (bvh.wgsl:268-272), occlusion stops at the first hit in range
// that no subtraction shape removes (bvh.wgsl:334-341).  Returns true when the traversal is over.
PT_DEV bool trav_apply(const DScene& sc, Trav& tv, bool hit, float ht, uint32_t sub) {
    if (!tv.shadow) {
        if (hit && ht < tv.bound && ht > PT_EPS) {
            tv.bound = ht;
            tv.slot = tv.pi;
            tv.sub = sub;
        }
        return false;
    }
    if (hit && ht > PT_EPS && ht < tv.bound - PT_EPS)
        if (sc.nneg == 0u || !inside_any_negative(sc, madd(tv.ray.o, tv.ray.d, ht))) {
            tv.slot = 0u;
            return true;
        }
    return false;
}

// The state of a traversing lane whose current primitive is done: in one-visit-per-step mode a lane with more
// primitives of its leaf to test waits for the next LEAF round, otherwise it walks on.
PT_DEV uint32_t after_prim(const Trav& tv, uint32_t mode) { return mode != 0u && tv.pi < tv.pend ? (uint32_t)LS_LEAF : (uint32_t)LS_WALK; }

// This wave's LDS operations have completed (the compiler puts no such wait before a workgroup-scope release).
PT_DEV void lds_drain() { __builtin_amdgcn_s_waitcnt(0xC07F); }

PT_DEV V3 tet_offset(uint32_t k_index, float k) {  // offsets e.xyy, e.yyx, e.yxy, e.xxx with e = (k, -k)
    switch (k_index) {
        case 0: return v3(k, -k, -k);
        case 1: return v3(-k, -k, k);
        case 2: return v3(-k, k, -k);
        default: return v3(k, k, k);
    }
}
// The r-th set bit of m (r < popcount(m)).
PT_DEV uint32_t nth_set_bit(unsigned long long m, uint32_t r) {
    uint32_t pos = 0u;
#pragma unroll
    for (uint32_t w = 32u; w >= 1u; w >>= 1) {
        const uint32_t c = (uint32_t)__popcll((m >> pos) & ((1ull << w) - 1ull));
        if (r >= c) { r -= c; pos += w; }
    }
    return pos;
}

// A march in registers, while a lane holds it.
struct FracTask {
    Ray ray;
    MarchState m;
    FracIter f;
    uint32_t phase, ref;
    uint32_t fast;  // 1: this march runs in the tolerance mode (set when it is posted; travels in TK_WORD)
    V3 nacc;  // normal tasks: the sum so far (kept in the owner's parked normal while nobody holds the task)
};
// Starts the estimator evaluation at distance t along the task's ray (normal phases: offset by their tetrahedron corner,
// in the figure's unit space): march_fractal's `p * inv_radius`, finish_fractal's `local + o_k`.
PT_DEV void task_begin_eval(const DPrim& g, FracTask& T, float t, uint32_t phase) {
    const float inv_radius = dm::rcp_(g.p[3]);
    V3 p = (madd(T.ray.o, T.ray.d, t) - v3p(g.p)) * inv_radius;
    if (phase >= FP_N0) p = p + tet_offset(phase - FP_N0, 0.5773f * PT_EPS * 2.0f);
    frac_eval_begin(g, p, T.f);
    T.phase = phase;
}
// The owning lane's parked fields, seen from a task slot (the fractal fields follow them).
PT_DEV float* task_park(float* st) { return st - SC_PARK_FIELDS * 64u; }
PT_DEV void task_store(float* st, const FracTask& T, uint32_t status) {
    uint32_t* su = (uint32_t*)st;
#pragma unroll
    for (int k = 0; k < 8; k++) st[(TK_S + k) * 64] = T.f.s[k];
    su[TK_REF * 64] = T.ref;
    st[(TK_O + 0) * 64] = T.ray.o.x; st[(TK_O + 1) * 64] = T.ray.o.y; st[(TK_O + 2) * 64] = T.ray.o.z;
    st[(TK_D + 0) * 64] = T.ray.d.x; st[(TK_D + 1) * 64] = T.ray.d.y; st[(TK_D + 2) * 64] = T.ray.d.z;
    st[TK_T * 64] = T.m.t; st[TK_PREV * 64] = T.m.prev_d;
    su[TK_STEP * 64] = T.m.i;
    if (T.phase > FP_N0) {  // the normal's partial sum (nothing yet in phase N0)
        float* pk = task_park(st);
        pk[(PK_N + 0) * 64] = T.nacc.x; pk[(PK_N + 1) * 64] = T.nacc.y; pk[(PK_N + 2) * 64] = T.nacc.z;
    }
    lds_drain();
    __hip_atomic_store(&su[TK_WORD * 64], T.f.i | (T.fast << 15) | (T.phase << 16) | (status << TK_STATUS_SHIFT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Takes the march at `st` (this lane's own or another lane's) if it is waiting for someone to run it.
PT_DEV bool task_claim(const DScene& sc, float* st, uint32_t march_steps, FracTask& T) {
    uint32_t* su = (uint32_t*)st;
    uint32_t word = __hip_atomic_load(&su[TK_WORD * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((word >> TK_STATUS_SHIFT) != (uint32_t)FS_READY) return false;
    const uint32_t busy = (word & ((1u << TK_STATUS_SHIFT) - 1u)) | ((uint32_t)FS_BUSY << TK_STATUS_SHIFT);
    if (!__hip_atomic_compare_exchange_strong(&su[TK_WORD * 64], &word, busy, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
        return false;
    PT_SITE(48);
#pragma unroll
    for (int k = 0; k < 8; k++) T.f.s[k] = st[(TK_S + k) * 64];
    T.f.i = word & 0x7FFFu;
    T.fast = (word >> 15) & 1u;
    T.phase = (word >> 16) & 0xFu;
    T.ref = su[TK_REF * 64];
    T.ray.o = v3(st[(TK_O + 0) * 64], st[(TK_O + 1) * 64], st[(TK_O + 2) * 64]);
    T.ray.d = v3(st[(TK_D + 0) * 64], st[(TK_D + 1) * 64], st[(TK_D + 2) * 64]);
    T.m.t = st[TK_T * 64]; T.m.prev_d = st[TK_PREV * 64];
    T.m.i = su[TK_STEP * 64];
    // what the slot leaves out, from the functions that made it
    const DPrim& g = slot_prim(sc, T.ref);
    T.f.s[8] = 0.0f;
    if (g.type == FIG_JULIA) {
        V4 z; z.x = T.f.s[0]; z.y = T.f.s[1]; z.z = T.f.s[2]; z.w = T.f.s[3];
        T.f.s[8] = dot4(z, z);
    }
    MarchState m0;
    (void)march_begin_fractal(T.ray, g, march_steps, m0);
    T.m.max_t = m0.max_t;
    T.nacc = v3(0.0f, 0.0f, 0.0f);
    if (T.phase > FP_N0) {
        const float* pk = task_park(st);
        T.nacc = v3(pk[(PK_N + 0) * 64], pk[(PK_N + 1) * 64], pk[(PK_N + 2) * 64]);
    }
    return true;
}
// The owner posts a march of its ray against figure g (primitive slot ref), begun by march_begin_fractal (m), ...
PT_DEV void task_start_march(const DPrim& g, uint32_t ref, const Ray& ray, const MarchState& m, float* frac, uint32_t fast) {
    FracTask T;
    T.ray = ray;
    T.m = m;
    T.ref = ref;
    T.fast = fast;
    T.nacc = v3(0.0f, 0.0f, 0.0f);
    task_begin_eval(g, T, m.t, FP_D);
    task_store(frac, T, FS_READY);
}
// ... or the four evaluations of the normal at the hit at distance t (finish_fractal / tet_normal).
PT_DEV void task_start_normal(const DPrim& g, uint32_t ref, const Ray& ray, float t, float* frac, uint32_t fast) {
    FracTask T;
    T.fast = fast;
    T.ray = ray;
    T.m.t = t; T.m.prev_d = 0.0f; T.m.max_t = 0.0f; T.m.i = 0u;
    T.ref = ref;
    T.nacc = v3(0.0f, 0.0f, 0.0f);
    task_begin_eval(g, T, t, FP_N0);
    task_store(frac, T, FS_READY);
}
// An estimator evaluation of a held march has completed with `value`: what the evaluation was for
// (march_step_fractal's logic around its two sdf calls, or finish_fractal's normal) and the march's next evaluation.
// Returns 0 while the march goes on, else its final status (FS_DONE_*).
PT_DEV uint32_t task_advance(const DScene& sc, FracTask& T, float value, uint32_t march_steps) {
    const DPrim& g = slot_prim(sc, T.ref);
    PT_SITE(42);
    if (T.phase >= FP_N0) {  // tet_normal: normalize(o0 d0 + o1 d1 + o2 d2 + o3 d3), summed left to right
        const uint32_t idx = T.phase - FP_N0;
        V3 term = tet_offset(idx, 0.5773f * PT_EPS * 2.0f) * value;
        T.nacc = idx == 0u ? term : T.nacc + term;
        if (idx == 3u) {
            T.nacc = normalize(T.nacc);
            return FS_DONE_NORMAL;
        }
        task_begin_eval(g, T, T.m.t, T.phase + 1u);
        return 0u;
    }
    const float omega = 1.3f;
    const float radius = g.p[3];
    MarchState& m = T.m;
    int r = MARCH_GO;
    if (T.phase == FP_D) {
        float d = value * radius;
        if (d < 0.0f && m.prev_d > 0.0f) {
            m.t -= m.prev_d * (omega - 1.0f);
            task_begin_eval(g, T, m.t, FP_D2);
            return 0u;
        }
        if (dm::abs_(d) < PT_EPS * m.t * 0.5f) r = MARCH_HIT;
        else {
            m.t += d * omega;
            m.prev_d = d;
            if (m.t > m.max_t) r = MARCH_MISS;
        }
    } else {
        float d2 = value * radius;
        m.t += d2;
        m.prev_d = d2;
    }
    if (r == MARCH_GO) {
        m.i += 1u;
        if (m.i < march_steps) {
            task_begin_eval(g, T, m.t, FP_D);
            return 0u;
        }
        r = MARCH_MISS;
    }
    return r == MARCH_HIT ? (uint32_t)FS_DONE_HIT : (uint32_t)FS_DONE_MISS;
}
// The held march is complete: its result goes where the owning lane looks for it.
PT_DEV void task_finish(float* st, const FracTask& T, uint32_t status) {
    if (status == FS_DONE_NORMAL) {
        float* pk = task_park(st);
        pk[(PK_N + 0) * 64] = T.nacc.x; pk[(PK_N + 1) * 64] = T.nacc.y; pk[(PK_N + 2) * 64] = T.nacc.z;
    } else {
        st[TK_T * 64] = T.m.t;
    }
    lds_drain();
    __hip_atomic_store(&((uint32_t*)st)[TK_WORD * 64], status << TK_STATUS_SHIFT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The owner of a march (a lane in an estimator state) looks at its slot: a finished march is applied to the lane's
// traversal (or, for a normal, flagged as parked) and the lane moves on.  Returns the lane's next state.
PT_DEV uint32_t task_consume(const DScene& sc, Trav& tv, uint32_t kind, float* frac, float* park, uint32_t mode) {
    uint32_t* frac_u = (uint32_t*)frac;
    const uint32_t status = __hip_atomic_load(&frac_u[TK_WORD * 64], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >> TK_STATUS_SHIFT;
    if (status < (uint32_t)FS_DONE_NORMAL) return kind;
    frac_u[TK_WORD * 64] = (uint32_t)FS_IDLE << TK_STATUS_SHIFT;
    if (status == (uint32_t)FS_DONE_NORMAL) {
        ((uint32_t*)park)[PK_MISC * 64] |= PK_MISC_NORMAL_READY;
        return LS_POST;
    }
    if (trav_apply(sc, tv, status == (uint32_t)FS_DONE_HIT, frac[TK_T * 64], 0u)) return LS_POST;
    tv.pi += 1u;
    return after_prim(tv, mode);
}

// One LEAF step: the lane tests the leaf primitive it stands at, or enters its marcher.  Returns the lane's next state.
// frac_fast: TraceParams::frac_fast (fractal kernels).
template <bool FRAC>
PT_DEV uint32_t leaf_step(const DScene& sc, Trav& tv, uint32_t march_steps, uint32_t mode, float* frac, uint32_t frac_fast = 0u) {
    const DPrim& g = sc.prims[tv.pi];
    const uint32_t type = g.type;
    bool hit = false;
    float ht = 0.0f;
    uint32_t sub = 0u;
    if (type == FIG_TORUS) {
        if (march_begin_torus(tv.ray, g, tv.m)) return LS_TORUS;
    } else if (type == FIG_MEBIUS) {
        if (march_begin_mebius(tv.ray, g, tv.m)) return LS_MEBIUS;
    } else if (FRAC && (type == FIG_MANDELBULB || type == FIG_JULIA)) {
        if (march_begin_fractal(tv.ray, g, march_steps, tv.m)) {
            // the tolerance mode never touches a camera ray: its closest hit is the pixel's primary id (the path's bounce
            // count is in the parked word; an occlusion ray belongs to a bounce's shading, not to the id)
            const uint32_t fast = frac_fast != 0u && (frac_fast >= 2u || tv.shadow || (((const uint32_t*)task_park(frac))[PK_MISC * 64] & 0xFFFFFFu) != 0u) ? 1u : 0u;
            task_start_march(g, tv.pi, tv.ray, tv.m, frac, fast);
            return type == FIG_MANDELBULB ? LS_MANDEL : LS_JULIA;
        }
    } else if (type != FIG_SKIP) {
        hit = intersect_closed(tv.ray, tv.inv_dir, g, ht, sub);
    }
    if (trav_apply(sc, tv, hit, ht, sub)) return LS_POST;
    tv.pi += 1u;
    return after_prim(tv, mode);
}

// One WALK step: walk inner nodes until the lane stands at a leaf primitive (same pops, pushes and guards as
// trace_positive_walk / trace_shadow_walk), then test that primitive, or enter its marcher.  Returns the lane's
// next state.  `single` (wave-uniform, for deep trees): a step is ONE node visit, so no lane waits inside the step
// for the lane with the longest way down to a leaf (10 M triangles: +10 %; shallow trees lose 1-5 % to the extra
// scheduling rounds), and a lane that arrives at a leaf goes to LS_LEAF: node visits and primitive tests are
// different code, and in one step a wave would run both for a few lanes each (measured on the 10 M-triangle
// scene: 9.7 lanes per node-pair test, 2.6 per triangle test); served as states of their own, the primitive
// tests run for the lanes that have gathered at leaves.
template <bool FRAC>
PT_DEV uint32_t walk_step(const DScene& sc, Trav& tv, const WalkStack& st, uint32_t march_steps, uint32_t mode, float* frac, uint32_t frac_fast = 0u) {
    const bool single = mode == 1u;
    bool walked = false;
    while (tv.pi >= tv.pend) {
        if (single && walked) return LS_WALK;
        walked = true;
        PT_SITE(1);
        if (!tv.have) {
            if (tv.sp == 0) return LS_POST;
            tv.sp -= 1;
            StackEntry e = stack_get(st, tv.sp);
            tv.cur = e.x;
            tv.cur_t = __uint_as_float(e.y);
        }
        tv.have = false;
        if (tv.shadow ? !(tv.cur_t < tv.bound) : (tv.cur_t >= tv.bound)) continue;
        uint32_t cnt = tv.cur >> PT_REF_BITS, ref = tv.cur & PT_REF_MASK;
        if (cnt > 0u) {
            tv.pi = ref;
            tv.pend = ref + cnt;
        } else {
            PT_SITE(2);
            const DNodePair& nd = sc.pairs[ref];
            float tl = intersect_aabb(tv.ray.o, tv.inv_dir, nd.lmin, nd.lmax);
            float tr = intersect_aabb(tv.ray.o, tv.inv_dir, nd.rmin, nd.rmax);
            uint32_t le = nd.lref | (nd.lcnt << PT_REF_BITS), re = nd.rref | (nd.rcnt << PT_REF_BITS);
            // closest: far child pushed, near child continued, both only if nearer than the best hit;
            // occlusion: left pushed, right continued, unconditionally (bvh.wgsl:283-300 / 346-353).
            bool left_near = !tv.shadow && tl < tr;
            uint32_t far_e = left_near ? re : le, near_e = left_near ? le : re;
            float far_t = left_near ? tr : tl, near_t = left_near ? tl : tr;
            if ((tv.shadow || far_t < tv.bound) && tv.sp < 31) {
                stack_put(st, tv.sp, make_uint2(far_e, __float_as_uint(far_t)));
                tv.sp += 1;
            }
            if ((tv.shadow || near_t < tv.bound) && tv.sp < 31) {
                tv.cur = near_e;
                tv.cur_t = near_t;
                tv.have = true;
            }
        }
    }
    if (mode != 0u && walked) return LS_LEAF;
    return leaf_step<FRAC>(sc, tv, march_steps, mode, frac, frac_fast);
}

// One MARCH step of marcher `kind` (wave-uniform).
template <bool FRAC>
PT_DEV uint32_t march_step(const DScene& sc, Trav& tv, uint32_t kind, uint32_t march_steps, uint32_t mode) {
    const DPrim& g = sc.prims[tv.pi];
    int r;
    if (kind == LS_TORUS) { PT_SITE(40); r = march_step_torus(tv.ray, g, tv.m); }
    else if (kind == LS_MEBIUS) { PT_SITE(41); r = march_step_mebius(tv.ray, g, tv.m); }
    else r = MARCH_MISS;  // the fractal kinds step through frac_step
    if (r == MARCH_GO) return kind;
    if (trav_apply(sc, tv, r == MARCH_HIT, tv.m.t, 0u)) return LS_POST;
    tv.pi += 1u;
    return after_prim(tv, mode);
}

// Pixel jobs.  A lane is not tied to one pixel: when its pixel has all its samples it writes the pixel back and
// claims the next unclaimed one (claim_pixel), so that a wave stays full until the frame runs out
// of pixels, however unequal the pixels are (a Mandelbulb's silhouette costs a hundred times a background
// pixel).  Job j is pixel (j & 255) of this launch's 16x16 block (j >> 8), 8x8 quadrant by quadrant; all the
// samples of a pixel are still folded in order by the one lane that owns it, so results do not depend on which
// lane that is.
struct PixelJob { uint32_t tx, ty, px, py, k; size_t pix; };
PT_DEV bool job_pixel(const TraceParams& P, uint32_t job, PixelJob& J, bool split = false) {
    J.k = 0u;
    if (split) {
        uint32_t pj = job / P.split_n;
        J.k = job - pj * P.split_n;
        if (P.job_transposed != 0u) {
            // Pools reserve thousands of consecutive jobs at a time.  In block order those are a few neighbouring 16x16 blocks, and
            // a workgroup whose blocks lie on the scene's expensive figures ends long after the others.  Here consecutive pixel jobs
            // are the SAME pixel position of consecutive blocks -- a reservation samples the frame at a stride of sixteen pixels, so
            // every pool gets the frame's mix of cheap and costly pixels.  (The pools sort their rays by key anyway: nothing is lost
            // by not starting from neighbouring pixels.)
            const uint32_t p = pj / P.n_blocks;
            pj = ((pj - p * P.n_blocks) << 8) | p;
        }
        job = pj;
    }
    const uint32_t blk = (job >> 8) * P.part_world + P.part_rank, p = job & 255u;
    const uint32_t bx = blk % P.blocks_x, by = blk / P.blocks_x, q = p >> 6, l = p & 63u;
    J.tx = bx * 16u + (q & 1u) * 8u + (l & 7u);
    J.ty = by * 16u + (q >> 1) * 8u + (l >> 3);
    J.px = P.tile_x0 + J.tx;
    J.py = P.tile_y0 + J.ty;
    J.pix = (size_t)J.py * P.cam.width + J.px;
    return J.tx < P.tile_w && J.ty < P.tile_h;
}
// A wave's claim state and path tally: one 32-byte record per wave of the grid in global memory ({next job, end} of the
// jobs the wave has reserved, paths finished), which only that wave touches -- one lane writes, all its lanes read, in the
// wave's program order -- through relaxed device-scope atomics, so that neither the compiler nor the per-CU cache keeps an
// earlier value.  (The trace kernels use every byte of LDS that four workgroups per CU leave: 48 bytes of LDS for this
// state cost the global-scene kernels a workgroup per CU, 12 % on the 10 M-triangle scene.)
struct WaveState {
    uint32_t next, end, paths;  // the jobs it has reserved and not handed out; paths finished
    uint32_t pace;              // paced reservations: when the previous one was made and how large it was (claim_jobs)
    uint32_t range;             // the counter it reserves from
    uint32_t pad[3];
};
static_assert(sizeof(WaveState) == 32, "WaveState");
PT_DEV WaveState* wave_state(const TraceParams& P) { return P.wave_state + (size_t)blockIdx.x * PTX_WAVES_PER_BLOCK + (threadIdx.x >> 6); }
// `ranged`: the kernel reserves from several counters (CLAIM_RANGED, below)
PT_DEV void wave_claim_init(const TraceParams& P, bool ranged) {
    if ((threadIdx.x & 63u) == 0u) {
        WaveState* ws = wave_state(P);
        __hip_atomic_store(&ws->next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws->end, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws->paths, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ranged) {
            const uint32_t wave = blockIdx.x * PTX_WAVES_PER_BLOCK + (threadIdx.x >> 6);
            __hip_atomic_store(&ws->pace, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws->range, P.claim_ranges > 1u ? wave % P.claim_ranges : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// One job number for every lane that calls (the lanes of a wave whose pixels finished together).  Jobs come from the wave's
// reserved range; when that runs out the first calling lane takes a new range from the launch-wide counter with ONE device
// atomic.  Atomics on one address are served one at a time, about 10 ns each on MI355X: with one claim per finished pixel a
// 1080p frame at one sample per launch spent 2.0 of its 2.2 ms there (and as long again on the path tally, which shared the
// counter's line).  With claim_tile == 1 a range is exactly what the calling lanes need -- heavy pixels spread over all waves,
// which serves the fractal scenes (a whole quadrant of silhouette pixels in one wave costs 23 %) and long launches of the
// Cornell box.  Otherwise a range is claim_tile jobs (64: one 8x8 quadrant, so the lanes of a wave stay on neighbouring
// pixels, which serves meshes and the Moebius marcher: +4 % / +18 %), or a size the kernel's claim mode works out (below).
#define PT_JOB_RETRY 0xFFFFFFFFu  // claim_jobs: the counter ran out under this lane's feet, there may be jobs at another
// MODE: what the kernel is compiled for (each costs registers: the wave-scheduled kernels have none to spare, and the lockstep
// kernel's code for the Cornell box is sensitive to all of it) --
// CLAIM_PLAIN   one counter, which starts at the first unassigned job, reservations of claim_tile jobs (the fractal kernels);
// CLAIM_GUIDED  the same with sizes by the jobs left (claim_shift; the other wave-scheduled kernels);
// CLAIM_PACED   the same with sizes by the wave's pace (claim_period; the lockstep kernels);
// CLAIM_RANGED  several counters, which start at zero, sizes by the wave's pace (the lockstep kernel for scenes without lights).
// The host sets the parameters to match (launch_trace).
enum : int { CLAIM_PLAIN = 0, CLAIM_GUIDED = 1, CLAIM_PACED = 2, CLAIM_RANGED = 3 };
// The wave's own pace sets the size of its reservations: its record keeps when the previous one was made (24 bits of the
// 100 MHz clock, which wrap after 168 ms) and how large it was (in 64 jobs, the top byte).  Used up in under 2/3 of claim_period:
// the next is twice as large; it lasted over 3/2: half; claim_tile to 512 jobs, and no more than the jobs left >> claim_cap_shift.
// Scalar integer work only (shifts and compares): the kernels have no vector register to spare here.
PT_DEV uint32_t paced_size(const TraceParams& P, WaveState* ws, uint32_t first_size, uint32_t left, bool leader) {
    const uint32_t now = (uint32_t)wall_clock64() & 0xFFFFFFu;
    const uint32_t rec = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ws->pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    uint32_t units = rec >> 24;
    if (units) {
        const uint32_t dt = (now - rec) & 0xFFFFFFu;
        if (dt * 3u < P.claim_period * 2u) units <<= 1;
        else if (dt * 2u > P.claim_period * 3u) units >>= 1;
        units = units > 8u ? 8u : units;
    } else {
        units = first_size >> 6;
    }
    if (P.claim_cap_shift) {
        const uint32_t cap = left >> P.claim_cap_shift;
        units = units > cap ? cap : units;
    }
    const uint32_t least = first_size >> 6;  // never below the fixed size of this launch (128 jobs for frames of 4 M pixels and more)
    units = units < least ? least : units;
    if (leader) __hip_atomic_store(&ws->pace, units << 24 | now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return units << 6;
}
template <int MODE>
PT_DEV uint32_t claim_jobs(const TraceParams& P) {
    const unsigned long long need = __ballot(true);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cnt = (uint32_t)__popcll(need), rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
    const uint32_t leader = (uint32_t)__ffsll((long long)need) - 1u;
    if (P.claim_tile <= 1u) {  // no reserve to keep: the wave's state is not touched
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(P.job_counter, cnt);
        return (MODE == CLAIM_RANGED ? P.job_first : 0u) + (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader) + rank;
    }
    WaveState* ws = wave_state(P);
    // the record is the wave's own: every lane reads the same words, kept in scalar registers from here on
    const uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ws->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ws->end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const uint32_t avail = end - next;
    if (cnt <= avail) {
        if (lane == leader) __hip_atomic_store(&ws->next, next + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return next + rank;
    }
    const uint32_t want = cnt - avail;
    uint32_t chunk = P.claim_tile;
    if (MODE != CLAIM_RANGED) {
        // `end` is where the counter stood after this wave's previous reservation
        if (MODE == CLAIM_GUIDED && P.claim_shift) {
            const uint32_t share = ((P.total_jobs - (end < P.total_jobs ? end : P.total_jobs)) >> P.claim_shift) & ~63u;
            chunk = share > chunk ? share : chunk;
        }
        if (MODE == CLAIM_PACED && P.claim_period) chunk = paced_size(P, ws, chunk, P.total_jobs - (end < P.total_jobs ? end : P.total_jobs), lane == leader);
        chunk = want > chunk ? want : chunk;
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(P.job_counter, chunk);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
        if (lane == leader) {
            __hip_atomic_store(&ws->next, base + want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws->end, base + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return rank < avail ? next + rank : base + (rank - avail);
    }
    // Which counter the wave reserves from: its own word.  Which counters have run out: a word all waves share, behind the
    // counters -- read here (a load, not an atomic), so that at the end of the launch a wave sees every bit set and asks no more
    // (finding out counter by counter cost every wave sixteen failed atomics at the end of every launch).
    uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&ws->range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    uint32_t* out_mask = P.job_counter + P.claim_ranges * 64u;
    const uint32_t all = (1u << P.claim_ranges) - 1u;
    const uint32_t open = ~(uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(out_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & all;
    uint32_t base = P.total_jobs, got = 0u;
    if (open != 0u) {
        // the first counter at or after r that has not run out, cyclically
        const uint32_t ahead = open >> r << r;
        r = (uint32_t)__builtin_ctz(ahead ? ahead : open);
        if (P.claim_period) {
            // the jobs left, estimated from where this wave's counter stood after its previous reservation (`end`): the other
            // counters stand about as far into their ranges
            uint32_t r_end = P.job_first + (r + 1u) * P.range_len;
            r_end = r_end < P.total_jobs ? r_end : P.total_jobs;
            uint32_t left = (end == 0u || end > r_end || end + P.range_len < r_end ? P.range_len : r_end - end) * P.claim_ranges;
            left = left < P.total_jobs ? left : P.total_jobs;
            chunk = paced_size(P, ws, chunk, left, lane == leader);
        }
        chunk = want > chunk ? want : chunk;
        // one device atomic on that counter; if it has just run out, the wave says so and its lanes ask again (PT_JOB_RETRY)
        uint32_t local = 0u;
        if (lane == leader) local = atomicAdd(P.job_counter + r * 64u, chunk);
        local = (uint32_t)__builtin_amdgcn_readlane((int)local, (int)leader);
        const uint32_t start = P.job_first + r * P.range_len;
        uint32_t r_end = start + P.range_len;
        r_end = r_end < P.total_jobs ? r_end : P.total_jobs;
        if (local < P.range_len && start + local < r_end) {
            base = start + local;
            got = r_end - base < chunk ? r_end - base : chunk;
        } else if (lane == leader) {
            atomicOr(out_mask, 1u << r);
        }
        if (lane == leader) {
            __hip_atomic_store(&ws->next, base + (want < got ? want : got), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws->end, base + got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ws->range, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (avail != 0u && lane == leader) {
        __hip_atomic_store(&ws->next, end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the last of its reserve goes to the first lanes below
    }
    if (rank < avail) return next + rank;
    const uint32_t k = rank - avail;
    if (k < got) return base + k;
    return open != 0u ? PT_JOB_RETRY : P.total_jobs;
}
// Claims jobs until one lies inside the tile; loads that pixel's accumulator into the lane's LDS slots.
template <int MODE>
PT_DEV bool claim_pixel(const TraceParams& P, uint32_t& job, bool have_candidate, float* px_f, PixelJob& J, bool split = false) {
    uint32_t* px_u = (uint32_t*)px_f;
    for (;;) {
        if (!have_candidate) job = claim_jobs<MODE>(P);
        have_candidate = false;
        if (MODE == CLAIM_RANGED && job == PT_JOB_RETRY) continue;
        if (job >= P.total_jobs) return false;
        if (job_pixel(P, job, J, split)) break;
    }
    if (split) {  // the pixel's accumulator stays in memory until the fold; its last sample owns the primary id, staged with it
        if (J.k + 1u == P.split_n) ((uint32_t*)&P.staging[(size_t)J.k * P.frame_pixels + J.pix])[3] = 0xFFFFFFFFu;
        return true;
    }
    float4 a = P.accum[J.pix];
    px_f[0 * 64] = a.x; px_f[1 * 64] = a.y; px_f[2 * 64] = a.z;
    if (P.m2) { px_f[3 * 64] = P.m2[J.pix * 3]; px_f[4 * 64] = P.m2[J.pix * 3 + 1]; px_f[5 * 64] = P.m2[J.pix * 3 + 2]; }
    px_u[6 * 64] = 0u;
    P.primary_ids[J.pix] = 0xFFFFFFFFu;
    return true;
}

// Tallies the samples of a finished pixel for all the lanes of the wave that finish a pixel together (in the wave's own
// record: the launch-wide tally takes one atomic per wave, at its end).
PT_DEV void add_paths(const TraceParams& P, uint32_t samples) {
    const unsigned long long m = __ballot(true);
    if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u)
        __hip_atomic_fetch_add(&wave_state(P)->paths, samples * (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
PT_DEV void flush_paths(const TraceParams& P) {
    if ((threadIdx.x & 63u) == 0u && P.stats)
        atomicAdd(&P.stats[0], (unsigned long long)__hip_atomic_load(&wave_state(P)->paths, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// The figure of a path's first hit: the "primary id" of the pixel's latest sample.  Sample-granular jobs stage it with the
// sample (launches overlap there: the fold, which runs in launch order, writes the id), whole-pixel jobs write it directly.
PT_DEV void write_primary(const TraceParams& P, uint32_t job, bool split, uint32_t fig) {
    PixelJob J;
    job_pixel(P, job, J, split);
    if (!split) P.primary_ids[J.pix] = fig;
    else if (J.k + 1u == P.split_n) ((uint32_t*)&P.staging[(size_t)J.k * P.frame_pixels + J.pix])[3] = fig;
}

// Ends paths and starts new ones: while `ended`, folds the finished sample's radiance into the pixel's running
// mean (path_trace.wgsl:181, :44-49), writes a finished pixel back and claims the next, and regenerates the
// camera ray of the next sample (random.wgsl:12-14, camera.wgsl:6-19); then starts the closest-hit traversal of
// `nray`.  `fresh`: the lane has just claimed `job` and has no sample to fold yet.  Returns the lane's next state.
template <int MODE>
PT_DEV uint32_t next_path(const DScene& sc, const TraceParams& P, Trav& tv, float* park, float* px_f, uint32_t& job, bool fresh, bool ended,
                          Ray nray, V3 thr, V3 rad, uint32_t rng, uint32_t bounce, bool split = false) {
    const ptx_camera& cam = P.cam;
    uint32_t* park_u = (uint32_t*)park;
    uint32_t* px_u = (uint32_t*)px_f;
    while (ended || fresh) {
        PixelJob J;
        uint32_t k_next = 0u;
        if (!fresh && split) {
            PT_SITE(36);
            job_pixel(P, job, J, true);
            V3 r = v3(dm::fmax_(rad.x, 0.0f), dm::fmax_(rad.y, 0.0f), dm::fmax_(rad.z, 0.0f));
            float* slot = (float*)&P.staging[(size_t)J.k * P.frame_pixels + J.pix];  // w: the primary hit, written at the first bounce
            slot[0] = r.x; slot[1] = r.y; slot[2] = r.z;
            px_u[6 * 64] += 1u;  // here: paths this lane has finished
            if (!claim_pixel<MODE>(P, job, false, px_f, J, true)) return LS_IDLE;
            k_next = J.k;
        } else if (!fresh) {
            PT_SITE(36);
            job_pixel(P, job, J);
            const uint32_t k = px_u[6 * 64];
            V3 mean = v3(px_f[0 * 64], px_f[1 * 64], px_f[2 * 64]);
            V3 r = v3(dm::fmax_(rad.x, 0.0f), dm::fmax_(rad.y, 0.0f), dm::fmax_(rad.z, 0.0f));
            float nn = dm::fmax_((float)(P.first_sample + k), 1.0f);
            V3 delta = r - mean;
            V3 nm = mean + vdivs(delta, nn);
            if (P.m2) {
                V3 d2 = r - nm;
                px_f[3 * 64] = dm::fma_(delta.x, d2.x, px_f[3 * 64]);
                px_f[4 * 64] = dm::fma_(delta.y, d2.y, px_f[4 * 64]);
                px_f[5 * 64] = dm::fma_(delta.z, d2.z, px_f[5 * 64]);
            }
            px_f[0 * 64] = nm.x; px_f[1 * 64] = nm.y; px_f[2 * 64] = nm.z;
            if (P.dbg_radiance && k < P.dbg_capacity) {
                size_t o = ((size_t)k * P.tile_h * P.tile_w + (size_t)J.ty * P.tile_w + J.tx) * 3;
                P.dbg_radiance[o] = r.x;
                P.dbg_radiance[o + 1] = r.y;
                P.dbg_radiance[o + 2] = r.z;
            }
            px_u[6 * 64] = k + 1u;
            k_next = k + 1u;
            if (k_next >= P.n) {  // pixel complete: write it back, claim another
                P.accum[J.pix] = make_float4(nm.x, nm.y, nm.z, 1.0f);
                if (P.m2) { P.m2[J.pix * 3] = px_f[3 * 64]; P.m2[J.pix * 3 + 1] = px_f[4 * 64]; P.m2[J.pix * 3 + 2] = px_f[5 * 64]; }
                P.out_rgba8[J.pix] = tonemap_rgba8(cam, nm);
                add_paths(P, k_next);
                if (!claim_pixel<MODE>(P, job, false, px_f, J)) return LS_IDLE;
                k_next = 0u;
            }
        } else {
            job_pixel(P, job, J, split);
            k_next = J.k;
        }
        fresh = false;
        rng = pcg_hash(J.px + J.py * 65536u + (P.first_frame + k_next) * 16777259u);
        nray = generate_ray(cam, rng, (float)J.px, (float)J.py);
        thr = v3(1.0f, 1.0f, 1.0f);
        rad = v3(0.0f, 0.0f, 0.0f);
        bounce = 0u;
        ended = cam.max_bounces == 0u;
    }
    park[(PK_THR + 0) * 64] = thr.x; park[(PK_THR + 1) * 64] = thr.y; park[(PK_THR + 2) * 64] = thr.z;
    park[(PK_RAD + 0) * 64] = rad.x; park[(PK_RAD + 1) * 64] = rad.y; park[(PK_RAD + 2) * 64] = rad.z;
    park_u[PK_RNG * 64] = rng;
    park_u[PK_MISC * 64] = bounce;  // csg attempt 0, lit 0
    tv.ray = nray;
    tv.inv_dir = vrcp(nray.d);
    trav_begin(sc, tv, false, PT_MAX_T);
    return LS_WALK;
}

// One POST step: everything between the end of a traversal and the start of the lane's next one
// (path_trace.wgsl:63-179 around the two trace calls; bvh.wgsl:143-232 after the BVH loop).
template <bool FRAC>
PT_DEV uint32_t post_step(const DScene& sc, const TraceParams& P, Trav& tv, float* park, float* frac, float* px_f, uint32_t& job, uint32_t& n_shadow, bool split) {
    PT_SITE(43);
    const ptx_camera& cam = P.cam;
    uint32_t* park_u = (uint32_t*)park;
    uint32_t* px_u = (uint32_t*)px_f;
    V3 thr = v3(park[(PK_THR + 0) * 64], park[(PK_THR + 1) * 64], park[(PK_THR + 2) * 64]);
    V3 rad = v3(park[(PK_RAD + 0) * 64], park[(PK_RAD + 1) * 64], park[(PK_RAD + 2) * 64]);
    uint32_t rng = park_u[PK_RNG * 64];
    const uint32_t misc = park_u[PK_MISC * 64];
    uint32_t bounce = misc & 0xFFFFFFu, attempt = (misc >> 24) & 0xFu;  // bit 28 is PK_MISC_NORMAL_READY
    const bool lit = (misc >> 31) != 0u;
    bool need_b = false, ended = false, new_ray = false;
    Ray nray;
    nray.o = v3(0.0f, 0.0f, 0.0f);
    nray.d = v3(0.0f, 0.0f, 1.0f);
    if (!tv.shadow) {
        // A lane whose closest hit is a fractal leaves this step once, to evaluate the normal (four estimator
        // evaluations) as schedulable iterations, and comes back with the normal parked and this flag set.
        const bool normal_ready = FRAC && (misc & PK_MISC_NORMAL_READY) != 0u;
        // bvh.wgsl:192-209: planes, tested linearly after the BVH.
        for (uint32_t i = 0; i < sc.ninf && !normal_ready; i++) {
            PT_SITE(30);
            const DPrim& g = sc.inf[i];
            float ht;
            uint32_t sub;
            if (intersect_listed<false>(tv.ray, tv.inv_dir, g, 0u, ht, sub) && ht > PT_EPS && ht < tv.bound) {
                tv.bound = ht;
                tv.slot = PT_SLOT_INF | i;
                tv.sub = sub;
            }
        }
        // bvh.wgsl:143-188: a hit inside a subtraction shape restarts the traversal behind that shape, up to 8 times.
        if (sc.nneg > 0u && tv.slot != PT_SLOT_NONE && attempt < 8u && !normal_ready) {
            V3 hp = madd(tv.ray.o, tv.ray.d, tv.bound);
            if (inside_any_negative(sc, hp)) {
                float min_exit = PT_MAX_T;
                for (uint32_t i = 0; i < sc.nneg; i++)
                    if (is_inside_neg(sc.negs[i], hp)) min_exit = dm::fmin_(min_exit, find_exit_t(sc.negs[i], tv.ray));
                if (!(min_exit >= PT_MAX_T)) {
                    float advance = min_exit + PT_EPS * 2.0f;
                    tv.ray.o = madd(tv.ray.o, tv.ray.d, advance);
                    park_u[PK_MISC * 64] = bounce | ((attempt + 1u) << 24);
                    trav_begin(sc, tv, false, PT_MAX_T);
                    return LS_WALK;
                }
            }
        }
        if (FRAC && !normal_ready && tv.slot != PT_SLOT_NONE) {
            const DPrim& gw = slot_prim(sc, tv.slot);
            if (gw.type == FIG_MANDELBULB || gw.type == FIG_JULIA) {
                park_u[PK_MISC * 64] = bounce | (attempt << 24);  // the planes may have moved bound / slot: those live in tv
                task_start_normal(gw, tv.slot, tv.ray, tv.bound, frac, P.frac_fast != 0u && (P.frac_fast >= 2u || bounce != 0u) ? 1u : 0u);
                return gw.type == FIG_MANDELBULB ? LS_MANDEL : LS_JULIA;
            }
        }
        PT_SITE(32);
        px_u[7 * 64] += 1u;
        const bool first = bounce == 0u;
        uint32_t fig = 0xFFFFFFFFu;
        if (tv.slot == PT_SLOT_NONE) {
            rad = rad + thr * sample_skybox(sc, cam, tv.ray.d);
            ended = true;
        } else {
            const DPrim& g = slot_prim(sc, tv.slot);
            fig = g.fig;
            const DShade& sh = sc.shade[fig];
            const ptx_material mt = sh.mat;
            bool need_uv = mt.texture_id >= 0;
            HitFull h;
            if (normal_ready) {  // finish_fractal with the normal evaluated by frac_step
                h.pos = madd(tv.ray.o, tv.ray.d, tv.bound);
                h.normal = v3(park[(PK_N + 0) * 64], park[(PK_N + 1) * 64], park[(PK_N + 2) * 64]);
                h.uv.x = 0.0f;
                h.uv.y = 0.0f;
            } else {
                resolve_hit<FRAC>(tv.ray, tv.inv_dir, g, tv.bound, tv.sub, need_uv, &sh, &h);
            }
            V3 base_color = v3p(mt.base_color);
            if (need_uv) {
                V2 tuv;
                tuv.x = h.uv.x * sh.texture_scale;
                tuv.y = h.uv.y * sh.texture_scale;
                base_color = base_color * sample_texture(sc, mt.texture_id, tuv);
            }
            if (mt.emission_strength > 0.0f) {  // path_trace.wgsl:83-87
                V3 le = v3p(mt.emission) * mt.emission_strength;
                rad = rad + thr * le;
                ended = true;
            } else {
                V3 wo = -tv.ray.d;
                V3 n = h.normal;
                if (dot(n, wo) < 0.0f) n = -n;
                if (mt.transmission > 0.5f) {  // path_trace.wgsl:97-105
                    PT_SITE(37);
                    V3 dir;
                    if (!sample_glass(rng, wo, n, mt.ior, dir)) {
                        ended = true;
                    } else {
                        thr = thr * base_color;
                        nray.o = madd(h.pos, dir, PT_EPS * 2.0f);
                        nray.d = dir;
                        bounce += 1u;
                        new_ray = true;
                    }
                } else {
                    V3 hpos = h.pos;
                    park[(PK_HPOS + 0) * 64] = hpos.x; park[(PK_HPOS + 1) * 64] = hpos.y; park[(PK_HPOS + 2) * 64] = hpos.z;
                    park[(PK_N + 0) * 64] = n.x; park[(PK_N + 1) * 64] = n.y; park[(PK_N + 2) * 64] = n.z;
                    park[(PK_BASE + 0) * 64] = base_color.x; park[(PK_BASE + 1) * 64] = base_color.y; park[(PK_BASE + 2) * 64] = base_color.z;
                    park_u[PK_FIG * 64] = fig;
                    park[(PK_WO + 0) * 64] = wo.x; park[(PK_WO + 1) * 64] = wo.y; park[(PK_WO + 2) * 64] = wo.z;
                    need_b = true;
                    if (mt.roughness > 0.04f) {  // path_trace.wgsl:108-154
                        PT_SITE(33);
                        MatD mat;
                        mat.base_color = base_color;
                        mat.metallic = mt.metallic;
                        mat.roughness = mt.roughness;
                        mat.ior = mt.ior;
                        const uint32_t num_lights = sc.nlight > 0u ? sc.nlight : 1u;  // empty list = one-element placeholder
                        float pick = rand_f32(rng);
                        uint32_t light_pick = (uint32_t)(pick * (float)num_lights);
                        uint32_t li = light_pick < num_lights - 1u ? light_pick : num_lights - 1u;
                        float r1 = rand_f32(rng), r2 = rand_f32(rng);
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
                                PT_SITE(34);
                                V3 light_normal = normalize(light_point - lpos);
                                float cos_light = dm::abs_(dot(-light_dir, light_normal));
                                float light_solid_pdf = cos_light <= 0.0f ? 0.0f : dm::div_(L.area_pdf * (light_dist * light_dist), cos_light);
                                V3 brdf = eval_brdf(wo, light_dir, n, mat);
                                float brdf_pdf_val = n_dot_l * PT_INV_PI;
                                float pa = light_solid_pdf * (float)num_lights;
                                float a2 = pa * pa, b2 = brdf_pdf_val * brdf_pdf_val;  // mis.wgsl:3-7
                                float w = dm::div_(a2, dm::fmax_(a2 + b2, 1e-10f));
                                V3 contrib = vdivs(thr * v3p(L.le) * brdf * n_dot_l * w, pa);
                                park[(PK_CONTRIB + 0) * 64] = contrib.x; park[(PK_CONTRIB + 1) * 64] = contrib.y; park[(PK_CONTRIB + 2) * 64] = contrib.z;
                                n_shadow += 1u;
                                if (first) write_primary(P, job, split, fig);
                                park[(PK_THR + 0) * 64] = thr.x; park[(PK_THR + 1) * 64] = thr.y; park[(PK_THR + 2) * 64] = thr.z;
                                park[(PK_RAD + 0) * 64] = rad.x; park[(PK_RAD + 1) * 64] = rad.y; park[(PK_RAD + 2) * 64] = rad.z;
                                park_u[PK_RNG * 64] = rng;
                                park_u[PK_MISC * 64] = bounce | (light_solid_pdf > 0.0f ? 0x80000000u : 0u);
                                tv.ray.o = madd(hpos, n, PT_EPS * 2.0f);
                                tv.ray.d = light_dir;
                                tv.inv_dir = vrcp(light_dir);
                                trav_begin(sc, tv, true, light_dist);
                                return LS_WALK;
                            }
                        }
                    }
                }
            }
        }
        if (first) write_primary(P, job, split, fig);
    } else {
        // bvh.wgsl:212-232: the plane list of the occlusion query, unless the BVH already gave an occluder.
        bool occluded = tv.slot != PT_SLOT_NONE;
        for (uint32_t i = 0; i < sc.ninf && !occluded; i++) {
            PT_SITE(30);
            const DPrim& g = sc.inf[i];
            float ht;
            uint32_t sub;
            if (intersect_listed<false>(tv.ray, tv.inv_dir, g, 0u, ht, sub) && ht > PT_EPS && ht < tv.bound - PT_EPS)
                if (sc.nneg == 0u || !inside_any_negative(sc, madd(tv.ray.o, tv.ray.d, ht))) occluded = true;
        }
        if (!occluded && lit)  // path_trace.wgsl:148: only when light_solid_pdf > 0
            rad = rad + v3(park[(PK_CONTRIB + 0) * 64], park[(PK_CONTRIB + 1) * 64], park[(PK_CONTRIB + 2) * 64]);
        need_b = true;
    }
    if (need_b) {  // path_trace.wgsl:156-178
        PT_SITE(35);
        V3 hpos = v3(park[(PK_HPOS + 0) * 64], park[(PK_HPOS + 1) * 64], park[(PK_HPOS + 2) * 64]);
        V3 n = v3(park[(PK_N + 0) * 64], park[(PK_N + 1) * 64], park[(PK_N + 2) * 64]);
        V3 wo = v3(park[(PK_WO + 0) * 64], park[(PK_WO + 1) * 64], park[(PK_WO + 2) * 64]);
        const ptx_material& mt = sc.shade[park_u[PK_FIG * 64]].mat;
        MatD mat;
        mat.base_color = v3(park[(PK_BASE + 0) * 64], park[(PK_BASE + 1) * 64], park[(PK_BASE + 2) * 64]);
        mat.metallic = mt.metallic;
        mat.roughness = mt.roughness;
        mat.ior = mt.ior;
        V3 wi, weight;
        if (!sample_brdf(rng, wo, n, mat, wi, weight)) {
            ended = true;
        } else {
            thr = thr * weight;
            nray.o = madd(hpos, wi, PT_EPS * 2.0f);
            nray.d = wi;
            if (bounce >= 3u) {  // path_trace.wgsl:166-172
                float survival = dm::fmin_(dm::fmax_(thr.x, dm::fmax_(thr.y, thr.z)), 0.95f);
                if (rand_f32(rng) > survival) ended = true;
                else thr = vdivs(thr, survival);
            }
            if (!ended) {
                float lum = dot(thr, v3(0.2126f, 0.7152f, 0.0722f));
                if (lum > cam.firefly_clamp) thr = thr * dm::div_(cam.firefly_clamp, lum);
                bounce += 1u;
                new_ray = true;
            }
        }
    }
    if (new_ray && bounce >= cam.max_bounces) ended = true;  // path_trace.wgsl:63
    return next_path<FRAC ? CLAIM_PLAIN : CLAIM_GUIDED>(sc, P, tv, park, px_f, job, false, ended, nray, thr, rad, rng, bounce, split);
}
