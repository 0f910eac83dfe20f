// ptx355: kernels and the C ABI (include/ptx355.h) of the MI355X path-tracing compute pass.
//
// Execution model.  One lane owns one pixel for the whole launch and walks a flattened state machine:
// each loop iteration is ONE bounce of the lane's current sample; when a path ends the lane folds the
// sample into its running mean (kept in registers) and immediately regenerates the camera ray of its
// next sample.  A wave therefore stays full until its lanes run out of samples, instead of idling
// behind the longest path of every sample as a one-sample-per-dispatch kernel does; with n samples per
// launch the accumulation buffer is read and written once instead of n times.  A wave covers an 8x8
// pixel block (primary rays stay coherent), a 256-thread workgroup a 16x16 block, and workgroups are
// dealt so that each XCD's L2 serves one contiguous band of the frame.  Traversal stacks live in LDS
// (lane-interleaved 8-byte entries, conflict-free); scene data is re-laid-out at upload into 64-byte
// records (dscene.h).  No MFMA: nothing here is a dense contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <chrono>
#include <vector>
#include "sched_dev.h"
#include "post_dev.h"
#include "bvh_build.h"
#include "relayout.h"
#include "collective.h"

#define PT_DEEP_TREE 20u  // stack entries from which a tree counts as deep (millions of primitives): one node visit per WALK step
#define PT_PIXEL_FIELDS 9u  // per-lane LDS slots: mean xyz, M2 xyz, sample index, path-bounces, primary id
#ifndef PTX_CLAIM_LDS
#define PTX_CLAIM_LDS true  // the lockstep LDS kernels claim pixels dynamically
#endif
#ifndef FRAC_HELD_MIN
#define FRAC_HELD_MIN 56  // a wave goes on serving an estimator while at least this many of its lanes hold a march (measured: 16 .. 56 rising, flat from 48)
#endif
#ifndef FRAC_EXIT_NUM
#define FRAC_EXIT_NUM 3u  // a marcher is served while at least NUM / DEN of the lanes it started with still march
#define FRAC_EXIT_DEN 4u
#endif
#ifndef PTX_MIN_WAVES
#define PTX_MIN_WAVES 3  // waves per SIMD the register allocator must leave room for
#endif

// Scratch of ptx_trace_n_static, the entry point that allocates nothing: the launch-wide pixel-claim counter, the
// per-wave claim records and the deep traversal-stack entries of up to PTX_STATIC_MAX_WAVES waves (a 256-CU device
// runs at most 4 workgroups of 4 waves per compute unit with these kernels' LDS use).  One copy per device, held
// from module load to process exit; every call on a device shares it.
#define PTX_FLIGHTS 8  // launches of sample-granular jobs that may be in flight at once (their staging buffers: 8 GiB together at most)
#define PTX_FLIGHTS_DEFAULT 6u
#define PTX_CLAIM_RANGES 16  // job counters of a launch at most (their run-out bits share one 32-bit word)
#define PTX_STATIC_MAX_WAVES 4096u
#define PTX_STATIC_WAVES_OFFSET 256u
#define PTX_STATIC_SPILL_OFFSET (PTX_STATIC_WAVES_OFFSET + PTX_STATIC_MAX_WAVES * 32u)  // sizeof(WaveState)
#define PTX_STATIC_BYTES (PTX_STATIC_SPILL_OFFSET + (size_t)PTX_STATIC_MAX_WAVES * (31u - PT_LDS_STACK) * 64u * 8u)
static __device__ __align__(256) unsigned char ptx_static_scratch[PTX_STATIC_BYTES];


// Bijective block remap: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), so give the
// blocks of one XCD a contiguous range of logical tile indices.
PT_DEV uint32_t xcd_remap(uint32_t b, uint32_t n) {
    uint32_t q = n / 8u, r = n % 8u, x = b % 8u;
    uint32_t base = x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q;
    return base + b / 8u;
}

PT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Cooperative copy of `bytes` (a multiple of 16) from global memory into LDS.
PT_DEV void copy_to_lds(void* dst, const void* src, uint32_t bytes) {
    uint4* d = (uint4*)dst;
    const uint4* s = (const uint4*)src;
    for (uint32_t i = threadIdx.x; i < bytes / 16u; i += PTX_BLOCK) d[i] = s[i];
}

// LDS_SCENE: the whole scene (primitives, planes, node pairs, shading records, lights, subtraction
// shapes) is small enough to be staged in LDS once per workgroup; every later scene read is then an
// LDS read (~100 cycles) instead of a dependent L1/L2 round trip, which is what the traversal waits on.
// CLAIM: the grid is sized to the GPU and a lane whose pixel has all its samples claims the next unclaimed
// pixel (job_pixel / claim_pixel in sched_dev.h) instead of idling until the slowest pixel of its wave is done.
template <bool LDS_SCENE, bool FRAC, bool CLAIM, int CM = FRAC ? CLAIM_PLAIN : CLAIM_PACED>
PT_DEV void trace_body(const TraceParams& P) {
    extern __shared__ StackEntry lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    WalkStack st;
    st.lds = lds_stack + (size_t)wave * P.sc.lds_stack_entries * 64u + lane;
    st.glb = walk_stack_glb(P.stack_spill, blockIdx.x * PTX_WAVES_PER_BLOCK + wave, lane);
    st.nl = P.sc.lds_stack_entries;
    DScene sc = P.sc;
    if (LDS_SCENE) {
        char* base = (char*)(lds_stack + (size_t)PTX_WAVES_PER_BLOCK * P.sc.lds_stack_entries * 64u) + (size_t)PTX_WAVES_PER_BLOCK * (PT_PIXEL_FIELDS + PT_PARK_FIELDS) * 64u * 4u;
        uint32_t off = 0;
        auto stage = [&](const void* src, uint32_t bytes) -> void* {
            void* dst = base + off;
            copy_to_lds(dst, src, bytes);
            off += bytes;
            return dst;
        };
        sc.prims = (const DPrim*)stage(P.sc.prims, P.sc.nprim * (uint32_t)sizeof(DPrim));
        sc.inf = (const DPrim*)stage(P.sc.inf, P.sc.ninf * (uint32_t)sizeof(DPrim));
        sc.pairs = (const DNodePair*)stage(P.sc.pairs, P.sc.npair * (uint32_t)sizeof(DNodePair));
        sc.shade = (const DShade*)stage(P.sc.shade, P.sc.nfig * (uint32_t)sizeof(DShade));
        sc.lights = (const DLight*)stage(P.sc.lights, P.sc.nlight * (uint32_t)sizeof(DLight));
        sc.negs = (const DNeg*)stage(P.sc.negs, P.sc.nneg * (uint32_t)sizeof(DNeg));
        __syncthreads();
    }

    // Per-pixel state that is only touched when a path ends (running mean, M2, sample index, tallies)
    // lives in LDS, one lane-interleaved slot per field, so it does not occupy registers during traversal.
    float* px_f = (float*)(lds_stack + (size_t)PTX_WAVES_PER_BLOCK * P.sc.lds_stack_entries * 64u) + (size_t)wave * (PT_PIXEL_FIELDS + PT_PARK_FIELDS) * 64u + lane;
    uint32_t* px_u = (uint32_t*)px_f;
    // The lane's pixel: with CLAIM a job number (coordinates are recomputed from it when a path ends), else fixed.
    uint32_t job = CLAIM ? blockIdx.x * PTX_BLOCK + threadIdx.x
                         : (xcd_remap(blockIdx.x, P.n_blocks) << 8) | threadIdx.x;
    PixelJob J;
    bool active = false;
    const bool split = CLAIM && P.split_n > 0u;  // jobs are single samples, staged for ptx_fold_kernel (TraceParams::split_n)
    wave_claim_init(P, CM == CLAIM_RANGED || CM == CLAIM_PACED);
    px_f[3 * 64] = 0.0f; px_f[4 * 64] = 0.0f; px_f[5 * 64] = 0.0f;
    px_u[6 * 64] = 0u;            // k: samples of the current pixel finished
    px_u[7 * 64] = 0u;            // path-bounces
    px_u[8 * 64] = 0xFFFFFFFFu;   // primary hit of the latest sample
    if (P.n > 0u) {
        if (CLAIM) {
            active = claim_pixel<CM>(P, job, true, px_f, J, split);
        } else if (job_pixel(P, job, J)) {
            active = true;
            float4 a = P.accum[J.pix];
            px_f[0 * 64] = a.x; px_f[1 * 64] = a.y; px_f[2 * 64] = a.z;
            if (P.m2) { px_f[3 * 64] = P.m2[J.pix * 3]; px_f[4 * 64] = P.m2[J.pix * 3 + 1]; px_f[5 * 64] = P.m2[J.pix * 3 + 2]; }
        }
    }
    uint32_t n_shadow = 0;
    PathState ps;
    ps.rng = 0u;
    ps.bounce = 0u;
    ps.throughput = v3(1.0f, 1.0f, 1.0f);
    ps.radiance = v3(0.0f, 0.0f, 0.0f);
    ps.ray.o = v3(0.0f, 0.0f, 0.0f);
    ps.ray.d = v3(0.0f, 0.0f, 1.0f);
    if (active) {
        ps.rng = pcg_hash(J.px + J.py * 65536u + (P.first_frame + (split ? J.k : 0u)) * 16777259u);  // random.wgsl:12-14
        ps.ray = generate_ray(P.cam, ps.rng, (float)J.px, (float)J.py);
    }
#ifdef PTX_PHASE_TIMING
    for (int i = 0; i < PT_NTICKS; i++) ps.tick_acc[i] = 0;
    ps.tick_last = clock64();
    const unsigned long long wall0 = wall_clock64();
#endif
    uint32_t local_bounces = 0;
    while (active) {
        bool done = true;
        if (ps.bounce < P.cam.max_bounces) {
            uint32_t fig;
            local_bounces += 1u;
            const bool first = ps.bounce == 0u;
            done = bounce_step<FRAC, !LDS_SCENE>(sc, P.cam, ps, st, px_f + PT_PIXEL_FIELDS * 64u, fig, n_shadow);
            PT_TICK(3);
            if (first) px_u[8 * 64] = fig;
            if (!done && ps.bounce >= P.cam.max_bounces) done = true;
        }
        PT_TICK(4);
        if (done && split) {
            PT_SITE(36);
            // the sample goes to the staging buffer, with the figure of its first hit; the fold kernel does path_trace.wgsl:181
            float* slot = (float*)&P.staging[(size_t)J.k * P.frame_pixels + J.pix];
            slot[0] = dm::fmax_(ps.radiance.x, 0.0f); slot[1] = dm::fmax_(ps.radiance.y, 0.0f); slot[2] = dm::fmax_(ps.radiance.z, 0.0f);
            ((uint32_t*)slot)[3] = px_u[8 * 64];
            px_u[7 * 64] += local_bounces;
            local_bounces = 0u;
            add_paths(P, 1u);
            active = claim_pixel<CM>(P, job, false, px_f, J, true);
            px_u[8 * 64] = 0xFFFFFFFFu;
            if (active) {
                ps.rng = pcg_hash(J.px + J.py * 65536u + (P.first_frame + J.k) * 16777259u);
                ps.ray = generate_ray(P.cam, ps.rng, (float)J.px, (float)J.py);
                ps.throughput = v3(1.0f, 1.0f, 1.0f);
                ps.radiance = v3(0.0f, 0.0f, 0.0f);
                ps.bounce = 0u;
            }
        } else if (done) {
            PT_SITE(36);
            // path_trace.wgsl:181 and :44-49
            const uint32_t k = px_u[6 * 64];
            V3 mean = v3(px_f[0 * 64], px_f[1 * 64], px_f[2 * 64]);
            V3 rad = v3(dm::fmax_(ps.radiance.x, 0.0f), dm::fmax_(ps.radiance.y, 0.0f), dm::fmax_(ps.radiance.z, 0.0f));
            float nn = dm::fmax_((float)(P.first_sample + k), 1.0f);
            V3 delta = rad - mean;
            V3 nm = mean + vdivs(delta, nn);
            if (P.m2) {
                V3 d2 = rad - nm;
                px_f[3 * 64] = dm::fma_(delta.x, d2.x, px_f[3 * 64]);
                px_f[4 * 64] = dm::fma_(delta.y, d2.y, px_f[4 * 64]);
                px_f[5 * 64] = dm::fma_(delta.z, d2.z, px_f[5 * 64]);
            }
            px_f[0 * 64] = nm.x; px_f[1 * 64] = nm.y; px_f[2 * 64] = nm.z;
            if (P.dbg_radiance && k < P.dbg_capacity) {
                size_t r = ((size_t)k * P.tile_h * P.tile_w + (size_t)J.ty * P.tile_w + J.tx) * 3;
                P.dbg_radiance[r] = rad.x;
                P.dbg_radiance[r + 1] = rad.y;
                P.dbg_radiance[r + 2] = rad.z;
            }
            px_u[6 * 64] = k + 1u;
            px_u[7 * 64] += local_bounces;
            local_bounces = 0u;
            uint32_t k_next = k + 1u;
            if (k_next >= P.n) {  // pixel complete: write it back; with CLAIM go on with another pixel
                P.accum[J.pix] = make_float4(nm.x, nm.y, nm.z, 1.0f);
                if (P.m2) { P.m2[J.pix * 3] = px_f[3 * 64]; P.m2[J.pix * 3 + 1] = px_f[4 * 64]; P.m2[J.pix * 3 + 2] = px_f[5 * 64]; }
                P.out_rgba8[J.pix] = tonemap_rgba8(P.cam, nm);
                P.primary_ids[J.pix] = px_u[8 * 64];
                add_paths(P, k_next);
                active = CLAIM && claim_pixel<CM>(P, job, false, px_f, J);
                px_u[8 * 64] = 0xFFFFFFFFu;
                k_next = 0u;
            }
            if (active) {
                ps.rng = pcg_hash(J.px + J.py * 65536u + (P.first_frame + k_next) * 16777259u);
                ps.ray = generate_ray(P.cam, ps.rng, (float)J.px, (float)J.py);
                ps.throughput = v3(1.0f, 1.0f, 1.0f);
                ps.radiance = v3(0.0f, 0.0f, 0.0f);
                ps.bounce = 0u;
            }
        }
        PT_TICK(5);
    }
#ifdef PTX_PHASE_TIMING
    if (lane == 0u && P.stats) {
        for (int i = 0; i < 6; i++) atomicAdd(&P.stats[4 + i], (unsigned long long)ps.tick_acc[i]);
        for (int i = 6; i < PT_NTICKS; i++) atomicAdd(&P.stats[10 + i], (unsigned long long)ps.tick_acc[i]);
        const unsigned long long wall1 = wall_clock64();  // 100 MHz
        atomicMax(&P.stats[10], ~wall0);
        atomicMax(&P.stats[11], wall1);
        atomicAdd(&P.stats[12], wall1 - wall0);
        atomicAdd(&P.stats[13], 1ull);
    }
#endif
    const uint32_t n_bounce = px_u[7 * 64];
    uint32_t sb = wave_sum(n_bounce), ss = wave_sum(n_shadow);
    flush_paths(P);
    if (lane == 0u && P.stats) {
        atomicAdd(&P.stats[1], (unsigned long long)sb);
        atomicAdd(&P.stats[2], (unsigned long long)ss);
    }
}

// Lockstep kernel over SORTED PATH POOLS (sample-granular jobs, scenes staged in LDS, long launches).
//
// In trace_body a wave's lanes hold unrelated rays after the first bounce: on the Cornell box the closest-hit query of a wave
// visits every leaf for a few lanes each.  The same query over the same rays runs 1.95 times as fast when each wave's rays
// share the leaf boxes they enter and the order they enter them in (tools/coherence_probe.py).  Sorting a workgroup's 256
// paths before every bounce gets part of that but stalls on a barrier per bounce (measured: fewer instructions, lower issue
// rate).  Here a workgroup owns a POOL of pool_m paths (4096: sixteen per lane) in global memory, two buffers of 64-byte
// records, and works in rounds:
//   top-up    the pool is refilled to pool_m with new paths -- ONE device atomic per workgroup and round reserves the jobs,
//             every lane generates camera rays (the code that ends and starts paths, which trace_body runs for 14 of 64 lanes
//             after every bounce, runs here for all lanes a few times per round);
//   sort      counting sort of the pool's keys (regroup_key: which key boxes the ray's box test hits, in which order) through
//             an LDS histogram: the order the buffer is traced in;
//   trace     waves take chunks of 64 paths of the sorted order from an LDS counter (a wave that is done early takes the next
//             chunk: the waves of a workgroup finish a round within one chunk of each other), run ONE bounce for each path,
//             stage finished samples, and append the paths that go on, with the key of their new ray, to the other buffer.
// One barrier sequence per ROUND (sixteen wave-bounces per wave), not per bounce.  A path's arithmetic is the lockstep
// kernel's whichever lane runs it and samples are folded in order by ptx_fold_kernel: bit-identical frames.
#define RG_BINS 128u
PT_DEV uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= (uint32_t)off) v += u;
    }
    return v;
}
PT_DEV void lds_settle() { __builtin_amdgcn_s_waitcnt(0xC07F); }  // this wave's LDS operations have been performed
PT_DEV void wg_sync() { lds_settle(); __syncthreads(); }
// kbox: four boxes {min xyz, -, max xyz, -}; valid: bit i set when box i exists (wave-uniform).
PT_DEV uint32_t regroup_key(const float* kbox, uint32_t valid, const Ray& ray) {
    const V3 inv = vrcp(ray.d);
    float t[4];
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++) {
        t[i] = PT_MAX_T;
        if (valid & (1u << i)) t[i] = intersect_aabb(ray.o, inv, kbox + i * 8u, kbox + i * 8u + 4u);
    }
    uint32_t key = 0u;
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++) key |= t[i] < PT_MAX_T ? 8u << i : 0u;
    key |= t[0] < t[1] ? 1u : 0u;
    key |= t[2] < t[3] ? 2u : 0u;
    key |= dm::fmin_(t[0], t[1]) < dm::fmin_(t[2], t[3]) ? 4u : 0u;
    return key;  // < RG_BINS
}
PT_DEV void pool_store(PoolPath* dst, const PathState& ps, uint32_t job, uint32_t primary) {
    float4* q = (float4*)dst;
    q[0] = make_float4(ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, __uint_as_float(ps.rng));
    q[1] = make_float4(ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, __uint_as_float(ps.bounce));
    q[2] = make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, __uint_as_float(job));
    q[3] = make_float4(ps.radiance.x, ps.radiance.y, ps.radiance.z, __uint_as_float(primary));
}
template <bool LDS_SCENE, bool FRAC>
PT_DEV void sorted_body(const TraceParams& P) {
    extern __shared__ StackEntry lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t kSlots = PT_PIXEL_FIELDS + PT_PARK_FIELDS;
    WalkStack st;
    st.lds = lds_stack + (size_t)wave * P.sc.lds_stack_entries * 64u + lane;
    st.glb = LDS_SCENE ? nullptr : walk_stack_glb(P.stack_spill, blockIdx.x * PTX_WAVES_PER_BLOCK + wave, lane);
    st.nl = P.sc.lds_stack_entries;
    float* fields = (float*)(lds_stack + (size_t)PTX_WAVES_PER_BLOCK * P.sc.lds_stack_entries * 64u);
    DScene sc = P.sc;
    if (LDS_SCENE) {
        char* base = (char*)(fields + (size_t)PTX_WAVES_PER_BLOCK * kSlots * 64u);
        uint32_t off = 0;
        auto stage = [&](const void* src, uint32_t bytes) -> void* {
            void* dst = base + off;
            copy_to_lds(dst, src, bytes);
            off += bytes;
            return dst;
        };
        sc.prims = (const DPrim*)stage(P.sc.prims, P.sc.nprim * (uint32_t)sizeof(DPrim));
        sc.inf = (const DPrim*)stage(P.sc.inf, P.sc.ninf * (uint32_t)sizeof(DPrim));
        sc.pairs = (const DNodePair*)stage(P.sc.pairs, P.sc.npair * (uint32_t)sizeof(DNodePair));
        sc.shade = (const DShade*)stage(P.sc.shade, P.sc.nfig * (uint32_t)sizeof(DShade));
        sc.lights = (const DLight*)stage(P.sc.lights, P.sc.nlight * (uint32_t)sizeof(DLight));
        sc.negs = (const DNeg*)stage(P.sc.negs, P.sc.nneg * (uint32_t)sizeof(DNeg));
    }
    // Sample jobs keep no running mean in the lane: the per-pixel slots 0-5 of every wave are free.  Wave 0's hold the
    // histogram, the bins' offsets and the round's counters; wave 1's the key boxes.
    uint32_t* hist = (uint32_t*)fields;
    uint32_t* offs = hist + RG_BINS;
    uint32_t* cv = offs + RG_BINS;  // 0 paths in the current buffer, 1 paths appended to the other, 2 next chunk, 3 first reserved job, 4 no job left, 5 jobs reserved this round
    float* kbox = fields + (size_t)kSlots * 64u;
    float* px_f = fields + (size_t)wave * kSlots * 64u + lane;
    float* park = px_f + PT_PIXEL_FIELDS * 64u;
    const uint32_t M = P.pool_m;
    PoolPath* const buf0 = P.pool + (size_t)blockIdx.x * 2u * M;
    uint8_t* const keys0 = P.pool_keys + (size_t)blockIdx.x * 2u * M;
    uint16_t* const order = P.pool_order + (size_t)blockIdx.x * M;  // pool_m <= 16384
    if (threadIdx.x < 8u) cv[threadIdx.x] = 0u;
    if (threadIdx.x == 0u) {
        // the key boxes: the children of the root's children (a child that is a leaf stands for itself)
        uint32_t valid = 0u;
        auto put = [&](uint32_t i, const float* mn, const float* mx) {
            for (int a = 0; a < 3; a++) { kbox[i * 8u + a] = mn[a]; kbox[i * 8u + 4u + a] = mx[a]; }
            valid |= 1u << i;
        };
        if (P.sc.nprim > 0u && (P.sc.root_ref >> PT_REF_BITS) == 0u) {
            const DNodePair& r = P.sc.pairs[P.sc.root_ref & PT_REF_MASK];
            if (r.lcnt == 0u) { const DNodePair& q = P.sc.pairs[r.lref]; put(0u, q.lmin, q.lmax); put(1u, q.rmin, q.rmax); }
            else put(0u, r.lmin, r.lmax);
            if (r.rcnt == 0u) { const DNodePair& q = P.sc.pairs[r.rref]; put(2u, q.lmin, q.lmax); put(3u, q.rmin, q.rmax); }
            else put(2u, r.rmin, r.rmax);
        }
        ((uint32_t*)kbox)[32] = valid;
    }
    wg_sync();
    const uint32_t valid = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t*)kbox)[32]);
    uint32_t n_shadow = 0u, n_bounce = 0u, n_paths = 0u, parity = 0u;
    PathState ps;
#ifdef PTX_PHASE_TIMING
    for (int i = 0; i < PT_NTICKS; i++) ps.tick_acc[i] = 0;
    ps.tick_last = clock64();
    const unsigned long long wall0 = wall_clock64();
#define WG_SYNC_T(slot) do { PT_TICK(slot); wg_sync(); PT_TICK(10); } while (0)
#else
#define WG_SYNC_T(slot) wg_sync()
#endif
    for (;;) {
        PoolPath* const cur = buf0 + (size_t)parity * M;
        PoolPath* const nxt = buf0 + (size_t)(parity ^ 1u) * M;
        uint8_t* const cur_keys = keys0 + (size_t)parity * M;
        uint8_t* const nxt_keys = keys0 + (size_t)(parity ^ 1u) * M;
        // ---- top-up: new paths for the places of those that ended
        if (threadIdx.x == 0u) {
            const uint32_t have = cv[0];
            uint32_t need = (cv[4] != 0u || P.n == 0u || have >= M) ? 0u : M - have;
            if (need > 0u && P.pool_guided != 0u) {
                // Never more than 1 / (pool_guided x workgroups) of the jobs that were left at this workgroup's previous reservation
                // (at least 256): large while there is plenty, small at the end, so that the launch's last jobs are spread over all
                // workgroups instead of filling the pools of a few (a short launch is a handful of fills per workgroup, and pixels
                // differ in cost).
                const uint32_t seen = cv[6];
                uint32_t cap = (P.total_jobs > seen ? P.total_jobs - seen : 0u) / (gridDim.x * P.pool_guided);
                cap = cap < 256u ? 256u : cap;
                need = need < cap ? need : cap;
            }
            cv[5] = need;
            if (need > 0u) {
                const uint32_t base = atomicAdd(P.job_counter, need);
                cv[3] = base;
                cv[6] = base + need;
                if (base + need >= P.total_jobs) cv[4] = 1u;  // the launch's last jobs are among these
            }
        }
        WG_SYNC_T(6);
        {
            const uint32_t need = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv[5]);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv[3]);
            for (uint32_t j = threadIdx.x; j < need; j += PTX_BLOCK) {
                const uint32_t job = base + j;
                PixelJob J;
                if (job < P.total_jobs && job_pixel(P, job, J, true)) {
                    ps.rng = pcg_hash(J.px + J.py * 65536u + (P.first_frame + J.k) * 16777259u);  // random.wgsl:12-14
                    ps.ray = generate_ray(P.cam, ps.rng, (float)J.px, (float)J.py);
                    ps.throughput = v3(1.0f, 1.0f, 1.0f);
                    ps.radiance = v3(0.0f, 0.0f, 0.0f);
                    ps.bounce = 0u;
                    const uint32_t pos = atomicAdd(&cv[0], 1u);
                    if (pos < M) {
                        pool_store(cur + pos, ps, job, 0xFFFFFFFFu);
                        cur_keys[pos] = (uint8_t)regroup_key(kbox, valid, ps.ray);
                    }
                }
            }
        }
        if (threadIdx.x < RG_BINS) hist[threadIdx.x] = 0u;
        if (threadIdx.x == RG_BINS) { cv[1] = 0u; cv[2] = 0u; }
        WG_SYNC_T(6);
        uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv[0]);
        const uint32_t no_jobs_left = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv[4]);
        if (count > M) count = M;
        if (count == 0u) {  // the same for every wave of the workgroup
            // An empty pool ends the workgroup only when the launch has no jobs left to reserve: a reservation may lie wholly
            // outside the tile (the right and bottom quadrants of edge blocks: 64 * n consecutive jobs), and then the next
            // top-up has to go on reserving.  The barrier keeps thread 0 from writing the next round's flags while the other
            // waves still read this round's.
            if (no_jobs_left != 0u || P.n == 0u) break;
            WG_SYNC_T(6);
            continue;
        }
        // The launch's last paths: no job is left to top the pool up with and what remains is a fraction of a round.  Rounds of a
        // shrinking pool cost their sort and barriers for a few paths each; instead every path left is traced to its end by the
        // lane that takes it, unsorted, and the workgroup is done.
        const bool finish = no_jobs_left != 0u && count <= P.pool_finish;
        // ---- sort: the order the current buffer is traced in
        if (!finish) {
            for (uint32_t i = threadIdx.x; i < count; i += PTX_BLOCK) atomicAdd(&hist[(uint32_t)cur_keys[i] & (RG_BINS - 1u)], 1u);
            WG_SYNC_T(7);
            if (wave == 0u) {
                const uint32_t a = hist[2u * lane], b = hist[2u * lane + 1u];
                const uint32_t incl = wave_incl_scan(a + b);
                offs[2u * lane] = incl - (a + b);
                offs[2u * lane + 1u] = incl - b;
            }
            WG_SYNC_T(7);
            for (uint32_t i = threadIdx.x; i < count; i += PTX_BLOCK) {
                const uint32_t pos = atomicAdd(&offs[(uint32_t)cur_keys[i] & (RG_BINS - 1u)], 1u);
                if (pos < M) order[pos] = (uint16_t)i;
            }
            WG_SYNC_T(7);
        }
        // ---- trace: one bounce of every path of the buffer, 64 at a time in sorted order.  A wave claims its next chunk and
        // reads that chunk's places in the order before it traces the current one: only the records' load waits on memory.
        auto claim_chunk = [&](uint32_t& chunk, uint32_t& idx) {
            uint32_t c2 = 0u;
            if (lane == 0u) c2 = atomicAdd(&cv[2], 1u);
            chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)c2);
            const uint32_t i2 = chunk * 64u + lane;
            idx = (chunk * 64u < count && i2 < count) ? (finish ? i2 : (uint32_t)order[i2]) : 0u;
        };
        uint32_t chunk_n, idx_n;
        claim_chunk(chunk_n, idx_n);
        for (;;) {
            const uint32_t chunk = chunk_n;
            uint32_t idx = idx_n;
            if (chunk * 64u >= count) break;
            const uint32_t i = chunk * 64u + lane;
            const bool have = i < count;
            if (idx >= M) idx = 0u;
            const float4* q = (const float4*)(cur + idx);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            ps.ray.o = v3(q0.x, q0.y, q0.z); ps.rng = __float_as_uint(q0.w);
            ps.ray.d = v3(q1.x, q1.y, q1.z); ps.bounce = __float_as_uint(q1.w);
            ps.throughput = v3(q2.x, q2.y, q2.z);
            const uint32_t job = __float_as_uint(q2.w);
            ps.radiance = v3(q3.x, q3.y, q3.z);
            uint32_t primary = __float_as_uint(q3.w);
            claim_chunk(chunk_n, idx_n);
            PT_TICK(8);
            bool keep = false;
            if (have) {
                bool done = true;
                do {
                    if (ps.bounce < P.cam.max_bounces) {
                        uint32_t fig;
                        n_bounce += 1u;
                        const bool first = ps.bounce == 0u;
                        done = bounce_step<FRAC, !LDS_SCENE>(sc, P.cam, ps, st, park, fig, n_shadow);
                        PT_TICK(3);
                        if (first) primary = fig;
                        if (!done && ps.bounce >= P.cam.max_bounces) done = true;
                    }
                } while (finish && !done);
                if (done) {
                    PT_SITE(36);
                    PixelJob J;
                    job_pixel(P, job, J, true);
                    float* slot = (float*)&P.staging[(size_t)J.k * P.frame_pixels + J.pix];
                    slot[0] = dm::fmax_(ps.radiance.x, 0.0f); slot[1] = dm::fmax_(ps.radiance.y, 0.0f); slot[2] = dm::fmax_(ps.radiance.z, 0.0f);
                    ((uint32_t*)slot)[3] = primary;
                    n_paths += 1u;
                } else {
                    keep = true;
                }
            }
            const unsigned long long going = __ballot(keep);
            if (going != 0ull) {
                uint32_t pos0 = 0u;
                if (lane == (uint32_t)__ffsll((long long)going) - 1u) pos0 = atomicAdd(&cv[1], (uint32_t)__popcll(going));
                pos0 = (uint32_t)__builtin_amdgcn_readlane((int)pos0, (int)((uint32_t)__ffsll((long long)going) - 1u));
                if (keep) {
                    const uint32_t pos = pos0 + (uint32_t)__popcll(going & ((1ull << lane) - 1ull));
                    if (pos < M) {
                        pool_store(nxt + pos, ps, job, primary);
                        nxt_keys[pos] = (uint8_t)regroup_key(kbox, valid, ps.ray);
                    }
                }
            }
            PT_TICK(9);
        }
        WG_SYNC_T(9);  // every chunk is done, every append made
        if (threadIdx.x == 0u) cv[0] = cv[1];
        parity ^= 1u;
    }
#ifdef PTX_PHASE_TIMING
    if (lane == 0u && P.stats) {
        for (int i = 0; i < 6; i++) atomicAdd(&P.stats[4 + i], (unsigned long long)ps.tick_acc[i]);
        for (int i = 6; i < PT_NTICKS; i++) atomicAdd(&P.stats[10 + i], (unsigned long long)ps.tick_acc[i]);
        const unsigned long long wall1 = wall_clock64();
        atomicMax(&P.stats[10], ~wall0);
        atomicMax(&P.stats[11], wall1);
        atomicAdd(&P.stats[12], wall1 - wall0);
        atomicAdd(&P.stats[13], 1ull);
    }
#endif
    const uint32_t sp = wave_sum(n_paths), sb = wave_sum(n_bounce), ss = wave_sum(n_shadow);
    if (lane == 0u && P.stats) {
        atomicAdd(&P.stats[0], (unsigned long long)sp);
        atomicAdd(&P.stats[1], (unsigned long long)sb);
        atomicAdd(&P.stats[2], (unsigned long long)ss);
    }
}

// The wave-scheduled kernel body (sched_dev.h): same buffers and results as trace_body; the grid is sized to
// the GPU, not to the frame, and lanes claim pixels as they go.
static_assert(SC_PIXEL_FIELDS + SC_PARK_FIELDS == PT_PIXEL_FIELDS + PT_PARK_FIELDS, "both kernel families share one LDS budget");
template <bool LDS_SCENE, bool FRAC, bool SAMPLES>
PT_DEV void sched_body(const TraceParams& P) {
    extern __shared__ StackEntry lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    WalkStack st;
    st.lds = lds_stack + (size_t)wave * P.sc.lds_stack_entries * 64u + lane;
    st.glb = walk_stack_glb(P.stack_spill, blockIdx.x * PTX_WAVES_PER_BLOCK + wave, lane);
    st.nl = P.sc.lds_stack_entries;
    // Sample jobs of the non-fractal kernels keep no running mean in the lane: pixel slots 0-3 do not exist there (the slot
    // base is shifted, so that the remaining slots keep their numbers: 4 job number, 5 occlusion queries, 6 paths finished,
    // 7 path-bounces), and the four slots go to the traversal stack instead -- two more entries in LDS for the global-scene
    // kernel (launch_trace sets lds_stack_entries), within the same 40 slots per lane.
    constexpr bool PARK_JOB = SAMPLES && !FRAC;
    constexpr uint32_t kNoSlots = PARK_JOB ? SC_SAMPLE_JOB_FREE : 0u;
    constexpr uint32_t kFields = SC_PIXEL_FIELDS - kNoSlots + SC_PARK_FIELDS + (FRAC ? SC_FRAC_FIELDS : 0u);  // per-lane LDS slots
    float* fields = (float*)(lds_stack + (size_t)PTX_WAVES_PER_BLOCK * P.sc.lds_stack_entries * 64u);
    DScene sc = P.sc;
    if (LDS_SCENE) {
        char* base = (char*)(fields + (size_t)PTX_WAVES_PER_BLOCK * kFields * 64u);
        uint32_t off = 0;
        auto stage = [&](const void* src, uint32_t bytes) -> void* {
            void* dst = base + off;
            copy_to_lds(dst, src, bytes);
            off += bytes;
            return dst;
        };
        sc.prims = (const DPrim*)stage(P.sc.prims, P.sc.nprim * (uint32_t)sizeof(DPrim));
        sc.inf = (const DPrim*)stage(P.sc.inf, P.sc.ninf * (uint32_t)sizeof(DPrim));
        sc.pairs = (const DNodePair*)stage(P.sc.pairs, P.sc.npair * (uint32_t)sizeof(DNodePair));
        sc.shade = (const DShade*)stage(P.sc.shade, P.sc.nfig * (uint32_t)sizeof(DShade));
        sc.lights = (const DLight*)stage(P.sc.lights, P.sc.nlight * (uint32_t)sizeof(DLight));
        sc.negs = (const DNeg*)stage(P.sc.negs, P.sc.nneg * (uint32_t)sizeof(DNeg));
        __syncthreads();
    }
    float* px_f = fields + (size_t)wave * kFields * 64u + lane - kNoSlots * 64u;
    uint32_t* px_u = (uint32_t*)px_f;
    float* park = px_f + SC_PIXEL_FIELDS * 64u;
    float* frac = park + SC_PARK_FIELDS * 64u;  // FRAC kernels only
    if (!PARK_JOB) { px_f[3 * 64] = 0.0f; px_f[4 * 64] = 0.0f; px_f[5 * 64] = 0.0f; }
    px_u[6 * 64] = 0u;  // samples of the current pixel finished
    px_u[7 * 64] = 0u;  // path-bounces, all pixels of this lane
    // FRAC: which lanes of each wave stand in an estimator, i.e. own a fractal march (republished every scheduling round),
    // for the other waves' free lanes to take marches from (sched_dev.h, FS_*); every status word starts idle before any wave looks.
    __shared__ unsigned long long est_lanes[2][PTX_WAVES_PER_BLOCK];
    __shared__ uint32_t est_serving[PTX_WAVES_PER_BLOCK];  // the estimator state a wave is serving right now (its own lanes are at their evaluations), or 0
    if (FRAC) {
        ((uint32_t*)frac)[TK_WORD * 64] = (uint32_t)FS_IDLE << TK_STATUS_SHIFT;
        if (lane == 0u) { est_lanes[0][wave] = 0ull; est_lanes[1][wave] = 0ull; est_serving[wave] = 0u; }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __syncthreads();
    }
    uint32_t n_shadow = 0;
    Trav tv;
    tv.ray.o = v3(0.0f, 0.0f, 0.0f);
    tv.ray.d = v3(0.0f, 0.0f, 1.0f);
    tv.inv_dir = v3(0.0f, 0.0f, 0.0f);
    tv.bound = 0.0f; tv.slot = PT_SLOT_NONE; tv.sub = 0u; tv.sp = 0; tv.cur = 0u; tv.cur_t = 0.0f; tv.pi = 0u; tv.pend = 0u;
    tv.have = false; tv.shadow = false;
    tv.m.t = 0.0f; tv.m.prev_d = 0.0f; tv.m.max_t = 0.0f; tv.m.i = 0u;
    uint32_t S = LS_IDLE;
    wave_claim_init(P, false);
    uint32_t job = blockIdx.x * PTX_BLOCK + threadIdx.x;  // the first job of every lane; the counter starts behind them
    // jobs are single samples (TraceParams::split_n): the fractal kernels decide per launch, the others are compiled for one
    // form or the other (the unused form's code cost the teapot 2.5 %)
    const bool split = SAMPLES && (!FRAC || P.split_n > 0u);
    if (P.n > 0u) {
        PixelJob J;
        if (claim_pixel<FRAC ? CLAIM_PLAIN : CLAIM_GUIDED>(P, job, true, px_f, J, split)) {
            Ray r0;
            r0.o = v3(0.0f, 0.0f, 0.0f);
            r0.d = v3(0.0f, 0.0f, 1.0f);
            S = next_path<FRAC ? CLAIM_PLAIN : CLAIM_GUIDED>(sc, P, tv, park, px_f, job, true, false, r0, v3(1.0f, 1.0f, 1.0f), v3(0.0f, 0.0f, 0.0f), 0u, 0u, split);
        }
    }
    if (PARK_JOB) { px_u[4 * 64] = job; px_u[5 * 64] = 0u; }
    // The scheduler.  A state is served when at least 1/share of the wave's live lanes wait in it, WALK before
    // POST before the fullest marcher; a marcher then keeps stepping while three quarters of its lanes are
    // still marching.  When no state reaches its share the fullest one is served, so every round makes
    // progress.  The shares are set per scene class by the host (measured): meshes, with or without ray-marched figures,
    // 8, 2, 4 (teapot 4K: 7 110 -> 7 900 Mpaths/s against 64, 2, 4, which was best before the per-wave claims), fractal scenes 64, 8, 4, small scenes with a Moebius strip 1, 2, 8
    // (with sample jobs: the fullest state unless half the lanes stand in POST or an eighth in a marcher; all.yaml 1 174 -> 1 340 Mpaths/s).
    for (;;) {
        uint32_t pop[LS_COUNT];
        uint32_t live = 0u, fullest = LS_IDLE, fullest_n = 0u, march = LS_IDLE, march_n = 0u;
#pragma unroll
        for (uint32_t s = LS_WALK; s < (uint32_t)LS_COUNT; s++) {
            pop[s] = (!FRAC && (s == (uint32_t)LS_MANDEL || s == (uint32_t)LS_JULIA)) ? 0u : (uint32_t)__popcll(__ballot(S == s));
            live += pop[s];
            // an estimator counts for what can be worked on: owners whose march another wave is running do not call for service
            if (FRAC && (s == (uint32_t)LS_MANDEL || s == (uint32_t)LS_JULIA))
                pop[s] = (uint32_t)__popcll(__ballot(S == s && (__hip_atomic_load(&((uint32_t*)frac)[TK_WORD * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> TK_STATUS_SHIFT) != (uint32_t)FS_BUSY));
            if (pop[s] > fullest_n) { fullest_n = pop[s]; fullest = s; }
            if (s >= (uint32_t)LS_TORUS && s <= (uint32_t)LS_JULIA && pop[s] > march_n) { march_n = pop[s]; march = s; }
        }
        if (FRAC) {
            const unsigned long long in_mandel = __ballot(S == LS_MANDEL), in_julia = __ballot(S == LS_JULIA);
            if (lane == 0u) {  // read by the other waves' lanes without a barrier: published and read as atomics
                __hip_atomic_store(&est_lanes[0][wave], in_mandel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&est_lanes[1][wave], in_julia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (live == 0u) break;
        const uint32_t single = P.walk_single;  // WALK mode
        // FRAC: a march another wave has finished for one of this wave's lanes is acted on whatever state is served next
        if (FRAC && (S == (uint32_t)LS_MANDEL || S == (uint32_t)LS_JULIA)) S = task_consume(sc, tv, S, frac, park, single);
        uint32_t pick = fullest;
        if (pop[LS_LEAF] * P.share_leaf >= live) pick = LS_LEAF;
        else if (pop[LS_WALK] * P.share_walk >= live) pick = LS_WALK;
        else if (pop[LS_POST] * P.share_post >= live) pick = LS_POST;
        else if (march_n * P.share_march >= live) pick = march;
        if (pick == LS_IDLE) {  // every live lane waits for a march that another wave is running
            __builtin_amdgcn_s_sleep(8);
        } else if (pick == LS_WALK) {
            for (;;) {
                if (S == LS_WALK) S = walk_step<FRAC>(sc, tv, st, P.cam.fractal_march_steps, single, frac, FRAC ? P.frac_fast : 0u);
                const uint32_t w = (uint32_t)__popcll(__ballot(S == LS_WALK));
                if (w == 0u || w * P.share_walk < live) break;
                if (single != 0u && (uint32_t)__popcll(__ballot(S == LS_LEAF)) * P.share_leaf >= live) break;
            }
        } else if (pick == LS_LEAF) {
            if (S == LS_LEAF) S = leaf_step<FRAC>(sc, tv, P.cam.fractal_march_steps, single, frac, FRAC ? P.frac_fast : 0u);
        } else if (pick == LS_POST) {
            if (S == LS_POST) {
                if (PARK_JOB) {
                    // sample jobs of the non-fractal kernels: the lane's job number and its tally of occlusion queries are read
                    // only here, so they wait in LDS slots instead of in registers (the kernel then fits four waves per SIMD
                    // without scratch)
                    uint32_t job_ = px_u[4 * 64], shadow_ = px_u[5 * 64];
                    S = post_step<FRAC>(sc, P, tv, park, frac, px_f, job_, shadow_, split);
                    px_u[4 * 64] = job_; px_u[5 * 64] = shadow_;
                } else {
                    S = post_step<FRAC>(sc, P, tv, park, frac, px_f, job, n_shadow, split);
                }
            }
        } else {
            const uint32_t n0 = pop[pick];
            if (FRAC && pick >= (uint32_t)LS_MANDEL) {
                // Serving an estimator: every lane of the wave works on a fractal march of that estimator.  A lane that owns one
                // takes its own; the others (and owners whose march another wave is running) take marches of the workgroup's
                // other waves.  A march stays in the lane's registers, evaluation after evaluation and marching step after
                // marching step, until it is finished or the wave turns to other states; a lane whose march finishes takes the next.
                if (lane == 0u) __hip_atomic_store(&est_serving[wave], pick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                float* cur = nullptr;  // the march this lane holds
                FracTask T;
                T.ray.o = v3(0.0f, 0.0f, 0.0f); T.ray.d = v3(0.0f, 0.0f, 1.0f);
                T.m.t = 0.0f; T.m.prev_d = 0.0f; T.m.max_t = 0.0f; T.m.i = 0u;
                for (int k9 = 0; k9 < 9; k9++) T.f.s[k9] = 0.0f;
                T.f.i = 0u; T.phase = 0u; T.ref = 0u; T.fast = 0u; T.nacc = v3(0.0f, 0.0f, 0.0f);
                uint32_t rot = wave * 16u;  // where this wave's free lanes start looking: moves on every pass, differs between waves
                for (;;) {
                    PT_SITE_HIST(60, live);
                    if (live >= 57u) {  // what a full wave's lanes are doing while an estimator is served
                        PT_SITE_ADD(70, (uint32_t)__popcll(__ballot(S == LS_WALK)));
                        PT_SITE_ADD(71, (uint32_t)__popcll(__ballot(S == LS_POST)));
                        PT_SITE_ADD(72, (uint32_t)__popcll(__ballot(S == LS_MANDEL)));
                        PT_SITE_ADD(73, (uint32_t)__popcll(__ballot(S == LS_JULIA)));
                        PT_SITE_ADD(74, (uint32_t)__popcll(__ballot(S == LS_IDLE)));
                    }
                    // owners of either estimator act on marches that have been finished (by this wave or another)
                    if (S == (uint32_t)LS_MANDEL || S == (uint32_t)LS_JULIA) S = task_consume(sc, tv, S, frac, park, single);
                    // lanes without a march: their own first ...
                    if (cur == nullptr && S == pick && task_claim(sc, frac, P.cam.fractal_march_steps, T)) cur = frac;
                    // ... then other waves': the r-th free lane takes the r-th march owner of the waves that are not serving this
                    // estimator themselves, counted from a moving start.  A stale or torn view of the published words costs a failed
                    // claim, never a wrong result: a march is taken by the READY -> BUSY exchange on its own status word, and all it
                    // needs is read from the claimed slot, not from the estimator being served here.
                    const unsigned long long free_lanes = __ballot(cur == nullptr);
                    const uint32_t n_free = (uint32_t)__popcll(free_lanes);
                    if (n_free >= (uint32_t)FRAC_ADOPT_MIN) {
                        unsigned long long cand[PTX_WAVES_PER_BLOCK - 1];
                        uint32_t total = 0u;
#pragma unroll
                        for (uint32_t k = 1u; k < (uint32_t)PTX_WAVES_PER_BLOCK; k++) {
                            const uint32_t w2 = (wave + k) % (uint32_t)PTX_WAVES_PER_BLOCK;
                            const bool theirs_too = __hip_atomic_load(&est_serving[w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == pick;
                            const unsigned long long seen = theirs_too ? 0ull : __hip_atomic_load(&est_lanes[pick - (uint32_t)LS_MANDEL][w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            // every lane read the same words: as scalars, so that the search below is scalar arithmetic
                            cand[k - 1u] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seen >> 32)) << 32) |
                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seen);
                            total += (uint32_t)__popcll(cand[k - 1u]);
                        }
                        if (cur == nullptr && total > 0u) {
                            uint32_t r = (uint32_t)__popcll(free_lanes & ((1ull << lane) - 1ull));
                            if (r < total) {
                                r += rot;  // rot < total
                                if (r >= total) r -= total;
                                float* theirs = nullptr;
#pragma unroll
                                for (uint32_t k = 1u; k < (uint32_t)PTX_WAVES_PER_BLOCK; k++) {
                                    const uint32_t c2 = (uint32_t)__popcll(cand[k - 1u]);
                                    if (theirs == nullptr) {
                                        if (r < c2) {
                                            const uint32_t w2 = (wave + k) % (uint32_t)PTX_WAVES_PER_BLOCK;
                                            theirs = fields + (size_t)w2 * kFields * 64u + (SC_PIXEL_FIELDS + SC_PARK_FIELDS) * 64u + nth_set_bit(cand[k - 1u], r);
                                        } else {
                                            r -= c2;
                                        }
                                    }
                                }
                                if (theirs && task_claim(sc, theirs, P.cam.fractal_march_steps, T)) cur = theirs;
                            }
                        }
                        rot += n_free;
                        while (total > 0u && rot >= total) rot -= total;
                        if (total == 0u) rot = 0u;
                    }
                    // a few estimator iterations of every march held; a finished march is published at once
                    PT_SITE_HIST(50, (uint32_t)__popcll(__ballot(cur != nullptr)));
                    if (cur != nullptr) {
                        // What follows a completed evaluation (what it was for, the march's next evaluation) is code for the few lanes
                        // whose evaluation has just ended: run after every iteration it cost more than the iterations themselves, run
                        // once per six iterations it left the finished lanes idle (40 of 55 held marches iterating).
                        const DPrim& g = slot_prim(sc, T.ref);
                        float value = 0.0f;
                        bool ended = false;
                        // lanes whose evaluation has ended wait for company: the code that follows an evaluation runs when
                        // FRAC_GLUE_MIN lanes want it (or the pass is over), and those lanes then iterate on with their next evaluation
                        for (int it = 0; it < FRAC_ITERS_PER_PASS && cur != nullptr; it++) {
                            if (!ended) ended = frac_eval_iter(g, T.f, value, T.fast != 0u);
                            const bool last = it + 1 == FRAC_ITERS_PER_PASS;
                            if ((uint32_t)__popcll(__ballot(ended)) >= (uint32_t)FRAC_GLUE_MIN || last) {
                                if (ended) {
                                    const uint32_t fin = task_advance(sc, T, value, P.cam.fractal_march_steps);
                                    ended = false;
                                    if (fin != 0u) {
                                        task_finish(cur, T, fin);
                                        cur = nullptr;
                                    }
                                }
                            }
                        }
                    }
                    const uint32_t m = (uint32_t)__popcll(__ballot(S == pick));
                    const uint32_t held = (uint32_t)__popcll(__ballot(cur != nullptr));
                    if (held < (uint32_t)FRAC_HELD_MIN) break;  // too little held and nothing more to be had: other states, or other waves, have to move first
                    if (m * FRAC_EXIT_DEN < n0 * FRAC_EXIT_NUM) break;
                }
                if (cur != nullptr) task_store(cur, T, FS_READY);
                if (S == (uint32_t)LS_MANDEL || S == (uint32_t)LS_JULIA) S = task_consume(sc, tv, S, frac, park, single);
                if (lane == 0u) __hip_atomic_store(&est_serving[wave], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                for (;;) {
                    PT_SITE_HIST(50, (uint32_t)__popcll(__ballot(S == pick)));
                    PT_SITE_HIST(60, live);
                    if (S == pick) S = march_step<FRAC>(sc, tv, pick, P.cam.fractal_march_steps, single);
                    const uint32_t m = (uint32_t)__popcll(__ballot(S == pick));
                    if (m == 0u || m * FRAC_EXIT_DEN < n0 * FRAC_EXIT_NUM) break;
                }
            }
        }
    }
    const uint32_t n_bounce = px_u[7 * 64];
    if (PARK_JOB) n_shadow = px_u[5 * 64];
    uint32_t sb = wave_sum(n_bounce), ss = wave_sum(n_shadow);
    const uint32_t sp = split ? wave_sum(px_u[6 * 64]) : 0u;
    if (!split) flush_paths(P);
    if (lane == 0u && P.stats) {
        if (split) atomicAdd(&P.stats[0], (unsigned long long)sp);
        atomicAdd(&P.stats[1], (unsigned long long)sb);
        atomicAdd(&P.stats[2], (unsigned long long)ss);
    }
}

// Folds the staged samples of each pixel job into the pixel's running mean (and second moment), in sample
// order, with the arithmetic of next_path; one lane per pixel job of the launch.
extern "C" __global__ void __launch_bounds__(256) ptx_fold_kernel(const TraceParams P) {
    PixelJob J;
    if (!job_pixel(P, blockIdx.x * 256u + threadIdx.x, J)) return;
    float4 a = P.accum[J.pix];
    V3 mean = v3(a.x, a.y, a.z);
    V3 m2 = v3(0.0f, 0.0f, 0.0f);
    if (P.m2) m2 = v3(P.m2[J.pix * 3], P.m2[J.pix * 3 + 1], P.m2[J.pix * 3 + 2]);
    for (uint32_t k = 0; k < P.n; k++) {
        float4 s = P.staging[(size_t)k * P.frame_pixels + J.pix];
        V3 r = v3(s.x, s.y, s.z);
        float nn = dm::fmax_((float)(P.first_sample + k), 1.0f);
        V3 delta = r - mean;
        V3 nm = mean + vdivs(delta, nn);
        if (P.m2) {
            V3 d2 = r - nm;
            m2.x = dm::fma_(delta.x, d2.x, m2.x);
            m2.y = dm::fma_(delta.y, d2.y, m2.y);
            m2.z = dm::fma_(delta.z, d2.z, m2.z);
        }
        mean = nm;
    }
    P.accum[J.pix] = make_float4(mean.x, mean.y, mean.z, 1.0f);
    if (P.m2) { P.m2[J.pix * 3] = m2.x; P.m2[J.pix * 3 + 1] = m2.y; P.m2[J.pix * 3 + 2] = m2.z; }
    P.out_rgba8[J.pix] = tonemap_rgba8(P.cam, mean);
    P.primary_ids[J.pix] = __float_as_uint(P.staging[(size_t)(P.n - 1u) * P.frame_pixels + J.pix].w);
}

// Four variants: scene records in global memory or staged in LDS; with or without the fractal marchers.
// Without them the kernel fits 128 VGPRs (4 waves per SIMD); with them 3 waves.
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 4) ptx_trace_kernel(const TraceParams P) { trace_body<false, false, false>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 4) ptx_trace_kernel_lds(const TraceParams P) { trace_body<true, false, PTX_CLAIM_LDS>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 4) ptx_trace_kernel_lds_ranged(const TraceParams P) { trace_body<true, false, PTX_CLAIM_LDS, CLAIM_RANGED>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 4) ptx_trace_kernel_lds_sorted(const TraceParams P) { sorted_body<true, false>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 4) ptx_trace_kernel_sorted(const TraceParams P) { sorted_body<false, false>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 3) ptx_trace_kernel_frac(const TraceParams P) { trace_body<false, true, false>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 3) ptx_trace_kernel_lds_frac(const TraceParams P) { trace_body<true, true, PTX_CLAIM_LDS>(P); }
#ifndef PTX_SCHED_WAVES
#define PTX_SCHED_WAVES 4
#endif
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, PTX_SCHED_WAVES) ptx_sched_kernel(const TraceParams P) { sched_body<false, false, false>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, PTX_SCHED_WAVES) ptx_sched_kernel_lds(const TraceParams P) { sched_body<true, false, false>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, PTX_SCHED_WAVES) ptx_sched_kernel_samples(const TraceParams P) { sched_body<false, false, true>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, PTX_SCHED_WAVES) ptx_sched_kernel_lds_samples(const TraceParams P) { sched_body<true, false, true>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 3) ptx_sched_kernel_frac(const TraceParams P) { sched_body<false, true, true>(P); }
extern "C" __global__ void __launch_bounds__(PTX_BLOCK, 3) ptx_sched_kernel_lds_frac(const TraceParams P) { sched_body<true, true, true>(P); }

// Chan's parallel update of (mean, M2) with another partial result, then re-tonemap.
extern "C" __global__ void ptx_merge_kernel(float4* accum, float* m2, uint32_t* out_rgba8, uint32_t first, uint32_t n_pixels,
                                            float n_a, const float4* o_mean, const float* o_m2, float n_b, const ptx_camera cam) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    size_t p = (size_t)first + i;
    float4 a = accum[p], b = o_mean[i];
    float n = n_a + n_b;
    float wb = n_b / n, wab = n_a * n_b / n;
    V3 ma = v3(a.x, a.y, a.z), mb = v3(b.x, b.y, b.z);
    V3 d = mb - ma;
    V3 m = ma + d * wb;
    if (n_a == 0.0f) m = mb;  // an accumulator without samples contributes nothing, whatever it holds
    if (m2 && o_m2) {
        if (n_a == 0.0f) {
            m2[p * 3] = o_m2[(size_t)i * 3]; m2[p * 3 + 1] = o_m2[(size_t)i * 3 + 1]; m2[p * 3 + 2] = o_m2[(size_t)i * 3 + 2];
        } else {
            m2[p * 3] = m2[p * 3] + o_m2[(size_t)i * 3] + d.x * d.x * wab;
            m2[p * 3 + 1] = m2[p * 3 + 1] + o_m2[(size_t)i * 3 + 1] + d.y * d.y * wab;
            m2[p * 3 + 2] = m2[p * 3 + 2] + o_m2[(size_t)i * 3 + 2] + d.z * d.z * wab;
        }
    }
    accum[p] = make_float4(m.x, m.y, m.z, 1.0f);
    out_rgba8[p] = tonemap_rgba8(cam, m);
}

// Re-derives the rgba8 image from the accumulation buffer (tonemap.wgsl + rgba8unorm store).
extern "C" __global__ void ptx_tonemap_kernel(const float4* accum, uint32_t* out_rgba8, uint32_t first, uint32_t n_pixels, const ptx_camera cam) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = accum[(size_t)first + i];
    out_rgba8[(size_t)first + i] = tonemap_rgba8(cam, v3(a.x, a.y, a.z));
}

// Tile partition (ptx_set_partition): the 16x16-pixel blocks of the tile whose row-major index is congruent to `rank` modulo
// `world` belong to that rank.  pack: the j-th block of `rank` goes to packed[j * 256 ..], one workgroup per block, rows of
// 16 pixels (256 contiguous bytes); unpack: the reverse, into the accumulation buffer.  Pixels of an edge block outside
// the tile are packed as zeros and skipped on the way back.
struct BlockPart { uint32_t tile_x0, tile_y0, tile_w, tile_h, blocks_x, width, rank, world; };
PT_DEV bool block_pixel(const BlockPart& B, uint32_t j, uint32_t p, size_t& pix) {
    const uint32_t blk = j * B.world + B.rank, bx = blk % B.blocks_x, by = blk / B.blocks_x;
    const uint32_t tx = bx * 16u + (p & 15u), ty = by * 16u + (p >> 4);
    pix = (size_t)(B.tile_y0 + ty) * B.width + B.tile_x0 + tx;
    return tx < B.tile_w && ty < B.tile_h;
}
extern "C" __global__ void __launch_bounds__(256) ptx_pack_blocks_kernel(const float4* accum, float4* packed, const BlockPart B) {
    size_t pix;
    const bool in = block_pixel(B, blockIdx.x, threadIdx.x, pix);
    packed[(size_t)blockIdx.x * 256u + threadIdx.x] = in ? accum[pix] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
extern "C" __global__ void __launch_bounds__(256) ptx_unpack_blocks_kernel(float4* accum, const float4* packed, const BlockPart B) {
    size_t pix;
    if (block_pixel(B, blockIdx.x, threadIdx.x, pix)) accum[pix] = packed[(size_t)blockIdx.x * 256u + threadIdx.x];
}

// Post-process pass: one 16 x (16 ROWS)-pixel tile per workgroup, ROWS vertically adjacent pixels per thread (4 for chains
// with an oil painting, whose pixels then share their columns; 1 otherwise: more pixels per thread cost the pointwise
// chains 3-8 % in registers).  TILE: the tile and a halo of `halo` texels are staged in LDS (clamped to the image
// edge) and every effect reads from there; otherwise reads go to the accumulation buffer.
template <bool TILE, int ROWS>
PT_DEV void post_body(const float4* accum, uint32_t* out_rgba8, const ptx_post_params pp, int halo) {
    PostImage im;
    im.accum = accum;
    im.w = (int)pp.width;
    im.h = (int)pp.height;
    im.tx0 = (int)blockIdx.x * PP_TILE;
    im.ty0 = (int)blockIdx.y * PP_TILE * ROWS;
    im.halo = halo;
    if (TILE) {
        const int side = PP_TILE + 2 * halo, rows = PP_TILE * ROWS + 2 * halo;
        for (int i = (int)threadIdx.x; i < side * rows; i += PP_TILE * PP_TILE) {
            int x = im.tx0 - halo + i % side, y = im.ty0 - halo + i / side;
            int cx = x < 0 ? 0 : (x > im.w - 1 ? im.w - 1 : x), cy = y < 0 ? 0 : (y > im.h - 1 ? im.h - 1 : y);
            pp_tile[i] = accum[(size_t)cy * im.w + cx];
        }
        __syncthreads();
    }
    const int x = im.tx0 + (int)(threadIdx.x % PP_TILE), y0 = im.ty0 + (int)(threadIdx.x / PP_TILE) * ROWS;
    if (x >= im.w || y0 >= im.h) return;
    V3 result[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) result[k] = pp_read_clamped<TILE>(im, x, y0 + k < im.h ? y0 + k : im.h - 1);  // post_process.wgsl:95-101
    for (uint32_t i = 0; i < pp.effect_count; i++) {
        if (TILE && ROWS > 1 && pp.effects[i] == 5u) {  // the oil painting shares its columns between the thread's pixels
            pp_oil_rows<ROWS>(im, x, y0, (int)pp.oil_radius, result);
            continue;
        }
#pragma unroll
        for (int k = 0; k < ROWS; k++)
            if (y0 + k < im.h) result[k] = pp_effect<TILE>(im, result[k], x, y0 + k, pp.effects[i], pp);
    }
#pragma unroll
    for (int k = 0; k < ROWS; k++)
        if (y0 + k < im.h)
            out_rgba8[(size_t)(y0 + k) * im.w + x] = unorm8(result[k].x) | (unorm8(result[k].y) << 8) | (unorm8(result[k].z) << 16) | 0xFF000000u;
}
extern "C" __global__ void __launch_bounds__(PP_TILE* PP_TILE) ptx_post_kernel_tile(const float4* accum, uint32_t* out_rgba8,
                                                                                    const ptx_post_params pp, int halo) {
    post_body<true, 1>(accum, out_rgba8, pp, halo);
}
extern "C" __global__ void __launch_bounds__(PP_TILE* PP_TILE) ptx_post_kernel_tile_rows(const float4* accum, uint32_t* out_rgba8,
                                                                                         const ptx_post_params pp, int halo) {
    post_body<true, PP_ROWS>(accum, out_rgba8, pp, halo);
}
extern "C" __global__ void __launch_bounds__(PP_TILE* PP_TILE) ptx_post_kernel(const float4* accum, uint32_t* out_rgba8,
                                                                               const ptx_post_params pp, int halo) {
    post_body<false, 1>(accum, out_rgba8, pp, halo);
}

extern "C" __global__ void ptx_kat_math_kernel(int op, uint32_t n, const float* a, const float* b, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i], r = 0.0f;
    switch (op) {
        case 0: r = dm::sin_(x); break;
        case 1: r = dm::cos_(x); break;
        case 2: r = dm::asin_(x); break;
        case 3: r = dm::acos_(x); break;
        case 4: r = dm::log_(x); break;
        case 5: r = dm::log2_(x); break;
        case 6: r = dm::exp2_(x); break;
        case 7: r = dm::atan2_(x, y); break;
        case 8: r = dm::pow_(x, y); break;
        case 9: r = dm::fmin_(x, y); break;
        case 10: r = dm::fmax_(x, y); break;
        case 11: r = dm::sqrt_(x); break;
        case 12: r = dm::div_(x, y); break;
        case 13: r = dm::fma_(x, y, x); break;
        case 14: r = dm::rcp_(x); break;
        default: break;
    }
    out[i] = r;
}

// Exhaustive check of the short reciprocal / square-root sequences: every one of the 2^32 inputs against the
// definition (division contract in dmath_dev.h) evaluated with the compiler's IEEE divide and square root.
PT_DEV float rcp_definition(float y) {
    if (y != y) return y;
    const uint32_t sign = __float_as_uint(y) & 0x80000000u;
    if (dm::abs_(y) < 1.17549435e-38f) return __uint_as_float(sign | 0x7f800000u);
    float r = 1.0f / y;
    if (dm::abs_(r) < 1.17549435e-38f) return __uint_as_float(sign);
    return r;
}
extern "C" __global__ void ptx_kat_exhaustive_kernel(int op, uint32_t hi, unsigned long long* n_bad, uint32_t* first_bad) {
    const uint32_t bits = (hi << 24) | (blockIdx.x * blockDim.x + threadIdx.x);
    const float x = __uint_as_float(bits);
    const float got = op == 0 ? dm::rcp_(x) : dm::sqrt_(x);
    const float want = op == 0 ? rcp_definition(x) : __builtin_sqrtf(x);
    const bool same = (got != got && want != want) || __float_as_uint(got) == __float_as_uint(want);
    if (!same && atomicAdd(n_bad, 1ull) == 0ull) *first_bad = bits;
}

// Unit hook: one ray per lane through the same traversal code as the render kernels (WALK as they pick it).
template <bool WALK>
PT_DEV void kat_trace_body(const DScene& sc, StackEntry* stack_spill, const ptx_camera& cam, uint32_t n, const float* rays, const float* max_t,
                           uint32_t* hit_id, float* hit_t, uint32_t* shadow) {
    extern __shared__ StackEntry lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    WalkStack st;
    // the loop variant keeps the whole stack in LDS, the walk variant its first PT_LDS_STACK entries
    st.lds = lds_stack + (size_t)wave * (WALK ? min(sc.stack_entries, PT_LDS_STACK) : sc.stack_entries) * 64u + lane;
    st.glb = walk_stack_glb(stack_spill, blockIdx.x * PTX_WAVES_PER_BLOCK + wave, lane);
    st.nl = WALK ? min(sc.stack_entries, PT_LDS_STACK) : sc.stack_entries;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray ray;
    ray.o = v3p(rays + 6 * (size_t)i);
    ray.d = v3p(rays + 6 * (size_t)i + 3);
    Ray r2 = ray;
    float t;
    uint32_t slot, sub;
    trace_closest<true, WALK>(sc, r2, vrcp(ray.d), cam.fractal_march_steps, st, t, slot, sub);
    hit_id[i] = slot == PT_SLOT_NONE ? 0xFFFFFFFFu : slot_prim(sc, slot).fig;
    hit_t[i] = slot == PT_SLOT_NONE ? PT_MAX_T : t;
    if (shadow && max_t) shadow[i] = trace_shadow<true, WALK>(sc, ray, max_t[i], cam.fractal_march_steps, st) ? 1u : 0u;
}
extern "C" __global__ void __launch_bounds__(PTX_BLOCK) ptx_kat_trace_kernel(const DScene sc, StackEntry* stack_spill, const ptx_camera cam,
                                                                            uint32_t n, const float* rays, const float* max_t,
                                                                            uint32_t* hit_id, float* hit_t, uint32_t* shadow) {
    kat_trace_body<true>(sc, stack_spill, cam, n, rays, max_t, hit_id, hit_t, shadow);
}
extern "C" __global__ void __launch_bounds__(PTX_BLOCK) ptx_kat_trace_kernel_loop(const DScene sc, const ptx_camera cam, uint32_t n,
                                                                                 const float* rays, const float* max_t, uint32_t* hit_id,
                                                                                 float* hit_t, uint32_t* shadow) {
    kat_trace_body<false>(sc, nullptr, cam, n, rays, max_t, hit_id, hit_t, shadow);
}

// ================================================================================================
// Host side
// ================================================================================================
// ---- launch policy ----------------------------------------------------------------------------------------------------
// What a ptx_trace* call runs is a function of the uploaded scene's class, the options, the launch's length and the
// sequence of calls on the context -- of nothing else (not of host timing).  configure() evaluates that function once per
// class of launch length and keeps the result as a table (ptx_ctx::policy: rows by ascending `upto`), together with the
// one form every launch can fall back to, whole-pixel jobs (ptx_ctx::pixels_form: ptx_trace_n_static, radiance capture,
// a host that waits for every frame).  launch_trace looks the row up; what depends on the frame's size (reservation
// sizes, chunks, launches in flight within the staging memory) is worked out from the row there (resolve_launch).
typedef void (*TraceKernel)(const TraceParams);
enum : uint8_t { JOBS_PIXELS = 0, JOBS_SAMPLES = 1, JOBS_POOLS = 2 };
struct LaunchForm {
    uint8_t jobs = JOBS_PIXELS;   // whole pixels folded in the lane | single samples staged for ptx_fold_kernel | the same from sorted path pools
    bool sched = false;           // the wave-scheduled family (else lockstep)
    bool claim = false;           // the grid is the resident workgroups and lanes claim jobs (else one workgroup per 16x16 block)
    TraceKernel kernel = nullptr;
    uint32_t counters = 1;        // job counters the reservations are spread over (ptx_trace_kernel_lds_ranged: up to sixteen)
    uint32_t guided_k = 0;        // reservations of 1/(k * waves) of the jobs left (0: fixed size)
    uint32_t paced_m = 0;         // reservations sized to come m * waves * 10 ns apart (0: fixed size)
    uint32_t pool_m = 0;          // JOBS_POOLS: paths per workgroup pool
    bool spread = false;          // JOBS_POOLS: a reservation samples the frame at a stride (else block order)
    uint32_t upto_jobs = 0;       // launches of more jobs (pixels of the tile's blocks x samples) take the row's other form (0: any size)
    uint32_t flights = 0;         // launches of sample jobs in flight at most; 0 = PTX_FLIGHTS_DEFAULT within the staging memory
};
struct PolicyRow { uint32_t upto; LaunchForm form, big; };  // launches of up to `upto` samples

struct ptx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scene
    bool have_scene = false;
    DScene sc{};
    std::vector<void*> scene_allocs;
    // frame
    uint32_t w = 0, h = 0;
    uint32_t tx0 = 0, ty0 = 0, tw = 0, th = 0;
    uint32_t part_rank = 0, part_world = 1;
    float4* accum = nullptr;
    uint32_t* out = nullptr;
    uint32_t* ids = nullptr;
    float* m2 = nullptr;
    bool moments = false;
    float* dbg = nullptr;
    uint32_t dbg_cap = 0;
    unsigned long long* stats = nullptr;
    bool sched = false;           // the uploaded scene runs the wave-scheduled kernels
    uint32_t shares[3] = {4u, 2u, 4u};  // their scheduling shares (walk, post, march)
    uint32_t share_leaf = 4u;           // and the share of lanes at leaf primitives (one-visit-per-step mode)
    uint32_t claim_tile = 1u;     // pixel jobs a wave claims at a time (1 or 64)
    uint32_t n_cu = 0;            // compute units of the device
    bool split = true;            // sample-granular jobs + fold where they pay (launch_trace)
    uint32_t split_upto = 0x7FFFFFFFu;  // launches of up to this many samples use them (the split_small option)
    bool plain_mesh = false;      // wave-scheduled, no marched figures, shallow tree: the scenes whose paths cost least
    uint32_t type_mask = 0;       // 1 << type over the BVH primitives of the uploaded scene
    uint32_t n_marched = 0;       // torus / Moebius figures among them
    const char* last_kernel = nullptr;  // the trace kernel of the latest launch since the scene was uploaded
    std::string last_info;              // ... and what else the launch policy chose for it (ptx_last_launch_info)
    std::string plan_info;              // ptx_launch_plan's answer
    std::vector<PolicyRow> policy;      // the launch policy of the uploaded scene under the current options (configure())
    LaunchForm pixels_form;             // ... and its whole-pixel form
    bool drained = false;               // the host has synchronised with the context since its latest trace launch
    // ptx_set_option: explicit overrides of the per-scene choices (tests and measurement runs); the defaults mean "as measured"
    struct Options {
        int family = -1;          // kernel_family: -1 per scene, 0 lockstep, 1 wave-scheduled
        int upload = -1;          // upload_path: -1 by size, 0 re-layout on the host, 1 on the device
        uint32_t claim_tile = 0;  // 0 per scene, 1 or 64
        uint32_t claim_ranges = 0; // 0 per scene, else 1..16: job counters the reservations are spread over
        int claim_paced = -1;     // -1 per scene and launch, 0 off, m: a wave's reservations are sized to come m * waves * 10 ns apart
        int claim_cap = -1;       // with claim_paced: -1 = 8, 0 none, k: a reservation is at most 1/(k * waves) of the jobs left
        int claim_guided = -1;    // -1 per scene and launch, 0 reservations of claim_tile jobs, k >= 4: of 1/(k * waves) of the jobs left
        uint32_t split_chunk = 0; // 0 = as many samples per chunk as the staging bound allows
        uint32_t shares[4] = {0u, 0u, 0u, 0u};  // walk, post, march, leaf; 0 per scene
        int walk_mode = -1;       // -1 per scene, 0 / 1 / 2 (TraceParams::walk_single)
        int sorted_pool = -1;     // -1 per scene and launch, 0 / 1: lockstep LDS kernel over sorted path pools (sample jobs, long launches)
        int arith_mode = 0;       // 0 exact (bit for bit the oracle's arithmetic), 1 tolerance mode of the fractal estimators (sched_dev.h, frac_fast)
        int host_waits = -1;      // host_waits_per_frame: -1 by the call sequence, 0 one-sample launches always overlap, 1 they never do (launch_trace)
        int pool_over = -1;       // -1 per scene (4; lit meshes 0): launches of more samples than this may trace from pools
        uint32_t pool_paths = 0;  // 0 = 4096, else paths per workgroup pool (a power of two from 256 to 16384)
        int pool_guided = -1;     // -1 = 0: a top-up reserves what the pool lacks; k: at most 1/(k * workgroups) of the jobs left (measured: helps
                                  // rings.yaml at one sample per launch, 1.50 -> 0.90 ms with k = 4, costs the Cornell box 0.855 -> 0.917)
        int pool_spread = -1;     // -1 one-sample launches only; 1: a pool's reservation samples the frame at a stride (job_pixel, job_transposed); 0: block order
        int pool_finish = -1;     // -1 = 256; with no job left, a pool of up to this many paths is traced to the end without further rounds
        uint32_t grid_per_cu = 0; // 0 = every resident workgroup
        int64_t split_upto = -1;  // split_small: -1 per scene, else launches of up to this many samples run sample jobs (fractal scenes on the scheduled kernels always do)
        uint32_t flights = 0;     // launches of sample-granular jobs in flight at once; 0 = as many as the staging bound allows (at most PTX_FLIGHTS)
        bool verbose = false, bvh_timing = false;
    } opt;
    // Launches of sample-granular jobs in flight (launch_trace): each traces on a stream of its own into a staging buffer of
    // its own; the folds follow one another on `stream`.
    struct Flight {
        hipStream_t stream = nullptr;
        hipEvent_t traced = nullptr, folded = nullptr;  // its trace kernel / the fold of its staging buffer has finished
        bool used = false;
        float4* staging = nullptr;  // staged samples, [sample][frame pixel]: radiance in xyz, the primary hit's figure in w
        size_t staging_n = 0;
        WaveState* wave_state = nullptr;
        size_t wave_state_n = 0;
        StackEntry* spill = nullptr;  // deep-stack entries of its waves (scenes that need them)
        size_t spill_waves = 0;
        // its set of sorted path pools (launches that trace from pools): for pool_wgs workgroups, pool_m paths each
        PoolPath* pool = nullptr;
        uint8_t* pool_keys = nullptr;
        uint16_t* pool_order = nullptr;
        size_t pool_wgs = 0;
        uint32_t pool_m = 0;
    };
    Flight flights[PTX_FLIGHTS];
    uint32_t flight_seq = 0;
    uint32_t* flight_counters = nullptr;  // PTX_CLAIM_RANGES job counters per flight and for the context's own stream, 256 bytes apart
    hipEvent_t barrier = nullptr;         // everything enqueued on `stream` by calls other than those launches, up to the last of them
    bool order_dirty = true;              // such a call has been made since `barrier` was recorded
    const void* occ_kernel = nullptr;  // the last occupancy query: kernel, LDS bytes -> resident workgroups per compute unit
    size_t occ_lds = 0;
    int occ_per_cu = 0;
    StackEntry* spill = nullptr;  // stack entries beyond the LDS part, one area per wave of the largest launch so far
    size_t spill_waves = 0;
    WaveState* wave_state = nullptr;  // claim record of every wave of the largest grid so far (the kernels initialise their own)
    size_t wave_state_n = 0;
    unsigned char* static_scratch = nullptr;  // this device's copy of ptx_static_scratch (its address, not the host-side symbol)
    // multi-GPU exchange (collective.h)
    ncclComm_t comm = nullptr;
    bool comm_owned = false;      // created by ptx_comm_init (destroyed with the context) or handed in by ptx_comm_adopt
    int comm_rank = 0, comm_world = 1;
    float* xchg = nullptr;        // slices received from the other ranks
    size_t xchg_floats = 0;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    bool found_idle = false;       // the latest one-sample launch came after a synchronising call (launch_trace)
    std::vector<hipEvent_t> free_events;
    double acc_ms = 0.0;
    uint32_t acc_launches = 0;
    std::vector<float> launch_ms;  // duration of every trace call since the last ptx_kernel_times
};

namespace {

int fail(ptx_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PTX_ERR_OOM : PTX_ERR_HIP,                          \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                  \
    } while (0)

void free_scene(ptx_ctx* c) {
    if (c->spill) (void)hipFree(c->spill);
    c->spill = nullptr;
    c->spill_waves = 0;
    for (void* p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->have_scene = false;
    memset(&c->sc, 0, sizeof(c->sc));
}
void free_frame(ptx_ctx* c) {
    if (c->accum) (void)hipFree(c->accum);
    if (c->out) (void)hipFree(c->out);
    if (c->ids) (void)hipFree(c->ids);
    if (c->m2) (void)hipFree(c->m2);
    if (c->dbg) (void)hipFree(c->dbg);
    c->accum = nullptr; c->out = nullptr; c->ids = nullptr; c->m2 = nullptr; c->dbg = nullptr;
    c->dbg_cap = 0;
    for (auto& f : c->flights) {  // callers have drained `stream`, whose folds wait for every trace
        if (f.staging) (void)hipFree(f.staging);
        f.staging = nullptr; f.staging_n = 0;
    }
    c->order_dirty = true;
}

// Spreads the jobs behind the lanes' first ones (jobs 0 .. first - 1) over `ranges` counters.
static void set_ranges(TraceParams& P, uint32_t ranges, uint32_t first) {
    P.job_first = first;
    P.claim_ranges = P.claim_tile > 1u ? ranges : 1u;
    const uint32_t rest = P.total_jobs > first ? P.total_jobs - first : 0u;
    P.range_len = ((rest + P.claim_ranges - 1u) / P.claim_ranges + 63u) & ~63u;
}

static uint32_t log2_ceil(size_t v) {
    uint32_t s = 0;
    while (((size_t)1 << s) < v) s++;
    return s;
}

// Makes room for the deep stack entries of `waves` waves (only scenes whose tree is deeper than PT_LDS_STACK
// and that run the global-memory kernels need any).
int ensure_spill(ptx_ctx* c, size_t waves, StackEntry** out) {
    *out = nullptr;
    if (c->sc.lds_scene_bytes > 0u || c->sc.stack_entries <= PT_LDS_STACK) return PTX_OK;
    if (waves > c->spill_waves) {
        if (c->spill) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->spill); c->spill = nullptr; c->spill_waves = 0; }
        hipError_t e = hipMalloc((void**)&c->spill, waves * (31u - PT_LDS_STACK) * 64u * sizeof(StackEntry));
        if (e != hipSuccess) return fail(c, PTX_ERR_OOM, "stack spill area: hipMalloc");
        c->spill_waves = waves;
    }
    *out = c->spill;
    return PTX_OK;
}

template <class T>
int upload_vec(ptx_ctx* c, const std::vector<T>& v, const T** dst) {
    *dst = nullptr;
    size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(T);
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, bytes));
    c->scene_allocs.push_back(p);
    if (!v.empty()) HIP_TRY(c, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T*)p;
    return PTX_OK;
}

int collect_timing(ptx_ctx* c) {
    for (auto& pr : c->pending) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventSynchronize(pr.second));
        HIP_TRY(c, hipEventElapsedTime(&ms, pr.first, pr.second));
        c->acc_ms += ms;
        c->acc_launches += 1;
        if (c->launch_ms.size() < 65536u) c->launch_ms.push_back(ms);
        c->free_events.push_back(pr.first);
        c->free_events.push_back(pr.second);
    }
    if (!c->pending.empty()) c->drained = true;  // the latest launch's last event has completed
    c->pending.clear();
    return PTX_OK;
}
int get_event(ptx_ctx* c, hipEvent_t* e) {
    if (!c->free_events.empty()) { *e = c->free_events.back(); c->free_events.pop_back(); return PTX_OK; }
    HIP_TRY(c, hipEventCreate(e));
    return PTX_OK;
}

struct DevBufs {  // device allocations of one build, released on every exit path
    std::vector<void*> ptrs;
    ~DevBufs() { release(); }
    void release() { for (void* p : ptrs) (void)hipFree(p); ptrs.clear(); }
    hipError_t alloc(void** out, size_t bytes) {
        hipError_t e = hipMalloc(out, bytes ? bytes : 4);
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
};
int scan_u32(ptx_ctx* c, const uint32_t* in, uint32_t n, uint32_t* out, uint32_t* sums) {
    const uint32_t nblocks = (n + bvhb::SCAN_TILE - 1u) / bvhb::SCAN_TILE;
    hipLaunchKernelGGL(bvhb::scan_sums_kernel, dim3(nblocks), dim3(256), 0, c->stream, in, n, sums);
    hipLaunchKernelGGL(bvhb::scan_top_kernel, dim3(1), dim3(256), 0, c->stream, sums, nblocks, out + n);
    hipLaunchKernelGGL(bvhb::scan_final_kernel, dim3(nblocks), dim3(256), 0, c->stream, in, n, (const uint32_t*)sums, out);
    HIP_TRY(c, hipGetLastError());
    return PTX_OK;
}

// The re-layout on the device (relayout.h): the raw arrays are copied over, one lane per record builds the shading,
// primitive and node-pair records and makes ptx_upload_scene's checks; what is small (lights, planes, subtraction
// shapes, textures) stays with the caller.  Leaves sc.prims / sc.shade / sc.pairs set and owned by the scene.
int relayout_on_device(ptx_ctx* c, const ptx_shape* figs, uint32_t nfig, const ptx_material* mats, uint32_t nmat,
                       const ptx_bvh_node* nodes, uint32_t nnode, const uint32_t* prims, uint32_t nprim, bool have_tree,
                       DScene& sc, uint32_t& npair, uint32_t& depth, uint32_t& type_mask, uint32_t& root_ref,
                       uint32_t& neg_count, uint32_t& sky_index) {
    using namespace relay;
    DevBufs tmp;  // the raw copies and the scratch: released when the records are built
    ptx_shape* d_figs;
    ptx_material* d_mats;
    ptx_bvh_node* d_nodes;
    uint32_t *d_prims, *d_res, *d_inner, *d_pair_of, *d_sums, *d_depth;
    hipStream_t st = c->stream;
    HIP_TRY(c, tmp.alloc((void**)&d_figs, (size_t)nfig * sizeof(ptx_shape)));
    HIP_TRY(c, tmp.alloc((void**)&d_mats, (size_t)nmat * sizeof(ptx_material)));
    HIP_TRY(c, tmp.alloc((void**)&d_nodes, (size_t)nnode * sizeof(ptx_bvh_node)));
    HIP_TRY(c, tmp.alloc((void**)&d_prims, (size_t)nprim * 4));
    HIP_TRY(c, tmp.alloc((void**)&d_res, R_WORDS * 4));
    HIP_TRY(c, tmp.alloc((void**)&d_inner, (size_t)nnode * 4));
    HIP_TRY(c, tmp.alloc((void**)&d_pair_of, ((size_t)nnode + 1) * 4));
    HIP_TRY(c, tmp.alloc((void**)&d_sums, ((size_t)nnode / bvhb::SCAN_TILE + 2) * 4));
    HIP_TRY(c, tmp.alloc((void**)&d_depth, (size_t)nnode * 4));
    auto scene_alloc = [&](void** out, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(out, bytes ? bytes : 16);
        if (e == hipSuccess) c->scene_allocs.push_back(*out);
        return e;
    };
    DShade* d_shade;
    DPrim* d_dprims;
    HIP_TRY(c, scene_alloc((void**)&d_shade, (size_t)nfig * sizeof(DShade)));
    HIP_TRY(c, scene_alloc((void**)&d_dprims, (size_t)nprim * sizeof(DPrim)));
    if (nfig) HIP_TRY(c, hipMemcpyAsync(d_figs, figs, (size_t)nfig * sizeof(ptx_shape), hipMemcpyHostToDevice, st));
    if (nmat) HIP_TRY(c, hipMemcpyAsync(d_mats, mats, (size_t)nmat * sizeof(ptx_material), hipMemcpyHostToDevice, st));
    if (nnode) HIP_TRY(c, hipMemcpyAsync(d_nodes, nodes, (size_t)nnode * sizeof(ptx_bvh_node), hipMemcpyHostToDevice, st));
    if (nprim) HIP_TRY(c, hipMemcpyAsync(d_prims, prims, (size_t)nprim * 4, hipMemcpyHostToDevice, st));
    uint32_t res0[R_WORDS] = {0u};
    res0[R_SKY_INDEX] = 0xFFFFFFFFu;
    HIP_TRY(c, hipMemcpyAsync(d_res, res0, sizeof(res0), hipMemcpyHostToDevice, st));
    auto blocks = [](uint32_t n) { return dim3((n + 255u) / 256u); };
    if (nfig) hipLaunchKernelGGL(shade_kernel, blocks(nfig), dim3(256), 0, st, (const ptx_shape*)d_figs, nfig, (const ptx_material*)d_mats, nmat, d_shade, d_res);
    if (nprim) hipLaunchKernelGGL(prim_kernel, blocks(nprim), dim3(256), 0, st, (const ptx_shape*)d_figs, nfig, (const uint32_t*)d_prims, nprim, d_dprims, d_res);
    if (have_tree) hipLaunchKernelGGL(node_kernel, blocks(nnode), dim3(256), 0, st, (const ptx_bvh_node*)d_nodes, nnode, nprim, d_inner, d_res);
    HIP_TRY(c, hipGetLastError());
    uint32_t res[R_WORDS];
    HIP_TRY(c, hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    // the first failing check in the order the host loops make them
    if (res[R_ERRORS] & E_MATERIAL) return fail(c, PTX_ERR_BAD_ARG, "figure material_idx out of range");
    if (res[R_ERRORS] & E_PRIM) return fail(c, PTX_ERR_BAD_ARG, "bvh_prims entry out of range");
    if (res[R_ERRORS] & E_CHILD) return fail(c, PTX_ERR_BAD_ARG, "inner BVH node with child index out of order or range");
    if (res[R_ERRORS] & E_LEAF_BIG) return fail(c, PTX_ERR_UNSUPPORTED, "BVH leaf with more than 7 primitives");
    if (res[R_ERRORS] & E_LEAF_RANGE) return fail(c, PTX_ERR_BAD_ARG, "BVH leaf range outside bvh_prims");
    neg_count = res[R_NEG_COUNT];
    sky_index = res[R_SKY_INDEX];
    type_mask = res[R_TYPE_MASK];
    npair = 0u;
    depth = 0u;
    root_ref = 0u;
    DNodePair* d_pairs = nullptr;
    if (have_tree) {
        { int rc = scan_u32(c, d_inner, nnode, d_pair_of, d_sums); if (rc) return rc; }
        HIP_TRY(c, hipMemcpyAsync(&npair, d_pair_of + nnode, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        HIP_TRY(c, scene_alloc((void**)&d_pairs, (size_t)npair * sizeof(DNodePair)));
        hipLaunchKernelGGL(pair_kernel, blocks(nnode), dim3(256), 0, st, (const ptx_bvh_node*)d_nodes, nnode, (const uint32_t*)d_pair_of, d_pairs);
        hipLaunchKernelGGL(depth_init_kernel, blocks(nnode), dim3(256), 0, st, d_depth, nnode);
        HIP_TRY(c, hipGetLastError());
        root_ref = nodes[0].prim_count > 0u ? (nodes[0].left_or_prim | (nodes[0].prim_count << PT_REF_BITS)) : 0u;  // the root's pair is record 0
        // Longest chain of inner nodes from the root (bounds the stack occupancy): one level per pass, every pass
        // flags its level when it holds an inner node, one read-back at the end.  The traversal stack never holds more
        // than 31 entries, so levels beyond R_LEVELS need not be told apart.
        if (nodes[0].prim_count == 0u) {
            for (uint32_t level = 0u; level < R_LEVELS; level++)
                hipLaunchKernelGGL(depth_pass_kernel, blocks(nnode), dim3(256), 0, st, (const ptx_bvh_node*)d_nodes, nnode, d_depth, level, d_res);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(res, d_res, sizeof(res), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            for (uint32_t level = 0u; level < R_LEVELS; level++)
                if (res[R_LEVEL0 + level]) depth = level + 1u;
        }
    } else {
        HIP_TRY(c, scene_alloc((void**)&d_pairs, 0));
    }
    HIP_TRY(c, hipStreamSynchronize(st));
    sc.prims = d_dprims;
    sc.shade = d_shade;
    sc.pairs = d_pairs;
    return PTX_OK;
}
}  // namespace

extern "C" {

int ptx_create(int device, ptx_ctx** out) {
    if (!out) return PTX_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return PTX_ERR_HIP;
    ptx_ctx* c = new ptx_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&c->stats, 24 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->stats, 0, 24 * sizeof(unsigned long long)) != hipSuccess) {
        delete c;
        return PTX_ERR_HIP;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    c->n_cu = (uint32_t)cus;
    // Every device holds its own copy of the static scratch array at its own address: ask for it with this device current.
    void* scratch_ptr = nullptr;
    if (hipGetSymbolAddress(&scratch_ptr, HIP_SYMBOL(ptx_static_scratch)) != hipSuccess || !scratch_ptr) {
        ptx_destroy(c);
        return PTX_ERR_HIP;
    }
    c->static_scratch = (unsigned char*)scratch_ptr;
    *out = c;
    return PTX_OK;
}

void ptx_destroy(ptx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& pr : c->pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    free_scene(c);
    free_frame(c);
    if (c->comm && c->comm_owned && coll::api()) (void)coll::api()->CommDestroy(c->comm);
    if (c->xchg) (void)hipFree(c->xchg);
    if (c->stats) (void)hipFree(c->stats);
    if (c->wave_state) (void)hipFree(c->wave_state);
    for (auto& f : c->flights) {
        if (f.stream) { (void)hipStreamSynchronize(f.stream); (void)hipStreamDestroy(f.stream); }
        if (f.traced) (void)hipEventDestroy(f.traced);
        if (f.folded) (void)hipEventDestroy(f.folded);
        if (f.wave_state) (void)hipFree(f.wave_state);
        if (f.spill) (void)hipFree(f.spill);
        if (f.pool) (void)hipFree(f.pool);
        if (f.pool_keys) (void)hipFree(f.pool_keys);
        if (f.pool_order) (void)hipFree(f.pool_order);
    }
    if (c->flight_counters) (void)hipFree(c->flight_counters);
    if (c->barrier) (void)hipEventDestroy(c->barrier);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* ptx_last_error(const ptx_ctx* c) { return c ? c->err.c_str() : "null context"; }

static int configure(ptx_ctx* c);

int ptx_set_option(ptx_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return PTX_ERR_BAD_ARG;
    const std::string k(key);
    auto in = [&](int64_t lo, int64_t hi) { return value >= lo && value <= hi; };
    ptx_ctx::Options& o = c->opt;
    bool ok = true;
    if (k == "kernel_family") { if ((ok = in(-1, 1))) o.family = (int)value; }
    else if (k == "upload_path") { if ((ok = in(-1, 1))) o.upload = (int)value; }
    else if (k == "claim_tile") { if ((ok = value == 0 || value == 1 || (value >= 16 && value <= 1024 && (value & (value - 1)) == 0))) o.claim_tile = (uint32_t)value; }
    else if (k == "claim_guided") { if ((ok = value == -1 || value == 0 || in(4, 64))) o.claim_guided = (int)value; }
    else if (k == "claim_ranges") { if ((ok = in(0, PTX_CLAIM_RANGES))) o.claim_ranges = (uint32_t)value; }
    else if (k == "claim_paced") { if ((ok = in(-1, 64))) o.claim_paced = (int)value; }
    else if (k == "claim_cap") { if ((ok = in(-1, 64))) o.claim_cap = (int)value; }
    else if (k == "split_samples") { if ((ok = in(0, 1))) c->split = value != 0; }
    else if (k == "split_small") { if ((ok = in(-1, 0x7FFFFFFF))) o.split_upto = (int64_t)value; }
    else if (k == "split_chunk") { if ((ok = in(0, 0x7FFFFFFF))) o.split_chunk = (uint32_t)value; }
    else if (k == "share_walk") { if ((ok = in(0, 64))) o.shares[0] = (uint32_t)value; }
    else if (k == "share_post") { if ((ok = in(0, 64))) o.shares[1] = (uint32_t)value; }
    else if (k == "share_march") { if ((ok = in(0, 64))) o.shares[2] = (uint32_t)value; }
    else if (k == "share_leaf") { if ((ok = in(0, 64))) o.shares[3] = (uint32_t)value; }
    else if (k == "walk_mode") { if ((ok = in(-1, 2))) o.walk_mode = (int)value; }
    else if (k == "sorted_pool") { if ((ok = in(-1, 1))) o.sorted_pool = (int)value; }
    else if (k == "arith_mode") { if ((ok = in(0, 2))) o.arith_mode = (int)value; }
    else if (k == "host_waits_per_frame") { if ((ok = in(-1, 1))) o.host_waits = (int)value; }
    else if (k == "idle_whole") { if ((ok = in(0, 1))) o.host_waits = value ? -1 : 0; }  // the option's earlier name
    else if (k == "pool_over") { if ((ok = in(-1, 0x7FFFFFFF))) o.pool_over = (int)value; }
    else if (k == "pool_paths") { if ((ok = value == 0 || (value >= 256 && value <= 16384 && (value & (value - 1)) == 0))) o.pool_paths = (uint32_t)value; }
    else if (k == "pool_guided") { if ((ok = in(-1, 64))) o.pool_guided = (int)value; }
    else if (k == "pool_spread") { if ((ok = in(-1, 1))) o.pool_spread = (int)value; }
    else if (k == "pool_finish") { if ((ok = in(-1, 16384))) o.pool_finish = (int)value; }
    else if (k == "grid_per_cu") { if ((ok = in(0, 64))) o.grid_per_cu = (uint32_t)value; }
    else if (k == "flights") { if ((ok = in(0, PTX_FLIGHTS))) o.flights = (uint32_t)value; }
    else if (k == "verbose") { o.verbose = value != 0; }
    else if (k == "bvh_timing") { o.bvh_timing = value != 0; }
    else return fail(c, PTX_ERR_BAD_ARG, "ptx_set_option: unknown key '" + k + "'");
    if (!ok) return fail(c, PTX_ERR_BAD_ARG, "ptx_set_option: value " + std::to_string((long long)value) + " out of range for '" + k + "'");
    if (c->have_scene) {
        HIP_TRY(c, hipSetDevice(c->device));
        return configure(c);
    }
    return PTX_OK;
}

// Which kernel family runs a scene.  The wave-scheduled kernels win wherever lanes need very unequal numbers of
// steps: trees read from global memory (meshes) and the long ray marchers (Moebius, Mandelbulb, Julia); small
// closed-form scenes (the Cornell box, torus included) run faster in lockstep.  The kernel_family option
// overrides the rule (tests, measurement runs).
// type_mask: 1 << type over the BVH primitives that can be hit.
static bool pick_sched(const DScene& sc, uint32_t type_mask, int forced) {
    if (forced >= 0) return forced == 1;
    if (sc.lds_scene_bytes == 0u || sc.has_fractals != 0u) return true;
    return (type_mask & (1u << FIG_MEBIUS)) != 0u;
}

static const char* trace_kernel_label(TraceKernel k) {
#define PTX_NAME(f) if (k == f) return #f;
    PTX_NAME(ptx_trace_kernel) PTX_NAME(ptx_trace_kernel_lds) PTX_NAME(ptx_trace_kernel_lds_ranged) PTX_NAME(ptx_trace_kernel_lds_sorted) PTX_NAME(ptx_trace_kernel_sorted) PTX_NAME(ptx_trace_kernel_frac)
    PTX_NAME(ptx_trace_kernel_lds_frac) PTX_NAME(ptx_sched_kernel) PTX_NAME(ptx_sched_kernel_lds) PTX_NAME(ptx_sched_kernel_samples)
    PTX_NAME(ptx_sched_kernel_lds_samples) PTX_NAME(ptx_sched_kernel_frac) PTX_NAME(ptx_sched_kernel_lds_frac)
#undef PTX_NAME
    return "";
}
static TraceKernel trace_kernel_of(const ptx_ctx* c) {
    const bool frac = c->sc.has_fractals != 0u, lds = c->sc.lds_scene_bytes > 0u;
    return c->sched ? (lds ? (frac ? ptx_sched_kernel_lds_frac : ptx_sched_kernel_lds) : (frac ? ptx_sched_kernel_frac : ptx_sched_kernel))
                    : (lds ? (frac ? ptx_trace_kernel_lds_frac : ptx_trace_kernel_lds) : (frac ? ptx_trace_kernel_frac : ptx_trace_kernel));
}
// Dynamic LDS of a trace workgroup for the uploaded scene: stack entries + per-lane fields (+ the staged scene, + the
// estimator states of the scheduled fractal kernels).
static size_t trace_lds_bytes(const ptx_ctx* c) {
    const size_t lds = (size_t)PTX_WAVES_PER_BLOCK * 64u * (c->sc.lds_stack_entries * sizeof(StackEntry) + (PT_PIXEL_FIELDS + PT_PARK_FIELDS) * 4u);
    return lds + c->sc.lds_scene_bytes + (c->sched && c->sc.has_fractals != 0u ? (size_t)PTX_WAVES_PER_BLOCK * 64u * SC_FRAC_FIELDS * 4u : 0u);
}


// The form a launch of n samples takes on the uploaded scene under the current options.  samples_allowed false: the
// whole-pixel form (what ptx_trace_n_static, radiance capture and a host that waits for every frame run), which does not
// depend on n.  Every choice below was measured; frames are identical under all of them (tested).
// Which launch lengths trace from sorted path pools when the pool_over option leaves it to the library.
//  * Scenes staged in LDS that run the lockstep kernels (the Cornell box and most of the reference's small scenes): every launch.
//    Short launches overlap (a pool set per flight) and beat lane-owned paths back to back -- 1080p, ms per launch of 1 / 2 / 4 samples:
//    Cornell box 0.99 / 1.80 / 3.42 -> 0.85 / 1.53 / 2.89 (10.4 Gpath-bounce/s at the reference's own one-sample dispatch), demo.yaml
//    0.4 / 0.7 / 1.0 -> 0.3 / 0.4 / 0.6 (profiles/r03_short_pools.txt).  With one pool set per context pools lose here: short launches
//    then run in turn, and pools of 4096 paths leave half the workgroups of a one-sample launch without a job.
//  * ... unless several ray-marched figures make pixels differ a hundredfold in cost (rings.yaml: sixteen tori): a pool reserves
//    thousands of jobs at a time, and at one or two samples per launch a workgroup whose jobs lie on the figures ends long after
//    the rest (0.78 / 1.46 ms lane-owned against 0.91 / 1.47 from pools); from three samples up pools win there too (2.69 -> 2.12 at four).
//  * A lit mesh's paths are long enough for pools to win at every length (unit_logo.yaml, one / four samples per launch:
//    430 -> 610 / 471 -> 750 Mpaths/s).  Other scenes, when the options ask for pools at all: launches of more than four samples.
static bool pools_by_default(const ptx_ctx* c, uint32_t n, bool big) {
    const bool lit_mesh = c->plain_mesh && c->sc.nlight > 0u;
    if (c->opt.pool_over >= 0) return n > (uint32_t)c->opt.pool_over;
    if (lit_mesh) return true;
    if (c->sc.lds_scene_bytes > 0u && c->sc.has_fractals == 0u) {
        // Small scenes with a Moebius strip (wave-scheduled by the per-scene rule): every length too, now that short launches overlap
        // -- 1080p, ms per launch of 1 / 2 / 4 samples back to back, wave-scheduled -> pools: all.yaml 1.83 / 3.30 / 6.26 -> 1.78 / 2.75 / 5.27,
        // mebius.yaml 1.21 / 2.24 / 4.28 -> 1.16 / 1.89 / 3.56.
        if (c->sched) return true;
        if (c->n_marched >= 2u && n <= 2u) return false;
        // ... and launches of three and four samples keep lane-owned paths when the scene is walls, a light and a figure or two
        // (fewer than four primitives in the tree): there is little for a sort to bring together, and back to back the pools'
        // rounds cost 6-12 % (cube.yaml 1.27 -> 1.39 ms at four samples per launch; the Cornell box 3.43 -> 2.89, demo.yaml
        // 1.03 -> 0.65, and the scenes without lights, whose paths end after a bounce or two, 0.52 -> 0.42, gain).
        if (n == 3u || n == 4u) return c->sc.nprim >= 4u || c->sc.nlight == 0u;
        return true;
    }
    // Shallow meshes without lights (a bounce or two into the sky): launches of one and two samples, whose time under the
    // wave-scheduled kernels is the hand-out of jobs and the launch's last waves -- 1080p, ms per launch of 1 / 2 samples back to
    // back: teapot.yaml 0.45 / 0.66 -> 0.36 / 0.59, iron_man.yaml 0.43 / 0.51 -> 0.26 / 0.37, car.yaml 0.37 / 0.53 -> 0.35 / 0.52.
    // At four samples pools win or lose by a third with the scene (teapot 1.13 -> 0.87, iron_man 0.65 -> 0.90), from five up they lose,
    // and so do launches of more than PTX_MESH_POOL_JOBS jobs whatever their length: past the launch's start and end the
    // wave-scheduled kernels trace a mesh faster (4K, one sample: teapot 1.20 -> 1.23 ms, iron_man 0.64 -> 0.76, car 0.88 -> 1.04).
    if (c->plain_mesh) return (n <= 2u && !big) || n > 4u;
    return n > 4u;
}
// ... and on meshes without lights only those short launches do without being asked (options sorted_pool 1 + kernel_family 0 ask).
static constexpr uint32_t PTX_MESH_POOL_JOBS = 4608u << 10;  // two samples of a 1080p frame, and a little
static bool short_mesh_pools(const ptx_ctx* c, uint32_t n, bool big) {
    return !big && c->plain_mesh && c->sc.nlight == 0u && c->opt.pool_over < 0 && n <= 2u;
}

// big: the launch has more jobs than the form for its length allows (LaunchForm::upto_jobs).
static LaunchForm form_for(const ptx_ctx* c, uint32_t n, bool samples_allowed, bool big = false) {
    const DScene& sc = c->sc;
    const bool frac = sc.has_fractals != 0u, lds = sc.lds_scene_bytes > 0u;
    LaunchForm f;
    // Small scenes with a Moebius strip (staged in LDS, wave-scheduled by the per-scene rule): launches of more than four samples
    // run the lockstep kernel over sorted path pools instead (all.yaml 25.1 -> 21.8 ms per 16 samples at 1080p, mebius.yaml
    // 17.3 -> 14.6; lane-owned lockstep 27.5 / 15.8), shorter ones stay wave-scheduled.  Same LDS bytes, same occupancy.
    // Shallow meshes with lights (unit_logo.yaml: six bounces and five occlusion queries per path) likewise: 72 -> 44 ms per 16
    // samples at 1080p; without lights (a bounce or two into the sky) the wave-scheduled kernels are up to twice as fast as pools.
    const bool lit_mesh = c->plain_mesh && sc.nlight > 0u;
    const bool by_default = pools_by_default(c, n, big), short_mesh = short_mesh_pools(c, n, big);
    // Sample jobs + fold against whole-pixel jobs: configure().  (Sample jobs need a claiming grid: below.)
    const bool can_split = samples_allowed && c->split && n <= c->split_upto;
    const bool pools_instead = c->opt.family < 0 && c->opt.sorted_pool != 0 && (lds || lit_mesh || short_mesh) && !frac && by_default && can_split;
    f.sched = c->sched && !pools_instead;
    // The claiming kernels run a grid sized to the GPU: exactly the workgroups that are resident at once (a
    // workgroup that had to wait for another to finish would start its pre-assigned first jobs at the very end),
    // whose lanes claim pixels from a counter that starts behind the grid's own first jobs.  The pools' grid is the resident
    // workgroups too -- but only a launch that does trace from pools may be given that grid: the lockstep kernel for scenes in
    // global memory does not claim, it runs one workgroup per 16x16 block.
    const bool pools_global = c->opt.sorted_pool == 1 || (c->opt.sorted_pool != 0 && (lit_mesh || short_mesh) && !f.sched);
    f.claim = f.sched || (PTX_CLAIM_LDS && lds) || (pools_global && !frac && by_default && can_split);
    const bool split = f.claim && can_split;
    // Long launches of sample jobs on the lockstep kernels trace from sorted path pools (sorted_body); short ones, whose
    // time is the launch's first and last paths, keep the lane-owned form, which overlaps with its neighbours.
    const bool sorted = split && !f.sched && !frac && by_default && (lds ? c->opt.sorted_pool != 0 : pools_global);
    f.jobs = sorted ? JOBS_POOLS : (split ? JOBS_SAMPLES : JOBS_PIXELS);
    f.kernel = f.sched == c->sched ? trace_kernel_of(c) : (lds ? ptx_trace_kernel_lds : ptx_trace_kernel);
    if (split && f.sched && !frac) f.kernel = lds ? ptx_sched_kernel_lds_samples : ptx_sched_kernel_samples;  // same registers, LDS and occupancy
    if (sorted) {
        f.kernel = lds ? ptx_trace_kernel_lds_sorted : ptx_trace_kernel_sorted;  // same LDS, four waves per SIMD
        // measured on the Cornell box, Gpath-bounce/s at 16 / 128 samples per launch: 1024 paths 11.6 / 12.3, 2048 12.2 / 13.2,
        // 4096 12.5 / 13.6, 8192 12.0 / 13.8, 16384 11.0 / 13.9 (a longer launch amortises a larger pool's first and last rounds)
        // (one-sample launches of a mesh in global memory, 1024 against 2048: teapot.yaml 0.41 -> 0.36 ms, iron_man.yaml 0.29 -> 0.26)
        const bool unlit_mesh = !lds && !lit_mesh;  // paths of a bounce or two: a launch is over before a large pool has been filled twice
        f.pool_m = c->opt.pool_paths ? c->opt.pool_paths : (n >= 64u ? 8192u : (n > 4u ? 4096u : (n == 1u && unlit_mesh ? 1024u : 2048u)));
        // A pool's reservation samples the whole frame at a stride (job_pixel) in one-sample launches (Cornell box 0.857 -> 0.847 ms,
        // rings.yaml 1.48 -> 0.91) and in two-sample launches of a mesh in global memory (iron_man.yaml 0.81 -> 0.37, teapot.yaml 0.62 -> 0.59);
        // longer launches lose the locality of neighbouring pixels' staged samples (rings.yaml 128 samples 51.8 -> 55.4 ms).
        f.upto_jobs = short_mesh && c->opt.sorted_pool != 1 ? PTX_MESH_POOL_JOBS : 0u;
        // A lit mesh's long paths want neither (unit_logo.yaml, one / two samples: 3.05 / 5.59 ms with both, 2.77 / 5.38 with pools of 2048 in block order).
        f.spread = c->opt.pool_spread >= 0 ? c->opt.pool_spread != 0 : (lit_mesh ? false : (n == 1u || (n == 2u && unlit_mesh)));
    }
    // Deep trees: a wave's reservations start large and shrink with the jobs left, so its lanes stay on neighbouring pixels for
    // longer while the launch still ends evenly (all.yaml + 10 M triangles, 8 samples per launch: 36.6 -> 36.2 ms; shallow meshes,
    // whose pixels differ a hundredfold in cost, lose up to half with reservations of thousands of jobs).
    f.guided_k = !f.sched || frac ? 0u : (c->opt.claim_guided >= 0 ? (uint32_t)c->opt.claim_guided : (sc.stack_entries >= PT_DEEP_TREE ? 8u : 0u));
    // The lockstep kernels with the scene in LDS, lane-owned paths (launches of up to four samples): half of the reference's
    // scenes finish a path in less time than its share of a 64-job reservation takes at the counter (4.5 G paths/s whatever the
    // scene).  There a wave paces itself: reservations used up in under 2/3 of 4 * waves * 10 ns double (up to 512 jobs), never
    // beyond 1/(8 * waves) of the jobs left.  1080p, G paths/s at 4 samples per launch, in turn / back to back: test.yaml 4.1 / 6.2
    // -> 5.4 / 8.1, aelita.yaml 4.1 / 6.4 -> 4.8 / 7.4, earth.yaml 4.0 / 5.6 -> 4.3 / 5.9, Cornell box and rings.yaml unchanged.
    // A looser cap (1/(2 * waves)) reaches 9.2 / 16.3 on test.yaml and halves rings.yaml, whose pixels differ most in cost.
    f.paced_m = f.sched || frac ? 0u : (c->opt.claim_paced >= 0 ? (uint32_t)c->opt.claim_paced : (lds ? 4u : 0u));
    // One counter hands out jobs in frame order: all waves of the GPU work on the same few rows at a time, which a deep tree in
    // memory wants (every wave walks the same part of it).  Scenes whose paths cost little want reservations served faster than one
    // address can (88 M a second): sixteen counters, each handing out a sixteenth of the jobs, a wave moving on when its counter runs out.
    // Only the lockstep kernel has the registers for it, in a variant of its own (CLAIM_RANGED, sched_dev.h: its counters count
    // from zero, the others' one counter from the first unassigned job), and only scenes without lights run it: their paths end
    // at the sky after a bounce or two, and the hand-out is what bounds them (test.yaml, planet.yaml: 1.5 to 2.5 times the rate at
    // every frame size, in turn and back to back).  Scenes with lights (three bounces per path and more) gain up to a quarter at
    // 1080p in turn and lose 5-12 % at 4K and back to back: the variant's code is a few per cent slower than the plain kernel's.
    const uint32_t counters_wanted = f.sched || frac || !lds ? 1u : (c->opt.claim_ranges ? c->opt.claim_ranges : (sc.nlight == 0u ? (uint32_t)PTX_CLAIM_RANGES : 1u));
    if (f.kernel == ptx_trace_kernel_lds && counters_wanted > 1u) {
        f.kernel = ptx_trace_kernel_lds_ranged;  // same LDS, same occupancy
        f.counters = counters_wanted;
    }
    // Overlap hides a launch's last paths behind the next launch: a quarter of a one-sample launch of the Cornell box, but
    // 0.4 % of a 128-sample one (0.7 % of 8 samples of the 10 M-triangle scene); fractal launches end on tens of milliseconds
    // of a few lanes marching whatever their length (1.9 % of 128 samples).  Longer launches of the other scenes run in turn,
    // which also keeps every kernel's duration a quantity of its own.  One set of pools per context: pools run in turn.
    // Pools of short launches: a pool set per flight, four in flight (1080p Cornell box, one sample per launch, ms with 1 / 2 / 4 / 6
    // in flight: 1.12 / 0.93 / 0.86 / 0.90 with 2048 paths per pool).
    f.flights = c->opt.flights >= 1u ? c->opt.flights : (!frac && n > 4u ? 1u : (sorted ? 4u : 0u));
    return f;
}

// The table: one row per class of launch length.  The thresholds are those form_for() compares n with.
static void build_policy(ptx_ctx* c) {
    const uint32_t pool_over = c->opt.pool_over >= 0 ? (uint32_t)c->opt.pool_over : 0u;
    std::vector<uint32_t> upto = {1u, 2u, 3u, 4u, 16u, 63u, pool_over, c->split_upto, 0xFFFFFFFFu};  // the lengths form_for() compares n with
    std::sort(upto.begin(), upto.end());
    upto.erase(std::unique(upto.begin(), upto.end()), upto.end());
    c->policy.clear();
    for (uint32_t u : upto) {
        if (u == 0u) continue;
        PolicyRow r;
        r.upto = u;
        r.form = form_for(c, u, true);
        r.big = r.form.upto_jobs ? form_for(c, u, true, true) : r.form;
        c->policy.push_back(r);
    }
    c->pixels_form = form_for(c, 1u, false);
}

// Jobs a wave reserves per device atomic (claim_jobs): 1 = just what its lanes need at that moment.  m: samples of the
// launch (JOBS_PIXELS) or of its chunk.
static uint32_t reservation_jobs(const ptx_ctx* c, const LaunchForm& f, uint32_t m, uint32_t n_blocks) {
    if (c->opt.claim_tile) return c->opt.claim_tile;
    const bool frac = c->sc.has_fractals != 0u;
    if (f.jobs == JOBS_PIXELS) {
        uint32_t t = c->claim_tile;
        // Lockstep kernels: short launches are bound by the rate of the claims (one device atomic per finished pixel and wave),
        // long ones lose 4 % when a wave's lanes sit on neighbouring pixels (Cornell box, 1080p: 1.3 ms against 2.0 at one sample
        // per launch, 15.6 against 17.1 at 16, 30.1 against 29.1 at 32).
        if (!f.sched && !frac && m <= 16u) t = 64u;
        // Scheduled kernels on frames of 4 M pixels and more, one or two samples per launch: even one claim per 64 pixels is
        // 130 k atomics on one address (teapot 4K, one sample per launch: 2.1 ms, 1.7 ms with 128 pixels per claim; longer
        // launches and smaller frames are 3-6 % faster with 64).
        if (f.sched && !frac && m <= 2u && n_blocks >= 16384u) t = 128u;
        return t;
    }
    uint32_t t = !frac ? (n_blocks >= 16384u ? 128u : 64u) : (m >= 64u ? 256u : 128u);
    // Shallow meshes finish 8 G paths a second and more: 64 jobs per reservation is more reservations than the counter serves
    // (one device atomic on one address: about 10 ns), while a wave that reserves 512 jobs at the end of a short launch keeps
    // the launch waiting.  About 49 k reservations per launch, 64 to 512 jobs each -- measured at 1080p, ms per launch of
    // 1 / 2 / 4 / 8 / 16 / 64 samples with 64 jobs: 0.9 / 1.3 / 2.3 / 3.6 / 7.0 / 27.6, 128: 0.9 / 1.0 / 1.8 / 2.7 / 5.2 / 19.5,
    // 256: 1.2 / 1.3 / 1.8 / 2.3 / 4.3 / 16.1, 512: 1.6 / 1.9 / 2.3 / 2.7 / 4.0 / 14.8 (teapot.yaml; iron_man.yaml and car.yaml alike).
    if (c->plain_mesh && f.sched) {
        const uint32_t per = (n_blocks * 256u * m / 49152u + 63u) & ~63u;
        t = per < 64u ? 64u : (per > 512u ? 512u : per);
    }
    return t;
}

// Everything launch_trace decides before it touches a buffer.
struct LaunchChoice {
    LaunchForm f;
    bool alone = false;            // a one-sample launch of a host that waits for every frame
    uint32_t chunk = 0, depth = 1;  // samples per kernel launch; launches of sample jobs in flight
    uint32_t tile0 = 0;            // jobs per reservation (of the first chunk)
    uint32_t lds_stack = 0;        // traversal-stack entries per lane in LDS
    uint32_t grid = 0;             // workgroups
};
// fixed: ptx_trace_n_static.  peek: ptx_launch_plan -- the context's record of the call sequence is read, not updated.
static LaunchChoice resolve_launch(ptx_ctx* c, uint32_t n, bool fixed, uint32_t n_blocks, bool peek) {
    LaunchChoice L;
    const bool frac = c->sc.has_fractals != 0u;
    const bool lit_mesh = c->plain_mesh && c->sc.nlight > 0u;
    // One sample per launch with the GPU idle -- a host that reads every frame back before it asks for the next, as the
    // reference's own frame loop presents every frame -- has no neighbour to overlap with: there the pixel folds its sample in
    // the lane and no fold kernel follows (1080p, in turn: Cornell box 1.42 -> 1.24 ms, cube.yaml 0.67 -> 0.59, test.yaml
    // 0.43 -> 0.31, teapot 0.85 -> 0.76; deep trees and the Moebius scenes lose 1-4 % and keep sample jobs).  Which host this
    // is follows from the calls it makes, not from a clock: a launch is alone when the host has synchronised with the
    // context (ptx_sync, a ptx_read_*, ptx_kernel_time*) since the launch before it, and did so before that one as well --
    // the first launch after a read-back may be the head of a back-to-back burst, whose second launch should start beside
    // it.  The host_waits_per_frame option states it outright: 1 every one-sample launch is alone, 0 none is.
    const bool eligible = n == 1u && !fixed && !frac && (!c->sched || (c->plain_mesh && !lit_mesh));
    if (eligible && c->opt.host_waits != 0) {
        const bool idle = c->drained;
        L.alone = c->opt.host_waits == 1 || (idle && c->found_idle);
        if (!peek) c->found_idle = idle;
    } else if (!peek) {
        c->found_idle = false;
    }
    if (!peek) c->drained = false;
    const PolicyRow* row = &c->policy.back();
    for (const PolicyRow& r : c->policy)
        if (n <= r.upto) { row = &r; break; }
    // whole-pixel jobs: the static entry point (no staging buffer), radiance capture (its sample index is launch-relative), a launch alone
    const bool big = row->form.upto_jobs && (uint64_t)n_blocks * 256u * n > row->form.upto_jobs;
    L.f = fixed || c->dbg || L.alone ? c->pixels_form : (big ? row->big : row->form);
    if (fixed && L.f.kernel == ptx_trace_kernel_lds_ranged) { L.f.kernel = ptx_trace_kernel_lds; L.f.counters = 1u; }  // its scratch array holds one counter
    L.chunk = n;
    L.depth = 1u;
    L.lds_stack = c->sc.lds_stack_entries;
    if (L.f.jobs != JOBS_PIXELS) {
        // Sample jobs are staged in chunks that bound the staging buffer (4 GiB of the 288: 128 samples of a 1080p frame) and the
        // job count; every chunk traces on a stream of its own (a "flight") into a staging buffer of its own, beside the tails of
        // the chunks and calls before it, and only the folds follow one another, on the context's stream.
        const size_t frame_pixels = (size_t)c->w * c->h;
        const size_t by_mem = ((size_t)4 << 30) / (frame_pixels * sizeof(float4)), by_jobs = 0x7FFFFFFFu / ((size_t)n_blocks * 256u);
        size_t m = by_mem < by_jobs ? by_mem : by_jobs;
        if (c->opt.split_chunk >= 1u && c->opt.split_chunk < m) m = c->opt.split_chunk;
        if (m < 1) m = 1;
        if (m < L.chunk) L.chunk = (uint32_t)m;
        const size_t in_flight = ((size_t)8 << 30) / (L.chunk * frame_pixels * sizeof(float4));
        const uint32_t by_staging = in_flight < 2 ? 2u : (uint32_t)in_flight;
        // measured, fractals 1080p, one sample per launch, ms per frame with 1 / 2 / 4 / 6 / 8 in flight: 58 / 28.5 / 29 / 20.6 / 21.9
        // (HIP's default of four hardware queues per process; with GPU_MAX_HW_QUEUES=8 in the host's environment: 58 / 27.5 / 17.4 / 14.1 / 16.8)
        if (L.f.flights == 0u) L.depth = in_flight >= PTX_FLIGHTS_DEFAULT ? PTX_FLIGHTS_DEFAULT : by_staging;
        else L.depth = L.f.flights < in_flight || L.f.flights <= 2u ? L.f.flights : by_staging;
        // the slots of the running mean, which sample jobs of the wave-scheduled kernels do not keep, hold two more stack entries (same LDS bytes: sched_body)
        if (L.f.sched && !frac && c->sc.lds_scene_bytes == 0u && c->sc.stack_entries > PT_LDS_STACK) {
            const uint32_t more = PT_LDS_STACK + SC_SAMPLE_JOB_FREE / 2u;
            L.lds_stack = c->sc.stack_entries < more ? c->sc.stack_entries : more;
        }
    }
    L.tile0 = L.f.claim ? reservation_jobs(c, L.f, n < L.chunk ? n : L.chunk, n_blocks) : 0u;
    // The claiming kernels run a grid sized to the GPU: exactly the workgroups that are resident at once.
    L.grid = n_blocks;
    if (L.f.claim && c->occ_per_cu >= 1) {
        uint32_t per_cu = (uint32_t)c->occ_per_cu;  // configure(): queried when the scene was uploaded
        if (c->opt.grid_per_cu >= 1u && c->opt.grid_per_cu < per_cu) per_cu = c->opt.grid_per_cu;
        const uint32_t resident = c->n_cu * per_cu;
        L.grid = n_blocks < resident ? n_blocks : resident;
        if (fixed && L.grid > PTX_STATIC_MAX_WAVES / PTX_WAVES_PER_BLOCK) L.grid = PTX_STATIC_MAX_WAVES / PTX_WAVES_PER_BLOCK;  // fewer workgroups claim the same jobs
    }
    // A pool is never larger than its workgroup's share of the launch's jobs: a pool that holds twice that leaves half the
    // workgroups without a job (a one-sample launch is 2025 jobs per workgroup at 1080p, 256 at 512 x 512).
    if (L.f.jobs == JOBS_POOLS && !c->opt.pool_paths && L.grid > 0u) {
        const uint64_t share = ((uint64_t)n_blocks * 256u * (n < L.chunk ? n : L.chunk) / L.grid + 63u) & ~63ull;
        const uint32_t least = 256u;
        const uint32_t fit = share < least ? least : (share > L.f.pool_m ? L.f.pool_m : (uint32_t)share);
        L.f.pool_m = fit;
    }
    return L;
}
static std::string describe_launch(const LaunchChoice& L) {
    char buf[256];
    const LaunchForm& f = L.f;
    snprintf(buf, sizeof(buf), "%s jobs=%s claim_tile=%u counters=%u guided=%u paced=%u pool=%u spread=%u flights=%u chunk=%u lds_stack=%u",
             trace_kernel_label(f.kernel), f.jobs == JOBS_POOLS ? "pools" : (f.jobs == JOBS_SAMPLES ? "samples" : "pixels"), L.tile0,
             f.claim ? f.counters : 0u, f.claim && L.tile0 > 1u ? f.guided_k : 0u, f.claim && L.tile0 > 1u ? f.paced_m : 0u, f.pool_m,
             f.jobs == JOBS_POOLS && f.spread ? 1u : 0u, f.jobs != JOBS_PIXELS ? L.depth : 0u, L.chunk, L.lds_stack);
    return buf;
}

// Per-scene choices (kernel family, scheduling shares, claim size, the launch policy's table, resident workgroups per compute
// unit) from the uploaded scene's class and the options; run at upload and whenever an option changes.
static int configure(ptx_ctx* c) {
    if (!c->have_scene) return PTX_OK;
    const DScene& sc = c->sc;
    const uint32_t type_mask = c->type_mask;
    const uint32_t fractal_bits = (1u << FIG_MANDELBULB) | (1u << FIG_JULIA);
    c->sched = pick_sched(sc, type_mask, c->opt.family);
    const uint32_t mesh[3] = {8u, 2u, 4u}, mixed[3] = {8u, 2u, 4u}, fractal[3] = {64u, 8u, 4u}, small[3] = {1u, 2u, 8u};
    // deep trees (one node visit per step, lanes at leaf primitives served as a state of their own): walk, post, march, leaf
    const uint32_t deep_mesh[4] = {12u, 2u, 4u, 4u}, deep_mixed[4] = {4u, 2u, 16u, 6u};
    const bool marchers = (type_mask & ((1u << FIG_TORUS) | (1u << FIG_MEBIUS) | fractal_bits)) != 0u;
    const uint32_t* pick = sc.lds_scene_bytes == 0u ? (marchers ? mixed : mesh) : (sc.has_fractals ? fractal : small);
    c->share_leaf = 4u;
    if (sc.stack_entries >= PT_DEEP_TREE) {
        pick = marchers ? deep_mixed : deep_mesh;
        c->share_leaf = pick[3];
    }
    memcpy(c->shares, pick, sizeof(c->shares));
    for (int i = 0; i < 3; i++)
        if (c->opt.shares[i]) c->shares[i] = c->opt.shares[i];
    if (c->opt.shares[3]) c->share_leaf = c->opt.shares[3];
    // Sample jobs + fold with overlapping launches against whole-pixel jobs (launch_trace), ms per sample at 1080p unless said:
    // Cornell box 0.98 / 0.80 against 1.23 / 0.87 (1 / 128 samples per launch); all.yaml 2.04 / 1.75 against 2.67 / 2.47 (1 / 16);
    // all.yaml + 10 M triangles 5.6 / 4.7 against 6.7 / 5.7 (1 / 16); fractals: see there.  A shallow tree without marched
    // figures traces a sample in about the time one reservation of 64 jobs takes at the launch-wide counter: there the
    // reservations grow with the launch (launch_trace), and with them sample jobs win as well (teapot.yaml 4K, ms per launch
    // of 4 / 16 samples: 4.4 / 15.2 against 5.0 / 16.9; iron_man.yaml 1080p, 16 / 64 samples: 3.6 / 6.0 against 8.8 / 34.3).
    c->plain_mesh = c->sched && !sc.has_fractals && !marchers && sc.stack_entries < PT_DEEP_TREE;
    c->split_upto = 0x7FFFFFFFu;
    if (c->opt.split_upto >= 0 && !(c->sched && sc.has_fractals)) c->split_upto = (uint32_t)c->opt.split_upto;
    c->claim_tile = c->sched && !sc.has_fractals ? 64u : 1u;  // measured: see claim_pixel
    if (c->opt.claim_tile) c->claim_tile = c->opt.claim_tile;
    build_policy(c);
    // Resident workgroups per compute unit of the kernel this scene runs: the claiming kernels' grid.
    const TraceKernel kernel = trace_kernel_of(c);
    const size_t lds_all = trace_lds_bytes(c);
    if (c->occ_kernel != (const void*)kernel || c->occ_lds != lds_all) {
        HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->occ_per_cu, kernel, (int)PTX_BLOCK, lds_all));
        c->occ_kernel = (const void*)kernel;
        c->occ_lds = lds_all;
        if (c->opt.verbose) fprintf(stderr, "ptx355: %s, %d workgroups per compute unit, %zu B LDS each\n", ptx_trace_kernel_name(c), c->occ_per_cu, lds_all);
    }
    return PTX_OK;
}


int ptx_upload_scene(ptx_ctx* c, const ptx_shape* figs, uint32_t nfig, const ptx_material* mats, uint32_t nmat,
                     const ptx_bvh_node* nodes, uint32_t nnode, const uint32_t* prims, uint32_t nprim,
                     const uint32_t* lights, uint32_t nlight, const uint32_t* infinite, uint32_t ninf,
                     const uint32_t* tex_pixels, uint32_t npix, const ptx_texture_info* tex_infos, uint32_t ntex) {
    if (c) c->order_dirty = true;
    if (!c) return PTX_ERR_BAD_ARG;
    if ((nfig && !figs) || (nmat && !mats) || (nnode && !nodes) || (nprim && !prims) || (nlight && !lights) ||
        (ninf && !infinite) || (npix && !tex_pixels) || (ntex && !tex_infos))
        return fail(c, PTX_ERR_BAD_ARG, "null array with non-zero count");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    if (nprim > PT_REF_MASK || nnode > PT_REF_MASK) return fail(c, PTX_ERR_UNSUPPORTED, "more than 2^29 nodes or primitives");
    // Where the re-layout runs: large scenes on the device (relayout.h), small ones on the host, whose loops cost
    // less than the launches.  Both give the same records.  The upload_path option overrides (tests, measurements).
    bool on_device = nfig >= 65536u || nprim >= 65536u;
    if (c->opt.upload >= 0) on_device = c->opt.upload == 1;

    std::vector<DNeg> negs;
    int32_t sky_tex = -1;
    float sky_color[3] = {0, 0, 0};
    auto add_neg = [&](const ptx_shape& f) {
        DNeg n;
        memset(&n, 0, sizeof(n));
        n.type = f.shape_type; n.radius = f.radius; n.radius2 = f.radius2; n.height = f.height;
        memcpy(n.pos, f.position, 12);
        memcpy(n.normal, f.normal, 12);
        negs.push_back(n);
    };
    DScene sc;
    memset(&sc, 0, sizeof(sc));
    int rc;
    uint32_t root_ref = 0, depth = 0, eff_nprim = 0, npair = 0, type_mask = 0;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    const bool have_tree = nnode > 0 && nprim > 0;
    if (!on_device) {
        for (uint32_t i = 0; i < nfig; i++)
            if (figs[i].material_idx >= nmat) return fail(c, PTX_ERR_BAD_ARG, "figure material_idx out of range");
        std::vector<DShade> shade(nfig);
        for (uint32_t i = 0; i < nfig; i++) {
            const ptx_shape& f = figs[i];
            const ptx_material& m = mats[f.material_idx];
            shade[i].mat = m;
            shade[i].texture_scale = f.texture_scale;
            shade[i].uv0 = f._pad2; shade[i].uv1 = f._pad3; shade[i].uv2 = f._pad4;
            if (f.csg_op == 1u) add_neg(f);
            if (sky_tex < 0 && f.shape_type == FIG_SKYBOX && m.texture_id >= 0) {
                sky_tex = m.texture_id;
                memcpy(sky_color, m.base_color, 12);
            }
        }
        std::vector<DPrim> dprims(nprim);
        for (uint32_t s = 0; s < nprim; s++) {
            if (prims[s] >= nfig) return fail(c, PTX_ERR_BAD_ARG, "bvh_prims entry out of range");
            dprims[s] = make_prim(figs[prims[s]], prims[s]);
            if (dprims[s].type < 32u) type_mask |= 1u << dprims[s].type;
        }
        // Node pairs.  Children must have larger indices than their parent (true of the reference's
        // pre-order flatten, bvh.rs:188-213), which also rules out cycles.
        std::vector<DNodePair> pairs;
        if (have_tree) {
            std::vector<uint32_t> pair_of(nnode, 0xFFFFFFFFu);
            uint32_t np = 0;
            for (uint32_t i = 0; i < nnode; i++) {
                const ptx_bvh_node& nd = nodes[i];
                if (nd.prim_count == 0u) {
                    if (i + 1 >= nnode || nd.left_or_prim >= nnode || nd.left_or_prim <= i)
                        return fail(c, PTX_ERR_BAD_ARG, "inner BVH node with child index out of order or range");
                    pair_of[i] = np++;
                } else {
                    if (nd.prim_count > 7u) return fail(c, PTX_ERR_UNSUPPORTED, "BVH leaf with more than 7 primitives");
                    if ((uint64_t)nd.left_or_prim + nd.prim_count > nprim) return fail(c, PTX_ERR_BAD_ARG, "BVH leaf range outside bvh_prims");
                }
            }
            pairs.resize(np);
            auto child = [&](uint32_t ci, float* mn, float* mx, uint32_t* ref, uint32_t* cnt) {
                const ptx_bvh_node& ch = nodes[ci];
                memcpy(mn, ch.aabb_min, 12);
                memcpy(mx, ch.aabb_max, 12);
                if (ch.prim_count > 0u) { *ref = ch.left_or_prim; *cnt = ch.prim_count; }
                else { *ref = pair_of[ci]; *cnt = 0u; }
            };
            for (uint32_t i = 0; i < nnode; i++) {
                if (nodes[i].prim_count != 0u) continue;
                DNodePair& pr = pairs[pair_of[i]];
                child(i + 1, pr.lmin, pr.lmax, &pr.lref, &pr.lcnt);
                child(nodes[i].left_or_prim, pr.rmin, pr.rmax, &pr.rref, &pr.rcnt);
            }
            root_ref = nodes[0].prim_count > 0u ? (nodes[0].left_or_prim | (nodes[0].prim_count << PT_REF_BITS)) : pair_of[0];
            // Longest chain of inner nodes from the root: bounds the stack occupancy.
            std::vector<std::pair<uint32_t, uint32_t>> todo;
            todo.push_back({0u, 0u});
            while (!todo.empty()) {
                auto [ni, d] = todo.back();
                todo.pop_back();
                if (nodes[ni].prim_count != 0u) continue;
                depth = d + 1 > depth ? d + 1 : depth;
                todo.push_back({ni + 1, d + 1});
                todo.push_back({nodes[ni].left_or_prim, d + 1});
            }
        }
        npair = (uint32_t)pairs.size();
        if ((rc = upload_vec(c, dprims, &sc.prims)) || (rc = upload_vec(c, shade, &sc.shade)) || (rc = upload_vec(c, pairs, &sc.pairs))) {
            free_scene(c);
            return rc;
        }
    } else {
        uint32_t neg_count = 0u, sky_index = 0xFFFFFFFFu;
        rc = relayout_on_device(c, figs, nfig, mats, nmat, nodes, nnode, prims, nprim, have_tree, sc, npair, depth, type_mask, root_ref, neg_count, sky_index);
        if (rc) { free_scene(c); return rc; }
        if (neg_count > 0u)  // rare: list the subtraction shapes in figure order
            for (uint32_t i = 0; i < nfig; i++)
                if (figs[i].csg_op == 1u) add_neg(figs[i]);
        if (sky_index != 0xFFFFFFFFu) {
            const ptx_material& m = mats[figs[sky_index].material_idx];
            sky_tex = m.texture_id;
            memcpy(sky_color, m.base_color, 12);
        }
    }
    if (have_tree) {
        memcpy(root_min, nodes[0].aabb_min, 12);
        memcpy(root_max, nodes[0].aabb_max, 12);
        eff_nprim = nprim;
    }
    std::vector<DPrim> dinf;
    for (uint32_t i = 0; i < ninf; i++) {
        uint32_t fi = infinite[i];
        if (fi == 0xFFFFFFFFu) continue;  // the reference's empty-list placeholder (bvh.wgsl:196-199)
        if (fi >= nfig) { free_scene(c); return fail(c, PTX_ERR_BAD_ARG, "infinite_indices entry out of range"); }
        if (figs[fi].csg_op == 1u) continue;  // bvh.wgsl:200-203
        dinf.push_back(make_prim(figs[fi], fi));
    }
    std::vector<DLight> dl;
    for (uint32_t i = 0; i < nlight; i++) {
        uint32_t fi = lights[i];
        if (fi == 0xFFFFFFFFu && nlight == 1u) break;  // placeholder list
        if (fi >= nfig) { free_scene(c); return fail(c, PTX_ERR_BAD_ARG, "light_indices entry out of range"); }
        const ptx_shape& f = figs[fi];
        const ptx_material& m = mats[f.material_idx];
        DLight L;
        memcpy(L.pos, f.position, 12);
        L.radius = f.radius;
        volatile float le0 = m.emission[0] * m.emission_strength, le1 = m.emission[1] * m.emission_strength,
                       le2 = m.emission[2] * m.emission_strength;
        L.le[0] = le0; L.le[1] = le1; L.le[2] = le2;
        volatile float a = 4.0f * PT_PI;  // lighting.wgsl:24: 4.0 * PI * r * r, left to right
        a = a * f.radius;
        a = a * f.radius;
        volatile float pdf = pt_host_rcp(a);
        L.area_pdf = pdf;
        dl.push_back(L);
    }
    if ((rc = upload_vec(c, dinf, &sc.inf)) || (rc = upload_vec(c, dl, &sc.lights)) || (rc = upload_vec(c, negs, &sc.negs))) {
        free_scene(c);
        return rc;
    }
    std::vector<uint32_t> px(tex_pixels, tex_pixels + npix);
    std::vector<ptx_texture_info> ti(tex_infos, tex_infos + ntex);
    for (auto& t : ti)
        if (t.width == 0 || t.height == 0 || (uint64_t)t.offset + (uint64_t)t.width * t.height > npix) {
            free_scene(c);
            return fail(c, PTX_ERR_BAD_ARG, "texture info outside the pixel array");
        }
    if ((rc = upload_vec(c, px, &sc.tex_pixels)) || (rc = upload_vec(c, ti, &sc.tex_infos))) {
        free_scene(c);
        return rc;
    }
    sc.nprim = eff_nprim;
    sc.ninf = (uint32_t)dinf.size();
    sc.nlight = (uint32_t)dl.size();
    sc.nneg = (uint32_t)negs.size();
    sc.ntex = ntex;
    sc.npix = npix;
    sc.nfig = nfig;
    sc.root_ref = root_ref;
    memcpy(sc.root_min, root_min, 12);
    memcpy(sc.root_max, root_max, 12);
    sc.sky_tex = sky_tex;
    memcpy(sc.sky_color, sky_color, 12);
    // a lane's stack holds at most one entry per inner node on the way down; one spare, none for a tree that is a single leaf
    uint32_t entries = depth == 0u ? 0u : depth + 1;
    sc.stack_entries = entries > 31u ? 31u : entries;
    sc.lds_stack_entries = sc.stack_entries;  // reduced below for the global-memory kernels
    sc.npair = npair;
    const uint32_t fractal_bits = (1u << FIG_MANDELBULB) | (1u << FIG_JULIA);
    sc.has_fractals = (type_mask & fractal_bits) != 0u ? 1u : 0u;
    {
        size_t bytes = (size_t)nprim * sizeof(DPrim) + dinf.size() * sizeof(DPrim) + (size_t)npair * sizeof(DNodePair) +
                       (size_t)nfig * sizeof(DShade) + dl.size() * sizeof(DLight) + negs.size() * sizeof(DNeg);
        sc.lds_scene_bytes = bytes <= PT_LDS_SCENE_MAX ? (uint32_t)bytes : 0u;
        if (sc.lds_scene_bytes == 0u && sc.lds_stack_entries > PT_LDS_STACK) sc.lds_stack_entries = PT_LDS_STACK;
    }
    c->sc = sc;
    c->have_scene = true;
    c->last_kernel = nullptr;
    c->last_info.clear();
    c->type_mask = type_mask;
    c->n_marched = 0u;  // torus / Moebius figures in the tree (their rays cost up to hundreds of times a closed-form figure's)
    for (uint32_t s_ = 0; s_ < nprim; s_++) {
        const uint32_t t_ = figs[prims[s_]].shape_type;
        if (t_ == FIG_TORUS || t_ == FIG_MEBIUS) c->n_marched += 1u;
    }
    if ((rc = configure(c))) { free_scene(c); return rc; }
    return PTX_OK;
}

int ptx_resize(ptx_ctx* c, uint32_t w, uint32_t h) {
    if (!c) return PTX_ERR_BAD_ARG;
    if (w == 0 || h == 0 || (uint64_t)w * h > 0x7FFFFFFFull) return fail(c, PTX_ERR_BAD_ARG, "bad frame size");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_frame(c);
    c->w = 0; c->h = 0; c->tx0 = 0; c->ty0 = 0; c->tw = 0; c->th = 0;  // no frame until every buffer of the new one exists
    const size_t n = (size_t)w * h;
    hipError_t e = hipMalloc((void**)&c->accum, n * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&c->out, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&c->ids, n * 4);
    if (e == hipSuccess && c->moments) e = hipMalloc((void**)&c->m2, n * 12);
    if (e != hipSuccess) {
        free_frame(c);
        return fail(c, e == hipErrorOutOfMemory ? PTX_ERR_OOM : PTX_ERR_HIP, std::string("ptx_resize: hipMalloc: ") + hipGetErrorString(e));
    }
    c->w = w; c->h = h;
    c->tx0 = 0; c->ty0 = 0; c->tw = w; c->th = h;
    const int rc = ptx_clear(c);
    if (rc != PTX_OK) { free_frame(c); c->w = 0; c->h = 0; c->tw = 0; c->th = 0; }
    return rc;
}

int ptx_set_tile(ptx_ctx* c, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
    if (!c) return PTX_ERR_BAD_ARG;
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_set_tile before ptx_resize");
    if (w == 0 || h == 0 || (uint64_t)x0 + w > c->w || (uint64_t)y0 + h > c->h) return fail(c, PTX_ERR_BAD_ARG, "tile outside the frame");
    if (c->dbg && (w != c->tw || h != c->th)) return fail(c, PTX_ERR_STATE, "tile size change while radiance capture is enabled");
    c->tx0 = x0; c->ty0 = y0; c->tw = w; c->th = h;
    return PTX_OK;
}

int ptx_set_partition(ptx_ctx* c, uint32_t rank, uint32_t world) {
    if (!c) return PTX_ERR_BAD_ARG;
    if (world == 0 || rank >= world) return fail(c, PTX_ERR_BAD_ARG, "partition rank outside [0, world)");
    c->part_rank = rank; c->part_world = world;
    return PTX_OK;
}

int ptx_clear(ptx_ctx* c) {
    if (c) c->order_dirty = true;
    if (!c) return PTX_ERR_BAD_ARG;
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_clear before ptx_resize");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)c->w * c->h;
    HIP_TRY(c, hipMemsetAsync(c->accum, 0, n * 16, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->out, 0, n * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->ids, 0xFF, n * 4, c->stream));
    if (c->m2) HIP_TRY(c, hipMemsetAsync(c->m2, 0, n * 12, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->stats, 0, 24 * sizeof(unsigned long long), c->stream));
    return PTX_OK;
}

// ptx_trace_n and ptx_trace_n_static.  `fixed`: the launch's scratch (pixel-claim counter, per-wave claim state, deep
// stack entries) is the file-scope array ptx_static_scratch instead of buffers the context allocates and grows, no
// events are created, and fractal scenes run whole-pixel jobs (no staging buffer): nothing is allocated.
static int launch_trace(ptx_ctx* c, const ptx_camera* cam, uint32_t first_frame, uint32_t first_sample, uint32_t n, bool fixed) {
    if (!c || !cam) return PTX_ERR_BAD_ARG;
    if (!c->have_scene) return fail(c, PTX_ERR_STATE, "ptx_trace before ptx_upload_scene");
    if (!c->accum || !c->out || !c->ids) return fail(c, PTX_ERR_STATE, "ptx_trace before ptx_resize");
    if (cam->width != c->w || cam->height != c->h) return fail(c, PTX_ERR_BAD_ARG, "camera width/height differ from the frame size");
    if (n == 0) return PTX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    TraceParams P;
    memset(&P, 0, sizeof(P));
    P.sc = c->sc;
    P.cam = *cam;
    P.first_frame = first_frame; P.first_sample = first_sample; P.n = n;
    P.tile_x0 = c->tx0; P.tile_y0 = c->ty0; P.tile_w = c->tw; P.tile_h = c->th;
    P.blocks_x = (c->tw + 15u) / 16u;
    uint32_t blocks_y = (c->th + 15u) / 16u;
    uint32_t total_blocks = P.blocks_x * blocks_y;
    P.part_rank = c->part_rank; P.part_world = c->part_world;
    P.n_blocks = total_blocks > c->part_rank ? (total_blocks - c->part_rank + c->part_world - 1u) / c->part_world : 0u;
    if (P.n_blocks == 0u) return PTX_OK;
    P.accum = c->accum; P.out_rgba8 = c->out; P.primary_ids = c->ids;
    P.m2 = c->moments ? c->m2 : nullptr;
    P.dbg_radiance = c->dbg; P.dbg_capacity = c->dbg_cap;
    P.stats = c->stats;
    const bool frac = c->sc.has_fractals != 0u;
    const size_t lds_all = trace_lds_bytes(c);
    const LaunchChoice L = resolve_launch(c, n, fixed, P.n_blocks, false);
    const LaunchForm& form = L.f;
    const TraceKernel kernel = form.kernel;
    const bool claim = form.claim, split = form.jobs != JOBS_PIXELS, sorted = form.jobs == JOBS_POOLS;
    if (claim && c->occ_per_cu < 1) return fail(c, PTX_ERR_STATE, "the trace kernel does not fit a compute unit with this scene's LDS size");
    const uint32_t grid = L.grid;
    const size_t waves = (size_t)grid * PTX_WAVES_PER_BLOCK;
    if (fixed) {
        if (!c->static_scratch) return fail(c, PTX_ERR_STATE, "ptx_trace_n_static: the static scratch array has no address on this device");
        if (waves > PTX_STATIC_MAX_WAVES)
            return fail(c, PTX_ERR_UNSUPPORTED, "ptx_trace_n_static: this launch runs " + std::to_string(waves) + " waves, the static scratch array holds the state of " +
                                                    std::to_string(PTX_STATIC_MAX_WAVES) + " (use ptx_trace_n)");
        if (c->sc.lds_scene_bytes == 0u && c->sc.stack_entries > PT_LDS_STACK) P.stack_spill = (StackEntry*)(c->static_scratch + PTX_STATIC_SPILL_OFFSET);
    } else {
        int rc = ensure_spill(c, waves, &P.stack_spill);
        if (rc) return rc;
    }
    if (fixed) {
        P.wave_state = (WaveState*)(c->static_scratch + PTX_STATIC_WAVES_OFFSET);
    } else {
        if (waves > c->wave_state_n) {
            if (c->wave_state) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->wave_state); c->wave_state = nullptr; c->wave_state_n = 0; }
            HIP_TRY(c, hipMalloc((void**)&c->wave_state, waves * sizeof(WaveState)));
            c->wave_state_n = waves;
        }
        P.wave_state = c->wave_state;
    }
    const uint32_t paced_cap = c->opt.claim_cap >= 0 ? (uint32_t)c->opt.claim_cap : 8u;
    if (claim) {
        if (!fixed && !c->flight_counters) {
            HIP_TRY(c, hipMalloc((void**)&c->flight_counters, (PTX_FLIGHTS + 1) * (PTX_CLAIM_RANGES + 1) * 256));
            HIP_TRY(c, hipMemset(c->flight_counters, 0, (PTX_FLIGHTS + 1) * (PTX_CLAIM_RANGES + 1) * 256));
        }
        P.job_counter = fixed ? (uint32_t*)c->static_scratch : c->flight_counters + PTX_FLIGHTS * (PTX_CLAIM_RANGES + 1) * 64;
        P.total_jobs = P.n_blocks * 256u;
        P.claim_tile = L.tile0;
        P.claim_shift = form.guided_k && P.claim_tile > 1u ? log2_ceil(form.guided_k * waves) : 0u;
        P.claim_period = form.paced_m && P.claim_tile > 1u ? form.paced_m * (uint32_t)waves : 0u;
        P.claim_cap_shift = paced_cap ? log2_ceil(paced_cap * waves) + 6u : 0u;
        P.share_walk = c->shares[0]; P.share_post = c->shares[1]; P.share_march = c->shares[2];
        P.walk_single = c->sc.stack_entries >= PT_DEEP_TREE ? 1u : 0u;
        if (c->opt.walk_mode >= 0) P.walk_single = (uint32_t)c->opt.walk_mode;
        P.share_leaf = c->share_leaf;
    }
    P.sc.lds_stack_entries = L.lds_stack;
    P.frac_fast = c->opt.arith_mode >= 1 && form.sched && frac ? (uint32_t)c->opt.arith_mode : 0u;  // the wave-scheduled fractal kernels have the mode; the others are always exact
    if (sorted) {
        P.pool_m = form.pool_m;
        P.pool_guided = c->opt.pool_guided >= 0 ? (uint32_t)c->opt.pool_guided : 0u;
        P.pool_finish = c->opt.pool_finish >= 0 ? (uint32_t)c->opt.pool_finish : 256u;
        P.job_transposed = form.spread ? 1u : 0u;
    }
    const bool zero_based = kernel == ptx_trace_kernel_lds_ranged;
    const uint32_t ranges = form.counters;
    c->last_kernel = trace_kernel_label(kernel);
    c->last_info = describe_launch(L) + "; grid=" + std::to_string(grid);
    const uint32_t chunk = L.chunk, depth = L.depth;
    const size_t frame_pixels = (size_t)c->w * c->h;
    if (split) {
        P.frame_pixels = (uint32_t)frame_pixels;
        if (!c->barrier) HIP_TRY(c, hipEventCreateWithFlags(&c->barrier, hipEventDisableTiming));
    } else {
        c->order_dirty = true;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!fixed) {
        if (c->pending.size() >= 256) { int rc = collect_timing(c); if (rc) return rc; }
        int rc;
        if ((rc = get_event(c, &e0)) || (rc = get_event(c, &e1))) return rc;
        HIP_TRY(c, hipEventRecord(e0, c->stream));
    }
    for (uint32_t done = 0; done < n; done += chunk) {
        const uint32_t m = n - done < chunk ? n - done : chunk;
        P.first_frame = first_frame + done; P.first_sample = first_sample + done; P.n = m;
        if (!split) {
            P.split_n = 0u; P.total_jobs = P.n_blocks * 256u;
            if (claim) {
                set_ranges(P, ranges, grid * 256u);
                if (zero_based) HIP_TRY(c, hipMemsetAsync(P.job_counter, 0, (P.claim_ranges + 1u) * 256u, c->stream));  // the counters and the word that says which have run out
                else HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)P.job_counter, (int)(grid * 256u), 1, c->stream));
            }
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(PTX_BLOCK), lds_all, c->stream, P);
            HIP_TRY(c, hipGetLastError());
            continue;
        }
        P.split_n = m; P.total_jobs = P.n_blocks * 256u * m;
        P.claim_tile = reservation_jobs(c, form, m, P.n_blocks);
        P.claim_shift = form.guided_k && P.claim_tile > 1u ? log2_ceil(form.guided_k * waves) : 0u;
        P.claim_period = form.paced_m && P.claim_tile > 1u ? form.paced_m * (uint32_t)waves : 0u;
        ptx_ctx::Flight& f = c->flights[c->flight_seq % depth];
        const uint32_t slot = (uint32_t)(&f - c->flights);
        if (!f.stream) {
            HIP_TRY(c, hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&f.traced, hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&f.folded, hipEventDisableTiming));
        }
        const bool needs_spill = P.stack_spill != nullptr || (c->sc.lds_scene_bytes == 0u && c->sc.stack_entries > PT_LDS_STACK);
        if ((size_t)m * frame_pixels > f.staging_n || waves > f.wave_state_n || (needs_spill && waves > f.spill_waves)) {  // grow: not while the flight's last launch uses them
            if (f.used) HIP_TRY(c, hipEventSynchronize(f.folded));
            if ((size_t)m * frame_pixels > f.staging_n) {
                if (f.staging) { (void)hipFree(f.staging); f.staging = nullptr; f.staging_n = 0; }
                HIP_TRY(c, hipMalloc((void**)&f.staging, (size_t)m * frame_pixels * sizeof(float4)));
                f.staging_n = (size_t)m * frame_pixels;
            }
            if (waves > f.wave_state_n) {
                if (f.wave_state) { (void)hipFree(f.wave_state); f.wave_state = nullptr; f.wave_state_n = 0; }
                HIP_TRY(c, hipMalloc((void**)&f.wave_state, waves * sizeof(WaveState)));
                f.wave_state_n = waves;
            }
            if (needs_spill && waves > f.spill_waves) {
                if (f.spill) { (void)hipFree(f.spill); f.spill = nullptr; f.spill_waves = 0; }
                HIP_TRY(c, hipMalloc((void**)&f.spill, waves * (31u - PT_LDS_STACK) * 64u * sizeof(StackEntry)));
                f.spill_waves = waves;
            }
        }
        if (needs_spill) P.stack_spill = f.spill;
        if (sorted) {
            const uint32_t pm = form.pool_m;
            if ((size_t)grid > f.pool_wgs || pm > f.pool_m) {  // the buffers only grow: a smaller pool uses a part of each workgroup's share
                const size_t wgs = (size_t)grid > f.pool_wgs ? (size_t)grid : f.pool_wgs;
                const uint32_t cap = pm > f.pool_m ? pm : f.pool_m;
                if (f.used) HIP_TRY(c, hipEventSynchronize(f.folded));
                if (f.pool) (void)hipFree(f.pool);
                if (f.pool_keys) (void)hipFree(f.pool_keys);
                if (f.pool_order) (void)hipFree(f.pool_order);
                f.pool = nullptr; f.pool_keys = nullptr; f.pool_order = nullptr; f.pool_wgs = 0; f.pool_m = 0;
                HIP_TRY(c, hipMalloc((void**)&f.pool, wgs * 2u * cap * sizeof(PoolPath)));
                HIP_TRY(c, hipMalloc((void**)&f.pool_keys, wgs * 2u * cap * sizeof(uint8_t)));
                HIP_TRY(c, hipMalloc((void**)&f.pool_order, wgs * cap * sizeof(uint16_t)));
                f.pool_wgs = wgs; f.pool_m = cap;
            }
            P.pool = f.pool; P.pool_keys = f.pool_keys; P.pool_order = f.pool_order;
        }
        // order: after whatever other calls have enqueued on the context's stream (a clear, an upload, other kernels) and after the
        // fold that read this flight's staging buffer last
        if (c->order_dirty) { HIP_TRY(c, hipEventRecord(c->barrier, c->stream)); c->order_dirty = false; }
        HIP_TRY(c, hipStreamWaitEvent(f.stream, c->barrier, 0));
        if (f.used) HIP_TRY(c, hipStreamWaitEvent(f.stream, f.folded, 0));
        P.staging = f.staging;
        P.wave_state = f.wave_state;
        P.job_counter = c->flight_counters + slot * ((PTX_CLAIM_RANGES + 1) * 64u);
        // the lane-owned kernels start with jobs 0 .. grid * 256 - 1 in their lanes; the pools reserve every job they trace
        set_ranges(P, sorted ? 1u : ranges, sorted ? 0u : grid * 256u);
        if (zero_based || sorted) HIP_TRY(c, hipMemsetAsync(P.job_counter, 0, (P.claim_ranges + 1u) * 256u, f.stream));
        else HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)P.job_counter, (int)(grid * 256u), 1, f.stream));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(PTX_BLOCK), lds_all, f.stream, P);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(f.traced, f.stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, f.traced, 0));
        TraceParams F = P;
        F.split_n = 0u;
        F.job_transposed = 0u;
        hipLaunchKernelGGL(ptx_fold_kernel, dim3(P.n_blocks), dim3(256), 0, c->stream, F);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(f.folded, c->stream));
        f.used = true;
        c->flight_seq += 1u;
    }
    if (!fixed) {
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        c->pending.push_back({e0, e1});
    }
    return PTX_OK;
}

int ptx_trace_n(ptx_ctx* c, const ptx_camera* cam, uint32_t first_frame, uint32_t first_sample, uint32_t n) {
    return launch_trace(c, cam, first_frame, first_sample, n, false);
}

int ptx_trace_n_static(ptx_ctx* c, const ptx_camera* cam, uint32_t first_frame, uint32_t first_sample, uint32_t n) {
    return launch_trace(c, cam, first_frame, first_sample, n, true);
}

int ptx_trace(ptx_ctx* c, const ptx_camera* cam) {
    if (!c || !cam) return PTX_ERR_BAD_ARG;
    return ptx_trace_n(c, cam, cam->frame_index, cam->sample_count, 1);
}

int ptx_sync(ptx_ctx* c) {
    if (!c) return PTX_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->drained = true;
    return PTX_OK;
}

static int read_back(ptx_ctx* c, void* dst, const void* src, size_t bytes, const char* what) {
    if (!c || !dst) return PTX_ERR_BAD_ARG;
    if (!src) return fail(c, PTX_ERR_STATE, std::string(what) + ": buffer not allocated");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->drained = true;
    return PTX_OK;
}
int ptx_read_accum(ptx_ctx* c, float* o) { return read_back(c, o, c ? c->accum : nullptr, c ? (size_t)c->w * c->h * 16 : 0, "ptx_read_accum"); }
int ptx_write_accum(ptx_ctx* c, const float* rgba, const float* m2) {
    if (c) c->order_dirty = true;
    if (!c || !rgba) return PTX_ERR_BAD_ARG;
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_write_accum before ptx_resize");
    if (m2 && !c->m2) return fail(c, PTX_ERR_STATE, "ptx_write_accum with moments, but moments are not enabled");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->accum, rgba, (size_t)c->w * c->h * 16, hipMemcpyHostToDevice));
    if (m2) HIP_TRY(c, hipMemcpy(c->m2, m2, (size_t)c->w * c->h * 12, hipMemcpyHostToDevice));
    return PTX_OK;
}
int ptx_read_output(ptx_ctx* c, uint8_t* o) { return read_back(c, o, c ? c->out : nullptr, c ? (size_t)c->w * c->h * 4 : 0, "ptx_read_output"); }
int ptx_read_primary_ids(ptx_ctx* c, uint32_t* o) { return read_back(c, o, c ? c->ids : nullptr, c ? (size_t)c->w * c->h * 4 : 0, "ptx_read_primary_ids"); }
int ptx_read_moments(ptx_ctx* c, float* o) { return read_back(c, o, c ? c->m2 : nullptr, c ? (size_t)c->w * c->h * 12 : 0, "ptx_read_moments"); }

int ptx_debug_capture(ptx_ctx* c, uint32_t max_samples) {
    if (c) c->order_dirty = true;
    if (!c) return PTX_ERR_BAD_ARG;
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_debug_capture before ptx_resize");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->dbg) { (void)hipFree(c->dbg); c->dbg = nullptr; c->dbg_cap = 0; }
    if (max_samples == 0) return PTX_OK;
    size_t bytes = (size_t)max_samples * c->tw * c->th * 12;
    HIP_TRY(c, hipMalloc((void**)&c->dbg, bytes));
    HIP_TRY(c, hipMemset(c->dbg, 0, bytes));
    c->dbg_cap = max_samples;
    return PTX_OK;
}
int ptx_read_debug_radiance(ptx_ctx* c, float* out, uint32_t n) {
    if (!c || !out) return PTX_ERR_BAD_ARG;
    if (!c->dbg || n > c->dbg_cap) return fail(c, PTX_ERR_STATE, "radiance capture not enabled for that many samples");
    return read_back(c, out, c->dbg, (size_t)n * c->tw * c->th * 12, "ptx_read_debug_radiance");
}

int ptx_enable_moments(ptx_ctx* c, int enable) {
    if (c) c->order_dirty = true;
    if (!c) return PTX_ERR_BAD_ARG;
    c->moments = enable != 0;
    if (c->moments && c->accum && !c->m2) {
        HIP_TRY(c, hipSetDevice(c->device));
        size_t n = (size_t)c->w * c->h;
        HIP_TRY(c, hipMalloc((void**)&c->m2, n * 12));
        HIP_TRY(c, hipMemsetAsync(c->m2, 0, n * 12, c->stream));
    }
    return PTX_OK;
}

void* ptx_device_accum(ptx_ctx* c) { return c ? (void*)c->accum : nullptr; }
void* ptx_device_moments(ptx_ctx* c) { return c ? (void*)c->m2 : nullptr; }

int ptx_merge_moments(ptx_ctx* c, uint32_t first_pixel, uint32_t n_pixels, uint32_t n_a, const void* other_mean4,
                      const void* other_m2, uint32_t n_b, const ptx_camera* cam) {
    if (c) c->order_dirty = true;
    if (!c || !other_mean4 || !cam) return PTX_ERR_BAD_ARG;
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_merge_moments before ptx_resize");
    if ((uint64_t)first_pixel + n_pixels > (uint64_t)c->w * c->h) return fail(c, PTX_ERR_BAD_ARG, "pixel range outside the frame");
    if (n_pixels == 0 || n_b == 0) return PTX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(ptx_merge_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, c->stream, c->accum,
                       (c->moments && other_m2) ? c->m2 : nullptr, c->out, first_pixel, n_pixels, (float)n_a,
                       (const float4*)other_mean4, (const float*)other_m2, (float)n_b, *cam);
    HIP_TRY(c, hipGetLastError());
    return PTX_OK;
}

int ptx_tonemap(ptx_ctx* c, uint32_t first_pixel, uint32_t n_pixels, const ptx_camera* cam) {
    if (c) c->order_dirty = true;
    if (!c || !cam) return PTX_ERR_BAD_ARG;
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_tonemap before ptx_resize");
    if ((uint64_t)first_pixel + n_pixels > (uint64_t)c->w * c->h) return fail(c, PTX_ERR_BAD_ARG, "pixel range outside the frame");
    if (n_pixels == 0) return PTX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(ptx_tonemap_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, c->stream, c->accum, c->out, first_pixel,
                       n_pixels, *cam);
    HIP_TRY(c, hipGetLastError());
    return PTX_OK;
}

// ---- multi-GPU exchange over RCCL (collective.h) ---------------------------------------------------------------------

#define NCCL_TRY(ctx, api_, expr)                                                                              \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(ctx, PTX_ERR_HIP, std::string(#expr) + ": " + (api_)->GetErrorString(r_)); \
    } while (0)

int ptx_comm_unique_id(void* id128) {
    if (!id128) return PTX_ERR_BAD_ARG;
    coll::Api* a = coll::api();
    if (!a) return PTX_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 opaque bytes");
    ncclUniqueId id;
    if (a->GetUniqueId(&id) != ncclSuccess) return PTX_ERR_HIP;
    memcpy(id128, &id, sizeof(id));
    return PTX_OK;
}

static int comm_release(ptx_ctx* c) {
    if (c->comm && c->comm_owned) {
        coll::Api* a = coll::api();
        if (a) { (void)hipStreamSynchronize(c->stream); (void)a->CommDestroy(c->comm); }
    }
    c->comm = nullptr; c->comm_owned = false; c->comm_rank = 0; c->comm_world = 1;
    return PTX_OK;
}

int ptx_comm_init(ptx_ctx* c, const void* id128, uint32_t rank, uint32_t world) {
    if (!c || !id128) return PTX_ERR_BAD_ARG;
    if (world == 0 || rank >= world) return fail(c, PTX_ERR_BAD_ARG, "ptx_comm_init: rank outside [0, world)");
    coll::Api* a = coll::api();
    if (!a) return fail(c, PTX_ERR_UNSUPPORTED, "ptx_comm_init: " + coll::api_error());
    HIP_TRY(c, hipSetDevice(c->device));
    comm_release(c);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    NCCL_TRY(c, a, a->CommInitRank(&c->comm, (int)world, id, (int)rank));
    c->comm_owned = true; c->comm_rank = (int)rank; c->comm_world = (int)world;
    return PTX_OK;
}

int ptx_comm_adopt(ptx_ctx* c, void* nccl_comm) {
    if (!c) return PTX_ERR_BAD_ARG;
    comm_release(c);
    if (!nccl_comm) return PTX_OK;
    coll::Api* a = coll::api();
    if (!a) return fail(c, PTX_ERR_UNSUPPORTED, "ptx_comm_adopt: " + coll::api_error());
    int world = 0, rank = 0;
    NCCL_TRY(c, a, a->CommCount((ncclComm_t)nccl_comm, &world));
    NCCL_TRY(c, a, a->CommUserRank((ncclComm_t)nccl_comm, &rank));
    c->comm = (ncclComm_t)nccl_comm; c->comm_owned = false; c->comm_rank = rank; c->comm_world = world;
    return PTX_OK;
}

int ptx_comm_free(ptx_ctx* c) {
    if (!c) return PTX_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    return comm_release(c);
}

// Runs `body` (which returns the first failing ncclResult_t of its sends and receives) between ncclGroupStart and
// ncclGroupEnd.  The group is closed on every path: a call that failed inside an open group must not leave the
// communicator waiting for the rest of it.
extern "C++" template <class F>
static int nccl_grouped(ptx_ctx* c, coll::Api* a, const char* what, F body) {
    NCCL_TRY(c, a, a->GroupStart());
    const ncclResult_t inside = body();
    const ncclResult_t closed = a->GroupEnd();
    if (inside != ncclSuccess) return fail(c, PTX_ERR_HIP, std::string(what) + ": " + a->GetErrorString(inside));
    if (closed != ncclSuccess) return fail(c, PTX_ERR_HIP, std::string(what) + ": ncclGroupEnd: " + a->GetErrorString(closed));
    return PTX_OK;
}

void ptx_slice_of(uint64_t n_pixels, uint32_t world, uint32_t rank, uint64_t* first, uint64_t* count) {
    uint64_t f = 0, k = 0;
    if (world > 0u && rank < world) coll::slice_of(n_pixels, world, rank, &f, &k);
    if (first) *first = f;
    if (count) *count = k;
}

int ptx_merge_sample_split(ptx_ctx* c, const uint32_t* counts, const ptx_camera* cam) {
    if (c) c->order_dirty = true;
    if (!c || !counts || !cam) return PTX_ERR_BAD_ARG;
    if (!c->comm) return fail(c, PTX_ERR_STATE, "ptx_merge_sample_split before ptx_comm_init / ptx_comm_adopt");
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_merge_sample_split before ptx_resize");
    coll::Api* a = coll::api();
    if (!a) return fail(c, PTX_ERR_UNSUPPORTED, "ptx_merge_sample_split: " + coll::api_error());
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t world = (uint32_t)c->comm_world, rank = (uint32_t)c->comm_rank;
    const uint64_t n = (uint64_t)c->w * c->h;
    const bool with_m2 = c->moments && c->m2;
    uint64_t my_first, my_cnt;
    coll::slice_of(n, world, rank, &my_first, &my_cnt);
    const size_t need = (size_t)world * my_cnt * (with_m2 ? 7u : 4u);
    if (need > c->xchg_floats) {
        if (c->xchg) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->xchg); c->xchg = nullptr; c->xchg_floats = 0; }
        HIP_TRY(c, hipMalloc((void**)&c->xchg, (need ? need : 4) * sizeof(float)));
        c->xchg_floats = need;
    }
    float* acc = (float*)c->accum;
    auto recv_a = [&](uint32_t r) { return c->xchg + (size_t)r * my_cnt * 4u; };
    auto recv_m = [&](uint32_t r) { return c->xchg + (size_t)world * my_cnt * 4u + (size_t)r * my_cnt * 3u; };
    hipStream_t st = c->stream;
    // 1. every rank receives its own slice of every rank's partial (its own by a device copy)
    if (my_cnt) {
        HIP_TRY(c, hipMemcpyAsync(recv_a(rank), acc + my_first * 4u, my_cnt * 16u, hipMemcpyDeviceToDevice, st));
        if (with_m2) HIP_TRY(c, hipMemcpyAsync(recv_m(rank), c->m2 + my_first * 3u, my_cnt * 12u, hipMemcpyDeviceToDevice, st));
    }
    if (world > 1u) {
        const int rc = nccl_grouped(c, a, "ptx_merge_sample_split: slice exchange", [&]() -> ncclResult_t {
            ncclResult_t r = ncclSuccess;
            for (uint32_t j = 0; j < world && r == ncclSuccess; j++) {
                if (j == rank) continue;
                uint64_t f, k;
                coll::slice_of(n, world, j, &f, &k);
                if (k) r = a->Send(acc + f * 4u, k * 4u, ncclFloat, (int)j, c->comm, st);
                if (my_cnt && r == ncclSuccess) r = a->Recv(recv_a(j), my_cnt * 4u, ncclFloat, (int)j, c->comm, st);
                if (with_m2) {
                    if (k && r == ncclSuccess) r = a->Send(c->m2 + f * 3u, k * 3u, ncclFloat, (int)j, c->comm, st);
                    if (my_cnt && r == ncclSuccess) r = a->Recv(recv_m(j), my_cnt * 3u, ncclFloat, (int)j, c->comm, st);
                }
            }
            return r;
        });
        if (rc) return rc;
    }
    // 2. Chan's update in fixed rank order.  A rank that holds no samples (counts[r] == 0) contributes nothing, whatever its
    // buffers hold: the first rank with samples is the start, the later ones are folded in.  (No rank with samples: rank 0's slice.)
    if (my_cnt) {
        uint32_t r0 = 0u;
        while (r0 + 1u < world && counts[r0] == 0u) r0++;
        if (counts[r0] == 0u) r0 = 0u;
        HIP_TRY(c, hipMemcpyAsync(acc + my_first * 4u, recv_a(r0), my_cnt * 16u, hipMemcpyDeviceToDevice, st));
        if (with_m2) HIP_TRY(c, hipMemcpyAsync(c->m2 + my_first * 3u, recv_m(r0), my_cnt * 12u, hipMemcpyDeviceToDevice, st));
        uint64_t n_a = counts[r0];
        for (uint32_t r = r0 + 1u; r < world; r++) {
            if (counts[r] == 0u) continue;
            hipLaunchKernelGGL(ptx_merge_kernel, dim3((uint32_t)((my_cnt + 255u) / 256u)), dim3(256), 0, st, c->accum, with_m2 ? c->m2 : nullptr, c->out,
                               (uint32_t)my_first, (uint32_t)my_cnt, (float)n_a, (const float4*)recv_a(r), with_m2 ? (const float*)recv_m(r) : nullptr,
                               (float)counts[r], *cam);
            HIP_TRY(c, hipGetLastError());
            n_a += counts[r];
        }
    }
    // 3. the merged slices go to everyone
    if (world > 1u) {
        const int rc = nccl_grouped(c, a, "ptx_merge_sample_split: merged slices", [&]() -> ncclResult_t {
            ncclResult_t r = ncclSuccess;
            for (uint32_t j = 0; j < world && r == ncclSuccess; j++) {
                if (j == rank) continue;
                uint64_t f, k;
                coll::slice_of(n, world, j, &f, &k);
                if (my_cnt) r = a->Send(acc + my_first * 4u, my_cnt * 4u, ncclFloat, (int)j, c->comm, st);
                if (k && r == ncclSuccess) r = a->Recv(acc + f * 4u, k * 4u, ncclFloat, (int)j, c->comm, st);
                if (with_m2) {
                    if (my_cnt && r == ncclSuccess) r = a->Send(c->m2 + my_first * 3u, my_cnt * 3u, ncclFloat, (int)j, c->comm, st);
                    if (k && r == ncclSuccess) r = a->Recv(c->m2 + f * 3u, k * 3u, ncclFloat, (int)j, c->comm, st);
                }
            }
            return r;
        });
        if (rc) return rc;
    }
    // 4. the rgba8 image of the whole frame from the merged means
    hipLaunchKernelGGL(ptx_tonemap_kernel, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, st, c->accum, c->out, 0u, (uint32_t)n, *cam);
    HIP_TRY(c, hipGetLastError());
    return PTX_OK;
}

int ptx_assemble_tiles(ptx_ctx* c, const ptx_camera* cam) {
    if (c) c->order_dirty = true;
    if (!c || !cam) return PTX_ERR_BAD_ARG;
    if (!c->comm) return fail(c, PTX_ERR_STATE, "ptx_assemble_tiles before ptx_comm_init / ptx_comm_adopt");
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_assemble_tiles before ptx_resize");
    coll::Api* a = coll::api();
    if (!a) return fail(c, PTX_ERR_UNSUPPORTED, "ptx_assemble_tiles: " + coll::api_error());
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t n = (uint64_t)c->w * c->h;
    const uint32_t world = (uint32_t)c->comm_world, rank = (uint32_t)c->comm_rank;
    hipStream_t st = c->stream;
    if (world > 1u && c->part_world == world && c->part_rank == rank) {
        // Ownership is known (ptx_set_partition with the communicator's rank and size): every rank packs the blocks it traced,
        // sends them to every peer and receives the peers' -- an all-gather of uneven shares, every peer link busy at once,
        // (world - 1) / world of the frame in and 1 / world of it out per link pair, half of what a ring all-reduce of the whole
        // buffer moves -- and unpacks them into its accumulation buffer.  Exact: no arithmetic is done on the pixels.
        BlockPart B;
        B.tile_x0 = c->tx0; B.tile_y0 = c->ty0; B.tile_w = c->tw; B.tile_h = c->th;
        B.blocks_x = (c->tw + 15u) / 16u; B.width = c->w; B.world = world;
        const uint32_t total_blocks = B.blocks_x * ((c->th + 15u) / 16u);
        auto blocks_of = [&](uint32_t r) { return total_blocks > r ? (total_blocks - r + world - 1u) / world : 0u; };
        std::vector<size_t> first(world + 1u, 0);
        for (uint32_t r = 0; r < world; r++) first[r + 1u] = first[r] + (size_t)blocks_of(r) * 256u;
        const size_t need = first[world] * 4u;  // floats
        if (need > c->xchg_floats) {
            if (c->xchg) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->xchg); c->xchg = nullptr; c->xchg_floats = 0; }
            HIP_TRY(c, hipMalloc((void**)&c->xchg, (need ? need : 4) * sizeof(float)));
            c->xchg_floats = need;
        }
        float4* packed = (float4*)c->xchg;
        const uint32_t mine = blocks_of(rank);
        if (mine) {
            B.rank = rank;
            hipLaunchKernelGGL(ptx_pack_blocks_kernel, dim3(mine), dim3(256), 0, st, (const float4*)c->accum, packed + first[rank], B);
            HIP_TRY(c, hipGetLastError());
        }
        const int rc = nccl_grouped(c, a, "ptx_assemble_tiles: block exchange", [&]() -> ncclResult_t {
            ncclResult_t r = ncclSuccess;
            for (uint32_t j = 0; j < world && r == ncclSuccess; j++) {
                if (j == rank) continue;
                if (mine) r = a->Send(packed + first[rank], (size_t)mine * 1024u, ncclFloat, (int)j, c->comm, st);
                if (blocks_of(j) && r == ncclSuccess) r = a->Recv(packed + first[j], (size_t)blocks_of(j) * 1024u, ncclFloat, (int)j, c->comm, st);
            }
            return r;
        });
        if (rc) return rc;
        for (uint32_t j = 0; j < world; j++) {
            if (j == rank || blocks_of(j) == 0u) continue;
            B.rank = j;
            hipLaunchKernelGGL(ptx_unpack_blocks_kernel, dim3(blocks_of(j)), dim3(256), 0, st, c->accum, (const float4*)(packed + first[j]), B);
            HIP_TRY(c, hipGetLastError());
        }
    } else if (world > 1u) {
        // Any other disjoint split (tiles set with ptx_set_tile, a partition that is not the communicator's): the ranks' buffers
        // are disjoint and untouched pixels are zero, so one SUM all-reduce of the whole buffer is exact.
        NCCL_TRY(c, a, a->AllReduce(c->accum, c->accum, n * 4u, ncclFloat, ncclSum, c->comm, st));
    }
    hipLaunchKernelGGL(ptx_tonemap_kernel, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, st, c->accum, c->out, 0u, (uint32_t)n, *cam);
    HIP_TRY(c, hipGetLastError());
    return PTX_OK;
}

#ifdef PTX_SITE_STATS
// Diagnostic builds only: reads and clears the per-site counters ({wave executions, lanes} per site).
extern "C" int ptx_debug_sites(unsigned long long* out, int n_pairs) {
    if (n_pairs > PT_NSITES) n_pairs = PT_NSITES;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pt_sites), (size_t)n_pairs * 16) != hipSuccess) return -1;
    static unsigned long long zeros[PT_NSITES * 2];
    if (hipMemcpyToSymbol(HIP_SYMBOL(pt_sites), zeros, sizeof(zeros)) != hipSuccess) return -1;
    return 0;
}
#endif

// ---- GPU BVH build (bvh_build.h) ---------------------------------------------------------------------------------------

int ptx_bvh_build(ptx_ctx* c, const float* aabbs, uint32_t n, ptx_bvh_node* nodes_out, uint32_t* n_nodes_out, uint32_t* prim_indices_out) {
    if (c) c->order_dirty = true;
    if (!c || !nodes_out || !n_nodes_out || (n > 0u && (!aabbs || !prim_indices_out))) return PTX_ERR_BAD_ARG;
    if (n == 0u) {  // bvh.rs:38-43
        memset(nodes_out, 0, sizeof(ptx_bvh_node));
        *n_nodes_out = 1u;
        return PTX_OK;
    }
    if (n > 0x3FFFFFFFu) return fail(c, PTX_ERR_BAD_ARG, "ptx_bvh_build: more than 2^30 primitives");
    HIP_TRY(c, hipSetDevice(c->device));
    using namespace bvhb;
    // the bvh_timing option: wall time of the phases on stderr (measurement runs)
    const bool timing = c->opt.bvh_timing;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    const auto t_start = now();
    DevBufs mem;
    float *d_aabb, *d_cent;
    uint32_t *d_idx[2], *d_seg, *d_pass, *d_P, *d_X, *d_Y, *d_flag, *d_rank, *d_sums, *d_bins, *d_counter;
    Rec* d_recs;
    FlatNode* d_nodes;
    const size_t big_cap = (size_t)n / SMALL_MAX + 2u;
    HIP_TRY(c, mem.alloc((void**)&d_aabb, (size_t)n * 24));
    HIP_TRY(c, mem.alloc((void**)&d_cent, (size_t)n * 12));
    HIP_TRY(c, mem.alloc((void**)&d_idx[0], (size_t)n * 4));
    HIP_TRY(c, mem.alloc((void**)&d_idx[1], (size_t)n * 4));
    HIP_TRY(c, mem.alloc((void**)&d_seg, (size_t)n * 4));
    HIP_TRY(c, mem.alloc((void**)&d_pass, (size_t)n * 4));
    HIP_TRY(c, mem.alloc((void**)&d_P, ((size_t)n + 1) * 4));
    HIP_TRY(c, mem.alloc((void**)&d_X, (size_t)n * 4));
    HIP_TRY(c, mem.alloc((void**)&d_Y, (size_t)n * 4));
    HIP_TRY(c, mem.alloc((void**)&d_flag, (size_t)n * 4));
    HIP_TRY(c, mem.alloc((void**)&d_rank, ((size_t)n + 1) * 4));
    HIP_TRY(c, mem.alloc((void**)&d_sums, ((size_t)n / SCAN_TILE + 2) * 4));
    HIP_TRY(c, mem.alloc((void**)&d_bins, big_cap * BIN_WORDS * 4));
    HIP_TRY(c, mem.alloc((void**)&d_counter, 8));  // [0] ranges above SMALL_MAX of the next level, [1] NaN seen
    HIP_TRY(c, mem.alloc((void**)&d_recs, (size_t)n * 2 * sizeof(Rec)));
    HIP_TRY(c, mem.alloc((void**)&d_nodes, (size_t)n * 2 * sizeof(FlatNode)));
    hipStream_t st = c->stream;
    const double ms_alloc = ms_since(t_start);
    const uint32_t pos_blocks = (n + 255u) / 256u;
    // the chunked kernels: at most 2048 workgroups, chunks a whole number of 256-position tiles
    const uint32_t chunk_blocks = pos_blocks < 2048u ? pos_blocks : 2048u;
    const uint32_t chunk = ((pos_blocks + chunk_blocks - 1u) / chunk_blocks) * 256u;
    const uint32_t chunk_grid = (n + chunk - 1u) / chunk;
    HIP_TRY(c, hipMemcpyAsync(d_aabb, aabbs, (size_t)n * 24, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(d_counter, 0, 8, st));
    hipLaunchKernelGGL(prep_kernel, dim3(pos_blocks), dim3(256), 0, st, n, (const float*)d_aabb, d_cent, d_idx[0], d_seg, d_counter + 1);
    hipLaunchKernelGGL(root_kernel, dim3(1), dim3(1), 0, st, d_recs, n, d_counter);
    HIP_TRY(c, hipGetLastError());
    uint32_t has_nan = 0u;
    HIP_TRY(c, hipMemcpyAsync(&has_nan, d_counter + 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    if (has_nan) return fail(c, PTX_ERR_BAD_ARG, "ptx_bvh_build: a box holds a NaN");
    const double ms_in = ms_since(t_start);
    std::vector<std::pair<uint32_t, uint32_t>> levels;  // {first record, records} per level
    uint32_t base = 0u, nseg = 1u, nbig = n > SMALL_MAX ? 1u : 0u, cur = 0u;
    for (;;) {
        if (levels.size() >= 8192u) return fail(c, PTX_ERR_STATE, "ptx_bvh_build: tree deeper than 8192 levels");
        levels.push_back({base, nseg});
        const uint32_t seg_blocks = (nseg + 255u) / 256u;
        // bounds of every range of the level; bins and the split of its inner ranges
        if (nbig > 0u) {
            hipLaunchKernelGGL(bounds_big_kernel, dim3(chunk_grid), dim3(256), 0, st, n, chunk, (const uint32_t*)d_seg, d_recs, (const uint32_t*)d_idx[cur], (const float*)d_aabb);
            hipLaunchKernelGGL(bins_clear_kernel, dim3(nbig), dim3(256), 0, st, d_bins, nbig);
            hipLaunchKernelGGL(bins_big_kernel, dim3(chunk_grid), dim3(256), 0, st, n, chunk, (const uint32_t*)d_seg, (const Rec*)d_recs, (const uint32_t*)d_idx[cur],
                               (const float*)d_aabb, (const float*)d_cent, d_bins);
            hipLaunchKernelGGL(split_big_kernel, dim3(seg_blocks), dim3(256), 0, st, d_recs, base, nseg, (const uint32_t*)d_bins);
        }
        hipLaunchKernelGGL(seg_small_kernel, dim3((nseg + 3u) / 4u), dim3(256), 0, st, d_recs, base, nseg, (const uint32_t*)d_idx[cur], (const float*)d_aabb, (const float*)d_cent);
        hipLaunchKernelGGL(inner_flag_kernel, dim3(seg_blocks), dim3(256), 0, st, (const Rec*)d_recs, base, nseg, d_flag);
        HIP_TRY(c, hipGetLastError());
        { int rc = scan_u32(c, d_flag, nseg, d_rank, d_sums); if (rc) return rc; }
        uint32_t inner = 0u;
        HIP_TRY(c, hipMemcpyAsync(&inner, d_rank + nseg, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (inner == 0u) break;
        // the partition of every inner range, then its two children
        hipLaunchKernelGGL(flag_kernel, dim3(pos_blocks), dim3(256), 0, st, n, (const uint32_t*)d_seg, (const Rec*)d_recs, (const uint32_t*)d_idx[cur], (const float*)d_cent, d_pass);
        HIP_TRY(c, hipGetLastError());
        { int rc = scan_u32(c, d_pass, n, d_P, d_sums); if (rc) return rc; }
        hipLaunchKernelGGL(pair_kernel, dim3(pos_blocks), dim3(256), 0, st, n, (const uint32_t*)d_seg, (const Rec*)d_recs, (const uint32_t*)d_pass, (const uint32_t*)d_P, d_X, d_Y);
        hipLaunchKernelGGL(move_kernel, dim3(pos_blocks), dim3(256), 0, st, n, (const uint32_t*)d_seg, (const Rec*)d_recs, (const uint32_t*)d_pass, (const uint32_t*)d_P,
                           (const uint32_t*)d_X, (const uint32_t*)d_Y, (const uint32_t*)d_idx[cur], d_idx[cur ^ 1u]);
        cur ^= 1u;
        HIP_TRY(c, hipMemsetAsync(d_counter, 0, 4, st));
        hipLaunchKernelGGL(children_kernel, dim3(seg_blocks), dim3(256), 0, st, d_recs, base, nseg, base + nseg, (const uint32_t*)d_rank, (const uint32_t*)d_P, d_counter);
        hipLaunchKernelGGL(mark_kernel, dim3(pos_blocks), dim3(256), 0, st, n, d_seg, (const Rec*)d_recs);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(&nbig, d_counter, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (nbig > big_cap) return fail(c, PTX_ERR_STATE, "ptx_bvh_build: more large ranges than primitives allow");
        base += nseg;
        nseg = 2u * inner;
    }
    const double ms_levels = ms_since(t_start);
    // bvh.rs:188-213: the flat array is the pre-order of the tree
    for (size_t l = levels.size(); l-- > 0;)
        hipLaunchKernelGGL(size_kernel, dim3((levels[l].second + 255u) / 256u), dim3(256), 0, st, d_recs, levels[l].first, levels[l].second);
    for (size_t l = 0; l < levels.size(); l++)
        hipLaunchKernelGGL(preorder_kernel, dim3((levels[l].second + 255u) / 256u), dim3(256), 0, st, d_recs, levels[l].first, levels[l].second);
    const uint32_t nrec = base + nseg;
    hipLaunchKernelGGL(emit_kernel, dim3((nrec + 255u) / 256u), dim3(256), 0, st, (const Rec*)d_recs, nrec, d_nodes);
    HIP_TRY(c, hipGetLastError());
    static_assert(sizeof(FlatNode) == sizeof(ptx_bvh_node), "flat node layout");
    if (timing) HIP_TRY(c, hipStreamSynchronize(st));
    const double ms_flat = ms_since(t_start);
    HIP_TRY(c, hipMemcpyAsync(nodes_out, d_nodes, (size_t)nrec * sizeof(FlatNode), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(prim_indices_out, d_idx[cur], (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    *n_nodes_out = nrec;  // every record is a node: the root's subtree size
    if (timing) {
        const double ms_out = ms_since(t_start);
        mem.release();
        fprintf(stderr, "ptx_bvh_build: n %u, %zu levels, %u nodes; ms: alloc %.1f, boxes in %.1f, levels %.1f, flatten %.1f, arrays out %.1f, free %.1f\n",
                n, levels.size(), nrec, ms_alloc, ms_in - ms_alloc, ms_levels - ms_in, ms_flat - ms_levels, ms_out - ms_flat, ms_since(t_start) - ms_out);
    }
    return PTX_OK;
}

int ptx_post_process(ptx_ctx* c, const ptx_post_params* pp) {
    if (c) c->order_dirty = true;
    if (!c || !pp) return PTX_ERR_BAD_ARG;
    if (!c->accum) return fail(c, PTX_ERR_STATE, "ptx_post_process before ptx_resize");
    if (pp->width != c->w || pp->height != c->h) return fail(c, PTX_ERR_BAD_ARG, "post params width/height differ from the frame size");
    if (pp->effect_count > 8u) return fail(c, PTX_ERR_BAD_ARG, "more than 8 post effects");
    HIP_TRY(c, hipSetDevice(c->device));
    // halo the chain needs: 1 for the 3x3 / cross stencils, oil_radius for the oil painting
    uint32_t halo = 0;
    for (uint32_t i = 0; i < pp->effect_count; i++) {
        const uint32_t e = pp->effects[i];
        if (e == 4u || e == 7u || e == 8u) halo = halo < 1u ? 1u : halo;
        if (e == 5u) halo = halo < pp->oil_radius ? pp->oil_radius : halo;
    }
    if (halo > 0x3FFFFFFFu) return fail(c, PTX_ERR_BAD_ARG, "oil_radius too large");
    bool oil = false;
    for (uint32_t i = 0; i < pp->effect_count; i++) oil = oil || pp->effects[i] == 5u;
    const uint32_t rows_per_thread = oil && halo <= PP_MAX_HALO ? (uint32_t)PP_ROWS : 1u, tile_h = PP_TILE * rows_per_thread;
    const dim3 grid((c->w + PP_TILE - 1u) / PP_TILE, (c->h + tile_h - 1u) / tile_h), block(PP_TILE * PP_TILE);
    if (halo <= PP_MAX_HALO) {
        const size_t lds = (PP_TILE + 2u * halo) * (tile_h + 2u * halo) * sizeof(float4);
        if (rows_per_thread > 1u) hipLaunchKernelGGL(ptx_post_kernel_tile_rows, grid, block, lds, c->stream, c->accum, c->out, *pp, (int)halo);
        else hipLaunchKernelGGL(ptx_post_kernel_tile, grid, block, lds, c->stream, c->accum, c->out, *pp, (int)halo);
    } else {
        hipLaunchKernelGGL(ptx_post_kernel, grid, block, 0, c->stream, c->accum, c->out, *pp, (int)halo);
    }
    HIP_TRY(c, hipGetLastError());
    return PTX_OK;
}

const char* ptx_trace_kernel_name(const ptx_ctx* c) {
    if (!c || !c->have_scene) return "";
    if (c->last_kernel) return c->last_kernel;  // what the latest launch ran (the form depends on the launch's length)
    const bool frac = c->sc.has_fractals != 0u, lds = c->sc.lds_scene_bytes > 0u;
    const bool samples = c->split && c->split_upto > 2u;  // launches of every size run sample jobs (launch_trace)
    if (c->sched && !frac) return lds ? (samples ? "ptx_sched_kernel_lds_samples" : "ptx_sched_kernel_lds") : (samples ? "ptx_sched_kernel_samples" : "ptx_sched_kernel");
    if (c->sched) return lds ? "ptx_sched_kernel_lds_frac" : "ptx_sched_kernel_frac";
    return lds ? (frac ? "ptx_trace_kernel_lds_frac" : "ptx_trace_kernel_lds") : (frac ? "ptx_trace_kernel_frac" : "ptx_trace_kernel");
}

const char* ptx_last_launch_info(const ptx_ctx* c) { return c ? c->last_info.c_str() : ""; }

const char* ptx_launch_plan(ptx_ctx* c, uint32_t n) {
    if (!c || !c->have_scene || !c->accum || n == 0u) return "";
    const uint32_t total_blocks = ((c->tw + 15u) / 16u) * ((c->th + 15u) / 16u);
    const uint32_t n_blocks = total_blocks > c->part_rank ? (total_blocks - c->part_rank + c->part_world - 1u) / c->part_world : 0u;
    c->plan_info = describe_launch(resolve_launch(c, n, false, n_blocks, true));
    return c->plan_info.c_str();
}

int ptx_read_stats(ptx_ctx* c, ptx_stats* out) {
    if (!c || !out) return PTX_ERR_BAD_ARG;
    unsigned long long v[24];
    int rc = read_back(c, v, c->stats, sizeof(v), "ptx_read_stats");
    if (rc) return rc;
#ifdef PTX_PHASE_TIMING
    fprintf(stderr, "phase cycles: closest %llu shadeA %llu shadow %llu shadeB %llu loop-rest %llu finalize %llu\n", v[4], v[5], v[6], v[7], v[8], v[9]);
    if (v[16] | v[17] | v[18] | v[19] | v[20])
        fprintf(stderr, "pool cycles: top-up %llu sort %llu record-load %llu record-store+key %llu barrier-wait %llu\n", v[16], v[17], v[18], v[19], v[20]);
    if (v[13]) fprintf(stderr, "waves %llu: first start to last end %.3f ms, mean wave lifetime %.3f ms (lane 0 of each wave)\n", v[13], (double)(v[11] - ~v[10]) * 1e-5, (double)v[12] / (double)v[13] * 1e-5);
#endif
    out->paths = v[0]; out->bounces = v[1]; out->shadow_rays = v[2];
    return PTX_OK;
}

int ptx_kernel_time(ptx_ctx* c, float* total_ms, uint32_t* launches) {
    if (!c) return PTX_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = collect_timing(c);
    if (rc) return rc;
    if (total_ms) *total_ms = (float)c->acc_ms;
    if (launches) *launches = c->acc_launches;
    c->acc_ms = 0.0;
    c->acc_launches = 0;
    return PTX_OK;
}

int ptx_kernel_times(ptx_ctx* c, float* ms_out, uint32_t capacity, uint32_t* count) {
    if (!c || !count || (capacity && !ms_out)) return PTX_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = collect_timing(c);
    if (rc) return rc;
    const uint32_t n = (uint32_t)c->launch_ms.size();
    for (uint32_t i = 0; i < n && i < capacity; i++) ms_out[i] = c->launch_ms[i];
    *count = n;
    c->launch_ms.clear();
    return PTX_OK;
}

int ptx_kat_math(ptx_ctx* c, int op, uint32_t n, const float* a, const float* b, float* out) {
    if (!c || !a || !b || !out) return PTX_ERR_BAD_ARG;
    if (n == 0) return PTX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    size_t bytes = (size_t)n * 4;
    int rc = PTX_OK;
    if (hipMalloc((void**)&da, bytes) != hipSuccess || hipMalloc((void**)&db, bytes) != hipSuccess ||
        hipMalloc((void**)&dout, bytes) != hipSuccess) {
        rc = fail(c, PTX_ERR_OOM, "ptx_kat_math: hipMalloc");
    } else if (hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(db, b, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        rc = fail(c, PTX_ERR_HIP, "ptx_kat_math: copy in");
    } else {
        hipLaunchKernelGGL(ptx_kat_math_kernel, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, op, n, da, db, dout);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(c, PTX_ERR_HIP, "ptx_kat_math: kernel or copy out");
    }
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return rc;
}

int ptx_kat_exhaustive(ptx_ctx* c, int op, uint64_t* n_bad, uint32_t* first_bad) {
    if (!c || !n_bad || !first_bad || (op != 0 && op != 1)) return PTX_ERR_BAD_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* dn = nullptr;
    uint32_t* df = nullptr;
    int rc = PTX_OK;
    if (hipMalloc((void**)&dn, 8) != hipSuccess || hipMalloc((void**)&df, 4) != hipSuccess) {
        rc = fail(c, PTX_ERR_OOM, "ptx_kat_exhaustive: hipMalloc");
    } else {
        (void)hipMemsetAsync(dn, 0, 8, c->stream);
        (void)hipMemsetAsync(df, 0, 4, c->stream);
        for (uint32_t hi = 0; hi < 256u; hi++)
            hipLaunchKernelGGL(ptx_kat_exhaustive_kernel, dim3(1u << 16), dim3(256), 0, c->stream, op, hi, dn, df);
        unsigned long long nb = 0;
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&nb, dn, 8, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(first_bad, df, 4, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(c, PTX_ERR_HIP, "ptx_kat_exhaustive: kernel or copy out");
        *n_bad = nb;
    }
    if (dn) (void)hipFree(dn);
    if (df) (void)hipFree(df);
    return rc;
}

int ptx_kat_trace(ptx_ctx* c, const ptx_camera* cam, uint32_t n, const float* rays, const float* max_t, uint32_t* hit_id,
                  float* hit_t, uint32_t* shadow) {
    if (!c || !cam || !rays || !hit_id || !hit_t) return PTX_ERR_BAD_ARG;
    if (!c->have_scene) return fail(c, PTX_ERR_STATE, "ptx_kat_trace before ptx_upload_scene");
    if (n == 0) return PTX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    float *dr = nullptr, *dm_ = nullptr, *dt = nullptr;
    uint32_t *did = nullptr, *dsh = nullptr;
    int rc = PTX_OK;
    bool want_shadow = shadow && max_t;
    if (hipMalloc((void**)&dr, (size_t)n * 24) != hipSuccess || hipMalloc((void**)&dm_, (size_t)n * 4) != hipSuccess ||
        hipMalloc((void**)&dt, (size_t)n * 4) != hipSuccess || hipMalloc((void**)&did, (size_t)n * 4) != hipSuccess ||
        hipMalloc((void**)&dsh, (size_t)n * 4) != hipSuccess) {
        rc = fail(c, PTX_ERR_OOM, "ptx_kat_trace: hipMalloc");
    } else if (hipMemcpy(dr, rays, (size_t)n * 24, hipMemcpyHostToDevice) != hipSuccess ||
               (want_shadow && hipMemcpy(dm_, max_t, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess)) {
        rc = fail(c, PTX_ERR_HIP, "ptx_kat_trace: copy in");
    } else {
        size_t lds = (size_t)PTX_WAVES_PER_BLOCK * c->sc.stack_entries * 64u * sizeof(StackEntry);
        const uint32_t blocks = (n + PTX_BLOCK - 1u) / PTX_BLOCK;
        StackEntry* spill = nullptr;
        // Same traversal variant as ptx_trace picks for this scene (the records themselves stay in global memory here).
        if (c->sc.lds_scene_bytes > 0u)
            hipLaunchKernelGGL(ptx_kat_trace_kernel_loop, dim3(blocks), dim3(PTX_BLOCK), lds, c->stream, c->sc, *cam, n, dr,
                               want_shadow ? dm_ : nullptr, did, dt, want_shadow ? dsh : nullptr);
        else if ((rc = ensure_spill(c, (size_t)blocks * PTX_WAVES_PER_BLOCK, &spill)) == PTX_OK)
            hipLaunchKernelGGL(ptx_kat_trace_kernel, dim3(blocks), dim3(PTX_BLOCK), lds, c->stream, c->sc, spill, *cam, n, dr,
                               want_shadow ? dm_ : nullptr, did, dt, want_shadow ? dsh : nullptr);
        if (rc == PTX_OK &&
            (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(hit_id, did, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hit_t, dt, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
             (want_shadow && hipMemcpy(shadow, dsh, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)))
            rc = fail(c, PTX_ERR_HIP, "ptx_kat_trace: kernel or copy out");
    }
    if (dr) (void)hipFree(dr);
    if (dm_) (void)hipFree(dm_);
    if (dt) (void)hipFree(dt);
    if (did) (void)hipFree(did);
    if (dsh) (void)hipFree(dsh);
    return rc;
}

}  // extern "C"
