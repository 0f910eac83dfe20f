// Issue cost of vector instructions on one SIMD of gfx950 with 1 to 8 waves resident: cycles per instruction of one wave
// (s_memtime around an unrolled loop) and per SIMD (that divided by the waves on the SIMD).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/issue_probe tools/probes/issue_probe.hip
// One workgroup per compute unit (it asks for 96 KB of LDS), 64 x 4 x W lanes: W waves on every SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define ITERS 65536

template <int KIND>
__global__ void probe(float* out, unsigned long long* cycles, float seed) {
    extern __shared__ float lds[];
    float a0 = seed, a1 = seed + 1.0f, a2 = seed + 2.0f, a3 = seed + 3.0f, a4 = seed + 4.0f, a5 = seed + 5.0f, a6 = seed + 6.0f, a7 = seed + 7.0f;
    float b = 0.999f, c = 0.001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < ITERS; i++) {
        if (KIND == 0) {  // eight independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (KIND == 1) {  // one dependent chain of eight v_fma_f32
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
        } else if (KIND == 2) {  // four independent v_pk_fma_f32 (eight fma)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (KIND == 3) {  // a dependent chain of four v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(p0) : "v"(pb), "v"(pc));
        } else if (KIND == 4) {  // Horner steps with the coefficient as a literal: four v_fmaak_f32 in a chain
            asm volatile("v_fmaak_f32 %0, %0, %1, 0x3d9021bb\n v_fmaak_f32 %0, %0, %1, 0x3def251a\n v_fmaak_f32 %0, %0, %1, 0xbdfe5d4f\n v_fmaak_f32 %0, %0, %1, 0x3e11e9bf\n"
                         : "+v"(a0) : "v"(b));
        } else if (KIND == 5) {  // the packed Horner step as the compiler writes it: s_mov_b32 of the coefficient + v_pk_fma_f32, four times
            asm volatile("s_mov_b32 s20, 0x3d9021bb\n v_pk_fma_f32 %0, %0, %1, s[20:21] op_sel_hi:[1,1,0]\n s_mov_b32 s20, 0x3def251a\n v_pk_fma_f32 %0, %0, %1, s[20:21] op_sel_hi:[1,1,0]\n"
                         "s_mov_b32 s20, 0xbdfe5d4f\n v_pk_fma_f32 %0, %0, %1, s[20:21] op_sel_hi:[1,1,0]\n s_mov_b32 s20, 0x3e11e9bf\n v_pk_fma_f32 %0, %0, %1, s[20:21] op_sel_hi:[1,1,0]\n"
                         : "+v"(p0) : "v"(pb) : "s20", "s21");
        } else if (KIND == 6) {  // compare + select through an SGPR pair, with the two wait states the hardware wants: four pairs in a chain
            asm volatile(REP8("v_cmp_lt_f32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n") : "+v"(a0) : "v"(b), "v"(c) : "s20", "s21");
        } else if (KIND == 7) {  // the same selects as v_max_f32 (no SGPR round trip)
            asm volatile(REP8("v_max_f32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
        } else if (KIND == 8) {  // quarter-rate: eight independent v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (KIND == 9) {  // v_rcp_f32 followed by its consumer (the wait state between a transcendental and its use), a chain of four pairs
            asm volatile("v_rcp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_rcp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_rcp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_rcp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if (KIND == 10) {  // eight scalar instructions
            asm volatile(REP8("s_add_u32 s20, s20, 1\n") : : : "s20", "scc");
        } else if (KIND == 11) {  // a scalar and a vector instruction alternating (independent), four pairs
            asm volatile("s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %4, %5\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 s20, s20, 1\n v_fma_f32 %2, %2, %4, %5\n s_add_u32 s20, s20, 1\n v_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "scc");
        } else if (KIND == 12) {  // exec-mask branch idiom that is not taken: s_and_saveexec + s_cbranch_execz + restore, around one v_fma, four times
            asm volatile(REP8("s_and_saveexec_b64 s[20:21], exec\n s_cbranch_execz 1f\n v_fma_f32 %0, %0, %1, %2\n1:\n s_or_b64 exec, exec, s[20:21]\n") : "+v"(a0) : "v"(b), "v"(c) : "s20", "s21");
        } else if (KIND == 13) {  // LDS read with its wait, then a dependent fma: four in a chain
            float v;
            asm volatile("ds_read_b32 %1, %3\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2\n ds_read_b32 %1, %3\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2\n"
                         "ds_read_b32 %1, %3\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2\n ds_read_b32 %1, %3\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0), "=&v"(v) : "v"(c), "v"((unsigned)(threadIdx.x * 4u)) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + lds[threadIdx.x & 63];
}

struct Kind { const char* name; int per_iter; };
static const Kind kinds[] = {
    {"v_fma_f32 x8 independent", 8}, {"v_fma_f32 x8 dependent chain", 8}, {"v_pk_fma_f32 x4 independent", 4}, {"v_pk_fma_f32 x4 dependent chain", 4},
    {"v_fmaak_f32 x4 chain (literal coefficient)", 4}, {"(s_mov_b32 + v_pk_fma_f32) x4 chain", 8}, {"(v_cmp + s_nop 1 + v_cndmask) x8 chain", 24},
    {"v_max_f32 x8 chain", 8}, {"v_rcp_f32 x8 independent", 8}, {"(v_rcp_f32 + s_nop 0 + v_fma_f32) x4 chain", 12}, {"s_add_u32 x8", 8},
    {"(s_add_u32 + v_fma_f32) x4", 8}, {"(saveexec + cbranch_execz + v_fma + s_or exec) x8", 32}, {"(ds_read_b32 + wait + v_fma_f32) x4 chain", 12}};

template <int KIND>
static void run(int waves, float* out, unsigned long long* cyc, int n_cu) {
    const size_t lds = 96 << 10;
    (void)hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(probe<KIND>, dim3(n_cu), dim3(256 * waves), lds, 0, out, cyc, 1.0f);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part; the shader clock is read from the device properties
    printf("  %d waves/SIMD: %8llu ticks", waves, h);
    fflush(stdout);
    (void)h;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate / 1e6;
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc((void**)&out, (size_t)n_cu * 1024 * 4);
    (void)hipMalloc((void**)&cyc, 8);
    printf("%s, %d CUs, %.2f GHz shader clock (device property)\n", prop.name, n_cu, clock_ghz);
    // calibrate the tick: a kernel of known duration by events
    for (int k = 0; k < (int)(sizeof(kinds) / sizeof(kinds[0])); k++) {
        printf("%s (%d instructions per iteration, %d iterations)\n", kinds[k].name, kinds[k].per_iter, ITERS);
        for (int w : {1, 2, 3, 4}) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            // warm + measure wall time by events as well: ticks may not be shader cycles
            switch (k) {
#define CASE(K) case K: run<K>(w, out, cyc, n_cu); (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(probe<K>, dim3(n_cu), dim3(256 * w), 96 << 10, 0, out, cyc, 1.0f); (void)hipEventRecord(e1, 0); break;
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
#undef CASE
            }
            (void)hipEventSynchronize(e1);
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)ITERS * kinds[k].per_iter;
            // wall-clock cycles per instruction of one wave (launch overhead of a few microseconds included), and per SIMD
            const double cyc_wave = ms * 1e-3 * clock_ghz * 1e9 / instr;
            printf("   %.3f ms -> %.2f cycles per instruction per wave, %.2f per SIMD\n", ms, cyc_wave, cyc_wave / w);
        }
    }
    return 0;
}
