// How many device atomics per second do R counters at a given stride serve?  4096 waves, lane 0 of each adds K times to counter
// (wave % R).  One address serves about 100 M/s on MI355X; this measures whether R addresses serve R times as many.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void probe(unsigned* ctr, unsigned ranges, unsigned stride_words, unsigned k, unsigned by_block, unsigned* sink) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned r = (by_block ? blockIdx.x : wave) % ranges;
    unsigned acc = 0;
    if ((threadIdx.x & 63u) == 0u)
        for (unsigned i = 0; i < k; i++) acc += atomicAdd(&ctr[(size_t)r * stride_words], 64u);
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main() {
    unsigned *ctr, *sink;
    CK(hipMalloc((void**)&ctr, 64 << 20));
    CK(hipMalloc((void**)&sink, 256));
    CK(hipMemset(ctr, 0, 64 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = 1024, k = 64;
    for (unsigned by_block = 0; by_block < 2; by_block++)
        for (unsigned stride : {16u, 64u, 1024u, 16384u})
            for (unsigned ranges : {1u, 2u, 4u, 8u, 16u, 32u}) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, ctr, ranges, stride, k, by_block, sink);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                printf("%s stride %6u B ranges %2u: %.3f ms  %.1f M atomics/s\n", by_block ? "by block" : "by wave ", stride * 4, ranges, best,
                       grid * 4.0 * k / best / 1e3);
            }
    return 0;
}
