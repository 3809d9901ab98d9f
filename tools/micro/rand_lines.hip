// rand_lines.hip -- how fast does MI355X deliver SCATTERED 128-byte lines, and does it depend on the size of the buffer they are scattered over?
// k_ext_trace_pk, k_p3_lanes and k_p1_assemble all sit at 2-2.5 TB/s of line fetches with most wave cycles waiting for memory: is that the memory system's rate
// for lines nobody else wants, or the address translation of a 30 GB arena?  Every lane reads one 16-byte piece of a pseudo-random line (one load in flight per
// lane and iteration, `depth` independent loads per lane), over footprints of 1 / 4 / 16 / 32 / 64 GB.
// Run: ./rand_lines            (prints GB/s of 128-byte lines per footprint and depth)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ void __launch_bounds__(256) k_rand(const uint4 *p, unsigned long long lines, int iters, unsigned *sink)
{
    unsigned long long s = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const unsigned long long line = (s >> 20) % lines;
            v[d] = p[line * 8ull + ((s >> 8) & 7ull)];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++) acc += v[d].x ^ v[d].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int nCU = pr.multiProcessorCount;
    unsigned *sink; CK(hipMalloc(&sink, 4));
    const double gbs[] = {1, 4, 16, 32, 64, 128};
    for (double gb : gbs) {
        const size_t bytes = (size_t)(gb * (1ull << 30));
        uint4 *p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("%.0f GB: hipMalloc failed\n", gb); continue; }
        CK(hipMemset(p, 1, bytes));
        const unsigned long long lines = bytes / 128;
        for (int depth : {1, 4}) {
            for (int wavesPerSimd : {2, 8}) {
                const int blocks = nCU * wavesPerSimd, iters = 2000 / depth;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    if (depth == 1) hipLaunchKernelGGL(k_rand<1>, dim3(blocks), dim3(256), 0, 0, p, lines, iters, sink);
                    else hipLaunchKernelGGL(k_rand<4>, dim3(blocks), dim3(256), 0, 0, p, lines, iters, sink);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                }
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                const double n = (double)blocks * 256.0 * iters * depth;
                printf("footprint %5.0f GB  loads in flight per lane %d  waves per SIMD %d : %7.1f G lines/s = %6.2f TB/s of 128-byte lines  (%.2f ms)\n", gb, depth, wavesPerSimd, n / ms / 1e6, n * 128.0 / ms / 1e9, ms);
            }
        }
        CK(hipFree(p));
    }
    return 0;
}
