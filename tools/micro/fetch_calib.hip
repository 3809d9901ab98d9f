// fetch_calib.hip -- what does FETCH_SIZE mean for k_ext_rows' access shape?  (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".)
// k_ext_rows reads its query codes and reference nibbles ONE BYTE PER LANE PER ROW, every lane walking its own sequential stream.  Three kernels read the
// same N bytes exactly once each:
//   k_wide         16 B per lane, coalesced                      -> the documented case: N = 2 x FETCH_SIZE
//   k_byte_sparse  one byte per lane per iteration, every lane its own 2 KB stream, ONE wave per SIMD on a quarter of the CUs: the lines in use
//                  (64 lanes x 128 B per wave) stay in L1/L2, so every line crosses the fabric once -> N = f x FETCH_SIZE gives f for byte loads
//   k_byte_dense   the same at k_ext_rows' occupancy (3 waves per SIMD on every CU, two streams per lane): lines are evicted between a lane's visits;
//                  FETCH_SIZE x f / N = the re-fetch factor of the access shape
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./fetch_calib      (prints the byte counts to compare with the counter)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_wide(const uint4 *p, size_t n16, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
// every thread owns stream t of `len` bytes and reads it byte by byte; a second stream (the "reference" of k_ext_rows) half a byte per iteration
template <bool TWO>
__global__ void k_byte(const uint8_t *p, const uint8_t *p2, size_t len, size_t streams, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < streams; t += (size_t)gridDim.x * blockDim.x) {
        const uint8_t *s = p + t * len, *s2 = p2 + t * (len / 2);
        for (size_t i = 0; i < len; i++) {
            acc += s[i];
            if (TWO) acc ^= s2[i >> 1];
            // ~600 dependent ALU operations per row in k_ext_rows: keep some distance between the loads
            for (int k = 0; k < 16; k++) acc = acc * 1664525u + 1013904223u;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const size_t N = 3ull << 30;                                   // 3 GiB: far beyond the 256 MiB Infinity Cache
    uint8_t *buf; unsigned *sink; CK(hipMalloc(&buf, N + N / 2)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, N + N / 2));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_wide, dim3(cus * 8), dim3(256), 0, 0, (const uint4 *)buf, N / 16, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"kernel\": \"k_wide\", \"bytes_read_once\": %zu, \"ms\": %.3f}\n", N, ms);
    const size_t len = 2048;
    {   // sparse: 64 blocks of 256 threads (one wave per SIMD on 64 CUs); 16 384 streams in flight, 2 MB of lines in use chip-wide
        const size_t streams = (N / 8) / len;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_byte<false>, dim3(64), dim3(256), 0, 0, buf, buf + N, len, streams, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"k_byte<false> sparse\", \"bytes_read_once\": %zu, \"ms\": %.3f}\n", streams * len, ms);
    }
    {   // dense: k_ext_rows' occupancy and both of its streams
        const size_t streams = (N / 2) / len;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_byte<true>, dim3(cus * 3), dim3(256), 0, 0, buf, buf + N, len, streams, sink); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"k_byte<true> dense\", \"bytes_read_once\": %zu, \"ms\": %.3f}\n", streams * len + streams * (len / 2), ms);
    }
    return 0;
}
