// prims.hip -- the exclusive sums and orderings of the batch layout (scan.h: one launch a sum, two an ordering; their work words clean themselves up).
// The reference handles one read at a time and has no such step; every stage of the batched path uses these between its count and fill launches.
#include "ctx.h"
#include "scan.h"
static_assert(kBucketMax == YD_BKT_MAX, "ctx.h and scan.h disagree on the buckets of an ordering");

template <class T> static int ownScanT(DevBuf &scanState, std::string &err, unsigned int *failed, hipStream_t st, const T *in, T *out, uint32_t n)
{
    if (n == 0) return 0;
    const size_t need = scanStateBytes(n);
    if (scanState.cap < need) {                                              // (zeroed when it is made; every launch leaves it zero)
        if (scanState.ensure(std::max<size_t>(need, 1u << 16))) { err = "hipMalloc(scan state)"; return YGPU_ENOMEM; }
        const hipError_t e0 = hipMemsetAsync(scanState.p, 0, scanState.cap, st);
        if (e0 != hipSuccess) { err = std::string("hipMemsetAsync(scan state): ") + hipGetErrorString(e0); return YGPU_ENODEV; }
    }
    hipLaunchKernelGGL((k_scan_excl<T>), dim3(scanTiles(n, (int)sizeof(T))), dim3(YD_SCAN_BS), 0, st, in, out, n, (unsigned long long *)scanState.p, failed);
    const hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { err = std::string("launch of k_scan_excl failed: ") + hipGetErrorString(e_); return YGPU_ENODEV; }
    return 0;
}
int ydScan32(DevBuf &scanState, std::string &err, unsigned int *failed, hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n)
{ return ownScanT<uint32_t>(scanState, err, failed, st, in, out, n); }
int ydScan64(DevBuf &scanState, std::string &err, unsigned int *failed, hipStream_t st, const unsigned long long *in, unsigned long long *out, uint32_t n)
{ return ownScanT<unsigned long long>(scanState, err, failed, st, in, out, n); }
size_t ydBucketWorkBytes() { return bucketWorkBytes(); }

// order[] = the items' values grouped by bucket((key - sub) >> shift), ascending; vals == nullptr: the values are the items' indices + valBase
int bucketOrder(ygpu_ctx *ctx, const uint32_t *keys, const uint32_t *vals, uint32_t valBase, uint32_t n, uint32_t sub, int shift, uint32_t nb,
                uint32_t *outVals, hipStream_t st)
{
    if (n == 0) return 0;
    nb = std::min<uint32_t>(std::max<uint32_t>(nb, 1u), YD_BKT_MAX);
    if (!ctx->bucketWork.p) {
        if (ctx->bucketWork.ensure(bucketWorkBytes())) { ctx->err = "hipMalloc(bucket work)"; return YGPU_ENOMEM; }
        HIPCHK(hipMemsetAsync(ctx->bucketWork.p, 0, ctx->bucketWork.cap, st));
    }
    const unsigned grid = (unsigned)((n + YD_BKT_TILE - 1) / YD_BKT_TILE);
    hipLaunchKernelGGL(k_bucket_count, dim3(grid), dim3(YD_BKT_BS), 0, st, keys, n, sub, shift, nb, ctx->bucketWork.as<unsigned int>());
    hipLaunchKernelGGL(k_bucket_scatter, dim3(grid), dim3(YD_BKT_BS), 0, st, keys, vals, valBase, n, sub, shift, nb, ctx->bucketWork.as<unsigned int>(),
                       outVals, (uint32_t *)nullptr);
    const hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { ctx->err = std::string("launch of k_bucket_count / k_bucket_scatter failed: ") + hipGetErrorString(e_); return YGPU_ENODEV; }
    return 0;
}

// ---- YGPU_CHECK_STATE=1: "the work words clean themselves up" as a checked invariant ----------------------------------------------------------------------------
// The look-back words of the sums (scanState: tile words, ticket, done counter) and the work words of the orderings (bucketWork) are zeroed ONCE, when they are
// made, and every launch is trusted to leave them zero.  A launch that does not -- or a buffer that was made without being zeroed (round 5: a presized context's)
// -- gives wrong sums in the NEXT call and nothing notices: wrong SAM, exit code 0.  With the switch on, every ygpu_run / ygpu_postfilter ends with a pass over
// those words on the device and fails when one is not zero.  (The reference has no such state: one read at a time, Query.c:306-497.)
__global__ void __launch_bounds__(256) k_check_zero(const uint32_t *p, uint32_t nWords, unsigned int *bad /* [0] count, [1] first index + 1 */)
{
    uint32_t cnt = 0, first = 0xFFFFFFFFu;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nWords; i += gridDim.x * blockDim.x) if (p[i] != 0u) { cnt++; first = min(first, i); }
    if (cnt) { atomicAdd(&bad[0], cnt); atomicMin(&bad[1], first); }
}
bool ydCheckStateOn() { static const bool on = getenv("YGPU_CHECK_STATE") != nullptr && atoi(getenv("YGPU_CHECK_STATE")) != 0; return on; }
// 0: every word of every buffer is zero; YGPU_EINTERNAL with `err` naming the first buffer and word otherwise.  `scratch`: 8 bytes of device memory.
int ydCheckZero(hipStream_t st, std::string &err, unsigned int *scratch, const DevBuf *const *bufs, const char *const *names, int n, const char *when)
{
    for (int k = 0; k < n; k++) {
        if (!bufs[k]->p || bufs[k]->cap < 4) continue;
        const uint32_t nWords = (uint32_t)std::min<size_t>(bufs[k]->cap / 4, 0xFFFFFFF0u);
        const unsigned int init[2] = {0u, 0xFFFFFFFFu}; unsigned int got[2] = {0u, 0u};
        if (hipMemcpyAsync(scratch, init, 8, hipMemcpyHostToDevice, st) != hipSuccess) { err = "state check: copy failed"; return YGPU_ENODEV; }
        hipLaunchKernelGGL(k_check_zero, dim3((unsigned)std::min<uint32_t>(1024u, (nWords + 255u) / 256u)), dim3(256), 0, st, (const uint32_t *)bufs[k]->p, nWords, scratch);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(got, scratch, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            err = "state check: launch failed"; (void)hipGetLastError(); return YGPU_ENODEV;
        }
        if (got[0]) {
            char m[256]; snprintf(m, sizeof m, "YGPU_CHECK_STATE: %u of the %u work words of %s are not zero %s (the first: word %u): the next call on them would be wrong",
                                  got[0], nWords, names[k], when, got[1]);
            err = m; return YGPU_EINTERNAL;
        }
    }
    return 0;
}
