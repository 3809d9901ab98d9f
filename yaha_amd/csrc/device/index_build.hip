// index_build.hip -- the hash index (reference Index.c:49-335) built on the MI355X: count -> scan -> fill -> order -> sample.
//
// The reference's builder is three single-threaded passes over the genome (count the k-mers, fill the reference-offset array ROA in ascending
// offset order per k-mer, Floyd-sample k-mers with more than maxHits occurrences).  For an hg18-scale genome (3.1 Gbp, 4^15 = 1 G k-mers, 16.7 GB
// file) that is minutes of pointer-chasing; here every base offset is one unit of work:
//   k_ix_count   thread per 16 consecutive offsets: rolling 2-bit hash of the 4-bit reference (Index.c:32-43), one atomicAdd per valid k-mer
//   [exclusive scan of the 4^k counters -> startingOffs]
//   k_ix_fill    the same walk; slot = startingOffs[h] + atomicAdd(cursor[h]) -- the order inside a k-mer's list is whatever the atomics gave
//   k_ix_order*  every list is put into ascending offset order, which is the ONLY order the reference can produce (its fill pass scans the genome
//                left to right, Index.c:201-229), so the result is independent of the atomics: lists of <= 32 entries are insertion-sorted by one
//                thread (they arrive almost sorted), up to 8 192 entries by one workgroup in LDS (bitonic), longer ones by a device radix sort each
//   sampling     k-mers with more than maxHits occurrences keep a Floyd sample drawn with the default-seeded Marsaglia generator, consumed in
//                k-mer order (Index.c:271-315, Math.c:304-343): inherently sequential, but it concerns a handful of k-mers -- their lists go to
//                the host, the samples come back, and one gather pass compacts ROA (k_ix_compact).
// k-mers never span sequences and skip any window holding a code > 3 (Index.c:98-127); any skip distance (walkKmers states which starts the reference's scan
// visits when -S > 1).  The host builder (host/formats.cpp) remains for machines without a GPU.  Output: the complete file image, byte-identical to the reference's.
#include <hip/hip_runtime.h>
#include "scan.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <ctime>
#include <cstdlib>
#include "../host/yaha_host.h"

namespace {
#define IXCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return false; } } while (0)
#define IX_PER_THREAD 16
#define IX_SMALL 32u
#define IX_BLOCK 8192u

// skip > 1: firstBad[s] = offset of the first code > 3 of sequence s (0xFFFFFFFF: none)
struct SeqTab { const uint32_t *start, *len; uint32_t n; int skip; const uint32_t *firstBad; };

__device__ __forceinline__ uint32_t nibAt(const uint8_t *b, uint64_t off) { const uint8_t v = b[off >> 1]; return (off & 1) ? (uint32_t)(v & 15u) : (uint32_t)(v >> 4); }

// Calls visit(hash, offset) for every k-mer start in [p0, p0 + IX_PER_THREAD) that the reference's scan visits (Index.c:98-127).  Skip distance 1: every start
// whose window is clean.  Skip distance S > 1: the scan steps S from the sequence's start until a window holds a code > 3, and restarts behind that run of codes
// at the next ABSOLUTE multiple of S (`((bad + S-1) / S) * S`) -- so a start is visited iff its window is clean and it is (start of sequence) + jS before the
// sequence's first bad code, or a multiple of S after it.
template <class Visit> __device__ __forceinline__ void walkKmers(const uint8_t *bases, SeqTab T, int k, uint64_t p0, uint64_t nOffsets, Visit visit)
{
    if (p0 >= nOffsets) return;
    // the sequence that holds p0 (sequences are ascending; a thread's span may cross into the next one)
    uint32_t lo = 0, hi = T.n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)T.start[mid] + T.len[mid] <= p0) lo = mid + 1; else hi = mid; }
    uint32_t si = lo;
    const uint32_t mask = 0xFFFFFFFFu >> (32 - 2 * k);
    uint32_t h = 0; int good = 0;                       // good = valid codes accumulated in h, inside the current sequence
    uint64_t p = p0;                                    // next offset whose code is to be shifted in
    uint64_t segEnd = 0, segStart = 0; bool inSeq = false;
    const uint64_t pEnd = p0 + IX_PER_THREAD + (uint64_t)k - 1;      // the last k-mer start of this thread needs codes up to here
    for (; p < pEnd && p < nOffsets; p++) {
        while (si < T.n && p >= (uint64_t)T.start[si] + T.len[si]) { si++; inSeq = false; }
        if (si >= T.n) break;
        if (!inSeq) { segStart = T.start[si]; segEnd = segStart + T.len[si]; inSeq = true; good = 0; }
        if (p < segStart) { good = 0; continue; }       // padding between sequences
        const uint32_t c = nibAt(bases, p);
        if (c > 3u) { good = 0; continue; }
        h = ((h << 2) | c) & mask; good++;
        if (good >= k) {
            const uint64_t s = p + 1 - (uint64_t)k;
            if (s >= p0 && s < p0 + IX_PER_THREAD) {
                bool ok = true;
                if (T.skip > 1) { const uint64_t fb = T.firstBad[si]; ok = (s + (uint64_t)k <= fb) ? ((s - segStart) % (uint64_t)T.skip == 0) : (s % (uint64_t)T.skip == 0); }
                if (ok) visit(h, (uint32_t)s);
            }
        }
    }
    (void)segEnd;
}

// skip distance > 1: the first code > 3 of every sequence (one atomicMin per thread and sequence that has one in the thread's span)
__global__ void k_ix_first_bad(const uint8_t *bases, SeqTab T, uint64_t nOffsets, uint32_t *firstBad)
{
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * IX_PER_THREAD;
    if (p0 >= nOffsets) return;
    uint32_t lo = 0, hi = T.n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)T.start[mid] + T.len[mid] <= p0) lo = mid + 1; else hi = mid; }
    uint32_t si = lo; uint32_t best = 0xFFFFFFFFu;
    for (uint64_t p = p0; p < p0 + IX_PER_THREAD && p < nOffsets; p++) {
        while (si < T.n && p >= (uint64_t)T.start[si] + T.len[si]) { if (best != 0xFFFFFFFFu) { atomicMin(&firstBad[si], best); best = 0xFFFFFFFFu; } si++; }
        if (si >= T.n) break;
        if (p < T.start[si]) continue;
        if (best == 0xFFFFFFFFu && nibAt(bases, p) > 3u) best = (uint32_t)p;
    }
    if (best != 0xFFFFFFFFu && si < T.n) atomicMin(&firstBad[si], best);
}
__global__ void k_ix_count(const uint8_t *bases, SeqTab T, int k, uint64_t nOffsets, uint32_t *counts)
{
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * IX_PER_THREAD;
    walkKmers(bases, T, k, p0, nOffsets, [&](uint32_t h, uint32_t) { atomicAdd(&counts[h], 1u); });
}
__global__ void k_ix_fill(const uint8_t *bases, SeqTab T, int k, uint64_t nOffsets, const uint32_t *so, uint32_t *cursor, uint32_t *roa)
{
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * IX_PER_THREAD;
    walkKmers(bases, T, k, p0, nOffsets, [&](uint32_t h, uint32_t off) { roa[so[h] + atomicAdd(&cursor[h], 1u)] = off; });
}
// lists of 2 .. IX_SMALL entries: in-place insertion sort by one thread; longer ones are listed for the workgroup / device sorts
__global__ void k_ix_order_small(const uint32_t *so, uint64_t nKmers, uint32_t *roa, uint32_t *bigList, unsigned int *nBig, uint32_t bigCap, uint32_t *overList,
    unsigned int *nOver, uint32_t overCap, uint32_t maxHits)
{
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= nKmers) return;
    const uint32_t b = so[h], n = so[h + 1] - b;
    if (n > maxHits) { const unsigned s = atomicAdd(nOver, 1u); if (s < overCap) overList[s] = (uint32_t)h; }
    if (n < 2u) return;
    if (n > IX_SMALL) { const unsigned s = atomicAdd(nBig, 1u); if (s < bigCap) bigList[s] = (uint32_t)h; return; }
    uint32_t *a = roa + b;
    for (uint32_t i = 1; i < n; i++) { const uint32_t v = a[i]; uint32_t j = i; while (j > 0 && a[j - 1] > v) { a[j] = a[j - 1]; j--; } a[j] = v; }
}
// lists of IX_SMALL+1 .. IX_BLOCK entries: bitonic sort in LDS, one workgroup per list
__global__ void __launch_bounds__(256) k_ix_order_block(const uint32_t *so, const uint32_t *bigList, uint32_t nBig, uint32_t *roa, uint32_t hugeMin)
{
    __shared__ uint32_t s[IX_BLOCK];
    for (uint32_t li = blockIdx.x; li < nBig; li += gridDim.x) {
        const uint32_t h = bigList[li], b = so[h], n = so[h + 1] - b;
        if (n > hugeMin) continue;                                            // left to the multi-workgroup sort below
        uint32_t m = 64; while (m < n) m <<= 1;
        for (uint32_t i = threadIdx.x; i < m; i += 256) s[i] = i < n ? roa[b + i] : 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t size = 2; size <= m; size <<= 1)
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t t = threadIdx.x; t < m / 2; t += 256) {
                    const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
                    const bool up = (i & size) == 0; const uint32_t x = s[i], y = s[j];
                    if ((x > y) == up) { s[i] = y; s[j] = x; }
                }
                __syncthreads();
            }
        for (uint32_t i = threadIdx.x; i < n; i += 256) roa[b + i] = s[i];
        __syncthreads();
    }
}
// A list beyond a workgroup's LDS (a satellite k-mer: tens of thousands to millions of offsets): a bitonic network over a copy padded to a power of two with
// 0xFFFFFFFF -- the strides of 4 096 and below of a merge step inside workgroups of 8 192 elements (LDS), the wider ones one launch each.  A few dozen launches
// for a million entries; the hot path never comes here (once per index, a handful of lists).
__global__ void __launch_bounds__(256) k_ix_bitonic_global(uint32_t *a, uint32_t m, uint32_t size, uint32_t stride)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m / 2) return;
    const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
    const bool up = (i & size) == 0; const uint32_t x = a[i], y = a[j];
    if ((x > y) == up) { a[i] = y; a[j] = x; }
}
// first == true: the whole network up to merge size IX_BLOCK inside the workgroup's 8 192 elements; else the strides IX_BLOCK / 2 .. 1 of merge step `size`
__global__ void __launch_bounds__(256) k_ix_bitonic_local(uint32_t *a, uint32_t m, uint32_t size, bool first)
{
    __shared__ uint32_t s[IX_BLOCK];
    const uint32_t base = blockIdx.x * IX_BLOCK, cnt = min(IX_BLOCK, m - base);      // (m is a power of two: cnt = IX_BLOCK, or m itself when m < IX_BLOCK)
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) s[i] = a[base + i];
    __syncthreads();
    for (uint32_t sz = first ? 2u : size; sz <= (first ? min(cnt, size) : size); sz <<= 1) {
        for (uint32_t stride = min(sz >> 1, cnt >> 1); stride > 0; stride >>= 1) {
            for (uint32_t t = threadIdx.x; t < cnt / 2; t += 256) {
                const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = ((base + i) & sz) == 0; const uint32_t x = s[i], y = s[j];
                if ((x > y) == up) { s[i] = y; s[j] = x; }
            }
            __syncthreads();
        }
        if (!first) break;
    }
    for (uint32_t i = threadIdx.x; i < cnt; i += 256) a[base + i] = s[i];
}
__global__ void k_ix_pad_copy(const uint32_t *src, uint32_t n, uint32_t *dst, uint32_t m)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[i] = i < n ? src[i] : 0xFFFFFFFFu;
}
__global__ void k_ix_clamp(const uint32_t *so, uint64_t nKmers, uint32_t maxHits, uint32_t *cnt2)
{
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h < nKmers) { const uint32_t n = so[h + 1] - so[h]; cnt2[h] = n > maxHits ? maxHits : n; }
    if (h == nKmers) cnt2[h] = 0;
}
// ROA' = the lists of all k-mers that were not sampled, at their new places.  A thread per k-mer: at -L 15 nearly every list has 0..3 entries and neighbouring
// threads write neighbouring words; a list of more than 32 entries is copied by the whole wave instead (the lanes that own such lists take turns).
__global__ void k_ix_compact(const uint32_t *so, const uint32_t *so2, uint64_t nKmers, uint32_t maxHits, const uint32_t *roa, uint32_t *roa2)
{
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    uint32_t b = 0, n = 0, d = 0;
    if (h < nKmers) { b = so[h]; n = so[h + 1] - b; d = so2[h]; if (n > maxHits) n = 0; }      // sampled lists are written from the host's samples
    if (n <= 32u) for (uint32_t i = 0; i < n; i++) roa2[d + i] = roa[b + i];
    unsigned long long longLists = __ballot(n > 32u);
    while (longLists) {
        const int src = __builtin_ctzll(longLists); longLists &= longLists - 1ull;
        const uint32_t bb = (uint32_t)__shfl((int)b, src, 64), nn = (uint32_t)__shfl((int)n, src, 64), dd = (uint32_t)__shfl((int)d, src, 64);
        for (uint32_t i = (uint32_t)lane; i < nn; i += 64u) roa2[dd + i] = roa[bb + i];
    }
}
// The lists of the over-represented k-mers, packed one after the other (a wave per list), and the way back for their samples: one transfer each way per
// group of lists instead of two blocking copies per k-mer (a small -H on a large genome has millions of such lists).
__global__ void k_ix_pack(const uint32_t *roa, const uint32_t *desc /* per list: begin, length, packed offset (64-bit as two words) */, uint32_t nLists, uint32_t *packed)
{
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; const int lane = (int)(threadIdx.x & 63);
    if (w >= nLists) return;
    const uint32_t b = desc[4 * w], n = desc[4 * w + 1]; const uint64_t o = (uint64_t)desc[4 * w + 2] | ((uint64_t)desc[4 * w + 3] << 32);
    for (uint32_t i = (uint32_t)lane; i < n; i += 64u) packed[o + i] = roa[b + i];
}
__global__ void k_ix_unpack(const uint32_t *samples, const uint32_t *dst /* per list: its place in ROA' */, uint32_t nLists, uint32_t maxHits, uint32_t *roa2)
{
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; const int lane = (int)(threadIdx.x & 63);
    if (w >= nLists) return;
    const uint32_t d = dst[w]; const uint64_t o = (uint64_t)w * maxHits;
    for (uint32_t i = (uint32_t)lane; i < maxHits; i += 64u) roa2[d + i] = samples[o + i];
}

// (begin, end, new begin) of the listed k-mers in one array: the host loops over a handful of long lists without a copy per list
__global__ void k_ix_gather(const uint32_t *so, const uint32_t *so2, const uint32_t *list, uint32_t n, uint32_t minLen, uint32_t *out, unsigned int *nOut)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = list[i], b = so[h], e = so[h + 1];
    if (e - b <= minLen) return;
    const unsigned s = atomicAdd(nOut, 1u);
    out[4 * (size_t)s] = h; out[4 * (size_t)s + 1] = b; out[4 * (size_t)s + 2] = e; out[4 * (size_t)s + 3] = so2 ? so2[h] : 0u;
}

struct Buf { void *p = nullptr; ~Buf() { if (p) hipFree(p); } void release() { if (p) hipFree(p); p = nullptr; } template <class T> T *as() { return (T *)p; } };
}  // namespace

namespace yaha {
static double wallNow() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int visibleDevices() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

// Builds the complete index file image {-1, wordLen, maxHits, total} + startingOffs[4^k + 1] + ROA[total] on HIP device `device`.
bool buildIndexDevice(int device, const Genome &g, int wordLen, int skipDist, int maxHits, IndexImage &image, FILE *log, std::string &err)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { err = "no such HIP device"; return false; }
    const bool timing = getenv("YAHA_TIMING") != nullptr; double tLast = wallNow();
    auto lap = [&](const char *what) { if (timing && log) { hipDeviceSynchronize(); const double t = wallNow();
        fprintf(log, "[yaha]   index on GPU: %-28s %7.2f s\n", what, t - tLast); tLast = t; } };
    IXCHK(hipSetDevice(device));
    const uint64_t HT = 1ull << (2 * wordLen), nOffsets = g.nBaseBytes * 2;
    std::vector<uint32_t> st, ln; for (auto &s : g.seqs) if ((int64_t)s.length >= wordLen) { st.push_back(s.start); ln.push_back(s.length); }
    // a sequence shorter than a k-mer holds none; the table keeps only the others (ascending starts)
    const uint32_t nSeq = (uint32_t)st.size();
    Buf dFirstBad, dBases, dStart, dLen, dCnt, dSO, dCur, dROA, dTemp, dBig, dOver, dN, dSO2, dROA2;
    IXCHK(hipMalloc(&dBases.p, g.nBaseBytes + 64)); IXCHK(hipMemcpy(dBases.p, g.bases, g.nBaseBytes, hipMemcpyHostToDevice));
    IXCHK(hipMalloc(&dStart.p, 4ull * (nSeq + 1))); IXCHK(hipMalloc(&dLen.p, 4ull * (nSeq + 1)));
    if (nSeq) { IXCHK(hipMemcpy(dStart.p, st.data(), 4ull * nSeq, hipMemcpyHostToDevice)); IXCHK(hipMemcpy(dLen.p, ln.data(), 4ull * nSeq, hipMemcpyHostToDevice)); }
    SeqTab T; T.start = dStart.as<uint32_t>(); T.len = dLen.as<uint32_t>(); T.n = nSeq; T.skip = skipDist; T.firstBad = nullptr;
    lap("reference to HBM");
    const uint64_t nThreads0 = (nOffsets + IX_PER_THREAD - 1) / IX_PER_THREAD; const unsigned grid0 = (unsigned)((nThreads0 + 255) / 256);
    if (skipDist > 1) {
        IXCHK(hipMalloc(&dFirstBad.p, 4ull * (nSeq + 1))); IXCHK(hipMemset(dFirstBad.p, 0xFF, 4ull * (nSeq + 1)));
        if (grid0) hipLaunchKernelGGL(k_ix_first_bad, dim3(grid0), dim3(256), 0, 0, dBases.as<uint8_t>(), T, nOffsets, dFirstBad.as<uint32_t>());
        IXCHK(hipGetLastError());
        T.firstBad = dFirstBad.as<uint32_t>();
    }
    IXCHK(hipMalloc(&dCnt.p, 4ull * (HT + 1))); IXCHK(hipMemset(dCnt.p, 0, 4ull * (HT + 1)));
    const uint64_t nThreads = (nOffsets + IX_PER_THREAD - 1) / IX_PER_THREAD; const unsigned grid = (unsigned)((nThreads + 255) / 256);
    if (grid) hipLaunchKernelGGL(k_ix_count, dim3(grid), dim3(256), 0, 0, dBases.as<uint8_t>(), T, wordLen, nOffsets, dCnt.as<uint32_t>());
    IXCHK(hipGetLastError());
    // startingOffs = exclusive prefix sums (4^k + 1 entries, the last one = total)
    IXCHK(hipMalloc(&dSO.p, 4ull * (HT + 1)));
    // (scan.h: single pass, decoupled look-back; its state words clean themselves up, so the second sum below needs no memset)
    const size_t tb = scanStateBytes(HT + 1) + 64; IXCHK(hipMalloc(&dTemp.p, tb)); IXCHK(hipMemset(dTemp.p, 0, tb));
    unsigned int *scanFail = (unsigned int *)((char *)dTemp.p + scanStateBytes(HT + 1));
    hipLaunchKernelGGL((k_scan_excl<uint32_t>), dim3(scanTiles(HT + 1)), dim3(YD_SCAN_BS), 0, 0, (const uint32_t *)dCnt.as<uint32_t>(), dSO.as<uint32_t>(), (uint32_t)(HT + 1),
        dTemp.as<unsigned long long>(), scanFail);
    IXCHK(hipGetLastError());
    uint32_t total = 0; IXCHK(hipMemcpy(&total, dSO.as<uint32_t>() + HT, 4, hipMemcpyDeviceToHost));
    lap("count + scan");
    // fill (the counters become the cursors)
    IXCHK(hipMemset(dCnt.p, 0, 4ull * (HT + 1)));
    IXCHK(hipMalloc(&dROA.p, 4ull * ((uint64_t)total + 16)));
    if (grid) hipLaunchKernelGGL(k_ix_fill, dim3(grid), dim3(256), 0, 0, dBases.as<uint8_t>(), T, wordLen, nOffsets, dSO.as<uint32_t>(), dCnt.as<uint32_t>(), dROA.as<uint32_t>());
    IXCHK(hipGetLastError());
    lap("fill");
    // order every list
    uint32_t bigCap = 1u << 22, overCap = 1u << 20; unsigned int two[2] = {0, 0};
    IXCHK(hipMalloc(&dBig.p, 4ull * bigCap)); IXCHK(hipMalloc(&dOver.p, 4ull * overCap)); IXCHK(hipMalloc(&dN.p, 8));
    for (int attempt = 0;; attempt++) {
        IXCHK(hipMemset(dN.p, 0, 8));
        hipLaunchKernelGGL(k_ix_order_small, dim3((unsigned)((HT + 255) / 256)), dim3(256), 0, 0, dSO.as<uint32_t>(), HT, dROA.as<uint32_t>(), dBig.as<uint32_t>(),
            dN.as<unsigned int>(), bigCap,
                           dOver.as<uint32_t>(), dN.as<unsigned int>() + 1, overCap, (uint32_t)maxHits);
        IXCHK(hipGetLastError()); IXCHK(hipMemcpy(two, dN.p, 8, hipMemcpyDeviceToHost));
        if (two[0] <= bigCap && two[1] <= overCap) break;
        if (attempt) { err = "index build: list of long k-mer lists overflows"; return false; }
        // (the small lists are sorted already; a second pass over them is harmless) regrow the lists and redo
        if (two[0] > bigCap) { bigCap = two[0] + 1024; hipFree(dBig.p); dBig.p = nullptr; IXCHK(hipMalloc(&dBig.p, 4ull * bigCap)); }
        if (two[1] > overCap) { overCap = two[1] + 1024; hipFree(dOver.p); dOver.p = nullptr; IXCHK(hipMalloc(&dOver.p, 4ull * overCap)); }
    }
    const uint32_t nBig = two[0], nOver = two[1];
    std::vector<uint32_t> hSOpair;                       // startingOffs of the long lists
    if (nBig) {
        // (YAHA_IX_HUGE_MIN: a test hook -- lists above that many entries take the multi-workgroup sort, 8 192 = what a workgroup's LDS holds otherwise)
        uint32_t hugeMin = IX_BLOCK; if (const char *e = getenv("YAHA_IX_HUGE_MIN")) { const long v = atol(e); if (v >= (long)IX_SMALL && v < (long)IX_BLOCK) hugeMin = (uint32_t)v;
            }
        hipLaunchKernelGGL(k_ix_order_block, dim3(std::min<uint32_t>(nBig, 4096u)), dim3(256), 0, 0, dSO.as<uint32_t>(), dBig.as<uint32_t>(), nBig, dROA.as<uint32_t>(), hugeMin);
        IXCHK(hipGetLastError());
        // the few lists beyond a workgroup's LDS: a bitonic network over a padded copy each (k_ix_bitonic_*)
        Buf dHuge; IXCHK(hipMalloc(&dHuge.p, 16ull * nBig + 16)); IXCHK(hipMemset(dN.p, 0, 4));
        hipLaunchKernelGGL(k_ix_gather, dim3((nBig + 255) / 256), dim3(256), 0, 0, dSO.as<uint32_t>(), (const uint32_t *)nullptr, dBig.as<uint32_t>(), nBig, hugeMin,
            dHuge.as<uint32_t>(), dN.as<unsigned int>());
        IXCHK(hipGetLastError());
        unsigned int nHuge = 0; IXCHK(hipMemcpy(&nHuge, dN.p, 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> huge(4ull * nHuge); if (nHuge) IXCHK(hipMemcpy(huge.data(), dHuge.p, 16ull * nHuge, hipMemcpyDeviceToHost));
        Buf dAlt; size_t altCap = 0;
        for (unsigned int k = 0; k < nHuge; k++) {
            const uint32_t b0 = huge[4ull * k + 1], n = huge[4ull * k + 2] - b0;
            uint32_t m = 64; while (m < n) m <<= 1;
            if (m > altCap) { if (dAlt.p) hipFree(dAlt.p); dAlt.p = nullptr; altCap = m; IXCHK(hipMalloc(&dAlt.p, 4ull * altCap)); }
            uint32_t *seg = dROA.as<uint32_t>() + b0, *alt = dAlt.as<uint32_t>();
            hipLaunchKernelGGL(k_ix_pad_copy, dim3((m + 255) / 256), dim3(256), 0, 0, (const uint32_t *)seg, n, alt, m);
            const unsigned lblocks = (m + IX_BLOCK - 1) / IX_BLOCK;
            hipLaunchKernelGGL(k_ix_bitonic_local, dim3(lblocks), dim3(256), 0, 0, alt, m, (uint32_t)IX_BLOCK, true);
            for (uint32_t size = 2 * IX_BLOCK; size <= m && size != 0; size <<= 1) {
                for (uint32_t stride = size >> 1; stride >= IX_BLOCK; stride >>= 1) hipLaunchKernelGGL(k_ix_bitonic_global, dim3((m / 2 + 255) / 256), dim3(256), 0, 0, alt, m,
                    size, stride);
                hipLaunchKernelGGL(k_ix_bitonic_local, dim3(lblocks), dim3(256), 0, 0, alt, m, size, false);
            }
            IXCHK(hipGetLastError());
            IXCHK(hipMemcpyAsync(seg, alt, 4ull * n, hipMemcpyDeviceToDevice, 0));
        }
    }
    IXCHK(hipDeviceSynchronize());
    lap("order the lists");
    if (log) fprintf(log, "Randomly Sampling hits for %d-mers that occur more than %d times in the reference.\n", wordLen, maxHits);
    uint32_t newTotal = total; const uint32_t *finalSO = dSO.as<uint32_t>(); const uint32_t *finalROA = dROA.as<uint32_t>();
    if (nOver) {
        // sampling pass (Index.c:271-315): the over-represented k-mers in ascending order, one generator for all of them
        std::vector<uint32_t> over(nOver); IXCHK(hipMemcpy(over.data(), dOver.p, 4ull * nOver, hipMemcpyDeviceToHost));
        std::sort(over.begin(), over.end());
        IXCHK(hipMalloc(&dSO2.p, 4ull * (HT + 1)));
        hipLaunchKernelGGL(k_ix_clamp, dim3((unsigned)((HT + 1 + 255) / 256)), dim3(256), 0, 0, dSO.as<uint32_t>(), HT, (uint32_t)maxHits, dCnt.as<uint32_t>());
        IXCHK(hipGetLastError());
        hipLaunchKernelGGL((k_scan_excl<uint32_t>), dim3(scanTiles(HT + 1)), dim3(YD_SCAN_BS), 0, 0, (const uint32_t *)dCnt.as<uint32_t>(), dSO2.as<uint32_t>(), (uint32_t)(HT + 1),
            dTemp.as<unsigned long long>(), scanFail);
        IXCHK(hipGetLastError());
        IXCHK(hipMemcpy(&newTotal, dSO2.as<uint32_t>() + HT, 4, hipMemcpyDeviceToHost));
        IXCHK(hipMalloc(&dROA2.p, 4ull * ((uint64_t)newTotal + 16)));
        hipLaunchKernelGGL(k_ix_compact, dim3((unsigned)((HT + 255) / 256)), dim3(256), 0, 0, dSO.as<uint32_t>(), dSO2.as<uint32_t>(), HT, (uint32_t)maxHits, dROA.as<uint32_t>(),
            dROA2.as<uint32_t>());
        IXCHK(hipGetLastError());
        RandState rs; randInitDefault(rs);
        Buf dOv; IXCHK(hipMalloc(&dOv.p, 16ull * nOver + 16)); IXCHK(hipMemset(dN.p, 0, 4));
        IXCHK(hipMemcpy(dOver.p, over.data(), 4ull * nOver, hipMemcpyHostToDevice));        // ascending now
        hipLaunchKernelGGL(k_ix_gather, dim3((nOver + 255) / 256), dim3(256), 0, 0, dSO.as<uint32_t>(), dSO2.as<uint32_t>(), dOver.as<uint32_t>(), nOver, 0u, dOv.as<uint32_t>(),
            dN.as<unsigned int>());
        IXCHK(hipGetLastError());
        std::vector<uint32_t> ov(4ull * nOver); IXCHK(hipMemcpy(ov.data(), dOv.p, 16ull * nOver, hipMemcpyDeviceToHost));
        std::vector<size_t> idx(nOver); for (size_t k = 0; k < nOver; k++) idx[k] = k;
        // the gather's atomics shuffled them: back to k-mer order (the generator's order)
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ov[4 * a] < ov[4 * b]; });
        // groups of lists of at most 64 M entries: pack on the device, one copy down, sample on the host in k-mer order (the generator is sequential by
        // definition), one copy of the samples up, scatter on the device
        const uint64_t groupCap = 64ull << 20;
        std::vector<uint32_t> desc, dst, packed, samples; Buf dDesc, dDst, dPacked, dSamples;
        size_t g0 = 0;
        while (g0 < idx.size()) {
            size_t g1 = g0; uint64_t tot = 0; desc.clear(); dst.clear();
            while (g1 < idx.size()) {
                const size_t k = idx[g1]; const uint32_t b0 = ov[4 * k + 1], n = ov[4 * k + 2] - b0;
                if (g1 > g0 && (tot + n > groupCap || (uint64_t)(g1 - g0 + 1) * (uint64_t)maxHits > groupCap)) break;
                desc.push_back(b0); desc.push_back(n); desc.push_back((uint32_t)tot); desc.push_back((uint32_t)(tot >> 32)); dst.push_back(ov[4 * k + 3]); tot += n; g1++;
            }
            const uint32_t nl = (uint32_t)(g1 - g0);
            dDesc.release(); dDst.release(); dPacked.release(); dSamples.release();
            IXCHK(hipMalloc(&dDesc.p, 16ull * nl)); IXCHK(hipMalloc(&dDst.p, 4ull * nl)); IXCHK(hipMalloc(&dPacked.p, 4ull * tot + 16));
                IXCHK(hipMalloc(&dSamples.p, 4ull * (uint64_t)nl * (uint64_t)maxHits + 16));
            IXCHK(hipMemcpy(dDesc.p, desc.data(), 16ull * nl, hipMemcpyHostToDevice)); IXCHK(hipMemcpy(dDst.p, dst.data(), 4ull * nl, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_ix_pack, dim3((unsigned)(((uint64_t)nl * 64 + 255) / 256)), dim3(256), 0, 0, dROA.as<uint32_t>(), dDesc.as<uint32_t>(), nl,
                dPacked.as<uint32_t>());
            IXCHK(hipGetLastError());
            packed.resize(tot); IXCHK(hipMemcpy(packed.data(), dPacked.p, 4ull * tot, hipMemcpyDeviceToHost));
            samples.resize((size_t)nl * (size_t)maxHits);
            for (uint32_t l = 0; l < nl; l++) {
                const uint64_t o = (uint64_t)desc[4 * l + 2] | ((uint64_t)desc[4 * l + 3] << 32);
                randSample(rs, packed.data() + o, (int)desc[4 * l + 1], samples.data() + (size_t)l * (size_t)maxHits, maxHits);
            }
            IXCHK(hipMemcpy(dSamples.p, samples.data(), 4ull * samples.size(), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_ix_unpack, dim3((unsigned)(((uint64_t)nl * 64 + 255) / 256)), dim3(256), 0, 0, dSamples.as<uint32_t>(), dDst.as<uint32_t>(), nl, (uint32_t)maxHits,
                dROA2.as<uint32_t>());
            IXCHK(hipGetLastError());
            g0 = g1;
        }
        IXCHK(hipDeviceSynchronize());
        finalSO = dSO2.as<uint32_t>(); finalROA = dROA2.as<uint32_t>();
    }
    if (log) fprintf(log, "%u %d-mers had more than %d hits.\n", nOver, wordLen, maxHits);
    lap("sampling + compaction");
    if (!image.alloc(4 + HT + 1 + (uint64_t)newTotal)) { err = "Insufficient memory to build the index."; return false; }
    image[0] = 0xFFFFFFFFu; image[1] = (uint32_t)wordLen; image[2] = (uint32_t)maxHits; image[3] = newTotal;
    const bool pinned = hipHostRegister(image.p, image.bytes, hipHostRegisterDefault) == hipSuccess;      // DMA straight into the image instead of staging through a bounce buffer
    if (!pinned) (void)hipGetLastError();
    IXCHK(hipMemcpy(image.p + 4, finalSO, 4ull * (HT + 1), hipMemcpyDeviceToHost));
    if (newTotal) IXCHK(hipMemcpy(image.p + 4 + HT + 1, finalROA, 4ull * newTotal, hipMemcpyDeviceToHost));
    if (pinned) hipHostUnregister(image.p);
    lap("image to host memory");
    return true;
}
}  // namespace yaha
