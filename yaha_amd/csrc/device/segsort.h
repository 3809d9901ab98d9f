// segsort.h -- A2: the hits of one (read, strand) sorted inside one workgroup.
//
// The join of A2 (findFragmentsSort, QueryMatch.c:52-121 with the heap of QueryHeap.inl:70-134) is the multiset of hits in ascending
// (diagonal, query offset) order.  k_expand_hits (seed.h) writes the hits of one (read, strand) -- one segment, a few thousand 64-bit keys --
// in ascending query offset, so a STABLE sort on the 32 diagonal bits [15, 47) of the key finishes the job.  A segment of up to 15 872 keys
// fits in a workgroup's registers and LDS: one read and one write of HBM per key instead of the library's digit passes over global memory.
// Longer segments (repeat-rich reads) are first cut by diagonal into buckets that fit (k_seg_split); a bucket that still does not fit is cut again by its OWN range of
// diagonals (k_seg_split_range), level by level, until every piece fits or holds a single diagonal (then it is in order as it stands: the cuts are stable).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "wgsort.h"

#ifndef YD_SORT_TOP
#define YD_SORT_TOP 31                 // hits a thread of the largest class (512 threads)
#endif
// 512 threads x 31 hits: 62 KB of LDS for the diagonals -- the most a workgroup may have -- and two registers a hit
#define YD_SEGSORT_MAX (512u * YD_SORT_TOP)

// One workgroup sorts one segment (wgsort.h: four 8-bit passes over LDS, this file's only dependency).  ATOMIC: the ranking of a pass by LDS atomics (the default) or by
// ballots (the fallback that needs nothing the manuals do not promise; stage_seed.hip switches to it for good when the order check of k_frag_scan_build ever fails).
// (four waves a SIMD: two workgroups of 512 threads a CU -- 128 registers, which two registers a hit leave room for)
template <unsigned BS, unsigned IPT, bool ATOMIC>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(4))) k_seg_sort(const unsigned long long *in, unsigned long long *out, const uint32_t *segB,
                                                                                         const uint32_t *segE, const uint32_t *list, unsigned int *outOfOrder)
{
    YD_HIGH_PRIO();
    using Sort = WgSort<BS, IPT, ATOMIC>;
    __shared__ typename Sort::Storage st;
    const uint32_t seg = list[blockIdx.x];                                   // the segments of this launch's size class (k_seg_classify)
    const uint32_t b = segB[seg], len = segE[seg] - b;
    // the (read, strand) bits [47, 64): the same in every key of a segment (a scalar)
    const uint32_t rsHi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(in[b] >> 32)) & 0xFFFF8000u;
    // HBM -> registers (wave-striped: coalesced) -> four digit passes over LDS -> HBM (striped over the workgroup: coalesced); in == out is fine: the workgroup has read
    // its whole segment before it stores.  What moves through LDS is the 32-bit diagonal with the 15-bit query offset as payload (6 bytes a hit).
    uint32_t dg[IPT], qo[IPT];
    const uint32_t w0 = (threadIdx.x >> 6) * (64u * IPT) + (threadIdx.x & 63u);
#pragma unroll
    for (unsigned k = 0; k < IPT; k++) {
        const uint32_t idx = w0 + k * 64u;
        const unsigned long long key = idx < len ? in[b + idx] : 0ull;
        dg[k] = (uint32_t)(key >> 15); qo[k] = (uint32_t)(key & 0x7FFFull);
    }
    bool bad = false;
    Sort::sort(dg, qo, len, st, [&](uint32_t pos, uint32_t d, uint32_t q) { out[b + pos] = ((unsigned long long)(rsHi | (d >> 17)) << 32) | (unsigned long long)((d << 15) | q); },
        bad);
    if (ATOMIC && bad) atomicOr(outOfOrder, 1u);
}

// Size classes of the segments [segB[s], segE[s]): class c (0..YD_SEG_NCLASS-1) = at most hi[c] hits -> lists[c] (the workgroup sorts above: one launch per class
// over exactly its segments; launching every class over all segments and letting the wrong ones leave cost 0.1 ms per 10 000 workgroups of 128 KB of LDS); longer
// ones = the last class -> lists[YD_SEG_NCLASS] (bigB / bigE, when given: their begin / end, empty for every other segment).  Twelve workgroup
// shapes (a sort costs what its shape holds, not what the segment has: with four shapes a segment filled its workgroup to 75 % on average).
// counts[0..YD_SEG_NCLASS]; one atomic per wave and class.
#define YD_SEG_NCLASS 12
struct SegClassHi { uint32_t hi[YD_SEG_NCLASS]; };
__global__ void __launch_bounds__(256) k_seg_classify(const uint32_t *segB, const uint32_t *segE, uint32_t nSeg, SegClassHi H,
                                                      uint32_t *lists, uint32_t *bigB, uint32_t *bigE, unsigned int *counts)
{
    YD_HIGH_PRIO();
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; const int lane = (int)(threadIdx.x & 63u);
    uint32_t b = 0, e = 0; int cls = -1;
    if (s < nSeg) {
        b = segB[s]; e = segE[s]; const uint32_t len = e - b;
        if (len) { cls = YD_SEG_NCLASS;
#pragma unroll
            for (int c = YD_SEG_NCLASS - 1; c >= 0; c--) if (len <= H.hi[c]) cls = c; }
        if (bigB) { bigB[s] = cls == YD_SEG_NCLASS ? b : 0u; bigE[s] = cls == YD_SEG_NCLASS ? e : 0u; }
    }
#pragma unroll
    for (int c = 0; c <= YD_SEG_NCLASS; c++) {
        const unsigned long long m = __ballot(cls == c);
        if (m == 0ull) continue;                                             // wave-uniform
        unsigned base = 0; const int first = __builtin_ctzll(m);
        if (lane == first) base = atomicAdd(&counts[c], (unsigned)__builtin_popcountll(m));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, first);
        if (cls == c) lists[(size_t)c * nSeg + base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = s;
    }
}

// A long segment (more than 15 872 hits: 44 % of the hits of a repeat-rich batch) is cut by DIAGONAL into up to 16 buckets that fit the workgroup sort: one
// stable counting pass by a workgroup per segment -- thread t owns a contiguous run of the segment's hits (count per bucket, exclusive scan over (bucket,
// thread), scatter in order) -- instead of the library's four digit passes.  Bucket = the diagonal's top bits (monotone, so sorting the buckets one by one
// sorts the segment); sub-segment sb * 16 + k = bucket k of the sb-th long segment.  A bucket that still exceeds the workgroup sort is cut again (k_seg_split_range).
#define YD_SPLIT_NB 16
__global__ void __launch_bounds__(1024) k_seg_split(const unsigned long long *in, unsigned long long *out, const uint32_t *segB, const uint32_t *segE, const uint32_t *bigList,
                                                    int diagBits, uint32_t *subB, uint32_t *subE)
{
    YD_HIGH_PRIO();
    // Wave w owns the w-th sixteenth of the segment and walks it 64 hits at a time (coalesced); the stable order is (wave, group, lane).
    __shared__ uint32_t sCnt[16][YD_SPLIT_NB];                                // [wave][bucket]: counts, then running write positions (a wave's 64 atomics go to 16 banks; as
                                                                              // [bucket][wave] they went to two: 85 % of the kernel's LDS cycles were bank conflicts)
    const uint32_t seg = bigList[blockIdx.x], b = segB[seg], len = segE[seg] - b, t = threadIdx.x, lane = t & 63u, w = t >> 6;
    int lg = 0; while (lg < 4 && ((len + 6143u) / 6144u) > (1u << lg)) lg++;                     // ~6 k hits a bucket on average, 16 buckets at most
    const uint32_t nb = 1u << lg; const int sh = diagBits > lg ? diagBits - lg : 0;
    const uint32_t per = ((len + 15u) / 16u + 63u) & ~63u, k0 = min(len, w * per), k1 = min(len, k0 + per);   // the wave's range, whole groups of 64
    if (t < YD_SPLIT_NB * 16u) sCnt[t & 15u][t >> 4] = 0u;
    __syncthreads();
    // pass 1: hits per (bucket, wave)
    for (uint32_t k = k0 + lane; k < k1; k += 64u) { const uint32_t d = (uint32_t)(in[b + k] >> 15); atomicAdd(&sCnt[w][min(d >> sh, nb - 1u)], 1u); }
    __syncthreads();
    // exclusive scan in (bucket, wave) order: 256 entries, one wave does it (4 per lane)
    if (w == 0) {
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = sum; sum += sCnt[(4u * lane + (uint32_t)k) & 15u][lane >> 2]; }      // entry e = bucket * 16 + wave, e = 4 lane + k
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += x; }
        const uint32_t excl = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) sCnt[(4u * lane + (uint32_t)k) & 15u][lane >> 2] = excl + v[k];
    }
    __syncthreads();
    if (t < YD_SPLIT_NB) { subB[blockIdx.x * YD_SPLIT_NB + t] = b + sCnt[0][t]; subE[blockIdx.x * YD_SPLIT_NB + t] = t + 1u < YD_SPLIT_NB ? b + sCnt[0][t + 1u] : b + len; }
    __syncthreads();
    // pass 2: every group of 64 is split by bucket with four ballots (the lanes of one bucket keep their order), written behind what the wave has already
    // written into that bucket
    for (uint32_t g = k0; g < k1; g += 64u) {
        const uint32_t k = g + lane; const bool live = k < k1;
        const unsigned long long key = live ? in[b + k] : 0ull;
        const uint32_t bk = live ? min((uint32_t)(key >> 15) >> sh, nb - 1u) : 0u;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 4; bit++) { const unsigned long long m = __ballot((bk >> bit) & 1u); peers &= ((bk >> bit) & 1u) ? m : ~m; }
        const uint32_t rank = (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1ull));
        if (live) out[b + sCnt[w][bk] + rank] = key;
        __builtin_amdgcn_wave_barrier();
        if (live && rank == 0u) sCnt[w][bk] += (uint32_t)__builtin_popcountll(peers);          // the bucket's first lane moves the wave's position on
        __builtin_amdgcn_wave_barrier();
    }
}
// A piece that is still too long for the workgroup sort after k_seg_split (a read inside a tandem repeat: tens of thousands of hits on neighbouring diagonals): cut
// again into 16 buckets over the piece's OWN range of diagonals [mn, mx] -- bucket = (d - mn) * 16 / (mx - mn + 1), monotone; the first and the last bucket are never
// empty, so every bucket is shorter than the piece, and the range a bucket spans shrinks sixteen-fold a level: at most eight levels for 32-bit diagonals.  A piece of ONE
// diagonal (mn == mx) is in order already -- k_expand_hits wrote ascending query offsets and every cut so far was stable: it is copied to `final` as it stands and
// gets no sub-pieces.  in -> out for the cut pieces (the caller sorts them out -> final or cuts them again), sub-piece sb * 16 + k = bucket k of the sb-th listed piece.
__global__ void __launch_bounds__(1024) k_seg_split_range(const unsigned long long *in, unsigned long long *out, unsigned long long *final, const uint32_t *segB,
    const uint32_t *segE, const uint32_t *list,
                                                          uint32_t *subB, uint32_t *subE)
{
    YD_HIGH_PRIO();
    __shared__ uint32_t sCnt[16][YD_SPLIT_NB]; __shared__ uint32_t sMin, sMax;
    const uint32_t seg = list[blockIdx.x], b = segB[seg], len = segE[seg] - b, t = threadIdx.x, lane = t & 63u, w = t >> 6;
    if (t < YD_SPLIT_NB * 16u) sCnt[t & 15u][t >> 4] = 0u;
    if (t == 0) { sMin = 0xFFFFFFFFu; sMax = 0u; }
    __syncthreads();
    { uint32_t mn = 0xFFFFFFFFu, mx = 0u;
      for (uint32_t k = t; k < len; k += 1024u) { const uint32_t d = (uint32_t)(in[b + k] >> 15); mn = min(mn, d); mx = max(mx, d); }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, 64)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64)); }
      if (lane == 0u) { atomicMin(&sMin, mn); atomicMax(&sMax, mx); } }
    __syncthreads();
    const uint32_t mn = sMin, mx = sMax;
    if (mn == mx) {                                                          // (workgroup-uniform)
        if (final != in) for (uint32_t k = t; k < len; k += 1024u) final[b + k] = in[b + k];
        if (t < YD_SPLIT_NB) { subB[blockIdx.x * YD_SPLIT_NB + t] = b; subE[blockIdx.x * YD_SPLIT_NB + t] = b; }
        return;
    }
    const unsigned long long range = (unsigned long long)(mx - mn) + 1ull;
    auto bucket = [&](unsigned long long key) { return (uint32_t)((((unsigned long long)((uint32_t)(key >> 15) - mn)) * (unsigned long long)YD_SPLIT_NB) / range); };
    const uint32_t per = ((len + 15u) / 16u + 63u) & ~63u, k0 = min(len, w * per), k1 = min(len, k0 + per);   // the wave's range, whole groups of 64 (as k_seg_split)
    for (uint32_t k = k0 + lane; k < k1; k += 64u) atomicAdd(&sCnt[w][bucket(in[b + k])], 1u);
    __syncthreads();
    if (w == 0) {
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = sum; sum += sCnt[(4u * lane + (uint32_t)k) & 15u][lane >> 2]; }      // entry e = bucket * 16 + wave, e = 4 lane + k
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += x; }
        const uint32_t excl = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) sCnt[(4u * lane + (uint32_t)k) & 15u][lane >> 2] = excl + v[k];
    }
    __syncthreads();
    if (t < YD_SPLIT_NB) { subB[blockIdx.x * YD_SPLIT_NB + t] = b + sCnt[0][t]; subE[blockIdx.x * YD_SPLIT_NB + t] = t + 1u < YD_SPLIT_NB ? b + sCnt[0][t + 1u] : b + len; }
    __syncthreads();
    for (uint32_t g = k0; g < k1; g += 64u) {
        const uint32_t k = g + lane; const bool live = k < k1;
        const unsigned long long key = live ? in[b + k] : 0ull;
        const uint32_t bk = live ? bucket(key) : 0u;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 4; bit++) { const unsigned long long m = __ballot((bk >> bit) & 1u); peers &= ((bk >> bit) & 1u) ? m : ~m; }
        const uint32_t rank = (uint32_t)__builtin_popcountll(peers & ((1ull << lane) - 1ull));
        if (live) out[b + sCnt[w][bk] + rank] = key;
        __builtin_amdgcn_wave_barrier();
        if (live && rank == 0u) sCnt[w][bk] += (uint32_t)__builtin_popcountll(peers);
        __builtin_amdgcn_wave_barrier();
    }
}
