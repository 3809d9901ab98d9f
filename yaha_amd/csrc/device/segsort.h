// segsort.h -- A2: the hits of one (read, strand) sorted inside one workgroup.
//
// The join of A2 (findFragmentsSort, QueryMatch.c:52-121 with the heap of QueryHeap.inl:70-134) is the multiset of hits in ascending
// (diagonal, query offset) order.  k_expand_hits (seed.h) writes the hits of one (read, strand) -- one segment, a few thousand 64-bit keys --
// in ascending query offset, so a STABLE sort on the 32 diagonal bits [15, 47) of the key finishes the job.  A segment of up to 16 384 keys
// fits in a workgroup's registers and LDS: one read and one write of HBM per key instead of the library's digit passes over global memory.
// Longer segments (repeat-rich reads) are left to hipcub::DeviceSegmentedRadixSort through begin/end arrays that are empty for all others.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/block/block_load.hpp>
#include <rocprim/block/block_store.hpp>
#include <rocprim/block/block_radix_sort.hpp>
#include <cstdint>

#define YD_SEGSORT_MAX 16384u

template <unsigned BS, unsigned IPT>
__global__ void __launch_bounds__(BS) k_seg_sort(const unsigned long long *in, unsigned long long *out, const uint32_t *segOff, const uint32_t *list)
{
    using Load = rocprim::block_load<unsigned long long, BS, IPT, rocprim::block_load_method::block_load_transpose>;
    using Store = rocprim::block_store<unsigned long long, BS, IPT, rocprim::block_store_method::block_store_transpose>;
    using Sort = rocprim::block_radix_sort<uint32_t, BS, IPT, uint16_t>;
    __shared__ union { typename Load::storage_type load; typename Store::storage_type store; typename Sort::storage_type sort; } st;
    const uint32_t seg = list[blockIdx.x];                                   // the segments of this launch's size class (k_seg_classify)
    const uint32_t b = segOff[seg], len = segOff[seg + 1] - b;
    unsigned long long keys[IPT];
    // blocked arrangement = the order the hits were written in; the padding keys sort last and, the sort being stable, stay behind real keys with the same bits
    Load().load(in + b, keys, len, ~0ull, st.load);
    __syncthreads();
    // what moves through the sort's LDS passes is the 32-bit diagonal with the 15-bit query offset as payload (6 bytes a hit instead of 8); the 17 (read, strand)
    // bits are the segment's number
    uint32_t dg[IPT]; uint16_t qo[IPT];
#pragma unroll
    for (unsigned k = 0; k < IPT; k++) { dg[k] = (uint32_t)(keys[k] >> 15); qo[k] = (uint16_t)(keys[k] & 0x7FFFull); }
    Sort().sort(dg, qo, st.sort, 0, 32);
    __syncthreads();
    const unsigned long long rs = (unsigned long long)seg << 47;
#pragma unroll
    for (unsigned k = 0; k < IPT; k++) keys[k] = rs | ((unsigned long long)dg[k] << 15) | (unsigned long long)qo[k];
    Store().store(out + b, keys, len, st.store);
}

// Size classes of the segments: class c (0..3) = at most hi[c] hits -> lists[c] (the workgroup sorts above: one launch per class over exactly its segments;
// launching every class over all segments and letting the wrong ones leave cost 0.1 ms per 10 000 workgroups of 128 KB of LDS); longer ones = class 4: begin/end
// offsets for the library's segmented sort, every other segment empty there.  counts[0..4]; one atomic per wave and class.
__global__ void __launch_bounds__(256) k_seg_classify(const uint32_t *segOff, uint32_t nSeg, uint32_t hi0, uint32_t hi1, uint32_t hi2, uint32_t hi3,
                                                      uint32_t *lists, uint32_t *bigB, uint32_t *bigE, unsigned int *counts)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; const int lane = (int)(threadIdx.x & 63u);
    uint32_t b = 0, e = 0; int cls = -1;
    if (s < nSeg) { b = segOff[s]; e = segOff[s + 1]; const uint32_t len = e - b; cls = len == 0 ? -1 : (len <= hi0 ? 0 : (len <= hi1 ? 1 : (len <= hi2 ? 2 : (len <= hi3 ? 3 : 4)))); }
    if (s < nSeg) { bigB[s] = cls == 4 ? b : 0u; bigE[s] = cls == 4 ? e : 0u; }
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const unsigned long long m = __ballot(cls == c);
        if (m == 0ull) continue;                                             // wave-uniform
        unsigned base = 0; const int first = __builtin_ctzll(m);
        if (lane == first) base = atomicAdd(&counts[c], (unsigned)__builtin_popcountll(m));
        base = (unsigned)__builtin_amdgcn_readlane((int)base, first);
        if (c < 4 && cls == c) lists[(size_t)c * nSeg + base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = s;
    }
}
