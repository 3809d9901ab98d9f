// layout.h -- A10 on the device: where the clumps of a batch go in the reference's output order (QueryMatch.c:306-331, QueryState.c:156-161).
// Included by stage_align.hip, behind the align headers (ChainClumpRec).
#pragma once
#include "common.h"

// final layout: clump ci of root r with push number p goes to rootBase[r] + (pushCount[r] - 1 - p)
__global__ void k_out_layout(const uint32_t *outRoot, const uint32_t *outPush, const uint32_t *rootBase, const unsigned int *rootPushCount, uint32_t nOut, uint32_t *dstIdx)
{
    YD_HIGH_PRIO();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nOut) return;
    const uint32_t r = outRoot[c];
    dstIdx[c] = rootBase[r] + (rootPushCount[r] - 1u - outPush[c]);
}
__global__ void k_out_scatter(const ygpu_clump *src, const uint32_t *dstIdx, uint32_t nOut, ygpu_clump *dst)
{
    YD_HIGH_PRIO();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nOut) return;
    dst[dstIdx[c]] = src[c];
}
// clumps per read: root r belongs to read (sorted[r].rs >> 1) (the records in rank order, k_clump_order)
__global__ void k_read_counts(const ChainClumpRec *sorted, const unsigned int *rootPushCount, uint32_t nRoots, unsigned int *readCount)
{
    YD_HIGH_PRIO();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nRoots) return;
    const unsigned n = rootPushCount[r];
    if (n) atomicAdd(&readCount[sorted[r].rs >> 1], n);
}
