// regions.h -- A3 on the device: region boundaries of the fresh fragment array, the regions by size class, and the creation-order rank of the root clumps
// (QueryMatch.c:146-158, 224-303).  Included by stage_chain.hip, behind chain.h (ChainClumpRec).
#pragma once
#include "lookback.h"

// A3: regions = runs of consecutive fragments of one (read, strand) whose diagonals differ by <= maxGap (QueryMatch.c:146-158).  One pass over the fresh
// fragment array: head flags, their exclusive scan (the look-back of k_frag_scan_build) and regStart[r] = first fragment of region r; the same pass sets
// refLen (setRefLen, FragsClumps.inl:44-47), which needs the head's and the last hit's stores of the build kernel to have landed.  regStart[nRegions] = nFrags
// is set by the host.  A tile = 4 waves x 8 rows of 64 fragments.
#define YD_REG_IPT 8
#define YD_REG_TILE (256 * YD_REG_IPT)
static_assert(4 * YD_REG_IPT <= 64, "k_region_scan: one lane of wave 0 per (wave, row) count");
__global__ void __launch_bounds__(256) k_region_scan(DevFrag *frags, uint32_t nFrags, int maxGap, uint32_t *regStart, unsigned long long *tileState,
    unsigned int *total /* [0] the count, [1] raised when the look-back gave up */)
{
    YD_HIGH_PRIO();
    __shared__ uint32_t sCnt[4 * YD_REG_IPT]; __shared__ uint32_t sPrefix;
    const uint32_t tile = tileTicket(tileState + gridDim.x, &sPrefix), t = threadIdx.x, lane = t & 63u, w = t >> 6,
        wbase = tile * (uint32_t)YD_REG_TILE + w * (uint32_t)(64 * YD_REG_IPT);
    uint32_t rs[YD_REG_IPT], dg[YD_REG_IPT]; unsigned long long headMask[YD_REG_IPT];
#pragma unroll
    for (int k = 0; k < YD_REG_IPT; k++) {
        const uint32_t f = wbase + (uint32_t)k * 64u + lane; rs[k] = 0xFFFFFFFFu; dg[k] = 0;
        if (f < nFrags) {
            const uint4 v = *(const uint4 *)&frags[f];                        // sro | sqo, eqo | refLen, used | rs
            const uint32_t sqo = v.y & 0xFFFFu, eqo = v.y >> 16;
            rs[k] = v.w; dg[k] = v.x - sqo;
            frags[f].refLen = (uint16_t)(1u + eqo - sqo);
        }
    }
    uint32_t eRs = 0xFFFFFFFFu, eDg = 0;                                        // lane 0: the fragment before the wave's range
    if (lane == 0u && wbase > 0u && wbase <= nFrags) { const DevFrag a = frags[wbase - 1u]; eRs = a.rs; eDg = a.sro - (uint32_t)a.sqo; }
#pragma unroll
    for (int k = 0; k < YD_REG_IPT; k++) {
        const uint32_t f = wbase + (uint32_t)k * 64u + lane;
        const uint32_t upRs = k > 0 ? rs[k - 1] : eRs, upDg = k > 0 ? dg[k - 1] : eDg;
        const uint32_t a0rs = (uint32_t)__builtin_amdgcn_readlane((int)upRs, k > 0 ? 63 : 0), a0dg = (uint32_t)__builtin_amdgcn_readlane((int)upDg, k > 0 ? 63 : 0);
        const uint32_t prs = (uint32_t)laneUp1((int)rs[k], (int)a0rs), pdg = (uint32_t)laneUp1((int)dg[k], (int)a0dg);
        const bool head = f < nFrags && (f == 0u || prs != rs[k] || absDiffU(pdg, dg[k]) > (uint32_t)maxGap);
        const unsigned long long m = __ballot(head); headMask[k] = m;
        if (lane == 0u) sCnt[(int)w * YD_REG_IPT + k] = (uint32_t)__builtin_popcountll(m);
    }
    __syncthreads();
    if (w == 0u) {
        const uint32_t v = lane < 4u * YD_REG_IPT ? sCnt[lane] : 0u; uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += x; }
        if (lane < 4u * YD_REG_IPT) sCnt[lane] = incl - v;
        const uint32_t agg = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t excl = tileLookBack(tileState, tile, agg, lane, total + 1);
        if (lane == 0u) { sPrefix = excl; if (tile + 1u == gridDim.x) *total = excl + agg; }
    }
    __syncthreads();
    const uint32_t prefix = sPrefix; const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < YD_REG_IPT; k++)
        if ((headMask[k] >> lane) & 1ull) regStart[prefix + sCnt[(int)w * YD_REG_IPT + k] + (uint32_t)__builtin_popcountll(headMask[k] & below)] = wbase + (uint32_t)k * 64u + lane;
}
// multi-fragment region lists + largest region
// smallList: regions with 2..8 fragments (k_chain_lanes<8>); the middle class, 9..16, fills the same array from its top downwards (midTop[-1 - i]: k_chain_lanes<16>; the
// two lists together are no longer than the regions are many); multiList: 17..64 (k_chain); bigList: more than 64 (k_chain_big)
__global__ void __launch_bounds__(1024) k_region_classify(const uint32_t *regStart, uint32_t nRegions, uint32_t *multiList, unsigned int *nMulti, unsigned int *maxN,
    uint32_t *bigList, unsigned int *nBig, uint32_t *smallList, unsigned long long *nSmallMid /* low half: the small regions, high half: the middle ones */, uint32_t *midTop)
{
    YD_HIGH_PRIO();
    __shared__ unsigned sM[16], sS[16], sD[16], sBase[3];                    // one atomic per list and 1024-thread block (a single L2 word takes ~88 atomics per microsecond)
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    const uint32_t n = r < nRegions ? regStart[r + 1] - regStart[r] : 0u;
    const bool big = n > 64, small = n >= 2 && n <= 8, mid = n > 8 && n <= 16, multi = n > 16 && !big;
    if (big) { unsigned p = atomicAdd(nBig, 1u); bigList[p] = r; atomicMax(maxN, n); }        // rare
    const unsigned long long mm = __ballot(multi), ms = __ballot(small), md = __ballot(mid);
    if (lane == 0) { sM[wv] = (unsigned)__builtin_popcountll(mm); sS[wv] = (unsigned)__builtin_popcountll(ms); sD[wv] = (unsigned)__builtin_popcountll(md); }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tm = 0, ts = 0, td = 0; for (unsigned k = 0; k < blockDim.x / 64u; k++) { tm += sM[k]; ts += sS[k]; td += sD[k]; }
        sBase[0] = tm ? atomicAdd(nMulti, tm) : 0u;
        // (the two lists of the lane kernels in one atomic: a block's atomics on the batch's counters are what this kernel waits for)
        const unsigned long long b = (ts | td) ? atomicAdd(nSmallMid, (unsigned long long)ts | ((unsigned long long)td << 32)) : 0ull;
        sBase[1] = (unsigned)b; sBase[2] = (unsigned)(b >> 32);
    }
    __syncthreads();
    unsigned bm = 0, bs = 0, bd = 0; for (int k = 0; k < wv; k++) { bm += sM[k]; bs += sS[k]; bd += sD[k]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (multi) multiList[sBase[0] + bm + (unsigned)__builtin_popcountll(mm & below)] = r;
    if (small) smallList[sBase[1] + bs + (unsigned)__builtin_popcountll(ms & below)] = r;
    if (mid) midTop[-1 - (long)(sBase[2] + bd + (unsigned)__builtin_popcountll(md & below))] = r;
}
// order[base[region] + seq] = clump index  (rank of a root clump = creation order)
// ... and sorted[rank] = the record itself: what the align stage reads (a wave's 64 roots in one stretch; through `order` they are 64 places of the arena)
__global__ void k_clump_order(const ChainClumpRec *clumps, uint32_t nClumps, const uint32_t *regionBase, uint32_t *order, ChainClumpRec *sorted)
{
    YD_HIGH_PRIO();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nClumps) return;
    const ChainClumpRec rec = clumps[c];
    if (rec.nFrags == 0xFFFFFFFFu) return;                                    // unused slot of a wave's reservation chunk
    const uint32_t rank = regionBase[rec.region] + rec.seq;
    order[rank] = c; sorted[rank] = rec;
}
