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
// Four lists in two arrays (a list from the bottom and one from the top of each: the lists together are no longer than the regions are many): 2..4 fragments from
// the top of multiList (multiTop[-1 - i]: k_chain_lanes<4>), 5..8 in smallList (k_chain_lanes<8>), 9..16 from its top (smallTop[-1 - i]: k_chain_lanes<16>), 17..64 in
// multiList (k_chain); bigList: more than 64 (k_chain_big).  cnt2: two 64-bit words, (multi | tiny << 32) and (small | middle << 32).
// (regions a thread: a block's two atomics on the batch's counters are what the kernel waits for -- one L2 word takes ~88 a microsecond)
#define YD_RCLS_IPT 4
__global__ void __launch_bounds__(1024) k_region_classify(const uint32_t *regStart, uint32_t nRegions, uint32_t *multiList, uint32_t *multiTop, uint32_t *smallList,
    uint32_t *smallTop, unsigned long long *cnt2, unsigned int *maxN, uint32_t *bigList, unsigned int *nBig)
{
    YD_HIGH_PRIO();
    __shared__ unsigned sC[4][16], sBase[4];
    const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
    const uint32_t r0 = blockIdx.x * (1024u * YD_RCLS_IPT) + threadIdx.x;
    int cls[YD_RCLS_IPT]; unsigned long long m[YD_RCLS_IPT][4]; unsigned tot[4] = {0, 0, 0, 0};      // (the masks and counts are wave-uniform: scalar registers)
#pragma unroll
    for (int k = 0; k < YD_RCLS_IPT; k++) {
        const uint32_t r = r0 + (uint32_t)k * 1024u;
        const uint32_t n = r < nRegions ? regStart[r + 1] - regStart[r] : 0u;
        const bool big = n > 64;
        cls[k] = n < 2 || big ? -1 : (n > 16 ? 0 : (n <= 4 ? 1 : (n <= 8 ? 2 : 3)));          // multi, tiny, small, middle
        if (big) { unsigned p = atomicAdd(nBig, 1u); bigList[p] = r; atomicMax(maxN, n); }    // rare
#pragma unroll
        for (int c = 0; c < 4; c++) { m[k][c] = __ballot(cls[k] == c); tot[c] += (unsigned)__builtin_popcountll(m[k][c]); }
    }
    if (lane == 0) { sC[0][wv] = tot[0]; sC[1][wv] = tot[1]; sC[2][wv] = tot[2]; sC[3][wv] = tot[3]; }
    __syncthreads();
    // lane = class * 16 + wave: exclusive sums over the waves of a class, the class totals to the batch's counters
    if (threadIdx.x < 64u) {
        const unsigned c = sC[lane >> 4][lane & 15]; unsigned incl = c;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { const unsigned y = (unsigned)__shfl_up((int)incl, d, 16); if ((lane & 15) >= d) incl += y; }
        sC[lane >> 4][lane & 15] = incl - c;
        const unsigned t0 = (unsigned)__shfl((int)incl, 15, 64), t1 = (unsigned)__shfl((int)incl, 31, 64), t2 = (unsigned)__shfl((int)incl, 47, 64),
            t3 = (unsigned)__shfl((int)incl, 63, 64);
        if (lane == 0) {
            const unsigned long long a = (t0 | t1) ? atomicAdd(cnt2, (unsigned long long)t0 | ((unsigned long long)t1 << 32)) : 0ull;
            const unsigned long long b = (t2 | t3) ? atomicAdd(cnt2 + 1, (unsigned long long)t2 | ((unsigned long long)t3 << 32)) : 0ull;
            sBase[0] = (unsigned)a; sBase[1] = (unsigned)(a >> 32); sBase[2] = (unsigned)b; sBase[3] = (unsigned)(b >> 32);
        }
    }
    __syncthreads();
    unsigned at[4]; for (int c = 0; c < 4; c++) at[c] = sBase[c] + sC[c][wv];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < YD_RCLS_IPT; k++) {
        const uint32_t r = r0 + (uint32_t)k * 1024u;
        if (cls[k] == 0) multiList[at[0] + (unsigned)__builtin_popcountll(m[k][0] & below)] = r;
        else if (cls[k] == 1) multiTop[-1 - (long)(at[1] + (unsigned)__builtin_popcountll(m[k][1] & below))] = r;
        else if (cls[k] == 2) smallList[at[2] + (unsigned)__builtin_popcountll(m[k][2] & below)] = r;
        else if (cls[k] == 3) smallTop[-1 - (long)(at[3] + (unsigned)__builtin_popcountll(m[k][3] & below))] = r;
#pragma unroll
        for (int c = 0; c < 4; c++) at[c] += (unsigned)__builtin_popcountll(m[k][c]);
    }
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
