// lookback.h -- the decoupled look-back of the one-pass scans over tiles (k_frag_scan_build in seed.h, k_region_scan in regions.h).
#pragma once
#include "common.h"

// Decoupled look-back of a single-pass scan over tiles (one 64-bit state word per tile, zeroed before the launch: status in the high half -- 1 = the tile's own
// count, 2 = its inclusive prefix -- and the value in the low half, so that both arrive together).  Called by one whole wave of tile `tile` with the tile's
// count; returns the sum of the counts of all tiles before it.  A workgroup's tile is the TICKET it draws when it starts (tileTicket), not its blockIdx: every
// tile before it has then started as well and publishes its count without waiting for anybody.  (With tile = blockIdx two such kernels running side by side --
// two batches in flight, or two processes on one device -- can fill each other's XCD with waiting workgroups while the tile both are waiting for has not been
// dispatched there: seen as a stall with two processes on one GPU.  *failed is raised, after a bounded wait, should a tile ever not show up.)
__device__ __forceinline__ uint32_t tileTicket(unsigned long long *ticketWord /* zeroed with the tile states */, uint32_t *sSlot)
{
    if (threadIdx.x == 0) *sSlot = (uint32_t)atomicAdd(ticketWord, 1ull);
    __syncthreads();
    return *sSlot;
}
__device__ __forceinline__ uint32_t tileLookBack(unsigned long long *tileState, uint32_t tile, uint32_t agg, uint32_t lane, unsigned int *failed)
{
    uint32_t excl = 0;
    if (tile == 0u) { if (lane == 0u) __hip_atomic_store(&tileState[0], (2ull << 32) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 0u; }
    if (lane == 0u) __hip_atomic_store(&tileState[tile], (1ull << 32) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int back = (int)tile - 1;                                            // lane l looks at tile back - l
    for (;;) {
        const int j = back - (int)lane;
        unsigned long long st = 2ull << 32;                              // before the first tile: a known prefix of zero
        if (j >= 0) {
            // (every tile waited for has drawn its ticket, so its workgroup is resident and publishes without waiting for anybody: the wait is bounded by WALL TIME
            // only -- a device that is time-sliced between processes, or stopped under a debugger, may take long -- 30 s of the 100 MHz clock, asleep between polls
            // after the first few; the caller sees the flag)
            st = __hip_atomic_load(&tileState[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((st >> 32) == 0ull) {
                const unsigned long long t0 = wall_clock64(); unsigned polls = 0;
                do { if (++polls > 64u) __builtin_amdgcn_s_sleep(32); st = __hip_atomic_load(&tileState[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                while ((st >> 32) == 0ull && wall_clock64() - t0 < 3000000000ull);
                if ((st >> 32) == 0ull) { st = 2ull << 32; atomicMax(failed, 1u); }
            }
        }
        const unsigned long long known = __ballot((st >> 32) == 2ull);
        const int stop = __builtin_ctzll(known | (1ull << 63));          // the nearest tile that knows its prefix (lane 63 at the latest if any)
        const bool use = known ? (int)lane <= stop : true;
        // (DPP: six shuffles through the LDS crossbar here were most of a look-back round, and the rounds are a chain)
        excl += waveTotalSumU(use ? (uint32_t)st : 0u);
        if (known) break;
        back -= 64;
    }
    if (lane == 0u) __hip_atomic_store(&tileState[tile], (2ull << 32) | (unsigned long long)(excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}
