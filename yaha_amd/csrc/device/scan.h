// scan.h -- the batch-wide exclusive sums and the orderings of the hot path, written for it (rounds 1-4 called the library for these: nine call sites, 44 launches a
// batch).  The reference has neither: it walks one read at a time (Query.c:306-497) and merges already-sorted lists (QueryMatch.c:52-121); the sums and orders here
// exist only because a batch's variable-size outputs are laid out as count -> exclusive sum -> fill, and because lanes of a wave should run problems of one shape.
//
//   k_scan_excl<T>      out[i] = in[0] + ... + in[i-1] over n elements of u32 or u64, ONE launch, one read and one write of HBM per element: a workgroup owns a tile
//                       of 4 096 elements (a wave moves 1 KB of consecutive memory per instruction; sums inside 16-byte pieces, over the lanes by DPP, the four
// waves' through LDS), the tile's offset comes from the tiles before it by decoupled look-back over 64-bit state words (status : 2 | value : 62).  The tile is
//                       the TICKET a workgroup draws when it starts (seed.h: tileTicket -- no assumption about dispatch order).  The state is SELF-CLEANING: the last
//                       workgroup to finish zeroes the words the launch used, so that the next launch needs no memset (the host zeroes the buffer when it makes it and
//                       after a launch that reported a failure).
//   k_bucket_count /    an ORDER by a small key (up to 12 bits: 4 096 buckets): problem indices grouped by bucket, ascending buckets, any order inside a bucket -- all the
//   k_bucket_scatter    hot path needs (longest bound first, equal walk lengths together, joints of one shape together: every result is written to its own problem's
//                       place, so the order inside a bucket changes nothing that is returned).  Two launches: per-workgroup histograms in LDS added to a global one;
//                       then every workgroup scans the global histogram itself (LDS), reserves its items' places with one atomic per non-empty bucket and scatters.
//                       The counters are self-cleaning like the scan's.
#pragma once
#include "common.h"

#define YD_SCAN_BS 512                       // two waves per SIMD, ~100 registers: fits beside a rows launch that shares the device (a 1 024-thread workgroup needs four waves on
                                             // every SIMD of one CU at once and found room on a third of the CUs only: 0.05 -> 0.61 ms a sum with four batches in flight)
#ifndef YD_SCAN_IPT
#define YD_SCAN_IPT 48                       // u32 elements a thread (192 bytes; u64: 24).  The tiles' look-back is a chain that advances 64 tiles a round trip: 50 ns a
#endif
                                             // tile, measured (4 096-element tiles: 400 us for 32 M elements) -- so a tile is 96 KB
#define YD_SCAN_TILE (YD_SCAN_BS * YD_SCAN_IPT)
__host__ __device__ inline uint32_t scanTiles(uint64_t n, int elemBytes = 4) { const uint64_t tile = (uint64_t)YD_SCAN_TILE * 4u / (unsigned)elemBytes;
    return (uint32_t)((n + tile - 1) / tile); }
__host__ __device__ inline size_t scanStateBytes(uint64_t n) { return 8ull * ((size_t)scanTiles(n, 8) + 4); }       // tile words (at most: the u64 tiling), ticket, done counter

__device__ __forceinline__ unsigned long long shflUp64(unsigned long long v, int d)
{ return ((unsigned long long)(uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, d, 64); }
// sum over all lanes of a 62-bit value, by DPP: three 32-bit sums (16 + 16 + 30 bits: none can overflow over 64 lanes)
__device__ __forceinline__ unsigned long long waveSum64(unsigned long long v)
{
    const uint32_t a = waveInclSumU((uint32_t)v & 0xFFFFu), b = waveInclSumU(((uint32_t)v >> 16) & 0xFFFFu), c = waveInclSumU((uint32_t)(v >> 32));
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)a, 63) + ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)b,
        63) << 16) + ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)c, 63) << 32);
}
// inclusive sum over the lanes in the width of T: u32 by DPP, u64 by shuffles (one or two such sums a batch)
__device__ __forceinline__ uint32_t waveInclSumT(uint32_t v, uint32_t) { return waveInclSumU(v); }
__device__ __forceinline__ unsigned long long waveInclSumT(unsigned long long v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long x = shflUp64(v, d); if ((int)lane >= d) v += x; }
    return v;
}
__device__ __forceinline__ uint32_t readLaneT(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ unsigned long long readLaneT(unsigned long long v, int l) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l); }

// decoupled look-back on 62-bit values; called by one whole wave; returns the sum of the aggregates of all tiles before `tile`
__device__ __forceinline__ unsigned long long tileLookBack64(unsigned long long *state, uint32_t tile, unsigned long long agg, uint32_t lane, unsigned int *failed)
{
    constexpr unsigned long long VMASK = (1ull << 62) - 1ull;
    if (tile == 0u) { if (lane == 0u) __hip_atomic_store(&state[0], (2ull << 62) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return 0ull; }
    if (lane == 0u) __hip_atomic_store(&state[tile], (1ull << 62) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long excl = 0ull; int back = (int)tile - 1;
    for (;;) {
        const int j = back - (int)lane;
        unsigned long long st = 2ull << 62;                                  // before the first tile: a known prefix of zero
        if (j >= 0) {
            st = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((st >> 62) == 0ull) {                                        // (bounded by wall time only, as in seed.h: every ticket holder before this one is resident)
                const unsigned long long t0 = wall_clock64(); unsigned polls = 0;
                do { if (++polls > 64u) __builtin_amdgcn_s_sleep(32); st = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                while ((st >> 62) == 0ull && wall_clock64() - t0 < 3000000000ull);
                if ((st >> 62) == 0ull) { st = 2ull << 62; if (failed) atomicMax(failed, 1u); }
            }
        }
        const unsigned long long known = __ballot((st >> 62) == 2ull);
        const int stop = __builtin_ctzll(known | (1ull << 63));
        const bool use = known ? (int)lane <= stop : true;
        excl += waveSum64(use ? (st & VMASK) : 0ull);
        if (known) break;
        back -= 64;
    }
    if (lane == 0u) __hip_atomic_store(&state[tile], (2ull << 62) | ((excl + agg) & VMASK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// T = uint32_t: every sum is kept in 32 bits (the caller's total fits its own output type); T = unsigned long long: 62 bits.
// A wave owns 64 * IPT consecutive elements and moves them in 16-byte pieces, piece (k, lane) = piece number k * 64 + lane of the wave's range: every load and
// store instruction covers 1 KB of consecutive memory (a thread that owned IPT consecutive elements touched a 64-byte sector per lane and instruction: twice
// as slow, measured).  Order of the sum: pieces by (k, lane), elements inside a piece.
template <class T>
__global__ void __launch_bounds__(YD_SCAN_BS) k_scan_excl(const T *in, T *out, uint32_t n, unsigned long long *state /* scanStateBytes(n), zero */, unsigned int *failed)
{
    YD_HIGH_PRIO();
    // elements a piece, pieces a lane (twelve 16-byte pieces), elements a lane
    constexpr int NW = YD_SCAN_BS / 64, PER = 16 / (int)sizeof(T), NP = YD_SCAN_IPT / 4, IPT = NP * PER;
    __shared__ T sWave[NW]; __shared__ T sPrefix; __shared__ uint32_t sTile;
    const uint32_t nTiles = gridDim.x, t = threadIdx.x, lane = t & 63u, w = t >> 6;
    if (t == 0) sTile = (uint32_t)atomicAdd(&state[nTiles], 1ull);
    __syncthreads();
    const uint32_t tile = sTile;
    const uint64_t wbase = ((uint64_t)tile * YD_SCAN_BS + (uint64_t)w * 64u) * IPT;
    const bool aligned = ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0u;
    T v[NP][PER], pre[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint64_t g0 = wbase + ((uint64_t)k * 64u + lane) * PER;
        if (aligned && g0 + PER <= n) {
            const uint4 a = *(const uint4 *)(in + g0);
            if (sizeof(T) == 4) { v[k][0] = (T)a.x; v[k][1 % PER] = (T)a.y; v[k][2 % PER] = (T)a.z; v[k][3 % PER] = (T)a.w; }
            else { v[k][0] = (T)(((unsigned long long)a.y << 32) | a.x); v[k][1] = (T)(((unsigned long long)a.w << 32) | a.z); }
        } else {
#pragma unroll
            for (int j = 0; j < PER; j++) v[k][j] = g0 + j < n ? in[g0 + j] : (T)0;
        }
    }
    T carry = 0;                                                             // sum of the wave's pieces (k' < k, any lane)
#pragma unroll
    for (int k = 0; k < NP; k++) {
        T s = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { const T x = v[k][j]; v[k][j] = s; s += x; }      // inside the piece: exclusive
        const T incl = waveInclSumT(s, lane);
        pre[k] = carry + incl - s;
        carry += readLaneT(incl, 63);
    }
    if (lane == 0u) sWave[w] = carry;
    __syncthreads();
    if (w == 0u) {
        const T mine = lane < (uint32_t)NW ? sWave[lane] : (T)0;             // lane k: wave k's sum
        const T wincl = waveInclSumT(mine, lane), agg = readLaneT(wincl, 63);
        const unsigned long long excl = tileLookBack64(state, tile, (unsigned long long)agg, lane, failed);
        if (lane < (uint32_t)NW) sWave[lane] = wincl - mine;                  // exclusive over the waves
        if (lane == 0u) sPrefix = (T)excl;
    }
    __syncthreads();
    const T base = sPrefix + sWave[w];
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint64_t g0 = wbase + ((uint64_t)k * 64u + lane) * PER; const T b = base + pre[k];
        if (aligned && g0 + PER <= n) {
            uint4 a;
            if (sizeof(T) == 4) { a.x = (uint32_t)(b + v[k][0]); a.y = (uint32_t)(b + v[k][1 % PER]); a.z = (uint32_t)(b + v[k][2 % PER]); a.w = (uint32_t)(b + v[k][3 % PER]); }
            else { const unsigned long long p = (unsigned long long)(b + v[k][0]), q = (unsigned long long)(b + v[k][1]); a.x = (uint32_t)p; a.y = (uint32_t)(p >> 32);
                a.z = (uint32_t)q; a.w = (uint32_t)(q >> 32); }
            *(uint4 *)(out + g0) = a;
        } else {
#pragma unroll
            for (int j = 0; j < PER; j++) if (g0 + j < n) out[g0 + j] = b + v[k][j];
        }
    }
    // self-cleaning: whoever finishes last has seen every other workgroup leave its look-back
    __syncthreads();
    if (t == 0) { __threadfence(); sTile = (uint32_t)atomicAdd(&state[nTiles + 1u], 1ull); }
    __syncthreads();
    if (sTile == nTiles - 1u) for (uint32_t k = t; k < nTiles + 2u; k += YD_SCAN_BS) __hip_atomic_store(&state[k], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- ordering by a small key ------------------------------------------------------------------------------------------------------------------------------
#define YD_BKT_BS 512                        // (workgroups of at most 512 threads: see YD_SCAN_BS)
#define YD_BKT_IPT 16
#define YD_BKT_TILE (YD_BKT_BS * YD_BKT_IPT)
#define YD_BKT_MAX 4096                         // buckets (12 key bits)
// work words of one ordering: hist[nb] | cursor[nb] | done; zero before the first use, left zero by k_bucket_scatter
__host__ __device__ inline size_t bucketWorkBytes() { return 4ull * (2 * YD_BKT_MAX + 4); }
// bucket of a key: (key - sub) >> shift, clamped to [0, nb) (keys below `sub` go to bucket 0)
__device__ __forceinline__ uint32_t bucketOf(uint32_t key, uint32_t sub, int shift, uint32_t nb) { const uint32_t b = (key > sub ? key - sub : 0u) >> shift;
    return b < nb ? b : nb - 1u; }

static __global__ void __launch_bounds__(YD_BKT_BS) k_bucket_count(const uint32_t *keys, uint32_t n, uint32_t sub, int shift, uint32_t nb, unsigned int *work)
{
    YD_HIGH_PRIO();
    __shared__ unsigned int sHist[YD_BKT_MAX];
    for (uint32_t b = threadIdx.x; b < nb; b += YD_BKT_BS) sHist[b] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)YD_BKT_TILE;
#pragma unroll
    for (int k = 0; k < YD_BKT_IPT; k++) { const uint32_t i = base + (uint32_t)k * YD_BKT_BS + threadIdx.x; if (i < n) atomicAdd(&sHist[bucketOf(keys[i], sub, shift, nb)], 1u); }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += YD_BKT_BS) { const unsigned c = sHist[b]; if (c) atomicAdd(&work[b], c); }
}
// vals == nullptr: the value of item i is i + valBase
static __global__ void __launch_bounds__(YD_BKT_BS) k_bucket_scatter(const uint32_t *keys, const uint32_t *vals, uint32_t valBase, uint32_t n, uint32_t sub, int shift, uint32_t nb,
    unsigned int *work, uint32_t *outVals, uint32_t *outKeys /* or nullptr */)
{
    YD_HIGH_PRIO();
    __shared__ unsigned int sPos[YD_BKT_MAX]; __shared__ unsigned int sCnt[YD_BKT_MAX]; __shared__ unsigned int sWave[YD_BKT_BS / 64]; __shared__ unsigned int sLast;
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    // the tile's keys first (their latency covers the histogram's), then every workgroup scans the global histogram itself: thread t owns buckets [t * per, (t + 1) * per)
    const uint32_t base = blockIdx.x * (uint32_t)YD_BKT_TILE;
    uint32_t key[YD_BKT_IPT];
#pragma unroll
    for (int k = 0; k < YD_BKT_IPT; k++) { const uint32_t i = base + (uint32_t)k * YD_BKT_BS + t; key[k] = i < n ? keys[i] : 0u; }
    const uint32_t per = (nb + YD_BKT_BS - 1u) / YD_BKT_BS;               // 1 .. 4
    unsigned int sum = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t b = t * per + k; const unsigned c = b < nb ? __hip_atomic_load(&work[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (b < nb) { sPos[b] = sum; sCnt[b] = 0u; } sum += c; }
    const unsigned int incl = waveInclSumU(sum);
    if (lane == 63u) sWave[w] = incl;
    __syncthreads();
    unsigned int wbase = 0; for (uint32_t k = 0; k < w; k++) wbase += sWave[k];
    const unsigned int tbase = wbase + incl - sum;
    for (uint32_t k = 0; k < per; k++) { const uint32_t b = t * per + k; if (b < nb) sPos[b] += tbase; }
#pragma unroll
    // (key[] holds the bucket from here on)
    for (int k = 0; k < YD_BKT_IPT; k++) { const uint32_t i = base + (uint32_t)k * YD_BKT_BS + t; if (i < n) { key[k] = bucketOf(key[k], sub, shift, nb);
        atomicAdd(&sCnt[key[k]], 1u); } }
    __syncthreads();
    // one reservation per non-empty bucket: sPos[b] = where this workgroup's items of bucket b go
    for (uint32_t b = t; b < nb; b += YD_BKT_BS) { const unsigned c = sCnt[b]; if (c) sPos[b] += atomicAdd(&work[YD_BKT_MAX + b], c); }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < YD_BKT_IPT; k++) {
        const uint32_t i = base + (uint32_t)k * YD_BKT_BS + t;
        if (i < n) { const unsigned p = atomicAdd(&sPos[key[k]], 1u); outVals[p] = vals ? vals[i] : i + valBase; if (outKeys) outKeys[p] = keys[i]; }
    }
    // self-cleaning (the last workgroup to finish: every other one has read the histogram and made its reservations)
    __syncthreads();
    if (t == 0) { __threadfence(); sLast = atomicAdd(&work[2 * YD_BKT_MAX], 1u); }
    __syncthreads();
    if (sLast == gridDim.x - 1u) { for (uint32_t b = t; b < nb; b += YD_BKT_BS) { work[b] = 0u; work[YD_BKT_MAX + b] = 0u; } if (t == 0) work[2 * YD_BKT_MAX] = 0u; }
}
