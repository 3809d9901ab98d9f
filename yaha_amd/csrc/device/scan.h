// scan.h -- the batch-wide exclusive sums and the orderings of the hot path, written for it (rounds 1-4 called the library for these: nine call sites, 44 launches a
// batch).  The reference has neither: it walks one read at a time (Query.c:306-497) and merges already-sorted lists (QueryMatch.c:52-121); the sums and orders here
// exist only because a batch's variable-size outputs are laid out as count -> exclusive sum -> fill, and because lanes of a wave should run problems of one shape.
//
//   k_scan_excl<T>      out[i] = in[0] + ... + in[i-1] over n elements of u32 or u64, ONE launch, one read and one write of HBM per element: a workgroup owns a tile
//                       of 8 192 elements (eight consecutive ones a thread: a serial prefix in registers, the lanes' totals by DPP-free shuffles, the waves' through
//                       LDS), the tile's offset comes from the tiles before it by decoupled look-back over 64-bit state words (status : 2 | value : 62).  The tile is
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

#define YD_SCAN_BS 1024
#define YD_SCAN_IPT 8
#define YD_SCAN_TILE (YD_SCAN_BS * YD_SCAN_IPT)
__host__ __device__ inline uint32_t scanTiles(uint64_t n) { return (uint32_t)((n + YD_SCAN_TILE - 1) / YD_SCAN_TILE); }
__host__ __device__ inline size_t scanStateBytes(uint64_t n) { return 8ull * ((size_t)scanTiles(n) + 4); }       // tile words, ticket, done counter

__device__ __forceinline__ unsigned long long shflUp64(unsigned long long v, int d)
{ return ((unsigned long long)(uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, d, 64); }
__device__ __forceinline__ unsigned long long waveSum64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64);
    return v;
}

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

template <class T>
__global__ void __launch_bounds__(YD_SCAN_BS) k_scan_excl(const T *in, T *out, uint32_t n, unsigned long long *state /* scanStateBytes(n), zero */, unsigned int *failed)
{
    YD_HIGH_PRIO();
    constexpr int NW = YD_SCAN_BS / 64;
    __shared__ unsigned long long sWave[NW]; __shared__ unsigned long long sPrefix; __shared__ uint32_t sTile;
    const uint32_t nTiles = gridDim.x, t = threadIdx.x, lane = t & 63u, w = t >> 6;
    if (t == 0) sTile = (uint32_t)atomicAdd(&state[nTiles], 1ull);
    __syncthreads();
    const uint32_t tile = sTile;
    const uint64_t i0 = (uint64_t)tile * YD_SCAN_TILE + (uint64_t)t * YD_SCAN_IPT;
    T v[YD_SCAN_IPT];
    const bool vec = i0 + YD_SCAN_IPT <= n && ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0u;      // whole threads of 16-byte aligned arrays move as vectors
    if (vec) {
        if (sizeof(T) == 4) { const uint4 a = *(const uint4 *)(in + i0), b = *(const uint4 *)(in + i0 + 4); v[0] = (T)a.x; v[1] = (T)a.y; v[2] = (T)a.z; v[3] = (T)a.w; v[4] = (T)b.x; v[5] = (T)b.y; v[6] = (T)b.z; v[7] = (T)b.w; }
        else {
#pragma unroll
            for (int k = 0; k < YD_SCAN_IPT; k += 2) { const ulonglong2 a = *(const ulonglong2 *)(in + i0 + k); v[k] = (T)a.x; v[k + 1] = (T)a.y; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < YD_SCAN_IPT; k++) v[k] = i0 + k < n ? in[i0 + k] : (T)0;
    }
    unsigned long long sum = 0ull;                                           // the thread's serial prefix: v[k] becomes the sum of the elements before it in the thread
#pragma unroll
    for (int k = 0; k < YD_SCAN_IPT; k++) { const T x = v[k]; v[k] = (T)sum; sum += (unsigned long long)x; }
    unsigned long long incl = sum;                                           // inclusive over the wave's lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long x = shflUp64(incl, d); if ((int)lane >= d) incl += x; }
    if (lane == 63u) sWave[w] = incl;
    __syncthreads();
    if (w == 0u) {
        const unsigned long long wv = lane < (uint32_t)NW ? sWave[lane] : 0ull; unsigned long long wi = wv;
#pragma unroll
        for (int d = 1; d < NW; d <<= 1) { const unsigned long long x = shflUp64(wi, d); if ((int)lane >= d) wi += x; }
        if (lane < (uint32_t)NW) sWave[lane] = wi - wv;                       // exclusive over the waves
        const unsigned long long agg = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(wi >> 32), NW - 1) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)wi, NW - 1);
        const unsigned long long excl = tileLookBack64(state, tile, agg, lane, failed);
        if (lane == 0u) sPrefix = excl;
    }
    __syncthreads();
    const unsigned long long base = sPrefix + sWave[w] + (incl - sum);
    if (vec) {
        if (sizeof(T) == 4) {
            uint4 a, b; a.x = (uint32_t)(base + v[0]); a.y = (uint32_t)(base + v[1]); a.z = (uint32_t)(base + v[2]); a.w = (uint32_t)(base + v[3]);
            b.x = (uint32_t)(base + v[4]); b.y = (uint32_t)(base + v[5]); b.z = (uint32_t)(base + v[6]); b.w = (uint32_t)(base + v[7]);
            *(uint4 *)(out + i0) = a; *(uint4 *)(out + i0 + 4) = b;
        } else {
#pragma unroll
            for (int k = 0; k < YD_SCAN_IPT; k += 2) { ulonglong2 a; a.x = base + v[k]; a.y = base + v[k + 1]; *(ulonglong2 *)(out + i0 + k) = a; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < YD_SCAN_IPT; k++) if (i0 + k < n) out[i0 + k] = (T)(base + v[k]);
    }
    // self-cleaning: whoever finishes last has seen every other workgroup leave its look-back
    __syncthreads();
    if (t == 0) { __threadfence(); sTile = (uint32_t)atomicAdd(&state[nTiles + 1u], 1ull); }
    __syncthreads();
    if (sTile == nTiles - 1u) for (uint32_t k = t; k < nTiles + 2u; k += YD_SCAN_BS) __hip_atomic_store(&state[k], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- ordering by a small key ------------------------------------------------------------------------------------------------------------------------------
#define YD_BKT_BS 256
#define YD_BKT_IPT 16
#define YD_BKT_TILE (YD_BKT_BS * YD_BKT_IPT)
#define YD_BKT_MAX 4096                         // buckets (12 key bits)
// work words of one ordering: hist[nb] | cursor[nb] | done; zero before the first use, left zero by k_bucket_scatter
__host__ __device__ inline size_t bucketWorkBytes() { return 4ull * (2 * YD_BKT_MAX + 4); }
// bucket of a key: (key - sub) >> shift, clamped to [0, nb) (keys below `sub` go to bucket 0)
__device__ __forceinline__ uint32_t bucketOf(uint32_t key, uint32_t sub, int shift, uint32_t nb) { const uint32_t b = (key > sub ? key - sub : 0u) >> shift; return b < nb ? b : nb - 1u; }

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
static __global__ void __launch_bounds__(YD_BKT_BS) k_bucket_scatter(const uint32_t *keys, const uint32_t *vals, uint32_t valBase, uint32_t n, uint32_t sub, int shift, uint32_t nb, unsigned int *work, uint32_t *outVals, uint32_t *outKeys /* or nullptr */)
{
    YD_HIGH_PRIO();
    __shared__ unsigned int sPos[YD_BKT_MAX]; __shared__ unsigned int sCnt[YD_BKT_MAX]; __shared__ unsigned int sWave[YD_BKT_BS / 64]; __shared__ unsigned int sLast;
    const uint32_t t = threadIdx.x, lane = t & 63u, w = t >> 6;
    // every workgroup scans the global histogram itself: thread t owns the buckets [t * per, (t + 1) * per)
    const uint32_t per = (nb + YD_BKT_BS - 1u) / YD_BKT_BS;
    unsigned int sum = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t b = t * per + k; const unsigned c = b < nb ? __hip_atomic_load(&work[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; if (b < nb) { sPos[b] = sum; sCnt[b] = 0u; } sum += c; }
    unsigned int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned x = (unsigned)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += x; }
    if (lane == 63u) sWave[w] = incl;
    __syncthreads();
    unsigned int wbase = 0; for (uint32_t k = 0; k < w; k++) wbase += sWave[k];
    const unsigned int tbase = wbase + incl - sum;
    for (uint32_t k = 0; k < per; k++) { const uint32_t b = t * per + k; if (b < nb) sPos[b] += tbase; }
    __syncthreads();
    // the tile's own histogram, then one reservation per non-empty bucket: sPos[b] = where this workgroup's items of bucket b go
    const uint32_t base = blockIdx.x * (uint32_t)YD_BKT_TILE;
    uint32_t key[YD_BKT_IPT];
#pragma unroll
    for (int k = 0; k < YD_BKT_IPT; k++) { const uint32_t i = base + (uint32_t)k * YD_BKT_BS + t; key[k] = i < n ? keys[i] : 0u; if (i < n) atomicAdd(&sCnt[bucketOf(key[k], sub, shift, nb)], 1u); }
    __syncthreads();
    for (uint32_t b = t; b < nb; b += YD_BKT_BS) { const unsigned c = sCnt[b]; if (c) sPos[b] += atomicAdd(&work[YD_BKT_MAX + b], c); }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < YD_BKT_IPT; k++) {
        const uint32_t i = base + (uint32_t)k * YD_BKT_BS + t;
        if (i < n) { const unsigned p = atomicAdd(&sPos[bucketOf(key[k], sub, shift, nb)], 1u); outVals[p] = vals ? vals[i] : i + valBase; if (outKeys) outKeys[p] = key[k]; }
    }
    // self-cleaning (the last workgroup to finish: every other one has read the histogram and made its reservations)
    __syncthreads();
    if (t == 0) { __threadfence(); sLast = atomicAdd(&work[2 * YD_BKT_MAX], 1u); }
    __syncthreads();
    if (sLast == gridDim.x - 1u) { for (uint32_t b = t; b < nb; b += YD_BKT_BS) { work[b] = 0u; work[YD_BKT_MAX + b] = 0u; } if (t == 0) work[2 * YD_BKT_MAX] = 0u; }
}
