// common.h -- shared device-side definitions for the gfx950 hot path (wave64, CDNA4).
// Everything here is written for 64-wide wavefronts; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/yaha_hip.h"

#define YD_WORST   (-(0x7fffff00))          // reference DPWorstScore, SW.cpp:356
#define YD_BIAS    (1 << 24)                 // keeps (score + BIAS) positive and below 2^25 (checked in ygpu_init)
#define YD_WAVE    64
// Wave priority.  A SIMD issues for the wave of the highest priority that is ready, the oldest one among equals -- and the waves of the X-drop rows kernels
// (k_ext_rows*: persistent, 3 per SIMD, always an instruction ready) are older than anything launched beside them: with every wave at the default priority a kernel of
// another batch in flight that shares their SIMDs is starved of issue slots until the rows launch ends (a 0.3 ms kernel took 17 ms there, a buffer copy 14 ms,
// profiles/r04_timeline_four_contexts_before_priorities.txt).  So every kernel BUT the rows kernels raises its waves at entry: memory-bound as they are, they take
// the few issue slots they need when they need them and wait for memory the rest of the time, and the rows kernel runs in what they leave.
#ifndef YD_PRIO
#define YD_PRIO 3
#endif
#define YD_HIGH_PRIO() __builtin_amdgcn_s_setprio(YD_PRIO)
// Stores of data that the storing kernel never reads again and that no kernel reads soon (trace records, sorted keys, ...): non-temporal, so that they do not push
// out of L2 what the kernel does come back to.  k_ext_rows_pk's 20.5 GB of trace records a launch evicted the sectors of its lanes' query / reference streams
// between two of their dword refills: 25.5 GB fetched for 2.2 GB of input, 15.2 GB with the records stored non-temporally (round 4, profiles/r04_nt_stores.txt).
#define YD_STORE_NT(ptr, val) __builtin_nontemporal_store((val), (ptr))
// (Only there: the same on the sorted keys, k_expand_hits' keys and the traceback's staged ops made a step 7 ms slower -- partial lines written non-temporally do not
// combine in L2, k_ext_trace_pk wrote 6.96 GB instead of 4.47.)

enum { stReversed = 0x01, stAligned = 0x04, stScored = 0x08, stSplit = 0x10 };   // FragsClumps.inl:235-240
enum { OP_M = 0, OP_R = 1, OP_D = 2, OP_I = 3 };

// explicit address spaces: a generic (flat) store also counts on lgkmcnt and forces a full wait per DP row
#define YD_GLOBAL __attribute__((address_space(1)))
#define YD_LDS    __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ YD_GLOBAL T *toGlobal(T *p) { return (YD_GLOBAL T *)p; }
#define YD_LBAND 10                 // extension bandwidth 2 * BW of the lane kernels (-BW 5): columns left and right of the origin
#define YD_MEMO 12                  // careful-extension problems listed per split root
#define YD_LDS_CELLS 5120            // uint16 trace cells kept in LDS per wavefront (10 KB): 160 rows of a 21-wide extension strip

struct DevParams {
    int wordLen, maxHits, bandWidth, maxGap, maxIntron, minMatch, maxDesert, minNonOverlap, minRawScore, minExtLength;
    int GO, GE, RC, MS, X; float minIdentity;
    uint32_t maxROff, totalMatches;
};

struct DevFrag { uint32_t sro; uint16_t sqo, eqo; uint16_t refLen; uint16_t used; uint32_t rs; };   // = ygpu_fragment (16 B)

// per-batch device view handed to the kernels
struct DevBatch {
    const uint8_t  *fwd, *rev;     // 4-bit codes, one per byte, both strands, same offsets
    const uint32_t *readOff;       // nReads + 1
    uint32_t nReads;
};

struct DevCounters { unsigned long long v[16]; };   // same order as ygpu_counters
enum { C_KMER = 0, C_HITS, C_FRAGS, C_REGIONS, C_FORMED, C_SCORED, C_EXT_CALLS, C_EXT_ROWS, C_EXT_CELLS, C_GAP_CALLS, C_GAP_ROWS, C_GAP_CELLS, C_PERFECT, C_TOUCHED, C_OPS,
    C_SPLITS };

// ---- wave64 helpers ----------------------------------------------------------------------------------------
__device__ __forceinline__ int  laneId() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int  uni(int v) { return __builtin_amdgcn_readfirstlane(v); }            // make a wave-uniform value scalar
__device__ __forceinline__ unsigned uniU(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
// wave-uniform condition -> scalar branch.  Every branch of wave-uniform control flow goes through this: a vector-masked
// loop exit lets single lanes drop out of a loop (and out of the lane-0 queue pop) if anything ever differs between lanes.
#define UNI_B(c) (__builtin_amdgcn_readfirstlane((c) ? 1 : 0) != 0)
__device__ __forceinline__ int  bcast(int v, int srcLane) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(srcLane)); }

// ---- DPP (data-parallel primitives) cross-lane moves: VALU-rate, no LDS crossbar round trip ---------------------
// gfx9/CDNA dpp_ctrl encodings: row_shr:n 0x110+n, wave_shl:1 0x130, wave_shr:1 0x138, row_bcast:15 0x142, row_bcast:31 0x143
template <int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ int dppMov(int oldv, int src)
{ return __builtin_amdgcn_update_dpp(oldv, src, CTRL, ROWMASK, 0xf, false); }
__device__ __forceinline__ int laneUp1(int v, int fill) { return dppMov<0x138>(fill, v); }     // lane i <- lane i-1 (lane 0 <- fill)
__device__ __forceinline__ int laneDown1(int v, int fill) { return dppMov<0x130>(fill, v); }   // lane i <- lane i+1 (lane 63 <- fill)
// inclusive prefix max over the 64 lanes (identity 0): row_shr 1,2,4,8 inside rows of 16, then row_bcast 15 / 31
__device__ __forceinline__ unsigned waveInclMaxU(unsigned v)
{
    unsigned t;
    t = (unsigned)dppMov<0x111>(0, (int)v); v = v > t ? v : t;
    t = (unsigned)dppMov<0x112>(0, (int)v); v = v > t ? v : t;
    t = (unsigned)dppMov<0x114>(0, (int)v); v = v > t ? v : t;
    t = (unsigned)dppMov<0x118>(0, (int)v); v = v > t ? v : t;
    t = (unsigned)dppMov<0x142, 0xa>(0, (int)v); v = v > t ? v : t;
    t = (unsigned)dppMov<0x143, 0xc>(0, (int)v); v = v > t ? v : t;
    return v;
}
// inclusive sum over the 64 lanes by DPP (VALU rate; __shfl_up is a trip through the LDS crossbar): row_shr 1, 2, 4, 8 inside rows of 16, then row_bcast 15 / 31
__device__ __forceinline__ uint32_t waveInclSumU(uint32_t v)
{
    v += (uint32_t)dppMov<0x111>(0, (int)v); v += (uint32_t)dppMov<0x112>(0, (int)v); v += (uint32_t)dppMov<0x114>(0, (int)v); v += (uint32_t)dppMov<0x118>(0, (int)v);
    v += (uint32_t)dppMov<0x142, 0xa>(0, (int)v); v += (uint32_t)dppMov<0x143, 0xc>(0, (int)v);
    return v;
}
__device__ __forceinline__ uint32_t waveTotalSumU(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)waveInclSumU(v), 63); }      // the sum over all lanes, as a scalar
// The ds_bpermute forms below are kept for code that is not on the per-row critical path.
__device__ __forceinline__ unsigned waveMaxU(unsigned v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { unsigned t = (unsigned)__shfl_xor((int)v, d, 64); v = v > t ? v : t; }
    return v;
}
__device__ __forceinline__ int waveMaxI(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { int t = __shfl_xor(v, d, 64); v = v > t ? v : t; }
    return v;
}
__device__ __forceinline__ int waveSumI(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// exclusive prefix max over lanes (identity 0)
__device__ __forceinline__ unsigned waveExclMaxU(unsigned v, int lane)
{ (void)lane; return (unsigned)laneUp1((int)waveInclMaxU(v), 0); }
// maximum over all lanes, returned as a scalar
__device__ __forceinline__ unsigned waveTotalMaxU(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)waveInclMaxU(v), 63); }

__device__ __forceinline__ uint8_t ref4(const uint8_t *bases, uint32_t off)          // getFrom4Code, Math.c:180-188
{ uint8_t b = bases[off >> 1]; return (off & 1) ? (uint8_t)(b & 0xF) : (uint8_t)(b >> 4); }

__device__ __forceinline__ int      fragQLen(int sqo, int eqo) { return 1 + eqo - sqo; }
__device__ __forceinline__ uint32_t absDiffU(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }
__device__ __forceinline__ uint32_t gapI(int lo, int hi) { return hi > lo ? (uint32_t)(hi - lo) - 1u : 0u; }      // calcGap, FragsClumps.inl:158
__device__ __forceinline__ uint32_t gapU(uint32_t lo, uint32_t hi) { return hi > lo ? (hi - lo) - 1u : 0u; }
__device__ __forceinline__ uint32_t ovlI(int lo, int hi) { return lo >= hi ? (uint32_t)(lo - hi) + 1u : 0u; }     // calcOverlap :159
__device__ __forceinline__ uint32_t ovlU(uint32_t lo, uint32_t hi) { return lo >= hi ? (lo - hi) + 1u : 0u; }

__device__ __forceinline__ uint32_t opMake(int code, int len) { return ((uint32_t)code << 16) | ((uint32_t)len & 0xFFFFu); }  // code = OP_*
__device__ __forceinline__ int      opCode(uint32_t o) { return (int)(o >> 16); }
__device__ __forceinline__ int      opLen(uint32_t o) { return (int)(o & 0xFFFFu); }
