// phase_lanes.h -- everything of the per-root work around the X-drop extensions (ext_lanes.h), for the default band:
//
//   phase 1 = alignClump up to the extensions (AlignHelpers.c:205-272, AlignExtFrag.cpp:164-234)
//     k_joint_counts   joints (fragment pairs) per root clump -> scan -> joint slots
//     k_p1_joints      lane per root: the exact-match extensions of every joint, then the joint's gap is classified:
//                      nothing / one D, I or R op / pure diagonal (see below) / a DP problem (sort key = strip width, rows)
//     [radix sort of the DP joints by size, so that the lanes of a wave run problems of the same shape]
//     k_gap_lanes<GW>  lane per DP joint: gapDPLane (the sequential recurrence, strip state and both sequences in LDS)
//     k_gap_wave       wave per DP joint for the few that exceed gapDPLane's limits (dp_wave.h)
//     k_p1_assemble    lane per root: edit list = M ops + joint ops, the clump's exact-match end extensions, the two
//                      X-drop extension problems for k_ext_rows
//   phase 3 = the tail of extendClumpForwardReverse + scoreClump / splitClump (AlignExtFrag.cpp:112-141, AlignHelpers.c:302-579)
//     k_p3_lanes       lane per root: merge the extension results, scoreClump; accepted clumps are written by their lane;
//                      for a root that needs splitClump the careful extensions it will ask for are listed (predictCarefulDPs)
//     [second k_ext_rows / k_ext_trace round on those problems]
//     k_split_lanes    (split_lanes.h) lane per split root: the splitClump state machine on the stored results
//     k_align_p3       wave per root (align.h state machine) for whatever k_split_lanes gives back
//
// "Pure diagonal": an equal-length gap of g bases whose diagonal has mm mismatches needs no DP when
//     mm * (MS + RC) <= MS + 2 * (GO + GE):
// every gapped path aligns at most g-1 pairs and opens at least one insertion and one deletion, so it scores at most
// MS*(g-1) - 2*(GO+GE), which is <= the diagonal's score at every prefix; the reference's DP (ties go to the diagonal: E and F
// win only with '>', SW.cpp:1036-1060) then returns exactly the diagonal.  Its work counters are charged as the reference
// would have counted them.
#pragma once
#include "ext_lanes.h"

// a root after phase 1, 32 bytes (two to the 64-byte sector; it used to be a whole 100-byte Frame, of which the phase-3 kernels read these fields)
struct RootState { uint32_t sro, listOff; int32_t score; uint16_t sqo, eqo, refLen, len; uint8_t status, pad[3]; uint32_t pad2[2]; };
enum { JK_NONE = 0, JK_D, JK_I, JK_R, JK_DIAG, JK_DP };
struct JointRec {                                           // 32 B
    uint32_t nsro, qBase; uint16_t nsqo, qGap, rGap; uint8_t kind, flags;   // flags: bit0 strand, bit1 banded
    uint32_t opsOff; uint16_t nOps, pad; int32_t score; uint32_t cells;
};
struct PhaseArgs {
    RootState *state; uint32_t *stateOps; unsigned int *stateOpsCount; uint32_t stateOpsCap;
    ExtProb *probs; unsigned long long *rowsBound;      // 2 per root
    const ExtRes *res; const uint32_t *extOps;          // extension results; extOps = the trace arena's base, which their op lists' offsets refer to (extOpsPtr)
    uint32_t *slowList; unsigned int *slowCount;        // roots (k_p3_lanes) or joints (k_gap_lanes) handed to the wave kernels
    int useList;                                        // k_align_p3: take roots from slowList[0 .. *slowCount)
    uint32_t rootBegin;                                 // k_p3_lanes: first root of the chunk (A.nRoots = its end)
    const uint32_t *p3Order;                            // k_p3_lanes: the chunk's roots by descending length of their merged edit list (k_p3_keys), or nullptr
    // splitClump in lanes (split_lanes.h): the careful extensions of the split roots are listed by k_p3_lanes, computed by a second
    // k_ext_rows / k_ext_trace round, and consumed by k_split_lanes
    uint32_t *memoKeys; unsigned int *memoCount; ExtProb *probs2; unsigned long long *rowsBound2; unsigned int *nProb2; uint32_t probs2Cap;
    // joints
    uint32_t *jointCount; const uint32_t *jointBase; JointRec *joints; uint32_t nJoints;
    uint32_t *sortKeys, *sortVals; const uint32_t *sortedVals; unsigned int *nDP, *nDPb; uint32_t band24;      // band24: k_gap_band_pk<24> runs (see k_gap_lanes)
    uint32_t *gapOps; unsigned int *gapOpsCount; uint32_t gapOpsCap;
    uint32_t *extKeys, *extVals;                        // k_ext_rows takes the problems longest-bound first (keys = 0xFFFF - qLen)
    uint8_t *gapScratch;                                // YD_GAP_SCRATCH bytes per k_gap_lanes thread (trace strip + op list of gapDPLane)
};

// ---- wave helpers ----
__device__ __forceinline__ unsigned waveSumU(unsigned v) { return (unsigned)waveSumI((int)v); }
// rank of this lane among the set lanes of `mask`, and one atomicAdd of the total by the first set lane
__device__ __forceinline__ unsigned waveReserve(unsigned long long mask, unsigned int *counter, int lane, unsigned each = 1u)
{
    if (!mask) return 0u;
    const int first = __builtin_ctzll(mask); unsigned base = 0;
    if (lane == first) base = atomicAdd(counter, each * (unsigned)__builtin_popcountll(mask));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, first);
    return base + each * (unsigned)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
}


// Work counters of a 256-thread block: the waves add their (wave-reduced) values to LDS, then one atomic per counter and
// block instead of one per wave (the counters are single L2 words).  Every thread of the block must call it.
template <int N> __device__ __forceinline__ void blockCounters(unsigned long long *const (&dst)[N], const unsigned (&val)[N])
{
    __shared__ unsigned sAcc[N];
    if (threadIdx.x < (unsigned)N) sAcc[threadIdx.x] = 0;
    __syncthreads();
    if (laneId() == 0) { for (int k = 0; k < N; k++) if (val[k]) atomicAdd(&sAcc[k], val[k]); }
    __syncthreads();
    if (threadIdx.x < (unsigned)N && sAcc[threadIdx.x]) atomicAdd(dst[threadIdx.x], (unsigned long long)sAcc[threadIdx.x]);
}

template <int N> __device__ __forceinline__ void blockCountersU32(unsigned int *const (&dst)[N], const unsigned (&val)[N])
{
    __shared__ unsigned sAcc32[N];
    if (threadIdx.x < (unsigned)N) sAcc32[threadIdx.x] = 0;
    __syncthreads();
    if (laneId() == 0) { for (int k = 0; k < N; k++) if (val[k]) atomicAdd(&sAcc32[k], val[k]); }
    __syncthreads();
    if (threadIdx.x < (unsigned)N && sAcc32[threadIdx.x]) atomicAdd(dst[threadIdx.x], sAcc32[threadIdx.x]);
}

// Arena space for a 256-thread block: the waves' totals meet in LDS, one atomic per block; returns this wave's base.
// Every thread of the block must call it.
__device__ __forceinline__ unsigned blockReserve(unsigned int *counter, unsigned waveTotal)
{
    __shared__ unsigned sTot[4], sBase;
    const int wv = (int)(threadIdx.x >> 6);
    if (laneId() == 0) sTot[wv] = waveTotal;
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned t = sTot[0] + sTot[1] + sTot[2] + sTot[3]; sBase = t ? atomicAdd(counter, t) : 0u; }
    __syncthreads();
    unsigned b = sBase; for (int k = 0; k < wv; k++) b += sTot[k];
    return b;
}

// ---- gap-fill DP of one lane (findAGSAlignment / findAGSAlignmentBanded, SW.cpp:462-477, 798-1208) ---------------------------
// The sequential recurrence, one problem per lane: strip state (PV/PF int32, PI uint8, GW+1 columns) and the reference
// segment in LDS laid out [column][lane] (conflict-free whatever column each lane is at), trace cells (op | run << 2, one
// byte) in a lane-private HBM strip written four cells per store, traceback as the reference's.  Two instances: GW = 16
// (14 KB of LDS per wave: most gaps) and GW = 32 (23 KB).  Limits: W <= GW, qLen <= 62, rLen <= 64; anything larger
// goes to the wave kernel.  Returns the ops in emission order (far end first) in tmp[0..nOps).
#define YD_GROWS 60
#define YD_GREF 64
#define YD_GQW 16                                // dwords of query codes (60 + alignment slack)
#define YD_GRW 10                                // dwords of packed reference (64 bases + alignment slack)
#define YD_GAP_SCRATCH 3072                    // per lane: (YD_GROWS + 1) * 32 trace bytes + 192 ops; a wave's 64 strips are interleaved dword by dword,
                                               // so that lanes at the same cell of their (size-sorted) problems store to one line
#define TR_U8 0xFFu
struct GapLaneMem { int *pv, *pf; uint8_t *pi; uint32_t *refw, *qw; uint8_t *T; uint32_t *tmp; };   // LDS pointers already offset by lane (stride 64); T, tmp in HBM
template <int GW>
__device__ __forceinline__ int gapDPLane(const DevParams &P, YD_GLOBAL const uint8_t *gB, YD_GLOBAL const uint8_t *q, bool banded,
                                         uint32_t rOff, int rLen, int qOff, int qLen, const GapLaneMem &M, int &nOps, unsigned &cellsOut)
{
    const int GO = P.GO, GE = P.GE, RC = P.RC, MS = P.MS, maxIntron = P.maxIntron, maxGapP = P.maxGap;
    int left = 0, right = 0;
    if (banded) { const int bw = P.bandWidth; if (rLen > qLen) { right = bw + (rLen - qLen); left = bw; } else { left = bw + (qLen - rLen); right = bw; } }
    const int W = banded ? left + right + 1 : rLen + 1;
    int *PV = M.pv, *PF = M.pf; uint8_t *PI = M.pi; YD_GLOBAL uint32_t *T32 = (YD_GLOBAL uint32_t *)toGlobal(M.T);
#define GPV(j) PV[(j) * 64]
#define GPF(j) PF[(j) * 64]
#define GPI(j) PI[(j) * 64]
    // both sequences come in as whole dwords issued back to back (one memory round trip each), then live in LDS
    const uint32_t rByte0 = (rOff >> 1) & ~3u; const int rNib0 = (int)(rOff - 2u * rByte0);             // nibble index of base 0 inside the dwords
    { YD_GLOBAL const uint32_t *g = (YD_GLOBAL const uint32_t *)(gB + rByte0); uint32_t w[YD_GRW];
#pragma unroll
      for (int k = 0; k < YD_GRW; k++) w[k] = g[k];
#pragma unroll
      for (int k = 0; k < YD_GRW; k++) M.refw[k * 64] = w[k]; }
    const size_t qAddr = (size_t)(q + qOff); const int qSkew = (int)(qAddr & 3u);
    { YD_GLOBAL const uint32_t *g = (YD_GLOBAL const uint32_t *)(qAddr - (size_t)qSkew); uint32_t w[YD_GQW];
#pragma unroll
      for (int k = 0; k < YD_GQW; k++) w[k] = g[k];
#pragma unroll
      for (int k = 0; k < YD_GQW; k++) M.qw[k * 64] = w[k]; }
    auto refAt = [&](int t) -> int { const int nb = rNib0 + t; const uint32_t w = M.refw[(nb >> 3) * 64]; return (int)((w >> (8 * ((nb >> 1) & 3) + ((nb & 1) ? 0 : 4))) & 15u); };
    auto qAt = [&](int t) -> int { const int bb = qSkew + t; return (int)((M.qw[(bb >> 2) * 64] >> (8 * (bb & 3))) & 0xFFu); };
    // row 0 (SW.cpp:905-935): U at the origin, deletions to its right
    const int startInit = banded ? left + 1 : 1;
    if (banded) { GPF(W) = YD_WORST; GPV(W) = YD_WORST; GPI(W) = 0; }
    {
        uint32_t acc = 0;
        for (int j = 0; j < W; j++) {
            uint32_t cell = 0;
            if (j == startInit - 1) cell = TR_U8; else if (j >= startInit) { const int dc = j - startInit + 1; cell = (uint32_t)(OP_D | (dc << 2)); GPV(j) = -(GO + dc * GE);
                GPF(j) = YD_WORST; GPI(j) = 0; }
            acc |= cell << (8 * (j & 3));
            if ((j & 3) == 3 || j == W - 1) { T32[(j >> 2) * 64] = acc; acc = 0; }
        }
    }
    GPF(startInit - 1) = 0; GPI(startInit - 1) = 0; GPV(startInit - 1) = 0;
    int V = 0, PVCol = YD_WORST, startCol = 1, endCol = W - 1; unsigned cells = 0;
    for (int i = 1; i <= qLen; i++) {
        int PDCol = 0, PECol = YD_WORST, bl = -1;                           // bl = the row's boundary insertion cell (SW.cpp:937-949)
        if (banded) {
            startCol = left + 1 - i;
            if (startCol <= 0) { startCol = 0; PVCol = YD_WORST; } else { PVCol = -(GO + i * GE); GPV(startCol - 1) = PVCol; bl = startCol - 1; }
            endCol = min(left + rLen - i, W - 1);
        } else { PVCol = -(GO + i * GE); bl = 0; }
        const int qc = qAt(i - 1);
        const int rRow = banded ? i - left - 1 : 0;
        uint32_t acc = bl >= 0 ? (uint32_t)(OP_I | (i << 2)) << (8 * (bl & 3)) : 0u;
        if (bl >= 0 && ((bl & 3) == 3 || startCol > endCol)) { T32[((i * GW + bl) >> 2) * 64] = acc; acc = 0; }
        for (int j = startCol; j <= endCol; j++) {
            const int RM = banded ? j : j - 1, IO = RM + 1; int op;
            V = GPV(RM);
            const int rc = refAt(banded ? rRow + j : j - 1);
            if (qc == rc) { V += MS; op = OP_M; } else { V -= RC; op = OP_R; }
            int len = 0;
            const int CE = PECol - GE, NE = PVCol - (GO + GE);
            if (CE >= NE && (PDCol + 1) <= maxIntron) { PECol = CE; PDCol = PDCol + 1; } else { PECol = NE; PDCol = 1; }
            if (PECol > V) { V = PECol; op = OP_D; len = PDCol; }
            int F, I; const int CF = GPF(IO) - GE, NF = GPV(IO) - (GO + GE), pio = (int)GPI(IO);
            if (CF >= NF && (pio + 1) <= maxGapP) { F = CF; I = pio + 1; } else { F = NF; I = 1; }
            if (F > V) { V = F; op = OP_I; len = I; }
            GPF(j) = F; GPI(j) = (uint8_t)I;
            acc |= (uint32_t)(op | (len << 2)) << (8 * (j & 3));
            if ((j & 3) == 3 || j == endCol) { T32[((i * GW + j) >> 2) * 64] = acc; acc = 0; }
            if (banded) GPV(j) = V; else GPV(j - 1) = PVCol;
            PVCol = V; cells++;
        }
        if (!banded) GPV(endCol) = V;
    }
    cellsOut = cells;
    // traceback from the end cell (SW.cpp:1138-1195)
    int x = banded ? right : W - 1, y = qLen;
    auto cellAt = [&](int yy, int xx) -> unsigned { const int c = yy * GW + xx; return (T32[(c >> 2) * 64] >> (8 * (c & 3))) & 0xFFu; };
    unsigned cell = cellAt(y, x);
    int prev = cell == TR_U8 ? -1 : (int)(cell & 3u), acc2 = 0, n = 0;
    for (int guard = 0; cell != TR_U8 && guard < 4096; guard++) {
        const int code = (int)(cell & 3u); int len = (int)(cell >> 2);
        if (banded) { if (code == OP_D) x -= len; else if (code == OP_I) { x += len; y -= len; } else { y -= 1; len = 1; } }
        else        { if (code == OP_D) x -= len; else if (code == OP_I) { y -= len; } else { x -= 1; y -= 1; len = 1; } }
        if (prev != code) { M.tmp[n * 64] = opMake(prev, acc2); n++; prev = code; acc2 = len; } else acc2 += len;
        if (y < 0 || x < 0 || x >= GW || n >= 190) break;
        cell = cellAt(y, x);
    }
    M.tmp[n * 64] = opMake(prev, acc2); n++;
    nOps = n;
#undef GPV
#undef GPF
#undef GPI
    return V;
}

__global__ void k_joint_counts(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < A.nRoots) X.jointCount[r] = YD_ROOT_REC(A, r).nFrags - 1u;
    if (r == A.nRoots) X.jointCount[r] = 0u;
}

// Exact-match run: the number of leading positions t in [0, maxLen) with q[qi + DIR*t] == reference base at ro + DIR*t (the reference's byte-by-byte loops of
// AlignHelpers.c:216-232 and AlignExtFrag.cpp:88-104).  Eight positions per step: 8 query codes (one per byte, < 16: include/yaha_hip.h) folded into a nibble
// stream, 8 reference nibbles (two per byte, high nibble first) swapped into the same order, one XOR, count of matching nibbles from the near end.  The
// byte loop had two dependent loads per base.  A window is only read when all eight positions are inside [0, maxLen): no access the byte loop would not make,
// except the bytes that complete the reference window's 8-byte load (inside the image's slack).
typedef unsigned long long yd_u64u __attribute__((aligned(1)));
template <int DIR>
__device__ __forceinline__ int matchRun(YD_GLOBAL const uint8_t *q, int qi, YD_GLOBAL const uint8_t *gB, uint32_t ro, int maxLen)
{
    int m = 0;
    while (maxLen - m >= 8) {
        const int qa = DIR > 0 ? qi + m : qi - m - 7; const uint32_t ra = DIR > 0 ? ro + (uint32_t)m : ro - (uint32_t)m - 7u;
        unsigned long long qx = *(YD_GLOBAL const yd_u64u *)(q + qa);
        qx = (qx | (qx >> 4)) & 0x00FF00FF00FF00FFull; qx = (qx | (qx >> 8)) & 0x0000FFFF0000FFFFull;
        const uint32_t qs = (uint32_t)(qx | (qx >> 16));
        unsigned long long rx = *(YD_GLOBAL const yd_u64u *)(gB + (ra >> 1));
        rx = ((rx & 0x0F0F0F0F0F0F0F0Full) << 4) | ((rx >> 4) & 0x0F0F0F0F0F0F0F0Full);
        const uint32_t rs = (uint32_t)(rx >> (4u * (ra & 1u)));
        const uint32_t diff = qs ^ rs;
        if (diff) return m + (DIR > 0 ? (__builtin_ctz(diff) >> 2) : (__builtin_clz(diff) >> 2));
        m += 8;
    }
    for (; m < maxLen; m++) {
        const uint32_t off = DIR > 0 ? ro + (uint32_t)m : ro - (uint32_t)m; const uint32_t b = gB[off >> 1];
        if ((uint32_t)q[DIR > 0 ? qi + m : qi - m] != ((off & 1u) ? (b & 15u) : (b >> 4))) break;
    }
    return m;
}

// order key of a DP joint, 12 bits (one bucket pass, scan.h): class : 2 | strip width : 5 | rows / 2 : 5, both clamped (the order only groups similar shapes; the
// class boundaries are exact).  Class 0 / 1: banded, within k_gap_band's limits, W <= 12 / 16; 2: other W <= 16; 3: the rest.  Joints without a DP: YD_JKEY_NONE, last.
#define YD_JKEY_NONE 0xFFFu
#define YD_JKEY_BITS 12
__host__ __device__ __forceinline__ uint32_t gapJointClass(uint32_t key) { return key >> 10; }
__host__ __device__ __forceinline__ uint32_t gapJointKey(const DevParams &P, bool banded, int qGap, int rGap)
{
    const int lenDiff = qGap > rGap ? qGap - rGap : rGap - qGap;
    const int W = banded ? 2 * P.bandWidth + lenDiff + 1 : rGap + 1;
    uint32_t cls = W <= 16 ? 2u : 3u;
    const bool lim = banded && P.bandWidth >= 5 && P.maxGap >= 16 && qGap <= YD_GROWS && rGap <= YD_GREF;      // what the register-strip band kernels take
    if (lim && W <= 16) cls = W <= 12 ? 0u : 1u;
    // the width field: W itself -- except in the last class, where the banded joints of W <= 24 inside the limits come FIRST (0 .. 7 = W - 17: k_gap_band_pk<24>,
    // gap_band_pk.h) and everything else behind them (8 .. 31)
    uint32_t wf = (uint32_t)(W < 31 ? W : 31);
    if (cls == 3u) wf = (lim && W <= 24) ? (uint32_t)(W - 17) : 8u + (uint32_t)(W - 17 < 23 ? W - 17 : 23);
    return (cls << 10) | (wf << 5) | (uint32_t)((qGap >> 1) < 30 ? (qGap >> 1) : 30);      // (0xFFF = class 3, width 31, rows 31 is kept for YD_JKEY_NONE)
}
__host__ __device__ __forceinline__ bool gapJointBand24(uint32_t key) { return (key >> 10) == 3u && ((key >> 5) & 31u) < 8u; }

// A lane-per-root kernel walks, in every wave, as long as the wave's longest root.  wgDeal hands the 256 roots of a workgroup to its lanes by ascending class (0 = the longest
// walk ... 63; 64 = no root): a counting sort in LDS, so that the waves of a workgroup hold roots of like length and three of four are done early.  The roots stay the
// workgroup's own (their records are in the lines its lanes have just read); nothing such a kernel writes may depend on which lane a root sits in.  Every thread of the
// workgroup calls it; returns the root for this lane, or 0xFFFFFFFF.  (Round 6: the step is bound by the sum of vector instructions, and the instructions idle lanes sit
// through are issued all the same -- k_p3_lanes 3.06 -> 2.62 ms, -0.55 ms a step, profiles/r06_rows_kernel_passes.txt.)
template <int BS = 256>
__device__ __forceinline__ uint32_t wgDeal(uint32_t root, uint32_t cls)
{
    __shared__ uint32_t sHist[65], sOrder[BS];
    const int lane = laneId();
    if (threadIdx.x < 65u) sHist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t inCls = atomicAdd(&sHist[cls], 1u);
    __syncthreads();
    if (threadIdx.x < 64u) {                                                 // exclusive sums of the 65 counts (the last class needs none behind it)
        const uint32_t c = sHist[lane]; uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += y; }
        sHist[lane] = incl - c; if (lane == 63) sHist[64] = incl;
    }
    __syncthreads();
    sOrder[sHist[cls] + inCls] = cls < 64u ? root : 0xFFFFFFFFu;
    __syncthreads();
    return sOrder[threadIdx.x];
}

// lane per root: exact-match extensions of every joint (AlignHelpers.c:216-232), then the gap's kind (AlignExtFrag.cpp:190-231)
__global__ void __launch_bounds__(256) k_p1_joints(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    // (wgDeal by the number of fragments, measured: 1.64 -> 1.61 ms here, 2.39 -> 2.66 ms in k_p1_assemble -- their per-root records are written side by side by
    // neighbouring lanes -- and the step 0.2 ms slower: not used)
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < A.nRoots; const DevParams &P = A.P;
    unsigned perfect = 0, touched = 0, nDP = 0, nDP16 = 0, nB12 = 0, nB16 = 0, nB24 = 0;
    if (live) {
        const ChainClumpRec rec = YD_ROOT_REC(A, r); const int n = (int)rec.nFrags;
        if (n > 1) {
            const uint32_t read = rec.rs >> 1, r0 = A.B.readOff[read];
            YD_GLOBAL const uint8_t *q = toGlobal((rec.rs & 1u) ? A.B.rev : A.B.fwd) + r0; YD_GLOBAL const uint8_t *gB = toGlobal(A.bases);
            auto refAt = [&](uint32_t off) -> uint32_t { const uint32_t b = gB[off >> 1]; return (off & 1u) ? (b & 15u) : (b >> 4); };
            DevFrag *F = A.clumpFrags + rec.fragOff; const uint32_t jb = X.jointBase[r];
            DevFrag cur = F[0];
            for (int k = 1; k < n; k++) {
                DevFrag nxt = F[k];
                int gap = (int)min(gapI(cur.eqo, nxt.sqo), gapU(cur.sro + (uint32_t)cur.refLen - 1u, nxt.sro));
                { const int c = matchRun<-1>(q, (int)nxt.sqo - 1, gB, nxt.sro - 1u, gap);
                  perfect += c; touched += c + (c < gap);
                  if (c > 0) { nxt.sqo = (uint16_t)(nxt.sqo - c); nxt.sro -= (uint32_t)c; nxt.refLen = (uint16_t)(nxt.refLen + c); } gap -= c; }
                { const uint32_t eRO = cur.sro + (uint32_t)cur.refLen - 1u; const int c = matchRun<1>(q, (int)cur.eqo + 1, gB, eRO + 1u, gap);
                  perfect += c; touched += c + (c < gap);
                  if (c > 0) { cur.eqo = (uint16_t)(cur.eqo + c); cur.refLen = (uint16_t)(cur.refLen + c); } }
                F[k - 1] = cur;
                const uint32_t eRO = cur.sro + (uint32_t)cur.refLen - 1u;
                const int qGap = (int)(gapI(cur.eqo, nxt.sqo) & 0xFFFF), rGap = (int)(gapU(eRO, nxt.sro) & 0xFFFF);
                JointRec j; j.nsro = eRO + 1u; j.qBase = r0; j.nsqo = (uint16_t)((cur.eqo + 1) & 0xFFFF); j.qGap = (uint16_t)qGap; j.rGap = (uint16_t)rGap;
                j.flags = (uint8_t)(rec.rs & 1u); j.opsOff = 0; j.nOps = 0; j.pad = 0; j.score = 0; j.cells = 0; j.kind = JK_NONE;
                uint32_t key = YD_JKEY_NONE;
                if (qGap == 0 && rGap == 0) { }
                else if (qGap == 0) j.kind = JK_D;
                else if (rGap == 0) j.kind = JK_I;
                else if (rGap == 1 && qGap == 1) j.kind = JK_R;
                else {
                    const int lenDiff = qGap > rGap ? qGap - rGap : rGap - qGap;
                    const bool banded = lenDiff + P.bandWidth * 2 + 1 < rGap;
                    j.flags |= banded ? 2u : 0u; j.kind = JK_DP;
                    if (qGap == rGap) {
                        int mm = 0;
                        for (int t = 0; t < qGap; t++) mm += (uint32_t)q[(int)cur.eqo + 1 + t] != refAt(eRO + 1u + (uint32_t)t);
                        if (mm * (P.MS + P.RC) <= P.MS + 2 * (P.GO + P.GE)) { j.kind = JK_DIAG; j.score = P.MS * (qGap - mm) - P.RC * mm; }
                    }
                    if (j.kind == JK_DP) { key = gapJointKey(P, banded, qGap, rGap); const uint32_t cls = gapJointClass(key); nDP++; nDP16 += cls <= 2u; nB12 += cls == 0u;
                        nB16 += cls <= 1u; nB24 += gapJointBand24(key) ? 1u : 0u; }
                }
                X.joints[jb + (uint32_t)(k - 1)] = j; X.sortKeys[jb + (uint32_t)(k - 1)] = key; X.sortVals[jb + (uint32_t)(k - 1)] = jb + (uint32_t)(k - 1);
                cur = nxt;
            }
            F[n - 1] = cur;
        }
    }
    perfect = waveSumU(perfect); touched = waveSumU(touched); nDP = waveSumU(nDP); nDP16 = waveSumU(nDP16); nB12 = waveSumU(nB12); nB16 = waveSumU(nB16); nB24 = waveSumU(nB24);
    { unsigned long long *const dst[2] = {&A.ctr->v[C_PERFECT], &A.ctr->v[C_TOUCHED]}; const unsigned val[2] = {perfect, touched}; blockCounters<2>(dst, val); }
    { unsigned int *const dst[5] = {X.nDP, X.nDP + 1, X.nDPb, X.nDPb + 1, X.nDPb + 2}; const unsigned val[5] = {nDP, nDP16, nB12, nB16, nB24}; blockCountersU32<5>(dst, val); }
}

// lane per DP joint, in size order (key = class, strip width, rows).  Persistent 64-thread blocks.  The sorted joints are [0, nB12) banded with W <= 12 and
// [nB12, nB16) banded with W <= 16: k_gap_band<12|16> (gap_band_lanes.h);  [nB16, n16) the other W <= 16: the GW = 16 instance of this kernel;  [n16, nDP): GW = 32
// (X.nDP[0] = nDP, X.nDP[1] = n16, X.nDPb[0] = nB12, X.nDPb[1] = nB16, counted by k_p1_joints); with X.band24 the first X.nDPb[2] joints of [n16, nDP) -- banded, W <= 24 --
// are k_gap_band_pk<24>'s (gap_band_pk.h) and GW = 32 starts behind them.
template <int GW>
__global__ void __launch_bounds__(64) k_gap_lanes(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    __shared__ int sPV[GW + 1][64], sPF[GW + 1][64];
    __shared__ uint8_t sPI[GW + 1][64]; __shared__ uint32_t sRefW[YD_GRW][64], sQW[YD_GQW][64];
    const int lane = laneId(); const DevParams &P = A.P;
    GapLaneMem GM; GM.pv = &sPV[0][lane]; GM.pf = &sPF[0][lane]; GM.pi = &sPI[0][lane]; GM.refw = &sRefW[0][lane]; GM.qw = &sQW[0][lane];
    { uint32_t *sp = (uint32_t *)(X.gapScratch + (size_t)blockIdx.x * 64u * YD_GAP_SCRATCH) + lane; GM.T = (uint8_t *)sp; GM.tmp = sp + (size_t)((YD_GROWS + 1) * 32 / 4) * 64; }
    YD_GLOBAL const uint8_t *gB = toGlobal(A.bases);
    const uint32_t nAll = X.nDP[0], n16 = X.nDP[1];
    const uint32_t tBegin = GW == 16 ? X.nDPb[1] : n16 + (X.band24 ? X.nDPb[2] : 0u), tEnd = GW == 16 ? n16 : nAll;
    for (uint32_t base = tBegin + blockIdx.x * 64u; base < tEnd; base += gridDim.x * 64u) {
        const uint32_t t = base + (uint32_t)lane; const bool live = t < tEnd;
        int nT = 0, score = 0; unsigned cells = 0; bool tooBig = false; uint32_t ji = 0;
        if (live) {
            ji = X.sortedVals[t]; const JointRec j = X.joints[ji];
            const bool banded = (j.flags & 2u) != 0; const int qGap = j.qGap, rGap = j.rGap;
            const int lenDiff = qGap > rGap ? qGap - rGap : rGap - qGap;
            const int W = banded ? 2 * P.bandWidth + lenDiff + 1 : rGap + 1;
            if (W > GW || qGap > YD_GROWS || rGap > YD_GREF) tooBig = true;
            else {
                YD_GLOBAL const uint8_t *q = toGlobal((j.flags & 1u) ? A.B.rev : A.B.fwd) + j.qBase;
                score = gapDPLane<GW>(P, gB, q, banded, j.nsro, rGap, (int)j.nsqo, qGap, GM, nT, cells);
            }
        }
        { const unsigned long long mm = __ballot(tooBig); const unsigned sl = waveReserve(mm, X.slowCount, lane); if (tooBig) X.slowList[sl] = ji; }
        // op slots: wave prefix sum of nT
        int incl = nT;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
        const int total = __shfl(incl, 63, 64); unsigned ob = 0;
        if (lane == 63 && total) ob = atomicAdd(X.gapOpsCount, (unsigned)total);
        ob = (unsigned)__shfl((int)ob, 63, 64);
        if (live && !tooBig) {
            const unsigned off = ob + (unsigned)(incl - nT);
            if ((unsigned long long)off + (unsigned)nT > (unsigned long long)X.gapOpsCap) atomicCAS(A.errFlag, 0, (int)YERR_OUT);
            else {
                for (int k = 0; k < nT; k++) X.gapOps[off + k] = GM.tmp[(nT - 1 - k) * 64];    // list order
                JointRec *jp = X.joints + ji; jp->opsOff = off; jp->nOps = (uint16_t)nT; jp->score = score; jp->cells = cells;
            }
        }
    }
}

// wave per DP joint (dp_wave.h) for the joints beyond gapDPLane's limits
__global__ void __launch_bounds__(64) k_gap_wave(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    const int lane = laneId();
    WaveMem M = carveScratch(A.scratch + (size_t)blockIdx.x * A.scratchPerWave, A.maxQ, A.traceRows, A.listCap, A.genCap);
    __shared__ uint16_t sTrace[YD_LDS_CELLS];
    int err = 0; WaveScratch S; S.ldsTrace = sTrace; S.trace = M.trace; S.traceRows = A.traceRows; S.tmpOps = M.tmpOps; S.tmpCap = 2 * A.maxQ + 512; S.gen = M.gen;
        S.genCap = A.genCap; S.err = &err;
    const unsigned n = uniU(*X.slowCount);
    for (;;) {
        if (__ballot(1) != ~0ull) { atomicCAS(A.errFlag, 0, (int)YERR_EXEC); break; }
        unsigned t = 0; if (lane == 0) t = atomicAdd(A.queueHead, 1u);
        const unsigned i = uniU(t);
        if (i >= n) break;
        const uint32_t ji = uniU(X.slowList[i]); const JointRec j = X.joints[ji];
        const uint8_t *q = ((j.flags & 1u) ? A.B.rev : A.B.fwd) + j.qBase;
        DPOut o = dpWave(A.P, A.bases, q, (j.flags & 2u) ? YGPU_DP_BANDED : YGPU_DP_FULL, j.nsro, (int)j.rGap, (int)j.nsqo, (int)j.qGap, S);
        if (UNI_B(err != 0)) { if (lane == 0) atomicCAS(A.errFlag, 0, err); break; }
        const int nT = uni(o.nOps); unsigned ob = 0;
        if (lane == 0) ob = atomicAdd(X.gapOpsCount, (unsigned)nT); ob = uniU(ob);
        if (UNI_B((unsigned long long)ob + (unsigned)nT > (unsigned long long)X.gapOpsCap)) { if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_OUT); break; }
        for (int k = lane; k < nT; k += 64) X.gapOps[ob + k] = dpOp(S, o, false, k);
        if (lane == 0) { JointRec *jp = X.joints + ji; jp->opsOff = ob; jp->nOps = (uint16_t)nT; jp->score = uni(o.score); jp->cells = (uint32_t)uni(o.cells); }
    }
}

// lane per root: the edit list, the clump's fragment, its exact-match end extensions (AlignExtFrag.cpp:76-107), the
// two X-drop extension problems
__global__ void __launch_bounds__(256) k_p1_assemble(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    const int lane = laneId(); const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < A.nRoots; const DevParams &P = A.P;
    ChainClumpRec rec; rec.nFrags = 0; rec.rs = 0; rec.fragOff = 0;
    if (live) rec = YD_ROOT_REC(A, r);
    const int n = (int)rec.nFrags; const uint32_t jb = live ? X.jointBase[r] : 0u;
    // exact upper bound of the list length
    unsigned want = 0;
    if (live) { want = (unsigned)n + 1u; for (int k = 0; k + 1 < n; k++) { const JointRec &j = X.joints[jb + (uint32_t)k];
        want += j.kind == JK_DP ? (unsigned)j.nOps : (j.kind == JK_DIAG ? (unsigned)j.qGap : (j.kind != JK_NONE ? 1u : 0u)); } }
    unsigned incl = want;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { unsigned t = (unsigned)__shfl_up((int)incl, d, 64); if (lane >= d) incl += t; }
    const unsigned total = (unsigned)__shfl((int)incl, 63, 64);
    const unsigned base = blockReserve(X.stateOpsCount, total);
    const unsigned slot = base + incl - want;
    unsigned perfect = 0, touched = 0, gapCalls = 0, gapRows = 0, gapCells = 0;
    if (live) {
        if ((unsigned long long)slot + want > (unsigned long long)X.stateOpsCap) atomicCAS(A.errFlag, 0, (int)YERR_OUT);
        else {
            const uint32_t read = rec.rs >> 1, r0 = A.B.readOff[read]; const int qlen = (int)(A.B.readOff[read + 1] - r0);
            YD_GLOBAL const uint8_t *q = toGlobal((rec.rs & 1u) ? A.B.rev : A.B.fwd) + r0; YD_GLOBAL const uint8_t *gB = toGlobal(A.bases);
            auto refAt = [&](uint32_t off) -> uint32_t { const uint32_t b = gB[off >> 1]; return (off & 1u) ? (b & 15u) : (b >> 4); };
            const DevFrag *F = A.clumpFrags + rec.fragOff;
            uint32_t *ops = X.stateOps + slot; int nOut = 0, pc = -1, pl = 0, score = 0;
            auto put = [&](int code, int len) { if (pc == code) { pl = (pl + len) & 0xFFFF; return; } if (pc >= 0) { ops[nOut] = opMake(pc, pl); nOut++; } pc = code;
                pl = len & 0xFFFF; };
            const DevFrag firstF = F[0]; DevFrag cur = firstF;
            for (int k = 1; k <= n; k++) {
                { const int ql = fragQLen(cur.sqo, cur.eqo); put(OP_M, ql); score += P.MS * ql; }
                if (k == n) break;
                const JointRec j = X.joints[jb + (uint32_t)(k - 1)]; const int qGap = j.qGap, rGap = j.rGap;
                if (j.kind == JK_D) { put(OP_D, rGap); score -= P.GO + rGap * P.GE; }
                else if (j.kind == JK_I) { put(OP_I, qGap); score -= P.GO + qGap * P.GE; }
                else if (j.kind == JK_R) { put(OP_R, 1); score -= P.RC; }
                else if (j.kind == JK_DIAG) {
                    const int g = qGap;
                    for (int t = 0; t < g; t++) put((uint32_t)q[(int)j.nsqo + t] == refAt(j.nsro + (uint32_t)t) ? OP_M : OP_R, 1);
                    score += j.score; gapCalls++; gapRows += (unsigned)g; touched += (unsigned)g;
                    if (P.bandWidth * 2 + 1 < g) { const int bw = P.bandWidth, W = 2 * bw + 1; for (int i = 1; i <= g; i++) { int sc = bw + 1 - i; if (sc < 0) sc = 0;
                        int ec = bw + g - i; if (ec > W - 1) ec = W - 1; if (ec >= sc) gapCells += (unsigned)(ec - sc + 1); } }
                    else gapCells += (unsigned)(g * g);
                } else if (j.kind == JK_DP) {
                    for (int t = 0; t < (int)j.nOps; t++) { const uint32_t op = X.gapOps[j.opsOff + (uint32_t)t]; put(opCode(op), opLen(op)); }
                    score += j.score; gapCalls++; gapRows += (unsigned)qGap; gapCells += j.cells; touched += (unsigned)rGap;
                }
                cur = F[k];
            }
            uint32_t sro = firstF.sro; int sqo = firstF.sqo, eqo = cur.eqo; int refLen = (int)((1u + (cur.sro + (uint32_t)cur.refLen - 1u) - firstF.sro) & 0xFFFFu);
            int firstAdd = 0;
            int backLen = (int)((uint32_t)sqo < sro ? (uint32_t)sqo : sro), forwLen;
            if (backLen > 0) {
                const int m = matchRun<-1>(q, sqo - 1, gB, sro - 1u, backLen);
                perfect += m; touched += m + (m < backLen);
                if (m > 0) { firstAdd = m; score += m * P.MS; backLen -= m; sqo -= m; sro -= (uint32_t)m; refLen = (refLen + m) & 0xFFFF; }
            }
            {
                const uint32_t eRO = sro + (uint32_t)refLen - 1u;
                const uint32_t qrem = (uint32_t)(((qlen - 1) - eqo) & 0xFFFF), rrem = P.maxROff - eRO;
                forwLen = (int)(qrem < rrem ? qrem : rrem);
                if (forwLen > 0) {
                    const int m = matchRun<1>(q, eqo + 1, gB, eRO + 1u, forwLen);
                    perfect += m; touched += m + (m < forwLen);
                    if (m > 0) { pl = (pl + m) & 0xFFFF; score += m * P.MS; forwLen -= m; eqo += m; refLen = (refLen + m) & 0xFFFF; }
                }
            }
            if (nOut == 0) pl = (pl + firstAdd) & 0xFFFF;
            ops[nOut] = opMake(pc, pl); nOut++;
            if (nOut > 1 && firstAdd) ops[0] = opMake(opCode(ops[0]), (opLen(ops[0]) + firstAdd) & 0xFFFF);
            RootState s; memset(&s, 0, sizeof s);
            s.sro = sro; s.sqo = (uint16_t)sqo; s.eqo = (uint16_t)eqo; s.refLen = (uint16_t)refLen; s.score = score; s.status = (rec.rs & 1u) ? stReversed : 0;
                s.len = (uint16_t)nOut;
            s.listOff = slot;
            X.state[r] = s;
            const uint32_t strand = (rec.rs & 1u) ? XP_STRAND : 0u; const bool vb = backLen >= P.minExtLength, vf = forwLen >= P.minExtLength;
            ExtProb pb; pb.qBase = r0; pb.rOff = sro - 1u; pb.qOff = (uint16_t)((sqo - 1) & 0xFFFF); pb.qLen = (uint16_t)(backLen & 0xFFFF);
                pb.flags = strand | XP_REV | (vb ? XP_VALID : 0u);
            ExtProb pf; pf.qBase = r0; pf.rOff = sro + (uint32_t)refLen; pf.qOff = (uint16_t)((eqo + 1) & 0xFFFF); pf.qLen = (uint16_t)(forwLen & 0xFFFF);
                pf.flags = strand | (vf ? XP_VALID : 0u);
            X.probs[2 * (size_t)r] = pb; X.probs[2 * (size_t)r + 1] = pf;
            // 16 bits: two radix passes (a valid problem has qLen >= 1)
            X.extKeys[2 * (size_t)r] = vb ? 0xFFFFu - pb.qLen : 0xFFFFu; X.extKeys[2 * (size_t)r + 1] = vf ? 0xFFFFu - pf.qLen : 0xFFFFu;
            X.extVals[2 * (size_t)r] = 2u * r; X.extVals[2 * (size_t)r + 1] = 2u * r + 1u;
            // trace blocks of 10 rows: up to 9 rows of phase in front, one spare row behind
            X.rowsBound[2 * (size_t)r] = vb ? (unsigned long long)((pb.qLen + 19u) / 10u) : 0ull;
                X.rowsBound[2 * (size_t)r + 1] = vf ? (unsigned long long)((pf.qLen + 19u) / 10u) : 0ull;
        }
    }
    perfect = waveSumU(perfect); touched = waveSumU(touched); gapCalls = waveSumU(gapCalls); gapRows = waveSumU(gapRows); gapCells = waveSumU(gapCells);
    { unsigned long long *c = A.ctr->v; unsigned long long *const dst[5] = {&c[C_PERFECT], &c[C_TOUCHED], &c[C_GAP_CALLS], &c[C_GAP_ROWS], &c[C_GAP_CELLS]};
      const unsigned val[5] = {perfect, touched, gapCalls, gapRows, gapCells}; blockCounters<5>(dst, val); }
}

// Phase 3 for roots that scoreClump accepts or rejects without a split (AlignHelpers.c:302-366): the merged edit list
// [backward extension ops][phase-1 ops][forward extension ops] is scanned in place; accepted clumps are written out by
// their lane.  A root that needs splitClump goes to slowList for k_align_p3.
// score and counts of one edit op without a branch a code (the lanes of a wave hold different codes: four branches were four masked passes over the same three
// instructions, and as many jumps)
__device__ __forceinline__ int opScore(const DevParams &P, int code, int len) { return len * (code == OP_M ? P.MS : (code == OP_R ? -P.RC : -P.GE)) + (code >= OP_D ? -P.GO : 0); }
#define YD_OP_COUNT(code, len, m, r, i, d) \
    do { m += (code) == OP_M ? (len) : 0; r += (code) == OP_R ? (len) : 0; i += (code) == OP_I ? (len) : 0; d += (code) == OP_D ? (len) : 0; } while (0)
typedef uint32_t yd_u32x4u __attribute__((ext_vector_type(4), aligned(4)));
struct MergedOps {
    const uint32_t *a, *b, *c; int na, nb, nc; int jab, jbc;      // junction merges (mergeEOLToFront / mergeEOLToBack, SW.cpp:151-261)
    // a = the backward extension's ops as k_ext_trace leaves them (list order reversed): list element k = a[na-1-k]
    // The lists are read through a window of four ops (one 16-byte load, aligned to the segment's start): a lane's list shares its lines with nobody, the
    // lanes of 20 waves per CU evict each other's lines between two of their dependent one-op reads, and every such read was a fetch from beyond L2.
    // (Reads up to 12 bytes past a segment's end: inside the arenas' slack.)
    mutable const uint32_t *wp = nullptr; mutable uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    __device__ __forceinline__ uint32_t fetch(const uint32_t *seg, int idx) const
    {
        const uint32_t *base = seg + (idx & ~3);
        if (base != wp) { const yd_u32x4u v = *(YD_GLOBAL const yd_u32x4u *)toGlobal(base); w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w; wp = base; }
        const int sl = idx & 3;
        return sl == 0 ? w0 : (sl == 1 ? w1 : (sl == 2 ? w2 : w3));
    }
    __device__ int count() const { return (na - jab) + nb + (nc - jbc); }
    // (no branches but the window's: segment and index by selects, the junctions' lengths -- lenA0 / lenC0, set by setJunctions -- added where they belong.  With the
    // three segment cases as branches an op cost ~70 vector and ~120 scalar instructions in forty basic blocks: k_p3_lanes 2.61 -> 2.01 ms with this form, round 6)
    int lenA0 = 0, lenC0 = 0;
    __device__ __forceinline__ void setJunctions()
    {
        jab = (na > 0 && opCode(a[0]) == opCode(b[0])) ? 1 : 0; lenA0 = jab ? opLen(a[0]) : 0;
        jbc = (nc > 0 && opCode(c[0]) == opCode(b[nb - 1])) ? 1 : 0; lenC0 = jbc ? opLen(c[0]) : 0;
    }
    // the same with a branch a case: what predictCarefulDPs keeps (k_p3_predict: 1.27 ms with this form, 1.46 with the selects)
    __device__ uint32_t at(int k) const
    {
        const int ka = na - jab;
        if (k < ka) return fetch(a, na - 1 - k);
        k -= ka;
        if (k < nb) {
            uint32_t op = fetch(b, k); int len = opLen(op);
            if (k == 0 && jab) len += lenA0;
            if (k == nb - 1 && jbc) len += lenC0;
            return opMake(opCode(op), len & 0xFFFF);
        }
        return fetch(c, k - nb + jbc);
    }
    __device__ __forceinline__ uint32_t atLean(int k) const
    {
        const int ka = na - jab, kb = ka + nb;
        const bool inA = k < ka, inB = !inA & (k < kb);
        const int idx = inA ? na - 1 - k : (inB ? k - ka : k - kb + jbc);
        const uint32_t op = fetch(inA ? a : (inB ? b : c), idx);
        const int add = ((inB & (k == ka)) ? lenA0 : 0) + ((inB & (k == kb - 1)) ? lenC0 : 0);
        return (op & 0xFFFF0000u) | ((op + (uint32_t)add) & 0xFFFFu);
    }
};
// Which careful extensions will splitClump ask for?  The frames it visits (the root, then recursively the head and tail
// remainders that still hold a seed-length match, AlignHelpers.c:374-557) and their cores depend only on the edit list as it is
// before any careful extension, so their X-drop problems (direction, rOff, qOff, qLen) can be listed ahead of time and run
// through the lane kernels.  k_split_lanes takes a stored result only when its own arguments match one exactly; a root with
// a request that is not in its list (a re-split after an extension, deeper recursion) goes to the wave kernel instead.
struct PredFrame { int start, len, sqo, eqo, refLen; uint32_t sro; int phase, minItem, maxItem, sQO, eQO; uint32_t sRO, eRO; };
__device__ inline int predictCarefulDPs(const DevParams &P, const MergedOps &L, int n0, int sqo0, int eqo0, uint32_t sro0, int refLen0,
                                        YD_GLOBAL const uint8_t *q, int qlen, YD_GLOBAL const uint8_t *gB, ExtProb *out, uint32_t qBase, uint32_t strand)
{
    PredFrame st[5]; int depth = 0, nOut = 0; const int wS = sqo0, wE = eqo0;
    PredFrame f; f.start = 0; f.len = n0; f.sqo = sqo0; f.eqo = eqo0; f.sro = sro0; f.refLen = refLen0; f.phase = 0;
    for (int guard = 0; guard < 64; guard++) {
        if (f.phase == 0) {                                                  // splitClumpHelper: the best-scoring core
            int matches = 0, mism = 0, ins = 0, del = 0, AGS = 0, maxAGS = -10000, maxItem = -1, minItem = -1, eQO = 0, sQO = 0; uint32_t eRO = 0, sRO = 0;
            for (int k = 0; k < f.len; k++) {
                const uint32_t op = L.at(f.start + k); const int code = opCode(op), len = opLen(op);
                YD_OP_COUNT(code, len, matches, mism, ins, del);
                AGS += opScore(P, code, len); if (AGS < 0) AGS = 0;
                if (AGS > maxAGS) { maxAGS = AGS; maxItem = k; eQO = (f.sqo + matches + mism + ins - 1) & 0xFFFF; eRO = f.sro + (uint32_t)(matches + mism + del) - 1u; }
            }
            AGS = maxAGS; matches = mism = ins = del = 0; int maxMatch = 0;
            for (int k = maxItem; k >= 0; k--) {
                const uint32_t op = L.at(f.start + k); const int code = opCode(op), len = opLen(op);
                YD_OP_COUNT(code, len, matches, mism, ins, del);
                AGS -= opScore(P, code, len); maxMatch = (code == OP_M && len > maxMatch) ? len : maxMatch;
                if (AGS <= 0) { minItem = k; sQO = (eQO - (matches + mism + ins - 1)) & 0xFFFF; sRO = eRO - (uint32_t)(matches + mism + del - 1); break; }
            }
            if (maxMatch < P.wordLen || minItem < 0) f.phase = 4;
            else {
                f.minItem = minItem; f.maxItem = maxItem; f.sQO = sQO; f.eQO = eQO; f.sRO = sRO; f.eRO = eRO; f.phase = 1;
                bool has = false;
                if (minItem != 0) for (int k = 0; k < minItem && !has; k++) { const uint32_t op = L.at(f.start + k); has = opCode(op) == OP_M && opLen(op) >= P.wordLen; }
                if (has) {
                    if (depth >= 5) return nOut;
                    PredFrame c; c.start = f.start; c.len = minItem; c.sqo = f.sqo; c.eqo = (sQO - 1) & 0xFFFF; c.sro = f.sro;
                        c.refLen = (int)((1u + (sRO - 1u) - f.sro) & 0xFFFFu); c.phase = 0;
                    st[depth++] = f; f = c;
                }
            }
            continue;
        }
        if (f.phase == 1) {                                                  // tail remainder
            f.phase = 2;
            const int t0 = f.maxItem + 1, tl = f.len - t0; bool has = false;
            if (f.maxItem != f.len - 1) for (int k = 0; k < tl && !has; k++) { const uint32_t op = L.at(f.start + t0 + k); has = opCode(op) == OP_M && opLen(op) >= P.wordLen; }
            if (has) {
                if (depth >= 5) return nOut;
                PredFrame c; c.start = f.start + t0; c.len = tl; c.sqo = (f.eQO + 1) & 0xFFFF; c.eqo = f.eqo; c.sro = f.eRO + 1u;
                c.refLen = (int)((1u + (f.sro + (uint32_t)f.refLen - 1u) - (f.eRO + 1u)) & 0xFFFFu); c.phase = 0;
                st[depth++] = f; f = c;
            }
            continue;
        }
        if (f.phase == 2) {                                                  // the core's careful extensions (extendClump, AlignExtFrag.cpp:64-156)
            int sqo = f.sQO, eqo = f.eQO, refLen = (int)((1u + f.eRO - f.sRO) & 0xFFFFu); uint32_t sro = f.sRO;
            const bool cutB = f.sQO != wS, cutF = f.eQO != wE;
            const bool goBack = cutB, goForw = cutF || !cutB;                // sic: forward also when neither end was cut
            if (goBack) {
                int backLen = (int)((uint32_t)sqo < sro ? (uint32_t)sqo : sro);
                if (backLen > 0) { const int m = matchRun<-1>(q, sqo - 1, gB, sro - 1u, backLen); backLen -= m; sqo -= m; sro -= (uint32_t)m; refLen = (refLen + m) & 0xFFFF; }
                if (backLen >= P.minExtLength && nOut < YD_MEMO) { ExtProb p; p.qBase = qBase; p.rOff = sro - 1u; p.qOff = (uint16_t)((sqo - 1) & 0xFFFF);
                    p.qLen = (uint16_t)(backLen & 0xFFFF); p.flags = strand | XP_REV | XP_VALID; out[nOut++] = p; }
            }
            if (goForw) {
                const uint32_t eRO = sro + (uint32_t)refLen - 1u;
                const uint32_t qrem = (uint32_t)(((qlen - 1) - eqo) & 0xFFFF), rrem = P.maxROff - eRO;
                int forwLen = (int)(qrem < rrem ? qrem : rrem);
                if (forwLen > 0) { const int m = matchRun<1>(q, eqo + 1, gB, eRO + 1u, forwLen); forwLen -= m; eqo += m; refLen = (refLen + m) & 0xFFFF; }
                if (forwLen >= P.minExtLength && nOut < YD_MEMO) { ExtProb p; p.qBase = qBase; p.rOff = sro + (uint32_t)refLen; p.qOff = (uint16_t)((eqo + 1) & 0xFFFF);
                    p.qLen = (uint16_t)(forwLen & 0xFFFF); p.flags = strand | XP_VALID; out[nOut++] = p; }
            }
            f.phase = 4; continue;
        }
        if (depth == 0) break;                                               // frame done
        f = st[--depth];
    }
    return nOut;
}

// a root after phase 2: its frame and the merged edit list [backward extension][phase-1 list][forward extension] (AlignExtFrag.cpp:112-141)
struct P3Root { MergedOps L; uint32_t sro; int sqo, eqo, refLen, status, score; };
__device__ __forceinline__ P3Root p3Merged(const PhaseArgs &X, uint32_t r)
{
    P3Root o; MergedOps &L = o.L; L.a = L.b = L.c = nullptr; L.na = L.nb = L.nc = L.jab = L.jbc = 0;
    const RootState S = X.state[r];
    o.sro = S.sro; o.sqo = S.sqo; o.eqo = S.eqo; o.refLen = S.refLen; o.status = S.status; o.score = S.score;
    L.b = X.stateOps + S.listOff; L.nb = S.len;
    const ExtRes rb = X.res[2 * (size_t)r], rf = X.res[2 * (size_t)r + 1];
    if (rb.score > 0) {                                                  // AlignExtFrag.cpp:112-125
        const int aQ = rb.maxi, aR = rb.maxi + (rb.maxj - YD_LBAND);
        L.a = extOpsPtr(X.extOps, rb); L.na = (int)rb.nOps;
        o.score += rb.score; o.sqo = (o.sqo - aQ) & 0xFFFF; o.sro -= (uint32_t)aR; o.refLen = (o.refLen + aR) & 0xFFFF;
    }
    if (rf.score > 0) {                                                  // AlignExtFrag.cpp:128-141
        const int aQ = rf.maxi, aR = rf.maxi + (rf.maxj - YD_LBAND);
        L.c = extOpsPtr(X.extOps, rf); L.nc = (int)rf.nOps;
        o.score += rf.score; o.eqo = (o.eqo + aQ) & 0xFFFF; o.refLen = (o.refLen + aR) & 0xFFFF;
    }
    L.setJunctions();
    return o;
}

// k_p3_lanes walks every root's merged edit list op by op, and a wave walks as long as its longest list: 57 ops a root on average, ~240 for the longest of 64 roots in
// rank order -- three quarters of the lane slots of its loop idle (round 6: 11 100 vector instructions a wave for 64 x 57 ops).  So the roots are taken by
// descending list length (one key per root, one bucket pass, scan.h): 4 095 - min(ops, 4 095).  The order of the roots does not matter to anything the kernel
// writes -- accepted clumps and split roots take their places by atomic reservation, and the final layout orders by (root rank, push number).
__global__ void __launch_bounds__(256) k_p3_keys(PhaseArgs X, uint32_t rootEnd, uint32_t *keys)
{
    YD_HIGH_PRIO();
    const uint32_t r = X.rootBegin + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rootEnd) return;
    const RootState S = X.state[r]; const ExtRes rb = X.res[2 * (size_t)r], rf = X.res[2 * (size_t)r + 1];
    const uint32_t n = (uint32_t)S.len + (rb.score > 0 ? rb.nOps : 0u) + (rf.score > 0 ? rf.nOps : 0u);
    keys[r - X.rootBegin] = 4095u - min(n, 4095u);
}

#ifndef YD_P3_LOCAL_SORT
#define YD_P3_LOCAL_SORT 1
#endif
#ifndef YD_P3_BS
#define YD_P3_BS 256
#endif
__global__ void __launch_bounds__(YD_P3_BS) k_p3_lanes(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    const int lane = laneId(); const uint32_t t = X.rootBegin + blockIdx.x * blockDim.x + threadIdx.x;
    const DevParams &P = A.P;
#if YD_P3_LOCAL_SORT
    // (wgDeal: by descending length of the merged edit list -- 57 ops a root on average, ~240 for the longest of 64 in rank order -- in fours)
    uint32_t r; bool live;
    {
        const bool in = t < A.nRoots;
        uint32_t cls = 64u;
        if (in) {
            const RootState S = X.state[t]; const ExtRes rb = X.res[2 * (size_t)t], rf = X.res[2 * (size_t)t + 1];
            const uint32_t n0 = (uint32_t)S.len + (rb.score > 0 ? rb.nOps : 0u) + (rf.score > 0 ? rf.nOps : 0u);
            cls = 63u - min(n0 >> 2, 63u);
        }
        r = wgDeal<YD_P3_BS>(t, cls); live = r != 0xFFFFFFFFu;
    }
#else
    const bool live = t < A.nRoots;
    const uint32_t r = (live && X.p3Order) ? X.p3Order[t - X.rootBegin] : t;
#endif
    int verdict = -1;                                  // -1 none, 0 rejected, 1 split needed, 2 scored
    MergedOps L; L.a = L.b = L.c = nullptr; L.na = L.nb = L.nc = L.jab = L.jbc = 0;
    uint32_t sro = 0; int sqo = 0, eqo = 0, refLen = 0, status = 0, n = 0;
    int matches = 0, mism = 0, ins = 0, del = 0, AGS = 0;
    if (live) {
        const P3Root R0 = p3Merged(X, r);
        L = R0.L; sro = R0.sro; sqo = R0.sqo; eqo = R0.eqo; refLen = R0.refLen; status = R0.status; const int score = R0.score;
        status |= stAligned;
        // scoreClump
        n = L.count(); const int aligned = score; int maxAGS = 0; verdict = 0;
        // (the walk without branches but the window's: MergedOps::at by selects, the four counts and the score by selects)
        {
            for (int k = 0; k < n; k++) {
                const uint32_t op = L.atLean(k); const int code = opCode(op), len = opLen(op);
                YD_OP_COUNT(code, len, matches, mism, ins, del);
                AGS += opScore(P, code, len);
                if (AGS <= 0 || (AGS >= aligned && k != n - 1)) { verdict = 1; break; }
                maxAGS = AGS > maxAGS ? AGS : maxAGS;
            }
        }
        if (verdict == 0) {
            if (matches >= P.minRawScore && maxAGS > AGS) verdict = 1;
            else if (matches >= P.minRawScore) {
                const int tot = (matches + mism + ins + del) & 0xFFFF;
                const double percent = (double)(matches & 0xFFFF) / (double)tot;
                if (!(percent < (double)P.minIdentity)) verdict = 2;
            }
        }
    }
    // a root that needs splitClump goes to the split list; k_p3_predict lists its careful extensions
    // (the split list in the order the workgroup's lanes hold the roots; giving the verdicts back to the roots' own lanes -- the list order of round 5 -- takes a barrier
    // at the end of the walk, behind which the early waves wait for the longest: 2.62 -> 3.23 ms for this kernel, and the step loses what the dealing gained)
    { const unsigned long long sm = __ballot(verdict == 1); const unsigned slot = waveReserve(sm, X.slowCount, lane); if (verdict == 1) X.slowList[slot] = r; }
    // emit the accepted clumps (emit() in align.h)
    const bool acc = verdict == 2;
    const unsigned long long am = __ballot(acc);
    unsigned nOpsTot = 0;
    if (am) {
        const unsigned ci = waveReserve(am, &A.outCounts[0], lane);
        int incl = acc ? n : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const int total = __shfl(incl, 63, 64); unsigned ob = 0;
        if (lane == 63) ob = atomicAdd(&A.outCounts[1], (unsigned)total);
        ob = (unsigned)__shfl((int)ob, 63, 64);
        const unsigned oi = ob + (unsigned)(incl - (acc ? n : 0));
        if (acc) {
            if (ci >= A.outClumpCap || (unsigned long long)oi + (unsigned)n > (unsigned long long)A.outOpsCap) atomicCAS(A.errFlag, 0, (int)YERR_OUT);
            else {
                const char codes[4] = {'M', 'R', 'D', 'I'};
                for (int k = 0; k < n; k++) { const uint32_t op = L.atLean(k); A.outOps[oi + k] = ((uint32_t)(uint8_t)codes[opCode(op) & 3] << 16) | (uint32_t)opLen(op); }
                ygpu_clump c; c.sro = sro; c.sqo = (uint16_t)sqo; c.eqo = (uint16_t)eqo; c.refLen = (uint16_t)refLen; c.totScore = (uint16_t)(AGS & 0xFFFF);
                c.totLength = (uint16_t)((matches + mism + ins + del) & 0xFFFF); c.matchedBases = (uint16_t)(matches & 0xFFFF); c.mismatchedBases = (uint16_t)(mism & 0xFFFF);
                c.gapBases = (uint16_t)((ins + del) & 0xFFFF); c.status = (uint8_t)(status | stScored); c.reserved = 0; c.op_start = oi; c.n_ops = (uint32_t)n;
                A.outClumps[ci] = c; A.outRoot[ci] = r; A.outPush[ci] = 0;
                A.rootPushCount[r] = 1; nOpsTot = (unsigned)n;
            }
        }
    }
    const unsigned nAcc = (unsigned)__builtin_popcountll(am); nOpsTot = waveSumU(nOpsTot);
    if (lane == 0 && nAcc) { atomicAdd(&A.ctr->v[C_SCORED], (unsigned long long)nAcc); atomicAdd(&A.ctr->v[C_OPS], (unsigned long long)nOpsTot); }
}


// lane per split root (compacted: in k_p3_lanes the 4 % of lanes that need this kept nearly every wave waiting for it): the careful extensions splitClump will
// ask for -> probs2 (the second k_ext_rows / k_ext_trace round) and the root's memo keys.  Persistent 64-thread blocks over the split list.
__global__ void __launch_bounds__(64) k_p3_predict(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    const int lane = laneId(); const DevParams &P = A.P; const uint32_t nSlow = *X.slowCount;
    for (uint32_t base = blockIdx.x * 64u; base < nSlow; base += gridDim.x * 64u) {
        const uint32_t slot = base + (uint32_t)lane; const bool live = slot < nSlow;
        ExtProb pp[YD_MEMO]; int np = 0;
        if (live) {
            const uint32_t r = X.slowList[slot]; const P3Root R0 = p3Merged(X, r);
            const ChainClumpRec rec = YD_ROOT_REC(A, r); const uint32_t r0 = A.B.readOff[rec.rs >> 1]; const int qlen = (int)(A.B.readOff[(rec.rs >> 1) + 1] - r0);
            np = predictCarefulDPs(P, R0.L, R0.L.count(), R0.sqo, R0.eqo, R0.sro, R0.refLen, toGlobal((rec.rs & 1u) ? A.B.rev : A.B.fwd) + r0, qlen, toGlobal(A.bases), pp, r0,
                (rec.rs & 1u) ? XP_STRAND : 0u);
        }
        int incl = np;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const int total = __shfl(incl, 63, 64); unsigned pb = 0;
        if (lane == 63 && total) pb = atomicAdd(X.nProb2, (unsigned)total);
        pb = (unsigned)__shfl((int)pb, 63, 64);
        if (live) {
            const unsigned first = pb + (unsigned)(incl - np); int kept = 0;
            for (int k = 0; k < np; k++) {
                const unsigned idx = first + (unsigned)k; if (idx >= X.probs2Cap) break;
                X.probs2[idx] = pp[k]; X.rowsBound2[idx] = (unsigned long long)((pp[k].qLen + 19u) / 10u);
                uint32_t *key = X.memoKeys + ((size_t)slot * YD_MEMO + (size_t)k) * 3;
                key[0] = pp[k].rOff; key[1] = (uint32_t)pp[k].qOff | ((uint32_t)pp[k].qLen << 16); key[2] = ((pp[k].flags & XP_REV) ? 1u : 0u) | (idx << 1);
                kept++;
            }
            X.memoCount[slot] = (unsigned)kept;
        }
    }
}

// merge the extension results, then scoreClump / splitClump as in k_align
__global__ void __launch_bounds__(64) k_align_p3(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    const unsigned wave = blockIdx.x; const int lane = laneId();
    WaveMem M = carveScratch(A.scratch + (size_t)wave * A.scratchPerWave, A.maxQ, A.traceRows, A.listCap, A.genCap);
    __shared__ uint16_t sTrace[YD_LDS_CELLS];
    Aligner al(A, M, sTrace);
    PROF_INIT();
    const unsigned nRoots = X.useList ? uniU(*X.slowCount) : uniU(A.nRoots);
    for (;;) {
        if (__ballot(1) != ~0ull) { atomicCAS(A.errFlag, 0, (int)YERR_EXEC); break; }
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(A.queueHead, 4u);
        const unsigned r0 = uniU(t);
        if (r0 >= nRoots) break;
        const unsigned r1 = min(r0 + 4u, nRoots);
        for (unsigned ri = r0; ri < r1; ri++) {
            const unsigned r = X.useList ? uniU(X.slowList[ri]) : ri;
            const ChainClumpRec rec = YD_ROOT_REC(A, r);
            al.setRead(rec); al.rootRank = r; al.pushes = 0;
            const RootState S0 = X.state[r]; const uint32_t listOff = uniU(S0.listOff);
            Frame f; memset(&f, 0, sizeof f); f.sro = S0.sro; f.sqo = S0.sqo; f.eqo = S0.eqo; f.refLen = S0.refLen; f.score = S0.score; f.status = S0.status; f.phase = PH_NONE;
            f.start = A.front; f.len = uni((int)S0.len);
            uint32_t *b = al.buf(0);
            for (int k = lane; k < f.len; k += 64) b[f.start + k] = X.stateOps[listOff + k];
            __threadfence_block();
            const ExtRes rb = X.res[2 * (size_t)r], rf = X.res[2 * (size_t)r + 1];
            int score = f.score;
            if (UNI_B(rb.score > 0)) {                                      // AlignExtFrag.cpp:112-125
                const int aQ = rb.maxi, aR = rb.maxi + (rb.maxj - YD_LBAND);
                al.mergeFrontSrc(b, f.start, f.len, extOpsPtr(X.extOps, rb), (int)rb.nOps, true);
                score += rb.score; f.sqo = (f.sqo - aQ) & 0xFFFF; f.sro -= (uint32_t)aR; f.refLen = (f.refLen + aR) & 0xFFFF;
            }
            if (UNI_B(rf.score > 0)) {                                      // AlignExtFrag.cpp:128-141
                const int aQ = rf.maxi, aR = rf.maxi + (rf.maxj - YD_LBAND);
                al.mergeBackSrc(b, f.start, f.len, extOpsPtr(X.extOps, rf), (int)rf.nOps);
                score += rf.score; f.eqo = (f.eqo + aQ) & 0xFFFF; f.refLen = (f.refLen + aR) & 0xFFFF;
            }
            f.score = uni(score); f.sqo = uni(f.sqo); f.eqo = uni(f.eqo); f.refLen = uni(f.refLen); f.sro = uniU(f.sro);
            f.status |= stAligned;
            if (!UNI_B(al.err != 0)) al.finishRoot(f);
            if (lane == 0) A.rootPushCount[r] = al.pushes;
            if (UNI_B(al.err != 0)) break;
        }
        if (UNI_B(al.err != 0)) { if (lane == 0) atomicCAS(A.errFlag, 0, al.err); break; }
    }
    al.flushCounters();
    PROF_FLUSH();
}
