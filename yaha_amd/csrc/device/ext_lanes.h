// ext_lanes.h -- the X-drop extensions of alignClump (reference extendClumpForwardReverse -> findAGSForward/BackwardExtension
// -> findAffineGapScore<banded, extension>, AlignExtFrag.cpp:109-141, SW.cpp:479-516, 798-1208) as ONE PROBLEM PER LANE.
//
// These two calls per root clump are >90 % of all DP cells of the hot path, and their strip is narrow (W = 4*BW + 1 = 21
// columns for the default -BW 5), so columns-as-lanes (dp_wave.h) leaves 2/3 of the wave idle and pays a cross-lane scan
// per row.  Here every lane runs the reference's sequential recurrence for its own problem with the whole strip
// (PV/PF[/PI] of 21 columns) in registers: no cross-lane traffic at all, the serial E/D chain is just program order, the
// run caps (maxIntron / maxGap) are applied exactly (compiled out when they cannot bind inside 21 columns).
//
//   k_ext_rows   forward pass, persistent lanes.  Per row and lane: 1 query code, 1 reference nibble, 21 cells, 12 bytes of
//                trace (4 bits per cell: op | E-run-continues | F-run-continues).  Lanes take new problems from a per-wave
//                pool of 64 pre-loaded problems (one atomic and one round of loads per 64), longest row bound first.
//                Trace rows are staged in LDS and leave as whole 128-byte blocks of 10 rows; all global stores of an
//                iteration are issued at its top.  It also stamps its own start / end time (wall_clock64) for bench.py.
//   k_ext_trace  lane per problem: walks the 4-bit cells back to the origin (run lengths are recovered from the continue
//                bits; the next 8 rows of a straight run are fetched together) and writes the ops INTO THE STRIP, over
//                rows the walk has already consumed.
//
// The kernel is specialised for the default band (-BW 5: bandwidth 10, W = 21, origin column 10) and needs maxGap >= 10.
// The reference's boundary insertions V(i, left - i) = -(GO + i*GE) are not special-cased: with PF(0, left) = -GO the
// ordinary F recurrence produces exactly that chain (F = -(GO + i*GE), I = i, op I), and every cell left of it stays at
// the "worst" sentinel, so all 21 columns run the same code on every row.  Only the columns left of the origin have to be
// kept out of the row maximum, and only in the first rows; the right edge is always inside the band (rLen = qLen + 2*BW).
// Other bands, and all gap-fill calls, are elsewhere (align.h / dp_wave.h, phase_lanes.h).
#pragma once
#include "align.h"

#define YD_REFILL_MIN 4                        // idle lanes of a wave before it runs a refill pass
#define YD_LW 21                               // register columns of the lane kernel = strip width for -BW 5
#define YD_LWORST (-(1 << 28))                 // sentinel: far below any reachable score (|score| < 2^23), no overflow when it decays

typedef uint32_t yd_u32x4 __attribute__((ext_vector_type(4)));
struct ExtProb { uint32_t qBase, rOff; uint16_t qOff, qLen; uint32_t flags; };           // 16 B; qBase = offset of the read in fwd/rev
enum { XP_STRAND = 1, XP_REV = 2, XP_VALID = 4 };
struct ExtRes { int score, maxi, maxj; uint32_t opsOff, nOps, rLen, rows, cells; };       // 32 B; maxj in register columns; rows/cells = work of this call

struct ExtArgs {
    DevParams P; const uint8_t *bases; const uint8_t *fwd, *rev;
    const ExtProb *probs; uint32_t nProb; const unsigned long long *stripOff; unsigned long long stripBase;
    const uint32_t *order;                      // problem indices in processing order (longest bound first), or nullptr
    unsigned long long *clock;                  // optional: [0] = earliest start, [1] = latest end of the launch in wall_clock64() ticks (100 MHz)
    uint32_t *trace;                            // 128-byte blocks of 10 rows (3 dwords each, 2 dwords of padding); stripOff counts blocks
    ExtRes *res; unsigned int *queue; DevCounters *ctr;      // ctr == nullptr: the consumer of the results accounts for the work (careful extensions)
    int *errFlag;
};

// CAPS = false when neither run cap can bind inside a 21-column strip (maxGap >= 21 and maxIntron >= 21: a run spans at most 20
// columns): the run-length state (PI, PD) is then dead and is compiled out.
// SECOND = the careful-extension round of splitClump (split_lanes.h): same code, its own kernel name in profiles, work counted by
// the consumer of the results.
template <bool CAPS, bool SECOND>
__global__ void __launch_bounds__(256) k_ext_rows(ExtArgs A)
{
    __shared__ uint32_t sBlk[32][256];          // per lane: the current 10-row trace block, [dword][thread] (conflict-free for any row slot)
    __shared__ uint32_t sList[4][3][64];        // per wave: the blocks to write out (owner thread, destination)
    if (!SECOND && A.clock && threadIdx.x == 0) atomicMin(&A.clock[0], (unsigned long long)wall_clock64());
    const int lane = laneId();
    const int GO = A.P.GO, GE = A.P.GE, GOE = A.P.GO + A.P.GE, RC = A.P.RC, MS = A.P.MS, XC = A.P.X, maxIntron = A.P.maxIntron, maxGap = A.P.maxGap;
    constexpr int bandwidth = YD_LBAND, leftR = YD_LBAND;
    const uint32_t maxROff = A.P.maxROff;
    YD_GLOBAL const uint8_t *gBases = toGlobal(A.bases);
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;

    // The strip state lives in registers and is written by the row code only (a fresh problem selects its row-0 values at
    // the top of its first row): one definition per loop iteration keeps the register allocator from duplicating it.
    int PV[YD_LW], PF[YD_LW], PI[YD_LW];
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    int p = -1, i = 0, qLen = 0, rLen = 0, maxScore = YD_LWORST, maxi = 0, maxj = 0, qStep = 0, qcNext = 0;
    uint32_t rOff = 0; bool rev = false, done = false;
    YD_GLOBAL const uint8_t *q = toGlobal(A.fwd); YD_GLOBAL uint32_t *strip = toGlobal(A.trace);
    unsigned calls = 0, rows = 0, cells = 0;
#pragma unroll
    for (int j = 0; j < YD_LW; j++) { PV[j] = YD_LWORST; PF[j] = YD_LWORST; PI[j] = 0; }

    // Pool of claimed problems: lane l holds entry l, completely set up (clamped lengths, first reference window, first
    // query base), so that handing an entry to an idle lane is eight cross-lane moves and no memory latency.  One atomic
    // and one round of dependent loads per 64 problems.
    unsigned poolBase = 0; int poolCount = 0, poolNext = 0; bool exhausted = false;
    uint32_t eLens = 0, eROff = 0, eQ = 0, eMisc = 0, eW1 = 0, eW2 = 0, eSLo = 0, eSHi = 0, ePidx = 0;
    // deferred stores (see the row code)
    bool pendFlush = false; YD_GLOBAL uint32_t *pendBlk = toGlobal(A.trace); int pendRes = -1, pendScore = 0, pendI = 0, pendJ = 0; unsigned pendRows = 0, pendCells = 0, pCells = 0;
    // The 10-row trace blocks of all lanes of a wave are in phase: row i of a problem that started when the wave was at row slot `phase` sits in
    // strip row i - 1 + phase, and every lane writes LDS row slot `wslot` in the same iteration.  All blocks of the wave are then complete in the
    // same iteration (wslot == 9) and leave together, once every ten rows; a problem that ends in between leaves its partial block behind
    // (`dirty`), which goes out with the next refill pass or the next common hand-over, whichever comes first.
    int wslot = 0, phase = 0; bool dirty = false; YD_GLOBAL uint32_t *curBlk = toGlobal(A.trace);

    // Finished blocks leave the wave together: the lanes that have one list it in LDS, then eight lanes write each block, 16 bytes apiece, so
    // that a store instruction carries whole 128-byte lines (a lane writing its own block alone sends eight 16-byte pieces in eight instructions).
    auto flushBlocks = [&]() {
        const unsigned long long f = __ballot(pendFlush);
        const int n = __builtin_popcountll(f);
        if (n == 0) return;
        const int tid = (int)threadIdx.x, wv = tid >> 6;
        if (pendFlush) {
            const int r = __builtin_popcountll(f & lanesBelow); const unsigned long long a = (unsigned long long)pendBlk;
            sList[wv][0][r] = (uint32_t)tid; sList[wv][1][r] = (uint32_t)a; sList[wv][2][r] = (uint32_t)(a >> 32);
            pendFlush = false;
        }
        __builtin_amdgcn_wave_barrier();                                     // LDS operations of a wave execute in order; keep the compiler from moving them
        const uint32_t *flat = &sBlk[0][0];
        const int piece = lane & 7;
        for (int g0 = 0; g0 * 8 < n; g0 += 2) {                              // two groups of eight blocks per pass: their LDS reads overlap
            uint32_t b[2]; unsigned long long a[2]; yd_u32x4 v[2]; bool on[2];
#pragma unroll
            for (int t = 0; t < 2; t++) { const int e = (g0 + t) * 8 + (lane >> 3); on[t] = e < n; const int ee = on[t] ? e : 0; b[t] = sList[wv][0][ee]; a[t] = (unsigned long long)sList[wv][1][ee] | ((unsigned long long)sList[wv][2][ee] << 32); }
#pragma unroll
            for (int t = 0; t < 2; t++) { v[t].x = flat[b[t] + (piece * 4 + 0) * 256]; v[t].y = flat[b[t] + (piece * 4 + 1) * 256]; v[t].z = flat[b[t] + (piece * 4 + 2) * 256]; v[t].w = flat[b[t] + (piece * 4 + 3) * 256]; }
#pragma unroll
            for (int t = 0; t < 2; t++) if (on[t]) *(YD_GLOBAL yd_u32x4 *)((YD_GLOBAL uint32_t *)a[t] + piece * 4) = v[t];
        }
        __builtin_amdgcn_wave_barrier();
    };
    bool firstFill = true;
    for (;;) {
        // ---- refill: until every lane is busy or nothing is left ----
        for (;;) {
            const unsigned long long need = __ballot(p < 0 && !done);
            if (!need) break;
            // a refill pass costs ~100 wave instructions whatever the number of lanes it serves: wait until a few are idle
            if (__builtin_popcountll(need) < YD_REFILL_MIN && __ballot(p >= 0) != 0ull && !firstFill) break;
            if (p < 0 && dirty) { pendFlush = true; dirty = false; }         // the lanes about to start a problem still hold the last rows of their previous one
            flushBlocks();
            if (poolNext >= poolCount) {                                     // wave-uniform: claim and set up the next 64 problems
                unsigned base = 0;
                if (!exhausted) { if (lane == 0) base = atomicAdd(A.queue, 64u); base = uniU(base); if (base >= A.nProb) exhausted = true; }
                if (exhausted) { if (p < 0) done = true; break; }
                poolBase = base; poolCount = (int)min(64u, A.nProb - base); poolNext = 0;
                eLens = 0;
                if (lane < poolCount) {
                    const unsigned np = A.order ? A.order[base + (unsigned)lane] : base + (unsigned)lane;
                    ePidx = np;
                    const ExtProb pr = A.probs[np];
                    int ql = 0; uint32_t rl = 0; const bool rv_ = (pr.flags & XP_REV) != 0;
                    if (pr.flags & XP_VALID) {                              // findAGSExtension, SW.cpp:479-516
                        calls++;
                        ql = pr.qLen;
                        rl = (uint32_t)(ql + bandwidth);
                        if (rv_ && rl > pr.rOff) { rl = pr.rOff + 1; ql = (int)rl - bandwidth; }
                        if (!rv_ && (pr.rOff + rl) > maxROff) { rl = maxROff - pr.rOff; ql = (int)rl - bandwidth; }
                        if (ql > 0) { ql &= 0xFFFF; rl &= 0xFFFF; }
                    }
                    if (ql <= 0) { ExtRes r; r.score = 0; r.maxi = r.maxj = 0; r.opsOff = r.nOps = 0; r.rLen = 0; r.rows = r.cells = 0; A.res[np] = r; }
                    else {
                        eLens = (uint32_t)ql | (rl << 16); eROff = pr.rOff; eQ = pr.qBase + pr.qOff;
                        YD_GLOBAL const uint8_t *qp = toGlobal((pr.flags & XP_STRAND) ? A.rev : A.fwd) + eQ;
                        eMisc = (pr.flags & 3u) | ((uint32_t)qp[0] << 8);
                        const unsigned long long so = A.stripOff[np] - A.stripBase; eSLo = (uint32_t)so; eSHi = (uint32_t)(so >> 32);
                        // reference window of row 1: register column c holds reference index c - leftR
                        eW1 = 0; eW2 = 0;
                        for (int c = leftR; c < YD_LW; c++) {
                            const int idx = c - leftR; uint32_t nib = 15u;
                            if (idx < (int)rl) { const uint32_t off = rv_ ? pr.rOff - (uint32_t)idx : pr.rOff + (uint32_t)idx; const uint32_t b = gBases[off >> 1]; nib = (off & 1u) ? (b & 15u) : (b >> 4); }
                            const uint32_t sh = (uint32_t)(c & 7) * 4u;
                            if (c < 16) eW1 |= nib << sh; else eW2 |= nib << sh;
                        }
                    }
                }
            }
            // the k-th idle lane takes entry poolNext + k
            const int nNeed = __builtin_popcountll(need), avail = poolCount - poolNext;
            const int e = poolNext + __builtin_popcountll(need & lanesBelow);
            const bool take = (p < 0 && !done) && e < poolCount;
            const int src = take ? e : lane;
            const uint32_t gLens = (uint32_t)__shfl((int)eLens, src, 64), gROff = (uint32_t)__shfl((int)eROff, src, 64), gQ = (uint32_t)__shfl((int)eQ, src, 64), gMisc = (uint32_t)__shfl((int)eMisc, src, 64);
            const uint32_t gW1 = (uint32_t)__shfl((int)eW1, src, 64), gW2 = (uint32_t)__shfl((int)eW2, src, 64), gSLo = (uint32_t)__shfl((int)eSLo, src, 64), gSHi = (uint32_t)__shfl((int)eSHi, src, 64);
            const uint32_t gPidx = (uint32_t)__shfl((int)ePidx, src, 64);
            const bool init = take && gLens != 0u;
            // row 0 of the strip (SW.cpp:905-935; PF(0, left) = -GO, see the header) for the lanes that start a problem: plain selects,
            // so that the state registers have one definition here and one in the row code
#pragma unroll
            for (int j = 0; j < YD_LW; j++) {
                const int iV = j == leftR ? 0 : (j > leftR ? -(GO + (j - leftR) * GE) : YD_LWORST), iF = j == leftR ? -GO : YD_LWORST;
                PV[j] = init ? iV : PV[j]; PF[j] = init ? iF : PF[j]; if (CAPS) PI[j] = init ? 0 : PI[j];
            }
            if (init) {
                p = (int)gPidx; qLen = (int)(gLens & 0xFFFFu); rLen = (int)(gLens >> 16); i = 0; maxScore = YD_LWORST; maxi = 0; maxj = 0;
                rev = (gMisc & XP_REV) != 0; rOff = gROff; pCells = 0;
                q = toGlobal((gMisc & XP_STRAND) ? A.rev : A.fwd) + gQ; qStep = rev ? -1 : 1; qcNext = (int)((gMisc >> 8) & 0xFFu);
                strip = toGlobal(A.trace) + (((unsigned long long)gSHi << 32) | gSLo) * 32ull; curBlk = strip; phase = wslot;
                w0 = 0; w1 = gW1; w2 = gW2;
            }
            poolNext += nNeed < avail ? nNeed : avail;
        }
        firstFill = false;
        if (__ballot(p >= 0) == 0ull) break;

        // ---- one DP row in every lane (lanes without a problem run on idle state; their stores are masked) ----
        // All memory operations of an iteration are issued here at the top: the previous row's trace cells and a finished
        // problem's result (both deferred), and the loads the row needs at its END (next query base, next reference base).
        // The wait the compiler puts at the loop header then finds them ~1000 instructions old.
        flushBlocks();                                                       // deferred from the previous row
        if (pendRes >= 0) {
            ExtRes r; r.score = pendScore > 0 ? pendScore : 0; r.maxi = pendI; r.maxj = pendJ; r.opsOff = 0; r.nOps = 0; r.rLen = pendRows >> 20; r.rows = pendRows & 0xFFFFFu; r.cells = pendCells;      // rLen: the strip row of row 1
            A.res[pendRes] = r; pendRes = -1;
        }
        const bool busy = p >= 0;
        ++i;
        const int qc = qcNext;
        { const int ni = i < qLen ? i : (qLen > 0 ? qLen - 1 : 0); qcNext = (int)q[ni * qStep]; }   // next row's query base
        uint32_t nbByte, nbOdd;                                                                     // next row's top reference base (index i + right)
        { const int idx = i + bandwidth; const bool in = busy && idx < rLen; const uint32_t off = in ? (rev ? rOff - (uint32_t)idx : rOff + (uint32_t)idx) : 0u;
          nbByte = gBases[off >> 1]; nbOdd = in ? (off & 1u) : 2u; }
        // Real cells of row i: columns startCol = max(left + 1 - i, 0) .. endCol.  findAGSExtension always passes rLen = qLen + 2*BW
        // (both clamps keep that relation, SW.cpp:499-516), so endCol = min(left + rLen - i, W - 1) = W - 1 on every row i <= qLen:
        // only the first `left` rows have cells to keep out of the row maximum, and only in the columns left of the origin.
        int sc = leftR + 1 - i; if (sc < 0) sc = 0;
        if (busy) { const unsigned nc = (unsigned)(YD_LW - sc); rows++; cells += nc; pCells += nc; }
        int PVCol = YD_LWORST, PE = YD_LWORST, PD = 0;
        uint32_t t0 = 0, t1 = 0, t2 = 0, rowKey = 0;
        int dV = PV[0];
#pragma unroll
        for (int j = 0; j < YD_LW; j++) {
            const uint32_t wsrc = j < 8 ? w0 : (j < 16 ? w1 : w2);
            const int rc = (int)((wsrc >> ((j & 7) * 4)) & 15u);
            const bool eq = rc == qc;
            int V = dV + (eq ? MS : -RC);
            const int CE = PE - GE, NE = PVCol - GOE;
            const bool cE = CE >= NE && (!CAPS || PD < maxIntron);
            PE = cE ? CE : NE; if (CAPS) PD = cE ? PD + 1 : 1;
            const bool tE = PE >= V; V = tE ? PE : V;
            int upV, upF, upI;
            if (j + 1 < YD_LW) { upV = PV[j + 1]; upF = PF[j + 1]; upI = PI[j + 1]; } else { upV = YD_LWORST; upF = YD_LWORST; upI = 0; }
            const int CF = upF - GE, NF = upV - GOE;
            const bool cF = CF >= NF && (!CAPS || upI < maxGap);
            const int F = cF ? CF : NF, I = CAPS ? (cF ? upI + 1 : 1) : 0;
            const bool tF = F >= V; V = tF ? F : V;
            uint32_t nib = eq ? (uint32_t)OP_M : (uint32_t)OP_R; nib = tE ? (uint32_t)OP_D : nib; nib = tF ? (uint32_t)OP_I : nib;
            nib |= (cE ? 4u : 0u) | (cF ? 8u : 0u);
            if (j < 8) { t0 |= nib << ((j & 7) * 4); asm volatile("" : "+v"(t0)); } else if (j < 16) { t1 |= nib << ((j & 7) * 4); asm volatile("" : "+v"(t1)); } else { t2 |= nib << ((j & 7) * 4); asm volatile("" : "+v"(t2)); }   // pinned: the condition masks die here
            // row-major first maximum over the real cells: key = (V + BIAS) << 5 | (31 - j)
            uint32_t key = ((uint32_t)(V + YD_BIAS)) << 5 | (uint32_t)(31 - j);
            if (j < leftR) key = j >= sc ? key : 0u;
            rowKey = key > rowKey ? key : rowKey;
            PV[j] = V; PF[j] = F; PI[j] = I; PVCol = V;
            dV = upV;                                                        // the next column's diagonal predecessor
            __builtin_amdgcn_sched_barrier(0);                               // keep the cells in program order: their many condition masks stay short-lived
        }
        { const int slot = wslot * 3, tid = (int)threadIdx.x;               // this row's cells go to the lane's LDS block (the same row slot in every lane)
          sBlk[slot][tid] = t0; sBlk[slot + 1][tid] = t1; sBlk[slot + 2][tid] = t2; }
        int rv = YD_LWORST, rj = 0;
        if (rowKey) { rv = (int)(rowKey >> 5) - YD_BIAS; rj = 31 - (int)(rowKey & 31u); }
        if (rv > maxScore) { maxScore = rv; maxi = i; maxj = rj; }
        // slide the window: column c takes column c+1, the top column takes the new base
        const uint32_t nb = nbOdd == 2u ? 15u : (nbOdd ? (nbByte & 15u) : (nbByte >> 4));
        w0 = (w0 >> 4) | (w1 << 28); w1 = (w1 >> 4) | (w2 << 28); w2 = (w2 >> 4) | (nb << 16);
        const bool fin = busy && (rv < maxScore - XC || i >= qLen);
        if (busy) { dirty = true; pendBlk = curBlk; }
        if (wslot == 9) { pendFlush = dirty; dirty = false; if (busy) curBlk += 32; wslot = 0; } else wslot++;      // wave-uniform
        if (fin) {
            pendRes = p; pendScore = maxScore; pendI = maxi; pendJ = maxj; pendRows = (unsigned)i | ((unsigned)phase << 20); pendCells = pCells;
            p = -1; qStep = 0; rLen = 0; qLen = 0; i = 0;
        }
    }
    // the last deferred stores
    if (dirty) pendFlush = true;
    flushBlocks();
    if (pendRes >= 0) {
        ExtRes r; r.score = pendScore > 0 ? pendScore : 0; r.maxi = pendI; r.maxj = pendJ; r.opsOff = 0; r.nOps = 0; r.rLen = pendRows >> 20; r.rows = pendRows & 0xFFFFFu; r.cells = pendCells;
        A.res[pendRes] = r;
    }
    if (!SECOND && A.clock && lane == 0) atomicMax(&A.clock[1], (unsigned long long)wall_clock64());
    // work counters
    unsigned c0 = (unsigned)waveSumI((int)calls), c1 = (unsigned)waveSumI((int)rows);
    unsigned long long cc = cells;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { cc += (unsigned long long)__shfl_xor((long long)cc, d, 64); }
    if (lane == 0 && !SECOND && A.ctr) {
        unsigned long long *c = A.ctr->v;
        atomicAdd(&c[C_EXT_CALLS], (unsigned long long)c0); atomicAdd(&c[C_EXT_ROWS], (unsigned long long)c1); atomicAdd(&c[C_EXT_CELLS], cc);
        atomicAdd(&c[C_TOUCHED], (unsigned long long)c1 + (unsigned long long)c0 * (unsigned long long)(4 * A.P.bandWidth + 1));
    }
}

// ---- traceback, lane per problem (SW.cpp:1138-1195) -----------------------------------------------------------------
struct ExtRowBits { uint32_t a, b, c; };
__device__ __forceinline__ ExtRowBits extLoadRow(YD_GLOBAL const uint32_t *strip, int y)
{ ExtRowBits r; YD_GLOBAL const uint32_t *t = strip + (size_t)((y - 1) / 10) * 32u + (size_t)((y - 1) % 10) * 3u; r.a = t[0]; r.b = t[1]; r.c = t[2]; return r; }
__device__ __forceinline__ uint32_t extNib(const ExtRowBits &r, int x) { const uint32_t w = x < 8 ? r.a : (x < 16 ? r.b : r.c); return (w >> ((x & 7) * 4)) & 15u; }

// Walks from (y, x) back to the origin (0, leftR).  The ops are written INTO THE STRIP, over rows the walk has already
// consumed: emission k (far end first) goes to dword E-1-k, E = end of row maxi+1 (the strip has one spare row).  After
// k rows are consumed at most 2k+1 ops exist (every op but a D needs a row of its own, and two D ops never touch) and
// 3k+3 dwords are free, so the walk never overwrites a row it still has to read.  The result is the ascending array
// strip[opsOff .. opsOff+nOps): the forward extension's list in order (ops are added to the front, SW.cpp:1186), the
// backward extension's list reversed (added to the back, SW.cpp:1190).
__device__ __forceinline__ int extRowWord(int y) { return ((y - 1) / 10) * 32 + ((y - 1) % 10) * 3; }
#define YD_TRACE_DEPTH 8
__global__ void __launch_bounds__(256) k_ext_trace(ExtArgs A)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.nProb) return;
    ExtRes r = A.res[p];
    if (r.score <= 0) return;
    YD_GLOBAL uint32_t *strip = toGlobal(A.trace) + (A.stripOff[p] - A.stripBase) * 32ull;
    constexpr int leftR = YD_LBAND;
    const int ph = (int)r.rLen;                                              // row i of the problem is strip row i - 1 + ph (k_ext_rows keeps the blocks of a wave in phase)
    const int E = extRowWord(r.maxi + 1 + ph) + 3;
    int y = r.maxi, x = r.maxj, prev = -1, acc = 0, n = 0; bool bad = false;
    // w = word offset of row y inside the strip, rr = its row inside the 10-row block (kept incrementally: no divisions in the loops)
    int rr = (y - 1 + ph) % 10, w = ((y - 1 + ph) / 10) * 32 + rr * 3;
    auto flush = [&]() { const int wp = E - 1 - n; if (wp < w + 3) bad = true; else strip[wp] = opMake(prev, acc); n++; };   // rows above row y are consumed
    auto put = [&](int code, int len) { if (prev != code) { if (prev >= 0) flush(); prev = code; acc = len; } else acc += len; };
    auto stepUp = [&](int &ww, int &r2) { if (r2 == 0) { r2 = 9; ww -= 5; } else { r2--; ww -= 3; } };                        // one row towards the origin
    // The kernel is bound by the latency of dependent loads (one per path cell).  Most of a path is straight runs of M / R cells
    // in one column, so the cells of the next YD_TRACE_DEPTH rows in column x are fetched together (one dword each) and consumed in turn.
    for (int guard = 0; guard < 70000 && y > 0 && x >= 0 && x < YD_LW; guard++) {
        const int ws = x >> 3, sh = (x & 7) * 4;
        uint32_t d[YD_TRACE_DEPTH];
        { int wk = w, rk = rr;
#pragma unroll
          for (int k = 0; k < YD_TRACE_DEPTH; k++) { d[k] = strip[wk + ws]; if (y > k + 1) stepUp(wk, rk); } }
        uint32_t nib = (d[0] >> sh) & 15u; int took = 0;
#pragma unroll
        for (int k = 0; k < YD_TRACE_DEPTH; k++) {
            nib = (d[k] >> sh) & 15u;
            const int op = (int)(nib & 3u);
            if (op >= OP_D || y <= 0) break;
            if (prev != op) { if (prev >= 0) flush(); prev = op; acc = 1; } else acc++;
            y--; stepUp(w, rr); took++;
        }
        if (took == YD_TRACE_DEPTH || y <= 0) continue;                                   // still in a straight run (or at the origin row)
        if ((nib & 3u) == (uint32_t)OP_D) {                                  // deletion run: walk the continue bits along the row
            ExtRowBits rb; rb.a = strip[w]; rb.b = strip[w + 1]; rb.c = strip[w + 2];
            int run = 1, xx = x;
            while (extNib(rb, xx) & 4u) { xx--; if (xx < 0) break; run++; }
            put(OP_D, run); x -= run;
        } else {                                                            // insertion run: walk the continue bits up and to the right
            int run = 1, yy = y, xx = x, ww = w, q2 = rr; uint32_t nb2 = nib;
            while (nb2 & 8u) {
                yy--; xx++; if (yy <= 0 || xx >= YD_LW) break;
                run++; stepUp(ww, q2);
                nb2 = (strip[ww + (xx >> 3)] >> ((xx & 7) * 4)) & 15u;
            }
            put(OP_I, run); for (int t = 0; t < run; t++) stepUp(w, rr); y -= run; x += run;
        }
    }
    if (y <= 0 && x > leftR) put(OP_D, x - leftR);                           // row 0: deletions back to the origin (SW.cpp:905-935)
    if (prev >= 0) { const int wp = E - 1 - n; if (wp < 0) bad = true; else strip[wp] = opMake(prev, acc); n++; }
    if (bad) { atomicCAS(A.errFlag, 0, (int)YERR_TRACE); return; }
    r.opsOff = (uint32_t)(E - n); r.nOps = (uint32_t)n; A.res[p] = r;
}

// order values are global problem indices; a chunk's kernels index from the chunk's first problem
__global__ void k_rebase_u32(uint32_t *v, uint32_t n, uint32_t sub)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] -= sub;
}

// sort keys for a problem list that has none yet: longest row bound first
__global__ void k_prob_keys(const ExtProb *probs, uint32_t n, uint32_t *keys, uint32_t *vals)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) { keys[p] = (probs[p].flags & XP_VALID) ? 0xFFFFu - probs[p].qLen : 0x10000u; vals[p] = p; }
}
