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
//                bits; the next 8 rows of a straight run are fetched together), stages the ops in rows the walk has already
//                consumed, then copies them to an exactly-sized slot of the ops arena.
//
// TRACE MEMORY GROWS WITH THE ROWS THAT ARE COMPUTED, not with the rows an extension may reach (an X-drop run stops after a median of 60
// rows of a bound of ~500): a wave hands its 64 lanes' blocks over together, once every ten rows ("flush"), into one 8 KB slot of the
// wave's current arena chunk -- lane l's block of the chunk's k-th flush at chunk + l * 2 KB + k * 128 B.  A chunk is YD_CHUNK_FLUSHES consecutive
// flushes of one wave (128 KB); waves take chunks from a common counter as they go and note them in a per-wave table.  Row i of a problem that started
// in flush f0 at row slot `phase` of lane l is row (i - 1 + phase) % 10 of the lane-l block of flush f0 + (i - 1 + phase) / 10, whatever
// the other lanes did meanwhile: (wave, lane, f0, phase) is all the traceback needs (carried in ExtRes).  Consecutive problems of a lane
// share a block (the earlier one's last rows, the later one's first); a lane starts its next problem no sooner than two row slots after
// the previous one's last row, because the traceback uses the row after a problem's best row as spare space.
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
#define YD_CHUNK_FLUSHES 16                    // flushes of one wave per arena chunk
#define YD_LANE_DWORDS (YD_CHUNK_FLUSHES * 32)  // inside a chunk a lane's blocks are consecutive (2 KB = 160 rows): a traceback reads neighbouring lines
#define YD_CHUNK_DWORDS (64 * YD_LANE_DWORDS)  // 128 KB: 64 lanes x 16 flushes x 128 B
#define YD_LDS_STRIDE 33                       // dwords per lane in the LDS staging block: odd, so that the row code (a lane per bank) and the hand-over
                                               // (eight lanes read one lane's block) are both free of bank conflicts

typedef uint32_t yd_u32x4 __attribute__((ext_vector_type(4)));
struct ExtProb { uint32_t qBase, rOff; uint16_t qOff, qLen; uint32_t flags; };           // 16 B; qBase = offset of the read in fwd/rev
enum { XP_STRAND = 1, XP_REV = 2, XP_VALID = 4 };
// 32 B.  maxj in register columns; rows/cells = work of this call.  where = phase | lane << 4 | wave << 10 (the problem's place in the arena);
// opsOff = the flush index of its first block as k_ext_rows leaves it.  After k_ext_trace (where, opsOff) = high and low word of the op list's place:
// a signed dword offset from the trace arena's base (the list stays where the walk staged it unless it straddles two chunks; then it is in the ops arena).
struct ExtRes { int score, maxi, maxj; uint32_t opsOff, nOps, where, rows, cells; };
static_assert(sizeof(ExtProb) == 16 && sizeof(ExtRes) == 32, "k_ext_rows_pk moves these as 16-byte vectors (non-temporal loads / stores)");
__device__ __forceinline__ const uint32_t *extOpsPtr(const uint32_t *traceBase, const ExtRes &r) {
    return traceBase + (long long)(((unsigned long long)r.where << 32) | (unsigned long long)r.opsOff); }

struct ExtArgs {
    DevParams P; const uint8_t *bases; const uint8_t *fwd, *rev;
    const uint8_t *fwd4, *rev4;                 // the same codes packed two to the byte (high nibble = even offset, as the reference's bases): k_ext_rows_pk's query stream
    const ExtProb *probs; uint32_t nProb;
    const uint32_t *order;                      // problem indices in processing order (longest bound first), or nullptr
    unsigned long long *clock;                  // optional: [0] = earliest start, [1] = latest end of the launch in wall_clock64() ticks (100 MHz)
    uint32_t *trace; uint32_t nChunks;          // the arena: nChunks chunks of YD_CHUNK_DWORDS
    unsigned int *chunkCount;                   // chunks handed out so far
    uint32_t *waveChunks; uint32_t maxCh;       // [wave][maxCh]: the chunk of the wave's flushes [16 k, 16 k + 16)
    uint32_t *ops; unsigned int *opsCount; uint32_t opsCap;     // k_ext_trace: the op lists, exactly sized, in list order (forward) / reversed (backward)
    ExtRes *res; unsigned int *queue; DevCounters *ctr;      // ctr == nullptr: the consumer of the results accounts for the work (careful extensions)
    int *errFlag; int dbgMode;                  // dbgMode: experiments only (YGPU_TRACE_MODE)
};
enum { YERR_TRACEMEM = 9 };                     // the trace arena (or a wave's chunk table) is full: the host grows it / cuts the batch and redoes the stage

// CAPS = false when neither run cap can bind inside a 21-column strip (maxGap >= 21 and maxIntron >= 21: a run spans at most 20
// columns): the run-length state (PI, PD) is then dead and is compiled out.
// SECOND = the careful-extension round of splitClump (split_lanes.h): same code, its own kernel name in profiles, work counted by
// the consumer of the results.
// three waves per SIMD: the register allocation is held at 168 (a fourth wave would cost more in spills than it hides, a third one is needed to cover the row's loads).
// With the run-length state (CAPS: 22 more registers, -G or the intron cap below 21 -- rare) two waves per SIMD: at three it spilled four registers to scratch memory.
template <bool CAPS, bool SECOND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CAPS ? 2 : 3, CAPS ? 2 : 3))) k_ext_rows(ExtArgs A)
{
    __shared__ uint32_t sBlk[256 * YD_LDS_STRIDE];   // per lane: the current 10-row trace block (30 dwords), lane stride 33
    if (!SECOND && A.clock && threadIdx.x == 0) atomicMin(&A.clock[0], (unsigned long long)wall_clock64());
    const int lane = laneId();
    const int GO = A.P.GO, GE = A.P.GE, GOE = A.P.GO + A.P.GE, XC = A.P.X, maxIntron = A.P.maxIntron, maxGap = A.P.maxGap;
    // (the strip keeps Vg = V - GOE, as k_ext_rows_pk does since round 5: one subtraction per cell instead of one per use -- the gap a neighbour opens to the right
    // and below-left -- and the diagonal step adds GOE back inside its two constants)
    const int MSG = A.P.MS + GOE, RCG = GOE - A.P.RC, LWG = YD_LWORST - GOE;
    constexpr int bandwidth = YD_LBAND, leftR = YD_LBAND;
    const uint32_t maxROff = A.P.maxROff;
    YD_GLOBAL const uint8_t *gBases = toGlobal(A.bases);
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t *const myBlk = &sBlk[threadIdx.x * YD_LDS_STRIDE];

    // The strip state lives in registers and is written by the row code only (a fresh problem selects its row-0 values at
    // the top of its first row): one definition per loop iteration keeps the register allocator from duplicating it.
    int PV[YD_LW], PF[YD_LW], PI[YD_LW];
    uint32_t w0 = 0, w1 = 0, w2 = 0;
    int p = -1, i = 0, qLen = 0, rLen = 0, maxScore = YD_LWORST, maxi = 0, maxj = 0, qStep = 0, qcNext = 0;
    uint32_t rOff = 0; bool rev = false, done = false;
    YD_GLOBAL const uint8_t *q = toGlobal(A.fwd);
    unsigned calls = 0, rows = 0, cells = 0;
#pragma unroll
    for (int j = 0; j < YD_LW; j++) { PV[j] = LWG; PF[j] = YD_LWORST; PI[j] = 0; }

    // Pool of claimed problems: lane l holds entry l, completely set up (clamped lengths, first reference window, first
    // query base), so that handing an entry to an idle lane is a few cross-lane moves and no memory latency.  One atomic
    // and one round of dependent loads per 64 problems.
    int poolCount = 0, poolNext = 0; bool exhausted = false;
    uint32_t eLens = 0, eROff = 0, eQ = 0, eMisc = 0, eW1 = 0, eW2 = 0, ePidx = 0;
    // deferred stores (see the row code)
    // (a finished problem's result is stored in the lane's NEXT pass; the refill leaves such a lane alone for that pass -- justDone -- so maxScore / maxi / maxj, the row
    // count i and pStart are still the problem's when the store is issued: no copies of them are kept -- six registers that the allocation of 168 did not have)
    bool pendFlush = false; int pendRes = -1; unsigned pStart = 0;   // pStart = the flush and the row slot a problem started in: flush << 4 | slot
    auto storeResult = [&]() {
        // work of this call: i rows; row r has 21 - max(11 - r, 0) real cells (the columns left of the origin come into the band one row at a time)
        const unsigned m = i < leftR ? (unsigned)i : (unsigned)leftR, nCells = (unsigned)YD_LW * (unsigned)i - ((unsigned)(leftR + 1) * m - m * (m + 1u) / 2u);
        rows += (unsigned)i; cells += nCells;
        ExtRes r; r.score = maxScore > 0 ? maxScore : 0; r.maxi = maxi; r.maxj = maxj; r.opsOff = pStart >> 4; r.nOps = 0;
        r.where = (pStart & 15u) | ((uint32_t)lane << 4) | (wave << 10); r.rows = (unsigned)i; r.cells = nCells;
        A.res[pendRes] = r; pendRes = -1; i = 0;
    };
    // Trace memory: the wave's 64 blocks of ten rows leave together (a "flush"), into slot (flush & 15) of the wave's current arena chunk.  wslot = the row
    // slot all lanes write in this iteration, flush = flushes done so far; a problem notes (flush, wslot) when it starts.  A lane whose problem ended inside
    // the block keeps the block `dirty` until the next hand-over; a lane that starts a problem inside a block shares it with its previous problem.
    int wslot = 0; unsigned flush = 0; bool dirty = false, justDone = false;
    YD_GLOBAL uint32_t *chunkPtr = toGlobal(A.trace); bool noMem = false;
    auto takeChunk = [&]() {                                                 // wave-uniform: the chunk of flushes [flush, flush + 16)
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(A.chunkCount, 1u);
        c = uniU(c);
        const unsigned ci = flush / YD_CHUNK_FLUSHES;
        if (c >= A.nChunks || ci >= A.maxCh) { noMem = true; c = 0; if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_TRACEMEM); }
        if (lane == 0 && ci < A.maxCh) A.waveChunks[(size_t)wave * A.maxCh + ci] = c;       // (a valid entry also when the arena is full: nothing may follow a stale one)
        chunkPtr = toGlobal(A.trace) + (size_t)c * YD_CHUNK_DWORDS;
    };
    takeChunk();
    // Hand-over: eight lanes write each block, 16 bytes apiece, so that one store instruction carries eight whole 128-byte lines.
    auto flushBlocks = [&]() {
        const unsigned long long f = __ballot(pendFlush);
        pendFlush = false;
        if (f != 0ull && !noMem) {
            YD_GLOBAL uint32_t *slot = chunkPtr + (size_t)(flush % YD_CHUNK_FLUSHES) * 32u;
            const int piece = lane & 7, wv = (int)(threadIdx.x >> 6);
#pragma unroll
            for (int g = 0; g < 8; g++) {
                if (((f >> (g * 8)) & 0xFFull) == 0ull) continue;            // wave-uniform
                const int o = g * 8 + (lane >> 3);
                const uint32_t *src = &sBlk[(wv * 64 + o) * YD_LDS_STRIDE + piece * 4];
                yd_u32x4 v; v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                if ((f >> o) & 1ull) YD_STORE_NT((YD_GLOBAL yd_u32x4 *)(slot + o * YD_LANE_DWORDS + piece * 4), v);      // (non-temporal: see common.h)
            }
        }
    };
    auto nextFlush = [&]() { flush++; if (flush % YD_CHUNK_FLUSHES == 0u) takeChunk(); };     // wave-uniform
    bool firstFill = true;
    for (;;) {
        if (noMem) break;                                                    // wave-uniform: the arena is full, the host redoes the stage with more memory or fewer roots
        // the blocks the previous iteration completed leave first: a problem that starts now notes the flush its own first block will go out with
        if (wslot == 0 && __ballot(pendFlush) != 0ull) { flushBlocks(); nextFlush(); }
        // ---- refill: until every lane is busy or nothing is left ----
        for (;;) {
            const unsigned long long need = __ballot(p < 0 && !done && !justDone);
            if (!need) break;
            // a refill pass costs ~100 wave instructions whatever the number of lanes it serves: wait until a few are idle
            if (__builtin_popcountll(need) < YD_REFILL_MIN && __ballot(p >= 0) != 0ull && !firstFill) break;
            if (poolNext >= poolCount) {                                     // wave-uniform: claim and set up the next 64 problems
                unsigned base = 0;
                if (!exhausted) { if (lane == 0) base = atomicAdd(A.queue, 64u); base = uniU(base); if (base >= A.nProb) exhausted = true; }
                if (exhausted) { if (p < 0) done = true; break; }
                poolCount = (int)min(64u, A.nProb - base); poolNext = 0;
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
                    if (ql <= 0) { ExtRes r; r.score = 0; r.maxi = r.maxj = 0; r.opsOff = r.nOps = 0; r.where = 0; r.rows = r.cells = 0; A.res[np] = r; }
                    else {
                        eLens = (uint32_t)ql | (rl << 16); eROff = pr.rOff; eQ = pr.qBase + pr.qOff;
                        YD_GLOBAL const uint8_t *qp = toGlobal((pr.flags & XP_STRAND) ? A.rev : A.fwd) + eQ;
                        eMisc = (pr.flags & 3u) | ((uint32_t)qp[0] << 8);
                        // reference window of row 1: register column c holds reference index c - leftR
                        eW1 = 0; eW2 = 0;
                        for (int c = leftR; c < YD_LW; c++) {
                            const int idx = c - leftR; uint32_t nib = 15u;
                            if (idx < (int)rl) { const uint32_t off = rv_ ? pr.rOff - (uint32_t)idx : pr.rOff + (uint32_t)idx; const uint32_t b = gBases[off >> 1];
                                nib = (off & 1u) ? (b & 15u) : (b >> 4); }
                            const uint32_t sh = (uint32_t)(c & 7) * 4u;
                            if (c < 16) eW1 |= nib << sh; else eW2 |= nib << sh;
                        }
                    }
                }
            }
            // the k-th idle lane takes entry poolNext + k
            const int nNeed = __builtin_popcountll(need), avail = poolCount - poolNext;
            const int e = poolNext + __builtin_popcountll(need & lanesBelow);
            const bool take = (p < 0 && !done && !justDone) && e < poolCount;
            const int src = take ? e : lane;
            const uint32_t gLens = (uint32_t)__shfl((int)eLens, src, 64), gROff = (uint32_t)__shfl((int)eROff, src, 64), gQ = (uint32_t)__shfl((int)eQ, src, 64),
                gMisc = (uint32_t)__shfl((int)eMisc, src, 64);
            const uint32_t gW1 = (uint32_t)__shfl((int)eW1, src, 64), gW2 = (uint32_t)__shfl((int)eW2, src, 64);
            const uint32_t gPidx = (uint32_t)__shfl((int)ePidx, src, 64);
            const bool init = take && gLens != 0u;
            // row 0 of the strip (SW.cpp:905-935; PF(0, left) = -GO, see the header) for the lanes that start a problem: plain selects,
            // so that the state registers have one definition here and one in the row code
#pragma unroll
            for (int j = 0; j < YD_LW; j++) {
                const int iV = (j == leftR ? 0 : (j > leftR ? -(GO + (j - leftR) * GE) : YD_LWORST)) - GOE, iF = j == leftR ? -GO : YD_LWORST;      // (Vg)
                PV[j] = init ? iV : PV[j]; PF[j] = init ? iF : PF[j]; if (CAPS) PI[j] = init ? 0 : PI[j];
            }
            if (init) {
                p = (int)gPidx; qLen = (int)(gLens & 0xFFFFu); rLen = (int)(gLens >> 16); i = 0; maxScore = YD_LWORST; maxi = 0; maxj = 0;
                rev = (gMisc & XP_REV) != 0; rOff = gROff;
                q = toGlobal((gMisc & XP_STRAND) ? A.rev : A.fwd) + gQ; qStep = rev ? -1 : 1; qcNext = (int)((gMisc >> 8) & 0xFFu);
                pStart = (flush << 4) | (unsigned)wslot;
                w0 = 0; w1 = gW1; w2 = gW2;
            }
            poolNext += nNeed < avail ? nNeed : avail;
        }
        firstFill = false;
        if (__ballot(p >= 0) == 0ull && __ballot(justDone) == 0ull) break;   // nothing runs, nothing left (a lane waiting out its gap row keeps the wave alive)

        // ---- one DP row in every lane (lanes without a problem run on idle state; their stores are masked) ----
        // All memory operations of an iteration are issued at its top: the previous ten rows' trace cells (above) and a finished
        // problem's result (both deferred), and the loads the row needs at its END (next query base, next reference base).
        // The wait the compiler puts at the loop header then finds them ~1000 instructions old.
        if (pendRes >= 0) storeResult();
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
        int PVCol = LWG, PE = YD_LWORST, PD = 0;
        uint32_t t0 = 0, t1 = 0, t2 = 0, rowKey = 0;
        int dV = PV[0];
#pragma unroll
        for (int j = 0; j < YD_LW; j++) {
            const uint32_t wsrc = j < 8 ? w0 : (j < 16 ? w1 : w2);
            const int rc = (int)((wsrc >> ((j & 7) * 4)) & 15u);
            const bool eq = rc == qc;
            int V = dV + (eq ? MSG : RCG);
            const int CE = PE - GE, NE = PVCol;
            const bool cE = CE >= NE && (!CAPS || PD < maxIntron);
            PE = cE ? CE : NE; if (CAPS) PD = cE ? PD + 1 : 1;
            const bool tE = PE >= V; V = tE ? PE : V;
            int upV, upF, upI;
            if (j + 1 < YD_LW) { upV = PV[j + 1]; upF = PF[j + 1]; upI = PI[j + 1]; } else { upV = LWG; upF = YD_LWORST; upI = 0; }
            const int CF = upF - GE, NF = upV;
            const bool cF = CF >= NF && (!CAPS || upI < maxGap);
            const int F = cF ? CF : NF, I = CAPS ? (cF ? upI + 1 : 1) : 0;
            const bool tF = F >= V; V = tF ? F : V;
            uint32_t nib = eq ? (uint32_t)OP_M : (uint32_t)OP_R; nib = tE ? (uint32_t)OP_D : nib; nib = tF ? (uint32_t)OP_I : nib;
            nib |= (cE ? 4u : 0u) | (cF ? 8u : 0u);
            // pinned: the condition masks die here
            if (j < 8) { t0 |= nib << ((j & 7) * 4); asm volatile("" : "+v"(t0)); } else if (j < 16) { t1 |= nib << ((j & 7) * 4); asm volatile("" : "+v"(t1));
                } else { t2 |= nib << ((j & 7) * 4); asm volatile("" : "+v"(t2)); }
            // row-major first maximum over the real cells: key = (V + BIAS) << 5 | (31 - j)
            uint32_t key = ((uint32_t)(V + YD_BIAS)) << 5 | (uint32_t)(31 - j);
            if (j < leftR) key = j >= sc ? key : 0u;
            rowKey = key > rowKey ? key : rowKey;
            { const int Vg = V - GOE; PV[j] = Vg; PVCol = Vg; } PF[j] = F; PI[j] = I;
            dV = upV;                                                        // the next column's diagonal predecessor
            __builtin_amdgcn_sched_barrier(0);                               // keep the cells in program order: their many condition masks stay short-lived
        }
        { const int slot = wslot * 3;                                        // this row's cells go to the lane's LDS block (the same row slot in every lane)
          myBlk[slot] = t0; myBlk[slot + 1] = t1; myBlk[slot + 2] = t2; }
        int rv = YD_LWORST, rj = 0;
        if (rowKey) { rv = (int)(rowKey >> 5) - YD_BIAS; rj = 31 - (int)(rowKey & 31u); }
        if (rv > maxScore) { maxScore = rv; maxi = i; maxj = rj; }
        // slide the window: column c takes column c+1, the top column takes the new base
        const uint32_t nb = nbOdd == 2u ? 15u : (nbOdd ? (nbByte & 15u) : (nbByte >> 4));
        w0 = (w0 >> 4) | (w1 << 28); w1 = (w1 >> 4) | (w2 << 28); w2 = (w2 >> 4) | (nb << 16);
        const bool fin = busy && (rv < maxScore - XC || i >= qLen);
        if (busy) dirty = true;
        if (wslot == 9) { pendFlush = dirty; dirty = false; wslot = 0; } else wslot++;      // wave-uniform
        justDone = fin;                                                      // the next row slot stays empty behind a finished problem (its traceback's spare row)
        if (fin) { pendRes = p; p = -1; qStep = 0; rLen = 0; qLen = 0; }      // (i, maxScore, maxi, maxj, pStart stay as they are until storeResult)
    }
    // the last deferred stores; one more flush slot stays reserved behind the last block (the spare row of a problem that ended in row slot 9)
    if (wslot != 0 && dirty) pendFlush = true;
    if (!noMem && __ballot(pendFlush) != 0ull) { flushBlocks(); nextFlush(); }
    if (pendRes >= 0) storeResult();
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

struct ExtRowBits { uint32_t a, b, c; };
__device__ __forceinline__ uint32_t extNib(const ExtRowBits &r, int x) { const uint32_t w = x < 8 ? r.a : (x < 16 ? r.b : r.c); return (w >> ((x & 7) * 4)) & 15u; }

// ---- traceback, lane per problem (SW.cpp:1138-1195) -----------------------------------------------------------------
// Logical strip of a problem: blocks of 32 dwords (10 rows of 3 dwords + 2 spare), block b = the lane's block of the wave's flush f0 + b.
// Logical dword d lives at chunk(f0 + d / 32) + lane * 512 + ((f0 + d / 32) % 16) * 32 + d % 32.
__device__ unsigned int gTraceDbg[8];             // diagnostics of the first out-of-strip access (YGPU_TRACE)
struct ExtStrip {
    YD_GLOBAL uint32_t *arena; YD_GLOBAL const uint32_t *tab; unsigned f0, laneOff; int cIdx; YD_GLOBAL uint32_t *cBase; int limit; bool wild;
    __device__ __forceinline__ YD_GLOBAL uint32_t *at(int d)
    {
        if ((unsigned)d >= (unsigned)limit) { wild = true; d = 0; }         // outside the problem's strip: a corrupted trace; never follow it into the table
        const unsigned f = f0 + (unsigned)(d >> 5); const int ci = (int)(f / YD_CHUNK_FLUSHES);
        if (ci != cIdx) { cIdx = ci; cBase = arena + (size_t)tab[ci] * YD_CHUNK_DWORDS + laneOff; }
        return cBase + (f % YD_CHUNK_FLUSHES) * 32u + (unsigned)(d & 31);
    }
};

// Walks from (y, x) back to the origin (0, leftR).  The ops are staged INSIDE THE STRIP, over rows the walk has already
// consumed: emission k (far end first) goes to logical dword E-1-k, E = end of row maxi+1 (the strip has one spare row).  After
// k rows are consumed at most 2k+1 ops exist (every op but a D needs a row of its own, and two D ops never touch) and
// 3k+3 dwords are free, so the walk never overwrites a row it still has to read.  The finished list -- ascending dwords: the forward
// extension's list in order (ops are added to the front, SW.cpp:1186), the backward extension's list reversed (added to the back,
// SW.cpp:1190) -- is then copied to a slot of the ops arena that is exactly its size (one reservation per wave).
__device__ __forceinline__ int extRowWord(int y) { return ((y - 1) / 10) * 32 + ((y - 1) % 10) * 3; }
#define YD_TRACE_DEPTH 8
__global__ void __launch_bounds__(256) k_ext_trace(ExtArgs A)
{
    YD_HIGH_PRIO();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; const int lane = laneId();
    // k_ext_rows ran out of arena (or an earlier kernel failed): the stage is redone, its strips are incomplete
    if (*toGlobal(A.errFlag) != 0) return;
    // problems in the order k_ext_rows took them: the 64 problems of a pool ran in one wave at the same time, so the lanes of a wave here walk neighbouring blocks
    const bool live = t < A.nProb;
    const uint32_t p = live ? ((A.order && !(A.dbgMode & 2)) ? A.order[t] : t) : 0u;
    ExtRes r; r.score = 0; r.nOps = 0;
    if (live) r = A.res[p];
    const bool walk = live && r.score > 0;
    constexpr int leftR = YD_LBAND;
    int n = 0; bool bad = false; int E = 0;
    ExtStrip S; S.arena = toGlobal(A.trace); S.cIdx = -1; S.cBase = S.arena; S.f0 = 0; S.laneOff = 0; S.tab = toGlobal(A.waveChunks); S.limit = 0; S.wild = false;
    if (walk) {
        const int ph = (int)(r.where & 15u);                                 // row i of the problem is strip row i - 1 + ph (k_ext_rows keeps the blocks of a wave in phase)
        S.f0 = r.opsOff; S.laneOff = ((r.where >> 4) & 63u) * YD_LANE_DWORDS; S.tab = toGlobal(A.waveChunks) + (size_t)(r.where >> 10) * A.maxCh;
        E = extRowWord(r.maxi + 1 + ph) + 3; S.limit = E;
        int y = r.maxi, x = r.maxj, prev = -1, acc = 0;
        // The walk keeps a PHYSICAL cursor on row y: cp = its first dword, rr = its row inside the 10-row block, w = its logical word offset (for the
        // staging bound), fb = its block's flush index.  One row up is cp - 3; across a block boundary cp - 5 (the lane's previous block is the line below),
        // across a chunk boundary one table look-up.
        struct Cur { YD_GLOBAL uint32_t *cp; int rr, w; unsigned fb; };
        Cur c; c.rr = (y - 1 + ph) % 10; c.w = ((y - 1 + ph) / 10) * 32 + c.rr * 3; c.fb = S.f0 + (unsigned)((y - 1 + ph) / 10); c.cp = S.at(c.w);
        auto stepUp = [&](Cur &u) {                                          // one row towards the origin
            if (u.rr != 0) { u.rr--; u.w -= 3; u.cp -= 3; return; }
            u.rr = 9; u.w -= 5; u.fb--;
            if (u.fb % YD_CHUNK_FLUSHES == YD_CHUNK_FLUSHES - 1u) u.cp = S.arena + (size_t)S.tab[u.fb / YD_CHUNK_FLUSHES] * YD_CHUNK_DWORDS + S.laneOff + (YD_CHUNK_FLUSHES - 1u) *
                32u + 27u;
            else u.cp -= 5;
        };
        const int row0w = extRowWord(1 + ph) - 3;                               // the logical position of row 0 (not in the strip): with the cursor there every row is consumed
        auto flush = [&]() { const int wp = E - 1 - n; if (wp < c.w + 3) bad = true; else *S.at(wp) = opMake(prev, acc); n++; };   // rows above row y are consumed
        auto put = [&](int code, int len) { if (prev != code) { if (prev >= 0) flush(); prev = code; acc = len; } else acc += len; };
        // The kernel is bound by the latency of dependent loads (one per path cell).  Most of a path is straight runs of M / R cells
        // in one column, so the cells of the next YD_TRACE_DEPTH rows in column x are fetched together (one dword each) and consumed in turn.
        for (int guard = 0; guard < 70000 && y > 0 && x >= 0 && x < YD_LW; guard++) {
            const int ws = x >> 3, sh = (x & 7) * 4;
            uint32_t d[YD_TRACE_DEPTH];
            // rows fetched together: as many as the walk has left, and none beyond the chunk (the cursor steps without branches or table look-ups inside it,
            // so that the loads are issued back to back)
            int lim = (int)(c.fb % YD_CHUNK_FLUSHES) * 10 + c.rr + 1; lim = lim < y ? lim : y; lim = lim < YD_TRACE_DEPTH ? lim : YD_TRACE_DEPTH;
            { YD_GLOBAL uint32_t *up = c.cp; int ur = c.rr;
#pragma unroll
              for (int k = 0; k < YD_TRACE_DEPTH; k++) { d[k] = up[ws]; const bool go = k + 1 < lim; const int dec = ur == 0 ? 5 : 3; up -= go ? dec : 0;
                  ur = go ? (ur == 0 ? 9 : ur - 1) : ur; } }
            uint32_t nib = (d[0] >> sh) & 15u; int took = 0;
#pragma unroll
            for (int k = 0; k < YD_TRACE_DEPTH; k++) {
                if (k >= lim) break;
                nib = (d[k] >> sh) & 15u;
                const int op = (int)(nib & 3u);
                if (op >= OP_D || y <= 0) break;
                if (prev != op) { if (prev >= 0) flush(); prev = op; acc = 1; } else acc++;
                y--; took++;
                // inside the chunk: no look-up
                if (took < lim) { const int dec = c.rr == 0 ? 5 : 3; c.cp -= dec; c.w -= dec; c.fb -= c.rr == 0 ? 1u : 0u; c.rr = c.rr == 0 ? 9 : c.rr - 1; }
            }
            // the whole batch was a straight run: one full step (it may leave the chunk), next batch
            if (took == lim) { if (y > 0) stepUp(c); else c.w = row0w; continue; }
            if ((nib & 3u) == (uint32_t)OP_D) {                                  // deletion run: walk the continue bits along the row
                ExtRowBits rb; rb.a = c.cp[0]; rb.b = c.cp[1]; rb.c = c.cp[2];
                int run = 1, xx = x;
                while (extNib(rb, xx) & 4u) { xx--; if (xx < 0) break; run++; }
                put(OP_D, run); x -= run;
            } else {                                                            // insertion run: walk the continue bits up and to the right
                int run = 1, yy = y, xx = x; Cur u = c; uint32_t nb2 = nib;
                while (nb2 & 8u) {
                    yy--; xx++; if (yy <= 0 || xx >= YD_LW) break;
                    run++; stepUp(u);
                    nb2 = (u.cp[xx >> 3] >> ((xx & 7) * 4)) & 15u;
                }
                put(OP_I, run); y -= run; x += run;
                if (y > 0) { for (int k = 0; k < run; k++) stepUp(c); } else c.w = row0w;
            }
        }
        if (y <= 0 && x > leftR) put(OP_D, x - leftR);                           // row 0: deletions back to the origin (SW.cpp:905-935)
        if (prev >= 0) { const int wp = E - 1 - n; if (wp < 0) bad = true; else *S.at(wp) = opMake(prev, acc); n++; }
        if (bad || S.wild) { bad = true; atomicCAS(A.errFlag, 0, (int)YERR_TRACE); n = 0; if (S.wild && atomicCAS(&gTraceDbg[5], 0u, 1u) == 0u) { gTraceDbg[6] = p;
            gTraceDbg[7] = r.where; } }
    }
    // The finished list is contiguous where it was staged unless it straddles two chunks; only then it is copied, to an exactly-sized slot of the ops arena
    // (one reservation per wave).  Either way the result carries its address as an offset from the trace arena's base.
    if (A.dbgMode & 1) n = 0;
    const bool inPlace = walk && !bad && n > 0 && (S.f0 + (unsigned)((E - n) >> 5)) / YD_CHUNK_FLUSHES == (S.f0 + (unsigned)((E - 1) >> 5)) / YD_CHUNK_FLUSHES;
    const int nCopy = (walk && !bad && !inPlace) ? n : 0;
    int incl = nCopy;
#pragma unroll
    for (int d2 = 1; d2 < 64; d2 <<= 1) { const int v = __shfl_up(incl, d2, 64); if (lane >= d2) incl += v; }
    const int total = __shfl(incl, 63, 64); unsigned ob = 0;
    if (lane == 63 && total) ob = atomicAdd(A.opsCount, (unsigned)total);
    ob = (unsigned)__shfl((int)ob, 63, 64);
    if (walk && !bad) {
        long long place;
        if (inPlace) place = (long long)(S.at(E - n) - S.arena);
        else {
            const unsigned off = ob + (unsigned)(incl - nCopy);
            if ((unsigned long long)off + (unsigned)n > (unsigned long long)A.opsCap) { atomicCAS(A.errFlag, 0, (int)YERR_OUT); return; }
            YD_GLOBAL uint32_t *dst = toGlobal(A.ops) + off;
            for (int k0 = 0; k0 < n; k0 += 8) {                              // eight loads in flight, then eight stores (the staged list is a few lines in L2)
                uint32_t v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = (k0 + j < n) ? *S.at(E - n + k0 + j) : 0u;
#pragma unroll
                for (int j = 0; j < 8; j++) if (k0 + j < n) dst[k0 + j] = v[j];
            }
            place = (long long)(dst - S.arena);
        }
        r.opsOff = (uint32_t)(unsigned long long)place; r.where = (uint32_t)((unsigned long long)place >> 32); r.nOps = (uint32_t)n; A.res[p] = r;
    }
}

// Traceback order: a wave of k_ext_trace takes as long as its longest walk, and the walks (best row maxi; none at all without a score) differ by two orders
// of magnitude among the problems of a pool.  Re-sorting by length alone scatters a wave's lanes over the arena (measured: 2-3 times slower -- the 64 lanes then
// read 64 different chunks); so the problems are grouped by the ARENA CHUNK their strip starts in (the wave of k_ext_rows that ran them and the chunk it was
// writing) and sorted by walk length, descending, inside the group: lanes of a wave read one chunk and walk paths of similar length.
// key = arena region (physical chunk >> regionLog) | length bucket : 5 (32 buckets of the longest read's length: two radix passes for the bench batch)
__global__ void k_trace_keys(const ExtRes *res, const uint32_t *order, uint32_t n, const uint32_t *waveChunks, uint32_t maxCh, int regionLog, int lenShift, int lenBits,
    uint32_t *keys, uint32_t *vals)
{
    YD_HIGH_PRIO();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint32_t p = order ? order[t] : t;
    const ExtRes r = res[p];
    const uint32_t top = (1u << lenBits) - 1u;
    const uint32_t len = r.score > 0 ? min((uint32_t)r.maxi >> lenShift, top) : 0u;
    const uint32_t ci = r.opsOff / YD_CHUNK_FLUSHES;
    const uint32_t phys = (regionLog < 20 && r.score > 0 && ci < maxCh) ? waveChunks[(size_t)(r.where >> 10) * maxCh + ci] : 0u;     // (regionLog >= 20: one region)
    keys[t] = (min(phys >> regionLog, 0xFFFFFFu) << lenBits) | (top - len);
    vals[t] = p;
}

// Trace volume of a batch: 16-byte (k_ext_rows_pk) / 12-byte (k_ext_rows) records the rows kernel WROTE -- one per iteration of every problem it ran: rows + 1 --
// against the records a traceback can VISIT: those of rows 0 .. maxi of the problems that end with a score above zero (SW.cpp:1091-1111 returns before any
// traceback otherwise; rows behind the maximum belong to the X-drop tail).  out[0] = written, out[1] = visitable, out[2] = problems run, out[3] = walkers.
__global__ void __launch_bounds__(256) k_trace_volume(const ExtRes *res, uint32_t n, unsigned long long *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long w = 0, v = 0, np = 0, nw = 0;
    if (t < n) { const ExtRes r = res[t]; if (r.rows) { w = r.rows + 1ull; np = 1; if (r.score > 0) { v = (unsigned long long)r.maxi + 1ull; nw = 1; } } }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        w += (unsigned long long)__shfl_xor((long long)w, d, 64); v += (unsigned long long)__shfl_xor((long long)v, d, 64);
        np += (unsigned long long)__shfl_xor((long long)np, d, 64); nw += (unsigned long long)__shfl_xor((long long)nw, d, 64);
    }
    if ((threadIdx.x & 63u) == 0u && np) { atomicAdd(&out[0], w); atomicAdd(&out[1], v); atomicAdd(&out[2], np); atomicAdd(&out[3], nw); }
}

// order values are global problem indices; a chunk's kernels index from the chunk's first problem
__global__ void k_rebase_u32(uint32_t *v, uint32_t n, uint32_t sub)
{
    YD_HIGH_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] -= sub;
}

// sort keys for a problem list that has none yet: longest row bound first
__global__ void k_prob_keys(const ExtProb *probs, uint32_t n, uint32_t *keys, uint32_t *vals)
{
    YD_HIGH_PRIO();
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) { keys[p] = (probs[p].flags & XP_VALID) ? 0xFFFFu - probs[p].qLen : 0xFFFFu; vals[p] = p; }
}
