// ext_lanes_pk.h -- k_ext_rows in PACKED 16-BIT arithmetic, one problem per lane, the strip SKEWED over the two halves of each register.
//
// k_ext_rows (ext_lanes.h) is bound by vector-instruction issue: ~30 instructions per cell at ~4 cycles each.  v_pk_*_i16 instructions run at the same
// rate and serve two cells.  The serial E chain of a row (column c needs column c-1) rules out packing two columns of the SAME row, so the 22 register
// columns are split in two halves that run one row apart:
//
//     register pair k (k = 0..10):   low half  = column k      of row t       high half = column k + 11 of row t - 1        (iteration t of the problem)
//
// Inside an iteration the pairs are computed left to right, as the columns are in k_ext_rows.  For both halves the diagonal neighbour is the pair's own old
// value and the upper neighbour the next pair's old value, exactly as in the 32-bit kernel; the two places where the halves meet:
//   * the left neighbour of column 11 (pair 0, high) is column 10 of the same row t-1 = pair 10's low half as the PREVIOUS iteration left it (V, and the E
//     value carried in a register of its own);
//   * the upper neighbour of column 10 (pair 10, low) is column 11 of row t-1 = pair 0's high half as THIS iteration has just computed it.
// Column 21 (pair 10, high) does not exist: it is held at the sentinel and stands for "no cell above to the right" of column 20.
// Row 0 of the high half is not special either: started from the sentinel, the E chain that enters from the origin (0, 10) produces exactly
// V(0, j) = -(GO + (j - 10) GE) in the problem's first iteration.  A problem of i rows takes i + 1 iterations; row t's maximum is known after iteration t + 1.
//
// Decisions are the sign bits of saturating differences (no compares, no SGPR pairs): bit 15 / 31 of  dT = E - G (set: E does NOT win), dU = F - max(G, E),
// dE = (PE - GE) - (PV - GOE) (set: the E run does NOT continue), dF likewise.  v_perm_b32 gathers the sign BYTES of two registers (both halves), v_bfi shifts
// them into bit planes: one trace record (16 bytes) per iteration and lane,
//     dword 0 (AB2): pairs 8..10: bits 5..7 of the A bytes, bits 2..4 of the B bytes
//     dword 1 (M):   mismatch bits, pair k at bit 10 - k (low half) and 26 - k (high half)
//     dword 2 (A):   byte 0 = notT low half, byte 1 = notT high half, byte 2 = notU low, byte 3 = notU high;  bit k of a byte = pair k (k = 0..7)
//     dword 3 (B):   as A for notContE / notContF
// (the traceback of a straight run needs notT, notU and the mismatch bit of one pair per record: dwords 0,1 or 1,2 -- one 8-byte load)
// eight records to the 128-byte block; everything else -- the pool of pre-loaded problems, the wave-wide hand-over of blocks into arena chunks, ExtRes --
// is k_ext_rows'.  Record t holds row t's columns 0..10 and row t-1's columns 11..20; the traceback (k_ext_trace_pk) reads cell (y, x) in record y + (x >= 11).
//
// Used when the scores fit 16 bits with room for the sentinel (-16000; all arithmetic saturates, so a difference with a sentinel operand keeps its sign):
// MS * (longest read) <= 15000, RC + X + GO + 21 * GE <= 4000, and neither run cap can bind (maxGap, maxIntron >= 21); otherwise k_ext_rows.
#pragma once
#include "ext_lanes.h"

typedef short yd_s16x2 __attribute__((ext_vector_type(2)));
#define YD_LW16 (-16000)
#define YD_NP 11                               // register pairs
#define YD_RCREAL 0x7FF0u                      // reference code of a matrix cell: 0x7FF0 | nibble (as a score bound: never below a real V); not a cell: 0x800F

__device__ __forceinline__ yd_s16x2 pkS(uint32_t v) { return __builtin_bit_cast(yd_s16x2, v); }
__device__ __forceinline__ uint32_t pkU(yd_s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pkAdd(uint32_t a, uint32_t b) { return pkU(__builtin_elementwise_add_sat(pkS(a), pkS(b))); }
__device__ __forceinline__ uint32_t pkSub(uint32_t a, uint32_t b) { return pkU(__builtin_elementwise_sub_sat(pkS(a), pkS(b))); }
__device__ __forceinline__ uint32_t pkMax(uint32_t a, uint32_t b) { return pkU(__builtin_elementwise_max(pkS(a), pkS(b))); }
// (inline assembly with register operands: written as vector expressions these two are scalarised into compares and selects)
__device__ __forceinline__ uint32_t pkMin(uint32_t a, uint32_t b) { return pkU(__builtin_elementwise_min(pkS(a), pkS(b))); }
__device__ __forceinline__ uint32_t pkMinU(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pkMad(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// 0xFFFF in every negative half.  (Not inline assembly with the literal 15: in a packed instruction an inline constant feeds the low half only.)
__device__ __forceinline__ uint32_t pkSignMask(uint32_t a) { return pkU(pkS(a) >> (yd_s16x2)(15)); }
// the same as one opaque instruction (c15 = 0x000F000F in a register): written as an expression the compiler turns sub + shift + select into two
// 16-bit compares and two selects, which cost twice as much here (v_cmp writes an SGPR pair)
__device__ __forceinline__ uint32_t pkSignMaskAsm(uint32_t a, uint32_t c15) { uint32_t d; asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(d) : "v"(c15), "v"(a)); return d; }
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }                      // v_bfi_b32
__device__ __forceinline__ uint32_t pk2(int v) { return ((uint32_t)v & 0xFFFFu) * 0x10001u; }

// YD_ROWS_LDSWIN (round 5): the two nibble streams of a lane -- query codes, reference bases -- come from HBM in ALIGNED 16-BYTE PIECES (one request per lane and 32
// rows, a 64-byte sector is visited four times instead of sixteen) that are staged in LDS ([dword][thread]: conflict-free for the one-dword reads); every 8th row a lane
// takes its next dword of eight entries from there.  0: every refill is a dword load from HBM (rounds 2-4).
// Built, bit-exact, and SLOWER (profiles/r05_rows_lds_windows.txt: k_ext_rows_pk 14.68 -> 15.62 ms a launch, 44.5 -> 46.1 ms a step with four contexts): a lane's
// stream is private to it, so the staging buys no reuse, only larger requests -- and the kernel is bound by instruction issue, not by its requests.  Off.
#ifndef YD_ROWS_LDSWIN
#define YD_ROWS_LDSWIN 0
#endif
// YD_ROWS_RESBATCH (round 6): a finished problem's result -- the column of its maximum looked up in the kept strip (55 instructions), 32 bytes stored -- used to
// be written in the pass after the problem ended: with 64 lanes and ~110 rows a problem some lane has one pending in 43 % of the passes, and the whole wave walks
// through the block every time.  The lane is idle until the next refill anyway (its state stays its own), so the results now wait for each other: they are written
// when YD_REFILL_MIN lanes hold one (or nothing runs any more), and a lane is not refilled before its result is out.  0: the pass after the end (rounds 2-5).
#ifndef YD_ROWS_RESBATCH
#define YD_ROWS_RESBATCH 1
#endif
#ifdef YD_PROF
// k_ext_trace_pk (YD_PROF): waves that walk, passes, active lane-passes, group-loop rounds, rounds with a flush, passes with a deletion run, with an insertion run, rows
__device__ unsigned long long gTraceProf[8];
__device__ unsigned long long gRowsProf[8];      // passes, passes that wrote results, refill rounds, passes with a new maximum, busy lane-passes, flushes, pool loads
#endif
// YD_ROWS_UNI_EXIT=1: only the pass loop's two exits as scalar branches -- 14.38 -> 14.32 ms a launch (everything at once was slower, see the loop's head); and a
// scalar exit is the safe form of a wave-uniform one
#ifndef YD_ROWS_UNI_EXIT
#define YD_ROWS_UNI_EXIT 1
#endif
#if YD_ROWS_UNI_EXIT
#define YD_ROWS_EXIT(c) UNI_B(c)
#else
#define YD_ROWS_EXIT(c) (c)
#endif
#ifndef YD_ROWS_WAVES
#define YD_ROWS_WAVES 3                        // waves per SIMD of k_ext_rows_pk (a build switch for experiments: make variant VARIANT_FLAGS=-DYD_ROWS_WAVES=2)
#endif
// BS = threads of a workgroup: 256 (a wave per SIMD), or 512 (two waves per SIMD of ONE CU: what the launch uses when it shares the device and takes a part of the
// CUs only -- a wave that is alone on its SIMD runs at a third of the rate, so the waves of a partial launch should come in pairs; see stage_align.hip)
template <bool SECOND, int BS = 256>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(YD_ROWS_WAVES, YD_ROWS_WAVES))) k_ext_rows_pk(ExtArgs A)
{
    __shared__ uint32_t sBlk[BS * YD_LDS_STRIDE];    // per lane: the current block of eight 16-byte records, lane stride 33
#if YD_ROWS_LDSWIN
    __shared__ uint32_t sWin[8][BS];                 // per thread: the current 16-byte piece of its query stream (dwords 0..3) and of its reference stream (4..7)
    yd_u32x4 GQ = {0u, 0u, 0u, 0u}, GR = {0u, 0u, 0u, 0u}; int jQ = 0, jR = 0; bool newQ = false, newR = false, freshQ = false, freshR = false;
    const unsigned tid = threadIdx.x;
#endif
    if (!SECOND && A.clock && threadIdx.x == 0) atomicMin(&A.clock[0], (unsigned long long)wall_clock64());
    const int lane = laneId();
    const int GO = A.P.GO, GE = A.P.GE, XC = A.P.X;
    constexpr int bandwidth = YD_LBAND, leftR = YD_LBAND;
    const uint32_t maxROff = A.P.maxROff;
    YD_GLOBAL const uint8_t *gBases = toGlobal(A.bases);
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    const unsigned wave = blockIdx.x * (unsigned)(BS / 64) + (threadIdx.x >> 6);
    uint32_t *const myBlk = &sBlk[threadIdx.x * YD_LDS_STRIDE];
    const uint32_t GEp = pk2(GE), GOEp = pk2(GO + GE), NEGK = pk2(-(A.P.MS + A.P.RC)), LWp = pk2(YD_LW16), ONEp = 0x00010001u, LWlo = (uint32_t)YD_LW16 & 0xFFFFu;
    // The strip keeps every cell's value as Vg = V - (GO + GE) (round 5): that is what both of a cell's later uses subtract -- the gap a neighbour opens from it, to its
    // right in the same row (NE) and below-left in the next (NF) -- so one subtraction per cell replaces two; the diagonal step adds GO + GE back inside the constant of
    // its multiply-add (MS + GO + GE | -RC + GO + GE), for nothing.  Eleven instructions fewer a row; all values as before (nothing here saturates: see the header).
    const uint32_t MSGp = pk2(A.P.MS + GO + GE), LWg = pk2(YD_LW16 - (GO + GE)), LWgLo = (uint32_t)(YD_LW16 - (GO + GE)) & 0xFFFFu;

    // operands of the inline-assembly instructions: kept in VGPRs (copied from SGPRs at every use otherwise)
    uint32_t NEGKv = NEGK, MSv = MSGp, ONEv = ONEp, C15v = 0x000F000Fu;
    asm volatile("" : "+v"(NEGKv), "+v"(MSv), "+v"(ONEv), "+v"(C15v));
    uint32_t PV[YD_NP], PF[YD_NP], rc[YD_NP], carryE = LWp, nbNext = 15u;
    int p = -1, i = 0, qLen = 0, maxScore = YD_LWORST, maxi = 0, qcNext = 0, qcPrev = 0, rvLo = YD_LWORST;
    // the strip of the iteration that set the maximum (the column is found when the problem ends): only the half the maximum is in, two pairs to the register
    // (round 6: six v_perm instead of eleven moves in 99.9 % of the passes, five registers fewer)
    constexpr int YD_NSV = (YD_NP + 1) / 2;
    uint32_t SV[YD_NSV]; int maxSide = 0;
#pragma unroll
    for (int k = 0; k < YD_NSV; k++) SV[k] = 0;
    // The two input streams -- a query code and a reference nibble per iteration -- come through per-lane WINDOWS: 64-bit shift registers of the next sixteen
    // entries in the order the lane consumes them (low end first), refilled with whole aligned dwords of eight entries every 8th iteration (the iterations are
    // the wave's: wslot == 0).  Both are NIBBLE streams: the reference's packed bases, and the batch's query codes packed the same way (k_pack4; round 4 -- a
    // byte per code was a dword refill every 4th iteration and a 64-byte sector every 64 rows; packed, a lane comes back to a sector for 128 rows, as for the
    // reference).  A byte load per lane and iteration (rounds 1-2) made every lane's stream a line of its own in L1 and L2, 196-262 k lines that evict each other:
    // 61 GB of fetches per launch for 2 GB of input.  qHave / rHave = buffered entries, qNext / rNext = index (inside the extension) of the first one not
    // buffered; after the pool's first fill (a partial dword + a full one) the buffered end is dword-aligned, so every refill is one full dword.  Nothing outside
    // the extension's own range is ever addressed (a dword is loaded only if it holds an entry of the range).
    uint32_t qwLo = 0, qwHi = 0, rwLo = 0, rwHi = 0, qLd = 0, rLd = 0, rOffP = 0, qPos = 0;
    int qHave = 0, rHave = 0, qNext = 0, rNext = 0, rLenP = 0, qStep = 0, rLeft = 0; bool pendQ = false, pendR = false, done = false;
    bool insQ = false, insR = false;                                         // wave-uniform: the previous pass ran the query / reference refill
    // the strand's PACKED codes (two to the byte, as the reference: k_pack4): entry idx of the extension = nibble qPos +- idx
    YD_GLOBAL const uint8_t *q4 = toGlobal(A.fwd4);
    unsigned calls = 0, rows = 0, cells = 0;
#ifdef YD_PROF
    unsigned pfPass = 0, pfRes = 0, pfRefill = 0, pfSnap = 0, pfBusy = 0, pfFlush = 0, pfPool = 0;      // (wave-uniform)
#endif
#pragma unroll
    for (int k = 0; k < YD_NP; k++) { PV[k] = LWg; PF[k] = LWp; rc[k] = 0x7FFF7FFFu; }      // (PV holds Vg = V - GOE throughout)

    int poolCount = 0, poolNext = 0; bool exhausted = false;
    // eMisc: flags | first query code << 8 | buffered entries (query | reference << 8) << 16
    uint32_t eLens = 0, eROff = 0, eQ = 0, eMisc = 0, eW1 = 0, eW2 = 0, ePidx = 0, eQwLo = 0, eQwHi = 0, eRwLo = 0, eRwHi = 0;
    // pendRes: the problem whose result this lane has yet to store (its state stays untouched until then); rowsFin: its last row
    bool pendFlush = false; int pendRes = -1, rowsFin = 0; unsigned pStart = 0;
    int wslot = 0; unsigned flush = 0; bool dirty = false, justDone = false;
    YD_GLOBAL uint32_t *chunkPtr = toGlobal(A.trace); bool noMem = false;
    auto takeChunk = [&]() {                                                 // wave-uniform: the chunk of flushes [flush, flush + 16)
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(A.chunkCount, 1u);
        c = uniU(c);
        const unsigned ci = flush / YD_CHUNK_FLUSHES;
        if (c >= A.nChunks || ci >= A.maxCh) { noMem = true; c = 0; if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_TRACEMEM); }
        if (lane == 0 && ci < A.maxCh) A.waveChunks[(size_t)wave * A.maxCh + ci] = c;
        chunkPtr = toGlobal(A.trace) + (size_t)c * YD_CHUNK_DWORDS;
    };
    takeChunk();
    // Hand-over: eight lanes write each block, 16 bytes apiece (see k_ext_rows).  f = the lanes whose block leaves, slot = where the wave's blocks of this flush go.
    auto flushBlocks = [&](unsigned long long f, YD_GLOBAL uint32_t *slot) {
        if (f != 0ull && !noMem) {
            int ln = lane; asm volatile("" : "+v"(ln));                      // (opaque: the eight address pairs are computed here, not kept in registers across the row code)
            const int piece = ln & 7, wv = (int)(threadIdx.x >> 6);
#pragma unroll
            for (int g = 0; g < 8; g++) {
                if (((f >> (g * 8)) & 0xFFull) == 0ull) continue;            // wave-uniform
                const int o = g * 8 + (ln >> 3);
                const uint32_t *src = &sBlk[(wv * 64 + o) * YD_LDS_STRIDE + piece * 4];
                yd_u32x4 v; v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                if ((f >> o) & 1ull) YD_STORE_NT((YD_GLOBAL yd_u32x4 *)(slot + o * YD_LANE_DWORDS + piece * 4), v);      // (non-temporal: see common.h)
            }
        }
    };
    auto nextFlush = [&]() { flush++; if (flush % YD_CHUNK_FLUSHES == 0u) takeChunk(); };     // wave-uniform
    bool firstFill = true;
    for (;;) {
        // (Round 6, measured and dropped: every wave-uniform condition of this loop through UNI_B and every uniform counter through uni().  Left to itself the compiler
        // takes the loop's exits for lane-dependent -- an EXEC-masked loop, the uniform state in vector registers, twenty moves and thirty mask instructions a pass --
        // but the scalar version waits for a v_readfirstlane before every branch: 14.42 -> 15.84 ms a launch, profiles/r06_rows_kernel_passes.txt.)
        if (YD_ROWS_EXIT(noMem)) break;
        // What the previous iteration loaded is consumed HERE, before this iteration issues any store (the memory counter is in-order: a wait for these loads
        // further down would also wait for the stores issued in between).  Slide the reference window: pair k takes pair k+1; pair 10's low half takes what
        // was pair 1's high half, its high half the new base.  (A lane that starts a problem below overwrites the window.)
        {
#pragma unroll
            for (int k = 0; k + 1 < YD_NP; k++) rc[k] = rc[k + 1];
            rc[YD_NP - 1] = (rc[0] >> 16) | ((nbNext | YD_RCREAL) << 16);
        }
        // the dwords the previous pass loaded go to the ends of the windows (entries in consumption order: a reverse extension's bytes, and its nibbles, swapped)
        if (insQ) {
            if (pendQ) {
#if YD_ROWS_LDSWIN
                // a fresh piece: into the lane's slot; its dword jQ is the one due
                if (newQ) { sWin[0][tid] = GQ.x; sWin[1][tid] = GQ.y; sWin[2][tid] = GQ.z; sWin[3][tid] = GQ.w; qLd = sWin[jQ][tid]; }
#endif
                const uint32_t sw = ((qLd & 0x0F0F0F0Fu) << 4) | ((qLd >> 4) & 0x0F0F0F0Fu);      // forward: the even offset (high nibble) first
                const uint32_t v = qStep < 0 ? __builtin_amdgcn_perm(0u, qLd, 0x00010203u) : sw;   // reverse: bytes swapped, each byte's low nibble (the higher offset) first
                const unsigned long long t = (unsigned long long)v << (4 * (qHave & 15));
                qwLo |= (uint32_t)t; qwHi |= (uint32_t)(t >> 32); qHave += 8; qNext += 8;
            }
            insQ = false;
        }
        if (insR) {
            if (pendR) {
#if YD_ROWS_LDSWIN
                if (newR) { sWin[4][tid] = GR.x; sWin[5][tid] = GR.y; sWin[6][tid] = GR.z; sWin[7][tid] = GR.w; rLd = sWin[4 + jR][tid]; }
#endif
                const uint32_t sw = ((rLd & 0x0F0F0F0Fu) << 4) | ((rLd >> 4) & 0x0F0F0F0Fu);      // forward: the even offset (high nibble) first
                const uint32_t v = qStep < 0 ? __builtin_amdgcn_perm(0u, rLd, 0x00010203u) : sw;   // reverse: bytes swapped, each byte's low nibble (the higher offset) first
                const unsigned long long t = (unsigned long long)v << (4 * (rHave & 15));
                rwLo |= (uint32_t)t; rwHi |= (uint32_t)(t >> 32); rHave += 8; rNext += 8;
            }
            insR = false;
        }
        // the blocks the previous iteration completed: their place is fixed now (a problem that starts below notes the flush ITS first block will go out
        // with), the stores themselves are issued after this iteration's loads (program order = the order the memory counter retires in)
        unsigned long long flushNow = 0ull; YD_GLOBAL uint32_t *flushSlot = chunkPtr;
        if (wslot == 0) { flushNow = __ballot(pendFlush); pendFlush = false; if (flushNow != 0ull) { flushSlot = chunkPtr + (size_t)(flush % YD_CHUNK_FLUSHES) * 32u; nextFlush();
            } }
        // ---- refill (k_ext_rows') ----
        for (;;) {
            const unsigned long long need = __ballot(p < 0 && !done && !justDone && (!YD_ROWS_RESBATCH || pendRes < 0));
            if (!need) break;
            if (__builtin_popcountll(need) < YD_REFILL_MIN && __ballot(p >= 0) != 0ull && !firstFill) break;
#ifdef YD_PROF
            pfRefill++; if (poolNext >= poolCount) pfPool++;
#endif
            if (poolNext >= poolCount) {
                unsigned base = 0;
                if (!exhausted) { if (lane == 0) base = atomicAdd(A.queue, 64u); base = uniU(base); if (base >= A.nProb) exhausted = true; }
                if (exhausted) { if (p < 0) done = true; break; }
                poolCount = (int)min(64u, A.nProb - base); poolNext = 0;
                eLens = 0; bool isCall = false;
                if (lane < poolCount) {
                    const unsigned np = A.order ? __builtin_nontemporal_load(toGlobal(&A.order[base + (unsigned)lane])) : base + (unsigned)lane;
                    ePidx = np;
                    ExtProb pr; { const yd_u32x4 v = __builtin_nontemporal_load((YD_GLOBAL const yd_u32x4 *)toGlobal(&A.probs[np])); pr.qBase = v.x; pr.rOff = v.y;
                        pr.qOff = (uint16_t)(v.z & 0xFFFFu); pr.qLen = (uint16_t)(v.z >> 16); pr.flags = v.w; }
                    int ql = 0; uint32_t rl = 0; const bool rv_ = (pr.flags & XP_REV) != 0;
                    if (pr.flags & XP_VALID) {                              // findAGSExtension, SW.cpp:479-516
                        isCall = true;
                        ql = pr.qLen;
                        rl = (uint32_t)(ql + bandwidth);
                        if (rv_ && rl > pr.rOff) { rl = pr.rOff + 1; ql = (int)rl - bandwidth; }
                        if (!rv_ && (pr.rOff + rl) > maxROff) { rl = maxROff - pr.rOff; ql = (int)rl - bandwidth; }
                        if (ql > 0) { ql &= 0xFFFF; rl &= 0xFFFF; }
                    }
                    if (ql <= 0) { ExtRes r; r.score = 0; r.maxi = r.maxj = 0; r.opsOff = r.nOps = 0; r.where = 0; r.rows = r.cells = 0; A.res[np] = r; }
                    else {
                        eLens = (uint32_t)ql | (rl << 16); eROff = pr.rOff; eQ = pr.qBase + pr.qOff;
                        YD_GLOBAL const uint8_t *qb4 = toGlobal((pr.flags & XP_STRAND) ? A.rev4 : A.fwd4);
                        { const uint32_t b0 = qb4[eQ >> 1]; eMisc = (pr.flags & 3u) | (((eQ & 1u) ? (b0 & 15u) : (b0 >> 4)) << 8); }
                        eW1 = 0; eW2 = 0;                                     // reference indices 0..10 (nibble c - leftR of the window, as in k_ext_rows)
                        for (int c = leftR; c < YD_LW; c++) {
                            const int idx = c - leftR; uint32_t nib = 15u;
                            if (idx < (int)rl) { const uint32_t off = rv_ ? pr.rOff - (uint32_t)idx : pr.rOff + (uint32_t)idx; const uint32_t b = gBases[off >> 1];
                                nib = (off & 1u) ? (b & 15u) : (b >> 4); }
                            const uint32_t sh = (uint32_t)(c & 7) * 4u;
                            if (c < 16) eW1 |= nib << sh; else eW2 |= nib << sh;
                        }
                        // first fill of the stream windows: query codes from index 1 (index 0 is in eMisc), reference nibbles from index 11 (0..10 are in eW1/eW2);
                        // a partial aligned dword, then a full one -- each only if it holds an entry of the extension
                        uint32_t qh = 0, rh = 0; eQwLo = eQwHi = eRwLo = eRwHi = 0;
                        {   // query codes from index 1, out of the packed array: the reference's nibble logic below with 1 for 11
                            const uint32_t n1 = rv_ ? eQ - 1u : eQ + 1u, lo3 = n1 & 7u, c1 = rv_ ? lo3 + 1u : 8u - lo3;
                            uint32_t v1 = 0, v2 = 0;
                            if (ql >= 2) {
                                YD_GLOBAL const uint32_t *d1 = (YD_GLOBAL const uint32_t *)(qb4 + ((n1 >> 1) & ~3u));
                                const uint32_t w = *d1;
                                v1 = rv_ ? __builtin_amdgcn_perm(0u, w, 0x00010203u) >> (4u * (7u - lo3)) : (((w & 0x0F0F0F0Fu) << 4) | ((w >> 4) & 0x0F0F0F0Fu)) >> (4u * lo3);
                                qh = c1;
                                if (ql > (int)(1u + c1)) { const uint32_t w2 = rv_ ? d1[-1] : d1[1];
                                    v2 = rv_ ? __builtin_amdgcn_perm(0u, w2, 0x00010203u) : (((w2 & 0x0F0F0F0Fu) << 4) | ((w2 >> 4) & 0x0F0F0F0Fu)); qh = c1 + 8u; }
                            }
                            const unsigned long long t = (unsigned long long)v1 | (c1 < 16u ? ((unsigned long long)v2 << (4u * c1)) : 0ull);
                            eQwLo = (uint32_t)t; eQwHi = (uint32_t)(t >> 32);
                        }
                        {
                            const uint32_t n11 = rv_ ? pr.rOff - 11u : pr.rOff + 11u, lo3 = n11 & 7u, c1 = rv_ ? lo3 + 1u : 8u - lo3;
                            uint32_t v1 = 0, v2 = 0;
                            if (rl > 11u) {
                                YD_GLOBAL const uint32_t *d1 = (YD_GLOBAL const uint32_t *)(gBases + ((n11 >> 1) & ~3u));
                                const uint32_t w = *d1;
                                v1 = rv_ ? __builtin_amdgcn_perm(0u, w, 0x00010203u) >> (4u * (7u - lo3)) : (((w & 0x0F0F0F0Fu) << 4) | ((w >> 4) & 0x0F0F0F0Fu)) >> (4u * lo3);
                                rh = c1;
                                if (rl > 11u + c1) { const uint32_t w2 = rv_ ? d1[-1] : d1[1];
                                    v2 = rv_ ? __builtin_amdgcn_perm(0u, w2, 0x00010203u) : (((w2 & 0x0F0F0F0Fu) << 4) | ((w2 >> 4) & 0x0F0F0F0Fu)); rh = c1 + 8u; }
                            }
                            const unsigned long long t = (unsigned long long)v1 | (c1 < 16u ? ((unsigned long long)v2 << (4u * c1)) : 0ull);
                            eRwLo = (uint32_t)t; eRwHi = (uint32_t)(t >> 32);
                        }
                        eMisc |= (qh | (rh << 8)) << 16;
                    }
                }
                calls += (unsigned)__builtin_popcountll(__ballot(isCall));      // (wave-uniform: a scalar register)
            }
            const int nNeed = __builtin_popcountll(need), avail = poolCount - poolNext;
            const int e = poolNext + __builtin_popcountll(need & lanesBelow);
            const bool take = (p < 0 && !done && !justDone && (!YD_ROWS_RESBATCH || pendRes < 0)) && e < poolCount;
            const int src = take ? e : lane;
            const uint32_t gLens = (uint32_t)__shfl((int)eLens, src, 64), gROff = (uint32_t)__shfl((int)eROff, src, 64), gQ = (uint32_t)__shfl((int)eQ, src, 64),
                gMisc = (uint32_t)__shfl((int)eMisc, src, 64);
            const uint32_t gW1 = (uint32_t)__shfl((int)eW1, src, 64), gW2 = (uint32_t)__shfl((int)eW2, src, 64);
            const uint32_t gPidx = (uint32_t)__shfl((int)ePidx, src, 64);
            const uint32_t gQwLo = (uint32_t)__shfl((int)eQwLo, src, 64), gQwHi = (uint32_t)__shfl((int)eQwHi, src, 64), gRwLo = (uint32_t)__shfl((int)eRwLo, src, 64),
                gRwHi = (uint32_t)__shfl((int)eRwHi, src, 64);
            const uint32_t gHave = gMisc >> 16;
            const bool init = take && gLens != 0u;
            // A fresh problem: low halves = row 0 of columns 0..10 (the origin (0, 10): V = 0, F = -GO; sentinel left of it), high halves = the sentinel
            // ("row -1"; the first iteration turns it into row 0 of columns 11..20).  Reference codes: low halves of pairs 0..9 lie left of the matrix in
            // row 1 (bit 15: not a cell; it slides out with the window), pair 10 low = index 0, pair k high = index k.
#pragma unroll
            for (int k = 0; k < YD_NP; k++) {
                // the origin: V = 0, i.e. Vg = -GOE
                const uint32_t iV = k == leftR ? ((LWg & 0xFFFF0000u) | ((uint32_t)(-(GO + GE)) & 0xFFFFu)) : LWg, iF = k == leftR
                    ? ((LWp & 0xFFFF0000u) | ((uint32_t)(-GO) & 0xFFFFu)) : LWp;
                const int c = leftR + k;                                      // window nibble of reference index k
                const uint32_t nibHi = (((c < 16 ? gW1 : gW2) >> ((c & 7) * 4)) & 15u) | YD_RCREAL, nibLo = k == leftR ? (((gW1 >> ((leftR & 7) * 4)) & 15u) | YD_RCREAL) : 0x800Fu;
                PV[k] = init ? iV : PV[k]; PF[k] = init ? iF : PF[k]; rc[k] = init ? (nibLo | (nibHi << 16)) : rc[k];
            }
            if (init) {
                p = (int)gPidx; qLen = (int)(gLens & 0xFFFFu); i = 0; maxScore = YD_LWORST; maxi = 0; carryE = LWp; rvLo = YD_LWORST;
                const bool rev = (gMisc & XP_REV) != 0;
                q4 = toGlobal((gMisc & XP_STRAND) ? A.rev4 : A.fwd4); qPos = gQ; qStep = rev ? -1 : 1;
                qcNext = (int)((gMisc >> 8) & 0xFFu); qcPrev = 0;
                qwLo = gQwLo; qwHi = gQwHi; qHave = (int)(gHave & 0xFFu); qNext = 1 + qHave; pendQ = false;
                // iteration i takes reference index i + 10 from the window (valid while below rLen)
                rOffP = gROff; rLenP = (int)(gLens >> 16); rLeft = rLenP - bandwidth;
                rwLo = gRwLo; rwHi = gRwHi; rHave = (int)(gHave >> 8); rNext = 11 + rHave; pendR = false;
                pStart = (flush << 4) | (unsigned)wslot;
#if YD_ROWS_LDSWIN
                freshQ = true; freshR = true;                                // (the lane's staged pieces are the previous problem's)
#endif
            }
            poolNext += nNeed < avail ? nNeed : avail;
        }
        firstFill = false;
        const bool last = __ballot(p >= 0) == 0ull && __ballot(justDone) == 0ull;      // wave-uniform: nothing runs, nothing left; leave once this pass' stores are out

        // ---- one iteration in every lane: row i in the low halves, row i - 1 in the high halves ----
        const bool busy = p >= 0;
        ++i;
        const int qc = qcNext;
        // next iteration's query code (index i) and the reference nibble that enters the window (index i + 10; 15 beyond the reference): off the windows' low ends
        qcNext = (int)(qwLo & 15u); qwLo = __builtin_amdgcn_alignbit(qwHi, qwLo, 4); qwHi >>= 4; qHave--;
        { rLeft--; nbNext = rLeft > 0 ? (rwLo & 15u) : 15u; rwLo = __builtin_amdgcn_alignbit(rwHi, rwLo, 4); rwHi >>= 4; rHave--; }
        // refills (wave-uniform schedule): the loads are consumed at the top of the next pass
        if (wslot == 0) {
            pendQ = qHave <= 8 && qNext < qLen;
#if YD_ROWS_LDSWIN
            // the dword that is due: out of the lane's staged piece -- or, when it is the first dword of the next aligned 16-byte piece (the last one going down), or the
            // first refill of a problem, that whole piece from HBM (it goes to LDS at the top of the next pass).  An aligned piece that holds a dword of the
            // extension lies inside the arrays (they start on 256-byte boundaries and have slack behind).
            newQ = false;
            if (pendQ) {
                // (da: dword-aligned byte offset)
                const uint32_t n = qStep < 0 ? qPos - (uint32_t)qNext - 7u : qPos + (uint32_t)qNext, da = n >> 1; const int j = (int)((da >> 2) & 3u);
                if (freshQ || j == (qStep < 0 ? 3 : 0)) { GQ = *(YD_GLOBAL const yd_u32x4 *)(q4 + (da & ~15u)); newQ = true; freshQ = false; jQ = j; }
                else qLd = sWin[j][tid];
            }
            insQ = true;
            pendR = rHave <= 8 && rNext < rLenP;
            newR = false;
            if (pendR) {
                const uint32_t n = qStep < 0 ? rOffP - (uint32_t)rNext - 7u : rOffP + (uint32_t)rNext, da = n >> 1; const int j = (int)((da >> 2) & 3u);
                if (freshR || j == (qStep < 0 ? 3 : 0)) { GR = *(YD_GLOBAL const yd_u32x4 *)(gBases + (da & ~15u)); newR = true; freshR = false; jR = j; }
                else rLd = sWin[4 + j][tid];
            }
            insR = true;
#else
            // (nibble qPos + qNext * qStep is dword-aligned going up, the last nibble of a dword going down)
            if (pendQ) { const uint32_t n = qStep < 0 ? qPos - (uint32_t)qNext - 7u : qPos + (uint32_t)qNext; qLd = *(YD_GLOBAL const uint32_t *)(q4 + (n >> 1)); }
            insQ = true;
            pendR = rHave <= 8 && rNext < rLenP;
            if (pendR) { const uint32_t n = qStep < 0 ? rOffP - (uint32_t)rNext - 7u : rOffP + (uint32_t)rNext; rLd = *(YD_GLOBAL const uint32_t *)(gBases + (n >> 1)); }
            insR = true;
#endif
        }
        // the iteration's stores, behind its loads: a finished problem's result and the blocks that leave
#if YD_ROWS_RESBATCH
        // (wave-uniform: the results wait until as many lanes are idle as a refill asks for -- the pass before that refill -- or nothing runs any more)
        const bool resNow = __ballot(pendRes >= 0) != 0ull && (__builtin_popcountll(__ballot(p < 0 && !done)) >= YD_REFILL_MIN || __ballot(p >= 0) == 0ull);
#else
        const bool resNow = true;
#endif
#ifdef YD_PROF
        pfPass++; if (resNow && __ballot(pendRes >= 0) != 0ull) pfRes++; if (flushNow != 0ull) pfFlush++;
#endif
        if (resNow && pendRes >= 0) {
            // the problem ended in an earlier pass at row rowsFin (the lane sat out every refill since, so maxScore / maxi / maxj / pStart
            // are still its own).  Work of the call: row r has 21 - max(11 - r, 0) real cells.
            const unsigned rowF = YD_ROWS_RESBATCH ? (unsigned)rowsFin : (unsigned)(i - 2), m = rowF < (unsigned)leftR ? rowF : (unsigned)leftR, nCells = __umul24((unsigned)YD_LW,
                rowF) - (__umul24((unsigned)(leftR + 1), m) - __umul24(m, m + 1u) / 2u);
            rows += rowF; cells += nCells;
            int maxj = 0;                                                     // the first column of the kept strip's half that holds the maximum
#pragma unroll
            for (int k = YD_NP - 1; k >= 0; k--) {      // (the strip holds Vg; pair k's value of the maximum's side: half k & 1 of SV[k / 2])
                const int v = (k & 1) ? (int)(short)(SV[k >> 1] >> 16) : (int)(short)(SV[k >> 1] & 0xFFFFu);
                if (v == maxScore - (GO + GE)) maxj = k + (maxSide ? YD_NP : 0);
            }
            ExtRes r; r.score = maxScore > 0 ? maxScore : 0; r.maxi = maxi; r.maxj = maxj; r.opsOff = pStart >> 4; r.nOps = 0;
            r.where = (pStart & 15u) | ((uint32_t)lane << 4) | (wave << 10); r.rows = rowF; r.cells = nCells;
            // (the result, like the trace blocks, is stored non-temporally, and the pool's problem records are loaded so: neither comes back to this kernel's L2)
            { const yd_u32x4 lo = {(uint32_t)r.score, (uint32_t)r.maxi, (uint32_t)r.maxj, r.opsOff}, hi = {r.nOps, r.where, r.rows, r.cells};
                YD_GLOBAL yd_u32x4 *dst = (YD_GLOBAL yd_u32x4 *)toGlobal(&A.res[pendRes]); YD_STORE_NT(dst, lo); YD_STORE_NT(dst + 1, hi); } pendRes = -1;
        }
        flushBlocks(flushNow, flushSlot);
        if (YD_ROWS_EXIT(last)) break;
        const uint32_t qcP = ((uint32_t)qc | ((uint32_t)qcPrev << 16)) ^ (YD_RCREAL * 0x10001u);      // code ^ qcP = nibble ^ query code
        qcPrev = qc;
        uint32_t qcPv = qcP; asm volatile("" : "+v"(qcPv));                   // (opaque: else the constant is re-applied in every pair)
        uint32_t PVCol = (PV[YD_NP - 1] << 16) | LWgLo;                      // (Vg) low: nothing left of column 0; high: V(i-1, 10)
        uint32_t PE = (carryE << 16) | LWlo;                                 //                                       E(i-1, 10)
        uint32_t rowMax = LWp, dV = PV[0];
        uint32_t accA = 0, accB = 0, accA2 = 0, accB2 = 0, accM = 0;
#pragma unroll
        for (int k = 0; k < YD_NP; k++) {
            const uint32_t mm = pkMinU(rc[k] ^ qcPv, ONEv);                   // 0 = match, 1 = mismatch, per half
            uint32_t V = pkAdd(dV, pkMad(mm, NEGKv, MSv));                     // G = diagonal's Vg + (MS + GOE | -RC + GOE)
            const uint32_t CE = pkSub(PE, GEp), NE = PVCol;                  // (the left neighbour's Vg IS the gap opened from it)
            PE = pkMax(CE, NE);
            const uint32_t dE = pkSub(CE, NE);                               // >= 0: the E run continues (ties continue, SW.cpp:1029-1033)
            const uint32_t dT = pkSub(PE, V);                                // >= 0: E wins over G ('>=' in extension mode, SW.cpp:1036)
            V = pkMax(V, PE);
            uint32_t upV, upF;
            // column 10's upper neighbour = column 11 of row i-1, just computed
            if (k + 1 < YD_NP) { upV = PV[k + 1]; upF = PF[k + 1]; } else { upV = PV[0] >> 16; upF = PF[0] >> 16; }
            const uint32_t CF = pkSub(upF, GEp), NF = upV;
            const uint32_t F = pkMax(CF, NF);
            const uint32_t dF = pkSub(CF, NF);
            const uint32_t dU = pkSub(F, V);
            V = pkMax(V, F);
            // sign bytes of both halves -> bit planes
            const uint32_t S1 = __builtin_amdgcn_perm(dU, dT, 0x07050301u), S2 = __builtin_amdgcn_perm(dF, dE, 0x07050301u);
            if (k < 8) { accA = bfi(0x80808080u, S1, accA >> 1); accB = bfi(0x80808080u, S2, accB >> 1); asm volatile("" : "+v"(accA), "+v"(accB)); }
            else { accA2 = bfi(0x80808080u, S1, accA2 >> 1); accB2 = bfi(0x80808080u, S2, accB2 >> 1); asm volatile("" : "+v"(accA2), "+v"(accB2)); }
            accM = (accM << 1) | mm;
            // row-major first maximum over the real cells, per half.  Low half: the columns left of the matrix (one of them is the boundary column, with a
            // real-sized value) have a reference code that is negative as a score (0x800F), the cells of the matrix one that is above every score (0x7FF0 | nibble):
            // one minimum.  High half of pair 10: the column that does not exist.
            uint32_t Vm = V;
            if (k < leftR) Vm = pkMin(V, rc[k]);
            if (k == YD_NP - 1) { V = bfi(0x0000FFFFu, V, LWp); Vm = V; }
            rowMax = pkMax(rowMax, Vm);
            { const uint32_t Vg = pkSub(V, GOEp); PV[k] = Vg; PVCol = Vg; } PF[k] = k == YD_NP - 1 ? bfi(0x0000FFFFu, F, LWp) : F; dV = upV;
            __builtin_amdgcn_sched_barrier(0);
        }
        carryE = PE;
        { const int slot = wslot * 4;                                        // this iteration's record goes to the lane's LDS block (the same slot in every lane)
          myBlk[slot] = accA2 | (accB2 >> 3); myBlk[slot + 1] = accM; myBlk[slot + 2] = accA; myBlk[slot + 3] = accB; }
        // row i - 1 is complete now: its maximum (columns 0..10 from the previous iteration, 11..20 from this one; the first one in column order), X-drop test
        // The scan is row-major with a strict '>': row i-1's columns 11..20 (this iteration's high halves) come after its columns 0..10 (seen one iteration ago),
        // and before row i's columns 0..10 (this iteration's low halves, counted unless row i-1 ends the problem).  The strip of the iteration that set the
        // maximum is kept; the column is looked up in it when the problem ends.
        const int rvHi = (int)(short)(rowMax >> 16);
        const int rv = rvHi > rvLo ? rvHi : rvLo;
        rvLo = (int)(short)(rowMax & 0xFFFFu);
        const int row = i - 1;
        bool snap = false;
        if (busy && row >= 1 && rvHi > maxScore) { maxScore = rvHi; maxi = row; maxSide = 1; snap = true; }
        const bool fin = busy && row >= 1 && (rv < maxScore - XC || row >= qLen);
        if (busy && !fin && i >= 1 && rvLo > maxScore) { maxScore = rvLo; maxi = i; maxSide = 0; snap = true; }
        if (snap) {
            const uint32_t sel = maxSide ? 0x07060302u : 0x05040100u;          // v_perm_b32 (S0, S1, sel): bytes 0..3 of S1, 4..7 of S0 -> the side's half of pairs 2 j and 2 j + 1
#pragma unroll
            for (int k = 0; k < YD_NSV; k++) SV[k] = __builtin_amdgcn_perm(PV[2 * k + 1 < YD_NP ? 2 * k + 1 : 2 * k], PV[2 * k], sel);
        }
        if (busy) dirty = true;
        if (wslot == 7) { pendFlush = dirty; dirty = false; wslot = 0; } else wslot++;      // wave-uniform
        justDone = fin;                                                      // the next record slot stays empty behind a finished problem (its traceback's spare record)
        // A finished lane keeps its state: the result is stored from it in the next pass; its byte streams stay inside the finished extension (clamped index,
        // rLeft <= 0), so an idle lane's loads are harmless.
        if (fin) { pendRes = p; p = -1; rowsFin = row; }
#ifdef YD_PROF
        if (__ballot(snap) != 0ull) pfSnap++; pfBusy += (unsigned)__builtin_popcountll(__ballot(busy));
#endif
    }
    if (wslot != 0 && dirty) pendFlush = true;
    { const unsigned long long f = __ballot(pendFlush); if (!noMem && f != 0ull) { flushBlocks(f, chunkPtr + (size_t)(flush % YD_CHUNK_FLUSHES) * 32u); nextFlush(); } }
    // (a result still pending here belongs to a launch that ran out of arena: the host zeroes the results and redoes the stage)
    if (!SECOND && A.clock && lane == 0) atomicMax(&A.clock[1], (unsigned long long)wall_clock64());
    unsigned c0 = calls, c1 = (unsigned)waveSumI((int)rows);
    unsigned long long cc = cells;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { cc += (unsigned long long)__shfl_xor((long long)cc, d, 64); }
#ifdef YD_PROF
    if (lane == 0 && !SECOND) { const unsigned v[7] = {pfPass, pfRes, pfRefill, pfSnap, pfBusy, pfFlush, pfPool};
        for (int k = 0; k < 7; k++) atomicAdd(&gRowsProf[k], (unsigned long long)v[k]); }
#endif
    if (lane == 0 && !SECOND && A.ctr) {
        unsigned long long *c = A.ctr->v;
        atomicAdd(&c[C_EXT_CALLS], (unsigned long long)c0); atomicAdd(&c[C_EXT_ROWS], (unsigned long long)c1); atomicAdd(&c[C_EXT_CELLS], cc);
        atomicAdd(&c[C_TOUCHED], (unsigned long long)c1 + (unsigned long long)c0 * (unsigned long long)(4 * A.P.bandWidth + 1));
    }
}

// ---- traceback over the packed records, lane per problem (SW.cpp:1138-1195; see k_ext_trace for the staging of the ops inside the strip) ---------------
// Logical strip: record n (1-based, + phase) = dwords [(n-1)/8 * 32 + (n-1)%8 * 4, +4).  Cell (y, x) is pair k = x % 11 (x < 22), half h = x / 11, of record y + h.
struct PkRec { uint32_t a, m, ab2, b; };
__device__ __forceinline__ PkRec pkLoadRec(YD_GLOBAL const uint32_t *cp) { const yd_u32x4 v = *(YD_GLOBAL const yd_u32x4 *)cp; PkRec r; r.ab2 = v.x; r.m = v.y; r.a = v.z;
    r.b = v.w; return r; }
__device__ __forceinline__ int pkRecWord(int n) { return ((n - 1) >> 3) * 32 + ((n - 1) & 7) * 4; }
__device__ __forceinline__ bool pkContE(const PkRec &r, int k, int h) { const uint32_t d = k < 8 ? r.b : r.ab2; const int bit = k < 8 ? k : k - 6;
    return ((d >> (8 * h + bit)) & 1u) == 0u; }
__device__ __forceinline__ bool pkContF(const PkRec &r, int k, int h) { const uint32_t d = k < 8 ? r.b : r.ab2; const int bit = k < 8 ? k : k - 6;
    return ((d >> (16 + 8 * h + bit)) & 1u) == 0u; }
// dwords per problem in the traceback's block cache (32 + padding; 16-byte aligned).  (32 with an XOR swizzle -- a fifth workgroup per CU -- was no faster.)
#define YD_TSTRIDE 36
// Inside a problem's block the four dwords of every 16-byte record are stored XOR-ed by (problem >> 3) & 3 (YD_TSWZ).  A lane reads single dwords of its own
// problem's records (ds_read_b32: 32 lanes a group, bank = dword address mod 32): with stride 36 and whole records in place lane l's dword c of slot s sits on bank
// 4 (l + s) + c -- the 32 lanes of a group on the eight banks that are c modulo 4, four lanes each (33.6 % of the kernel's LDS cycles were such conflicts, rounds 3-4).
// With the dword index XOR-ed by the lane's (l >> 3) & 3 the four lanes that shared a bank read four different ones.  The staging store writes whole records, eight
// lanes to a problem, (problem >> 3) & 3 = g & 3 a compile-time constant of the unrolled loop: the permutation is a renaming of the four registers it stores.
#ifndef YD_TRACE_SWZ
// (measured: 3.23 -> 3.39 ms a launch WITH the swizzle, profiles/r05_trace_swizzle.txt -- the XORs and six more registers cost more than the conflicts)
#define YD_TRACE_SWZ 0
#endif
#define YD_TSWZ(problem) (YD_TRACE_SWZ ? (((problem) >> 3) & 3) : 0)
#ifndef YD_TRACE_BS
// threads of a traceback workgroup (its waves are independent: the size only sets the granule of LDS -- 9 KB a wave -- a CU hands out)
#define YD_TRACE_BS 256
#endif
__global__ void __launch_bounds__(YD_TRACE_BS) k_ext_trace_pk(ExtArgs A)
{
    YD_HIGH_PRIO();
    // The 64 lanes of a wave read 64 different 128-byte blocks per pass.  Read by their own lanes -- eight 8-byte pieces each -- that is 512 line requests a
    // pass, and the kernel's time followed the number of such requests, not the bytes.  Here the wave fetches the blocks TOGETHER: eight loads of 16 bytes per
    // lane, each covering eight whole blocks (eight lanes per block), through LDS ([problem][36 dwords], 36 KB a workgroup); a lane then reads its own records there, and a gap
    // run that stays inside the block needs no further load.
    __shared__ uint32_t sBlkT[YD_TRACE_BS / 64][64 * YD_TSTRIDE];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; const int lane = laneId();
    uint32_t *const wBlk = sBlkT[threadIdx.x >> 6]; const uint32_t *const myRec = wBlk + lane * YD_TSTRIDE;
    if (*toGlobal(A.errFlag) != 0) return;
    const bool live = t < A.nProb;
    const uint32_t p = live ? ((A.order && !(A.dbgMode & 2)) ? A.order[t] : t) : 0u;
    ExtRes r; r.score = 0; r.nOps = 0;
    if (live) r = A.res[p];
    const bool walk = live && r.score > 0;
    constexpr int leftR = YD_LBAND;
    int n = 0; bool bad = false; int E = 0;
    ExtStrip S; S.arena = toGlobal(A.trace); S.cIdx = -1; S.cBase = S.arena; S.f0 = 0; S.laneOff = 0; S.tab = toGlobal(A.waveChunks); S.limit = 0; S.wild = false;
    struct Cur { YD_GLOBAL uint32_t *cp; int rr, w; unsigned fb; };
    int y = 0, x = 0, prev = -1, acc = 0, row0w = 0;
    Cur u; u.cp = S.arena; u.rr = 0; u.w = 0; u.fb = 0;
    if (walk) {
        const int ph = (int)(r.where & 15u);
        S.f0 = r.opsOff; S.laneOff = ((r.where >> 4) & 63u) * YD_LANE_DWORDS; S.tab = toGlobal(A.waveChunks) + (size_t)(r.where >> 10) * A.maxCh;
        E = pkRecWord(r.maxi + 2 + ph) + 4; S.limit = E;                         // record maxi + 2 is the spare one (computed, or the idle slot behind the problem)
        y = r.maxi; x = r.maxj;
        // physical cursor on the record of the current cell (record y + h): cp = its first dword, rr = its slot in the block, w = its logical offset, fb = its flush
        { const int idx = y + (x >= YD_NP ? 1 : 0) - 1 + ph; u.rr = idx & 7; u.w = (idx >> 3) * 32 + u.rr * 4; u.fb = S.f0 + (unsigned)(idx >> 3); u.cp = S.at(u.w); }
        row0w = pkRecWord(1 + ph) - 4;                                      // with the cursor there every record is consumed
    }
    {
        auto stepUp = [&](Cur &c) {                                          // one record towards the origin (a lane's blocks are consecutive inside a chunk)
            c.w -= 4;
            if (c.rr != 0) { c.rr--; c.cp -= 4; return; }
            c.rr = 7; c.fb--;
            if (c.fb % YD_CHUNK_FLUSHES == YD_CHUNK_FLUSHES - 1u) c.cp = S.arena + (size_t)S.tab[c.fb / YD_CHUNK_FLUSHES] * YD_CHUNK_DWORDS + S.laneOff + (YD_CHUNK_FLUSHES - 1u) *
                32u + 28u;
            else c.cp -= 4;
        };
        // free for staging: the records above record (cursor + 1) -- the record after the cursor's may still hold the high half of the cursor's row
        auto flush = [&]() { const int wp = E - 1 - n; if (wp < u.w + 8) bad = true; else *S.at(wp) = opMake(prev, acc); n++; };
        auto put = [&](int code, int len) { if (prev != code) { if (prev >= 0) flush(); prev = code; acc = len; } else acc += len; };
#ifdef YD_PROF
        unsigned tpPass = 0, tpAct = 0, tpRounds = 0, tpFlushR = 0, tpDel = 0, tpIns = 0, tpRows = 0; const bool tpWalks = __ballot(walk) != 0ull;
#endif
        for (int guard = 0;; guard++) {
            const bool act = walk && guard < 70000 && y > 0 && x >= 0 && x < YD_LW;
            if (__ballot(act) == 0ull) break;                                  // wave-uniform
#ifdef YD_PROF
            tpPass++; tpAct += (unsigned)__builtin_popcountll(__ballot(act));
#endif
            {   // the wave's 64 current blocks -> LDS
                const unsigned long long myBase = (unsigned long long)(act ? u.cp - u.rr * 4 : S.arena);
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const int src = g * 8 + (lane >> 3);
                    const unsigned long long b = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(myBase >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)myBase, src,
                        64);
                    const yd_u32x4 v = *(YD_GLOBAL const yd_u32x4 *)((YD_GLOBAL const uint32_t *)b + (lane & 7) * 4);
                    // YD_TSWZ(src) = g & 3 (a renaming: g is a constant of the unrolled loop)
                    yd_u32x4 sw; { const uint32_t c[4] = {v.x, v.y, v.z, v.w}; constexpr int z = YD_TRACE_SWZ ? 3 : 0; sw.x = c[(0 ^ g) & z]; sw.y = c[1 ^ (g & z)];
                        sw.z = c[2 ^ (g & z)]; sw.w = c[3 ^ (g & z)]; }
                    *(yd_u32x4 *)(wBlk + src * YD_TSTRIDE + (lane & 7) * 4) = sw;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!act) continue;
            const int rr0 = u.rr;                                            // the cursor's slot when the block was fetched
            const int swz = YD_TSWZ(lane);                                   // this lane's records: dword c is stored at c ^ swz
            auto blkRec = [&](int slot, YD_GLOBAL const uint32_t *cp) -> PkRec {  // the record in `slot` of the fetched block, or (slot < 0: a block further up) the one at cp
                if (slot >= 0) { const uint32_t *q = myRec + slot * 4; PkRec r2; r2.ab2 = q[0 ^ swz]; r2.m = q[1 ^ swz]; r2.a = q[2 ^ swz]; r2.b = q[3 ^ swz]; return r2; }
                return pkLoadRec(cp);
            };
            const int h = x >= YD_NP ? 1 : 0, k = x - YD_NP * h;
            const int second = k < 8 ? 0 : 1, bt = k < 8 ? k : k - 3, sT = 8 * h + bt, sU = 16 + sT, sM = 16 * h + 10 - k;
            int lim = u.rr + 1; lim = lim < y ? lim : y;                      // to the start of the 128-byte block: a line is fetched once, by one batch
            // The straight run of this block, all rows at once: bit d of `diag` = the cell d rows up in this column took the diagonal (notT and notU), bit d of `mis` =
            // it is a mismatch.  The run's length is the number of trailing ones of diag below lim; its ops are the groups of equal bits of mis.  (Row by row this was
            // ~25 instructions a row with a branch each; a run of eight rows is the common case: reads differ from the reference every sixty bases.)
            uint32_t diag = 0u, mis = 0u;
            const int iTU = (second ? 0 : 2) ^ swz, iM = 1 ^ swz;
#pragma unroll
            for (int d = 0; d < YD_TRACE_DEPTH; d++) {
                const int sl = rr0 - d < 0 ? 0 : rr0 - d;
                const uint32_t tu = myRec[sl * 4 + iTU], mw = myRec[sl * 4 + iM];
                diag |= (((tu >> sT) & (tu >> sU)) & 1u) << d; mis |= ((mw >> sM) & 1u) << d;
            }
            const uint32_t inLim = (1u << lim) - 1u;
            const int took = __builtin_ctz(~diag | ~inLim);                  // rows of the run (0 .. lim)
            int op = 0;
            // what stops the run: a gap op (the row is inside the block: took < lim <= rr0 + 1)
            if (took < lim) { const int sl = rr0 - took; const uint32_t tu = myRec[sl * 4 + iTU]; op = ((tu >> sU) & 1u) == 0u ? OP_I : OP_D; }
            y -= took;
            if (took > 0) {                                                   // the cursor moves up took rows inside the block (the step out of it is stepUp's); before the
                const int mv = took < lim ? took : took - 1;                  // ops are staged: every row of the run has been decoded, its records are free
                u.cp -= 4 * mv; u.w -= 4 * mv; u.rr -= mv;
            }
            {
                const uint32_t m = mis & ((1u << took) - 1u); int pos = 0;
                while (pos < took) {
#ifdef YD_PROF
                    tpRounds++;                                                    // (per lane here; summed over the wave below: lane-rounds)
#endif
                    const uint32_t cur = (m >> pos) & 1u;
                    const uint32_t same = (cur ? ~m : m) >> pos;              // zeros where the following rows have the same bit
                    const int g = __builtin_ctz(same | (1u << (took - pos)));
                    const int code = cur ? OP_R : OP_M;
                    if (prev != code) { if (prev >= 0) flush(); prev = code; acc = g; } else acc += g;
                    pos += g;
                }
            }
#ifdef YD_PROF
            tpRows += (unsigned)took;
            if (took != lim) { if (op == OP_D) tpDel++; else tpIns++; }
#endif
            if (took == lim) { if (y > 0) stepUp(u); else u.w = row0w; continue; }
            if (op == OP_D) {                                                    // deletion run: the continue bits along the row, leftwards (the row's low half is one record up)
                int sl = rr0 - took;                                             // the cursor's slot in the fetched block
                PkRec rb = blkRec(sl, u.cp);
                int run = 1, xx = x, hh = h; Cur v = u;
                for (;;) {
                    if (!pkContE(rb, xx - YD_NP * hh, hh)) break;
                    xx--; if (xx < 0) break;
                    run++;
                    if (hh && xx < YD_NP) { hh = 0; stepUp(v); rb = blkRec(--sl, v.cp); }
                }
                put(OP_D, run); x -= run;
                if (h && x < YD_NP) stepUp(u);
            } else {                                                            // insertion run: the continue bits up and to the right
                int run = 1, yy = y, xx = x, hh = h; Cur v = u;
                int sl = rr0 - took;
                PkRec rb = blkRec(sl, v.cp);
                while (pkContF(rb, xx - YD_NP * hh, hh)) {
                    yy--; xx++; if (yy <= 0 || xx >= YD_LW) break;
                    run++;
                    if (!hh && xx >= YD_NP) hh = 1;                              // (y-1, 11) is in the record of (y, 10)
                    else { stepUp(v); rb = blkRec(--sl, v.cp); }
                }
                put(OP_I, run); y -= run; x += run;
                const int h2 = x >= YD_NP ? 1 : 0, steps = run - (h2 - h);
                if (y > 0) { for (int s = 0; s < steps; s++) stepUp(u); } else u.w = row0w;
            }
        }
#ifdef YD_PROF
        if (tpWalks) {
            const unsigned r4 = (unsigned)waveSumI((int)tpRounds), d4 = (unsigned)waveSumI((int)tpDel), i4 = (unsigned)waveSumI((int)tpIns), w4 = (unsigned)waveSumI((int)tpRows);
            if (lane == 0) { atomicAdd(&gTraceProf[0], 1ull); atomicAdd(&gTraceProf[1], (unsigned long long)tpPass); atomicAdd(&gTraceProf[2], (unsigned long long)tpAct);
                atomicAdd(&gTraceProf[3], (unsigned long long)r4); atomicAdd(&gTraceProf[5], (unsigned long long)d4); atomicAdd(&gTraceProf[6], (unsigned long long)i4);
                atomicAdd(&gTraceProf[7], (unsigned long long)w4); }
        }
        (void)tpFlushR;
#endif
        if (walk) {
        if (y <= 0 && x > leftR) put(OP_D, x - leftR);                           // row 0: deletions back to the origin (SW.cpp:905-935)
        if (prev >= 0) { const int wp = E - 1 - n; if (wp < 0) bad = true; else *S.at(wp) = opMake(prev, acc); n++; }
        if (bad || S.wild) { bad = true; atomicCAS(A.errFlag, 0, (int)YERR_TRACE); n = 0; }
        }
    }
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
            for (int k0 = 0; k0 < n; k0 += 8) {
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

