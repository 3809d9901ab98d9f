// ext_lanes16.h -- k_ext_rows with TWO problems per lane in packed 16-bit arithmetic (v_pk_*_i16).
//
// k_ext_rows (ext_lanes.h) is bound by vector-instruction issue: ~32 instructions per cell.  Here every 32-bit register holds the
// same quantity of two independent problems (low half: slot 0, high half: slot 1), so the adds, subtractions and maxima of the
// recurrence serve two cells at once; the decisions are taken from the sign bits of saturating differences instead of compares.
// Everything else is as in k_ext_rows: whole strip in registers, per-wave pool of pre-loaded problems with refill (per slot),
// LDS-staged 128-byte trace blocks, deferred stores.  Used when the scores fit: MS * (longest read) <= 15000 and
// RC + X + GO + 21*GE <= 4000 (sentinel -16000, all arithmetic saturating) and the run caps cannot bind; otherwise k_ext_rows.
//
// Trace format: 5 bits per cell (3 cells per halfword, 7 halfwords = 16 bytes per row): bit0 = E did NOT win over G, bit1 = F did
// NOT win, bit2 = E-run did NOT continue, bit3 = F-run did NOT continue, bit4 = mismatch.  k_ext_trace<true> decodes it.
#pragma once
#include "ext_lanes.h"

typedef short yd_s16x2 __attribute__((ext_vector_type(2)));
#define YD_LW16 (-16000)

__device__ __forceinline__ yd_s16x2 pkS(uint32_t v) { return __builtin_bit_cast(yd_s16x2, v); }
__device__ __forceinline__ uint32_t pkU(yd_s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pkAdd(uint32_t a, uint32_t b) { return pkU(__builtin_elementwise_add_sat(pkS(a), pkS(b))); }
__device__ __forceinline__ uint32_t pkSub(uint32_t a, uint32_t b) { return pkU(__builtin_elementwise_sub_sat(pkS(a), pkS(b))); }
__device__ __forceinline__ uint32_t pkMax(uint32_t a, uint32_t b) { return pkU(__builtin_elementwise_max(pkS(a), pkS(b))); }
// (inline assembly with register operands: written as vector expressions these two are scalarised into compares and selects)
__device__ __forceinline__ uint32_t pkMinU(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pkMad(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// 0xFFFF in every negative half.  (Not inline assembly with the literal 15: in a packed instruction an inline constant feeds the low half only.)
__device__ __forceinline__ uint32_t pkSignMask(uint32_t a) { return pkU(pkS(a) >> (yd_s16x2)(15)); }
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }                      // v_bfi_b32
__device__ __forceinline__ uint32_t pk2(int v) { return ((uint32_t)v & 0xFFFFu) * 0x10001u; }

struct Slot16 {
    int p, i, qLen, maxScore, maxi, maxj, qStep, qcNext, pendRes, pendScore, pendI, pendJ; unsigned pCells, pendRows, pendCells; uint32_t rOff; bool rev, pendFlush, done;
    YD_GLOBAL const uint8_t *q; YD_GLOBAL uint32_t *strip, *pendBlk;
};

template <bool SECOND>
__global__ void __launch_bounds__(256) k_ext_rows16(ExtArgs A)
{
    __shared__ uint32_t sBlk[2][32][256];      // per lane and slot: the current 8-row trace block, [dword][thread]
    if (!SECOND && A.clock && threadIdx.x == 0) atomicMin(&A.clock[0], (unsigned long long)wall_clock64());
    const int lane = laneId(), tid = (int)threadIdx.x;
    const int GO = A.P.GO, GE = A.P.GE, XC = A.P.X;
    constexpr int bandwidth = YD_LBAND, leftR = YD_LBAND;
    const uint32_t maxROff = A.P.maxROff;
    YD_GLOBAL const uint8_t *gBases = toGlobal(A.bases);
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    const unsigned nProbEff = A.nProb;
    const uint32_t GEp = pk2(GE), GOEp = pk2(GO + GE), MSp = pk2(A.P.MS), NEGK = pk2(-(A.P.MS + A.P.RC)), LWp = pk2(YD_LW16), ONEp = 0x00010001u;

    uint32_t PV[YD_LW], PF[YD_LW], rc[YD_LW];
#pragma unroll
    for (int j = 0; j < YD_LW; j++) { PV[j] = LWp; PF[j] = LWp; rc[j] = 0x000F000Fu; }
    Slot16 S[2];
#pragma unroll
    for (int s = 0; s < 2; s++) { S[s].p = -1; S[s].i = 0; S[s].qLen = 0; S[s].maxScore = YD_LW16; S[s].maxi = S[s].maxj = 0; S[s].qStep = 0; S[s].qcNext = 0; S[s].pendRes = -1; S[s].pendScore = S[s].pendI = S[s].pendJ = 0;
        S[s].pCells = S[s].pendRows = S[s].pendCells = 0; S[s].rOff = 0; S[s].rev = false; S[s].pendFlush = false; S[s].done = false; S[s].q = toGlobal(A.fwd); S[s].strip = toGlobal(A.trace); S[s].pendBlk = toGlobal(A.trace); }
    unsigned calls = 0, rows = 0, cells = 0;
    unsigned poolBase = 0; int poolCount = 0, poolNext = 0; bool exhausted = false, firstFill = true;
    uint32_t eLens = 0, eROff = 0, eQ = 0, eMisc = 0, eW1 = 0, eW2 = 0, eSLo = 0, eSHi = 0, ePidx = 0;

    for (;;) {
        // ---- refill (see k_ext_rows): the idle slots of the wave, slot 0 lanes first ----
        for (;;) {
            const unsigned long long need0 = __ballot(S[0].p < 0 && !S[0].done), need1 = __ballot(S[1].p < 0 && !S[1].done);
            const int n0 = __builtin_popcountll(need0), n1 = __builtin_popcountll(need1);
            if (n0 + n1 == 0) break;
            if (n0 + n1 < 2 * YD_REFILL_MIN && __ballot(S[0].p >= 0 || S[1].p >= 0) != 0ull && !firstFill) break;
            if (poolNext >= poolCount) {
                unsigned base = 0;
                if (!exhausted) { if (lane == 0) base = atomicAdd(A.queue, 64u); base = uniU(base); if (base >= nProbEff) exhausted = true; }
                if (exhausted) { if (S[0].p < 0) S[0].done = true; if (S[1].p < 0) S[1].done = true; break; }
                poolBase = base; poolCount = (int)min(64u, nProbEff - base); poolNext = 0;
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
                        eW1 = 0; eW2 = 0;                                     // reference window of row 1: register column c holds reference index c - leftR
                        for (int c = leftR; c < YD_LW; c++) {
                            const int idx = c - leftR; uint32_t nib = 15u;
                            if (idx < (int)rl) { const uint32_t off = rv_ ? pr.rOff - (uint32_t)idx : pr.rOff + (uint32_t)idx; const uint32_t b = gBases[off >> 1]; nib = (off & 1u) ? (b & 15u) : (b >> 4); }
                            const uint32_t sh = (uint32_t)(c & 7) * 4u;
                            if (c < 16) eW1 |= nib << sh; else eW2 |= nib << sh;
                        }
                    }
                }
            }
            const int avail = poolCount - poolNext;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const unsigned long long need = s == 0 ? need0 : need1;
                const int e = poolNext + (s == 0 ? 0 : n0) + __builtin_popcountll(need & lanesBelow);
                const bool take = (S[s].p < 0 && !S[s].done) && e < poolCount;
                const int src = take ? e : lane;
                const uint32_t gLens = (uint32_t)__shfl((int)eLens, src, 64), gROff = (uint32_t)__shfl((int)eROff, src, 64), gQ = (uint32_t)__shfl((int)eQ, src, 64), gMisc = (uint32_t)__shfl((int)eMisc, src, 64);
                const uint32_t gW1 = (uint32_t)__shfl((int)eW1, src, 64), gW2 = (uint32_t)__shfl((int)eW2, src, 64), gSLo = (uint32_t)__shfl((int)eSLo, src, 64), gSHi = (uint32_t)__shfl((int)eSHi, src, 64);
                const uint32_t gPidx = (uint32_t)__shfl((int)ePidx, src, 64);
                const bool init = take && gLens != 0u;
                const uint32_t hm = init ? (s == 0 ? 0x0000FFFFu : 0xFFFF0000u) : 0u;       // this slot's half of every state register
#pragma unroll
                for (int j = 0; j < YD_LW; j++) {                            // row 0 of the strip (SW.cpp:905-935; PF(0, left) = -GO, see ext_lanes.h)
                    const int iV = j == leftR ? 0 : (j > leftR ? -(GO + (j - leftR) * GE) : YD_LW16), iF = j == leftR ? -GO : YD_LW16;
                    PV[j] = bfi(hm, pk2(iV), PV[j]); PF[j] = bfi(hm, pk2(iF), PF[j]);
                    const uint32_t nib = j < leftR ? 0x800Fu : ((j < 16 ? gW1 : gW2) >> ((j & 7) * 4)) & 15u;      // bit 15: not a cell of the matrix
                    rc[j] = bfi(hm, nib * 0x10001u, rc[j]);
                }
                if (init) {
                    S[s].p = (int)gPidx; S[s].qLen = (int)(gLens & 0xFFFFu); S[s].i = 0; S[s].maxScore = YD_LW16; S[s].maxi = 0; S[s].maxj = 0;
                    S[s].rev = (gMisc & XP_REV) != 0; S[s].rOff = gROff; S[s].pCells = 0;
                    S[s].q = toGlobal((gMisc & XP_STRAND) ? A.rev : A.fwd) + gQ; S[s].qStep = S[s].rev ? -1 : 1; S[s].qcNext = (int)((gMisc >> 8) & 0xFFu);
                    S[s].strip = toGlobal(A.trace) + (((unsigned long long)gSHi << 32) | gSLo) * 32ull;
                }
            }
            poolNext += (n0 + n1) < avail ? (n0 + n1) : avail;
        }
        firstFill = false;
        if (__ballot(S[0].p >= 0 || S[1].p >= 0) == 0ull) break;

        // ---- one DP row in both slots of every lane ----
        uint32_t qcP = 0, nbByte[2], nbOdd[2]; bool busy[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (S[s].pendFlush) {                                            // deferred: a finished 8-row block, LDS -> one 128-byte line
#pragma unroll
                for (int d = 0; d < 32; d += 4) { yd_u32x4 v; v.x = sBlk[s][d][tid]; v.y = sBlk[s][d + 1][tid]; v.z = sBlk[s][d + 2][tid]; v.w = sBlk[s][d + 3][tid]; *(YD_GLOBAL yd_u32x4 *)(S[s].pendBlk + d) = v; }
                S[s].pendFlush = false;
            }
            if (S[s].pendRes >= 0) {
                ExtRes r; r.score = S[s].pendScore > 0 ? S[s].pendScore : 0; r.maxi = S[s].pendI; r.maxj = S[s].pendJ; r.opsOff = 0; r.nOps = 0; r.rLen = 0; r.rows = S[s].pendRows; r.cells = S[s].pendCells;
                A.res[S[s].pendRes] = r; S[s].pendRes = -1;
            }
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            busy[s] = S[s].p >= 0;
            const int i = ++S[s].i;
            const int qc = S[s].qcNext;
            { const int ni = i < S[s].qLen ? i : (S[s].qLen > 0 ? S[s].qLen - 1 : 0); S[s].qcNext = (int)S[s].q[ni * S[s].qStep]; }
            { const int idx = i + bandwidth; const bool in = busy[s] && idx < S[s].qLen + bandwidth; const uint32_t off = in ? (S[s].rev ? S[s].rOff - (uint32_t)idx : S[s].rOff + (uint32_t)idx) : 0u;
              nbByte[s] = gBases[off >> 1]; nbOdd[s] = in ? (off & 1u) : 2u; }      // unconditional load, decoded after the cells (no wait here)
            qcP |= (uint32_t)qc << (16 * s);
            int sc = leftR + 1 - i; if (sc < 0) sc = 0;
            if (busy[s]) { const unsigned nc = (unsigned)(YD_LW - sc); rows++; cells += nc; S[s].pCells += nc; }
        }
        uint32_t PVCol = LWp, PE = LWp, rowMax = LWp, jbest = 0, dV = PV[0];
        uint32_t acc[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < YD_LW; j++) {
            const uint32_t mm = pkMinU(rc[j] ^ qcP, ONEp);                   // 0 = match, 1 = mismatch, per slot
            uint32_t V = pkAdd(dV, pkMad(mm, NEGK, MSp));                     // G = diagonal + (MS | -RC)
            const uint32_t CE = pkSub(PE, GEp), NE = pkSub(PVCol, GOEp);
            PE = pkMax(CE, NE);
            const uint32_t dE = pkSub(CE, NE);                               // >= 0: the E run continues (ties continue, SW.cpp:1029-1033)
            const uint32_t dT = pkSub(PE, V);                                // >= 0: E wins over G ('>=' in extension mode, SW.cpp:1036)
            V = pkMax(V, PE);
            const uint32_t upV = j + 1 < YD_LW ? PV[j + 1] : LWp, upF = j + 1 < YD_LW ? PF[j + 1] : LWp;
            const uint32_t CF = pkSub(upF, GEp), NF = pkSub(upV, GOEp);
            const uint32_t F = pkMax(CF, NF);
            const uint32_t dF = pkSub(CF, NF);
            const uint32_t dU = pkSub(F, V);
            V = pkMax(V, F);
            // 5-bit trace cell of both slots from the sign bits (set = "did not")
            uint32_t f = (dT >> 15) & 0x00010001u;
            f = ((dU >> 14) & 0x00020002u) | f;
            f = ((dE >> 13) & 0x00040004u) | f;
            f = ((dF >> 12) & 0x00080008u) | f;
            f = (mm << 4) | f;
            acc[j / 3] = (acc[j / 3] << 5) | f;
            asm volatile("" : "+v"(acc[j / 3]));                             // pinned (see k_ext_rows)
            // row-major first maximum over the real cells: the columns left of the origin (one of them is the boundary column, with a
            // real-sized value) carry bit 15 in their reference code, which has slid along with the window
            const uint32_t Vm = j < leftR ? bfi(pkSignMask(rc[j]), LWp, V) : V;
            const uint32_t gt = pkSignMask(pkSub(rowMax, Vm));               // 0xFFFF where this cell beats the row's maximum so far
            jbest = bfi(gt, (uint32_t)j * 0x10001u, jbest);
            rowMax = pkMax(rowMax, Vm);
            PV[j] = V; PF[j] = F; PVCol = V; dV = upV;
            __builtin_amdgcn_sched_barrier(0);
        }
        // slide the reference windows: column c takes column c+1, the top column takes the new bases
#pragma unroll
        for (int j = 0; j + 1 < YD_LW; j++) rc[j] = rc[j + 1];
        { const uint32_t nb0 = nbOdd[0] == 2u ? 15u : (nbOdd[0] ? (nbByte[0] & 15u) : (nbByte[0] >> 4)), nb1 = nbOdd[1] == 2u ? 15u : (nbOdd[1] ? (nbByte[1] & 15u) : (nbByte[1] >> 4));
          rc[YD_LW - 1] = nb0 | (nb1 << 16); }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int i = S[s].i;
            {   // this row's cells: halfwords 0..6 of the slot -> 4 dwords of its LDS block
                const int slot = ((i - 1) & 7) * 4;
                auto half = [&](int k) { return s == 0 ? (acc[k] & 0xFFFFu) : (acc[k] >> 16); };
                sBlk[s][slot][tid] = half(0) | (half(1) << 16); sBlk[s][slot + 1][tid] = half(2) | (half(3) << 16); sBlk[s][slot + 2][tid] = half(4) | (half(5) << 16); sBlk[s][slot + 3][tid] = half(6);
            }
            const int rv = (int)(short)(s == 0 ? (rowMax & 0xFFFFu) : (rowMax >> 16)), rj = (int)(s == 0 ? (jbest & 0xFFFFu) : (jbest >> 16));
            if (rv > S[s].maxScore) { S[s].maxScore = rv; S[s].maxi = i; S[s].maxj = rj; }
            const bool fin = busy[s] && (rv < S[s].maxScore - XC || i >= S[s].qLen);
            if (busy[s] && (fin || (i & 7) == 0)) { S[s].pendFlush = true; S[s].pendBlk = S[s].strip + (size_t)((i - 1) >> 3) * 32u; }
            if (fin) {
                S[s].pendRes = S[s].p; S[s].pendScore = S[s].maxScore; S[s].pendI = S[s].maxi; S[s].pendJ = S[s].maxj; S[s].pendRows = (unsigned)i; S[s].pendCells = S[s].pCells;
                S[s].p = -1; S[s].qStep = 0; S[s].qLen = 0; S[s].i = 0;
            }
        }
    }
    // the last deferred stores
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (S[s].pendFlush) { for (int d = 0; d < 32; d += 4) { yd_u32x4 v; v.x = sBlk[s][d][tid]; v.y = sBlk[s][d + 1][tid]; v.z = sBlk[s][d + 2][tid]; v.w = sBlk[s][d + 3][tid]; *(YD_GLOBAL yd_u32x4 *)(S[s].pendBlk + d) = v; } }
        if (S[s].pendRes >= 0) {
            ExtRes r; r.score = S[s].pendScore > 0 ? S[s].pendScore : 0; r.maxi = S[s].pendI; r.maxj = S[s].pendJ; r.opsOff = 0; r.nOps = 0; r.rLen = 0; r.rows = S[s].pendRows; r.cells = S[s].pendCells;
            A.res[S[s].pendRes] = r;
        }
    }
    if (!SECOND && A.clock && lane == 0) atomicMax(&A.clock[1], (unsigned long long)wall_clock64());
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
