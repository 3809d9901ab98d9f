// gap_band_pk.h -- k_gap_band (gap_band_lanes.h) in PACKED 16-BIT arithmetic: the banded gap fills (findAGSAlignmentBanded -> findAffineGapScore<banded, global>,
// SW.cpp:470-475, 798-1208), one joint per lane, the strip in registers and SKEWED over the two halves of each register exactly as k_ext_rows_pk's (ext_lanes_pk.h):
//
//     register pair k (k = 0 .. H-1, H = GW / 2):   low half = column k of row t       high half = column k + H of row t - 1        (iteration t of the joint)
//
// Inside an iteration the pairs go left to right.  For both halves the diagonal neighbour is the pair's own old value and the upper neighbour the next pair's old
// value; the halves meet in two places: the left neighbour of column H (pair 0, high) is column H-1 of the same row t-1 -- pair H-1's low half as the PREVIOUS
// iteration left it (carried: value, E state, E run) -- and the upper neighbour of column H-1 (pair H-1, low) is column H of row t-1 -- pair 0's high half as THIS
// iteration has just computed it.  Column GW does not exist: the sentinel.  A joint of qGap rows takes qGap + 1 iterations.
//
// What differs from the extension kernel: the global mode's tie rules ('>' for E and F against the cell, SW.cpp:1036,1054: a sign mask of cell - E instead of
// E - cell), run lengths (a trace cell is op | run << 2 in one byte, as k_gap_band's, so the traceback needs no continue bits), the run caps (maxIntron on E, maxGap
// on F: compiled out for a wave none of whose joints can reach them), and row 0, which is not produced by the recurrence when the origin lies in the high half:
// the first iteration computes row 1 in the low halves and has its high halves REPLACED by the explicit row 0 (SW.cpp:905-935) as each pair is finished.
// Trace: one record of GW bytes an iteration -- bytes 0 .. H-1 the low halves' cells (row t), bytes H .. GW-1 the high halves' (row t-1): cell (y, x) is byte x of
// record y + (x >= H).  Scores: |V| <= 64 max(MS, RC) + GO + 64 GE, sentinel -16000, all arithmetic saturating (the host checks the range: gapBandPacked()).
// 29 vector instructions a pair = 14.5 a cell where k_gap_band has 32 (round 6: the gap fills were 1.0 G of the step's vector instructions for 6.5 % of its DP cells).
#pragma once
#include "gap_band_lanes.h"
#include "ext_lanes_pk.h"

// rows of one joint (iterations 1 .. qGap + 1).  CAPS: the run caps can bind for some joint of the wave.  All state by reference: the caller owns it.
template <int GW, bool CAPS, class QAt, class RefAt>
__device__ __forceinline__ int gapBandPkRows(const DevParams &P, int qGap, int left, int right, int W, QAt qAt, RefAt refAt, YD_GLOBAL uint32_t *T32)
{
    constexpr int H = GW / 2, RD = GW / 4;
    const int GO = P.GO, GE = P.GE, GOE = GO + GE;
    const uint32_t GEp = pk2(GE), GOEp = pk2(GOE), LWp = pk2(YD_LW16), LWg = pk2(YD_LW16 - GOE), ONEp = 0x00010001u;
    uint32_t NEGKv = pk2(-(P.MS + P.RC)), MSGv = pk2(P.MS + GOE), ONEv = ONEp, C15v = 0x000F000Fu, FOURv = 0x00040004u, OPDv = pk2(OP_D), OPIv = pk2(OP_I);
    asm volatile("" : "+v"(NEGKv), "+v"(MSGv), "+v"(ONEv), "+v"(C15v), "+v"(FOURv), "+v"(OPDv), "+v"(OPIv));      // operands of the inline-assembly instructions: kept in VGPRs
    const uint32_t maxIp = pk2(P.maxIntron < 30000 ? P.maxIntron : 30000), maxGp = pk2(P.maxGap < 30000 ? P.maxGap : 30000);
    // explicit row 0 (SW.cpp:905-935): the origin at column `left` (V = 0, F = -GO, cell U), deletions to its right inside the band, the sentinel elsewhere
    auto onChain = [&](int c) { return c > left && c < W; };
    auto v0 = [&](int c) -> int { return c == left ? 0 : (onChain(c) ? -(GO + (c - left) * GE) : YD_LW16); };
    auto f0 = [&](int c) -> int { return c == left ? -GO : YD_LW16; };
    auto e0 = [&](int c) -> int { return onChain(c) ? -(GO + (c - left) * GE) : YD_LW16; };                       // (on the chain E == V)
    auto d0 = [&](int c) -> int { return onChain(c) ? c - left : 0; };
    auto cell0 = [&](int c) -> uint32_t { return c == left ? (uint32_t)TR_U8 : (onChain(c) ? (uint32_t)(OP_D | ((c - left) << 2)) : 0u); };
    auto lo16 = [](int v) -> uint32_t { return (uint32_t)v & 0xFFFFu; };
    uint32_t PVg[H], PF[H], PI[H], rc[H], keep[H];
#pragma unroll
    for (int k = 0; k < H; k++) {
        PVg[k] = lo16(v0(k) - GOE) | (LWg & 0xFFFF0000u); PF[k] = lo16(f0(k)) | (LWp & 0xFFFF0000u); PI[k] = 0u;
        // reference codes: low = (row 1, column k) -> index k - left; high = (row 0, column k + H) -> index k + H - 1 - left (it slides into place, see below)
        rc[k] = refAt(k - left) | (refAt(k + H - 1 - left) << 16);
        keep[k] = (k + H < 11 || k + H < W) ? 0xFFFFFFFFu : 0x0000FFFFu;                                        // columns >= W (>= 11 at the default band) stay at the sentinel
    }
    // row 0's cells of columns 0 .. H-1: the low bytes of record 0 (its high bytes belong to a row that does not exist)
#pragma unroll
    for (int d = 0; d < RD; d++) {
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) { const int c = 4 * d + b; if (c < H) acc |= cell0(c) << (8 * b); }
        T32[d * 64] = acc;
    }
    uint32_t carryV = LWg, carryE = LWp, carryD = 0u;                                                           // column H-1 of the row the high halves are about to compute
    int qcHi = 0, scoreLo = 0;
    for (int t = 1; t <= qGap + 1; t++) {
        const int qcLo = qAt(t <= qGap ? t - 1 : qGap - 1);
        const uint32_t qcP = (uint32_t)qcLo | ((uint32_t)qcHi << 16);
        uint32_t qcPv = qcP; asm volatile("" : "+v"(qcPv));
        uint32_t PVCol = (carryV << 16) | (LWg & 0xFFFFu), PE = (carryE << 16) | (LWp & 0xFFFFu), PD = carryD << 16;
        uint32_t cells[H];
        const bool first = t == 1;
#pragma unroll
        for (int k = 0; k < H; k++) {
            const uint32_t mm = pkMinU(rc[k] ^ qcPv, ONEv);                                                    // 0 = match, 1 = mismatch = OP_M / OP_R
            // the diagonal's Vg (the pair's own old value) + (MS + GOE | -RC + GOE)
            const uint32_t G = pkAdd(PVg[k], pkMad(mm, NEGKv, MSGv));
            // E: the deletion run along the row
            const uint32_t CE = pkSub(PE, GEp), NE = PVCol;
            const uint32_t PD1 = pkAdd(PD, ONEp);
            uint32_t notE = pkSignMaskAsm(pkSub(CE, NE), C15v);                                                 // set: the run does NOT continue ('>=' continues, SW.cpp:1029-1033)
            if (CAPS) notE |= pkSignMaskAsm(pkSub(maxIp, PD1), C15v);
            PE = CAPS ? bfi(notE, NE, CE) : pkMax(CE, NE);
            PD = bfi(notE, ONEp, PD1);
            uint32_t V = pkMax(G, PE), cell = bfi(pkSignMaskAsm(pkSub(G, PE), C15v), pkMad(PD, FOURv, OPDv), mm);   // E wins only when greater (SW.cpp:1036, global mode)
            // F: the insertion run from the row above
            uint32_t upV, upF, upI;
            if (k + 1 < H) { upV = PVg[k + 1]; upF = PF[k + 1]; upI = PI[k + 1]; }
            else { upV = (PVg[0] >> 16) | (LWg & 0xFFFF0000u); upF = (PF[0] >> 16) | (LWp & 0xFFFF0000u); upI = PI[0] >> 16; }      // column H of the row above: just computed
            const uint32_t CF = pkSub(upF, GEp), NF = upV, I1 = pkAdd(upI, ONEp);
            uint32_t notF = pkSignMaskAsm(pkSub(CF, NF), C15v);
            if (CAPS) notF |= pkSignMaskAsm(pkSub(maxGp, I1), C15v);
            uint32_t F = CAPS ? bfi(notF, NF, CF) : pkMax(CF, NF);
            uint32_t I = bfi(notF, ONEp, I1);
            cell = bfi(pkSignMaskAsm(pkSub(V, F), C15v), pkMad(I, FOURv, OPIv), cell);                           // F wins only when greater (SW.cpp:1054)
            V = pkMax(V, F);
            uint32_t Vg = pkSub(V, GOEp);
            if (first) {                                                                                        // the high halves of the first iteration ARE row 0
                const int c = k + H;
                Vg = (Vg & 0xFFFFu) | (lo16(v0(c) - GOE) << 16); F = (F & 0xFFFFu) | (lo16(f0(c)) << 16); I &= 0xFFFFu;
                PE = (PE & 0xFFFFu) | (lo16(e0(c)) << 16); PD = (PD & 0xFFFFu) | ((uint32_t)d0(c) << 16); cell = (cell & 0xFFFFu) | (cell0(c) << 16);
            }
            if (k + H >= 11) { Vg = bfi(keep[k], Vg, LWg); F = bfi(keep[k], F, LWp); I &= keep[k]; }
            PVg[k] = Vg; PF[k] = F; PI[k] = I; PVCol = Vg; cells[k] = cell;
        }
        // the iteration's record: [low cells of pairs 0 .. H-1][high cells of pairs 0 .. H-1], a byte each
        {
            uint32_t X[H / 2];                                                                                  // pairs 2 j, 2 j + 1 -> bytes (lo, lo, hi, hi)
#pragma unroll
            for (int j = 0; j < H / 2; j++) X[j] = __builtin_amdgcn_perm(cells[2 * j + 1], cells[2 * j], 0x06020400u);
            uint32_t D[RD];
            if constexpr (GW == 12) {
                D[0] = __builtin_amdgcn_perm(X[1], X[0], 0x05040100u);                                          // lo 0 1 2 3
                D[1] = __builtin_amdgcn_perm(X[0], X[2], 0x07060100u);                                          // lo 4 5, hi 0 1
                D[2] = __builtin_amdgcn_perm(X[2], X[1], 0x07060302u);                                          // hi 2 3 4 5
            // (H a multiple of four: the low cells' dwords, then the high cells')
            } else {
#pragma unroll
                for (int d = 0; d < RD / 2; d++) {
                    D[d] = __builtin_amdgcn_perm(X[2 * d + 1], X[2 * d], 0x05040100u); D[RD / 2 + d] = __builtin_amdgcn_perm(X[2 * d + 1], X[2 * d], 0x07060302u);
                }
            }
#pragma unroll
            for (int d = 0; d < RD; d++) T32[(t * RD + d) * 64] = D[d];
        }
        // column H-1 of the row just finished in the low halves: what pair 0's high half starts from in the next iteration
        carryV = PVg[H - 1] & 0xFFFFu; carryE = PE & 0xFFFFu; carryD = PD & 0xFFFFu;
        if (t == qGap && right < H) {                                                                           // the end cell (qGap, right) in a low half
            scoreLo = 0;
#pragma unroll
            for (int k = 0; k < H; k++) if (k == right) scoreLo = (int)(short)(PVg[k] & 0xFFFFu) + GOE;
        }
        // slide the reference codes one row on: pair k takes pair k + 1's; pair H-1's low half what was pair 1's high half, its high half the new base
        if (t <= qGap) {                                                                                        // (nothing beyond the staged bases is asked for)
            const uint32_t nb = refAt(t + GW - 2 - left), was1 = rc[1] >> 16;
#pragma unroll
            for (int k = 0; k + 1 < H; k++) rc[k] = rc[k + 1];
            rc[H - 1] = was1 | (nb << 16);
        }
        qcHi = qcLo;
    }
    int score = scoreLo;
    if (right >= H) {
#pragma unroll
        for (int k = 0; k < H; k++) if (k + H == right) score = (int)(short)(PVg[k] >> 16) + GOE;
    }
    return score;
}

template <int GW>
__global__ void __launch_bounds__(64) k_gap_band_pk(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    static_assert(GW == 12 || GW == 16 || GW == 24, "three instances");
    constexpr int H = GW / 2;
    const int lane = laneId(); const DevParams &P = A.P;
    const int bw = P.bandWidth;
    uint32_t *sp = (uint32_t *)(X.gapScratch + (size_t)blockIdx.x * 64u * YD_GAP_SCRATCH) + lane;
    YD_GLOBAL uint32_t *T32 = toGlobal(sp); uint32_t *tmp = sp + (size_t)((YD_GROWS + 1) * 32 / 4) * 64;
    YD_GLOBAL const uint8_t *gB = toGlobal(A.bases);
    // GW = 12, 16: the joints of class 0 / 1 (gapJointKey: banded, W <= 12 / 16, within the limits).  GW = 24: the joints of the LAST class -- everything wider than 16 --
    // that are banded with W <= 24 inside the same limits: the order key puts them first in their class (96 % of it on the bench batch; k_gap_lanes<32> ran them at 60
    // instructions a cell with its strip in LDS), X.nDPb[2] counts them.
    const uint32_t tBegin = GW == 12 ? 0u : (GW == 16 ? X.nDPb[0] : X.nDP[1]), tEnd = GW == 12 ? X.nDPb[0] : (GW == 16 ? X.nDPb[1] : X.nDP[1] + X.nDPb[2]);
    // the joint's query codes and reference bytes, [dword][lane] (k_gap_band)
    __shared__ uint32_t sQ[16 * 64], sR[13 * 64];
    typedef uint32_t yd_u32u __attribute__((aligned(1)));
    for (uint32_t base = tBegin + blockIdx.x * 64u; base < tEnd; base += gridDim.x * 64u) {
        const uint32_t t = base + (uint32_t)lane; const bool live = t < tEnd;
        int nT = 0, score = 0; unsigned cells = 0; uint32_t ji = 0;
        JointRec j; j.qGap = 0; j.rGap = 0; j.nsro = 0; j.flags = 0; j.qBase = 0; j.nsqo = 0;
        if (live) { ji = X.sortedVals[t]; j = X.joints[ji]; }

        // (wave-uniform: can a run cap bind for any joint of this wave?  an E run spans at most GW - 1 columns, an F run at most qGap rows)
        const bool caps = __ballot(live && ((int)j.qGap > P.maxGap || P.maxIntron < GW)) != 0ull;
        if (live) {
            const int qGap = j.qGap, rGap = j.rGap;
            int left, right; if (rGap > qGap) { right = bw + (rGap - qGap); left = bw; } else { left = bw + (qGap - rGap); right = bw; }
            const int W = left + right + 1;
            YD_GLOBAL const uint8_t *q = toGlobal((j.flags & 1u) ? A.B.rev : A.B.fwd) + j.qBase + j.nsqo;
            const uint32_t rB0 = ((j.nsro >= (uint32_t)left ? j.nsro - (uint32_t)left : 0u) >> 1) & ~3u;      // first staged byte of the reference (dword-aligned)
            {
                const int nQ = (qGap + 3) >> 2, nR = (int)((((j.nsro + (uint32_t)(qGap - left + GW - 1)) >> 1) - rB0) >> 2) + 1;
#pragma unroll
                for (int k = 0; k < 16; k++) if (k < nQ) sQ[k * 64 + lane] = *(YD_GLOBAL const yd_u32u *)(q + 4 * k);
#pragma unroll
                for (int k = 0; k < 13; k++) if (k < nR) sR[k * 64 + lane] = *(YD_GLOBAL const uint32_t *)(gB + rB0 + 4u * (uint32_t)k);
            }
            auto qAt = [&](int idx) -> int { return (int)((sQ[(idx >> 2) * 64 + lane] >> (8 * (idx & 3))) & 0xFFu); };
            auto refAt = [&](int idx) -> uint32_t {
                if (idx < 0 && (uint32_t)(-idx) > j.nsro) return 15u;
                const uint32_t off = j.nsro + (uint32_t)idx, rel = (off >> 1) - rB0; const uint32_t b = (sR[(rel >> 2) * 64 + lane] >> (8u * (rel & 3u))) & 0xFFu;
                return (off & 1u) ? (b & 15u) : (b >> 4);
            };
            for (int i = 1; i <= qGap; i++) { int sc = left + 1 - i; if (sc < 0) sc = 0; int ec = left + rGap - i; if (ec > W - 1) ec = W - 1;
                if (ec >= sc) cells += (unsigned)(ec - sc + 1); }
            score = caps ? gapBandPkRows<GW, true>(P, qGap, left, right, W, qAt, refAt, T32) : gapBandPkRows<GW, false>(P, qGap, left, right, W, qAt, refAt, T32);
            // traceback from the end cell (SW.cpp:1138-1195), as k_gap_band's: cell (y, x) is byte x of record y + (x >= H)
            int x = right, y = qGap;
            auto cellAt = [&](int yy, int xx) -> unsigned { const int c = (yy + (xx >= H ? 1 : 0)) * GW + xx; return (T32[(c >> 2) * 64] >> (8 * (c & 3))) & 0xFFu; };
            unsigned cell = cellAt(y, x);
            int prev = cell == TR_U8 ? -1 : (int)(cell & 3u), acc2 = 0, n = 0;
            for (int guard = 0; cell != TR_U8 && guard < 4096; guard++) {
                const int code = (int)(cell & 3u); int len = (int)(cell >> 2);
                if (code == OP_D) x -= len; else if (code == OP_I) { x += len; y -= len; } else { y -= 1; len = 1; }
                if (prev != code) { tmp[n * 64] = opMake(prev, acc2); n++; prev = code; acc2 = len; } else acc2 += len;
                if (y < 0 || x < 0 || x >= GW || n >= 190) break;
                cell = cellAt(y, x);
            }
            tmp[n * 64] = opMake(prev, acc2); n++;
            nT = n;
        }
        // op slots: wave prefix sum of nT (as k_gap_band)
        int incl = nT;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
        const int total = __shfl(incl, 63, 64); unsigned ob = 0;
        if (lane == 63 && total) ob = atomicAdd(X.gapOpsCount, (unsigned)total);
        ob = (unsigned)__shfl((int)ob, 63, 64);
        if (live) {
            const unsigned off = ob + (unsigned)(incl - nT);
            if ((unsigned long long)off + (unsigned)nT > (unsigned long long)X.gapOpsCap) atomicCAS(A.errFlag, 0, (int)YERR_OUT);
            else {
                for (int k = 0; k < nT; k++) X.gapOps[off + k] = tmp[(nT - 1 - k) * 64];    // list order
                JointRec *jp = X.joints + ji; jp->opsOff = off; jp->nOps = (uint16_t)nT; jp->score = score; jp->cells = cells;
            }
        }
    }
}
