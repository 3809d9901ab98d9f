// gap_band_lanes.h -- banded gap fills (findAGSAlignmentBanded -> findAffineGapScore<banded, global>, SW.cpp:470-475, 798-1208) with the strip IN REGISTERS,
// one joint per lane: the gap-fill counterpart of k_ext_rows (ext_lanes.h).
//
// 85 % of the gap-fill DP calls are the plain 11-wide band (W = 2*BW + 1 + |qGap - rGap|), ~30 rows.  k_gap_lanes keeps the strip state in LDS because W varies
// from lane to lane: 4 LDS reads and 3 writes per cell, ~60 instructions.  Here the strip is GW register columns (GW = 12 or 16, two instances; the joints are
// sorted by class and shape, so the lanes of a wave run the same instance on similar problems), every row computes all GW columns with static register
// indices, and what the reference special-cases falls out of the recurrence itself, exactly as in k_ext_rows:
//   * the boundary insertions V(i, left - i) = -(GO + i*GE) (SW.cpp:937-949): with PF(0, left) = -GO the ordinary F recurrence produces that chain (op I, run i)
//     -- the cap maxGap cannot bind on it because i <= left <= BW + 10 here -- and everything left of it stays at the sentinel;
//   * the right edge endCol = min(left + rLen - i, W - 1): cells beyond it are computed but never read by a real cell (a real cell's "up" neighbour is one
//     column further right in the previous row, where the edge was one column further right as well);
//   * columns >= W (W < GW) are held at the sentinel, so that column W - 1 sees the reference's "no cell above to the right".
// Tie rules of the global mode ('>' for E and F, SW.cpp:1036,1054), run caps (maxIntron on E, maxGap on F), one-byte trace cells (op | run << 2) in the same
// lane-private strip layout and the same traceback as gapDPLane (phase_lanes.h).  Limits: W <= GW, qGap <= 60, rGap <= 64 (class 0 / 1 of gapJointClass).
#pragma once
#include "phase_lanes.h"

template <int GW>
__global__ void __launch_bounds__(64) k_gap_band(AlignArgs A, PhaseArgs X)
{
    YD_HIGH_PRIO();
    static_assert(GW == 12 || GW == 16, "two instances");
    const int lane = laneId(); const DevParams &P = A.P;
    const int GO = P.GO, GE = P.GE, GOE = P.GO + P.GE, RC = P.RC, MS = P.MS, maxIntron = P.maxIntron, maxGapP = P.maxGap, bw = P.bandWidth;
    uint32_t *sp = (uint32_t *)(X.gapScratch + (size_t)blockIdx.x * 64u * YD_GAP_SCRATCH) + lane;
    YD_GLOBAL uint32_t *T32 = toGlobal(sp); uint32_t *tmp = sp + (size_t)((YD_GROWS + 1) * 32 / 4) * 64;
    YD_GLOBAL const uint8_t *gB = toGlobal(A.bases);
    const uint32_t tBegin = GW == 12 ? 0u : X.nDPb[0], tEnd = GW == 12 ? X.nDPb[0] : X.nDPb[1];
    constexpr int RD = GW / 4;                                              // trace dwords per row
    // The joint's query codes (at most 60) and reference nibbles (indices -left .. qGap - left + GW - 1: at most 91) are staged in LDS once, [dword][lane], by
    // whole dwords issued together.  A byte load per row for each of them was a memory round trip per row: the loads of a row return behind the trace stores
    // of the row before (one counter, in order).
    __shared__ uint32_t sQ[16 * 64], sR[13 * 64];
    typedef uint32_t yd_u32u __attribute__((aligned(1)));
    for (uint32_t base = tBegin + blockIdx.x * 64u; base < tEnd; base += gridDim.x * 64u) {
        const uint32_t t = base + (uint32_t)lane; const bool live = t < tEnd;
        int nT = 0, score = 0; unsigned cells = 0; uint32_t ji = 0;
        if (live) {
            ji = X.sortedVals[t]; const JointRec j = X.joints[ji];
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
            int PV[GW], PF[GW], PI[GW];
#pragma unroll
            for (int c = 0; c < GW; c++) { PV[c] = c == left ? 0 : ((c > left && c < W) ? -(GO + (c - left) * GE) : YD_LWORST); PF[c] = c == left ? -GO : YD_LWORST; PI[c] = 0; }
            // row 0 of the trace (SW.cpp:905-935): U at the origin, deletions to its right
#pragma unroll
            for (int k = 0; k < RD; k++) {
                uint32_t acc = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) { const int c = 4 * k + b; const uint32_t cell = c == left ? TR_U8 : ((c > left && c < W) ? (uint32_t)(OP_D | ((c - left) << 2)) : 0u);
                    acc |= cell << (8 * b); }
                T32[k * 64] = acc;
            }
            // reference window of row 1: column c holds reference index c - left
            unsigned long long win = 0;
#pragma unroll
            for (int c = 0; c < GW; c++) win |= (unsigned long long)refAt(c - left) << (4 * c);
            int qc = qAt(0);
            for (int i = 1; i <= qGap; i++) {
                const int qcNext = qAt(i < qGap ? i : qGap - 1);              // next row's query code and top reference base
                const uint32_t nbNext = refAt(i - left + GW - 1);
                { int sc = left + 1 - i; if (sc < 0) sc = 0; int ec = left + rGap - i; if (ec > W - 1) ec = W - 1; if (ec >= sc) cells += (unsigned)(ec - sc + 1); }
                int PVCol = YD_LWORST, PE = YD_LWORST, PD = 0, dV = PV[0];
                uint32_t acc = 0;
#pragma unroll
                for (int c = 0; c < GW; c++) {
                    const int rc = (int)((win >> (4 * c)) & 15ull);
                    const bool eq = rc == qc;
                    int V = dV + (eq ? MS : -RC); uint32_t cell = eq ? (uint32_t)OP_M : (uint32_t)OP_R;
                    const int CE = PE - GE, NE = PVCol - GOE;
                    const bool cE = CE >= NE && (PD + 1) <= maxIntron;
                    PE = cE ? CE : NE; PD = cE ? PD + 1 : 1;
                    if (PE > V) { V = PE; cell = (uint32_t)OP_D | ((uint32_t)PD << 2); }
                    int upV, upF, upI;
                    if (c + 1 < GW) { upV = PV[c + 1]; upF = PF[c + 1]; upI = PI[c + 1]; } else { upV = YD_LWORST; upF = YD_LWORST; upI = 0; }
                    const int CF = upF - GE, NF = upV - GOE;
                    const bool cF = CF >= NF && (upI + 1) <= maxGapP;
                    const int F = cF ? CF : NF, I = cF ? upI + 1 : 1;
                    if (F > V) { V = F; cell = (uint32_t)OP_I | ((uint32_t)I << 2); }
                    acc |= (cell & 0xFFu) << (8 * (c & 3));
                    if ((c & 3) == 3) { T32[(i * RD + (c >> 2)) * 64] = acc; acc = 0; }
                    const bool in = c < 11 || c < W;                          // W >= 2*BW + 1 = 11 at the default band; beyond W the sentinel stays
                    PV[c] = in ? V : YD_LWORST; PF[c] = in ? F : YD_LWORST; PI[c] = in ? I : 0;
                    PVCol = V; dV = upV;
                }
                win = (win >> 4) | ((unsigned long long)nbNext << (4 * (GW - 1)));
                qc = qcNext;
            }
            // the result is the cell (qGap, right)
            score = PV[0];
#pragma unroll
            for (int c = 1; c < GW; c++) score = c == right ? PV[c] : score;
            // traceback from the end cell (SW.cpp:1138-1195), as in gapDPLane
            int x = right, y = qGap;
            auto cellAt = [&](int yy, int xx) -> unsigned { const int c = yy * GW + xx; return (T32[(c >> 2) * 64] >> (8 * (c & 3))) & 0xFFu; };
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
        // op slots: wave prefix sum of nT (as k_gap_lanes)
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
