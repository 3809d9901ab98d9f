// chain_lanes.h -- stage A4 for the small multi-fragment regions (2..8 fragments: the great majority): ONE REGION PER LANE.
//
// Same algorithm as chainSmall (chain.h) -- the best-chain DP over the region's unused fragments in (SQO, diag) order with the
// reference's tie rules, backtrack with overlap trimming, cleanUpClump, coverage-based elimination, repeated extraction
// (GraphPath.cpp:134-292, AlignHelpers.c:60-193, QueryMatch.c:170-215) -- written as the sequential code it is in the
// reference, with every per-region array in LDS laid out [index][lane].  A wave-per-region kernel spends a 64-wide
// instruction on two or three fragments; here the 64 lanes are 64 regions.
#pragma once
#include "chain.h"

// A wave of this kernel takes as long as its region with the most fragments, squared: three instances over three lists (regions.h: k_region_classify)
#define YD_CLT4 4                                 // fragments per region of the smallest class (2 .. 4: 72 % of the regions with more than one fragment) ...
#define YD_CL 8                                   // ... of the small class (5 .. 8) ...
#define YD_CLM 16                                 // ... and of the middle class (9 .. 16: 84 % of what k_chain took, a wave-wide instruction for a dozen fragments)

template <int YD_CLT>
struct ChainLaneLds {
    uint32_t fsro[YD_CLT][64]; uint16_t fsqo[YD_CLT][64], feqo[YD_CLT][64], frl[YD_CLT][64]; uint8_t used[YD_CLT][64];
    int16_t best[YD_CLT][64]; int8_t prev[YD_CLT][64], ord[YD_CLT][64]; uint16_t psqo[YD_CLT][64];
    uint32_t lsro[YD_CLT][64]; uint16_t lsqo[YD_CLT][64], leqo[YD_CLT][64], lrl[YD_CLT][64]; int8_t nx[YD_CLT][64], pv[YD_CLT][64];
    uint16_t ivS[YD_CLT + 1][64], ivL[YD_CLT + 1][64];
};

template <int YD_CLT>
__global__ void __launch_bounds__(64) k_chain_lanes(ChainArgs A, const uint32_t *smallList, uint32_t nSmall)
{
    YD_HIGH_PRIO();
    __shared__ ChainLaneLds<YD_CLT> T;
    const int lane = laneId(); const DevParams &P = A.P;
    const int MS = P.MS, GO = P.GO, GE = P.GE, maxGap = P.maxGap, maxDesert = P.maxDesert, minMatch = P.minMatch, minLeft = P.minNonOverlap - 1;
    unsigned formed = 0;
    unsigned cBase = 0, cLeft = 0, fBase = 0, fLeft = 0;                     // this wave's arena chunks (see ChainAlloc)
    for (uint32_t base = blockIdx.x * 64u; base < nSmall; base += gridDim.x * 64u) {
        const uint32_t w = base + (uint32_t)lane; const bool live = w < nSmall;
        uint32_t reg = 0, s = 0, rs = 0; int n0 = 0;
        if (live) {
            reg = smallList[w]; s = A.regStart[reg]; n0 = (int)(A.regStart[reg + 1] - s);
            for (int i = 0; i < n0; i++) { const DevFrag f = A.frags[s + (uint32_t)i]; T.fsro[i][lane] = f.sro; T.fsqo[i][lane] = f.sqo; T.feqo[i][lane] = f.eqo;
                T.frl[i][lane] = f.refLen; T.used[i][lane] = f.used != 0; if (i == 0) rs = f.rs; }
        }
        int nIv = 0; uint32_t seq = 0; bool active = live;
        for (int iter = 0; iter <= YD_CLT; iter++) {
            if (iter > n0) active = false;
            int head = -1, tail = -1, mm = 0, matched = 0; bool emit = false;
            if (active) {
                // the unused fragments in (SQO, diag) order (compareFragsByQueryOffsets, GraphPath.cpp:148-159)
                int cnt = 0;
                for (int i = 0; i < n0; i++) if (!T.used[i][lane]) {
                    const int q = T.fsqo[i][lane]; const uint32_t d = T.fsro[i][lane] - (uint32_t)q; int k = cnt;
                    while (k > 0) { const int o = T.ord[k - 1][lane]; const int oq = T.fsqo[o][lane]; const uint32_t od = T.fsro[o][lane] - (uint32_t)oq;
                        if (oq < q || (oq == q && od < d)) break; T.ord[k][lane] = (int8_t)o; k--; }
                    T.ord[k][lane] = (int8_t)i; cnt++;
                }
                if (cnt == 0) active = false;
                else {
                    for (int a = 0; a < cnt; a++) { const int i = T.ord[a][lane]; T.best[i][lane] = (int16_t)((int)(int16_t)T.frl[i][lane] * MS); T.prev[i][lane] = -1;
                        T.psqo[i][lane] = T.fsqo[i][lane]; }
                    int bestScore = YD_WORST, bestNode = -1, bestEQO = 0, bestPSQO = 0;
                    for (int a = 0; a < cnt; a++) {                           // chain DP, GraphPath.cpp:194-266
                        const int i = T.ord[a][lane];
                        const int lSQO = T.fsqo[i][lane], lEQO = T.feqo[i][lane], lbest = T.best[i][lane], lps = T.psqo[i][lane];
                        const uint32_t ld = T.fsro[i][lane] - (uint32_t)lSQO, lSRO = ld + (uint32_t)lSQO, lERO = ld + (uint32_t)lEQO;
                        for (int b = a + 1; b < cnt; b++) {
                            const int j = T.ord[b][lane]; const int fsqo = T.fsqo[j][lane];
                            if (fsqo == lSQO) continue;
                            const uint32_t ndiag = T.fsro[j][lane] - (uint32_t)fsqo; const int nlen = (int)(int16_t)T.frl[j][lane];
                            const uint32_t diagGap = absDiffU(ld, ndiag), rSRO = ndiag + (uint32_t)fsqo;
                            bool ok = diagGap <= (uint32_t)maxGap && lSRO < rSRO;
                            if (ok) { const uint32_t g1 = gapI(lEQO, fsqo), g2 = gapU(lERO, rSRO); ok = (int)(g1 < g2 ? g1 : g2) <= maxDesert; }
                            int newbases = 0;
                            if (ok) { const uint32_t o1 = ovlI(lEQO, fsqo), o2 = ovlU(lERO, rSRO); newbases = nlen - (int)(o1 > o2 ? o1 : o2); ok = newbases >= 1; }
                            if (!ok) continue;
                            const int newScore = lbest + newbases * MS + ((int)diagGap > 0 ? -(GO + (int)diagGap * GE) : 0);
                            const int best = T.best[j][lane], prevL = T.prev[j][lane];
                            bool take = true;
                            if (best > newScore) take = false;
                            else if (best == newScore) {
                                if (prevL < 0) take = false;
                                else {
                                    const int pq = T.fsqo[prevL][lane]; const uint32_t pdiag = T.fsro[prevL][lane] - (uint32_t)pq;
                                        const int pEQO = T.feqo[prevL][lane], pps = T.psqo[prevL][lane];
                                    const int dc = (int)(absDiffU(ld, ndiag) - absDiffU(pdiag, ndiag));
                                    if (dc > 0) take = false;
                                    else if (dc == 0) { const int gc = (int)(gapI(lEQO, fsqo) - gapI(pEQO, fsqo)); if (gc > 0) take = false;
                                        else if (gc == 0 && lps <= pps) take = false; }
                                }
                            }
                            if (take) { T.best[j][lane] = (int16_t)newScore; T.prev[j][lane] = (int8_t)i; T.psqo[j][lane] = (uint16_t)lps; }
                        }
                        if (!(lbest < bestScore)) {
                            bool better = lbest > bestScore;
                            if (!better) better = (lEQO != bestEQO) ? (lEQO < bestEQO) : (lps > bestPSQO);
                            if (better) { bestNode = i; bestScore = lbest; bestEQO = lEQO; bestPSQO = lps; }
                        }
                    }
                    // processBestFragmentPath / insertFragment (GraphPath.cpp:134-146, AlignHelpers.c:60-90)
                    for (int cur = bestNode; cur >= 0 && mm < YD_CLT; ) {
                        uint32_t s1 = T.fsro[cur][lane]; int q1 = T.fsqo[cur][lane], e1 = T.feqo[cur][lane], r1 = T.frl[cur][lane];
                        if (head >= 0) {
                            uint32_t s2 = T.lsro[head][lane]; int q2 = T.lsqo[head][lane], e2 = T.leqo[head][lane], r2 = T.lrl[head][lane];
                            const uint32_t o1 = ovlI(e1, q2), o2 = ovlU(s1 + (uint32_t)r1 - 1u, s2); const int mo = (int)(o1 > o2 ? o1 : o2);
                            if (mo > 0) {
                                const int l1 = fragQLen(q1, e1), l2 = fragQLen(q2, e2);
                                const bool chop1 = (l1 != l2) ? (l1 < l2) : (T.nx[head][lane] < 0);
                                if (chop1) { e1 = (e1 - mo) & 0xFFFF; r1 = (r1 - mo) & 0xFFFF; T.feqo[cur][lane] = (uint16_t)e1; T.frl[cur][lane] = (uint16_t)r1; }
                                else { q2 = (q2 + mo) & 0xFFFF; s2 += (uint32_t)mo; r2 = (r2 - mo) & 0xFFFF; T.lsqo[head][lane] = (uint16_t)q2; T.lsro[head][lane] = s2;
                                    T.lrl[head][lane] = (uint16_t)r2; }
                            }
                        }
                        matched = (matched + r1) & 0xFFFF;
                        const int id = mm++;
                        T.lsro[id][lane] = s1; T.lsqo[id][lane] = (uint16_t)q1; T.leqo[id][lane] = (uint16_t)e1; T.lrl[id][lane] = (uint16_t)r1; T.nx[id][lane] = (int8_t)head;
                            T.pv[id][lane] = -1;
                        if (head >= 0) T.pv[head][lane] = (int8_t)id; else tail = id;
                        head = id;
                        cur = T.prev[cur][lane];
                    }
                    if (matched < minMatch) active = false;
                    else {
                        // cleanUpClump, AlignHelpers.c:92-193 (same as cleanUpList in chain.h)
                        auto qlenOf = [&](int id) { return fragQLen(T.lsqo[id][lane], T.leqo[id][lane]); };
                        auto diagOf = [&](int id) { return T.lsro[id][lane] - (uint32_t)T.lsqo[id][lane]; };
                        auto nxt = [&](int id) { return (int)T.nx[id][lane]; };
                        auto removeNode = [&](int id) { const int n = T.nx[id][lane], p = T.pv[id][lane]; if (p < 0) head = n; else T.nx[p][lane] = (int8_t)n; if (n < 0) tail = p;
                            else T.pv[n][lane] = (int8_t)p; };
                        int S1 = head, S2 = S1 >= 0 ? nxt(S1) : -1, S3 = S2 >= 0 ? nxt(S2) : -1, guard = 0;
                        while (S2 >= 0 && S3 >= 0 && ++guard < 1000) {
                            if (qlenOf(S2) < P.wordLen) {
                                int anchor = S3;
                                while (qlenOf(anchor) < P.wordLen && nxt(anchor) >= 0) anchor = nxt(anchor);
                                const uint32_t f1 = diagOf(S1), ad = diagOf(anchor);
                                if (absDiffU(f1, ad) <= (uint32_t)P.maxGap) {
                                    int del = S2;
                                    while (del != anchor) {
                                        const int dn = nxt(del); const uint32_t dd = diagOf(del);
                                        const uint32_t m1 = absDiffU(f1, dd), m2 = absDiffU(dd, ad);
                                        if (!((dd < f1 && dd < ad) || (dd > f1 && dd > ad)) || ((m1 < m2 ? m1 : m2) <= (uint32_t)P.bandWidth)) removeNode(del);
                                        del = dn;
                                    }
                                }
                                S1 = anchor; S2 = nxt(anchor);
                            } else { S1 = S2; S2 = S3; }
                            if (S2 >= 0) S3 = nxt(S2);
                        }
                        auto endGapSmall = [&](int a, int b) {
                            const int qGap = (int)gapI(T.leqo[a][lane], T.lsqo[b][lane]), rGap = (int)gapU(T.lsro[a][lane] + (uint32_t)T.lrl[a][lane] - 1u, T.lsro[b][lane]);
                            return (qGap == 0 && rGap <= 2 * P.bandWidth) || (rGap == 0 && qGap <= 2 * P.bandWidth);
                        };
                        S1 = head;
                        if (qlenOf(S1) < P.wordLen && nxt(S1) >= 0) { if (endGapSmall(S1, nxt(S1))) removeNode(S1); }
                        S2 = tail;
                        if (qlenOf(S2) < P.wordLen) { S1 = T.pv[S2][lane]; if (S1 >= 0 && endGapSmall(S1, S2)) removeNode(S2); }
                        mm = 0; for (int id = head; id >= 0; id = nxt(id)) mm++;
                        const int cSQO = T.lsqo[head][lane], cLen = (1 + (int)T.leqo[tail][lane] - cSQO) & 0xFFFF;
                        T.ivS[nIv][lane] = (uint16_t)cSQO; T.ivL[nIv][lane] = (uint16_t)cLen; nIv++;
                        // eliminateFragments / checkStartEndCoverage, QueryMatch.c:177-215
                        for (int a = 0; a < cnt; a++) {
                            const int i = T.ord[a][lane]; const int fsqo = T.fsqo[i][lane], feqo = T.feqo[i][lane];
                            bool keep;
                            if (feqo - fsqo < minLeft) keep = false;
                            else {
                                bool aFree = true, bFree = true;
                                for (int e = 0; e < nIv; e++) {
                                    const int S0 = T.ivS[e][lane], E1 = S0 + (int)T.ivL[e][lane] - 1;
                                    if (S0 <= fsqo + minLeft && E1 >= fsqo) aFree = false;
                                    if (S0 <= feqo && E1 >= feqo - minLeft) bFree = false;
                                }
                                keep = aFree || bFree;
                            }
                            if (!keep) T.used[i][lane] = 1;
                        }
                        emit = true;
                    }
                }
            }
            // arena space for the clumps of this round: wave prefix sums inside the wave's chunks
            const unsigned long long em = __ballot(emit);
            if (em) {
                const unsigned nC = (unsigned)__builtin_popcountll(em);
                int incl = emit ? mm : 0;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
                const unsigned nF = (unsigned)__shfl(incl, 63, 64);
                if (cLeft < nC) { unsigned b = 0; if (lane == 0) b = atomicAdd(&A.counts[0], 512u); cBase = uniU(b); cLeft = 512u; }
                if (fLeft < nF) { unsigned b = 0; if (lane == 0) b = atomicAdd(&A.counts[1], 2048u); fBase = uniU(b); fLeft = 2048u; }
                const unsigned ci = cBase + (unsigned)__builtin_popcountll(em & ((1ull << lane) - 1ull)), fi = fBase + (unsigned)(incl - (emit ? mm : 0));
                cBase += nC; cLeft -= nC; fBase += nF; fLeft -= nF;
                if (emit) {
                    if (ci >= A.clumpCap || fi + (unsigned)mm > A.fragCap) atomicCAS(A.errFlag, 0, (int)YERR_CHAIN);
                    else {
                        int id = head;
                        for (int k = 0; k < mm; k++) { DevFrag f; f.sro = T.lsro[id][lane]; f.sqo = T.lsqo[id][lane]; f.eqo = T.leqo[id][lane]; f.refLen = T.lrl[id][lane];
                            f.used = 0; f.rs = rs; A.clumpFrags[fi + (unsigned)k] = f; id = T.nx[id][lane]; }
                        ChainClumpRec r; r.rs = rs; r.fragOff = fi; r.nFrags = (uint32_t)mm; r.region = reg; r.seq = seq; r.matched = (uint32_t)matched; A.clumps[ci] = r;
                        seq++;
                    }
                }
            }
            if (!__ballot(active)) break;
        }
        if (live) { A.regionClumpCount[reg] = seq; formed += seq; }
    }
    formed = (unsigned)waveSumI((int)formed);
    if (lane == 0 && formed) atomicAdd(&A.ctr->v[C_FORMED], (unsigned long long)formed);
}
