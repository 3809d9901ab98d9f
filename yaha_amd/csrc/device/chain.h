// chain.h -- stages A3 + A4 on the device: diagonal-region segmentation, the O(n^2) best-chain DP over the
// fragments of a region with repeated extraction, overlap trimming, clean-up and coverage-based elimination
// (reference QueryMatch.c:146-303, GraphPath.cpp:57-292, AlignHelpers.c:48-193).
//
// One wavefront owns one multi-fragment region.  The chain DP is sequential in the left node i but independent
// across right nodes j, so lanes take the j's (node state in the wave's HBM/L2 scratch, SoA).  Single-fragment
// regions (62 % on human-like data) are handled by a thread-per-region kernel.
#pragma once
#include "align.h"

struct ChainArgs {
    DevParams P; DevBatch B; DevFrag *frags; const uint32_t *regStart; uint32_t nRegions;
    const uint32_t *multiList; uint32_t nMulti; unsigned int *queueHead;
    uint8_t *scratch; size_t scratchPerWave; int maxN, maxQ;
    ChainClumpRec *clumps; DevFrag *clumpFrags; unsigned int *counts; uint32_t clumpCap, fragCap;
    uint32_t *regionClumpCount; int *errFlag; DevCounters *ctr;
};

struct ChainMem { unsigned long long *key; uint32_t *fidx, *sidx, *diag; int *sqo, *eqo, *len, *best, *prev, *psqo; DevFrag *L; int *nx, *pv; int *ivS, *ivL; };
__host__ __device__ inline size_t chainScratchBytes(int maxN, int maxQ)
{ (void)maxQ; size_t b = (size_t)maxN * 80 + 320; return (b + 255) & ~(size_t)255; }
__device__ inline ChainMem carveChain(uint8_t *p, int maxN)
{
    ChainMem m; size_t N = (size_t)maxN;
    m.key = (unsigned long long *)p; p += N * 8;
    m.L = (DevFrag *)p; p += N * 16;
    m.fidx = (uint32_t *)p; p += N * 4; m.sidx = (uint32_t *)p; p += N * 4; m.diag = (uint32_t *)p; p += N * 4;
    m.sqo = (int *)p; p += N * 4; m.eqo = (int *)p; p += N * 4; m.len = (int *)p; p += N * 4; m.best = (int *)p; p += N * 4; m.prev = (int *)p; p += N * 4; m.psqo = (int *)p;
        p += N * 4;
    m.nx = (int *)p; p += N * 4; m.pv = (int *)p; p += N * 4;
    m.ivS = (int *)p; p += (N + 2) * 4; m.ivL = (int *)p;      // one interval per extraction (at most n0 + 1)
    return m;
}

// Arena space is reserved per wavefront in chunks (one atomic per 32 clump records / 256 fragments instead of two per
// clump: a single L2 word takes only ~88 atomics/us).  Unused slots of a chunk stay marked invalid (nFrags = ~0).
struct ChainAlloc { unsigned cBase = 0, cLeft = 0, fBase = 0, fLeft = 0; };
#define YD_CLUMP_CHUNK 32u
#define YD_FRAG_CHUNK 256u
__device__ inline bool emitChainClump(const ChainArgs &A, ChainAlloc &al, uint32_t rs, uint32_t region, uint32_t seq, int matched, const DevFrag *list, const int *nx, int head,
    int m, int lane)
{
    if (al.cLeft == 0) { unsigned b = 0; if (lane == 0) b = atomicAdd(&A.counts[0], YD_CLUMP_CHUNK); al.cBase = uniU(b); al.cLeft = YD_CLUMP_CHUNK; }
    if (al.fLeft < (unsigned)m) { const unsigned want = (unsigned)m > YD_FRAG_CHUNK ? (unsigned)m : YD_FRAG_CHUNK; unsigned b = 0; if (lane == 0) b = atomicAdd(&A.counts[1], want);
        al.fBase = uniU(b); al.fLeft = want; }
    const unsigned ci = al.cBase, fi = al.fBase;
    if (ci >= A.clumpCap || fi + (unsigned)m > A.fragCap) return false;
    al.cBase++; al.cLeft--; al.fBase += (unsigned)m; al.fLeft -= (unsigned)m;
    int id = head;
    for (int k = 0; k < m; k++) { DevFrag f = list[id]; f.used = 0; f.rs = rs; A.clumpFrags[fi + k] = f; id = uni(nx[id]); }
    if (lane == 0) { ChainClumpRec r; r.rs = rs; r.fragOff = fi; r.nFrags = (uint32_t)m; r.region = region; r.seq = seq; r.matched = (uint32_t)matched; A.clumps[ci] = r; }
    return true;
}

// cleanUpClump, AlignHelpers.c:92-193, on an index-linked list (wave-uniform)
__device__ inline void cleanUpList(const DevParams &P, DevFrag *L, int *nx, int *pv, int &head, int &tail)
{
    auto qlenOf = [&](int id) { return fragQLen(L[id].sqo, L[id].eqo); };
    auto diagOf = [&](int id) { return L[id].sro - (uint32_t)L[id].sqo; };
    auto removeNode = [&](int id) { int n = nx[id], p = pv[id]; if (p < 0) head = n; else nx[p] = n; if (n < 0) tail = p; else pv[n] = p; };
    int S1 = head, S2 = S1 >= 0 ? nx[S1] : -1, S3 = S2 >= 0 ? nx[S2] : -1, guard = 0;
    while (S2 >= 0 && S3 >= 0 && ++guard < 1000000) {
        if (qlenOf(S2) < P.wordLen) {
            int anchor = S3;
            while (qlenOf(anchor) < P.wordLen && nx[anchor] >= 0) anchor = nx[anchor];
            uint32_t f1 = diagOf(S1), ad = diagOf(anchor);
            if (absDiffU(f1, ad) <= (uint32_t)P.maxGap) {
                int del = S2;
                while (del != anchor) {
                    int dn = nx[del]; uint32_t dd = diagOf(del);
                    uint32_t m1 = absDiffU(f1, dd), m2 = absDiffU(dd, ad);
                    if (!((dd < f1 && dd < ad) || (dd > f1 && dd > ad)) || ((m1 < m2 ? m1 : m2) <= (uint32_t)P.bandWidth)) removeNode(del);
                    del = dn;
                }
            }
            S1 = anchor; S2 = nx[anchor];
        } else { S1 = S2; S2 = S3; }
        if (S2 >= 0) S3 = nx[S2];
    }
    S1 = head;
    if (qlenOf(S1) < P.wordLen && nx[S1] >= 0) {
        const DevFrag a = L[S1], b = L[nx[S1]];
        int qGap = (int)gapI(a.eqo, b.sqo), rGap = (int)gapU(a.sro + a.refLen - 1u, b.sro);
        if ((qGap == 0 && rGap <= 2 * P.bandWidth) || (rGap == 0 && qGap <= 2 * P.bandWidth)) removeNode(S1);
    }
    S2 = tail;
    if (qlenOf(S2) < P.wordLen) {
        S1 = pv[S2]; if (S1 < 0) return;
        const DevFrag a = L[S1], b = L[S2];
        int qGap = (int)gapI(a.eqo, b.sqo), rGap = (int)gapU(a.sro + a.refLen - 1u, b.sro);
        if ((qGap == 0 && rGap <= 2 * P.bandWidth) || (rGap == 0 && qGap <= 2 * P.bandWidth)) removeNode(S2);
    }
}


// ---- fast path: a region with <= 64 fragments lives in registers (lane j = fragment j) and a little LDS -------------
// Nothing is physically sorted: every node knows its rank in (SQO, diag) order; the left node of step i is the lane whose
// rank is i (ballot + readlane broadcast), right nodes are the lanes with a larger rank.  The coverage map of
// processFragmentRangeUsingGraph (GraphPath.cpp:272-292) is the union of the extracted clumps' query spans, so it is
// kept as an interval list (lane e = interval e).  Trimming by insertFragment persists in the lane's registers,
// which is what the reference's in-place edit of fragArray amounts to within one region.
struct ChainLds { DevFrag L[64]; int nx[64], pv[64]; };

// `f` = this lane's fragment of the region (lane < n0), loaded by the caller ahead of time
__device__ inline uint32_t chainSmall(const ChainArgs &A, ChainAlloc &al, ChainLds &T, uint32_t reg, uint32_t s, int n0, uint32_t rs, DevFrag f, bool &fail)
{
    const DevParams &P = A.P; const int lane = laneId();
    const int MS = uni(P.MS), GO = uni(P.GO), GE = uni(P.GE), maxGap = uni(P.maxGap), maxDesert = uni(P.maxDesert), minMatch = uni(P.minMatch), minLeft = uni(P.minNonOverlap) - 1;
    if (lane >= n0) { f.sro = 0; f.sqo = 0; f.eqo = 0; f.refLen = 0; f.used = 1; f.rs = rs; }
    int fsro = (int)f.sro, fsqo = f.sqo, feqo = f.eqo, frl = f.refLen; bool used = lane >= n0 || f.used != 0;
    int ivS = 0, ivL = 0, nIv = 0;                                   // coverage intervals
    uint32_t seq = 0;
    for (int iter = 0; iter <= n0; iter++) {
        const bool valid = !used;
        const unsigned long long vmask = __ballot(valid);
        const int cnt = __popcll(vmask);
        if (cnt == 0) break;
        // rank in (SQO, diag) order (compareFragsByQueryOffsets, GraphPath.cpp:148-159; keys are unique)
        const uint32_t ndiag = (uint32_t)fsro - (uint32_t)fsqo;
        int r = 0;
        for (unsigned long long m = vmask; m; m &= m - 1) {
            const int k = __builtin_ctzll(m);
            const int oq = bcast(fsqo, k); const uint32_t od = (uint32_t)bcast((int)ndiag, k);
            r += (oq < fsqo || (oq == fsqo && od < ndiag)) ? 1 : 0;
        }
        const int nlen = (int)(int16_t)frl;
        int best = (int)(int16_t)(nlen * MS), prevL = -1, psqo = fsqo;
        int bestScore = YD_WORST, bestLane = -1, bestEQO = 0, bestPSQO = 0;
        for (int i = 0; i < cnt; i++) {                               // chain DP, GraphPath.cpp:194-266
            const int src = __builtin_ctzll(__ballot(valid && r == i));
            const uint32_t ld = (uint32_t)bcast((int)ndiag, src); const int lSQO = bcast(fsqo, src), lEQO = bcast(feqo, src), lbest = bcast(best, src), lps = bcast(psqo, src);
            const uint32_t lSRO = ld + (uint32_t)lSQO, lERO = ld + (uint32_t)lEQO;
            const int pl = prevL < 0 ? 0 : prevL;
            const uint32_t pdiag = (uint32_t)__shfl((int)ndiag, pl, 64); const int pEQO = __shfl(feqo, pl, 64), pps = __shfl(psqo, pl, 64);
            if (valid && r > i && fsqo != lSQO) {
                const uint32_t diagGap = absDiffU(ld, ndiag); const uint32_t rSRO = ndiag + (uint32_t)fsqo;
                bool ok = diagGap <= (uint32_t)maxGap && lSRO < rSRO;
                if (ok) { uint32_t g1 = gapI(lEQO, fsqo), g2 = gapU(lERO, rSRO); ok = (int)(g1 < g2 ? g1 : g2) <= maxDesert; }
                int newbases = 0;
                if (ok) { uint32_t o1 = ovlI(lEQO, fsqo), o2 = ovlU(lERO, rSRO); newbases = nlen - (int)(o1 > o2 ? o1 : o2); ok = newbases >= 1; }
                if (ok) {
                    const int newScore = lbest + newbases * MS + ((int)diagGap > 0 ? -(GO + (int)diagGap * GE) : 0);
                    bool take = true;
                    if (best > newScore) take = false;
                    else if (best == newScore) {
                        if (prevL < 0) take = false;
                        else {
                            const int dc = (int)(absDiffU(ld, ndiag) - absDiffU(pdiag, ndiag));
                            if (dc > 0) take = false;
                            else if (dc == 0) { const int gc = (int)(gapI(lEQO, fsqo) - gapI(pEQO, fsqo)); if (gc > 0) take = false; else if (gc == 0 && lps <= pps) take = false; }
                        }
                    }
                    if (take) { best = (int)(int16_t)newScore; prevL = src; psqo = lps; }
                }
            }
            if (!(lbest < bestScore)) {
                bool better = lbest > bestScore;
                if (!better) better = (lEQO != bestEQO) ? (lEQO < bestEQO) : (lps > bestPSQO);
                if (better) { bestLane = src; bestScore = lbest; bestEQO = lEQO; bestPSQO = lps; }
            }
        }
        // processBestFragmentPath / insertFragment (GraphPath.cpp:134-146, AlignHelpers.c:60-90)
        int head = -1, tail = -1, m = 0, matched = 0;
        for (int cur = bestLane; cur >= 0 && m < 64; ) {
            DevFrag f1; f1.sro = (uint32_t)bcast(fsro, cur); f1.sqo = (uint16_t)bcast(fsqo, cur); f1.eqo = (uint16_t)bcast(feqo, cur); f1.refLen = (uint16_t)bcast(frl, cur);
                f1.used = 0; f1.rs = rs;
            if (head >= 0) {
                DevFrag f2 = T.L[head];
                uint32_t o1 = ovlI(f1.eqo, f2.sqo), o2 = ovlU(f1.sro + f1.refLen - 1u, f2.sro); const int mo = uni((int)(o1 > o2 ? o1 : o2));
                if (mo > 0) {
                    const int l1 = fragQLen(f1.sqo, f1.eqo), l2 = fragQLen(f2.sqo, f2.eqo);
                    const bool chop1 = UNI_B((l1 != l2) ? (l1 < l2) : (T.nx[head] < 0));
                    if (chop1) { f1.eqo = (uint16_t)(f1.eqo - mo); f1.refLen = (uint16_t)(f1.refLen - mo); if (lane == cur) { feqo = f1.eqo; frl = f1.refLen; } }
                    else { f2.sqo = (uint16_t)(f2.sqo + mo); f2.sro += (uint32_t)mo; f2.refLen = (uint16_t)(f2.refLen - mo); T.L[head] = f2; }
                }
            }
            matched = (matched + f1.refLen) & 0xFFFF;
            const int id = m++;
            T.L[id] = f1; T.nx[id] = head; T.pv[id] = -1;
            if (head >= 0) T.pv[head] = id; else tail = id;
            head = id;
            cur = bcast(prevL, cur);
        }
        if (UNI_B(matched < minMatch)) break;
        cleanUpList(P, T.L, T.nx, T.pv, head, tail);
        head = uni(head); tail = uni(tail);
        int mm = 0; for (int id = head; id >= 0; id = uni(T.nx[id])) mm++;
        const int cSQO = uni((int)T.L[head].sqo), cLen = uni((1 + (int)T.L[tail].eqo - (int)T.L[head].sqo) & 0xFFFF);
        if (lane == nIv) { ivS = cSQO; ivL = cLen; }
        nIv++;
        if (valid) {                                                  // eliminateFragments / checkStartEndCoverage, QueryMatch.c:177-215
            bool keep;
            if (feqo - fsqo < minLeft) keep = false;
            else {
                bool aFree = true, bFree = true;
                for (int e = 0; e < nIv; e++) {
                    const int S0 = bcast(ivS, e), S1 = S0 + bcast(ivL, e) - 1;      // covered [S0, S1]
                    if (S0 <= fsqo + minLeft && S1 >= fsqo) aFree = false;
                    if (S0 <= feqo && S1 >= feqo - minLeft) bFree = false;
                }
                keep = aFree || bFree;
            }
            if (!keep) used = true;
        }
        if (!emitChainClump(A, al, rs, reg, seq, matched, T.L, T.nx, head, mm, lane)) { fail = true; break; }
        seq++;
    }
    return seq;
}

// general path (any region size): node state in the arrays of ChainMem -- LDS for regions up to YD_CHAIN_LDS_N nodes,
// the wave's HBM scratch beyond that.
__device__ inline uint32_t chainGeneral(const ChainArgs &A, ChainAlloc &al, const ChainMem &M, uint32_t reg, uint32_t s, int n0, uint32_t rs, bool &fail)
{
    const DevParams &P = A.P; const int lane = laneId();
        uint32_t seq = 0; int nIv = 0;                                          // coverage = interval list (see chainSmall)
    for (int iter = 0; iter <= n0; iter++) {                                    // processFragmentRangeUsingGraph, GraphPath.cpp:272-292
        // ---- node list = unused fragments (buildBestClumpFromFragmentRange :173-187)
        int cnt = 0;
        for (int base = 0; base < n0; base += 64) {
            int k = base + lane; bool valid = false; DevFrag f;
            if (k < n0) { f = A.frags[s + k]; valid = f.used == 0; }
            unsigned long long mask = __ballot(valid);
            if (valid) { int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull)); M.key[pos] = ((unsigned long long)f.sqo << 32) | (unsigned long long)(f.sro - (uint32_t)f.sqo);
                M.fidx[pos] = s + (uint32_t)k; }
            cnt += __popcll(mask);
        }
        cnt = uni(cnt);
        if (cnt == 0) break;
        __threadfence_block();
        // ---- sort by (SQO, diag) (compareFragsByQueryOffsets :148-159; keys are unique) -> rank sort
        for (int base = 0; base < cnt; base += 64) {
            int j = base + lane;
            if (j < cnt) {
                unsigned long long kj = M.key[j]; int r = 0;
                for (int k = 0; k < cnt; k++) r += (M.key[k] < kj) ? 1 : 0;
                const uint32_t fi = M.fidx[j]; const DevFrag f = A.frags[fi];
                M.sidx[r] = fi;                                             // node r <-> fragment fi
                M.diag[r] = f.sro - (uint32_t)f.sqo; M.sqo[r] = f.sqo; M.eqo[r] = f.eqo; M.len[r] = (int)(int16_t)f.refLen;
                M.best[r] = (int)(int16_t)((int)(int16_t)f.refLen * P.MS); M.prev[r] = -1; M.psqo[r] = f.sqo;
            }
        }
        __threadfence_block();
        // ---- chain DP (:194-266): left node i sequential, right nodes j across lanes
        int bestScore = YD_WORST, bestNode = -1, bestEQO = 0, bestPSQO = 0;
        for (int i = 0; i < cnt; i++) {
            const uint32_t ld = M.diag[i]; const int lSQO = M.sqo[i], lEQO = M.eqo[i], lbest = M.best[i], lps = M.psqo[i];
            const uint32_t lSRO = ld + (uint32_t)lSQO, lERO = ld + (uint32_t)lEQO;
            for (int base = (i + 1) & ~63; base < cnt; base += 64) {
                const int j = base + lane;
                if (j > i && j < cnt) {
                    const int rSQO = M.sqo[j];
                    if (rSQO != lSQO) {
                        const uint32_t rd = M.diag[j]; const uint32_t diagGap = absDiffU(ld, rd);
                        const uint32_t rSRO = rd + (uint32_t)rSQO;
                        bool ok = diagGap <= (uint32_t)P.maxGap && lSRO < rSRO;
                        if (ok) { uint32_t g1 = gapI(lEQO, rSQO), g2 = gapU(lERO, rSRO); int desert = (int)(g1 < g2 ? g1 : g2); ok = desert <= P.maxDesert; }
                        int newbases = 0;
                        if (ok) { uint32_t o1 = ovlI(lEQO, rSQO), o2 = ovlU(lERO, rSRO); int mo = (int)(o1 > o2 ? o1 : o2); newbases = M.len[j] - mo; ok = newbases >= 1; }
                        if (ok) {
                            const int newScore = lbest + newbases * P.MS + ((int)diagGap > 0 ? -(P.GO + (int)diagGap * P.GE) : 0);
                            const int rbest = M.best[j];
                            bool take = true;
                            if (rbest > newScore) take = false;
                            else if (rbest == newScore) {
                                const int pb = M.prev[j];
                                if (pb < 0) take = false;
                                else {
                                    const int dc = (int)(absDiffU(ld, rd) - absDiffU(M.diag[pb], rd));
                                    if (dc > 0) take = false;
                                    else if (dc == 0) {
                                        const int gc = (int)(gapI(lEQO, rSQO) - gapI(M.eqo[pb], rSQO));
                                        if (gc > 0) take = false; else if (gc == 0 && lps <= M.psqo[pb]) take = false;
                                    }
                                }
                            }
                            if (take) { M.best[j] = (int)(int16_t)newScore; M.prev[j] = i; M.psqo[j] = lps; }
                        }
                    }
                }
            }
            if (!(lbest < bestScore)) {
                bool better = lbest > bestScore;
                if (!better) better = (lEQO != bestEQO) ? (lEQO < bestEQO) : (lps > bestPSQO);     // differentiateEqualFragNodesDuringBacktrack :88-94
                if (better) { bestNode = i; bestScore = lbest; bestEQO = lEQO; bestPSQO = lps; }
            }
            __threadfence_block();
        }
        // ---- processBestFragmentPath (:134-146): insertFragment front to back with overlap trimming (AlignHelpers.c:60-90)
        int head = -1, tail = -1, m = 0, matched = 0;
        {
            for (int cur = uni(bestNode); cur >= 0; cur = uni(M.prev[cur])) {
                const uint32_t fi = uniU(M.sidx[cur]);
                DevFrag f1 = A.frags[fi];
                if (head >= 0) {
                    DevFrag f2 = M.L[head];
                    uint32_t o1 = ovlI(f1.eqo, f2.sqo), o2 = ovlU(f1.sro + f1.refLen - 1u, f2.sro); const int mo = (int)(o1 > o2 ? o1 : o2);   // calcMaxOverlap
                    if (mo > 0) {
                        const int l1 = fragQLen(f1.sqo, f1.eqo), l2 = fragQLen(f2.sqo, f2.eqo);
                        const bool chop1 = (l1 != l2) ? (l1 < l2) : (M.nx[head] < 0);
                        if (chop1) { f1.eqo = (uint16_t)(f1.eqo - mo); f1.refLen = (uint16_t)(f1.refLen - mo); A.frags[fi] = f1; }              // trims fragArray in place
                        else { f2.sqo = (uint16_t)(f2.sqo + mo); f2.sro += (uint32_t)mo; f2.refLen = (uint16_t)(f2.refLen - mo); M.L[head] = f2; }
                    }
                }
                matched = (matched + f1.refLen) & 0xFFFF;
                const int id = m++;                                            // new list node
                M.L[id] = f1; M.nx[id] = head; M.pv[id] = -1;
                if (head >= 0) M.pv[head] = id; else tail = id;
                head = id;
            }
        }
        if (UNI_B(matched < P.minMatch)) break;                                       // resetClump -> empty -> region finished (:142-143, 281-285)
        cleanUpList(P, M.L, M.nx, M.pv, head, tail);
        head = uni(head); tail = uni(tail);
        int mm = 0; for (int id = head; id >= 0; id = uni(M.nx[id])) mm++;
        const DevFrag first = M.L[head], last = M.L[tail];
        const int cSQO = uni((int)first.sqo), cLen = uni((1 + (int)last.eqo - (int)first.sqo) & 0xFFFF);
        M.ivS[nIv] = cSQO; M.ivL[nIv] = cLen; nIv++;                                        // setCoverage, :287
        const int minLeft = P.minNonOverlap - 1;                              // eliminateFragments / checkStartEndCoverage, QueryMatch.c:177-215
        for (int base = 0; base < n0; base += 64) {
            int k = base + lane;
            if (k < n0) {
                DevFrag f = A.frags[s + k];
                if (!f.used) {
                    const int SQO = f.sqo, EQO = f.eqo; bool keep;
                    if (EQO - SQO < minLeft) keep = false;
                    else {
                        bool aFree = true, bFree = true;
                        for (int e2 = 0; e2 < nIv; e2++) { const int S0 = M.ivS[e2], S1 = S0 + M.ivL[e2] - 1; if (S0 <= SQO + minLeft && S1 >= SQO) aFree = false;
                            if (S0 <= EQO && S1 >= EQO - minLeft) bFree = false; }
                        keep = aFree || bFree;
                    }
                    if (!keep) { f.used = 1; A.frags[s + k] = f; }
                }
            }
        }
        __threadfence_block();
        if (!emitChainClump(A, al, rs, reg, seq, matched, M.L, M.nx, head, mm, lane)) { fail = true; break; }
        seq++;
    }
    return seq;
}

__global__ void __launch_bounds__(64) k_chain(ChainArgs A)
{
    YD_HIGH_PRIO();
    __shared__ ChainLds sT;
    const int lane = laneId();
    unsigned formed = 0; ChainAlloc al;
    const unsigned nMulti = uniU(A.nMulti);
    for (;;) {
        if (__ballot(1) != ~0ull) { atomicCAS(A.errFlag, 0, (int)YERR_EXEC); break; }
        unsigned t = 0; if (lane == 0) t = atomicAdd(A.queueHead, 8u);           // eight regions per pop (queue word contention)
        const unsigned w0 = uniU(t);
        if (w0 >= nMulti) break;
        bool failS = false;
        // the headers of the eight regions in one round of loads (lane k: region k), each region's fragments one region ahead
        const unsigned cntR = min(8u, nMulti - w0);
        uint32_t hReg = 0, hS = 0, hE = 0, hRs = 0;
        if ((unsigned)lane < cntR) { hReg = A.multiList[w0 + (unsigned)lane]; hS = A.regStart[hReg]; hE = A.regStart[hReg + 1]; hRs = A.frags[hS].rs; }
        DevFrag fNext; fNext.sro = 0; fNext.sqo = 0; fNext.eqo = 0; fNext.refLen = 0; fNext.used = 1; fNext.rs = 0;
        { const uint32_t s0 = (uint32_t)bcast((int)hS, 0), e0 = (uint32_t)bcast((int)hE, 0); if ((uint32_t)lane < e0 - s0) fNext = A.frags[s0 + (uint32_t)lane]; }
        for (unsigned k = 0; k < cntR && !failS; k++) {
            const uint32_t reg = (uint32_t)bcast((int)hReg, (int)k), s = (uint32_t)bcast((int)hS, (int)k), e = (uint32_t)bcast((int)hE, (int)k), rs = (uint32_t)bcast((int)hRs,
                (int)k); const int n0 = (int)(e - s);
            const DevFrag fCur = fNext;
            if (k + 1 < cntR) { const uint32_t s1 = (uint32_t)bcast((int)hS, (int)k + 1), e1 = (uint32_t)bcast((int)hE, (int)k + 1);
                if ((uint32_t)lane < e1 - s1) fNext = A.frags[s1 + (uint32_t)lane]; }
            const uint32_t seqS = chainSmall(A, al, sT, reg, s, n0, rs, fCur, failS);
            if (lane == 0) A.regionClumpCount[reg] = seqS;
            formed += seqS; failS = UNI_B(failS);
        }
        if (failS) { if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_CHAIN); break; }
    }
    if (lane == 0 && formed) atomicAdd(&A.ctr->v[C_FORMED], (unsigned long long)formed);
}

#define YD_CHAIN_LDS_N 512
// regions with more than 64 fragments: one wavefront each, node arrays in 40 KB of LDS (up to 512 nodes) so that the
// serial steps of the chain DP wait on LDS instead of L2
__global__ void __launch_bounds__(64) k_chain_big(ChainArgs A, const uint32_t *bigList, uint32_t nBig, unsigned int *queueHead)
{
    YD_HIGH_PRIO();
    __shared__ __attribute__((aligned(16))) uint8_t smem[YD_CHAIN_LDS_N * 80 + 64];
    const int lane = laneId();
    ChainMem G = carveChain(A.scratch + (size_t)blockIdx.x * A.scratchPerWave, A.maxN), Lm = carveChain(smem, YD_CHAIN_LDS_N);
    unsigned formed = 0; const unsigned nb = uniU(nBig); ChainAlloc al;
    for (;;) {
        if (__ballot(1) != ~0ull) { atomicCAS(A.errFlag, 0, (int)YERR_EXEC); break; }
        unsigned t = 0; if (lane == 0) t = atomicAdd(queueHead, 1u);
        const unsigned w = uniU(t);
        if (w >= nb) break;
        const uint32_t reg = uniU(bigList[w]); const uint32_t s = uniU(A.regStart[reg]), e = uniU(A.regStart[reg + 1]); const int n0 = (int)(e - s);
        const uint32_t rs = uniU(A.frags[s].rs);
        if (n0 > A.maxN) { if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_CHAIN); break; }
        bool fail = false;
        const uint32_t seq = chainGeneral(A, al, n0 <= YD_CHAIN_LDS_N ? Lm : G, reg, s, n0, rs, fail);
        if (lane == 0) A.regionClumpCount[reg] = seq;
        formed += seq;
        if (UNI_B(fail)) { if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_CHAIN); break; }
    }
    if (lane == 0 && formed) atomicAdd(&A.ctr->v[C_FORMED], (unsigned long long)formed);
}

// single-fragment regions: a clump iff refLen >= minMatch (QueryMatch.c:281-290); one allocation per wavefront
// Two atomics per 1024-thread block (a single L2 word takes only ~88 atomics/us; there are ~10^5 waves here).
__global__ void __launch_bounds__(1024) k_regions_single(ChainArgs A)
{
    YD_HIGH_PRIO();
    __shared__ unsigned sCnt[16], sBase[2];
    const uint32_t reg = blockIdx.x * blockDim.x + threadIdx.x; const int lane = laneId(), wv = (int)(threadIdx.x >> 6);
    bool make = false; DevFrag f; f.refLen = 0; f.rs = 0;
    if (reg < A.nRegions) {
        const uint32_t s = A.regStart[reg], e = A.regStart[reg + 1];
        if (e - s == 1) { f = A.frags[s]; make = (int)f.refLen >= A.P.minMatch; A.regionClumpCount[reg] = make ? 1u : 0u; }
    }
    const unsigned long long m = __ballot(make);
    if (lane == 0) sCnt[wv] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned n = 0; for (unsigned k = 0; k < blockDim.x / 64u; k++) n += sCnt[k];
        // clump slots and fragment slots advance together: one 64-bit add on the pair of counters (counts is 8-byte aligned)
        if (n) { const unsigned long long old = atomicAdd((unsigned long long *)&A.counts[0], (unsigned long long)n | ((unsigned long long)n << 32)); sBase[0] = (unsigned)old;
            sBase[1] = (unsigned)(old >> 32); atomicAdd(&A.ctr->v[C_FORMED], (unsigned long long)n); }
    }
    __syncthreads();
    if (make) {
        unsigned before = 0; for (int k = 0; k < wv; k++) before += sCnt[k];
        const unsigned k = before + (unsigned)__popcll(m & ((1ull << lane) - 1ull)); const unsigned ci = sBase[0] + k, fi = sBase[1] + k;
        if (ci >= A.clumpCap || fi >= A.fragCap) { atomicCAS(A.errFlag, 0, (int)YERR_CHAIN); return; }
        DevFrag g = f; g.used = 0; A.clumpFrags[fi] = g;
        ChainClumpRec r; r.rs = f.rs; r.fragOff = fi; r.nFrags = 1; r.region = reg; r.seq = 0; r.matched = f.refLen; A.clumps[ci] = r;
    }
}
