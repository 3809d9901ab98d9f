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

struct ChainMem { unsigned long long *key; uint32_t *fidx, *sidx, *diag; int *sqo, *eqo, *len, *best, *prev, *psqo; DevFrag *L; int *nx, *pv; uint8_t *cov; };
__host__ __device__ inline size_t chainScratchBytes(int maxN, int maxQ)
{ size_t b = (size_t)maxN * 72 + (size_t)maxQ + 320; return (b + 255) & ~(size_t)255; }
__device__ inline ChainMem carveChain(uint8_t *p, int maxN)
{
    ChainMem m; size_t N = (size_t)maxN;
    m.key = (unsigned long long *)p; p += N * 8;
    m.L = (DevFrag *)p; p += N * 16;
    m.fidx = (uint32_t *)p; p += N * 4; m.sidx = (uint32_t *)p; p += N * 4; m.diag = (uint32_t *)p; p += N * 4;
    m.sqo = (int *)p; p += N * 4; m.eqo = (int *)p; p += N * 4; m.len = (int *)p; p += N * 4; m.best = (int *)p; p += N * 4; m.prev = (int *)p; p += N * 4; m.psqo = (int *)p; p += N * 4;
    m.nx = (int *)p; p += N * 4; m.pv = (int *)p; p += N * 4;
    m.cov = (uint8_t *)p;
    return m;
}

__device__ inline bool emitChainClump(const ChainArgs &A, uint32_t rs, uint32_t region, uint32_t seq, int matched, const DevFrag *list, const int *nx, int head, int m, int lane)
{
    unsigned ci = 0, fi = 0;
    if (lane == 0) { ci = atomicAdd(&A.counts[0], 1u); fi = atomicAdd(&A.counts[1], (unsigned)m); }
    ci = uniU(ci); fi = uniU(fi);
    if (ci >= A.clumpCap || fi + (unsigned)m > A.fragCap) return false;
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

__global__ void __launch_bounds__(64) k_chain(ChainArgs A)
{
    const int lane = laneId(); const DevParams &P = A.P;
    ChainMem M = carveChain(A.scratch + (size_t)blockIdx.x * A.scratchPerWave, A.maxN);
    unsigned formed = 0;
    const unsigned nMulti = uniU(A.nMulti);
    for (;;) {
        if (__ballot(1) != ~0ull) { atomicCAS(A.errFlag, 0, (int)YERR_EXEC); break; }
        unsigned t = 0; if (lane == 0) t = atomicAdd(A.queueHead, 1u);
        const unsigned w = uniU(t);
        if (w >= nMulti) break;
        const uint32_t reg = uniU(A.multiList[w]); const uint32_t s = uniU(A.regStart[reg]), e = uniU(A.regStart[reg + 1]); const int n0 = (int)(e - s);
        const uint32_t rs = A.frags[s].rs; const uint32_t read = rs >> 1; const int qlen = (int)(A.B.readOff[read + 1] - A.B.readOff[read]);
        if (n0 > A.maxN) { if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_CHAIN); break; }
        for (int k = lane; k <= qlen; k += 64) M.cov[k] = 0;                       // setCoverage(QS, 0, queryLen, FALSE), GraphPath.cpp:276
        __threadfence_block();
        uint32_t seq = 0; bool fail = false;
        for (int iter = 0; iter <= n0; iter++) {                                    // processFragmentRangeUsingGraph, GraphPath.cpp:272-292
            // ---- node list = unused fragments (buildBestClumpFromFragmentRange :173-187)
            int cnt = 0;
            for (int base = 0; base < n0; base += 64) {
                int k = base + lane; bool valid = false; DevFrag f;
                if (k < n0) { f = A.frags[s + k]; valid = f.used == 0; }
                unsigned long long mask = __ballot(valid);
                if (valid) { int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull)); M.key[pos] = ((unsigned long long)f.sqo << 32) | (unsigned long long)(f.sro - (uint32_t)f.sqo); M.fidx[pos] = s + (uint32_t)k; }
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
            const int cSQO = first.sqo, cLen = (1 + (int)last.eqo - (int)first.sqo) & 0xFFFF;
            for (int k = lane; k < cLen; k += 64) if (cSQO + k <= qlen) M.cov[cSQO + k] = 1;      // setCoverage, :287
            __threadfence_block();
            const int minLeft = P.minNonOverlap - 1;                              // eliminateFragments / checkStartEndCoverage, QueryMatch.c:177-215
            for (int base = 0; base < n0; base += 64) {
                int k = base + lane;
                if (k < n0) {
                    DevFrag f = A.frags[s + k];
                    if (!f.used) {
                        const int SQO = f.sqo, EQO = f.eqo; bool keep;
                        if (EQO - SQO < minLeft) keep = false;
                        else {
                            bool a = true; for (int c = SQO; c <= SQO + minLeft; c++) if (M.cov[c]) { a = false; break; }
                            if (a) keep = true; else { bool b = true; for (int c = EQO - minLeft; c <= EQO; c++) if (M.cov[c]) { b = false; break; } keep = b; }
                        }
                        if (!keep) { f.used = 1; A.frags[s + k] = f; }
                    }
                }
            }
            __threadfence_block();
            if (!emitChainClump(A, rs, reg, seq, matched, M.L, M.nx, head, mm, lane)) { fail = true; break; }
            seq++; formed++;
        }
        if (lane == 0) A.regionClumpCount[reg] = seq;
        if (UNI_B(fail)) { if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_CHAIN); break; }
    }
    if (lane == 0 && formed) atomicAdd(&A.ctr->v[C_FORMED], (unsigned long long)formed);
}

// single-fragment regions: a clump iff refLen >= minMatch (QueryMatch.c:281-290)
__global__ void k_regions_single(ChainArgs A)
{
    const uint32_t reg = blockIdx.x * blockDim.x + threadIdx.x;
    if (reg >= A.nRegions) return;
    const uint32_t s = A.regStart[reg], e = A.regStart[reg + 1];
    if (e - s != 1) return;
    const DevFrag f = A.frags[s];
    uint32_t count = 0;
    if ((int)f.refLen >= A.P.minMatch) {
        unsigned ci = atomicAdd(&A.counts[0], 1u), fi = atomicAdd(&A.counts[1], 1u);
        if (ci >= A.clumpCap || fi >= A.fragCap) { atomicCAS(A.errFlag, 0, (int)YERR_CHAIN); return; }
        DevFrag g = f; g.used = 0; A.clumpFrags[fi] = g;
        ChainClumpRec r; r.rs = f.rs; r.fragOff = fi; r.nFrags = 1; r.region = reg; r.seq = 0; r.matched = f.refLen; A.clumps[ci] = r;
        count = 1; atomicAdd(&A.ctr->v[C_FORMED], 1ull);
    }
    A.regionClumpCount[reg] = count;
}
