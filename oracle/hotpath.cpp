// oracle/hotpath.cpp -- TEST INFRASTRUCTURE ONLY (see hotpath.h).
// Sequential CPU restatement of the reference's hot path with flat arrays instead of the reference's
// slab-allocated linked lists.  Every function cites the reference lines it follows (paths relative to
// /root/reference/src).  Integer widths (uint16 query offsets / lengths, int16 chain scores, uint32
// reference offsets with wrap-around) are kept on purpose: they are observable.
#include "hotpath.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>

namespace yoracle {

#define DPWorstScore (-(0x7fffff00))                        // SW.cpp:356, GraphPath.cpp:55
enum { stReversed = 0x01, stAligned = 0x04, stScored = 0x08, stSplit = 0x10 };   // FragsClumps.inl:235-240

static inline uint8_t ref4(const uint8_t *bases, uint32_t off)      // getFrom4Code, Math.c:180-188
{ uint8_t b = bases[off >> 1]; return (off & 1) ? (b & 0xF) : (uint8_t)(b >> 4); }
static const uint8_t COMP4[16] = {2, 3, 0, 1, 4, 12, 7, 6, 9, 8, 15, 11, 5, 13, 14, 10};   // fourBitCompCodes, Math.c:156

// ---- fragment geometry, FragsClumps.inl:35-193 -------------------------------------------------------
static inline int      fragQueryLen(const Frag &f) { return 1 + (int)f.eqo - (int)f.sqo; }
static inline uint32_t fragERO(const Frag &f) { return f.sro + f.refLen - 1; }
static inline uint32_t fragDiag(const Frag &f) { return f.sro - f.sqo; }
static inline uint32_t absDiff(uint32_t a, uint32_t b) { return a > b ? a - b : b - a; }          // :133-137
static inline int      uintDiff(uint32_t a, uint32_t b) { return a >= b ? (int)(a - b) : -((int)(b - a)); }   // :127-131
static inline uint32_t gapI(int lo, int hi) { return hi > lo ? (uint32_t)(hi - lo) - 1 : 0; }      // calcGap :158
static inline uint32_t gapU(uint32_t lo, uint32_t hi) { return hi > lo ? (hi - lo) - 1 : 0; }
static inline uint32_t ovlI(int lo, int hi) { return lo >= hi ? (uint32_t)(lo - hi) + 1 : 0; }     // calcOverlap :159
static inline uint32_t ovlU(uint32_t lo, uint32_t hi) { return lo >= hi ? (lo - hi) + 1 : 0; }
static inline int gapCost(int len, const ygpu_params &P) { return len > 0 ? -(P.GOCost + len * P.GECost) : 0; }   // :190-193

// ---- edit-op lists, SW.cpp:151-283 (knit-point merging of equal opcodes) -----------------------------
static void mergeToBack(OpList &target, OpList &source)              // mergeEOLToBack, SW.cpp:207-261
{
    if (source.empty()) return;
    if (target.empty()) { target.swap(source); return; }
    size_t from = 0;
    if (target.back().code == source.front().code) { target.back().len = (uint16_t)(target.back().len + source.front().len); from = 1; }
    target.insert(target.end(), source.begin() + from, source.end());
    source.clear();
}
static void mergeToFront(OpList &target, OpList &source)             // mergeEOLToFront, SW.cpp:151-205
{
    if (source.empty()) return;
    if (target.empty()) { target.swap(source); return; }
    size_t from = 0;
    if (source.back().code == target.front().code) { source.back().len = (uint16_t)(source.back().len + target.front().len); from = 1; }
    source.insert(source.end(), target.begin() + from, target.end());
    target.swap(source); source.clear();
}

// =======================================================================================================
// A1 + A2  k-mer lookup and seed-hit join
// =======================================================================================================
void seedJoin(const Index &ix, const ygpu_params &P, const uint8_t *codes, int qlen, std::vector<Frag> &frags,
              ygpu_counters *ctr)
{
    frags.clear();
    const int L = P.wordLen;
    if (qlen < L) return;
    const int nPos = qlen - L + 1;                                   // offsetCount, Query.c:341
    struct OC { uint32_t s, c; };
    std::vector<OC> oc(nPos);
    const uint32_t mask = 0xFFFFFFFFu >> (32 - 2 * L);              // Query.c:296
    // Query.c:365-412: a k-mer is dropped iff its window holds a code > 3 or its count exceeds maxHits.
    uint64_t total = 0; int lastBad = -1; uint32_t h = 0;
    for (int p = 0; p < qlen; p++) {
        uint8_t c = codes[p];
        if (c > 3) lastBad = p;
        h = ((h << 2) | (c & 3)) & mask;
        int i = p - L + 1;
        if (i < 0) continue;
        if (lastBad >= i) { oc[i] = {0, 0}; continue; }
        uint32_t s = ix.SO[h], cnt = ix.SO[h + 1] - s;
        if (ctr) ctr->kmer_lookups++;
        if (cnt <= (uint32_t)P.maxHits) { oc[i] = {s, cnt}; total += cnt; } else oc[i] = {0, 0};
    }
    if (total == 0) return;                                          // Query.c:426
    // QueryMatch.c:56-69: every hit list enters the merge; hits whose diagonal wraps below zero (roff < i)
    // are pre-loaded, and if *all* of a k-mer's hits wrap the loop keeps reading past the list's end until it
    // meets a non-wrapping entry, which joins the merge too.  The merge pops in ascending (diag<<32)+i
    // (QueryHeap.inl:70-73), i.e. the output is the sorted multiset.
    std::vector<uint64_t> keys; keys.reserve(total + 8);
    for (int i = 0; i < nPos; i++) {
        uint32_t c = oc[i].c, s = oc[i].s;
        if (c == 0) continue;
        uint32_t w = 0;
        while ((uint64_t)s + w < ix.totalMatches && ix.ROA[s + w] < (uint32_t)i) w++;
        uint32_t eff = (w < c) ? c : w + 1;
        for (uint32_t j = 0; j < eff && (uint64_t)s + j < ix.totalMatches; j++)
            keys.push_back(((uint64_t)(uint32_t)(ix.ROA[s + j] - (uint32_t)i) << 32) | (uint32_t)i);
    }
    if (ctr) ctr->hits += keys.size();
    std::sort(keys.begin(), keys.end());
    // QueryMatch.c:76-119: coalesce same-diagonal hits whose k-mers overlap or abut.
    uint32_t curDiag = (uint32_t)(keys[0] >> 32); uint16_t curQO = (uint16_t)(keys[0] & 0xFFFFFFFFu);
    Frag cur; cur.sqo = curQO; cur.sro = curDiag + curQO;
    uint16_t curEQO = (uint16_t)(curQO + L);
    auto finish = [&]() { cur.eqo = (uint16_t)(curEQO - 1); cur.refLen = (uint16_t)fragQueryLen(cur); frags.push_back(cur); };
    for (size_t k = 0; k < keys.size(); k++) {
        uint32_t d = (uint32_t)(keys[k] >> 32); uint16_t q = (uint16_t)(keys[k] & 0xFFFFFFFFu);
        if (d != curDiag || q > curEQO) { finish(); curDiag = d; curQO = q; cur.sqo = q; cur.sro = d + q; curEQO = (uint16_t)(q + L); }
        else curEQO = (uint16_t)(q + L);
    }
    finish();
    if (ctr) ctr->fragments += frags.size();
}

// =======================================================================================================
// A3 + A4  region segmentation, chain DP, clump construction
// =======================================================================================================
namespace {
struct FList {                                           // SFragmentList_t restated with index links
    std::vector<Frag> f; std::vector<int> nx, pv; int head = -1, tail = -1;
    bool empty() const { return head < 0; }
    void pushFront(const Frag &x) { int id = (int)f.size(); f.push_back(x); nx.push_back(head); pv.push_back(-1); if (head >= 0) pv[head] = id; else tail = id; head = id; }
    void remove(int id) { int n = nx[id], p = pv[id]; if (p < 0) head = n; else nx[p] = n; if (n < 0) tail = p; else pv[n] = p; }
};
struct GNode {                                           // fGraphNode, GraphPath.cpp:65-79
    int bestPrev; int fragIdx; int16_t bestScore, pathLength; uint16_t pathSQO; uint32_t diag; int16_t nodeLength; uint16_t SQO, EQO;
};
struct BuildClump { FList list; uint16_t matchedBases = 0; };

void insertFragment(BuildClump &c, Frag &frag1)          // AlignHelpers.c:48-90 (frag1 aliases fragArray)
{
    if (!c.list.empty()) {
        int nextId = c.list.head; Frag &frag2 = c.list.f[nextId];
        int maxOverlap = (int)std::max(ovlI(frag1.eqo, frag2.sqo), ovlU(fragERO(frag1), frag2.sro));   // calcMaxOverlap :161-164
        if (maxOverlap > 0) {
            int l1 = fragQueryLen(frag1), l2 = fragQueryLen(frag2);
            bool chop1 = (l1 != l2) ? (l1 < l2) : (c.list.nx[nextId] < 0);
            if (chop1) { frag1.eqo = (uint16_t)(frag1.eqo - maxOverlap); frag1.refLen = (uint16_t)(frag1.refLen - maxOverlap); }              // subLenFromBack
            else { frag2.sqo = (uint16_t)(frag2.sqo + maxOverlap); frag2.sro += maxOverlap; frag2.refLen = (uint16_t)(frag2.refLen - maxOverlap); } // subLenFromFront
        }
    }
    c.matchedBases = (uint16_t)(c.matchedBases + frag1.refLen);      // addFragment :48-56
    c.list.pushFront(frag1);
}

void cleanUpClump(BuildClump &c, const ygpu_params &P)   // AlignHelpers.c:92-193
{
    FList &l = c.list;
    int S1 = l.head, S2 = S1 >= 0 ? l.nx[S1] : -1, S3 = S2 >= 0 ? l.nx[S2] : -1;
    while (S2 >= 0 && S3 >= 0) {
        if (fragQueryLen(l.f[S2]) < P.wordLen) {
            int anchor = S3;
            while (fragQueryLen(l.f[anchor]) < P.wordLen && l.nx[anchor] >= 0) anchor = l.nx[anchor];
            uint32_t f1Diag = fragDiag(l.f[S1]), aDiag = fragDiag(l.f[anchor]);
            if (absDiff(f1Diag, aDiag) <= (uint32_t)P.maxGap) {
                int del = S2;
                while (del != anchor) {
                    int dn = l.nx[del]; uint32_t dDiag = fragDiag(l.f[del]);
                    if (!((dDiag < f1Diag && dDiag < aDiag) || (dDiag > f1Diag && dDiag > aDiag)) ||
                        (std::min(absDiff(f1Diag, dDiag), absDiff(dDiag, aDiag)) <= (uint32_t)P.bandWidth)) l.remove(del);
                    del = dn;
                }
            }
            S1 = anchor; S2 = l.nx[anchor];
        } else { S1 = S2; S2 = S3; }
        if (S2 >= 0) S3 = l.nx[S2];
    }
    S1 = l.head;
    if (fragQueryLen(l.f[S1]) < P.wordLen && l.nx[S1] >= 0) {       // (the reference would dereference NULL for a lone short frag)
        const Frag &a = l.f[S1], &b = l.f[l.nx[S1]];
        int qGap = (int)gapI(a.eqo, b.sqo), rGap = (int)gapU(fragERO(a), b.sro);
        if ((qGap == 0 && rGap <= 2 * P.bandWidth) || (rGap == 0 && qGap <= 2 * P.bandWidth)) l.remove(S1);
    }
    S2 = l.tail;
    if (fragQueryLen(l.f[S2]) < P.wordLen) {
        S1 = l.pv[S2]; if (S1 < 0) return;
        const Frag &a = l.f[S1], &b = l.f[S2];
        int qGap = (int)gapI(a.eqo, b.sqo), rGap = (int)gapU(fragERO(a), b.sro);
        if ((qGap == 0 && rGap <= 2 * P.bandWidth) || (rGap == 0 && qGap <= 2 * P.bandWidth)) l.remove(S2);
    }
}

// buildBestClumpFromFragmentRange, GraphPath.cpp:161-270
void buildBestClump(const ygpu_params &P, std::vector<Frag> &frags, const std::vector<uint8_t> &used, int startF, int endF,
                    BuildClump &clump, std::vector<GNode> &nodes)
{
    nodes.clear();
    for (int i = startF; i <= endF; i++) {
        if (used[i]) continue;
        const Frag &f = frags[i]; GNode n;
        n.bestPrev = -1; n.pathLength = 1; n.fragIdx = i; n.diag = fragDiag(f); n.nodeLength = (int16_t)f.refLen;
        n.bestScore = (int16_t)(n.nodeLength * P.MScore); n.SQO = f.sqo; n.EQO = f.eqo; n.pathSQO = n.SQO;
        nodes.push_back(n);
    }
    int nodeCount = (int)nodes.size();
    if (nodeCount == 0) return;
    std::sort(nodes.begin(), nodes.end(), [](const GNode &a, const GNode &b) {        // compareFragsByQueryOffsets :148-159
        int d = (int)a.SQO - (int)b.SQO; if (d != 0) return d < 0; return uintDiff(a.diag, b.diag) < 0; });
    int bestScore = DPWorstScore, bestNode = -1;
    const uint32_t maxGap = (uint32_t)P.maxGap; const int maxDesert = P.maxDesert;
    for (int i = 0; i < nodeCount; i++) {
        GNode &ln = nodes[i];
        int lSQO = ln.SQO, lEQO = ln.EQO; uint32_t lSRO = ln.diag + lSQO, lERO = ln.diag + ln.EQO;
        for (int j = nodeCount - 1; j > i; j--) {
            GNode &rn = nodes[j];
            int rSQO = rn.SQO;
            if (rSQO == lSQO) break;
            uint32_t diagGap = absDiff(ln.diag, rn.diag);
            if (diagGap > maxGap) continue;
            uint32_t rSRO = rn.diag + rSQO;
            if (lSRO >= rSRO) continue;
            int desert = (int)std::min(gapI(lEQO, rSQO), gapU(lERO, rSRO));
            if (desert > maxDesert) continue;
            int maxOverlap = (int)std::max(ovlI(lEQO, rSQO), ovlU(lERO, rSRO));
            int newbases = rn.nodeLength - maxOverlap;
            if (newbases < 1) continue;
            int newScore = ln.bestScore + newbases * P.MScore + gapCost((int)diagGap, P);
            if (rn.bestScore > newScore) continue;
            else if (rn.bestScore == newScore) {
                int pb = rn.bestPrev;
                if (pb < 0) continue;
                const GNode &pn = nodes[pb];
                int diagCompare = (int)(absDiff(ln.diag, rn.diag) - absDiff(pn.diag, rn.diag));
                if (diagCompare > 0) continue;
                else if (diagCompare == 0) {
                    int gapCompare = (int)(gapI(ln.EQO, rn.SQO) - gapI(pn.EQO, rn.SQO));
                    if (gapCompare > 0) continue;
                    else if (gapCompare == 0 && ln.pathSQO <= pn.pathSQO) continue;
                }
            }
            rn.bestScore = (int16_t)newScore; rn.bestPrev = i; rn.pathLength = (int16_t)(ln.pathLength + 1); rn.pathSQO = ln.pathSQO;
        }
        if (ln.bestScore < bestScore) continue;
        bool better = ln.bestScore > bestScore;
        if (!better) {                                                 // differentiateEqualFragNodesDuringBacktrack :88-94
            const GNode &bn = nodes[bestNode];
            better = (ln.EQO != bn.EQO) ? (ln.EQO < bn.EQO) : (ln.pathSQO > bn.pathSQO);
        }
        if (better) { bestNode = i; bestScore = ln.bestScore; }
    }
    for (int cur = bestNode; cur >= 0; cur = nodes[cur].bestPrev) insertFragment(clump, frags[nodes[cur].fragIdx]);   // :134-139
    if (clump.matchedBases < P.minMatch) { clump.list = FList(); clump.matchedBases = 0; }                           // :142-143
    else cleanUpClump(clump, P);
}
}  // namespace

void chainFragments(const ygpu_params &P, std::vector<Frag> &frags, int qlen, bool reversed,
                    std::vector<ChainClump> &out, ygpu_counters *ctr)
{
    const int fragCount = (int)frags.size();
    if (fragCount == 0) return;
    std::vector<uint8_t> used(fragCount, 0), coverage(qlen + 1, 0);
    std::vector<GNode> nodes;
    auto emit = [&](BuildClump &bc) {
        ChainClump cc; cc.matchedBases = bc.matchedBases; cc.reversed = reversed;
        for (int id = bc.list.head; id >= 0; id = bc.list.nx[id]) cc.frags.push_back(bc.list.f[id]);
        out.push_back(std::move(cc)); if (ctr) ctr->clumps_formed++;
    };
    int next = 0;
    while (next < fragCount) {                                          // QueryMatch.c:231-298
        int startF = next, endF = fragCount - 1;
        { uint32_t curDiag = fragDiag(frags[startF]);                   // findAlignableFragsForw :146-158
          for (int i = startF; i <= fragCount - 1; i++) { uint32_t d = fragDiag(frags[i]); if (absDiff(curDiag, d) > (uint32_t)P.maxGap) { endF = i - 1; break; } curDiag = d; } }
        if (ctr) ctr->regions++;
        if (endF == startF) {
            if ((int)frags[startF].refLen >= P.minMatch) { BuildClump bc; bc.matchedBases = frags[startF].refLen; bc.list.pushFront(frags[startF]); emit(bc); }
        } else {                                                        // processFragmentRangeUsingGraph, GraphPath.cpp:272-292
            std::fill(coverage.begin(), coverage.end(), 0);
            for (;;) {
                BuildClump bc;
                buildBestClump(P, frags, used, startF, endF, bc, nodes);
                if (bc.list.empty()) break;
                const Frag &first = bc.list.f[bc.list.head], &last = bc.list.f[bc.list.tail];
                uint16_t cSQO = first.sqo, cLen = (uint16_t)(1 + last.eqo - first.sqo);
                for (int k = 0; k < cLen && cSQO + k <= qlen; k++) coverage[cSQO + k] = 1;
                int minLeft = P.minNonOverlap - 1;                      // eliminateFragments / checkStartEndCoverage, QueryMatch.c:177-215
                for (int i = startF; i <= endF; i++) {
                    if (used[i]) continue;
                    int SQO = frags[i].sqo, EQO = frags[i].eqo; bool keep;
                    if (EQO - SQO < minLeft) keep = false;
                    else {
                        bool a = true; for (int k = SQO; k <= SQO + minLeft; k++) if (coverage[k]) { a = false; break; }
                        if (a) keep = true;
                        else { bool b = true; for (int k = EQO - minLeft; k <= EQO; k++) if (coverage[k]) { b = false; break; } keep = b; }
                    }
                    if (!keep) used[i] = 1;
                }
                emit(bc);
            }
        }
        next = endF + 1;
    }
}

// =======================================================================================================
// A6  findAffineGapScore<banded,extension,reverse,XCutoff>, SW.cpp:798-1208
// =======================================================================================================
namespace {
int affineGap(const ygpu_params &P, bool banded, bool extension, bool reverse, const uint8_t *q, int qBase, uint16_t qLen,
              const uint8_t *rStr, uint16_t rLen, OpList &list, uint16_t *addedQLen, uint16_t *addedRLen, uint64_t *rowsOut, uint64_t *cellsOut)
{
    const int GOCost = P.GOCost, GECost = P.GECost, RCost = P.RCost, MScore = P.MScore;
    int bandwidth = 0, leftBandwidth = 0, rightBandwidth = 0, arrWidth, maxi = 0, maxj = 0;
    const int arrHeight = qLen + 1;
    if (banded) {
        if (extension) { bandwidth = 2 * P.bandWidth; leftBandwidth = rightBandwidth = bandwidth; maxi = maxj = 0; }
        else {
            bandwidth = P.bandWidth;
            if (rLen > qLen) { rightBandwidth = bandwidth + (rLen - qLen); leftBandwidth = bandwidth; }
            else { leftBandwidth = bandwidth + (qLen - rLen); rightBandwidth = bandwidth; }
            maxi = qLen; maxj = rightBandwidth;
        }
        arrWidth = leftBandwidth + rightBandwidth + 1;
    } else { arrWidth = rLen + 1; maxi = arrHeight - 1; maxj = arrWidth - 1; }

    std::vector<char> EO((size_t)arrWidth * arrHeight + arrWidth + 2, 0);
    std::vector<int>  ID((size_t)arrWidth * arrHeight + arrWidth + 2, 0);
    std::vector<int>  PVs(arrWidth + 3, 0), PF(arrWidth + 3, 0), PI(arrWidth + 3, 0);
    int *PV = PVs.data() + 1; PV[-1] = DPWorstScore;                  // DPInit hack, SW.cpp:385-388
    EO[0] = 'U'; ID[0] = 0;                                           // DPInit, SW.cpp:382-383
    int PVCol = DPWorstScore;

    int startInit, endInit;
    if (banded) { startInit = leftBandwidth + 1; EO[leftBandwidth] = 'U'; ID[leftBandwidth] = 0; PF[arrWidth] = PV[arrWidth] = DPWorstScore; PI[arrWidth] = 0; }
    else startInit = 1;
    endInit = arrWidth;
    int deleteCount = 1;
    for (int j = startInit; j < endInit; j++) { EO[j] = 'D'; ID[j] = deleteCount; PV[j] = -(GOCost + deleteCount * GECost); deleteCount++; PF[j] = DPWorstScore; PI[j] = 0; }
    PF[startInit - 1] = 0; PI[startInit - 1] = 0; PV[startInit - 1] = 0;
    endInit = banded ? leftBandwidth : arrHeight - 1;
    for (int i = 1; i <= endInit && i < arrHeight; i++) { int lo = banded ? leftBandwidth - i : 0; EO[(size_t)i * arrWidth + lo] = 'I'; ID[(size_t)i * arrWidth + lo] = i; }

    int maxScore = 0, Cutoff = 0;
    if (extension) { maxScore = DPWorstScore; Cutoff = P.XCutoff; }
    int startCol = 0, endCol = 0;
    if (!banded) { startCol = 1; endCol = arrWidth - 1; }
    size_t rowOffset = 0; int V = 0; uint64_t rows = 0, cells = 0;
    for (int i = 1; i < arrHeight; i++) {
        rowOffset += arrWidth;
        int PDCol = 0, PECol = DPWorstScore;
        if (banded) {
            startCol = leftBandwidth + 1 - i;
            if (startCol <= 0) { startCol = 0; PVCol = DPWorstScore; }
            else PVCol = PV[startCol - 1] = -(GOCost + i * GECost);
            endCol = std::min(leftBandwidth + rLen - i, arrWidth - 1);
        } else PVCol = -(GOCost + i * GECost);
        int rowMaxScore = DPWorstScore;
        uint8_t qChar = reverse ? q[qBase + 1 - i] : q[qBase + i - 1];
        int rRowStartOff = banded ? i - leftBandwidth - 1 : 0;
        rows++;
        for (int j = startCol; j <= endCol; j++) {
            int RMOff = banded ? j : j - 1, IOff = RMOff + 1;
            char opcode;
            V = PV[RMOff];
            uint8_t rChar = banded ? rStr[rRowStartOff + j] : rStr[j - 1];
            if (qChar == rChar) { V += MScore; opcode = 'M'; } else { V -= RCost; opcode = 'R'; }
            int CE = PECol - GECost, NE = PVCol - (GOCost + GECost);
            if (CE >= NE && (PDCol + 1) <= P.maxIntron) { PECol = CE; PDCol = PDCol + 1; } else { PECol = NE; PDCol = 1; }
            if (extension ? (PECol >= V) : (PECol > V)) { V = PECol; opcode = 'D'; ID[rowOffset + j] = PDCol; }
            int F, I, CF = PF[IOff] - GECost, NF = PV[IOff] - (GOCost + GECost);
            if (CF >= NF && (PI[IOff] + 1) <= P.maxGap) { F = CF; I = PI[IOff] + 1; } else { F = NF; I = 1; }
            if (extension ? (F >= V) : (F > V)) { V = F; opcode = 'I'; ID[rowOffset + j] = I; }
            PF[j] = F; PI[j] = I;
            EO[rowOffset + j] = opcode;
            if (extension && V > rowMaxScore) rowMaxScore = V;
            if (extension && V > maxScore) { maxScore = V; maxi = i; maxj = j; }
            if (banded) PV[j] = V; else PV[j - 1] = PVCol;
            PVCol = V; cells++;
        }
        if (extension && rowMaxScore < (maxScore - Cutoff)) break;
        if (!banded) PV[endCol] = V;
    }
    if (rowsOut) *rowsOut += rows; if (cellsOut) *cellsOut += cells;
    int retval = extension ? maxScore : V;
    if (extension && retval <= 0) return 0;
    list.clear();
    if (extension) { *addedQLen = (uint16_t)maxi; *addedRLen = (uint16_t)(maxi + (maxj - bandwidth)); }
    // backtrack, SW.cpp:1138-1195
    int x = maxj; long rowBase = (long)maxi * arrWidth;
    char prev = EO[rowBase + x]; int opLen = 0; char code; OpList rev;
    while ((code = EO[rowBase + x]) != 'U') {
        int len = ID[rowBase + x];
        if (banded) { if (code == 'D') x -= len; else if (code == 'I') { x += len; rowBase -= (long)len * arrWidth; } else { rowBase -= arrWidth; len = 1; } }
        else        { if (code == 'D') x -= len; else if (code == 'I') { rowBase -= (long)len * arrWidth; } else { x -= 1; rowBase -= arrWidth; len = 1; } }
        if (prev != code) { rev.push_back({(uint16_t)opLen, prev}); prev = code; opLen = len; } else opLen += len;
    }
    rev.push_back({(uint16_t)opLen, prev});
    // forward: each op is added to the FRONT (so the list is the reverse of emission order); reverse: to the BACK.
    if (reverse) list = rev; else list.assign(rev.rbegin(), rev.rend());
    return retval;
}

void decompress(const Index &ix, bool reverse, uint32_t start, uint32_t rLen, std::vector<uint8_t> &out)   // SW.cpp:444-456
{ out.resize(rLen); for (uint32_t i = 0; i < rLen; i++) out[i] = ref4(ix.bases, reverse ? start - i : start + i); }
}  // namespace

DPOut dpFull(const Index &ix, const ygpu_params &P, const uint8_t *q, uint32_t rOff, uint16_t rLen, uint16_t qOff, uint16_t qLen)
{ DPOut o; std::vector<uint8_t> r; decompress(ix, false, rOff, rLen, r); o.score = affineGap(P, false, false, false, q, qOff, qLen, r.data(), rLen, o.ops, nullptr, nullptr, &o.rows, &o.cells); return o; }   // SW.cpp:462-468
DPOut dpBanded(const Index &ix, const ygpu_params &P, const uint8_t *q, uint32_t rOff, uint16_t rLen, uint16_t qOff, uint16_t qLen)
{ DPOut o; std::vector<uint8_t> r; decompress(ix, false, rOff, rLen, r); o.score = affineGap(P, true, false, false, q, qOff, qLen, r.data(), rLen, o.ops, nullptr, nullptr, &o.rows, &o.cells); return o; }    // SW.cpp:470-475

DPOut dpExtend(const Index &ix, const ygpu_params &P, const uint8_t *q, bool reverse, uint32_t rOff, uint16_t qOff, uint16_t qLenArg)
{                                                                       // findAGSExtension, SW.cpp:479-533
    DPOut o; int qLen = qLenArg;
    if (qLen <= 0) return o;
    int bandwidth = 2 * P.bandWidth; uint32_t rLen = (uint32_t)(qLen + bandwidth);
    if (reverse && rLen > rOff) { rLen = rOff + 1; qLen = (int)rLen - bandwidth; if (qLen <= 0) return o; }
    if (!reverse && (rOff + rLen) > ix.maxROff) { rLen = ix.maxROff - rOff; qLen = (int)rLen - bandwidth; if (qLen <= 0) return o; }
    std::vector<uint8_t> r; decompress(ix, reverse, rOff, rLen, r);
    int AGS = affineGap(P, true, true, reverse, q, qOff, (uint16_t)qLen, r.data(), (uint16_t)rLen, o.ops, &o.addedQ, &o.addedR, &o.rows, &o.cells);
    if (AGS <= 0) { o.score = 0; o.ops.clear(); o.addedQ = o.addedR = 0; return o; }   // (*addedQLen may be stale in the reference; callers ignore it when score <= 0)
    o.score = AGS; return o;
}

// =======================================================================================================
// A5 / A7 / A8 / A10
// =======================================================================================================
namespace {
struct WClump { Frag frag; int score = 0; OpList ops; uint8_t status = 0; uint16_t totScore = 0, totLength = 0, matched = 0, mismatched = 0, gapBases = 0; };

struct Aligner {
    const Index &ix; const ygpu_params &P; const uint8_t *fwd, *rev; int qlen; ygpu_counters *ctr;
    std::vector<WClump> pushed;                               // the new QS->clumps in push order (head = last)

    const uint8_t *qbuf(const WClump &c) const { return (c.status & stReversed) ? rev : fwd; }

    int extFwdPerfect(Frag &f, const uint8_t *q, int len)     // AlignExtFrag.cpp:30-38
    { uint16_t qOff = (uint16_t)(f.eqo + 1); uint32_t rOff = fragERO(f) + 1; int count = 0;
      while (count < len && q[qOff + count] == ref4(ix.bases, rOff + count)) count++;
      if (ctr) { ctr->perfect_ext_bases += count; ctr->ref_bases_touched += count + (count < len); }
      if (count > 0) { f.eqo = (uint16_t)(f.eqo + count); f.refLen = (uint16_t)(f.refLen + count); } return count; }
    int extBackPerfect(Frag &f, const uint8_t *q, int len)    // AlignExtFrag.cpp:40-48
    { uint16_t qOff = (uint16_t)(f.sqo - 1); uint32_t rOff = f.sro - 1; int count = 0;
      while (count < len && q[qOff - count] == ref4(ix.bases, rOff - count)) count++;
      if (ctr) { ctr->perfect_ext_bases += count; ctr->ref_bases_touched += count + (count < len); }
      if (count > 0) { f.sqo = (uint16_t)(f.sqo - count); f.sro -= count; f.refLen = (uint16_t)(f.refLen + count); } return count; }

    void tally(const DPOut &o, bool ext, uint32_t touched)
    { if (!ctr) return; if (ext) { ctr->dp_ext_calls++; ctr->dp_ext_rows += o.rows; ctr->dp_ext_cells += o.cells; } else { ctr->dp_gap_calls++; ctr->dp_gap_rows += o.rows; ctr->dp_gap_cells += o.cells; } ctr->ref_bases_touched += touched; }

    // findAGSForwardExtensionCarefully, SW.cpp:553-669
    int fwdCarefully(const uint8_t *q, uint32_t rOff, uint16_t qOff, uint16_t qLen, OpList &list, int score, uint16_t *aQ, uint16_t *aR)
    {
        DPOut o = dpExtend(ix, P, q, false, rOff, qOff, qLen); tally(o, true, (uint32_t)o.rows + 4 * P.bandWidth + 1);
        *aQ = o.addedQ; *aR = o.addedR;
        int initAGS = o.score; if (initAGS <= 0) return 0;
        int QLen = 0, RLen = 0, AGS = score, maxAGS = score, maxItem = -1, maxQLen = 0, maxRLen = 0;
        OpList &t = o.ops;
        for (int k = 0; k < (int)t.size(); k++) {
            char op = t[k].code; int len = t[k].len;
            if (op == 'M') { QLen += len; RLen += len; AGS += P.MScore * len; }
            else if (op == 'R') { QLen += len; RLen += len; AGS -= P.RCost * len; }
            else if (op == 'I') { QLen += len; AGS -= (P.GOCost + P.GECost * len); }
            else if (op == 'D') { RLen += len; AGS -= (P.GOCost + P.GECost * len); }
            if (AGS > maxAGS) { maxAGS = AGS; maxQLen = QLen; maxRLen = RLen; maxItem = k; }
            else if (AGS <= 0) {
                if (maxAGS <= score) { *aQ = 0; *aR = 0; return 0; }
                t.resize(maxItem + 1);                                  // splitEditOpListAfter + drop the tail
                *aQ = (uint16_t)maxQLen; *aR = (uint16_t)maxRLen; initAGS = maxAGS - score; break;
            }
        }
        mergeToBack(list, t);
        return initAGS;
    }
    // findAGSBackwardExtensionCarefully, SW.cpp:671-788
    int backCarefully(const uint8_t *q, uint32_t rOff, uint16_t qOff, uint16_t qLen, OpList &list, int score, uint16_t *aQ, uint16_t *aR)
    {
        DPOut o = dpExtend(ix, P, q, true, rOff, qOff, qLen); tally(o, true, (uint32_t)o.rows + 4 * P.bandWidth + 1);
        *aQ = o.addedQ; *aR = o.addedR;
        int initAGS = o.score; if (initAGS <= 0) return 0;
        int QLen = 0, RLen = 0, AGS = 0, maxAGS = 0, startItem = -1;
        OpList &t = o.ops;
        for (int k = 0; k < (int)t.size(); k++) {
            char op = t[k].code; int len = t[k].len;
            if (op == 'M') { QLen += len; RLen += len; AGS += P.MScore * len; }
            else if (op == 'R') { QLen += len; RLen += len; AGS -= P.RCost * len; }
            else if (op == 'I') { QLen += len; AGS -= (P.GOCost + P.GECost * len); }
            else if (op == 'D') { RLen += len; AGS -= (P.GOCost + P.GECost * len); }
            if (AGS <= 0) { AGS = 0; maxAGS = 0; QLen = 0; RLen = 0; startItem = k; }
            if (AGS > maxAGS) maxAGS = AGS;
        }
        if (AGS <= 0 || maxAGS >= AGS + score) { *aQ = 0; *aR = 0; return 0; }
        if (startItem >= 0) { OpList wanted(t.begin() + startItem + 1, t.end()); mergeToFront(list, wanted); }
        else mergeToFront(list, t);
        *aQ = (uint16_t)QLen; *aR = (uint16_t)RLen;
        return AGS;
    }

    // extendClumpForwardReverseTemplated<goBack,goForw,goCarefully>, AlignExtFrag.cpp:64-144
    void extendClump(WClump &c, bool goBack, bool goForw, bool carefully)
    {
        Frag &frag = c.frag; OpList &list = c.ops; const uint8_t *q = qbuf(c);
        int score = c.score, backLen = 0, forwLen = 0;
        if (goBack) {
            backLen = (int)std::min<uint32_t>(frag.sqo, frag.sro);
            if (backLen > 0) { int m = extBackPerfect(frag, q, backLen); if (m > 0) { list.front().len = (uint16_t)(list.front().len + m); score += m * P.MScore; backLen -= m; } }
        }
        if (goForw) {
            uint16_t ql = (uint16_t)((qlen - 1) - frag.eqo); uint32_t rl = ix.maxROff - fragERO(frag);
            forwLen = (int)std::min<uint32_t>(ql, rl);
            if (forwLen > 0) { int m = extFwdPerfect(frag, q, forwLen); if (m > 0) { list.back().len = (uint16_t)(list.back().len + m); score += m * P.MScore; forwLen -= m; } }
        }
        uint16_t aQ = 0, aR = 0;
        if (goBack && backLen >= P.minExtLength) {
            int ns;
            if (carefully) ns = backCarefully(q, frag.sro - 1, (uint16_t)(frag.sqo - 1), (uint16_t)backLen, list, score, &aQ, &aR);
            else { DPOut o = dpExtend(ix, P, q, true, frag.sro - 1, (uint16_t)(frag.sqo - 1), (uint16_t)backLen); tally(o, true, (uint32_t)o.rows + 4 * P.bandWidth + 1);
                   ns = o.score; aQ = o.addedQ; aR = o.addedR; if (ns > 0) mergeToFront(list, o.ops); }
            if (ns > 0) { score += ns; frag.sqo = (uint16_t)(frag.sqo - aQ); frag.sro -= aR; frag.refLen = (uint16_t)(frag.refLen + aR); }
        }
        if (goForw && forwLen >= P.minExtLength) {
            int ns;
            if (carefully) ns = fwdCarefully(q, fragERO(frag) + 1, (uint16_t)(frag.eqo + 1), (uint16_t)forwLen, list, score, &aQ, &aR);
            else { DPOut o = dpExtend(ix, P, q, false, fragERO(frag) + 1, (uint16_t)(frag.eqo + 1), (uint16_t)forwLen); tally(o, true, (uint32_t)o.rows + 4 * P.bandWidth + 1);
                   ns = o.score; aQ = o.addedQ; aR = o.addedR; if (ns > 0) mergeToBack(list, o.ops); }
            if (ns > 0) { score += ns; frag.eqo = (uint16_t)(frag.eqo + aQ); frag.refLen = (uint16_t)(frag.refLen + aR); }
        }
        c.score = score;
    }

    // alignClump, AlignHelpers.c:205-272 (+ makeAndAlignSFragmentToFillGap, AlignExtFrag.cpp:164-234; collapseSFragments :274-300)
    void alignClump(ChainClump &cc, WClump &c)
    {
        c.status = cc.reversed ? stReversed : 0; c.matched = cc.matchedBases;
        const uint8_t *q = qbuf(c);
        std::vector<Frag> &F = cc.frags; const int n = (int)F.size();
        for (int k = 1; k < n; k++) {                               // perfect extensions towards each other :226-237
            Frag &f1 = F[k - 1], &f2 = F[k];
            int gap = (int)std::min(gapI(f1.eqo, f2.sqo), gapU(fragERO(f1), f2.sro));
            gap -= extBackPerfect(f2, q, gap);
            gap -= extFwdPerfect(f1, q, gap);
        }
        OpList list; int total = 0;
        for (int k = 0; k < n; k++) {
            int ql = fragQueryLen(F[k]);
            OpList m{{(uint16_t)ql, 'M'}}; total += P.MScore * ql; mergeToBack(list, m);
            if (k + 1 == n) break;
            const Frag &f1 = F[k], &f2 = F[k + 1];
            uint16_t qGap = (uint16_t)gapI(f1.eqo, f2.sqo), rGap = (uint16_t)gapU(fragERO(f1), f2.sro);
            if (qGap == 0 && rGap == 0) continue;
            uint16_t nsqo = (uint16_t)(f1.eqo + 1); uint32_t nsro = fragERO(f1) + 1;
            OpList g; int gs;
            if (qGap == 0) { g.push_back({rGap, 'D'}); gs = gapCost(rGap, P); }
            else if (rGap == 0) { g.push_back({qGap, 'I'}); gs = gapCost(qGap, P); }
            else if (rGap == 1 && qGap == 1) { g.push_back({1, 'R'}); gs = -P.RCost; }
            else {
                int lenDiff = std::abs((int)qGap - (int)rGap);
                DPOut o = (lenDiff + P.bandWidth * 2 + 1 < rGap) ? dpBanded(ix, P, q, nsro, rGap, nsqo, qGap) : dpFull(ix, P, q, nsro, rGap, nsqo, qGap);
                tally(o, false, rGap);
                g = o.ops; gs = o.score;
            }
            total += gs; mergeToBack(list, g);
        }
        c.frag = F[0]; c.frag.eqo = F[n - 1].eqo; c.frag.refLen = (uint16_t)(1 + fragERO(F[n - 1]) - c.frag.sro);
        c.score = total; c.ops.swap(list);
        extendClump(c, true, true, false);
        c.status |= stAligned;
    }

    // scoreClump, AlignHelpers.c:302-366
    void scoreClump(WClump &c)
    {
        if (c.status & stScored) return;
        int AGS = 0, maxAGS = 0, matches = 0, mismatches = 0, inserts = 0, deletes = 0;
        const int alignedScore = c.score; const int last = (int)c.ops.size() - 1;
        for (int k = 0; k <= last; k++) {
            char op = c.ops[k].code; int len = c.ops[k].len;
            if (op == 'M') { matches += len; AGS += P.MScore * len; }
            else if (op == 'R') { mismatches += len; AGS -= P.RCost * len; }
            else if (op == 'I') { inserts += len; AGS -= (P.GOCost + P.GECost * len); }
            else if (op == 'D') { deletes += len; AGS -= (P.GOCost + P.GECost * len); }
            if (AGS <= 0 || (AGS >= alignedScore && k != last)) { splitClump(c); return; }
            if (AGS > maxAGS) maxAGS = AGS;
        }
        if (matches >= P.minRawScore && maxAGS > AGS) { splitClump(c); return; }
        if (matches < P.minRawScore) return;
        c.matched = (uint16_t)matches; c.mismatched = (uint16_t)mismatches; c.gapBases = (uint16_t)(inserts + deletes);
        c.totLength = (uint16_t)(matches + mismatches + inserts + deletes); c.totScore = (uint16_t)AGS;
        double percent = (double)c.matched / c.totLength;
        if (percent < P.minIdentity) return;                          // float promoted to double, AlignHelpers.c:359-360
        c.status |= stScored;
    }
    void splitClump(WClump &c) { if (ctr) ctr->splits++; splitHelper(c, c.frag.sqo, c.frag.eqo); }   // AlignHelpers.c:561-579

    static bool hasMaxMatch(const OpList &l, int min) { for (auto &o : l) if (o.code == 'M' && o.len >= min) return true; return false; }   // SW.cpp:1215-1222

    // splitClumpHelper, AlignHelpers.c:374-557
    void splitHelper(WClump &c, int wSQO, int wEQO)
    {
        OpList &list = c.ops; const Frag cur = c.frag;
        uint16_t sQO = 0, eQO = 0; uint32_t sRO = 0, eRO = 0;
        int matches = 0, mismatches = 0, inserts = 0, deletes = 0, AGS = 0, maxAGS = -10000, maxItem = -1, minItem = -1;
        for (int k = 0; k < (int)list.size(); k++) {
            char op = list[k].code; int len = list[k].len, ns = 0;
            if (op == 'M') { matches += len; ns = P.MScore * len; }
            else if (op == 'R') { mismatches += len; ns = -(P.RCost * len); }
            else if (op == 'I') { inserts += len; ns = -(P.GOCost + P.GECost * len); }
            else if (op == 'D') { deletes += len; ns = -(P.GOCost + P.GECost * len); }
            AGS += ns; if (AGS < 0) AGS = 0;
            if (AGS > maxAGS) { maxAGS = AGS; maxItem = k; eQO = (uint16_t)(cur.sqo + matches + mismatches + inserts - 1); eRO = cur.sro + matches + mismatches + deletes - 1; }
        }
        AGS = maxAGS; matches = mismatches = inserts = deletes = 0; int maxMatch = 0;
        for (int k = maxItem; k >= 0; k--) {
            char op = list[k].code; int len = list[k].len;
            if (op == 'M') { matches += len; AGS -= P.MScore * len; if (len > maxMatch) maxMatch = len; }
            else if (op == 'R') { mismatches += len; AGS += P.RCost * len; }
            else if (op == 'I') { inserts += len; AGS += (P.GOCost + P.GECost * len); }
            else if (op == 'D') { deletes += len; AGS += (P.GOCost + P.GECost * len); }
            if (AGS <= 0) { minItem = k; sQO = (uint16_t)(eQO - (matches + mismatches + inserts - 1)); sRO = eRO - (matches + mismatches + deletes - 1); break; }
        }
        if (maxMatch < P.wordLen) return;
        if (minItem < 0) return;                                      // cannot happen; the reference would crash
        if (minItem != 0) {                                           // head remainder :463-495
            WClump nc; nc.status = c.status & stReversed;
            nc.ops.assign(list.begin(), list.begin() + minItem); list.erase(list.begin(), list.begin() + minItem);
            maxItem -= minItem; minItem = 0;
            if (hasMaxMatch(nc.ops, P.wordLen)) {
                nc.frag.sqo = cur.sqo; nc.frag.eqo = (uint16_t)(sQO - 1); nc.frag.sro = cur.sro; nc.frag.refLen = (uint16_t)(1 + (sRO - 1) - cur.sro);
                splitHelper(nc, wSQO, wEQO);
            }
            if (nc.status & stScored) { nc.status |= stSplit | stAligned; pushed.push_back(std::move(nc)); }
        }
        if (maxItem != (int)list.size() - 1) {                        // tail remainder :500-531
            WClump nc; nc.status = c.status & stReversed;
            nc.ops.assign(list.begin() + maxItem + 1, list.end()); list.resize(maxItem + 1);
            if (hasMaxMatch(nc.ops, P.wordLen)) {
                nc.frag.sqo = (uint16_t)(eQO + 1); nc.frag.eqo = cur.eqo; nc.frag.sro = eRO + 1; nc.frag.refLen = (uint16_t)(1 + fragERO(cur) - (eRO + 1));
                splitHelper(nc, wSQO, wEQO);
            }
            if (nc.status & stScored) { nc.status |= stSplit | stAligned; pushed.push_back(std::move(nc)); }
        }
        c.frag.sqo = sQO; c.frag.eqo = eQO; c.frag.sro = sRO; c.frag.refLen = (uint16_t)(1 + eRO - sRO); c.score = maxAGS;
        bool goBack = (sQO != wSQO), goForw = (eQO != wEQO);
        if (goBack && goForw) extendClump(c, true, true, true);       // extendClumpForwardReverseCarefully, AlignExtFrag.cpp:151-156
        else if (goBack) extendClump(c, true, false, true);
        else extendClump(c, false, true, true);                       // sic: also when neither side was cut
        c.status |= stSplit;
        scoreClump(c);
    }
};
}  // namespace

void alignScoreClumps(const Index &ix, const ygpu_params &P, const uint8_t *fwd, const uint8_t *rev, int qlen,
                      std::vector<ChainClump> &created, std::vector<ScoredClump> &out, ygpu_counters *ctr)
{
    Aligner A{ix, P, fwd, rev, qlen, ctr, {}};
    // postProcessClumps walks QS->clumps head->tail = last created first, QueryMatch.c:306-331
    for (int k = (int)created.size() - 1; k >= 0; k--) {
        WClump c; A.alignClump(created[k], c); A.scoreClump(c);
        if (c.status & stScored) A.pushed.push_back(std::move(c));
    }
    out.clear();
    for (int k = (int)A.pushed.size() - 1; k >= 0; k--) {             // pushes go to the head
        WClump &w = A.pushed[k]; ScoredClump s;
        s.frag = w.frag; s.ops.swap(w.ops); s.totScore = w.totScore; s.totLength = w.totLength; s.matched = w.matched;
        s.mismatched = w.mismatched; s.gapBases = w.gapBases; s.status = w.status; out.push_back(std::move(s));
        if (ctr) { ctr->clumps_scored++; ctr->ops_out += out.back().ops.size(); }
    }
}

void processRead(const Index &ix, const ygpu_params &P, const uint8_t *fwd, int qlen, std::vector<ScoredClump> &out, ygpu_counters *ctr)
{
    std::vector<uint8_t> rev(qlen);
    for (int k = 0; k < qlen; k++) rev[k] = COMP4[fwd[qlen - 1 - k] & 0xF];      // Query.c:161-167
    std::vector<ChainClump> created; std::vector<Frag> frags;
    for (int strand = 0; strand < 2; strand++) {                                 // Query.c:342
        const uint8_t *codes = strand ? rev.data() : fwd;
        seedJoin(ix, P, codes, qlen, frags, ctr);
        chainFragments(P, frags, qlen, strand != 0, created, ctr);
    }
    alignScoreClumps(ix, P, fwd, rev.data(), qlen, created, out, ctr);
}
}  // namespace yoracle

// ===========================================================================================================
// C entry points
// ===========================================================================================================
using namespace yoracle;
static Index mkIndex(const ygpu_index_view *v) { Index ix; ix.bases = v->bases; ix.maxROff = v->maxROff; ix.SO = v->startingOffs; ix.ROA = v->ROA; ix.totalMatches = v->totalMatches; return ix; }
static void addCounters(ygpu_counters &a, const ygpu_counters &b) { uint64_t *x = (uint64_t *)&a; const uint64_t *y = (const uint64_t *)&b; for (size_t i = 0; i < sizeof(ygpu_counters) / 8; i++) x[i] += y[i]; }

extern "C" int yoracle_run(const ygpu_index_view *v, const ygpu_params *P, const ygpu_read_batch *b, int threads, yoracle_result *out)
{
    Index ix = mkIndex(v); const uint32_t n = b->n_reads;
    std::vector<std::vector<ScoredClump>> res(n);
    if (threads < 1) threads = 1;
    std::vector<ygpu_counters> ctrs(threads); memset(ctrs.data(), 0, sizeof(ygpu_counters) * threads);
    std::atomic<uint32_t> next(0);
    auto work = [&](int t) { for (;;) { uint32_t i = next.fetch_add(1); if (i >= n) break; processRead(ix, *P, b->codes + b->offsets[i], (int)(b->offsets[i + 1] - b->offsets[i]), res[i], &ctrs[t]); } };
    std::vector<std::thread> th; for (int t = 1; t < threads; t++) th.emplace_back(work, t); work(0); for (auto &x : th) x.join();
    memset(out, 0, sizeof *out); out->n_reads = n;
    for (int t = 0; t < threads; t++) addCounters(out->counters, ctrs[t]);
    uint64_t nc = 0, no = 0; for (auto &r : res) { nc += r.size(); for (auto &c : r) no += c.ops.size(); }
    out->clump_start = (uint32_t *)malloc(sizeof(uint32_t) * (n + 1)); out->clumps = (ygpu_clump *)malloc(sizeof(ygpu_clump) * (nc + 1)); out->ops = (uint32_t *)malloc(sizeof(uint32_t) * (no + 1));
    out->n_clumps = nc; out->n_ops = no; uint64_t ci = 0, oi = 0;
    for (uint32_t i = 0; i < n; i++) {
        out->clump_start[i] = (uint32_t)ci;
        for (auto &c : res[i]) {
            ygpu_clump &g = out->clumps[ci++]; memset(&g, 0, sizeof g);
            g.sro = c.frag.sro; g.sqo = c.frag.sqo; g.eqo = c.frag.eqo; g.refLen = c.frag.refLen; g.totScore = c.totScore; g.totLength = c.totLength;
            g.matchedBases = c.matched; g.mismatchedBases = c.mismatched; g.gapBases = c.gapBases; g.status = c.status; g.op_start = (uint32_t)oi; g.n_ops = (uint32_t)c.ops.size();
            for (auto &o : c.ops) out->ops[oi++] = YGPU_OP_MAKE(o.code, o.len);
        }
    }
    out->clump_start[n] = (uint32_t)ci;
    return 0;
}
extern "C" void yoracle_free_result(yoracle_result *r) { free(r->clump_start); free(r->clumps); free(r->ops); memset(r, 0, sizeof *r); }
extern "C" void yoracle_free(void *p) { free(p); }

static void revcompCodes(const uint8_t *fwd, int qlen, std::vector<uint8_t> &rev) { rev.resize(qlen); for (int k = 0; k < qlen; k++) rev[k] = COMP4[fwd[qlen - 1 - k] & 0xF]; }

extern "C" int yoracle_seed_join(const ygpu_index_view *v, const ygpu_params *P, const ygpu_read_batch *b, ygpu_fragment **frags, uint64_t *n)
{
    Index ix = mkIndex(v); std::vector<ygpu_fragment> all; std::vector<Frag> fr; std::vector<uint8_t> rev;
    for (uint32_t i = 0; i < b->n_reads; i++) {
        const uint8_t *fwd = b->codes + b->offsets[i]; int qlen = (int)(b->offsets[i + 1] - b->offsets[i]); revcompCodes(fwd, qlen, rev);
        for (int s = 0; s < 2; s++) { seedJoin(ix, *P, s ? rev.data() : fwd, qlen, fr, nullptr); for (auto &f : fr) all.push_back({f.sro, f.sqo, f.eqo, f.refLen, 0, i * 2 + (uint32_t)s}); }
    }
    *n = all.size(); *frags = (ygpu_fragment *)malloc(sizeof(ygpu_fragment) * (all.size() + 1)); memcpy(*frags, all.data(), sizeof(ygpu_fragment) * all.size());
    return 0;
}
extern "C" int yoracle_chain(const ygpu_index_view *v, const ygpu_params *P, const ygpu_read_batch *b, ygpu_fragment **cf, uint32_t **cfs, uint32_t **crs, uint64_t *nclumps)
{
    Index ix = mkIndex(v); std::vector<ygpu_fragment> all; std::vector<uint32_t> starts, rs; std::vector<Frag> fr; std::vector<uint8_t> rev;
    for (uint32_t i = 0; i < b->n_reads; i++) {
        const uint8_t *fwd = b->codes + b->offsets[i]; int qlen = (int)(b->offsets[i + 1] - b->offsets[i]); revcompCodes(fwd, qlen, rev);
        for (int s = 0; s < 2; s++) {
            seedJoin(ix, *P, s ? rev.data() : fwd, qlen, fr, nullptr); std::vector<ChainClump> cl; chainFragments(*P, fr, qlen, s != 0, cl, nullptr);
            for (auto &c : cl) { starts.push_back((uint32_t)all.size()); rs.push_back(i * 2 + s); for (auto &f : c.frags) all.push_back({f.sro, f.sqo, f.eqo, f.refLen, 0, i * 2 + (uint32_t)s}); }
        }
    }
    starts.push_back((uint32_t)all.size()); *nclumps = rs.size();
    *cf = (ygpu_fragment *)malloc(sizeof(ygpu_fragment) * (all.size() + 1)); memcpy(*cf, all.data(), sizeof(ygpu_fragment) * all.size());
    *cfs = (uint32_t *)malloc(4 * starts.size()); memcpy(*cfs, starts.data(), 4 * starts.size());
    *crs = (uint32_t *)malloc(4 * (rs.size() + 1)); memcpy(*crs, rs.data(), 4 * rs.size());
    return 0;
}
extern "C" int yoracle_dp_batch(const ygpu_index_view *v, const ygpu_params *P, const ygpu_read_batch *b, const ygpu_dp_problem *pr, uint32_t n,
                                ygpu_dp_result **res, uint32_t **ops, uint64_t *n_ops)
{
    Index ix = mkIndex(v); std::vector<uint32_t> allops; *res = (ygpu_dp_result *)malloc(sizeof(ygpu_dp_result) * (n + 1)); std::vector<uint8_t> rev;
    for (uint32_t k = 0; k < n; k++) {
        const ygpu_dp_problem &p = pr[k]; const uint8_t *fwd = b->codes + b->offsets[p.read]; int qlen = (int)(b->offsets[p.read + 1] - b->offsets[p.read]);
        const uint8_t *q = fwd; if (p.strand) { revcompCodes(fwd, qlen, rev); q = rev.data(); }
        DPOut o;
        switch (p.mode) { case YGPU_DP_FULL: o = dpFull(ix, *P, q, p.rOff, p.rLen, p.qOff, p.qLen); break; case YGPU_DP_BANDED: o = dpBanded(ix, *P, q, p.rOff, p.rLen, p.qOff, p.qLen); break;
                          case YGPU_DP_EXT_FWD: o = dpExtend(ix, *P, q, false, p.rOff, p.qOff, p.qLen); break; default: o = dpExtend(ix, *P, q, true, p.rOff, p.qOff, p.qLen); }
        ygpu_dp_result &r = (*res)[k]; r.score = o.score; r.addedQLen = o.addedQ; r.addedRLen = o.addedR; r.op_start = (uint32_t)allops.size(); r.n_ops = (uint32_t)o.ops.size();
        for (auto &e : o.ops) allops.push_back(YGPU_OP_MAKE(e.code, e.len));
    }
    *n_ops = allops.size(); *ops = (uint32_t *)malloc(4 * (allops.size() + 1)); memcpy(*ops, allops.data(), 4 * allops.size());
    return 0;
}
