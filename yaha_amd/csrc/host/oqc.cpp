// oqc.cpp -- host post-filter: Optimal Query Coverage, filter-by-similarity, duplicate removal and mapping
// quality (reference GraphPath.cpp:294-1175).  Consumes the clump records the device hot path returns (QS->clumps
// order) and yields the clumps to print, in print order.  SURVEY.md 8(f)-1: stays on the host (tiny,
// pointer-chasing, double/log10 arithmetic) but has to be restated exactly -- including the per-read RNG that
// breaks sort ties -- for bit-exact SAM.
#include "yaha_host.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace yaha {
namespace {
#define WorstScore (-(0x7fffff00))
enum { stReversed = 0x01, stPrimary = 0x20 };

struct CNode {                                           // cGraphNode, GraphPath.cpp:299-324
    int bestPrev; int clump;                             // clump < 0 = dead
    int16_t bestScore, pathLength; uint32_t SRO, ERO; uint16_t SQO, EQO; int16_t nodeLength, nodeScore; uint16_t qLenInOQC; uint8_t reversed, seqNum;
};
struct PAttr { int alignedQueryLength, numOutputSecondaries; int16_t secondScore, thirdScore; };   // :526-533

struct Ctx {
    const Args &a; const ygpu_clump *cl; const uint32_t *ops; std::vector<CNode> nodes;
    const uint32_t *opsOf(int c) const { return ops + cl[c].op_start; }
    int nOps(int c) const { return (int)cl[c].n_ops; }

    int scoreForLength(int c, int length, bool forward) const           // calcScoreForLength, :705-732
    {
        const uint32_t *o = opsOf(c); int n = nOps(c), k = forward ? 0 : n - 1, QLen = 0, AGS = 0;
        while (k >= 0 && k < n && QLen < length) {
            char op = YGPU_OP_CODE(o[k]); int len = (int)YGPU_OP_LEN(o[k]);
            if (op == 'D') AGS -= (a.GOCost + a.GECost * len);
            else {
                if (QLen + len > length) len = length - QLen;
                QLen += len;
                if (op == 'M') AGS += a.MScore * len; else if (op == 'R') AGS -= a.RCost * len; else if (op == 'I') AGS -= (a.GOCost + a.GECost * len);
            }
            k += forward ? 1 : -1;
        }
        return AGS;
    }
    int accurateOverlapScore(int left, int right, int overlap, bool *rightBest) const   // :744-800
    {
        const CNode &rn = nodes[right];
        int rightScore = scoreForLength(rn.clump, overlap, !rn.reversed);
        int pathScore = 0, remaining = overlap, cur = left;
        for (;;) {
            const CNode &cn = nodes[cur];
            int q = std::min(remaining, (int)cn.qLenInOQC); remaining -= q;
            pathScore += scoreForLength(cn.clump, q, cn.reversed != 0);
            if (remaining <= 0) break;
            cur = cn.bestPrev;
        }
        if (pathScore > rightScore) { *rightBest = false; return rightScore; }
        *rightBest = true; return pathScore;
    }
    void cacheReverse(int left, int right, int overlap, bool rightBest)                 // cacehQlenInOQCPathReverse :802-826
    {
        CNode &rn = nodes[right];
        if (rightBest) {
            rn.qLenInOQC = (uint16_t)(1 + rn.EQO - rn.SQO);
            int remaining = overlap, cur = left;
            for (;;) { CNode &cn = nodes[cur]; int q = std::min(remaining, (int)cn.qLenInOQC); cn.qLenInOQC = (uint16_t)(cn.qLenInOQC - q); remaining -= q; if (remaining <= 0) break; cur = cn.bestPrev; }
        } else rn.qLenInOQC = (uint16_t)((1 + rn.EQO - rn.SQO) - overlap);
    }
    int cachePath(int right)                                                             // cacheQlenInOQCPath :841-867
    {
        CNode &rn = nodes[right]; int qLen = 1 + rn.EQO - rn.SQO;
        if (rn.bestPrev < 0) { rn.qLenInOQC = (uint16_t)qLen; return right; }
        int left = cachePath(rn.bestPrev);
        const CNode &ln = nodes[left];
        int overlap = ((int)ln.EQO >= (int)rn.SQO) ? ((int)ln.EQO - (int)rn.SQO) + 1 : 0;
        if (overlap > 0) { bool rb; accurateOverlapScore(left, right, overlap, &rb); cacheReverse(left, right, overlap, rb); }
        else rn.qLenInOQC = (uint16_t)qLen;
        return right;
    }
};

inline uint64_t compareKey(const CNode &n)                              // getCompareKey :377-380
{ return ((((uint64_t)n.SQO << 16) + (uint16_t)(-(int)(int16_t)n.EQO)) << 16) + (uint16_t)(-(int)n.nodeScore); }
inline bool nodeLess(const CNode &x, const CNode &y, RandState &rs)
{ uint64_t k1 = compareKey(x), k2 = compareKey(y); if (k1 == k2) return (randBits(rs) & 1) != 0; return k1 < k2; }
void quickSort(CNode *arr, int left, int right, RandState &rs)          // myQuickSortHelper :427-453
{
    if (left >= right) return;
    int pivot = (left + right) / 2; std::swap(arr[pivot], arr[right]);
    int store = left;
    for (int i = left; i < right; i++) if (nodeLess(arr[i], arr[right], rs)) { std::swap(arr[i], arr[store]); store++; }
    std::swap(arr[store], arr[right]);
    quickSort(arr, left, store - 1, rs); quickSort(arr, store + 1, right, rs);
}

struct DupElem { int64_t clump; uint32_t SRO; int score; };             // dupArrayElem :1099-1104 (16 bytes like the reference's)
int cmpDup(const void *p1, const void *p2)
{ const DupElem *a = (const DupElem *)p1, *b = (const DupElem *)p2; if (a->SRO > b->SRO) return 1; if (a->SRO < b->SRO) return -1; return b->score - a->score; }
}  // namespace

void postFilter(const Args &a, const Genome &g, const Read &r, const ygpu_clump *cl, uint32_t n, const uint32_t *ops,
                std::vector<OutClump> &out, int &primaryCount)
{
    out.clear();
    auto mk = [&](int c) { OutClump o; o.c = cl[c]; o.ops = ops + cl[c].op_start; o.status = cl[c].status; return o; };
    const int qlen = r.len();
    if (!a.OQC) {                                                       // postFilterRemoveDups :1127-1174
        if (n < 2) { for (uint32_t i = 0; i < n; i++) out.push_back(mk((int)i)); return; }
        std::vector<DupElem> d(n);
        for (uint32_t i = 0; i < n; i++) d[i] = {(int64_t)i, cl[i].sro, (int)cl[i].totScore};
        qsort(d.data(), n, sizeof(DupElem), cmpDup);                    // same libc qsort the reference calls
        std::vector<int> keep;
        for (uint32_t i = 0; i < n; i++) {
            if (d[i].clump < 0) continue;
            const ygpu_clump &c1 = cl[d[i].clump];
            for (uint32_t j = i + 1; j < n; j++) {
                if (d[i].SRO < d[j].SRO) break;
                if (d[j].clump < 0) continue;
                const ygpu_clump &c2 = cl[d[j].clump];
                if (c1.sro == c2.sro && c1.sqo == c2.sqo && c1.eqo == c2.eqo && (c1.sro + c1.refLen) == (c2.sro + c2.refLen) && ((c1.status ^ c2.status) & stReversed) == 0) d[j].clump = -1;
            }
            keep.push_back((int)d[i].clump);
        }
        for (int k = (int)keep.size() - 1; k >= 0; k--) out.push_back(mk(keep[k]));    // pushes go to the head
        return;
    }
    if (n < 1) return;
    if (n == 1) { OutClump o = mk(0); o.status |= stPrimary; o.mapQuality = 250; o.numSecondaries = 0; o.matchedPrimary = 1; primaryCount = 1; out.push_back(o); return; }   // :907-916

    static thread_local std::vector<CNode> tlNodes, tlPrim; static thread_local std::vector<PAttr> tlPA; static thread_local std::vector<OutClump> tlPush;   // scratch reused from read to read
    Ctx X{a, cl, ops, std::move(tlNodes)};
    struct GiveBack { Ctx &x; ~GiveBack() { tlNodes = std::move(x.nodes); } } giveBack{X};
    X.nodes.resize(n);
    for (uint32_t i = 0; i < n; i++) {                                  // initcGraphNode :342-363, list walked head->tail :929-934
        CNode &nd = X.nodes[i]; const ygpu_clump &c = cl[i]; bool rev = (c.status & stReversed) != 0;
        nd.bestPrev = -1; nd.pathLength = 1; nd.clump = (int)i;
        nd.bestScore = nd.nodeScore = (int16_t)(int)c.totScore; nd.nodeLength = (int16_t)c.totLength;
        nd.SQO = rev ? (uint16_t)((qlen - 1) - c.eqo) : c.sqo; nd.EQO = rev ? (uint16_t)((qlen - 1) - c.sqo) : c.eqo;
        nd.SRO = c.sro; nd.ERO = c.sro + c.refLen - 1; nd.reversed = rev; nd.qLenInOQC = (uint16_t)(1 + c.eqo - c.sqo);
        nd.seqNum = (uint8_t)g.findSeq(nd.SRO);
    }
    RandState rs; seedFromRead(r, rs);
    quickSort(X.nodes.data(), 0, (int)n - 1, rs);
    // deleteSubsumedDups :488-517
    int cnt = 0;
    {
        std::vector<CNode> &gn = X.nodes; const int nodeCount = (int)n;
        for (int i = 0; i < nodeCount; i++) {
            if (gn[i].clump < 0) continue;
            if (cnt != i) gn[cnt] = gn[i];
            cnt++;
            const CNode cur = gn[i]; int thr = cur.nodeScore / 8;
            for (int j = i + 1; j < nodeCount; j++) {
                CNode &nx = gn[j];
                if (nx.clump < 0) continue;
                if (nx.EQO > cur.EQO) break;
                bool subsumed = (cur.EQO > nx.EQO && nx.nodeScore < thr);
                bool dups = (cur.SRO == nx.SRO && cur.ERO == nx.ERO && cur.reversed == nx.reversed && cur.SQO == nx.SQO && cur.EQO == nx.EQO);
                if (subsumed || dups) nx.clump = -1;
            }
        }
    }
    const int curNodeCount = cnt;
    int bestScore = WorstScore, bestNode = -1, startj = 1;
    const int minNonOverlap = a.OQCMinNonOverlap, BPCost = a.BPCost, MBPL = a.maxBPLog;
    for (int i = 0; i < curNodeCount; i++) {                            // :973-1063
        X.cachePath(i);
        CNode &ln = X.nodes[i];
        int leftSQO = ln.SQO, leftEQO = ln.EQO; bool foundstartj = false;
        for (int j = startj; j < curNodeCount; j++) {
            CNode &rn = X.nodes[j];
            int rightSQO = rn.SQO;
            if ((rightSQO - leftSQO) >= minNonOverlap) {
                if (!foundstartj) { startj = j; foundstartj = true; }
                int rightEQO = rn.EQO;
                if ((rightEQO - leftEQO) >= minNonOverlap) {
                    int16_t newScore = (int16_t)(ln.bestScore + rn.nodeScore);
                    if (rn.bestScore > newScore) continue;
                    int BPP;
                    if (ln.seqNum == rn.seqNum) {
                        uint32_t distance;
                        if (ln.SRO > rn.ERO) distance = ln.SRO - rn.ERO; else if (rn.SRO > ln.ERO) distance = rn.SRO - ln.ERO; else distance = 0;
                        if (distance <= 10) BPP = BPCost;
                        else { double lg = log10((double)distance); if (lg > MBPL) lg = (double)MBPL; BPP = (int)(lg * BPCost + 0.5); }
                    } else BPP = MBPL * BPCost;
                    newScore = (int16_t)(newScore - BPP);
                    if (rn.bestScore > newScore) continue;
                    int overlap = (leftEQO >= rightSQO) ? (leftEQO - rightSQO) + 1 : 0;
                    bool rightBest = false;
                    if (overlap > 0) { newScore = (int16_t)(newScore - X.accurateOverlapScore(i, j, overlap, &rightBest)); if (rn.bestScore > newScore) continue; }
                    if (rn.bestScore < newScore || (rn.bestPrev >= 0 && ln.pathLength < X.nodes[rn.bestPrev].pathLength)) {
                        if (overlap > 0) { int ql = 1 + rn.EQO - rn.SQO; rn.qLenInOQC = (uint16_t)(rightBest ? ql : ql - overlap); }   // cacheQlenInRightNode :873-878
                        rn.bestScore = newScore; rn.bestPrev = i; rn.pathLength = (int16_t)(ln.pathLength + 1);
                    }
                }
            }
            if (!foundstartj) startj = curNodeCount;
        }
        if (ln.bestScore < bestScore) continue;
        if (ln.bestScore > bestScore || (bestNode >= 0 && ln.pathLength < X.nodes[bestNode].pathLength)) { bestNode = i; bestScore = ln.bestScore; }
    }
    // filterBySimilarity :571-692
    std::vector<CNode> &gn = X.nodes;
    const int primeCount = gn[bestNode].pathLength;
    std::vector<CNode> &primaries = tlPrim; std::vector<PAttr> &PA = tlPA; primaries.assign(primeCount, CNode()); PA.assign(primeCount, PAttr());
    std::vector<OutClump> &pushOrder = tlPush; pushOrder.clear();      // push-to-head order; reversed at the end
    {
        int pi = primeCount - 1;
        for (int p = bestNode; p >= 0; p = gn[p].bestPrev) {
            primaries[pi] = gn[p];
            PA[pi] = {1 + gn[p].EQO - gn[p].SQO, 0, 0, 0};
            OutClump o = mk(gn[p].clump); o.status |= stPrimary; o.matchedPrimary = (uint16_t)(pi + 1); pushOrder.push_back(o);
            int keepPrev = gn[p].bestPrev; gn[p].clump = -1; pi--; (void)keepPrev;
        }
    }
    const double targetOverlap = a.FBS_PSLength;
    for (int i = 0; i < curNodeCount; i++) {
        const CNode &cn = gn[i];
        if (cn.clump < 0) continue;
        int curSQO = cn.SQO, curEQO = cn.EQO, curQLen = 1 + curEQO - curSQO, maxOverlap = 0, maxIndex = 0;
        for (int k = 0; k < primeCount; k++) {
            int overlap = 1 + std::min(curEQO, (int)primaries[k].EQO) - std::max(curSQO, (int)primaries[k].SQO);
            if (overlap > maxOverlap) { maxOverlap = overlap; maxIndex = k; }
        }
        if (maxOverlap > 0) {
            PAttr &pa = PA[maxIndex];
            if (cn.nodeScore > pa.secondScore) { pa.thirdScore = pa.secondScore; pa.secondScore = cn.nodeScore; }      // memoPAsFromOverlappingNode :545-557
            else if (cn.nodeScore > pa.thirdScore) pa.thirdScore = cn.nodeScore;
            const CNode &pn = primaries[maxIndex];
            if (((double)cn.nodeScore) / pn.nodeScore >= a.FBS_PSScore) {
                int overlap = 1 + std::min(curEQO, (int)pn.EQO) - std::max(curSQO, (int)pn.SQO);
                int pathQLen = pa.alignedQueryLength; double overlapD = overlap;
                if (overlapD / curQLen >= targetOverlap && overlapD / pathQLen >= targetOverlap) {
                    pa.numOutputSecondaries += 1;
                    if (a.FBS) { OutClump o = mk(cn.clump); o.matchedPrimary = (uint16_t)(maxIndex + 1); pushOrder.push_back(o); continue; }
                }
            }
        }
    }
    primaryCount = primeCount;
    // calcMQfromPAs :559-569 -- primaries are pushOrder[0..primeCount) holding index primeCount-1 .. 0
    for (int k = 0; k < primeCount; k++) {
        OutClump &o = pushOrder[k]; int pi = primeCount - 1 - k; const PAttr &pa = PA[pi]; const double ts = (double)o.c.totScore;
        if (pa.secondScore == 0) o.mapQuality = 250;
        else {
            double ratio = std::max(ts - pa.secondScore, 0.0) / ts;
            ratio = ratio * (1.0 + std::max(ts - pa.thirdScore, 0.0) / o.c.totScore) / 2.0;
            o.mapQuality = (uint8_t)((250.0 * ratio) + 0.5);
        }
        o.numSecondaries = (uint16_t)pa.numOutputSecondaries;
    }
    for (int k = (int)pushOrder.size() - 1; k >= 0; k--) out.push_back(pushOrder[k]);
}
}  // namespace yaha
