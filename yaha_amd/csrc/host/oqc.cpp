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
    // calcScoreForLength (:705-732) walks a clump's edit list from one end until `length` query bases are covered; the graph loop asks it ~120 times a read
    // for the same few clumps.  Here every clump that is asked about gets (once) the running sums Q[m], S[m] = query bases and score after its first m
    // ops, untruncated; a call is then a binary search for the op the walk stops in plus that op's truncated share.  The walk from the other end reads
    // the same table backwards (Qb[m] = Q[n] - Q[n-m]).  Same arithmetic, same results: only the order of the additions differs (integers).
    std::vector<int> &pfxOff; std::vector<int> &pool;                    // per clump: offset of its table in pool (-1: not built); pool: Q[0..n] then S[0..n]
    const uint32_t *opsOf(int c) const { return ops + cl[c].op_start; }
    int nOps(int c) const { return (int)cl[c].n_ops; }
    inline int opScore(char op, int len) const
    { return op == 'M' ? a.MScore * len : op == 'R' ? -(a.RCost * len) : op == 'I' ? -(a.GOCost + a.GECost * len) : 0; }
    const int *table(int c)
    {
        int off = pfxOff[c];
        if (off < 0) {
            const uint32_t *o = opsOf(c); const int n = nOps(c);
            off = (int)pool.size(); pfxOff[c] = off; pool.resize(pool.size() + 2 * (size_t)(n + 1));
            int *Q = pool.data() + off, *S = Q + n + 1; int q = 0, sc = 0; Q[0] = 0; S[0] = 0;
            for (int k = 0; k < n; k++) {
                const char op = YGPU_OP_CODE(o[k]); const int len = (int)YGPU_OP_LEN(o[k]);
                if (op == 'D') sc -= (a.GOCost + a.GECost * len); else { q += len; sc += opScore(op, len); }
                Q[k + 1] = q; S[k + 1] = sc;
            }
        }
        return pool.data() + off;
    }
    int scoreForLength(int c, int length, bool forward)                 // calcScoreForLength, :705-732
    {
        if (length <= 0) return 0;
        const int n = nOps(c); if (n <= 0) return 0;
        const uint32_t *o = opsOf(c);
        const int *Q = table(c), *S = Q + n + 1;
        if (Q[n] < length) return S[n];                                  // the list ends first: every op counted in full
        // (measured: forcing these searches branch-free -- an AND with the comparison's mask -- made a call slower, 14.9 -> 18.2 us a read: the walks of one read
        // stop in similar places, the branches predict, and the dependent loads are the longer chain; the reference's own loop for the first four ops before
        // any table is touched: 14.9 -> 16.7 -- the overlaps that reach this function are long)
        if (forward) {
            const int *base = Q + 1; int len = n;                        // first m in 1..n with Q[m] >= length (exists: Q[n] >= length); op m-1 is the one the walk stops in
            while (len > 1) { const int half = len >> 1; base = (base[half - 1] < length) ? base + half : base; len -= half; }
            const int j = (int)(base - Q) - 1;
            return S[j] + opScore(YGPU_OP_CODE(o[j]), length - Q[j]);
        }
        const int X = Q[n] - length;                                     // last t in 0..n-1 with Q[t] <= X (Q[0] = 0 <= X < Q[n]); op t is the one the backward walk stops in
        const int *base = Q; int len = n;
        while (len > 1) { const int half = len >> 1; base = (base[half] <= X) ? base + half : base; len -= half; }
        const int t = (int)(base - Q);
        return (S[n] - S[t + 1]) + opScore(YGPU_OP_CODE(o[t]), length - (Q[n] - Q[t + 1]));
    }
    int accurateOverlapScore(int left, int right, int overlap, bool *rightBest)   // :744-800
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

// The sort (myQuickSortHelper :427-453 on getCompareKey :377-380, ties broken by the per-read generator) decides which of two equal-keyed clumps survives, so
// its comparisons must happen in the reference's order.  They depend on keys and positions only: the same routine on (key, clump) pairs -- 16 bytes to swap
// instead of a 40-byte node, keys computed once -- yields the reference's permutation and consumes the same random bits.
struct SortKey { uint64_t key; int clump; };
inline uint64_t compareKey(int SQO, int EQO, int score)
{ return ((((uint64_t)(uint16_t)SQO << 16) + (uint16_t)(-(int)(int16_t)(uint16_t)EQO)) << 16) + (uint16_t)(-(int)(int16_t)score); }
inline bool keyLess(const SortKey &x, const SortKey &y, RandState &rs)
{ if (x.key == y.key) return (randBits(rs) & 1) != 0; return x.key < y.key; }
void quickSort(SortKey *arr, int left, int right, RandState &rs)
{
    if (left >= right) return;
    int pivot = (left + right) / 2; std::swap(arr[pivot], arr[right]);
    int store = left;
    const uint64_t pk = arr[right].key;
    for (int i = left; i < right; i++) {                                // "if less: swap(arr[i], arr[store]), store++" written with masks: the outcome of a
        const uint64_t xk = arr[i].key, yk = arr[store].key; const uint32_t xc = (uint32_t)arr[i].clump, yc = (uint32_t)arr[store].clump;   // comparison is a coin toss to the branch predictor, ~650 of them a read
        bool less = xk < pk;
        if (__builtin_expect(xk == pk, 0)) less = (randBits(rs) & 1) != 0;
        const uint64_t m = (uint64_t)0 - (uint64_t)less; const uint64_t dk = (xk ^ yk) & m; const uint32_t dc = (xc ^ yc) & (uint32_t)m;
        arr[i].key = xk ^ dk; arr[i].clump = (int)(xc ^ dc); arr[store].key = yk ^ dk; arr[store].clump = (int)(yc ^ dc); store += (int)less;
    }
    std::swap(arr[store], arr[right]);
    quickSort(arr, left, store - 1, rs); quickSort(arr, store + 1, right, rs);
}

// Break point penalty of two alignments `distance` (> 10) reference bases apart on one sequence: (int)(min(log10(distance), maxBPLog) * BPCost + 0.5), :1014-1025.
// The graph loop asks for it for a hundred node pairs a read, and log10 is by far the dearest thing in that loop.  The value is a non-decreasing step function
// of the integer distance with at most maxBPLog * BPCost steps, so the steps are found once -- by bisection with the very same floating-point expression, so that
// every distance gets the value the expression gives -- and a call is a search among ~20 thresholds.
struct BPPTable {
    int bpCost = -1, mbpl = -1, vmin = 0; std::vector<uint32_t> thr;
    static int exact(uint32_t distance, int BPCost, int MBPL) { double lg = log10((double)distance); if (lg > MBPL) lg = (double)MBPL; return (int)(lg * BPCost + 0.5); }
    void build(int BPCost, int MBPL)
    {
        bpCost = BPCost; mbpl = MBPL; thr.clear(); vmin = exact(11, BPCost, MBPL);
        const int vmax = exact(0xFFFFFFFFu, BPCost, MBPL);
        for (int v = vmin + 1; v <= vmax; v++) {                        // smallest distance whose penalty is >= v
            uint64_t lo = 11, hi = 0xFFFFFFFFull;
            while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (exact((uint32_t)mid, BPCost, MBPL) >= v) hi = mid; else lo = mid + 1; }
            thr.push_back((uint32_t)lo);
        }
    }
    int operator()(uint32_t distance) const { int k = 0; const int n = (int)thr.size(); while (k < n && thr[k] <= distance) k++; return vmin + k; }
};

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
    static thread_local std::vector<int> tlPfxOff, tlPool; static thread_local std::vector<SortKey> tlKeys;
    tlPfxOff.assign(n, -1); tlPool.clear();
    Ctx X{a, cl, ops, std::move(tlNodes), tlPfxOff, tlPool};
    struct GiveBack { Ctx &x; ~GiveBack() { tlNodes = std::move(x.nodes); } } giveBack{X};
    // keys in list order head->tail (:929-934), sorted
    tlKeys.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const ygpu_clump &c = cl[i]; const bool rev = (c.status & stReversed) != 0;
        tlKeys[i] = {compareKey(rev ? (qlen - 1) - c.eqo : c.sqo, rev ? (qlen - 1) - c.sqo : c.eqo, (int)c.totScore), (int)i};
    }
    RandState rs; seedFromRead(r, rs);
    quickSort(tlKeys.data(), 0, (int)n - 1, rs);
    // deleteSubsumedDups :488-517, run on the sorted keys: all it reads of a node -- SQO, EQO, score -- is in the key (SQO<<32 | (-EQO & 0xffff)<<16 | -score & 0xffff),
    // and only two nodes of equal SQO and EQO are ever compared by reference position.  A dead node is marked in its clump field (~clump); nodes are then made
    // (initcGraphNode :342-363) for the survivors only -- a quarter of what the device returns for a 1 kbp read.
    int cnt = 0;
    {
        SortKey *sk = tlKeys.data(); const int nodeCount = (int)n;
        auto eqoOf = [](uint64_t k) { return (int)(uint16_t)(0u - (uint32_t)((k >> 16) & 0xffff)); };
        auto scoreOf = [](uint64_t k) { return (int)(int16_t)(uint16_t)(0u - (uint32_t)(k & 0xffff)); };
        for (int i = 0; i < nodeCount; i++) {
            if (sk[i].clump < 0) continue;
            const uint64_t ck = sk[i].key; const int ci = sk[i].clump;
            sk[cnt++] = sk[i];                                          // survivors compacted in place (cnt <= i)
            const int curEQO = eqoOf(ck), thr = scoreOf(ck) / 8; const uint64_t cur32 = ck >> 16;
            for (int j = i + 1; j < nodeCount; j++) {
                if (sk[j].clump < 0) continue;
                const uint64_t k = sk[j].key; const int e = eqoOf(k);
                if (e > curEQO) break;
                bool kill = (curEQO > e && scoreOf(k) < thr);
                if (!kill && (k >> 16) == cur32) {                      // same SQO and EQO: duplicates if they are the same piece of the reference on the same strand
                    const ygpu_clump &c1 = cl[ci], &c2 = cl[sk[j].clump];
                    kill = (c1.sro == c2.sro && c1.refLen == c2.refLen && ((c1.status ^ c2.status) & stReversed) == 0);
                }
                if (kill) sk[j].clump = ~sk[j].clump;
            }
        }
    }
    X.nodes.resize(cnt);
    for (int p = 0; p < cnt; p++) {
        const int i = tlKeys[p].clump;
        CNode &nd = X.nodes[p]; const ygpu_clump &c = cl[i]; bool rev = (c.status & stReversed) != 0;
        nd.bestPrev = -1; nd.pathLength = 1; nd.clump = i;
        nd.bestScore = nd.nodeScore = (int16_t)(int)c.totScore; nd.nodeLength = (int16_t)c.totLength;
        nd.SQO = rev ? (uint16_t)((qlen - 1) - c.eqo) : c.sqo; nd.EQO = rev ? (uint16_t)((qlen - 1) - c.sqo) : c.eqo;
        nd.SRO = c.sro; nd.ERO = c.sro + c.refLen - 1; nd.reversed = rev; nd.qLenInOQC = (uint16_t)(1 + c.eqo - c.sqo);
        nd.seqNum = 0;
    }
    const int curNodeCount = cnt;
    { int last = 0;                                                     // the sequence of a node (break point penalty): most nodes of a read lie in one or two sequences
      for (int i = 0; i < curNodeCount; i++) { CNode &nd = X.nodes[i]; const BaseSeq &b = g.seqs[last]; if (nd.SRO >= b.start && nd.SRO < b.start + b.length) { nd.seqNum = (uint8_t)last; continue; } const int f = g.findSeq(nd.SRO); nd.seqNum = (uint8_t)f; if (f >= 0) last = f; } }
    int bestScore = WorstScore, bestNode = -1, startj = 1;
    const int minNonOverlap = a.OQCMinNonOverlap, BPCost = a.BPCost, MBPL = a.maxBPLog;
    static thread_local BPPTable bpp; const bool useTable = BPCost >= 0 && MBPL >= 0 && MBPL * (long)BPCost <= 4096;     // a step function only for non-negative costs
    if (useTable && (bpp.bpCost != BPCost || bpp.mbpl != MBPL)) bpp.build(BPCost, MBPL);
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
                        else BPP = useTable ? bpp(distance) : BPPTable::exact(distance, BPCost, MBPL);
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
