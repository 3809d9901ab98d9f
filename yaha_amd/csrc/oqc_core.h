// oqc_core.h -- the post-filter of one read (Optimal Query Coverage, filter-by-similarity, duplicate removal, mapping quality: reference
// GraphPath.cpp:294-1086) as ONE routine on plain arrays, without recursion, allocation or library calls, so that the same source runs
//   * on the device, one read per lane, right after the hot path (device/oqc_stage.h: only the clumps that are printed -- one or two of the ~75 a 1 kbp read
//     produces -- and their edit ops cross PCIe, and the host's share of a read shrinks to parsing and printing), and
//   * on the host (host/oqc.cpp: the Session / ctypes path, runs without a usable break-point table, and the referee of the device stage in the tests).
// Every step keeps the reference's order of operations where the result depends on it: the quicksort consumes the per-read generator's bits in the reference's
// comparison order (ties decide which duplicate survives), 16-bit score fields wrap as the reference's SINT fields do, and the floating-point expressions of
// the similarity filter and the mapping quality are evaluated in double precision without contraction (a fused multiply-add rounds once where the reference
// rounds twice).
#pragma once
#include <stdint.h>
#include <math.h>
#include "../../include/yaha_hip.h"

#if defined(__HIPCC__)
#define YQ_FN __host__ __device__ inline
#else
#define YQ_FN inline
#endif
#define YQ_INLINE __attribute__((always_inline))

namespace yoqc {

enum { stReversed = 0x01, stPrimary = 0x20 };
#define YQ_WORST (-(0x7fffff00))

struct Params {
    int GOCost, GECost, RCost, MScore;
    int minNonOverlap, BPCost, maxBPLog, FBS;
    float FBS_PSLength, FBS_PSScore;
    // break point penalty of a distance > 10 on one sequence, (int)(min(log10(d), maxBPLog) * BPCost + 0.5) (:1014-1025), as the step function it is: bppVmin +
    // the number of thresholds <= d.  The thresholds are found on the host by bisection with that very expression (host/oqc.cpp: BPPTable).
    int bppVmin, bppN; const uint32_t *bppThr;
};
struct Seqs { const uint32_t *start, *length; uint32_t n; };           // reference sequences (bases), ascending
struct SortKey { uint64_t key; int clump; int pad; };
struct CNode {                                                         // cGraphNode, GraphPath.cpp:299-324
    int bestPrev; int clump;                                           // clump < 0 = dead
    int16_t bestScore, pathLength; uint32_t SRO, ERO; uint16_t SQO, EQO; int16_t nodeLength, nodeScore; uint16_t qLenInOQC; uint8_t reversed, seqNum;
};
struct PAttr { int alignedQueryLength, numOutputSecondaries; int16_t secondScore, thirdScore; };   // :526-533
struct OutRec { int clump; uint8_t status, mapQuality; uint16_t numSecondaries, matchedPrimary, pad; };   // one clump as it reaches printClump, in print order
// Per-read work space (n = clumps of the read, cnt <= n = nodes that survive the duplicate removal):
//   keys n; the sort's stack: stackCap ints in `stack`, what goes beyond in `stack2` (4n + 8 ints hold any recursion); nodes, tbl, path: cnt each;
//   running-sum tables: `pool` while they fit its poolCap ints, `pool2` after that (the sum over the read's clumps of 2 n_ops + 3 ints holds every table);
//   prim, pa, push: n each (touched once at the end).
// The device keeps keys, stack, nodes, tbl, path and a small pool in LDS (device/oqc_stage.h); the host has one of everything.
struct Scratch { SortKey *keys; int *stack; int stackCap; int *stack2; CNode *nodes; int *tbl; int *path; int *pool; int poolCap; int *pool2; CNode *prim; PAttr *pa; OutRec *push;
    };

struct Rand { uint32_t s[5]; };                                        // Marsaglia xorshift, Math.c:257-290
YQ_FN uint32_t randBits(Rand &r)
{
    uint32_t t = r.s[0] ^ (r.s[0] >> 7);
    r.s[0] = r.s[1]; r.s[1] = r.s[2]; r.s[2] = r.s[3]; r.s[3] = r.s[4];
    r.s[4] = (r.s[4] ^ (r.s[4] << 6)) ^ (t ^ (t << 13));
    return (r.s[1] + r.s[1] + 1) * r.s[4];
}
// generateRandomSeed, QueryState.c:172-187: word i of the seed = the two low bits of codes 16 i .. 16 i + 15, the index wrapping around the read
YQ_FN uint32_t seedWord(const uint8_t *codes, int n, int i)
{
    uint32_t w = 0; int q = (16 * i) % n;
    for (int j = 0; j < 16; j++) { w = (w << 2) | (codes[q] & 0x3); q++; if (q >= n) q = 0; }
    return w;
}
YQ_FN void seedFromCodes(const uint8_t *codes, int n, Rand &rs) { for (int i = 0; i < 5; i++) rs.s[i] = seedWord(codes, n, i); }
YQ_FN uint64_t compareKey(int SQO, int EQO, int score)                // getCompareKey :377-380
{ return ((((uint64_t)(uint16_t)SQO << 16) + (uint16_t)(-(int)(int16_t)(uint16_t)EQO)) << 16) + (uint16_t)(-(int)(int16_t)score); }
YQ_FN int keyEQO(uint64_t k) { return (int)(uint16_t)(0u - (uint32_t)((k >> 16) & 0xffff)); }
YQ_FN int keyScore(uint64_t k) { return (int)(int16_t)(uint16_t)(0u - (uint32_t)(k & 0xffff)); }
YQ_FN int findSeq(const Seqs &g, uint32_t off)                        // findBaseSequenceNum, BaseSeq.c:81-90
{ for (uint32_t i = 0; i < g.n; i++) if (off >= g.start[i] && off < g.start[i] + g.length[i]) return (int)i; return -1; }

// double-precision pieces of the filter, defined at the end of this file under "no contraction" (a fused multiply-add would round once where the reference rounds twice)
YQ_FN uint8_t mapQuality(int totScore, int secondScore, int thirdScore);
YQ_FN bool similarEnough(int nodeScore, int primaryScore, float PSScore);
YQ_FN bool overlapsEnough(int overlap, int len, double target);

// bppN < 0 (negative costs: not a non-decreasing step function): the expression itself, on the host only -- the device stage is not used for such runs
#if defined(__HIP_DEVICE_COMPILE__)
YQ_FN int exactBPP(uint32_t, int, int) { return 0; }
#else
inline int exactBPP(uint32_t distance, int BPCost, int MBPL) { double lg = log10((double)distance); if (lg > MBPL) lg = (double)MBPL; return (int)(lg * BPCost + 0.5); }
#endif

// The state of one read's run, in the steps the reference takes (postFilterBySimilarity :897-1086).  run() below strings them together for one thread; the device
// stage calls the same steps, the independent ones -- the duplicate scan behind a node, the candidate successors of a node -- from the lanes of a wave.
struct Run {
    const Params &P; const ygpu_clump *cl; const uint32_t *ops; Scratch S; int poolUsed, pool2Used;
    YQ_FN int opScore(char op, int len) const
    { return op == 'M' ? P.MScore * len : op == 'R' ? -(P.RCost * len) : op == 'I' ? -(P.GOCost + P.GECost * len) : 0; }

    // ---- keys, sort ------------------------------------------------------------------------------------------------------------------------------------------
    YQ_FN uint64_t clumpKey(int i, int qlen) const                      // (48 bits: SQO, -EQO, -score in 16 each)
    {
        const ygpu_clump &c = cl[i]; const bool rev = (c.status & stReversed) != 0;
        return compareKey(rev ? (qlen - 1) - c.eqo : c.sqo, rev ? (qlen - 1) - c.sqo : c.eqo, (int)c.totScore);
    }
    YQ_FN void makeKey(int i, int qlen)                                 // keys in list order head->tail (:929-934)
    { SortKey k; k.key = clumpKey(i, qlen); k.clump = i; k.pad = 0; S.keys[i] = k; }
    // The sort (myQuickSortHelper :427-453 on getCompareKey, ties broken by the per-read generator) decides which of two equal-keyed clumps comes first -- which
    // duplicate survives, which copy of a repeat becomes the primary -- so its comparisons must happen in the reference's order.  They depend on keys and positions
    // only: the routine runs on (key, clump) pairs with the keys computed once; the reference's recursion (left part first) is an explicit stack of ranges (its first
    // stkCap ints in stk, what goes beyond in stk2).  Array and stack are parameters, and the function is always inlined: the device calls it once with pointers into
    // LDS and once with pointers into HBM, and the compiler turns each copy's accesses into the instructions of that address space (through a pointer that may be
    // either, every access is a "flat" one that waits for all memory traffic of the wave: 1 400 cycles a comparison, measured).
    YQ_FN YQ_INLINE static void stPush(int *stk, int stkCap, int *stk2, int &sp, int v) { if (sp < stkCap) stk[sp] = v; else stk2[sp - stkCap] = v; sp++; }
    YQ_FN YQ_INLINE static int stPop(int *stk, int stkCap, int *stk2, int &sp) { sp--; int v; if (sp < stkCap) v = stk[sp]; else v = stk2[sp - stkCap]; return v; }
    YQ_FN YQ_INLINE static void sortRange(SortKey *arr, int n, int *stk, int stkCap, int *stk2, Rand rs)
    {
        int sp = 0; stPush(stk, stkCap, stk2, sp, 0); stPush(stk, stkCap, stk2, sp, n - 1);
        while (sp > 0) {
            const int right = stPop(stk, stkCap, stk2, sp), left = stPop(stk, stkCap, stk2, sp);
            if (left >= right) continue;
            const int pivot = (left + right) / 2;
            // (elements move as (key, clump) scalars, never as structs: a SortKey temporary ends up in private memory -- scratch, on the device: a trip to HBM per access)
            { const uint64_t pk0 = arr[pivot].key, rk0 = arr[right].key; const int pc0 = arr[pivot].clump, rc0 = arr[right].clump;
              arr[pivot].key = rk0; arr[pivot].clump = rc0; arr[right].key = pk0; arr[right].clump = pc0; }
            int store = left;
            const uint64_t pk = arr[right].key;
            // "if less: swap(arr[i], arr[store]), store++".  The element at `store` and the next element of the scan are kept in registers: the next element is
            // fetched one iteration ahead (an iteration writes positions i and store <= i only); the one at `store` changes only after a swap.
            uint64_t yk = arr[store].key, nk = arr[left].key; int yc = arr[store].clump, nc = arr[left].clump;
            for (int i = left; i < right; i++) {
                const uint64_t xk = nk; const int xc = nc;
                if (i + 1 < right) { nk = arr[i + 1].key; nc = arr[i + 1].clump; }
                bool less = xk < pk;
                if (xk == pk) less = (randBits(rs) & 1) != 0;
                if (less) {
                    if (i != store) { arr[i].key = yk; arr[i].clump = yc; arr[store].key = xk; arr[store].clump = xc; }
                    store++;
                    // (store == i: the element just written there, read back in order)
                    if (store == i + 1) { yk = nk; yc = nc; } else { yk = arr[store].key; yc = arr[store].clump; }
                }
            }
            { const uint64_t sk0 = arr[store].key, rk0 = arr[right].key; const int sc0 = arr[store].clump, rc0 = arr[right].clump;
              arr[store].key = rk0; arr[store].clump = rc0; arr[right].key = sk0; arr[right].clump = sc0; }
            // the reference sorts [left, store-1] completely before it touches [store+1, right]: the right part goes on the stack first
            stPush(stk, stkCap, stk2, sp, store + 1); stPush(stk, stkCap, stk2, sp, right);
            stPush(stk, stkCap, stk2, sp, left); stPush(stk, stkCap, stk2, sp, store - 1);
        }
    }
    YQ_FN void sortKeys(int n, const uint8_t *fwdCodes, int qlen) { Rand rs; seedFromCodes(fwdCodes, qlen, rs); sortRange(S.keys, n, S.stack, S.stackCap, S.stack2, rs); }

    // ---- deleteSubsumedDups :488-517, on the sorted keys --------------------------------------------------------------------------------------------------------
    // All the scan reads of a node -- SQO, EQO, score -- is in the key (SQO<<32 | (-EQO & 0xffff)<<16 | -score & 0xffff); only two nodes of equal SQO and EQO are
    // ever compared by reference position.  A dead node is marked in its clump field (~clump).  dupKill(ck, ci, j): must the live node at j die behind the live
    // node (ck, ci)?  (The caller has checked that j's EQO does not exceed the node's: there the scan ends.)
    YQ_FN bool dupKillK(uint64_t ck, int ci, uint64_t k, int cj) const   // (the candidate given by value: key k, clump cj)
    {
        if (keyEQO(ck) > keyEQO(k) && keyScore(k) < keyScore(ck) / 8) return true;
        if ((k >> 16) != (ck >> 16)) return false;                      // same SQO and EQO: duplicates if they are the same piece of the reference on the same strand
        const ygpu_clump &c1 = cl[ci], &c2 = cl[cj];
        return c1.sro == c2.sro && c1.refLen == c2.refLen && ((c1.status ^ c2.status) & stReversed) == 0;
    }
    YQ_FN bool dupKill(uint64_t ck, int ci, int j) const { return dupKillK(ck, ci, S.keys[j].key, S.keys[j].clump); }
    YQ_FN int dedup(int n)                                              // survivors compacted to keys[0..return)
    {
        SortKey *sk = S.keys; int cnt = 0;
        for (int i = 0; i < n; i++) {
            if (sk[i].clump < 0) continue;
            const uint64_t ck = sk[i].key; const int ci = sk[i].clump, curEQO = keyEQO(ck);
            sk[cnt++] = sk[i];
            for (int j = i + 1; j < n; j++) {
                if (sk[j].clump < 0) continue;
                if (keyEQO(sk[j].key) > curEQO) break;
                if (dupKill(ck, ci, j)) sk[j].clump = ~sk[j].clump;
            }
        }
        return cnt;
    }

    // ---- nodes (initcGraphNode :342-363), for the survivors only ---------------------------------------------------------------------------------------------------
    YQ_FN void makeNode(int p, const Seqs &g, int qlen, int seqHint)
    {
        const int i = S.keys[p].clump;
        CNode nd; const ygpu_clump &c = cl[i]; const bool rev = (c.status & stReversed) != 0;
        nd.bestPrev = -1; nd.pathLength = 1; nd.clump = i;
        nd.bestScore = nd.nodeScore = (int16_t)(int)c.totScore; nd.nodeLength = (int16_t)c.totLength;
        nd.SQO = rev ? (uint16_t)((qlen - 1) - c.eqo) : c.sqo; nd.EQO = rev ? (uint16_t)((qlen - 1) - c.sqo) : c.eqo;
        nd.SRO = c.sro; nd.ERO = c.sro + c.refLen - 1; nd.reversed = rev; nd.qLenInOQC = (uint16_t)(1 + c.eqo - c.sqo);
        if (seqHint >= 0 && (uint32_t)seqHint < g.n && nd.SRO >= g.start[seqHint] && nd.SRO < g.start[seqHint] + g.length[seqHint]) nd.seqNum = (uint8_t)seqHint;
        else nd.seqNum = (uint8_t)findSeq(g, nd.SRO);
        S.nodes[p] = nd; S.tbl[p] = -1;
    }

    // ---- calcScoreForLength (:705-732) --------------------------------------------------------------------------------------------------------------------------
    // It walks a clump's edit list from one end until `length` query bases are covered; the graph loop asks it ~120 times a read for the same few clumps.  Every
    // node that is asked about gets (once) the running sums Q[m], S[m] = query bases and score after its first m ops, untruncated; a call is then a binary search
    // for the op the walk stops in plus that op's truncated share.  The walk from the other end reads the same table backwards (Qb[m] = Q[n] - Q[n-m]).  Same
    // arithmetic, same results: only the order of the integer additions differs.
    // table of a node with n ops: [n | Q[0..n] | S[0..n]], 2n + 3 ints; Q[m] = query bases after the first m ops in its low 24 bits and the code of op m-1 in its
    // high 8 (a call then needs nothing but the table: on the device the clump record and the op list are HBM round trips, the table sits in LDS)
    YQ_FN static int tableInts(int nOps) { return 2 * nOps + 3; }
    YQ_FN void assignTable(int p)                                       // a place for node p's table (sequential: the pools are bump allocators)
    {
        const int need = tableInts((int)cl[S.nodes[p].clump].n_ops);
        if (poolUsed + need <= S.poolCap) { S.tbl[p] = poolUsed; poolUsed += need; } else { S.tbl[p] = -2 - pool2Used; pool2Used += need; }
    }
    // (a table is reached through `pool` or through `pool2`, never through a pointer that may be either: on the device the first is LDS, the second HBM, and each
    // copy of the always-inlined bodies below compiles to the instructions of its address space)
    YQ_FN YQ_INLINE void fillInto(int *T, int p) const
    {
        const int c = S.nodes[p].clump; const uint32_t *o = ops + cl[c].op_start; const int n = (int)cl[c].n_ops;
        int *Q = T + 1, *Sc = Q + n + 1; int q = 0, sc = 0; T[0] = n; Q[0] = 0; Sc[0] = 0;
        for (int k = 0; k < n; k++) {
            const char op = YGPU_OP_CODE(o[k]); const int len = (int)YGPU_OP_LEN(o[k]);
            if (op == 'D') sc -= (P.GOCost + P.GECost * len); else { q += len; sc += opScore(op, len); }
            Q[k + 1] = q | ((int)(unsigned char)op << 24); Sc[k + 1] = sc;
        }
    }
    YQ_FN void fillTable(int p) { const int off = S.tbl[p]; if (off >= 0) fillInto(S.pool + off, p); else fillInto(S.pool2 + (-2 - off), p); }
    YQ_FN int scoreForLength(int p, int length, bool forward)           // node p's edit list
    {
        if (length <= 0) return 0;
        if (S.tbl[p] == -1) { assignTable(p); fillTable(p); }
        const int off = S.tbl[p];
        return off >= 0 ? scoreIn(S.pool + off, length, forward) : scoreIn(S.pool2 + (-2 - off), length, forward);
    }
    YQ_FN YQ_INLINE int scoreIn(const int *T, int length, bool forward) const
    {
        const int n = T[0]; if (n <= 0) return 0;
        const int *Q = T + 1, *Sc = Q + n + 1; const int M = 0xFFFFFF;
        if ((Q[n] & M) < length) return Sc[n];                           // the list ends first: every op counted in full
        // (host, measured: forcing these searches branch-free -- an AND with the comparison's mask -- made a call slower, 14.9 -> 18.2 us a read: the walks of one
        // read stop in similar places, the branches predict; the reference's own loop for the first four ops before any table is touched: 14.9 -> 16.7 -- the
        // overlaps that reach this function are long)
        if (forward) {
            const int *base = Q + 1; int len = n;                        // first m in 1..n with Q[m] >= length (exists: Q[n] >= length); op m-1 is the one the walk stops in
            while (len > 1) { const int half = len >> 1; base = ((base[half - 1] & M) < length) ? base + half : base; len -= half; }
            const int j = (int)(base - Q) - 1;
            return Sc[j] + opScore((char)((unsigned)Q[j + 1] >> 24), length - (Q[j] & M));
        }
        const int X = (Q[n] & M) - length;                               // last t in 0..n-1 with Q[t] <= X (Q[0] = 0 <= X < Q[n]); op t is the one the backward walk stops in
        const int *base = Q; int len = n;
        while (len > 1) { const int half = len >> 1; base = ((base[half] & M) <= X) ? base + half : base; len -= half; }
        const int t = (int)(base - Q);
        return (Sc[n] - Sc[t + 1]) + opScore((char)((unsigned)Q[t + 1] >> 24), length - ((Q[n] & M) - (Q[t + 1] & M)));
    }
    YQ_FN int accurateOverlapScore(int left, int right, int overlap, bool *rightBest)   // :744-800
    {
        const CNode &rn = S.nodes[right];
        int rightScore = scoreForLength(right, overlap, !rn.reversed);
        int pathScore = 0, remaining = overlap, cur = left;
        for (;;) {
            const CNode &cn = S.nodes[cur];
            int q = remaining < (int)cn.qLenInOQC ? remaining : (int)cn.qLenInOQC; remaining -= q;
            pathScore += scoreForLength(cur, q, cn.reversed != 0);
            if (remaining <= 0) break;
            cur = cn.bestPrev;
        }
        if (pathScore > rightScore) { *rightBest = false; return rightScore; }
        *rightBest = true; return pathScore;
    }
    YQ_FN void cacheReverse(int left, int right, int overlap, bool rightBest)            // cacehQlenInOQCPathReverse :802-826
    {
        CNode &rn = S.nodes[right];
        if (rightBest) {
            rn.qLenInOQC = (uint16_t)(1 + rn.EQO - rn.SQO);
            int remaining = overlap, cur = left;
            for (;;) {
                CNode &cn = S.nodes[cur];
                int q = remaining < (int)cn.qLenInOQC ? remaining : (int)cn.qLenInOQC;
                cn.qLenInOQC = (uint16_t)(cn.qLenInOQC - q); remaining -= q;
                if (remaining <= 0) break;
                cur = cn.bestPrev;
            }
        } else rn.qLenInOQC = (uint16_t)((1 + rn.EQO - rn.SQO) - overlap);
    }
    // cacheQlenInOQCPath :841-867: the reference recurses to the head of the best path to `right` and works its way back; here the path is written out first
    // (S.path, head last) and walked from its head
    YQ_FN void cachePath(int right)
    {
        int m = 0;
        for (int p = right; p >= 0; p = S.nodes[p].bestPrev) S.path[m++] = p;
        { CNode &hd = S.nodes[S.path[m - 1]]; hd.qLenInOQC = (uint16_t)(1 + hd.EQO - hd.SQO); }
        for (int k = m - 2; k >= 0; k--) {
            const int left = S.path[k + 1], rt = S.path[k];
            CNode &rn = S.nodes[rt]; const CNode &ln = S.nodes[left];
            const int qLen = 1 + rn.EQO - rn.SQO;
            const int overlap = ((int)ln.EQO >= (int)rn.SQO) ? ((int)ln.EQO - (int)rn.SQO) + 1 : 0;
            if (overlap > 0) { bool rb; accurateOverlapScore(left, rt, overlap, &rb); cacheReverse(left, rt, overlap, rb); }
            else rn.qLenInOQC = (uint16_t)qLen;
        }
    }
    // ---- the graph (:973-1063) ----------------------------------------------------------------------------------------------------------------------------------
    // node j as a successor of node i (i < j, SQO_j - SQO_i >= minNonOverlap already checked; cachePath(i) has run).  Reads node i and its path, writes node j only:
    // the successors of one node are independent of each other.
    YQ_FN void relax(int i, int j)
    {
        const CNode &ln = S.nodes[i]; CNode &rn = S.nodes[j];
        const int leftEQO = ln.EQO, rightSQO = rn.SQO, rightEQO = rn.EQO;
        if ((rightEQO - leftEQO) < P.minNonOverlap) return;
        int16_t newScore = (int16_t)(ln.bestScore + rn.nodeScore);
        if (rn.bestScore > newScore) return;
        int BPP;
        if (ln.seqNum == rn.seqNum) {
            uint32_t distance;
            if (ln.SRO > rn.ERO) distance = ln.SRO - rn.ERO; else if (rn.SRO > ln.ERO) distance = rn.SRO - ln.ERO; else distance = 0;
            if (distance <= 10) BPP = P.BPCost;
            else if (P.bppN >= 0) {                                    // number of thresholds <= distance (ascending): upper bound by bisection
                int lo = 0, hi = P.bppN;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.bppThr[mid] <= distance) lo = mid + 1; else hi = mid; }
                BPP = P.bppVmin + lo;
            }
            else BPP = exactBPP(distance, P.BPCost, P.maxBPLog);
        } else BPP = P.maxBPLog * P.BPCost;
        newScore = (int16_t)(newScore - BPP);
        if (rn.bestScore > newScore) return;
        const int overlap = (leftEQO >= rightSQO) ? (leftEQO - rightSQO) + 1 : 0;
        bool rightBest = false;
        if (overlap > 0) { newScore = (int16_t)(newScore - accurateOverlapScore(i, j, overlap, &rightBest)); if (rn.bestScore > newScore) return; }
        if (rn.bestScore < newScore || (rn.bestPrev >= 0 && ln.pathLength < S.nodes[rn.bestPrev].pathLength)) {
            if (overlap > 0) { const int ql = 1 + rn.EQO - rn.SQO; rn.qLenInOQC = (uint16_t)(rightBest ? ql : ql - overlap); }   // cacheQlenInRightNode :873-878
            rn.bestScore = newScore; rn.bestPrev = i; rn.pathLength = (int16_t)(ln.pathLength + 1);
        }
    }
    YQ_FN void considerBest(int i, int &bestScore, int &bestNode) const
    {
        const CNode &ln = S.nodes[i];
        if (ln.bestScore < bestScore) return;
        if (ln.bestScore > bestScore || (bestNode >= 0 && ln.pathLength < S.nodes[bestNode].pathLength)) { bestNode = i; bestScore = ln.bestScore; }
    }
    // ---- filterBySimilarity :571-692, calcMQfromPAs :559-569 ---------------------------------------------------------------------------------------------------------
    YQ_FN int finish(int curNodeCount, int bestNode, OutRec *out, int *primaryCount)
    {
        CNode *gn = S.nodes;
        const int primeCount = gn[bestNode].pathLength;
        CNode *primaries = S.prim; PAttr *PA = S.pa; OutRec *push = S.push; int nPush = 0;      // push-to-head order; reversed at the end
        {
            int pi = primeCount - 1;
            for (int p = bestNode; p >= 0; p = gn[p].bestPrev) {
                primaries[pi] = gn[p];
                PAttr a; a.alignedQueryLength = 1 + gn[p].EQO - gn[p].SQO; a.numOutputSecondaries = 0; a.secondScore = 0; a.thirdScore = 0; PA[pi] = a;
                OutRec o; o.clump = gn[p].clump; o.status = (uint8_t)(cl[gn[p].clump].status | stPrimary); o.mapQuality = 255; o.numSecondaries = 0;
                    o.matchedPrimary = (uint16_t)(pi + 1); o.pad = 0;
                push[nPush++] = o;
                gn[p].clump = -1; pi--;
            }
        }
        const double targetOverlap = (double)P.FBS_PSLength;
        for (int i = 0; i < curNodeCount; i++) {
            const CNode &cn = gn[i];
            if (cn.clump < 0) continue;
            const int curSQO = cn.SQO, curEQO = cn.EQO, curQLen = 1 + curEQO - curSQO; int maxOverlap = 0, maxIndex = 0;
            for (int k = 0; k < primeCount; k++) {
                const int e = curEQO < (int)primaries[k].EQO ? curEQO : (int)primaries[k].EQO, s = curSQO > (int)primaries[k].SQO ? curSQO : (int)primaries[k].SQO;
                const int overlap = 1 + e - s;
                if (overlap > maxOverlap) { maxOverlap = overlap; maxIndex = k; }
            }
            if (maxOverlap > 0) {
                PAttr &pa = PA[maxIndex];
                if (cn.nodeScore > pa.secondScore) { pa.thirdScore = pa.secondScore; pa.secondScore = cn.nodeScore; }      // memoPAsFromOverlappingNode :545-557
                else if (cn.nodeScore > pa.thirdScore) pa.thirdScore = cn.nodeScore;
                const CNode &pn = primaries[maxIndex];
                if (similarEnough(cn.nodeScore, pn.nodeScore, P.FBS_PSScore)) {
                    const int e = curEQO < (int)pn.EQO ? curEQO : (int)pn.EQO, s = curSQO > (int)pn.SQO ? curSQO : (int)pn.SQO;
                    const int overlap = 1 + e - s, pathQLen = pa.alignedQueryLength;
                    if (overlapsEnough(overlap, curQLen, targetOverlap) && overlapsEnough(overlap, pathQLen, targetOverlap)) {
                        pa.numOutputSecondaries += 1;
                        if (P.FBS) {
                            OutRec o; o.clump = cn.clump; o.status = cl[cn.clump].status; o.mapQuality = 255; o.numSecondaries = 0; o.matchedPrimary = (uint16_t)(maxIndex + 1);
                                o.pad = 0;
                            push[nPush++] = o; continue;
                        }
                    }
                }
            }
        }
        *primaryCount = primeCount;
        for (int k = 0; k < primeCount; k++) {                          // primaries are push[0..primeCount) holding index primeCount-1 .. 0
            OutRec &o = push[k]; const PAttr &pa = PA[primeCount - 1 - k];
            o.mapQuality = mapQuality((int)cl[o.clump].totScore, pa.secondScore, pa.thirdScore);
            o.numSecondaries = (uint16_t)pa.numOutputSecondaries;
        }
        for (int k = nPush - 1; k >= 0; k--) out[nPush - 1 - k] = push[k];
        return nPush;
    }
};

YQ_FN int single(const ygpu_clump *cl, OutRec *out, int *primaryCount)      // a read with one clump, :907-916
{
    OutRec o; o.clump = 0; o.status = (uint8_t)(cl[0].status | stPrimary); o.mapQuality = 250; o.numSecondaries = 0; o.matchedPrimary = 1; o.pad = 0;
    out[0] = o; *primaryCount = 1; return 1;
}
// postFilterBySimilarity (:897-1086) for the n clumps of one read (QS->clumps head -> tail), one thread.  out[0..return) = the clumps to print, in print order;
// *primaryCount = QS->primaryCount.  Needs n >= 1.
YQ_FN int run(const Params &P, const Seqs &g, const ygpu_clump *cl, int n, const uint32_t *ops, int qlen, const uint8_t *fwdCodes, Scratch S, OutRec *out, int *primaryCount)
{
    if (n == 1) return single(cl, out, primaryCount);
    Run X{P, cl, ops, S, 0, 0};
    for (int i = 0; i < n; i++) X.makeKey(i, qlen);
    X.sortKeys(n, fwdCodes, qlen);
    const int cnt = X.dedup(n);
    // (most nodes of a read lie in one or two sequences)
    { int last = 0; for (int p = 0; p < cnt; p++) { X.makeNode(p, g, qlen, last); const int f = S.nodes[p].seqNum; if (f != 255) last = f; } }
    int bestScore = YQ_WORST, bestNode = -1, startj = 1;
    for (int i = 0; i < cnt; i++) {                                     // :973-1063
        X.cachePath(i);
        const int leftSQO = S.nodes[i].SQO; bool foundstartj = false;
        for (int j = startj; j < cnt; j++) {
            if (((int)S.nodes[j].SQO - leftSQO) >= P.minNonOverlap) {
                if (!foundstartj) { startj = j; foundstartj = true; }
                X.relax(i, j);
            }
            if (!foundstartj) startj = cnt;
        }
        X.considerBest(i, bestScore, bestNode);
    }
    return X.finish(cnt, bestNode, out, primaryCount);
}
}  // namespace yoqc

namespace yoqc {
#if defined(__clang__)
#pragma clang fp contract(off)
#define YQ_NOFMA
#elif defined(__GNUC__)
#define YQ_NOFMA __attribute__((optimize("fp-contract=off")))
#else
#define YQ_NOFMA
#endif
YQ_NOFMA YQ_FN uint8_t mapQuality(int totScore, int secondScore, int thirdScore)                  // calcMQfromPAs :559-569
{
    if (secondScore == 0) return 250;
    const double ts = (double)totScore;
    double a = ts - (double)secondScore; if (!(a > 0.0)) a = 0.0;
    double ratio = a / ts;
    double b = ts - (double)thirdScore; if (!(b > 0.0)) b = 0.0;
    const double f = 1.0 + b / (double)totScore;
    ratio = ratio * f;
    ratio = ratio / 2.0;
    const double q = 250.0 * ratio;
    return (uint8_t)(q + 0.5);
}
YQ_NOFMA YQ_FN bool similarEnough(int nodeScore, int primaryScore, float PSScore) { return ((double)nodeScore) / (double)primaryScore >= (double)PSScore; }
YQ_NOFMA YQ_FN bool overlapsEnough(int overlap, int len, double target) { return ((double)overlap) / (double)len >= target; }

#if defined(__clang__)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(fast)
#else
#pragma clang fp contract(on)
#endif
#endif
}  // namespace yoqc
