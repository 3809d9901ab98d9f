// oqc_stage.h -- the post-filter on the device (SURVEY.md 8(f)-1 widened onto the GPU): Optimal Query Coverage, filter-by-similarity and mapping quality
// (reference GraphPath.cpp:897-1086) for every read of the batch, right where the hot path left its clumps, so that only the clumps that are PRINTED -- one or
// two of the ~75 a 1 kbp read produces -- and their edit ops travel to the host (9 KB a read -> ~0.2 KB) and the host's share of a read is parsing and
// printing.  The routine itself is ../oqc_core.h, the very source the host compiles (host/oqc.cpp), its steps called from k_oqc_wave: a wave per read, the work
// space in LDS, the independent steps on the wave's lanes.
#pragma once
#include "common.h"
#include "../oqc_core.h"

struct OqcArgs {
    yoqc::Params P; yoqc::Seqs G;
    // (the batch's results as ygpu_postfilter_snapshot copied them; five seed words and the length of every read)
    const uint32_t *cs; const ygpu_clump *cl; const uint32_t *ops; const uint32_t *seeds, *qlen; uint32_t nReads;
    const unsigned long long *poolOff;
    yoqc::SortKey *keys; int *stack; yoqc::CNode *nodes, *prim; yoqc::PAttr *pa; int *pfxOff, *path, *pool; yoqc::OutRec *push, *out;
    uint32_t *outCnt, *outOpsCnt; uint32_t *primCnt;
    int devMax;                        // reads with more clumps are left to the host (YQ_DEVICE_MAX; YGPU_OQC_MAX lowers it in tests)
    // test hook (YGPU_OQC_HBM=1): every read's nodes and tables in HBM -- the path of a read whose survivors do not fit its LDS, which real batches hardly ever take
    int graphInHbm;
    // YGPU_OQC_PROF=1: 100 MHz ticks per step of k_oqc_wave, summed over the reads and the largest of any read (keys, sort, duplicate scan, nodes + tables, walk along the path,
    // successors, finish), per class
    unsigned long long *prof;
};
// What the routine costs on a GPU, measured: one read per lane with its work space in HBM took 75 ms a batch (a few thousand DEPENDENT accesses a read, microseconds
// each); one read per wave, first lane only, with the work space in LDS still 39 ms for the reads of 320..640 clumps -- the graph loop is quadratic in the nodes that survive the
// duplicate scan, and reads in repeats have hundreds.  So a read gets a WAVE, the work space in LDS, and the steps oqc_core.h marks as independent run on its
// 64 lanes: the scan behind a node in the duplicate removal (64 candidates at a time, the ballot finds where the scan ends), the nodes and their running-sum
// tables, and the successors of a node in the graph loop (each lane relaxes a different successor: it reads the node and its path and writes its own successor
// only).  What must stay in the reference's order stays on the first lane: the sort (it consumes the read's random bits in comparison order), the walk along
// the best path before each node's successors, the choice of the best node, the similarity filter.
// Reads come in classes by their number of clumps (the LDS a workgroup gets is fixed at launch): keys + sort stack, then nodes + table index + path + tables share it.
#define YQ_NCLASS 5
// clumps of the largest read the device stage filters: its sort (waveSort: 36 bytes a clump) fits the 64 KB a workgroup may ask for (see k_oqc_raw)
#define YQ_DEVICE_MAX 1792
#define YQ_SORT_LDS 36u               // LDS bytes a clump during the sort: the key record (16), two packed sort entries (8 + 8), one position (4)
#define YQ_STACK_LDS 128              // ints of the sort's stack kept in LDS (depth ~2 log2 n ranges); deeper recursion continues in HBM
#define YQ_THR_LDS 64                 // break point thresholds copied to LDS (every lane searches them for every successor it relaxes); a longer table stays in HBM
#define YQ_LDS_MAX 65536u
__device__ __constant__ const int kOqcCapN[YQ_NCLASS] = {112, 224, 448, YQ_DEVICE_MAX, 0x7fffffff};      // clumps a read of the class may have (the last class: left to the host)
__host__ __device__ inline unsigned oqcLdsBytes(int capN) { return 64u * (unsigned)capN + 4u * (YQ_STACK_LDS + YQ_THR_LDS) + 64u; }
// the snapshot's share of the reads themselves: the five words the reference seeds a read's generator with (generateRandomSeed, QueryState.c:172-187) and its length
__global__ void k_oqc_seeds(const uint8_t *fwd, const uint32_t *readOff, uint32_t nReads, uint32_t *seeds, uint32_t *qlen)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nReads) return;
    const uint32_t o = readOff[r]; const int n = (int)(readOff[r + 1] - o);
    qlen[r] = (uint32_t)n;
#pragma unroll
    for (int k = 0; k < 5; k++) seeds[5ull * r + k] = n > 0 ? yoqc::seedWord(fwd + o, n, k) : 0u;
}
// classes of the reads with two or more clumps (lists[c * nReads ...], cnt[c]); reads with one clump are settled here (GraphPath.cpp:907-916); ints of running-sum
// tables a read may need in HBM: 2 n_ops + 3 per clump of the read (every clump's table built)
__global__ void k_oqc_classify(OqcArgs A, unsigned long long *need, uint32_t *lists, unsigned int *cnt)
{
    YD_HIGH_PRIO();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; const int lane = (int)(threadIdx.x & 63);
    unsigned long long v = 0; int cls = -1;
    if (r < A.nReads) {
        const uint32_t b = A.cs[r], n = A.cs[r + 1] - b;
        if (n >= 2) { for (uint32_t c = b; c < b + n; c++) v += 2ull * (unsigned long long)A.cl[c].n_ops + 3ull; cls = 0; while (n > (uint32_t)kOqcCapN[cls]) cls++;
            if (n > (uint32_t)A.devMax) cls = YQ_NCLASS - 1; }
        else {
            uint32_t m = 0, nops = 0; int pc = 0;
            if (n == 1) { m = (uint32_t)yoqc::single(A.cl + b, A.out + b, &pc); nops = A.cl[b].n_ops; }
            A.outCnt[r] = m; A.outOpsCnt[r] = nops; A.primCnt[r] = (uint32_t)pc;
        }
    }
    if (r <= A.nReads) need[r] = v;
#pragma unroll
    for (int c = 0; c < YQ_NCLASS; c++) {
        const unsigned long long mk = __ballot(cls == c);
        if (!mk) continue;
        unsigned base = 0; const int leader = __builtin_ctzll(mk);
        if (lane == leader) base = atomicAdd(&cnt[c], (unsigned)__builtin_popcountll(mk));
        base = (unsigned)__shfl((int)base, leader, 64);
        if (cls == c) lists[(size_t)c * A.nReads + base + (unsigned)__builtin_popcountll(mk & ((1ull << lane) - 1ull))] = r;
    }
}
// ---- the sort, one partition at a time on the wave's lanes ---------------------------------------------------------------------------------------------------
// The reference's quicksort (myQuickSortHelper, GraphPath.cpp:427-453; oqc_core.h sortRange is its restatement for one thread) must be followed comparison by
// comparison -- ties against the pivot consume the read's random bits in scan order, and which of two equal clumps comes first decides which duplicate survives --
// and a single GPU lane does only two million of its iterations a second (1 300 cycles each, rounds 3-4: it was 80 % of this stage).  But ONE PARTITION is a pure
// function of its inputs that the lanes can evaluate together.  With E[k] the scanned elements (pivot already swapped to the right end) and less[k] = E[k] < pivot, or,
// for a tie, the next random bit in order of k (the ties are counted with a ballot and the generator -- in scalar registers -- stepped that many times):
//   * the scan's `store` when it looks at k is left + rank(k), rank(k) = number of less elements before k: the less elements end up at left + rank(k), in order;
//   * position k holds, after step k, E[k] if it was not less, else what position rank(k) held (the swap): src(k) = less ? rank(k) : k, a forest whose roots are the
//     not-less elements; the lanes chase it by pointer doubling (__shfl; a handful of rounds, ends when no lane moved);
//   * positions above nL = rank(m) are never written again, so final[j] = E[root(j)] for nL < j < m; the closing swap puts the pivot at nL and what nL held at the right end.
// Ranges of up to 64 elements -- nearly all of a sort's partitions -- live in registers (one LDS read and at most two LDS writes a lane); longer ones go 64 at a time
// through an LDS copy.  The recursion is the reference's: the left part completely before the right part (the right part waits on the stack).
// A partition costs ~1 000 cycles whatever its length, a read of 50 clumps 20 us instead of 220, one of 550 under 0.2 ms instead of 2.
__device__ __forceinline__ uint32_t yqLanesBelow(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint64_t yqShfl64(uint64_t v, int src) {
    return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src, 64); }
__device__ __forceinline__ unsigned long long yqDraw(yoqc::Rand &rs, int t)        // the next t random bits (t <= 64, uniform), bit d = draw d
{ unsigned long long m = 0; for (int d = 0; d < t; d++) m |= (unsigned long long)(yoqc::randBits(rs) & 1u) << d; return m; }
// a[0..n): sort entries (48-bit key << 16 | clump) in LDS; a2 (n entries) and pos (n ints): scratch in LDS; stk/stk2: the stack of waiting right parts (stkCap even)
__device__ __forceinline__ void waveSort(uint64_t *a, uint64_t *a2, int *pos, int n, int *stk, int stkCap, int *stk2, yoqc::Rand rs, int lane)
{
#pragma unroll
    for (int k = 0; k < 5; k++) rs.s[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rs.s[k]);
    int sp = 0, left = 0, right = n - 1;
    for (;;) {
        if (left >= right) {
            if (sp == 0) break;
            sp -= 2; int l, r;
            if (sp < stkCap) { l = stk[sp]; r = stk[sp + 1]; } else { l = stk2[sp - stkCap]; r = stk2[sp + 1 - stkCap]; }
            left = __builtin_amdgcn_readfirstlane(l); right = __builtin_amdgcn_readfirstlane(r);
            continue;
        }
        const int m = right - left, pivotIdx = (left + right) >> 1;        // m scanned elements; the pivot moves to `right` first (:431-433), here by index
        int nL;
        if (m < 64) {
            const int k = lane; const bool valid = k < m; const int idx = left + k;
            const uint64_t E = k <= m ? a[idx == pivotIdx ? right : k == m ? pivotIdx : idx] : ~0ull; const uint64_t ek = E >> 16;      // (lane m reads the pivot: one trip to LDS)
            const uint64_t P = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(E >> 32), m) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)E, m);
                const uint64_t pk = P >> 16;
            bool less = valid && ek < pk; const bool eq = valid && ek == pk;
            const unsigned long long eqMask = __ballot(eq);
            if (eqMask) { const unsigned long long bits = yqDraw(rs, __builtin_popcountll(eqMask)); if (eq) less = ((bits >> yqLanesBelow(eqMask)) & 1ull) != 0; }
            const unsigned long long lessMask = __ballot(less);
            nL = __builtin_popcountll(lessMask); const int rank = (int)yqLanesBelow(lessMask);
            int src = less ? rank : k;
            for (;;) { const int s2 = __shfl(src, src, 64); if (!__ballot(s2 != src)) break; src = s2; }
            const int rootNL = __builtin_amdgcn_readlane(src, __builtin_amdgcn_readfirstlane(nL < 64 ? nL : 0));
            const uint64_t G = yqShfl64(E, lane == m ? rootNL : src);
            if (less) a[left + rank] = E;
            if (lane >= nL && lane <= m) a[left + lane] = lane == nL ? P : G;
        } else {
            const uint64_t P = a[pivotIdx]; const uint64_t pk = P >> 16;
            int base = 0;
            for (int c0 = 0; c0 < m; c0 += 64) {
                const int k = c0 + lane; const bool valid = k < m; const int idx = left + k;
                const uint64_t E = valid ? a[idx == pivotIdx ? right : idx] : ~0ull; const uint64_t ek = E >> 16;
                bool less = valid && ek < pk; const bool eq = valid && ek == pk;
                const unsigned long long eqMask = __ballot(eq);
                if (eqMask) { const unsigned long long bits = yqDraw(rs, __builtin_popcountll(eqMask)); if (eq) less = ((bits >> yqLanesBelow(eqMask)) & 1ull) != 0; }
                const unsigned long long lessMask = __ballot(less);
                const int rank = base + (int)yqLanesBelow(lessMask);
                if (valid) pos[k] = less ? rank : k;
                if (less) a2[rank] = E;
                base += __builtin_popcountll(lessMask);
            }
            nL = base;
            for (;;) {
                bool moved = false;
                for (int c0 = 0; c0 < m; c0 += 64) { const int k = c0 + lane; if (k < m) { const int p = pos[k], pp = pos[p]; if (pp != p) { pos[k] = pp; moved = true; } } }
                if (!__ballot(moved)) break;
            }
            const int rootNL = nL < m ? pos[nL] : 0;
            for (int j0 = nL; j0 <= m; j0 += 64) {
                const int j = j0 + lane;
                if (j <= m) { uint64_t v = P; if (j != nL) { const int idx = left + (j == m ? rootNL : pos[j]); v = a[idx == pivotIdx ? right : idx]; } a2[j] = v; }
            }
            for (int j0 = 0; j0 <= m; j0 += 64) { const int j = j0 + lane; if (j <= m) a[left + j] = a2[j]; }
        }
        nL = __builtin_amdgcn_readfirstlane(nL);
        const int store = left + nL;
        if (store + 1 < right) {                                          // the right part waits; the left part is sorted first (:449-452)
            if (lane == 0) { if (sp < stkCap) { stk[sp] = store + 1; stk[sp + 1] = right; } else { stk2[sp - stkCap] = store + 1; stk2[sp + 1 - stkCap] = right; } }
            sp += 2;
        }
        right = store - 1;
    }
}
// test entry (ygpu_selftest_primitives): array t = ent[off[t] .. off[t + 1]) sorted by waveSort with the generator seeded from seeds[5 t ..]; LDS: stack | entries | copy |
// positions
__global__ void __launch_bounds__(64) k_oqc_sort_test(uint64_t *ent, const uint32_t *off, const uint32_t *seeds, int *stack2, uint32_t count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sOqc[];
    if (blockIdx.x >= count) return;
    const int lane = (int)threadIdx.x; const uint32_t b = off[blockIdx.x]; const int n = (int)(off[blockIdx.x + 1] - b);
    int *const sStack = (int *)sOqc; uint64_t *const se = (uint64_t *)(sStack + YQ_STACK_LDS), *const se2 = se + n; int *const pos = (int *)(se2 + n);
    yoqc::Rand rs;
#pragma unroll
    for (int k = 0; k < 5; k++) rs.s[k] = seeds[5 * blockIdx.x + k];
    for (int i = lane; i < n; i += 64) se[i] = ent[b + i];
    __syncthreads();
    waveSort(se, se2, pos, n, sStack, 8, stack2 + 2ull * b + 8ull * blockIdx.x, rs, lane);      // (eight ints of stack in LDS: the part in HBM gets used)
    __syncthreads();
    for (int i = lane; i < n; i += 64) ent[b + i] = se[i];
}
// Reads with more than YQ_DEVICE_MAX clumps (a handful in a batch of 1 kbp reads) are not filtered here: they are handed to the host as they are, marked
// (primaryCount = 0xFFFF), and the host runs the same routine on them (oqc_core.h; a CPU core does such a read in a millisecond).
// the reads left to the host: all their clumps, in the hot path's order
__global__ void k_oqc_raw(OqcArgs A, const uint32_t *list, uint32_t count)
{
    YD_HIGH_PRIO();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t r = list[t], b = A.cs[r], n = A.cs[r + 1] - b; uint32_t nops = 0;
    for (uint32_t k = 0; k < n; k++) { yoqc::OutRec o; o.clump = (int)k; o.status = A.cl[b + k].status; o.mapQuality = 255; o.numSecondaries = 0; o.matchedPrimary = 0; o.pad = 0;
        A.out[b + k] = o; nops += A.cl[b + k].n_ops; }
    A.outCnt[r] = n; A.outOpsCnt[r] = nops; A.primCnt[r] = 0xFFFFu;
}
// filterBySimilarity + mapping quality (oqc_core.h finish, :571-692) with the wave: what a node contributes -- the primary it overlaps most and whether it is similar
// enough to be printed with it: a loop over the primaries and three double-precision divisions a node -- is independent of the other nodes and done a node a lane;
// what depends on the order of the nodes (second and third score of a primary, the order of the secondaries in the output) stays on the first lane, reading one
// word a node.  `prim` / `pa`: the primaries' copies and attributes, in LDS when the tables' place holds them.
__device__ __forceinline__ int waveFinish(yoqc::Run &X, int cnt, int bestNode, yoqc::CNode *prim, yoqc::PAttr *pa, yoqc::OutRec *out, int *primaryCount, int lane)
{
    using namespace yoqc;
    CNode *gn = X.S.nodes; OutRec *push = X.S.push; int *code = X.S.path; const ygpu_clump *cl = X.cl;
    const int primeCount = gn[bestNode].pathLength;
    if (lane == 0) {
        int pi = primeCount - 1;
        for (int p = bestNode; p >= 0; p = gn[p].bestPrev) {
            prim[pi] = gn[p];
            PAttr a; a.alignedQueryLength = 1 + gn[p].EQO - gn[p].SQO; a.numOutputSecondaries = 0; a.secondScore = 0; a.thirdScore = 0; pa[pi] = a;
            OutRec o; o.clump = gn[p].clump; o.status = 0; o.mapQuality = 255; o.numSecondaries = 0; o.matchedPrimary = (uint16_t)(pi + 1); o.pad = 0;
            push[primeCount - 1 - pi] = o;
            gn[p].clump = -1; pi--;
        }
    }
    __syncthreads();
    const double targetOverlap = (double)X.P.FBS_PSLength;
    for (int i = lane; i < cnt; i += 64) {
        const CNode cn = gn[i]; int c = -1;
        if (cn.clump >= 0) {
            const int curSQO = cn.SQO, curEQO = cn.EQO, curQLen = 1 + curEQO - curSQO; int maxOverlap = 0, maxIndex = 0;
            for (int k = 0; k < primeCount; k++) {
                const int pe = prim[k].EQO, ps = prim[k].SQO;
                const int e = curEQO < pe ? curEQO : pe, s = curSQO > ps ? curSQO : ps, overlap = 1 + e - s;
                if (overlap > maxOverlap) { maxOverlap = overlap; maxIndex = k; }
            }
            if (maxOverlap > 0) {
                c = maxIndex | (((int)cn.nodeScore & 0xFFFF) << 12);
                if (similarEnough(cn.nodeScore, prim[maxIndex].nodeScore, X.P.FBS_PSScore) && overlapsEnough(maxOverlap, curQLen, targetOverlap) && overlapsEnough(maxOverlap,
                    pa[maxIndex].alignedQueryLength, targetOverlap)) c |= 1 << 30;
            }
        }
        code[i] = c;
    }
    __syncthreads();
    int nPush = primeCount;
    if (lane == 0) {
        for (int i = 0; i < cnt; i++) {
            const int c = code[i];
            if (c < 0) continue;
            const int mi = c & 0xFFF; const int16_t ns = (int16_t)(uint16_t)((c >> 12) & 0xFFFF);
            PAttr a = pa[mi];
            if (ns > a.secondScore) { a.thirdScore = a.secondScore; a.secondScore = ns; } else if (ns > a.thirdScore) a.thirdScore = ns;      // memoPAsFromOverlappingNode :545-557
            if (c & (1 << 30)) {
                a.numOutputSecondaries += 1;
                if (X.P.FBS) { OutRec o; o.clump = gn[i].clump; o.status = 0; o.mapQuality = 255; o.numSecondaries = 0; o.matchedPrimary = (uint16_t)(mi + 1); o.pad = 0;
                    push[nPush++] = o; }
            }
            pa[mi] = a;
        }
    }
    nPush = __shfl(nPush, 0, 64);
    __syncthreads();
    // print order = the reverse of the push order; status and (primaries) mapping quality filled in on the way
    for (int k = lane; k < nPush; k += 64) {
        OutRec o = push[k]; const ygpu_clump &c = cl[o.clump];
        if (k < primeCount) { const PAttr a = pa[primeCount - 1 - k]; o.status = (uint8_t)(c.status | stPrimary);
            o.mapQuality = mapQuality((int)c.totScore, a.secondScore, a.thirdScore); o.numSecondaries = (uint16_t)a.numOutputSecondaries; }
        else o.status = c.status;
        out[nPush - 1 - k] = o;
    }
    *primaryCount = primeCount;
    return nPush;
}
// The rest of a read's run, once its survivors are known: nodes and tables, the graph loop, the filter.  LDS = true: nodes, table index, path and the first tables
// live in the workgroup's LDS (the surviving keys wait in HBM, the read's slice of the key array) and every access below is compiled as an LDS instruction; LDS =
// false: everything in the read's slices of the batch-wide HBM arrays.  (One body whose pointers may be either costs a `flat` access -- it waits for all of the
// wave's memory traffic -- at every step of the one-lane parts: the walk along the path, the tables' binary searches.)
// YQ_GRAPH_INLINE=0 (round 6, measured and dropped): the graph stage as a FUNCTION the kernel calls once a read instead of a part of its body.  Inlined twice (LDS /
// HBM work space) behind the sort and the duplicate scan it makes one body of 12 000 instructions whose wave-uniform state -- the parameters, thirteen work-space
// pointers, the arguments -- does not fit the scalar registers: 245 of them spill to lanes of vector registers (v_writelane / v_readlane, no memory).  As a function:
// no spill, but 616 bytes of scratch a lane for what is passed by reference, and k_oqc_wave 2.47 -> 2.93 ms a launch, the batch's slowest read 3.05 -> 3.37 ms
// (profiles/r06_postfilter_graph_noinline.txt): a spilled scalar costs one v_readlane, a scratch access a trip to memory.
#ifndef YQ_GRAPH_INLINE
#define YQ_GRAPH_INLINE 1
#endif
#if YQ_GRAPH_INLINE
#define YQ_GRAPH_FN __device__ __forceinline__
#else
#define YQ_GRAPH_FN __device__ __attribute__((noinline))
#endif
template <bool LDS>
YQ_GRAPH_FN void oqcGraph(const OqcArgs &A, const yoqc::Params &P, yoqc::Scratch S, unsigned char *sMain, unsigned mainBytes, uint32_t b, uint32_t r, int n, int cnt,
    int qlen, int lane, unsigned long long *tk)
{
    const bool prof = A.prof != nullptr; unsigned long long tPath = 0, tSucc = 0;
    if (LDS) {
        const yoqc::SortKey *lk = (const yoqc::SortKey *)sMain;
        for (int p = lane; p < cnt; p += 64) A.keys[b + p] = lk[p];
        __syncthreads();
        S.keys = A.keys + b;
        S.nodes = (yoqc::CNode *)sMain; S.tbl = (int *)(sMain + 40u * (unsigned)cnt); S.path = S.tbl + cnt;
        S.pool = S.path + cnt; S.poolCap = (int)((mainBytes - 48u * (unsigned)cnt) / 4u);      // the rest of the keys' place: tables
    }
    yoqc::Run X{P, A.cl + b, A.ops, S, 0, 0};
    for (int p = lane; p < cnt; p += 64) X.makeNode(p, A.G, qlen, -1);
    __syncthreads();
    if (lane == 0) for (int p = 0; p < cnt; p++) X.assignTable(p);          // bump allocation: sequential, a few instructions a node
    X.poolUsed = __shfl(X.poolUsed, 0, 64); X.pool2Used = __shfl(X.pool2Used, 0, 64);
    __syncthreads();
    for (int p = lane; p < cnt; p += 64) X.fillTable(p);
    __syncthreads();
    if (prof) tk[4] = wall_clock64();
    int bestScore = YQ_WORST, bestNode = -1, startj = 1;
    for (int i = 0; i < cnt; i++) {                                         // :973-1063
        unsigned long long t0 = 0; if (prof) t0 = wall_clock64();
        if (lane == 0) X.cachePath(i);
        __syncthreads();
        if (prof) { const unsigned long long t1 = wall_clock64(); tPath += t1 - t0; t0 = t1; }
        const int leftSQO = X.S.nodes[i].SQO; int first = -1;
        for (int base = startj; base < cnt; base += 64) {
            const int j = base + lane;
            const bool far = j < cnt && ((int)X.S.nodes[j].SQO - leftSQO) >= P.minNonOverlap;
            if (far) X.relax(i, j);
            const unsigned long long fm = __ballot(far);
            if (first < 0 && fm) first = base + __builtin_ctzll(fm);
        }
        if (startj < cnt) startj = first >= 0 ? first : cnt;                // (the reference moves startj only inside the loop over j)
        __syncthreads();
        X.considerBest(i, bestScore, bestNode);                              // (uniform: every lane reads the same node)
        if (prof) tSucc += wall_clock64() - t0;
    }
    if (prof) tk[5] = wall_clock64();
    int primary = 0; uint32_t m;
    {
        // the primaries' copies and attributes go where the tables were (the graph loop is over) when they fit
        const int primeCount = X.S.nodes[bestNode].pathLength;
        const unsigned at = (48u * (unsigned)cnt + 15u) & ~15u; constexpr unsigned nodeB = (unsigned)sizeof(yoqc::CNode), attrB = (unsigned)sizeof(yoqc::PAttr);
        static_assert(sizeof(yoqc::CNode) % 4 == 0 && sizeof(yoqc::CNode) <= 40, "nodes, table index and path take 48 bytes a survivor");
        if (primeCount > 4095) m = lane == 0 ? (uint32_t)X.finish(cnt, bestNode, A.out + b, &primary) : 0u;
        else if (LDS && at + (nodeB + attrB) * (unsigned)primeCount <= mainBytes) m = (uint32_t)waveFinish(X, cnt, bestNode, (yoqc::CNode *)(sMain + at),
            (yoqc::PAttr *)(sMain + at + nodeB * (unsigned)primeCount), A.out + b, &primary, lane);
        else m = (uint32_t)waveFinish(X, cnt, bestNode, X.S.prim, X.S.pa, A.out + b, &primary, lane);
        m = (uint32_t)__shfl((int)m, 0, 64); primary = __shfl(primary, 0, 64);
    }
    __syncthreads();
    uint32_t nops = 0;
    for (uint32_t k = (uint32_t)lane; k < m; k += 64) nops += A.cl[b + (uint32_t)A.out[b + k].clump].n_ops;
    nops = waveTotalSumU(nops);
    if (lane == 0) {
        A.outCnt[r] = m; A.outOpsCnt[r] = nops; A.primCnt[r] = (uint32_t)primary;
        if (prof) {
            const unsigned long long t6 = wall_clock64(); int cls = 0; while (n > kOqcCapN[cls]) cls++;
            unsigned long long *pp = A.prof + 32 * cls;
            const unsigned long long ph[7] = {tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tPath, tSucc, t6 - tk[5]};
            for (int k = 0; k < 7; k++) { atomicAdd(&pp[k], ph[k]); atomicMax(&pp[16 + k], ph[k]); }
            atomicAdd(&pp[7], 1ull); atomicAdd(&pp[8], (unsigned long long)n); atomicAdd(&pp[9], (unsigned long long)cnt);
            // the slowest read: ticks, clumps, survivors
            atomicMax(&pp[10], ((t6 - tk[0]) << 24) | ((unsigned long long)n << 12) | (unsigned long long)(cnt < 4095 ? cnt : 4095));
        }
    }
}
// One read per workgroup of one wave.  LDS (dynamic, `ldsBytes`): [sort stack | keys ...] during the sort and the duplicate scan, then [sort stack | nodes, tbl,
// path | running-sum tables ...] over the keys' place (the surviving keys are parked in HBM for the moment the nodes are made): a quarter of a read's clumps survive
// the duplicate scan, so most of the tables -- a lane's binary searches in the graph loop -- find room in LDS as well; what does not fit lives in the read's slices
// of the batch-wide HBM arrays.
__global__ void __launch_bounds__(64) k_oqc_wave(OqcArgs A, const uint32_t *list, uint32_t count, unsigned ldsBytes)
{
    YD_HIGH_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char sOqc[];
    if (blockIdx.x >= count) return;
    const int lane = (int)threadIdx.x;
    const uint32_t r = list[blockIdx.x], b = A.cs[r]; const int n = (int)(A.cs[r + 1] - b);
    const int qlen = (int)A.qlen[r];
    int *const sStack = (int *)sOqc; uint32_t *const sThr = (uint32_t *)(sStack + YQ_STACK_LDS); unsigned char *const sMain = (unsigned char *)(sThr + YQ_THR_LDS);
    yoqc::Params P = A.P;
    if (P.bppN <= YQ_THR_LDS) { if (lane < P.bppN) sThr[lane] = A.P.bppThr[lane]; P.bppThr = sThr; }
    const unsigned mainBytes = ldsBytes - (unsigned)(sMain - sOqc);
    yoqc::Scratch S;
    S.stack = sStack; S.stackCap = YQ_STACK_LDS; S.stack2 = A.stack + 4ull * b + 8ull * r;
    S.pool = A.pool + A.poolOff[r]; S.poolCap = 0; S.pool2 = S.pool;            // (the LDS pool is set up once the nodes are placed)
    S.prim = A.prim + b; S.pa = A.pa + b; S.push = A.push + b;
    const bool keysInLds = 16u * (unsigned)n <= mainBytes;
    S.keys = keysInLds ? (yoqc::SortKey *)sMain : A.keys + b;
    S.nodes = A.nodes + b; S.tbl = A.pfxOff + b; S.path = A.path + b;             // set for good once the number of survivors is known
    yoqc::Run X{P, A.cl + b, A.ops, S, 0, 0};
    unsigned long long tk[8]; const bool prof = A.prof != nullptr;
    if (prof) tk[0] = wall_clock64();
    yoqc::Rand rs; { const uint32_t w = lane < 5 ? A.seeds[5ull * r + (uint32_t)lane] : 0u;
#pragma unroll
        for (int k = 0; k < 5; k++) rs.s[k] = (uint32_t)__shfl((int)w, k, 64); }
    const bool sortOnWave = keysInLds && YQ_SORT_LDS * (unsigned)n <= mainBytes && n < 65536;
    if (sortOnWave) {
        // sort entries (key << 16 | clump: the key has 48 bits) behind the key records' place, then the records written out in sorted order
        uint64_t *const se = (uint64_t *)(sMain + 16u * (unsigned)n), *const se2 = se + n; int *const pos = (int *)(se2 + n);
        for (int i = lane; i < n; i += 64) se[i] = (X.clumpKey(i, qlen) << 16) | (uint64_t)(uint32_t)i;
        __syncthreads();
        if (prof) tk[1] = wall_clock64();
        waveSort(se, se2, pos, n, sStack, YQ_STACK_LDS, X.S.stack2, rs, lane);
        __syncthreads();
        for (int p = lane; p < n; p += 64) { const uint64_t v = se[p]; yoqc::SortKey k; k.key = v >> 16; k.clump = (int)(v & 0xffffu); k.pad = 0; ((yoqc::SortKey *)sMain)[p] = k; }
    } else {                                                             // (a read whose sort does not fit the LDS it was given: the one-lane routine)
        for (int i = lane; i < n; i += 64) X.makeKey(i, qlen);
        __syncthreads();
        if (prof) tk[1] = wall_clock64();
        if (lane == 0) {
            if (keysInLds) yoqc::Run::sortRange((yoqc::SortKey *)sMain, n, sStack, YQ_STACK_LDS, X.S.stack2, rs);
            else yoqc::Run::sortRange(A.keys + b, n, sStack, YQ_STACK_LDS, X.S.stack2, rs);
        }
    }
    __syncthreads();
    if (prof) tk[2] = wall_clock64();
    // deleteSubsumedDups: the scan behind node i, 64 candidates at a time; it ends at the first live candidate whose EQO exceeds the node's.  The nodes are taken 64 at a
    // time into registers (a lane each): node i is then read with v_readlane, the candidates of its own block are compared and killed in registers, and only a scan
    // that runs past the block's end goes to LDS -- most end within a few candidates, and a node costs some forty instructions instead of two trips to LDS and a barrier.
    int cnt = 0;
    for (int blk = 0; blk < n; blk += 64) {
        uint64_t kk = 0; int kc = -1;                                      // (lanes past the end: dead)
        if (blk + lane < n) { const yoqc::SortKey t = X.S.keys[blk + lane]; kk = t.key; kc = t.clump; }
        const int lim = n - blk < 64 ? n - blk : 64;
        for (int l = 0; l < lim; l++) {
            const int ci = __builtin_amdgcn_readlane(kc, l);
            if (ci < 0) continue;
            const uint64_t ck = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(kk >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)kk, l);
            const int curEQO = yoqc::keyEQO(ck);
            // survivors compacted in place (cnt <= blk + l: a place already in registers)
            if (lane == 0) { yoqc::SortKey ki; ki.key = ck; ki.clump = ci; ki.pad = 0; X.S.keys[cnt] = ki; }
            cnt++;
            {
                const bool live = lane > l && kc >= 0, ends = live && yoqc::keyEQO(kk) > curEQO;
                const unsigned long long em = __ballot(ends);
                const int stop = em ? __builtin_ctzll(em) : 64;
                if (live && lane < stop && X.dupKillK(ck, ci, kk, kc)) kc = ~kc;
                if (em) continue;
            }
            for (int base = blk + 64; base < n; base += 64) {
                const int j = base + lane; bool live = false, ends = false; yoqc::SortKey kj; kj.key = 0; kj.clump = -1;
                if (j < n) { kj = X.S.keys[j]; live = kj.clump >= 0; ends = live && yoqc::keyEQO(kj.key) > curEQO; }
                const unsigned long long em = __ballot(ends);
                const int stop = em ? __builtin_ctzll(em) : 64;
                if (live && lane < stop && X.dupKillK(ck, ci, kj.key, kj.clump)) X.S.keys[j].clump = ~kj.clump;
                if (em) break;
            }
        }
        __syncthreads();
    }
    if (prof) tk[3] = wall_clock64();
    // nodes, table index and path take the keys' place when they fit (48 bytes a survivor: always, but for a read of the last class nearly all of whose clumps survive)
    if (!A.graphInHbm && keysInLds && 48u * (unsigned)cnt <= mainBytes) oqcGraph<true>(A, P, X.S, sMain, mainBytes, b, r, n, cnt, qlen, lane, tk);
    else oqcGraph<false>(A, P, X.S, sMain, mainBytes, b, r, n, cnt, qlen, lane, tk);
}
// the printed clumps of read r, in print order, with their ops copied behind one another: out clump k of the read = fClumps[outStart[r] + k].  A wave per read, a lane
// per clump (a read handed to the host unfiltered brings hundreds): the ops' places from a scan of the clumps' op counts across the wave.
__global__ void __launch_bounds__(256) k_oqc_gather(OqcArgs A, const uint32_t *outStart, const uint32_t *opsStart, ygpu_out_clump *fClumps, uint32_t *fOps)
{
    YD_HIGH_PRIO();
    const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; const int lane = (int)(threadIdx.x & 63);
    if (r >= A.nReads) return;
    const uint32_t b = A.cs[r], m = A.outCnt[r], d = outStart[r]; uint32_t od = opsStart[r]; const uint16_t pc = (uint16_t)A.primCnt[r];
    for (uint32_t k0 = 0; k0 < m; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane; const bool have = k < m;
        yoqc::OutRec o; ygpu_clump c; uint32_t nops = 0;
        if (have) { o = A.out[b + k]; c = A.cl[b + (uint32_t)o.clump]; nops = c.n_ops; }
        uint32_t incl = nops;
#pragma unroll
        for (int s2 = 1; s2 < 64; s2 <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, s2, 64); if (lane >= s2) incl += v; }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        if (have) {
            const uint32_t dst = od + incl - nops; const uint32_t *src = A.ops + c.op_start;
            for (uint32_t i = 0; i < nops; i++) fOps[dst + i] = src[i];
            c.op_start = dst;
            ygpu_out_clump f; f.c = c; f.status = o.status; f.mapQuality = o.mapQuality; f.numSecondaries = o.numSecondaries; f.matchedPrimary = o.matchedPrimary;
                f.primaryCount = pc;
            fClumps[d + k] = f;
        }
        od += total;
    }
}
