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
    const uint32_t *cs; const ygpu_clump *cl; const uint32_t *ops; const uint8_t *fwd; const uint32_t *readOff; uint32_t nReads;
    const unsigned long long *poolOff;
    yoqc::SortKey *keys; int *stack; yoqc::CNode *nodes, *prim; yoqc::PAttr *pa; int *pfxOff, *path, *pool; yoqc::OutRec *push, *out;
    uint32_t *outCnt, *outOpsCnt; uint32_t *primCnt;
    int devMax;                        // reads with more clumps are left to the host (YQ_DEVICE_MAX; YGPU_OQC_MAX lowers it in tests)
    unsigned long long *prof;          // YGPU_OQC_PROF=1: 100 MHz ticks per step of k_oqc_wave, summed over the reads (keys, sort, duplicate scan, nodes + tables, walk along the path, successors, finish) and per class
};
// What the routine costs on a GPU, measured: one read per lane with its work space in HBM took 75 ms a batch (a few thousand DEPENDENT accesses a read, microseconds
// each); one read per wave, first lane only, with the work space in LDS still 39 ms for the reads of 320..640 clumps -- the graph loop is quadratic in the nodes that survive the
// duplicate scan, and reads in repeats have hundreds.  So a read gets a WAVE, the work space in LDS, and the steps oqc_core.h marks as independent run on its
// 64 lanes: the scan behind a node in the duplicate removal (64 candidates at a time, the ballot finds where the scan ends), the nodes and their running-sum
// tables, and the successors of a node in the graph loop (each lane relaxes a different successor: it reads the node and its path and writes its own successor
// only).  What must stay in the reference's order stays on the first lane: the sort (it consumes the read's random bits in comparison order), the walk along
// the best path before each node's successors, the choice of the best node, the similarity filter.
// Reads come in classes by their number of clumps (the LDS a workgroup gets is fixed at launch): keys + sort stack, then nodes + table index + path + tables share it.
#define YQ_NCLASS 4
#define YQ_DEVICE_MAX 448            // clumps of the largest read the device stage filters (see k_oqc_raw)
#define YQ_STACK_LDS 128              // ints of the sort's stack kept in LDS (depth ~2 log2 n ranges); deeper recursion continues in HBM
#define YQ_THR_LDS 64                 // break point thresholds copied to LDS (every lane searches them for every successor it relaxes); a longer table stays in HBM
#define YQ_LDS_MAX 65536u
__device__ __constant__ const int kOqcCapN[YQ_NCLASS] = {112, 224, YQ_DEVICE_MAX, 0x7fffffff};      // clumps a read of the class may have (the last class: left to the host)
__host__ __device__ inline unsigned oqcLdsBytes(int capN) { return 64u * (unsigned)capN + 4u * (YQ_STACK_LDS + YQ_THR_LDS) + 64u; }
// classes of the reads with two or more clumps (lists[c * nReads ...], cnt[c]); reads with one clump are settled here (GraphPath.cpp:907-916); ints of running-sum
// tables a read may need in HBM: 2 n_ops + 3 per clump of the read (every clump's table built)
__global__ void k_oqc_classify(OqcArgs A, unsigned long long *need, uint32_t *lists, unsigned int *cnt)
{
    YD_HIGH_PRIO();
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; const int lane = (int)(threadIdx.x & 63);
    unsigned long long v = 0; int cls = -1;
    if (r < A.nReads) {
        const uint32_t b = A.cs[r], n = A.cs[r + 1] - b;
        if (n >= 2) { for (uint32_t c = b; c < b + n; c++) v += 2ull * (unsigned long long)A.cl[c].n_ops + 3ull; cls = 0; while (n > (uint32_t)kOqcCapN[cls]) cls++; if (n > (uint32_t)A.devMax) cls = YQ_NCLASS - 1; }
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
// The sort cannot be spread over lanes -- every comparison's outcome moves elements the next comparison reads, and ties consume the read's random bits in
// comparison order -- and a single lane of a GPU does about two million of its iterations a second (1 300 cycles each: ~60 dependent instructions and three LDS
// round trips, measured alike with one such wave on a SIMD and with five, with flat and with ds_ instructions, on lane 0 of a wave and with sixteen reads a wave in
// a kernel of its own).  It is 80 % of this stage: 0.2 ms for a read of 50 clumps, 2 ms for one of 550 -- and a kernel lasts as long as its slowest read.  Reads
// with more than YQ_DEVICE_MAX clumps (1 % of the reads of a 1 kbp batch) are therefore not filtered here: they are handed to the host as they are, marked
// (primaryCount = 0xFFFF), and the host runs the same routine on them (oqc_core.h; a CPU core does such a read in ~0.3 ms).
// the reads left to the host: all their clumps, in the hot path's order
__global__ void k_oqc_raw(OqcArgs A, const uint32_t *list, uint32_t count)
{
    YD_HIGH_PRIO();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint32_t r = list[t], b = A.cs[r], n = A.cs[r + 1] - b; uint32_t nops = 0;
    for (uint32_t k = 0; k < n; k++) { yoqc::OutRec o; o.clump = (int)k; o.status = A.cl[b + k].status; o.mapQuality = 255; o.numSecondaries = 0; o.matchedPrimary = 0; o.pad = 0; A.out[b + k] = o; nops += A.cl[b + k].n_ops; }
    A.outCnt[r] = n; A.outOpsCnt[r] = nops; A.primCnt[r] = 0xFFFFu;
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
    const uint32_t o = A.readOff[r]; const int qlen = (int)(A.readOff[r + 1] - o);
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
    unsigned long long tk[8]; const bool prof = A.prof != nullptr; unsigned long long tPath = 0, tSucc = 0;
    if (prof) tk[0] = wall_clock64();
    for (int i = lane; i < n; i += 64) X.makeKey(i, qlen);
    __syncthreads();
    if (prof) tk[1] = wall_clock64();
    // (the seed's five words a lane each: 80 code bytes from HBM, one after the other on a single lane, were a third of a light read's sort)
    yoqc::Rand rs; { const uint32_t w = lane < 5 ? yoqc::seedWord(A.fwd + o, qlen, lane) : 0u;
#pragma unroll
        for (int k = 0; k < 5; k++) rs.s[k] = (uint32_t)__shfl((int)w, k, 64); }
    if (lane == 0) {
        if (keysInLds) yoqc::Run::sortRange((yoqc::SortKey *)sMain, n, sStack, YQ_STACK_LDS, X.S.stack2, rs);
        else yoqc::Run::sortRange(A.keys + b, n, sStack, YQ_STACK_LDS, X.S.stack2, rs);
    }
    __syncthreads();
    if (prof) tk[2] = wall_clock64();
    // deleteSubsumedDups: the scan behind node i, 64 candidates at a time; it ends at the first live candidate whose EQO exceeds the node's
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        const yoqc::SortKey ki = X.S.keys[i];                              // (uniform)
        if (ki.clump < 0) continue;
        const int curEQO = yoqc::keyEQO(ki.key);
        if (lane == 0) X.S.keys[cnt] = ki;                                 // survivors compacted in place (cnt <= i)
        cnt++;
        for (int base = i + 1; base < n; base += 64) {
            const int j = base + lane; bool live = false, ends = false;
            if (j < n) { const yoqc::SortKey kj = X.S.keys[j]; live = kj.clump >= 0; ends = live && yoqc::keyEQO(kj.key) > curEQO; }
            const unsigned long long em = __ballot(ends);
            const int stop = em ? __builtin_ctzll(em) : 64;
            if (live && lane < stop && X.dupKill(ki.key, ki.clump, j)) X.S.keys[j].clump = ~X.S.keys[j].clump;
            if (em) break;
        }
        __syncthreads();
    }
    if (prof) tk[3] = wall_clock64();
    // nodes, table index and path take the keys' place when they fit (48 bytes a survivor); the surviving keys wait in HBM (the read's slice of the key array)
    const bool nodesInLds = 48u * (unsigned)cnt <= mainBytes;
    if (nodesInLds && keysInLds) {
        for (int p = lane; p < cnt; p += 64) A.keys[b + p] = X.S.keys[p];
        __syncthreads();
        X.S.keys = A.keys + b;
    }
    if (nodesInLds) {
        X.S.nodes = (yoqc::CNode *)sMain; X.S.tbl = (int *)(sMain + 40u * (unsigned)cnt); X.S.path = X.S.tbl + cnt;
        if (keysInLds) { X.S.pool = X.S.path + cnt; X.S.poolCap = (int)((mainBytes - 48u * (unsigned)cnt) / 4u); }      // the rest of the keys' place: tables
    }
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
    if (lane == 0) {
        int primary = 0;
        const uint32_t m = (uint32_t)X.finish(cnt, bestNode, A.out + b, &primary);
        uint32_t nops = 0;
        for (uint32_t k = 0; k < m; k++) nops += A.cl[b + (uint32_t)A.out[b + k].clump].n_ops;
        A.outCnt[r] = m; A.outOpsCnt[r] = nops; A.primCnt[r] = (uint32_t)primary;
        if (prof) {
            const unsigned long long t6 = wall_clock64(); int cls = 0; while (n > kOqcCapN[cls]) cls++;
            unsigned long long *pp = A.prof + 16 * cls;
            atomicAdd(&pp[0], tk[1] - tk[0]); atomicAdd(&pp[1], tk[2] - tk[1]); atomicAdd(&pp[2], tk[3] - tk[2]); atomicAdd(&pp[3], tk[4] - tk[3]); atomicAdd(&pp[4], tPath); atomicAdd(&pp[5], tSucc); atomicAdd(&pp[6], t6 - tk[5]);
            atomicAdd(&pp[7], 1ull); atomicAdd(&pp[8], (unsigned long long)n); atomicAdd(&pp[9], (unsigned long long)cnt); atomicMax(&pp[10], t6 - tk[0]);
        }
    }
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
            ygpu_out_clump f; f.c = c; f.status = o.status; f.mapQuality = o.mapQuality; f.numSecondaries = o.numSecondaries; f.matchedPrimary = o.matchedPrimary; f.primaryCount = pc;
            fClumps[d + k] = f;
        }
        od += total;
    }
}
