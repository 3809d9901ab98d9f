// ygpu.hip -- device context, stage orchestration and the C-ABI of include/yaha_hip.h.
//
// HBM layout per context (one or more per GPU -- ygpu_clone shares the index image; reads shard across contexts and GPUs, no collective):
//   index   : packed 4-bit reference, startingOffs[4^L+1], ROA[totalMatches]           (resident for the whole run)
//   batch   : forward + reverse-complement codes (1 B/base), read offsets, k-mer offsets
//   stage arenas, grown on demand and reused across batches:
//     A1  posS/posC/posRsI per k-mer  -> exclusive scan -> hit offsets
//     A2  64-bit hit keys (double buffer for the radix sort) -> fragment array (16 B each)
//     A3  region starts, multi-fragment region list
//     A4  clump records + clump fragment lists (atomic arenas), per-region counts -> creation-order ranks
//     A5-8 default band: joint records + gap-op arena, root states + phase-1 lists, extension problems / results, extension trace
//          strips (128 B per 10 rows, sized by row bound; the extension ops are written into them), split-root scratch;
//          general path and leftovers: per-wave scratch (trace strip, DP temp list, frame stack with edit-list buffers); output arenas
//   results : clump records in QS->clumps order, ops arena, clump_start per read
// Every stage is a handful of launches on one stream; sizes that the next stage needs cross the PCIe as single words.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include "chain.h"
#include "chain_lanes.h"
#include "seed.h"
#include "scan.h"
#include "segsort.h"
#include "phase_lanes.h"
#include "gap_band_lanes.h"
#include "ext_lanes_pk.h"
#include "split_lanes.h"
#include "dp_stage.h"
#include "oqc_stage.h"

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return YGPU_ENODEV; } } while (0)

// Every kernel launch is followed by a check of the submit status: a launch the runtime rejects (too much LDS, a grid that is too large, a code object
// for another architecture) would otherwise leave the stage running on unwritten buffers, and the later stream synchronisation reports nothing.
#define KL(kern, grid, block, shmem, st, ...) do { hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__); hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { ctx->err = std::string("launch of " #kern " failed: ") + hipGetErrorString(e_); return YGPU_ENODEV; } } while (0)

#define YD_MAX_CHUNK_EV 16
namespace {
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t bytes, bool keep = false, hipStream_t st = 0)
    {
        if (bytes <= cap) return 0;
        size_t ncap = bytes + bytes / 4 + 256; void *np = nullptr;
        if (hipMalloc(&np, ncap) != hipSuccess) return -1;
        if (keep && p && cap) { hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st); hipStreamSynchronize(st); }
        if (p) hipFree(p);
        p = np; cap = ncap; return 0;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    // exactly `bytes` (no growth margin), the old buffer freed FIRST: for buffers sized against what is free on the device -- the request may then reuse the
    // buffer's own memory.  The contents are lost; on failure the buffer is empty.
    int ensureExact(size_t bytes) { if (bytes <= cap) return 0; release(); void *np = nullptr; if (hipMalloc(&np, bytes) != hipSuccess) return -1; p = np; cap = bytes; return 0; }
    template <class T> T *as() const { return (T *)p; }
};
enum { CNT_NMULTI = 0, CNT_MAXN, CNT_CLUMPS, CNT_CFRAGS, CNT_QCHAIN, CNT_QALIGN, CNT_OUTCLUMPS, CNT_OUTOPS, CNT_QDP, CNT_DPOPS, CNT_NBIG, CNT_QBIG, CNT_STATEOPS, CNT_EXTOPS, CNT_QEXT, CNT_SLOW, CNT_NDP, CNT_NDP16, CNT_GAPOPS, CNT_NSMALL, CNT_NB12, CNT_NB16, CNT_SEGC /* YD_SEG_NCLASS + 1: the segments of the workgroup-sort classes, the long ones */, CNT_NFRAGS = CNT_SEGC + 16 /* + the look-back's flag */, CNT_NREG = CNT_NFRAGS + 2 /* + flag */, CNT_SCANFAIL = CNT_NREG + 2 /* raised by a look-back of scan.h that gave up */, CNT_N = CNT_NREG + 4 };
// the first T_TOP entries partition a run; the rest are sub-intervals of align_dp (lane-extension pipeline)
enum { T_SEED = 0, T_SORT, T_FRAGS, T_CHAIN, T_ALIGN, T_LAYOUT, T_TOP, T_P1 = T_TOP, T_XROWS, T_XTRACE, T_P3, T_XROWS_DEV, T_XROWS_PK, T_N };
std::atomic<int> gCtxPerDevice[64];      // live contexts per device of this process: they share the device's free memory
// One rows launch at a time per device (YGPU_ROWS_SERIAL): a context's main rows launch waits for the one launched before it on the device, whichever context that
// was -- two of them side by side take the whole chip between them and leave the other batches' kernels nothing, which is what the half-size launch is there to avoid.
std::mutex gRowsMu[64]; hipEvent_t gRowsEv[64][4]; bool gRowsEvValid[64][4]; unsigned long long gRowsSeq[64];      // (a ring of events: YGPU_ROWS_SERIAL=k lets k launches overlap)
std::atomic<int> gActiveRuns[64];        // contexts of this process inside ygpu_run on the device right now: a rows launch shares the device when there are two or more
const char *const kStageNames[T_N] = {"seed_lookup", "hit_sort", "fragments_regions", "chain", "align_dp", "layout", "align_p1_gapfill", "ext_rows", "ext_trace", "align_p3_score_split", "ext_rows_device_clock", "ext_rows_packed16"};      // the last one is a flag, not a time: 1 when k_ext_rows_pk ran (ext_lanes_pk.h)
}  // namespace

// What the post-filter's host code works with in place of the context: the second stream, its own pinned slot, wait event, look-back words and message -- the
// stage runs on a SNAPSHOT of a batch's results (ygpu_postfilter_snapshot) and may therefore run on a thread of its own while the context itself is already
// uploading and running the next batch.  (Same member names as the context's, so the macros and the small helpers below serve both.)
struct PfSide {
    std::string err; hipStream_t stream = nullptr; uint32_t *pinned = nullptr; hipEvent_t evSync = nullptr; DevBuf scanState, counters; int device = 0;
};

struct ygpu_ctx {
    int device = 0; hipStream_t stream = nullptr; DevParams P{}; std::string err; int nCU = 256;
    DevBuf dBases, dSO, dROA, dLow;
    // batch
    uint32_t nReads = 0; int maxQ = 0; uint64_t totalBases = 0; uint32_t nKmers = 0;
    std::vector<uint32_t> hReadOff, hKmerOff;
    DevBuf dFwd, dRev, dFwd4, dRev4, dReadOff, dKmerOff;
    // arenas
    DevBuf bigB, bigE;
    DevBuf posS, posC, posRsI, hitOff, expandStart, keysA, keysB, segOff, isHead, tileState, frags, regStart, multiList, smallList, bigList, regionCount, regionBase;
    DevBuf clumps, clumpFrags, clumpFrags0, order, rootPush, rootBase, outClumps, outClumps2, outOps, outRoot, outPush, dstIdx, readCount, readStart;
    DevBuf counters, ctr, errFlag, cubTemp, scanState, bucketWork, scratchAlign, scratchChain, dpProbs, dpRes, dpOps;
    DevBuf segLists, subB, subE, subLists, subBigB, subBigE, sub2B, sub2E, sub2Lists, sub3B, sub3E, sub3Lists, kmerParts, rootState, stateOps, extProbs, rowsBound, stripOff, extRes, extTrace, chunkCnt, cubTemp2, memoKeys, memoCount, probs2, rowsBound2, stripOff2, extRes2, extTrace2, splitScratch, fallList, keys2a, keys2b, vals2a, vals2b, extKeys, extVals, extKeys2, extOrder, slowList, gapScratch, jointCount, jointBase, joints, sortKeys, sortVals, sortKeys2, sortVals2, gapOps;
    bool evUsed[16] = {false}; double traceT = 0; hipStream_t stream2 = nullptr; hipEvent_t evChunk[YD_MAX_CHUNK_EV], evTail; bool sharedIndex = false; bool counted = false; bool parked = false; long long traceBudgetBlocks = 0; uint32_t lastClumpSlots = 0; bool keepAllFrags = false; DevBuf rowsClock; unsigned long long hRowsClock[2] = {0, 0}; int laneChunks = 0; int segSort = 2; uint32_t segSortMax = YD_SEGSORT_MAX; int splitLanes = 1; int rows2PerCU = 0; int alignWavesPerCU = 0; int laneExt = 1; std::vector<unsigned long long> hStripOff; int runsDone = 0; double traceRatio = 0.0, opsRatio = 0.03; int statRanges = 0, statAttempts = 0; double statT0 = 0; DevBuf waveChunks, extOps, traceCnt;
    // post-filter stage (oqc_stage.h)
    DevBuf oqProf, oqLists, oqClsCnt, oqThr, oqSeqStart, oqSeqLen, oqNeed, oqPoolOff, oqKeys, oqStack, oqNodes, oqPrim, oqPA, oqPfx, oqPath, oqPool, oqPush, oqOut, oqOutCnt, oqOutOps, oqPrimCnt, oqOutStart, oqOpsStart, oqFClumps, oqFOps;
    bool oqSet = false, oqDone = false; yoqc::Params oqP{}; yoqc::Seqs oqG{}; uint32_t nFOut = 0, nFOps = 0;
    // the snapshot the stage works on (taken by the context's thread) and the copy of its sizes the stage runs with (its own thread)
    PfSide pf; std::atomic<bool> pfSnap{false}; hipEvent_t evSnap = nullptr; uint32_t snapN = 0, snapC = 0, snapOps = 0, pfN = 0; ygpu_counters pfCounters{};
    DevCounters *snapCtr = nullptr; DevCounters snapCtrPlain{}; unsigned long long snapHits = 0, snapFrags = 0, snapRegions = 0;
    DevBuf oqCs, oqCl, oqOpsIn, oqSeeds, oqQlen;
    // stage state
    uint32_t hOutCounts[2] = {0, 0}, hOutEf = 0; bool hOutValid = false;
    uint32_t nHits = 0, nFrags = 0, nRegions = 0, nMulti = 0, nSmall = 0, nBig = 0, maxN = 0, nClumpSlots = 0, nClumps = 0, nClumpFrags = 0, nOut = 0, nOutOps = 0;
    int stageDone = 0;     // 0 none, 1 fragments, 2 chain, 3 all
    // host results
    std::vector<uint32_t> hClumpStart, hOps, hClumpFragStart, hClumpRS, hDpOps; std::vector<ygpu_clump> hClumps; std::vector<ygpu_fragment> hFrags, hClumpFrags;
    std::vector<ygpu_dp_result> hDpRes; ygpu_counters hCounters{};
    // asynchronous tickets (ygpu_submit / ygpu_wait): one worker thread per context, started on first use
    std::thread worker; std::mutex aMu; std::condition_variable aCv; const ygpu_read_batch *aBatch = nullptr; uint64_t aTicket = 0; int aRc = 0; bool aOpen = false, aDone = false, aQuit = false, aWaiting = false; ygpu_result_batch aOut{};
    // timing
    long long lastFall = -1; unsigned int hFall = 0; uint32_t *pinned = nullptr; hipEvent_t evSync = nullptr; hipEvent_t ev[T_N][2]; float ms[T_N] = {0}; float totalMs = 0; const char *names[T_N]; bool rowsPacked = false;
};

static DevBatch devBatch(ygpu_ctx *c) { DevBatch b; b.fwd = c->dFwd.as<uint8_t>(); b.rev = c->dRev.as<uint8_t>(); b.readOff = c->dReadOff.as<uint32_t>(); b.nReads = c->nReads; return b; }
static inline unsigned gridFor(uint64_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// Exclusive sums and orderings of the hot path: scan.h (own kernels; one launch a scan, two an ordering; their work words clean themselves up)
template <class T, class C> static int ownScanT(C *ctx, const T *in, T *out, uint32_t n, hipStream_t st)
{
    if (n == 0) return 0;
    const size_t need = scanStateBytes(n);
    if (ctx->scanState.cap < need) {                                         // (zeroed when it is made; every launch leaves it zero)
        if (ctx->scanState.ensure(std::max<size_t>(need, 1u << 16))) { ctx->err = "hipMalloc(scan state)"; return YGPU_ENOMEM; }
        HIPCHK(hipMemsetAsync(ctx->scanState.p, 0, ctx->scanState.cap, st));
    }
    hipLaunchKernelGGL((k_scan_excl<T>), dim3(scanTiles(n, (int)sizeof(T))), dim3(YD_SCAN_BS), 0, st, in, out, n, (unsigned long long *)ctx->scanState.p, (unsigned int *)ctx->counters.p + CNT_SCANFAIL);
    hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ctx->err = std::string("launch of k_scan_excl failed: ") + hipGetErrorString(e_); return YGPU_ENODEV; }
    return 0;
}
template <class C> static int cubScan(C *ctx, const uint32_t *in, uint32_t *out, uint32_t n) { return ownScanT<uint32_t>(ctx, in, out, n, ctx->stream); }
template <class C> static int cubScan64(C *ctx, const unsigned long long *in, unsigned long long *out, uint32_t n) { return ownScanT<unsigned long long>(ctx, in, out, n, ctx->stream); }
// order[] = the items' values grouped by bucket((key - sub) >> shift), ascending; vals == nullptr: the values are the items' indices + valBase
static int bucketOrder(ygpu_ctx *ctx, const uint32_t *keys, const uint32_t *vals, uint32_t valBase, uint32_t n, uint32_t sub, int shift, uint32_t nb, uint32_t *outVals, hipStream_t st)
{
    if (n == 0) return 0;
    nb = std::min<uint32_t>(std::max<uint32_t>(nb, 1u), YD_BKT_MAX);
    if (!ctx->bucketWork.p) { if (ctx->bucketWork.ensure(bucketWorkBytes())) { ctx->err = "hipMalloc(bucket work)"; return YGPU_ENOMEM; } HIPCHK(hipMemsetAsync(ctx->bucketWork.p, 0, ctx->bucketWork.cap, st)); }
    const unsigned grid = (unsigned)((n + YD_BKT_TILE - 1) / YD_BKT_TILE);
    hipLaunchKernelGGL(k_bucket_count, dim3(grid), dim3(YD_BKT_BS), 0, st, keys, n, sub, shift, nb, ctx->bucketWork.as<unsigned int>());
    hipLaunchKernelGGL(k_bucket_scatter, dim3(grid), dim3(YD_BKT_BS), 0, st, keys, vals, valBase, n, sub, shift, nb, ctx->bucketWork.as<unsigned int>(), outVals, (uint32_t *)nullptr);
    hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ctx->err = std::string("launch of k_bucket_count / k_bucket_scatter failed: ") + hipGetErrorString(e_); return YGPU_ENODEV; }
    return 0;
}
#include <chrono>
static double nowMs() { using namespace std::chrono; return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count(); }
static const bool kTrace = getenv("YGPU_TRACE") != nullptr;
static const bool kStats = getenv("YGPU_STATS") != nullptr;      // one line per ygpu_run: attempts of the align stage, ranges, arena size
#define TRACE(what) do { if (kTrace) { streamSync(ctx); double t_ = nowMs(); fprintf(stderr, "[ygpu] %-28s %9.3f ms\n", what, t_ - ctx->traceT); ctx->traceT = t_; } } while (0)
#define ENSURE(buf, bytes) do { const size_t was_ = (buf).cap; if ((buf).ensure(bytes)) { ctx->err = "hipMalloc failed for " #buf; return YGPU_ENOMEM; } if (kStats && (buf).cap != was_ && (buf).cap >= (1ull << 30)) fprintf(stderr, "[ygpu] ctx %p: " #buf " grows %.2f -> %.2f GB\n", (void *)ctx, was_ / 1e9, (buf).cap / 1e9); } while (0)
#define EV0(t) (ctx->evUsed[t] = true, hipEventRecord(ctx->ev[t][0], ctx->stream))
#define EV1(t) hipEventRecord(ctx->ev[t][1], ctx->stream)

// Waits of the host for its stream: on an event created with hipEventBlockingSync, so that the thread sleeps instead of spinning -- a context has ~15 such
// waits per batch, each tens of milliseconds long, and a node runs (GPUs x contexts) of these threads (YGPU_SPIN_SYNC=1: plain hipStreamSynchronize).
template <class C> static hipError_t streamSync(C *ctx)
{
    static const bool spin = getenv("YGPU_SPIN_SYNC") != nullptr;
    if (spin || !ctx->evSync) return hipStreamSynchronize(ctx->stream);
    hipError_t e = hipEventRecord(ctx->evSync, ctx->stream);
    return e != hipSuccess ? e : hipEventSynchronize(ctx->evSync);
}

// the next stage's sizes cross PCIe as a few words, through a pinned slot (a pageable destination goes through a staging kernel and a second copy)
template <class C> static int fetchU32(C *ctx, const void *dptr, uint32_t *out, size_t n = 1)
{
    if (ctx->pinned && n <= 64) {
        HIPCHK(hipMemcpyAsync(ctx->pinned, dptr, 4 * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        memcpy(out, ctx->pinned, 4 * n); return 0;
    }
    HIPCHK(hipMemcpyAsync(out, dptr, 4 * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx)); return 0;
}

// several small pieces in ONE wait (the copies queue up behind the kernels, one event is waited for): every wait of the host is a gap on the device when one context runs alone
struct FetchPiece { const void *src; uint32_t *dst; uint32_t n; };
template <class C> static int fetchMany(C *ctx, const FetchPiece *pc, int np)
{
    uint32_t tot = 0; for (int k = 0; k < np; k++) tot += pc[k].n;
    if (!ctx->pinned || tot > 64) { for (int k = 0; k < np; k++) { int rc = fetchU32(ctx, pc[k].src, pc[k].dst, pc[k].n); if (rc) return rc; } return 0; }
    uint32_t o = 0; for (int k = 0; k < np; k++) { HIPCHK(hipMemcpyAsync(ctx->pinned + o, pc[k].src, 4ull * pc[k].n, hipMemcpyDeviceToHost, ctx->stream)); o += pc[k].n; }
    HIPCHK(streamSync(ctx));
    o = 0; for (int k = 0; k < np; k++) { memcpy(pc[k].dst, ctx->pinned + o, 4ull * pc[k].n); o += pc[k].n; }
    return 0;
}

// ---- A1 + A2 (+ fragment array) --------------------------------------------------------------------------------
// maxGap for the dead-single test of seed.h, or -1: every fragment is kept (the fragments themselves are asked for, or one word can be a whole match)
static int fragDropGap(const ygpu_ctx *ctx) { return (ctx->keepAllFrags || ctx->P.wordLen >= ctx->P.minMatch) ? -1 : ctx->P.maxGap; }
static int stageSeed(ygpu_ctx *ctx)
{
    const uint32_t n = ctx->nReads, K = ctx->nKmers; DevBatch B = devBatch(ctx);
    HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 4 * CNT_N, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->ctr.p, 0, sizeof(DevCounters), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->errFlag.p, 0, 4, ctx->stream));
    ctx->nHits = ctx->nFrags = ctx->nRegions = ctx->nMulti = ctx->maxN = ctx->nClumps = ctx->nClumpFrags = ctx->nOut = ctx->nOutOps = 0;
    for (int t = 0; t < T_N; t++) { ctx->ms[t] = 0; ctx->evUsed[t] = false; }
    if (K == 0 || n == 0) return 0;
    EV0(T_SEED);
    ENSURE(ctx->posS, 4ull * (K + 1)); ENSURE(ctx->posC, 4ull * (K + 1)); ENSURE(ctx->posRsI, 4ull * (K + 1)); ENSURE(ctx->hitOff, 4ull * (K + 1));
    HIPCHK(hipMemsetAsync(ctx->posC.p, 0, 4ull * (K + 1), ctx->stream));
    ENSURE(ctx->kmerParts, 4096); HIPCHK(hipMemsetAsync(ctx->kmerParts.p, 0, 4096, ctx->stream));
    KL(k_kmer_lookup, dim3(2 * n), dim3(128), 0, ctx->stream, ctx->P, B, ctx->dSO.as<uint32_t>(), ctx->dROA.as<uint32_t>(), ctx->dLow.as<uint32_t>(), ctx->dKmerOff.as<uint32_t>(),
                       ctx->posS.as<uint32_t>(), ctx->posC.as<uint32_t>(), ctx->posRsI.as<uint32_t>(), ctx->kmerParts.as<unsigned int>());
    KL(k_sum_parts, dim3(1), dim3(1024), 0, ctx->stream, ctx->kmerParts.as<unsigned int>(), ctx->ctr.as<DevCounters>()->v + C_KMER);
    int rc = cubScan(ctx, ctx->posC.as<uint32_t>(), ctx->hitOff.as<uint32_t>(), K + 1); if (rc) return rc;
    EV1(T_SEED);
    uint32_t H = 0; rc = fetchU32(ctx, ctx->hitOff.as<uint32_t>() + K, &H); if (rc) return rc;
    ctx->nHits = H;
    if (H == 0) return 0;
    if (H > 0x7FFFFFF0u) { ctx->err = "too many seed hits in one batch; use a smaller batch"; return YGPU_EOVERFLOW; }
    EV0(T_SORT);
    ENSURE(ctx->keysA, 8ull * H); ENSURE(ctx->keysB, 8ull * H);
    ENSURE(ctx->expandStart, 4ull * (gridFor(H, YD_EXPAND_HITS) + 1));
    KL(k_expand_starts, dim3(gridFor(K, 256)), dim3(256), 0, ctx->stream, ctx->hitOff.as<uint32_t>(), K, ctx->expandStart.as<uint32_t>());
    KL(k_expand_hits, dim3(gridFor(H, YD_EXPAND_HITS)), dim3(256), 0, ctx->stream, ctx->dROA.as<uint32_t>(), ctx->posS.as<uint32_t>(), ctx->hitOff.as<uint32_t>(), ctx->posRsI.as<uint32_t>(), ctx->expandStart.as<uint32_t>(), K, H, ctx->keysA.as<unsigned long long>());
    {
        // The sort is stable and k_expand_hits writes the hits of one (read, strand) in ascending query offset (k-mers in order, each
        // k-mer's reference offsets ascending), so two hits of one diagonal are already in qo order: the low 15 key bits need no pass.
        // The hits of one (read, strand) are one segment: sorted on the 32 diagonal bits only, inside one workgroup (one pass over HBM), instead of a
        // batch-wide sort that also has to order the (read, strand) bits.
        {
            ENSURE(ctx->segOff, 4ull * (2 * n + 2));
            KL(k_seg_offsets, dim3(gridFor(2 * n + 1, 256)), dim3(256), 0, ctx->stream, ctx->dKmerOff.as<uint32_t>(), ctx->hitOff.as<uint32_t>(), 2 * n, ctx->segOff.as<uint32_t>());
            {
                // segments of up to 15 872 hits: one workgroup each (segsort.h), in twelve size classes, one launch per class over exactly its segments
                const unsigned long long *in = ctx->keysA.as<unsigned long long>(); unsigned long long *out = ctx->keysB.as<unsigned long long>(); const uint32_t *so = ctx->segOff.as<uint32_t>();
                ENSURE(ctx->segLists, 4ull * (YD_SEG_NCLASS + 1) * (2 * n + 1));
                uint32_t *segCnt = ctx->counters.as<uint32_t>() + CNT_SEGC;
                const uint32_t mx = ctx->segSortMax;                                  // YD_SEGSORT_MAX; lower only to drive the long-segment path in tests
                // threads x hits a thread: 128 x 8, 128 x 16, 256 x 12 / 16, 512 x 10 / 12 / 14 / 16 and, for the four largest classes, 512 x 20 / 24 / 28 / 31 (384 and 768
                // threads x 16 sorted slower than the next shape up).  The largest classes had 1 024-thread workgroups (x 10 / 12 / 14 / 16): four waves a SIMD with 72-112
                // registers each, which beside a rows launch -- one or two waves of 152 registers on every SIMD of the device while it runs -- found room on the CUs with one
                // rows workgroup (x 10, x 12) or NOWHERE (x 14, x 16: their launch, first in the stream, then waited for the rows launch to end -- 0.36 ms alone, 4.7 ms in
                // the four-context run, and every smaller class behind it).  512 threads x twice the hits: two waves a SIMD of 136-176 registers, room beside one rows
                // workgroup for all four (x 32 would need 177 registers, eight a SIMD too many: hence 15 872 hits as the limit of a single workgroup's sort); the same speed alone, 0.3-0.5 ms a step with four contexts (profiles/r05_sort_shapes.txt).  YGPU_SORT_WIDE=0: the old shapes.
                static const uint32_t kShape[YD_SEG_NCLASS] = {1024u, 2048u, 3072u, 4096u, 5120u, 6144u, 7168u, 8192u, 10240u, 12288u, 14336u, YD_SEGSORT_MAX};
                SegClassHi HI; for (int c = 0; c < YD_SEG_NCLASS; c++) HI.hi[c] = std::min(mx, kShape[c]);
                // one launch per class over exactly its segments: in[inB..inE) sorted into out[inB..)
                auto sortClasses = [&](const unsigned long long *src, unsigned long long *dst, const uint32_t *sB, const uint32_t *sE, uint32_t nSeg, const uint32_t *lists, const uint32_t *nc) -> int {
#define YD_SORT_CLASS(c, BS, IPT) if (nc[c]) KL((k_seg_sort<BS, IPT>), dim3(nc[c]), dim3(BS), 0, ctx->stream, src, dst, sB, sE, lists + (size_t)(c) * nSeg)
                    const char *ws = getenv("YGPU_SORT_WIDE"); const int wideShapes = ws ? atoi(ws) : 1;      // (read at every call: the tests run both)
                    if (wideShapes) { YD_SORT_CLASS(11, 512, YD_SORT_TOP); YD_SORT_CLASS(10, 512, 28); YD_SORT_CLASS(9, 512, 24); YD_SORT_CLASS(8, 512, 20); }
                    else { YD_SORT_CLASS(11, 1024, 16); YD_SORT_CLASS(10, 1024, 14); YD_SORT_CLASS(9, 1024, 12); YD_SORT_CLASS(8, 1024, 10); }
                    YD_SORT_CLASS(7, 512, 16);
                    YD_SORT_CLASS(6, 512, 14); YD_SORT_CLASS(5, 512, 12); YD_SORT_CLASS(4, 512, 10);
                    YD_SORT_CLASS(3, 256, 16); YD_SORT_CLASS(2, 256, 12); YD_SORT_CLASS(1, 128, 16); YD_SORT_CLASS(0, 128, 8);
#undef YD_SORT_CLASS
                    return 0;
                };
                HIPCHK(hipMemsetAsync(segCnt, 0, 4 * (YD_SEG_NCLASS + 1), ctx->stream));
                uint32_t *lists = ctx->segLists.as<uint32_t>();
                KL(k_seg_classify, dim3(gridFor(2 * n, 256)), dim3(256), 0, ctx->stream, so, so + 1, 2 * n, HI, lists, (uint32_t *)nullptr, (uint32_t *)nullptr, segCnt);
                uint32_t nc[YD_SEG_NCLASS + 1] = {0}; rc = fetchU32(ctx, segCnt, nc, YD_SEG_NCLASS + 1); if (rc) return rc;
                rc = sortClasses(in, out, so, so + 1, 2 * n, lists, nc); if (rc) return rc;
                const uint32_t nBig = nc[YD_SEG_NCLASS];
                if (kTrace) { std::vector<uint32_t> so2(2 * (size_t)n + 1); hipMemcpy(so2.data(), so, 4ull * (2 * n + 1), hipMemcpyDeviceToHost); unsigned long long hb = 0, mxl = 0, cl[YD_SEG_NCLASS] = {0};
                    for (uint32_t k = 0; k < 2 * n; k++) { const unsigned long long l = so2[k + 1] - so2[k]; if (l > mx) { hb += l; mxl = std::max(mxl, l); } else for (int c = 0; c < YD_SEG_NCLASS; c++) if (l <= HI.hi[c]) { cl[c] += l; break; } }
                    fprintf(stderr, "[ygpu] hit sort: %u hits; segments above %u hits: %u holding %llu hits (%.1f%%, longest %llu); %% of the hits by class:", H, mx, nBig, hb, 100.0 * hb / H, mxl);
                    for (int c = 0; c < YD_SEG_NCLASS; c++) fprintf(stderr, " <=%u: %.1f", HI.hi[c], 100.0 * cl[c] / H);
                    fprintf(stderr, "\n"); }
                if (nBig) {
                    // long segments: cut by diagonal into buckets that fit the workgroup sort (k_seg_split: keysA -> keysB), the buckets sorted in place
                    const uint32_t nSub = nBig * YD_SPLIT_NB;
                    ENSURE(ctx->subB, 4ull * nSub + 64); ENSURE(ctx->subE, 4ull * nSub + 64); ENSURE(ctx->subLists, 4ull * (YD_SEG_NCLASS + 1) * nSub + 64);
                    int diagBits = 1; while (diagBits < 32 && (ctx->P.maxROff >> diagBits)) diagBits++;
                    uint32_t *sB = ctx->subB.as<uint32_t>(), *sE = ctx->subE.as<uint32_t>(), *l2 = ctx->subLists.as<uint32_t>();
                    KL(k_seg_split, dim3(nBig), dim3(1024), 0, ctx->stream, in, out, so, so + 1, lists + (size_t)YD_SEG_NCLASS * (2 * n), diagBits, sB, sE);
                    HIPCHK(hipMemsetAsync(segCnt, 0, 4 * (YD_SEG_NCLASS + 1), ctx->stream));
                    KL(k_seg_classify, dim3(gridFor(nSub, 256)), dim3(256), 0, ctx->stream, sB, sE, nSub, HI, l2, (uint32_t *)nullptr, (uint32_t *)nullptr, segCnt);
                    uint32_t ns[YD_SEG_NCLASS + 1] = {0}; rc = fetchU32(ctx, segCnt, ns, YD_SEG_NCLASS + 1); if (rc) return rc;
                    rc = sortClasses(out, out, sB, sE, nSub, l2, ns); if (rc) return rc;
                    // pieces that still do not fit: cut again over their own range of diagonals, level by level (segsort.h: k_seg_split_range), until every piece
                    // fits the workgroup sort or holds one diagonal only.  Level L reads the pieces where level L-1 left them (keysB after the first cut, then
                    // keysA / keysB in turn) and sorts what fits into keysB.
                    uint32_t nOver = ns[YD_SEG_NCLASS]; const uint32_t *oB = sB, *oE = sE, *oList = l2 + (size_t)YD_SEG_NCLASS * nSub;
                    const unsigned long long *cur = out; unsigned long long *other = ctx->keysA.as<unsigned long long>();
                    if (kTrace) { fprintf(stderr, "[ygpu] hit sort: %u long segments cut into buckets, by class:", nBig); for (int c = 0; c < YD_SEG_NCLASS; c++) fprintf(stderr, " %u", ns[c]); fprintf(stderr, "; %u to be cut again\n", nOver); }
                    for (int level = 0; nOver; level++) {
                        if (level >= 12) { ctx->err = "hit sort: a segment does not fit the workgroup sort after twelve cuts"; return YGPU_EINTERNAL; }
                        DevBuf &xB = level & 1 ? ctx->sub3B : ctx->sub2B, &xE = level & 1 ? ctx->sub3E : ctx->sub2E, &xL = level & 1 ? ctx->sub3Lists : ctx->sub2Lists;
                        const uint32_t nSub2 = nOver * YD_SPLIT_NB;
                        ENSURE(xB, 4ull * nSub2 + 64); ENSURE(xE, 4ull * nSub2 + 64); ENSURE(xL, 4ull * (YD_SEG_NCLASS + 1) * nSub2 + 64);
                        KL(k_seg_split_range, dim3(nOver), dim3(1024), 0, ctx->stream, cur, other, out, oB, oE, oList, xB.as<uint32_t>(), xE.as<uint32_t>());
                        HIPCHK(hipMemsetAsync(segCnt, 0, 4 * (YD_SEG_NCLASS + 1), ctx->stream));
                        KL(k_seg_classify, dim3(gridFor(nSub2, 256)), dim3(256), 0, ctx->stream, xB.as<uint32_t>(), xE.as<uint32_t>(), nSub2, HI, xL.as<uint32_t>(), (uint32_t *)nullptr, (uint32_t *)nullptr, segCnt);
                        uint32_t n3[YD_SEG_NCLASS + 1] = {0}; rc = fetchU32(ctx, segCnt, n3, YD_SEG_NCLASS + 1); if (rc) return rc;
                        rc = sortClasses(other, out, xB.as<uint32_t>(), xE.as<uint32_t>(), nSub2, xL.as<uint32_t>(), n3); if (rc) return rc;
                        if (kTrace) fprintf(stderr, "[ygpu] hit sort: cut %d: %u pieces cut again, %u of their buckets still too long\n", level + 2, nOver, n3[YD_SEG_NCLASS]);
                        nOver = n3[YD_SEG_NCLASS]; oB = xB.as<uint32_t>(); oE = xE.as<uint32_t>(); oList = xL.as<uint32_t>() + (size_t)YD_SEG_NCLASS * nSub2;
                        const unsigned long long *was = cur; cur = other; other = (unsigned long long *)was;
                    }
                }
            }
        }
    }
    EV1(T_SORT);
    return 0;
}
// (Re)creates the fragment array from the sorted keys -- the chain stage trims it in place, so a redo of that stage comes back here.  One kernel (seed.h:
// k_frag_scan_build) counts and writes; the array is sized from the last batch's count, and a batch that needs more is run again with room (the first batch of
// a context always is: its first pass only counts).
static int buildFrags(ygpu_ctx *ctx, bool redo = false)      // redo: the regions stand, only the records are rebuilt (refLen is otherwise set by k_region_scan)
{
    const uint32_t H = ctx->nHits;
    ctx->nFrags = 0;
    if (!H) return 0;
    const uint32_t nTiles = (uint32_t)gridFor(H, YD_FRAG_TILE);
    ENSURE(ctx->tileState, 8ull * (nTiles + 1));                             // + the ticket word
    unsigned int *total = ctx->counters.as<unsigned int>() + CNT_NFRAGS;
    for (int pass = 0;; pass++) {
        const uint32_t cap = ctx->frags.cap >= 32 ? (uint32_t)std::min<uint64_t>(ctx->frags.cap / 16 - 1, 0xFFFFFFF0u) : 0u;
        // the fragments that are dropped (seed.h: hitClass) are counted: the counters report every fragment and region of the reference
        ENSURE(ctx->kmerParts, 4096); HIPCHK(hipMemsetAsync(ctx->kmerParts.p, 0, 4096, ctx->stream)); HIPCHK(hipMemsetAsync(ctx->ctr.as<DevCounters>()->v + C_FRAGS, 0, 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->tileState.p, 0, 8ull * (nTiles + 1), ctx->stream));
        KL(k_frag_scan_build, dim3(nTiles), dim3(YD_FRAG_BS), 0, ctx->stream, ctx->keysB.as<unsigned long long>(), H, ctx->P.wordLen, fragDropGap(ctx), ctx->frags.as<DevFrag>(), cap,
           ctx->tileState.as<unsigned long long>(), total, ctx->kmerParts.as<unsigned int>());
        uint32_t two[2] = {0, 0}; int rc = fetchU32(ctx, total, two, 2); if (rc) return rc;
        if (two[1]) { ctx->err = "fragment scan: a tile was not published within 30 s (look-back gave up)"; return YGPU_EINTERNAL; }
        const uint32_t F = two[0];
        if (F <= cap) { ctx->nFrags = F; break; }
        if (pass >= 2) { ctx->err = "fragment build: the count changed between passes"; return YGPU_EINTERNAL; }
        ENSURE(ctx->frags, 16ull * ((uint64_t)F + F / 8 + 4096));
    }
    KL(k_sum_parts, dim3(1), dim3(1024), 0, ctx->stream, ctx->kmerParts.as<unsigned int>(), ctx->ctr.as<DevCounters>()->v + C_FRAGS);
    if (ctx->nFrags && (redo || ctx->keepAllFrags)) KL(k_frag_finish, dim3(gridFor(ctx->nFrags, 256)), dim3(256), 0, ctx->stream, ctx->frags.as<DevFrag>(), ctx->nFrags);
    return 0;
}

// ---- A3 + A4 ------------------------------------------------------------------------------------------------------
static int stageChain(ygpu_ctx *ctx)
{
    const uint32_t F = ctx->nFrags; DevBatch B = devBatch(ctx);
    if (!F) return 0;
    int rc;
    // region boundaries (uses a second head/scan pair sized by F; the hit-level pair is still needed by buildFrags on a retry)
    ENSURE(ctx->regStart, 4ull * (F + 2)); ENSURE(ctx->multiList, 4ull * (F + 1)); ENSURE(ctx->smallList, 4ull * (F + 1)); ENSURE(ctx->bigList, 4ull * (F / 64 + 2)); ENSURE(ctx->regionCount, 4ull * (F + 2)); ENSURE(ctx->regionBase, 4ull * (F + 2));
    // (the fragment scan's tile states are free again: reused for the region scan)
    const uint32_t nRegTiles = (uint32_t)gridFor(F, YD_REG_TILE);
    ENSURE(ctx->tileState, 8ull * (nRegTiles + 1));
    uint32_t *cnt = ctx->counters.as<uint32_t>();
    HIPCHK(hipMemsetAsync(ctx->tileState.p, 0, 8ull * (nRegTiles + 1), ctx->stream));
    KL(k_region_scan, dim3(nRegTiles), dim3(256), 0, ctx->stream, ctx->frags.as<DevFrag>(), F, ctx->P.maxGap, ctx->regStart.as<uint32_t>(), ctx->tileState.as<unsigned long long>(), cnt + CNT_NREG);
    uint32_t R = 0; { uint32_t two[2] = {0, 0}; rc = fetchU32(ctx, cnt + CNT_NREG, two, 2); if (rc) return rc; if (two[1]) { ctx->err = "region scan: a tile was not published within 30 s (look-back gave up)"; return YGPU_EINTERNAL; } R = two[0]; }
    ctx->nRegions = R;
    HIPCHK(hipMemcpyAsync((uint32_t *)ctx->regStart.p + R, &ctx->nFrags, 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(cnt + CNT_NMULTI, 0, 8, ctx->stream)); HIPCHK(hipMemsetAsync(cnt + CNT_NBIG, 0, 8, ctx->stream)); HIPCHK(hipMemsetAsync(cnt + CNT_NSMALL, 0, 4, ctx->stream));
    KL(k_region_classify, dim3(gridFor(R, 1024)), dim3(1024), 0, ctx->stream, ctx->regStart.as<uint32_t>(), R, ctx->multiList.as<uint32_t>(), cnt + CNT_NMULTI, cnt + CNT_MAXN, ctx->bigList.as<uint32_t>(), cnt + CNT_NBIG, ctx->smallList.as<uint32_t>(), cnt + CNT_NSMALL);
    uint32_t two[2] = {0, 0};
    { const FetchPiece pc[3] = {{cnt + CNT_NMULTI, two, 2}, {cnt + CNT_NBIG, &ctx->nBig, 1}, {cnt + CNT_NSMALL, &ctx->nSmall, 1}}; rc = fetchMany(ctx, pc, 3); if (rc) return rc; }
    ctx->nMulti = two[0]; ctx->maxN = two[1];
    EV1(T_FRAGS);

    EV0(T_CHAIN);
    // (clump slots are pre-set to "invalid", so their number is a 16-byte store each: after the first batch the bound is twice what the last batch used -- most
    // fragments of a large genome are single hits that form no clump: 128 M fragments, 5.4 M clumps a batch at 3.1 Gbp -- and a batch that overflows it is redone at the full bound)
    const uint32_t clumpSlack = 1024 + 32 * (ctx->nCU * 27 + 64) + 512 * (ctx->nCU * 8), clumpCapFull = F + R / 2 + clumpSlack;
    uint32_t clumpCap = ctx->lastClumpSlots ? (uint32_t)std::min<uint64_t>(clumpCapFull, 2ull * ctx->lastClumpSlots + clumpSlack) : clumpCapFull, fragCap = 2 * F + 1024 + 256 * (ctx->nCU * 27 + 64) + 2048 * (ctx->nCU * 8);   // + one open chunk per wave (k_chain: 32/256, k_chain_lanes: 512/2048)
    if (const char *e = getenv("YGPU_CLUMP_BOUND")) { const long v = atol(e); if (v > 0 && (uint64_t)v < clumpCapFull) clumpCap = (uint32_t)v; }      // tests: a first bound that overflows
    const unsigned waves = (unsigned)std::min<uint64_t>(std::max<uint32_t>(ctx->nMulti, 1u), (uint64_t)ctx->nCU * 24);     // latency-bound serial work: 6 waves per SIMD
    const unsigned wavesBig = (unsigned)std::min<uint64_t>(std::max<uint32_t>(ctx->nBig, 1u), (uint64_t)ctx->nCU * 3);        // 40 KB of LDS each
    for (int attempt = 0;; attempt++) {
        ENSURE(ctx->clumps, sizeof(ChainClumpRec) * (uint64_t)clumpCap); ENSURE(ctx->clumpFrags, 16ull * fragCap);
        const int maxN = (int)std::max<uint32_t>(ctx->maxN, 2u);
        const size_t per = chainScratchBytes(maxN, ctx->maxQ);
        ENSURE(ctx->scratchChain, per * wavesBig);
        HIPCHK(hipMemsetAsync(cnt + CNT_CLUMPS, 0, 12, ctx->stream));          // clumps, cfrags, qchain
        HIPCHK(hipMemsetAsync(cnt + CNT_QBIG, 0, 4, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->clumps.p, 0xFF, sizeof(ChainClumpRec) * (uint64_t)clumpCap, ctx->stream));        // invalid until written
        HIPCHK(hipMemsetAsync(ctx->regionCount.p, 0, 4ull * (R + 1), ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->errFlag.p, 0, 4, ctx->stream));
        ChainArgs A; A.P = ctx->P; A.B = B; A.frags = ctx->frags.as<DevFrag>(); A.regStart = ctx->regStart.as<uint32_t>(); A.nRegions = R;
        A.multiList = ctx->multiList.as<uint32_t>(); A.nMulti = ctx->nMulti; A.queueHead = cnt + CNT_QCHAIN;
        A.scratch = ctx->scratchChain.as<uint8_t>(); A.scratchPerWave = per; A.maxN = maxN; A.maxQ = ctx->maxQ;
        A.clumps = ctx->clumps.as<ChainClumpRec>(); A.clumpFrags = ctx->clumpFrags.as<DevFrag>(); A.counts = cnt + CNT_CLUMPS; A.clumpCap = clumpCap; A.fragCap = fragCap;
        A.regionClumpCount = ctx->regionCount.as<uint32_t>(); A.errFlag = ctx->errFlag.as<int>(); A.ctr = ctx->ctr.as<DevCounters>();
        KL(k_regions_single, dim3(gridFor(R, 1024)), dim3(1024), 0, ctx->stream, A);
        if (ctx->nSmall) KL(k_chain_lanes, dim3((unsigned)std::min<uint64_t>(gridFor(ctx->nSmall, 64), (uint64_t)ctx->nCU * 8)), dim3(64), 0, ctx->stream, A, ctx->smallList.as<uint32_t>(), ctx->nSmall);
        if (ctx->nMulti) KL(k_chain, dim3(waves), dim3(64), 0, ctx->stream, A);
        if (ctx->nBig) KL(k_chain_big, dim3(wavesBig), dim3(64), 0, ctx->stream, A, ctx->bigList.as<uint32_t>(), ctx->nBig, cnt + CNT_QBIG);
        // creation-order rank of every root clump: the sum over the regions' counts is launched before anybody knows whether the attempt fitted -- an attempt that did not
        // is redone, sum included -- so that the counts, the error flag and the number of clumps cross in ONE wait
        rc = cubScan(ctx, ctx->regionCount.as<uint32_t>(), ctx->regionBase.as<uint32_t>(), R + 1); if (rc) return rc;
        uint32_t got[2] = {0, 0}, ef = 0;
        { const FetchPiece pc[3] = {{cnt + CNT_CLUMPS, got, 2}, {ctx->errFlag.p, &ef, 1}, {ctx->regionBase.as<uint32_t>() + R, &ctx->nClumps, 1}}; rc = fetchMany(ctx, pc, 3); if (rc) return rc; }
        if (ef == 0 && got[0] <= clumpCap && got[1] <= fragCap) { ctx->nClumpSlots = got[0]; ctx->nClumpFrags = got[1]; ctx->lastClumpSlots = got[0]; break; }
        if (attempt >= 6) { ctx->err = "chain stage: arena overflow persists"; return YGPU_EOVERFLOW; }
        if (clumpCap < clumpCapFull) clumpCap = clumpCapFull; else { clumpCap *= 2; fragCap *= 2; }      // grow and redo: the fragment array was modified in place
        rc = buildFrags(ctx, true); if (rc) return rc;
    }
    // (ctx->nClumps: clumps actually formed -- slots minus chunk slack)
    ENSURE(ctx->order, 4ull * (ctx->nClumps + 1));
    if (ctx->nClumpSlots) KL(k_clump_order, dim3(gridFor(ctx->nClumpSlots, 256)), dim3(256), 0, ctx->stream, ctx->clumps.as<ChainClumpRec>(), ctx->nClumpSlots, ctx->regionBase.as<uint32_t>(), ctx->order.as<uint32_t>());
    EV1(T_CHAIN);
    return 0;
}

// Per-wave scratch of the wave kernels, sized from the parameters.  A gap fill between two chained fragments has min(qGap, rGap) <= maxDesert
// and |qGap - rGap| <= maxGap (GraphPath.cpp:211-230), so its strip is at most MD + G + 2*BW + 3 columns wide (banded: 2*BW + 1 + |qGap - rGap|,
// full: rGap + 1) and rows x width <= (MD + 2) * (MD + G + 2*BW + 3) cells; the extensions need (maxQ + 2) rows of 64 cells.
// X-drop extensions in packed 16-bit arithmetic (ext_lanes_pk.h) when every score fits with room for the sentinel
static bool extRowsPacked(const ygpu_ctx *ctx, bool caps)
{
    const bool force32 = getenv("YGPU_EXT32") != nullptr;            // (read at every call: the tests run both kernel families in one process)
    const DevParams &P = ctx->P;
    return !force32 && !caps && P.MS >= 0 && (long long)P.MS * std::max(1, ctx->maxQ) <= 15000 && P.RC >= 0 && P.GO >= 0 && P.GE >= 0 && P.X >= 0 && (long long)P.RC + P.X + P.GO + 21ll * P.GE <= 4000;
}

static void alignDims(ygpu_ctx *ctx, int &listCap, int &front, int &genCap, int &traceRows)
{
    front = 2 * ctx->maxQ + 8 * ctx->P.bandWidth + 64; listCap = 2 * front + 3 * ctx->maxQ + 1024;
    const long long md = std::min<long long>(ctx->P.maxDesert, 32000), g = std::min<long long>(ctx->P.maxGap, 32000), wMax = md + g + 2 * ctx->P.bandWidth + 3;
    genCap = (int)std::max<long long>(1024, wMax + 1);
    // rows of 64 cells: an X-drop extension of a whole read, 4 * BW + 1 columns wide (more than one 64-cell row per DP row when BW > 15), or the widest gap fill
    const long long wExt = 4ll * ctx->P.bandWidth + 1;
    traceRows = (int)std::max<long long>((ctx->maxQ + 2) * ((wExt + 63) / 64), ((md + 2) * wMax + 63) / 64 + 1);
}


// alignClump with the two X-drop extensions of every root done one problem per lane (ext_lanes.h):
//   k_align_p1 (wave/root: gap fills, exact-match extensions) -> scan of the strip sizes -> k_ext_rows + k_ext_trace
//   (lane/problem, in chunks that fit the trace memory) -> k_align_p3 (wave/root: merge, scoreClump/splitClump).
// returns -2 when an arena was too small (the caller grows and redoes the stage)
static int alignWithLaneExtensions(ygpu_ctx *ctx, AlignArgs &A, unsigned waves, uint32_t stateOpsCap, uint32_t gapOpsPerJoint)
{
    const uint32_t NC = ctx->nClumps; const uint32_t nProb = 2 * NC; int rc;
    uint32_t *cnt = ctx->counters.as<uint32_t>();
    ENSURE(ctx->rootState, sizeof(RootState) * (uint64_t)NC); ENSURE(ctx->stateOps, 4ull * stateOpsCap); ENSURE(ctx->extProbs, sizeof(ExtProb) * (uint64_t)nProb);
    ENSURE(ctx->rowsBound, 8ull * (nProb + 1)); ENSURE(ctx->stripOff, 8ull * (nProb + 1)); ENSURE(ctx->extRes, sizeof(ExtRes) * (uint64_t)nProb);
    ENSURE(ctx->slowList, 4ull * (NC + 1));
    HIPCHK(hipMemsetAsync(cnt + CNT_STATEOPS, 0, 16, ctx->stream));            // stateops, extops, qext, slow
    HIPCHK(hipMemsetAsync(cnt + CNT_QALIGN, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync((unsigned long long *)ctx->rowsBound.p + nProb, 0, 8, ctx->stream));
    PhaseArgs X; X.state = ctx->rootState.as<RootState>(); X.stateOps = ctx->stateOps.as<uint32_t>(); X.stateOpsCount = cnt + CNT_STATEOPS; X.stateOpsCap = stateOpsCap;
    X.probs = ctx->extProbs.as<ExtProb>(); X.rowsBound = ctx->rowsBound.as<unsigned long long>(); X.res = ctx->extRes.as<ExtRes>(); X.extOps = nullptr; X.rootBegin = 0;
    TRACE("lanes: ensure");
    X.slowList = ctx->slowList.as<uint32_t>(); X.slowCount = cnt + CNT_SLOW; X.useList = 1;
    EV0(T_P1);
    // joints of all roots
    ENSURE(ctx->jointCount, 4ull * (NC + 2)); ENSURE(ctx->jointBase, 4ull * (NC + 2));
    X.jointCount = ctx->jointCount.as<uint32_t>(); X.jointBase = ctx->jointBase.as<uint32_t>();
    KL(k_joint_counts, dim3(gridFor(NC + 1, 256)), dim3(256), 0, ctx->stream, A, X);
    rc = cubScan(ctx, ctx->jointCount.as<uint32_t>(), ctx->jointBase.as<uint32_t>(), NC + 1); if (rc) return rc;
    uint32_t J = 0; rc = fetchU32(ctx, ctx->jointBase.as<uint32_t>() + NC, &J); if (rc) return rc;
    const uint32_t gapOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, (uint64_t)gapOpsPerJoint * J + (1u << 20));
    ENSURE(ctx->joints, sizeof(JointRec) * (uint64_t)(J + 1)); ENSURE(ctx->sortKeys, 4ull * (J + 1)); ENSURE(ctx->sortVals, 4ull * (J + 1)); ENSURE(ctx->sortKeys2, 4ull * (J + 1)); ENSURE(ctx->sortVals2, 4ull * (J + 1));
    ENSURE(ctx->gapOps, 4ull * gapOpsCap); ENSURE(ctx->slowList, 4ull * (std::max(NC, J) + 1));
    X.slowList = ctx->slowList.as<uint32_t>();
    X.joints = ctx->joints.as<JointRec>(); X.nJoints = J; X.sortKeys = ctx->sortKeys.as<uint32_t>(); X.sortVals = ctx->sortVals.as<uint32_t>(); X.sortedVals = ctx->sortVals2.as<uint32_t>();
    X.nDP = cnt + CNT_NDP; X.nDPb = cnt + CNT_NB12; X.gapOps = ctx->gapOps.as<uint32_t>(); X.gapOpsCount = cnt + CNT_GAPOPS; X.gapOpsCap = gapOpsCap;
    HIPCHK(hipMemsetAsync(cnt + CNT_NDP, 0, 12, ctx->stream)); HIPCHK(hipMemsetAsync(cnt + CNT_NB12, 0, 8, ctx->stream));      // ndp, ndp16, gapops; nb12, nb16
    if (J) {
        KL(k_p1_joints, dim3(gridFor(NC, 256)), dim3(256), 0, ctx->stream, A, X);
        rc = bucketOrder(ctx, X.sortKeys, X.sortVals, 0, J, 0, 0, 1u << YD_JKEY_BITS, ctx->sortVals2.as<uint32_t>(), ctx->stream); if (rc) return rc;      // joints of one (class, width, rows / 2) together
        const unsigned gBlocks16 = (unsigned)std::min<uint64_t>(gridFor(J, 64), (uint64_t)ctx->nCU * 9), gBlocks32 = (unsigned)std::min<uint64_t>(gridFor(J, 64), (uint64_t)ctx->nCU * 6);   // 17 / 26 KB of LDS per 64-thread block
        ENSURE(ctx->gapScratch, (size_t)YD_GAP_SCRATCH * 64 * std::max(gBlocks16, gBlocks32)); X.gapScratch = ctx->gapScratch.as<uint8_t>();
        KL(k_gap_band<12>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_band<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_lanes<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_lanes<32>, dim3(gBlocks32), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_wave, dim3(std::min(waves, 512u)), dim3(64), 0, ctx->stream, A, X);
    }
    ENSURE(ctx->extKeys, 4ull * (nProb + 1)); ENSURE(ctx->extVals, 4ull * (nProb + 1)); ENSURE(ctx->extKeys2, 4ull * (nProb + 1)); ENSURE(ctx->extOrder, 4ull * (nProb + 1));
    X.extKeys = ctx->extKeys.as<uint32_t>(); X.extVals = ctx->extVals.as<uint32_t>();
    KL(k_p1_assemble, dim3(gridFor(NC, 256)), dim3(256), 0, ctx->stream, A, X);
    rc = cubScan64(ctx, ctx->rowsBound.as<unsigned long long>(), ctx->stripOff.as<unsigned long long>(), nProb + 1); if (rc) return rc;
    EV1(T_P1);
    TRACE("lanes: p1+scan");
    if (kTrace) { uint32_t v[3] = {0, 0, 0}; fetchU32(ctx, cnt + CNT_SLOW, &v[0]); uint32_t w[3] = {0, 0, 0}; fetchU32(ctx, cnt + CNT_NDP, w, 3); v[1] = w[0]; v[2] = w[2]; fprintf(stderr, "[ygpu] roots %u, joints %u, DP joints %u (W<=16: %u, wave fallback %u), gap ops %u\n", NC, J, v[1], w[1], v[0], v[2]); }
    unsigned long long boundBlocks = 0;                                      // sum over the problems of the 10-row blocks each may reach (its row BOUND)
    HIPCHK(hipMemcpyAsync(&boundBlocks, ctx->stripOff.as<unsigned long long>() + nProb, 8, hipMemcpyDeviceToHost, ctx->stream));
    uint32_t ef = 0; rc = fetchU32(ctx, ctx->errFlag.p, &ef); if (rc) return rc;
    if (ef == YERR_OUT) return -2;
    if (ef) return 0;                                                         // reported by the caller
    // (the kernels of the X-drop extensions: packed 16-bit rows when the scores fit, see ext_lanes_pk.h; YGPU_EXT32=1 forces the 32-bit kernels)
    // ---- trace memory (ext_lanes.h): an arena of 128 KB chunks that the waves of k_ext_rows take as their rows are computed -------------------------
    // What a launch will need is not known before it ran (an X-drop run stops where it stops); the arena is sized from the bound scaled by the ratio
    // the last batches showed (ctx->traceRatio; the first batch guesses from the mean bound) and the stage is redone with a larger one when it overflows.
    // When even the budget (this context's share of the free memory) is not enough, the roots are cut into ranges that use the arena one after the other.
    ExtArgs E; E.P = ctx->P; E.bases = ctx->dBases.as<uint8_t>(); E.fwd = ctx->dFwd.as<uint8_t>(); E.rev = ctx->dRev.as<uint8_t>(); E.fwd4 = ctx->dFwd4.as<uint8_t>(); E.rev4 = ctx->dRev4.as<uint8_t>();
    const bool caps = ctx->P.maxGap < YD_LW || ctx->P.maxIntron < YD_LW;
    const bool pk = extRowsPacked(ctx, caps); ctx->rowsPacked = pk;
    // (YGPU_ROWS_BS=512: the main rows launch in workgroups of eight waves -- two per SIMD of one CU -- when it shares the device)
    static const int rowsBSenv = getenv("YGPU_ROWS_BS") ? atoi(getenv("YGPU_ROWS_BS")) : 256;
    const bool rowsShare = gActiveRuns[ctx->device & 63].load() >= 2;      // (decided when the launch is sized: a batch that is alone in flight takes the whole device)
    const unsigned rowsBS = (pk && rowsBSenv == 512 && rowsShare) ? 512u : 256u;
    auto rowsKernel = pk ? (rowsBS == 512u ? k_ext_rows_pk<false, 512> : k_ext_rows_pk<false, 256>) : (caps ? k_ext_rows<true, false> : k_ext_rows<false, false>);
    // (the careful-extension round is a small launch of a wave per SIMD; YGPU_ROWS2_BS=512 pairs its waves: workgroups of eight waves on half as many CUs)
    static const int rows2BSenv = getenv("YGPU_ROWS2_BS") ? atoi(getenv("YGPU_ROWS2_BS")) : 256;
    const unsigned rows2BS = (pk && rows2BSenv == 512) ? 512u : 256u;
    auto rowsKernel2 = pk ? (rows2BS == 512u ? k_ext_rows_pk<true, 512> : k_ext_rows_pk<true, 256>) : (caps ? k_ext_rows<true, true> : k_ext_rows<false, true>);
    auto traceKernel = pk ? k_ext_trace_pk : k_ext_trace; const unsigned traceBS = pk ? (unsigned)YD_TRACE_BS : 256u;
    // the traceback's order: 0 = k_ext_rows' order; n > 0: by arena region of 2^n chunks, then by walk length (YGPU_TRACE_LENBITS bits).  With the wave-wide block
    // fetch of k_ext_trace_pk a wave walks in lock step, so what counts is that its lanes' walks are equally long: the default is the length alone (n = 20: one region),
    // in 128 classes -- one radix pass (3.1 Gbp, three contexts: 52.0 ms a step with regions of 128 chunks and 32 classes, 57.3 in the rows kernel's order, 51.2 so)
    static const int traceSort = getenv("YGPU_TRACE_SORT") ? atoi(getenv("YGPU_TRACE_SORT")) : 20;
    int perCU = 2; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, rowsKernel, (int)rowsBS, 0) != hipSuccess || perCU < 1) perCU = rowsBS == 512u ? 1 : 2;
    if (const char *e = getenv("YGPU_ROWS_PER_CU")) { int v = atoi(e); if (v >= 1 && v < perCU) perCU = v; }
    unsigned maxBlocksK = (unsigned)((uint64_t)ctx->nCU * perCU);
    // With other batches in flight on the device the persistent launch takes 9/16 of what fits (1.7 workgroups a CU), and only ONE such launch runs at a time on the
    // device (below: gRowsEv).  Its waves hold their registers and LDS until the launch ends; what they leave is all the other batches' latency-bound kernels get to
    // run in meanwhile -- and two rows launches side by side would take the whole chip between them again.  Four contexts, 3.1 Gbp, ms a step
    // (profiles/r05_rows_blocks_sweep.txt): the full launch, free-running (rounds 1-4) 44.4-45.0; 384 workgroups free-running 43.6-44.0; one at a time: 352 workgroups
    // 44.2-44.3, 384 43.4-43.8, 416 42.5-43.0, 448 42.2-43.0, 480 43.5-43.9.  Alone on the device the full launch is 2.7 ms a step faster than half of it.
    // (YGPU_ROWS_BLOCKS: the workgroups as a count, for such sweeps.)
    if (rowsShare && rowsBS == 256u) maxBlocksK = std::max(64u, maxBlocksK * 9u / 16u);
    if (const char *e = getenv("YGPU_ROWS_BLOCKS")) { const long v = atol(e); if (v >= 64 && v <= (long)ctx->nCU * perCU) maxBlocksK = (unsigned)v; }
    const unsigned maxWavesK = maxBlocksK * (rowsBS / 64u);
    const double chunkBlocks = (double)YD_CHUNK_FLUSHES * 64.0;             // lane blocks (128 B) per chunk
    size_t freeB = 0, totB = 0; hipMemGetInfo(&freeB, &totB);
    const int nShare = std::max(1, gCtxPerDevice[ctx->device & 63].load());
    // this context's budget: an equal share of 60 % of the device's memory whatever the order the contexts get here in (the first one used to take most of what
    // was free and left the others to cut their batches into ranges), and no more than what is free now
    const size_t fairB = (size_t)((double)totB * 0.6 / nShare), availB = (size_t)((double)(freeB + ctx->extTrace.cap) * 0.8);
    unsigned long long budgetChunks = std::max<unsigned long long>(maxWavesK + 64ull, (unsigned long long)std::min(fairB, availB) / (YD_CHUNK_DWORDS * 4ull));
    budgetChunks = std::min<unsigned long long>(budgetChunks, (96ull << 30) / (YD_CHUNK_DWORDS * 4ull));
    if (ctx->traceRatio <= 0.0) {
        // first batch: an X-drop run stops after ~100-200 rows whatever its bound (most roots are chance hits), so the share of the bound that gets used
        // follows the mean bound; too small an estimate costs a redo of this stage (the arena doubles), too large a one memory the other contexts need
        const double meanBoundRows = 10.0 * (double)boundBlocks / std::max(1u, nProb);
        // (rows an X-drop run computes, as the first batches of real runs showed them, with the 1.3 margin: 170 for 1 kbp reads -- mean bound 425 rows -- and 230 for
        // 10 kbp reads -- mean bound 5 000; a guess that is too small costs a redo of the stage with an arena half as large again, and an arena that was made too
        // large stays: giving 40 GB back and asking for 26 stalled every context of the device for 3.4 s, measured, profiles/r04_cli_10kbp.txt)
        ctx->traceRatio = std::min(0.6, std::max(0.02, (165.0 + 0.02 * meanBoundRows) / std::max(1.0, meanBoundRows)));
    }
    const double slackChunks = (double)maxWavesK + (double)ctx->nCU * 8.0 + 64.0;   // every wave's open chunk, and the careful-extension round's
    const double wantChunks = (double)boundBlocks * ctx->traceRatio / chunkBlocks + slackChunks;
    std::vector<uint32_t> cuts;                                               // root indices
    unsigned long long nChunksArena = 0; size_t nRanges = 1; bool haveBounds = false;
    for (;;) {
        cuts.assign(1, 0);
        if (wantChunks <= (double)budgetChunks && ctx->traceBudgetBlocks <= 0) { cuts.push_back(NC); nChunksArena = (unsigned long long)wantChunks; }
        else {
            // ranges of roots whose estimated need fits the budget (the estimate follows the problems' bounds); YGPU_TRACE_BUDGET_BLOCKS (test hook) sets the
            // bound blocks per range directly, so that small inputs take this path
            double perRange = std::max(1.0, ((double)budgetChunks - slackChunks) * chunkBlocks / ctx->traceRatio);                                  // bound blocks per range
            if (ctx->traceBudgetBlocks > 0) perRange = std::min(perRange, (double)ctx->traceBudgetBlocks);
            nChunksArena = (unsigned long long)std::min((double)budgetChunks, perRange * ctx->traceRatio / chunkBlocks + slackChunks);
            if (!haveBounds) {
                ctx->hStripOff.resize(nProb + 1);
                HIPCHK(hipMemcpyAsync(ctx->hStripOff.data(), ctx->stripOff.p, 8ull * (nProb + 1), hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx)); haveBounds = true;
            }
            uint32_t r0 = 0;
            while (r0 < NC) {
                uint32_t lo = r0 + 1, hi = NC;                                    // largest r1 with bound(r0 .. r1) <= perRange (at least one root)
                while (lo < hi) { uint32_t mid = lo + (hi - lo + 1) / 2; if ((double)(ctx->hStripOff[2 * (size_t)mid] - ctx->hStripOff[2 * (size_t)r0]) <= perRange) lo = mid; else hi = mid - 1; }
                cuts.push_back(lo); r0 = lo;
            }
        }
        nRanges = cuts.size() - 1;
        const unsigned long long minChunks = maxWavesK + 64ull;
        nChunksArena = std::min<unsigned long long>(std::max<unsigned long long>(nChunksArena, minChunks), 0xFFFFFFF0ull);
        {   // an arena that is there and within the estimate's safety margin is not re-allocated for the margin's sake (freeing and allocating tens of GB stalls
            // every context of the device; should it overflow, the stage is redone with twice as much)
            const unsigned long long capChunks = ctx->extTrace.cap > 256 ? (unsigned long long)((ctx->extTrace.cap - 256) / (YD_CHUNK_DWORDS * 4ull)) : 0ull;
            if (nRanges == 1 && capChunks > minChunks && nChunksArena > capChunks && (double)nChunksArena <= 1.3 * (double)capChunks) nChunksArena = capChunks;
        }
        // The arena is idle here (the stage has not started): the old one is freed before the new one is asked for, and the request is exact, so that a
        // budget that counts the old arena as reusable can be met.  A request the device refuses although the budget allowed it (another process took the
        // memory meanwhile, fragmentation) is halved and the roots are cut into ranges for what there is; only an arena that cannot even hold every wave's
        // open chunk is an error.
        if (ctx->extTrace.ensureExact((size_t)nChunksArena * YD_CHUNK_DWORDS * 4ull + 256) == 0) break;
        (void)hipGetLastError();
        if (nChunksArena <= minChunks) { ctx->err = "hipMalloc failed for the extension trace arena (not even one chunk per wave fits)"; return YGPU_ENOMEM; }
        budgetChunks = std::max<unsigned long long>(minChunks, nChunksArena / 2);
        if (kTrace) fprintf(stderr, "[ygpu] trace arena of %llu chunks refused by the device: retrying with %llu and ranges\n", nChunksArena, budgetChunks);
    }
    ctx->statRanges = (int)nRanges;
    nChunksArena = std::min<unsigned long long>(0xFFFFFFF0ull, (unsigned long long)((ctx->extTrace.cap - 256) / (YD_CHUNK_DWORDS * 4ull)));
    const uint32_t maxCh = (uint32_t)std::min<unsigned long long>(nChunksArena, std::max<unsigned long long>(64ull, 4ull * nChunksArena / std::max(1u, maxWavesK) + 64ull));
    ENSURE(ctx->waveChunks, 4ull * (size_t)maxWavesK * maxCh + 64);
    // the op lists of the extensions (exactly sized slots, k_ext_trace): sized like the arena, from the ratio of the last batches
    const uint32_t extOpsCap = (uint32_t)std::min<double>(2.0e9, (double)boundBlocks * 10.0 * ctx->opsRatio + 4.0e6);
    ENSURE(ctx->extOps, 4ull * extOpsCap + 64);
    ENSURE(ctx->chunkCnt, 32ull * (nRanges + 2));                            // 8 words per range: queues and counts of its kernels
    HIPCHK(hipMemsetAsync(ctx->chunkCnt.p, 0, 32ull * (nRanges + 2), ctx->stream));
    ENSURE(ctx->traceCnt, 64); HIPCHK(hipMemsetAsync(ctx->traceCnt.p, 0, 64, ctx->stream));       // [0] chunks handed out, [1] ops handed out, [2] the high-water mark of [0] over the ranges
    TRACE("lanes: trace arena");
    if (kTrace && getenv("YGPU_COUNT_DUPS")) {                                // diagnostics: how many extension problems of the batch are exact duplicates (direction, strand, read, rOff, qOff, qLen)?
        std::vector<ExtProb> hp(nProb); hipMemcpy(hp.data(), ctx->extProbs.p, sizeof(ExtProb) * (size_t)nProb, hipMemcpyDeviceToHost);
        std::vector<std::array<uint32_t, 4>> keys; keys.reserve(nProb);
        for (auto &e : hp) if (e.flags & XP_VALID) keys.push_back({e.qBase, e.rOff, (uint32_t)e.qOff | ((uint32_t)e.qLen << 16), e.flags & 3u});
        std::sort(keys.begin(), keys.end()); size_t dup = 0, sameStart = 0;
        for (size_t k = 1; k < keys.size(); k++) { dup += keys[k] == keys[k - 1]; sameStart += keys[k][0] == keys[k - 1][0] && keys[k][1] == keys[k - 1][1] && (keys[k][2] & 0xFFFF) == (keys[k - 1][2] & 0xFFFF) && keys[k][3] == keys[k - 1][3]; }
        fprintf(stderr, "[ygpu] extension problems: %zu valid, %zu exact duplicates (%.2f%%), %zu share (read, strand, direction, rOff, qOff) with their predecessor (%.2f%%)\n", keys.size(), dup, 100.0 * dup / std::max<size_t>(1, keys.size()), sameStart, 100.0 * sameStart / std::max<size_t>(1, keys.size()));
    }
    if (kTrace) fprintf(stderr, "[ygpu] trace bound %.2f GB, arena %.2f GB (%llu chunks, ratio %.3f), %zu range(s); ext ops cap %u\n", boundBlocks * 128.0 / 1e9, nChunksArena * (YD_CHUNK_DWORDS * 4.0) / 1e9, nChunksArena, ctx->traceRatio, nRanges, extOpsCap);
    ENSURE(ctx->rowsClock, 16); { const unsigned long long init[2] = {~0ull, 0ull}; HIPCHK(hipMemcpyAsync(ctx->rowsClock.p, init, 16, hipMemcpyHostToDevice, ctx->stream)); }
    E.clock = ctx->rowsClock.as<unsigned long long>();
    E.trace = ctx->extTrace.as<uint32_t>(); E.nChunks = (uint32_t)nChunksArena; E.chunkCount = ctx->traceCnt.as<unsigned int>(); E.waveChunks = ctx->waveChunks.as<uint32_t>(); E.maxCh = maxCh;
    E.ops = ctx->extOps.as<uint32_t>(); E.opsCount = ctx->traceCnt.as<unsigned int>() + 1; E.opsCap = extOpsCap;
    E.ctr = ctx->ctr.as<DevCounters>(); E.errFlag = ctx->errFlag.as<int>(); E.dbgMode = getenv("YGPU_TRACE_MODE") ? atoi(getenv("YGPU_TRACE_MODE")) : 0;
    X.extOps = ctx->extTrace.as<uint32_t>();                                 // the base the op lists' offsets refer to
    uint32_t *cc = ctx->chunkCnt.as<uint32_t>();
    unsigned long long usedChunksMax = 0;
    EV0(T_XROWS);
    for (size_t c = 0; c < nRanges; c++) {
        const uint32_t r0 = cuts[c], r1 = cuts[c + 1], p0 = 2 * r0, np = 2 * (r1 - r0);
        E.probs = ctx->extProbs.as<ExtProb>() + p0; E.nProb = np; E.res = ctx->extRes.as<ExtRes>() + p0;
        E.queue = cc + 8 * c;
        HIPCHK(hipMemsetAsync(ctx->traceCnt.p, 0, 4, ctx->stream));          // the arena starts empty for every range (the previous one's lists are in the ops arena)
        HIPCHK(hipMemsetAsync(E.res, 0, sizeof(ExtRes) * (uint64_t)np, ctx->stream));   // a launch that runs out of arena leaves problems unfinished: they must read as "no extension", not as the last batch's results
        {   // longest bound first: the launch's drain phase is then left with short problems only.  Keys 0xFFFF - qLen (invalid: 0xFFFF, last): one bucket per
            // length while the longest read has fewer than 4 096 bases, per 2^k lengths beyond; the values are the problems' indices inside the range
            const uint32_t sub = 0xFFFFu - (uint32_t)std::min(ctx->maxQ, 0xFFFF); int shift = 0; while (((uint32_t)ctx->maxQ >> shift) >= YD_BKT_MAX - 1u) shift++;
            uint32_t *v1 = ctx->extOrder.as<uint32_t>() + p0;
            rc = bucketOrder(ctx, ctx->extKeys.as<uint32_t>() + p0, nullptr, 0, np, sub, shift, ((uint32_t)ctx->maxQ >> shift) + 2u, v1, ctx->stream); if (rc) return rc;
            E.order = v1;
        }
        const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)np + rowsBS - 1) / rowsBS, (uint64_t)maxBlocksK);
        static const int rowsSerial = getenv("YGPU_ROWS_SERIAL") ? atoi(getenv("YGPU_ROWS_SERIAL")) : 1;
        // (one at a time only for short reads: a launch of 10 kbp problems ends in a long tail of a few lanes, and the next one would wait for all of it -- 10 kbp reads,
        // four contexts, ms a step: 432 workgroups free-running 29.4, one at a time 30.6, the full launch free-running 30.6)
        if (rowsSerial && rowsShare && ctx->maxQ <= 4096) {
            std::lock_guard<std::mutex> lk(gRowsMu[ctx->device & 63]); const int dv = ctx->device & 63;
            const int depth = std::min(4, std::max(1, rowsSerial)); const int slot = (int)(gRowsSeq[dv]++ % (unsigned long long)depth);      // launch n waits for launch n - depth
            if (!gRowsEvValid[dv][slot]) { if (hipEventCreateWithFlags(&gRowsEv[dv][slot], hipEventDisableTiming) == hipSuccess) gRowsEvValid[dv][slot] = true; }
            else HIPCHK(hipStreamWaitEvent(ctx->stream, gRowsEv[dv][slot], 0));
            KL(rowsKernel, dim3(blocks), dim3(rowsBS), 0, ctx->stream, E);
            if (gRowsEvValid[dv][slot]) HIPCHK(hipEventRecord(gRowsEv[dv][slot], ctx->stream));
        } else
        KL(rowsKernel, dim3(blocks), dim3(rowsBS), 0, ctx->stream, E);
        if (c + 1 == nRanges) EV1(T_XROWS);
        TRACE("lanes: ext_rows");
        if (c == 0) { ctx->evUsed[T_XTRACE] = true; hipEventRecord(ctx->ev[T_XTRACE][0], ctx->stream); }
        if (kTrace && getenv("YGPU_TRACE_LENS")) {                            // diagnostics: the walks of the traceback, per problem and per wave of 64 in k_ext_rows' order
            HIPCHK(streamSync(ctx));
            std::vector<ExtRes> hr(np); std::vector<uint32_t> ho(np);
            hipMemcpy(hr.data(), E.res, sizeof(ExtRes) * (size_t)np, hipMemcpyDeviceToHost); hipMemcpy(ho.data(), E.order, 4ull * np, hipMemcpyDeviceToHost);
            unsigned long long walkers = 0, sumLen = 0, sumWaveMax = 0, sumRows = 0, hist[8] = {0}; std::vector<uint32_t> lens; lens.reserve(np);
            for (uint32_t w = 0; w < np; w += 64) { uint32_t mx = 0; for (uint32_t k = w; k < std::min(np, w + 64); k++) { const ExtRes &r = hr[ho[k]]; const uint32_t len = r.score > 0 ? (uint32_t)r.maxi : 0u; walkers += r.score > 0; sumLen += len; sumRows += r.rows; mx = std::max(mx, len); int b = 0; while (b < 7 && (len >> (b + 3))) b++; hist[len ? b : 0] += 1; } sumWaveMax += mx; }
            fprintf(stderr, "[ygpu] traceback: %u problems, %llu walk (%.1f%%), mean walk %.1f rows (all) / %.1f (walkers), rows computed mean %.1f; sum over waves of the longest walk %llu = %.1f x the lanes' mean\n",
                    np, walkers, 100.0 * walkers / np, (double)sumLen / np, (double)sumLen / std::max(1ull, walkers), (double)sumRows / np, sumWaveMax, (double)sumWaveMax * 64.0 / std::max(1ull, sumLen));
            fprintf(stderr, "[ygpu] walk length histogram (0..7, 8.., 16.., 32.., 64.., 128.., 256.., 512..):"); for (int b = 0; b < 8; b++) fprintf(stderr, " %llu", hist[b]); fprintf(stderr, "\n");
        }
        if (traceSort > 0 && np > 4096u) {                                    // traceback order: by arena region, then by walk length (k_trace_keys)
            uint32_t *k0 = ctx->extKeys.as<uint32_t>() + p0, *v0 = ctx->extVals.as<uint32_t>() + p0, *v1 = ctx->extKeys2.as<uint32_t>() + p0;      // (v1: not extOrder, which k_trace_keys reads)
            static const int lenBits = getenv("YGPU_TRACE_LENBITS") ? std::min(8, std::max(1, atoi(getenv("YGPU_TRACE_LENBITS")))) : 7;
            int lenShift = 0; while ((ctx->maxQ >> lenShift) > (1 << lenBits) - 1) lenShift++;
            int keyBits = lenBits; while (keyBits < 32 && ((unsigned long long)E.nChunks >> std::min(traceSort, 31)) >> (keyBits - lenBits)) keyBits++;        // region bits above the length bits
            KL(k_trace_keys, dim3(gridFor(np, 256)), dim3(256), 0, ctx->stream, E.res, E.order, np, E.waveChunks, E.maxCh, std::min(traceSort, 31), lenShift, lenBits, k0, v0);
            rc = bucketOrder(ctx, k0, v0, 0, np, 0, std::max(0, keyBits - 12), 1u << std::min(keyBits, 12), v1, ctx->stream); if (rc) return rc;
            E.order = v1;
        }
        // (the tracebacks of the contexts one at a time, as the rows launches: 43.20 against 43.19 ms a step; in the rows launches' chain: 44.89 -- profiles/r05_trace_chain.txt, commit 8c679ac)
        KL(traceKernel, dim3(gridFor(np, traceBS)), dim3(traceBS), 0, ctx->stream, E);
        if (c + 1 == nRanges) hipEventRecord(ctx->ev[T_XTRACE][1], ctx->stream);
        TRACE("lanes: ext_trace");
        if (kTrace) { unsigned w8[8]; hipMemcpyFromSymbol(w8, HIP_SYMBOL(gTraceDbg), sizeof w8); if (w8[0]) { ExtRes rr; hipMemcpy(&rr, E.res + w8[6], sizeof rr, hipMemcpyDeviceToHost); ExtProb pp; hipMemcpy(&pp, E.probs + w8[6], sizeof pp, hipMemcpyDeviceToHost);
            fprintf(stderr, "[ygpu] k_ext_trace left its strip: dword %d of %u, f0 %u, laneOff %u; problem %u where %08x (wave %u lane %u phase %u) score %d maxi %d maxj %d rows %u qLen %u flags %u\n", (int)w8[1], w8[2], w8[3], w8[4], w8[6], w8[7], w8[7] >> 10, (w8[7] >> 4) & 63, w8[7] & 15, rr.score, rr.maxi, rr.maxj, rr.rows, pp.qLen, pp.flags);
            memset(w8, 0, sizeof w8); hipMemcpyToSymbol(HIP_SYMBOL(gTraceDbg), w8, sizeof w8); } }
        AlignArgs Ac = A; Ac.nRoots = r1; Ac.queueHead = cc + 8 * c + 1;
        PhaseArgs Xc = X; Xc.rootBegin = r0; Xc.slowList = ctx->slowList.as<uint32_t>() + r0; Xc.slowCount = cc + 8 * c + 2; Xc.useList = 1;
        if (c == 0) { ctx->evUsed[T_P3] = true; hipEventRecord(ctx->ev[T_P3][0], ctx->stream); }
        const uint32_t nr = r1 - r0, cap2 = nr / 4 + 1024;
        if (ctx->splitLanes) {
            ENSURE(ctx->memoKeys, 12ull * YD_MEMO * (nr + 1)); ENSURE(ctx->memoCount, 4ull * (nr + 1)); ENSURE(ctx->probs2, sizeof(ExtProb) * (uint64_t)cap2);
            ENSURE(ctx->rowsBound2, 8ull * (cap2 + 1)); ENSURE(ctx->extRes2, sizeof(ExtRes) * (uint64_t)cap2); ENSURE(ctx->fallList, 4ull * (nr + 1));
            HIPCHK(hipMemsetAsync(ctx->memoCount.p, 0, 4ull * (nr + 1), ctx->stream)); HIPCHK(hipMemsetAsync(ctx->rowsBound2.p, 0, 8ull * (cap2 + 1), ctx->stream));
            Xc.memoKeys = ctx->memoKeys.as<uint32_t>(); Xc.memoCount = ctx->memoCount.as<unsigned int>(); Xc.probs2 = ctx->probs2.as<ExtProb>(); Xc.rowsBound2 = ctx->rowsBound2.as<unsigned long long>();
            Xc.nProb2 = cc + 8 * c + 3; Xc.probs2Cap = cap2;
        } else { Xc.memoKeys = nullptr; Xc.memoCount = nullptr; Xc.probs2 = nullptr; Xc.rowsBound2 = nullptr; Xc.nProb2 = nullptr; Xc.probs2Cap = 0; }
        KL(k_p3_lanes, dim3(gridFor(nr, 256)), dim3(256), 0, ctx->stream, Ac, Xc);
        if (ctx->splitLanes) KL(k_p3_predict, dim3((unsigned)std::min<uint64_t>(gridFor(nr, 64), (uint64_t)ctx->nCU * 8)), dim3(64), 0, ctx->stream, Ac, Xc);
        PhaseArgs Xw = Xc;                                                    // what k_align_p3 gets: all split roots, or only those k_split_lanes gives back
        // the range's use of the arena (for the next batch's estimate), then the careful-extension round starts it afresh
        unsigned int used[2] = {0, 0}; uint32_t three[3] = {0, 0, 0};       // slow roots, predicted problems
        HIPCHK(hipMemcpyAsync(used, ctx->traceCnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (ctx->splitLanes) HIPCHK(hipMemcpyAsync(three, cc + 8 * c + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(streamSync(ctx));
        usedChunksMax = std::max<unsigned long long>(usedChunksMax, used[0]);
        if (ctx->splitLanes) {
            // splitClump in lanes: the careful extensions the split roots will ask for go through a second k_ext_rows / k_ext_trace round
            const uint32_t nSlow = three[0], n2 = std::min(three[1], cap2);
            if (kTrace) fprintf(stderr, "[ygpu] range %zu: chunks used %u, ext ops %u; split roots %u, careful extensions listed %u\n", c, used[0], used[1], nSlow, n2);
            if (nSlow) {
                ExtArgs E2 = E;
                if (n2) {
                    E2.probs = ctx->probs2.as<ExtProb>(); E2.nProb = n2;
                    HIPCHK(hipMemsetAsync(ctx->extRes2.p, 0, sizeof(ExtRes) * (uint64_t)n2, ctx->stream));        // (the arena goes on: the first round's lists stay in it)
                    {   // longest bound first here too: this launch is small and ends when its longest problem ends
                        ENSURE(ctx->keys2a, 4ull * (cap2 + 1)); ENSURE(ctx->keys2b, 4ull * (cap2 + 1)); ENSURE(ctx->vals2a, 4ull * (cap2 + 1)); ENSURE(ctx->vals2b, 4ull * (cap2 + 1));
                        KL(k_prob_keys, dim3(gridFor(n2, 256)), dim3(256), 0, ctx->stream, ctx->probs2.as<ExtProb>(), n2, ctx->keys2a.as<uint32_t>(), ctx->vals2a.as<uint32_t>());
                        { const uint32_t sub = 0xFFFFu - (uint32_t)std::min(ctx->maxQ, 0xFFFF); int shift = 0; while (((uint32_t)ctx->maxQ >> shift) >= YD_BKT_MAX - 1u) shift++;
                          rc = bucketOrder(ctx, ctx->keys2a.as<uint32_t>(), nullptr, 0, n2, sub, shift, ((uint32_t)ctx->maxQ >> shift) + 2u, ctx->vals2b.as<uint32_t>(), ctx->stream); if (rc) return rc; }
                    }
                    E2.order = ctx->vals2b.as<uint32_t>(); E2.res = ctx->extRes2.as<ExtRes>(); E2.queue = cc + 8 * c + 4; E2.ctr = nullptr;   // counted by k_split_lanes
                    {   // a small launch: a few problems per lane, so its length is set by the lanes' chains of problems, not by the chip's throughput.  One wave
                        // per SIMD runs a row 2.4x faster than three sharing it (a lone wave issues every ~5 cycles) and gives every lane more problems to balance.
                        const uint64_t blocks2 = ctx->rows2PerCU > 0 ? (uint64_t)ctx->rows2PerCU : (uint64_t)ctx->nCU;
                        const uint64_t b2 = rows2BS == 512u ? std::max<uint64_t>(1, blocks2 / 2) : blocks2;
                        KL(rowsKernel2, dim3((unsigned)std::min<uint64_t>(std::min<uint64_t>(((uint64_t)n2 + rows2BS - 1) / rows2BS, b2), (uint64_t)maxBlocksK)), dim3(rows2BS), 0, ctx->stream, E2); }
                    KL(traceKernel, dim3(gridFor(n2, traceBS)), dim3(traceBS), 0, ctx->stream, E2);
                }
                ENSURE(ctx->splitScratch, (size_t)YD_SL_BYTES * (((size_t)nSlow + 63) / 64 * 64));
                SplitArgs Sx; Sx.scratch = ctx->splitScratch.as<uint8_t>(); Sx.memoKeys = ctx->memoKeys.as<uint32_t>(); Sx.memoCount = ctx->memoCount.as<unsigned int>();
                Sx.res2 = ctx->extRes2.as<ExtRes>(); Sx.ops2 = ctx->extTrace.as<uint32_t>(); Sx.nProb2 = n2;
                Sx.fallList = ctx->fallList.as<uint32_t>(); Sx.fallCount = cc + 8 * c + 5; Sx.nSlots = nSlow;
                KL(k_split_lanes, dim3(gridFor(nSlow, 64)), dim3(64), 0, ctx->stream, Ac, Xc, Sx);
                Xw.slowList = ctx->fallList.as<uint32_t>(); Xw.slowCount = cc + 8 * c + 5;
                if (kTrace) { uint32_t fc = 0; HIPCHK(hipMemcpyAsync(&fc, cc + 8 * c + 5, 4, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx)); unsigned w8[8]; hipMemcpyFromSymbol(w8, HIP_SYMBOL(gFallWhy), sizeof w8); fprintf(stderr, "[ygpu] roots left to the wave kernel %u (other %u, DP not listed %u, second split %u, depth/list %u)\n", fc, w8[0], w8[1], w8[2], w8[3]); memset(w8, 0, sizeof w8); hipMemcpyToSymbol(HIP_SYMBOL(gFallWhy), w8, sizeof w8); }
            }
        }
        {   // the wave-per-root kernel takes what the lane kernels hand back: nothing at all on ordinary batches, and 1 024 waves that only find an empty list cost
            // 0.3 ms -- so its grid follows what the last batch handed back (all split roots when k_split_lanes is off)
            unsigned p3Waves = std::min<unsigned>(waves, std::max<unsigned>(64u, (r1 - r0) / 8u));
            if (ctx->splitLanes && ctx->lastFall >= 0) p3Waves = std::min<unsigned>(p3Waves, std::max<unsigned>(64u, (unsigned)std::min<long long>(1ll << 20, 2ll * ctx->lastFall)));
            KL(k_align_p3, dim3(p3Waves), dim3(64), 0, ctx->stream, Ac, Xw);
            if (ctx->splitLanes && c + 1 == nRanges) HIPCHK(hipMemcpyAsync(&ctx->hFall, Xw.slowCount, 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (c + 1 == nRanges) hipEventRecord(ctx->ev[T_P3][1], ctx->stream);
        TRACE("lanes: p3");
        if (kTrace && getenv("YGPU_LIST_HIST")) {      // what k_p3_lanes walks: lengths of the three lists of a root (backward extension, phase-1 list, forward extension)
            const uint32_t nr2 = r1 - r0; std::vector<ExtRes> hr2(2 * (size_t)nr2); std::vector<RootState> hs(nr2);
            hipMemcpy(hr2.data(), ctx->extRes.as<ExtRes>() + 2 * (size_t)r0, sizeof(ExtRes) * hr2.size(), hipMemcpyDeviceToHost); hipMemcpy(hs.data(), ctx->rootState.as<RootState>() + r0, sizeof(RootState) * nr2, hipMemcpyDeviceToHost);
            unsigned long long hx[8] = {0}, hb[8] = {0}, ht[8] = {0}, sumx = 0, sumb = 0; const unsigned edges[7] = {0, 1, 2, 4, 8, 16, 32};
            auto bin = [&](unsigned v) { int k = 0; while (k < 7 && v > edges[k]) k++; return k; };
            for (uint32_t k = 0; k < nr2; k++) { const unsigned a = hr2[2 * k].score > 0 ? hr2[2 * k].nOps : 0u, c2 = hr2[2 * k + 1].score > 0 ? hr2[2 * k + 1].nOps : 0u, b = hs[k].len; hx[bin(a)]++; hx[bin(c2)]++; hb[bin(b)]++; ht[bin(a + b + c2)]++; sumx += a + c2; sumb += b; }
            fprintf(stderr, "[ygpu] list lengths over %u roots (bins: 0, 1, 2, 3-4, 5-8, 9-16, 17-32, more): extension lists", nr2); for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", hx[k]);
            fprintf(stderr, "; phase-1 lists"); for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", hb[k]); fprintf(stderr, "; merged"); for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", ht[k]);
            fprintf(stderr, "; mean ops per root: extensions %.1f, phase 1 %.1f\n", (double)sumx / nr2, (double)sumb / nr2);
        }
    }
    HIPCHK(hipMemcpyAsync(ctx->hRowsClock, ctx->rowsClock.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    // errors of the trace memory: grow what overflowed and have the caller redo the stage
    unsigned int usedOps = 0; HIPCHK(hipMemcpyAsync(&usedOps, ctx->traceCnt.as<unsigned int>() + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    { const FetchPiece pc[2] = {{ctx->errFlag.p, &ef, 1}, {cnt + CNT_OUTCLUMPS, ctx->hOutCounts, 2}}; rc = fetchMany(ctx, pc, 2); if (rc) return rc; }      // (stageAlign reads the output counts from here)
    ctx->hOutEf = ef; ctx->hOutValid = true;
    if (ctx->splitLanes) ctx->lastFall = (long long)ctx->hFall;              // (the fetch above synchronised the stream)
    if (ef == YERR_TRACEMEM) {
        if (ctx->traceRatio >= 64.0) { ctx->err = "the extension trace arena overflows even at 64 times the problems' bound"; return YGPU_ENOMEM; }
        ctx->traceRatio = std::min(64.0, ctx->traceRatio * 1.5); return -3;
    }
    if (ef == YERR_OUT && usedOps > extOpsCap) { ctx->opsRatio = std::min(4.0, std::max(ctx->opsRatio * 2.0, 1.3 * (double)usedOps / std::max(1.0, (double)boundBlocks * 10.0))); return -3; }
    if (ef == 0 && boundBlocks) {
        // next batch's estimate: what this one used, with a margin
        const double usedRatio = ((double)usedChunksMax - (double)std::min<unsigned long long>(usedChunksMax, maxWavesK)) * chunkBlocks * (double)nRanges / (double)boundBlocks;
        // (followed at once: an estimate that turns out too small costs one redo of the stage, one that stays too large costs memory and, for long reads whose
        // bound is a hundred times their use, forces the ranges)
        ctx->traceRatio = std::max(0.01, usedRatio * 1.3);
        ctx->opsRatio = std::max(0.002, std::max(1.3 * (double)usedOps / ((double)boundBlocks * 10.0), ctx->opsRatio * 0.7));
    }
    TRACE("lanes: ranges done");
    return 0;
}

// ---- A5..A8 + layout ---------------------------------------------------------------------------------------------
static int stageAlign(ygpu_ctx *ctx)
{
    const uint32_t n = ctx->nReads, NC = ctx->nClumps; DevBatch B = devBatch(ctx); int rc;
    uint32_t *cnt = ctx->counters.as<uint32_t>();
    ENSURE(ctx->readCount, 4ull * (n + 1)); ENSURE(ctx->readStart, 4ull * (n + 1));
    HIPCHK(hipMemsetAsync(ctx->readCount.p, 0, 4ull * (n + 1), ctx->stream));
    ctx->nOut = ctx->nOutOps = 0;
    if (NC) {
        TRACE("before align");
        EV0(T_ALIGN);
        int listCap, front, genCap, traceRows; alignDims(ctx, listCap, front, genCap, traceRows);
        const size_t per = alignScratchBytes(ctx->maxQ, traceRows, listCap, genCap);
        size_t freeB = 0, totB = 0; hipMemGetInfo(&freeB, &totB);
        uint64_t maxWaves = std::max<uint64_t>(64, (uint64_t)((freeB / std::max(1, gCtxPerDevice[ctx->device & 63].load()) + ctx->scratchAlign.cap) * 6 / 10) / per);
        // the default band runs its X-drop extensions one problem per lane (ext_lanes.h); other bands stay on the wave kernel
        const bool useLanes = ctx->laneExt && ctx->P.bandWidth == 5 && ctx->P.maxGap >= YD_LBAND;
        // waves of the wave-per-root kernels (and their scratch, ~1 KB per query base each): the whole stage without the lane kernels, only the roots those
        // hand back with them
        const unsigned wavesPerCU = ctx->alignWavesPerCU > 0 ? (unsigned)ctx->alignWavesPerCU : (useLanes ? 4u : 12u);
        unsigned waves = (unsigned)std::min<uint64_t>(std::min<uint64_t>(NC, (uint64_t)ctx->nCU * wavesPerCU), maxWaves);      // 3 waves per SIMD (137 VGPRs)
        // (with the lane kernels doing the bulk the wave kernels see the roots those hand back -- none on ordinary batches -- and the gap fills beyond the lane kernels'
        // limits: their scratch, ~1 KB per query base and wave, is held to 3 GB -- 10 kbp reads took 10.7 GB a context for 1 024 waves that had nothing to do)
        if (useLanes && per * waves > (3ull << 30)) waves = (unsigned)std::min<uint64_t>(waves, std::max<uint64_t>(64, (3ull << 30) / per));      // (the cap only ever lowers the count)
        ENSURE(ctx->scratchAlign, per * waves);
        ENSURE(ctx->clumpFrags0, 16ull * (ctx->nClumpFrags + 1));
        HIPCHK(hipMemcpyAsync(ctx->clumpFrags0.p, ctx->clumpFrags.p, 16ull * ctx->nClumpFrags, hipMemcpyDeviceToDevice, ctx->stream));
        ENSURE(ctx->rootPush, 4ull * (NC + 1)); ENSURE(ctx->rootBase, 4ull * (NC + 1));
        uint32_t stateOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 32ull * NC + 8ull * ctx->nClumpFrags + 65536);
        uint32_t gapOpsPerJoint = 16;
        uint32_t outClumpCap = NC + NC / 2 + 1024; uint32_t outOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 32ull * NC + ctx->totalBases / 2 + 65536);
        for (int attempt = 0;; attempt++) {
            ctx->statAttempts = attempt + 1;
            ENSURE(ctx->outClumps, sizeof(ygpu_clump) * (uint64_t)outClumpCap); ENSURE(ctx->outClumps2, sizeof(ygpu_clump) * (uint64_t)outClumpCap);
            ENSURE(ctx->outOps, 4ull * outOpsCap); ENSURE(ctx->outRoot, 4ull * outClumpCap); ENSURE(ctx->outPush, 4ull * outClumpCap); ENSURE(ctx->dstIdx, 4ull * outClumpCap);
            HIPCHK(hipMemsetAsync(cnt + CNT_QALIGN, 0, 12, ctx->stream));      // qalign, outclumps, outops
            HIPCHK(hipMemsetAsync(ctx->rootPush.p, 0, 4ull * (NC + 1), ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->errFlag.p, 0, 4, ctx->stream));
            AlignArgs A; A.P = ctx->P; A.bases = ctx->dBases.as<uint8_t>(); A.B = B; A.order = ctx->order.as<uint32_t>(); A.nRoots = NC;
            A.clumps = ctx->clumps.as<ChainClumpRec>(); A.clumpFrags = ctx->clumpFrags.as<DevFrag>(); A.queueHead = cnt + CNT_QALIGN;
            A.scratch = ctx->scratchAlign.as<uint8_t>(); A.scratchPerWave = per; A.maxQ = ctx->maxQ; A.listCap = listCap; A.front = front; A.genCap = genCap; A.traceRows = traceRows;
            A.outClumps = ctx->outClumps.as<ygpu_clump>(); A.outOps = ctx->outOps.as<uint32_t>(); A.outRoot = ctx->outRoot.as<uint32_t>(); A.outPush = ctx->outPush.as<uint32_t>();
            A.outCounts = cnt + CNT_OUTCLUMPS; A.outClumpCap = outClumpCap; A.outOpsCap = outOpsCap; A.rootPushCount = ctx->rootPush.as<unsigned int>();
            A.ctr = ctx->ctr.as<DevCounters>(); A.errFlag = ctx->errFlag.as<int>();
#ifdef YD_PROF
            { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(gProf), z, sizeof z); hipMemcpyToSymbol(HIP_SYMBOL(gRowsProf), z, sizeof(unsigned long long) * 8); }
#endif
            bool laneOverflow = false, traceOverflow = false; ctx->hOutValid = false;
            if (!useLanes) KL(k_align, dim3(waves), dim3(64), 0, ctx->stream, A);
            else { rc = alignWithLaneExtensions(ctx, A, waves, stateOpsCap, gapOpsPerJoint); if (rc == -2) laneOverflow = true; else if (rc == -3) traceOverflow = true; else if (rc) return rc; }
#ifdef YD_PROF
            { streamSync(ctx); unsigned long long z[16]; hipMemcpyFromSymbol(z, HIP_SYMBOL(gProf), sizeof z);
              const char *nm[10] = {"root_total", "dp_rows", "traceback", "perfect_ext", "score", "emit", "split", "merge", "dp_calls", "roots"};
              fprintf(stderr, "[YD_PROF] waves %u:", waves); for (int i = 0; i < 10; i++) fprintf(stderr, " %s=%llu", nm[i], z[i]); fprintf(stderr, "\n");
              unsigned long long q[8]; hipMemcpyFromSymbol(q, HIP_SYMBOL(gRowsProf), sizeof q);      // k_ext_rows_pk: where its passes go
              if (q[0]) fprintf(stderr, "[YD_PROF] k_ext_rows_pk: wave passes %llu; of them writing results %.1f %%, with a new maximum in some lane %.1f %%, handing blocks over %.1f %%; refill rounds %.3f a pass (pool loads %.4f); busy lanes %.1f of 64\n",
                                q[0], 100.0 * q[1] / q[0], 100.0 * q[3] / q[0], 100.0 * q[5] / q[0], (double)q[2] / q[0], (double)q[6] / q[0], (double)q[4] / q[0]); }
#endif
            uint32_t got[2] = {0, 0}, ef = 0;
            if (laneOverflow || traceOverflow) ef = YERR_OUT;
            else if (ctx->hOutValid) { got[0] = ctx->hOutCounts[0]; got[1] = ctx->hOutCounts[1]; ef = ctx->hOutEf; }      // (fetched with the lane pipeline's last wait)
            else { const FetchPiece pc[2] = {{cnt + CNT_OUTCLUMPS, got, 2}, {ctx->errFlag.p, &ef, 1}}; rc = fetchMany(ctx, pc, 2); if (rc) return rc; }
            if (ef == 0) { ctx->nOut = got[0]; ctx->nOutOps = got[1]; break; }
            if (kStats) fprintf(stderr, "[ygpu] ctx %p: align attempt %d repeated (%s); trace ratio %.3f, ops ratio %.4f\n", (void *)ctx, attempt + 1, traceOverflow ? "trace / extension-op arena" : (laneOverflow ? "phase-1 arenas (state ops, gap ops)" : "output arenas"), ctx->traceRatio, ctx->opsRatio);
            if (ef != YERR_OUT || attempt >= 24)      /* the trace estimate grows by half a time: 1.5^20 covers the floor-to-cap range */ { char b[96]; snprintf(b, sizeof b, "align stage failed with device error %u (see dp_wave.h YERR_*)", ef); ctx->err = b; return ef == YERR_OUT ? YGPU_EOVERFLOW : YGPU_EINTERNAL; }
            if (!traceOverflow) {                                            // (a full trace arena has grown its own estimate)
                outClumpCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 2ull * outClumpCap); outOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 2ull * outOpsCap);      // (capped: up to 24 attempts, and a wrapped bound would never fit)
                gapOpsPerJoint = std::min<uint32_t>(gapOpsPerJoint * 2u, 1u << 16); stateOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 2ull * stateOpsCap);
            }
            HIPCHK(hipMemcpyAsync(ctx->clumpFrags.p, ctx->clumpFrags0.p, 16ull * ctx->nClumpFrags, hipMemcpyDeviceToDevice, ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->ctr.as<DevCounters>()->v + C_SCORED, 0, 8 * (16 - C_SCORED), ctx->stream));
        }
        TRACE("align: fetch");
        EV1(T_ALIGN);
        EV0(T_LAYOUT);
        rc = cubScan(ctx, ctx->rootPush.as<uint32_t>(), ctx->rootBase.as<uint32_t>(), NC + 1); if (rc) return rc;
        if (ctx->nOut) {
            KL(k_out_layout, dim3(gridFor(ctx->nOut, 256)), dim3(256), 0, ctx->stream, ctx->outRoot.as<uint32_t>(), ctx->outPush.as<uint32_t>(), ctx->rootBase.as<uint32_t>(), ctx->rootPush.as<unsigned int>(), ctx->nOut, ctx->dstIdx.as<uint32_t>());
            KL(k_out_scatter, dim3(gridFor(ctx->nOut, 256)), dim3(256), 0, ctx->stream, ctx->outClumps.as<ygpu_clump>(), ctx->dstIdx.as<uint32_t>(), ctx->nOut, ctx->outClumps2.as<ygpu_clump>());
        }
        KL(k_read_counts, dim3(gridFor(NC, 256)), dim3(256), 0, ctx->stream, ctx->clumps.as<ChainClumpRec>(), ctx->order.as<uint32_t>(), ctx->rootPush.as<unsigned int>(), NC, ctx->readCount.as<unsigned int>());
    }
    rc = cubScan(ctx, ctx->readCount.as<uint32_t>(), ctx->readStart.as<uint32_t>(), n + 1); if (rc) return rc;
    if (NC) EV1(T_LAYOUT);
    return 0;
}

static int runTo(ygpu_ctx *ctx, int stage)
{
    int rc;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->stageDone < 1) { rc = stageSeed(ctx); if (rc) return rc; EV0(T_FRAGS); rc = buildFrags(ctx); if (rc) return rc; if (!ctx->nFrags) EV1(T_FRAGS); ctx->stageDone = 1; }
    if (stage >= 2 && ctx->stageDone < 2) { if (ctx->nFrags) { rc = stageChain(ctx); if (rc) return rc; } ctx->stageDone = 2; }
    if (stage >= 3 && ctx->stageDone < 3) { rc = stageAlign(ctx); if (rc) return rc; ctx->stageDone = 3; }
    // (the stream is drained by this fetch: the flag a look-back of scan.h raises when a tile never showed up -- the state words are then made clean again)
    uint32_t scanFail = 0; rc = fetchU32(ctx, ctx->counters.as<uint32_t>() + CNT_SCANFAIL, &scanFail); if (rc) return rc;
    if (scanFail) { if (ctx->scanState.p) HIPCHK(hipMemsetAsync(ctx->scanState.p, 0, ctx->scanState.cap, ctx->stream)); ctx->err = "exclusive sum: a tile was not published within 30 s (look-back gave up)"; return YGPU_EINTERNAL; }
    return 0;
}

static int initCommon(ygpu_ctx *ctx, int device)
{
    const bool phases = getenv("YGPU_INIT_PHASES") != nullptr; double tPh = nowMs();
    auto phase = [&](const char *what) { if (phases) { const double t = nowMs(); fprintf(stderr, "[ygpu] ctx %p: %-20s %8.1f ms\n", (void *)ctx, what, t - tPh); tPh = t; } };
    HIPCHK(hipSetDevice(device));
    // (Measured in round 4 and dropped: the context's streams at the highest priority and the long X-drop kernels on a stream of the lowest --
    // hipStreamCreateWithPriority, range -1..1 here -- 46.2-46.4 ms a step against 45.0-45.9 with four contexts, profiles/r04_ab_prio_and_waves.txt: the rows kernel's
    // waves are persistent, a slot they hold is not handed to anybody before the launch ends.)
    // (two streams a context and no more: the runtime spreads streams over four hardware queues, and with two a context the main streams of contexts 0 and 2, 1 and 3
    // share one -- their kernels take turns -- which is worth 2 ms a step against a queue for every stream and 5 against all main streams on one queue:
    // profiles/r05_hw_queues.txt)
    HIPCHK(hipStreamCreate(&ctx->stream)); HIPCHK(hipStreamCreate(&ctx->stream2));
    phase("two streams");
    for (int i = 0; i < YD_MAX_CHUNK_EV; i++) HIPCHK(hipEventCreateWithFlags(&ctx->evChunk[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->evTail, hipEventDisableTiming));
    if (const char *e = getenv("YGPU_LANE_CHUNKS")) ctx->laneChunks = atoi(e);
    if (const char *e = getenv("YGPU_TRACE_BUDGET_BLOCKS")) ctx->traceBudgetBlocks = atoll(e);
        if (const char *e = getenv("YGPU_SEGSORT_MAX")) { long v = atol(e); if (v >= 1 && v <= (long)YD_SEGSORT_MAX) ctx->segSortMax = (uint32_t)v; }
    if (const char *e = getenv("YGPU_SPLIT_LANES")) ctx->splitLanes = atoi(e);
    if (const char *e = getenv("YGPU_ROWS2_PER_CU")) ctx->rows2PerCU = atoi(e);
    if (const char *e = getenv("YGPU_ALIGN_WAVES")) ctx->alignWavesPerCU = atoi(e);
    gCtxPerDevice[device & 63]++; ctx->counted = true;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { cu = 0; (void)hipGetLastError(); } ctx->nCU = cu > 0 ? cu : 256; }
    phase("device attribute");
    for (int t = 0; t < T_N; t++) { HIPCHK(hipEventCreate(&ctx->ev[t][0])); HIPCHK(hipEventCreate(&ctx->ev[t][1])); ctx->names[t] = kStageNames[t]; }
    if (hipEventCreateWithFlags(&ctx->evSync, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { ctx->evSync = nullptr; (void)hipGetLastError(); }
    phase("events");
    if (hipHostMalloc((void **)&ctx->pinned, 256, hipHostMallocDefault) != hipSuccess) { ctx->pinned = nullptr; (void)hipGetLastError(); }
    // the post-filter's side: the second stream, a pinned slot, a wait event and look-back words of its own
    ctx->pf.stream = ctx->stream2; ctx->pf.device = device;
    if (hipHostMalloc((void **)&ctx->pf.pinned, 256, hipHostMallocDefault) != hipSuccess) { ctx->pf.pinned = nullptr; (void)hipGetLastError(); }
    if (hipEventCreateWithFlags(&ctx->pf.evSync, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) { ctx->pf.evSync = nullptr; (void)hipGetLastError(); }
    HIPCHK(hipEventCreateWithFlags(&ctx->evSnap, hipEventDisableTiming));
    if (hipHostMalloc((void **)&ctx->snapCtr, sizeof(DevCounters), hipHostMallocDefault) != hipSuccess) { ctx->snapCtr = nullptr; (void)hipGetLastError(); }
    if (ctx->pf.counters.ensure(4 * CNT_N)) { ctx->err = "hipMalloc failed"; return YGPU_ENOMEM; }
    HIPCHK(hipMemsetAsync(ctx->pf.counters.p, 0, 4 * CNT_N, ctx->stream2));
    phase("pinned words");
    return 0;
}


// ---- the index image: one copy from the host, the other devices from their neighbour ---------------------------------------------------------------------
// The reference maps the index once for all its threads (Query.c:565-626).  Here every device needs the image in its own HBM -- 16.7 GB at hg18 scale -- and N uploads
// from the host at once share the host's memory and its PCIe root ports.  So the image is cut into pieces (one linear sequence over bases, table, offsets), the FIRST
// device takes them from the host, and every further device takes piece k from the device before it as soon as that one has it (hipMemcpyPeerAsync over xGMI; a chain,
// pipelined by piece: the last device has the image a few pieces after the first).  A device that cannot reach its neighbour (hipDeviceCanAccessPeer) uploads from the
// host itself.  While the pieces travel the device's thread creates streams and events and has the code object loaded (a first kernel launch), which used to follow the copy.
namespace {
struct ImagePiece { size_t part; size_t off, bytes; };                                    // part: 0 bases, 1 table, 2 offsets
struct ImagePlan {
    const char *src[3]; size_t bytes[3]; std::vector<ImagePiece> pieces;
    void build(const ygpu_index_view *ix, size_t pieceBytes)
    {
        const uint64_t HT = 1ull << (2 * ix->wordLen);
        src[0] = (const char *)ix->bases; bytes[0] = (size_t)ix->n_base_bytes; src[1] = (const char *)ix->startingOffs; bytes[1] = (size_t)(4 * (HT + 1)); src[2] = (const char *)ix->ROA; bytes[2] = (size_t)(4ull * ix->totalMatches);
        for (size_t part = 0; part < 3; part++) for (size_t o = 0; o < bytes[part]; o += pieceBytes) pieces.push_back({part, o, std::min(pieceBytes, bytes[part] - o)});
    }
};
struct ImageState {                                                                       // one per device of a ygpu_init_multi call
    std::unique_ptr<std::atomic<int>[]> done; std::atomic<int> failed{0}; char *dst[3] = {nullptr, nullptr, nullptr};
};
__global__ void k_touch(unsigned int *p) { if (p && threadIdx.x == 1000) *p = 0; }      // (the first launch loads the library's code object: ~20 ms that need not follow the image)

// pieces from host memory: `nt` threads, each with a stream of its own, pieces taken from a common counter (the runtime's own path for unpinned memory pins a piece
// and lets the DMA engines read it in place: 55 GB/s for a single hipMemcpy of a mapped file, tools/micro/h2d_probe.hip); staged = the threads copy the pieces
// into page-locked buffers of their own first (YGPU_UPLOAD=staged:T)
static void uploadFromHost(int device, const ImagePlan &plan, ImageState &me, int nt, bool staged)
{
    std::atomic<size_t> next(0);
    auto work = [&]() {
        if (hipSetDevice(device) != hipSuccess) { me.failed = 1; return; }
        hipStream_t st; if (hipStreamCreate(&st) != hipSuccess) { me.failed = 1; return; }
        char *buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; size_t pend[2] = {0, 0}; size_t maxPiece = 0; for (auto &q : plan.pieces) maxPiece = std::max(maxPiece, q.bytes);
        if (staged) for (int k = 0; k < 2; k++) {
            if (hipHostMalloc((void **)&buf[k], maxPiece, hipHostMallocDefault) != hipSuccess) { buf[k] = nullptr; me.failed = 1; }
            if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { ev[k] = nullptr; me.failed = 1; }
        }
        for (int k = 0; !me.failed; k ^= 1) {
            const size_t i = next.fetch_add(1); if (i >= plan.pieces.size()) break;
            const ImagePiece &q = plan.pieces[i]; char *d = me.dst[q.part] + q.off; const char *sp = plan.src[q.part] + q.off;
            if (!staged) { if (hipMemcpyAsync(d, sp, q.bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) me.failed = 1; else me.done[i].store(1, std::memory_order_release); continue; }
            if (used[k]) { if (hipEventSynchronize(ev[k]) != hipSuccess) { me.failed = 1; break; } me.done[pend[k]].store(1, std::memory_order_release); used[k] = false; }
            memcpy(buf[k], sp, q.bytes);
            if (hipMemcpyAsync(d, buf[k], q.bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { me.failed = 1; break; }
            used[k] = true; pend[k] = i;
        }
        if (staged) { if (hipStreamSynchronize(st) != hipSuccess) me.failed = 1; for (int k = 0; k < 2; k++) { if (used[k] && !me.failed) me.done[pend[k]].store(1, std::memory_order_release); if (buf[k]) (void)hipHostFree(buf[k]); if (ev[k]) (void)hipEventDestroy(ev[k]); } }
        (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th; for (int t = 1; t < nt; t++) th.emplace_back(work);
    work(); for (auto &x : th) x.join();
    if (me.failed) (void)hipGetLastError();
}
// pieces from the neighbour's image, each as soon as the neighbour has it
static void copyFromPeer(int device, int srcDevice, const ImagePlan &plan, ImageState &me, ImageState &from)
{
    if (hipSetDevice(device) != hipSuccess) { me.failed = 1; return; }
    hipStream_t st; if (hipStreamCreate(&st) != hipSuccess) { me.failed = 1; return; }
    // (a window of copies in flight: the events of the last W pieces; a piece is published once its event has completed)
    const int W = 4; hipEvent_t ev[W]; size_t pend[W]; bool used[W]; for (int k = 0; k < W; k++) { used[k] = false; if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) me.failed = 1; }
    for (size_t i = 0; i < plan.pieces.size() && !me.failed; i++) {
        const int k = (int)(i % W);
        if (used[k]) { if (hipEventSynchronize(ev[k]) != hipSuccess) { me.failed = 1; break; } me.done[pend[k]].store(1, std::memory_order_release); used[k] = false; }
        while (!from.done[i].load(std::memory_order_acquire)) { if (from.failed) { me.failed = 1; break; } std::this_thread::yield(); }
        if (me.failed) break;
        const ImagePiece &q = plan.pieces[i];
        const hipError_t e = device == srcDevice ? hipMemcpyAsync(me.dst[q.part] + q.off, from.dst[q.part] + q.off, q.bytes, hipMemcpyDeviceToDevice, st)
                                                 : hipMemcpyPeerAsync(me.dst[q.part] + q.off, device, from.dst[q.part] + q.off, srcDevice, q.bytes, st);
        if (e != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { me.failed = 1; break; }
        used[k] = true; pend[k] = i;
    }
    if (hipStreamSynchronize(st) != hipSuccess) me.failed = 1;
    for (int k = 0; k < W; k++) { if (used[k] && !me.failed) me.done[pend[k]].store(1, std::memory_order_release); (void)hipEventDestroy(ev[k]); }
    (void)hipStreamDestroy(st);
    if (me.failed) (void)hipGetLastError();
}
static int checkParams(ygpu_ctx *ctx, const ygpu_index_view *ix, const ygpu_params *p)
{
    // supported parameter ranges of the wave-parallel DP (dp_wave.h)
    long big = 32000L * std::max(std::max(p->MScore, p->RCost), p->GECost) + p->GOCost + 128L * p->GECost;
    if (p->wordLen < 1 || p->wordLen > 15 || ix->wordLen != p->wordLen) { ctx->err = "wordLen must be 1..15 and match the index"; return YGPU_EINVAL; }
    // (any band the reference accepts, Main.c:324-327: an extension strip of 4 * BW + 1 <= 64 columns runs with a column per lane, a wider one goes through the
    // sequential recurrence of dp_wave.h with its scratch sized by alignDims below; the bound here only keeps that scratch within a few MB per wave)
    if (p->bandWidth < 0 || p->bandWidth > 255) { ctx->err = "bandWidth must be between 0 and 255"; return YGPU_EINVAL; }
    if (p->maxGap < 0 || p->maxGap > 16383 || p->maxIntron < 0 || p->maxHits < 0 || p->maxHits > 65525) { ctx->err = "maxGap/maxIntron/maxHits out of range"; return YGPU_EINVAL; }
    if (p->MScore < 0 || p->RCost < 0 || p->GECost < 0 || p->GOCost < 0 || big >= (1L << 23)) { ctx->err = "scoring parameters out of the supported range"; return YGPU_EINVAL; }
    DevParams &P = ctx->P;
    P.wordLen = p->wordLen; P.maxHits = p->maxHits; P.bandWidth = p->bandWidth; P.maxGap = p->maxGap; P.maxIntron = p->maxIntron; P.minMatch = p->minMatch; P.maxDesert = p->maxDesert;
    P.minNonOverlap = p->minNonOverlap; P.minRawScore = p->minRawScore; P.minExtLength = p->minExtLength & 0xFF; P.GO = p->GOCost; P.GE = p->GECost; P.RC = p->RCost; P.MS = p->MScore; P.X = p->XCutoff;
    P.minIdentity = p->minIdentity; P.maxROff = ix->maxROff; P.totalMatches = ix->totalMatches;
    return 0;
}
// one device of ygpu_init_multi: image memory, then the copy (its own thread) beside streams / events / code object, then the bit table of seed.h
static void shareImage(ygpu_ctx *ctx, const ygpu_ctx *parent)
{
    ctx->P = parent->P;
    ctx->dBases.p = parent->dBases.p; ctx->dBases.cap = parent->dBases.cap; ctx->dSO.p = parent->dSO.p; ctx->dSO.cap = parent->dSO.cap; ctx->dROA.p = parent->dROA.p; ctx->dROA.cap = parent->dROA.cap; ctx->dLow.p = parent->dLow.p; ctx->dLow.cap = parent->dLow.cap;
    ctx->sharedIndex = true;
}
static int initDevice(ygpu_ctx *ctx, int device, int srcIndex /* -1: the host */, int srcDevice, const ygpu_index_view *ix, const ImagePlan &plan, ImageState *states, int self, std::atomic<int> *imageReady, ygpu_ctx **more, int nMore)
{
    // the device's further contexts (they share this one's image): streams, events and counters are made beside the copy as well
    std::vector<int> moreRc(nMore, 0); std::vector<std::thread> moreTh;
    for (int j = 0; j < nMore; j++) moreTh.emplace_back([&, j]() { ygpu_ctx *c = more[j]; int rc = initCommon(c, device); if (rc == 0 && (c->counters.ensure(4 * CNT_N) || c->ctr.ensure(sizeof(DevCounters)) || c->errFlag.ensure(64))) { c->err = "hipMalloc failed"; rc = YGPU_ENOMEM; } if (rc == 0 && hipMemsetAsync(c->counters.p, 0, 4 * CNT_N, c->stream) != hipSuccess) { c->err = "hipMemset failed"; rc = YGPU_ENODEV; } moreRc[j] = rc; });
    struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{moreTh};
    const bool phases = getenv("YGPU_INIT_PHASES") != nullptr; double tPh = nowMs(); const double tPh0 = tPh;      // where a context's start-up goes
    auto phase = [&](const char *what) { if (phases) { const double t = nowMs(); fprintf(stderr, "[ygpu] init device %d: %-40s %8.1f ms\n", device, what, t - tPh); tPh = t; } };
    ImageState &me = states[self];
    auto giveUp = [&](int rc) { me.failed = 1; imageReady[self] = -1; return rc; };
    if (hipSetDevice(device) != hipSuccess) { ctx->err = "hipSetDevice failed"; (void)hipGetLastError(); return giveUp(YGPU_ENODEV); }
    const uint64_t HT = 1ull << (2 * ix->wordLen);
    if (ctx->dBases.ensureExact(ix->n_base_bytes + 4096) || ctx->dSO.ensureExact(4 * (HT + 1) + 64) || ctx->dROA.ensureExact(4ull * ix->totalMatches + 64)) {      /* exact: a growth margin on 16.7 GB is 4 GB */ ctx->err = "hipMalloc failed for the index image"; (void)hipGetLastError(); return giveUp(YGPU_ENOMEM); }
    me.dst[0] = (char *)ctx->dBases.p; me.dst[1] = (char *)ctx->dSO.p; me.dst[2] = (char *)ctx->dROA.p;
    imageReady[self] = 1;                                                    // the memory is there: the next device in the chain may start asking for pieces
    // (slack behind the bases reads as 0xEE: the lane kernels load whole dwords around a window)
    if (hipMemset((char *)ctx->dBases.p + ix->n_base_bytes, 0xEE, ctx->dBases.cap - ix->n_base_bytes) != hipSuccess) { ctx->err = "hipMemset failed"; (void)hipGetLastError(); return giveUp(YGPU_ENODEV); }
    phase("device memory for the image");
    // One thread, one plain copy per piece: the runtime pins the piece and the DMA engines read it in place -- 41-48 GB/s for the 16.7 GB index out of the page cache
    // (55 GB/s, the link's rate, for a file whose pages the kernel could keep in large folios; tools/micro/h2d_probe.hip).  Measured on the command line, 16.7 GB, runs
    // 6 s apart: one thread 404 ms, two 450-500, four 640; six threads staging through page-locked buffers of their own 590-630 (profiles/r04_index_upload.txt).
    bool staged = false; int nt = 1;                                         // YGPU_UPLOAD=direct:T | staged:T  (YGPU_UPLOAD_THREADS=T: the earlier spelling of direct:T)
    if (const char *e = getenv("YGPU_UPLOAD_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 32) nt = v; }
    if (const char *e = getenv("YGPU_UPLOAD")) { staged = strncmp(e, "staged", 6) == 0; const char *c = strchr(e, ':'); if (c) { const int v = atoi(c + 1); if (v >= 1 && v <= 32) nt = v; } }
    std::thread copier;
    if (srcIndex < 0) copier = std::thread([&, nt, staged]() { uploadFromHost(device, plan, me, nt, staged); });
    else copier = std::thread([&]() { while (imageReady[srcIndex].load() == 0) std::this_thread::yield(); if (imageReady[srcIndex].load() < 0) { me.failed = 1; return; } copyFromPeer(device, srcDevice, plan, me, states[srcIndex]); });
    int rc0 = initCommon(ctx, device);
    if (rc0 == 0) { if (ctx->counters.ensure(4 * CNT_N) || ctx->ctr.ensure(sizeof(DevCounters)) || ctx->errFlag.ensure(64) || ctx->dLow.ensure(YD_LOW_BITS / 8)) { ctx->err = "hipMalloc failed"; rc0 = YGPU_ENOMEM; } }
    if (rc0 == 0) { hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, ctx->stream, (unsigned int *)nullptr); if (hipMemsetAsync(ctx->counters.p, 0, 4 * CNT_N, ctx->stream) != hipSuccess || hipMemsetAsync(ctx->dLow.p, 0, YD_LOW_BITS / 8, ctx->stream) != hipSuccess || streamSync(ctx) != hipSuccess) { ctx->err = "first kernel launch failed"; (void)hipGetLastError(); rc0 = YGPU_ENODEV; } }
    phase("streams, events, code object (beside the copy)");
    copier.join();
    if (rc0) { me.failed = 1; return rc0; }
    if (me.failed) { ctx->err = srcIndex < 0 ? "copying the index image to the device failed" : "copying the index image from the neighbouring device failed"; return YGPU_ENODEV; }
    phase(srcIndex < 0 ? "image copied from the host" : "image copied from the device before");
    if (ix->totalMatches) KL(k_low_offsets, dim3((unsigned)std::min<uint64_t>(gridFor(ix->totalMatches, 256), (uint64_t)ctx->nCU * 64)), dim3(256), 0, ctx->stream, ctx->dSO.as<uint32_t>(), (uint32_t)HT, ctx->dROA.as<uint32_t>(), (uint32_t)ix->totalMatches, ctx->dLow.as<uint32_t>());
    HIPCHK(streamSync(ctx));
    phase("low-offset bit table");
    for (auto &x : moreTh) x.join();
    for (int j = 0; j < nMore; j++) { if (moreRc[j]) { ctx->err = "a further context of the device failed: " + more[j]->err; return moreRc[j]; } shareImage(more[j], ctx); }
    if (phases) fprintf(stderr, "[ygpu] init device %d: total %.1f ms (%d contexts)\n", device, nowMs() - tPh0, 1 + nMore);
    return 0;
}
}  // namespace

extern "C" {

int ygpu_init_multi(const int *devices, int n, int ctx_per_device, const ygpu_index_view *ix, const ygpu_params *p, ygpu_ctx **all, int *rc_each)
{
    if (!all || n < 1 || n > 64 || !devices || ctx_per_device < 1 || ctx_per_device > 16) return YGPU_EINVAL;
    const int cpd = ctx_per_device;
    for (int k = 0; k < n * cpd; k++) all[k] = nullptr;
    for (int k = 0; k < n; k++) if (rc_each) rc_each[k] = YGPU_EINVAL;
    if (!ix || !p) return YGPU_EINVAL;
    for (int k = 0; k < n * cpd; k++) { all[k] = new ygpu_ctx; all[k]->device = devices[k / cpd]; }
    std::vector<ygpu_ctx *> out(n); for (int k = 0; k < n; k++) out[k] = all[k * cpd];      // every device's first context: the one that owns its image
    std::vector<int> rcs(n, 0);
    // a failure before anything was started: the devices it is about say why, the others that they were not started
    auto notStarted = [&](int rc) { for (int k = 0; k < n; k++) { if (!rcs[k]) { rcs[k] = YGPU_EINVAL; out[k]->err = "not started: another device of the call failed"; } if (rc_each) rc_each[k] = rcs[k]; for (int j = 1; j < cpd; j++) all[k * cpd + j]->err = out[k]->err; } return rc; };
    const double t0 = nowMs();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { for (int k = 0; k < n; k++) { rcs[k] = YGPU_ENODEV; out[k]->err = "no HIP device visible: the hot path needs an MI355X (there is no CPU fallback)"; } return notStarted(YGPU_ENODEV); }
    if (getenv("YGPU_INIT_PHASES")) fprintf(stderr, "[ygpu] init: runtime up (hipGetDeviceCount) %8.1f ms\n", nowMs() - t0);
    { bool bad = false; for (int k = 0; k < n; k++) if (devices[k] < 0 || devices[k] >= ndev) { rcs[k] = YGPU_ENODEV; out[k]->err = "device index out of range"; bad = true; } if (bad) return notStarted(YGPU_ENODEV); }
    { int bad = 0; for (int k = 0; k < n; k++) { rcs[k] = checkParams(out[k], ix, p); if (rcs[k]) bad = rcs[k]; } if (bad) return notStarted(bad); }
    ImagePlan plan; plan.build(ix, n > 1 ? (64ull << 20) : (1024ull << 20));
    std::vector<ImageState> states(n); for (auto &st : states) { st.done.reset(new std::atomic<int>[plan.pieces.size() + 1]); for (size_t i = 0; i <= plan.pieces.size(); i++) st.done[i] = 0; }
    std::unique_ptr<std::atomic<int>[]> ready(new std::atomic<int>[n]); for (int k = 0; k < n; k++) ready[k] = 0;
    // the chain: device k takes the image from device k - 1 when it can reach it (YGPU_PEER_COPY=0: every device from the host)
    std::vector<int> srcIndex(n, -1); const bool peer = !(getenv("YGPU_PEER_COPY") && atoi(getenv("YGPU_PEER_COPY")) == 0);
    for (int k = 1; k < n && peer; k++) {
        int can = devices[k] == devices[k - 1] ? 1 : 0;
        if (!can && hipDeviceCanAccessPeer(&can, devices[k], devices[k - 1]) != hipSuccess) { can = 0; (void)hipGetLastError(); }
        if (can && devices[k] != devices[k - 1]) { if (hipSetDevice(devices[k]) == hipSuccess) { const hipError_t e = hipDeviceEnablePeerAccess(devices[k - 1], 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0; } else can = 0; (void)hipGetLastError(); }
        if (can) srcIndex[k] = k - 1;
    }
    if (getenv("YGPU_INIT_PHASES")) { fprintf(stderr, "[ygpu] init: image sources:"); for (int k = 0; k < n; k++) { if (srcIndex[k] < 0) fprintf(stderr, " device %d <- host;", devices[k]); else fprintf(stderr, " device %d <- device %d;", devices[k], devices[srcIndex[k]]); } fprintf(stderr, " %zu pieces\n", plan.pieces.size()); }
    std::vector<std::thread> th;
    for (int k = 1; k < n; k++) th.emplace_back([&, k]() { rcs[k] = initDevice(out[k], devices[k], srcIndex[k], srcIndex[k] >= 0 ? devices[srcIndex[k]] : -1, ix, plan, states.data(), k, ready.get(), all + k * cpd + 1, cpd - 1); });
    rcs[0] = initDevice(out[0], devices[0], -1, -1, ix, plan, states.data(), 0, ready.get(), all + 1, cpd - 1);
    for (auto &x : th) x.join();
    int rc = 0; for (int k = 0; k < n; k++) { if (rc_each) rc_each[k] = rcs[k]; if (rcs[k] && !rc) rc = rcs[k]; if (rcs[k]) for (int j = 1; j < cpd; j++) if (all[k * cpd + j]->err.empty()) all[k * cpd + j]->err = "the device's first context failed: " + out[k]->err; }
    return rc;
}

int ygpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

/* Create a context on HIP device `device`: the index image goes from the host to that device. */
int ygpu_init(int device, const ygpu_index_view *ix, const ygpu_params *p, ygpu_ctx **out)
{
    if (!out) return YGPU_EINVAL;
    return ygpu_init_multi(&device, 1, 1, ix, p, out, nullptr);
}

/* A second context on the same device that shares the parent's index image in HBM (nothing is uploaded again).  Two contexts on
 * one GPU, each driven by its own host thread with its own batches, keep the device busy while one of them is in a latency-bound
 * stage or waiting for its host.  The parent must outlive its clones. */
int ygpu_clone(const ygpu_ctx *parent, ygpu_ctx **out)
{
    *out = nullptr;
    if (!parent || !parent->stream) return YGPU_EINVAL;
    ygpu_ctx *ctx = new ygpu_ctx; *out = ctx; ctx->device = parent->device;
    int rc = initCommon(ctx, parent->device); if (rc) return rc;
    shareImage(ctx, parent);
    ENSURE(ctx->counters, 4 * CNT_N); ENSURE(ctx->ctr, sizeof(DevCounters)); ENSURE(ctx->errFlag, 64);
    HIPCHK(hipMemsetAsync(ctx->counters.p, 0, 4 * CNT_N, ctx->stream));
    HIPCHK(streamSync(ctx));
    return 0;
}

}  // extern "C"
static std::vector<DevBuf *> allBuffers(ygpu_ctx *ctx)
{
    DevBuf *all[] = {&ctx->dBases, &ctx->dSO, &ctx->dROA, &ctx->dLow, &ctx->dFwd, &ctx->dRev, &ctx->dFwd4, &ctx->dRev4, &ctx->dReadOff, &ctx->dKmerOff, &ctx->posS, &ctx->posC, &ctx->posRsI, &ctx->hitOff, &ctx->expandStart, &ctx->keysA, &ctx->keysB, &ctx->segOff, &ctx->bigB, &ctx->bigE, &ctx->isHead, &ctx->tileState,
                         &ctx->frags, &ctx->regStart, &ctx->multiList, &ctx->smallList, &ctx->bigList, &ctx->regionCount, &ctx->regionBase, &ctx->clumps, &ctx->clumpFrags, &ctx->clumpFrags0, &ctx->order, &ctx->rootPush, &ctx->rootBase, &ctx->outClumps,
                         &ctx->outClumps2, &ctx->outOps, &ctx->outRoot, &ctx->outPush, &ctx->dstIdx, &ctx->readCount, &ctx->readStart, &ctx->counters, &ctx->ctr, &ctx->errFlag, &ctx->cubTemp, &ctx->scanState, &ctx->bucketWork, &ctx->scratchAlign,
                         &ctx->segLists, &ctx->subB, &ctx->subE, &ctx->subLists, &ctx->subBigB, &ctx->subBigE, &ctx->sub2B, &ctx->sub2E, &ctx->sub2Lists, &ctx->sub3B, &ctx->sub3E, &ctx->sub3Lists, &ctx->kmerParts, &ctx->scratchChain, &ctx->dpProbs, &ctx->dpRes, &ctx->dpOps, &ctx->rootState, &ctx->stateOps, &ctx->extProbs, &ctx->rowsBound, &ctx->stripOff, &ctx->extRes, &ctx->extTrace, &ctx->chunkCnt, &ctx->cubTemp2, &ctx->memoKeys, &ctx->memoCount, &ctx->probs2, &ctx->rowsBound2, &ctx->stripOff2, &ctx->extRes2, &ctx->extTrace2, &ctx->rowsClock, &ctx->splitScratch, &ctx->fallList, &ctx->keys2a, &ctx->keys2b, &ctx->vals2a, &ctx->vals2b, &ctx->extKeys, &ctx->extVals, &ctx->extKeys2, &ctx->extOrder, &ctx->slowList, &ctx->gapScratch, &ctx->jointCount, &ctx->jointBase, &ctx->joints, &ctx->sortKeys, &ctx->sortVals, &ctx->sortKeys2, &ctx->sortVals2, &ctx->gapOps, &ctx->waveChunks, &ctx->extOps, &ctx->traceCnt,
                         &ctx->oqCs, &ctx->oqCl, &ctx->oqOpsIn, &ctx->oqSeeds, &ctx->oqQlen, &ctx->pf.scanState, &ctx->pf.counters, &ctx->oqProf, &ctx->oqLists, &ctx->oqClsCnt, &ctx->oqThr, &ctx->oqSeqStart, &ctx->oqSeqLen, &ctx->oqNeed, &ctx->oqPoolOff, &ctx->oqKeys, &ctx->oqStack, &ctx->oqNodes, &ctx->oqPrim, &ctx->oqPA, &ctx->oqPfx, &ctx->oqPath, &ctx->oqPool, &ctx->oqPush, &ctx->oqOut, &ctx->oqOutCnt, &ctx->oqOutOps, &ctx->oqPrimCnt, &ctx->oqOutStart, &ctx->oqOpsStart, &ctx->oqFClumps, &ctx->oqFOps};
    return std::vector<DevBuf *>(all, all + sizeof all / sizeof all[0]);
}
extern "C" {

void ygpu_destroy(ygpu_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->worker.joinable()) { { std::lock_guard<std::mutex> lk(ctx->aMu); ctx->aQuit = true; } ctx->aCv.notify_all(); ctx->worker.join(); }
    if (ctx->counted) gCtxPerDevice[ctx->device & 63]--;
    if (ctx->stream) {
        hipSetDevice(ctx->device);
        if (ctx->sharedIndex) { ctx->dBases.p = nullptr; ctx->dBases.cap = 0; ctx->dSO.p = nullptr; ctx->dSO.cap = 0; ctx->dROA.p = nullptr; ctx->dROA.cap = 0; ctx->dLow.p = nullptr; ctx->dLow.cap = 0; }
        const std::vector<DevBuf *> all = allBuffers(ctx);
        for (auto b : all) b->release();
        for (int t = 0; t < T_N; t++) { hipEventDestroy(ctx->ev[t][0]); hipEventDestroy(ctx->ev[t][1]); }
        if (ctx->evSync) hipEventDestroy(ctx->evSync);
        if (ctx->pinned) hipHostFree(ctx->pinned);
        if (ctx->pf.pinned) hipHostFree(ctx->pf.pinned);
        if (ctx->snapCtr) hipHostFree(ctx->snapCtr);
        if (ctx->pf.evSync) hipEventDestroy(ctx->pf.evSync);
        if (ctx->evSnap) hipEventDestroy(ctx->evSnap);
        for (int i = 0; i < YD_MAX_CHUNK_EV; i++) hipEventDestroy(ctx->evChunk[i]);
        hipEventDestroy(ctx->evTail); hipStreamDestroy(ctx->stream2);
        hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}
/* Device memory: free and total bytes of the context's device, and what this context's own buffers hold (a shared index image counts for the context that owns it). */
int ygpu_memory(ygpu_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes, uint64_t *ctx_bytes)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    size_t fb = 0, tb = 0; HIPCHK(hipMemGetInfo(&fb, &tb));
    uint64_t mine = 0; for (DevBuf *b : allBuffers(ctx)) if (b->p && !(ctx->sharedIndex && (b == &ctx->dBases || b == &ctx->dSO || b == &ctx->dROA || b == &ctx->dLow))) mine += b->cap;
    if (free_bytes) *free_bytes = fb; if (total_bytes) *total_bytes = tb; if (ctx_bytes) *ctx_bytes = mine;
    return 0;
}
/* A context the host decides not to use (no room for its arenas beside the others): its own buffers are released and it no longer counts among the contexts that
 * share the device's memory budget (the trace arena's fair share, the wave kernels' scratch).  The index image stays (other contexts may share it); the context
 * can only be destroyed afterwards. */
int ygpu_park(ygpu_ctx *ctx)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device)); HIPCHK(streamSync(ctx));
    for (DevBuf *b : allBuffers(ctx)) if (b != &ctx->dBases && b != &ctx->dSO && b != &ctx->dROA && b != &ctx->dLow) b->release();
    if (ctx->counted) { gCtxPerDevice[ctx->device & 63]--; ctx->counted = false; }
    ctx->stageDone = 0; ctx->parked = true;
    return 0;
}
/* What a context's arenas hold after a batch, and the estimates it carries from batch to batch -- so that the device's other contexts can be given the same
 * capacities in one go (ygpu_presize) instead of growing theirs buffer by buffer during a first batch of their own. */
int ygpu_get_arena_profile(ygpu_ctx *ctx, ygpu_arena_profile *out)
{
    if (!ctx || !ctx->stream || !out) return YGPU_EINVAL;
    memset(out, 0, sizeof *out);
    const std::vector<DevBuf *> all = allBuffers(ctx);
    if (all.size() > sizeof out->cap / sizeof out->cap[0]) { ctx->err = "arena profile: more buffers than the profile holds"; return YGPU_EINTERNAL; }
    out->n = (uint32_t)all.size();
    for (size_t k = 0; k < all.size(); k++) { DevBuf *b = all[k]; out->cap[k] = (b == &ctx->dBases || b == &ctx->dSO || b == &ctx->dROA || b == &ctx->dLow) ? 0ull : (uint64_t)b->cap; }
    out->trace_ratio = ctx->traceRatio; out->ops_ratio = ctx->opsRatio; out->last_clump_slots = ctx->lastClumpSlots; out->last_fall = ctx->lastFall; out->bases = ctx->totalBases;
    return 0;
}
int ygpu_presize(ygpu_ctx *ctx, const ygpu_arena_profile *prof)
{
    if (!ctx || !ctx->stream || !prof) return YGPU_EINVAL;
    if (ctx->parked) { ctx->err = "the context was parked (ygpu_park)"; return YGPU_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    const std::vector<DevBuf *> all = allBuffers(ctx);
    if (prof->n != all.size()) { ctx->err = "arena profile of another build"; return YGPU_EINVAL; }
    for (size_t k = 0; k < all.size(); k++) {
        DevBuf *b = all[k];
        if (b == &ctx->dBases || b == &ctx->dSO || b == &ctx->dROA || b == &ctx->dLow || b == &ctx->counters || b == &ctx->ctr || b == &ctx->errFlag || b == &ctx->pf.counters) continue;
        if (prof->cap[k] > b->cap && b->ensureExact((size_t)prof->cap[k])) { (void)hipGetLastError(); ctx->err = "hipMalloc failed while presizing the arenas"; return YGPU_ENOMEM; }
    }
    // (work words that their kernels expect zeroed when they are made)
    if (ctx->scanState.p) HIPCHK(hipMemsetAsync(ctx->scanState.p, 0, ctx->scanState.cap, ctx->stream));
    if (ctx->bucketWork.p) HIPCHK(hipMemsetAsync(ctx->bucketWork.p, 0, ctx->bucketWork.cap, ctx->stream));
    if (ctx->pf.scanState.p) HIPCHK(hipMemsetAsync(ctx->pf.scanState.p, 0, ctx->pf.scanState.cap, ctx->stream));      // (the post-filter side's look-back words: the same rule)
    if (ctx->runsDone == 0) { ctx->traceRatio = prof->trace_ratio; ctx->opsRatio = prof->ops_ratio > 0 ? prof->ops_ratio : ctx->opsRatio; ctx->lastClumpSlots = prof->last_clump_slots; ctx->lastFall = (long long)prof->last_fall; }
    HIPCHK(streamSync(ctx));
    return 0;
}
static thread_local const ygpu_ctx *tlsPfFailed = nullptr;                   // the context whose post-filter side failed last on this thread: ygpu_last_error then reports that side's message
const char *ygpu_last_error(const ygpu_ctx *ctx) { return !ctx ? "null context" : (tlsPfFailed == ctx && !ctx->pf.err.empty()) ? ctx->pf.err.c_str() : ctx->err.c_str(); }

} // extern "C"
static int uploadBatch(ygpu_ctx *ctx, const ygpu_read_batch *b, bool wait)
{
    if (!ctx || !ctx->stream || !b) return YGPU_EINVAL;
    if (ctx->parked) { ctx->err = "the context was parked (ygpu_park)"; return YGPU_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = b->n_reads;
    if (n > 65536) { ctx->err = "at most 65536 reads per batch"; return YGPU_EINVAL; }
    ctx->nReads = n; ctx->stageDone = 0; ctx->hReadOff.assign(n + 1, 0); ctx->hKmerOff.assign(2 * n + 1, 0); ctx->maxQ = 0;
    const uint64_t base0 = n ? b->offsets[0] : 0;
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t len = b->offsets[i + 1] - b->offsets[i];
        if (len > 32000) { ctx->err = "read longer than 32000 bases"; return YGPU_EINVAL; }
        ctx->hReadOff[i + 1] = (uint32_t)(b->offsets[i + 1] - base0);
        ctx->maxQ = std::max(ctx->maxQ, (int)len);
        uint32_t np = len >= (uint64_t)ctx->P.wordLen ? (uint32_t)(len - ctx->P.wordLen + 1) : 0;
        ctx->hKmerOff[2 * i] = k; k += np; ctx->hKmerOff[2 * i + 1] = k; k += np;
    }
    ctx->hKmerOff[2 * n] = k; ctx->nKmers = k; ctx->totalBases = n ? b->offsets[n] - base0 : 0;
    if (ctx->totalBases > 0x7FFFFFF0ull) { ctx->err = "batch larger than 2 Gbases"; return YGPU_EINVAL; }
    // (a snapshot taken without a wait may still be reading the previous batch's codes and offsets on this stream: before any of these buffers is replaced by a
    // larger one, the stream is drained -- hipFree waits for the device by itself, this does not rely on it)
    if (ctx->dFwd.cap < ctx->totalBases + 256 || ctx->dReadOff.cap < 4ull * (n + 2)) HIPCHK(streamSync(ctx));
    ENSURE(ctx->dFwd, ctx->totalBases + 256); ENSURE(ctx->dRev, ctx->totalBases + 256); ENSURE(ctx->dFwd4, ctx->totalBases / 2 + 256); ENSURE(ctx->dRev4, ctx->totalBases / 2 + 256);   /* slack: lane kernels read whole dwords around a segment */ ENSURE(ctx->dReadOff, 4ull * (n + 1)); ENSURE(ctx->dKmerOff, 4ull * (2 * n + 1));
    if (n) {
        HIPCHK(hipMemcpyAsync(ctx->dFwd.p, b->codes + base0, ctx->totalBases, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->dReadOff.p, ctx->hReadOff.data(), 4ull * (n + 1), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->dKmerOff.p, ctx->hKmerOff.data(), 4ull * (2 * n + 1), hipMemcpyHostToDevice, ctx->stream));
        KL(k_revcomp, dim3(n), dim3(256), 0, ctx->stream, ctx->dFwd.as<uint8_t>(), ctx->dRev.as<uint8_t>(), ctx->dReadOff.as<uint32_t>(), n);
        // (both strands packed two codes to the byte for the X-drop kernel's query windows; the byte arrays have 256 bytes of slack behind the last code)
        const uint32_t nPacked = (uint32_t)((ctx->totalBases + 1) / 2);
        KL(k_pack4, dim3(gridFor(nPacked, 256)), dim3(256), 0, ctx->stream, ctx->dFwd.as<uint8_t>(), ctx->dFwd4.as<uint8_t>(), nPacked, (uint32_t)ctx->totalBases);
        KL(k_pack4, dim3(gridFor(nPacked, 256)), dim3(256), 0, ctx->stream, ctx->dRev.as<uint8_t>(), ctx->dRev4.as<uint8_t>(), nPacked, (uint32_t)ctx->totalBases);
    }
    if (wait) HIPCHK(streamSync(ctx));
    return 0;
}
extern "C" {
int ygpu_upload(ygpu_ctx *ctx, const ygpu_read_batch *b) { return uploadBatch(ctx, b, true); }
/* ygpu_upload without its wait: returns as soon as the copies are queued.  The batch's memory must stay unchanged until the ygpu_run that follows has returned. */
int ygpu_upload_nowait(ygpu_ctx *ctx, const ygpu_read_batch *b) { static const bool waitAnyway = getenv("YGPU_UPLOAD_WAIT") != nullptr; return uploadBatch(ctx, b, waitAnyway); }

int ygpu_run(ygpu_ctx *ctx)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    ctx->stageDone = 0; const double t0 = nowMs(); ctx->statAttempts = 0; ctx->statRanges = 0;
    struct Active { int d; explicit Active(int dv) : d(dv) { gActiveRuns[d]++; } ~Active() { gActiveRuns[d]--; } } active(ctx->device & 63);
    int rc = runTo(ctx, 3);
    if (kStats) { size_t fb = 0, tb = 0; hipMemGetInfo(&fb, &tb); fprintf(stderr, "[ygpu] ctx %p run: %u reads, rc %d, %.1f ms; align attempts %d, ranges %d, trace arena %.2f GB (ratio %.3f), free %.1f GB\n", (void *)ctx, ctx->nReads, rc, nowMs() - t0, ctx->statAttempts, ctx->statRanges, ctx->extTrace.cap / 1e9, ctx->traceRatio, fb / 1e9); }
    if (rc) return rc;
    ctx->runsDone++;
    ctx->totalMs = 0;
    for (int t = 0; t < T_N; t++) {
        float m = 0; ctx->ms[t] = (ctx->evUsed[t] && hipEventElapsedTime(&m, ctx->ev[t][0], ctx->ev[t][1]) == hipSuccess) ? m : 0;
        if (t == T_XROWS_DEV) ctx->ms[t] = (ctx->evUsed[T_XROWS] && ctx->hRowsClock[1] > ctx->hRowsClock[0] && ctx->hRowsClock[0] != ~0ull) ? (float)((double)(ctx->hRowsClock[1] - ctx->hRowsClock[0]) / 1.0e5) : 0;   // 100 MHz ticks -> ms
        if (t == T_XROWS_PK) ctx->ms[t] = ctx->rowsPacked ? 1.0f : 0.0f;
        if (t < T_TOP) ctx->totalMs += ctx->ms[t];
    }
    return 0;
}

int ygpu_collect(ygpu_ctx *ctx, ygpu_result_batch *out)
{
    if (!ctx || !out || ctx->stageDone < 3) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->nReads;
    ctx->hClumpStart.assign(n + 1, 0); ctx->hClumps.resize(ctx->nOut); ctx->hOps.resize(ctx->nOutOps);
    HIPCHK(hipMemcpyAsync(ctx->hClumpStart.data(), ctx->readStart.p, 4ull * (n + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->nOut) HIPCHK(hipMemcpyAsync(ctx->hClumps.data(), ctx->outClumps2.p, sizeof(ygpu_clump) * (uint64_t)ctx->nOut, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->nOutOps) HIPCHK(hipMemcpyAsync(ctx->hOps.data(), ctx->outOps.p, 4ull * ctx->nOutOps, hipMemcpyDeviceToHost, ctx->stream));
    DevCounters dc; HIPCHK(hipMemcpyAsync(&dc, ctx->ctr.p, sizeof dc, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(streamSync(ctx));
    { const unsigned long long dropped = dc.v[C_FRAGS];                      // dead single-hit fragments (each one a region of its own) that were counted, not written
      dc.v[C_HITS] = ctx->nHits; dc.v[C_FRAGS] = ctx->nFrags + dropped; dc.v[C_REGIONS] = ctx->nRegions + dropped; }
    memcpy(&ctx->hCounters, dc.v, sizeof(ygpu_counters));
    out->n_reads = n; out->clump_start = ctx->hClumpStart.data(); out->clumps = ctx->hClumps.data(); out->ops = ctx->hOps.data();
    out->n_clumps = ctx->nOut; out->n_ops = ctx->nOutOps; out->counters = ctx->hCounters;
    return 0;
}

int ygpu_result_size(ygpu_ctx *ctx, uint64_t *n_clumps, uint64_t *n_ops)
{
    if (!ctx || ctx->stageDone < 3) return YGPU_EINVAL;
    if (n_clumps) *n_clumps = ctx->nOut; if (n_ops) *n_ops = ctx->nOutOps;
    return 0;
}
int ygpu_collect_into(ygpu_ctx *ctx, uint32_t *clump_start, ygpu_clump *clumps, uint32_t *ops, ygpu_result_batch *out)
{
    if (!ctx || !out || !clump_start || ctx->stageDone < 3 || (ctx->nOut && !clumps) || (ctx->nOutOps && !ops)) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->nReads;
    HIPCHK(hipMemcpyAsync(clump_start, ctx->readStart.p, 4ull * (n + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->nOut) HIPCHK(hipMemcpyAsync(clumps, ctx->outClumps2.p, sizeof(ygpu_clump) * (uint64_t)ctx->nOut, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->nOutOps) HIPCHK(hipMemcpyAsync(ops, ctx->outOps.p, 4ull * ctx->nOutOps, hipMemcpyDeviceToHost, ctx->stream));
    DevCounters dc; HIPCHK(hipMemcpyAsync(&dc, ctx->ctr.p, sizeof dc, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(streamSync(ctx));
    { const unsigned long long dropped = dc.v[C_FRAGS];
      dc.v[C_HITS] = ctx->nHits; dc.v[C_FRAGS] = ctx->nFrags + dropped; dc.v[C_REGIONS] = ctx->nRegions + dropped; }
    memcpy(&ctx->hCounters, dc.v, sizeof(ygpu_counters));
    out->n_reads = n; out->clump_start = clump_start; out->clumps = clumps; out->ops = ops;
    out->n_clumps = ctx->nOut; out->n_ops = ctx->nOutOps; out->counters = ctx->hCounters;
    return 0;
}
void *ygpu_host_alloc(size_t bytes) { void *p = nullptr; if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return p; }
void ygpu_host_free(void *p) { if (p) (void)hipHostFree(p); }

// ---- post-filter on the device (oqc_stage.h; reference GraphPath.cpp:897-1086) ----------------------------------------------------------------------
int ygpu_set_postfilter(ygpu_ctx *ctx, const ygpu_postfilter_params *p)
{
    if (!ctx || !ctx->stream || !p) return YGPU_EINVAL;
    if (p->bppN < 0 || p->bppN > 65536 || (p->bppN && !p->bppThr) || (p->n_seqs && (!p->seq_start || !p->seq_length))) { ctx->err = "ygpu_set_postfilter: bad break point table or sequence table"; return YGPU_EINVAL; }
    // (the wave's successor relaxation writes node j > i only while it reads node i: with a non-overlap requirement below one base a node could be its own successor, oqc_stage.h)
    if (p->minNonOverlap < 1) { ctx->err = "ygpu_set_postfilter: minNonOverlap (-MNO) must be at least 1 for the device stage; use the host filter"; return YGPU_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    ENSURE(ctx->oqThr, 4ull * (p->bppN + 1)); ENSURE(ctx->oqSeqStart, 4ull * (p->n_seqs + 1)); ENSURE(ctx->oqSeqLen, 4ull * (p->n_seqs + 1));
    if (p->bppN) HIPCHK(hipMemcpyAsync(ctx->oqThr.p, p->bppThr, 4ull * p->bppN, hipMemcpyHostToDevice, ctx->stream));
    if (p->n_seqs) { HIPCHK(hipMemcpyAsync(ctx->oqSeqStart.p, p->seq_start, 4ull * p->n_seqs, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipMemcpyAsync(ctx->oqSeqLen.p, p->seq_length, 4ull * p->n_seqs, hipMemcpyHostToDevice, ctx->stream)); }
    HIPCHK(streamSync(ctx));
    yoqc::Params &P = ctx->oqP;
    P.GOCost = ctx->P.GO; P.GECost = ctx->P.GE; P.RCost = ctx->P.RC; P.MScore = ctx->P.MS;
    P.minNonOverlap = p->minNonOverlap; P.BPCost = p->BPCost; P.maxBPLog = p->maxBPLog; P.FBS = p->FBS; P.FBS_PSLength = p->FBS_PSLength; P.FBS_PSScore = p->FBS_PSScore;
    P.bppVmin = p->bppVmin; P.bppN = p->bppN; P.bppThr = ctx->oqThr.as<uint32_t>();
    ctx->oqG.start = ctx->oqSeqStart.as<uint32_t>(); ctx->oqG.length = ctx->oqSeqLen.as<uint32_t>(); ctx->oqG.n = p->n_seqs;
    ctx->oqSet = true; return 0;
}
/* The stage works on a SNAPSHOT of the batch's results -- clump lists, edit ops, the reads' lengths and generator seeds, the work counters: 150 MB copied inside
 * the device in ~0.1 ms -- so that the context can take its next batch (ygpu_upload, ygpu_run) while another thread filters this one: ygpu_postfilter_snapshot on
 * the context's thread after ygpu_run, then ygpu_postfilter / ygpu_filtered_size / ygpu_collect_filtered on any thread.  (A ygpu_postfilter without a snapshot
 * takes one itself: the sequential use.)  One snapshot at a time: the next may be taken once the filtered results of this one have been collected. */
int ygpu_postfilter_snapshot(ygpu_ctx *ctx)
{
    if (!ctx || !ctx->stream || ctx->stageDone < 3) return YGPU_EINVAL;
    if (!ctx->oqSet) { ctx->err = "ygpu_postfilter_snapshot: ygpu_set_postfilter has not been called on this context"; return YGPU_EINVAL; }
    if (ctx->pfSnap.load()) { ctx->err = "ygpu_postfilter_snapshot: the previous snapshot has not been filtered yet"; return YGPU_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->nReads, C = ctx->nOut, O = ctx->nOutOps;
    ENSURE(ctx->oqCs, 4ull * (n + 2)); ENSURE(ctx->oqCl, sizeof(ygpu_clump) * ((uint64_t)C + 1)); ENSURE(ctx->oqOpsIn, 4ull * ((uint64_t)O + 1)); ENSURE(ctx->oqSeeds, 20ull * (n + 1)); ENSURE(ctx->oqQlen, 4ull * (n + 1));
    HIPCHK(hipMemcpyAsync(ctx->oqCs.p, ctx->readStart.p, 4ull * (n + 1), hipMemcpyDeviceToDevice, ctx->stream));
    if (C) HIPCHK(hipMemcpyAsync(ctx->oqCl.p, ctx->outClumps2.p, sizeof(ygpu_clump) * (uint64_t)C, hipMemcpyDeviceToDevice, ctx->stream));
    if (O) HIPCHK(hipMemcpyAsync(ctx->oqOpsIn.p, ctx->outOps.p, 4ull * O, hipMemcpyDeviceToDevice, ctx->stream));
    if (n) KL(k_oqc_seeds, dim3(gridFor(n, 256)), dim3(256), 0, ctx->stream, ctx->dFwd.as<uint8_t>(), ctx->dReadOff.as<uint32_t>(), n, ctx->oqSeeds.as<uint32_t>(), ctx->oqQlen.as<uint32_t>());
    // (no wait here: the context's thread goes straight on to its next batch -- whatever it queues on this stream follows the copies -- and the work counters land
    // in a pinned slot the filter's side reads after its own first wait; without the slot, a wait it is)
    if (ctx->snapCtr) HIPCHK(hipMemcpyAsync(ctx->snapCtr, ctx->ctr.p, sizeof(DevCounters), hipMemcpyDeviceToHost, ctx->stream));
    else { DevCounters dc; HIPCHK(hipMemcpyAsync(&dc, ctx->ctr.p, sizeof dc, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx)); memcpy(&ctx->snapCtrPlain, &dc, sizeof dc); }
    HIPCHK(hipEventRecord(ctx->evSnap, ctx->stream));
    { static const bool waitHere = getenv("YGPU_SNAPSHOT_WAIT") != nullptr; if (waitHere) HIPCHK(streamSync(ctx)); }
    ctx->snapHits = ctx->nHits; ctx->snapFrags = ctx->nFrags; ctx->snapRegions = ctx->nRegions;
    ctx->snapN = n; ctx->snapC = C; ctx->snapOps = O; ctx->oqDone = false;
    ctx->pfSnap.store(true);
    return 0;
}
static int postfilterBody(ygpu_ctx *full);
int ygpu_postfilter(ygpu_ctx *full)
{
    if (!full || !full->stream) return YGPU_EINVAL;
    if (!full->pfSnap.load()) { const int rc = ygpu_postfilter_snapshot(full); if (rc) { tlsPfFailed = nullptr; return rc; } }      // (the snapshot's message is the context's own)
    const int rc = postfilterBody(full);
    full->pfSnap.store(false);
    tlsPfFailed = rc ? full : nullptr;
    return rc;
}
static int postfilterBody(ygpu_ctx *full)
{
    PfSide *ctx = &full->pf;                                                 // (every macro and helper below: the post-filter's side)
    HIPCHK(hipSetDevice(full->device));
    HIPCHK(hipStreamWaitEvent(ctx->stream, full->evSnap, 0));
    const uint32_t n = full->snapN, C = full->snapC; full->pfN = n; full->nFOut = full->nFOps = 0; full->oqDone = false;
    auto takeCounters = [&]() {                                              // (after a wait of this side's stream: the snapshot's copies are done)
        DevCounters dc = full->snapCtr ? *full->snapCtr : full->snapCtrPlain;
        const unsigned long long dropped = dc.v[C_FRAGS]; dc.v[C_HITS] = full->snapHits; dc.v[C_FRAGS] = full->snapFrags + dropped; dc.v[C_REGIONS] = full->snapRegions + dropped;
        memcpy(&full->pfCounters, dc.v, sizeof(ygpu_counters));
    };
    ENSURE(full->oqOutStart, 4ull * (n + 2)); ENSURE(full->oqOpsStart, 4ull * (n + 2));
    if (n == 0 || C == 0) { HIPCHK(hipMemsetAsync(full->oqOutStart.p, 0, 4ull * (n + 2), ctx->stream)); HIPCHK(streamSync(ctx)); takeCounters(); full->oqDone = true; return 0; }
    ENSURE(full->oqNeed, 8ull * (n + 2)); ENSURE(full->oqPoolOff, 8ull * (n + 2)); ENSURE(full->oqLists, 4ull * YQ_NCLASS * (uint64_t)n + 64); ENSURE(full->oqClsCnt, 64);
    ENSURE(full->oqPrim, sizeof(yoqc::CNode) * (uint64_t)C); ENSURE(full->oqPA, sizeof(yoqc::PAttr) * (uint64_t)C); ENSURE(full->oqPush, sizeof(yoqc::OutRec) * (uint64_t)C); ENSURE(full->oqOut, sizeof(yoqc::OutRec) * (uint64_t)C);
    ENSURE(full->oqOutCnt, 4ull * (n + 2)); ENSURE(full->oqOutOps, 4ull * (n + 2)); ENSURE(full->oqPrimCnt, 4ull * (n + 2));
    HIPCHK(hipMemsetAsync((uint32_t *)full->oqOutCnt.p + n, 0, 8, ctx->stream)); HIPCHK(hipMemsetAsync((uint32_t *)full->oqOutOps.p + n, 0, 8, ctx->stream)); HIPCHK(hipMemsetAsync(full->oqClsCnt.p, 0, 64, ctx->stream));
    OqcArgs A; A.P = full->oqP; A.G = full->oqG; A.cs = full->oqCs.as<uint32_t>(); A.cl = full->oqCl.as<ygpu_clump>(); A.ops = full->oqOpsIn.as<uint32_t>(); A.seeds = full->oqSeeds.as<uint32_t>(); A.qlen = full->oqQlen.as<uint32_t>(); A.nReads = n;
    A.poolOff = full->oqPoolOff.as<unsigned long long>(); A.prim = full->oqPrim.as<yoqc::CNode>(); A.pa = full->oqPA.as<yoqc::PAttr>(); A.push = full->oqPush.as<yoqc::OutRec>(); A.out = full->oqOut.as<yoqc::OutRec>();
    A.outCnt = full->oqOutCnt.as<uint32_t>(); A.outOpsCnt = full->oqOutOps.as<uint32_t>(); A.primCnt = full->oqPrimCnt.as<uint32_t>();
    A.keys = nullptr; A.stack = nullptr; A.nodes = nullptr; A.pfxOff = nullptr; A.path = nullptr; A.pool = nullptr; A.prof = nullptr;
    { const char *e = getenv("YGPU_OQC_MAX"); const int v = e ? atoi(e) : YQ_DEVICE_MAX; A.devMax = v >= 1 && v < YQ_DEVICE_MAX ? v : YQ_DEVICE_MAX; }     // (read at every call: tests lower it to send small reads down the hand-over path)
    { const char *e = getenv("YGPU_OQC_HBM"); A.graphInHbm = e && atoi(e) ? 1 : 0; }      // (read at every call, as YGPU_OQC_MAX)
    static const bool oqProf = getenv("YGPU_OQC_PROF") != nullptr;
    if (oqProf) { ENSURE(full->oqProf, 8ull * 32 * YQ_NCLASS); HIPCHK(hipMemsetAsync(full->oqProf.p, 0, 8ull * 32 * YQ_NCLASS, ctx->stream)); A.prof = full->oqProf.as<unsigned long long>(); }
    uint32_t *lists = full->oqLists.as<uint32_t>();
    KL(k_oqc_classify, dim3(gridFor(n + 1, 256)), dim3(256), 0, ctx->stream, A, full->oqNeed.as<unsigned long long>(), lists, full->oqClsCnt.as<unsigned int>());
    int rc = cubScan64(ctx, full->oqNeed.as<unsigned long long>(), full->oqPoolOff.as<unsigned long long>(), n + 1); if (rc) return rc;
    unsigned long long poolInts = 0; uint32_t nCls[YQ_NCLASS] = {0, 0, 0, 0, 0};
    { uint32_t w[2] = {0, 0}; const FetchPiece pc[2] = {{full->oqPoolOff.as<unsigned long long>() + n, w, 2}, {full->oqClsCnt.p, nCls, YQ_NCLASS}};
      rc = fetchMany(ctx, pc, 2); if (rc) return rc; poolInts = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32); }
    takeCounters();
    ENSURE(full->oqPool, 4ull * (poolInts + 16)); A.pool = full->oqPool.as<int>();
    // work space of the reads in HBM: what a wave's LDS does not hold (the survivors' keys while the nodes are made; everything for the reads of the last class)
    ENSURE(full->oqKeys, sizeof(yoqc::SortKey) * (uint64_t)C); ENSURE(full->oqStack, 4ull * (4ull * C + 8ull * n + 16)); ENSURE(full->oqNodes, sizeof(yoqc::CNode) * (uint64_t)C); ENSURE(full->oqPfx, 4ull * C); ENSURE(full->oqPath, 4ull * C);
    A.keys = full->oqKeys.as<yoqc::SortKey>(); A.stack = full->oqStack.as<int>(); A.nodes = full->oqNodes.as<yoqc::CNode>(); A.pfxOff = full->oqPfx.as<int>(); A.path = full->oqPath.as<int>();
    // the classes: clumps a read may have -> LDS of its workgroup; ints of LDS pool (the first tables; later ones go to the read's slice of the HBM pool)
    static const int capN[YQ_NCLASS] = {112, 224, 448, YQ_DEVICE_MAX, 0};
    // A wave a read, every read of a class resident at once: a launch lasts as long as its slowest read (one of 400 clumps with 280 survivors: 3 ms), and the classes
    // follow one another on the post-filter's one stream.  That latency is off the context's path -- the next batch is running meanwhile -- and a stream of its own for
    // every class is not worth having: streams share four hardware queues, and more than two a context put all contexts' main streams on one (profiles/r05_hw_queues.txt).
    if (nCls[YQ_NCLASS - 1]) KL(k_oqc_raw, dim3(gridFor(nCls[YQ_NCLASS - 1], 64)), dim3(64), 0, ctx->stream, A, lists + (size_t)(YQ_NCLASS - 1) * n, nCls[YQ_NCLASS - 1]);      // left to the host, marked
    for (int c = YQ_NCLASS - 2; c >= 0; c--) if (nCls[c]) {
        const unsigned lds = std::min(YQ_LDS_MAX, oqcLdsBytes(capN[c]));
        KL(k_oqc_wave, dim3(nCls[c]), dim3(64), lds, ctx->stream, A, lists + (size_t)c * n, nCls[c], lds);
    }
    if (kTrace) fprintf(stderr, "[ygpu] post-filter: %u reads with two or more clumps in classes of <= 112 / 224 / 448 / %d clumps: %u / %u / %u / %u, left to the host %u; pool %.1f MB\n", nCls[0] + nCls[1] + nCls[2] + nCls[3] + nCls[4], YQ_DEVICE_MAX, nCls[0], nCls[1], nCls[2], nCls[3], nCls[4], poolInts * 4.0 / 1e6);
    rc = cubScan(ctx, full->oqOutCnt.as<uint32_t>(), full->oqOutStart.as<uint32_t>(), n + 1); if (rc) return rc;
    rc = cubScan(ctx, full->oqOutOps.as<uint32_t>(), full->oqOpsStart.as<uint32_t>(), n + 1); if (rc) return rc;
    uint32_t tot[2] = {0, 0}, scanFail = 0;
    { const FetchPiece pc[3] = {{full->oqOutStart.as<uint32_t>() + n, &tot[0], 1}, {full->oqOpsStart.as<uint32_t>() + n, &tot[1], 1}, {ctx->counters.as<uint32_t>() + CNT_SCANFAIL, &scanFail, 1}}; rc = fetchMany(ctx, pc, 3); if (rc) return rc; }
    if (scanFail) { ctx->err = "post-filter: a look-back of an exclusive sum gave up"; return YGPU_EINTERNAL; }
    full->nFOut = tot[0]; full->nFOps = tot[1];
    ENSURE(full->oqFClumps, sizeof(ygpu_out_clump) * ((uint64_t)tot[0] + 1)); ENSURE(full->oqFOps, 4ull * ((uint64_t)tot[1] + 1));
    KL(k_oqc_gather, dim3(gridFor((uint64_t)n * 64, 256)), dim3(256), 0, ctx->stream, A, full->oqOutStart.as<uint32_t>(), full->oqOpsStart.as<uint32_t>(), full->oqFClumps.as<ygpu_out_clump>(), full->oqFOps.as<uint32_t>());
    if (oqProf) {
        unsigned long long h[32 * YQ_NCLASS]; HIPCHK(hipMemcpyAsync(h, full->oqProf.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        static const char *nm[7] = {"keys", "sort", "dup scan", "nodes+tables", "path walk", "successors", "finish"};
        for (int c = 0; c < YQ_NCLASS; c++) if (h[32 * c + 7]) {
            const unsigned long long *q = h + 32 * c; fprintf(stderr, "[ygpu] post-filter class %d: %llu reads, %.0f clumps, %.0f survivors a read; us a read (largest of any read):", c, q[7], (double)q[8] / q[7], (double)q[9] / q[7]);
            for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.1f (%.0f)", nm[k], q[k] / 100.0 / q[7], q[16 + k] / 100.0);
            fprintf(stderr, "; slowest read %.0f us: %llu clumps, %llu survivors\n", (q[10] >> 24) / 100.0, (q[10] >> 12) & 4095ull, q[10] & 4095ull);
        }
    }
    full->oqDone = true;
    return 0;
}
/* Stage-level test entry for the path's own exclusive sums and orderings (scan.h): n pseudo-random elements from `seed` -- a u32 sum, a u64 sum whose values pass
 * 2^32, an in-place sum, and an ordering by a `key_bits`-bit key (with an offset and, above 12 bits, a shift) -- each checked against the plain host loop. */
int ygpu_selftest_primitives(ygpu_ctx *ctx, uint32_t n, uint32_t seed, int key_bits)
{
    if (!ctx || !ctx->stream || n == 0 || key_bits < 1 || key_bits > 16) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1; auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    std::vector<uint32_t> h32(n), o32(n); std::vector<unsigned long long> h64(n), o64(n);
    for (uint32_t i = 0; i < n; i++) { const uint64_t r = rnd(); h32[i] = (uint32_t)(r % 97u) * ((r >> 40) % 5u == 0 ? 1000u : 1u); h64[i] = (r >> 8) % (1ull << 36); }
    HIPCHK(hipMemsetAsync(ctx->counters.as<uint32_t>() + CNT_SCANFAIL, 0, 4, ctx->stream));
    DevBuf a, b; struct Rel { DevBuf &a, &b; ~Rel() { a.release(); b.release(); } } rel{a, b};
    if (a.ensure(8ull * n + 64) || b.ensure(8ull * n + 64)) { ctx->err = "hipMalloc failed"; return YGPU_ENOMEM; }
    auto fail = [&](const char *what, uint64_t at) { char m[160]; snprintf(m, sizeof m, "selftest: %s differs from the host at element %llu of %u", what, (unsigned long long)at, n); ctx->err = m; return YGPU_EINTERNAL; };
    int rc;
    HIPCHK(hipMemcpyAsync(a.p, h32.data(), 4ull * n, hipMemcpyHostToDevice, ctx->stream));
    rc = cubScan(ctx, a.as<uint32_t>(), b.as<uint32_t>(), n); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(o32.data(), b.p, 4ull * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
    { uint32_t sum = 0; for (uint32_t i = 0; i < n; i++) { if (o32[i] != sum) return fail("the u32 sum", i); sum += h32[i]; } }
    rc = cubScan(ctx, a.as<uint32_t>(), a.as<uint32_t>(), n); if (rc) return rc;                                   // in place
    HIPCHK(hipMemcpyAsync(o32.data(), a.p, 4ull * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
    { uint32_t sum = 0; for (uint32_t i = 0; i < n; i++) { if (o32[i] != sum) return fail("the in-place u32 sum", i); sum += h32[i]; } }
    HIPCHK(hipMemcpyAsync(a.p, h64.data(), 8ull * n, hipMemcpyHostToDevice, ctx->stream));
    rc = cubScan64(ctx, a.as<unsigned long long>(), b.as<unsigned long long>(), n); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(o64.data(), b.p, 8ull * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
    { unsigned long long sum = 0; for (uint32_t i = 0; i < n; i++) { if (o64[i] != sum) return fail("the u64 sum", i); sum += h64[i]; } }
    // ordering: keys in [sub, sub + 2^key_bits), values = indices + 7; shift as the callers choose it (at most 4 096 buckets)
    const uint32_t sub = 1000u, span = 1u << key_bits; const int shift = std::max(0, key_bits - 12); const uint32_t nb = (span >> shift) + 1u;
    for (uint32_t i = 0; i < n; i++) { const uint64_t r = rnd(); h32[i] = sub + (uint32_t)((r >> 20) % span) / ((r & 3u) == 0 ? 7u : 1u); }      // (skewed: some buckets crowded)
    HIPCHK(hipMemcpyAsync(a.p, h32.data(), 4ull * n, hipMemcpyHostToDevice, ctx->stream));
    rc = bucketOrder(ctx, a.as<uint32_t>(), nullptr, 7u, n, sub, shift, nb, b.as<uint32_t>(), ctx->stream); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(o32.data(), b.p, 4ull * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
    { std::vector<uint8_t> seen(n, 0); uint32_t last = 0;
      for (uint32_t i = 0; i < n; i++) { const uint32_t v = o32[i] - 7u; if (v >= n || seen[v]) return fail("the ordering (not a permutation)", i); seen[v] = 1; const uint32_t bk = std::min((h32[v] - sub) >> shift, nb - 1u); if (bk < last) return fail("the ordering (buckets not ascending)", i); last = bk; } }
    // and once more right away: the work words of both must have cleaned themselves up
    rc = bucketOrder(ctx, a.as<uint32_t>(), nullptr, 7u, n, sub, shift, nb, b.as<uint32_t>(), ctx->stream); if (rc) return rc;
    std::vector<uint32_t> again(n); HIPCHK(hipMemcpyAsync(again.data(), b.p, 4ull * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
    { uint32_t last = 0; for (uint32_t i = 0; i < n; i++) { const uint32_t v = again[i] - 7u; if (v >= n) return fail("the second ordering", i); const uint32_t bk = std::min((h32[v] - sub) >> shift, nb - 1u); if (bk < last) return fail("the second ordering (buckets not ascending)", i); last = bk; } }
    uint32_t sf = 0; rc = fetchU32(ctx, ctx->counters.as<uint32_t>() + CNT_SCANFAIL, &sf); if (rc) return rc;
    if (sf) { ctx->err = "selftest: a look-back gave up"; return YGPU_EINTERNAL; }
    // the post-filter's sort on the wave (oqc_stage.h waveSort) against the one-thread routine it stands for (oqc_core.h sortRange, the reference's quicksort with its
    // random tie breaks): arrays of 2 .. YQ_DEVICE_MAX entries around the 64-lane edges, keys from 2 to 4 096 distinct values (ties by the hundred down to none), random,
    // ascending and descending; every entry must land where the routine puts it
    {
        static const int fixedLen[] = {2, 3, 4, 5, 9, 17, 33, 63, 64, 65, 66, 100, 127, 128, 129, 130, 191, 192, 193, 300, 448, 449, 700, 1000, 1500, YQ_DEVICE_MAX};
        const uint32_t nArr = 56; std::vector<uint32_t> off(nArr + 1, 0), seeds(5 * nArr); std::vector<uint64_t> ent;
        for (uint32_t t = 0; t < nArr; t++) {
            const int len = t < sizeof fixedLen / sizeof fixedLen[0] ? fixedLen[t] : 2 + (int)(rnd() % (YQ_DEVICE_MAX - 1));
            const uint64_t span = 1ull << (1 + (seed * 7u + t) % 12u); const int shape = (int)(rnd() % 5);
            for (int i = 0; i < len; i++) { uint64_t k = rnd() % span; if (shape == 3) k = (uint64_t)i * span / len; if (shape == 4) k = (uint64_t)(len - 1 - i) * span / len; ent.push_back((k << 16) | (uint64_t)i); }
            for (int k = 0; k < 5; k++) seeds[5 * t + k] = (uint32_t)rnd();
            off[t + 1] = off[t] + (uint32_t)len;
        }
        std::vector<uint64_t> want(ent.size()), got(ent.size());
        for (uint32_t t = 0; t < nArr; t++) {
            const int len = (int)(off[t + 1] - off[t]); std::vector<yoqc::SortKey> sk(len); std::vector<int> stk(4 * len + 16);
            for (int i = 0; i < len; i++) { sk[i].key = ent[off[t] + i] >> 16; sk[i].clump = i; sk[i].pad = 0; }
            yoqc::Rand rs; for (int k = 0; k < 5; k++) rs.s[k] = seeds[5 * t + k];
            yoqc::Run::sortRange(sk.data(), len, stk.data(), (int)stk.size(), stk.data(), rs);
            for (int i = 0; i < len; i++) want[off[t] + i] = (sk[i].key << 16) | (uint64_t)(uint32_t)sk[i].clump;
        }
        DevBuf dOff, dSeeds, dStack; struct Rel2 { DevBuf &a, &b, &c; ~Rel2() { a.release(); b.release(); c.release(); } } rel2{dOff, dSeeds, dStack};
        if (a.ensure(8ull * ent.size()) || dOff.ensure(4ull * off.size()) || dSeeds.ensure(4ull * seeds.size()) || dStack.ensure(4ull * (2ull * ent.size() + 8ull * nArr + 16))) { ctx->err = "hipMalloc failed"; return YGPU_ENOMEM; }
        HIPCHK(hipMemcpyAsync(a.p, ent.data(), 8ull * ent.size(), hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipMemcpyAsync(dOff.p, off.data(), 4ull * off.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(dSeeds.p, seeds.data(), 4ull * seeds.size(), hipMemcpyHostToDevice, ctx->stream));
        KL(k_oqc_sort_test, dim3(nArr), dim3(64), 4u * YQ_STACK_LDS + 20u * YQ_DEVICE_MAX, ctx->stream, a.as<uint64_t>(), dOff.as<uint32_t>(), dSeeds.as<uint32_t>(), dStack.as<int>(), nArr);
        HIPCHK(hipMemcpyAsync(got.data(), a.p, 8ull * ent.size(), hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        for (uint32_t t = 0; t < nArr; t++) for (uint32_t i = off[t]; i < off[t + 1]; i++) if (got[i] != want[i]) {
            char m[200]; snprintf(m, sizeof m, "selftest: the sort on the wave differs from the one-thread routine: array %u (%u entries), position %u", t, off[t + 1] - off[t], i - off[t]); ctx->err = m; return YGPU_EINTERNAL; }
    }
    return 0;
}
int ygpu_inject_results(ygpu_ctx *ctx, const ygpu_result_batch *r)
{
    if (!ctx || !ctx->stream || !r || r->n_reads != ctx->nReads || !r->clump_start || (r->n_clumps && !r->clumps) || (r->n_ops && !r->ops) || r->n_clumps > 0x7FFFFFF0ull || r->n_ops > 0x7FFFFFF0ull) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->nReads;
    ENSURE(ctx->readStart, 4ull * (n + 1)); ENSURE(ctx->outClumps2, sizeof(ygpu_clump) * (r->n_clumps + 1)); ENSURE(ctx->outOps, 4ull * (r->n_ops + 1)); ENSURE(ctx->ctr, sizeof(DevCounters));
    HIPCHK(hipMemcpyAsync(ctx->readStart.p, r->clump_start, 4ull * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (r->n_clumps) HIPCHK(hipMemcpyAsync(ctx->outClumps2.p, r->clumps, sizeof(ygpu_clump) * r->n_clumps, hipMemcpyHostToDevice, ctx->stream));
    if (r->n_ops) HIPCHK(hipMemcpyAsync(ctx->outOps.p, r->ops, 4ull * r->n_ops, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(streamSync(ctx));
    ctx->nOut = (uint32_t)r->n_clumps; ctx->nOutOps = (uint32_t)r->n_ops; ctx->stageDone = 3;
    return 0;
}
int ygpu_filtered_size(ygpu_ctx *ctx, uint64_t *n_clumps, uint64_t *n_ops)
{
    if (!ctx || !ctx->oqDone) return YGPU_EINVAL;
    if (n_clumps) *n_clumps = ctx->nFOut; if (n_ops) *n_ops = ctx->nFOps;
    return 0;
}
int ygpu_collect_filtered(ygpu_ctx *full, uint32_t *clump_start, ygpu_out_clump *clumps, uint32_t *ops, ygpu_filtered_batch *out)
{
    if (!full || !out || !clump_start || !full->oqDone || (full->nFOut && !clumps) || (full->nFOps && !ops)) return YGPU_EINVAL;
    PfSide *ctx = &full->pf;
    tlsPfFailed = full;
    HIPCHK(hipSetDevice(full->device));
    const uint32_t n = full->pfN;
    HIPCHK(hipMemcpyAsync(clump_start, full->oqOutStart.p, 4ull * (n + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (full->nFOut) HIPCHK(hipMemcpyAsync(clumps, full->oqFClumps.p, sizeof(ygpu_out_clump) * (uint64_t)full->nFOut, hipMemcpyDeviceToHost, ctx->stream));
    if (full->nFOps) HIPCHK(hipMemcpyAsync(ops, full->oqFOps.p, 4ull * full->nFOps, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(streamSync(ctx));
    tlsPfFailed = nullptr;
    out->n_reads = n; out->clump_start = clump_start; out->clumps = clumps; out->ops = ops; out->n_clumps = full->nFOut; out->n_ops = full->nFOps; out->counters = full->pfCounters;
    return 0;
}

static void asyncWorker(ygpu_ctx *ctx)
{
    std::unique_lock<std::mutex> lk(ctx->aMu);
    for (;;) {
        ctx->aCv.wait(lk, [&] { return ctx->aQuit || (ctx->aOpen && !ctx->aDone && ctx->aBatch); });
        if (ctx->aQuit) return;
        const ygpu_read_batch *b = ctx->aBatch; ctx->aBatch = nullptr;
        lk.unlock();
        int rc = ygpu_upload(ctx, b); if (rc == 0) rc = ygpu_run(ctx); if (rc == 0) rc = ygpu_collect(ctx, &ctx->aOut);
        lk.lock();
        ctx->aRc = rc; ctx->aDone = true; ctx->aCv.notify_all();
    }
}
int ygpu_submit(ygpu_ctx *ctx, const ygpu_read_batch *batch, ygpu_ticket *ticket)
{
    if (!ctx || !ctx->stream || !batch || !ticket) return YGPU_EINVAL;
    std::unique_lock<std::mutex> lk(ctx->aMu);
    if (ctx->aOpen) return YGPU_EBUSY;                                       // (the context's error text is the worker's while a ticket is open: the code says it all)
    if (!ctx->worker.joinable()) ctx->worker = std::thread(asyncWorker, ctx);
    ctx->aBatch = batch; ctx->aOpen = true; ctx->aDone = false; ctx->aRc = 0; *ticket = ++ctx->aTicket;
    ctx->aCv.notify_all();
    return 0;
}
int ygpu_poll(ygpu_ctx *ctx, ygpu_ticket ticket)
{
    if (!ctx) return YGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->aMu);
    if (!ctx->aOpen || ticket != ctx->aTicket) return YGPU_EINVAL;
    return ctx->aDone ? 1 : 0;
}
int ygpu_wait(ygpu_ctx *ctx, ygpu_ticket ticket, ygpu_result_batch *out)
{
    if (!ctx || !out) return YGPU_EINVAL;
    std::unique_lock<std::mutex> lk(ctx->aMu);
    if (!ctx->aOpen || ctx->aWaiting || ticket != ctx->aTicket) return YGPU_EINVAL;      // no such open ticket, or another thread is already waiting for it
    ctx->aWaiting = true;
    ctx->aCv.wait(lk, [&] { return ctx->aDone; });
    ctx->aOpen = false; ctx->aWaiting = false; *out = ctx->aOut;
    return ctx->aRc;
}

int ygpu_last_timing(ygpu_ctx *ctx, float *total_ms, int *n_stages, const char *const **names, const float **ms)
{
    if (!ctx) return YGPU_EINVAL;
    if (total_ms) *total_ms = ctx->totalMs; if (n_stages) *n_stages = T_N; if (names) *names = ctx->names; if (ms) *ms = ctx->ms;
    return 0;
}

int ygpu_seed_join(ygpu_ctx *ctx, const ygpu_fragment **frags, uint64_t *n_frags)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    ctx->stageDone = 0; ctx->keepAllFrags = true; int rc = runTo(ctx, 1); ctx->keepAllFrags = false; if (rc) return rc;
    ctx->hFrags.resize(ctx->nFrags);
    if (ctx->nFrags) HIPCHK(hipMemcpy(ctx->hFrags.data(), ctx->frags.p, 16ull * ctx->nFrags, hipMemcpyDeviceToHost));
    for (auto &f : ctx->hFrags) f.reserved = 0;
    *frags = ctx->hFrags.data(); *n_frags = ctx->nFrags; return 0;
}

int ygpu_chain(ygpu_ctx *ctx, const ygpu_fragment **clump_frags, const uint32_t **clump_frag_start, const uint32_t **clump_read_strand, uint64_t *n_clumps)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    ctx->stageDone = 0; int rc = runTo(ctx, 2); if (rc) return rc;
    const uint32_t NC = ctx->nClumps;
    std::vector<ChainClumpRec> recs(ctx->nClumpSlots); std::vector<uint32_t> order(NC); std::vector<ygpu_fragment> cf(ctx->nClumpFrags);
    if (NC) {
        HIPCHK(hipMemcpy(recs.data(), ctx->clumps.p, sizeof(ChainClumpRec) * (uint64_t)ctx->nClumpSlots, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(order.data(), ctx->order.p, 4ull * NC, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(cf.data(), ctx->clumpFrags.p, 16ull * ctx->nClumpFrags, hipMemcpyDeviceToHost));
    }
    ctx->hClumpFrags.clear(); ctx->hClumpFragStart.assign(1, 0); ctx->hClumpRS.clear();
    for (uint32_t r = 0; r < NC; r++) {
        const ChainClumpRec &c = recs[order[r]];
        for (uint32_t k = 0; k < c.nFrags; k++) { ygpu_fragment f = cf[c.fragOff + k]; f.reserved = 0; f.read_strand = c.rs; ctx->hClumpFrags.push_back(f); }
        ctx->hClumpFragStart.push_back((uint32_t)ctx->hClumpFrags.size()); ctx->hClumpRS.push_back(c.rs);
    }
    *clump_frags = ctx->hClumpFrags.data(); *clump_frag_start = ctx->hClumpFragStart.data(); *clump_read_strand = ctx->hClumpRS.data(); *n_clumps = NC;
    return 0;
}

}  // extern "C" (the stage-level DP entry follows its two implementations)

static int dpBatchWave(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n, const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops)
{
    uint32_t *cnt = ctx->counters.as<uint32_t>();
    int listCap, front, genCap, traceRows; alignDims(ctx, listCap, front, genCap, traceRows); listCap = 64;     // no frame stack needed here
    const size_t per = alignScratchBytes(ctx->maxQ, traceRows, listCap, genCap);
    const unsigned waves = (unsigned)std::min<uint64_t>(std::max<uint32_t>(n, 1u), (uint64_t)ctx->nCU * 4);
    uint32_t opsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, (uint64_t)n * (2ull * ctx->maxQ + 64));
    ENSURE(ctx->scratchAlign, per * waves); ENSURE(ctx->dpProbs, sizeof(ygpu_dp_problem) * (uint64_t)(n + 1)); ENSURE(ctx->dpRes, sizeof(ygpu_dp_result) * (uint64_t)(n + 1)); ENSURE(ctx->dpOps, 4ull * opsCap + 64);
    HIPCHK(hipMemcpyAsync(ctx->dpProbs.p, problems, sizeof(ygpu_dp_problem) * (uint64_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(cnt + CNT_QDP, 0, 8, ctx->stream)); HIPCHK(hipMemsetAsync(ctx->errFlag.p, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->dpRes.p, 0, sizeof(ygpu_dp_result) * (uint64_t)(n + 1), ctx->stream));
    DPBatchArgs A; A.P = ctx->P; A.bases = ctx->dBases.as<uint8_t>(); A.B = devBatch(ctx); A.probs = ctx->dpProbs.as<ygpu_dp_problem>(); A.n = n; A.queueHead = cnt + CNT_QDP;
    A.scratch = ctx->scratchAlign.as<uint8_t>(); A.scratchPerWave = per; A.maxQ = ctx->maxQ; A.listCap = listCap; A.genCap = genCap; A.traceRows = traceRows;
    A.res = ctx->dpRes.as<ygpu_dp_result>(); A.ops = ctx->dpOps.as<uint32_t>(); A.opsCount = cnt + CNT_DPOPS; A.opsCap = opsCap; A.errFlag = ctx->errFlag.as<int>();
    if (n) KL(k_dp_batch, dim3(waves), dim3(64), 0, ctx->stream, A);
    uint32_t no = 0, ef = 0; int rc = fetchU32(ctx, cnt + CNT_DPOPS, &no); if (rc) return rc; rc = fetchU32(ctx, ctx->errFlag.p, &ef); if (rc) return rc;
    if (ef) { char b[64]; snprintf(b, sizeof b, "dp batch failed with device error %u", ef); ctx->err = b; return YGPU_EINTERNAL; }
    ctx->hDpRes.resize(n); ctx->hDpOps.resize(no);
    if (n) HIPCHK(hipMemcpy(ctx->hDpRes.data(), ctx->dpRes.p, sizeof(ygpu_dp_result) * (uint64_t)n, hipMemcpyDeviceToHost));
    if (no) HIPCHK(hipMemcpy(ctx->hDpOps.data(), ctx->dpOps.p, 4ull * no, hipMemcpyDeviceToHost));
    *results = ctx->hDpRes.data(); *ops = ctx->hDpOps.data(); *n_ops = no;
    return 0;
}

// The same calls through the kernels ygpu_run uses at the default band (dp_stage.h).  second = the careful-extension instantiation of k_ext_rows.
static int dpBatchLanes(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n, bool second, const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops)
{
    std::vector<ExtProb> xp; std::vector<uint32_t> xdst; std::vector<unsigned long long> xrows; std::vector<JointRec> jp; std::vector<uint32_t> jdst;
    uint64_t gapOpsBound = 64;
    for (uint32_t k = 0; k < n; k++) {
        const ygpu_dp_problem &p = problems[k]; const uint32_t base = ctx->hReadOff[p.read];
        if (p.mode >= YGPU_DP_EXT_FWD) {
            ExtProb e; e.qBase = base; e.rOff = p.rOff; e.qOff = p.qOff; e.qLen = p.qLen; e.flags = (p.strand ? XP_STRAND : 0u) | (p.mode == YGPU_DP_EXT_REV ? XP_REV : 0u) | XP_VALID;
            xp.push_back(e); xdst.push_back(k); xrows.push_back((unsigned long long)((p.qLen + 19u) / 10u));
        } else {
            JointRec j; memset(&j, 0, sizeof j); j.nsro = p.rOff; j.qBase = base; j.nsqo = p.qOff; j.qGap = p.qLen; j.rGap = p.rLen; j.kind = JK_DP; j.flags = (uint8_t)((p.strand ? 1u : 0u) | (p.mode == YGPU_DP_BANDED ? 2u : 0u));
            jp.push_back(j); jdst.push_back(k); gapOpsBound += (uint64_t)p.qLen + p.rLen + 2;
        }
    }
    const uint32_t nX = (uint32_t)xp.size(), nJ = (uint32_t)jp.size(); int rc;
    uint32_t *cnt = ctx->counters.as<uint32_t>(); DevBatch B = devBatch(ctx);
    HIPCHK(hipMemsetAsync(ctx->errFlag.p, 0, 4, ctx->stream));
    std::vector<ExtRes> hres(nX); std::vector<JointRec> hj(nJ); std::vector<uint32_t> xOff(nX + 1, 0), jOff(nJ + 1, 0);
    if (nX) {
        ENSURE(ctx->extProbs, sizeof(ExtProb) * (uint64_t)nX); ENSURE(ctx->extRes, sizeof(ExtRes) * (uint64_t)nX); ENSURE(ctx->chunkCnt, 64);
        HIPCHK(hipMemcpyAsync(ctx->extProbs.p, xp.data(), sizeof(ExtProb) * (uint64_t)nX, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->chunkCnt.p, 0, 64, ctx->stream)); HIPCHK(hipMemsetAsync(ctx->extRes.p, 0, sizeof(ExtRes) * (uint64_t)nX, ctx->stream));
        unsigned long long blocks = 0, opsBound = 64; for (uint32_t k = 0; k < nX; k++) { blocks += xrows[k]; opsBound += 2ull * xp[k].qLen + 4; }
        const bool caps = ctx->P.maxGap < YD_LW || ctx->P.maxIntron < YD_LW;
        const bool pk = extRowsPacked(ctx, caps);
        auto rowsKernel = pk ? k_ext_rows_pk<false> : (caps ? k_ext_rows<true, false> : k_ext_rows<false, false>);
        auto rowsKernel2 = pk ? k_ext_rows_pk<true> : (caps ? k_ext_rows<true, true> : k_ext_rows<false, true>);
        auto traceKernel = pk ? k_ext_trace_pk : k_ext_trace;
        int perCU = 2; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, rowsKernel, 256, 0) != hipSuccess || perCU < 1) perCU = 2;
        const unsigned blocksK = (unsigned)std::min<uint64_t>(((uint64_t)nX + 255) / 256, (uint64_t)ctx->nCU * (second ? 1 : perCU)), wavesK = blocksK * 4u;
        // the arena at the problems' full bound (every lane slot of a flush counts, so twice that) plus a chunk of slack per wave: test-sized batches
        for (unsigned long long mult = 2;; mult *= 8) {
        const unsigned long long nCh = mult * blocks / (YD_CHUNK_FLUSHES * 64ull) + 2ull * wavesK + 64ull;
        if (nCh * (YD_CHUNK_DWORDS * 4ull) > (64ull << 30) || opsBound > 0x7FFFFFF0ull) { ctx->err = "too many extension rows in one ygpu_dp_batch call"; return YGPU_EINVAL; }
        const uint32_t maxCh = (uint32_t)nCh;
        ENSURE(ctx->extTrace, nCh * (YD_CHUNK_DWORDS * 4ull) + 256); ENSURE(ctx->waveChunks, 4ull * (size_t)wavesK * maxCh + 64); ENSURE(ctx->extOps, 4ull * opsBound + 64); ENSURE(ctx->traceCnt, 64);
        HIPCHK(hipMemsetAsync(ctx->traceCnt.p, 0, 64, ctx->stream));
        ExtArgs E; E.P = ctx->P; E.bases = ctx->dBases.as<uint8_t>(); E.fwd = ctx->dFwd.as<uint8_t>(); E.rev = ctx->dRev.as<uint8_t>(); E.fwd4 = ctx->dFwd4.as<uint8_t>(); E.rev4 = ctx->dRev4.as<uint8_t>(); E.probs = ctx->extProbs.as<ExtProb>(); E.nProb = nX;
        E.order = nullptr; E.clock = nullptr; E.trace = ctx->extTrace.as<uint32_t>(); E.nChunks = (uint32_t)nCh; E.chunkCount = ctx->traceCnt.as<unsigned int>(); E.waveChunks = ctx->waveChunks.as<uint32_t>(); E.maxCh = maxCh;
        E.ops = ctx->extOps.as<uint32_t>(); E.opsCount = ctx->traceCnt.as<unsigned int>() + 1; E.opsCap = (uint32_t)opsBound; E.res = ctx->extRes.as<ExtRes>();
        E.queue = ctx->chunkCnt.as<unsigned int>(); E.ctr = nullptr; E.errFlag = ctx->errFlag.as<int>(); E.dbgMode = 0;
        if (second) KL(rowsKernel2, dim3(blocksK), dim3(256), 0, ctx->stream, E); else KL(rowsKernel, dim3(blocksK), dim3(256), 0, ctx->stream, E);
        { const unsigned tbs = pk ? (unsigned)YD_TRACE_BS : 256u; KL(traceKernel, dim3(gridFor(nX, tbs)), dim3(tbs), 0, ctx->stream, E); }
        uint32_t ef2 = 0; rc = fetchU32(ctx, ctx->errFlag.p, &ef2); if (rc) return rc;
        if (ef2 != YERR_TRACEMEM) break;                                    // (other errors are reported below)
        if (mult >= 1024) { ctx->err = "extension trace arena overflows"; return YGPU_ENOMEM; }
        HIPCHK(hipMemsetAsync(ctx->errFlag.p, 0, 4, ctx->stream)); HIPCHK(hipMemsetAsync(ctx->chunkCnt.p, 0, 64, ctx->stream)); HIPCHK(hipMemsetAsync(ctx->extRes.p, 0, sizeof(ExtRes) * (uint64_t)nX, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(hres.data(), ctx->extRes.p, sizeof(ExtRes) * (uint64_t)nX, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        for (uint32_t k = 0; k < nX; k++) xOff[k + 1] = xOff[k] + (hres[k].score > 0 ? hres[k].nOps : 0u);
    }
    if (nJ) {
        const uint32_t gapOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, gapOpsBound);
        ENSURE(ctx->joints, sizeof(JointRec) * (uint64_t)(nJ + 1)); ENSURE(ctx->sortKeys, 4ull * (nJ + 1)); ENSURE(ctx->sortVals, 4ull * (nJ + 1)); ENSURE(ctx->sortVals2, 4ull * (nJ + 1));
        ENSURE(ctx->gapOps, 4ull * gapOpsCap); ENSURE(ctx->slowList, 4ull * (nJ + 1));
        HIPCHK(hipMemcpyAsync(ctx->joints.p, jp.data(), sizeof(JointRec) * (uint64_t)nJ, hipMemcpyHostToDevice, ctx->stream));
        KL(k_dp_classify, dim3(gridFor(nJ, 256)), dim3(256), 0, ctx->stream, ctx->P, ctx->dBases.as<uint8_t>(), ctx->dFwd.as<uint8_t>(), ctx->dRev.as<uint8_t>(), ctx->joints.as<JointRec>(), nJ, ctx->sortKeys.as<uint32_t>(), ctx->sortVals.as<uint32_t>());
        std::vector<uint32_t> keys(nJ), idx(nJ);
        rc = fetchU32(ctx, ctx->sortKeys.p, keys.data(), nJ); if (rc) return rc;
        for (uint32_t k = 0; k < nJ; k++) idx[k] = k;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });           // the production path sorts the DP joints by (strip width, rows) as well
        uint32_t nd[3] = {0, 0, 0}, nb[2] = {0, 0}; for (uint32_t k = 0; k < nJ; k++) if (keys[k] != YD_JKEY_NONE) { const uint32_t cls = gapJointClass(keys[k]); nd[0]++; nd[1] += cls <= 2u; nb[0] += cls == 0u; nb[1] += cls <= 1u; }
        HIPCHK(hipMemcpyAsync(cnt + CNT_NB12, nb, 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->sortVals2.p, idx.data(), 4ull * nJ, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(cnt + CNT_NDP, nd, 12, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(cnt + CNT_SLOW, 0, 4, ctx->stream)); HIPCHK(hipMemsetAsync(cnt + CNT_QALIGN, 0, 4, ctx->stream));
        int listCap, front, genCap, traceRows; alignDims(ctx, listCap, front, genCap, traceRows); listCap = 64;
        const size_t per = alignScratchBytes(ctx->maxQ, traceRows, listCap, genCap); const unsigned waves = 512;
        ENSURE(ctx->scratchAlign, per * waves);
        AlignArgs A; memset(&A, 0, sizeof A); A.P = ctx->P; A.bases = ctx->dBases.as<uint8_t>(); A.B = B; A.queueHead = cnt + CNT_QALIGN; A.scratch = ctx->scratchAlign.as<uint8_t>(); A.scratchPerWave = per;
        A.maxQ = ctx->maxQ; A.listCap = listCap; A.front = front; A.genCap = genCap; A.traceRows = traceRows; A.ctr = ctx->ctr.as<DevCounters>(); A.errFlag = ctx->errFlag.as<int>();
        PhaseArgs X; memset(&X, 0, sizeof X); X.joints = ctx->joints.as<JointRec>(); X.nJoints = nJ; X.sortedVals = ctx->sortVals2.as<uint32_t>(); X.nDP = cnt + CNT_NDP; X.nDPb = cnt + CNT_NB12;
        X.gapOps = ctx->gapOps.as<uint32_t>(); X.gapOpsCount = cnt + CNT_GAPOPS; X.gapOpsCap = gapOpsCap; X.slowList = ctx->slowList.as<uint32_t>(); X.slowCount = cnt + CNT_SLOW;
        const unsigned gBlocks16 = (unsigned)std::min<uint64_t>(gridFor(nJ, 64), (uint64_t)ctx->nCU * 9), gBlocks32 = (unsigned)std::min<uint64_t>(gridFor(nJ, 64), (uint64_t)ctx->nCU * 6);
        ENSURE(ctx->gapScratch, (size_t)YD_GAP_SCRATCH * 64 * std::max(gBlocks16, gBlocks32)); X.gapScratch = ctx->gapScratch.as<uint8_t>();
        KL(k_gap_band<12>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_band<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_lanes<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_lanes<32>, dim3(gBlocks32), dim3(64), 0, ctx->stream, A, X);
        KL(k_gap_wave, dim3(waves), dim3(64), 0, ctx->stream, A, X);
        HIPCHK(hipMemcpyAsync(hj.data(), ctx->joints.p, sizeof(JointRec) * (uint64_t)nJ, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        for (uint32_t k = 0; k < nJ; k++) jOff[k + 1] = jOff[k] + hj[k].nOps;
    }
    uint32_t ef = 0; rc = fetchU32(ctx, ctx->errFlag.p, &ef); if (rc) return rc;
    if (ef) { char b[64]; snprintf(b, sizeof b, "dp batch (lane kernels) failed with device error %u", ef); ctx->err = b; return YGPU_EINTERNAL; }
    const uint32_t totX = xOff[nX], tot = totX + jOff[nJ];
    for (auto &v : jOff) v += totX;
    ENSURE(ctx->dpRes, sizeof(ygpu_dp_result) * (uint64_t)(n + 1)); ENSURE(ctx->dpOps, 4ull * tot + 64); ENSURE(ctx->dpProbs, 8ull * (n + 2));
    uint32_t *dOff = ctx->dpProbs.as<uint32_t>(), *dDst = dOff + (n + 2);       // per-list offsets and destinations (the lists are done one after the other)
    if (nX) {
        HIPCHK(hipMemcpyAsync(dOff, xOff.data(), 4ull * nX, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipMemcpyAsync(dDst, xdst.data(), 4ull * nX, hipMemcpyHostToDevice, ctx->stream));
        KL(k_dp_gather_ext, dim3(gridFor(nX, 256)), dim3(256), 0, ctx->stream, ctx->extProbs.as<ExtProb>(), ctx->extRes.as<ExtRes>(), ctx->extTrace.as<uint32_t>(), dOff, dDst, nX,
           ctx->dpRes.as<ygpu_dp_result>(), ctx->dpOps.as<uint32_t>());
        HIPCHK(streamSync(ctx));
    }
    if (nJ) {
        HIPCHK(hipMemcpyAsync(dOff, jOff.data(), 4ull * nJ, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(hipMemcpyAsync(dDst, jdst.data(), 4ull * nJ, hipMemcpyHostToDevice, ctx->stream));
        KL(k_dp_gather_gap, dim3(gridFor(nJ, 256)), dim3(256), 0, ctx->stream, ctx->P, ctx->dBases.as<uint8_t>(), ctx->dFwd.as<uint8_t>(), ctx->dRev.as<uint8_t>(), ctx->joints.as<JointRec>(), ctx->gapOps.as<uint32_t>(), dOff, dDst, nJ,
           ctx->dpRes.as<ygpu_dp_result>(), ctx->dpOps.as<uint32_t>());
        HIPCHK(streamSync(ctx));
    }
    ctx->hDpRes.resize(n); ctx->hDpOps.resize(tot);
    if (n) HIPCHK(hipMemcpy(ctx->hDpRes.data(), ctx->dpRes.p, sizeof(ygpu_dp_result) * (uint64_t)n, hipMemcpyDeviceToHost));
    if (tot) HIPCHK(hipMemcpy(ctx->hDpOps.data(), ctx->dpOps.p, 4ull * tot, hipMemcpyDeviceToHost));
    *results = ctx->hDpRes.data(); *ops = ctx->hDpOps.data(); *n_ops = tot;
    return 0;
}

extern "C" {
int ygpu_dp_batch_ex(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n, int kernels, const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops)
{
    if (!ctx || !ctx->stream || !ctx->nReads || kernels < YGPU_DP_KERNELS_AUTO || kernels > YGPU_DP_KERNELS_LANES_CAREFUL) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    for (uint32_t k = 0; k < n; k++) if (problems[k].read >= ctx->nReads || problems[k].mode > 3) { ctx->err = "bad DP problem"; return YGPU_EINVAL; }
    const bool useLanes = ctx->laneExt && ctx->P.bandWidth == 5 && ctx->P.maxGap >= YD_LBAND;       // the same choice stageAlign makes
    if (kernels == YGPU_DP_KERNELS_WAVE || (kernels == YGPU_DP_KERNELS_AUTO && !useLanes)) return dpBatchWave(ctx, problems, n, results, ops, n_ops);
    if (!useLanes) { ctx->err = "the lane kernels need -BW 5 and -G >= 10"; return YGPU_EINVAL; }
    return dpBatchLanes(ctx, problems, n, kernels == YGPU_DP_KERNELS_LANES_CAREFUL, results, ops, n_ops);
}
int ygpu_dp_batch(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n, const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops)
{ return ygpu_dp_batch_ex(ctx, problems, n, YGPU_DP_KERNELS_AUTO, results, ops, n_ops); }
}  // extern "C"
