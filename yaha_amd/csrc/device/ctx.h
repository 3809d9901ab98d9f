// ctx.h -- the device context and what the stage files share: buffers, counters, the macros every launch and wait goes through.
//
// The C-ABI of include/yaha_hip.h is implemented by six translation units over this header (round 6; one file of 1 750 lines before):
//   abi.hip          contexts: index image (one upload, chain of peer copies), clone / park / presize / destroy, ygpu_run, tickets, timing
//   stage_seed.hip   batch upload, A1 + A2 (k-mer lookup, hit expansion, the segments' sort) and the fragment array      seed.h, segsort.h
//   stage_chain.hip  A3 + A4 (regions, chain DP, creation-order ranks)                                                 chain.h, chain_lanes.h, regions.h
//   stage_align.hip  A5..A8 + A10's layout, the stage-level DP entry                                                   phase_lanes.h ... ext_lanes_pk.h, layout.h
//   stage_out.hip    results to the host, the post-filter stage on a snapshot                                          oqc_stage.h
//   prims.hip        exclusive sums and orderings of the batch layout                                                  scan.h
// Every kernel header is included by exactly one of them (a kernel is one symbol of the library).
//
// HBM layout per context (one or more per GPU -- ygpu_clone shares the index image; reads shard across contexts and GPUs, no collective):
//   index   : packed 4-bit reference, startingOffs[4^L+1], ROA[totalMatches]           (resident for the whole run)
//   batch   : forward + reverse-complement codes (1 B/base), read offsets, k-mer offsets
//   stage arenas, grown on demand and reused across batches:
//     A1  posS/posC/posRsI per k-mer  -> exclusive scan -> hit offsets
//     A2  64-bit hit keys (double buffer for the sort) -> fragment array (16 B each)
//     A3  region starts, multi-fragment region list
//     A4  clump records + clump fragment lists (atomic arenas), per-region counts -> creation-order ranks
//     A5-8 default band: joint records + gap-op arena, root states + phase-1 lists, extension problems / results, extension trace
//          arena (128 B per 8 rows, chunks taken as rows are computed; the extension ops are written into it), split-root scratch;
//          general path and leftovers: per-wave scratch (trace strip, DP temp list, frame stack with edit-list buffers); output arenas
//   results : clump records in QS->clumps order, ops arena, clump_start per read
// Every stage is a handful of launches on one stream; sizes that the next stage needs cross the PCIe as single words.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include "common.h"
#include "../oqc_core.h"

#define HIPCHK(call) do { hipError_t e_ = (call); \
    if (e_ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e_); return YGPU_ENODEV; } } while (0)

// Every kernel launch is followed by a check of the submit status: a launch the runtime rejects (too much LDS, a grid that is too large, a code object
// for another architecture) would otherwise leave the stage running on unwritten buffers, and the later stream synchronisation reports nothing.
#define KL(kern, grid, block, shmem, st, ...) do { hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__); hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { ctx->err = std::string("launch of " #kern " failed: ") + hipGetErrorString(e_); return YGPU_ENODEV; } } while (0)

#define YD_MAX_CHUNK_EV 16

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
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
    int ensureExact(size_t bytes)
    {
        if (bytes <= cap) return 0;
        release();
        void *np = nullptr;
        if (hipMalloc(&np, bytes) != hipSuccess) return -1;
        p = np; cap = bytes; return 0;
    }
    template <class T> T *as() const { return (T *)p; }
};

// words of ygpu_ctx::counters (device): queue heads, arena counts, list lengths of one batch
enum {
    CNT_NMULTI = 0, CNT_MAXN, CNT_CLUMPS, CNT_CFRAGS, CNT_QCHAIN, CNT_QALIGN, CNT_OUTCLUMPS, CNT_OUTOPS, CNT_QDP, CNT_DPOPS, CNT_NBIG, CNT_QBIG,
    CNT_STATEOPS, CNT_EXTOPS, CNT_QEXT, CNT_SLOW, CNT_NDP, CNT_NDP16, CNT_GAPOPS, CNT_PAD_EVEN, CNT_NMULTI2 /* (multi | tiny << 32), (small | middle << 32): two 64-bit words */,
        CNT_NTINY, CNT_NSMALL, CNT_NMID, CNT_NB12, CNT_NB16,
        CNT_NB24,
    CNT_SEGC,                            // YD_SEG_NCLASS + 1 words: the segments of the workgroup-sort classes, the long ones
    CNT_NFRAGS = CNT_SEGC + 16,          // + the look-back's flag + the order check's (wgsort.h: a sorted hit that is not above the hit before it)
    CNT_NREG = CNT_NFRAGS + 3,           // + flag
    CNT_SCANFAIL = CNT_NREG + 2,         // raised by a look-back of scan.h that gave up
    CNT_N = CNT_NREG + 4
};
// the first T_TOP entries partition a run; the rest are sub-intervals of align_dp (lane-extension pipeline)
enum { T_SEED = 0, T_SORT, T_FRAGS, T_CHAIN, T_ALIGN, T_LAYOUT, T_TOP, T_P1 = T_TOP, T_XROWS, T_XTRACE, T_P3, T_XROWS_DEV, T_XROWS_PK, T_N };

// per-process state of the devices (abi.hip)
extern std::atomic<int> gCtxPerDevice[64];      // live contexts per device of this process: they share the device's free memory
// One rows launch at a time per device (YGPU_ROWS_SERIAL): a context's main rows launch waits for the one launched before it on the device, whichever context that
// was -- two of them side by side take the whole chip between them and leave the other batches' kernels nothing, which is what the partial launch is there to avoid.
// (a ring of events: YGPU_ROWS_SERIAL=k lets k launches overlap)
extern std::mutex gRowsMu[64];
extern hipEvent_t gRowsEv[64][4];
extern bool gRowsEvValid[64][4];
extern unsigned long long gRowsSeq[64];
extern std::atomic<int> gActiveRuns[64];        // contexts of this process inside ygpu_run on the device right now: a rows launch shares the device when there are two or more
extern const char *const kStageNames[T_N];

// What the post-filter's host code works with in place of the context: the second stream, its own pinned slot, wait event, look-back words and message -- the
// stage runs on a SNAPSHOT of a batch's results (ygpu_postfilter_snapshot) and may therefore run on a thread of its own while the context itself is already
// uploading and running the next batch.  (Same member names as the context's, so the macros and the small helpers below serve both.)
struct PfSide {
    std::string err;
    hipStream_t stream = nullptr;
    uint32_t *pinned = nullptr;
    hipEvent_t evSync = nullptr;
    DevBuf scanState, counters;
    int device = 0;
};

struct ygpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevParams P{};
    std::string err;
    int nCU = 256;
    // index image (shared by the device's clones)
    DevBuf dBases, dSO, dROA, dLow;
    bool sharedIndex = false;
    // batch
    uint32_t nReads = 0;
    int maxQ = 0;
    uint64_t totalBases = 0;
    uint32_t nKmers = 0;
    std::vector<uint32_t> hReadOff, hKmerOff;
    DevBuf dFwd, dRev, dFwd4, dRev4, dReadOff, dKmerOff;
    // arenas: seed stage
    DevBuf bigB, bigE;
    DevBuf posS, posC, posRsI, hitOff, expandStart, keysA, keysB, segOff, isHead, tileState, frags;
    DevBuf segLists, subB, subE, subLists, subBigB, subBigE, sub2B, sub2E, sub2Lists, sub3B, sub3E, sub3Lists, kmerParts;
    // chain stage
    DevBuf regStart, multiList, smallList, bigList, regionCount, regionBase;
    DevBuf clumps, clumpFrags, clumpFrags0, order, clumpsSorted, scratchChain;
    // align stage
    DevBuf rootPush, rootBase, outClumps, outClumps2, outOps, outRoot, outPush, dstIdx, readCount, readStart;
    DevBuf scratchAlign, dpProbs, dpRes, dpOps;
    DevBuf rootState, stateOps, extProbs, rowsBound, stripOff, extRes, extTrace, chunkCnt;
    DevBuf memoKeys, memoCount, probs2, rowsBound2, stripOff2, extRes2, extTrace2, splitScratch, fallList;
    DevBuf keys2a, keys2b, vals2a, vals2b, extKeys, extVals, extKeys2, extOrder, slowList;
    DevBuf gapScratch, jointCount, jointBase, joints, sortKeys, sortVals, sortKeys2, sortVals2, gapOps;
    DevBuf waveChunks, extOps, traceCnt, rowsClock;
    // counters and work words
    DevBuf counters, ctr, errFlag, scanState, bucketWork;
    // stage bookkeeping
    bool evUsed[16] = {false};
    double traceT = 0;
    hipStream_t stream2 = nullptr;
    hipEvent_t evChunk[YD_MAX_CHUNK_EV], evTail;
    bool counted = false;
    bool parked = false;
    long long traceBudgetBlocks = 0;
    uint32_t lastClumpSlots = 0;
    bool keepAllFrags = false;
    unsigned long long hRowsClock[2] = {0, 0};
    int laneChunks = 0;
    uint32_t segSortMax = 0;                     // (set by initCommon: YD_SEGSORT_MAX, or YGPU_SEGSORT_MAX)
    int splitLanes = 1;
    int rows2PerCU = 0;
    int alignWavesPerCU = 0;
    int laneExt = 1;
    std::vector<unsigned long long> hStripOff;
    int runsDone = 0;
    double traceRatio = 0.0, opsRatio = 0.03;
    int statRanges = 0, statAttempts = 0;
    double statT0 = 0;
    // post-filter stage (oqc_stage.h)
    DevBuf oqProf, oqLists, oqClsCnt, oqThr, oqSeqStart, oqSeqLen, oqNeed, oqPoolOff, oqKeys, oqStack, oqNodes, oqPrim, oqPA, oqPfx, oqPath, oqPool;
    DevBuf oqPush, oqOut, oqOutCnt, oqOutOps, oqPrimCnt, oqOutStart, oqOpsStart, oqFClumps, oqFOps;
    bool oqSet = false, oqDone = false;
    yoqc::Params oqP{};
    yoqc::Seqs oqG{};
    uint32_t nFOut = 0, nFOps = 0;
    // the snapshot the stage works on (taken by the context's thread) and the copy of its sizes the stage runs with (its own thread)
    PfSide pf;
    std::atomic<bool> pfSnap{false};
    hipEvent_t evSnap = nullptr;
    uint32_t snapN = 0, snapC = 0, snapOps = 0, pfN = 0;
    unsigned long long runGen = 0, snapGen = 0, filtGen = 0;       // ygpu_run / ygpu_inject_results count; the count the snapshot / the filtered results belong to
    ygpu_counters pfCounters{};
    DevCounters *snapCtr = nullptr;
    DevCounters snapCtrPlain{};
    unsigned long long snapHits = 0, snapFrags = 0, snapRegions = 0;
    DevBuf oqCs, oqCl, oqOpsIn, oqSeeds, oqQlen;
    // stage state
    uint32_t hOutCounts[2] = {0, 0}, hOutEf = 0;
    bool hOutValid = false;
    uint32_t nHits = 0, nFrags = 0, nRegions = 0, nMulti = 0, nTiny = 0, nSmall = 0, nMid = 0, nBig = 0, maxN = 0;
    int sortRankUsed = 0;                // the ranking the batch's hits were sorted with (stage_seed.hip: 0 = LDS atomics, 1 = ballots)
    bool sortRankForced = false;         // ... because YGPU_SORT_RANK said so
    uint32_t nClumpSlots = 0, nClumps = 0, nClumpFrags = 0, nOut = 0, nOutOps = 0;
    int stageDone = 0;     // 0 none, 1 fragments, 2 chain, 3 all
    // host results
    std::vector<uint32_t> hClumpStart, hOps, hClumpFragStart, hClumpRS, hDpOps;
    std::vector<ygpu_clump> hClumps;
    std::vector<ygpu_fragment> hFrags, hClumpFrags;
    std::vector<ygpu_dp_result> hDpRes;
    ygpu_counters hCounters{};
    // asynchronous tickets (ygpu_submit / ygpu_wait): one worker thread per context, started on first use
    std::thread worker;
    std::mutex aMu;
    std::condition_variable aCv;
    const ygpu_read_batch *aBatch = nullptr;
    uint64_t aTicket = 0;
    int aRc = 0;
    bool aOpen = false, aDone = false, aQuit = false, aWaiting = false;
    ygpu_result_batch aOut{};
    // timing
    long long lastFall = -1;
    unsigned int hFall = 0;
    uint32_t *pinned = nullptr;
    hipEvent_t evSync = nullptr;
    hipEvent_t ev[T_N][2];
    float ms[T_N] = {0};
    float totalMs = 0;
    const char *names[T_N];
    bool rowsPacked = false;
};

static inline DevBatch devBatch(ygpu_ctx *c)
{
    DevBatch b; b.fwd = c->dFwd.as<uint8_t>(); b.rev = c->dRev.as<uint8_t>(); b.readOff = c->dReadOff.as<uint32_t>(); b.nReads = c->nReads; return b;
}
static inline unsigned gridFor(uint64_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
static inline double nowMs() { using namespace std::chrono; return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count(); }
static const bool kTrace = getenv("YGPU_TRACE") != nullptr;
static const bool kStats = getenv("YGPU_STATS") != nullptr;      // one line per ygpu_run: attempts of the align stage, ranges, arena size

// Waits of the host for its stream: on an event created with hipEventBlockingSync, so that the thread sleeps instead of spinning -- a context has ~12 such
// waits per batch, each tens of milliseconds long, and a node runs (GPUs x contexts) of these threads (YGPU_SPIN_SYNC=1: plain hipStreamSynchronize).
template <class C> static hipError_t streamSync(C *ctx)
{
    static const bool spin = getenv("YGPU_SPIN_SYNC") != nullptr;
    if (spin || !ctx->evSync) return hipStreamSynchronize(ctx->stream);
    hipError_t e = hipEventRecord(ctx->evSync, ctx->stream);
    return e != hipSuccess ? e : hipEventSynchronize(ctx->evSync);
}

#define TRACE(what) do { if (kTrace) { streamSync(ctx); double t_ = nowMs(); \
    fprintf(stderr, "[ygpu] %-28s %9.3f ms\n", what, t_ - ctx->traceT); ctx->traceT = t_; } } while (0)
#define ENSURE(buf, bytes) do { const size_t was_ = (buf).cap; \
    if ((buf).ensure(bytes)) { ctx->err = "hipMalloc failed for " #buf; return YGPU_ENOMEM; } \
    if (kStats && (buf).cap != was_ && (buf).cap >= (1ull << 30)) \
        fprintf(stderr, "[ygpu] ctx %p: " #buf " grows %.2f -> %.2f GB\n", (void *)ctx, was_ / 1e9, (buf).cap / 1e9); } while (0)
#define EV0(t) (ctx->evUsed[t] = true, hipEventRecord(ctx->ev[t][0], ctx->stream))
#define EV1(t) hipEventRecord(ctx->ev[t][1], ctx->stream)

// the next stage's sizes cross PCIe as a few words, through a pinned slot (a pageable destination goes through a staging kernel and a second copy)
template <class C> static int fetchU32(C *ctx, const void *dptr, uint32_t *out, size_t n = 1)
{
    if (ctx->pinned && n <= 64) {
        HIPCHK(hipMemcpyAsync(ctx->pinned, dptr, 4 * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        memcpy(out, ctx->pinned, 4 * n); return 0;
    }
    HIPCHK(hipMemcpyAsync(out, dptr, 4 * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx)); return 0;
}

// several small pieces in ONE wait (the copies queue up behind the kernels, one event is waited for): every wait of the host is a gap on the device when one
// context runs alone
struct FetchPiece { const void *src; uint32_t *dst; uint32_t n; };
template <class C> static int fetchMany(C *ctx, const FetchPiece *pc, int np)
{
    uint32_t tot = 0; for (int k = 0; k < np; k++) tot += pc[k].n;
    if (!ctx->pinned || tot > 64) { for (int k = 0; k < np; k++) { int rc = fetchU32(ctx, pc[k].src, pc[k].dst, pc[k].n); if (rc) return rc; } return 0; }
    uint32_t o = 0;
    for (int k = 0; k < np; k++) { HIPCHK(hipMemcpyAsync(ctx->pinned + o, pc[k].src, 4ull * pc[k].n, hipMemcpyDeviceToHost, ctx->stream)); o += pc[k].n; }
    HIPCHK(streamSync(ctx));
    o = 0; for (int k = 0; k < np; k++) { memcpy(pc[k].dst, ctx->pinned + o, 4ull * pc[k].n); o += pc[k].n; }
    return 0;
}

// ---- prims.hip: exclusive sums and orderings (scan.h; one launch a scan, two an ordering; their work words clean themselves up) ---------------------------
int ydScan32(DevBuf &scanState, std::string &err, unsigned int *failed, hipStream_t st, const uint32_t *in, uint32_t *out, uint32_t n);
int ydScan64(DevBuf &scanState, std::string &err, unsigned int *failed, hipStream_t st, const unsigned long long *in, unsigned long long *out, uint32_t n);
template <class C> static int cubScan(C *ctx, const uint32_t *in, uint32_t *out, uint32_t n)
{ return ydScan32(ctx->scanState, ctx->err, (unsigned int *)ctx->counters.p + CNT_SCANFAIL, ctx->stream, in, out, n); }
template <class C> static int cubScan64(C *ctx, const unsigned long long *in, unsigned long long *out, uint32_t n)
{ return ydScan64(ctx->scanState, ctx->err, (unsigned int *)ctx->counters.p + CNT_SCANFAIL, ctx->stream, in, out, n); }
// order[] = the items' values grouped by bucket((key - sub) >> shift), ascending; vals == nullptr: the values are the items' indices + valBase
int bucketOrder(ygpu_ctx *ctx, const uint32_t *keys, const uint32_t *vals, uint32_t valBase, uint32_t n, uint32_t sub, int shift, uint32_t nb,
                uint32_t *outVals, hipStream_t st);
size_t ydBucketWorkBytes();
bool ydCheckStateOn();                                        // YGPU_CHECK_STATE=1
int ydCheckZero(hipStream_t st, std::string &err, unsigned int *scratch, const DevBuf *const *bufs, const char *const *names, int n, const char *when);
enum { kBucketMax = 4096 };                                   // buckets of an ordering (scan.h YD_BKT_MAX: 12 key bits)

// ---- the stages (runTo in abi.hip drives them) -------------------------------------------------------------------------------------------------------------
int stageSeed(ygpu_ctx *ctx);                                 // stage_seed.hip: A1 + A2
int ydSortRank();                                             //                 the ranking the workgroup sort uses now: 0 = LDS atomics, 1 = ballots
// buildFrags: the keys were not in order behind the atomic ranking; the caller runs stageSeed again (never leaves the library)
enum { YD_RESORT = 1000 };
int buildFrags(ygpu_ctx *ctx, bool redo = false);             //                 the fragment array from the sorted keys (redo: the regions stand, the records are rebuilt)
int uploadBatch(ygpu_ctx *ctx, const ygpu_read_batch *b, bool wait);
int ydLowOffsets(ygpu_ctx *ctx, const ygpu_index_view *ix);   //                 the bit table of k_kmer_lookup (once per image)
int ydFirstLaunch(ygpu_ctx *ctx);                             //                 a first, empty launch: loads the library's code object beside the image's copy
uint32_t ydSegSortMax();
size_t ydLowTableBytes();
int stageChain(ygpu_ctx *ctx);                                // stage_chain.hip: A3 + A4
int stageAlign(ygpu_ctx *ctx);                                // stage_align.hip: A5..A8, layout
int runTo(ygpu_ctx *ctx, int stage);                          // abi.hip
int ydSelftestSegSort(ygpu_ctx *ctx, uint64_t &x);                    // stage_seed.hip: the workgroup sort of A2, both rankings, against std::stable_sort
int ydSelftestWaveSort(ygpu_ctx *ctx, uint32_t seed, uint64_t &x);      // stage_out.hip: the post-filter's sort on the wave against the one-thread routine
std::vector<DevBuf *> allBuffers(ygpu_ctx *ctx);              // abi.hip: every device buffer of a context, in the order of the arena profile
extern thread_local const ygpu_ctx *tlsPfFailed;              // the context whose post-filter side failed last on this thread: ygpu_last_error then reports that side's message
