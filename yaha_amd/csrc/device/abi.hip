// abi.hip -- contexts and the C-ABI of include/yaha_hip.h around the stage files (see ctx.h for the map): the index image's way into HBM (one upload, a chain of
// peer copies for further devices), clones / parking / presizing, ygpu_run, the asynchronous tickets, timing.  The reference's counterpart is the set-up and the
// thread loop of Query.c:565-690; nothing here computes.
#include "ctx.h"
#include <memory>

std::atomic<int> gCtxPerDevice[64];
std::mutex gRowsMu[64];
hipEvent_t gRowsEv[64][4];
bool gRowsEvValid[64][4];
unsigned long long gRowsSeq[64];
std::atomic<int> gActiveRuns[64];
// (the last one is a flag, not a time: 1 when k_ext_rows_pk ran, ext_lanes_pk.h)
const char *const kStageNames[T_N] = {"seed_lookup", "hit_sort", "fragments_regions", "chain", "align_dp", "layout", "align_p1_gapfill", "ext_rows", "ext_trace",
                                      "align_p3_score_split", "ext_rows_device_clock", "ext_rows_packed16"};
thread_local const ygpu_ctx *tlsPfFailed = nullptr;

int runTo(ygpu_ctx *ctx, int stage)
{
    int rc;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->stageDone < 1) {
        rc = stageSeed(ctx); if (rc) return rc;
        EV0(T_FRAGS); rc = buildFrags(ctx);
        // (the order check behind the sort failed with the ranking by LDS atomics: once more from the k-mers, now -- and from now on -- with the ballots; wgsort.h)
        if (rc == YD_RESORT) { rc = stageSeed(ctx); if (rc) return rc; EV0(T_FRAGS); rc = buildFrags(ctx);
            if (rc == YD_RESORT) { ctx->err = "hit sort: the sorted keys are not in ascending order"; rc = YGPU_EINTERNAL; } }
        if (rc) return rc;
        if (!ctx->nFrags) EV1(T_FRAGS);
        ctx->stageDone = 1;
    }
    if (stage >= 2 && ctx->stageDone < 2) { if (ctx->nFrags) { rc = stageChain(ctx); if (rc) return rc; } ctx->stageDone = 2; }
    if (stage >= 3 && ctx->stageDone < 3) { rc = stageAlign(ctx); if (rc) return rc; ctx->stageDone = 3; }
    // (the stream is drained by this fetch: the flag a look-back of scan.h raises when a tile never showed up -- the state words are then made clean again)
    uint32_t scanFail = 0; rc = fetchU32(ctx, ctx->counters.as<uint32_t>() + CNT_SCANFAIL, &scanFail); if (rc) return rc;
    if (scanFail) { if (ctx->scanState.p) HIPCHK(hipMemsetAsync(ctx->scanState.p, 0, ctx->scanState.cap, ctx->stream));
        ctx->err = "exclusive sum: a tile was not published within 30 s (look-back gave up)"; return YGPU_EINTERNAL; }
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
    ctx->segSortMax = ydSegSortMax();
    if (const char *e = getenv("YGPU_SEGSORT_MAX")) { long v = atol(e); if (v >= 1 && v <= (long)ydSegSortMax()) ctx->segSortMax = (uint32_t)v; }
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
        src[0] = (const char *)ix->bases; bytes[0] = (size_t)ix->n_base_bytes; src[1] = (const char *)ix->startingOffs; bytes[1] = (size_t)(4 * (HT + 1));
            src[2] = (const char *)ix->ROA; bytes[2] = (size_t)(4ull * ix->totalMatches);
        for (size_t part = 0; part < 3; part++) for (size_t o = 0; o < bytes[part]; o += pieceBytes) pieces.push_back({part, o, std::min(pieceBytes, bytes[part] - o)});
    }
};
struct ImageState {                                                                       // one per device of a ygpu_init_multi call
    std::unique_ptr<std::atomic<int>[]> done; std::atomic<int> failed{0}; char *dst[3] = {nullptr, nullptr, nullptr};
};

// pieces from host memory: `nt` threads, each with a stream of its own, pieces taken from a common counter (the runtime's own path for unpinned memory pins a piece
// and lets the DMA engines read it in place: 55 GB/s for a single hipMemcpy of a mapped file, tools/micro/h2d_probe.hip); staged = the threads copy the pieces
// into page-locked buffers of their own first (YGPU_UPLOAD=staged:T)
static void uploadFromHost(int device, const ImagePlan &plan, ImageState &me, int nt, bool staged)
{
    std::atomic<size_t> next(0);
    auto work = [&]() {
        if (hipSetDevice(device) != hipSuccess) { me.failed = 1; return; }
        hipStream_t st; if (hipStreamCreate(&st) != hipSuccess) { me.failed = 1; return; }
        char *buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; size_t pend[2] = {0, 0}; size_t maxPiece = 0;
            for (auto &q : plan.pieces) maxPiece = std::max(maxPiece, q.bytes);
        if (staged) for (int k = 0; k < 2; k++) {
            if (hipHostMalloc((void **)&buf[k], maxPiece, hipHostMallocDefault) != hipSuccess) { buf[k] = nullptr; me.failed = 1; }
            if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { ev[k] = nullptr; me.failed = 1; }
        }
        for (int k = 0; !me.failed; k ^= 1) {
            const size_t i = next.fetch_add(1); if (i >= plan.pieces.size()) break;
            if (me.done[i].load(std::memory_order_acquire)) continue;          // (already here: the pieces a broken peer chain delivered before it broke)
            const ImagePiece &q = plan.pieces[i]; char *d = me.dst[q.part] + q.off; const char *sp = plan.src[q.part] + q.off;
            if (!staged) { if (hipMemcpyAsync(d, sp, q.bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) me.failed = 1;
                else me.done[i].store(1, std::memory_order_release); continue; }
            if (used[k]) { if (hipEventSynchronize(ev[k]) != hipSuccess) { me.failed = 1; break; } me.done[pend[k]].store(1, std::memory_order_release); used[k] = false; }
            memcpy(buf[k], sp, q.bytes);
            if (hipMemcpyAsync(d, buf[k], q.bytes, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { me.failed = 1; break; }
            used[k] = true; pend[k] = i;
        }
        if (staged) { if (hipStreamSynchronize(st) != hipSuccess) me.failed = 1;
            for (int k = 0; k < 2; k++) { if (used[k] && !me.failed) me.done[pend[k]].store(1, std::memory_order_release); if (buf[k]) (void)hipHostFree(buf[k]);
            if (ev[k]) (void)hipEventDestroy(ev[k]); } }
        (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th; for (int t = 1; t < nt; t++) th.emplace_back(work);
    work(); for (auto &x : th) x.join();
    if (me.failed) (void)hipGetLastError();
}
// pieces from the neighbour's image, each as soon as the neighbour has it
// Returns the piece at which the chain broke (a copy the runtime refused, a source device that failed, YGPU_PEER_FAIL_AT), or -1: the pieces published so far stay,
// the caller takes the rest from the host.  Only a failure of THIS device's own set-up is final (me.failed).
static long copyFromPeer(int device, int srcDevice, const ImagePlan &plan, ImageState &me, ImageState &from, long failAt)
{
    if (hipSetDevice(device) != hipSuccess) { me.failed = 1; return -1; }
    hipStream_t st; if (hipStreamCreate(&st) != hipSuccess) { me.failed = 1; return -1; }
    long broke = -1;
    // (a window of copies in flight: the events of the last W pieces; a piece is published once its event has completed)
    const int W = 4; hipEvent_t ev[W]; size_t pend[W]; bool used[W]; for (int k = 0; k < W; k++) { used[k] = false;
        if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) me.failed = 1; }
    for (size_t i = 0; i < plan.pieces.size() && !me.failed; i++) {
        const int k = (int)(i % W);
        if (used[k]) { if (hipEventSynchronize(ev[k]) != hipSuccess) { used[k] = false; broke = (long)pend[k]; break; } me.done[pend[k]].store(1, std::memory_order_release);
            used[k] = false; }
        while (!from.done[i].load(std::memory_order_acquire)) { if (from.failed) { broke = (long)i; break; } std::this_thread::yield(); }
        if (broke >= 0) break;
        const ImagePiece &q = plan.pieces[i];
        const hipError_t e = (long)i == failAt ? hipErrorUnknown      // (YGPU_PEER_FAIL_AT: the fault a test injects)
                           : device == srcDevice ? hipMemcpyAsync(me.dst[q.part] + q.off, from.dst[q.part] + q.off, q.bytes, hipMemcpyDeviceToDevice, st)
                                                 : hipMemcpyPeerAsync(me.dst[q.part] + q.off, device, from.dst[q.part] + q.off, srcDevice, q.bytes, st);
        if (e != hipSuccess || hipEventRecord(ev[k], st) != hipSuccess) { broke = (long)i; break; }
        used[k] = true; pend[k] = i;
    }
    // what is in flight either lands (and is published) or is taken again from the host
    const bool landed = hipStreamSynchronize(st) == hipSuccess;
    for (int k = 0; k < W; k++) { if (used[k] && landed && !me.failed) me.done[pend[k]].store(1, std::memory_order_release); (void)hipEventDestroy(ev[k]); }
    if (!landed && broke < 0) broke = 0;
    (void)hipStreamDestroy(st);
    (void)hipGetLastError();
    return me.failed ? -1 : broke;
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
    P.wordLen = p->wordLen; P.maxHits = p->maxHits; P.bandWidth = p->bandWidth; P.maxGap = p->maxGap; P.maxIntron = p->maxIntron; P.minMatch = p->minMatch;
        P.maxDesert = p->maxDesert;
    P.minNonOverlap = p->minNonOverlap; P.minRawScore = p->minRawScore; P.minExtLength = p->minExtLength & 0xFF; P.GO = p->GOCost; P.GE = p->GECost; P.RC = p->RCost;
        P.MS = p->MScore; P.X = p->XCutoff;
    P.minIdentity = p->minIdentity; P.maxROff = ix->maxROff; P.totalMatches = ix->totalMatches;
    return 0;
}
// one device of ygpu_init_multi: image memory, then the copy (its own thread) beside streams / events / code object, then the bit table of seed.h
static void shareImage(ygpu_ctx *ctx, const ygpu_ctx *parent)
{
    ctx->P = parent->P;
    ctx->dBases.p = parent->dBases.p; ctx->dBases.cap = parent->dBases.cap; ctx->dSO.p = parent->dSO.p; ctx->dSO.cap = parent->dSO.cap; ctx->dROA.p = parent->dROA.p;
        ctx->dROA.cap = parent->dROA.cap; ctx->dLow.p = parent->dLow.p; ctx->dLow.cap = parent->dLow.cap;
    ctx->sharedIndex = true;
}
static int initDevice(ygpu_ctx *ctx, int device, int srcIndex /* -1: the host */, int srcDevice, const ygpu_index_view *ix, const ImagePlan &plan, ImageState *states, int self,
    std::atomic<int> *imageReady, ygpu_ctx **more, int nMore)
{
    // the device's further contexts (they share this one's image): streams, events and counters are made beside the copy as well
    std::vector<int> moreRc(nMore, 0); std::vector<std::thread> moreTh;
    for (int j = 0; j < nMore; j++) moreTh.emplace_back([&, j]() { ygpu_ctx *c = more[j]; int rc = initCommon(c, device); if (rc == 0 && (c->counters.ensure(4 * CNT_N)
        || c->ctr.ensure(sizeof(DevCounters)) || c->errFlag.ensure(64))) { c->err = "hipMalloc failed"; rc = YGPU_ENOMEM; } if (rc == 0 && hipMemsetAsync(c->counters.p, 0,
        4 * CNT_N, c->stream) != hipSuccess) { c->err = "hipMemset failed"; rc = YGPU_ENODEV; } moreRc[j] = rc; });
    struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{moreTh};
    const bool phases = getenv("YGPU_INIT_PHASES") != nullptr; double tPh = nowMs(); const double tPh0 = tPh;      // where a context's start-up goes
    auto phase = [&](const char *what) { if (phases) { const double t = nowMs(); fprintf(stderr, "[ygpu] init device %d: %-40s %8.1f ms\n", device, what, t - tPh); tPh = t; } };
    ImageState &me = states[self];
    auto giveUp = [&](int rc) { me.failed = 1; imageReady[self] = -1; return rc; };
    if (hipSetDevice(device) != hipSuccess) { ctx->err = "hipSetDevice failed"; (void)hipGetLastError(); return giveUp(YGPU_ENODEV); }
    const uint64_t HT = 1ull << (2 * ix->wordLen);
    if (ctx->dBases.ensureExact(ix->n_base_bytes + 4096) || ctx->dSO.ensureExact(4 * (HT + 1) + 64) || ctx->dROA.ensureExact(4ull * ix->totalMatches + 64)) {
        /* exact: a growth margin on 16.7 GB is 4 GB */ ctx->err = "hipMalloc failed for the index image"; (void)hipGetLastError(); return giveUp(YGPU_ENOMEM); }
    me.dst[0] = (char *)ctx->dBases.p; me.dst[1] = (char *)ctx->dSO.p; me.dst[2] = (char *)ctx->dROA.p;
    imageReady[self] = 1;                                                    // the memory is there: the next device in the chain may start asking for pieces
    // (slack behind the bases reads as 0xEE: the lane kernels load whole dwords around a window)
    if (hipMemset((char *)ctx->dBases.p + ix->n_base_bytes, 0xEE, ctx->dBases.cap - ix->n_base_bytes) != hipSuccess) { ctx->err = "hipMemset failed"; (void)hipGetLastError();
        return giveUp(YGPU_ENODEV); }
    phase("device memory for the image");
    // One thread, one plain copy per piece: the runtime pins the piece and the DMA engines read it in place -- 41-48 GB/s for the 16.7 GB index out of the page cache
    // (55 GB/s, the link's rate, for a file whose pages the kernel could keep in large folios; tools/micro/h2d_probe.hip).  Measured on the command line, 16.7 GB, runs
    // 6 s apart: one thread 404 ms, two 450-500, four 640; six threads staging through page-locked buffers of their own 590-630 (profiles/r04_index_upload.txt).
    bool staged = false; int nt = 1;                                         // YGPU_UPLOAD=direct:T | staged:T  (YGPU_UPLOAD_THREADS=T: the earlier spelling of direct:T)
    if (const char *e = getenv("YGPU_UPLOAD_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 32) nt = v; }
    if (const char *e = getenv("YGPU_UPLOAD")) { staged = strncmp(e, "staged", 6) == 0; const char *c = strchr(e, ':'); if (c) { const int v = atoi(c + 1);
        if (v >= 1 && v <= 32) nt = v; } }
    std::thread copier;
    if (srcIndex < 0) copier = std::thread([&, nt, staged]() { uploadFromHost(device, plan, me, nt, staged); });
    else copier = std::thread([&, nt, staged]() {
        // (YGPU_PEER_FAIL_AT=k or k:j -- piece k of the chain fails, on every chained device or on the j-th device of the call only: the fall-back's test)
        long failAt = -1; if (const char *e = getenv("YGPU_PEER_FAIL_AT")) { const char *c = strchr(e, ':'); if (!c || atoi(c + 1) == self) failAt = atol(e); }
        while (imageReady[srcIndex].load() == 0) std::this_thread::yield();
        long broke = imageReady[srcIndex].load() < 0 ? 0 : copyFromPeer(device, srcDevice, plan, me, states[srcIndex], failAt);
        if (broke >= 0 && !me.failed) {      // the chain broke: this device takes what it still lacks from the host (and goes on serving the device behind it)
            fprintf(stderr, "[ygpu] device %d: the copy of the index image from device %d broke at piece %ld of %zu; the rest comes from the host\n",
                    device, srcDevice, broke, plan.pieces.size());
            uploadFromHost(device, plan, me, nt, staged);
        }
    });
    int rc0 = initCommon(ctx, device);
    if (rc0 == 0) { if (ctx->counters.ensure(4 * CNT_N) || ctx->ctr.ensure(sizeof(DevCounters)) || ctx->errFlag.ensure(64) || ctx->dLow.ensure(ydLowTableBytes())) {
        ctx->err = "hipMalloc failed"; rc0 = YGPU_ENOMEM; } }
    if (rc0 == 0) {      // (the first launch loads the library's code object: ~20 ms that need not follow the image)
        if (ydFirstLaunch(ctx) || hipMemsetAsync(ctx->counters.p, 0, 4 * CNT_N, ctx->stream) != hipSuccess
            || hipMemsetAsync(ctx->dLow.p, 0, ydLowTableBytes(), ctx->stream) != hipSuccess || streamSync(ctx) != hipSuccess) {
            ctx->err = "first kernel launch failed"; (void)hipGetLastError(); rc0 = YGPU_ENODEV;
        }
    }
    phase("streams, events, code object (beside the copy)");
    copier.join();
    if (rc0) { me.failed = 1; return rc0; }
    if (me.failed) { ctx->err = srcIndex < 0 ? "copying the index image to the device failed" : "copying the index image from the neighbouring device failed"; return YGPU_ENODEV; }
    phase(srcIndex < 0 ? "image copied from the host" : "image copied from the device before");
    { const int rcL = ydLowOffsets(ctx, ix); if (rcL) return rcL; }
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
    auto notStarted = [&](int rc) {
        for (int k = 0; k < n; k++) {
            if (!rcs[k]) { rcs[k] = YGPU_EINVAL; out[k]->err = "not started: another device of the call failed"; }
            if (rc_each) rc_each[k] = rcs[k];
            for (int j = 1; j < cpd; j++) all[k * cpd + j]->err = out[k]->err;
        }
        return rc;
    };
    const double t0 = nowMs();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { for (int k = 0; k < n; k++) { rcs[k] = YGPU_ENODEV;
        out[k]->err = "no HIP device visible: the hot path needs an MI355X (there is no CPU fallback)"; } return notStarted(YGPU_ENODEV); }
    if (getenv("YGPU_INIT_PHASES")) fprintf(stderr, "[ygpu] init: runtime up (hipGetDeviceCount) %8.1f ms\n", nowMs() - t0);
    {
        bool bad = false;
        for (int k = 0; k < n; k++) if (devices[k] < 0 || devices[k] >= ndev) { rcs[k] = YGPU_ENODEV; out[k]->err = "device index out of range"; bad = true; }
        if (bad) return notStarted(YGPU_ENODEV);
    }
    { int bad = 0; for (int k = 0; k < n; k++) { rcs[k] = checkParams(out[k], ix, p); if (rcs[k]) bad = rcs[k]; } if (bad) return notStarted(bad); }
    ImagePlan plan; plan.build(ix, n > 1 ? (64ull << 20) : (1024ull << 20));
    std::vector<ImageState> states(n); for (auto &st : states) { st.done.reset(new std::atomic<int>[plan.pieces.size() + 1]);
        for (size_t i = 0; i <= plan.pieces.size(); i++) st.done[i] = 0; }
    std::unique_ptr<std::atomic<int>[]> ready(new std::atomic<int>[n]); for (int k = 0; k < n; k++) ready[k] = 0;
    // the chain: device k takes the image from device k - 1 when it can reach it (YGPU_PEER_COPY=0: every device from the host)
    std::vector<int> srcIndex(n, -1); const bool peer = !(getenv("YGPU_PEER_COPY") && atoi(getenv("YGPU_PEER_COPY")) == 0);
    for (int k = 1; k < n && peer; k++) {
        int can = devices[k] == devices[k - 1] ? 1 : 0;
        if (!can && hipDeviceCanAccessPeer(&can, devices[k], devices[k - 1]) != hipSuccess) { can = 0; (void)hipGetLastError(); }
        if (can && devices[k] != devices[k - 1]) { if (hipSetDevice(devices[k]) == hipSuccess) { const hipError_t e = hipDeviceEnablePeerAccess(devices[k - 1], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0; } else can = 0; (void)hipGetLastError(); }
        if (can) srcIndex[k] = k - 1;
    }
    if (getenv("YGPU_INIT_PHASES")) { fprintf(stderr, "[ygpu] init: image sources:");
        for (int k = 0; k < n; k++) { if (srcIndex[k] < 0) fprintf(stderr, " device %d <- host;", devices[k]);
        else fprintf(stderr, " device %d <- device %d;", devices[k], devices[srcIndex[k]]); } fprintf(stderr, " %zu pieces\n", plan.pieces.size()); }
    std::vector<std::thread> th;
    for (int k = 1; k < n; k++) th.emplace_back([&, k]() { rcs[k] = initDevice(out[k], devices[k], srcIndex[k], srcIndex[k] >= 0 ? devices[srcIndex[k]] : -1, ix, plan,
        states.data(), k, ready.get(), all + k * cpd + 1, cpd - 1); });
    rcs[0] = initDevice(out[0], devices[0], -1, -1, ix, plan, states.data(), 0, ready.get(), all + 1, cpd - 1);
    for (auto &x : th) x.join();
    int rc = 0; for (int k = 0; k < n; k++) { if (rc_each) rc_each[k] = rcs[k]; if (rcs[k] && !rc) rc = rcs[k];
        if (rcs[k]) for (int j = 1; j < cpd; j++) if (all[k * cpd + j]->err.empty()) all[k * cpd + j]->err = "the device's first context failed: " + out[k]->err; }
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
std::vector<DevBuf *> allBuffers(ygpu_ctx *ctx)
{
    DevBuf *all[] = {&ctx->dBases, &ctx->dSO, &ctx->dROA, &ctx->dLow, &ctx->dFwd, &ctx->dRev, &ctx->dFwd4, &ctx->dRev4, &ctx->dReadOff, &ctx->dKmerOff, &ctx->posS, &ctx->posC,
        &ctx->posRsI, &ctx->hitOff, &ctx->expandStart, &ctx->keysA, &ctx->keysB, &ctx->segOff, &ctx->bigB, &ctx->bigE, &ctx->isHead, &ctx->tileState,
                         &ctx->frags, &ctx->regStart, &ctx->multiList, &ctx->smallList, &ctx->bigList, &ctx->regionCount, &ctx->regionBase, &ctx->clumps, &ctx->clumpFrags,
                             &ctx->clumpFrags0, &ctx->order, &ctx->clumpsSorted, &ctx->rootPush, &ctx->rootBase, &ctx->outClumps,
                         &ctx->outClumps2, &ctx->outOps, &ctx->outRoot, &ctx->outPush, &ctx->dstIdx, &ctx->readCount, &ctx->readStart, &ctx->counters, &ctx->ctr, &ctx->errFlag,
                             &ctx->scanState, &ctx->bucketWork, &ctx->scratchAlign,
                         &ctx->segLists, &ctx->subB, &ctx->subE, &ctx->subLists, &ctx->subBigB, &ctx->subBigE, &ctx->sub2B, &ctx->sub2E, &ctx->sub2Lists, &ctx->sub3B, &ctx->sub3E,
                             &ctx->sub3Lists, &ctx->kmerParts, &ctx->scratchChain, &ctx->dpProbs, &ctx->dpRes, &ctx->dpOps, &ctx->rootState, &ctx->stateOps, &ctx->extProbs,
                             &ctx->rowsBound, &ctx->stripOff, &ctx->extRes, &ctx->extTrace, &ctx->chunkCnt, &ctx->memoKeys, &ctx->memoCount, &ctx->probs2, &ctx->rowsBound2,
                             &ctx->stripOff2, &ctx->extRes2, &ctx->extTrace2, &ctx->rowsClock, &ctx->splitScratch, &ctx->fallList, &ctx->keys2a, &ctx->keys2b, &ctx->vals2a,
                             &ctx->vals2b, &ctx->extKeys, &ctx->extVals, &ctx->extKeys2, &ctx->extOrder, &ctx->slowList, &ctx->gapScratch, &ctx->jointCount, &ctx->jointBase,
                             &ctx->joints, &ctx->sortKeys, &ctx->sortVals, &ctx->sortKeys2, &ctx->sortVals2, &ctx->gapOps, &ctx->waveChunks, &ctx->extOps, &ctx->traceCnt,
                         &ctx->oqCs, &ctx->oqCl, &ctx->oqOpsIn, &ctx->oqSeeds, &ctx->oqQlen, &ctx->pf.scanState, &ctx->pf.counters, &ctx->oqProf, &ctx->oqLists, &ctx->oqClsCnt,
                             &ctx->oqThr, &ctx->oqSeqStart, &ctx->oqSeqLen, &ctx->oqNeed, &ctx->oqPoolOff, &ctx->oqKeys, &ctx->oqStack, &ctx->oqNodes, &ctx->oqPrim, &ctx->oqPA,
                             &ctx->oqPfx, &ctx->oqPath, &ctx->oqPool, &ctx->oqPush, &ctx->oqOut, &ctx->oqOutCnt, &ctx->oqOutOps, &ctx->oqPrimCnt, &ctx->oqOutStart,
                             &ctx->oqOpsStart, &ctx->oqFClumps, &ctx->oqFOps};
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
        {   // the device's ring of rows-launch events was recorded on streams of its contexts: with the last of them it goes too (a later context starts a fresh one)
            const int dv = ctx->device & 63; std::lock_guard<std::mutex> lk(gRowsMu[dv]);
            (void)hipStreamSynchronize(ctx->stream);
            if (gCtxPerDevice[dv].load() <= 0) for (int k = 0; k < 4; k++) if (gRowsEvValid[dv][k]) { (void)hipEventDestroy(gRowsEv[dv][k]); gRowsEvValid[dv][k] = false; }
        }
        if (ctx->sharedIndex) { ctx->dBases.p = nullptr; ctx->dBases.cap = 0; ctx->dSO.p = nullptr; ctx->dSO.cap = 0; ctx->dROA.p = nullptr; ctx->dROA.cap = 0;
            ctx->dLow.p = nullptr; ctx->dLow.cap = 0; }
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
    uint64_t mine = 0; for (DevBuf *b : allBuffers(ctx)) if (b->p && !(ctx->sharedIndex && (b == &ctx->dBases || b == &ctx->dSO || b == &ctx->dROA
        || b == &ctx->dLow))) mine += b->cap;
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
    for (size_t k = 0; k < all.size(); k++) { DevBuf *b = all[k];
        out->cap[k] = (b == &ctx->dBases || b == &ctx->dSO || b == &ctx->dROA || b == &ctx->dLow) ? 0ull : (uint64_t)b->cap; }
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
        if (b == &ctx->dBases || b == &ctx->dSO || b == &ctx->dROA || b == &ctx->dLow || b == &ctx->counters || b == &ctx->ctr || b == &ctx->errFlag
            || b == &ctx->pf.counters) continue;
        if (prof->cap[k] > b->cap && b->ensureExact((size_t)prof->cap[k])) { (void)hipGetLastError(); ctx->err = "hipMalloc failed while presizing the arenas"; return YGPU_ENOMEM;
            }
    }
    // (work words that their kernels expect zeroed when they are made)
    if (ctx->scanState.p) HIPCHK(hipMemsetAsync(ctx->scanState.p, 0, ctx->scanState.cap, ctx->stream));
    if (ctx->bucketWork.p) HIPCHK(hipMemsetAsync(ctx->bucketWork.p, 0, ctx->bucketWork.cap, ctx->stream));
    if (ctx->pf.scanState.p) HIPCHK(hipMemsetAsync(ctx->pf.scanState.p, 0, ctx->pf.scanState.cap, ctx->stream));      // (the post-filter side's look-back words: the same rule)
    if (ctx->runsDone == 0) { ctx->traceRatio = prof->trace_ratio; ctx->opsRatio = prof->ops_ratio > 0 ? prof->ops_ratio : ctx->opsRatio;
        ctx->lastClumpSlots = prof->last_clump_slots; ctx->lastFall = (long long)prof->last_fall; }
    HIPCHK(streamSync(ctx));
    return 0;
}
const char *ygpu_last_error(const ygpu_ctx *ctx) { return !ctx ? "null context" : (tlsPfFailed == ctx && !ctx->pf.err.empty()) ? ctx->pf.err.c_str() : ctx->err.c_str(); }

} // extern "C"
extern "C" {

int ygpu_run(ygpu_ctx *ctx)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    ctx->stageDone = 0; const double t0 = nowMs(); ctx->statAttempts = 0; ctx->statRanges = 0;
    struct Active { int d; explicit Active(int dv) : d(dv) { gActiveRuns[d]++; } ~Active() { gActiveRuns[d]--; } } active(ctx->device & 63);
    int rc = runTo(ctx, 3);
    if (kStats) { size_t fb = 0, tb = 0; hipMemGetInfo(&fb, &tb);
        fprintf(stderr, "[ygpu] ctx %p run: %u reads, rc %d, %.1f ms; align attempts %d, ranges %d, trace arena %.2f GB (ratio %.3f), free %.1f GB\n", (void *)ctx, ctx->nReads, rc,
        nowMs() - t0, ctx->statAttempts, ctx->statRanges, ctx->extTrace.cap / 1e9, ctx->traceRatio, fb / 1e9); }
    // (a failed run may leave the batch's host-to-device copies queued -- ygpu_upload_nowait -- and the caller recycles the batch's memory next: drained here)
    if (rc) { (void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError(); return rc; }
    if (ydCheckStateOn()) {                                                  // (debug switch, on in the whole GPU tier: prims.hip)
        if (getenv("YGPU_CHECK_STATE_INJECT") && ctx->scanState.p) (void)hipMemsetAsync((char *)ctx->scanState.p + 8, 0x01, 4, ctx->stream);      // (tests: a word left dirty)
        const DevBuf *const bufs[2] = {&ctx->scanState, &ctx->bucketWork}; static const char *const names[2] = {"the sums' look-back state", "the orderings' bucket counters"};
        rc = ydCheckZero(ctx->stream, ctx->err, (unsigned int *)ctx->errFlag.p + 2, bufs, names, 2, "after ygpu_run");
        if (rc) { if (ctx->scanState.p) (void)hipMemsetAsync(ctx->scanState.p, 0, ctx->scanState.cap, ctx->stream); return rc; }
    }
    ctx->runsDone++;
    ctx->totalMs = 0;
    for (int t = 0; t < T_N; t++) {
        float m = 0; ctx->ms[t] = (ctx->evUsed[t] && hipEventElapsedTime(&m, ctx->ev[t][0], ctx->ev[t][1]) == hipSuccess) ? m : 0;
        // 100 MHz ticks -> ms
        if (t == T_XROWS_DEV) ctx->ms[t] = (ctx->evUsed[T_XROWS] && ctx->hRowsClock[1] > ctx->hRowsClock[0] && ctx->hRowsClock[0] != ~0ull)
            ? (float)((double)(ctx->hRowsClock[1] - ctx->hRowsClock[0]) / 1.0e5) : 0;
        if (t == T_XROWS_PK) ctx->ms[t] = ctx->rowsPacked ? 1.0f : 0.0f;
        if (t < T_TOP) ctx->totalMs += ctx->ms[t];
    }
    return 0;
}
void *ygpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void ygpu_host_free(void *p) { if (p) (void)hipHostFree(p); }
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
    auto fail = [&](const char *what, uint64_t at) { char m[160];
        snprintf(m, sizeof m, "selftest: %s differs from the host at element %llu of %u", what, (unsigned long long)at, n); ctx->err = m; return YGPU_EINTERNAL; };
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
      for (uint32_t i = 0; i < n; i++) { const uint32_t v = o32[i] - 7u; if (v >= n || seen[v]) return fail("the ordering (not a permutation)", i); seen[v] = 1;
          const uint32_t bk = std::min((h32[v] - sub) >> shift, nb - 1u); if (bk < last) return fail("the ordering (buckets not ascending)", i); last = bk; } }
    // and once more right away: the work words of both must have cleaned themselves up
    rc = bucketOrder(ctx, a.as<uint32_t>(), nullptr, 7u, n, sub, shift, nb, b.as<uint32_t>(), ctx->stream); if (rc) return rc;
    std::vector<uint32_t> again(n); HIPCHK(hipMemcpyAsync(again.data(), b.p, 4ull * n, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
    { uint32_t last = 0; for (uint32_t i = 0; i < n; i++) { const uint32_t v = again[i] - 7u; if (v >= n) return fail("the second ordering", i);
        const uint32_t bk = std::min((h32[v] - sub) >> shift, nb - 1u); if (bk < last) return fail("the second ordering (buckets not ascending)", i); last = bk; } }
    uint32_t sf = 0; rc = fetchU32(ctx, ctx->counters.as<uint32_t>() + CNT_SCANFAIL, &sf); if (rc) return rc;
    if (sf) { ctx->err = "selftest: a look-back gave up"; return YGPU_EINTERNAL; }
    rc = ydSelftestSegSort(ctx, x); if (rc) return rc;
    return ydSelftestWaveSort(ctx, seed, x);
}
}  // extern "C"
