// stage_out.hip -- what leaves the device: the batch's results as ygpu_run left them (ygpu_collect*), and the post-filter stage (postFilterBySimilarity,
// GraphPath.cpp:897-1086, Query.c:450) on a snapshot of them -- oqc_stage.h -- with its own stream, wait slot and look-back words (PfSide).
#include "ctx.h"
#include "oqc_stage.h"

extern "C" {
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

// ---- post-filter on the device (oqc_stage.h; reference GraphPath.cpp:897-1086) ----------------------------------------------------------------------
int ygpu_set_postfilter(ygpu_ctx *ctx, const ygpu_postfilter_params *p)
{
    if (!ctx || !ctx->stream || !p) return YGPU_EINVAL;
    if (p->bppN < 0 || p->bppN > 65536 || (p->bppN && !p->bppThr) || (p->n_seqs && (!p->seq_start || !p->seq_length))) {
        ctx->err = "ygpu_set_postfilter: bad break point table or sequence table"; return YGPU_EINVAL; }
    // (the wave's successor relaxation writes node j > i only while it reads node i: with a non-overlap requirement below one base a node could be its own successor, oqc_stage.h)
    if (p->minNonOverlap < 1) { ctx->err = "ygpu_set_postfilter: minNonOverlap (-MNO) must be at least 1 for the device stage; use the host filter"; return YGPU_EINVAL; }
    HIPCHK(hipSetDevice(ctx->device));
    ENSURE(ctx->oqThr, 4ull * (p->bppN + 1)); ENSURE(ctx->oqSeqStart, 4ull * (p->n_seqs + 1)); ENSURE(ctx->oqSeqLen, 4ull * (p->n_seqs + 1));
    if (p->bppN) HIPCHK(hipMemcpyAsync(ctx->oqThr.p, p->bppThr, 4ull * p->bppN, hipMemcpyHostToDevice, ctx->stream));
    if (p->n_seqs) { HIPCHK(hipMemcpyAsync(ctx->oqSeqStart.p, p->seq_start, 4ull * p->n_seqs, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->oqSeqLen.p, p->seq_length, 4ull * p->n_seqs, hipMemcpyHostToDevice, ctx->stream)); }
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
    ENSURE(ctx->oqCs, 4ull * (n + 2)); ENSURE(ctx->oqCl, sizeof(ygpu_clump) * ((uint64_t)C + 1)); ENSURE(ctx->oqOpsIn, 4ull * ((uint64_t)O + 1));
        ENSURE(ctx->oqSeeds, 20ull * (n + 1)); ENSURE(ctx->oqQlen, 4ull * (n + 1));
    HIPCHK(hipMemcpyAsync(ctx->oqCs.p, ctx->readStart.p, 4ull * (n + 1), hipMemcpyDeviceToDevice, ctx->stream));
    if (C) HIPCHK(hipMemcpyAsync(ctx->oqCl.p, ctx->outClumps2.p, sizeof(ygpu_clump) * (uint64_t)C, hipMemcpyDeviceToDevice, ctx->stream));
    if (O) HIPCHK(hipMemcpyAsync(ctx->oqOpsIn.p, ctx->outOps.p, 4ull * O, hipMemcpyDeviceToDevice, ctx->stream));
    if (n) KL(k_oqc_seeds, dim3(gridFor(n, 256)), dim3(256), 0, ctx->stream, ctx->dFwd.as<uint8_t>(), ctx->dReadOff.as<uint32_t>(), n, ctx->oqSeeds.as<uint32_t>(),
        ctx->oqQlen.as<uint32_t>());
    // (no wait here: the context's thread goes straight on to its next batch -- whatever it queues on this stream follows the copies -- and the work counters land
    // in a pinned slot the filter's side reads after its own first wait; without the slot, a wait it is)
    if (ctx->snapCtr) HIPCHK(hipMemcpyAsync(ctx->snapCtr, ctx->ctr.p, sizeof(DevCounters), hipMemcpyDeviceToHost, ctx->stream));
    else { DevCounters dc; HIPCHK(hipMemcpyAsync(&dc, ctx->ctr.p, sizeof dc, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        memcpy(&ctx->snapCtrPlain, &dc, sizeof dc); }
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
    int rc = postfilterBody(full);
    if (rc == 0 && ydCheckStateOn()) {                                       // (debug switch: the post-filter side's own look-back words)
        const DevBuf *const bufs[1] = {&full->pf.scanState}; static const char *const names[1] = {"the post-filter side's look-back state"};
        if (full->oqClsCnt.ensure(64)) { full->pf.err = "hipMalloc failed"; rc = YGPU_ENOMEM; }
        else rc = ydCheckZero(full->pf.stream, full->pf.err, (unsigned int *)full->oqClsCnt.p + 8, bufs, names, 1, "after ygpu_postfilter");
        if (rc) full->oqDone = false;
    }
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
    ENSURE(full->oqPrim, sizeof(yoqc::CNode) * (uint64_t)C); ENSURE(full->oqPA, sizeof(yoqc::PAttr) * (uint64_t)C); ENSURE(full->oqPush, sizeof(yoqc::OutRec) * (uint64_t)C);
        ENSURE(full->oqOut, sizeof(yoqc::OutRec) * (uint64_t)C);
    ENSURE(full->oqOutCnt, 4ull * (n + 2)); ENSURE(full->oqOutOps, 4ull * (n + 2)); ENSURE(full->oqPrimCnt, 4ull * (n + 2));
    HIPCHK(hipMemsetAsync((uint32_t *)full->oqOutCnt.p + n, 0, 8, ctx->stream)); HIPCHK(hipMemsetAsync((uint32_t *)full->oqOutOps.p + n, 0, 8, ctx->stream));
        HIPCHK(hipMemsetAsync(full->oqClsCnt.p, 0, 64, ctx->stream));
    OqcArgs A; A.P = full->oqP; A.G = full->oqG; A.cs = full->oqCs.as<uint32_t>(); A.cl = full->oqCl.as<ygpu_clump>(); A.ops = full->oqOpsIn.as<uint32_t>();
        A.seeds = full->oqSeeds.as<uint32_t>(); A.qlen = full->oqQlen.as<uint32_t>(); A.nReads = n;
    A.poolOff = full->oqPoolOff.as<unsigned long long>(); A.prim = full->oqPrim.as<yoqc::CNode>(); A.pa = full->oqPA.as<yoqc::PAttr>(); A.push = full->oqPush.as<yoqc::OutRec>();
        A.out = full->oqOut.as<yoqc::OutRec>();
    A.outCnt = full->oqOutCnt.as<uint32_t>(); A.outOpsCnt = full->oqOutOps.as<uint32_t>(); A.primCnt = full->oqPrimCnt.as<uint32_t>();
    A.keys = nullptr; A.stack = nullptr; A.nodes = nullptr; A.pfxOff = nullptr; A.path = nullptr; A.pool = nullptr; A.prof = nullptr;
    // (read at every call: tests lower it to send small reads down the hand-over path)
    { const char *e = getenv("YGPU_OQC_MAX"); const int v = e ? atoi(e) : YQ_DEVICE_MAX; A.devMax = v >= 1 && v < YQ_DEVICE_MAX ? v : YQ_DEVICE_MAX; }
    { const char *e = getenv("YGPU_OQC_HBM"); A.graphInHbm = e && atoi(e) ? 1 : 0; }      // (read at every call, as YGPU_OQC_MAX)
    static const bool oqProf = getenv("YGPU_OQC_PROF") != nullptr;
    if (oqProf) { ENSURE(full->oqProf, 8ull * 32 * YQ_NCLASS); HIPCHK(hipMemsetAsync(full->oqProf.p, 0, 8ull * 32 * YQ_NCLASS, ctx->stream));
        A.prof = full->oqProf.as<unsigned long long>(); }
    uint32_t *lists = full->oqLists.as<uint32_t>();
    KL(k_oqc_classify, dim3(gridFor(n + 1, 256)), dim3(256), 0, ctx->stream, A, full->oqNeed.as<unsigned long long>(), lists, full->oqClsCnt.as<unsigned int>());
    int rc = cubScan64(ctx, full->oqNeed.as<unsigned long long>(), full->oqPoolOff.as<unsigned long long>(), n + 1); if (rc) return rc;
    unsigned long long poolInts = 0; uint32_t nCls[YQ_NCLASS] = {0, 0, 0, 0, 0};
    { uint32_t w[2] = {0, 0}; const FetchPiece pc[2] = {{full->oqPoolOff.as<unsigned long long>() + n, w, 2}, {full->oqClsCnt.p, nCls, YQ_NCLASS}};
      rc = fetchMany(ctx, pc, 2); if (rc) return rc; poolInts = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32); }
    takeCounters();
    ENSURE(full->oqPool, 4ull * (poolInts + 16)); A.pool = full->oqPool.as<int>();
    // work space of the reads in HBM: what a wave's LDS does not hold (the survivors' keys while the nodes are made; everything for the reads of the last class)
    ENSURE(full->oqKeys, sizeof(yoqc::SortKey) * (uint64_t)C); ENSURE(full->oqStack, 4ull * (4ull * C + 8ull * n + 16)); ENSURE(full->oqNodes, sizeof(yoqc::CNode) * (uint64_t)C);
        ENSURE(full->oqPfx, 4ull * C); ENSURE(full->oqPath, 4ull * C);
    A.keys = full->oqKeys.as<yoqc::SortKey>(); A.stack = full->oqStack.as<int>(); A.nodes = full->oqNodes.as<yoqc::CNode>(); A.pfxOff = full->oqPfx.as<int>();
        A.path = full->oqPath.as<int>();
    // the classes: clumps a read may have -> LDS of its workgroup; ints of LDS pool (the first tables; later ones go to the read's slice of the HBM pool)
    static const int capN[YQ_NCLASS] = {112, 224, 448, YQ_DEVICE_MAX, 0};
    // A wave a read, every read of a class resident at once: a launch lasts as long as its slowest read (one of 400 clumps with 280 survivors: 3 ms), and the classes
    // follow one another on the post-filter's one stream.  That latency is off the context's path -- the next batch is running meanwhile -- and a stream of its own for
    // every class is not worth having: streams share four hardware queues, and more than two a context put all contexts' main streams on one (profiles/r05_hw_queues.txt).
    // left to the host, marked
    if (nCls[YQ_NCLASS - 1]) KL(k_oqc_raw, dim3(gridFor(nCls[YQ_NCLASS - 1], 64)), dim3(64), 0, ctx->stream, A, lists + (size_t)(YQ_NCLASS - 1) * n, nCls[YQ_NCLASS - 1]);
    for (int c = YQ_NCLASS - 2; c >= 0; c--) if (nCls[c]) {
        const unsigned lds = std::min(YQ_LDS_MAX, oqcLdsBytes(capN[c]));
        KL(k_oqc_wave, dim3(nCls[c]), dim3(64), lds, ctx->stream, A, lists + (size_t)c * n, nCls[c], lds);
    }
    if (kTrace) fprintf(stderr,
        "[ygpu] post-filter: %u reads with two or more clumps in classes of <= 112 / 224 / 448 / %d clumps: %u / %u / %u / %u, left to the host %u; pool %.1f MB\n",
        nCls[0] + nCls[1] + nCls[2] + nCls[3] + nCls[4], YQ_DEVICE_MAX, nCls[0], nCls[1], nCls[2], nCls[3], nCls[4], poolInts * 4.0 / 1e6);
    rc = cubScan(ctx, full->oqOutCnt.as<uint32_t>(), full->oqOutStart.as<uint32_t>(), n + 1); if (rc) return rc;
    rc = cubScan(ctx, full->oqOutOps.as<uint32_t>(), full->oqOpsStart.as<uint32_t>(), n + 1); if (rc) return rc;
    uint32_t tot[2] = {0, 0}, scanFail = 0;
    { const FetchPiece pc[3] = {{full->oqOutStart.as<uint32_t>() + n, &tot[0], 1}, {full->oqOpsStart.as<uint32_t>() + n, &tot[1], 1}, {ctx->counters.as<uint32_t>() + CNT_SCANFAIL,
        &scanFail, 1}}; rc = fetchMany(ctx, pc, 3); if (rc) return rc; }
    if (scanFail) {      // (not sticky: the flag and the look-back words -- stale tickets and statuses -- are made clean again for the next batch, as runTo does on its side)
        HIPCHK(hipMemsetAsync(ctx->counters.as<uint32_t>() + CNT_SCANFAIL, 0, 4, ctx->stream));
        if (ctx->scanState.p) HIPCHK(hipMemsetAsync(ctx->scanState.p, 0, ctx->scanState.cap, ctx->stream));
        HIPCHK(streamSync(ctx));
        ctx->err = "post-filter: a look-back of an exclusive sum gave up"; return YGPU_EINTERNAL;
    }
    full->nFOut = tot[0]; full->nFOps = tot[1];
    ENSURE(full->oqFClumps, sizeof(ygpu_out_clump) * ((uint64_t)tot[0] + 1)); ENSURE(full->oqFOps, 4ull * ((uint64_t)tot[1] + 1));
    KL(k_oqc_gather, dim3(gridFor((uint64_t)n * 64, 256)), dim3(256), 0, ctx->stream, A, full->oqOutStart.as<uint32_t>(), full->oqOpsStart.as<uint32_t>(),
        full->oqFClumps.as<ygpu_out_clump>(), full->oqFOps.as<uint32_t>());
    if (oqProf) {
        unsigned long long h[32 * YQ_NCLASS]; HIPCHK(hipMemcpyAsync(h, full->oqProf.p, sizeof h, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        static const char *nm[7] = {"keys", "sort", "dup scan", "nodes+tables", "path walk", "successors", "finish"};
        for (int c = 0; c < YQ_NCLASS; c++) if (h[32 * c + 7]) {
            const unsigned long long *q = h + 32 * c;
                fprintf(stderr, "[ygpu] post-filter class %d: %llu reads, %.0f clumps, %.0f survivors a read; us a read (largest of any read):", c, q[7], (double)q[8] / q[7],
                (double)q[9] / q[7]);
            for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.1f (%.0f)", nm[k], q[k] / 100.0 / q[7], q[16 + k] / 100.0);
            fprintf(stderr, "; slowest read %.0f us: %llu clumps, %llu survivors\n", (q[10] >> 24) / 100.0, (q[10] >> 12) & 4095ull, q[10] & 4095ull);
        }
    }
    full->oqDone = true;
    return 0;
}
int ygpu_inject_results(ygpu_ctx *ctx, const ygpu_result_batch *r)
{
    if (!ctx || !ctx->stream || !r || r->n_reads != ctx->nReads || !r->clump_start || (r->n_clumps && !r->clumps) || (r->n_ops && !r->ops) || r->n_clumps > 0x7FFFFFF0ull
        || r->n_ops > 0x7FFFFFF0ull) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->nReads;
    ENSURE(ctx->readStart, 4ull * (n + 1)); ENSURE(ctx->outClumps2, sizeof(ygpu_clump) * (r->n_clumps + 1)); ENSURE(ctx->outOps, 4ull * (r->n_ops + 1));
        ENSURE(ctx->ctr, sizeof(DevCounters));
    HIPCHK(hipMemcpyAsync(ctx->readStart.p, r->clump_start, 4ull * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (r->n_clumps) HIPCHK(hipMemcpyAsync(ctx->outClumps2.p, r->clumps, sizeof(ygpu_clump) * r->n_clumps, hipMemcpyHostToDevice, ctx->stream));
    if (r->n_ops) HIPCHK(hipMemcpyAsync(ctx->outOps.p, r->ops, 4ull * r->n_ops, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(streamSync(ctx));
    ctx->nOut = (uint32_t)r->n_clumps; ctx->nOutOps = (uint32_t)r->n_ops; ctx->stageDone = 3;
    return 0;
}
int ygpu_postfilter_drop(ygpu_ctx *ctx)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    ctx->pfSnap.store(false); ctx->oqDone = false; ctx->nFOut = ctx->nFOps = 0;
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
    // (collected: a later ygpu_filtered_size without a new ygpu_postfilter is an error, not the previous batch once more)
    full->oqDone = false;
    out->n_reads = n; out->clump_start = clump_start; out->clumps = clumps; out->ops = ops; out->n_clumps = full->nFOut; out->n_ops = full->nFOps; out->counters = full->pfCounters;
    return 0;
}
}  // extern "C"

// the post-filter's sort on the wave (oqc_stage.h waveSort) against the one-thread routine it stands for (oqc_core.h sortRange, the reference's quicksort with its
// random tie breaks): arrays of 2 .. YQ_DEVICE_MAX entries around the 64-lane edges, keys from 2 to 4 096 distinct values (ties by the hundred down to none), random,
// ascending and descending; every entry must land where the routine puts it
int ydSelftestWaveSort(ygpu_ctx *ctx, uint32_t seed, uint64_t &x)
{
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    DevBuf a; struct Rel { DevBuf &a; ~Rel() { a.release(); } } rel{a};
    {
        static const int fixedLen[] = {2, 3, 4, 5, 9, 17, 33, 63, 64, 65, 66, 100, 127, 128, 129, 130, 191, 192, 193, 300, 448, 449, 700, 1000, 1500, YQ_DEVICE_MAX};
        const uint32_t nArr = 56; std::vector<uint32_t> off(nArr + 1, 0), seeds(5 * nArr); std::vector<uint64_t> ent;
        for (uint32_t t = 0; t < nArr; t++) {
            const int len = t < sizeof fixedLen / sizeof fixedLen[0] ? fixedLen[t] : 2 + (int)(rnd() % (YQ_DEVICE_MAX - 1));
            const uint64_t span = 1ull << (1 + (seed * 7u + t) % 12u); const int shape = (int)(rnd() % 5);
            for (int i = 0; i < len; i++) { uint64_t k = rnd() % span; if (shape == 3) k = (uint64_t)i * span / len; if (shape == 4) k = (uint64_t)(len - 1 - i) * span / len;
                ent.push_back((k << 16) | (uint64_t)i); }
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
        if (a.ensure(8ull * ent.size()) || dOff.ensure(4ull * off.size()) || dSeeds.ensure(4ull * seeds.size()) || dStack.ensure(4ull * (2ull * ent.size() + 8ull * nArr + 16))) {
            ctx->err = "hipMalloc failed"; return YGPU_ENOMEM; }
        HIPCHK(hipMemcpyAsync(a.p, ent.data(), 8ull * ent.size(), hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(dOff.p, off.data(), 4ull * off.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(dSeeds.p, seeds.data(), 4ull * seeds.size(), hipMemcpyHostToDevice, ctx->stream));
        KL(k_oqc_sort_test, dim3(nArr), dim3(64), 4u * YQ_STACK_LDS + 20u * YQ_DEVICE_MAX, ctx->stream, a.as<uint64_t>(), dOff.as<uint32_t>(), dSeeds.as<uint32_t>(),
            dStack.as<int>(), nArr);
        HIPCHK(hipMemcpyAsync(got.data(), a.p, 8ull * ent.size(), hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        for (uint32_t t = 0; t < nArr; t++) for (uint32_t i = off[t]; i < off[t + 1]; i++) if (got[i] != want[i]) {
            char m[200]; snprintf(m, sizeof m, "selftest: the sort on the wave differs from the one-thread routine: array %u (%u entries), position %u", t, off[t + 1] - off[t],
                i - off[t]); ctx->err = m; return YGPU_EINTERNAL; }
    }
    return 0;
}
