// stage_align.hip -- stages A5..A8 of the hot path and A10's layout: gap fills between chained fragments, the two X-drop extensions of every root one problem per
// lane (k_ext_rows_pk / k_ext_rows + tracebacks), scoreClump / splitClump with the careful-extension round (AlignHelpers.c:205-579, AlignExtFrag.cpp:30-234,
// SW.cpp:462-1208); the output order (QueryMatch.c:306-331); and the stage-level DP entry ygpu_dp_batch[_ex] over the same kernels.
#include "ctx.h"
#include "phase_lanes.h"
#include "gap_band_lanes.h"
#include "gap_band_pk.h"
#include "ext_lanes_pk.h"
#include "split_lanes.h"
#include "dp_stage.h"
#include "align_kernels.h"
#include "layout.h"

// Per-wave scratch of the wave kernels, sized from the parameters.  A gap fill between two chained fragments has min(qGap, rGap) <= maxDesert
// and |qGap - rGap| <= maxGap (GraphPath.cpp:211-230), so its strip is at most MD + G + 2*BW + 3 columns wide (banded: 2*BW + 1 + |qGap - rGap|,
// full: rGap + 1) and rows x width <= (MD + 2) * (MD + G + 2*BW + 3) cells; the extensions need (maxQ + 2) rows of 64 cells.
// X-drop extensions in packed 16-bit arithmetic (ext_lanes_pk.h) when every score fits with room for the sentinel
static bool extRowsPacked(const ygpu_ctx *ctx, bool caps)
{
    const bool force32 = getenv("YGPU_EXT32") != nullptr;            // (read at every call: the tests run both kernel families in one process)
    const DevParams &P = ctx->P;
    return !force32 && !caps && P.MS >= 0 && (long long)P.MS * std::max(1, ctx->maxQ) <= 15000 && P.RC >= 0 && P.GO >= 0 && P.GE >= 0 && P.X >= 0
        && (long long)P.RC + P.X + P.GO + 21ll * P.GE <= 4000;
}

// banded gap fills in packed 16-bit arithmetic (gap_band_pk.h) when every score of a joint of at most 64 bases fits with room for the sentinel; YGPU_EXT32=1 forces the
// 32-bit kernels here as it does for the extensions
static bool gapBandPacked(const ygpu_ctx *ctx)
{
    const DevParams &P = ctx->P;
    return getenv("YGPU_EXT32") == nullptr && getenv("YGPU_GAP32") == nullptr && P.MS >= 0 && P.RC >= 0 && P.GO >= 0 && P.GE >= 0 && 64ll * std::max(P.MS,
        P.RC) + P.GO + 64ll * P.GE <= 12000;
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
        X.p3Order = nullptr; X.band24 = 0;
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
    ENSURE(ctx->joints, sizeof(JointRec) * (uint64_t)(J + 1)); ENSURE(ctx->sortKeys, 4ull * (J + 1)); ENSURE(ctx->sortVals, 4ull * (J + 1)); ENSURE(ctx->sortKeys2, 4ull * (J + 1));
        ENSURE(ctx->sortVals2, 4ull * (J + 1));
    ENSURE(ctx->gapOps, 4ull * gapOpsCap); ENSURE(ctx->slowList, 4ull * (std::max(NC, J) + 1));
    X.slowList = ctx->slowList.as<uint32_t>();
    X.joints = ctx->joints.as<JointRec>(); X.nJoints = J; X.sortKeys = ctx->sortKeys.as<uint32_t>(); X.sortVals = ctx->sortVals.as<uint32_t>();
        X.sortedVals = ctx->sortVals2.as<uint32_t>();
    X.nDP = cnt + CNT_NDP; X.nDPb = cnt + CNT_NB12; X.gapOps = ctx->gapOps.as<uint32_t>(); X.gapOpsCount = cnt + CNT_GAPOPS; X.gapOpsCap = gapOpsCap;
    HIPCHK(hipMemsetAsync(cnt + CNT_NDP, 0, 12, ctx->stream)); HIPCHK(hipMemsetAsync(cnt + CNT_NB12, 0, 12, ctx->stream));      // ndp, ndp16, gapops; nb12, nb16, nb24
    if (J) {
        KL(k_p1_joints, dim3(gridFor(NC, 256)), dim3(256), 0, ctx->stream, A, X);
        // joints of one (class, width, rows / 2) together
        rc = bucketOrder(ctx, X.sortKeys, X.sortVals, 0, J, 0, 0, 1u << YD_JKEY_BITS, ctx->sortVals2.as<uint32_t>(), ctx->stream); if (rc) return rc;
        // 17 / 26 KB of LDS per 64-thread block
        const unsigned gBlocks16 = (unsigned)std::min<uint64_t>(gridFor(J, 64), (uint64_t)ctx->nCU * 9), gBlocks32 = (unsigned)std::min<uint64_t>(gridFor(J, 64),
            (uint64_t)ctx->nCU * 6);
        ENSURE(ctx->gapScratch, (size_t)YD_GAP_SCRATCH * 64 * std::max(gBlocks16, gBlocks32)); X.gapScratch = ctx->gapScratch.as<uint8_t>();
        X.band24 = (gapBandPacked(ctx) && !getenv("YGPU_GAP24_OFF")) ? 1u : 0u;
        if (gapBandPacked(ctx)) {
            KL(k_gap_band_pk<12>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X); KL(k_gap_band_pk<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
            if (X.band24) KL(k_gap_band_pk<24>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        }
        else { KL(k_gap_band<12>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X); KL(k_gap_band<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X); }
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
    if (kTrace) { uint32_t v[3] = {0, 0, 0}; fetchU32(ctx, cnt + CNT_SLOW, &v[0]); uint32_t w[3] = {0, 0, 0}; fetchU32(ctx, cnt + CNT_NDP, w, 3); v[1] = w[0]; v[2] = w[2];
        fprintf(stderr, "[ygpu] roots %u, joints %u, DP joints %u (W<=16: %u, wave fallback %u), gap ops %u\n", NC, J, v[1], w[1], v[0], v[2]); }
    if (kTrace && getenv("YGPU_JOINT_HIST")) {      // diagnostics: the DP joints by kernel class and shape (which kernel takes what: gapJointKey)
        std::vector<JointRec> hj(J); hipMemcpy(hj.data(), ctx->joints.p, sizeof(JointRec) * (size_t)J, hipMemcpyDeviceToHost);
        unsigned long long nj[4] = {0, 0, 0, 0}, cj[4] = {0, 0, 0, 0}, b3 = 0, b3lim = 0, c3lim = 0, f3 = 0, wh[5] = {0, 0, 0, 0, 0};
        for (auto &j : hj) if (j.kind == JK_DP) {
            const bool banded = (j.flags & 2u) != 0; const int q = j.qGap, r = j.rGap, ld = q > r ? q - r : r - q, W = banded ? 2 * ctx->P.bandWidth + ld + 1 : r + 1;
            const uint32_t cls = gapJointClass(gapJointKey(ctx->P, banded, q, r)); nj[cls]++; cj[cls] += (unsigned long long)q * (unsigned long long)std::min(W, r + 1);
            if (cls == 3) { if (banded) { b3++; if (W <= 32 && q <= YD_GROWS - 1 && r <= YD_GREF) { b3lim++; c3lim += (unsigned long long)q * W; } } else f3++;
                wh[std::min(4, (W - 17) / 8)]++; }
        }
        fprintf(stderr, "[ygpu] DP joints by class (band <= 12 / band <= 16 / other W <= 16 / the rest): %llu %llu %llu %llu; strip cells %llu %llu %llu %llu; "
                        "the rest: banded %llu (W <= 32 within the band kernels' limits: %llu, %llu cells), full %llu; "
                        "W 17-24 / 25-32 / 33-40 / 41-48 / more: %llu %llu %llu %llu %llu\n",
                nj[0], nj[1], nj[2], nj[3], cj[0], cj[1], cj[2], cj[3], b3, b3lim, c3lim, f3, wh[0], wh[1], wh[2], wh[3], wh[4]);
    }
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
    ExtArgs E; E.P = ctx->P; E.bases = ctx->dBases.as<uint8_t>(); E.fwd = ctx->dFwd.as<uint8_t>(); E.rev = ctx->dRev.as<uint8_t>(); E.fwd4 = ctx->dFwd4.as<uint8_t>();
        E.rev4 = ctx->dRev4.as<uint8_t>();
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
    // With other batches in flight on the device the persistent launch takes 3/5 of what fits (1.8 workgroups a CU; 9/16 until round 6: the kernel got 3 % shorter and
    // the optimum moved -- 432 workgroups 41.82, 448 41.71, 464 41.43, 480 41.45 ms a step, three alternating rounds, profiles/r06_rows_blocks_sweep.txt), and only ONE such launch
    // runs at a time on the
    // device (below: gRowsEv).  Its waves hold their registers and LDS until the launch ends; what they leave is all the other batches' latency-bound kernels get to
    // run in meanwhile -- and two rows launches side by side would take the whole chip between them again.  Four contexts, 3.1 Gbp, ms a step
    // (profiles/r05_rows_blocks_sweep.txt): the full launch, free-running (rounds 1-4) 44.4-45.0; 384 workgroups free-running 43.6-44.0; one at a time: 352 workgroups
    // 44.2-44.3, 384 43.4-43.8, 416 42.5-43.0, 448 42.2-43.0, 480 43.5-43.9.  Alone on the device the full launch is 2.7 ms a step faster than half of it.
    // (YGPU_ROWS_BLOCKS: the workgroups as a count, for such sweeps.)
    if (rowsShare && rowsBS == 256u) maxBlocksK = std::max(64u, maxBlocksK * 3u / 5u);
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
                while (lo < hi) { uint32_t mid = lo + (hi - lo + 1) / 2; if ((double)(ctx->hStripOff[2 * (size_t)mid] - ctx->hStripOff[2 * (size_t)r0]) <= perRange) lo = mid;
                    else hi = mid - 1; }
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
    // [0] chunks handed out, [1] ops handed out, [2] the high-water mark of [0] over the ranges
    ENSURE(ctx->traceCnt, 64); HIPCHK(hipMemsetAsync(ctx->traceCnt.p, 0, 64, ctx->stream));
    TRACE("lanes: trace arena");
    // diagnostics: how many extension problems of the batch are exact duplicates (direction, strand, read, rOff, qOff, qLen)?
    if (kTrace && getenv("YGPU_COUNT_DUPS")) {
        std::vector<ExtProb> hp(nProb); hipMemcpy(hp.data(), ctx->extProbs.p, sizeof(ExtProb) * (size_t)nProb, hipMemcpyDeviceToHost);
        std::vector<std::array<uint32_t, 4>> keys; keys.reserve(nProb);
        for (auto &e : hp) if (e.flags & XP_VALID) keys.push_back({e.qBase, e.rOff, (uint32_t)e.qOff | ((uint32_t)e.qLen << 16), e.flags & 3u});
        std::sort(keys.begin(), keys.end()); size_t dup = 0, sameStart = 0;
        for (size_t k = 1; k < keys.size(); k++) { dup += keys[k] == keys[k - 1];
            sameStart += keys[k][0] == keys[k - 1][0] && keys[k][1] == keys[k - 1][1] && (keys[k][2] & 0xFFFF) == (keys[k - 1][2] & 0xFFFF) && keys[k][3] == keys[k - 1][3]; }
        fprintf(stderr, "[ygpu] extension problems: %zu valid, %zu exact duplicates (%.2f%%), %zu share (read, strand, direction, rOff, qOff) with their predecessor (%.2f%%)\n",
            keys.size(), dup, 100.0 * dup / std::max<size_t>(1, keys.size()), sameStart, 100.0 * sameStart / std::max<size_t>(1, keys.size()));
    }
    if (kTrace) fprintf(stderr, "[ygpu] trace bound %.2f GB, arena %.2f GB (%llu chunks, ratio %.3f), %zu range(s); ext ops cap %u\n", boundBlocks * 128.0 / 1e9,
        nChunksArena * (YD_CHUNK_DWORDS * 4.0) / 1e9, nChunksArena, ctx->traceRatio, nRanges, extOpsCap);
    ENSURE(ctx->rowsClock, 16); { const unsigned long long init[2] = {~0ull, 0ull}; HIPCHK(hipMemcpyAsync(ctx->rowsClock.p, init, 16, hipMemcpyHostToDevice, ctx->stream)); }
    E.clock = ctx->rowsClock.as<unsigned long long>();
    E.trace = ctx->extTrace.as<uint32_t>(); E.nChunks = (uint32_t)nChunksArena; E.chunkCount = ctx->traceCnt.as<unsigned int>(); E.waveChunks = ctx->waveChunks.as<uint32_t>();
        E.maxCh = maxCh;
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
        // a launch that runs out of arena leaves problems unfinished: they must read as "no extension", not as the last batch's results
        HIPCHK(hipMemsetAsync(E.res, 0, sizeof(ExtRes) * (uint64_t)np, ctx->stream));
        {   // longest bound first: the launch's drain phase is then left with short problems only.  Keys 0xFFFF - qLen (invalid: 0xFFFF, last): one bucket per
            // length while the longest read has fewer than 4 096 bases, per 2^k lengths beyond; the values are the problems' indices inside the range
            const uint32_t sub = 0xFFFFu - (uint32_t)std::min(ctx->maxQ, 0xFFFF); int shift = 0; while (((uint32_t)ctx->maxQ >> shift) >= kBucketMax - 1u) shift++;
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
            for (uint32_t w = 0; w < np; w += 64) { uint32_t mx = 0; for (uint32_t k = w; k < std::min(np, w + 64); k++) { const ExtRes &r = hr[ho[k]];
                const uint32_t len = r.score > 0 ? (uint32_t)r.maxi : 0u; walkers += r.score > 0; sumLen += len; sumRows += r.rows; mx = std::max(mx, len); int b = 0;
                while (b < 7 && (len >> (b + 3))) b++; hist[len ? b : 0] += 1; } sumWaveMax += mx; }
            fprintf(stderr, "[ygpu] traceback: %u problems, %llu walk (%.1f%%), mean walk %.1f rows (all) / %.1f (walkers), rows computed mean %.1f; sum over waves of "
                "the longest walk %llu = %.1f x the lanes' mean\n",
                    np, walkers, 100.0 * walkers / np, (double)sumLen / np, (double)sumLen / std::max(1ull, walkers), (double)sumRows / np, sumWaveMax,
                        (double)sumWaveMax * 64.0 / std::max(1ull, sumLen));
            fprintf(stderr, "[ygpu] walk length histogram (0..7, 8.., 16.., 32.., 64.., 128.., 256.., 512..):"); for (int b = 0; b < 8; b++) fprintf(stderr, " %llu", hist[b]);
                fprintf(stderr, "\n");
        }
        if (traceSort > 0 && np > 4096u) {                                    // traceback order: by arena region, then by walk length (k_trace_keys)
            // (v1: not extOrder, which k_trace_keys reads)
            uint32_t *k0 = ctx->extKeys.as<uint32_t>() + p0, *v0 = ctx->extVals.as<uint32_t>() + p0, *v1 = ctx->extKeys2.as<uint32_t>() + p0;
            static const int lenBits = getenv("YGPU_TRACE_LENBITS") ? std::min(8, std::max(1, atoi(getenv("YGPU_TRACE_LENBITS")))) : 7;
            int lenShift = 0; while ((ctx->maxQ >> lenShift) > (1 << lenBits) - 1) lenShift++;
            // region bits above the length bits
            int keyBits = lenBits; while (keyBits < 32 && ((unsigned long long)E.nChunks >> std::min(traceSort, 31)) >> (keyBits - lenBits)) keyBits++;
            KL(k_trace_keys, dim3(gridFor(np, 256)), dim3(256), 0, ctx->stream, E.res, E.order, np, E.waveChunks, E.maxCh, std::min(traceSort, 31), lenShift, lenBits, k0, v0);
            rc = bucketOrder(ctx, k0, v0, 0, np, 0, std::max(0, keyBits - 12), 1u << std::min(keyBits, 12), v1, ctx->stream); if (rc) return rc;
            E.order = v1;
        }
        // (the tracebacks of the contexts one at a time, as the rows launches: 43.20 against 43.19 ms a step; in the rows launches' chain: 44.89 -- profiles/r05_trace_chain.txt,
        // commit 8c679ac)
        KL(traceKernel, dim3(gridFor(np, traceBS)), dim3(traceBS), 0, ctx->stream, E);
        if (c + 1 == nRanges) hipEventRecord(ctx->ev[T_XTRACE][1], ctx->stream);
        TRACE("lanes: ext_trace");
        if (kTrace) { unsigned w8[8]; hipMemcpyFromSymbol(w8, HIP_SYMBOL(gTraceDbg), sizeof w8); if (w8[0]) { ExtRes rr;
            hipMemcpy(&rr, E.res + w8[6], sizeof rr, hipMemcpyDeviceToHost); ExtProb pp; hipMemcpy(&pp, E.probs + w8[6], sizeof pp, hipMemcpyDeviceToHost);
            fprintf(stderr, "[ygpu] k_ext_trace left its strip: dword %d of %u, f0 %u, laneOff %u; problem %u where %08x (wave %u lane %u phase %u) score %d maxi %d maxj "
                "%d rows %u qLen %u flags %u\n", (int)w8[1], w8[2], w8[3], w8[4], w8[6], w8[7], w8[7] >> 10, (w8[7] >> 4) & 63, w8[7] & 15, rr.score, rr.maxi, rr.maxj, rr.rows,
                    pp.qLen, pp.flags);
            memset(w8, 0, sizeof w8); hipMemcpyToSymbol(HIP_SYMBOL(gTraceDbg), w8, sizeof w8); } }
        AlignArgs Ac = A; Ac.nRoots = r1; Ac.queueHead = cc + 8 * c + 1;
        PhaseArgs Xc = X; Xc.rootBegin = r0; Xc.slowList = ctx->slowList.as<uint32_t>() + r0; Xc.slowCount = cc + 8 * c + 2; Xc.useList = 1;
        if (c == 0) { ctx->evUsed[T_P3] = true; hipEventRecord(ctx->ev[T_P3][0], ctx->stream); }
        const uint32_t nr = r1 - r0, cap2 = nr / 4 + 1024;
        if (ctx->splitLanes) {
            ENSURE(ctx->memoKeys, 12ull * YD_MEMO * (nr + 1)); ENSURE(ctx->memoCount, 4ull * (nr + 1)); ENSURE(ctx->probs2, sizeof(ExtProb) * (uint64_t)cap2);
            ENSURE(ctx->rowsBound2, 8ull * (cap2 + 1)); ENSURE(ctx->extRes2, sizeof(ExtRes) * (uint64_t)cap2); ENSURE(ctx->fallList, 4ull * (nr + 1));
            HIPCHK(hipMemsetAsync(ctx->memoCount.p, 0, 4ull * (nr + 1), ctx->stream)); HIPCHK(hipMemsetAsync(ctx->rowsBound2.p, 0, 8ull * (cap2 + 1), ctx->stream));
            Xc.memoKeys = ctx->memoKeys.as<uint32_t>(); Xc.memoCount = ctx->memoCount.as<unsigned int>(); Xc.probs2 = ctx->probs2.as<ExtProb>();
                Xc.rowsBound2 = ctx->rowsBound2.as<unsigned long long>();
            Xc.nProb2 = cc + 8 * c + 3; Xc.probs2Cap = cap2;
        } else { Xc.memoKeys = nullptr; Xc.memoCount = nullptr; Xc.probs2 = nullptr; Xc.rowsBound2 = nullptr; Xc.nProb2 = nullptr; Xc.probs2Cap = 0; }
        {   // the roots by descending length of their merged lists (phase_lanes.h k_p3_keys): keys and order in the traceback's key arrays, which are free by now
            // (YGPU_P3_SORT=1; measured in round 6 and OFF: k_p3_lanes 3.07 -> 2.76 ms, but the key and ordering kernels cost 0.22 and k_split_lanes, which takes the split
            // roots in the order k_p3_lanes lists them, goes 1.46 -> 1.97 ms: the kernels are bound by their scattered fetches, not by the idle lanes of the loop)
            static const bool p3Sort = getenv("YGPU_P3_SORT") && atoi(getenv("YGPU_P3_SORT")) != 0;
            Xc.p3Order = nullptr;
            if (p3Sort && nr > 4096u) {
                uint32_t *kk = ctx->extKeys.as<uint32_t>() + p0, *oo = ctx->extVals.as<uint32_t>() + p0;
                KL(k_p3_keys, dim3(gridFor(nr, 256)), dim3(256), 0, ctx->stream, Xc, r1, kk);
                rc = bucketOrder(ctx, kk, nullptr, r0, nr, 0, 0, 4096u, oo, ctx->stream); if (rc) return rc;
                Xc.p3Order = oo;
            }
        }
        KL(k_p3_lanes, dim3(gridFor(nr, YD_P3_BS)), dim3(YD_P3_BS), 0, ctx->stream, Ac, Xc);
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
                        ENSURE(ctx->keys2a, 4ull * (cap2 + 1)); ENSURE(ctx->keys2b, 4ull * (cap2 + 1)); ENSURE(ctx->vals2a, 4ull * (cap2 + 1));
                            ENSURE(ctx->vals2b, 4ull * (cap2 + 1));
                        KL(k_prob_keys, dim3(gridFor(n2, 256)), dim3(256), 0, ctx->stream, ctx->probs2.as<ExtProb>(), n2, ctx->keys2a.as<uint32_t>(), ctx->vals2a.as<uint32_t>());
                        { const uint32_t sub = 0xFFFFu - (uint32_t)std::min(ctx->maxQ, 0xFFFF); int shift = 0; while (((uint32_t)ctx->maxQ >> shift) >= kBucketMax - 1u) shift++;
                          rc = bucketOrder(ctx, ctx->keys2a.as<uint32_t>(), nullptr, 0, n2, sub, shift, ((uint32_t)ctx->maxQ >> shift) + 2u, ctx->vals2b.as<uint32_t>(),
                              ctx->stream); if (rc) return rc; }
                    }
                    E2.order = ctx->vals2b.as<uint32_t>(); E2.res = ctx->extRes2.as<ExtRes>(); E2.queue = cc + 8 * c + 4; E2.ctr = nullptr;   // counted by k_split_lanes
                    {   // a small launch: a few problems per lane, so its length is set by the lanes' chains of problems, not by the chip's throughput.  One wave
                        // per SIMD runs a row 2.4x faster than three sharing it (a lone wave issues every ~5 cycles) and gives every lane more problems to balance.
                        const uint64_t blocks2 = ctx->rows2PerCU > 0 ? (uint64_t)ctx->rows2PerCU : (uint64_t)ctx->nCU;
                        const uint64_t b2 = rows2BS == 512u ? std::max<uint64_t>(1, blocks2 / 2) : blocks2;
                        KL(rowsKernel2, dim3((unsigned)std::min<uint64_t>(std::min<uint64_t>(((uint64_t)n2 + rows2BS - 1) / rows2BS, b2), (uint64_t)maxBlocksK)), dim3(rows2BS), 0,
                            ctx->stream, E2); }
                    KL(traceKernel, dim3(gridFor(n2, traceBS)), dim3(traceBS), 0, ctx->stream, E2);
                }
                ENSURE(ctx->splitScratch, (size_t)YD_SL_BYTES * (((size_t)nSlow + 63) / 64 * 64));
                SplitArgs Sx; Sx.scratch = ctx->splitScratch.as<uint8_t>(); Sx.memoKeys = ctx->memoKeys.as<uint32_t>(); Sx.memoCount = ctx->memoCount.as<unsigned int>();
                Sx.res2 = ctx->extRes2.as<ExtRes>(); Sx.ops2 = ctx->extTrace.as<uint32_t>(); Sx.nProb2 = n2;
                Sx.fallList = ctx->fallList.as<uint32_t>(); Sx.fallCount = cc + 8 * c + 5; Sx.nSlots = nSlow;
                KL(k_split_lanes, dim3(gridFor(nSlow, 64)), dim3(64), 0, ctx->stream, Ac, Xc, Sx);
                Xw.slowList = ctx->fallList.as<uint32_t>(); Xw.slowCount = cc + 8 * c + 5;
                if (kTrace) { uint32_t fc = 0; HIPCHK(hipMemcpyAsync(&fc, cc + 8 * c + 5, 4, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx)); unsigned w8[8];
                    hipMemcpyFromSymbol(w8, HIP_SYMBOL(gFallWhy), sizeof w8);
                    fprintf(stderr, "[ygpu] roots left to the wave kernel %u (other %u, DP not listed %u, second split %u, depth/list %u)\n", fc, w8[0], w8[1], w8[2], w8[3]);
                    memset(w8, 0, sizeof w8); hipMemcpyToSymbol(HIP_SYMBOL(gFallWhy), w8, sizeof w8); }
            }
        }
        {   // the wave-per-root kernel takes what the lane kernels hand back: nothing at all on ordinary batches, and 1 024 waves that only find an empty list cost
            // 0.3 ms -- so its grid follows what the last batch handed back (all split roots when k_split_lanes is off)
            unsigned p3Waves = std::min<unsigned>(waves, std::max<unsigned>(64u, (r1 - r0) / 8u));
            if (ctx->splitLanes && ctx->lastFall >= 0) p3Waves = std::min<unsigned>(p3Waves, std::max<unsigned>(64u, (unsigned)std::min<long long>(1ll << 20,
                2ll * ctx->lastFall)));
            KL(k_align_p3, dim3(p3Waves), dim3(64), 0, ctx->stream, Ac, Xw);
            if (ctx->splitLanes && c + 1 == nRanges) HIPCHK(hipMemcpyAsync(&ctx->hFall, Xw.slowCount, 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (c + 1 == nRanges) hipEventRecord(ctx->ev[T_P3][1], ctx->stream);
        TRACE("lanes: p3");
        if (kTrace && getenv("YGPU_LIST_HIST")) {      // what k_p3_lanes walks: lengths of the three lists of a root (backward extension, phase-1 list, forward extension)
            const uint32_t nr2 = r1 - r0; std::vector<ExtRes> hr2(2 * (size_t)nr2); std::vector<RootState> hs(nr2);
            hipMemcpy(hr2.data(), ctx->extRes.as<ExtRes>() + 2 * (size_t)r0, sizeof(ExtRes) * hr2.size(), hipMemcpyDeviceToHost);
                hipMemcpy(hs.data(), ctx->rootState.as<RootState>() + r0, sizeof(RootState) * nr2, hipMemcpyDeviceToHost);
            unsigned long long hx[8] = {0}, hb[8] = {0}, ht[8] = {0}, sumx = 0, sumb = 0; const unsigned edges[7] = {0, 1, 2, 4, 8, 16, 32};
            auto bin = [&](unsigned v) { int k = 0; while (k < 7 && v > edges[k]) k++; return k; };
            for (uint32_t k = 0; k < nr2; k++) {
                const unsigned a = hr2[2 * k].score > 0 ? hr2[2 * k].nOps : 0u, c2 = hr2[2 * k + 1].score > 0 ? hr2[2 * k + 1].nOps : 0u, b = hs[k].len; hx[bin(a)]++;
                hx[bin(c2)]++; hb[bin(b)]++; ht[bin(a + b + c2)]++; sumx += a + c2; sumb += b; }
            fprintf(stderr, "[ygpu] list lengths over %u roots (bins: 0, 1, 2, 3-4, 5-8, 9-16, 17-32, more): extension lists", nr2);
                for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", hx[k]);
            fprintf(stderr, "; phase-1 lists"); for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", hb[k]); fprintf(stderr, "; merged");
                for (int k = 0; k < 8; k++) fprintf(stderr, " %llu", ht[k]);
            fprintf(stderr, "; mean ops per root: extensions %.1f, phase 1 %.1f\n", (double)sumx / nr2, (double)sumb / nr2);
        }
    }
    HIPCHK(hipMemcpyAsync(ctx->hRowsClock, ctx->rowsClock.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    // errors of the trace memory: grow what overflowed and have the caller redo the stage
    unsigned int usedOps = 0; HIPCHK(hipMemcpyAsync(&usedOps, ctx->traceCnt.as<unsigned int>() + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
    // (stageAlign reads the output counts from here)
    { const FetchPiece pc[2] = {{ctx->errFlag.p, &ef, 1}, {cnt + CNT_OUTCLUMPS, ctx->hOutCounts, 2}}; rc = fetchMany(ctx, pc, 2); if (rc) return rc; }
    ctx->hOutEf = ef; ctx->hOutValid = true;
    if (ctx->splitLanes) ctx->lastFall = (long long)ctx->hFall;              // (the fetch above synchronised the stream)
    if (ef == YERR_TRACEMEM) {
        if (ctx->traceRatio >= 64.0) { ctx->err = "the extension trace arena overflows even at 64 times the problems' bound"; return YGPU_ENOMEM; }
        ctx->traceRatio = std::min(64.0, ctx->traceRatio * 1.5); return -3;
    }
    if (ef == YERR_OUT && usedOps > extOpsCap) { ctx->opsRatio = std::min(4.0, std::max(ctx->opsRatio * 2.0, 1.3 * (double)usedOps / std::max(1.0, (double)boundBlocks * 10.0)));
        return -3; }
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
int stageAlign(ygpu_ctx *ctx)
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
        // (the cap only ever lowers the count)
        if (useLanes && per * waves > (3ull << 30)) waves = (unsigned)std::min<uint64_t>(waves, std::max<uint64_t>(64, (3ull << 30) / per));
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
            // (the roots' records in rank order: clumpsSorted, below)
            AlignArgs A; A.P = ctx->P; A.bases = ctx->dBases.as<uint8_t>(); A.B = B; A.order = nullptr; A.nRoots = NC;
            A.clumps = ctx->clumpsSorted.as<ChainClumpRec>(); A.clumpFrags = ctx->clumpFrags.as<DevFrag>(); A.queueHead = cnt + CNT_QALIGN;
            A.scratch = ctx->scratchAlign.as<uint8_t>(); A.scratchPerWave = per; A.maxQ = ctx->maxQ; A.listCap = listCap; A.front = front; A.genCap = genCap;
                A.traceRows = traceRows;
            A.outClumps = ctx->outClumps.as<ygpu_clump>(); A.outOps = ctx->outOps.as<uint32_t>(); A.outRoot = ctx->outRoot.as<uint32_t>(); A.outPush = ctx->outPush.as<uint32_t>();
            A.outCounts = cnt + CNT_OUTCLUMPS; A.outClumpCap = outClumpCap; A.outOpsCap = outOpsCap; A.rootPushCount = ctx->rootPush.as<unsigned int>();
            A.ctr = ctx->ctr.as<DevCounters>(); A.errFlag = ctx->errFlag.as<int>();
#ifdef YD_PROF
            { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(gProf), z, sizeof z); hipMemcpyToSymbol(HIP_SYMBOL(gRowsProf), z, sizeof(unsigned long long) * 8);
              hipMemcpyToSymbol(HIP_SYMBOL(gTraceProf), z, sizeof(unsigned long long) * 8); }
#endif
            bool laneOverflow = false, traceOverflow = false; ctx->hOutValid = false;
            if (!useLanes) KL(k_align, dim3(waves), dim3(64), 0, ctx->stream, A);
            else { rc = alignWithLaneExtensions(ctx, A, waves, stateOpsCap, gapOpsPerJoint); if (rc == -2) laneOverflow = true; else if (rc == -3) traceOverflow = true;
                else if (rc) return rc; }
#ifdef YD_PROF
            { streamSync(ctx); unsigned long long z[16]; hipMemcpyFromSymbol(z, HIP_SYMBOL(gProf), sizeof z);
              const char *nm[10] = {"root_total", "dp_rows", "traceback", "perfect_ext", "score", "emit", "split", "merge", "dp_calls", "roots"};
              fprintf(stderr, "[YD_PROF] waves %u:", waves); for (int i = 0; i < 10; i++) fprintf(stderr, " %s=%llu", nm[i], z[i]); fprintf(stderr, "\n");
              unsigned long long q[8]; hipMemcpyFromSymbol(q, HIP_SYMBOL(gRowsProf), sizeof q);      // k_ext_rows_pk: where its passes go
              if (q[0]) fprintf(stderr,
                  "[YD_PROF] k_ext_rows_pk: wave passes %llu; of them writing results %.1f %%, with a new maximum in some lane %.1f %%, handing blocks over %.1f %%; "
                      "refill rounds %.3f a pass (pool loads %.4f); busy lanes %.1f of 64\n",
                                q[0], 100.0 * q[1] / q[0], 100.0 * q[3] / q[0], 100.0 * q[5] / q[0], (double)q[2] / q[0], (double)q[6] / q[0], (double)q[4] / q[0]);
              unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(gTraceProf), sizeof t);      // k_ext_trace_pk: where its passes go
              if (t[0]) fprintf(stderr, "[YD_PROF] k_ext_trace_pk: %llu waves that walk, %.1f passes a wave, %.1f active lanes a pass, %.2f rows a lane and pass; op groups %.2f a "
                                        "lane-pass; lane-passes that end in a deletion run %.1f %%, in an insertion run %.1f %%\n",
                                t[0], (double)t[1] / t[0], (double)t[2] / t[1], (double)t[7] / t[2], (double)t[3] / t[2], 100.0 * t[5] / t[2], 100.0 * t[6] / t[2]); }
#endif
            uint32_t got[2] = {0, 0}, ef = 0;
            if (laneOverflow || traceOverflow) ef = YERR_OUT;
            else if (ctx->hOutValid) { got[0] = ctx->hOutCounts[0]; got[1] = ctx->hOutCounts[1]; ef = ctx->hOutEf; }      // (fetched with the lane pipeline's last wait)
            else { const FetchPiece pc[2] = {{cnt + CNT_OUTCLUMPS, got, 2}, {ctx->errFlag.p, &ef, 1}}; rc = fetchMany(ctx, pc, 2); if (rc) return rc; }
            if (ef == 0) { ctx->nOut = got[0]; ctx->nOutOps = got[1]; break; }
            if (kStats) fprintf(stderr, "[ygpu] ctx %p: align attempt %d repeated (%s); trace ratio %.3f, ops ratio %.4f\n", (void *)ctx, attempt + 1, traceOverflow
                ? "trace / extension-op arena" : (laneOverflow ? "phase-1 arenas (state ops, gap ops)" : "output arenas"), ctx->traceRatio, ctx->opsRatio);
            if (ef != YERR_OUT || attempt >= 24)      /* the trace estimate grows by half a time: 1.5^20 covers the floor-to-cap range */ { char b[96];
                snprintf(b, sizeof b, "align stage failed with device error %u (see dp_wave.h YERR_*)", ef); ctx->err = b; return ef == YERR_OUT ? YGPU_EOVERFLOW : YGPU_EINTERNAL;
                }
            if (!traceOverflow) {                                            // (a full trace arena has grown its own estimate)
                // (capped: up to 24 attempts, and a wrapped bound would never fit)
                outClumpCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 2ull * outClumpCap); outOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 2ull * outOpsCap);
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
            KL(k_out_layout, dim3(gridFor(ctx->nOut, 256)), dim3(256), 0, ctx->stream, ctx->outRoot.as<uint32_t>(), ctx->outPush.as<uint32_t>(), ctx->rootBase.as<uint32_t>(),
                ctx->rootPush.as<unsigned int>(), ctx->nOut, ctx->dstIdx.as<uint32_t>());
            KL(k_out_scatter, dim3(gridFor(ctx->nOut, 256)), dim3(256), 0, ctx->stream, ctx->outClumps.as<ygpu_clump>(), ctx->dstIdx.as<uint32_t>(), ctx->nOut,
                ctx->outClumps2.as<ygpu_clump>());
        }
        KL(k_read_counts, dim3(gridFor(NC, 256)), dim3(256), 0, ctx->stream, ctx->clumpsSorted.as<ChainClumpRec>(), ctx->rootPush.as<unsigned int>(), NC,
            ctx->readCount.as<unsigned int>());
    }
    rc = cubScan(ctx, ctx->readCount.as<uint32_t>(), ctx->readStart.as<uint32_t>(), n + 1); if (rc) return rc;
    if (NC) EV1(T_LAYOUT);
    return 0;
}

// ---- the stage-level DP entry (tests: every reference DP call through every kernel family) ------------------------------------------------------------------
static int dpBatchWave(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n, const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops)
{
    uint32_t *cnt = ctx->counters.as<uint32_t>();
    int listCap, front, genCap, traceRows; alignDims(ctx, listCap, front, genCap, traceRows); listCap = 64;     // no frame stack needed here
    const size_t per = alignScratchBytes(ctx->maxQ, traceRows, listCap, genCap);
    const unsigned waves = (unsigned)std::min<uint64_t>(std::max<uint32_t>(n, 1u), (uint64_t)ctx->nCU * 4);
    uint32_t opsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, (uint64_t)n * (2ull * ctx->maxQ + 64));
    ENSURE(ctx->scratchAlign, per * waves); ENSURE(ctx->dpProbs, sizeof(ygpu_dp_problem) * (uint64_t)(n + 1)); ENSURE(ctx->dpRes, sizeof(ygpu_dp_result) * (uint64_t)(n + 1));
        ENSURE(ctx->dpOps, 4ull * opsCap + 64);
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
            ExtProb e; e.qBase = base; e.rOff = p.rOff; e.qOff = p.qOff; e.qLen = p.qLen;
                e.flags = (p.strand ? XP_STRAND : 0u) | (p.mode == YGPU_DP_EXT_REV ? XP_REV : 0u) | XP_VALID;
            xp.push_back(e); xdst.push_back(k); xrows.push_back((unsigned long long)((p.qLen + 19u) / 10u));
        } else {
            JointRec j; memset(&j, 0, sizeof j); j.nsro = p.rOff; j.qBase = base; j.nsqo = p.qOff; j.qGap = p.qLen; j.rGap = p.rLen; j.kind = JK_DP;
                j.flags = (uint8_t)((p.strand ? 1u : 0u) | (p.mode == YGPU_DP_BANDED ? 2u : 0u));
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
        ENSURE(ctx->extTrace, nCh * (YD_CHUNK_DWORDS * 4ull) + 256); ENSURE(ctx->waveChunks, 4ull * (size_t)wavesK * maxCh + 64); ENSURE(ctx->extOps, 4ull * opsBound + 64);
            ENSURE(ctx->traceCnt, 64);
        HIPCHK(hipMemsetAsync(ctx->traceCnt.p, 0, 64, ctx->stream));
        ExtArgs E; E.P = ctx->P; E.bases = ctx->dBases.as<uint8_t>(); E.fwd = ctx->dFwd.as<uint8_t>(); E.rev = ctx->dRev.as<uint8_t>(); E.fwd4 = ctx->dFwd4.as<uint8_t>();
            E.rev4 = ctx->dRev4.as<uint8_t>(); E.probs = ctx->extProbs.as<ExtProb>(); E.nProb = nX;
        E.order = nullptr; E.clock = nullptr; E.trace = ctx->extTrace.as<uint32_t>(); E.nChunks = (uint32_t)nCh; E.chunkCount = ctx->traceCnt.as<unsigned int>();
            E.waveChunks = ctx->waveChunks.as<uint32_t>(); E.maxCh = maxCh;
        E.ops = ctx->extOps.as<uint32_t>(); E.opsCount = ctx->traceCnt.as<unsigned int>() + 1; E.opsCap = (uint32_t)opsBound; E.res = ctx->extRes.as<ExtRes>();
        E.queue = ctx->chunkCnt.as<unsigned int>(); E.ctr = nullptr; E.errFlag = ctx->errFlag.as<int>(); E.dbgMode = 0;
        if (second) KL(rowsKernel2, dim3(blocksK), dim3(256), 0, ctx->stream, E); else KL(rowsKernel, dim3(blocksK), dim3(256), 0, ctx->stream, E);
        { const unsigned tbs = pk ? (unsigned)YD_TRACE_BS : 256u; KL(traceKernel, dim3(gridFor(nX, tbs)), dim3(tbs), 0, ctx->stream, E); }
        uint32_t ef2 = 0; rc = fetchU32(ctx, ctx->errFlag.p, &ef2); if (rc) return rc;
        if (ef2 != YERR_TRACEMEM) break;                                    // (other errors are reported below)
        if (mult >= 1024) { ctx->err = "extension trace arena overflows"; return YGPU_ENOMEM; }
        HIPCHK(hipMemsetAsync(ctx->errFlag.p, 0, 4, ctx->stream)); HIPCHK(hipMemsetAsync(ctx->chunkCnt.p, 0, 64, ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->extRes.p, 0, sizeof(ExtRes) * (uint64_t)nX, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(hres.data(), ctx->extRes.p, sizeof(ExtRes) * (uint64_t)nX, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        for (uint32_t k = 0; k < nX; k++) xOff[k + 1] = xOff[k] + (hres[k].score > 0 ? hres[k].nOps : 0u);
    }
    if (nJ) {
        const uint32_t gapOpsCap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, gapOpsBound);
        ENSURE(ctx->joints, sizeof(JointRec) * (uint64_t)(nJ + 1)); ENSURE(ctx->sortKeys, 4ull * (nJ + 1)); ENSURE(ctx->sortVals, 4ull * (nJ + 1));
            ENSURE(ctx->sortVals2, 4ull * (nJ + 1));
        ENSURE(ctx->gapOps, 4ull * gapOpsCap); ENSURE(ctx->slowList, 4ull * (nJ + 1));
        HIPCHK(hipMemcpyAsync(ctx->joints.p, jp.data(), sizeof(JointRec) * (uint64_t)nJ, hipMemcpyHostToDevice, ctx->stream));
        KL(k_dp_classify, dim3(gridFor(nJ, 256)), dim3(256), 0, ctx->stream, ctx->P, ctx->dBases.as<uint8_t>(), ctx->dFwd.as<uint8_t>(), ctx->dRev.as<uint8_t>(),
            ctx->joints.as<JointRec>(), nJ, ctx->sortKeys.as<uint32_t>(), ctx->sortVals.as<uint32_t>());
        std::vector<uint32_t> keys(nJ), idx(nJ);
        rc = fetchU32(ctx, ctx->sortKeys.p, keys.data(), nJ); if (rc) return rc;
        for (uint32_t k = 0; k < nJ; k++) idx[k] = k;
        // the production path sorts the DP joints by (strip width, rows) as well
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
        uint32_t nd[3] = {0, 0, 0}, nb[3] = {0, 0, 0}; for (uint32_t k = 0; k < nJ; k++) if (keys[k] != YD_JKEY_NONE) { const uint32_t cls = gapJointClass(keys[k]); nd[0]++;
            nd[1] += cls <= 2u; nb[0] += cls == 0u; nb[1] += cls <= 1u; nb[2] += gapJointBand24(keys[k]) ? 1u : 0u; }
        HIPCHK(hipMemcpyAsync(cnt + CNT_NB12, nb, 12, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->sortVals2.p, idx.data(), 4ull * nJ, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(cnt + CNT_NDP, nd, 12, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemsetAsync(cnt + CNT_SLOW, 0, 4, ctx->stream)); HIPCHK(hipMemsetAsync(cnt + CNT_QALIGN, 0, 4, ctx->stream));
        int listCap, front, genCap, traceRows; alignDims(ctx, listCap, front, genCap, traceRows); listCap = 64;
        const size_t per = alignScratchBytes(ctx->maxQ, traceRows, listCap, genCap); const unsigned waves = 512;
        ENSURE(ctx->scratchAlign, per * waves);
        AlignArgs A; memset(&A, 0, sizeof A); A.P = ctx->P; A.bases = ctx->dBases.as<uint8_t>(); A.B = B; A.queueHead = cnt + CNT_QALIGN;
            A.scratch = ctx->scratchAlign.as<uint8_t>(); A.scratchPerWave = per;
        A.maxQ = ctx->maxQ; A.listCap = listCap; A.front = front; A.genCap = genCap; A.traceRows = traceRows; A.ctr = ctx->ctr.as<DevCounters>();
            A.errFlag = ctx->errFlag.as<int>();
        PhaseArgs X; memset(&X, 0, sizeof X); X.joints = ctx->joints.as<JointRec>(); X.nJoints = nJ; X.sortedVals = ctx->sortVals2.as<uint32_t>(); X.nDP = cnt + CNT_NDP;
            X.nDPb = cnt + CNT_NB12;
        X.gapOps = ctx->gapOps.as<uint32_t>(); X.gapOpsCount = cnt + CNT_GAPOPS; X.gapOpsCap = gapOpsCap; X.slowList = ctx->slowList.as<uint32_t>(); X.slowCount = cnt + CNT_SLOW;
        const unsigned gBlocks16 = (unsigned)std::min<uint64_t>(gridFor(nJ, 64), (uint64_t)ctx->nCU * 9), gBlocks32 = (unsigned)std::min<uint64_t>(gridFor(nJ, 64),
            (uint64_t)ctx->nCU * 6);
        ENSURE(ctx->gapScratch, (size_t)YD_GAP_SCRATCH * 64 * std::max(gBlocks16, gBlocks32)); X.gapScratch = ctx->gapScratch.as<uint8_t>();
        X.band24 = (gapBandPacked(ctx) && !getenv("YGPU_GAP24_OFF")) ? 1u : 0u;
        if (gapBandPacked(ctx)) {
            KL(k_gap_band_pk<12>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X); KL(k_gap_band_pk<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
            if (X.band24) KL(k_gap_band_pk<24>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X);
        }
        else { KL(k_gap_band<12>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X); KL(k_gap_band<16>, dim3(gBlocks16), dim3(64), 0, ctx->stream, A, X); }
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
        HIPCHK(hipMemcpyAsync(dOff, xOff.data(), 4ull * nX, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(dDst, xdst.data(), 4ull * nX, hipMemcpyHostToDevice, ctx->stream));
        KL(k_dp_gather_ext, dim3(gridFor(nX, 256)), dim3(256), 0, ctx->stream, ctx->extProbs.as<ExtProb>(), ctx->extRes.as<ExtRes>(), ctx->extTrace.as<uint32_t>(), dOff, dDst, nX,
           ctx->dpRes.as<ygpu_dp_result>(), ctx->dpOps.as<uint32_t>());
        HIPCHK(streamSync(ctx));
    }
    if (nJ) {
        HIPCHK(hipMemcpyAsync(dOff, jOff.data(), 4ull * nJ, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(dDst, jdst.data(), 4ull * nJ, hipMemcpyHostToDevice, ctx->stream));
        KL(k_dp_gather_gap, dim3(gridFor(nJ, 256)), dim3(256), 0, ctx->stream, ctx->P, ctx->dBases.as<uint8_t>(), ctx->dFwd.as<uint8_t>(), ctx->dRev.as<uint8_t>(),
            ctx->joints.as<JointRec>(), ctx->gapOps.as<uint32_t>(), dOff, dDst, nJ,
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
/* The trace stream of the last ygpu_run (see include/yaha_hip.h): counted from the extension results, which stay on the device until the next run. */
int ygpu_trace_volume(ygpu_ctx *ctx, uint64_t out[6])
{
    if (!ctx || !ctx->stream || !out || ctx->stageDone < 3) return YGPU_EINVAL;
    HIPCHK(hipSetDevice(ctx->device));
    for (int k = 0; k < 6; k++) out[k] = 0;
    const bool useLanes = ctx->laneExt && ctx->P.bandWidth == 5 && ctx->P.maxGap >= YD_LBAND;
    const uint32_t nProb = 2u * ctx->nClumps;
    if (!useLanes || !nProb || ctx->extRes.cap < sizeof(ExtRes) * (uint64_t)nProb) return 0;
    ENSURE(ctx->traceCnt, 64);
    unsigned long long *d = (unsigned long long *)((char *)ctx->traceCnt.p + 32);       // (behind the arena's three counters)
    HIPCHK(hipMemsetAsync(d, 0, 32, ctx->stream));
    KL(k_trace_volume, dim3(gridFor(nProb, 256)), dim3(256), 0, ctx->stream, ctx->extRes.as<ExtRes>(), nProb, d);
    unsigned long long h[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(h, d, 32, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];
    out[4] = ctx->rowsPacked ? 16 : 12;                                           // bytes a record
    out[5] = ctx->extTrace.cap;                                                   // the arena the records went to
    return 0;
}
int ygpu_dp_batch(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n, const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops)
{ return ygpu_dp_batch_ex(ctx, problems, n, YGPU_DP_KERNELS_AUTO, results, ops, n_ops); }
}  // extern "C"
