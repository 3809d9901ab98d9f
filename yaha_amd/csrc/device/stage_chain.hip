// stage_chain.hip -- stages A3 + A4 of the hot path: regions of the fragment array (QueryMatch.c:146-158, 224-303), the chain DP with its trimming and
// elimination (GraphPath.cpp:161-292, AlignHelpers.c:48-193, QueryMatch.c:170-215), the creation-order ranks of the root clumps.
#include "ctx.h"
#include "chain.h"
#include "chain_lanes.h"
#include "regions.h"

// ---- A3 + A4 ------------------------------------------------------------------------------------------------------
int stageChain(ygpu_ctx *ctx)
{
    const uint32_t F = ctx->nFrags; DevBatch B = devBatch(ctx);
    if (!F) return 0;
    int rc;
    // region boundaries (uses a second head/scan pair sized by F; the hit-level pair is still needed by buildFrags on a retry)
    ENSURE(ctx->regStart, 4ull * (F + 2)); ENSURE(ctx->multiList, 4ull * (F + 1)); ENSURE(ctx->smallList, 4ull * (F + 1)); ENSURE(ctx->bigList, 4ull * (F / 64 + 2));
        ENSURE(ctx->regionCount, 4ull * (F + 2)); ENSURE(ctx->regionBase, 4ull * (F + 2));
    // (the fragment scan's tile states are free again: reused for the region scan)
    const uint32_t nRegTiles = (uint32_t)gridFor(F, YD_REG_TILE);
    ENSURE(ctx->tileState, 8ull * (nRegTiles + 1));
    uint32_t *cnt = ctx->counters.as<uint32_t>();
    HIPCHK(hipMemsetAsync(ctx->tileState.p, 0, 8ull * (nRegTiles + 1), ctx->stream));
    KL(k_region_scan, dim3(nRegTiles), dim3(256), 0, ctx->stream, ctx->frags.as<DevFrag>(), F, ctx->P.maxGap, ctx->regStart.as<uint32_t>(), ctx->tileState.as<unsigned long long>(),
        cnt + CNT_NREG);
    uint32_t R = 0; { uint32_t two[2] = {0, 0}; rc = fetchU32(ctx, cnt + CNT_NREG, two, 2); if (rc) return rc;
        if (two[1]) { ctx->err = "region scan: a tile was not published within 30 s (look-back gave up)"; return YGPU_EINTERNAL; } R = two[0]; }
    ctx->nRegions = R;
    HIPCHK(hipMemcpyAsync((uint32_t *)ctx->regStart.p + R, &ctx->nFrags, 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(cnt + CNT_NMULTI, 0, 8, ctx->stream)); HIPCHK(hipMemsetAsync(cnt + CNT_NBIG, 0, 8, ctx->stream));
        HIPCHK(hipMemsetAsync(cnt + CNT_NMULTI2, 0, 16, ctx->stream));
    static_assert(CNT_NMULTI2 % 2 == 0 && CNT_NTINY == CNT_NMULTI2 + 1 && CNT_NSMALL == CNT_NMULTI2 + 2 && CNT_NMID == CNT_NMULTI2 + 3, "two 64-bit words: one memset, one fetch");
    KL(k_region_classify, dim3(gridFor(R, 1024 * YD_RCLS_IPT)), dim3(1024), 0, ctx->stream, ctx->regStart.as<uint32_t>(), R, ctx->multiList.as<uint32_t>(),
        ctx->multiList.as<uint32_t>() + F,
        ctx->smallList.as<uint32_t>(), ctx->smallList.as<uint32_t>() + F, (unsigned long long *)(cnt + CNT_NMULTI2), cnt + CNT_MAXN, ctx->bigList.as<uint32_t>(), cnt + CNT_NBIG);
    uint32_t four[4] = {0, 0, 0, 0}, mx = 0;
    { const FetchPiece pc[3] = {{cnt + CNT_NMULTI2, four, 4}, {cnt + CNT_NBIG, &ctx->nBig, 1}, {cnt + CNT_MAXN, &mx, 1}}; rc = fetchMany(ctx, pc, 3); if (rc) return rc; }
    uint32_t two[2] = {four[0], mx};
    ctx->nTiny = four[1]; ctx->nSmall = four[2]; ctx->nMid = four[3];
    ctx->nMulti = two[0]; ctx->maxN = two[1];
    if (kTrace && getenv("YGPU_REGION_HIST")) {                                  // fragments a region: how the three chain kernels' shares lie (diagnostic)
        std::vector<uint32_t> rs((size_t)R + 1); hipMemcpy(rs.data(), ctx->regStart.p, 4ull * (R + 1), hipMemcpyDeviceToHost);
        unsigned long long h[12] = {0}, fr[12] = {0}; static const uint32_t hi[12] = {1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 512, 0xFFFFFFFFu};
        for (uint32_t r = 0; r < R; r++) { const uint32_t n = rs[r + 1] - rs[r]; for (int c = 0; c < 12; c++) if (n <= hi[c]) { h[c]++; fr[c] += n; break; } }
        fprintf(stderr, "[ygpu] regions by fragments (count / fragments):");
        for (int c = 0; c < 12; c++) fprintf(stderr, " <=%u: %llu / %llu", hi[c], h[c], fr[c]);
        fprintf(stderr, "\n");
    }
    EV1(T_FRAGS);

    EV0(T_CHAIN);
    // (clump slots are pre-set to "invalid", so their number is a 16-byte store each: after the first batch the bound is twice what the last batch used -- most
    // fragments of a large genome are single hits that form no clump: 128 M fragments, 5.4 M clumps a batch at 3.1 Gbp -- and a batch that overflows it is redone at the full
    // bound)
    const uint32_t clumpSlack = 1024 + 32 * (ctx->nCU * 27 + 64) + 512 * (ctx->nCU * 8), clumpCapFull = F + R / 2 + clumpSlack;
    // + one open chunk per wave (k_chain: 32/256, k_chain_lanes: 512/2048)
    uint32_t clumpCap = ctx->lastClumpSlots ? (uint32_t)std::min<uint64_t>(clumpCapFull, 2ull * ctx->lastClumpSlots + clumpSlack) : clumpCapFull,
        fragCap = 2 * F + 1024 + 256 * (ctx->nCU * 27 + 64) + 2048 * (ctx->nCU * 8);
    // tests: a first bound that overflows
    if (const char *e = getenv("YGPU_CLUMP_BOUND")) { const long v = atol(e); if (v > 0 && (uint64_t)v < clumpCapFull) clumpCap = (uint32_t)v; }
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
        // (the lane kernels: 2 .. 4, 5 .. 8 and 9 .. 16 fragments a region -- 8.5, 17 and 34 KB of LDS a wave)
        if (ctx->nTiny) KL(k_chain_lanes<YD_CLT4>, dim3((unsigned)std::min<uint64_t>(gridFor(ctx->nTiny, 64), (uint64_t)ctx->nCU * 16)), dim3(64), 0, ctx->stream, A,
            ctx->multiList.as<uint32_t>() + F - ctx->nTiny, ctx->nTiny);
        if (ctx->nSmall) KL(k_chain_lanes<YD_CL>, dim3((unsigned)std::min<uint64_t>(gridFor(ctx->nSmall, 64), (uint64_t)ctx->nCU * 8)), dim3(64), 0, ctx->stream, A,
            ctx->smallList.as<uint32_t>(), ctx->nSmall);
        if (ctx->nMid) KL(k_chain_lanes<YD_CLM>, dim3((unsigned)std::min<uint64_t>(gridFor(ctx->nMid, 64), (uint64_t)ctx->nCU * 4)), dim3(64), 0, ctx->stream, A,
            ctx->smallList.as<uint32_t>() + F - ctx->nMid, ctx->nMid);
        if (ctx->nMulti) KL(k_chain, dim3(waves), dim3(64), 0, ctx->stream, A);
        if (ctx->nBig) KL(k_chain_big, dim3(wavesBig), dim3(64), 0, ctx->stream, A, ctx->bigList.as<uint32_t>(), ctx->nBig, cnt + CNT_QBIG);
        // creation-order rank of every root clump: the sum over the regions' counts is launched before anybody knows whether the attempt fitted -- an attempt that did not
        // is redone, sum included -- so that the counts, the error flag and the number of clumps cross in ONE wait
        rc = cubScan(ctx, ctx->regionCount.as<uint32_t>(), ctx->regionBase.as<uint32_t>(), R + 1); if (rc) return rc;
        uint32_t got[2] = {0, 0}, ef = 0;
        { const FetchPiece pc[3] = {{cnt + CNT_CLUMPS, got, 2}, {ctx->errFlag.p, &ef, 1}, {ctx->regionBase.as<uint32_t>() + R, &ctx->nClumps, 1}}; rc = fetchMany(ctx, pc, 3);
            if (rc) return rc; }
        if (ef == 0 && got[0] <= clumpCap && got[1] <= fragCap) { ctx->nClumpSlots = got[0]; ctx->nClumpFrags = got[1]; ctx->lastClumpSlots = got[0]; break; }
        if (attempt >= 6) { ctx->err = "chain stage: arena overflow persists"; return YGPU_EOVERFLOW; }
        if (clumpCap < clumpCapFull) clumpCap = clumpCapFull; else { clumpCap *= 2; fragCap *= 2; }      // grow and redo: the fragment array was modified in place
        rc = buildFrags(ctx, true); if (rc) return rc;
    }
    // (ctx->nClumps: clumps actually formed -- slots minus chunk slack)
    ENSURE(ctx->order, 4ull * (ctx->nClumps + 1)); ENSURE(ctx->clumpsSorted, sizeof(ChainClumpRec) * ((uint64_t)ctx->nClumps + 1));
    if (ctx->nClumpSlots) KL(k_clump_order, dim3(gridFor(ctx->nClumpSlots, 256)), dim3(256), 0, ctx->stream, ctx->clumps.as<ChainClumpRec>(), ctx->nClumpSlots,
        ctx->regionBase.as<uint32_t>(), ctx->order.as<uint32_t>(), ctx->clumpsSorted.as<ChainClumpRec>());
    EV1(T_CHAIN);
    return 0;
}

extern "C" {
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

}  // extern "C"
