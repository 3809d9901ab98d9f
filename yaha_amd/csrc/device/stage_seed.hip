// stage_seed.hip -- the batch's way onto the device and stages A1 + A2 of the hot path: k-mer hashing and table lookup (Query.c:233-244, 341-412), the seed-hit join
// (QueryMatch.c:52-121, QueryHeap.inl:70-134: here the hits of a (read, strand) sorted inside one workgroup, segsort.h) and the fragment array.
#include "ctx.h"
#include "seed.h"
#include "segsort.h"

uint32_t ydSegSortMax() { return YD_SEGSORT_MAX; }
size_t ydLowTableBytes() { return YD_LOW_BITS / 8; }
// the bit table of k_kmer_lookup: which k-mers have a first offset in the low half of their table line (seed.h)
int ydLowOffsets(ygpu_ctx *ctx, const ygpu_index_view *ix)
{
    const uint64_t HT = 1ull << (2 * ix->wordLen);
    if (ix->totalMatches)
        KL(k_low_offsets, dim3((unsigned)std::min<uint64_t>(gridFor(ix->totalMatches, 256), (uint64_t)ctx->nCU * 64)), dim3(256), 0, ctx->stream,
           ctx->dSO.as<uint32_t>(), (uint32_t)HT, ctx->dROA.as<uint32_t>(), (uint32_t)ix->totalMatches, ctx->dLow.as<uint32_t>());
    return 0;
}
// (the first launch loads the library's code object: ~20 ms that need not follow the image)
__global__ void k_touch(unsigned int *p) { if (p && threadIdx.x == 1000) *p = 0; }
int ydFirstLaunch(ygpu_ctx *ctx) { KL(k_touch, dim3(1), dim3(64), 0, ctx->stream, (unsigned int *)nullptr); return 0; }

// ---- A1 + A2 (+ fragment array) --------------------------------------------------------------------------------
// The ranking of the workgroup sort's passes (wgsort.h): 0 = LDS atomics, 1 = ballots.  Atomics are the default; their stability rests on the order in which the LDS
// serves the lanes of one instruction, so every batch's keys are checked by k_frag_scan_build, and the first batch that fails the check switches the process to the
// ballots for good (buildFrags below; the batch is sorted again).  YGPU_SORT_RANK=ballots | atomic (read at every batch) overrides both the default and the switch -- with
// "atomic" a failed check is an error; YGPU_SORT_CHECK_INJECT=1 makes the check of the process' first batch sorted with atomics fail (the tests' way into the fallback).
static std::atomic<int> gSortFell{0}, gSortInject{-1};
static int sortRank(bool *forced = nullptr)
{
    const char *e = getenv("YGPU_SORT_RANK"); const bool b = e && (e[0] == 'b' || e[0] == 'B'), a = e && (e[0] == 'a' || e[0] == 'A');
    if (forced) *forced = a || b;
    return b ? 1 : (a ? 0 : gSortFell.load());
}
int ydSortRank() { return sortRank(); }
// maxGap for the dead-single test of seed.h, or -1: every fragment is kept (the fragments themselves are asked for, or one word can be a whole match)
static int fragDropGap(const ygpu_ctx *ctx) { return (ctx->keepAllFrags || ctx->P.wordLen >= ctx->P.minMatch) ? -1 : ctx->P.maxGap; }
int stageSeed(ygpu_ctx *ctx)
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
    ctx->sortRankUsed = sortRank(&ctx->sortRankForced);
    ENSURE(ctx->keysA, 8ull * H); ENSURE(ctx->keysB, 8ull * H);
    ENSURE(ctx->expandStart, 4ull * (gridFor(H, YD_EXPAND_HITS) + 1));
    KL(k_expand_starts, dim3(gridFor(K, 256)), dim3(256), 0, ctx->stream, ctx->hitOff.as<uint32_t>(), K, ctx->expandStart.as<uint32_t>());
    KL(k_expand_hits, dim3(gridFor(H, YD_EXPAND_HITS)), dim3(256), 0, ctx->stream, ctx->dROA.as<uint32_t>(), ctx->posS.as<uint32_t>(), ctx->hitOff.as<uint32_t>(),
        ctx->posRsI.as<uint32_t>(), ctx->expandStart.as<uint32_t>(), K, H, ctx->keysA.as<unsigned long long>());
    {
        // The sort is stable and k_expand_hits writes the hits of one (read, strand) in ascending query offset (k-mers in order, each
        // k-mer's reference offsets ascending), so two hits of one diagonal are already in qo order: the low 15 key bits need no pass.
        // The hits of one (read, strand) are one segment: sorted on the 32 diagonal bits only, inside one workgroup (one pass over HBM), instead of a
        // batch-wide sort that also has to order the (read, strand) bits.
        {
            ENSURE(ctx->segOff, 4ull * (2 * n + 2));
            KL(k_seg_offsets, dim3(gridFor(2 * n + 1, 256)), dim3(256), 0, ctx->stream, ctx->dKmerOff.as<uint32_t>(), ctx->hitOff.as<uint32_t>(), 2 * n,
                ctx->segOff.as<uint32_t>());
            {
                // segments of up to 15 872 hits: one workgroup each (segsort.h), in twelve size classes, one launch per class over exactly its segments
                const unsigned long long *in = ctx->keysA.as<unsigned long long>(); unsigned long long *out = ctx->keysB.as<unsigned long long>();
                    const uint32_t *so = ctx->segOff.as<uint32_t>();
                ENSURE(ctx->segLists, 4ull * (YD_SEG_NCLASS + 1) * (2 * n + 1));
                uint32_t *segCnt = ctx->counters.as<uint32_t>() + CNT_SEGC;
                const uint32_t mx = ctx->segSortMax;                                  // YD_SEGSORT_MAX; lower only to drive the long-segment path in tests
                // threads x hits a thread: 128 x 8, 128 x 16, 256 x 12 / 16, 512 x 10 / 12 / 14 / 16 and, for the four largest classes, 512 x 20 / 24 / 28 / 31 (384 and 768
                // threads x 16 sorted slower than the next shape up).  The largest classes had 1 024-thread workgroups (x 10 / 12 / 14 / 16): four waves a SIMD with 72-112
                // registers each, which beside a rows launch -- one or two waves of 152 registers on every SIMD of the device while it runs -- found room on the CUs with one
                // rows workgroup (x 10, x 12) or NOWHERE (x 14, x 16: their launch, first in the stream, then waited for the rows launch to end -- 0.36 ms alone, 4.7 ms in
                // the four-context run, and every smaller class behind it).  512 threads x twice the hits: two waves a SIMD of 136-176 registers, room beside one rows
                // workgroup for all four (x 32 would need 177 registers, eight a SIMD too many: hence 15 872 hits as the limit of a single workgroup's sort); the same speed alone,
                // 0.3-0.5 ms a step with four contexts (profiles/r05_sort_shapes.txt).  YGPU_SORT_WIDE=0: the old shapes.
                static const uint32_t kShape[YD_SEG_NCLASS] = {1024u, 2048u, 3072u, 4096u, 5120u, 6144u, 7168u, 8192u, 10240u, 12288u, 14336u, YD_SEGSORT_MAX};
                SegClassHi HI; for (int c = 0; c < YD_SEG_NCLASS; c++) HI.hi[c] = std::min(mx, kShape[c]);
                // one launch per class over exactly its segments: in[inB..inE) sorted into out[inB..)
                auto sortClasses = [&](const unsigned long long *src, unsigned long long *dst, const uint32_t *sB, const uint32_t *sE, uint32_t nSeg, const uint32_t *lists,
                    const uint32_t *nc) -> int {
#define YD_SORT_CLASS(c, BS, IPT) if (nc[c]) { const uint32_t *cl = lists + (size_t)(c) * nSeg; \
                    if (byBallots) KL((k_seg_sort<BS, IPT, false>), dim3(nc[c]), dim3(BS), 0, ctx->stream, src, dst, sB, sE, cl, ooo); \
                    else KL((k_seg_sort<BS, IPT, true>), dim3(nc[c]), dim3(BS), 0, ctx->stream, src, dst, sB, sE, cl, ooo); }
                    const char *ws = getenv("YGPU_SORT_WIDE"); const int wideShapes = ws ? atoi(ws) : 1;      // (read at every call: the tests run both)
                    const bool byBallots = ctx->sortRankUsed == 1;
                    unsigned int *ooo = ctx->counters.as<unsigned int>() + CNT_NFRAGS + 2;      // raised by the sort's order check, read with the fragment count (buildFrags)
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
                if (kTrace) { std::vector<uint32_t> so2(2 * (size_t)n + 1); hipMemcpy(so2.data(), so, 4ull * (2 * n + 1), hipMemcpyDeviceToHost);
                    unsigned long long hb = 0, mxl = 0, cl[YD_SEG_NCLASS] = {0};
                    for (uint32_t k = 0; k < 2 * n; k++) { const unsigned long long l = so2[k + 1] - so2[k]; if (l > mx) { hb += l; mxl = std::max(mxl, l);
                        } else for (int c = 0; c < YD_SEG_NCLASS; c++) if (l <= HI.hi[c]) { cl[c] += l; break; } }
                    fprintf(stderr, "[ygpu] hit sort: %u hits; segments above %u hits: %u holding %llu hits (%.1f%%, longest %llu); %% of the hits by class:", H, mx, nBig, hb,
                        100.0 * hb / H, mxl);
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
                    if (kTrace) { fprintf(stderr, "[ygpu] hit sort: %u long segments cut into buckets, by class:", nBig);
                        for (int c = 0; c < YD_SEG_NCLASS; c++) fprintf(stderr, " %u", ns[c]); fprintf(stderr, "; %u to be cut again\n", nOver); }
                    for (int level = 0; nOver; level++) {
                        if (level >= 12) { ctx->err = "hit sort: a segment does not fit the workgroup sort after twelve cuts"; return YGPU_EINTERNAL; }
                        DevBuf &xB = level & 1 ? ctx->sub3B : ctx->sub2B, &xE = level & 1 ? ctx->sub3E : ctx->sub2E, &xL = level & 1 ? ctx->sub3Lists : ctx->sub2Lists;
                        const uint32_t nSub2 = nOver * YD_SPLIT_NB;
                        ENSURE(xB, 4ull * nSub2 + 64); ENSURE(xE, 4ull * nSub2 + 64); ENSURE(xL, 4ull * (YD_SEG_NCLASS + 1) * nSub2 + 64);
                        KL(k_seg_split_range, dim3(nOver), dim3(1024), 0, ctx->stream, cur, other, out, oB, oE, oList, xB.as<uint32_t>(), xE.as<uint32_t>());
                        HIPCHK(hipMemsetAsync(segCnt, 0, 4 * (YD_SEG_NCLASS + 1), ctx->stream));
                        KL(k_seg_classify, dim3(gridFor(nSub2, 256)), dim3(256), 0, ctx->stream, xB.as<uint32_t>(), xE.as<uint32_t>(), nSub2, HI, xL.as<uint32_t>(),
                            (uint32_t *)nullptr, (uint32_t *)nullptr, segCnt);
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
// The workgroup sort against std::stable_sort on the host, both rankings, four shapes: segments of every fill (one hit, a partial last row, full), diagonals that are
// random, all equal, two values alternating lane by lane, equal inside a row and in runs across rows (what the lane order of the LDS atomics has to get right),
// query offsets ascending in input order.  (ygpu_selftest_primitives)
template <unsigned BS, unsigned IPT>
static int selftestSegSortShape(ygpu_ctx *ctx, uint64_t &x, char *why, size_t whyLen)
{
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    constexpr uint32_t N = BS * IPT; const uint32_t nSeg = 24;
    std::vector<uint32_t> segB(nSeg), segE(nSeg), list(nSeg); std::vector<unsigned long long> keys;
    for (uint32_t sI = 0; sI < nSeg; sI++) {
        const uint32_t pat = sI % 6u;
        uint32_t len = sI == 0 ? 1u : (sI == 1 ? N : (sI == 2 ? N - 63u : (sI == 3 ? 65u : 1u + (uint32_t)(rnd() % N))));
        segB[sI] = (uint32_t)keys.size(); list[sI] = sI;
        const uint32_t base = (uint32_t)rnd(), alt = (uint32_t)rnd();
        for (uint32_t i = 0; i < len; i++) {
            uint32_t d;
            switch (pat) {
            case 0: d = (uint32_t)rnd(); break;                                               // random
            case 1: d = base; break;                                                          // one diagonal
            case 2: d = (i & 1u) ? base : alt; break;                                         // two values, lane by lane
            case 3: d = base + ((i / 64u) % 3u) * 0x01010101u; break;                          // equal inside a row, three values over the rows
            case 4: d = (rnd() % 4u) ? base + (uint32_t)(rnd() % 5u) : (uint32_t)rnd(); break;  // a crowded neighbourhood and scattered singles
            default: d = base ^ ((uint32_t)(rnd() % 3u) << (8u * (uint32_t)(rnd() % 4u))); break; // ties in three of the four digits
            }
            keys.push_back(((unsigned long long)(sI & 0x1FFFFu) << 47) | ((unsigned long long)d << 15) | (unsigned long long)(i & 0x7FFFu));
        }
        segE[sI] = (uint32_t)keys.size();
    }
    std::vector<unsigned long long> want(keys);
    for (uint32_t sI = 0; sI < nSeg; sI++)
        std::stable_sort(want.begin() + segB[sI], want.begin() + segE[sI], [](unsigned long long a, unsigned long long b) { return (uint32_t)(a >> 15) < (uint32_t)(b >> 15); });
    DevBuf in, out, meta; struct Rel { DevBuf &a, &b, &c; ~Rel() { a.release(); b.release(); c.release(); } } rel{in, out, meta};
    if (in.ensure(8ull * keys.size() + 64) || out.ensure(8ull * keys.size() + 64) || meta.ensure(16ull * nSeg + 64)) { ctx->err = "hipMalloc failed"; return YGPU_ENOMEM; }
    HIPCHK(hipMemcpyAsync(in.p, keys.data(), 8ull * keys.size(), hipMemcpyHostToDevice, ctx->stream));
    uint32_t *m = meta.as<uint32_t>();
    HIPCHK(hipMemcpyAsync(m, segB.data(), 4ull * nSeg, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(m + nSeg, segE.data(), 4ull * nSeg, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(m + 2 * nSeg, list.data(), 4ull * nSeg, hipMemcpyHostToDevice, ctx->stream));
    std::vector<unsigned long long> got(keys.size());
    for (int ballots = 0; ballots < 2; ballots++) {
        HIPCHK(hipMemsetAsync(out.p, 0xFF, 8ull * keys.size(), ctx->stream)); HIPCHK(hipMemsetAsync(m + 3 * nSeg, 0, 4, ctx->stream));
        if (ballots) KL((k_seg_sort<BS, IPT, false>), dim3(nSeg), dim3(BS), 0, ctx->stream, in.as<unsigned long long>(), out.as<unsigned long long>(), m, m + nSeg, m + 2 * nSeg,
            m + 3 * nSeg);
        else KL((k_seg_sort<BS, IPT, true>), dim3(nSeg), dim3(BS), 0, ctx->stream, in.as<unsigned long long>(), out.as<unsigned long long>(), m, m + nSeg, m + 2 * nSeg,
            m + 3 * nSeg);
        uint32_t flag = 0; HIPCHK(hipMemcpyAsync(&flag, m + 3 * nSeg, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(got.data(), out.p, 8ull * keys.size(), hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(streamSync(ctx));
        for (size_t i = 0; i < keys.size(); i++) if (got[i] != want[i]) {
            uint32_t sI = 0; while (sI + 1 < nSeg && segB[sI + 1] <= i) sI++;
            snprintf(why, whyLen, "selftest: the workgroup sort (%u threads x %u, ranking by %s) differs from a stable sort at hit %zu of segment %u (pattern %u, %u hits)", BS,
                IPT,
                     ballots ? "ballots" : "LDS atomics", i - segB[sI], sI, sI % 6u, segE[sI] - segB[sI]);
            ctx->err = why; return YGPU_EINTERNAL;
        }
        if (flag) { snprintf(why, whyLen, "selftest: the workgroup sort (%u threads x %u) raised its order check on keys that came out in order", BS, IPT); ctx->err = why;
            return YGPU_EINTERNAL; }
    }
    return 0;
}
int ydSelftestSegSort(ygpu_ctx *ctx, uint64_t &x)
{
    static thread_local char why[256]; int rc;
    rc = selftestSegSortShape<128, 8>(ctx, x, why, sizeof why); if (rc) return rc;
    rc = selftestSegSortShape<256, 16>(ctx, x, why, sizeof why); if (rc) return rc;
    rc = selftestSegSortShape<512, 20>(ctx, x, why, sizeof why); if (rc) return rc;
    rc = selftestSegSortShape<512, YD_SORT_TOP>(ctx, x, why, sizeof why); if (rc) return rc;
    return selftestSegSortShape<1024, 16>(ctx, x, why, sizeof why);
}

// (Re)creates the fragment array from the sorted keys -- the chain stage trims it in place, so a redo of that stage comes back here.  One kernel (seed.h:
// k_frag_scan_build) counts and writes; the array is sized from the last batch's count, and a batch that needs more is run again with room (the first batch of
// a context always is: its first pass only counts).
int buildFrags(ygpu_ctx *ctx, bool redo)      // redo: the regions stand, only the records are rebuilt (refLen is otherwise set by k_region_scan)
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
        ENSURE(ctx->kmerParts, 4096); HIPCHK(hipMemsetAsync(ctx->kmerParts.p, 0, 4096, ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->ctr.as<DevCounters>()->v + C_FRAGS, 0, 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->tileState.p, 0, 8ull * (nTiles + 1), ctx->stream));
        KL(k_frag_scan_build, dim3(nTiles), dim3(YD_FRAG_BS), 0, ctx->stream, ctx->keysB.as<unsigned long long>(), H, ctx->P.wordLen, fragDropGap(ctx), ctx->frags.as<DevFrag>(),
            cap,
           ctx->tileState.as<unsigned long long>(), total, ctx->kmerParts.as<unsigned int>());
        uint32_t two[3] = {0, 0, 0}; int rc = fetchU32(ctx, total, two, 3); if (rc) return rc;
        if (two[1]) { ctx->err = "fragment scan: a tile was not published within 30 s (look-back gave up)"; return YGPU_EINTERNAL; }
        if (!redo && ctx->sortRankUsed == 0 && !ctx->sortRankForced) {
            if (gSortInject.load() < 0) { const char *e = getenv("YGPU_SORT_CHECK_INJECT"); gSortInject.store(e && atoi(e) ? 1 : 0); }
            int one = 1; if (gSortInject.compare_exchange_strong(one, 0)) two[2] = 1;
        }
        if (two[2]) {
            // (the words of the look-back are clean again -- every tile was published and read --, the count is not used)
            HIPCHK(hipMemsetAsync(total, 0, 12, ctx->stream));
            if (ctx->sortRankUsed == 0 && !ctx->sortRankForced && !redo) {
                if (gSortFell.exchange(1) != 1)
                    fprintf(stderr, "[ygpu] hit sort: keys out of order behind the ranking by LDS atomics; this batch is sorted again and the process "
                                    "stays with the ranking by ballots\n");
                return YD_RESORT;
            }
            ctx->err = "hit sort: the sorted keys are not in ascending order"; return YGPU_EINTERNAL;
        }
        const uint32_t F = two[0];
        if (F <= cap) { ctx->nFrags = F; break; }
        if (pass >= 2) { ctx->err = "fragment build: the count changed between passes"; return YGPU_EINTERNAL; }
        ENSURE(ctx->frags, 16ull * ((uint64_t)F + F / 8 + 4096));
    }
    KL(k_sum_parts, dim3(1), dim3(1024), 0, ctx->stream, ctx->kmerParts.as<unsigned int>(), ctx->ctr.as<DevCounters>()->v + C_FRAGS);
    if (ctx->nFrags && (redo || ctx->keepAllFrags)) KL(k_frag_finish, dim3(gridFor(ctx->nFrags, 256)), dim3(256), 0, ctx->stream, ctx->frags.as<DevFrag>(), ctx->nFrags);
    return 0;
}

// ---- the batch: codes, offsets, reverse complement, packed copies ---------------------------------------------------------------------------------------------
int uploadBatch(ygpu_ctx *ctx, const ygpu_read_batch *b, bool wait)
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
    ENSURE(ctx->dFwd, ctx->totalBases + 256); ENSURE(ctx->dRev, ctx->totalBases + 256); ENSURE(ctx->dFwd4, ctx->totalBases / 2 + 256);
        ENSURE(ctx->dRev4, ctx->totalBases / 2 + 256);   /* slack: lane kernels read whole dwords around a segment */ ENSURE(ctx->dReadOff, 4ull * (n + 1));
        ENSURE(ctx->dKmerOff, 4ull * (2 * n + 1));
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

int ygpu_seed_join(ygpu_ctx *ctx, const ygpu_fragment **frags, uint64_t *n_frags)
{
    if (!ctx || !ctx->stream) return YGPU_EINVAL;
    ctx->stageDone = 0; ctx->keepAllFrags = true; int rc = runTo(ctx, 1); ctx->keepAllFrags = false; if (rc) return rc;
    ctx->hFrags.resize(ctx->nFrags);
    if (ctx->nFrags) HIPCHK(hipMemcpy(ctx->hFrags.data(), ctx->frags.p, 16ull * ctx->nFrags, hipMemcpyDeviceToHost));
    for (auto &f : ctx->hFrags) f.reserved = 0;
    *frags = ctx->hFrags.data(); *n_frags = ctx->nFrags; return 0;
}
}  // extern "C"
