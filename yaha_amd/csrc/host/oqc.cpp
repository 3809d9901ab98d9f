// oqc.cpp -- host post-filter: Optimal Query Coverage, filter-by-similarity, duplicate removal and mapping
// quality (reference GraphPath.cpp:294-1175).  Consumes the clump records the device hot path returns (QS->clumps
// order) and yields the clumps to print, in print order.  SURVEY.md 8(f)-1: stays on the host (tiny,
// pointer-chasing, double/log10 arithmetic) but has to be restated exactly -- including the per-read RNG that
// breaks sort ties -- for bit-exact SAM.
#include "yaha_host.h"
#include "../oqc_core.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace yaha {
namespace {
// Break point penalty of two alignments `distance` (> 10) reference bases apart on one sequence: (int)(min(log10(distance), maxBPLog) * BPCost + 0.5), :1014-1025.
// The graph loop asks for it for a hundred node pairs a read, and log10 is the dearest thing in that loop -- and not something the device stage should evaluate
// with another math library.  The value is a non-decreasing step function of the integer distance with at most maxBPLog * BPCost steps, so the steps are found
// once -- by bisection with the very same floating-point expression, so that every distance gets the value the expression gives -- and a call is a search
// among ~20 thresholds (oqc_core.h).
struct BPPTable {
    int bpCost = -1, mbpl = -1, vmin = 0; std::vector<uint32_t> thr;
    void build(int BPCost, int MBPL)
    {
        bpCost = BPCost; mbpl = MBPL; thr.clear(); vmin = yoqc::exactBPP(11, BPCost, MBPL);
        const int vmax = yoqc::exactBPP(0xFFFFFFFFu, BPCost, MBPL);
        for (int v = vmin + 1; v <= vmax; v++) {                        // smallest distance whose penalty is >= v
            uint64_t lo = 11, hi = 0xFFFFFFFFull;
            while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (yoqc::exactBPP((uint32_t)mid, BPCost, MBPL) >= v) hi = mid; else lo = mid + 1; }
            thr.push_back((uint32_t)lo);
        }
    }
};

void oqcParams(const Args &a, yoqc::Params &P, BPPTable &bpp)
{
    const bool useTable = a.BPCost >= 0 && a.maxBPLog >= 0 && a.maxBPLog * (long)a.BPCost <= 4096;     // a step function only for non-negative costs
    if (useTable && (bpp.bpCost != a.BPCost || bpp.mbpl != a.maxBPLog)) bpp.build(a.BPCost, a.maxBPLog);
    P.GOCost = a.GOCost; P.GECost = a.GECost; P.RCost = a.RCost; P.MScore = a.MScore; P.minNonOverlap = a.OQCMinNonOverlap; P.BPCost = a.BPCost; P.maxBPLog = a.maxBPLog;
        P.FBS = a.FBS ? 1 : 0;
    P.FBS_PSLength = a.FBS_PSLength; P.FBS_PSScore = a.FBS_PSScore; P.bppVmin = bpp.vmin; P.bppN = useTable ? (int)bpp.thr.size() : -1; P.bppThr = bpp.thr.data();
}
struct DupElem { int64_t clump; uint32_t SRO; int score; };             // dupArrayElem :1099-1104 (16 bytes like the reference's)
int cmpDup(const void *p1, const void *p2)
{ const DupElem *a = (const DupElem *)p1, *b = (const DupElem *)p2; if (a->SRO > b->SRO) return 1; if (a->SRO < b->SRO) return -1; return b->score - a->score; }
}  // namespace

void postFilter(const Args &a, const Genome &g, const Read &r, const ygpu_clump *cl, uint32_t n, const uint32_t *ops,
                std::vector<OutClump> &out, int &primaryCount)
{
    out.clear();
    auto mk = [&](int c) { OutClump o; o.c = cl[c]; o.ops = ops + cl[c].op_start; o.status = cl[c].status; return o; };
    const int qlen = r.len();
    if (!a.OQC) {                                                       // postFilterRemoveDups :1127-1174
        if (n < 2) { for (uint32_t i = 0; i < n; i++) out.push_back(mk((int)i)); return; }
        std::vector<DupElem> d(n);
        for (uint32_t i = 0; i < n; i++) d[i] = {(int64_t)i, cl[i].sro, (int)cl[i].totScore};
        qsort(d.data(), n, sizeof(DupElem), cmpDup);                    // same libc qsort the reference calls
        std::vector<int> keep;
        for (uint32_t i = 0; i < n; i++) {
            if (d[i].clump < 0) continue;
            const ygpu_clump &c1 = cl[d[i].clump];
            for (uint32_t j = i + 1; j < n; j++) {
                if (d[i].SRO < d[j].SRO) break;
                if (d[j].clump < 0) continue;
                const ygpu_clump &c2 = cl[d[j].clump];
                if (c1.sro == c2.sro && c1.sqo == c2.sqo && c1.eqo == c2.eqo && (c1.sro + c1.refLen) == (c2.sro + c2.refLen)
                    && ((c1.status ^ c2.status) & yoqc::stReversed) == 0) d[j].clump = -1;
            }
            keep.push_back((int)d[i].clump);
        }
        for (int k = (int)keep.size() - 1; k >= 0; k--) out.push_back(mk(keep[k]));    // pushes go to the head
        return;
    }
    if (n < 1) return;
    // Optimal Query Coverage + filter by similarity + mapping quality: oqc_core.h, the routine the device stage runs as well
    static thread_local BPPTable bpp; static thread_local std::vector<uint32_t> tlSeqStart, tlSeqLen; static thread_local const Genome *tlGenome = nullptr;
    yoqc::Params P; oqcParams(a, P, bpp);
    if (tlGenome != &g || tlSeqStart.size() != g.seqs.size()) { tlSeqStart.clear(); tlSeqLen.clear(); for (auto &sq : g.seqs) { tlSeqStart.push_back(sq.start);
        tlSeqLen.push_back(sq.length); } tlGenome = &g; }
    yoqc::Seqs Sq{tlSeqStart.data(), tlSeqLen.data(), (uint32_t)tlSeqStart.size()};
    static thread_local std::vector<yoqc::SortKey> tlKeys; static thread_local std::vector<int> tlStack, tlPfx, tlPath, tlPool;
        static thread_local std::vector<yoqc::CNode> tlNodes, tlPrim;
    static thread_local std::vector<yoqc::PAttr> tlPA; static thread_local std::vector<yoqc::OutRec> tlPush, tlOut;      // scratch reused from read to read
    size_t poolInts = 0; for (uint32_t i = 0; i < n; i++) poolInts += 2 * (size_t)cl[i].n_ops + 3;
    if (tlKeys.size() < n) { tlKeys.resize(n); tlStack.resize(4 * (size_t)n + 8); tlPfx.resize(n); tlPath.resize(n); tlNodes.resize(n); tlPrim.resize(n); tlPA.resize(n);
        tlPush.resize(n); tlOut.resize(n); }
    if (tlPool.size() < poolInts) tlPool.resize(poolInts);
    yoqc::Scratch S{tlKeys.data(), tlStack.data(), 0x7fffffff, nullptr, tlNodes.data(), tlPfx.data(), tlPath.data(), tlPool.data(), 0x7fffffff, nullptr, tlPrim.data(), tlPA.data(),
        tlPush.data()};
    const int m = yoqc::run(P, Sq, cl, (int)n, ops, qlen, r.fwdCodes.data(), S, tlOut.data(), &primaryCount);
    for (int k = 0; k < m; k++) {
        const yoqc::OutRec &o = tlOut[k]; OutClump oc; oc.c = cl[o.clump]; oc.ops = ops + cl[o.clump].op_start; oc.status = o.status; oc.mapQuality = o.mapQuality;
            oc.numSecondaries = o.numSecondaries; oc.matchedPrimary = o.matchedPrimary;
        out.push_back(oc);
    }
}
// the post-filter's parameters for oqc_core.h; `bpp` holds the break point table the parameters point into (kept by the caller)
void oqcParamsFromArgs(const Args &a, yoqc::Params &P, std::vector<uint32_t> &thr)
{
    BPPTable t; const bool useTable = a.BPCost >= 0 && a.maxBPLog >= 0 && a.maxBPLog * (long)a.BPCost <= 4096;
    if (useTable) t.build(a.BPCost, a.maxBPLog);
    thr = t.thr;
    P.GOCost = a.GOCost; P.GECost = a.GECost; P.RCost = a.RCost; P.MScore = a.MScore; P.minNonOverlap = a.OQCMinNonOverlap; P.BPCost = a.BPCost; P.maxBPLog = a.maxBPLog;
        P.FBS = a.FBS ? 1 : 0;
    P.FBS_PSLength = a.FBS_PSLength; P.FBS_PSScore = a.FBS_PSScore; P.bppVmin = t.vmin; P.bppN = useTable ? (int)thr.size() : -1; P.bppThr = thr.data();
}
}  // namespace yaha
