// oracle_device.cpp -- TEST DOUBLE, CPU tier only (tests/test_pipeline_cpu.py builds it into a scratch directory; it is never part of libyaha_hip.so).
// The device entry points of include/yaha_hip.h that the command line's pipeline calls (ygpu_init / clone / upload / run / collect / destroy), answered by the
// oracle (oracle/hotpath.cpp) so that host/pipeline.cpp -- splitter, parsers, context threads, formatter pool, ordered writer, batch pool, -gpus N x -ctx M,
// failure handling -- can run as a whole on a box without a GPU, under ThreadSanitizer / AddressSanitizer, against the reference's golden SAM.
// What it proves is the HOST pipeline; the HIP path is proven by the -m gpu tests.
#include "../../yaha_amd/csrc/host/yaha_host.h"
#include "../../oracle/hotpath.h"
#include "../../yaha_amd/csrc/oqc_core.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <chrono>

struct ygpu_ctx { bool pfSet = false, pfDone = false; yoqc::Params pf; std::vector<uint32_t> thr, ss, sl; std::vector<uint32_t> fcs, fops; std::vector<ygpu_out_clump> fcl; ygpu_index_view V; ygpu_params P; std::vector<uint8_t> codes; std::vector<uint64_t> offs; uint32_t n = 0; yoracle_result res; bool have = false; int device = 0; std::string err;
                  // the post-filter's snapshot of a batch (results, reads) and what its collect reports: the stage may run on a thread of its own while the context runs the next batch
                  yoracle_result snap; std::atomic<bool> haveSnap{false}; std::vector<uint8_t> snapCodes; std::vector<uint64_t> snapOffs; uint32_t fReads = 0; ygpu_counters fCounters{}; };
static std::atomic<int> gInits(0), gRuns(0);
extern "C" {
int ygpu_init(int device, const ygpu_index_view *v, const ygpu_params *p, ygpu_ctx **out)
{
    *out = nullptr;
    const char *nd = getenv("YTEST_DEVICES"); if (device < 0 || device >= (nd ? atoi(nd) : 1)) return YGPU_ENODEV;
    ygpu_ctx *c = new ygpu_ctx; c->V = *v; c->P = *p; c->device = device; memset(&c->res, 0, sizeof c->res); memset(&c->snap, 0, sizeof c->snap); *out = c; gInits++;
    if (const char *ms = getenv("YTEST_INIT_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(ms)));
    return 0;
}
int ygpu_device_count(void) { const char *nd = getenv("YTEST_DEVICES"); return nd ? atoi(nd) : 1; }
int ygpu_clone(const ygpu_ctx *parent, ygpu_ctx **out);
int ygpu_init_multi(const int *devices, int n, int cpd, const ygpu_index_view *v, const ygpu_params *p, ygpu_ctx **out, int *rc_each)
{
    int rc = 0;
    for (int k = 0; k < n * cpd; k++) out[k] = nullptr;
    for (int k = 0; k < n; k++) { int r = ygpu_init(devices[k], v, p, &out[k * cpd]); for (int j = 1; j < cpd && r == 0; j++) r = ygpu_clone(out[k * cpd], &out[k * cpd + j]); if (rc_each) rc_each[k] = r; if (r && !rc) rc = r; }
    return rc;
}
int ygpu_clone(const ygpu_ctx *parent, ygpu_ctx **out) { ygpu_ctx *c = new ygpu_ctx; c->V = parent->V; c->P = parent->P; c->device = parent->device; memset(&c->res, 0, sizeof c->res); memset(&c->snap, 0, sizeof c->snap); *out = c; return 0; }
void ygpu_destroy(ygpu_ctx *c) { if (!c) return; if (c->have) yoracle_free_result(&c->res); if (c->haveSnap.load()) yoracle_free_result(&c->snap); delete c; }
// (YTEST_FREE_GB / YTEST_CTX_GB: what the double reports as free on the device and as held by a context -- drives the command line's "context left out" path)
int ygpu_memory(ygpu_ctx *, uint64_t *f, uint64_t *t, uint64_t *m)
{
    const char *fg = getenv("YTEST_FREE_GB"), *cg = getenv("YTEST_CTX_GB");
    if (const char *ms = getenv("YTEST_MEMORY_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(ms)));      // (a real hipSetDevice + hipMemGetInfo takes its time: the race of ADVICE r04 needed that)
    if (f) *f = fg ? (uint64_t)atoll(fg) << 30 : 1ull << 40; if (t) *t = 1ull << 40; if (m) *m = cg ? (uint64_t)atoll(cg) << 30 : 0;
    return 0;
}
static std::atomic<int> gParked(0);
int ygpu_park(ygpu_ctx *) { gParked++; return 0; }
static std::atomic<int> gPresized(0);
int ygpu_get_arena_profile(ygpu_ctx *, ygpu_arena_profile *p) { memset(p, 0, sizeof *p); p->n = 3; p->cap[0] = 1 << 20; return 0; }
int ygpu_presize(ygpu_ctx *, const ygpu_arena_profile *p) { if (p->n != 3) return YGPU_EINVAL; gPresized++; if (const char *ms = getenv("YTEST_PRESIZE_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(ms))); return 0; }
const char *ygpu_last_error(const ygpu_ctx *c) { return c ? c->err.c_str() : "no context"; }
int ygpu_upload(ygpu_ctx *c, const ygpu_read_batch *b)
{
    c->n = b->n_reads; c->offs.assign(b->offsets, b->offsets + b->n_reads + 1); c->codes.assign(b->codes + b->offsets[0], b->codes + b->offsets[b->n_reads]);
    const uint64_t o0 = c->offs[0]; for (auto &o : c->offs) o -= o0;
    return 0;
}
int ygpu_upload_nowait(ygpu_ctx *c, const ygpu_read_batch *b) { return ygpu_upload(c, b); }
int ygpu_run(ygpu_ctx *c)
{
    const int k = ++gRuns;
    if (const char *f = getenv("YTEST_FAIL_RUN")) if (k == atoi(f)) { c->err = "test double: injected device failure"; return YGPU_EINTERNAL; }
    if (c->have) { yoracle_free_result(&c->res); c->have = false; }
    ygpu_read_batch b{c->n, c->codes.data(), c->offs.data()};
    if (yoracle_run(&c->V, &c->P, &b, 1, &c->res) != 0) { c->err = "oracle failed"; return YGPU_EINTERNAL; }
    c->have = true; return 0;
}
int ygpu_collect(ygpu_ctx *c, ygpu_result_batch *r)
{
    if (!c->have) return YGPU_EINVAL;
    memset(r, 0, sizeof *r); r->n_reads = c->res.n_reads; r->clump_start = c->res.clump_start; r->clumps = c->res.clumps; r->ops = c->res.ops; r->n_clumps = c->res.n_clumps; r->n_ops = c->res.n_ops; r->counters = c->res.counters;
    return 0;
}
int ygpu_result_size(ygpu_ctx *c, uint64_t *nc, uint64_t *no) { if (!c->have) return YGPU_EINVAL; *nc = c->res.n_clumps; *no = c->res.n_ops; return 0; }
int ygpu_collect_into(ygpu_ctx *c, uint32_t *cs, ygpu_clump *cl, uint32_t *ops, ygpu_result_batch *r)
{
    if (!c->have) return YGPU_EINVAL;
    memcpy(cs, c->res.clump_start, 4ull * (c->res.n_reads + 1)); if (c->res.n_clumps) memcpy(cl, c->res.clumps, sizeof(ygpu_clump) * c->res.n_clumps); if (c->res.n_ops) memcpy(ops, c->res.ops, 4ull * c->res.n_ops);
    memset(r, 0, sizeof *r); r->n_reads = c->res.n_reads; r->clump_start = cs; r->clumps = cl; r->ops = ops; r->n_clumps = c->res.n_clumps; r->n_ops = c->res.n_ops; r->counters = c->res.counters;
    return 0;
}
// the post-filter stage: the same routine the device runs (oqc_core.h), here on the double's copy of the results
int ygpu_set_postfilter(ygpu_ctx *c, const ygpu_postfilter_params *p)
{
    c->thr.assign(p->bppThr, p->bppThr + p->bppN); c->ss.assign(p->seq_start, p->seq_start + p->n_seqs); c->sl.assign(p->seq_length, p->seq_length + p->n_seqs);
    yoqc::Params &P = c->pf; P.GOCost = c->P.GOCost; P.GECost = c->P.GECost; P.RCost = c->P.RCost; P.MScore = c->P.MScore; P.minNonOverlap = p->minNonOverlap; P.BPCost = p->BPCost; P.maxBPLog = p->maxBPLog; P.FBS = p->FBS;
    P.FBS_PSLength = p->FBS_PSLength; P.FBS_PSScore = p->FBS_PSScore; P.bppVmin = p->bppVmin; P.bppN = p->bppN; P.bppThr = c->thr.data(); c->pfSet = true; return 0;
}
int ygpu_postfilter_snapshot(ygpu_ctx *c)
{
    if (!c->have || !c->pfSet || c->haveSnap.load()) return YGPU_EINVAL;
    c->snap = c->res; memset(&c->res, 0, sizeof c->res); c->have = false;      // (the double hands its results over instead of copying them)
    c->snapCodes = c->codes; c->snapOffs = c->offs; c->pfDone = false;
    if (const char *ms = getenv("YTEST_SNAPSHOT_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(ms)));
    c->haveSnap.store(true); return 0;
}
int ygpu_postfilter(ygpu_ctx *c)
{
    if (!c->haveSnap.load()) { const int rc = ygpu_postfilter_snapshot(c); if (rc) return rc; }
    if (const char *ms = getenv("YTEST_POSTFILTER_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(ms)));      // (a stage that takes its time: the next batch's run overlaps it)
    struct Done { ygpu_ctx *c; ~Done() { yoracle_free_result(&c->snap); c->haveSnap.store(false); } } done{c};
    if (const char *f = getenv("YTEST_FAIL_POSTFILTER")) { static std::atomic<int> calls(0); if (++calls == atoi(f)) { c->err = "test double: injected post-filter failure"; return YGPU_EINTERNAL; } }
    const yoracle_result &R = c->snap; c->fcs.assign(1, 0); c->fcl.clear(); c->fops.clear(); c->fReads = R.n_reads; c->fCounters = R.counters;
    yoqc::Seqs G{c->ss.data(), c->sl.data(), (uint32_t)c->ss.size()};
    for (uint32_t r = 0; r < R.n_reads; r++) {
        const uint32_t b = R.clump_start[r], n = R.clump_start[r + 1] - b;
        const char *rawAbove = getenv("YTEST_RAW_ABOVE");               // reads with more clumps than this come back unfiltered and marked, as from the device stage
        if (n && rawAbove && n > (uint32_t)atoi(rawAbove)) {
            for (uint32_t k = 0; k < n; k++) {
                ygpu_out_clump f; f.c = R.clumps[b + k]; const uint32_t *src = R.ops + f.c.op_start; f.c.op_start = (uint32_t)c->fops.size(); c->fops.insert(c->fops.end(), src, src + f.c.n_ops);
                f.status = f.c.status; f.mapQuality = 255; f.numSecondaries = 0; f.matchedPrimary = 0; f.primaryCount = 0xFFFF; c->fcl.push_back(f);
            }
        } else if (n) {
            size_t pool = 0; for (uint32_t i = 0; i < n; i++) pool += 2 * (size_t)R.clumps[b + i].n_ops + 3;
            std::vector<yoqc::SortKey> keys(n); std::vector<int> stack(4 * (size_t)n + 8), pfx(n), path(n), pl(pool + 1); std::vector<yoqc::CNode> nodes(n), prim(n); std::vector<yoqc::PAttr> pa(n); std::vector<yoqc::OutRec> push(n), out(n);
            yoqc::Scratch S{keys.data(), stack.data(), 0x7fffffff, nullptr, nodes.data(), pfx.data(), path.data(), pl.data(), 0x7fffffff, nullptr, prim.data(), pa.data(), push.data()};
            int pc = 0; const int qlen = (int)(c->snapOffs[r + 1] - c->snapOffs[r]);
            const int m = yoqc::run(c->pf, G, R.clumps + b, (int)n, R.ops, qlen, c->snapCodes.data() + c->snapOffs[r], S, out.data(), &pc);
            for (int k = 0; k < m; k++) {
                ygpu_out_clump f; f.c = R.clumps[b + out[k].clump]; const uint32_t *src = R.ops + f.c.op_start; f.c.op_start = (uint32_t)c->fops.size(); c->fops.insert(c->fops.end(), src, src + f.c.n_ops);
                f.status = out[k].status; f.mapQuality = out[k].mapQuality; f.numSecondaries = out[k].numSecondaries; f.matchedPrimary = out[k].matchedPrimary; f.primaryCount = (uint16_t)pc; c->fcl.push_back(f);
            }
        }
        c->fcs.push_back((uint32_t)c->fcl.size());
    }
    c->pfDone = true; return 0;
}
int ygpu_postfilter_drop(ygpu_ctx *c) { if (!c) return YGPU_EINVAL; c->pfDone = false; return 0; }
int ygpu_filtered_size(ygpu_ctx *c, uint64_t *nc, uint64_t *no) { if (!c->pfDone) return YGPU_EINVAL; *nc = c->fcl.size(); *no = c->fops.size(); return 0; }
int ygpu_collect_filtered(ygpu_ctx *c, uint32_t *cs, ygpu_out_clump *cl, uint32_t *ops, ygpu_filtered_batch *r)
{
    if (!c->pfDone) return YGPU_EINVAL;
    memcpy(cs, c->fcs.data(), 4 * c->fcs.size()); if (!c->fcl.empty()) memcpy(cl, c->fcl.data(), sizeof(ygpu_out_clump) * c->fcl.size()); if (!c->fops.empty()) memcpy(ops, c->fops.data(), 4 * c->fops.size());
    memset(r, 0, sizeof *r); r->n_reads = c->fReads; r->clump_start = cs; r->clumps = cl; r->ops = ops; r->n_clumps = c->fcl.size(); r->n_ops = c->fops.size(); r->counters = c->fCounters;
    return 0;
}
static std::atomic<long> gHostAllocs(0);
void *ygpu_host_alloc(size_t n) { gHostAllocs++; return getenv("YTEST_NO_PINNED") ? nullptr : malloc(n ? n : 1); }      // (the double's "pinned" memory is plain memory; YTEST_NO_PINNED exercises the pipeline's fall-back to it)
void ygpu_host_free(void *p) { free(p); }
int  ygpu_selftest_primitives(ygpu_ctx *, uint32_t, uint32_t, int) { return YGPU_ENODEV; }
int  ygpu_trace_volume(ygpu_ctx *, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_inject_results(ygpu_ctx *, const ygpu_result_batch *) { return YGPU_ENODEV; }
int  ygpu_submit(ygpu_ctx *, const ygpu_read_batch *, ygpu_ticket *) { return YGPU_ENODEV; }
int  ygpu_poll(ygpu_ctx *, ygpu_ticket) { return YGPU_ENODEV; }
int  ygpu_wait(ygpu_ctx *, ygpu_ticket, ygpu_result_batch *) { return YGPU_ENODEV; }
int  ygpu_last_timing(ygpu_ctx *, float *, int *, const char *const **, const float **) { return YGPU_ENODEV; }
int  ygpu_seed_join(ygpu_ctx *, const ygpu_fragment **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_chain(ygpu_ctx *, const ygpu_fragment **, const uint32_t **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_dp_batch(ygpu_ctx *, const ygpu_dp_problem *, uint32_t, const ygpu_dp_result **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
int  ygpu_dp_batch_ex(ygpu_ctx *, const ygpu_dp_problem *, uint32_t, int, const ygpu_dp_result **, const uint32_t **, uint64_t *) { return YGPU_ENODEV; }
}
namespace yaha {
int  visibleDevices() { return 0; }
bool buildIndexDevice(int, const Genome &, int, int, int, IndexImage &, FILE *, std::string &err) { err = "test double: no device code"; return false; }
}
