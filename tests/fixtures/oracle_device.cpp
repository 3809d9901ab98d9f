// oracle_device.cpp -- TEST DOUBLE, CPU tier only (tests/test_pipeline_cpu.py builds it into a scratch directory; it is never part of libyaha_hip.so).
// The device entry points of include/yaha_hip.h that the command line's pipeline calls (ygpu_init / clone / upload / run / collect / destroy), answered by the
// oracle (oracle/hotpath.cpp) so that host/pipeline.cpp -- splitter, parsers, context threads, formatter pool, ordered writer, batch pool, -gpus N x -ctx M,
// failure handling -- can run as a whole on a box without a GPU, under ThreadSanitizer / AddressSanitizer, against the reference's golden SAM.
// What it proves is the HOST pipeline; the HIP path is proven by the -m gpu tests.
#include "../../yaha_amd/csrc/host/yaha_host.h"
#include "../../oracle/hotpath.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <chrono>

struct ygpu_ctx { ygpu_index_view V; ygpu_params P; std::vector<uint8_t> codes; std::vector<uint64_t> offs; uint32_t n = 0; yoracle_result res; bool have = false; int device = 0; std::string err; };
static std::atomic<int> gInits(0), gRuns(0);
extern "C" {
int ygpu_init(int device, const ygpu_index_view *v, const ygpu_params *p, ygpu_ctx **out)
{
    *out = nullptr;
    const char *nd = getenv("YTEST_DEVICES"); if (device < 0 || device >= (nd ? atoi(nd) : 1)) return YGPU_ENODEV;
    ygpu_ctx *c = new ygpu_ctx; c->V = *v; c->P = *p; c->device = device; memset(&c->res, 0, sizeof c->res); *out = c; gInits++;
    if (const char *ms = getenv("YTEST_INIT_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(ms)));
    return 0;
}
int ygpu_clone(const ygpu_ctx *parent, ygpu_ctx **out) { ygpu_ctx *c = new ygpu_ctx; c->V = parent->V; c->P = parent->P; c->device = parent->device; memset(&c->res, 0, sizeof c->res); *out = c; return 0; }
void ygpu_destroy(ygpu_ctx *c) { if (!c) return; if (c->have) yoracle_free_result(&c->res); delete c; }
const char *ygpu_last_error(const ygpu_ctx *c) { return c ? c->err.c_str() : "no context"; }
int ygpu_upload(ygpu_ctx *c, const ygpu_read_batch *b)
{
    c->n = b->n_reads; c->offs.assign(b->offsets, b->offsets + b->n_reads + 1); c->codes.assign(b->codes + b->offsets[0], b->codes + b->offsets[b->n_reads]);
    const uint64_t o0 = c->offs[0]; for (auto &o : c->offs) o -= o0;
    return 0;
}
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
static std::atomic<long> gHostAllocs(0);
void *ygpu_host_alloc(size_t n) { gHostAllocs++; return getenv("YTEST_NO_PINNED") ? nullptr : malloc(n ? n : 1); }      // (the double's "pinned" memory is plain memory; YTEST_NO_PINNED exercises the pipeline's fall-back to it)
void ygpu_host_free(void *p) { free(p); }
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
