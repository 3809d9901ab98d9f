// pipeline.cpp -- the batching host that replaces the reference's per-read loop (Query.c:255-543) and its
// run driver (Query.c:551-709): read a batch of queries, hand it to the device hot path through the C-ABI
// (ygpu_upload / ygpu_run / ygpu_collect), post-filter and format on host threads, write in input order.
// With -gpus N one context per device is driven by its own host thread; batches are dealt round-robin and the
// output is re-ordered by batch ticket, so the text equals a `-t 1` run of the reference.
#include "yaha_host.h"
#include "../oqc_core.h"
#include <cstring>
#include <cstdlib>
#include <thread>
#include <chrono>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <map>
#include <deque>
#include <memory>
#include <sys/stat.h>
#include <sched.h>

using namespace yaha;

struct yaha_session {
    Args args; Genome genome; IndexFile index; ReadReader reader; std::string err, header; Text text;
    std::vector<uint32_t> pfThr, pfSeqStart, pfSeqLen;               // what yaha_session_postfilter_params points into
    std::vector<Read> reads; std::vector<uint8_t> codes; std::vector<uint64_t> offsets;
    bool readerOpen = false;
};

namespace yaha {

static bool fileNewerThan(const std::string &f1, const std::string &f2)      // FileHelpers.c:127-146
{
    struct stat s1, s2;
    if (stat(f1.c_str(), &s1) != 0) return false;
    if (stat(f2.c_str(), &s2) != 0) return true;
    return s1.st_mtime > s2.st_mtime;
}

int runIndex(Args &a, FILE *log)                                              // Main.c:587-628
{
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool timing = getenv("YAHA_TIMING") != nullptr; double t0 = now();
    if (a.wordLen > 15) { fprintf(log, "Word Length (-L) for index creation is currently restricted to < 16.\n"); return 1; }
    if (a.skipDist < 1 || a.skipDist > a.wordLen) { fprintf(log, "Skip Distance (-S) for index creation must be between 1 and WordLength (inclusive).\n"); return 1; }
    size_t dot = a.gfileName.rfind('.'); std::string ext = dot == std::string::npos ? "" : a.gfileName.substr(dot);
    bool fasta = (ext == ".fna" || ext == ".fa" || ext == ".fasta");
    if (!fasta && ext != ".nib2") { fprintf(log, "Expecting a \".fa\", \".fna\", \".fasta\", or \".nib2\" genome file.\n"); return 1; }
    char oext[32]; snprintf(oext, sizeof oext, ".X%02d_%02d_%05dS", a.wordLen, a.skipDist, a.maxHits);
    std::string root = a.gfileName.substr(0, dot), nib2 = fasta ? root + ".nib2" : a.gfileName, xfile = root + oext, err;
    if (fasta) {
        if (fileNewerThan(a.gfileName, nib2)) {
            fprintf(log, "Compressing %s into %s.\n", a.gfileName.c_str(), nib2.c_str());
            std::vector<uint8_t> img;
            if (!compressFasta(a.gfileName.c_str(), img, err) || !writeFile(nib2.c_str(), img.data(), img.size(), err)) { fprintf(log, "%s\n", err.c_str()); return 1; }
            fprintf(log, "Finished compressing %s, now forming index.\n", nib2.c_str());
            if (timing) { fprintf(log, "[yaha] compress %.1f s\n", now() - t0); t0 = now(); }
        } else fprintf(log, "%s already exists.  Creating the index file.\n", nib2.c_str());
    }
    fprintf(log, "Creating index file %s.\n", xfile.c_str());
    Genome g; if (!loadNib2(nib2.c_str(), g, err)) { fprintf(log, "%s\n", err.c_str()); return 1; }
    IndexImage image;
    // count -> scan -> fill -> order -> sample on the GPU (device/index_build.hip) when there is one; the host
    // builder (the reference's three passes, formats.cpp) otherwise.  Both produce the reference's file byte for byte.
    bool onDevice = !a.cpuIndex && a.device >= 0 && getenv("YAHA_CPU_INDEX") == nullptr && visibleDevices() > a.device;
    if (onDevice) {
        fprintf(log, "Building the index on GPU %d.\n", a.device);
        if (!buildIndexDevice(a.device, g, a.wordLen, a.skipDist, a.maxHits, image, log, err)) {
            // (a shared or smaller device may lack the ~40 GB an hg18-scale build keeps resident: the host builder writes the same file, slower)
            fprintf(log, "Index build on the GPU failed (%s): building on the host instead.\n", err.c_str());
            image.release(); onDevice = false;
        }
    }
    if (!onDevice && !buildIndex(g, a.wordLen, a.skipDist, a.maxHits, image, log)) { fprintf(log, "Insufficient memory to build the index.\n"); return 1; }
    if (timing) { fprintf(log, "[yaha] index image (%.2f GB) built in %.1f s %s\n", image.words * 4 / 1e9, now() - t0, onDevice ? "on the GPU" : "on the host"); t0 = now(); }
    if (!writeFile(xfile.c_str(), image.p, image.words * 4, err)) { fprintf(log, "%s\n", err.c_str()); return 1; }
    if (timing) fprintf(log, "[yaha] index file written in %.1f s\n", now() - t0);
    fprintf(log, "Index %s created.\n", xfile.c_str());
    return 0;
}

int runCompress(Args &a, FILE *log)                                           // Main.c:572-577 -> compressFile
{
    std::string err; std::vector<uint8_t> img;
    if (!compressFasta(a.gfileName.c_str(), img, err) || !writeFile(a.ofileName.c_str(), img.data(), img.size(), err)) { fprintf(log, "%s\n", err.c_str()); return 1; }
    return 0;
}
int runUncompress(Args &a, FILE *log)                                         // uncompressFile, Compress.c:337-397
{
    std::string err; Genome g;
    if (!loadNib2(a.gfileName.c_str(), g, err)) { fprintf(log, "%s\n", err.c_str()); return 1; }
    if (a.verbose) fprintf(log, "Read in %zu reference sequences from %s.\n", g.seqs.size(), a.gfileName.c_str());
    std::string out; size_t tot = 0; for (auto &s : g.seqs) tot += s.name.size() + 2 + s.length + s.length / 50 + 1; out.reserve(tot + 16);
    for (auto &s : g.seqs) {
        out += '>'; out += s.name; out += '\n';
        int col = 0;
        for (uint32_t j = 0; j < s.length; j++) { if (col == 50) { out += '\n'; col = 0; } out += kFourBitChars[get4(g.bases, s.start + j)]; col++; }
        if (col != 0) out += '\n';
    }
    if (!writeFile(a.ofileName.c_str(), out.data(), out.size(), err)) { fprintf(log, "%s\n", err.c_str()); return 1; }
    return 0;
}

static bool sessionLoad(yaha_session *s)
{
    Args &a = s->args;
    if (!loadNib2(a.gfileName.c_str(), s->genome, s->err)) return false;
    if (!loadIndex(a.xfileName.c_str(), s->index, s->err)) return false;
    a.wordLen = s->index.wordLen;                                            // Query.c:603-610
    if (s->index.maxHits < a.maxHits) {
        fprintf(stderr, "WARNING: Index file made with maxHits of %d, while %d specified for this query run.\nMimimum of two (%d) will be used.\n", s->index.maxHits, a.maxHits,
            s->index.maxHits);
        a.maxHits = s->index.maxHits;
    }
    s->reader.maxQueryLength = a.maxQueryLength; s->reader.wordLen = a.wordLen;
    if (!s->reader.open(a.qfileName.c_str(), s->err)) return false;
    s->readerOpen = true; a.fastq = s->reader.fastq;
    s->header = samHeader(a, s->genome);
    return true;
}

// OQC/FBS filter + SAM text of one batch.  nt > 1: the reads of the batch are cut into nt contiguous ranges, one thread each (the ctypes/Session path, one
// batch at a time); the command line formats whole batches on a pool of threads instead (runQueries) and passes nt = 1.
static void formatRange(const yaha_session *s, const ygpu_result_batch *r, uint32_t i0, uint32_t i1, Text &text, std::vector<OutClump> &oc)
{
    const Args &a = s->args;
    for (uint32_t i = i0; i < i1; i++) {
        uint32_t c0 = r->clump_start[i], c1 = r->clump_start[i + 1]; int primaryCount = 0;
        postFilter(a, s->genome, s->reads[i], r->clumps + c0, c1 - c0, r->ops, oc, primaryCount);
        for (auto &o : oc) printClump(a, s->genome, s->reads[i], o, primaryCount, text);
    }
}
// SAM text of a batch whose post-filter ran on the device (ygpu_postfilter): the clumps arrive in print order with the filter's fields set
static void formatFiltered(const yaha_session *s, const ygpu_filtered_batch *r, Text &text)
{
    const Args &a = s->args; text.clear();
    std::vector<ygpu_clump> raw; std::vector<OutClump> oc;
    for (uint32_t i = 0; i < r->n_reads; i++) {
        const uint32_t k0 = r->clump_start[i], k1 = r->clump_start[i + 1];
        if (k1 > k0 && r->clumps[k0].primaryCount == 0xFFFFu) {
            // a read the device left unfiltered (more clumps than its stage takes, device/oqc_stage.h): all its clumps in the hot path's order -- the host's filter, same routine
            raw.resize(k1 - k0); for (uint32_t k = k0; k < k1; k++) raw[k - k0] = r->clumps[k].c;
            int primaryCount = 0;
            postFilter(a, s->genome, s->reads[i], raw.data(), k1 - k0, r->ops, oc, primaryCount);
            for (auto &o : oc) printClump(a, s->genome, s->reads[i], o, primaryCount, text);
            continue;
        }
        for (uint32_t k = k0; k < k1; k++) {
            const ygpu_out_clump &f = r->clumps[k];
            OutClump o; o.c = f.c; o.ops = r->ops + f.c.op_start; o.status = f.status; o.mapQuality = f.mapQuality; o.numSecondaries = f.numSecondaries;
                o.matchedPrimary = f.matchedPrimary;
            printClump(a, s->genome, s->reads[i], o, (int)f.primaryCount, text);
        }
    }
}
static void formatBatch(yaha_session *s, const ygpu_result_batch *r, Text &text, int nt)
{
    const uint32_t n = r->n_reads;
    text.clear();
    // straight into the batch's text (its buffer is reused from batch to batch)
    if (nt <= 1 || n < 2 * (uint32_t)nt) { std::vector<OutClump> oc; formatRange(s, r, 0, n, text, oc); return; }
    std::vector<Text> parts(nt);
    auto work = [&](int t) { std::vector<OutClump> oc; formatRange(s, r, (uint32_t)((uint64_t)n * t / nt), (uint32_t)((uint64_t)n * (t + 1) / nt), parts[t], oc); };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0); for (auto &x : th) x.join();
    size_t tot = 0; for (auto &p : parts) tot += p.len;
    text.room(tot); for (auto &p : parts) text.append(p.p, p.len);
}

namespace {
// a bounded hand-over queue between pipeline stages
template <class T> struct StageQueue {
    std::mutex mu; std::condition_variable cvPush, cvPop; std::deque<T> q; size_t cap; int producers;
    StageQueue(size_t c, int np) : cap(c), producers(np) {}
    void push(T &&v) { std::unique_lock<std::mutex> lk(mu); cvPush.wait(lk, [&] { return q.size() < cap; }); q.push_back(std::move(v)); cvPop.notify_one(); }
    bool pop(T &v) { std::unique_lock<std::mutex> lk(mu); cvPop.wait(lk, [&] { return !q.empty() || producers == 0; }); if (q.empty()) return false; v = std::move(q.front());
        q.pop_front(); cvPush.notify_one(); return true; }
    void producerDone() { std::lock_guard<std::mutex> lk(mu); if (--producers == 0) cvPop.notify_all(); }
};
}  // namespace

// CPUs this process may really use: the smaller of the affinity mask and the control group's CPU quota (cpu.max; v1: cfs_quota/cfs_period).  A container that
// shows 256 hardware threads and has a quota of 16 CPUs is throttled -- every thread of it, the ones that feed the GPUs included, stopped for the rest of each
// 100 ms period -- as soon as more than 16 threads are busy: the pools below are sized from this number, not from hardware_concurrency().
int effectiveCpus()
{
    int n = (int)std::thread::hardware_concurrency(); if (n < 1) n = 1;
    cpu_set_t set; if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c >= 1 && c < n) n = c; }
    long long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[64] = ""; if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q); fclose(f); }
    else {
        if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f1, "%lld", &quota) != 1) quota = -1; fclose(f1); }
        if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &period) != 1) period = -1; fclose(f2); }
    }
    if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
    if (const char *e = getenv("YAHA_CPUS")) { const int c = atoi(e); if (c >= 1) n = c; }
    return n;
}

int runQueries(Args &a, FILE *log)
{
    const bool timing = getenv("YAHA_TIMING") != nullptr, stats = timing || getenv("YAHA_STATS") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tEnter = now(); double tStep = tEnter;
    std::unique_ptr<yaha_session> S(new yaha_session); S->args = a;
    if (!sessionLoad(S.get())) { fprintf(log, "%s\n", S->err.c_str()); return 1; }
    if (timing) { fprintf(stderr, "[yaha] index and input opened in %.1f ms\n", now() - tStep); tStep = now(); }
    Args &A = S->args;
    FILE *out = (A.ofileName == "stdout") ? stdout : fopen(A.ofileName.c_str(), "w");
    if (!out) { fprintf(log, "Failure to open output file: %s.\n", A.ofileName.c_str()); return 1; }
    setvbuf(out, nullptr, _IONBF, 0);                                       // whole batches are written with one call each
    if (fputs(S->header.c_str(), out) < 0) { fprintf(log, "Failure writing the output file.\n"); return 1; }
    ygpu_params P; paramsFromArgs(A, P);
    ygpu_index_view V; yaha_session_index_view(S.get(), &V);
    // -gpus N devices x -ctx M contexts per device (default 3 -- four contexts of ~60 GB beside the image leave a later, heavier batch no memory to grow into: measured slower;
    // while one context's batch is in a latency-bound device stage the others'
    // batches compute).  Contexts of one device share its index image.
    const int perDev = std::max(1, A.ctxPerGpu), nDev = std::max(1, A.gpus), ngpu = nDev * perDev;
    std::vector<ygpu_ctx *> ctx(ngpu, nullptr);
    // Six stages, batches handed over through bounded queues, a ticket ordering the output (= the reference's -t 1 order):
    //   splitter   (1 thread)          record boundaries of the memory-mapped / block-read input (memchr; reader.cpp) -- the only serial part;
    //   parsers    (a few threads)     id, sequence, quality, codes and the skip rules of one batch of records;
    //   contexts   (1 thread each)     upload, run the hot path, results by DMA into the batch's page-locked buffers (asleep while the device works);
    //   formatters                     OQC/FBS filter and SAM text of one whole batch each, into the batch's own text buffer;
    //   writer     (1 thread)          batches in ticket order, one write each; the batch object -- reads, codes, results, text, all with their capacity --
    //                                  then goes back to the pool the splitter takes from, so that the steady state allocates nothing and maps no new pages.
    // The device never waits for parsing or formatting unless those stages as a whole are slower than it.  The first failure (a device error, a write
    // error) stops the run: nothing after the last complete batch before it is written, and the exit code is 1.
    // Result storage of a batch: page-locked host memory (ygpu_host_alloc) the device copies into directly (ygpu_collect_into) -- no staging copy by the context
    // thread, no copy out of the context afterwards (two passes over ~9 KB a read otherwise); plain memory when it cannot be locked.  Grows, never shrinks.
    struct ResBuf {
        void *p = nullptr; size_t cap = 0; bool pinned = false;
        ~ResBuf() { drop(); }
        void drop() { if (p) { if (pinned) ygpu_host_free(p); else free(p); } p = nullptr; cap = 0; }
        bool ensure(size_t bytes)
        {
            if (bytes <= cap) return true;
            drop();
            const size_t c = bytes + bytes / 4 + 4096;
            p = ygpu_host_alloc(c); pinned = p != nullptr;
            if (!p) p = malloc(c);
            cap = p ? c : 0; return p != nullptr;
        }
    };
    struct Batch { uint64_t ticket = 0; std::vector<Span> spans; std::vector<Read> reads; size_t nReads = 0; std::vector<uint8_t> codes; std::vector<uint64_t> offsets;
                   ResBuf clumpStart, ops, clumps; uint64_t nClumps = 0, nOps = 0; bool filtered = false; Text text; double tRead = 0, tDev = 0, tFmt = 0; };
    typedef std::unique_ptr<Batch> BatchP;
    struct Pool { std::mutex mu; std::vector<BatchP> free; BatchP get() { { std::lock_guard<std::mutex> lk(mu); if (!free.empty()) { BatchP b = std::move(free.back());
        free.pop_back(); return b; } } return BatchP(new Batch); }
                  void put(BatchP &&b) { std::lock_guard<std::mutex> lk(mu); free.push_back(std::move(b)); } } pool;
    // Thread counts follow the CPUs the process may use (effectiveCpus: affinity and the control group's quota).  -t is the reference's thread count and is
    // echoed in @PG; given explicitly (> 1) it is the number of formatter threads, otherwise every CPU that is not a parser, the splitter or the writer is one.
    const int cpus = effectiveCpus();
    // (a parser thread does 1.4 M reads/s, a formatter 0.1 M with the host's post-filter and 0.65 M when the filter ran on the device: with several devices
    // behind a small CPU quota the parsers are what must not run short -- one for every two devices)
    const int nParse = std::max(1, std::min(8, std::min(std::max((cpus + 7) / 10, (nDev + 1) / 2), std::max(1, cpus / 3)))), nFmt = A.numThreads > 1 ? A.numThreads : std::max(1,
        cpus - nParse - 2);
    StageQueue<BatchP> parseQ((size_t)nParse + 2, 1), inQ((size_t)ngpu + 2, nParse), fmtQ((size_t)nFmt + (size_t)ngpu, ngpu + nParse), outQ((size_t)nFmt + 4, nFmt);
    std::atomic<bool> stop(false); std::atomic<int> rcAll(0); std::atomic<uint64_t> ticketsIssued(0), ticketsWritten(0);
    auto fail = [&](const char *what) { if (!stop.exchange(true)) fprintf(log, "%s -- stopping; the output ends with the last batch completed before this one.\n", what); rcAll = 1;
        };
    // batches follow the read length: about 16 M bases each (16 384 reads of 1 kbp, 1 600 of 10 kbp, 65 536 of 100 bp), unless -batch gives a read count
    const size_t maxReads = A.batchReads > 0 ? (size_t)A.batchReads : 65536, maxBases = A.batchReads > 0 ? ~(size_t)0 : ((size_t)16 << 20);
    auto splitter = [&]() {
        uint64_t ticket = 0;
        while (!stop) {
            BatchP b = pool.get();
            if (S->reader.split.nextSpans(maxReads, maxBases, b->spans) == 0) break;
            b->ticket = ticket++; ticketsIssued = ticket;
            parseQ.push(std::move(b));
        }
        parseQ.producerDone();
    };
    auto parser = [&]() {
        BatchP b;
        while (parseQ.pop(b)) {
            b->nReads = 0;
            if (!stop) {
                const double t0 = now();
                if (b->reads.size() < b->spans.size()) b->reads.resize(b->spans.size());      // Read objects keep their strings' capacity from batch to batch
                b->offsets.assign(1, 0); size_t bases = 0, k = 0;
                for (auto &sp : b->spans) if (parseSpan(sp, S->reader.fastq, S->reader.maxQueryLength, S->reader.wordLen, b->reads[k])) { bases += b->reads[k].fwdCodes.size(); k++;
                    }
                b->nReads = k; b->codes.resize(bases); size_t o = 0;
                for (size_t i = 0; i < k; i++) { const Read &rd = b->reads[i]; memcpy(b->codes.data() + o, rd.fwdCodes.data(), rd.fwdCodes.size()); o += rd.fwdCodes.size();
                    b->offsets.push_back(o); }
                b->tRead = now() - t0;
            }
            b->spans.clear();                                                  // lets go of the input chunk
            // a batch whose records were all skipped still takes its place in the output order
            if (b->nReads == 0) { b->nClumps = b->nOps = 0; fmtQ.push(std::move(b)); } else inQ.push(std::move(b));
        }
        inQ.producerDone(); fmtQ.producerDone();
    };
    // Contexts are made by the threads that use them, all devices at once (an index image of 16.7 GB takes 0.65 s to reach a device; eight in a row would be
    // five seconds), while the splitter and the parsers already work on the first batches.  The first context of a device uploads the image, the others share it.
    // A context's first batch allocates its device buffers (a hundred hipMalloc calls, each of which waits for the device to go idle): first batches run one at
    // a time and hold back the other contexts' new batches meanwhile, instead of fighting their kernels for every allocation (0.6 s per context otherwise, measured).
    // The post-filter (OQC, filter by similarity, mapping quality) runs on the device behind the hot path (ygpu_postfilter; the same routine as the host's,
    // oqc_core.h): the results that cross PCIe and reach the formatters are the clumps that get printed.  The host filter remains for -OQC N (duplicate
    // removal only), for break point costs that are no step function, and on request (-dpf N, YAHA_HOST_OQC=1).
    yoqc::Params oqP; std::vector<uint32_t> oqThr, oqSeqStart, oqSeqLen; oqcParamsFromArgs(A, oqP, oqThr);
    for (auto &sq : S->genome.seqs) { oqSeqStart.push_back(sq.start); oqSeqLen.push_back(sq.length); }
    // (-MNO below 1: the graph's successor test then also lets a node relax itself and nodes before it, which the device stage's one-successor-per-lane step does not order
    // the way the sequential loop does, GraphPath.cpp:633-700 -- those runs keep the host filter, as ygpu_set_postfilter insists)
    const bool deviceFilter = A.OQC && A.devicePostFilter && oqP.bppN >= 0 && oqP.minNonOverlap >= 1 && getenv("YAHA_HOST_OQC") == nullptr;
    ygpu_postfilter_params PF; memset(&PF, 0, sizeof PF);
    PF.minNonOverlap = oqP.minNonOverlap; PF.BPCost = oqP.BPCost; PF.maxBPLog = oqP.maxBPLog; PF.FBS = oqP.FBS; PF.FBS_PSLength = oqP.FBS_PSLength;
        PF.FBS_PSScore = oqP.FBS_PSScore;
    PF.bppVmin = oqP.bppVmin; PF.bppN = std::max(0, oqP.bppN); PF.bppThr = oqThr.data(); PF.n_seqs = (uint32_t)oqSeqStart.size(); PF.seq_start = oqSeqStart.data();
        PF.seq_length = oqSeqLen.data();
    // ready: 0 = image not there yet, 1 = there, -1 = failed
    struct Warm { std::mutex mu, first; std::condition_variable cv; int firstRunning = 0; int ready = 0; uint64_t footprint = 0;
        bool claimed = false, measured = false, haveProfile = false; ygpu_arena_profile profile; };
    std::vector<std::unique_ptr<Warm>> warm; for (int k = 0; k < nDev; k++) warm.emplace_back(new Warm);
    std::atomic<int> ctxUp(0), parked(0); double tCtxUp = 0;
    std::vector<std::atomic<uint64_t>> devReads(nDev); for (auto &x : devReads) x = 0;       // reads each device took (the stats line: do all devices pull their weight?)
    // where a context thread's time goes, batches after a context's first (the stats line; microseconds): upload, run, waiting for the filter thread, snapshot; and the filter
    // thread's post-filter + collect
    std::atomic<uint64_t> usUpload(0), usRun(0), usWaitFilter(0), usSnapshot(0), usFilter(0), usIdle(0), nLater(0);
    // The index image reaches the devices through ONE call (ygpu_init_multi): the first device takes it from the host, the others from their neighbour over xGMI,
    // piece by piece -- the reference maps its index once for all threads (Query.c:565-626); N uploads of 16.7 GB at once would share the host's memory instead.
    std::vector<int> leadRc(nDev, 0);
    // logical GPU k of -gpus N is HIP device -device + k, or the k-th entry of YAHA_DEVICES (a comma-separated list; the same device may appear twice: two
    // images on one device, which is how the multi-device path is exercised on a box with one GPU)
    std::vector<int> devs(nDev); for (int k = 0; k < nDev; k++) devs[k] = A.device + k;
    if (const char *e = getenv("YAHA_DEVICES")) { int k = 0; for (const char *q = e; *q && k < nDev; k++) { devs[k] = atoi(q); while (*q && *q != ',') q++; if (*q == ',') q++; } }
    auto bringUpDevices = [&]() {
        const int rc = ygpu_init_multi(devs.data(), nDev, perDev, &V, &P, ctx.data(), leadRc.data());      // ctx[k * perDev + j] = context j of device k
        for (int k = 0; k < nDev; k++) if (rc != 0 && !ctx[k * perDev]) leadRc[k] = rc;
        // (the device that failed is reported before the ones that were merely not started because of it)
        if (rc != 0) for (int k = 0; k < nDev; k++) { ygpu_ctx *c = ctx[k * perDev]; if (leadRc[k] != 0 && c && strncmp(ygpu_last_error(c), "not started", 11) != 0) { char m[512];
            snprintf(m, sizeof m, "ygpu_init(device %d) failed: %d %s", devs[k], leadRc[k], ygpu_last_error(c)); fail(m); break; } }
    };
    auto device = [&](int d) {
        BatchP b; bool first = true; Warm &W = *warm[d / perDev]; const int dev = devs[d / perDev], leadCtx = d - d % perDev;
        int rc0;
        if (d == 0) {
            bringUpDevices();
            for (int k = 0; k < nDev; k++) { Warm &Wk = *warm[k]; { std::lock_guard<std::mutex> lk(Wk.mu); Wk.ready = leadRc[k] == 0 ? 1 : -1; } Wk.cv.notify_all(); }
        }
        { std::unique_lock<std::mutex> lk(W.mu); W.cv.wait(lk, [&] { return W.ready != 0; }); rc0 = W.ready == 1 ? 0 : (leadRc[d / perDev] ? leadRc[d / perDev] : YGPU_ENODEV); }
        if (rc0 == 0 && deviceFilter) rc0 = ygpu_set_postfilter(ctx[d], &PF);
        if (rc0 != 0) { char m[512];
            snprintf(m, sizeof m, "ygpu_init(device %d) failed: %d %s", dev, rc0, ctx[d] ? ygpu_last_error(ctx[d]) : (d == leadCtx ? "" : "(the device's first context failed)"));
            fail(m); }
        if (++ctxUp == ngpu) { tCtxUp = now();
            if (timing) fprintf(stderr, "[yaha] %d device contexts up (index image on %d device%s) %.1f ms after start\n", ngpu, nDev, nDev > 1 ? "s" : "", tCtxUp - tEnter); }
        // A context's FIRST batch.  The first context of a device to get here leads: its first batch allocates its arenas, and what it then holds is the measure of what
        // a context needs.  The others wait for that measure BEFORE they take a batch, then go one at a time: a context that would start with less free memory than 0.9
        // of the measure is left out -- parked without ever holding a batch (round 4 took the batch first and pushed it back: a queue whose other consumers had already
        // seen it empty and finished never delivered it, and the output ended early with exit code 0).  Its share of the device's memory budget goes back to the
        // contexts that run (ygpu_park).
        bool lead = false;
        auto measured = [&]() { { std::lock_guard<std::mutex> lk(W.mu); W.measured = true; } W.cv.notify_all(); };
        // The post-filter of batch n runs on a thread of its own, on the SNAPSHOT this thread takes of the batch's results (ygpu_postfilter_snapshot), while this
        // thread is already uploading and running batch n + 1: the stage is a chain of waits for a few slow reads -- 5 to 9 ms in which the device is nearly idle -- and
        // behind the hot path it made every batch of the context that much longer (the command line's steady rate 0.85 of the hot path's, round 4).  A context's first
        // batch keeps the sequential order (its arenas are measured after it).  YAHA_SERIAL_FILTER=1: every batch in the sequential order.
        struct FilterSide { std::mutex mu; std::condition_variable cv; BatchP b; double t0 = 0; bool busy = false, quit = false; std::thread th; } F;
        const bool overlapFilter = deviceFilter && getenv("YAHA_SERIAL_FILTER") == nullptr;
        // post-filter (of the snapshot, or of the context's last run) and its results into the batch's own buffers
        auto collectFiltered = [&](BatchP &fb, ygpu_result_batch &res) -> int {
            uint64_t nc = 0, no = 0;
            int rc = ygpu_postfilter(ctx[d]); if (rc == 0) rc = ygpu_filtered_size(ctx[d], &nc, &no); if (rc != 0) return rc;
            if (!fb->clumpStart.ensure(4 * (fb->nReads + 1)) || !fb->clumps.ensure(sizeof(ygpu_out_clump) * nc) || !fb->ops.ensure(4 * no)) return YGPU_ENOMEM;
            ygpu_filtered_batch fr; rc = ygpu_collect_filtered(ctx[d], (uint32_t *)fb->clumpStart.p, (ygpu_out_clump *)fb->clumps.p, (uint32_t *)fb->ops.p, &fr);
            res.n_clumps = fr.n_clumps; res.n_ops = fr.n_ops; return rc;
        };
        auto deliver = [&](BatchP &fb, int rc, const ygpu_result_batch &res, double t0) {
            if (rc != 0) { char m[512];
                snprintf(m, sizeof m, "context %d: hot path failed (%d): %s", d, rc, rc == YGPU_ENOMEM && !fb->ops.p ? "host memory for the results" : ygpu_last_error(ctx[d]));
                fail(m); fb->nReads = 0; fmtQ.push(std::move(fb)); return; }
            fb->nClumps = res.n_clumps; fb->nOps = res.n_ops;
            fb->tDev = now() - t0; devReads[d / perDev] += fb->nReads;
            fmtQ.push(std::move(fb));
        };
        auto filterLoop = [&]() {
            for (;;) {
                BatchP fb; double t0;
                { std::unique_lock<std::mutex> lk(F.mu); F.cv.wait(lk, [&] { return F.quit || F.b; }); if (!F.b) return; fb = std::move(F.b); t0 = F.t0; }
                ygpu_result_batch res; memset(&res, 0, sizeof res);
                const double f0 = now();
                const int rc = stop ? 0 : collectFiltered(fb, res);
                usFilter += (uint64_t)((now() - f0) * 1e3);
                if (stop && rc == 0) { fb->nReads = 0; fmtQ.push(std::move(fb)); } else deliver(fb, rc, res, t0);
                { std::lock_guard<std::mutex> lk(F.mu); F.busy = false; } F.cv.notify_all();
            }
        };
        auto filterIdle = [&]() { std::unique_lock<std::mutex> lk(F.mu); F.cv.wait(lk, [&] { return !F.busy; }); };
        for (;;) {
            std::unique_lock<std::mutex> one(W.first, std::defer_lock);
            if (first) {
                if (!lead) { std::unique_lock<std::mutex> lk(W.mu); if (!W.claimed) { W.claimed = true; lead = true; } else W.cv.wait(lk, [&] { return W.measured; }); }
                one.lock();
                // Does the device still have room for this context's arenas?  (~55 GB a context for 16 M bases of 1 kbp reads, ~75 GB for 10 kbp reads; squeezed into
                // what is left, a first batch of 10 kbp reads at -ctx 3 took 1.8 s, cut into ranges, with every other context waiting behind it.)
                bool roomy = false;                                          // twice a context's arenas free: room to take them in one go AND for the others to grow theirs later
                if (!lead && W.footprint > 0 && rc0 == 0) {
                    uint64_t fb = 0, tb = 0, mine = 0;
                    const bool known = ygpu_memory(ctx[d], &fb, &tb, &mine) == 0;
                    roomy = known && (double)fb >= 2.0 * (double)W.footprint;
                    if (known && (double)fb < 0.9 * (double)W.footprint) {
                        if (timing || stats) fprintf(stderr, "[yaha] context %d left out: %.1f GB free on device %d, the first context's arenas hold %.1f GB\n", d, fb / 1e9, dev,
                            W.footprint / 1e9);
                        (void)ygpu_park(ctx[d]); parked++; break;
                    }
                }
                // The other contexts of the device get the leader's arenas in ONE go (ygpu_presize: its capacities after its first batch, its estimates) while the
                // device's running contexts are held back for those few milliseconds; their own first batch then runs like any later batch, beside the others'.
                // (Before: every context grew its hundred buffers during a first batch of its own, first batches ran one at a time and held the others back --
                // 64 + 106 + 158 ms for the first three batches of a run, a fourth context cost more at the start than it gained later.)
                // (Only while the device is roomy: presized to the brim -- four contexts of 60 GB beside a 17 GB image, three of 90 GB on 10 kbp reads -- a later batch
                // that needs a larger arena finds no memory and is cut into ranges: 319 k -> 290 k reads/s steady, 27.6 k -> 12.4 k on 10 kbp reads, measured.  A
                // context that starts in a tighter device grows its arenas during a first batch of its own, one at a time, as before.)
                if (!lead && rc0 == 0 && W.haveProfile && roomy && getenv("YAHA_NO_PRESIZE") == nullptr) {
                    { std::lock_guard<std::mutex> lk(W.mu); W.firstRunning++; }
                    const double p0 = now(); const int prc = ygpu_presize(ctx[d], &W.profile);
                    { std::lock_guard<std::mutex> lk(W.mu); W.firstRunning--; } W.cv.notify_all();
                    if (timing) fprintf(stderr, "[yaha] context %d: arenas presized from the device's first context in %.1f ms (rc %d)\n", d, now() - p0, prc);
                    if (prc == YGPU_ENOMEM) { if (timing || stats) fprintf(stderr, "[yaha] context %d left out: no room for its arenas on device %d\n", d, dev);
                        (void)ygpu_park(ctx[d]); parked++; break; }
                    if (prc == 0) { first = false; one.unlock(); }
                }
            }
            const double w0 = now();
            if (!inQ.pop(b)) break;
            if (stop) { b->nReads = 0; fmtQ.push(std::move(b)); continue; }
            const double t0 = now(); if (!first) usIdle += (uint64_t)((t0 - w0) * 1e3);
            ygpu_read_batch rb{(uint32_t)b->nReads, b->codes.data(), b->offsets.data()}; ygpu_result_batch res; memset(&res, 0, sizeof res); bool handedOver = false;
            auto hotPath = [&]() -> int {                                      // upload, run (+ post-filter), results straight into the batch's own buffers
                const double h0 = now();
                // (the batch lives until it is printed: no wait for its bytes here)
                int rc = ygpu_upload_nowait(ctx[d], &rb); const double h1 = now(); if (rc == 0) rc = ygpu_run(ctx[d]); if (rc != 0) return rc;
                const double h2 = now();
                uint64_t nc = 0, no = 0; b->filtered = deviceFilter;
                if (!first) { usUpload += (uint64_t)((h1 - h0) * 1e3); usRun += (uint64_t)((h2 - h1) * 1e3); nLater++; }
                if (deviceFilter && overlapFilter && !first) {                 // the filter thread takes it from here; this thread goes on with the next batch
                    filterIdle();
                    const double h3 = now();
                    if (stop) return 0;
                    rc = ygpu_postfilter_snapshot(ctx[d]); if (rc != 0) return rc;
                    usWaitFilter += (uint64_t)((h3 - h2) * 1e3); usSnapshot += (uint64_t)((now() - h3) * 1e3);
                    { std::lock_guard<std::mutex> lk(F.mu); F.b = std::move(b); F.t0 = t0; F.busy = true; if (!F.th.joinable()) F.th = std::thread(filterLoop); }
                    F.cv.notify_all(); handedOver = true; return 0;
                }
                if (deviceFilter) {
                    rc = collectFiltered(b, res);
                    if (timing && first) fprintf(stderr, "[yaha] context %d, first batch: upload %.1f  run %.1f  post-filter and collect %.1f ms\n", d, h1 - h0, h2 - h1,
                        now() - h2);
                    return rc;
                }
                rc = ygpu_result_size(ctx[d], &nc, &no); if (rc != 0) return rc;
                if (!b->clumpStart.ensure(4 * (b->nReads + 1)) || !b->clumps.ensure(sizeof(ygpu_clump) * nc) || !b->ops.ensure(4 * no)) return YGPU_ENOMEM;
                return ygpu_collect_into(ctx[d], (uint32_t *)b->clumpStart.p, (ygpu_clump *)b->clumps.p, (uint32_t *)b->ops.p, &res);
            };
            int rc;
            if (first) {
                { std::lock_guard<std::mutex> lk(W.mu); W.firstRunning++; }
                rc = hotPath(); first = false;
                if (lead) {                                                  // the measure: what this context holds after its first batch (without the index image)
                    uint64_t fb = 0, tb = 0, mine = 0; const uint64_t image = (uint64_t)V.n_base_bytes + 4ull * V.totalMatches + 4ull * ((1ull << (2 * V.wordLen)) + 1);
                    if (rc == 0 && ygpu_memory(ctx[d], &fb, &tb, &mine) == 0) W.footprint = std::max<uint64_t>(1, d == leadCtx && mine > image ? mine - image : mine);
                    if (rc == 0 && ygpu_get_arena_profile(ctx[d], &W.profile) == 0) W.haveProfile = true;
                }
                { std::lock_guard<std::mutex> lk(W.mu); W.firstRunning--; } W.cv.notify_all();
                one.unlock();
                if (lead) measured();
            } else {
                { std::unique_lock<std::mutex> lk(W.mu); W.cv.wait(lk, [&] { return W.firstRunning == 0; }); }
                rc = hotPath();
            }
            if (handedOver) continue;
            deliver(b, rc, res, t0);
        }
        if (lead) measured();                                                 // (a leader that never saw a batch: the others must not wait for its measure)
        if (F.th.joinable()) { filterIdle(); { std::lock_guard<std::mutex> lk(F.mu); F.quit = true; } F.cv.notify_all(); F.th.join(); }
        fmtQ.producerDone();
    };
    auto formatter = [&]() {
        yaha_session local; local.args = A; local.genome.bases = S->genome.bases; local.genome.nBaseBytes = S->genome.nBaseBytes; local.genome.seqs = S->genome.seqs;
            local.genome.maxROff = S->genome.maxROff;
        BatchP b;
        while (fmtQ.pop(b)) {
            const double t0 = now(); b->text.clear();
            if (!stop && b->nReads && b->filtered) {
                ygpu_filtered_batch fr; memset(&fr, 0, sizeof fr);
                fr.n_reads = (uint32_t)b->nReads; fr.clump_start = (const uint32_t *)b->clumpStart.p; fr.clumps = (const ygpu_out_clump *)b->clumps.p;
                    fr.ops = (const uint32_t *)b->ops.p; fr.n_clumps = b->nClumps; fr.n_ops = b->nOps;
                local.reads.swap(b->reads); formatFiltered(&local, &fr, b->text); local.reads.swap(b->reads);
            } else if (!stop && b->nReads) {
                ygpu_result_batch res; memset(&res, 0, sizeof res);
                res.n_reads = (uint32_t)b->nReads; res.clump_start = (const uint32_t *)b->clumpStart.p; res.clumps = (const ygpu_clump *)b->clumps.p;
                    res.ops = (const uint32_t *)b->ops.p; res.n_clumps = b->nClumps; res.n_ops = b->nOps;
                local.reads.swap(b->reads); formatBatch(&local, &res, b->text, 1); local.reads.swap(b->reads);
            }
            b->tFmt = now() - t0;
            outQ.push(std::move(b));
        }
        outQ.producerDone();
    };
    uint64_t nWritten = 0, nFirst = 0; double tFirstOut = 0, tLastOut = 0;
    auto writer = [&]() {
        std::map<uint64_t, BatchP> done; uint64_t nextOut = 0; BatchP b;
        while (outQ.pop(b)) {
            done[b->ticket] = std::move(b);
            while (!done.empty() && done.begin()->first == nextOut) {
                BatchP w = std::move(done.begin()->second); done.erase(done.begin()); nextOut++; ticketsWritten = nextOut;
                if (!stop) {
                    if (w->text.len && fwrite(w->text.p, 1, w->text.len, out) != w->text.len) fail("Failure writing the output file");
                    else { const double t = now(); if (nWritten == 0) { tFirstOut = t; nFirst = w->nReads; } tLastOut = t; nWritten += w->nReads; }
                    if (timing) fprintf(stderr, "[yaha] ticket %llu: %zu reads  parse %.1f  device (upload, run, collect) %.1f  format %.1f ms  written at %.1f\n",
                        (unsigned long long)w->ticket, w->nReads, w->tRead, w->tDev, w->tFmt, now() - tEnter);
                }
                pool.put(std::move(w));
            }
        }
    };
    if (timing) fprintf(stderr, "[yaha] %d usable CPUs: %d parser, %d formatter threads, %d contexts\n", cpus, nParse, nFmt, ngpu);
    std::vector<std::thread> th;
    for (int d = 0; d < ngpu; d++) th.emplace_back(device, d);                 // the index image starts towards the devices first
    th.emplace_back(splitter);
    for (int i = 0; i < nParse; i++) th.emplace_back(parser);
    for (int i = 0; i < nFmt; i++) th.emplace_back(formatter);
    th.emplace_back(writer);
    for (auto &x : th) x.join();
    const double tDone = now();
    // every batch the splitter cut must have reached the writer, in order: a batch lost between two stages would otherwise be a shorter SAM with exit code 0
    if (!stop && ticketsWritten.load() != ticketsIssued.load()) {
        fprintf(log, "internal error: %llu of %llu batches were written -- the output is incomplete.\n", (unsigned long long)ticketsWritten.load(),
        (unsigned long long)ticketsIssued.load()); rcAll = 1; }
    // The command line (csrc/main.cpp) leaves right after this function: it sets YAHA_FAST_EXIT and lets the process exit release the device memory and the
    // page-locked buffers in one go, instead of a hipFree per buffer (a second of waiting at the end of every run, measured).  Library users get the orderly path.
    const bool fastExit = getenv("YAHA_FAST_EXIT") != nullptr;
    if (!fastExit) for (int d = ngpu - 1; d >= 0; d--) if (ctx[d]) ygpu_destroy(ctx[d]);     // clones before their parents
    // (the batches -- a million small strings, the page-locked buffers -- go with the process as well: freeing them one by one was 0.3 s)
    if (fastExit) (void)new std::vector<BatchP>(std::move(pool.free));
    if (fflush(out) != 0 || ferror(out)) { if (!stop) fprintf(log, "Failure writing the output file.\n"); rcAll = 1; }
    if (out != stdout && fclose(out) != 0) { fprintf(log, "Failure closing the output file.\n"); rcAll = 1; }
    if (timing) fprintf(stderr, "[yaha] batches done %.1f ms after start, teardown %.1f ms\n", tDone - tEnter, now() - tDone);
    if (stats) {    // one line for scripts (bench.py): steady = reads written after the first batch / time from the first batch's write to the last one's
        const double steady = (nWritten > nFirst && tLastOut > tFirstOut) ? (nWritten - nFirst) / ((tLastOut - tFirstOut) * 1e-3) : 0.0;
        std::string per = "[";
        for (int k = 0; k < nDev; k++) { char t[32]; snprintf(t, sizeof t, "%s%llu", k ? ", " : "", (unsigned long long)devReads[k].load()); per += t; }
        per += "]";
        fprintf(stderr, "[yaha] stats {\"reads\": %llu, \"contexts_up_ms\": %.1f, \"first_batch_written_ms\": %.1f, \"last_batch_written_ms\": %.1f, \"total_ms\": %.1f, "
            "\"steady_reads_per_s\": %.0f, \"cpus\": %d, \"formatters\": %d, \"parsers\": %d, \"gpus\": %d, \"ctx_per_gpu\": %d, \"ctx_left_out\": %d, "
            "\"reads_per_device\": %s, \"context_thread_ms_per_batch\": {\"wait_for_a_batch\": %.2f, \"upload\": %.2f, \"run\": %.2f, \"wait_for_filter_thread\": %.2f, "
            "\"snapshot\": %.2f}, \"filter_thread_ms_per_batch\": %.2f}\n",
                (unsigned long long)nWritten, tCtxUp - tEnter, tFirstOut - tEnter, tLastOut - tEnter, now() - tEnter, steady, cpus, nFmt, nParse, nDev, perDev, parked.load(),
                    per.c_str(),
                usIdle / 1e3 / std::max<uint64_t>(1, nLater), usUpload / 1e3 / std::max<uint64_t>(1, nLater), usRun / 1e3 / std::max<uint64_t>(1, nLater),
                    usWaitFilter / 1e3 / std::max<uint64_t>(1, nLater), usSnapshot / 1e3 / std::max<uint64_t>(1, nLater), usFilter / 1e3 / std::max<uint64_t>(1, nLater));
    }
    return rcAll;
}
}  // namespace yaha

// ---- C-ABI --------------------------------------------------------------------------------------------------
extern "C" {
int yaha_session_open(int argc, const char *const *argv, yaha_session **out)
{
    *out = nullptr; yaha_session *s = new yaha_session;
    std::vector<char *> av; av.push_back((char *)"yaha"); for (int i = 0; i < argc; i++) av.push_back((char *)argv[i]);
    int rc = parseArgs((int)av.size(), av.data(), s->args);
    if (rc != 0 || !s->args.query) { delete s; return YGPU_EINVAL; }
    *out = s;
    if (!sessionLoad(s)) return YGPU_EINVAL;
    return 0;
}
void yaha_session_close(yaha_session *s) { if (!s) return; if (s->readerOpen) s->reader.close(); delete s; }
const char *yaha_session_error(const yaha_session *s) { return s ? s->err.c_str() : "null session"; }
int yaha_session_params(const yaha_session *s, ygpu_params *p) { paramsFromArgs(s->args, *p); return 0; }
int yaha_session_index_view(const yaha_session *s, ygpu_index_view *v)
{
    v->bases = s->genome.bases; v->n_base_bytes = s->genome.nBaseBytes; v->maxROff = s->genome.maxROff;
    v->startingOffs = s->index.SO; v->ROA = s->index.ROA; v->totalMatches = s->index.totalMatches; v->wordLen = s->index.wordLen; return 0;
}
int yaha_session_header(yaha_session *s, const char **text, size_t *len) { *text = s->header.c_str(); *len = s->header.size(); return 0; }
int yaha_session_next_batch(yaha_session *s, uint32_t max_reads, ygpu_read_batch *b)
{
    s->reads.clear(); s->codes.clear(); s->offsets.assign(1, 0);
    Read r;
    while (s->reads.size() < max_reads && s->reader.next(r)) { s->codes.insert(s->codes.end(), r.fwdCodes.begin(), r.fwdCodes.end()); s->offsets.push_back(s->codes.size());
        s->reads.push_back(std::move(r)); r = Read(); }
    b->n_reads = (uint32_t)s->reads.size(); b->codes = s->codes.data(); b->offsets = s->offsets.data(); return 0;
}
int yaha_session_emit(yaha_session *s, const ygpu_result_batch *r, const char **text, size_t *len)
{
    if (r->n_reads != s->reads.size()) { s->err = "result batch does not match the current read batch"; return YGPU_EINVAL; }
    formatBatch(s, r, s->text, std::max(1, s->args.numThreads)); *s->text.room(1) = 0; *text = s->text.p; *len = s->text.len; return 0;
}
int yaha_session_postfilter_params(yaha_session *s, ygpu_postfilter_params *p)
{
    yoqc::Params P; oqcParamsFromArgs(s->args, P, s->pfThr);
    if (!s->args.OQC || P.bppN < 0 || P.minNonOverlap < 1) { s->err = "the device post-filter takes OQC runs with non-negative break point costs and -MNO of at least 1 only";
        return YGPU_EINVAL; }
    s->pfSeqStart.clear(); s->pfSeqLen.clear(); for (auto &sq : s->genome.seqs) { s->pfSeqStart.push_back(sq.start); s->pfSeqLen.push_back(sq.length); }
    memset(p, 0, sizeof *p);
    p->minNonOverlap = P.minNonOverlap; p->BPCost = P.BPCost; p->maxBPLog = P.maxBPLog; p->FBS = P.FBS; p->FBS_PSLength = P.FBS_PSLength; p->FBS_PSScore = P.FBS_PSScore;
    p->bppVmin = P.bppVmin; p->bppN = P.bppN; p->bppThr = s->pfThr.data(); p->n_seqs = (uint32_t)s->pfSeqStart.size(); p->seq_start = s->pfSeqStart.data();
        p->seq_length = s->pfSeqLen.data();
    return 0;
}
int yaha_session_emit_filtered(yaha_session *s, const ygpu_filtered_batch *r, const char **text, size_t *len)
{
    if (r->n_reads != s->reads.size()) { s->err = "result batch does not match the current read batch"; return YGPU_EINVAL; }
    formatFiltered(s, r, s->text); *s->text.room(1) = 0; *text = s->text.p; *len = s->text.len; return 0;
}
int yaha_build_index(int argc, const char *const *argv)
{
    Args a; std::vector<char *> av; av.push_back((char *)"yaha"); for (int i = 0; i < argc; i++) av.push_back((char *)argv[i]);
    int rc = parseArgs((int)av.size(), av.data(), a); if (rc != 0 || !a.index) return YGPU_EINVAL;
    return runIndex(a, stderr) == 0 ? 0 : YGPU_EINVAL;
}
int yaha_main(int argc, char **argv)
{
    fprintf(stderr, "YAHA (MI355X hot path) compatible with version 0.1.83\n");
    Args a; int rc = parseArgs(argc, argv, a);
    if (rc != 0) return rc - 1;
    if (a.query) return runQueries(a, stderr);
    if (a.compress) return runCompress(a, stderr);
    if (a.uncompress) return runUncompress(a, stderr);
    return runIndex(a, stderr);
}
}
