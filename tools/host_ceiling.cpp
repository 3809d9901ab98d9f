// host_ceiling.cpp -- how fast can the host stages around the device run?  (the input side must not cap a node of 8 GPUs.)
// A diagnostic that drives the PRODUCT's reader code (yaha::ReadSplitter + yaha::parseSpan from libyaha_hip.so) over a read file:
//   splitter alone (one thread: record boundaries), then splitter + N parser threads (id, sequence, codes, reverse complement, packing)
// exactly as host/pipeline.cpp runs them, without a device behind.  Formatting (OQC + SAM) needs device results and is measured by
// tools/host_ceiling.py on the GPU box.
//   g++ -O2 -std=c++17 -o tools/host_ceiling tools/host_ceiling.cpp -Lyaha_amd/csrc -lyaha_hip -Wl,-rpath,$PWD/yaha_amd/csrc -pthread
#include "../yaha_amd/csrc/host/yaha_host.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
using namespace yaha;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: host_ceiling reads.fa [batch=4096] [parserThreads...]\n"); return 2; }
    const char *path = argv[1]; const size_t batch = argc > 2 ? (size_t)atol(argv[2]) : 4096;
    std::string err;
    {   // page the file in once so that every leg sees the same (cached) input
        ReadSplitter sp; if (!sp.open(path, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        std::vector<Span> v; size_t n = 0; while (sp.nextSpans(batch, v)) n += v.size();
    }
    {
        ReadSplitter sp; sp.open(path, err); std::vector<Span> v; size_t n = 0; const double t0 = now();
        while (sp.nextSpans(batch, v)) n += v.size();
        const double dt = now() - t0;
        printf("{\"stage\": \"split\", \"threads\": 1, \"reads\": %zu, \"seconds\": %.4f, \"reads_per_s\": %.0f}\n", n, dt, n / dt);
    }
    for (int a = 3; a < argc || a == 3; a++) {
        const int P = a < argc ? atoi(argv[a]) : 4;
        ReadSplitter sp; sp.open(path, err);
        std::mutex mu; std::condition_variable cvPush, cvPop; std::deque<std::vector<Span>> q; bool eof = false; const size_t cap = (size_t)P + 2;
        std::atomic<size_t> reads(0), bases(0);
        const double t0 = now();
        std::thread splitter([&]() {
            for (;;) { std::vector<Span> v; if (!sp.nextSpans(batch, v)) break; std::unique_lock<std::mutex> lk(mu); cvPush.wait(lk, [&] { return q.size() < cap; }); q.push_back(std::move(v)); cvPop.notify_one(); }
            std::lock_guard<std::mutex> lk(mu); eof = true; cvPop.notify_all();
        });
        std::vector<std::thread> th;
        for (int t = 0; t < P; t++) th.emplace_back([&]() {
            for (;;) {
                std::vector<Span> v;
                { std::unique_lock<std::mutex> lk(mu); cvPop.wait(lk, [&] { return !q.empty() || eof; }); if (q.empty()) return; v = std::move(q.front()); q.pop_front(); cvPush.notify_one(); }
                std::vector<Read> rs; rs.reserve(v.size()); Read r; size_t nb = 0;
                for (auto &s : v) if (parseSpan(s, sp.fastq, 32000, 15, r)) { nb += r.fwdCodes.size(); rs.push_back(std::move(r)); }
                std::vector<uint8_t> codes(nb); size_t o = 0; for (auto &x : rs) { memcpy(codes.data() + o, x.fwdCodes.data(), x.fwdCodes.size()); o += x.fwdCodes.size(); }
                reads += rs.size(); bases += nb;
            }
        });
        splitter.join(); for (auto &x : th) x.join();
        const double dt = now() - t0;
        printf("{\"stage\": \"split+parse\", \"threads\": %d, \"reads\": %zu, \"bases\": %zu, \"seconds\": %.4f, \"reads_per_s\": %.0f}\n", P, reads.load(), bases.load(), dt, reads / dt);
        if (a >= argc) break;
    }
    return 0;
}
