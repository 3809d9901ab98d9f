// cpu_probe.cpp -- what the host of the GPU box can really run in parallel (a cgroup CPU quota, SMT and NUMA all show up here):
// T threads each do a fixed amount of (a) pure ALU work, (b) private 16 MB buffer streaming writes, (c) malloc/free of 16 MB blocks with first touch;
// prints the aggregate rate per T.   g++ -O2 -std=c++17 -pthread -o tools/cpu_probe tools/cpu_probe.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    std::vector<int> Ts; for (int a = 1; a < argc; a++) Ts.push_back(atoi(argv[a])); if (Ts.empty()) Ts = {1, 8, 16, 32, 64, 128, 256};
    printf("{\"hardware_concurrency\": %u}\n", std::thread::hardware_concurrency());
    for (int mode = 0; mode < 3; mode++) for (int T : Ts) {
        std::atomic<uint64_t> sink(0); const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
            if (mode == 0) { uint64_t x = t + 1; for (long i = 0; i < 200000000L; i++) x = x * 6364136223846793005ULL + 1442695040888963407ULL; sink += x; }
            else if (mode == 1) { std::vector<char> b(16 << 20); for (int r = 0; r < 40; r++) memset(b.data(), r, b.size()); sink += b[5]; }
            else { for (int r = 0; r < 40; r++) { char *p = (char *)malloc(16 << 20); memset(p, r, 16 << 20); sink += p[7]; free(p); } }
        });
        for (auto &x : th) x.join();
        const double dt = now() - t0;
        printf("{\"mode\": \"%s\", \"threads\": %d, \"seconds\": %.3f, \"thread_units_per_s\": %.2f}\n", mode == 0 ? "alu" : mode == 1 ? "memset_private_16MB_x40" : "malloc_touch_free_16MB_x40", T, dt, T / dt);
        fflush(stdout);
    }
    return 0;
}
