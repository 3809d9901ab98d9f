// h2d_probe.hip -- how fast can an index image in the page cache reach HBM on this box, and through which path?
// The command line's start-up is the 16.7 GB index image going from a memory-mapped file to the device (23-30 GB/s through the runtime's
// pageable path).  This program measures the alternatives on a file of its own:
//   pinned      hipHostMalloc'ed memory -> device: the link's ceiling
//   pageable    the memory-mapped file -> device with one plain hipMemcpy (what ygpu_init did)
//   register    hipHostRegister on the mapping (in slices, T threads), then asynchronous copies of the registered slices
//   pread       T threads pread() the file into page-locked double buffers of their own and copy those
//   d2d         device -> device on one GPU (the rate a peer copy cannot exceed here; the box has one GPU)
// Run: ./h2d_probe [GB (default 4)] [path of the scratch file (default /tmp/h2d_probe.bin)]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <algorithm>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { using namespace std::chrono; return duration<double>(steady_clock::now().time_since_epoch()).count(); }

#include <sched.h>
static void pinTo(int cpu) { cpu_set_t m; CPU_ZERO(&m); CPU_SET(cpu, &m); if (sched_setaffinity(0, sizeof m, &m) != 0) perror("sched_setaffinity"); }
// ./h2d_probe numa GB cpuA cpuB: does it matter on which NUMA node the page cache holds the file?  The file is written by a thread pinned to cpuA, then to cpuB (new
// pages each time), and copied to the device by a plain hipMemcpy of its mapping and by pread into page-locked buffers from threads pinned next to cpuA / cpuB.
static int numaMode(double gb, int cpuA, int cpuB)
{
    const size_t bytes = ((size_t)(gb * (1ull << 30))) & ~((size_t)(64 << 20) - 1);
    char *dev; CK(hipMalloc(&dev, bytes));
    { char *pin; CK(hipHostMalloc((void **)&pin, 1 << 30, hipHostMallocDefault)); memset(pin, 1, 1 << 30); CK(hipMemcpy(dev, pin, 1 << 30, hipMemcpyHostToDevice)); double t = now(); for (int r = 0; r < 4; r++) CK(hipMemcpy(dev, pin, 1 << 30, hipMemcpyHostToDevice)); printf("pinned (hipHostMalloc) -> device: %.1f GB/s\n", 4.0 * (1 << 30) / (now() - t) / 1e9); CK(hipHostFree(pin)); }
    for (int which = 0; which < 2; which++) {
        const int wcpu = which ? cpuB : cpuA; char path[64]; snprintf(path, sizeof path, "/tmp/h2d_numa_%d.bin", which);
        std::thread w([&]() { pinTo(wcpu); int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644); std::vector<char> blk(64 << 20); for (size_t i = 0; i < blk.size(); i++) blk[i] = (char)(i * 2654435761u >> 13);
                              for (size_t o = 0; o < bytes; o += blk.size()) if (write(fd, blk.data(), blk.size()) != (ssize_t)blk.size()) { perror("write"); exit(1); } close(fd); });
        w.join();
        int fd = open(path, O_RDONLY); char *map = (char *)mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; }
        for (int rep = 0; rep < 2; rep++) { double t = now(); CK(hipMemcpy(dev, map, bytes, hipMemcpyHostToDevice)); printf("file written on cpu %3d: plain hipMemcpy of the mapping            %7.1f ms  %5.1f GB/s\n", wcpu, (now() - t) * 1e3, bytes / (now() - t) / 1e9); }
        for (int rcpu : {cpuA, cpuB}) for (int T : {4, 8}) {
            const size_t piece = 8u << 20; std::atomic<size_t> next(0); std::vector<char *> bufs(2 * T);
            double t = now();
            auto work = [&](int id) {
                pinTo(rcpu + id); CK(hipSetDevice(0)); CK(hipHostMalloc((void **)&bufs[2 * id], piece, hipHostMallocDefault)); CK(hipHostMalloc((void **)&bufs[2 * id + 1], piece, hipHostMallocDefault));
                hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming)); bool used[2] = {false, false};
                for (int k = 0;; k ^= 1) { const size_t o = next.fetch_add(piece); if (o >= bytes) break; const size_t n = std::min(piece, bytes - o);
                    if (used[k]) CK(hipEventSynchronize(ev[k]));
                    char *b = bufs[2 * id + k]; size_t got = 0; while (got < n) { ssize_t r = pread(fd, b + got, n - got, o + got); if (r <= 0) { perror("pread"); exit(1); } got += r; }
                    CK(hipMemcpyAsync(dev + o, b, n, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[k], st)); used[k] = true; }
                CK(hipStreamSynchronize(st)); (void)hipStreamDestroy(st); CK(hipHostFree(bufs[2 * id])); CK(hipHostFree(bufs[2 * id + 1])); };
            std::vector<std::thread> th; for (int k = 1; k < T; k++) th.emplace_back(work, k); work(0); for (auto &x : th) x.join();
            printf("file written on cpu %3d: pread by %d threads on cpus %3d.., pinned 8 MB  %7.1f ms  %5.1f GB/s (incl. the buffers' hipHostMalloc)\n", wcpu, T, rcpu, (now() - t) * 1e3, bytes / (now() - t) / 1e9);
        }
        munmap(map, bytes); close(fd); unlink(path);
    }
    return 0;
}

// ./h2d_probe file PATH: an existing file (the bench's 16.7 GB index), mapped privately, copied whole by plain hipMemcpy calls -- what ygpu_init does, nothing else running
static int fileMode(const char *path)
{
    int fd = open(path, O_RDONLY); if (fd < 0) { perror("open"); return 1; }
    const size_t bytes = (size_t)lseek(fd, 0, SEEK_END); char *dev; CK(hipMalloc(&dev, bytes));
    for (int rep = 0; rep < 3; rep++) {
        char *map = (char *)mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; }
        double t = now();
        if (rep < 2) CK(hipMemcpy(dev, map, bytes, hipMemcpyHostToDevice));
        else for (size_t o = 0; o < bytes; o += (1ull << 30)) CK(hipMemcpy(dev + o, map + o, std::min<size_t>(1ull << 30, bytes - o), hipMemcpyHostToDevice));
        printf("%s (%.1f GB), %s: %7.1f ms  %5.1f GB/s\n", path, bytes / 1e9, rep < 2 ? "one hipMemcpy" : "1 GB pieces", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
        munmap(map, bytes);
    }
    return 0;
}

// ./h2d_probe write GB PATH [chunkMB]: writes a file of that size with write() calls of chunkMB (0: one call for everything, as the index writer does) and leaves it there
static int writeMode(double gb, const char *path, size_t chunkMB)
{
    const size_t bytes = ((size_t)(gb * (1ull << 30))) & ~((size_t)(64 << 20) - 1);
    std::vector<char> blk(chunkMB ? (chunkMB << 20) : bytes); for (size_t i = 0; i < blk.size(); i += 61) blk[i] = (char)(i * 2654435761u >> 13);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0744); if (fd < 0) { perror("open"); return 1; }
    double t = now();
    for (size_t o = 0; o < bytes;) { const size_t n = std::min(blk.size(), bytes - o); size_t done = 0; while (done < n) { ssize_t w = write(fd, blk.data() + done, n - done); if (w < 0) { perror("write"); return 1; } done += (size_t)w; } o += n; }
    close(fd); printf("wrote %s (%.1f GB) in %.2f s\n", path, bytes / 1e9, now() - t); return 0;
}

int main(int argc, char **argv)
{
    if (argc > 3 && strcmp(argv[1], "write") == 0) return writeMode(atof(argv[2]), argv[3], argc > 4 ? (size_t)atoi(argv[4]) : 0);
    if (argc > 2 && strcmp(argv[1], "file") == 0) return fileMode(argv[2]);
    if (argc > 1 && strcmp(argv[1], "numa") == 0) return numaMode(argc > 2 ? atof(argv[2]) : 8.0, argc > 3 ? atoi(argv[3]) : 0, argc > 4 ? atoi(argv[4]) : 64);
    const double gb = argc > 1 ? atof(argv[1]) : 4.0; const char *path = argc > 2 ? argv[2] : "/tmp/h2d_probe.bin";
    const size_t bytes = ((size_t)(gb * (1ull << 30))) & ~((size_t)(64 << 20) - 1);
    int ndev = 0; CK(hipGetDeviceCount(&ndev)); printf("devices: %d\n", ndev);
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); printf("stream priority range: least %d, greatest %d\n", lo, hi);
    for (int a = 0; a < ndev; a++) for (int b = 0; b < ndev; b++) if (a != b) { int can = 0; hipDeviceCanAccessPeer(&can, a, b); printf("peer %d -> %d: %d\n", a, b, can); }
    char *dev; CK(hipMalloc(&dev, bytes)); char *dev2; CK(hipMalloc(&dev2, bytes));
    // the file, written once and read back so that it sits in the page cache
    { int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644); if (fd < 0) { perror("open"); return 1; }
      std::vector<char> blk(64 << 20); for (size_t i = 0; i < blk.size(); i++) blk[i] = (char)(i * 2654435761u >> 13);
      for (size_t o = 0; o < bytes; o += blk.size()) if (write(fd, blk.data(), blk.size()) != (ssize_t)blk.size()) { perror("write"); return 1; }
      close(fd); }
    int fd = open(path, O_RDONLY); if (fd < 0) { perror("open"); return 1; }
    char *map = (char *)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; }
    { volatile char s = 0; for (size_t o = 0; o < bytes; o += 4096) s += map[o]; }
    auto report = [&](const char *what, double t) { printf("%-58s %7.1f ms  %6.1f GB/s\n", what, t * 1e3, bytes / t / 1e9); fflush(stdout); };
    // pinned
    { char *pin; double t = now(); CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault)); printf("hipHostMalloc of %.1f GB: %.1f ms\n", bytes / 1e9, (now() - t) * 1e3);
      memset(pin, 1, bytes);
      for (int rep = 0; rep < 2; rep++) { t = now(); CK(hipMemcpy(dev, pin, bytes, hipMemcpyHostToDevice)); report("pinned -> device, one copy", now() - t); }
      { hipStream_t s[2]; CK(hipStreamCreate(&s[0])); CK(hipStreamCreate(&s[1])); t = now();
        CK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, s[0])); CK(hipMemcpyAsync(dev + bytes / 2, pin + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s[1]));
        CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1])); report("pinned -> device, two streams", now() - t); }
      CK(hipHostFree(pin)); }
    // pageable
    for (int rep = 0; rep < 2; rep++) { double t = now(); CK(hipMemcpy(dev, map, bytes, hipMemcpyHostToDevice)); report("mapped file -> device, plain hipMemcpy", now() - t); }
    // register the mapping in slices, T threads, copies follow each slice's registration
    for (int T : {1, 4, 8, 16}) for (size_t sliceMB : {64, 256}) {
        const size_t slice = sliceMB << 20; std::atomic<size_t> next(0); std::atomic<int> bad(0); double tReg = 0; std::atomic<long long> regNs(0);
        double t = now();
        auto work = [&]() {
            CK(hipSetDevice(0)); hipStream_t st; CK(hipStreamCreate(&st)); std::vector<char *> mine;
            for (;;) { const size_t o = next.fetch_add(slice); if (o >= bytes || bad) break; const size_t n = std::min(slice, bytes - o);
                const double r0 = now();
                hipError_t e = hipHostRegister(map + o, n, hipHostRegisterReadOnly); if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(map + o, n, hipHostRegisterDefault); }
                regNs += (long long)((now() - r0) * 1e9);
                if (e != hipSuccess) { if (!bad.exchange(1)) fprintf(stderr, "hipHostRegister: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); break; }
                mine.push_back(map + o);
                if (hipMemcpyAsync(dev + o, map + o, n, hipMemcpyHostToDevice, st) != hipSuccess) { bad = 1; break; } }
            hipStreamSynchronize(st); for (char *p : mine) hipHostUnregister(p); hipStreamDestroy(st); };
        std::vector<std::thread> th; for (int k = 1; k < T; k++) th.emplace_back(work); work(); for (auto &x : th) x.join();
        tReg = regNs.load() / 1e9; char nm[128]; snprintf(nm, sizeof nm, "register %zu MB slices + copy, %d threads (reg %.0f ms cpu)%s", sliceMB, T, tReg * 1e3, bad ? " FAILED" : "");
        report(nm, now() - t);     // (includes the unregister calls)
    }
    // pread into page-locked double buffers
    for (int T : {2, 4, 8, 12, 16}) for (size_t pieceMB : {8, 32}) {
        const size_t piece = pieceMB << 20; std::atomic<size_t> next(0);
        std::vector<char *> bufs(2 * T); for (auto &b : bufs) CK(hipHostMalloc((void **)&b, piece, hipHostMallocDefault));
        double t = now();
        auto work = [&](int id) {
            CK(hipSetDevice(0)); hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming)); bool used[2] = {false, false};
            for (int k = 0;; k ^= 1) { const size_t o = next.fetch_add(piece); if (o >= bytes) break; const size_t n = std::min(piece, bytes - o);
                if (used[k]) CK(hipEventSynchronize(ev[k]));
                char *b = bufs[2 * id + k]; size_t got = 0; while (got < n) { ssize_t r = pread(fd, b + got, n - got, o + got); if (r <= 0) { perror("pread"); exit(1); } got += r; }
                CK(hipMemcpyAsync(dev + o, b, n, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[k], st)); used[k] = true; }
            CK(hipStreamSynchronize(st)); hipStreamDestroy(st); };
        std::vector<std::thread> th; for (int k = 1; k < T; k++) th.emplace_back(work, k); work(0); for (auto &x : th) x.join();
        char nm[128]; snprintf(nm, sizeof nm, "pread into pinned %zu MB double buffers, %d threads", pieceMB, T); report(nm, now() - t);
        for (auto &b : bufs) CK(hipHostFree(b));
    }
    // memcpy from the mapping into page-locked double buffers (no system call per piece)
    for (int T : {4, 8, 16}) {
        const size_t piece = 16 << 20; std::atomic<size_t> next(0);
        std::vector<char *> bufs(2 * T); for (auto &b : bufs) CK(hipHostMalloc((void **)&b, piece, hipHostMallocDefault));
        double t = now();
        auto work = [&](int id) {
            CK(hipSetDevice(0)); hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming)); bool used[2] = {false, false};
            for (int k = 0;; k ^= 1) { const size_t o = next.fetch_add(piece); if (o >= bytes) break; const size_t n = std::min(piece, bytes - o);
                if (used[k]) CK(hipEventSynchronize(ev[k]));
                memcpy(bufs[2 * id + k], map + o, n);
                CK(hipMemcpyAsync(dev + o, bufs[2 * id + k], n, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[k], st)); used[k] = true; }
            CK(hipStreamSynchronize(st)); hipStreamDestroy(st); };
        std::vector<std::thread> th; for (int k = 1; k < T; k++) th.emplace_back(work, k); work(0); for (auto &x : th) x.join();
        char nm[128]; snprintf(nm, sizeof nm, "memcpy from the mapping into pinned 16 MB buffers, %d threads", T); report(nm, now() - t);
        for (auto &b : bufs) CK(hipHostFree(b));
    }
    { double t = now(); CK(hipMemcpy(dev2, dev, bytes, hipMemcpyDeviceToDevice)); CK(hipDeviceSynchronize()); report("device -> device (same GPU)", now() - t); }
    // ---- what a fresh process pays: device memory that was never written, and a mapping whose pages were never touched by this process ----
    { char *pin; CK(hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault)); memset(pin, 1, bytes);
      for (int variant = 0; variant < 4; variant++) {
          char *fresh; double t = now(); CK(hipMalloc(&fresh, bytes)); const double tAlloc = now() - t; double tTouch = 0;
          if (variant == 1) { t = now(); CK(hipMemset(fresh, 0, bytes)); CK(hipDeviceSynchronize()); tTouch = now() - t; }
          if (variant == 2) { t = now(); for (size_t o = 0; o < bytes; o += (2u << 20)) CK(hipMemsetAsync(fresh + o, 0, 4, 0)); CK(hipDeviceSynchronize()); tTouch = now() - t; }
          if (variant == 3) { t = now(); CK(hipMemset(fresh, 0, bytes)); CK(hipDeviceSynchronize()); tTouch = now() - t; t = now(); CK(hipMemset(fresh, 0, bytes)); CK(hipDeviceSynchronize()); printf("   second hipMemset of the same buffer: %.1f ms\n", (now() - t) * 1e3); }
          t = now(); CK(hipMemcpy(fresh, pin, bytes, hipMemcpyHostToDevice)); const double tCopy = now() - t;
          t = now(); CK(hipMemcpy(fresh, pin, bytes, hipMemcpyHostToDevice)); const double tCopy2 = now() - t;
          printf("fresh hipMalloc (%.1f ms), %s (%.1f ms): pinned -> device %.1f ms (%.1f GB/s), again %.1f ms\n", tAlloc * 1e3,
                 variant == 0 ? "untouched" : variant == 2 ? "4 bytes set per 2 MB" : "hipMemset of all of it", tTouch * 1e3, tCopy * 1e3, bytes / tCopy / 1e9, tCopy2 * 1e3);
          CK(hipFree(fresh)); }
      CK(hipHostFree(pin)); }
    for (int variant = 0; variant < 5; variant++) {
        // a new mapping: its pages are in the page cache but not in this process's page table
        munmap(map, bytes); map = (char *)mmap(nullptr, bytes, PROT_READ, variant == 1 ? MAP_SHARED | MAP_POPULATE : MAP_SHARED, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; }
        double tPrep = 0, t = now();
        if (variant >= 2) {      // MADV_POPULATE_READ (22) by T threads over slices
            const int T = variant == 2 ? 1 : variant == 3 ? 4 : 16; std::atomic<size_t> next(0); const size_t slice = 64u << 20; std::atomic<int> bad(0);
            auto work = [&]() { for (;;) { const size_t o = next.fetch_add(slice); if (o >= bytes) break; if (madvise(map + o, std::min(slice, bytes - o), 22) != 0) bad = 1; } };
            std::vector<std::thread> th; for (int k = 1; k < T; k++) th.emplace_back(work); work(); for (auto &x : th) x.join();
            tPrep = now() - t; if (bad) printf("   (MADV_POPULATE_READ refused)\n");
        } else if (variant == 1) tPrep = 0;
        t = now(); CK(hipMemcpy(dev, map, bytes, hipMemcpyHostToDevice)); const double tCopy = now() - t;
        printf("new mapping, %s (%.1f ms): plain hipMemcpy %.1f ms (%.1f GB/s)\n", variant == 0 ? "nothing done" : variant == 1 ? "MAP_POPULATE (inside mmap)" : variant == 2 ? "MADV_POPULATE_READ, 1 thread" : variant == 3 ? "MADV_POPULATE_READ, 4 threads" : "MADV_POPULATE_READ, 16 threads",
               tPrep * 1e3, tCopy * 1e3, bytes / tCopy / 1e9);
    }
    for (int variant = 0; variant < 4; variant++) {
        // what ygpu_init did until round 4: a PRIVATE read-only mapping, copied with hipMemcpyAsync + hipStreamSynchronize in 256 MB slices by one or two threads
        munmap(map, bytes); map = (char *)mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0); if (map == MAP_FAILED) { perror("mmap"); return 1; }
        double t = now();
        if (variant == 0) CK(hipMemcpy(dev, map, bytes, hipMemcpyHostToDevice));
        else if (variant == 1) { CK(hipMemcpyAsync(dev, map, bytes, hipMemcpyHostToDevice, 0)); CK(hipStreamSynchronize(0)); }
        else { const int T = variant == 2 ? 1 : 2; std::atomic<size_t> next(0); const size_t slice = 256u << 20;
            auto work = [&]() { CK(hipSetDevice(0)); hipStream_t st; CK(hipStreamCreate(&st));
                for (;;) { const size_t o = next.fetch_add(slice); if (o >= bytes) break; CK(hipMemcpyAsync(dev + o, map + o, std::min(slice, bytes - o), hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); }
                (void)hipStreamDestroy(st); };
            std::vector<std::thread> th; for (int k = 1; k < T; k++) th.emplace_back(work); work(); for (auto &x : th) x.join(); }
        report(variant == 0 ? "PRIVATE mapping, plain hipMemcpy" : variant == 1 ? "PRIVATE mapping, one hipMemcpyAsync + synchronize" : variant == 2 ? "PRIVATE mapping, 256 MB slices async + sync, 1 thread" : "PRIVATE mapping, 256 MB slices async + sync, 2 threads", now() - t);
    }
    {   // pread into pinned buffers, 4 threads, into device memory that was never written
        char *fresh; CK(hipMalloc(&fresh, bytes)); const int T = 4; const size_t piece = 8u << 20; std::atomic<size_t> next(0);
        std::vector<char *> bufs(2 * T); for (auto &b : bufs) CK(hipHostMalloc((void **)&b, piece, hipHostMallocDefault));
        double t = now();
        auto work = [&](int id) {
            CK(hipSetDevice(0)); hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming)); bool used[2] = {false, false};
            for (int k = 0;; k ^= 1) { const size_t o = next.fetch_add(piece); if (o >= bytes) break; const size_t n = std::min(piece, bytes - o);
                if (used[k]) CK(hipEventSynchronize(ev[k]));
                char *b = bufs[2 * id + k]; size_t got = 0; while (got < n) { ssize_t r = pread(fd, b + got, n - got, o + got); if (r <= 0) { perror("pread"); exit(1); } got += r; }
                CK(hipMemcpyAsync(fresh + o, b, n, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[k], st)); used[k] = true; }
            CK(hipStreamSynchronize(st)); (void)hipStreamDestroy(st); };
        std::vector<std::thread> th; for (int k = 1; k < T; k++) th.emplace_back(work, k); work(0); for (auto &x : th) x.join();
        report("pread, 4 threads, 8 MB pieces, into never-written device memory", now() - t);
        for (auto &b : bufs) CK(hipHostFree(b)); CK(hipFree(fresh)); }
    munmap(map, bytes); close(fd); unlink(path);
    return 0;
}
