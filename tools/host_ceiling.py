"""Host ceiling on the GPU box: how many reads/s can the stages around the device sustain?
  parse : tools/host_ceiling (C++: the product's ReadSplitter + parseSpan) on 262 144 x 1 kbp reads, 1..32 parser threads
  format: real device results of one 65 536-read batch (ygpu_run on the bench genome), then the product's OQC/FBS filter + SAM text (yaha_session_emit)
          over them with 1..128 threads -- the recorded results are replayed, nothing of the product path is stubbed.
Run after bench.py has built the cache:  python tools/host_ceiling.py > gpurun_out/host_ceiling.jsonl"""
import json, os, subprocess, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import yaha_amd as ya
cache = os.environ.get("YAHA_BENCH_CACHE", "/tmp/yaha_bench_cache")
tag = "g3100m_s42" if os.path.exists(os.path.join(cache, "g3100m_s42.X15_01_65525S.done")) else "g100m_s42"
X = os.path.join(cache, tag + ".X15_01_65525S"); G = os.path.join(cache, tag + ".fa")
R = os.path.join(cache, "ceiling_%s_262144.fa" % tag)
if not os.path.exists(R):
    subprocess.check_call([os.path.join(root, "tools/yaha_sim"), "reads", "--genome", G, "--out", R, "--seed", "99", "--n", "262144", "--len", "1000", "--div", "0.017"])
exe = os.path.join(root, "tools", "host_ceiling")
if True:                                               # always rebuilt: it links the product library, whose headers move
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(root, "tools", "host_ceiling.cpp"), "-L" + os.path.join(root, "yaha_amd", "csrc"), "-lyaha_hip",
                           "-Wl,-rpath," + os.path.join(root, "yaha_amd", "csrc"), "-pthread"])
sys.stdout.write(subprocess.run([exe, R, "4096", "1", "2", "4", "8", "16", "32"], stdout=subprocess.PIPE, check=True).stdout.decode()); sys.stdout.flush()
# Formatting as the command line does it (host/pipeline.cpp): a pool of formatter threads, each takes WHOLE batches (OQC/FBS filter + SAM text of every
# read of the batch) -- so T sessions hold the same 8 192 reads, the device aligns them once, and T threads format those recorded results concurrently.
import threading
N = 8192
with ya.Session(["-x", X, "-q", R]) as s0:
    b0 = s0.next_batch(N)
    ctx = ya.Context(s0.index, s0.params)
    ctx.upload(b0); ctx.run(); r = ctx.collect()
    import bench
    print(json.dumps({"box": {"hardware_threads": os.cpu_count(), "usable_cpus": bench.usable_cpus(), "cpu.max": (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)}, "index": tag})); sys.stdout.flush()
    for T in (1, 2, 4, 8, 12, 16, 24, 32, 64):
        sessions = [ya.Session(["-x", X, "-q", R, "-t", "1"]) for _ in range(T)]
        for s in sessions:
            assert s.next_batch(N).n_reads == N
        reps = 4
        import ctypes as C
        def work(s):                                   # the C call only (ctypes drops the GIL for it); the text stays in the session's buffer
            t_, n_ = C.c_char_p(), C.c_size_t()
            for _ in range(reps):
                assert ya.lib().yaha_session_emit(s._h, C.byref(r), C.byref(t_), C.byref(n_)) == 0
        th = [threading.Thread(target=work, args=(s,)) for s in sessions]
        t = time.time()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.time() - t
        print(json.dumps({"stage": "format (OQC + SAM text), one whole batch per thread", "threads": T, "reads": N * T * reps, "clumps_in_per_batch": int(r.n_clumps), "seconds": dt, "reads_per_s": N * T * reps / dt})); sys.stdout.flush()
        for s in sessions: s.close()
    # the same with the post-filter done on the device (ygpu_postfilter, what the command line does): the host's share of a read is printing what is left
    ctx.set_postfilter(s0); f = ctx.postfilter()
    for T in (1, 2, 4, 8, 12, 16):
        sessions = [ya.Session(["-x", X, "-q", R, "-t", "1"]) for _ in range(T)]
        for s in sessions:
            assert s.next_batch(N).n_reads == N
        reps = 20
        def work2(s):
            t_, n_ = C.c_char_p(), C.c_size_t()
            for _ in range(reps):
                assert ya.lib().yaha_session_emit_filtered(s._h, C.byref(f), C.byref(t_), C.byref(n_)) == 0
        th = [threading.Thread(target=work2, args=(s,)) for s in sessions]
        t = time.time()
        for x in th: x.start()
        for x in th: x.join()
        dt = time.time() - t
        print(json.dumps({"stage": "format (SAM text of the device-filtered clumps), one whole batch per thread", "threads": T, "reads": N * T * reps, "clumps_in_per_batch": int(f.n_clumps), "seconds": dt, "reads_per_s": N * T * reps / dt})); sys.stdout.flush()
        for s in sessions: s.close()
    ctx.close()
