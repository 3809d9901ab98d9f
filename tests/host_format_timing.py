"""Helper script (not a test): time the product's host post-filter + SAM writer (yaha_session_emit) on recorded hot-path results.
    python tests/host_format_timing.py make  IDX READS N DUMP.npz     # run the oracle once (CPU), record its result arrays
    python tests/host_format_timing.py time  IDX READS N DUMP.npz [reps] [threads]
Lives under tests/ because the `make` leg uses the oracle as the producer of results on a box without a GPU."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yaha_amd as ya


def main():
    mode, idx, reads, n, dump = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    if mode == "make":
        import oracle
        with ya.Session(["-x", idx, "-q", reads]) as s:
            b = s.next_batch(n)
            t = time.time(); r, own = oracle.run(s.index, s.params, b, threads=8); print("oracle %.1fs" % (time.time() - t))
            cs = np.ctypeslib.as_array(r.clump_start, (r.n_reads + 1,)).copy()
            cl = np.frombuffer(C.string_at(r.clumps, r.n_clumps * 32), dtype=np.uint8).copy()
            ops = np.ctypeslib.as_array(r.ops, (r.n_ops,)).copy()
            np.savez(dump, cs=cs, cl=cl, ops=ops)
            print("reads %d clumps %d ops %d" % (r.n_reads, r.n_clumps, r.n_ops))
        return
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    T = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    d = np.load(dump)
    cs, cl, ops = d["cs"], d["cl"], d["ops"]
    r = ya.ResultBatch()
    r.n_reads = len(cs) - 1
    r.clump_start = cs.ctypes.data_as(C.POINTER(C.c_uint32)); r.clumps = C.cast(cl.ctypes.data, C.POINTER(ya.Clump)); r.ops = ops.ctypes.data_as(C.POINTER(C.c_uint32))
    r.n_clumps = len(cl) // 32; r.n_ops = len(ops)
    sessions = [ya.Session(["-x", idx, "-q", reads, "-t", "1"]) for _ in range(T)]
    for s in sessions:
        assert s.next_batch(n).n_reads == r.n_reads
    out = [None] * T

    def work(k):
        t_, n_ = C.c_char_p(), C.c_size_t()
        for _ in range(reps):
            assert ya.lib().yaha_session_emit(sessions[k]._h, C.byref(r), C.byref(t_), C.byref(n_)) == 0
        out[k] = C.string_at(t_, n_.value)
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    t = time.time()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.time() - t
    import hashlib
    print("threads %d reps %d: %.2f us/read/thread, %.0f reads/s, %d clumps/read in, %d text bytes, sha %s" % (T, reps, 1e6 * dt / (reps * r.n_reads), T * reps * r.n_reads / dt, r.n_clumps // r.n_reads, len(out[0]), hashlib.sha256(out[0]).hexdigest()[:16]))


main()
