"""oracle -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so (the CPU restatement of the hot
path, oracle/hotpath.cpp) and the path of the real reference binary oracle/_ref/yaha.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing in yaha_amd/ does."""
import ctypes as C
import os
import subprocess

import yaha_amd as ya

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "yaha")


class OracleResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("clump_start", C.POINTER(C.c_uint32)), ("clumps", C.POINTER(ya.Clump)),
                ("ops", C.POINTER(C.c_uint32)), ("n_clumps", C.c_uint64), ("n_ops", C.c_uint64), ("counters", ya.Counters)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE, "oracle"])
        _lib = C.CDLL(LIB_PATH)
    return _lib


def run(index_view, params, batch, threads=1):
    """Whole hot path on the CPU; returns (yaha_amd.ResultBatch view, owner) -- keep `owner` alive."""
    r = OracleResult()
    rc = lib().yoracle_run(C.byref(index_view), C.byref(params), C.byref(batch), threads, C.byref(r))
    assert rc == 0
    view = ya.ResultBatch(r.n_reads, r.clump_start, r.clumps, r.ops, r.n_clumps, r.n_ops, r.counters)
    return view, _Owner(r)


class _Owner:
    def __init__(self, r):
        self.r = r

    def __del__(self):
        try:
            lib().yoracle_free_result(C.byref(self.r))
        except Exception:
            pass


def seed_join(index_view, params, batch):
    f, n = C.POINTER(ya.Fragment)(), C.c_uint64()
    lib().yoracle_seed_join(C.byref(index_view), C.byref(params), C.byref(batch), C.byref(f), C.byref(n))
    out = [(f[i].startRefOff, f[i].startQueryOff, f[i].endQueryOff, f[i].refLen, f[i].read_strand) for i in range(n.value)]
    lib().yoracle_free(f)
    return out


def chain(index_view, params, batch):
    f, s, rs, n = C.POINTER(ya.Fragment)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.c_uint64()
    lib().yoracle_chain(C.byref(index_view), C.byref(params), C.byref(batch), C.byref(f), C.byref(s), C.byref(rs), C.byref(n))
    out = []
    for k in range(n.value):
        out.append((rs[k], tuple((f[i].startRefOff, f[i].startQueryOff, f[i].endQueryOff, f[i].refLen) for i in range(s[k], s[k + 1]))))
    for p in (f, s, rs):
        lib().yoracle_free(p)
    return out


def dp_batch(index_view, params, batch, problems):
    arr = (ya.DPProblem * len(problems))(*problems)
    res, ops, nops = C.POINTER(ya.DPResult)(), C.POINTER(C.c_uint32)(), C.c_uint64()
    lib().yoracle_dp_batch(C.byref(index_view), C.byref(params), C.byref(batch), arr, len(problems), C.byref(res), C.byref(ops), C.byref(nops))
    out = []
    for k in range(len(problems)):
        r = res[k]
        out.append((r.score, r.addedQLen, r.addedRLen, tuple((ops[r.op_start + j] & 0xFFFF, chr((ops[r.op_start + j] >> 16) & 0xFF)) for j in range(r.n_ops))))
    lib().yoracle_free(res)
    lib().yoracle_free(ops)
    return out


def have_reference():
    return os.path.exists(REF_BIN)


def run_reference(args, cwd=None):
    """Run the real reference binary (oracle/_ref/yaha); returns (stdout, stderr)."""
    p = subprocess.run([REF_BIN] + list(args), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    return p.stdout.decode(), p.stderr.decode()
