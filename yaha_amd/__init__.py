"""yaha_amd -- Python mirror (ctypes) of the C-ABI in include/yaha_hip.h.

The product is the shared library ``yaha_amd/csrc/libyaha_hip.so`` (hand-written HIP kernels for gfx950 plus
the host stages) and the ``yaha`` command line next to it.  This module only loads it and mirrors the
structs; it never falls back to anything else: if the library is missing, importing the bindings raises.
PyTorch is not needed here (bench.py uses it for torch.distributed plumbing only).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YAHA_HIP_LIB", os.path.join(_HERE, "csrc", "libyaha_hip.so"))   # override only for diagnostic builds
CLI_PATH = os.path.join(_HERE, "csrc", "yaha")


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "wordLen", "maxHits", "bandWidth", "maxGap", "maxIntron", "minMatch", "maxDesert", "minNonOverlap",
        "minRawScore", "minExtLength", "GOCost", "GECost", "RCost", "MScore", "XCutoff")] + [("minIdentity", C.c_float)]


class IndexView(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("n_base_bytes", C.c_uint64), ("maxROff", C.c_uint32),
                ("startingOffs", C.c_void_p), ("ROA", C.c_void_p), ("totalMatches", C.c_uint32), ("wordLen", C.c_int32)]


class ReadBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("codes", C.c_void_p), ("offsets", C.c_void_p)]


class Clump(C.Structure):
    _fields_ = [("sro", C.c_uint32), ("sqo", C.c_uint16), ("eqo", C.c_uint16), ("refLen", C.c_uint16),
                ("totScore", C.c_uint16), ("totLength", C.c_uint16), ("matchedBases", C.c_uint16),
                ("mismatchedBases", C.c_uint16), ("gapBases", C.c_uint16), ("status", C.c_uint8),
                ("reserved", C.c_uint8), ("op_start", C.c_uint32), ("n_ops", C.c_uint32)]


COUNTER_NAMES = ("kmer_lookups", "hits", "fragments", "regions", "clumps_formed", "clumps_scored",
                 "dp_ext_calls", "dp_ext_rows", "dp_ext_cells", "dp_gap_calls", "dp_gap_rows", "dp_gap_cells",
                 "perfect_ext_bases", "ref_bases_touched", "ops_out", "splits")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


class ResultBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("clump_start", C.POINTER(C.c_uint32)), ("clumps", C.POINTER(Clump)),
                ("ops", C.POINTER(C.c_uint32)), ("n_clumps", C.c_uint64), ("n_ops", C.c_uint64), ("counters", Counters)]


class Fragment(C.Structure):
    _fields_ = [("startRefOff", C.c_uint32), ("startQueryOff", C.c_uint16), ("endQueryOff", C.c_uint16),
                ("refLen", C.c_uint16), ("reserved", C.c_uint16), ("read_strand", C.c_uint32)]


class DPProblem(C.Structure):
    _fields_ = [("read", C.c_uint32), ("strand", C.c_uint8), ("mode", C.c_uint8), ("qOff", C.c_uint16),
                ("qLen", C.c_uint16), ("rLen", C.c_uint16), ("rOff", C.c_uint32)]


class DPResult(C.Structure):
    _fields_ = [("score", C.c_int32), ("addedQLen", C.c_uint16), ("addedRLen", C.c_uint16),
                ("op_start", C.c_uint32), ("n_ops", C.c_uint32)]


class PostfilterParams(C.Structure):
    _fields_ = [("minNonOverlap", C.c_int32), ("BPCost", C.c_int32), ("maxBPLog", C.c_int32), ("FBS", C.c_int32), ("FBS_PSLength", C.c_float), ("FBS_PSScore", C.c_float),
                ("bppVmin", C.c_int32), ("bppN", C.c_int32), ("bppThr", C.c_void_p), ("n_seqs", C.c_uint32), ("seq_start", C.c_void_p), ("seq_length", C.c_void_p)]


class OutClump(C.Structure):
    _fields_ = [("c", Clump), ("status", C.c_uint8), ("mapQuality", C.c_uint8), ("numSecondaries", C.c_uint16), ("matchedPrimary", C.c_uint16), ("primaryCount", C.c_uint16)]


class FilteredBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("clump_start", C.POINTER(C.c_uint32)), ("clumps", C.POINTER(OutClump)), ("ops", C.POINTER(C.c_uint32)),
                ("n_clumps", C.c_uint64), ("n_ops", C.c_uint64), ("counters", Counters)]


class ArenaProfile(C.Structure):
    _fields_ = [("n", C.c_uint32), ("last_clump_slots", C.c_uint32), ("cap", C.c_uint64 * 224), ("trace_ratio", C.c_double), ("ops_ratio", C.c_double), ("last_fall", C.c_int64), ("bases", C.c_uint64)]


DP_FULL, DP_BANDED, DP_EXT_FWD, DP_EXT_REV = 0, 1, 2, 3
DP_KERNELS_AUTO, DP_KERNELS_WAVE, DP_KERNELS_LANES, DP_KERNELS_LANES_CAREFUL = 0, 1, 2, 3

EXPORTS = (
    "ygpu_device_count", "ygpu_init", "ygpu_init_multi", "ygpu_clone", "ygpu_destroy", "ygpu_last_error", "ygpu_memory", "ygpu_park", "ygpu_get_arena_profile", "ygpu_presize", "ygpu_upload", "ygpu_upload_nowait", "ygpu_run", "ygpu_collect", "ygpu_result_size", "ygpu_collect_into", "ygpu_host_alloc", "ygpu_host_free", "ygpu_set_postfilter", "ygpu_postfilter_snapshot", "ygpu_postfilter", "ygpu_postfilter_drop", "ygpu_inject_results", "ygpu_selftest_primitives", "ygpu_trace_volume", "ygpu_filtered_size", "ygpu_collect_filtered", "ygpu_last_timing",
    "ygpu_submit", "ygpu_poll", "ygpu_wait", "ygpu_seed_join", "ygpu_chain", "ygpu_dp_batch", "ygpu_dp_batch_ex",
    "yaha_session_open", "yaha_session_close", "yaha_session_error", "yaha_session_params",
    "yaha_session_index_view", "yaha_session_header", "yaha_session_next_batch", "yaha_session_emit", "yaha_session_postfilter_params", "yaha_session_emit_filtered",
    "yaha_build_index", "yaha_main")

_lib = None


def lib():
    """Load libyaha_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(or make -C yaha_amd/csrc); there is no fallback path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ygpu_last_error.restype = C.c_char_p
        L.yaha_session_error.restype = C.c_char_p
        for name in EXPORTS:
            getattr(L, name)
        _lib = L
    return _lib


def _argv(args):
    arr = (C.c_char_p * len(args))(*[a.encode() if isinstance(a, str) else a for a in args])
    return len(args), arr


def build_index(args):
    """`yaha -g genome.fa [-L k] [-S s] [-H h]` (reference Main.c:554-628)."""
    n, arr = _argv(args)
    rc = lib().yaha_build_index(n, arr)
    if rc != 0:
        raise RuntimeError("yaha_build_index%r failed: %d" % (tuple(args), rc))


class Session:
    """Host stages around the hot path: argument parsing, .nib2/index mapping, read batching, OQC + SAM."""

    def __init__(self, args):
        self._h = C.c_void_p()
        n, arr = _argv(args)
        rc = lib().yaha_session_open(n, arr, C.byref(self._h))
        if rc != 0:
            msg = lib().yaha_session_error(self._h).decode() if self._h else "bad arguments"
            raise RuntimeError("yaha_session_open failed: %d %s" % (rc, msg))
        self.params = Params()
        lib().yaha_session_params(self._h, C.byref(self.params))
        self.index = IndexView()
        lib().yaha_session_index_view(self._h, C.byref(self.index))

    def header(self):
        t, n = C.c_char_p(), C.c_size_t()
        lib().yaha_session_header(self._h, C.byref(t), C.byref(n))
        return C.string_at(t, n.value).decode()

    def next_batch(self, max_reads):
        b = ReadBatch()
        lib().yaha_session_next_batch(self._h, max_reads, C.byref(b))
        return b

    def emit(self, result):
        t, n = C.c_char_p(), C.c_size_t()
        rc = lib().yaha_session_emit(self._h, C.byref(result), C.byref(t), C.byref(n))
        if rc != 0:
            raise RuntimeError("yaha_session_emit failed: %d" % rc)
        return C.string_at(t, n.value).decode()

    def emit_filtered(self, filtered):
        t, n = C.c_char_p(), C.c_size_t()
        rc = lib().yaha_session_emit_filtered(self._h, C.byref(filtered), C.byref(t), C.byref(n))
        if rc != 0:
            raise RuntimeError("yaha_session_emit_filtered failed: %d" % rc)
        return C.string_at(t, n.value).decode()

    def close(self):
        if self._h:
            lib().yaha_session_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def device_count():
    """HIP devices visible to this process (ygpu_device_count)."""
    return int(lib().ygpu_device_count())


class Context:
    """One device context = the reference's per-thread QueryState, batched (include/yaha_hip.h)."""

    def __init__(self, index_view, params, device=0, parent=None):
        self._h = C.c_void_p()
        if parent is not None:                      # second context on the parent's device, sharing its index image
            rc = lib().ygpu_clone(parent._h, C.byref(self._h))
        else:
            rc = lib().ygpu_init(device, C.byref(index_view), C.byref(params), C.byref(self._h))
        if rc != 0:
            msg = lib().ygpu_last_error(self._h).decode() if self._h else ""
            raise RuntimeError("ygpu_init failed: %d %s (the HIP path is mandatory; there is no CPU fallback)" % (rc, msg))

    @classmethod
    def on_devices(cls, index_view, params, devices, ctx_per_device=1):
        """Contexts on the listed devices through ygpu_init_multi: the first device takes the index image from the host, every further one from the device before
        it (the same device may be listed twice: two images on one device); ctx_per_device contexts a device share its image.  Returns them device-major."""
        n = len(devices); m = n * ctx_per_device
        devs = (C.c_int * n)(*devices); hs = (C.c_void_p * m)(); rcs = (C.c_int * n)()
        rc = lib().ygpu_init_multi(devs, n, ctx_per_device, C.byref(index_view), C.byref(params), hs, rcs)
        ctxs = []
        for k in range(m):
            c = cls.__new__(cls); c._h = C.c_void_p(hs[k]); ctxs.append(c)
        if rc != 0:
            msgs = ["device %d: %d %s" % (devices[k], rcs[k], lib().ygpu_last_error(ctxs[k * ctx_per_device]._h).decode() if hs[k * ctx_per_device] else "") for k in range(n) if rcs[k]]
            for c in reversed(ctxs):
                c.close()
            raise RuntimeError("ygpu_init_multi failed: %d (%s)" % (rc, "; ".join(msgs)))
        return ctxs

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %d %s" % (what, rc, lib().ygpu_last_error(self._h).decode()))

    def memory(self):
        """(free, total, this context's own) device bytes (ygpu_memory)."""
        f, t, m = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(lib().ygpu_memory(self._h, C.byref(f), C.byref(t), C.byref(m)), "ygpu_memory")
        return int(f.value), int(t.value), int(m.value)

    def selftest_primitives(self, n, seed=1, key_bits=8):
        """The path's own exclusive sums and orderings (device/scan.h), the post-filter's sort on the wave and A2's workgroup sort (device/wgsort.h, both rankings)
        against the host's loops (ygpu_selftest_primitives); raises on a difference."""
        self._check(lib().ygpu_selftest_primitives(self._h, C.c_uint32(n), C.c_uint32(seed), C.c_int(key_bits)), "ygpu_selftest_primitives")

    def arena_profile(self):
        """Capacities of this context's arenas and its batch-to-batch estimates (ygpu_get_arena_profile)."""
        p = ArenaProfile()
        self._check(lib().ygpu_get_arena_profile(self._h, C.byref(p)), "ygpu_get_arena_profile")
        return p

    def presize(self, profile):
        """Give this context the arenas of `profile` in one go (ygpu_presize)."""
        self._check(lib().ygpu_presize(self._h, C.byref(profile)), "ygpu_presize")

    def park(self):
        """Release this context's arenas and its share of the device's memory budget (ygpu_park); it can only be closed afterwards."""
        self._check(lib().ygpu_park(self._h), "ygpu_park")

    def upload(self, batch):
        self._n_reads = int(batch.n_reads)
        self._check(lib().ygpu_upload(self._h, C.byref(batch)), "ygpu_upload")

    def run(self):
        self._check(lib().ygpu_run(self._h), "ygpu_run")

    def collect(self):
        r = ResultBatch()
        self._check(lib().ygpu_collect(self._h, C.byref(r)), "ygpu_collect")
        return r

    def set_postfilter(self, session):
        """The session's OQC / FBS parameters for ygpu_postfilter on this context (raises for runs the device stage does not take)."""
        p = PostfilterParams()
        if lib().yaha_session_postfilter_params(session._h, C.byref(p)) != 0:
            raise RuntimeError("yaha_session_postfilter_params: " + lib().yaha_session_error(session._h).decode())
        self._check(lib().ygpu_set_postfilter(self._h, C.byref(p)), "ygpu_set_postfilter")

    def inject_results(self, result):
        """Stage-level test entry: a ResultBatch placed on the device as if ygpu_run had produced it for the uploaded reads."""
        self._check(lib().ygpu_inject_results(self._h, C.byref(result)), "ygpu_inject_results")

    def postfilter_snapshot(self):
        """On the context's thread, after run(): the stage's copy of the batch's results.  The context is then free for the next upload() / run() while
        postfilter() -- on another thread -- works on this one."""
        self._check(lib().ygpu_postfilter_snapshot(self._h), "ygpu_postfilter_snapshot")
        self._snap_reads = self._n_reads

    def postfilter(self):
        """OQC, filter by similarity and mapping quality on the device; returns the clumps that are printed (FilteredBatch; arrays owned by this object).
        Works on the snapshot taken by postfilter_snapshot(), or takes one of the last run()'s results itself."""
        n_reads = getattr(self, "_snap_reads", None)
        if n_reads is None:
            n_reads = self._n_reads
        self._snap_reads = None
        self._check(lib().ygpu_postfilter(self._h), "ygpu_postfilter")
        nc, no = C.c_uint64(), C.c_uint64()
        self._check(lib().ygpu_filtered_size(self._h, C.byref(nc), C.byref(no)), "ygpu_filtered_size")
        self._f_cs = (C.c_uint32 * (n_reads + 1))(); self._f_cl = (OutClump * max(1, nc.value))(); self._f_ops = (C.c_uint32 * max(1, no.value))()
        r = FilteredBatch()
        self._check(lib().ygpu_collect_filtered(self._h, self._f_cs, self._f_cl, self._f_ops, C.byref(r)), "ygpu_collect_filtered")
        return r

    def trace_volume(self):
        """The trace stream of the last run (ygpu_trace_volume): records written by the rows kernel, records a traceback can visit, calls, walkers, bytes a record, arena bytes."""
        v = (C.c_uint64 * 6)()
        self._check(lib().ygpu_trace_volume(self._h, v), "ygpu_trace_volume")
        return {"records_written": v[0], "records_visitable": v[1], "calls": v[2], "walkers": v[3], "record_bytes": v[4], "arena_bytes": v[5]}

    def timing(self):
        tot, n = C.c_float(), C.c_int()
        names, ms = C.POINTER(C.c_char_p)(), C.POINTER(C.c_float)()
        self._check(lib().ygpu_last_timing(self._h, C.byref(tot), C.byref(n), C.byref(names), C.byref(ms)), "ygpu_last_timing")
        return tot.value, {names[i].decode(): ms[i] for i in range(n.value)}

    def seed_join(self):
        f, n = C.POINTER(Fragment)(), C.c_uint64()
        self._check(lib().ygpu_seed_join(self._h, C.byref(f), C.byref(n)), "ygpu_seed_join")
        return f, n.value

    def chain(self):
        f, s, rs, n = C.POINTER(Fragment)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.c_uint64()
        self._check(lib().ygpu_chain(self._h, C.byref(f), C.byref(s), C.byref(rs), C.byref(n)), "ygpu_chain")
        return f, s, rs, n.value

    def dp_batch(self, problems, kernels=DP_KERNELS_AUTO):
        arr = (DPProblem * len(problems))(*problems)
        res, ops, nops = C.POINTER(DPResult)(), C.POINTER(C.c_uint32)(), C.c_uint64()
        self._check(lib().ygpu_dp_batch_ex(self._h, arr, len(problems), kernels, C.byref(res), C.byref(ops), C.byref(nops)), "ygpu_dp_batch_ex")
        return res, ops, nops.value

    def submit(self, batch):
        t = C.c_uint64()
        self._check(lib().ygpu_submit(self._h, C.byref(batch), C.byref(t)), "ygpu_submit")
        return t.value

    def poll(self, ticket):
        return lib().ygpu_poll(self._h, C.c_uint64(ticket))

    def wait(self, ticket):
        r = ResultBatch()
        self._check(lib().ygpu_wait(self._h, C.c_uint64(ticket), C.byref(r)), "ygpu_wait")
        return r

    def close(self):
        if self._h:
            lib().ygpu_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def result_records(r):
    """Flatten a ResultBatch into plain Python tuples (for bit-exact comparisons in tests)."""
    out = []
    for i in range(r.n_reads):
        rec = []
        for k in range(r.clump_start[i], r.clump_start[i + 1]):
            c = r.clumps[k]
            ops = tuple((r.ops[c.op_start + j] & 0xFFFF, chr((r.ops[c.op_start + j] >> 16) & 0xFF)) for j in range(c.n_ops))
            rec.append((c.sro, c.sqo, c.eqo, c.refLen, c.totScore, c.totLength, c.matchedBases, c.mismatchedBases,
                        c.gapBases, c.status, ops))
        out.append(rec)
    return out
