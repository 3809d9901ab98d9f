#!/usr/bin/env python3
"""How do the kernels of several batches in flight share the device?  Reads a rocprofv3 kernel trace (…_kernel_trace.csv of `tools/measure.sh kstats TAG N`) and prints, over
the run's busiest window (from the first to the last k_ext_rows_pk launch): the share of the time with at least one kernel running, with a rows kernel running, how much of
every kernel's duration overlaps a rows kernel, and each kernel's mean duration (to set beside the one-context run's).   tools/overlap.py trace.csv [trace_one_context.csv [timed steps (8)]]"""
import csv, sys, collections

def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "rocprim" in n or "hipcub" in n: n = "library (rocPRIM / hipCUB)"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort(); return rows

def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot

def overlap_with(iv, merged):      # total length of iv's intersection with the merged (disjoint, sorted) intervals
    tot = 0
    for s, e in iv:
        for ms, me in merged:
            if me <= s: continue
            if ms >= e: break
            tot += min(e, me) - max(s, ms)
    return tot

def merge(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out

rows = load(sys.argv[1])
rk = [(s, e) for s, e, n in rows if n.startswith("k_ext_rows_pk<false")]
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 8
t0, t1 = rk[-nlast][0], rk[-1][1]               # the timed steps (bench.py warms the contexts up one after the other: no overlap there)
if len(sys.argv) > 4:                            # a window by kernel name instead: from the first to the last launch of kernels whose name contains argv[4] (e.g. k_oqc: the post-filter leg)
    sel = [(s, e) for s, e, n in rows if sys.argv[4] in n]
    t0, t1 = sel[0][0], sel[-1][1]
    rk = [x for x in rk if x[0] >= t0 and x[1] <= t1]
win = [(max(s, t0), min(e, t1), n) for s, e, n in rows if e > t0 and s < t1]
span = t1 - t0
busy = union([(s, e) for s, e, n in win]); rowsM = merge([(s, e) for s, e, n in win if n.startswith("k_ext_rows_pk")]); rowsBusy = sum(e - s for s, e in rowsM)
nrows = sum(1 for s, e, n in win if n.startswith("k_ext_rows_pk<false")) - 1       # from the first launch's start to the last one's end: n - 1 batch periods, roughly
print("window %.1f ms, %d batches: %.2f ms a batch; some kernel running %.1f%% of it, a rows kernel %.1f%%; sum of kernel durations %.1f ms a batch" % (span / 1e6, nrows, span / 1e6 / nrows, 100.0 * busy / span, 100.0 * rowsBusy / span, sum(e - s for s, e, n in win) / 1e6 / nrows))
one = {}
if len(sys.argv) > 2:
    d = collections.defaultdict(list)
    for s, e, n in load(sys.argv[2]): d[n].append(e - s)
    one = {n: sum(v) / len(v) for n, v in d.items()}
by = collections.defaultdict(list)
for s, e, n in win: by[n].append((s, e))
print("%-44s %6s %9s %9s %8s %s" % ("kernel", "calls", "ms/batch", "mean ms", "in rows%", "mean ms with one context"))
for n, iv in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:32]:
    tot = sum(e - s for s, e in iv)
    print("%-44s %6d %9.3f %9.3f %8.1f %s" % (n[:44], len(iv), tot / 1e6 / nrows, tot / 1e6 / len(iv), 100.0 * overlap_with(iv, rowsM) / max(1, tot), ("%.3f" % (one[n] / 1e6)) if n in one else ""))
