#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection / kernel stats CSVs per kernel (sum over dispatches), and write the
per-launch numbers of one kernel's LARGEST dispatch as JSON (profiles/pmc_latest.json, read by bench.py).

usage: pmc_summary.py <dir> [--emit out.json --kernel k_ext_rows --reads N]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section),
so fetch bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024 is taken as is.  Both are calibrated for wide streaming accesses only."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
emit = kernel = None; reads = 0
a = sys.argv[2:]
while a:
    if a[0] == "--emit": emit = a[1]
    elif a[0] == "--kernel": kernel = a[1]
    elif a[0] == "--reads": reads = int(a[1])
    a = a[2:]
def short(n):
    return n.split("(")[0].replace("void ", "")[:60]
out = {}; perDispatch = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> pass -> dispatch -> counters
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    pas = os.path.basename(os.path.dirname(f))
    if pas == "calib" or os.sep + "calib" + os.sep in f: continue          # the FETCH_SIZE calibration program's kernels (summarised below) are not the hot path's
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = short(row.get("Kernel_Name", "?")); c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0); d = row.get("Dispatch_Id")
        agg[k][c] += v; cnt[k].add(d)
        perDispatch[k][pas].setdefault(d, collections.defaultdict(float))[c] += v
    for k in agg:
        out.setdefault(k, {}).update({c: v for c, v in agg[k].items()}); out[k]["_dispatches_" + pas] = len(cnt[k])
durations = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True), key=lambda f: ("stats1" in f, f)):      # the one-context run last: its durations are the ones kept (a launch that shares the device with three other batches lasts as long as they let it)
    for row in csv.DictReader(open(f)):
        k = short(row.get("Name", "?")); out.setdefault(k, {}).update({"calls": int(row["Calls"]), "total_ns": float(row["TotalDurationNs"]), "avg_ns": float(row["AverageNs"]), "max_ns": float(row.get("MaxNs", 0) or 0), "pct": float(row["Percentage"])})
keys = sorted(out, key=lambda k: -out[k].get("total_ns", 0))
for k in keys[:26]:
    print(k); print("   ", json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in sorted(out[k].items())}))
if emit and kernel:
    ks = [k for k in perDispatch if k.startswith(kernel)]
    if not ks: sys.exit("kernel %s not found" % kernel)
    k = max(ks, key=lambda x: out[x].get("total_ns", 0))
    def biggest(pas, counter):
        ds = perDispatch[k].get(pas, {})
        return max((d.get(counter, 0.0) for d in ds.values()), default=0.0)
    fetch = biggest("fetch", "FETCH_SIZE") * 1024.0 * 2.0; write = biggest("write", "WRITE_SIZE") * 1024.0
    valu = biggest("sq", "SQ_INSTS_VALU"); dur_ns = out[k].get("avg_ns") or out[k].get("max_ns", 0.0)
    ldsc, ldsa, gui = biggest("lds", "SQ_LDS_BANK_CONFLICT"), biggest("lds", "SQ_LDS_IDX_ACTIVE"), biggest("grbm", "GRBM_GUI_ACTIVE")
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ksrc = hashlib.sha256(b"".join(open(os.path.join(here, "yaha_amd", "csrc", "device", f), "rb").read() for f in ("ext_lanes.h", "ext_lanes_pk.h"))).hexdigest()[:16]
    js = {"kernel": kernel, "reads_per_gpu": reads, "git_head": os.environ.get("GIT_HEAD"), "kernel_source_sha16": ksrc, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
          "valu_insts_per_launch": valu, "kernel_ns_largest_launch": dur_ns,
          "lds_bank_conflict_cycles": ldsc, "lds_idx_active_cycles": ldsa, "lds_bank_conflict_frac": (ldsc / ldsa) if ldsa else None,
          "gpu_busy_cycles": gui,      # (no clock derived from it: the cycles are one pass's, the duration another's)
          "lds": {c: biggest("lds", c) for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")},
          "valu_cycles_per_inst": (1024 * 2.4 * dur_ns / valu) if valu else None,
          "valu_issue_frac": (valu * 4.0 / (1024 * 2.4 * dur_ns)) if dur_ns else None,
          "sq": {c: biggest("sq", c) for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU")},
          "source": "rocprofv3 --pmc, separate passes with --kernel-trace only (tools/pmc_pass.sh); FETCH_SIZE x2 per the gfx950 note (see the calibration of byte-wide loads in profiles/); valu_cycles_per_inst = SIMD cycles (1024 SIMDs, 2.4 GHz) per wave64 VALU instruction over the launch; valu_issue_frac prices every instruction at 4 cycles (tools/micro/pk_rate.hip measures 2.5-2.9 for plain two-operand 32-bit ops and 4.3-4.7 for compares, selects, max and three-operand ops)"}
    # the whole path: FETCH_SIZE x 2 + WRITE_SIZE summed over EVERY kernel of the pass, per step (a pass runs the pipeline once per launch of the dominant kernel)
    nsteps = max(1, out[k].get("_dispatches_fetch", 1))
    wf = sum(v.get("FETCH_SIZE", 0.0) for v in out.values()) * 1024.0 * 2.0 / nsteps; ww = sum(v.get("WRITE_SIZE", 0.0) for v in out.values()) * 1024.0 / max(1, out[k].get("_dispatches_write", nsteps))
    js["whole_path"] = {"fetch_bytes_per_step": wf, "write_bytes_per_step": ww, "hbm_bytes_per_step": wf + ww, "steps_in_the_pass": nsteps,
                        "note": "sum over all kernels of one context's pass (the first batch's extra counting passes included), FETCH_SIZE x 2 + WRITE_SIZE as for the kernel above"}
    json.dump(js, open(emit, "w"), indent=1); print("wrote", emit, js)

# FETCH_SIZE calibration (tools/micro/fetch_calib.hip): counter per kernel against the bytes each kernel reads exactly once
cal = os.path.join(root, "calib.json")
if os.path.exists(cal):
    want = {}
    for line in open(cal):
        try:
            j = json.loads(line); want[j["kernel"].split()[0].split("<")[0]] = j
        except Exception:
            pass
    for f in glob.glob(os.path.join(root, "calib", "**", "*counter_collection.csv"), recursive=True):
        print("FETCH_SIZE calibration:")
        seen = collections.OrderedDict()
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == "FETCH_SIZE":
                seen.setdefault((row.get("Dispatch_Id"), short(row.get("Kernel_Name", "?"))), 0.0); seen[(row.get("Dispatch_Id"), short(row.get("Kernel_Name", "?")))] += float(row.get("Counter_Value", 0) or 0)
        lines = [json.loads(l) for l in open(cal) if l.startswith("{")]
        # dispatches are matched to the program's own lines BY KERNEL NAME (the program also launches an initialisation kernel first: matching by position
        # shifted every label by one dispatch in round 2's summary)
        for (d, kn), v in seen.items():
            base = kn.split("(")[0].strip()
            exp = next((l for l in lines if l["kernel"].split()[0] == base), None)
            if exp is None:
                print("   %-28s FETCH_SIZE %.0f KiB (not a calibration kernel)" % (base, v)); continue
            print("   %-28s FETCH_SIZE %.0f KiB = %.3f GB; bytes read exactly once %.3f GB; bytes / (FETCH_SIZE*1024) = %.3f" % (exp["kernel"], v, v * 1024 / 1e9, exp["bytes_read_once"] / 1e9, exp["bytes_read_once"] / (v * 1024) if v else 0))
