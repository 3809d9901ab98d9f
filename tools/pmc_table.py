#!/usr/bin/env python3
"""Per-kernel table from the counter passes of tools/pmc_pass.sh: tools/pmc_table.py <dir>  (the first dispatches of a pass are a one-step run, one context).
Durations come from the --stats run of the same command with one context (stats1)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def short(n): return "rocPRIM / hipCUB kernels (sorts, scans)" if "rocprim" in n else n.split("(")[0].replace("void ", "")[:44]
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    # a pass runs the pipeline once per launch of the dominant kernel (the context's first pass + the timed step): its counters are divided by that number
    rows = list(csv.DictReader(open(f)))
    runs = max(1, len({r["Dispatch_Id"] for r in rows if r["Kernel_Name"].startswith("void k_ext_rows_pk<false") or r["Kernel_Name"].startswith("void k_ext_rows<false, false>")}))
    for row in rows:
        cnt[short(row["Kernel_Name"])][row["Counter_Name"]] += float(row["Counter_Value"] or 0) / runs
dur = {}
for f in glob.glob(os.path.join(root, "stats1", "*kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        k = short(row["Name"]); o = dur.get(k, (0.0, 0, 0.0)); dur[k] = (0.0, o[1] + int(row["Calls"]), o[2] + float(row["TotalDurationNs"]))
# steps of the --stats run: the launches of the dominant kernel there (timed steps + warm-up + the context's first pass)
steps = float(max([v[1] for k, v in dur.items() if k.startswith("k_ext_rows_pk<false") or k.startswith("k_ext_rows<false, false>")] or [8]))
print("per-kernel counters, one context, bench.py defaults (3.1 Gbp genome, 16 384 x 1 kbp reads) (rocprofv3 --pmc in separate passes with --kernel-trace only; tools/pmc_pass.sh, tools/pmc_table.py)")
print("counters = sums over the dispatches of one step; percentages of SQ_WAVE_CYCLES; VALU ms = instructions x 4.4 cycles / (1024 SIMDs x 2.4 GHz); fetch = FETCH_SIZE x 2 KiB, write = WRITE_SIZE KiB;")
print("LDS conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; ms/step = total duration of the kernel's launches over the %d steps of the --stats run / %d" % (steps, steps))
print("%-44s %8s %6s %9s %8s %8s %8s %9s %9s %9s %8s" % ("kernel", "ms/step", "calls", "VALU(M)", "VALU ms", "active%", "memwait%", "iss.wait%", "fetch GB", "write GB", "ldsconf%"))
for k in sorted(dur, key=lambda k: -dur[k][2])[:30]:
    c = cnt.get(k, {}); wc = c.get("SQ_WAVE_CYCLES", 0.0)
    pct = lambda x: ("%8.1f" % (100.0 * c.get(x, 0.0) / wc)) if wc else "       -"
    lds = ("%8.1f" % (100.0 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"])) if c.get("SQ_LDS_IDX_ACTIVE") else "       -"
    print("%-44s %8.2f %6d %9.1f %8.2f %s %s %s %9.2f %9.2f %s" % (k, dur[k][2] / steps / 1e6, dur[k][1], c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_INSTS_VALU", 0) * 4.4 / (1024 * 2.4e9) * 1e3,
          pct("SQ_ACTIVE_INST_ANY"), pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), c.get("FETCH_SIZE", 0) * 2048.0 / 1e9, c.get("WRITE_SIZE", 0) * 1024.0 / 1e9, lds))
