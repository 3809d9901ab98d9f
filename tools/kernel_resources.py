#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in the built library (no kernel may spill vector registers to scratch memory).
    python tools/kernel_resources.py [libyaha_hip.so]        -> one line per kernel, sorted by name; exit code 1 if any kernel spills"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "yaha_amd", "csrc", "libyaha_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    tmp = os.path.join(td, os.path.basename(lib)); subprocess.check_call(["cp", lib, tmp])
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
    bad = 0
    for f in sorted(os.listdir(td)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)], stdout=subprocess.PIPE).stdout.decode()
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], stdout=subprocess.PIPE).stdout.decode().strip()
            name = re.sub(r"\(.*", "", name)
            spill = int(g("vgpr_spill_count"))              # (SGPR "spills" go to lanes of a VGPR, no memory traffic: listed, not counted)
            bad += spill > 0
            print("%-46s vgpr %3s sgpr %3s spill v%s s%s scratch %5s B  lds %6s B%s" % (name[:46], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), "   <-- SPILLS" if spill else ""))
sys.exit(1 if bad else 0)
