"""CPU tier: host logic and the C-ABI surface (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import pytest

import yaha_amd as ya
from conftest import ROOT


def test_library_exports_every_symbol_the_header_declares():
    hdr = open(os.path.join(ROOT, "include", "yaha_hip.h")).read()
    declared = set(re.findall(r"\b(ygpu_[a-z_]+|yaha_[a-z_]+)\s*\(", hdr))
    declared -= {"ygpu_ctx", "yaha_session"}
    assert declared, "no declarations parsed"
    L = ya.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libyaha_hip.so does not export " + name
    assert set(ya.EXPORTS) == declared


def test_struct_layouts_match_the_header():
    assert C.sizeof(ya.Clump) == 32 and C.sizeof(ya.Fragment) == 16 and C.sizeof(ya.DPProblem) == 16 and C.sizeof(ya.DPResult) == 16
    assert C.sizeof(ya.Params) == 64 and C.sizeof(ya.Counters) == 128
    assert C.sizeof(ya.OutClump) == 40 and C.sizeof(ya.PostfilterParams) == 64 and ya.OutClump.status.offset == 32 and ya.OutClump.primaryCount.offset == 38


def test_derived_parameters_follow_the_reference_defaults(work, index11):
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s:
        p = s.params
        assert (p.wordLen, p.maxHits, p.bandWidth, p.maxGap, p.maxIntron, p.minMatch, p.maxDesert) == (11, 650, 5, 50, 50, 25, 50)
        assert (p.minNonOverlap, p.minRawScore, p.minExtLength, p.GOCost, p.GECost, p.RCost, p.MScore, p.XCutoff) == (25, 25, 5, 5, 2, 3, 1, 25)
        assert abs(p.minIdentity - 0.9) < 1e-6 and p.minIdentity != 0.9      # float, not double
        assert "@HD\tVN:1.0" in s.header() and "@SQ\tSN:chr1" in s.header()


def test_reader_rules(work, index11, tmp_path):
    q = tmp_path / "odd.fa"
    q.write_text(">first read with spaces\nACGTACGTACGTACGTACGTAC\nGTACGTNNACGT\n>tooshort\nACGT\n>" + "x" * 250 + "\nACGTACGTACGTACGTACGTACGTACGT\n>last\nacgtacgtacgtacgtacgtacgt")
    with ya.Session(["-x", index11, "-q", str(q)]) as s:
        b = s.next_batch(100)
        assert b.n_reads == 3                      # the 4-base read is skipped (Query.c:207-211)
        offs = C.cast(b.offsets, C.POINTER(C.c_uint64))
        assert [offs[i + 1] - offs[i] for i in range(3)] == [34, 28, 24]
        codes = C.cast(b.codes, C.POINTER(C.c_uint8))
        assert [codes[i] for i in range(4)] == [2, 1, 3, 0] and codes[28] == 4      # A C G T ; N


def test_context_creation_fails_loudly_without_a_gpu_or_with_bad_params(work, index11):
    import torch
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s:
        if not torch.cuda.is_available():
            with pytest.raises(RuntimeError):
                ya.Context(s.index, s.params)
        p = ya.Params.from_buffer_copy(s.params)
        p.bandWidth = 300
        with pytest.raises(RuntimeError):
            ya.Context(s.index, p)


def test_cli_index_then_usage(work, tmp_path):
    import subprocess
    r = subprocess.run([ya.CLI_PATH], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"Usage" in r.stderr
    r = subprocess.run([ya.CLI_PATH, "-x", "nonexistent.X11_01_65525S", "-q", "none.fa"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0


def test_cli_compress_and_uncompress_only(work, meta, tmp_path):
    """-c / -u (Main.c:284-293: builds of the reference without COMPILE_USER_MODE): genome.fa -> genome.nib2 only, genome.nib2 -> genome.fasta (50 bases a line,
    Compress.c:337-397).  The .nib2 is the golden one; the FASTA compresses back to the same bytes."""
    import hashlib, shutil, subprocess
    g = str(tmp_path / "g.fa"); shutil.copy(os.path.join(work, "genome_small.fa"), g)
    assert subprocess.run([ya.CLI_PATH, "-g", g, "-c"], stderr=subprocess.PIPE).returncode == 0
    nib = str(tmp_path / "g.nib2")
    assert hashlib.sha256(open(nib, "rb").read()).hexdigest() == meta["index"]["genome_small.nib2"]["sha256"]
    assert not [f for f in os.listdir(tmp_path) if ".X" in f]                      # no index was made
    assert subprocess.run([ya.CLI_PATH, "-g", nib, "-u"], stderr=subprocess.PIPE).returncode == 0
    fasta = open(tmp_path / "g.fasta").read().split("\n")
    assert fasta[0].startswith(">") and all(len(l) <= 50 for l in fasta if not l.startswith(">")) and len(fasta[1]) == 50
    orig = "".join(l for l in open(g).read().split("\n") if not l.startswith(">")).upper()
    assert "".join(l for l in fasta if not l.startswith(">")) == orig
    os.rename(nib, str(tmp_path / "first.nib2"))
    assert subprocess.run([ya.CLI_PATH, "-g", str(tmp_path / "g.fasta"), "-c"], stderr=subprocess.PIPE).returncode == 0
    assert open(nib, "rb").read() == open(tmp_path / "first.nib2", "rb").read()
    assert subprocess.run([ya.CLI_PATH, "-g", nib, "-c"], stderr=subprocess.PIPE).returncode != 0      # -c wants a FASTA file
    assert subprocess.run([ya.CLI_PATH, "-g", g, "-u"], stderr=subprocess.PIPE).returncode != 0        # -u a .nib2


def test_bench_picks_the_genome_the_box_can_hold(tmp_path, monkeypatch):
    # bench.py's default workload is G-hg18scale (3.1 Gbp) when the box has the disk and memory for it, else the 100 Mbp genome; a cache that already holds
    # the 3.1 Gbp index is always used
    import importlib, sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    import shutil as _sh
    usage = type("U", (), {})
    small = usage(); small.free = 10 << 30
    monkeypatch.setattr(_sh, "disk_usage", lambda p: small)
    assert bench.pick_genome_mbp(str(tmp_path), 42) == 100
    (tmp_path / "g3100m_s42.X15_01_65525S.done").write_text("ok")
    assert bench.pick_genome_mbp(str(tmp_path), 42) == 3100


def test_no_kernel_spills_vector_registers():
    """Every kernel of the built library keeps its vector registers out of scratch memory (tools/kernel_resources.py reads the code object's notes): round 4's 32-bit
    rows kernel spilled four and lost 1.2 %."""
    import subprocess, sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = p.stdout.decode()
    assert "k_ext_rows_pk" in out and "k_frag_scan_build" in out, p.stderr.decode()[-500:]
    assert p.returncode == 0 and "SPILLS" not in out, [l for l in out.split("\n") if "SPILLS" in l]
