"""Shared fixtures.  `-m "not gpu"` runs on the CPU container (oracle vs golden vectors, host logic, C-ABI
exports); `-m gpu` tests are the parity tests proper and call the HIP path through the C-ABI."""
import ctypes as C
import gzip
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")
    # The whole tier runs with the state-word check on (device/prims.hip: every ygpu_run / ygpu_postfilter ends with a pass over the look-back and bucket words
    # on the device and fails unless all are zero again) -- the library reads the switch once, and the command lines the tests start inherit it.
    os.environ.setdefault("YGPU_CHECK_STATE", "1")


def _build():
    lib = os.path.join(ROOT, "yaha_amd", "csrc", "libyaha_hip.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(ROOT, "yaha_amd", "csrc", "yaha")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "yaha_amd", "csrc"), "-j8"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    sim = os.path.join(ROOT, "tools", "yaha_sim")
    if not os.path.exists(sim):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", sim, os.path.join(ROOT, "tools", "yaha_sim.cpp")])


@pytest.fixture(scope="session")
def meta():
    return json.load(open(os.path.join(GOLDEN, "golden.json")))


@pytest.fixture(scope="session")
def work(tmp_path_factory):
    """Scratch directory holding the unpacked golden inputs and the index built by THIS repo's indexer."""
    _build()
    import yaha_amd as ya
    d = str(tmp_path_factory.mktemp("yaha"))
    for f in os.listdir(GOLDEN):
        if f.endswith(".gz") and not f.endswith(".out.gz"):
            with gzip.open(os.path.join(GOLDEN, f), "rb") as g, open(os.path.join(d, f[:-3]), "wb") as o:
                shutil.copyfileobj(g, o)
    ya.build_index(["-g", os.path.join(d, "genome_small.fa"), "-L", "11"])
    return d


@pytest.fixture(scope="session")
def index11(work):
    return os.path.join(work, "genome_small.X11_01_65525S")


def golden_lines(name):
    with gzip.open(os.path.join(GOLDEN, name + ".out.gz"), "rb") as g:
        return g.read().decode().split("\n")


def strip_pg(text):
    return [l for l in text.split("\n") if not l.startswith("@PG")]


def oflag_args(oflag):
    return [oflag, "stdout"]


def nib2_v1_copy(work, dst_dir):
    """The golden genome's `.nib2` rewritten as a VERSION 1 file (Compress.c:89-128: 12-byte sequence entries {start, length, nameOffset << 16 | nameLength} instead of
    version 2's 16-byte ones; the reference still loads both, nothing writes version 1 any more) with its index beside it.  Returns the index path."""
    import struct
    src = open(os.path.join(work, "genome_small.nib2"), "rb").read()
    magic, version, base_off, n = struct.unpack_from("<4I", src, 0)
    assert magic == 0x01020304 and version == 2
    ent = [struct.unpack_from("<4I", src, 16 + 16 * i) for i in range(n)]
    names_at = 16 + 16 * n + 4
    names = src[names_at:base_off]                                            # the name block, zero padded to 4 bytes
    assert all(e[2] < 65536 and e[3] < 65536 for e in ent)
    head = b"".join(struct.pack("<3I", e[0], e[1], (e[2] << 16) | e[3]) for e in ent) + struct.pack("<I", 0) + names
    v1 = struct.pack("<4I", magic, 1, 16 + len(head), n) + head + src[base_off:]
    os.makedirs(dst_dir, exist_ok=True)
    open(os.path.join(dst_dir, "genome_small.nib2"), "wb").write(v1)
    idx = [f for f in os.listdir(work) if f.startswith("genome_small.X11_")][0]
    shutil.copyfile(os.path.join(work, idx), os.path.join(dst_dir, idx))
    return os.path.join(dst_dir, idx)
