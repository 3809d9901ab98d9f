"""CPU tier: properties of the source tree the reviews asked for and that are easy to lose again."""
import os
import subprocess
import sys

from conftest import ROOT

CSRC = os.path.join(ROOT, "yaha_amd", "csrc")


def _sources():
    out = []
    for d in ("device", "host", "."):
        for f in sorted(os.listdir(os.path.join(CSRC, d))):
            if f.endswith((".h", ".hip", ".cpp")):
                out.append(os.path.join(CSRC, d, f))
    return out


def test_no_source_line_is_longer_than_180_columns():
    """Round 5's `ygpu.hip` had 112 lines over 200 characters (one of 914); since round 6 every line of csrc/ fits 180 columns (tools/wrap_lines.py cuts at token
    boundaries only, so the compiled code does not change)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wrap_lines.py"), "--check"] + _sources(), stdout=subprocess.PIPE)
    assert p.returncode == 0, p.stdout.decode()[:2000]


def test_every_kernel_header_has_one_owner():
    """A kernel is one symbol of the library: every device header that defines a non-template, non-static __global__ function is included by exactly one of the
    translation units (directly or through another header of that unit)."""
    dev = os.path.join(CSRC, "device")
    units = [f for f in os.listdir(dev) if f.endswith(".hip")]
    direct = {}
    for f in os.listdir(dev):
        if f.endswith((".h", ".hip")):
            direct[f] = [l.split('"')[1] for l in open(os.path.join(dev, f)) if l.startswith('#include "') and "/" not in l.split('"')[1]]

    def closure(f, seen):
        for g in direct.get(f, []):
            if g not in seen:
                seen.add(g); closure(g, seen)
        return seen

    import re
    for h in (f for f in direct if f.endswith(".h")):
        text = open(os.path.join(dev, h)).read()
        lines = text.split("\n")
        # (a kernel template may be instantiated by several units -- scan.h's sums in prims.hip and index_build.hip -- and a `static` kernel is a unit's own)
        plain = [i for i, l in enumerate(lines) if l.startswith("__global__") and not (i > 0 and lines[i - 1].lstrip().startswith("template"))]
        if not plain:
            continue
        owners = [u for u in units if h in closure(u, set())]
        assert len(owners) == 1, "%s defines kernels and is seen by %s" % (h, owners)
