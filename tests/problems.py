"""Helpers that derive realistic DP problems (gap fills between chained fragments, X-drop extensions from clump
ends) from the oracle's chain stage, used by the CPU lane-model test and the GPU dp_batch parity test."""
import ctypes as C
import random

import numpy as np

import oracle
import yaha_amd as ya

COMP = [2, 3, 0, 1, 4, 12, 7, 6, 9, 8, 15, 11, 5, 13, 14, 10]


def batch_arrays(s, b):
    nb = s.index.n_base_bytes
    bases = np.ctypeslib.as_array(C.cast(s.index.bases, C.POINTER(C.c_uint8)), shape=(nb,))
    offs = np.ctypeslib.as_array(C.cast(b.offsets, C.POINTER(C.c_uint64)), shape=(b.n_reads + 1,))
    codes = np.ctypeslib.as_array(C.cast(b.codes, C.POINTER(C.c_uint8)), shape=(int(offs[-1]),))
    return bases, offs, codes


def dp_problems_from_chain(s, b, limit=2000, seed=1):
    P = s.params
    offs = np.ctypeslib.as_array(C.cast(b.offsets, C.POINTER(C.c_uint64)), shape=(b.n_reads + 1,))
    probs = []
    for rs, frags in oracle.chain(s.index, P, b):
        read, strand = rs >> 1, rs & 1
        qlen = int(offs[read + 1] - offs[read])
        for a, bb in zip(frags, frags[1:]):
            qg = max(bb[1] - a[2] - 1, 0)
            ero = a[0] + a[3] - 1
            rg = max(bb[0] - ero - 1, 0)
            if qg > 0 and rg > 0 and not (qg == 1 and rg == 1):
                mode = ya.DP_BANDED if abs(qg - rg) + 2 * P.bandWidth + 1 < rg else ya.DP_FULL
                probs.append(ya.DPProblem(read, strand, mode, a[2] + 1, qg, rg, ero + 1))
        f0, fn = frags[0], frags[-1]
        if f0[1] >= 5 and f0[0] >= 1:
            probs.append(ya.DPProblem(read, strand, ya.DP_EXT_REV, f0[1] - 1, min(f0[1], f0[0]), 0, f0[0] - 1))
        ero = fn[0] + fn[3] - 1
        fl = min(qlen - 1 - fn[2], s.index.maxROff - ero)
        if fl >= 5:
            probs.append(ya.DPProblem(read, strand, ya.DP_EXT_FWD, fn[2] + 1, fl, 0, ero + 1))
    random.Random(seed).shuffle(probs)
    return probs[:limit]


def read_fasta(path):
    names, seqs, cur = [], [], []
    for line in open(path):
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
                cur = []
            names.append(line[1:].split()[0])
        else:
            cur.append(line.strip())
    seqs.append("".join(cur))
    return names, seqs


def write_long_indel_reads(genome_fa, out, n, seed, max_del=1250, junk=0, flank=450):
    """Reads that carry ONE long deletion (up to max_del reference bases) or insertion, with two substitutions right at the junction so that the
    seeds stop short of it and the gap between the chained fragments needs a DP whose strip is hundreds of columns wide (needs a large -G);
    junk > 0 additionally drops that many random query bases into the junction (a gap fill with many rows AND many columns; needs -MD >= junk)."""
    rnd = random.Random(seed)
    names, seqs = read_fasta(genome_fa)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}

    def mut(s, k):
        s = list(s)
        for p in rnd.sample(range(len(s)), k):
            s[p] = rnd.choice([c for c in "ACGT" if c != s[p]])
        return "".join(s)
    with open(out, "w") as f:
        for i in range(n):
            si = rnd.randrange(len(seqs))
            g = seqs[si]
            d = rnd.randrange(max_del // 2, max_del)
            fl, fr = flank - 30, flank + 30
            a = rnd.randrange(0, len(g) - fl - fr - 300 - d)
            left, right = g[a:a + fl], None
            if i % 3 != 2:                                       # deletion in the read
                right = g[a + fl + d:a + fl + d + fr]
                mid = "".join(rnd.choice("ACGT") for _ in range(junk))
            else:                                                # insertion in the read
                right = g[a + fl:a + fl + fr]
                mid = "".join(rnd.choice("ACGT") for _ in range(d))
            left = mut(left[:-9], fl // 100) + mut(left[-9:], 2)
            right = mut(right[:9], 2) + mut(right[9:], fr // 100)
            s = left + mid + right
            if "N" in s:
                s = s.replace("N", "A")
            if rnd.random() < 0.5:
                s = "".join(comp[c] for c in reversed(s))
            f.write(">indel_%s_%d_%d_%d\n%s\n" % (names[si], a, d, i, s))


def load_stage_golden(name, read_ids):
    """tests/golden/stage_<name>.json.gz (made by tests/golden/make_stage_golden.py from an instrumented build of the REAL reference):
    returns (fragments {read*2+strand: [(sro, sqo, eqo, refLen), ...]}, DP problems, expected DP results)."""
    import gzip
    import json
    import os
    d = json.load(gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage_%s.json.gz" % name), "rt"))
    idx = {rid: i for i, rid in enumerate(read_ids)}
    frags = {}
    for rid, strand, fl in d["fragments"]:
        frags[idx[rid] * 2 + strand] = [tuple(f) for f in fl]
    mode = {"full": ya.DP_FULL, "banded": ya.DP_BANDED, "ext_fwd": ya.DP_EXT_FWD, "ext_rev": ya.DP_EXT_REV}
    probs, exp = [], []
    for kind, rid, strand, rOff, rLen, qOff, qLen, score, aQ, aR, ops in d["dp"]:
        m = mode[kind]
        if m < ya.DP_EXT_FWD and score == 0:
            continue                                    # (the dump prints no list for a zero score; cannot be compared)
        probs.append(ya.DPProblem(idx[rid], strand, m, qOff, qLen, rLen, rOff))
        ol = tuple((int(o[:-1]), o[-1]) for o in ops.split()) if ops else ()
        exp.append((score, aQ, aR, ol))
    return d, frags, probs, exp


def fasta_ids(path, n):
    out = []
    for line in open(path):
        if line.startswith(">"):
            out.append(line[1:].strip().replace(" ", "_"))
            if len(out) == n:
                break
    return out
