#!/usr/bin/env python3
"""Stage-level golden vectors from an INSTRUMENTED build of the real reference.

Runs only in the build container (needs /root/reference).  The reference's sources are copied to a scratch directory under /tmp, print statements are
inserted there (nothing else changes; the patched copy is never committed), the binary runs on the committed golden inputs with -t 1, and what it
prints becomes the fixtures tests/golden/stage_<set>.json.gz:

  fragments : after findFragmentsSort (QueryMatch.c:52-121, called at Query.c:432) -- per read and strand the fragment array
              [startRefOff, startQueryOff, endQueryOff, refLen], in the reference's order;
  dp        : every call of the DP wrappers -- findAGSAlignment / findAGSAlignmentBanded (SW.cpp:462-475) and findAGSExtension<reverse> (SW.cpp:479-533,
              which the careful variants also go through) -- with its arguments as passed (strand, rOff, rLen, qOff, qLen) and its results
              (score, addedQLen, addedRLen, the edit list the call itself produced, head to tail, before any merge).

usage: python tests/golden/make_stage_golden.py      (writes tests/golden/stage_*.json.gz; the SAM of the instrumented run is checked against the goldens)
"""
import gzip, json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, ROOT)


def patch(path, anchor, insert, after=True, count=1):
    s = open(path).read()
    assert s.count(anchor) >= 1, (path, anchor)
    if after:
        s = s.replace(anchor, anchor + insert, count)
    else:
        s = s.replace(anchor, insert + anchor, count)
    open(path, "w").write(s)


def build_instrumented(work):
    src = os.path.join(work, "src")
    shutil.copytree(os.path.join(REF, "src"), src)
    subprocess.check_call(["chmod", "-R", "u+w", src])
    # fragments after findFragmentsSort
    patch(os.path.join(src, "Query.c"), "            int fragCount = findFragmentsSort(AAs, QS, offsetCount);\n",
          '            { fprintf(stderr, "@F %.*s %d %d", QS->queryIDLen, QS->queryID, QS->reversed ? 1 : 0, fragCount);\n'
          '              for (int zz = 0; zz < fragCount; zz++) fprintf(stderr, " %u,%u,%u,%u", (unsigned)QS->fragArray[zz].startRefOff, (unsigned)QS->fragArray[zz].startQueryOff, (unsigned)QS->fragArray[zz].endQueryOff, (unsigned)QS->fragArray[zz].refLen);\n'
          '              fprintf(stderr, "\\n"); }\n')
    # DP wrappers: a helper that prints one call
    helper = ('static void zzDump(QueryState_t * QS, const char * kind, char * qStr, unsigned rOff, unsigned rLen, unsigned qOff, unsigned qLen, int score, unsigned aQ, unsigned aR, EditOpList_t * list)\n'
              '{\n'
              '    fprintf(stderr, "@D %s %.*s %d %u %u %u %u %d %u %u", kind, QS->queryIDLen, QS->queryID, (qStr == QS->reverseCodeBuf) ? 1 : 0, rOff, rLen, qOff, qLen, score, aQ, aR);\n'
              '    if (list != NULL && score != 0) { forAllEditOpsInList(zzop, list) fprintf(stderr, " %u%c", (unsigned)zzop->length, zzop->opcode); }\n'
              '    fprintf(stderr, "\\n");\n'
              '}\n\n')
    sw = os.path.join(src, "SW.cpp")
    patch(sw, "int findAGSAlignment(QueryState_t * QS, ROFF rOff, SUINT rLen, char * qStr, SUINT qOff, SUINT qLen, EditOpList_t * list)\n", helper, after=False)
    patch(sw, "    int retval = findAffineGapScore<FALSE, FALSE, FALSE, FALSE>(QS, qStr+qOff, qLen, QS->refTemp, rLen, list);\n",
          '    zzDump(QS, "full", qStr, rOff, rLen, qOff, qLen, retval, 0, 0, list);\n')
    patch(sw, "    int retval = findAffineGapScore<TRUE, FALSE, FALSE, FALSE>(QS, qStr+qOff, qLen, QS->refTemp, rLen, list);\n",
          '    zzDump(QS, "banded", qStr, rOff, rLen, qOff, qLen, retval, 0, 0, list);\n')
    patch(sw, "    int AGS = findAffineGapScore<TRUE, TRUE, reverse, TRUE>(QS, qStr+qOff, qLen, QS->refTemp, rLen, tempList, addedQLen, addedRLen);\n",
          '    zzDump(QS, reverse ? "ext_rev" : "ext_fwd", qStr, rOff, 0, qOff, qLenArg, AGS > 0 ? AGS : 0, AGS > 0 ? *addedQLen : 0, AGS > 0 ? *addedRLen : 0, AGS > 0 ? tempList : NULL);\n')
    # early returns of findAGSExtension (nothing to extend / clamped to nothing) are calls too: score 0
    patch(sw, "    if (qLen <= 0) return 0;\n    AlignmentArgs_t * AAs = QS->AAs;\n", "", after=True)
    s = open(sw).read()
    s = s.replace("    *addedQLen = 0;\n    *addedRLen = 0;\n    if (qLen <= 0) return 0;\n    AlignmentArgs_t * AAs = QS->AAs;",
                  '    *addedQLen = 0;\n    *addedRLen = 0;\n    if (qLen <= 0) { zzDump(QS, reverse ? "ext_rev" : "ext_fwd", qStr, rOff, 0, qOff, qLenArg, 0, 0, 0, NULL); return 0; }\n    AlignmentArgs_t * AAs = QS->AAs;', 1)
    s = s.replace("        rLen = rOff + 1;\n        qLen = rLen - bandwidth;\n        if (qLen <= 0) return 0;",
                  '        rLen = rOff + 1;\n        qLen = rLen - bandwidth;\n        if (qLen <= 0) { zzDump(QS, "ext_rev", qStr, rOff, 0, qOff, qLenArg, 0, 0, 0, NULL); return 0; }', 1)
    s = s.replace("        rLen = AAs->maxROff - rOff;\n        qLen = rLen - bandwidth;\n        if (qLen <= 0) return 0;",
                  '        rLen = AAs->maxROff - rOff;\n        qLen = rLen - bandwidth;\n        if (qLen <= 0) { zzDump(QS, "ext_fwd", qStr, rOff, 0, qOff, qLenArg, 0, 0, 0, NULL); return 0; }', 1)
    open(sw, "w").write(s)
    objs = []
    for f in sorted(os.listdir(src)):
        if f.endswith(".c") and f != "FragsClumps.c":
            o = os.path.join(work, f + ".o"); subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-w", "-D", "COMPILE_USER_MODE", "-D", "BUILDNUM=83", "-c", os.path.join(src, f), "-o", o]); objs.append(o)
        elif f.endswith(".cpp"):
            o = os.path.join(work, f + ".o"); subprocess.check_call(["g++", "-O2", "-w", "-fpermissive", "-D", "COMPILE_USER_MODE", "-D", "BUILDNUM=83", "-c", os.path.join(src, f), "-o", o]); objs.append(o)
    exe = os.path.join(work, "yaha_instr")
    subprocess.check_call(["g++", "-o", exe] + objs + ["-pthread"])
    return exe


def main():
    import yaha_amd as ya
    work = tempfile.mkdtemp(prefix="yaha_instr_")
    exe = build_instrumented(work)
    for f in ("genome_small.fa", "r1k.fa", "rchim.fa"):
        with gzip.open(os.path.join(GOLD, f + ".gz"), "rb") as g, open(os.path.join(work, f), "wb") as o:
            shutil.copyfileobj(g, o)
    ya.build_index(["-g", os.path.join(work, "genome_small.fa"), "-L", "11", "-cpuindex"])
    idx = os.path.join(work, "genome_small.X11_01_65525S")
    for name, reads, nreads, golden in (("r1k", "r1k.fa", 24, "r1k_default"), ("rchim", "rchim.fa", 24, "rchim_default")):
        # the first `nreads` reads of the set
        sub = os.path.join(work, name + "_sub.fa"); k = 0
        with open(os.path.join(work, reads)) as f, open(sub, "w") as o:
            for line in f:
                if line.startswith(">"):
                    k += 1
                    if k > nreads:
                        break
                o.write(line)
        p = subprocess.run([exe, "-x", idx, "-q", sub, "-osh", os.path.join(work, "o.sam"), "-t", "1"], stderr=subprocess.PIPE, check=True)
        # the instrumented binary still produces the golden SAM (for these reads)
        with gzip.open(os.path.join(GOLD, golden + ".out.gz"), "rb") as g:
            gold = [l for l in g.read().decode().split("\n") if l and not l.startswith("@")]
        mine = [l.rstrip("\n") for l in open(os.path.join(work, "o.sam")) if not l.startswith("@")]
        assert mine == gold[:len(mine)] and len(mine) > 0, "instrumented reference diverges from the golden SAM"
        frags, dps = [], []
        for line in p.stderr.decode().split("\n"):
            if line.startswith("@F "):
                t = line.split(" ")
                frags.append([t[1], int(t[2]), [[int(x) for x in f.split(",")] for f in t[4:]]])
            elif line.startswith("@D "):
                t = line.split(" ")
                dps.append([t[1], t[2], int(t[3])] + [int(x) for x in t[4:11]] + [" ".join(t[11:])])
        out = {"reads": name + ".fa", "n_reads": nreads, "index": "genome_small -L 11", "args": [],
               "fragments_fields": ["read id", "strand", [["startRefOff", "startQueryOff", "endQueryOff", "refLen"]]],
               "dp_fields": ["kind", "read id", "strand", "rOff", "rLen", "qOff", "qLen", "score", "addedQLen", "addedRLen", "ops head..tail"],
               "fragments": frags, "dp": dps}
        with gzip.open(os.path.join(GOLD, "stage_%s.json.gz" % name), "wt", compresslevel=9) as g:
            json.dump(out, g, separators=(",", ":"))
        print(name, "fragment arrays", len(frags), "dp calls", len(dps), "bytes", os.path.getsize(os.path.join(GOLD, "stage_%s.json.gz" % name)))
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
