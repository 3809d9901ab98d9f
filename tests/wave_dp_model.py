"""Lane-level model (numpy, 64 'lanes') of the wave-parallel DP used by the HIP kernel
(yaha_amd/csrc/device/dp_wave.h).  It exists to prove on the CPU, against the oracle, that the re-formulation
of findAffineGapScore (reference SW.cpp:798-1208) is exact:

  * lanes = columns of the DP strip (banded: the reference's vertical strip layout, SW.cpp:1001-1012; full:
    reference columns); one row per step;
  * F/I (insertion, vertical) are lane-local with one neighbour exchange;
  * E/D (deletion, horizontal, serial in the reference: SW.cpp:1029-1033) becomes an exclusive max-plus
    prefix scan over lanes of key = ((H_k + GE*k + BIAS) << 6) | (63 - k): the maximum gives the best gap
    origin, the low bits break ties towards the smallest k = the longest run, which is what the reference's
    `CE >= NE -> continue` does.  Valid while the run-length cap (maxIntron) cannot bind, i.e. arrWidth-1 <=
    maxIntron; otherwise the kernel takes its sequential path (not modelled here);
  * row-major first-maximum for X-drop extension = per-row max-reduction of ((V + BIAS) << 6) | (63 - j);
  * boundary cells are never stored: the traceback synthesises them.
"""
import numpy as np

WORST = -(0x7fffff00)
BIAS = 1 << 24
FULL, BANDED, EXT_FWD, EXT_REV = 0, 1, 2, 3


def ref4(bases, off):
    b = int(bases[off >> 1])
    return (b & 0xF) if (off & 1) else (b >> 4)


def supported(P, mode, qLen, rLen):
    bw = P.bandWidth
    if mode in (EXT_FWD, EXT_REV):
        w = 4 * bw + 1
    elif mode == BANDED:
        w = 2 * bw + 1 + abs(rLen - qLen)
    else:
        w = rLen + 1
    return w <= 64 and (w - 1) <= P.maxIntron


def wave_dp(P, mode, q, qOff, qLen, bases, maxROff, rOff, rLen):
    """returns (score, addedQ, addedR, ops[list order]) -- mirrors the wrappers SW.cpp:462-547."""
    GO, GE, RC, MS = P.GOCost, P.GECost, P.RCost, P.MScore
    ext = mode in (EXT_FWD, EXT_REV)
    rev = mode == EXT_REV
    banded = mode != FULL
    if ext:
        if qLen <= 0:
            return 0, 0, 0, ()
        bandwidth = 2 * P.bandWidth
        rLen = qLen + bandwidth
        if rev and rLen > rOff:
            rLen = rOff + 1
            qLen = rLen - bandwidth
            if qLen <= 0:
                return 0, 0, 0, ()
        if (not rev) and rOff + rLen > maxROff:
            rLen = maxROff - rOff
            qLen = rLen - bandwidth
            if qLen <= 0:
                return 0, 0, 0, ()
        left = right = bandwidth
    elif banded:
        bandwidth = P.bandWidth
        if rLen > qLen:
            right, left = bandwidth + (rLen - qLen), bandwidth
        else:
            left, right = bandwidth + (qLen - rLen), bandwidth
    if banded:
        W = left + right + 1
    else:
        W = rLen + 1
        left = 0
    assert W <= 64
    lane = np.arange(64)

    def rbase(idx):  # decompressRef, SW.cpp:444-456
        return ref4(bases, rOff - idx if rev else rOff + idx)

    PV = np.full(64, WORST, dtype=np.int64)
    PF = np.full(64, WORST, dtype=np.int64)
    PI = np.zeros(64, dtype=np.int64)
    if banded:
        for j in range(left, W):
            PV[j] = 0 if j == left else -(GO + (j - left) * GE)
        PF[left] = 0
    else:
        for j in range(0, W):
            PV[j] = 0 if j == 0 else -(GO + j * GE)
        PF[0] = 0
    trace = np.zeros((qLen + 1, 64), dtype=np.int64)
    maxScore, maxi, maxj = WORST, 0, 0
    V = np.zeros(64, dtype=np.int64)
    lastV = 0
    for i in range(1, qLen + 1):
        qc = int(q[qOff + 1 - i]) if rev else int(q[qOff + i - 1])
        if banded:
            sc = left + 1 - i
            hasB = sc > 0
            sc = max(sc, 0)
            ec = min(left + rLen - i, W - 1)
            bl = sc - 1 if hasB else -1
            rc = np.array([rbase(i - left - 1 + j) if sc <= j <= ec else 255 for j in range(64)])
            diag = PV
            upV = np.append(PV[1:], WORST)
            upF = np.append(PF[1:], WORST)
            upI = np.append(PI[1:], 0)
            upV[W - 1], upF[W - 1], upI[W - 1] = WORST, WORST, 0
        else:
            sc, ec, bl = 1, W - 1, 0
            rc = np.array([rbase(j - 1) if 1 <= j <= ec else 255 for j in range(64)])
            diag = np.insert(PV[:-1], 0, WORST)
            upV, upF, upI = PV, PF, PI
        active = (lane >= sc) & (lane <= ec)
        G = diag + np.where(rc == qc, MS, -RC)
        isM = rc == qc
        CF, NF = upF - GE, upV - (GO + GE)
        cont = (CF >= NF) & (upI + 1 <= P.maxGap)
        F = np.where(cont, CF, NF)
        I = np.where(cont, upI + 1, 1)
        H = np.maximum(G, F)
        bval = -(GO + i * GE)
        key = np.where(active, ((H + GE * lane + BIAS) << 6) | (63 - lane), 0)
        if bl >= 0:
            key[bl] = ((bval + GE * bl + BIAS) << 6) | (63 - bl)
        M = np.zeros(64, dtype=np.int64)  # exclusive prefix max
        run = 0
        for j in range(64):
            M[j] = run
            run = max(run, int(key[j]))
        E = np.where(M > 0, (M >> 6) - BIAS - GE * lane - GO, WORST)
        D = lane - (63 - (M & 63))
        V = G.copy()
        op = np.where(isM, 0, 1)  # 0 M, 1 R, 2 D, 3 I
        ln = np.zeros(64, dtype=np.int64)
        takeE = (E >= V) if ext else (E > V)
        V = np.where(takeE, E, V); op = np.where(takeE, 2, op); ln = np.where(takeE, D, ln)
        takeF = (F >= V) if ext else (F > V)
        V = np.where(takeF, F, V); op = np.where(takeF, 3, op); ln = np.where(takeF, I, ln)
        trace[i] = np.where(active, op | (ln << 2), 0)
        if banded:
            PV = np.where(active, V, PV)
            if bl >= 0:
                PV = PV.copy(); PV[bl] = bval
        else:
            PV = np.where(active, V, PV).copy(); PV[0] = bval
        PF = np.where(active, F, PF)
        PI = np.where(active, I, PI)
        if ec >= sc:
            lastV = int(V[ec])
        if ext:
            rk = np.where(active, ((V + BIAS) << 6) | (63 - lane), 0).max()
            if rk > 0:
                rv, rj = (int(rk) >> 6) - BIAS, 63 - (int(rk) & 63)
            else:
                rv, rj = WORST, 0
            if rv > maxScore:
                maxScore, maxi, maxj = rv, i, rj
            if rv < maxScore - P.XCutoff:
                break
    if ext:
        if maxScore <= 0:
            return 0, 0, 0, ()
        y, x = maxi, maxj
        addedQ, addedR = maxi, maxi + (maxj - bandwidth)
        score = maxScore
    else:
        y, x = qLen, (right if banded else W - 1)
        addedQ = addedR = 0
        score = lastV

    def cell(y, x):
        if banded:
            if y == 0:
                return ('U', 0) if x == left else ('D', x - left)
            if x == left - y:
                return ('I', y)
        else:
            if y == 0:
                return ('U', 0) if x == 0 else ('D', x)
            if x == 0:
                return ('I', y)
        t = int(trace[y][x])
        return ("MRDI"[t & 3], t >> 2)

    emitted = []
    code, ln = cell(y, x)
    prev, opLen = code, 0
    while code != 'U':
        if banded:
            if code == 'D': x -= ln
            elif code == 'I': x += ln; y -= ln
            else: y -= 1; ln = 1
        else:
            if code == 'D': x -= ln
            elif code == 'I': y -= ln
            else: x -= 1; y -= 1; ln = 1
        if prev != code:
            pass
        nxt = cell(y, x)
        # the reference compares the op just consumed (code) with prevEOCode, SW.cpp:1182-1189
        if prev != code:
            emitted.append((opLen, prev)); prev = code; opLen = ln
        else:
            opLen += ln
        code, ln = nxt
    emitted.append((opLen, prev))
    ops = tuple(emitted) if rev else tuple(reversed(emitted))
    return score, addedQ, addedR, ops
