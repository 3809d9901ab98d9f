// dp_wave.h -- one wavefront computes one banded affine-gap DP (reference findAffineGapScore<banded,extension,
// reverse,XCutoff>, SW.cpp:798-1208, through its wrappers SW.cpp:462-547).  Integer DP: no MFMA.
//
// Layout (fast path): lane j = column j of the reference's DP strip (banded: the vertical strip of SW.cpp:1001-1012,
// W = leftBW + rightBW + 1 <= 64; full: reference columns 0..rLen).  One row per step:
//   * G (match/replace) needs the diagonal predecessor: the lane's own previous V (banded) or the left lane's (full);
//   * F / I (insertion run) need the "up" predecessor: lane j+1 (banded) or the lane itself (full) -> lane-local + 1 shift;
//   * E / D (deletion run) is serial along the row in the reference (SW.cpp:1029-1033).  Here it is an exclusive
//     max-plus prefix scan over lanes of key = ((H_k + GE*k + BIAS) << 6) | (63 - k), H = max(G, F): the maximum picks
//     the best gap origin, the low bits resolve ties towards the smallest k (= longest run = the reference's
//     "CE >= NE -> continue").  Exact while the run cap cannot bind (W - 1 <= maxIntron); otherwise, and for W > 64,
//     the wave runs dpGeneric (the sequential recurrence, wave-uniform, arrays in HBM scratch).
//   * X-drop extension: row-major first maximum = per-row max-reduction of ((V + BIAS) << 6) | (63 - j).
//   * traceback cells (op | runLength << 2, uint16) go to a per-wave HBM/L2 scratch strip, 128 B per row, coalesced;
//     boundary cells are synthesised during traceback.
// tests/wave_dp_model.py is the lane-level model of exactly this algorithm, checked against the oracle on the CPU.
#pragma once
#include "common.h"
// ---- optional in-kernel cycle accounting (diagnostic build only: make PROF=1) ----
#ifdef YD_PROF
__device__ unsigned long long gProf[16];
enum { PF_ROOT = 0, PF_DPROWS, PF_TRACEBACK, PF_PERFECT, PF_SCORE, PF_EMIT, PF_SPLIT, PF_MERGE, PF_DPCALLS, PF_ROOTS };
// per-wave accumulation in LDS (one wavefront per workgroup), flushed once per kernel: the instrumentation must not contend
__shared__ unsigned long long sProf[16];
#define PROF_INIT() do { if (laneId() < 16) sProf[laneId()] = 0; } while (0)
#define PROF_FLUSH() do { if (laneId() < 16 && sProf[laneId()]) atomicAdd(&gProf[laneId()], sProf[laneId()]); } while (0)
#define PROF_T0() unsigned long long pf_t0_ = clock64()
#define PROF_ADD(slot) do { unsigned long long pf_t1_ = clock64(); if (laneId() == 0) sProf[slot] += pf_t1_ - pf_t0_; pf_t0_ = pf_t1_; } while (0)
#define PROF_CNT(slot) do { if (laneId() == 0) sProf[slot] += 1ull; } while (0)
#else
#define PROF_INIT() do { } while (0)
#define PROF_FLUSH() do { } while (0)
#define PROF_T0() do { } while (0)
#define PROF_ADD(slot) do { } while (0)
#define PROF_CNT(slot) do { } while (0)
#endif
#ifdef YD_DEBUG
#define YDBG(...) do { if (laneId() == 0 && blockIdx.x == 0) printf(__VA_ARGS__); } while (0)
#else
#define YDBG(...) do { } while (0)
#endif

struct DPOut { int score, addedQ, addedR, nOps, rows, cells; };   // ops are left in S.tmpOps in EMISSION order (see dpOp)

struct WaveScratch {
    uint16_t *trace;  int traceRows;      // traceRows * 64 cells (HBM/L2)
    uint16_t *ldsTrace;                   // YD_LDS_CELLS cells of LDS: the first rows of every strip live here
    uint32_t *tmpOps; int tmpCap;         // DP result, emission order
    int      *gen;    int genCap;         // generic path rows: 3 * (genCap + 3) ints
    int      *err;                        // wave-local sticky error code (0 = ok)
};
enum { YERR_TRACE = 1, YERR_TMPOPS = 2, YERR_GEN = 3, YERR_ARENA = 4, YERR_DEPTH = 5, YERR_OUT = 6, YERR_CHAIN = 7, YERR_EXEC = 8 };

// k-th op of the last DP result in LIST order (head..tail).  The traceback emits from the alignment's far end towards
// the anchor; the reference adds each op to the front (forward) or to the back (reverse), SW.cpp:1182-1195.
__device__ __forceinline__ uint32_t dpOp(const WaveScratch &S, const DPOut &o, bool reverse, int k)
{ return toGlobal(S.tmpOps)[reverse ? k : (o.nOps - 1 - k)]; }

#define TR_U 0xFFFFu

__device__ __noinline__ DPOut dpGeneric(const DevParams &P, const uint8_t *__restrict__ bases, const uint8_t *__restrict__ q,
                                        bool banded, bool ext, bool rev, uint32_t rOff, int rLen, int qOff, int qLen,
                                        int bandwidth, int left, int right, int W, const WaveScratch &S)
{
    DPOut out = {0, 0, 0, 0, 0, 0};
    PROF_T0();
    const int GO = uni(P.GO), GE = uni(P.GE), RC = uni(P.RC), MS = uni(P.MS), maxIntron = uni(P.maxIntron), maxGapP = uni(P.maxGap), XC = uni(P.X);
    if ((long)(qLen + 1) * W > (long)S.traceRows * 64) { *S.err = YERR_TRACE; return out; }
    if (W > S.genCap) { *S.err = YERR_GEN; return out; }
    int *PV = S.gen + 1, *PF = S.gen + (S.genCap + 3), *PI = S.gen + 2 * (S.genCap + 3);
    uint16_t *T = S.trace;
    PV[-1] = YD_WORST;
    int maxi = 0, maxj = 0;
    if (!ext) { if (banded) { maxi = qLen; maxj = right; } else { maxi = qLen; maxj = W - 1; } }
    int startInit;
    T[0] = TR_U;
    if (banded) { startInit = left + 1; T[left] = TR_U; PF[W] = PV[W] = YD_WORST; PI[W] = 0; } else startInit = 1;
    { int dc = 1; for (int j = startInit; j < W; j++) { T[j] = (uint16_t)(OP_D | (dc << 2)); PV[j] = -(GO + dc * GE); dc++; PF[j] = YD_WORST; PI[j] = 0; } }
    PF[startInit - 1] = 0; PI[startInit - 1] = 0; PV[startInit - 1] = 0;
    { int endInit = banded ? left : qLen; for (int i = 1; i <= endInit && i <= qLen; i++) { int lo = banded ? left - i : 0; T[(long)i * W + lo] = (uint16_t)(OP_I | (i << 2)); } }
    int maxScore = ext ? YD_WORST : 0, V = 0, PVCol = YD_WORST, startCol = 1, endCol = W - 1;
    long rowOffset = 0;
    for (int i = 1; i <= qLen; i++) {
        rowOffset += W;
        int PDCol = 0, PECol = YD_WORST;
        if (banded) {
            startCol = left + 1 - i;
            if (startCol <= 0) { startCol = 0; PVCol = YD_WORST; } else PVCol = PV[startCol - 1] = -(GO + i * GE);
            endCol = min(left + rLen - i, W - 1);
        } else PVCol = -(GO + i * GE);
        int rowMax = YD_WORST;
        int qc = uni((int)q[rev ? qOff + 1 - i : qOff + i - 1]);
        int rRow = banded ? i - left - 1 : 0;
        out.rows++;
        for (int j = startCol; j <= endCol; j++) {
            int RM = banded ? j : j - 1, IO = RM + 1, op;
            V = uni(PV[RM]);
            int ridx = banded ? rRow + j : j - 1;
            int rc = uni((int)ref4(bases, rev ? rOff - (uint32_t)ridx : rOff + (uint32_t)ridx));
            if (qc == rc) { V += MS; op = OP_M; } else { V -= RC; op = OP_R; }
            int len = 0;
            int CE = PECol - GE, NE = PVCol - (GO + GE);
            if (CE >= NE && (PDCol + 1) <= maxIntron) { PECol = CE; PDCol = PDCol + 1; } else { PECol = NE; PDCol = 1; }
            if (ext ? (PECol >= V) : (PECol > V)) { V = PECol; op = OP_D; len = PDCol; }
            int F, I, CF = uni(PF[IO]) - GE, NF = uni(PV[IO]) - (GO + GE); const int pio = uni(PI[IO]);
            if (CF >= NF && (pio + 1) <= maxGapP) { F = CF; I = pio + 1; } else { F = NF; I = 1; }
            if (ext ? (F >= V) : (F > V)) { V = F; op = OP_I; len = I; }
            PF[j] = F; PI[j] = I;
            T[rowOffset + j] = (uint16_t)(op | (len << 2));
            if (ext && V > rowMax) rowMax = V;
            if (ext && V > maxScore) { maxScore = V; maxi = i; maxj = j; }
            if (banded) PV[j] = V; else PV[j - 1] = PVCol;
            PVCol = V; out.cells++;
        }
        if (ext && rowMax < (maxScore - XC)) break;
        if (!banded) PV[endCol] = V;
    }
    int retval = ext ? maxScore : V;
    if (ext && retval <= 0) return out;
    out.score = retval;
    if (ext) { out.addedQ = maxi; out.addedR = maxi + (maxj - bandwidth); }
    int x = maxj; long rowBase = (long)maxi * W;
    unsigned cell = uniU((unsigned)T[rowBase + x]);
    int prev = cell == TR_U ? -1 : (cell & 3), opLenAcc = 0, n = 0;
    for (int guard = 0; cell != TR_U; guard++) {
        if (guard > 200000) { *S.err = YERR_TRACE; out.score = 0; return out; }
        int code = cell & 3, len = cell >> 2;
        if (banded) { if (code == OP_D) x -= len; else if (code == OP_I) { x += len; rowBase -= (long)len * W; } else { rowBase -= W; len = 1; } }
        else        { if (code == OP_D) x -= len; else if (code == OP_I) { rowBase -= (long)len * W; } else { x -= 1; rowBase -= W; len = 1; } }
        if (prev != code) { if (n < S.tmpCap) S.tmpOps[n] = opMake(prev, opLenAcc); n++; prev = code; opLenAcc = len; } else opLenAcc += len;
        cell = uniU((unsigned)T[rowBase + x]);
    }
    if (n < S.tmpCap) S.tmpOps[n] = opMake(prev, opLenAcc); n++;
    if (n > S.tmpCap) { *S.err = YERR_TMPOPS; out.score = 0; n = 0; }
    out.nOps = n;
    PROF_ADD(PF_TRACEBACK);
    return out;
}

// mode: YGPU_DP_FULL / BANDED / EXT_FWD / EXT_REV.  q = the read's strand codes; offsets as the reference wrappers take them.
__device__ __noinline__ DPOut dpWave(const DevParams &P, const uint8_t *__restrict__ bases, const uint8_t *__restrict__ q, int mode,
                                     uint32_t rOff, int rLenArg, int qOff, int qLenArg, const WaveScratch &S)
{
    const int lane = laneId();
    const int GO = uni(P.GO), GE = uni(P.GE), RC = uni(P.RC), MS = uni(P.MS), maxGap = uni(P.maxGap), XC = uni(P.X), bwP = uni(P.bandWidth), maxIntron = uni(P.maxIntron);
    const uint32_t maxROff = uniU(P.maxROff);
    mode = uni(mode); rOff = uniU(rOff); qOff = uni(qOff);
    const bool ext = mode >= YGPU_DP_EXT_FWD, rev = mode == YGPU_DP_EXT_REV, banded = mode != YGPU_DP_FULL;
    DPOut out = {0, 0, 0, 0, 0, 0};
    int qLen = uni(qLenArg), rLen = uni(rLenArg), bandwidth = 0, left = 0, right = 0;
    if (ext) {                                                              // findAGSExtension, SW.cpp:479-516
        if (qLen <= 0) return out;
        bandwidth = 2 * bwP;
        uint32_t rl = (uint32_t)(qLen + bandwidth);
        if (rev && rl > rOff) { rl = rOff + 1; qLen = (int)rl - bandwidth; if (qLen <= 0) return out; }
        if (!rev && (rOff + rl) > maxROff) { rl = maxROff - rOff; qLen = (int)rl - bandwidth; if (qLen <= 0) return out; }
        qLen &= 0xFFFF; rLen = (int)(rl & 0xFFFF);                           // SUINT parameters of findAffineGapScore
        left = right = bandwidth;
    } else if (banded) {                                                    // SW.cpp:853-871
        bandwidth = bwP;
        if (rLen > qLen) { right = bandwidth + (rLen - qLen); left = bandwidth; } else { left = bandwidth + (qLen - rLen); right = bandwidth; }
    }
    qLen = uni(qLen); rLen = uni(rLen); left = uni(left); right = uni(right);
    const int W = banded ? left + right + 1 : rLen + 1;
    YDBG("dpWave mode %d qLen %d rLen %d W %d left %d right %d\n", mode, qLen, rLen, W, left, right);
    if (W > 64 || (W - 1) > maxIntron) return dpGeneric(P, bases, q, banded, ext, rev, rOff, rLen, qOff, qLen, bandwidth, left, right, W, S);
    if (qLen + 1 > S.traceRows) { *S.err = YERR_TRACE; return out; }

    YD_GLOBAL const uint8_t *gBases = toGlobal(bases), *gQ = toGlobal(q);
    YD_GLOBAL uint16_t *trace = toGlobal(S.trace);
    YD_LDS uint16_t *lds = (YD_LDS uint16_t *)S.ldsTrace;
    const int Wp = W <= 32 ? 32 : 64, ldsRows = YD_LDS_CELLS / Wp;      // rows [0, ldsRows) of the strip are kept in LDS
    auto loadRef = [&](int idx) -> int {
        int v = 0xFF;
        if (idx >= 0 && idx < rLen) { uint32_t off = rev ? rOff - (uint32_t)idx : rOff + (uint32_t)idx; uint32_t b = gBases[off >> 1];
            v = (int)((off & 1u) ? (b & 0xFu) : (b >> 4)); }
        asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(v));             // consume the load here, so that no wait lands at the row-loop header
        return v; };
    auto loadQ   = [&](int t) -> int { int v = (t < qLen) ? (int)gQ[rev ? qOff - t : qOff + t] : 0xFE; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(v)); return v; };

    int PV, PF, PI = 0, rc;
    if (banded) { PV = lane == left ? 0 : (lane > left ? -(GO + (lane - left) * GE) : YD_WORST); PF = lane == left ? 0 : YD_WORST; rc = loadRef(lane - left); }
    else        { PV = lane == 0 ? 0 : -(GO + lane * GE); PF = lane == 0 ? 0 : YD_WORST; rc = loadRef(lane - 1); }
    int rbuf = loadRef(lane), rcb = 0;          // stream of reference bases for the top lane of the strip
    int qbuf = loadQ(lane), qcb = 0;            // stream of query bases, one per row
    int maxScore = YD_WORST, maxi = 0, maxj = 0, lastEc = 0;
    int rows = 0, cells = 0;
    PROF_T0(); PROF_CNT(PF_DPCALLS);
    for (int i = 1; i <= qLen; ++i) {
        int sc, ec, bl;
        if (banded) { sc = left + 1 - i; bl = sc > 0 ? sc - 1 : -1; if (sc < 0) sc = 0; ec = min(left + rLen - i, W - 1); }
        else { sc = 1; ec = W - 1; bl = 0; }
        const int t = i - 1;
        if (t >= qcb + 64) { qcb += 64; qbuf = loadQ(qcb + lane); }
        const int qc = bcast(qbuf, t - qcb);
        const bool active = lane >= sc && lane <= ec;
        int diag, upV, upF, upI;
        if (banded) {
            diag = PV; upV = laneDown1(PV, YD_WORST); upF = laneDown1(PF, YD_WORST); upI = laneDown1(PI, 0);
            if (lane >= W - 1) { upV = YD_WORST; upF = YD_WORST; upI = 0; }
        } else { diag = laneUp1(PV, YD_WORST); upV = PV; upF = PF; upI = PI; }
        const bool isM = (rc == qc);
        const int G = diag + (isM ? MS : -RC);
        const int CF = upF - GE, NF = upV - (GO + GE);
        const bool cont = (CF >= NF) && (upI + 1 <= maxGap);
        const int F = cont ? CF : NF, I = cont ? upI + 1 : 1;
        const int H = G > F ? G : F;
        const int bval = -(GO + i * GE);
        unsigned key = active ? (((unsigned)(H + GE * lane + YD_BIAS) << 6) | (unsigned)(63 - lane)) : 0u;
        if (lane == bl) key = ((unsigned)(bval + GE * lane + YD_BIAS) << 6) | (unsigned)(63 - lane);
        const unsigned M = waveExclMaxU(key, lane);
        int E = YD_WORST, D = 0;
        if (M) { E = (int)(M >> 6) - YD_BIAS - GE * lane - GO; D = lane - (63 - (int)(M & 63u)); }
        int V = G, op = isM ? OP_M : OP_R, len = 0;
        if (ext ? (E >= V) : (E > V)) { V = E; op = OP_D; len = D; }
        if (ext ? (F >= V) : (F > V)) { V = F; op = OP_I; len = I; }
        if (active) { PV = V; PF = F; PI = I; }
        if (i < ldsRows) { if (active) lds[i * Wp + lane] = (uint16_t)(op | (len << 2)); }
        else if (active) trace[i * 64 + lane] = (uint16_t)(op | (len << 2));
        if (lane == bl) PV = bval;
        rows++; cells += (ec >= sc) ? (ec - sc + 1) : 0; if (ec >= sc) lastEc = ec;
        if (ext) {
            const unsigned rk = waveTotalMaxU(active ? (((unsigned)(V + YD_BIAS) << 6) | (unsigned)(63 - lane)) : 0u);
            int rv = YD_WORST, rj = 0;
            if (rk) { rv = (int)(rk >> 6) - YD_BIAS; rj = 63 - (int)(rk & 63u); }
            rv = uni(rv); rj = uni(rj);
            if (rv > maxScore) { maxScore = rv; maxi = i; maxj = rj; }
            if (rv < maxScore - XC) break;
        }
        if (banded) {                                                       // slide the reference window by one base
            const int ni = i + right;                                      // index the top lane needs for row i+1
            if (ni >= rcb + 64) { rcb += 64; rbuf = loadRef(rcb + lane); }
            const int nb = bcast(rbuf, ni - rcb);
            rc = laneDown1(rc, 0xFF);
            if (lane == W - 1) rc = nb;
        }
    }
    out.rows = rows; out.cells = cells;
    PROF_ADD(PF_DPROWS);
    YDBG("rows done %d cells %d maxScore %d maxi %d maxj %d\n", rows, cells, maxScore, maxi, maxj);
    int y, x, score;
    if (ext) { if (maxScore <= 0) return out; y = maxi; x = maxj; score = maxScore; out.addedQ = maxi; out.addedR = maxi + (maxj - bandwidth); }
    else { y = qLen; x = banded ? right : W - 1; score = bcast(PV, uni(lastEc)); }
    out.score = score;
    __threadfence_block();                                                  // other lanes' trace cells become visible to every lane
    // ---- traceback (wave-uniform), SW.cpp:1138-1195; boundary cells are synthesised ----
    auto cellAt = [&](int yy, int xx, int &code, int &len) {
        if (banded) {
            if (yy == 0) { if (xx == left) { code = -1; len = 0; } else { code = OP_D; len = xx - left; } return; }
            if (xx == left - yy) { code = OP_I; len = yy; return; }
        } else {
            if (yy == 0) { if (xx == 0) { code = -1; len = 0; } else { code = OP_D; len = xx; } return; }
            if (xx == 0) { code = OP_I; len = yy; return; }
        }
        const unsigned c = uniU(yy < ldsRows ? (unsigned)lds[yy * Wp + xx] : (unsigned)trace[yy * 64 + xx]); code = (int)(c & 3u); len = (int)(c >> 2);
    };
    YD_GLOBAL uint32_t *gTmp = toGlobal(S.tmpOps);
    int code, len; cellAt(y, x, code, len);
    int prev = code, acc = 0, n = 0;
    // The path is mostly straight runs of M/R cells (same strip column in banded mode, the diagonal in full mode).  64 cells of
    // the run are examined per step: lane l looks at the cell l rows up; a ballot finds where the run ends and which cells
    // are replacements; the run-length encoding of that bit string is scalar work.  Only D/I and boundary cells take the
    // one-cell path below.  Same emission rule as the reference loop (SW.cpp:1154-1195).
    for (int guard = 0; code >= 0; guard++) {
        if (guard > 200000 || y < 0 || x < 0 || x > 63) { *S.err = YERR_TRACE; out.score = 0; out.nOps = 0; return out; }
        if (code <= OP_R) {
            const int yy = y - lane, xx = banded ? x : x - lane;
            bool valid = yy >= 1 && (banded ? (xx != left - yy) : (xx >= 1));
            unsigned c = 3u;
            if (valid) c = yy < ldsRows ? (unsigned)lds[yy * Wp + xx] : (unsigned)trace[yy * 64 + xx];
            const bool isMR = valid && (c & 3u) < 2u;
            const unsigned long long stop = __ballot(!isMR), rb = __ballot(isMR && (c & 3u) == (unsigned)OP_R);
            const int run = stop ? __builtin_ctzll(stop) : 64;          // >= 1: the current cell is M or R
            for (int pos = 0; pos < run; ) {
                const int bit = (int)((rb >> pos) & 1ull);
                const unsigned long long rest = (bit ? ~rb : rb) >> pos;
                int seg = rest ? __builtin_ctzll(rest) : 64; if (seg > run - pos) seg = run - pos;
                const int cd = bit ? OP_R : OP_M;
                if (prev != cd) { if (n < S.tmpCap) gTmp[n] = opMake(prev, acc); n++; prev = cd; acc = seg; } else acc += seg;
                pos += seg;
            }
            y -= run; if (!banded) x -= run;
            cellAt(y, x, code, len);
            continue;
        }
        if (code == OP_D) x -= len; else y -= len, x += banded ? len : 0;  // D: back along the row; I: up `len` rows
        if (prev != code) { if (n < S.tmpCap) gTmp[n] = opMake(prev, acc); n++; prev = code; acc = len; } else acc += len;
        cellAt(y, x, code, len);
    }
    if (n < S.tmpCap) gTmp[n] = opMake(prev, acc); n++;
    if (n > S.tmpCap) { *S.err = YERR_TMPOPS; out.score = 0; n = 0; }
    out.nOps = n;
    PROF_ADD(PF_TRACEBACK);
    return out;
}
