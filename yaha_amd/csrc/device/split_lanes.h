// split_lanes.h -- splitClump with ONE ROOT PER LANE (reference AlignHelpers.c:302-579, AlignExtFrag.cpp:64-156, SW.cpp:553-788).
//
// The roots that scoreClump sends to splitClump (about 4 % of all roots) used to run one wavefront each; that path is bound by
// the latency of its serial steps.  Here a lane runs the same state machine (finishRoot in align.h) as plain sequential code:
// frame stack and edit-list buffers in lane-private HBM, every careful extension taken from the results that the lane kernels
// computed beforehand (predictCarefulDPs in phase_lanes.h -> second k_ext_rows / k_ext_trace round).  Clumps are buffered and
// published at the end; a root whose run asks for a DP that is not in its list, or overflows a lane buffer, publishes nothing
// and goes to the wave kernel (k_align_p3) instead.
#pragma once
#include "phase_lanes.h"

#define YD_SL_DEPTH 8                              // frames per lane
#define YD_SL_CAP 1024                             // ops per frame buffer
#define YD_SL_FRONT 320                            // head-room for ops merged to the front
#define YD_SL_OUT 1536                             // buffered output ops per lane
#define YD_SL_CLUMPS 8                             // buffered output clumps per lane
#define YD_SL_BYTES ((YD_SL_DEPTH * YD_SL_CAP + YD_SL_OUT) * 4 + YD_SL_CLUMPS * 32)

struct SplitArgs {
    uint8_t *scratch;                              // YD_SL_BYTES per thread
    const uint32_t *memoKeys; const unsigned int *memoCount; const ExtRes *res2; const uint32_t *ops2; uint32_t nProb2;
    uint32_t *fallList; unsigned int *fallCount;   // roots left to k_align_p3
    uint32_t nSlots;
};

__device__ unsigned int gFallWhy[8];                // diagnostics (YGPU_TRACE): why a root was left to the wave kernel
struct SFrame { uint32_t sro; int sqo, eqo, refLen, score, status, start, len, phase, minItem, maxItem, sQO, eQO, maxAGS, cSqo, cEqo, cRefLen; uint32_t sRO, eRO, cSro; };

__global__ void __launch_bounds__(64) k_split_lanes(AlignArgs A, PhaseArgs X, SplitArgs Sx)
{
    YD_HIGH_PRIO();
    const int lane = laneId(); const uint32_t slot = blockIdx.x * 64u + (uint32_t)lane; const DevParams &P = A.P;
    const bool live = slot < Sx.nSlots;
    uint32_t *lists = (uint32_t *)(Sx.scratch + (size_t)slot * YD_SL_BYTES); uint32_t *outOps = lists + YD_SL_DEPTH * YD_SL_CAP;
        ygpu_clump *outCl = (ygpu_clump *)(outOps + YD_SL_OUT);
    bool fall = false; int why = 0; int nCl = 0, nOutOps = 0; uint32_t r = 0;
    unsigned splits = 0, extCalls = 0, extRows = 0, extCells = 0, perfect = 0, touched = 0;
    if (live) {
        r = X.slowList[slot];
        const ChainClumpRec rec = YD_ROOT_REC(A, r);
        const uint32_t r0 = A.B.readOff[rec.rs >> 1]; const int qlen = (int)(A.B.readOff[(rec.rs >> 1) + 1] - r0);
        YD_GLOBAL const uint8_t *q = toGlobal((rec.rs & 1u) ? A.B.rev : A.B.fwd) + r0; YD_GLOBAL const uint8_t *gB = toGlobal(A.bases);
        // ---- the root as k_p3_lanes saw it: phase-1 list + the two extension results --------------------------------------------
        SFrame f; { const RootState S = X.state[r]; f.sro = S.sro; f.sqo = S.sqo; f.eqo = S.eqo; f.refLen = S.refLen; f.status = S.status; f.score = S.score; f.len = S.len;
            f.start = 0; f.phase = 0; }
        MergedOps L; L.a = L.c = nullptr; L.na = L.nc = L.jab = L.jbc = 0; L.b = X.stateOps + X.state[r].listOff; L.nb = f.len;
        const ExtRes rb = X.res[2 * (size_t)r], rf = X.res[2 * (size_t)r + 1];
        if (rb.score > 0) {
            const int aQ = rb.maxi, aR = rb.maxi + (rb.maxj - YD_LBAND);
            L.a = extOpsPtr(X.extOps, rb); L.na = (int)rb.nOps;
            f.score += rb.score; f.sqo = (f.sqo - aQ) & 0xFFFF; f.sro -= (uint32_t)aR; f.refLen = (f.refLen + aR) & 0xFFFF;
        }
        if (rf.score > 0) {
            const int aQ = rf.maxi, aR = rf.maxi + (rf.maxj - YD_LBAND);
            L.c = extOpsPtr(X.extOps, rf); L.nc = (int)rf.nOps;
            f.score += rf.score; f.eqo = (f.eqo + aQ) & 0xFFFF; f.refLen = (f.refLen + aR) & 0xFFFF;
        }
        f.status |= stAligned;
        L.setJunctions();
        const int n0 = L.count();
        if (YD_SL_FRONT + n0 > YD_SL_CAP - 64) fall = true;
        else { for (int k = 0; k < n0; k++) lists[YD_SL_FRONT + k] = L.atLean(k); f.start = YD_SL_FRONT; f.len = n0; }
        // ---- helpers on a frame's list buffer ---------------------------------------------------------------------------------------
        int depth = 0; SFrame st[YD_SL_DEPTH]; unsigned pushes = 0;
        int sm = -1, smm = 0, sg = 0, sl = 0, ss = 0;                          // results of the last scoreList
        auto buf = [&](int d) { return lists + d * YD_SL_CAP; };
        auto scoreList = [&](SFrame &fr, const uint32_t *b) -> int {        // scoreClump, AlignHelpers.c:302-366 (scoreListImpl in align.h)
            int AGS = 0, maxAGS = 0, matches = 0, mism = 0, ins = 0, del = 0; const int n = fr.len, aligned = fr.score;
            for (int k = 0; k < n; k++) {
                const uint32_t op = b[fr.start + k]; const int code = opCode(op), len = opLen(op);
                YD_OP_COUNT(code, len, matches, mism, ins, del);
                AGS += opScore(P, code, len);
                if (AGS <= 0 || (AGS >= aligned && k != n - 1)) return 1;
                if (AGS > maxAGS) maxAGS = AGS;
            }
            if (matches >= P.minRawScore && maxAGS > AGS) return 1;
            sm = -1;
            if (matches < P.minRawScore) return 0;
            const int tot = (matches + mism + ins + del) & 0xFFFF; matches &= 0xFFFF;
            sm = matches; smm = mism & 0xFFFF; sg = (ins + del) & 0xFFFF; sl = tot; ss = AGS & 0xFFFF;
            if ((double)matches / (double)tot < (double)P.minIdentity) return 0;
            fr.status |= stScored;
            return 0;
        };
        auto hasMaxMatch = [&](const uint32_t *b, int start, int n) { for (int k = 0; k < n; k++) { const uint32_t op = b[start + k];
            if (opCode(op) == OP_M && opLen(op) >= P.wordLen) return true; } return false; };
        auto emit = [&](const SFrame &fr, const uint32_t *b) {
            if (nCl >= YD_SL_CLUMPS || nOutOps + fr.len > YD_SL_OUT) { fall = true; return; }
            for (int k = 0; k < fr.len; k++) outOps[nOutOps + k] = b[fr.start + k];
            ygpu_clump c; c.sro = fr.sro; c.sqo = (uint16_t)fr.sqo; c.eqo = (uint16_t)fr.eqo; c.refLen = (uint16_t)fr.refLen; c.totScore = (uint16_t)ss; c.totLength = (uint16_t)sl;
            c.matchedBases = (uint16_t)sm; c.mismatchedBases = (uint16_t)smm; c.gapBases = (uint16_t)sg; c.status = (uint8_t)fr.status; c.reserved = 0;
                c.op_start = (uint32_t)nOutOps; c.n_ops = (uint32_t)fr.len;
            outCl[nCl++] = c; nOutOps += fr.len; pushes++;
        };
        // a careful extension = the stored result of exactly this X-drop problem (runDP in align.h); ops in LIST order through opAt
        struct DPRes { int score, addedQ, addedR, nOps; const uint32_t *arr; bool rev; };
        auto lookup = [&](bool rev, uint32_t rOff, int qOff, int qLen, DPRes &o) -> bool {
            const unsigned n = min(Sx.memoCount[slot], (unsigned)YD_MEMO);
            for (unsigned k = 0; k < n; k++) {
                const uint32_t *key = Sx.memoKeys + ((size_t)slot * YD_MEMO + k) * 3;
                if (key[0] == rOff && key[1] == (((uint32_t)qOff & 0xFFFFu) | ((uint32_t)qLen << 16)) && (key[2] & 1u) == (rev ? 1u : 0u)) {
                    const uint32_t idx = key[2] >> 1; if (idx >= Sx.nProb2) return false;
                    const ExtRes e = Sx.res2[idx];
                    extCalls++; extRows += e.rows; extCells += e.cells; touched += e.rows + 4 * (unsigned)P.bandWidth + 1u;
                    o.score = e.score > 0 ? e.score : 0; o.addedQ = o.addedR = o.nOps = 0; o.arr = nullptr; o.rev = rev;
                    if (o.score > 0) { o.addedQ = e.maxi; o.addedR = e.maxi + (e.maxj - YD_LBAND); o.nOps = (int)e.nOps; o.arr = extOpsPtr(Sx.ops2, e); }
                    return true;
                }
            }
            return false;
        };
        auto opAt = [&](const DPRes &o, int k) { return o.rev ? o.arr[o.nOps - 1 - k] : o.arr[k]; };      // the backward list is stored reversed (k_ext_trace)
        auto mergeBack = [&](uint32_t *b, int start, int &len, const DPRes &o, int t0, int t1) {           // mergeEOLToBack, SW.cpp:207-261
            if (t1 <= t0) return;
            if (len > 0) { const uint32_t last = b[start + len - 1], first = opAt(o, t0);
                if (opCode(last) == opCode(first)) { b[start + len - 1] = opMake(opCode(last), (opLen(last) + opLen(first)) & 0xFFFF); t0++; } }
            const int cnt = t1 - t0;
            if (start + len + cnt > YD_SL_CAP) { fall = true; return; }
            for (int k = 0; k < cnt; k++) b[start + len + k] = opAt(o, t0 + k);
            len += cnt;
        };
        auto mergeFront = [&](uint32_t *b, int &start, int &len, const DPRes &o, int t0, int t1) {         // mergeEOLToFront, SW.cpp:151-205
            if (t1 <= t0) return;
            if (len > 0) { const uint32_t first = b[start], last = opAt(o, t1 - 1);
                if (opCode(first) == opCode(last)) { b[start] = opMake(opCode(first), (opLen(first) + opLen(last)) & 0xFFFF); t1--; } }
            const int cnt = t1 - t0;
            if (start - cnt < 0) { fall = true; return; }
            start -= cnt;
            for (int k = 0; k < cnt; k++) b[start + k] = opAt(o, t0 + k);
            len += cnt;
        };
        auto opStep = [&](uint32_t op, int &QLen, int &RLen, int &AGS) {
            const int code = opCode(op), len = opLen(op);
            QLen += code != OP_D ? len : 0; RLen += code != OP_I ? len : 0; AGS += opScore(P, code, len);
        };
        // extendClump<goBack, goForw, carefully> (AlignExtFrag.cpp:64-156) with the careful variants of SW.cpp:553-788
        auto extendCarefully = [&](SFrame &fr, uint32_t *b, bool goBack, bool goForw) {
            int score = fr.score, backLen = 0, forwLen = 0;
            if (goBack) {
                backLen = (int)((uint32_t)fr.sqo < fr.sro ? (uint32_t)fr.sqo : fr.sro);
                if (backLen > 0) {
                    const int m = matchRun<-1>(q, fr.sqo - 1, gB, fr.sro - 1u, backLen);
                    perfect += m; touched += m + (m < backLen);
                    if (m > 0) { b[fr.start] = opMake(opCode(b[fr.start]), (opLen(b[fr.start]) + m) & 0xFFFF); score += m * P.MS; backLen -= m; fr.sqo -= m; fr.sro -= (uint32_t)m;
                        fr.refLen = (fr.refLen + m) & 0xFFFF; }
                }
            }
            if (goForw) {
                const uint32_t eRO = fr.sro + (uint32_t)fr.refLen - 1u;
                const uint32_t qrem = (uint32_t)(((qlen - 1) - fr.eqo) & 0xFFFF), rrem = P.maxROff - eRO;
                forwLen = (int)(qrem < rrem ? qrem : rrem);
                if (forwLen > 0) {
                    const int m = matchRun<1>(q, fr.eqo + 1, gB, eRO + 1u, forwLen);
                    perfect += m; touched += m + (m < forwLen);
                    if (m > 0) { const int li = fr.start + fr.len - 1; b[li] = opMake(opCode(b[li]), (opLen(b[li]) + m) & 0xFFFF); score += m * P.MS; forwLen -= m; fr.eqo += m;
                        fr.refLen = (fr.refLen + m) & 0xFFFF; }
                }
            }
            if (goBack && backLen >= P.minExtLength) {                          // findAGSBackwardExtensionCarefully, SW.cpp:671-788
                DPRes o; if (!lookup(true, fr.sro - 1u, (fr.sqo - 1) & 0xFFFF, backLen & 0xFFFF, o)) { fall = true; why = 1; return; }
                int ns = 0, aQ = 0, aR = 0;
                if (o.score > 0) {
                    int QLen = 0, RLen = 0, AGS = 0, maxAGS = 0, startItem = -1;
                    for (int k = 0; k < o.nOps; k++) {
                        opStep(opAt(o, k), QLen, RLen, AGS);
                        if (AGS <= 0) { AGS = 0; maxAGS = 0; QLen = 0; RLen = 0; startItem = k; }
                        if (AGS > maxAGS) maxAGS = AGS;
                    }
                    if (!(AGS <= 0 || maxAGS >= AGS + score)) { mergeFront(b, fr.start, fr.len, o, startItem + 1, o.nOps); aQ = QLen; aR = RLen; ns = AGS; }
                }
                if (ns > 0) { score += ns; fr.sqo = (fr.sqo - aQ) & 0xFFFF; fr.sro -= (uint32_t)aR; fr.refLen = (fr.refLen + aR) & 0xFFFF; }
            }
            if (fall) return;
            if (goForw && forwLen >= P.minExtLength) {                          // findAGSForwardExtensionCarefully, SW.cpp:553-669
                DPRes o; if (!lookup(false, fr.sro + (uint32_t)fr.refLen, (fr.eqo + 1) & 0xFFFF, forwLen & 0xFFFF, o)) { fall = true; why = 1; return; }
                int ns = 0, aQ = o.addedQ, aR = o.addedR;
                if (o.score > 0) {
                    int initAGS = o.score, QLen = 0, RLen = 0, AGS = score, maxAGS = score, maxItem = -1, maxQLen = 0, maxRLen = 0, nT = o.nOps; bool none = false;
                    for (int k = 0; k < o.nOps; k++) {
                        opStep(opAt(o, k), QLen, RLen, AGS);
                        if (AGS > maxAGS) { maxAGS = AGS; maxQLen = QLen; maxRLen = RLen; maxItem = k; }
                        else if (AGS <= 0) { if (maxAGS <= score) { none = true; break; } nT = maxItem + 1; aQ = maxQLen; aR = maxRLen; initAGS = maxAGS - score; break; }
                    }
                    if (!none) { mergeBack(b, fr.start, fr.len, o, 0, nT); ns = initAGS; }
                }
                if (ns > 0) { score += ns; fr.eqo = (fr.eqo + aQ) & 0xFFFF; fr.refLen = (fr.refLen + aR) & 0xFFFF; }
            }
            fr.score = score;
        };
        // ---- the state machine of finishRoot (align.h) ---------------------------------------------------------------------------------
        enum { ST_SCORE, ST_SPLIT_ENTER, ST_SPLIT_TAIL, ST_SPLIT_CORE, ST_RETURN } state = ST_SCORE;
        int wS = 0, wE = 0;                                                   // the root's span when it was sent to splitClump (inherited by every frame)
        for (int guard = 0; !fall; guard++) {
            if (guard > 4000) { fall = true; break; }
            uint32_t *b = buf(depth);
            if (state == ST_SCORE) {
                if (f.status & stScored) { state = ST_RETURN; continue; }
                const int rr = scoreList(f, b);
                if (rr == 1) {
                    if (depth != 0 || splits != 0) { fall = true; why = 2; break; }      // a second split of the same root: rare, left to the wave kernel
                    splits++; wS = f.sqo; wE = f.eqo; state = ST_SPLIT_ENTER;
                } else state = ST_RETURN;
                continue;
            }
            if (state == ST_SPLIT_ENTER) {                                    // splitClumpHelper, AlignHelpers.c:374-557
                int matches = 0, mism = 0, ins = 0, del = 0, AGS = 0, maxAGS = -10000, maxItem = -1, minItem = -1, eQO = 0, sQO = 0; uint32_t eRO = 0, sRO = 0; const int n = f.len;
                for (int k = 0; k < n; k++) {
                    const uint32_t op = b[f.start + k]; const int code = opCode(op), len = opLen(op);
                    YD_OP_COUNT(code, len, matches, mism, ins, del);
                    AGS += opScore(P, code, len); if (AGS < 0) AGS = 0;
                    if (AGS > maxAGS) { maxAGS = AGS; maxItem = k; eQO = (f.sqo + matches + mism + ins - 1) & 0xFFFF; eRO = f.sro + (uint32_t)(matches + mism + del) - 1u; }
                }
                AGS = maxAGS; matches = mism = ins = del = 0; int maxMatch = 0;
                for (int k = maxItem; k >= 0; k--) {
                    const uint32_t op = b[f.start + k]; const int code = opCode(op), len = opLen(op);
                    YD_OP_COUNT(code, len, matches, mism, ins, del);
                    AGS -= opScore(P, code, len); maxMatch = (code == OP_M && len > maxMatch) ? len : maxMatch;
                    if (AGS <= 0) { minItem = k; sQO = (eQO - (matches + mism + ins - 1)) & 0xFFFF; sRO = eRO - (uint32_t)(matches + mism + del - 1); break; }
                }
                if (maxMatch < P.wordLen || minItem < 0) { state = ST_RETURN; continue; }
                f.minItem = minItem; f.maxItem = maxItem; f.sQO = sQO; f.eQO = eQO; f.sRO = sRO; f.eRO = eRO; f.maxAGS = maxAGS;
                f.cSqo = f.sqo; f.cEqo = f.eqo; f.cSro = f.sro; f.cRefLen = f.refLen;
                if (minItem != 0 && hasMaxMatch(b, f.start, minItem)) {        // head remainder :463-495
                    if (depth + 1 >= YD_SL_DEPTH || YD_SL_FRONT + minItem > YD_SL_CAP - 64) { fall = true; why = 3; break; }
                    uint32_t *cb = buf(depth + 1);
                    for (int k = 0; k < minItem; k++) cb[YD_SL_FRONT + k] = b[f.start + k];
                    SFrame c = f; c.status = f.status & stReversed; c.sqo = f.cSqo; c.eqo = (sQO - 1) & 0xFFFF; c.sro = f.cSro;
                        c.refLen = (int)((1u + (sRO - 1u) - f.cSro) & 0xFFFFu);
                    c.score = 0; c.start = YD_SL_FRONT; c.len = minItem; c.phase = PH_NONE;
                    f.phase = PH_AFTER_HEAD; st[depth] = f; depth++; f = c; state = ST_SPLIT_ENTER; continue;
                }
                state = ST_SPLIT_TAIL; continue;
            }
            if (state == ST_SPLIT_TAIL) {                                     // tail remainder :500-531
                const int n = f.len;
                if (f.maxItem != n - 1) {
                    const int t0 = f.maxItem + 1, tl = n - t0;
                    if (hasMaxMatch(b, f.start + t0, tl)) {
                        if (depth + 1 >= YD_SL_DEPTH || YD_SL_FRONT + tl > YD_SL_CAP - 64) { fall = true; why = 3; break; }
                        uint32_t *cb = buf(depth + 1);
                        for (int k = 0; k < tl; k++) cb[YD_SL_FRONT + k] = b[f.start + t0 + k];
                        SFrame c = f; c.status = f.status & stReversed; c.sqo = (f.eQO + 1) & 0xFFFF; c.eqo = f.cEqo; c.sro = f.eRO + 1u;
                        c.refLen = (int)((1u + (f.cSro + (uint32_t)f.cRefLen - 1u) - (f.eRO + 1u)) & 0xFFFFu);
                        c.score = 0; c.start = YD_SL_FRONT; c.len = tl; c.phase = PH_NONE;
                        f.phase = PH_AFTER_TAIL; st[depth] = f; depth++; f = c; state = ST_SPLIT_ENTER; continue;
                    }
                }
                state = ST_SPLIT_CORE; continue;
            }
            if (state == ST_SPLIT_CORE) {
                f.start += f.minItem; f.len = f.maxItem - f.minItem + 1;      // the list keeps only the core
                f.sqo = f.sQO; f.eqo = f.eQO; f.sro = f.sRO; f.refLen = (int)((1u + f.eRO - f.sRO) & 0xFFFFu); f.score = f.maxAGS;
                const bool goBack = (f.sQO != wS), goForw = (f.eQO != wE);
                if (goBack && goForw) extendCarefully(f, b, true, true);
                else if (goBack) extendCarefully(f, b, true, false);
                else extendCarefully(f, b, false, true);                      // sic: also when neither end was cut
                f.status |= stSplit; f.phase = PH_NONE;
                state = ST_SCORE; continue;
            }
            // ST_RETURN
            if (depth == 0) { if (f.status & stScored) emit(f, b); break; }
            if (f.status & stScored) { f.status |= stSplit | stAligned; emit(f, b); }
            depth--; f = st[depth];
            state = f.phase == PH_AFTER_HEAD ? ST_SPLIT_TAIL : ST_SPLIT_CORE;
        }
        (void)pushes;
    }
    // ---- publish, or hand the root to the wave kernel ----------------------------------------------------------------------------------------
    if (live && fall) atomicAdd(&gFallWhy[why & 7], 1u);
    { const unsigned long long fm = __ballot(live && fall); const unsigned sl = waveReserve(fm, Sx.fallCount, lane); if (live && fall) Sx.fallList[sl] = r; }
    const bool pub = live && !fall;
    int inclC = pub ? nCl : 0, inclO = pub ? nOutOps : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inclC, d, 64), u = __shfl_up(inclO, d, 64); if (lane >= d) { inclC += t; inclO += u; } }
    const int totC = __shfl(inclC, 63, 64), totO = __shfl(inclO, 63, 64); unsigned cb = 0, ob = 0;
    if (lane == 63 && totC) { cb = atomicAdd(&A.outCounts[0], (unsigned)totC); ob = atomicAdd(&A.outCounts[1], (unsigned)totO); }
    cb = (unsigned)__shfl((int)cb, 63, 64); ob = (unsigned)__shfl((int)ob, 63, 64);
    if (pub) {
        const unsigned ci = cb + (unsigned)(inclC - nCl), oi = ob + (unsigned)(inclO - nOutOps);
        if (nCl && (ci + (unsigned)nCl > A.outClumpCap || (unsigned long long)oi + (unsigned)nOutOps > (unsigned long long)A.outOpsCap)) atomicCAS(A.errFlag, 0, (int)YERR_OUT);
        else {
            const char codes[4] = {'M', 'R', 'D', 'I'};
            for (int k = 0; k < nOutOps; k++) { const uint32_t op = outOps[k]; A.outOps[oi + (unsigned)k] = ((uint32_t)(uint8_t)codes[opCode(op) & 3] << 16) | (uint32_t)opLen(op);
                }
            for (int k = 0; k < nCl; k++) { ygpu_clump c = outCl[k]; c.op_start += oi; A.outClumps[ci + (unsigned)k] = c; A.outRoot[ci + (unsigned)k] = r;
                A.outPush[ci + (unsigned)k] = (uint32_t)k; }
            A.rootPushCount[r] = (unsigned)nCl;
        }
    }
    unsigned cScored = pub ? (unsigned)nCl : 0u, cOps = pub ? (unsigned)nOutOps : 0u;
    if (!pub) { splits = extCalls = extRows = extCells = perfect = touched = 0; }
    splits = waveSumU(splits); extCalls = waveSumU(extCalls); extRows = waveSumU(extRows); extCells = waveSumU(extCells); perfect = waveSumU(perfect); touched = waveSumU(touched);
        cScored = waveSumU(cScored); cOps = waveSumU(cOps);
    if (lane == 0 && (splits | extCalls | cScored)) {
        unsigned long long *c = A.ctr->v;
        atomicAdd(&c[C_SPLITS], (unsigned long long)splits); atomicAdd(&c[C_EXT_CALLS], (unsigned long long)extCalls); atomicAdd(&c[C_EXT_ROWS], (unsigned long long)extRows);
            atomicAdd(&c[C_EXT_CELLS], (unsigned long long)extCells);
        atomicAdd(&c[C_PERFECT], (unsigned long long)perfect); atomicAdd(&c[C_TOUCHED], (unsigned long long)touched); atomicAdd(&c[C_SCORED], (unsigned long long)cScored);
            atomicAdd(&c[C_OPS], (unsigned long long)cOps);
    }
}
