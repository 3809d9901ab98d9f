// align.h -- stages A5..A8 + A10 on the device: one wavefront owns one root clump from alignClump to the last
// split child (reference AlignHelpers.c:205-579, AlignExtFrag.cpp:30-234, SW.cpp:553-788).
//
// Control flow is wave-uniform (every lane computes the same scalars); the 64 lanes are used inside the DP
// (dp_wave.h), the exact-match extensions (64 bases per step + ballot) and list copies.  The reference's
// recursion (scoreClump -> splitClump -> splitClumpHelper -> scoreClump ...) is an explicit frame stack in the
// wave's HBM scratch; every frame owns a list buffer with head-room on both sides so that extensions can grow the
// edit list in place.  Results go to batch-wide arenas through one atomic per clump; (root rank, push number) is
// recorded so that a later pass can lay the clumps out in the reference's QS->clumps order.
#pragma once
#include "dp_wave.h"

struct ChainClumpRec { uint32_t rs, fragOff, nFrags, region, seq, matched; };
// root r's record: the records in RANK order (round 6: k_clump_order writes a copy in the order of the roots, so that a wave's 64 roots are one stretch of 1.5 KB --
// six kernels start their chain of dependent fetches here) -- order == nullptr -- or, for the arena as the chain stage left it, through the rank table
#define YD_ROOT_REC(A, r) ((A).order ? (A).clumps[(A).order[(r)]] : (A).clumps[(r)])

#define YD_DEPTH 24

struct Frame {
    uint32_t sro; int sqo, eqo, refLen; int score, status, wS, wE; int start, len;
    int phase, minItem, maxItem, sQO, eQO; uint32_t sRO, eRO; int maxAGS;
    int cSqo, cEqo, cRefLen; uint32_t cSro;               // the frame's fragment before the split rewrote it ("curFrag")
};
enum { PH_NONE = 0, PH_AFTER_HEAD = 1, PH_AFTER_TAIL = 2 };

struct AlignArgs {
    DevParams P; const uint8_t *bases; DevBatch B;
    const uint32_t *order; uint32_t nRoots; const ChainClumpRec *clumps; DevFrag *clumpFrags;
    unsigned int *queueHead;
    uint8_t *scratch; size_t scratchPerWave; int maxQ, listCap, front, genCap, traceRows;
    ygpu_clump *outClumps; uint32_t *outOps; uint32_t *outRoot; uint32_t *outPush; unsigned int *outCounts; uint32_t outClumpCap, outOpsCap;
    unsigned int *rootPushCount;
    DevCounters *ctr; int *errFlag;
};

// layout of one wave's scratch
struct WaveMem { uint16_t *trace; uint32_t *tmpOps; int *gen; uint32_t *arena; Frame *frames; };
// traceRows: rows of 64 trace cells (>= maxQ + 2; more when -G / -MD allow gap fills whose strip does not fit that, see alignDims in stage_align.hip)
__host__ __device__ inline size_t alignScratchBytes(int maxQ, int traceRows, int listCap, int genCap)
{
    size_t b = 0;
    b += (size_t)traceRows * 64 * 2; b = (b + 255) & ~(size_t)255;
    b += (size_t)(2 * maxQ + 512) * 4; b = (b + 255) & ~(size_t)255;
    b += (size_t)3 * (genCap + 3) * 4; b = (b + 255) & ~(size_t)255;
    b += (size_t)YD_DEPTH * listCap * 4; b = (b + 255) & ~(size_t)255;
    b += (size_t)YD_DEPTH * sizeof(Frame); b = (b + 255) & ~(size_t)255;
    return b;
}
__device__ inline WaveMem carveScratch(uint8_t *p, int maxQ, int traceRows, int listCap, int genCap)
{
    WaveMem m; size_t b = 0;
    m.trace = (uint16_t *)(p + b); b += (size_t)traceRows * 64 * 2; b = (b + 255) & ~(size_t)255;
    m.tmpOps = (uint32_t *)(p + b); b += (size_t)(2 * maxQ + 512) * 4; b = (b + 255) & ~(size_t)255;
    m.gen = (int *)(p + b); b += (size_t)3 * (genCap + 3) * 4; b = (b + 255) & ~(size_t)255;
    m.arena = (uint32_t *)(p + b); b += (size_t)YD_DEPTH * listCap * 4; b = (b + 255) & ~(size_t)255;
    m.frames = (Frame *)(p + b);
    return m;
}

// ---- exact-match extension: 64 bases per step (extendFragment{Forward,Backward}ToStopPerfectly, AlignExtFrag.cpp:30-48)
__device__ inline int perfectFwd(const uint8_t *bases, const uint8_t *q, int qOff0, uint32_t rOff0, int len)
{
    PROF_T0();
    const int lane = laneId(); int count = 0;
    while (count < len) {
        int k = count + lane;
        bool ok = (k < len) && (q[qOff0 + k] == ref4(bases, rOff0 + (uint32_t)k));
        unsigned long long bad = __ballot(!ok);
        if (bad) { count += __ffsll((long long)bad) - 1; break; }
        count += 64;
    }
    PROF_ADD(PF_PERFECT);
    return uni(count);
}
__device__ inline int perfectBack(const uint8_t *bases, const uint8_t *q, int qOff0, uint32_t rOff0, int len)
{
    PROF_T0();
    const int lane = laneId(); int count = 0;
    while (count < len) {
        int k = count + lane;
        bool ok = (k < len) && (q[qOff0 - k] == ref4(bases, rOff0 - (uint32_t)k));
        unsigned long long bad = __ballot(!ok);
        if (bad) { count += __ffsll((long long)bad) - 1; break; }
        count += 64;
    }
    PROF_ADD(PF_PERFECT);
    return uni(count);
}

struct Aligner {
    const AlignArgs &A; const DevParams &P; WaveMem M; WaveScratch S; int err;
    const uint8_t *q; int qlen; int lane;
    // local work counters
    unsigned extCalls, extRows, extCells, gapCalls, gapRows, gapCells, perfect, touched, opsOut, splits, scored;
    uint32_t rootRank; unsigned pushes;

    __device__ Aligner(const AlignArgs &a, WaveMem m, uint16_t *ldsTrace) : A(a), P(a.P), M(m), err(0), q(nullptr), qlen(0), lane(laneId()),
        extCalls(0), extRows(0), extCells(0), gapCalls(0), gapRows(0), gapCells(0), perfect(0), touched(0), opsOut(0), splits(0), scored(0), rootRank(0), pushes(0)
    { S.ldsTrace = ldsTrace; S.trace = M.trace; S.traceRows = A.traceRows; S.tmpOps = M.tmpOps; S.tmpCap = 2 * A.maxQ + 512; S.gen = M.gen; S.genCap = A.genCap; S.err = &err; }

    __device__ uint32_t *buf(int depth) const { return M.arena + (size_t)depth * A.listCap; }
    __device__ int gapCost(int len) const { return len > 0 ? -(P.GO + len * P.GE) : 0; }
    __device__ static uint32_t ero(uint32_t sro, int refLen) { return sro + (uint32_t)refLen - 1u; }

    // mergeEOLToBack with a one-op source (SW.cpp:207-261)
    __device__ void appendOp(uint32_t *b, int start, int &len, int code, int n)
    {
        if (len > 0 && opCode(b[start + len - 1]) == code) { b[start + len - 1] = opMake(code, (opLen(b[start + len - 1]) + n) & 0xFFFF); return; }
        if (start + len >= A.listCap) { err = YERR_ARENA; return; }
        b[start + len] = opMake(code, n & 0xFFFF); len++;
    }
    // mergeEOLToBack(list, T[t0,t1))   T = last DP result in list order
    __device__ void mergeBack(uint32_t *b, int start, int &len, const DPOut &o, bool rev, int t0, int t1)
    {
        if (t1 <= t0) return;
        if (len > 0) {
            uint32_t last = b[start + len - 1], first = dpOp(S, o, rev, t0);
            if (opCode(last) == opCode(first)) { b[start + len - 1] = opMake(opCode(last), (opLen(last) + opLen(first)) & 0xFFFF); t0++; }
        }
        int cnt = t1 - t0;
        if (start + len + cnt > A.listCap) { err = YERR_ARENA; return; }
        for (int k = lane; k < cnt; k += 64) b[start + len + k] = dpOp(S, o, rev, t0 + k);
        len += cnt; __threadfence_block();
    }
    // mergeEOLToFront(list, T[t0,t1))  (SW.cpp:151-205)
    __device__ void mergeFront(uint32_t *b, int &start, int &len, const DPOut &o, bool rev, int t0, int t1)
    {
        if (t1 <= t0) return;
        if (len > 0) {
            uint32_t first = b[start], last = dpOp(S, o, rev, t1 - 1);
            if (opCode(first) == opCode(last)) { b[start] = opMake(opCode(first), (opLen(first) + opLen(last)) & 0xFFFF); t1--; }
        }
        int cnt = t1 - t0;
        if (start - cnt < 0) { err = YERR_ARENA; return; }
        start -= cnt;
        for (int k = lane; k < cnt; k += 64) b[start + k] = dpOp(S, o, rev, t0 + k);
        len += cnt; __threadfence_block();
    }

    // the same two merges with the source already in list order in HBM (ext_lanes.h results)
    __device__ void mergeBackSrc(uint32_t *b, int start, int &len, const uint32_t *src, int cnt)
    {
        if (cnt <= 0) return;
        int t0 = 0;
        if (len > 0) {
            uint32_t last = b[start + len - 1], first = uniU(src[0]);
            if (opCode(last) == opCode(first)) { b[start + len - 1] = opMake(opCode(last), (opLen(last) + opLen(first)) & 0xFFFF); t0 = 1; }
        }
        const int n = cnt - t0;
        if (start + len + n > A.listCap) { err = YERR_ARENA; return; }
        for (int k = lane; k < n; k += 64) b[start + len + k] = src[t0 + k];
        len += n; __threadfence_block();
    }
    // reversed: list element k of the source is src[cnt-1-k]
    __device__ void mergeFrontSrc(uint32_t *b, int &start, int &len, const uint32_t *src, int cnt, bool reversed)
    {
        if (cnt <= 0) return;
        int t1 = cnt;
        if (len > 0) {
            uint32_t first = b[start], last = uniU(src[reversed ? 0 : cnt - 1]);
            if (opCode(first) == opCode(last)) { b[start] = opMake(opCode(first), (opLen(first) + opLen(last)) & 0xFFFF); t1--; }
        }
        if (start - t1 < 0) { err = YERR_ARENA; return; }
        start -= t1;
        for (int k = lane; k < t1; k += 64) b[start + k] = src[reversed ? cnt - 1 - k : k];
        len += t1; __threadfence_block();
    }

    __device__ DPOut runDP(int mode, uint32_t rOff, int rLen, int qOff, int qLen)
    {
        DPOut o = dpWave(P, A.bases, q, mode, rOff, rLen, qOff, qLen, S);
        o.score = uni(o.score); o.addedQ = uni(o.addedQ); o.addedR = uni(o.addedR); o.nOps = uni(o.nOps); o.rows = uni(o.rows); o.cells = uni(o.cells);
        if (mode >= YGPU_DP_EXT_FWD) { extCalls++; extRows += o.rows; extCells += o.cells; touched += o.rows + 4 * P.bandWidth + 1; }
        else { gapCalls++; gapRows += o.rows; gapCells += o.cells; touched += rLen; }
        return o;
    }

    // findAGSForwardExtensionCarefully, SW.cpp:553-669
    __device__ int fwdCarefully(Frame &f, uint32_t *b, uint32_t rOff, int qOff, int qLen, int score, int &aQ, int &aR)
    {
        DPOut o = runDP(YGPU_DP_EXT_FWD, rOff, 0, qOff, qLen);
        aQ = o.addedQ; aR = o.addedR;
        int initAGS = o.score; if (initAGS <= 0) { aQ = aR = 0; return 0; }
        int QLen = 0, RLen = 0, AGS = score, maxAGS = score, maxItem = -1, maxQLen = 0, maxRLen = 0, nT = o.nOps;
        for (int k = 0; k < o.nOps; k++) {
            uint32_t op = uniU(dpOp(S, o, false, k)); int code = opCode(op), len = opLen(op);
            if (code == OP_M) { QLen += len; RLen += len; AGS += P.MS * len; }
            else if (code == OP_R) { QLen += len; RLen += len; AGS -= P.RC * len; }
            else if (code == OP_I) { QLen += len; AGS -= (P.GO + P.GE * len); }
            else { RLen += len; AGS -= (P.GO + P.GE * len); }
            if (AGS > maxAGS) { maxAGS = AGS; maxQLen = QLen; maxRLen = RLen; maxItem = k; }
            else if (AGS <= 0) {
                if (maxAGS <= score) { aQ = aR = 0; return 0; }
                nT = maxItem + 1; aQ = maxQLen; aR = maxRLen; initAGS = maxAGS - score; break;
            }
        }
        mergeBack(b, f.start, f.len, o, false, 0, nT);
        return initAGS;
    }
    // findAGSBackwardExtensionCarefully, SW.cpp:671-788
    __device__ int backCarefully(Frame &f, uint32_t *b, uint32_t rOff, int qOff, int qLen, int score, int &aQ, int &aR)
    {
        DPOut o = runDP(YGPU_DP_EXT_REV, rOff, 0, qOff, qLen);
        aQ = o.addedQ; aR = o.addedR;
        if (o.score <= 0) { aQ = aR = 0; return 0; }
        int QLen = 0, RLen = 0, AGS = 0, maxAGS = 0, startItem = -1;
        for (int k = 0; k < o.nOps; k++) {
            uint32_t op = uniU(dpOp(S, o, true, k)); int code = opCode(op), len = opLen(op);
            if (code == OP_M) { QLen += len; RLen += len; AGS += P.MS * len; }
            else if (code == OP_R) { QLen += len; RLen += len; AGS -= P.RC * len; }
            else if (code == OP_I) { QLen += len; AGS -= (P.GO + P.GE * len); }
            else { RLen += len; AGS -= (P.GO + P.GE * len); }
            if (AGS <= 0) { AGS = 0; maxAGS = 0; QLen = 0; RLen = 0; startItem = k; }
            if (AGS > maxAGS) maxAGS = AGS;
        }
        if (AGS <= 0 || maxAGS >= AGS + score) { aQ = aR = 0; return 0; }
        mergeFront(b, f.start, f.len, o, true, startItem + 1, o.nOps);
        aQ = QLen; aR = RLen;
        return AGS;
    }

    // extendClumpForwardReverseTemplated<goBack,goForw,goCarefully>, AlignExtFrag.cpp:64-144
    // first half: exact-match extensions (lines 76-107); returns the lengths still open for the DP extensions
    __device__ void extendPerfect(Frame &f, uint32_t *b, bool goBack, bool goForw, int &backLen, int &forwLen)
    {
        int score = f.score; backLen = 0; forwLen = 0;
        if (goBack) {
            uint32_t bl = (uint32_t)f.sqo < f.sro ? (uint32_t)f.sqo : f.sro; backLen = (int)bl;
            if (backLen > 0) {
                int m = perfectBack(A.bases, q, f.sqo - 1, f.sro - 1u, backLen); perfect += m; touched += m + (m < backLen);
                if (m > 0) { b[f.start] = opMake(opCode(b[f.start]), (opLen(b[f.start]) + m) & 0xFFFF); score += m * P.MS; backLen -= m; f.sqo -= m; f.sro -= (uint32_t)m;
                    f.refLen = (f.refLen + m) & 0xFFFF; }
            }
        }
        if (goForw) {
            uint32_t ql = (uint32_t)(((qlen - 1) - f.eqo) & 0xFFFF), rl = P.maxROff - ero(f.sro, f.refLen);
            forwLen = (int)(ql < rl ? ql : rl);
            if (forwLen > 0) {
                int m = perfectFwd(A.bases, q, f.eqo + 1, ero(f.sro, f.refLen) + 1u, forwLen); perfect += m; touched += m + (m < forwLen);
                if (m > 0) { int li = f.start + f.len - 1; b[li] = opMake(opCode(b[li]), (opLen(b[li]) + m) & 0xFFFF); score += m * P.MS; forwLen -= m; f.eqo += m;
                    f.refLen = (f.refLen + m) & 0xFFFF; }
            }
        }
        f.score = score; backLen = uni(backLen); forwLen = uni(forwLen);
    }
    __device__ void extendClump(Frame &f, uint32_t *b, bool goBack, bool goForw, bool carefully)
    {
        int backLen, forwLen; extendPerfect(f, b, goBack, goForw, backLen, forwLen);
        int score = f.score, aQ = 0, aR = 0;
        if (goBack && backLen >= P.minExtLength) {
            int ns;
            if (carefully) ns = backCarefully(f, b, f.sro - 1u, (f.sqo - 1) & 0xFFFF, backLen & 0xFFFF, score, aQ, aR);
            else { DPOut o = runDP(YGPU_DP_EXT_REV, f.sro - 1u, 0, (f.sqo - 1) & 0xFFFF, backLen & 0xFFFF); ns = o.score; aQ = o.addedQ; aR = o.addedR;
                if (ns > 0) mergeFront(b, f.start, f.len, o, true, 0, o.nOps); }
            if (ns > 0) { score += ns; f.sqo = (f.sqo - aQ) & 0xFFFF; f.sro -= (uint32_t)aR; f.refLen = (f.refLen + aR) & 0xFFFF; }
        }
        if (goForw && forwLen >= P.minExtLength) {
            int ns;
            if (carefully) ns = fwdCarefully(f, b, ero(f.sro, f.refLen) + 1u, (f.eqo + 1) & 0xFFFF, forwLen & 0xFFFF, score, aQ, aR);
            else { DPOut o = runDP(YGPU_DP_EXT_FWD, ero(f.sro, f.refLen) + 1u, 0, (f.eqo + 1) & 0xFFFF, forwLen & 0xFFFF); ns = o.score; aQ = o.addedQ; aR = o.addedR;
                if (ns > 0) mergeBack(b, f.start, f.len, o, false, 0, o.nOps); }
            if (ns > 0) { score += ns; f.eqo = (f.eqo + aQ) & 0xFFFF; f.refLen = (f.refLen + aR) & 0xFFFF; }
        }
        f.score = score;
    }

    // alignClump, AlignHelpers.c:205-272 (+ makeAndAlignSFragmentToFillGap AlignExtFrag.cpp:164-234, collapseSFragments :274-300)
    // everything of alignClump before extendClumpForwardReverse
    __device__ void alignRootPre(const ChainClumpRec &rec, Frame &f)
    {
        DevFrag *F = A.clumpFrags + rec.fragOff; const int n = (int)rec.nFrags;
        for (int k = 1; k < n; k++) {
            DevFrag f1 = F[k - 1], f2 = F[k];
            int gap = (int)min(gapI(f1.eqo, f2.sqo), gapU(ero(f1.sro, f1.refLen), f2.sro));
            int c = perfectBack(A.bases, q, (int)f2.sqo - 1, f2.sro - 1u, gap); perfect += c; touched += c + (c < gap);
            if (c > 0) { f2.sqo = (uint16_t)(f2.sqo - c); f2.sro -= (uint32_t)c; f2.refLen = (uint16_t)(f2.refLen + c); }
            gap -= c;
            c = perfectFwd(A.bases, q, (int)f1.eqo + 1, ero(f1.sro, f1.refLen) + 1u, gap); perfect += c; touched += c + (c < gap);
            if (c > 0) { f1.eqo = (uint16_t)(f1.eqo + c); f1.refLen = (uint16_t)(f1.refLen + c); }
            F[k - 1] = f1; F[k] = f2;
        }
        uint32_t *b = buf(0); f.start = A.front; f.len = 0; int total = 0;
        for (int k = 0; k < n && !err; k++) {
            const DevFrag f1 = F[k];
            int ql = fragQLen(f1.sqo, f1.eqo);
            appendOp(b, f.start, f.len, OP_M, ql); total += P.MS * ql;
            if (k + 1 == n) break;
            const DevFrag f2 = F[k + 1];
            int qGap = (int)(gapI(f1.eqo, f2.sqo) & 0xFFFF), rGap = (int)(gapU(ero(f1.sro, f1.refLen), f2.sro) & 0xFFFF);
            if (qGap == 0 && rGap == 0) continue;
            int nsqo = (f1.eqo + 1) & 0xFFFF; uint32_t nsro = ero(f1.sro, f1.refLen) + 1u;
            if (qGap == 0) { appendOp(b, f.start, f.len, OP_D, rGap); total += gapCost(rGap); }
            else if (rGap == 0) { appendOp(b, f.start, f.len, OP_I, qGap); total += gapCost(qGap); }
            else if (rGap == 1 && qGap == 1) { appendOp(b, f.start, f.len, OP_R, 1); total -= P.RC; }
            else {
                int lenDiff = qGap > rGap ? qGap - rGap : rGap - qGap;
                int mode = (lenDiff + P.bandWidth * 2 + 1 < rGap) ? YGPU_DP_BANDED : YGPU_DP_FULL;
                DPOut o = runDP(mode, nsro, rGap, nsqo, qGap);
                mergeBack(b, f.start, f.len, o, false, 0, o.nOps); total += o.score;
            }
        }
        const DevFrag f0 = F[0], fn = F[n - 1];
        f.sro = f0.sro; f.sqo = f0.sqo; f.eqo = fn.eqo; f.refLen = (int)((1u + ero(fn.sro, fn.refLen) - f0.sro) & 0xFFFFu);
        f.score = total; f.status = (rec.rs & 1u) ? stReversed : 0; f.phase = PH_NONE;
    }
    __device__ void alignRoot(const ChainClumpRec &rec, Frame &f)
    {
        alignRootPre(rec, f);
        if (err) return;
        extendClump(f, buf(0), true, true, false);
        f.status |= stAligned;
    }

    // scoreClump, AlignHelpers.c:302-366.  returns 0 = done (scored or rejected), 1 = needs splitClump
    __device__ int scoreList(Frame &f, const uint32_t *b, int &oMatches, int &oMism, int &oGap, int &oLen, int &oScore)
    { PROF_T0(); int rr_ = scoreListImpl(f, b, oMatches, oMism, oGap, oLen, oScore); PROF_ADD(PF_SCORE); return rr_; }
    __device__ int scoreListImpl(Frame &f, const uint32_t *b, int &oMatches, int &oMism, int &oGap, int &oLen, int &oScore)
    {
        int AGS = 0, maxAGS = 0, matches = 0, mism = 0, ins = 0, del = 0; const int n = f.len, aligned = f.score;
        for (int base = 0; base < n; base += 64) {
            uint32_t reg = (base + lane < n) ? b[f.start + base + lane] : 0u; const int cnt = min(64, n - base);
            for (int k = 0; k < cnt; k++) {
                uint32_t op = (uint32_t)bcast((int)reg, k); int code = opCode(op), len = opLen(op);
                if (code == OP_M) { matches += len; AGS += P.MS * len; } else if (code == OP_R) { mism += len; AGS -= P.RC * len; }
                else if (code == OP_I) { ins += len; AGS -= (P.GO + P.GE * len); } else { del += len; AGS -= (P.GO + P.GE * len); }
                if (AGS <= 0 || (AGS >= aligned && (base + k) != n - 1)) return 1;
                if (AGS > maxAGS) maxAGS = AGS;
            }
        }
        if (matches >= P.minRawScore && maxAGS > AGS) return 1;
        oMatches = -1;
        if (matches < P.minRawScore) return 0;
        int tot = (matches + mism + ins + del) & 0xFFFF; matches &= 0xFFFF;
        oMatches = matches; oMism = mism & 0xFFFF; oGap = (ins + del) & 0xFFFF; oLen = tot; oScore = AGS & 0xFFFF;
        double percent = (double)matches / (double)tot;
        if (percent < (double)P.minIdentity) return 0;                    // float threshold widened to double, AlignHelpers.c:359-360
        f.status |= stScored;
        return 0;
    }

    __device__ bool hasMaxMatch(const uint32_t *b, int start, int n)      // EditOpList2Maxmatch, SW.cpp:1215-1222
    {
        bool any = false;
        for (int base = 0; base < n; base += 64) {
            int k = base + lane; bool hit = false;
            if (k < n) { uint32_t op = b[start + k]; hit = opCode(op) == OP_M && opLen(op) >= P.wordLen; }
            if (__ballot(hit)) { any = true; break; }
        }
        return any;
    }

    __device__ void emit(const Frame &f, const uint32_t *b, int matches, int mism, int gap, int totLen, int totScore)
    {
        PROF_T0();
        unsigned ci = 0, oi = 0;
        if (lane == 0) { ci = atomicAdd(&A.outCounts[0], 1u); oi = atomicAdd(&A.outCounts[1], (unsigned)f.len); }
        ci = uniU(ci); oi = uniU(oi);
        if (ci >= A.outClumpCap || oi + (unsigned)f.len > A.outOpsCap) { err = YERR_OUT; return; }
        const char codes[4] = {'M', 'R', 'D', 'I'};
        for (int k = lane; k < f.len; k += 64) { uint32_t op = b[f.start + k]; A.outOps[oi + k] = ((uint32_t)(uint8_t)codes[opCode(op) & 3] << 16) | (uint32_t)opLen(op); }
        if (lane == 0) {
            ygpu_clump c; c.sro = f.sro; c.sqo = (uint16_t)f.sqo; c.eqo = (uint16_t)f.eqo; c.refLen = (uint16_t)f.refLen; c.totScore = (uint16_t)totScore;
                c.totLength = (uint16_t)totLen;
            c.matchedBases = (uint16_t)matches; c.mismatchedBases = (uint16_t)mism; c.gapBases = (uint16_t)gap; c.status = (uint8_t)f.status; c.reserved = 0; c.op_start = oi;
                c.n_ops = (uint32_t)f.len;
            A.outClumps[ci] = c; A.outRoot[ci] = rootRank; A.outPush[ci] = pushes;
        }
        pushes++; scored++; opsOut += (unsigned)f.len;
        PROF_ADD(PF_EMIT);
    }

    // The whole life of one root clump.
    __device__ void processRoot(uint32_t rank)
    {
        const ChainClumpRec rec = YD_ROOT_REC(A, rank);
        const uint32_t read = rec.rs >> 1; const uint32_t r0 = A.B.readOff[read];
        qlen = (int)(A.B.readOff[read + 1] - r0); q = ((rec.rs & 1u) ? A.B.rev : A.B.fwd) + r0;
        rootRank = rank; pushes = 0;
        Frame f;
        alignRoot(rec, f);
        finishRoot(f);
    }
    __device__ void setRead(const ChainClumpRec &rec)
    {
        const uint32_t read = rec.rs >> 1; const uint32_t r0 = A.B.readOff[read];
        qlen = (int)(A.B.readOff[read + 1] - r0); q = ((rec.rs & 1u) ? A.B.rev : A.B.fwd) + r0;
    }
    // scoreClump / splitClump state machine on an aligned clump (frame 0)
    __device__ void finishRoot(Frame f)
    {
        int depth = 0;
        // per-frame results of the last scoreList
        int sm = -1, smm = 0, sg = 0, sl = 0, ss = 0;
        enum { ST_SCORE, ST_SPLIT_ENTER, ST_SPLIT_TAIL, ST_SPLIT_CORE, ST_RETURN } state = ST_SCORE;
        int guard = 0;
        while (!UNI_B(err != 0)) {
            if (++guard > 100000) { err = YERR_DEPTH; break; }
            state = (decltype(state))uni((int)state); depth = uni(depth);
            uint32_t *b = buf(depth);
            if (state == ST_SCORE) {
                if (UNI_B(f.status & stScored)) { state = ST_RETURN; continue; }
                int r = uni(scoreList(f, b, sm, smm, sg, sl, ss));
                if (r == 1) { splits++; f.wS = f.sqo; f.wE = f.eqo; state = ST_SPLIT_ENTER; }      // splitClump, AlignHelpers.c:561-579
                else state = ST_RETURN;
                continue;
            }
            if (state == ST_SPLIT_ENTER) {                                   // splitClumpHelper, AlignHelpers.c:374-557
                int matches = 0, mism = 0, ins = 0, del = 0, AGS = 0, maxAGS = -10000, maxItem = -1, minItem = -1;
                int eQO = 0, sQO = 0; uint32_t eRO = 0, sRO = 0; const int n = f.len;
                for (int base = 0; base < n; base += 64) {
                    uint32_t reg = (base + lane < n) ? b[f.start + base + lane] : 0u; const int cnt = min(64, n - base);
                    for (int k = 0; k < cnt; k++) {
                        uint32_t op = (uint32_t)bcast((int)reg, k); int code = opCode(op), len = opLen(op), ns;
                        if (code == OP_M) { matches += len; ns = P.MS * len; } else if (code == OP_R) { mism += len; ns = -(P.RC * len); }
                        else if (code == OP_I) { ins += len; ns = -(P.GO + P.GE * len); } else { del += len; ns = -(P.GO + P.GE * len); }
                        AGS += ns; if (AGS < 0) AGS = 0;
                        if (AGS > maxAGS) { maxAGS = AGS; maxItem = base + k; eQO = (f.sqo + matches + mism + ins - 1) & 0xFFFF;
                            eRO = f.sro + (uint32_t)(matches + mism + del) - 1u; }
                    }
                }
                AGS = maxAGS; matches = mism = ins = del = 0; int maxMatch = 0;
                for (int k = maxItem; k >= 0; k--) {
                    uint32_t op = uniU(b[f.start + k]); int code = opCode(op), len = opLen(op);
                    if (code == OP_M) { matches += len; AGS -= P.MS * len; if (len > maxMatch) maxMatch = len; } else if (code == OP_R) { mism += len; AGS += P.RC * len; }
                    else if (code == OP_I) { ins += len; AGS += (P.GO + P.GE * len); } else { del += len; AGS += (P.GO + P.GE * len); }
                    if (AGS <= 0) { minItem = k; sQO = (eQO - (matches + mism + ins - 1)) & 0xFFFF; sRO = eRO - (uint32_t)(matches + mism + del - 1); break; }
                }
                if (maxMatch < P.wordLen || minItem < 0) { state = ST_RETURN; continue; }
                f.minItem = minItem; f.maxItem = maxItem; f.sQO = sQO; f.eQO = eQO; f.sRO = sRO; f.eRO = eRO; f.maxAGS = maxAGS;
                f.cSqo = f.sqo; f.cEqo = f.eqo; f.cSro = f.sro; f.cRefLen = f.refLen;
                if (minItem != 0) {                                          // head remainder :463-495
                    if (hasMaxMatch(b, f.start, minItem)) {
                        if (depth + 1 >= YD_DEPTH) { err = YERR_DEPTH; break; }
                        uint32_t *cb = buf(depth + 1);
                        if (A.front + minItem > A.listCap) { err = YERR_ARENA; break; }
                        for (int k = lane; k < minItem; k += 64) cb[A.front + k] = b[f.start + k];
                        __threadfence_block();
                        Frame c; c.status = f.status & stReversed; c.sqo = f.cSqo; c.eqo = (sQO - 1) & 0xFFFF; c.sro = f.cSro;
                            c.refLen = (int)((1u + (sRO - 1u) - f.cSro) & 0xFFFFu);
                        c.score = 0; c.wS = f.wS; c.wE = f.wE; c.start = A.front; c.len = minItem; c.phase = PH_NONE;
                        f.phase = PH_AFTER_HEAD; M.frames[depth] = f; depth++; f = c; state = ST_SPLIT_ENTER; continue;
                    }
                }
                state = ST_SPLIT_TAIL; continue;
            }
            if (state == ST_SPLIT_TAIL) {                                    // tail remainder :500-531
                const int n = f.len;
                if (f.maxItem != n - 1) {
                    const int t0 = f.maxItem + 1, tl = n - t0;
                    if (hasMaxMatch(b, f.start + t0, tl)) {
                        if (depth + 1 >= YD_DEPTH) { err = YERR_DEPTH; break; }
                        uint32_t *cb = buf(depth + 1);
                        if (A.front + tl > A.listCap) { err = YERR_ARENA; break; }
                        for (int k = lane; k < tl; k += 64) cb[A.front + k] = b[f.start + t0 + k];
                        __threadfence_block();
                        Frame c; c.status = f.status & stReversed; c.sqo = (f.eQO + 1) & 0xFFFF; c.eqo = f.cEqo; c.sro = f.eRO + 1u;
                        c.refLen = (int)((1u + ero(f.cSro, f.cRefLen) - (f.eRO + 1u)) & 0xFFFFu);
                        c.score = 0; c.wS = f.wS; c.wE = f.wE; c.start = A.front; c.len = tl; c.phase = PH_NONE;
                        f.phase = PH_AFTER_TAIL; M.frames[depth] = f; depth++; f = c; state = ST_SPLIT_ENTER; continue;
                    }
                }
                state = ST_SPLIT_CORE; continue;
            }
            if (state == ST_SPLIT_CORE) {
                f.start += f.minItem; f.len = f.maxItem - f.minItem + 1;     // the list keeps only the core
                f.sqo = f.sQO; f.eqo = f.eQO; f.sro = f.sRO; f.refLen = (int)((1u + f.eRO - f.sRO) & 0xFFFFu); f.score = f.maxAGS;
                const bool goBack = (f.sQO != f.wS), goForw = (f.eQO != f.wE);
                if (goBack && goForw) extendClump(f, b, true, true, true);   // extendClumpForwardReverseCarefully, AlignExtFrag.cpp:151-156
                else if (goBack) extendClump(f, b, true, false, true);
                else extendClump(f, b, false, true, true);                   // sic: also when neither end was cut
                f.status |= stSplit; f.phase = PH_NONE;
                state = ST_SCORE; continue;
            }
            // ST_RETURN: this frame is finished
            if (depth == 0) { if (f.status & stScored) emit(f, b, sm, smm, sg, sl, ss);
                break; }
            if (f.status & stScored) { f.status |= stSplit | stAligned; emit(f, b, sm, smm, sg, sl, ss); }
            depth--; f = M.frames[depth];
            if (f.phase == PH_AFTER_HEAD) state = ST_SPLIT_TAIL; else state = ST_SPLIT_CORE;
        }
    }

    __device__ void flushCounters()
    {
        if (lane != 0) return;
        unsigned long long *c = A.ctr->v;
        atomicAdd(&c[C_EXT_CALLS], (unsigned long long)extCalls); atomicAdd(&c[C_EXT_ROWS], (unsigned long long)extRows); atomicAdd(&c[C_EXT_CELLS], (unsigned long long)extCells);
        atomicAdd(&c[C_GAP_CALLS], (unsigned long long)gapCalls); atomicAdd(&c[C_GAP_ROWS], (unsigned long long)gapRows); atomicAdd(&c[C_GAP_CELLS], (unsigned long long)gapCells);
        atomicAdd(&c[C_PERFECT], (unsigned long long)perfect); atomicAdd(&c[C_TOUCHED], (unsigned long long)touched); atomicAdd(&c[C_OPS], (unsigned long long)opsOut);
        atomicAdd(&c[C_SPLITS], (unsigned long long)splits); atomicAdd(&c[C_SCORED], (unsigned long long)scored);
    }
};
