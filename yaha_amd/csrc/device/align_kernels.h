// align_kernels.h -- the wave-per-root kernels over align.h: k_align (any band: the whole of A5-A8 for a root) and k_dp_batch (the stage-level DP entry).
// Included by stage_align.hip only; align.h itself (device functions, no kernel) is shared with the chain stage.
#pragma once
#include "align.h"

// Persistent waves pull root clumps from a queue (one 64-thread workgroup = one wavefront).
__global__ void __launch_bounds__(64) k_align(AlignArgs A)
{
    YD_HIGH_PRIO();
    const unsigned wave = blockIdx.x;
    WaveMem M = carveScratch(A.scratch + (size_t)wave * A.scratchPerWave, A.maxQ, A.traceRows, A.listCap, A.genCap);
    __shared__ uint16_t sTrace[YD_LDS_CELLS];
    Aligner al(A, M, sTrace);
    PROF_INIT();
    const unsigned nRoots = uniU(A.nRoots);
    for (;;) {
        if (__ballot(1) != ~0ull) { atomicCAS(A.errFlag, 0, (int)YERR_EXEC); break; }     // a lane left the wave-uniform flow: fail loudly
        unsigned t = 0;
        if (laneId() == 0) t = atomicAdd(A.queueHead, 1u);
        const unsigned r = uniU(t);
        if (r >= nRoots) break;
        { PROF_T0(); al.processRoot(r); PROF_ADD(PF_ROOT); PROF_CNT(PF_ROOTS); }
        if (laneId() == 0) A.rootPushCount[r] = al.pushes;
        if (UNI_B(al.err != 0)) { if (laneId() == 0) atomicCAS(A.errFlag, 0, al.err); break; }
    }
    al.flushCounters();
    PROF_FLUSH();
}

// ---- stage-level test entry: a batch of independent DP calls (ygpu_dp_batch) -------------------------------------------
struct DPBatchArgs {
    DevParams P; const uint8_t *bases; DevBatch B; const ygpu_dp_problem *probs; uint32_t n; unsigned int *queueHead;
    uint8_t *scratch; size_t scratchPerWave; int maxQ, listCap, genCap, traceRows;
    ygpu_dp_result *res; uint32_t *ops; unsigned int *opsCount; uint32_t opsCap; int *errFlag;
};
__global__ void __launch_bounds__(64) k_dp_batch(DPBatchArgs A)
{
    YD_HIGH_PRIO();
    const int lane = laneId();
    WaveMem M = carveScratch(A.scratch + (size_t)blockIdx.x * A.scratchPerWave, A.maxQ, A.traceRows, A.listCap, A.genCap);
    __shared__ uint16_t sTrace[YD_LDS_CELLS];
    int err = 0; WaveScratch S; S.ldsTrace = sTrace; S.trace = M.trace; S.traceRows = A.traceRows; S.tmpOps = M.tmpOps; S.tmpCap = 2 * A.maxQ + 512; S.gen = M.gen;
        S.genCap = A.genCap; S.err = &err;
    const unsigned nProb = uniU(A.n);
    for (;;) {
        if (__ballot(1) != ~0ull) { atomicCAS(A.errFlag, 0, (int)YERR_EXEC); break; }
        unsigned t = 0; if (lane == 0) t = atomicAdd(A.queueHead, 1u);
        const unsigned r = uniU(t);
        if (r >= nProb) break;
        const ygpu_dp_problem p = A.probs[r];
        YDBG("k_dp_batch r %u mode %d\n", r, (int)p.mode);
        const uint32_t r0 = A.B.readOff[p.read]; const uint8_t *q = (p.strand ? A.B.rev : A.B.fwd) + r0;
        DPOut o = dpWave(A.P, A.bases, q, p.mode, p.rOff, p.rLen, p.qOff, p.qLen, S);
        YDBG("dp done score %d nOps %d err %d\n", o.score, o.nOps, err);
        if (UNI_B(err != 0)) { if (lane == 0) atomicCAS(A.errFlag, 0, err); break; }
        const bool rev = p.mode == YGPU_DP_EXT_REV; unsigned oi = 0; const int n = uni(o.score != 0 || p.mode < YGPU_DP_EXT_FWD ? o.nOps : 0);
        if (lane == 0) oi = atomicAdd(A.opsCount, (unsigned)n); oi = uniU(oi);
        if (UNI_B(oi + (unsigned)n > A.opsCap)) { if (lane == 0) atomicCAS(A.errFlag, 0, (int)YERR_OUT); break; }
        const char codes[4] = {'M', 'R', 'D', 'I'};
        for (int k = lane; k < n; k += 64) { uint32_t op = dpOp(S, o, rev, k); A.ops[oi + k] = ((uint32_t)(uint8_t)codes[opCode(op) & 3] << 16) | (uint32_t)opLen(op); }
        if (lane == 0) { ygpu_dp_result rr; rr.score = o.score; rr.addedQLen = (uint16_t)o.addedQ; rr.addedRLen = (uint16_t)o.addedR; rr.op_start = oi; rr.n_ops = (uint32_t)n;
            A.res[r] = rr; }
    }
}
