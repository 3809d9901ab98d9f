// oqc_stage.h -- the post-filter on the device (SURVEY.md 8(f)-1 widened onto the GPU): Optimal Query Coverage, filter-by-similarity and mapping quality
// (reference GraphPath.cpp:897-1086) for every read of the batch, right where the hot path left its clumps, so that only the clumps that are PRINTED -- one or
// two of the ~75 a 1 kbp read produces -- and their edit ops travel to the host (9 KB a read -> ~0.2 KB) and the host's share of a read is parsing and
// printing.  The routine itself is ../oqc_core.h, the very source the host compiles (host/oqc.cpp): one read per lane, sequential, all of its work space in
// per-read slices of batch-wide arrays.  It is serial, branchy, latency-bound code -- a few hundred dependent steps a read -- that occupies one wave per CU
// and next to no issue slots: it runs underneath the other contexts' kernels.
#pragma once
#include "common.h"
#include "../oqc_core.h"

struct OqcArgs {
    yoqc::Params P; yoqc::Seqs G;
    const uint32_t *cs; const ygpu_clump *cl; const uint32_t *ops; const uint8_t *fwd; const uint32_t *readOff; uint32_t nReads;
    const unsigned long long *poolOff;
    yoqc::SortKey *keys; int *stack; yoqc::CNode *nodes, *prim; yoqc::PAttr *pa; int *pfxOff, *path, *pool; yoqc::OutRec *push, *out;
    uint32_t *outCnt, *outOpsCnt; uint32_t *primCnt;
};
// ints of running-sum tables a read may need: two per op and clump of the read (every clump's table built), none for reads of fewer than two clumps
__global__ void k_oqc_sizes(const uint32_t *cs, const ygpu_clump *cl, uint32_t nReads, unsigned long long *need)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nReads) return;
    unsigned long long v = 0;
    if (r < nReads) { const uint32_t b = cs[r], e = cs[r + 1]; if (e - b >= 2) for (uint32_t c = b; c < e; c++) v += 2ull * ((unsigned long long)cl[c].n_ops + 1ull); }
    need[r] = v;
}
__global__ void __launch_bounds__(64) k_oqc_run(OqcArgs A)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nReads) return;
    const uint32_t b = A.cs[r], n = A.cs[r + 1] - b;
    uint32_t m = 0, nops = 0; int primary = 0;
    if (n) {
        yoqc::Scratch S;
        S.keys = A.keys + b; S.stack = A.stack + 4ull * b + 8ull * r; S.nodes = A.nodes + b; S.prim = A.prim + b; S.pa = A.pa + b; S.pfxOff = A.pfxOff + b; S.path = A.path + b;
        S.pool = A.pool + A.poolOff[r]; S.push = A.push + b;
        const uint32_t o = A.readOff[r]; const int qlen = (int)(A.readOff[r + 1] - o);
        m = (uint32_t)yoqc::run(A.P, A.G, A.cl + b, (int)n, A.ops, qlen, A.fwd + o, S, A.out + b, &primary);
        for (uint32_t k = 0; k < m; k++) nops += A.cl[b + (uint32_t)A.out[b + k].clump].n_ops;
    }
    A.outCnt[r] = m; A.outOpsCnt[r] = nops; A.primCnt[r] = (uint32_t)primary;
}
// the printed clumps of read r, in print order, with their ops copied behind one another: out clump k of the read = fClumps[outStart[r] + k]
__global__ void k_oqc_gather(OqcArgs A, const uint32_t *outStart, const uint32_t *opsStart, ygpu_out_clump *fClumps, uint32_t *fOps)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.nReads) return;
    const uint32_t b = A.cs[r], m = A.outCnt[r]; uint32_t d = outStart[r], od = opsStart[r];
    for (uint32_t k = 0; k < m; k++) {
        const yoqc::OutRec o = A.out[b + k]; ygpu_clump c = A.cl[b + (uint32_t)o.clump];
        const uint32_t *src = A.ops + c.op_start;
        for (uint32_t i = 0; i < c.n_ops; i++) fOps[od + i] = src[i];
        c.op_start = od; od += c.n_ops;
        ygpu_out_clump f; f.c = c; f.status = o.status; f.mapQuality = o.mapQuality; f.numSecondaries = o.numSecondaries; f.matchedPrimary = o.matchedPrimary; f.primaryCount = (uint16_t)A.primCnt[r];
        fClumps[d + k] = f;
    }
}
