// dp_stage.h -- glue for the stage-level DP entry (ygpu_dp_batch) when it runs the PRODUCTION lane kernels: k_ext_rows / k_ext_trace for the X-drop
// extensions, the pure-diagonal shortcut of k_p1_joints and k_gap_lanes<16|32> / k_gap_wave for the gap fills -- exactly the kernels, launch shapes and
// data layouts ygpu_run uses at -BW 5 -- so that every findAffineGapScore call (SW.cpp:798-1208, wrappers :462-547) can be compared one by one.
#pragma once
#include "split_lanes.h"

// classify a gap problem as k_p1_joints does (phase_lanes.h): pure diagonal or DP; sort key = (class, strip width, rows)
__global__ void k_dp_classify(DevParams P, const uint8_t *bases, const uint8_t *fwd, const uint8_t *rev, JointRec *joints, uint32_t n, uint32_t *keys, uint32_t *diagOps)
{
    YD_HIGH_PRIO();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    JointRec j = joints[t];
    YD_GLOBAL const uint8_t *q = toGlobal((j.flags & 1u) ? rev : fwd) + j.qBase; YD_GLOBAL const uint8_t *gB = toGlobal(bases);
    auto refAt = [&](uint32_t off) -> uint32_t { const uint32_t b = gB[off >> 1]; return (off & 1u) ? (b & 15u) : (b >> 4); };
    const int qGap = j.qGap, rGap = j.rGap;
    const bool banded = (j.flags & 2u) != 0;
    j.kind = JK_DP; uint32_t key = YD_JKEY_NONE;
    if (qGap == rGap && qGap > 0) {
        int mm = 0, runs = 0, pc = -1;
        for (int k = 0; k < qGap; k++) { const int c = (uint32_t)q[(int)j.nsqo + k] != refAt(j.nsro + (uint32_t)k); mm += c; runs += c != pc; pc = c; }
        if (mm * (P.MS + P.RC) <= P.MS + 2 * (P.GO + P.GE)) { j.kind = JK_DIAG; j.score = P.MS * (qGap - mm) - P.RC * mm; j.nOps = (uint16_t)runs; }
    }
    if (j.kind == JK_DP) key = gapJointKey(P, banded, qGap, rGap);
    joints[t] = j; keys[t] = key; diagOps[t] = j.kind == JK_DIAG ? (uint32_t)j.nOps : 0u;
}

// results of the lane kernels -> ygpu_dp_result + ops in list order with the public op codes
__global__ void k_dp_gather_ext(const ExtProb *probs, const ExtRes *res, const uint32_t *extOps, const uint32_t *outOff, const uint32_t *dst, uint32_t n,
                                ygpu_dp_result *out, uint32_t *outOps)
{
    YD_HIGH_PRIO();
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const ExtRes r = res[p]; ygpu_dp_result o; o.score = 0; o.addedQLen = o.addedRLen = 0; o.op_start = outOff[p]; o.n_ops = 0;
    if (r.score > 0) {
        const bool rv = (probs[p].flags & XP_REV) != 0; const uint32_t *src = extOpsPtr(extOps, r); const char codes[4] = {'M', 'R', 'D', 'I'};
        o.score = r.score; o.addedQLen = (uint16_t)r.maxi; o.addedRLen = (uint16_t)(r.maxi + (r.maxj - YD_LBAND)); o.n_ops = r.nOps;
        for (uint32_t k = 0; k < r.nOps; k++) { const uint32_t op = src[rv ? r.nOps - 1u - k : k];
            outOps[o.op_start + k] = ((uint32_t)(uint8_t)codes[opCode(op) & 3] << 16) | (uint32_t)opLen(op); }
    }
    out[dst[p]] = o;
}
__global__ void k_dp_gather_gap(DevParams P, const uint8_t *bases, const uint8_t *fwd, const uint8_t *rev, const JointRec *joints, const uint32_t *gapOps, const uint32_t *outOff,
    const uint32_t *dst,
                                uint32_t n, ygpu_dp_result *out, uint32_t *outOps)
{
    YD_HIGH_PRIO();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const JointRec j = joints[t]; ygpu_dp_result o; o.score = j.score; o.addedQLen = o.addedRLen = 0; o.op_start = outOff[t]; o.n_ops = j.nOps;
    const char codes[4] = {'M', 'R', 'D', 'I'};
    if (j.kind == JK_DIAG) {                                                 // the list k_p1_assemble writes for a pure diagonal
        YD_GLOBAL const uint8_t *q = toGlobal((j.flags & 1u) ? rev : fwd) + j.qBase; YD_GLOBAL const uint8_t *gB = toGlobal(bases);
        int pc = -1, pl = 0; uint32_t w = o.op_start;
        for (int k = 0; k < (int)j.qGap; k++) {
            const uint32_t off = j.nsro + (uint32_t)k, b = gB[off >> 1], rc = (off & 1u) ? (b & 15u) : (b >> 4);
            const int c = (uint32_t)q[(int)j.nsqo + k] == rc ? OP_M : OP_R;
            if (c == pc) pl++; else { if (pc >= 0) outOps[w++] = ((uint32_t)(uint8_t)codes[pc] << 16) | (uint32_t)pl; pc = c; pl = 1; }
        }
        if (pc >= 0) outOps[w++] = ((uint32_t)(uint8_t)codes[pc] << 16) | (uint32_t)pl;
    } else for (uint32_t k = 0; k < j.nOps; k++) { const uint32_t op = gapOps[j.opsOff + k];
        outOps[o.op_start + k] = ((uint32_t)(uint8_t)codes[opCode(op) & 3] << 16) | (uint32_t)opLen(op); }
    out[dst[t]] = o;
}
