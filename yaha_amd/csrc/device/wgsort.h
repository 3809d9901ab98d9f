// wgsort.h -- the workgroup's own stable sort of A2's hits on the 32 diagonal bits: least-significant-digit radix passes of 8 bits over LDS, written for 64-lane waves.
//
// The items sit in the wave-striped arrangement (wave w holds positions [w * 64 * IPT, (w + 1) * 64 * IPT), item k of lane l = position k * 64 + l of them): the order of
// the hits IS the position, so a row of 64 lanes is 64 consecutive hits.  One pass:
//   rank    a lane's rank among the hits of its wave with the same digit, counters per wave ([wave][digit]; the rows of a wave go one after the other, and the LDS
//           operations of one wave execute in order).  Two ways, both compiled:
//             ATOMIC   one ds_add_rtn per lane.  The lanes of one instruction that meet on an address are served in ascending lane order on this hardware -- which is what
//                      makes the pass stable -- but no manual promises it: every sorted hit is therefore CHECKED against the hit before it as it leaves LDS
//                      ((diagonal, query offset) strictly ascending: no two hits of a segment are equal), and a batch that fails is sorted again with the other
//                      ranking, which then stays (stage_seed.hip).
//             ballots  the lanes of the row with the same digit by eight ballots (the set of lanes that differ in some bit is OR-ed up), the number of them below the
//                      lane (v_mbcnt) and their count; the counter is read by all of them and moved on by the lowest.  40 instructions a hit and pass more than ATOMIC.
//   scan    digit-major, wave-minor exclusive sums of the NW x 256 counters: thread d adds up the NW counters of digit d, the 256 sums are scanned by four waves.
//   move    rank = base[wave][digit] + rank in the wave; diagonals and payloads (the 15-bit query offset) go through LDS and come back in the wave-striped arrangement --
//           after the last pass striped over the workgroup (item k of thread t = position k * BS + t), which is what a coalesced store wants.
// LDS is at most 64 KB: where 6 bytes a hit fit, diagonals and payloads move together (FUSED: two barriers a pass less), where the counters fit beside them they are
// zeroed while the hits come back (SEPCNT: three less); the largest shapes (512 x 24 / 28 / 31) move the payloads through the area of the diagonals, the counters alias its
// head.  A hit costs two registers: the diagonal, and its rank above the payload.
// Only the valid positions (< len) take part: nothing is padded, nothing sorts "last".
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

template <unsigned BS, unsigned IPT, bool ATOMIC>
struct WgSort {
    static constexpr unsigned NW = BS / 64u, N = BS * IPT, CNT = NW * 256u;
    static_assert(BS % 64u == 0 && BS >= 128u, "whole waves, and at least one thread per two digits");
    static_assert(CNT <= N && N <= 16384u, "the counters may alias the head of the exchange area; a rank has 14 bits");
#ifndef YD_SORT_LDS_MAX
#define YD_SORT_LDS_MAX 65536u
#endif
    static constexpr bool FUSED = 6u * N + 16u <= YD_SORT_LDS_MAX;
    static constexpr bool SEPCNT = (FUSED ? 6u : 4u) * N + 4u * CNT + 16u <= YD_SORT_LDS_MAX;
    struct Storage { uint32_t x[N]; uint16_t p[FUSED ? N : 2u]; uint32_t c[SEPCNT ? CNT : 4u]; uint32_t wt[4]; };

    // the lanes of the row (among vlo | vhi) whose digit equals this lane's
    static __device__ __forceinline__ void match(uint32_t d, uint32_t vlo, uint32_t vhi, uint32_t &plo, uint32_t &phi)
    {
        uint32_t xlo = 0, xhi = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int bm = __builtin_amdgcn_sbfe((int)d, b, 1);                // 0 / -1
            const unsigned long long m = __ballot(bm != 0);
            xlo |= (uint32_t)m ^ (uint32_t)bm; xhi |= (uint32_t)(m >> 32) ^ (uint32_t)bm;
        }
        plo = vlo & ~xlo; phi = vhi & ~xhi;
    }

    // dg: the diagonals, rq: the payloads (16 bits; the upper half is the sort's), wave-striped.  The sorted hits are handed to emit(position, diagonal, payload), thread t
    // those of the positions k * BS + t (a coalesced store).  Every thread of the workgroup calls it (barriers inside).
    // ATOMIC: bad is raised where a sorted hit is not above the hit before it in (diagonal, payload) -- the check the lane order of the LDS atomics is trusted under.
    template <class Emit>
    static __device__ __forceinline__ void sort(uint32_t (&dg)[IPT], uint32_t (&rq)[IPT], uint32_t len, Storage &S, Emit emit, bool &bad)
    {
        const uint32_t t0 = threadIdx.x, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t0 >> 6));
        uint16_t *const x16 = (uint16_t *)S.x;
        uint32_t *const cnt = SEPCNT ? S.c : S.x;
        for (uint32_t e = t0; e < CNT; e += BS) cnt[e] = 0u;
        __syncthreads();
#pragma unroll 1
        for (uint32_t shift = 0; shift < 32u; shift += 8u) {
            // (opaque per pass: what follows from the thread's number and the length -- the rows' lane masks, every item's position and LDS address, the addresses of the
            // final store -- is worked out where it is used, not once before the loop and then kept in a hundred registers across the passes)
            uint32_t lenv = (uint32_t)__builtin_amdgcn_readfirstlane((int)len), t = t0, wv = w; asm volatile("" : "+s"(lenv), "+v"(t), "+s"(wv));
            const uint32_t lane = t & 63u, wbase = wv * (64u * IPT);
            uint32_t *const myCnt = cnt + wv * 256u;
            // ---- rank inside the wave ----
#pragma unroll
            for (unsigned k = 0; k < IPT; k++) {
                const uint32_t row = wbase + k * 64u;
                if (row < lenv) {                                              // wave-uniform
                    const bool valid = row + lane < lenv;
                    const uint32_t d = (dg[k] >> shift) & 0xFFu;
                    uint32_t r = 0;
                    if (ATOMIC) { if (valid) r = atomicAdd(&myCnt[d], 1u); }
                    else {
                        const unsigned long long vm = __ballot(valid);
                        uint32_t plo, phi; match(d, (uint32_t)vm, (uint32_t)(vm >> 32), plo, phi);
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                        const uint32_t n = (uint32_t)__builtin_popcount(plo) + (uint32_t)__builtin_popcount(phi);
                        const uint32_t old = myCnt[d];
                        if (valid && below == 0u) myCnt[d] = old + n;
                        r = old + below;
                    }
                    rq[k] = __builtin_amdgcn_perm(r, rq[k], 0x05040100u);      // rank << 16 | payload
                }
            }
            __syncthreads();
            // ---- scan: digit-major, wave-minor ----
            {
                constexpr unsigned DPT = BS >= 256u ? 1u : 256u / BS;           // digits a thread (BS = 128: two)
                uint32_t pre[DPT][NW], tot = 0;
                const bool scanner = t < 256u / DPT;
                if (scanner) {
#pragma unroll
                    for (unsigned j = 0; j < DPT; j++)
#pragma unroll
                        for (unsigned v = 0; v < NW; v++) { pre[j][v] = tot; tot += cnt[v * 256u + t * DPT + j]; }
                }
                uint32_t incl = tot;
#pragma unroll
                for (int s = 1; s < 64; s <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, s, 64); if ((int)lane >= s) incl += y; }
                if (lane == 63u && wv < 4u) S.wt[wv] = incl;
                __syncthreads();
                if (scanner) {
                    uint32_t off = incl - tot;
#pragma unroll
                    for (unsigned v = 0; v < 3u; v++) if (v < wv) off += S.wt[v];
#pragma unroll
                    for (unsigned j = 0; j < DPT; j++)
#pragma unroll
                        for (unsigned v = 0; v < NW; v++) cnt[v * 256u + t * DPT + j] = off + pre[j][v];
                }
            }
            __syncthreads();
            // ---- the rank in the workgroup ----
            uint32_t sh2 = shift; asm volatile("" : "+s"(sh2));                 // (opaque: the digits are extracted again, not kept in IPT registers across the scan)
#pragma unroll
            for (unsigned k = 0; k < IPT; k++) if (wbase + k * 64u < lenv) rq[k] += myCnt[(dg[k] >> sh2) & 0xFFu] << 16;
            if (!SEPCNT) __syncthreads();                                      // (the counters alias what the hits are written to)
            const bool lastPass = shift == 24u;
            // ---- the hits move ----
#pragma unroll
            for (unsigned k = 0; k < IPT; k++) if (wbase + k * 64u + lane < lenv) { const uint32_t r = rq[k] >> 16; S.x[r] = dg[k]; if (FUSED) S.p[r] = (uint16_t)rq[k]; }
            __syncthreads();
            if (FUSED) {
                if (lastPass) {                                                // straight from LDS to the caller: no registers in between
#pragma unroll 4
                    for (unsigned k = 0; k < IPT; k++) {
                        const uint32_t pos = k * BS + t;
                        if (pos < lenv) {
                            const uint32_t d = S.x[pos], q = S.p[pos]; emit(pos, d, q);
                            if (ATOMIC && pos > 0u) { const uint32_t pd = S.x[pos - 1u], pq = S.p[pos - 1u]; bad |= !(pd < d || (pd == d && pq < q)); }
                        }
                    }
                    return;
                }
#pragma unroll
                for (unsigned k = 0; k < IPT; k++) { const uint32_t pos = wbase + k * 64u + lane; dg[k] = S.x[pos]; rq[k] = S.p[pos]; }
            } else {
                if (!lastPass) {
#pragma unroll
                    for (unsigned k = 0; k < IPT; k++) dg[k] = S.x[wbase + k * 64u + lane];
                } else {
#pragma unroll
                    for (unsigned k = 0; k < IPT; k++) dg[k] = S.x[k * BS + t];
                }
                __syncthreads();
#pragma unroll
                for (unsigned k = 0; k < IPT; k++) if (wbase + k * 64u + lane < lenv) x16[rq[k] >> 16] = (uint16_t)rq[k];
                // (the order check below: the diagonal before a wave's first lane is its neighbour wave's last -- parked in the upper half of the area, which the
                // payloads leave free)
                uint32_t *const edge = S.x + (N - NW * IPT);
                static_assert(NW * IPT <= N / 2u, "the parked diagonals lie behind the payloads");
                if (ATOMIC && lastPass && lane == 63u) {
#pragma unroll
                    for (unsigned k = 0; k < IPT; k++) edge[wv * IPT + k] = dg[k];
                }
                __syncthreads();
                if (lastPass) {
#pragma unroll
                    for (unsigned k = 0; k < IPT; k++) {
                        const uint32_t pos = k * BS + t;
                        const uint32_t q = x16[pos < lenv ? pos : 0u];
                        if (pos < lenv) emit(pos, dg[k], q);
                        if (ATOMIC) {
                            // the hit before: lane - 1's (the wave before's last lane's; for the workgroup's first thread the last thread's previous item)
                            uint32_t pd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dg[k], 0x138, 0xF, 0xF, false);      // wave_shr:1
                            if (lane == 0u) pd = wv > 0u ? edge[(wv - 1u) * IPT + k] : (k > 0u ? edge[(NW - 1u) * IPT + (k - 1u)] : 0u);
                            if (pos < lenv && pos > 0u) { const uint32_t pq = x16[pos - 1u]; bad |= !(pd < dg[k] || (pd == dg[k] && pq < q)); }
                        }
                        if (k & 1u) __builtin_amdgcn_sched_barrier(0);         // (two at a time: else every 64-bit address and value of the thread is built before the first store)
                    }
                    return;
                }
#pragma unroll
                for (unsigned k = 0; k < IPT; k++) rq[k] = x16[wbase + k * 64u + lane];
            }
            // ---- the counters of the next pass ----
            if (!SEPCNT) __syncthreads();                                      // (every thread has read its hits back)
            for (uint32_t e = t; e < CNT; e += BS) cnt[e] = 0u;                // (SEPCNT: every thread read its bases before the barrier behind the move)
            __syncthreads();
        }
    }
};
