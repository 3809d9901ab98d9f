// seed.h -- stages A1 + A2 on the device: k-mer hashing of both strands, hash-table lookup, hit enumeration,
// (read, strand, diagonal, query offset) sort and coalescing into fragments (reference Query.c:341-412,
// QueryMatch.c:52-121, QueryHeap.inl:70-134), then diagonal-region boundaries (QueryMatch.c:146-158).
// The reference merges the per-k-mer hit lists through a binary heap; its output is just the sorted multiset of
// (diag << 32 | qo) keys, so a device radix sort of the batch-wide keys (read*2+strand in the top bits) gives the
// identical fragment array.  HBM-bound integer/byte work: coalesced streams + random 8-byte table probes.
#pragma once
#include "common.h"
#include "lookback.h"

__constant__ unsigned char kComp4[16] = {2, 3, 0, 1, 4, 12, 7, 6, 9, 8, 15, 11, 5, 13, 14, 10};   // fourBitCompCodes, Math.c:156

// reverse-complement codes (Query.c:161-167): one workgroup per read
__global__ void k_revcomp(const uint8_t *fwd, uint8_t *rev, const uint32_t *readOff, uint32_t nReads)
{
    const uint32_t r = blockIdx.x; if (r >= nReads) return;
    const uint32_t o = readOff[r], n = readOff[r + 1] - o;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) rev[o + k] = kComp4[fwd[o + n - 1 - k] & 0xF];
}

// the batch's codes packed two to the byte (high nibble = the even offset, as .nib2 packs the reference): what k_ext_rows_pk's query windows are refilled from -- a dword
// then holds eight codes, like a dword of reference bases, and a lane comes back to a 64-byte sector for 128 rows instead of 64
__global__ void k_pack4(const uint8_t *codes, uint8_t *packed, uint32_t nBytesOut, uint32_t nCodes)
{
    YD_HIGH_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // (an odd number of codes: the last byte's low nibble has no code -- 15, never the slack byte behind the codes, which nobody wrote)
    if (i < nBytesOut) packed[i] = (uint8_t)(((codes[2u * i] & 15u) << 4) | (2u * i + 1u < nCodes ? (codes[2u * i + 1u] & 15u) : 15u));
}

// Once per index: a bit per k-mer (its low 22 bits: exact up to -L 11, a filter above) that has a reference offset below 2^15 anywhere in its list -- the only
// lists whose first offset can be below a query offset (QueryMatch.c:56-69).  One pass over the offsets; the few that qualify find their k-mer by bisection.
#define YD_LOW_BITS (1u << 22)
__global__ void __launch_bounds__(256) k_low_offsets(const uint32_t *SO, uint32_t nKmerSlots /* 4^L */, const uint32_t *ROA, uint32_t total, uint32_t *low)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (uint64_t)gridDim.x * blockDim.x) {
        if (ROA[j] >= 32768u) continue;
        uint32_t lo = 0, hi = nKmerSlots;                                          // largest h with SO[h] <= j
        while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (SO[mid] <= (uint32_t)j) lo = mid; else hi = mid; }
        atomicOr(&low[(lo & (YD_LOW_BITS - 1u)) >> 5], 1u << (lo & 31u));
    }
}

// A1: one workgroup per (read, strand); thread per k-mer start.  posS/posC are indexed by kmerOff[rs] + i.
__global__ void k_kmer_lookup(DevParams P, DevBatch B, const uint32_t *SO, const uint32_t *ROA, const uint32_t *low, const uint32_t *kmerOff,
                              uint32_t *posS, uint32_t *posC, uint32_t *posRsI, unsigned int *parts)
{
    YD_HIGH_PRIO();
    const uint32_t rs = blockIdx.x; const uint32_t read = rs >> 1;
    const uint32_t o = B.readOff[read]; const int qlen = (int)(B.readOff[read + 1] - o);
    const int L = P.wordLen, nPos = qlen - L + 1;
    if (nPos <= 0) return;                                                        // (block-uniform)
    const uint8_t *codes = ((rs & 1u) ? B.rev : B.fwd) + o;
    const uint32_t base = kmerOff[rs];
    unsigned lookups = 0;
    for (int i = threadIdx.x; i < nPos; i += blockDim.x) {
        uint32_t h = 0; bool bad = false;
        for (int k = 0; k < L; k++) { uint32_t c = codes[i + k]; bad |= c > 3; h = (h << 2) | (c & 3u); }
        uint32_t s = 0, cnt = 0;
        if (!bad) {                                                               // Query.c:391-405
            s = SO[h]; cnt = SO[h + 1] - s; lookups++;
            if (cnt > (uint32_t)P.maxHits) cnt = 0;
            if (cnt) {                                                            // QueryMatch.c:56-69: wrapping diagonals / read past the list
                // (the walk only starts when the list's first offset is below i < 2^15: k_low_offsets marked the k-mers that have such an offset, for all others
                // the list is not touched here -- it was a second scattered line per k-mer in a kernel that runs at the chip's rate of scattered lines)
                uint32_t w = 0;
                if ((low[(h & (YD_LOW_BITS - 1u)) >> 5] >> (h & 31u)) & 1u)
                    while (s + w < P.totalMatches && ROA[s + w] < (uint32_t)i) w++;
                uint32_t eff = (w < cnt) ? cnt : w + 1;
                if (s + eff > P.totalMatches) eff = P.totalMatches - s;
                cnt = eff;
            }
        }
        posS[base + i] = s; posC[base + i] = cnt; posRsI[base + i] = (rs << 15) | (uint32_t)i;
    }
    // the work counter: one atomic per workgroup, spread over 1 024 words that k_sum_parts adds up (a single L2 word takes ~88 atomics per microsecond; one
    // per wave on the counter itself was most of this kernel's time on short reads: 262 k waves for 65 536 x 100 bp)
    __shared__ unsigned sLook;
    if (threadIdx.x == 0) sLook = 0;
    __syncthreads();
    lookups = (unsigned)waveSumI((int)lookups);
    if ((threadIdx.x & 63u) == 0 && lookups) atomicAdd(&sLook, lookups);
    __syncthreads();
    if (threadIdx.x == 0 && sLook) atomicAdd(&parts[blockIdx.x & 1023u], sLook);
}
// counter += sum of its 1 024 partial sums (one workgroup of 1 024 threads)
__global__ void __launch_bounds__(1024) k_sum_parts(const unsigned int *parts, unsigned long long *counter)
{
    YD_HIGH_PRIO();
    __shared__ unsigned long long sSum;
    if (threadIdx.x == 0) sSum = 0;
    __syncthreads();
    unsigned v = (unsigned)waveSumI((int)parts[threadIdx.x]);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&sSum, (unsigned long long)v);
    __syncthreads();
    if (threadIdx.x == 0 && sSum) atomicAdd(counter, sSum);
}

// A2a: thread per hit -> 64-bit key  rs(17) | diag(32) | qo(15).  A block owns 1024 consecutive hits; they belong to at most 1025 consecutive k-mers with hits,
// whose offsets, list starts and (read, strand, offset) words are staged in LDS.  The k-mer of every hit without a search: each k-mer of the window marks
// the slot of its first hit, a running maximum over the 1024 slots carries the mark forward (four slots per thread, one scan per wave, four waves).  The first
// k-mer of every block comes from k_expand_starts (one thread per k-mer writes the blocks that begin inside its list; a binary search over all k-mers by one
// thread per block was most of a block's life: 25 dependent loads).  Each thread then has four independent list reads in flight.
#define YD_EXPAND_HITS 1024
__global__ void k_expand_starts(const uint32_t *hitOff, uint32_t nKmers, uint32_t *blockG0)
{
    YD_HIGH_PRIO();
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nKmers) return;
    const uint32_t a = hitOff[g], b = hitOff[g + 1u];
    if (g == 0u) blockG0[(hitOff[nKmers] + (uint32_t)YD_EXPAND_HITS - 1u) / (uint32_t)YD_EXPAND_HITS] = nKmers;       // behind the last block: the end of the k-mers
    for (uint32_t blk = (a + (uint32_t)YD_EXPAND_HITS - 1u) / (uint32_t)YD_EXPAND_HITS; (unsigned long long)blk * YD_EXPAND_HITS < b; blk++) blockG0[blk] = g;
}
__global__ void __launch_bounds__(256) k_expand_hits(const uint32_t *ROA, const uint32_t *posS, const uint32_t *hitOff, const uint32_t *posRsI, const uint32_t *blockG0,
    uint32_t nKmers, uint32_t nHits, unsigned long long *keys)
{
    YD_HIGH_PRIO();
    __shared__ uint32_t sOff[YD_EXPAND_HITS + 2], sS[YD_EXPAND_HITS + 2], sRsI[YD_EXPAND_HITS + 2]; __shared__ __attribute__((aligned(16))) uint32_t sK[YD_EXPAND_HITS];
        __shared__ uint32_t sWave[4];
    const uint32_t t0 = blockIdx.x * YD_EXPAND_HITS, tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    // the window: from this block's first k-mer to the next block's (k-mers without hits in between included), at most 1026 offsets (hitOff has nKmers + 1 entries)
    const uint32_t g0 = blockG0[blockIdx.x], span = min(min(nKmers + 1u - g0, blockG0[blockIdx.x + 1u] + 2u - g0), (uint32_t)YD_EXPAND_HITS + 2u);
    for (uint32_t k = tid; k < span; k += 256u) { sOff[k] = hitOff[g0 + k]; sS[k] = posS[g0 + k]; sRsI[k] = posRsI[g0 + k]; }
    *(uint4 *)&sK[4u * tid] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (uint32_t k = 1u + tid; k + 1u < span; k += 256u) { const uint32_t o = sOff[k]; if (o > t0 && o - t0 < (uint32_t)YD_EXPAND_HITS && sOff[k + 1u] > o) sK[o - t0] = k; }
    __syncthreads();
    {
        const uint4 v = *(const uint4 *)&sK[4u * tid];
        const uint32_t a0 = v.x, a1 = max(a0, v.y), a2 = max(a1, v.z), a3 = max(a2, v.w);
        const uint32_t incl = waveInclMaxU(a3), excl = (uint32_t)laneUp1((int)incl, 0);
        if (lane == 63u) sWave[w] = incl;
        __syncthreads();
        uint32_t carry = excl;
        for (uint32_t k = 0; k < w; k++) carry = max(carry, sWave[k]);
        *(uint4 *)&sK[4u * tid] = make_uint4(max(carry, a0), max(carry, a1), max(carry, a2), max(carry, a3));
    }
    __syncthreads();
    const uint32_t endOff = g0 + span - 1u >= nKmers ? nHits : sOff[span - 1u];          // hits from here on belong to k-mers behind the window (long runs of k-mers without hits)
    uint32_t at[4], rsi[4]; bool in[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t p = tid + 256u * (uint32_t)i, t = t0 + p; in[i] = t < nHits; at[i] = 0; rsi[i] = 0;
        if (!in[i]) continue;
        const uint32_t k = sK[p];
        uint32_t off = sOff[k], s = sS[k]; rsi[i] = sRsI[k];
        if (t >= endOff) {
            uint32_t l2 = g0 + span - 1u, h2 = nKmers;                                  // largest g with hitOff[g] <= t
            while (h2 - l2 > 1) { const uint32_t mid = (l2 + h2) >> 1; if (hitOff[mid] <= t) l2 = mid; else h2 = mid; }
            off = hitOff[l2]; s = posS[l2]; rsi[i] = posRsI[l2];
        }
        at[i] = s + (t - off);
    }
    uint32_t roff[4];
#pragma unroll
    for (int i = 0; i < 4; i++) roff[i] = in[i] ? ROA[at[i]] : 0u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (!in[i]) continue;
        const uint32_t q = rsi[i] & 0x7FFFu, rs = rsi[i] >> 15;
        keys[t0 + tid + 256u * (uint32_t)i] = ((unsigned long long)rs << 47) | ((unsigned long long)(uint32_t)(roff[i] - q) << 15) | (unsigned long long)q;
    }
}

// segment offsets of the hit sort: the hits of (read, strand) rs are [hitOff[kmerOff[rs]], hitOff[kmerOff[rs + 1]])
__global__ void k_seg_offsets(const uint32_t *kmerOff, const uint32_t *hitOff, uint32_t nSeg, uint32_t *segOff)
{
    YD_HIGH_PRIO();
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= nSeg) segOff[s] = hitOff[kmerOff[s]];
}

// A2b: fragment heads.  A hit starts a new fragment when (read, strand) or diagonal changes or the k-mer neither overlaps nor abuts the previous one: a query
// offset more than one word beyond the previous hit's (QueryMatch.c:84-121).
// A fragment of ONE hit that is alone in its region (QueryMatch.c:146-158: the fragments before and after it are of another (read, strand) or more than maxGap
// diagonals away) can only become a clump if one word is a whole match (refLen = wordLen >= minMatch, QueryMatch.c:281-290).  With the defaults (15 < 25) it
// is dead on arrival, and on a large genome these chance hits are most fragments (3.1 Gbp: 128 M fragments a batch, ~100 M of them dead): they are counted
// and not written.  Dropping them changes nothing for the others: the diagonals are sorted, so the fragments on either side of a dropped run are further
// apart than the dropped fragment was from them -- every region boundary stays where it was.  maxGapDrop < 0: keep every fragment (ygpu_seed_join).
// A2b in one pass over the sorted keys: head flags, their batch-wide exclusive scan (= the fragment index of every hit) and the fragment records.  A workgroup
// of 16 waves owns a tile of 8 192 consecutive hits, each of its waves 8 rows of 64: a hit's neighbours are in the neighbouring lanes (or the edge lanes of the
// rows above and below), the rank of a head inside the tile comes from the ballots of the 16 x 8 (wave, row) groups -- their 128 counts scanned by one wave, two
// per lane -- and the tile's offset from the tiles before it by decoupled look-back: a tile publishes its own count as soon as it has it, then adds up the
// published counts behind it until it meets a tile that already knows its inclusive prefix (tile states: one 64-bit word, status in the high half so that value
// and status arrive together).  The tile IS the ticket the workgroup drew when it started (tileTicket): every earlier ticket holder is therefore resident (or
// done) and publishes before it waits for anything -- no assumption about the order in which workgroups are dispatched.  Before: the library's scan over the
// flags (2.6 GB of keys read, 1.3 GB of indices written) and a build kernel that read both again.  Records beyond cap are not written: the caller reads *total
// and comes back with room.
#ifndef YD_FRAG_BS
#define YD_FRAG_BS 1024
#endif
#ifndef YD_FRAG_IPT
#define YD_FRAG_IPT 8
#endif
#define YD_FRAG_TILE (YD_FRAG_BS * YD_FRAG_IPT)
// What links a hit to the one before it in the sorted order, from the two keys: `cont` -- the same fragment goes on (same (read, strand) and diagonal, the k-mer overlaps
// or abuts: QueryMatch.c:84-121) -- and `near` -- same (read, strand), at most maxGap diagonals apart (QueryMatch.c:146-158).  A hit is a fragment's head where cont is
// false, its last hit where the NEXT hit's cont is false, and a dead single (see above) where it is both and neither its own near nor the next hit's holds.
static_assert((YD_FRAG_IPT * (YD_FRAG_BS / 64)) % 64 == 0 && YD_FRAG_IPT <= 16, "k_frag_scan_build: wave 0 scans IPT x waves counts, a whole number per lane");
// The classes of a row of 64 hits are LANE MASKS in scalar registers: the two links of every hit come from six compares, everything after that -- the shift by one hit
// (a 64-bit shift with the next row's bit 0 coming in), head / last / dead, the counts -- is scalar arithmetic, and the masks come back as the conditions of the stores.
// (Round 3's form kept three class bits a hit in vector registers and worked out both neighbours of every hit: 132 vector instructions a hit, 668 M a step.)
// (80 scalar registers: with more than 96 -- the compiler takes what it is given -- eight waves no longer fit a SIMD and the workgroup of sixteen is alone on its CU)
__global__ void __launch_bounds__(YD_FRAG_BS) __attribute__((amdgpu_num_sgpr(80)))
k_frag_scan_build(const unsigned long long *keys, uint32_t nHits, int wordLen, int maxGapDrop, DevFrag *frags, uint32_t cap, unsigned long long *tileState,
                  unsigned int *total /* [0] the count, [1] raised when the look-back gave up */, unsigned int *deadParts /* [1024] partial counts of dropped fragments */)
{
    YD_HIGH_PRIO();
    typedef unsigned long long u64;
    constexpr int NW = YD_FRAG_BS / 64;
    __shared__ uint32_t sCnt[YD_FRAG_IPT * NW]; __shared__ uint32_t sPrefix; __shared__ unsigned sDead;
    const uint32_t tile = tileTicket(tileState + gridDim.x, &sPrefix), base = tile * (uint32_t)YD_FRAG_TILE, t = threadIdx.x, lane = t & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    if (t == 0) sDead = 0;
    // wave w owns the hits wbase + k * 64 + lane: all its loads are issued together (its rows and the two hits around the wave's range)
    const uint32_t wbase = base + w * (uint32_t)(64 * YD_FRAG_IPT);
    uint32_t klo[YD_FRAG_IPT], khi[YD_FRAG_IPT];
    {
        unsigned long long key[YD_FRAG_IPT];
#pragma unroll
        for (int k = 0; k < YD_FRAG_IPT; k++) { const uint32_t idx = wbase + (uint32_t)k * 64u + lane; key[k] = idx < nHits ? keys[idx] : 0ull; }
#pragma unroll
        for (int k = 0; k < YD_FRAG_IPT; k++) { klo[k] = (uint32_t)key[k]; khi[k] = (uint32_t)(key[k] >> 32); }
    }
    unsigned long long edge = 0ull;                                            // lane 0: the hit before the wave's range; lane 63: the one behind it
    if (lane == 0u && wbase > 0u && wbase <= nHits) edge = keys[wbase - 1u];
    if (lane == 63u && wbase + (uint32_t)(64 * YD_FRAG_IPT) < nHits) edge = keys[wbase + (uint32_t)(64 * YD_FRAG_IPT)];
    const uint32_t elo = (uint32_t)edge, ehi = (uint32_t)(edge >> 32), wl = (uint32_t)wordLen, G = (uint32_t)maxGapDrop;
    const bool drop = maxGapDrop >= 0;
    // the links of one row as masks (bit l = the hit of lane l against the hit before it), straight from the compares (v_cmp writes a lane mask; no predicate is
    // ever a vector register); the valid lanes of a row and "has a hit before it" are scalar arithmetic
    enum { CMP_EQ = 32, CMP_ULT = 36, CMP_ULE = 37 };                           // (LLVM's integer predicates)
    auto rowLinks = [&](int k, u64 &cM, u64 &nM, u64 &vM) {
        const uint32_t row = wbase + (uint32_t)k * 64u;
        const uint32_t nv = row < nHits ? (nHits - row < 64u ? nHits - row : 64u) : 0u;
        vM = nv >= 64u ? ~0ull : ((1ull << nv) - 1ull);
        const u64 hasM = row == 0u ? (vM & ~1ull) : vM;
        const uint32_t flo = (uint32_t)__builtin_amdgcn_readlane((int)(k > 0 ? klo[k > 0 ? k - 1 : 0] : elo), k > 0 ? 63 : 0),
                       fhi = (uint32_t)__builtin_amdgcn_readlane((int)(k > 0 ? khi[k > 0 ? k - 1 : 0] : ehi), k > 0 ? 63 : 0);
        const uint32_t alo = (uint32_t)laneUp1((int)klo[k], (int)flo), ahi = (uint32_t)laneUp1((int)khi[k], (int)fhi), blo = klo[k], bhi = khi[k];
        const uint32_t xh = ahi ^ bhi, xl = alo ^ blo;
        const uint32_t da = (uint32_t)__builtin_amdgcn_alignbit(ahi, alo, 15), db = (uint32_t)__builtin_amdgcn_alignbit(bhi, blo, 15);
        // (the keys ascend: inside one (read, strand) db >= da, and with the bits from 15 up equal the difference of the low words is that of the query offsets)
        cM = __builtin_amdgcn_uicmp(xh, 0u, CMP_EQ) & __builtin_amdgcn_uicmp(xl, 32768u, CMP_ULT) & __builtin_amdgcn_uicmp(blo - alo, wl, CMP_ULE) & hasM;
        nM = __builtin_amdgcn_uicmp(xh, 32768u, CMP_ULT) & __builtin_amdgcn_uicmp(db - da, G, CMP_ULE) & hasM;
    };
    // per row: the heads and the last hits of the fragments that are written -- the only hits that store anything (two masks a row: at 106 scalar registers a workgroup of
    // sixteen waves is alone on its CU, and the kernel a fifth slower than the form it replaces)
    u64 headM[YD_FRAG_IPT], lastM[YD_FRAG_IPT];
    unsigned nDead = 0;
    u64 cM, nM, vM; rowLinks(0, cM, nM, vM);
#pragma unroll
    for (int k = 0; k < YD_FRAG_IPT; k++) {
        u64 cN, nN, vN = 0ull;
        if (k + 1 < YD_FRAG_IPT) rowLinks(k + 1, cN, nN, vN);
        else {                                                                 // the hit behind the wave's range against the wave's last: scalars
            const uint32_t alo = (uint32_t)__builtin_amdgcn_readlane((int)klo[YD_FRAG_IPT - 1], 63), ahi = (uint32_t)__builtin_amdgcn_readlane((int)khi[YD_FRAG_IPT - 1], 63),
                           blo = (uint32_t)__builtin_amdgcn_readlane((int)elo, 63), bhi = (uint32_t)__builtin_amdgcn_readlane((int)ehi, 63);
            const uint32_t xh = ahi ^ bhi, xl = alo ^ blo, da = (ahi << 17) | (alo >> 15), db = (bhi << 17) | (blo >> 15);
            const bool has = wbase + (uint32_t)(64 * YD_FRAG_IPT) < nHits;
            cN = (has && xh == 0u && xl < 32768u && blo - alo <= wl) ? 1ull : 0ull; nN = (has && xh < 32768u && db - da <= G) ? 1ull : 0ull;
        }
        const u64 nextC = (cM >> 1) | (cN << 63), nextN = (nM >> 1) | (nN << 63);
        const u64 head = vM & ~cM, last = vM & ~nextC, dead = drop ? (head & last & ~nM & ~nextN) : 0ull;
        headM[k] = head & ~dead; lastM[k] = last & ~dead;
        if (lane == 0u) sCnt[(int)w * YD_FRAG_IPT + k] = (uint32_t)__builtin_popcountll(headM[k]);
        nDead += (unsigned)__builtin_popcountll(dead);
        cM = cN; nM = nN; vM = vN;
    }
    __syncthreads();
    if (w == 0u) {
        // exclusive scan of the NW x IPT (wave, row) counts, E consecutive ones per lane; then the look-back
        constexpr int E = YD_FRAG_IPT * NW / 64;
        uint32_t v[E], sum = 0;
#pragma unroll
        for (int e = 0; e < E; e++) { v[e] = sum; sum += sCnt[(int)lane * E + e]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += x; }
#pragma unroll
        for (int e = 0; e < E; e++) sCnt[(int)lane * E + e] = incl - sum + v[e];
        const uint32_t agg = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t excl = tileLookBack(tileState, tile, agg, lane, total + 1);
        if (lane == 0u) { sPrefix = excl; if (tile + 1u == gridDim.x) *total = excl + agg; }
    }
    __syncthreads();
    const uint32_t prefix = sPrefix;
#pragma unroll
    for (int k = 0; k < YD_FRAG_IPT; k++) {
        if ((headM[k] | lastM[k]) == 0ull) continue;                             // wave-uniform
        const bool head = __builtin_amdgcn_inverse_ballot_w64(headM[k]), last = __builtin_amdgcn_inverse_ballot_w64(lastM[k]), live = head | last;
        // the fragment this hit belongs to: the live heads before it in the tile, itself included
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(headM[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)headM[k], 0u));
        const uint32_t f = prefix + sCnt[(int)w * YD_FRAG_IPT + k] + below + (head ? 1u : 0u) - 1u;
        if (!live || f >= cap) continue;
        const uint32_t qo = klo[k] & 0x7FFFu, diag = (uint32_t)__builtin_amdgcn_alignbit(khi[k], klo[k], 15), rs = khi[k] >> 15;
        if (head && last) {                                                  // a fragment of one hit: the whole record in one 16-byte store
            const uint32_t eqo = qo + wl - 1u;
            uint4 v; v.x = diag + qo; v.y = qo | (eqo << 16); v.z = wl /* refLen, used = 0 */; v.w = rs;
            *(uint4 *)&frags[f] = v;
        } else {
            if (head) { frags[f].sro = diag + qo; frags[f].sqo = (uint16_t)qo; frags[f].rs = rs; frags[f].used = 0; }
            if (last) frags[f].eqo = (uint16_t)(qo + wl - 1u);
        }
    }
    if (lane == 0u && nDead) atomicAdd(&sDead, nDead);
    __syncthreads();
    if (t == 0 && sDead) atomicAdd(&deadParts[blockIdx.x & 1023u], sDead);
}
__global__ void k_frag_finish(DevFrag *frags, uint32_t nFrags)
{
    YD_HIGH_PRIO();
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nFrags) return;
    frags[f].refLen = (uint16_t)(1 + (int)frags[f].eqo - (int)frags[f].sqo);      // setRefLen, FragsClumps.inl:44-47
}
