// oracle/hotpath.h -- TEST INFRASTRUCTURE ONLY.
// CPU restatement of YAHA's per-read hot path.  It exists to referee the
// HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing under yaha_amd/ may
// include, link or call it.  Pinning: tests/test_oracle_golden.py replays it (through the product's host
// OQC + SAM writer) against SAM produced by the real reference binary (oracle/_ref/yaha, built from
// /root/reference by oracle/Makefile) -- see tests/golden/README.md.
#pragma once
#include <cstdint>
#include <vector>
#include "../include/yaha_hip.h"

namespace yoracle {

struct Frag { uint32_t sro; uint16_t sqo, eqo; uint16_t refLen; };          // Fragment_t, Math.h:448-456

struct Op { uint16_t len; char code; };                                      // EditOp_t, Math.h:371-377
typedef std::vector<Op> OpList;

struct DPOut { int score = 0; uint16_t addedQ = 0, addedR = 0; OpList ops; uint64_t rows = 0, cells = 0; };

struct ScoredClump {                                                         // what survives postProcessClumps
    Frag frag; OpList ops; uint16_t totScore, totLength, matched, mismatched, gapBases; uint8_t status;
};

struct Index {                                                               // borrowed view
    const uint8_t *bases; uint32_t maxROff; const uint32_t *SO; const uint32_t *ROA; uint32_t totalMatches;
};

struct ChainClump { std::vector<Frag> frags; uint16_t matchedBases; bool reversed; };

// A1 + A2: Query.c:341-412 + QueryMatch.c:52-121 for one strand.
void seedJoin(const Index &ix, const ygpu_params &P, const uint8_t *codes, int qlen, std::vector<Frag> &frags,
              ygpu_counters *ctr);
// A3 + A4: QueryMatch.c:224-303, GraphPath.cpp:161-292, AlignHelpers.c:60-193.  frags is modified in place
// exactly as the reference modifies fragArray (insertFragment trims through the pointer).
void chainFragments(const ygpu_params &P, std::vector<Frag> &frags, int qlen, bool reversed,
                    std::vector<ChainClump> &out, ygpu_counters *ctr);
// A6 through the wrappers SW.cpp:462-547.
DPOut dpFull(const Index &ix, const ygpu_params &P, const uint8_t *q, uint32_t rOff, uint16_t rLen, uint16_t qOff, uint16_t qLen);
DPOut dpBanded(const Index &ix, const ygpu_params &P, const uint8_t *q, uint32_t rOff, uint16_t rLen, uint16_t qOff, uint16_t qLen);
DPOut dpExtend(const Index &ix, const ygpu_params &P, const uint8_t *q, bool reverse, uint32_t rOff, uint16_t qOff, uint16_t qLenArg);
// A5 + A7 + A8 + A10 for one read: clumps in creation order in, QS->clumps (head->tail) out.
void alignScoreClumps(const Index &ix, const ygpu_params &P, const uint8_t *fwd, const uint8_t *rev, int qlen,
                      std::vector<ChainClump> &created, std::vector<ScoredClump> &out, ygpu_counters *ctr);
// whole path for one read
void processRead(const Index &ix, const ygpu_params &P, const uint8_t *fwd, int qlen, std::vector<ScoredClump> &out,
                 ygpu_counters *ctr);
}  // namespace yoracle

extern "C" {
// Batched C entry points used from Python (ctypes).  All outputs are malloc'ed; release with yoracle_free.
typedef struct yoracle_result {
    uint32_t n_reads; uint32_t *clump_start; ygpu_clump *clumps; uint32_t *ops; uint64_t n_clumps, n_ops;
    ygpu_counters counters;
} yoracle_result;
int  yoracle_run(const ygpu_index_view *ix, const ygpu_params *P, const ygpu_read_batch *b, int threads, yoracle_result *out);
void yoracle_free_result(yoracle_result *r);
int  yoracle_seed_join(const ygpu_index_view *ix, const ygpu_params *P, const ygpu_read_batch *b,
                       ygpu_fragment **frags, uint64_t *n);
int  yoracle_chain(const ygpu_index_view *ix, const ygpu_params *P, const ygpu_read_batch *b,
                   ygpu_fragment **clump_frags, uint32_t **clump_frag_start, uint32_t **clump_read_strand, uint64_t *n_clumps);
int  yoracle_dp_batch(const ygpu_index_view *ix, const ygpu_params *P, const ygpu_read_batch *b,
                      const ygpu_dp_problem *probs, uint32_t n, ygpu_dp_result **res, uint32_t **ops, uint64_t *n_ops);
void yoracle_free(void *p);
}
