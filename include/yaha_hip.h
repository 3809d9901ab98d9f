/* yaha_hip.h -- C-ABI of the MI355X-native hot path of YAHA (seed join -> clumps -> banded affine-gap DP ->
 * score/split).  Plain pointers and sizes only; no torch, no C++ types.
 *
 * The reference (GregoryFaust/yaha v0.1.83) has no plugin/FFI seam: its hot path is the body of the per-read
 * loop in src/Query.c:306-497, which calls (all declared in src/Math.h)
 *     findFragmentsSort        Math.h:554   (QueryMatch.c:52)     -> ygpu stage A1+A2
 *     processFragmentsGapped   Math.h:555   (QueryMatch.c:224)    -> ygpu stage A3+A4
 *     postProcessClumps        Math.h:539   (QueryMatch.c:306)    -> ygpu stage A5..A8
 *       alignClump/scoreClump  Math.h:537-538, findAGS* Math.h:401-410, extendClump* Math.h:535-536
 * on one QueryState_t (Math.h:587-666) per thread.  This header is the *batched* replacement of that loop
 * body: a whole batch of reads is handed over, and for every read the clump list that postProcessClumps
 * leaves in QS->clumps (head -> tail order, QueryMatch.c:309-330) comes back, ready for the host's
 * postFilterBySimilarity / printClump stages.
 *
 * Conventions: every function returns 0 on success or a negative YGPU_E* code; nothing ever calls exit().
 * The caller owns inputs until the call returns.  Results are owned by the context and stay valid until the
 * next ygpu_run on that context.  One context per device, one host thread per context (mirrors one
 * QueryState_t per thread, Query.c:642-684).
 */
#ifndef YAHA_HIP_H
#define YAHA_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define YGPU_OK            0
#define YGPU_EINVAL      (-1)   /* bad argument / unsupported parameter range            */
#define YGPU_ENODEV      (-2)   /* no HIP device / HIP runtime error                     */
#define YGPU_ENOMEM      (-3)   /* device or host allocation failed                      */
#define YGPU_EOVERFLOW   (-4)   /* a device arena overflowed even after regrowth         */
#define YGPU_EINTERNAL   (-5)
#define YGPU_EBUSY       (-6)   /* ygpu_submit while the context's previous ticket is still open */

/* Alignment parameters: the subset of AlignmentArgs_t (Math.h:257-334) the hot path reads, after
 * postProcessAlignmentArgs (AlignArgs.c:108-169) has filled the derived ones. */
typedef struct ygpu_params {
    int32_t wordLen;        /* -L, from the index header (Query.c:603)                         */
    int32_t maxHits;        /* min(-H, index header maxHits) (Query.c:604-610)                 */
    int32_t bandWidth;      /* -BW                                                             */
    int32_t maxGap;         /* -G                                                              */
    int32_t maxIntron;      /* -I, default = maxGap (AlignArgs.c:111)                          */
    int32_t minMatch;       /* -M                                                              */
    int32_t maxDesert;      /* -MD                                                             */
    int32_t minNonOverlap;  /* = OQCMinNonOverlap = -MNO, default minMatch (AlignArgs.c:115-125) */
    int32_t minRawScore;    /* -R, default = minMatch (AlignArgs.c:113)                        */
    int32_t minExtLength;   /* derived (AlignArgs.c:141-149); 5 at defaults                    */
    int32_t GOCost, GECost, RCost, MScore, XCutoff;
    float   minIdentity;    /* -P, kept as float on purpose (Math.h:292, AlignHelpers.c:359)   */
} ygpu_params;

/* Borrowed, read-only view of the loaded .nib2 bases and index (Query.c:565-626). */
typedef struct ygpu_index_view {
    const uint8_t  *bases;        /* AAs->basePtr: 4-bit codes, two per byte, high nibble first */
    uint64_t        n_base_bytes;
    uint32_t        maxROff;      /* AAs->maxROff (BaseSeq.c:121-125)                           */
    const uint32_t *startingOffs; /* 4^wordLen + 1 entries                                      */
    const uint32_t *ROA;          /* totalMatches entries                                       */
    uint32_t        totalMatches;
    int32_t         wordLen;
} ygpu_index_view;

/* A batch of reads: forward-strand 4-bit codes, one per byte (QS->forwardCodeBuf, Query.c:161-163);
 * read i occupies codes[offsets[i] .. offsets[i+1]).  The reverse complement (Query.c:164-167) is derived
 * on the device. */
typedef struct ygpu_read_batch {
    uint32_t        n_reads;
    const uint8_t  *codes;
    const uint64_t *offsets;      /* n_reads + 1 */
} ygpu_read_batch;

/* Edit operations are packed (len | opcode << 16), opcode one of 'M','R','I','D' (Math.h:353-360). */
#define YGPU_OP_LEN(x)  ((uint32_t)(x) & 0xFFFFu)
#define YGPU_OP_CODE(x) ((char)(((uint32_t)(x) >> 16) & 0xFFu))
#define YGPU_OP_MAKE(code, len) (((uint32_t)(uint8_t)(code) << 16) | ((uint32_t)(len) & 0xFFFFu))

/* One scored clump = the single SFragment + EditOpList + Clump_t fields (Math.h:448-456,512-527).
 * Offsets are strand-local exactly as in the reference (FragsClumps.inl:355-365 converts later). */
typedef struct ygpu_clump {
    uint32_t sro;          /* frag.startRefOff                       */
    uint16_t sqo, eqo;     /* frag.startQueryOff / endQueryOff       */
    uint16_t refLen;       /* frag.refLen (SUINT)                    */
    uint16_t totScore;     /* Clump_t::totScore (QOFF!)              */
    uint16_t totLength, matchedBases, mismatchedBases, gapBases;
    uint8_t  status;       /* clumpReversed 0x01 | Aligned 0x04 | Scored 0x08 | Split 0x10 */
    uint8_t  reserved;
    uint32_t op_start;     /* first op of this clump in ygpu_result_batch::ops */
    uint32_t n_ops;
} ygpu_clump;

/* Work counters (per batch) used for the algorithmic-bytes figure of SURVEY.md 8(d). */
typedef struct ygpu_counters {
    uint64_t kmer_lookups, hits, fragments, regions, clumps_formed, clumps_scored;
    uint64_t dp_ext_calls, dp_ext_rows, dp_ext_cells;
    uint64_t dp_gap_calls, dp_gap_rows, dp_gap_cells;
    uint64_t perfect_ext_bases, ref_bases_touched, ops_out, splits;
} ygpu_counters;

typedef struct ygpu_result_batch {
    uint32_t          n_reads;
    const uint32_t   *clump_start;   /* n_reads + 1; clumps of read i = [clump_start[i], clump_start[i+1]) in
                                        QS->clumps head->tail order after postProcessClumps */
    const ygpu_clump *clumps;
    const uint32_t   *ops;
    uint64_t          n_clumps, n_ops;
    ygpu_counters     counters;
} ygpu_result_batch;

typedef struct ygpu_ctx ygpu_ctx;

/* HIP devices visible to the process (0 when there is none or no runtime): what `-gpus N` may ask for. */
int  ygpu_device_count(void);
/* Create a context on HIP device `device`: copies the index view into HBM (replicated per GPU; reads shard
 * across GPUs, no collective).  Replaces the per-thread makeQueryState/initializeQueries (QueryState.c:36-104). */
int  ygpu_init(int device, const ygpu_index_view *index, const ygpu_params *params, ygpu_ctx **out);
/* The same for n devices at once, ctx_per_device contexts each -- the reference maps its index ONCE for all its threads (Query.c:565-626, 642-690); here every
 * device needs the image in its own HBM, and it crosses the host's memory and PCIe only once: devices[0] takes it from the host in pieces, every further device
 * takes each piece from the device before it as soon as that one has it (hipMemcpyPeerAsync over xGMI, a chain pipelined by piece).  A device that cannot reach
 * its neighbour (hipDeviceCanAccessPeer) uploads from the host itself.  The contexts of a device share its image (as ygpu_clone's do); their streams and events
 * are made while the image travels.  out[k * ctx_per_device + j] = context j of devices[k]; all are filled (also on failure, for ygpu_last_error; destroy them
 * all, a device's first context last); rc_each[0..n), when given, receives every device's own result code, the return value is the first that is not 0.  The
 * same device may be listed twice (two images on one device: what the 1-GPU tests use to drive the chain). */
int  ygpu_init_multi(const int *devices, int n, int ctx_per_device, const ygpu_index_view *index, const ygpu_params *params, ygpu_ctx **out, int *rc_each);
/* A further context on the same device sharing the parent's index image (no second upload; the parent must outlive it).  Two
 * contexts per GPU, one host thread and one stream of batches each, overlap one context's latency-bound stages and host work
 * with the other's compute -- the counterpart of running the reference with more threads than one per core is not needed:
 * there is no shared mutable state between contexts (as between the reference's per-thread QueryStates, Query.c:642-684). */
int  ygpu_clone(const ygpu_ctx *parent, ygpu_ctx **out);
void ygpu_destroy(ygpu_ctx *ctx);
const char *ygpu_last_error(const ygpu_ctx *ctx);
/* Device memory as the context sees it: free and total bytes of its device, and the bytes its own buffers hold (arenas grow with the first batches; the
 * reference's per-thread QueryState grows the same way, Query.c:81-100, 313).  A batching host uses it to decide how many contexts a device can carry. */
int  ygpu_memory(ygpu_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes, uint64_t *ctx_bytes);
/* A context the batching host leaves out (its device has no room for another set of arenas): releases the context's own buffers and takes it out of the count of
 * contexts that share the device's memory budget, so that the contexts that do run are not held to a share sized for it.  The reference has no counterpart -- a
 * thread of Query.c:642-690 that cannot allocate its QueryState is fatal (FragsClumps.c:74).  The context can only be destroyed afterwards. */
int  ygpu_park(ygpu_ctx *ctx);
/* The capacities of a context's arenas and the estimates it carries from batch to batch (trace rows per bound row, clump slots ...), as they stand after a batch;
 * ygpu_presize gives another context of the same build the same capacities in one go and seeds its estimates, so that its first batch runs like any later one
 * instead of growing a hundred buffers one by one (each growth frees a buffer, and a free waits for every kernel on the device: first batches used to run one
 * at a time with the device's other contexts held back).  The reference's per-thread QueryState grows lazily as well (Query.c:81-100, 313); its threads do not
 * share an allocator that stalls the others. */
typedef struct ygpu_arena_profile {
    uint32_t n; uint32_t last_clump_slots; uint64_t cap[224]; double trace_ratio, ops_ratio; int64_t last_fall; uint64_t bases;
} ygpu_arena_profile;
int  ygpu_get_arena_profile(ygpu_ctx *ctx, ygpu_arena_profile *out);
int  ygpu_presize(ygpu_ctx *ctx, const ygpu_arena_profile *profile);

/* Stage reads into HBM (H2D).  Separate from ygpu_run so that a benchmark can time the hot path with inputs
 * already resident. */
int  ygpu_upload(ygpu_ctx *ctx, const ygpu_read_batch *batch);
/* The same without the wait at its end: returns once the copies are queued (the reads' bytes may still be on their way).  The batch's memory must stay
 * unchanged until the ygpu_run that follows has returned.  (The command line's context threads: their batches live until they are printed.) */
int  ygpu_upload_nowait(ygpu_ctx *ctx, const ygpu_read_batch *batch);
/* Run the whole hot path (A1..A10) on the resident batch; results stay in HBM. */
int  ygpu_run(ygpu_ctx *ctx);
/* Copy results of the last ygpu_run to host memory owned by the context. */
int  ygpu_collect(ygpu_ctx *ctx, ygpu_result_batch *out);
/* The same into memory of the caller's: ygpu_result_size gives the element counts of the last ygpu_run; ygpu_collect_into copies clump_start[n_reads + 1],
 * clumps[n_clumps] and ops[n_ops] there and points `out` at them.  With buffers from ygpu_host_alloc (page-locked host memory) the copy is a DMA straight
 * into them -- no staging copy by the calling thread, none afterwards to get the results out of the context: a batching host keeps a result buffer per batch
 * in flight and hands it to its formatter threads while the context already runs the next batch (the reference's QueryState owns its clump list the same
 * way, QueryState.c:156-161).  Any memory works; pageable memory is staged by the runtime. */
int  ygpu_result_size(ygpu_ctx *ctx, uint64_t *n_clumps, uint64_t *n_ops);
int  ygpu_collect_into(ygpu_ctx *ctx, uint32_t *clump_start, ygpu_clump *clumps, uint32_t *ops, ygpu_result_batch *out);
void *ygpu_host_alloc(size_t bytes);           /* NULL when no device runtime is there or the memory cannot be locked */
void  ygpu_host_free(void *p);
/* ---- post-filter on the device (optional stage behind ygpu_run) ------------------------------------------------------------------------------------------
 * The reference's loop goes on, after the hot path, with postFilterBySimilarity (Query.c:450, GraphPath.cpp:897-1086: Optimal Query Coverage, filter by
 * similarity, mapping quality) and prints what is left -- for a 1 kbp read one or two of the ~75 clumps the hot path returns.  ygpu_postfilter runs that step
 * for the whole batch on the device (same routine as the host's, yaha_amd/csrc/oqc_core.h) and ygpu_collect_filtered returns, per read and in PRINT order,
 * only the clumps printClumps would see (QS->clumps after the filter) with the fields the filter sets (Math.h Clump_t: status, mapQuality, numSecondaries,
 * matchedPrimary; QS->primaryCount), and only their edit ops.  Bit-identical to filtering ygpu_collect's output on the host; -OQC N runs keep the host filter.
 * A read with more clumps than the device stage takes (1 792: what its sort holds in a workgroup's 64 KB of LDS) comes back UNFILTERED -- all its clumps in
 * ygpu_collect's order, primaryCount == 0xFFFF in each -- for the caller to filter (yaha_session_emit_filtered does).
 * The stage works on a SNAPSHOT of the batch's results, so a context can run its next batch while another thread filters this one (the reference's threads
 * each finish a read before they take the next, Query.c:430-470; here the two halves of a batch's life overlap):
 *   context's thread:  ygpu_upload, ygpu_run, ygpu_postfilter_snapshot  -> hand the batch to the filter thread -> ygpu_upload, ygpu_run of the next batch ...
 *   filter thread:     ygpu_postfilter, ygpu_filtered_size, ygpu_collect_filtered
 * One snapshot at a time: the next ygpu_postfilter_snapshot may be called once the filtered results of the previous one have been collected.  ygpu_postfilter
 * without a snapshot takes one itself (the sequential use: ygpu_run, ygpu_postfilter, ygpu_collect_filtered on one thread).  The filtered results are handed out
 * ONCE: after ygpu_collect_filtered, ygpu_filtered_size answers YGPU_EINVAL until the next ygpu_postfilter (it used to answer with the previous batch).  A snapshot
 * that will not be filtered after all (an error between the two calls) is dropped with ygpu_postfilter_drop -- otherwise the next ygpu_postfilter would filter IT. */
typedef struct ygpu_postfilter_params {
    int32_t  minNonOverlap, BPCost, maxBPLog, FBS;     /* AlignArgs: OQCMinNonOverlap, BPCost, maxBPLog, FBS (0/1) */
    float    FBS_PSLength, FBS_PSScore;
    int32_t  bppVmin, bppN;                            /* break point penalty (int)(min(log10(d), maxBPLog) * BPCost + 0.5) as a step function of the distance d > 10: */
    const uint32_t *bppThr;                            /*   bppVmin + number of thresholds <= d; bppN thresholds, ascending (host memory; copied) */
    uint32_t n_seqs; const uint32_t *seq_start, *seq_length;   /* reference sequences in bases, ascending (findBaseSequenceNum, BaseSeq.c:81-90; host memory; copied) */
} ygpu_postfilter_params;
typedef struct ygpu_out_clump {                        /* 40 bytes */
    ygpu_clump c;                                      /* op_start indexes the filtered batch's ops */
    uint8_t  status, mapQuality; uint16_t numSecondaries, matchedPrimary, primaryCount;
} ygpu_out_clump;
typedef struct ygpu_filtered_batch {
    uint32_t              n_reads;
    const uint32_t       *clump_start;   /* n_reads + 1 */
    const ygpu_out_clump *clumps;
    const uint32_t       *ops;
    uint64_t              n_clumps, n_ops;
    ygpu_counters         counters;
} ygpu_filtered_batch;
int  ygpu_set_postfilter(ygpu_ctx *ctx, const ygpu_postfilter_params *p);     /* once per context */
int  ygpu_postfilter_snapshot(ygpu_ctx *ctx);                                  /* after ygpu_run, on the context's thread */
int  ygpu_postfilter(ygpu_ctx *ctx);                                           /* after ygpu_run or ygpu_postfilter_snapshot */
int  ygpu_postfilter_drop(ygpu_ctx *ctx);                                      /* forget an unfiltered snapshot and uncollected results */
int  ygpu_filtered_size(ygpu_ctx *ctx, uint64_t *n_clumps, uint64_t *n_ops);
int  ygpu_collect_filtered(ygpu_ctx *ctx, uint32_t *clump_start, ygpu_out_clump *clumps, uint32_t *ops, ygpu_filtered_batch *out);

/* Stage-level entry for tests of the post-filter (as ygpu_dp_batch is for the DP kernels): place a result batch on the device as if ygpu_run had produced it
 * for the reads uploaded last (r->n_reads must equal the uploaded batch's; clump_start / clumps / ops as ygpu_collect returns them).  ygpu_postfilter,
 * ygpu_collect and their siblings then work on it -- so that the device filter can be driven with clump lists no real read produces (hundreds of exact ties,
 * chains of overlaps) and compared with the host's on the same data. */
int  ygpu_inject_results(ygpu_ctx *ctx, const ygpu_result_batch *r);
/* Stage-level test entry for the exclusive sums and orderings the hot path lays its variable-size outputs out with (device/scan.h: single-pass look-back sums of
 * u32 / u64, orderings by a small key) and for the post-filter's sort on the wave (device/oqc_stage.h waveSort, against oqc_core.h sortRange): runs them on n
 * pseudo-random elements (the sort: on arrays of 2 .. 1 792 entries with ties) and compares with the plain host loops; and for A2's workgroup sort (device/wgsort.h: five
 * shapes, both rankings -- LDS atomics and ballots -- over segments of every fill whose diagonals are made to tie, against std::stable_sort; its own order check must
 * stay silent).  0, or YGPU_EINTERNAL with the first difference in ygpu_last_error.  (The reference needs none of them: it handles one read at a time and merges
 * sorted k-mer lists, Query.c:306-497, QueryMatch.c:52-121.) */
int  ygpu_selftest_primitives(ygpu_ctx *ctx, uint32_t n, uint32_t seed, int key_bits);
/* The trace stream of the last ygpu_run -- the largest HBM stream of the path, an implementation choice and not algorithmic traffic: the X-drop rows kernel writes
 * a record per DP row it computes, the traceback (SW.cpp:1138-1195) comes back for those of rows 0 .. maxi of the calls that end above zero (SW.cpp:1091-1111 returns
 * before any traceback otherwise).  out[0] = records written, out[1] = records a traceback can visit, out[2] = extension calls run, out[3] = calls that walk,
 * out[4] = bytes a record, out[5] = bytes of the trace arena.  After ygpu_run, before the next upload. */
int  ygpu_trace_volume(ygpu_ctx *ctx, uint64_t out[6]);

/* Asynchronous form: ygpu_submit hands the batch to the context and returns at once; the context's own worker thread does
 * upload + run + collect; ygpu_wait blocks until the ticket is complete and returns the results (ygpu_poll: 1 = complete, 0 = still running).
 * One host thread can so keep several contexts (devices) busy -- the reference needs one thread per QueryState for that (Query.c:642-684).
 * One open ticket per context: the caller keeps the batch alive until ygpu_wait returns, results stay valid until the next ygpu_submit /
 * ygpu_run on the context; a second ygpu_submit before ygpu_wait returns YGPU_EBUSY, a ygpu_wait for a ticket that is not open (or that another
 * thread is already waiting for) YGPU_EINVAL.  ygpu_last_error describes the ticket's failure once ygpu_wait has returned it; while a ticket is open
 * the text belongs to the worker and must not be read. */
typedef uint64_t ygpu_ticket;
int  ygpu_submit(ygpu_ctx *ctx, const ygpu_read_batch *batch, ygpu_ticket *ticket);
int  ygpu_poll(ygpu_ctx *ctx, ygpu_ticket ticket);
int  ygpu_wait(ygpu_ctx *ctx, ygpu_ticket ticket, ygpu_result_batch *out);
/* Elapsed device time of the last ygpu_run in milliseconds, total and per kernel family (HIP events on the
 * context's stream). names/ms arrays are owned by the context. */
int  ygpu_last_timing(ygpu_ctx *ctx, float *total_ms, int *n_stages, const char *const **names, const float **ms);

/* ---- stage-level entry points (what golden vectors are replayed against) ------------------------------ */

/* Fragment_t (Math.h:448-456) as produced by findFragmentsSort, plus the owning read/strand. */
typedef struct ygpu_fragment {
    uint32_t startRefOff;
    uint16_t startQueryOff, endQueryOff;
    uint16_t refLen;
    uint16_t reserved;
    uint32_t read_strand;    /* read * 2 + strand */
} ygpu_fragment;

/* A1+A2 for the resident batch: fragments sorted by (read, strand, diag, SQO). Output owned by ctx. */
int  ygpu_seed_join(ygpu_ctx *ctx, const ygpu_fragment **frags, uint64_t *n_frags);

/* A3+A4 on the fragments of the last ygpu_seed_join: per clump (in creation order per read/strand) the
 * ordered fragment list before alignClump. clump_frag_start has n_clumps+1 entries. */
int  ygpu_chain(ygpu_ctx *ctx, const ygpu_fragment **clump_frags, const uint32_t **clump_frag_start,
                const uint32_t **clump_read_strand, uint64_t *n_clumps);

/* One DP call of findAffineGapScore (SW.cpp:798-1208) through its wrappers (SW.cpp:462-547). */
enum { YGPU_DP_FULL = 0, YGPU_DP_BANDED = 1, YGPU_DP_EXT_FWD = 2, YGPU_DP_EXT_REV = 3 };
typedef struct ygpu_dp_problem {
    uint32_t read;      /* index into the resident batch              */
    uint8_t  strand;    /* 0 forward codes, 1 reverse-complement      */
    uint8_t  mode;      /* YGPU_DP_*                                   */
    uint16_t qOff;      /* as passed to the reference wrapper          */
    uint16_t qLen;
    uint16_t rLen;      /* FULL/BANDED only                            */
    uint32_t rOff;
} ygpu_dp_problem;
typedef struct ygpu_dp_result {
    int32_t  score;
    uint16_t addedQLen, addedRLen;  /* extensions only */
    uint32_t op_start, n_ops;       /* ops in list order (head..tail) as the wrapper would merge them */
} ygpu_dp_result;
int  ygpu_dp_batch(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n,
                   const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops);
/* The same with the kernel family chosen explicitly.  AUTO = the kernels ygpu_run would use for these parameters: at the default band (-BW 5, -G >= 10)
 * the lane kernels -- k_ext_rows + k_ext_trace for extensions (findAGSForward/BackwardExtension, SW.cpp:479-533), the pure-diagonal shortcut and
 * k_gap_lanes / k_gap_wave for gap fills (findAGSAlignment[Banded], SW.cpp:462-475) -- otherwise the wave-per-problem DP.  WAVE = always the
 * wave-per-problem DP (dp_wave.h).  LANES_CAREFUL = the lane kernels with the k_ext_rows instantiation that serves splitClump's careful extensions
 * (SW.cpp:553-788 call the same findAGSExtension).  ygpu_dp_batch = AUTO. */
enum { YGPU_DP_KERNELS_AUTO = 0, YGPU_DP_KERNELS_WAVE = 1, YGPU_DP_KERNELS_LANES = 2, YGPU_DP_KERNELS_LANES_CAREFUL = 3 };
int  ygpu_dp_batch_ex(ygpu_ctx *ctx, const ygpu_dp_problem *problems, uint32_t n, int kernels,
                      const ygpu_dp_result **results, const uint32_t **ops, uint64_t *n_ops);

/* ---- host stages around the hot path (SURVEY.md 8(f) rows restated on the host) ------------------------
 * A session owns what processQueryFile (Query.c:551-709) sets up: parsed arguments, the mmap'ed .nib2 and
 * index, the query reader.  argv is the reference's own command line (`-x index -q reads -osh out ...`). */
typedef struct yaha_session yaha_session;
int  yaha_session_open(int argc, const char *const *argv, yaha_session **out);
void yaha_session_close(yaha_session *s);
const char *yaha_session_error(const yaha_session *s);
int  yaha_session_params(const yaha_session *s, ygpu_params *p);
int  yaha_session_index_view(const yaha_session *s, ygpu_index_view *v);
/* SAM header (@HD/@SQ/@PG, AlignOutput.c:30-111); text owned by the session. */
int  yaha_session_header(yaha_session *s, const char **text, size_t *len);
/* Read up to max_reads queries (readNextQuery, Query.c:102-228); b->n_reads == 0 at end of input.
 * The batch stays valid until the next call. */
int  yaha_session_next_batch(yaha_session *s, uint32_t max_reads, ygpu_read_batch *b);
/* Post-filter (OQC/FBS/dedup/MAPQ, GraphPath.cpp:897-1174) and format (printClump, AlignOutput.c:115-321)
 * the hot-path results of the current batch; text owned by the session until the next call. */
int  yaha_session_emit(yaha_session *s, const ygpu_result_batch *r, const char **text, size_t *len);
/* The same for a batch whose post-filter ran on the device: set a context up with the session's filter parameters (pointers into the session, valid until it is
 * closed; returns YGPU_EINVAL for runs the device stage does not take: -OQC N, break point costs that are no step function), then format what
 * ygpu_collect_filtered returned.  yaha_session_emit(ygpu_collect(...)) and yaha_session_emit_filtered(ygpu_collect_filtered(...)) give the same text. */
int  yaha_session_postfilter_params(yaha_session *s, ygpu_postfilter_params *p);
int  yaha_session_emit_filtered(yaha_session *s, const ygpu_filtered_batch *r, const char **text, size_t *len);
/* `yaha -g genome.fa [-L k] [-S s] [-H h]`: writes genome.nib2 and genome.X<LL>_<SS>_<HHHHH>S (Main.c:554-628). */
int  yaha_build_index(int argc, const char *const *argv);
/* The complete command-line program (index creation or query alignment on the GPU). */
int  yaha_main(int argc, char **argv);

#ifdef __cplusplus
}
#endif
#endif
