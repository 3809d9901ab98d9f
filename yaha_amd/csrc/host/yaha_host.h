// yaha_host.h -- host side of the MI355X-native YAHA hot path: everything the reference does *around* the
// per-read hot loop (file formats, argument handling, FASTA/FASTQ reading, OQC/FBS post-filter, SAM/Blast8
// output).  These stages are SURVEY.md 8(f) "next" rows restated on the host so that `yaha -x .. -q ..` keeps
// producing bit-identical output; the hot path itself (8(a) A1..A10) lives in ../device and is reached only
// through include/yaha_hip.h.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <new>
#include "../../../include/yaha_hip.h"

namespace yaha {

// ---- 4-bit code tables (data contract, reference Math.c:141-157) -----------------------------------------
extern const uint8_t kFourBitCodes[128];
extern const char    kFourBitChars[16];
extern const uint8_t kFourBitCompCodes[16];
inline uint8_t map8to4(int c) { return kFourBitCodes[c & 127]; }
inline uint8_t get4(const uint8_t *bases, uint32_t off) { uint8_t b = bases[off >> 1]; return (off & 1) ? (b & 0xF) : (uint8_t)(b >> 4); }

// ---- Marsaglia xorshift RNG (reference Math.c:238-343) ----------------------------------------------------
struct RandState { uint32_t s[5]; };
void     randInitDefault(RandState &r);
uint32_t randBits(RandState &r);
void     randSample(RandState &r, const uint32_t *in, int inLen, uint32_t *out, int outLen);

// ---- memory-mapped read-only file -----------------------------------------------------------------------
struct MMap { void *ptr = nullptr; size_t size = 0; int fd = -1; bool open(const char *path, std::string &err); void close(); ~MMap() { close(); } };

// ---- .nib2 genome (reference Compress.c, BaseSeq.c) -----------------------------------------------------
struct BaseSeq { std::string name; uint32_t start; uint32_t length; };      // start in BASES (already normalised)
struct Genome {
    MMap map; std::vector<uint8_t> owned;              // either mapped file or in-memory image
    const uint8_t *bases = nullptr; uint64_t nBaseBytes = 0;
    std::vector<BaseSeq> seqs; uint32_t maxROff = 0;
    int findSeq(uint32_t off) const;                   // findBaseSequenceNum, BaseSeq.c:81-90
};
bool compressFasta(const char *fastaPath, std::vector<uint8_t> &nib2Image, std::string &err);   // Compress.c:220-329
bool parseNib2(const uint8_t *img, size_t size, Genome &g, std::string &err);                   // Compress.c:76-134
bool loadNib2(const char *path, Genome &g, std::string &err);
bool writeFile(const char *path, const void *data, size_t size, std::string &err);             // mode 0744, FileHelpers.c:279

// ---- index (reference Index.c:49-335, Query.c:596-626) --------------------------------------------------
struct IndexFile {
    MMap map; std::vector<uint32_t> owned;
    int wordLen = 0; int maxHits = 0; uint32_t totalMatches = 0; const uint32_t *SO = nullptr; const uint32_t *ROA = nullptr;
};
// builds the complete file image {-1, wordLen, maxHits, total} + SO[4^L+1] + ROA[total] (huge-page backed: 4.3 GB at L=15)
struct IndexImage { uint32_t *p = nullptr; size_t words = 0, bytes = 0; bool alloc(size_t n); void release(); uint32_t &operator[](size_t i) { return p[i];
    } ~IndexImage() { release(); } };
bool buildIndex(const Genome &g, int wordLen, int skipDist, int maxHits, IndexImage &image, FILE *log);
// the same image built on HIP device `device` (device/index_build.hip)
bool buildIndexDevice(int device, const Genome &g, int wordLen, int skipDist, int maxHits, IndexImage &image, FILE *log, std::string &err);
int  visibleDevices();                                  // HIP devices this process can use (0 on a machine without a GPU)
bool parseIndex(const uint32_t *img, size_t bytes, IndexFile &ix, std::string &err);
bool loadIndex(const char *path, IndexFile &ix, std::string &err);

// ---- arguments (reference AlignArgs.c, Main.c) -----------------------------------------------------------
struct Args {
    std::string gfileName, xfileName, qfileName = "stdin", ofileName; bool haveG = false, haveX = false, haveO = false;
    int numThreads = 1; bool fastq = false;
    int wordLen = 15, skipDist = 1, maxHits = -1;
    int maxGap = 50, maxIntron = -1, minMatch = 25; float minIdentity = 0.9f; int bandWidth = 5, maxDesert = 50, minRawScore = -1, minNonOverlap = -1;
    bool affineGapScoring = true; int GOCost = 5, GECost = 2, RCost = 3, MScore = 1, XCutoff = 25; int minExtLength = 0;
    bool OQC = true; int OQCMinNonOverlap = -1, BPCost = 5, maxBPLog = 5; bool FBS = false; float FBS_PSLength = 0.90f, FBS_PSScore = 0.90f;
    int maxQueryLength = 32000; bool verbose = false, outputBlast8 = false, outputSAM = true, hardClip = true;
    // extensions of this implementation (not in the reference CLI)
    int batchReads = 0; int device = 0; int gpus = 1; int ctxPerGpu = 3; bool cpuIndex = false; bool devicePostFilter = true;      // batchReads 0: batches of ~16 M bases
    bool query = false, index = true, compress = false, uncompress = false;   // -c / -u: .fa -> .nib2 / .nib2 -> .fasta only (Main.c:284-293, non-user builds of the reference)
};
void postProcessArgs(Args &a, bool query);                                  // AlignArgs.c:108-169
// returns 0 to continue, >0 exit code+1 to stop (usage / error)
int  parseArgs(int argc, char **argv, Args &a);                            // Main.c:187-565
void paramsFromArgs(const Args &a, ygpu_params &p);
std::string samHeader(const Args &a, const Genome &g);                      // AlignOutput.c:30-111

// ---- reads (reference Query.c:63-228, QueryState.c:172-187) ---------------------------------------------
struct Read {
    std::string id; std::string fwd, rev; std::vector<uint8_t> fwdCodes; std::string qual;
    int len() const { return (int)fwd.size(); }
};
// One record of the input as byte ranges (offsets from base): produced serially by ReadSplitter, parsed by any thread.
struct Span { std::shared_ptr<std::vector<char>> hold; const char *base = nullptr; size_t idLen = 0, seq0 = 0, seqEnd = 0, qual0 = 0, qualEnd = 0; };
// Record boundaries of a FASTA/FASTQ stream (memory-mapped file, or stdin/pipe read in blocks); see reader.cpp for the rules.
struct ReadSplitter {
    size_t blockBytes = 32u << 20;                       // streaming sources are read in blocks of this size
    int fd = -1; bool ownFd = false; const char *mapPtr = nullptr; size_t mapLen = 0;
    std::shared_ptr<std::vector<char>> chunk; const char *cur = nullptr, *end = nullptr; bool atEof = true, done = true, fastq = false;
    bool open(const char *path, std::string &err);       // peeks '>' / '@' (Query.c:63-74)
    void close();
    bool nextSpan(Span &s);                              // false at the end of input (an empty sequence ends it, Query.c:216-217)
    size_t nextSpans(size_t maxSpans, std::vector<Span> &out) { return nextSpans(maxSpans, ~(size_t)0, out); }
    size_t nextSpans(size_t maxSpans, size_t maxBases, std::vector<Span> &out);   // up to maxSpans records, stopping once their sequences reach maxBases bytes
    ~ReadSplitter() { close(); }
  private:
    void fill(); bool seek(char c, size_t *pos); bool seekNlAt(size_t *pos);
};
// id / sequence / quality / codes of one record; false = the record is skipped (with the reference's warning on stderr)
bool parseSpan(const Span &s, bool fastq, int maxQueryLength, int wordLen, Read &r);
void finishRead(Read &r);                                // 4-bit codes + reverse-complement text from r.fwd
struct ReadReader {                                      // sequential reader: one accepted read per call
    ReadSplitter split; bool fastq = false; int maxQueryLength = 32000; int wordLen = 15;
    bool open(const char *path, std::string &err);
    bool next(Read &r);                                  // readNextQuery; false at EOF
    void close();
};
void seedFromRead(const Read &r, RandState &rs);          // generateRandomSeed

// ---- post filter + output (reference GraphPath.cpp:294-1175, AlignOutput.c:115-321) ---------------------
struct OutClump {                                        // one clump as it reaches printClump
    ygpu_clump c; const uint32_t *ops;                   // ops[0..c.n_ops)
    uint8_t status; uint8_t mapQuality = 255; uint16_t numSecondaries = 0, matchedPrimary = 0;
};
// clumps: QS->clumps head->tail after postProcessClumps.  Result: print order.  primaryCount = QS->primaryCount.
void postFilter(const Args &a, const Genome &g, const Read &r, const ygpu_clump *clumps, uint32_t n, const uint32_t *ops,
                std::vector<OutClump> &out, int &primaryCount);
// Output text of a batch: a plain growable byte buffer that is reused from batch to batch (no zero-fill on growth, no per-record allocation; a formatter
// thread keeps its buffers, so that the steady state touches no allocator and maps no new pages).
struct Text {
    char *p = nullptr; size_t len = 0, cap = 0;
    Text() {} Text(const Text &) = delete; Text &operator=(const Text &) = delete;
    Text(Text &&o) noexcept : p(o.p), len(o.len), cap(o.cap) { o.p = nullptr; o.len = o.cap = 0; }
    Text &operator=(Text &&o) noexcept { if (this != &o) { free(p); p = o.p; len = o.len; cap = o.cap; o.p = nullptr; o.len = o.cap = 0; } return *this; }
    ~Text() { free(p); }
    void clear() { len = 0; }
    char *room(size_t n) { if (len + n > cap) grow(len + n); return p + len; }      // at least n writable bytes at the end
    void append(const char *s, size_t n) { memcpy(room(n), s, n); len += n; }
  private:
    void grow(size_t need) { size_t c = cap ? cap : (1u << 16); while (c < need) c += c / 2; char *q = (char *)realloc(p, c); if (!q) throw std::bad_alloc(); p = q; cap = c; }
};
void printClump(const Args &a, const Genome &g, const Read &r, const OutClump &oc, int primaryCount, Text &out);
}  // namespace yaha
namespace yoqc { struct Params; }
namespace yaha {
// the post-filter's parameters in the form oqc_core.h takes them (host and device stage); thr receives the break point table P points into
void oqcParamsFromArgs(const Args &a, yoqc::Params &P, std::vector<uint32_t> &thr);

// ---- whole-run driver (replacement of processQueryFile, Query.c:551-709) --------------------------------
int effectiveCpus();                                    // affinity mask and control-group CPU quota
int runQueries(Args &a, FILE *log);
int runIndex(Args &a, FILE *log);
int runCompress(Args &a, FILE *log);                     // -c: compressFile only (Main.c:572-577)
int runUncompress(Args &a, FILE *log);                   // -u: uncompressFile, Compress.c:337-397 (50 bases a line)
}  // namespace yaha
