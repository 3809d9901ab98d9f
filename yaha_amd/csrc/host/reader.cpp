// reader.cpp -- FASTA/FASTQ query input with the reference's exact rules (readNextQuery, Query.c:63-228), built for a host
// that feeds several GPUs: the reference reads one character at a time under a file lock (Query.c:105-214), which caps it at a
// few 10^4 reads/s.  Here the input is memory-mapped (regular files) or read in large blocks (stdin, pipes), and the work is
// split in two:
//   * ReadSplitter (serial, one thread): finds the record boundaries with memchr -- the only part that is inherently
//     sequential -- and hands out batches of record spans;
//   * parseSpan (any number of threads): id, sequence with the newlines removed, quality, 4-bit codes, reverse complement,
//     and every skip rule, for one span.
// Rules kept (each was checked against the reference's loop):
//   * the first byte of the input only selects the format ('@' = FASTQ, anything else = FASTA) and is consumed (Query.c:63-74);
//   * id = the header line up to '\n', truncated to 200 characters, ' ' -> '_' (Query.c:120-135);
//   * FASTA: the sequence runs to the next '>' ANYWHERE (not only at a line start) or the end of input; only '\n' is skipped,
//     every other byte (also '\r') is a base (Query.c:140-160);
//   * FASTQ: the sequence runs to the next '+' anywhere; the rest of that line is skipped; the quality runs to the first '@'
//     that follows a '\n' -- where that '\n' has to be one the quality loop itself consumed, so a quality string that STARTS
//     with '@' is read correctly (Query.c:168-199, `prev` starts at 0);
//   * a sequence or quality longer than maxQueryLength, a length mismatch, or 0 < length < wordLen skips the record with the
//     reference's warnings (Query.c:148-156,181-213); an empty sequence (that is not skipped for another reason) ends the input.
#include "yaha_host.h"
#include <cstring>
#include <cstdlib>
#include <cerrno>
#include <algorithm>
#include <fcntl.h>
#include <poll.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>

namespace yaha {

static inline const char *findChar(const char *p, const char *end, char c) { return p < end ? (const char *)memchr(p, c, (size_t)(end - p)) : nullptr; }

// ---- byte source -----------------------------------------------------------------------------------------------------------
bool ReadSplitter::open(const char *path, std::string &err)
{
    close();
    if (const char *e = getenv("YAHA_READ_BLOCK")) { long v = atol(e); if (v >= 16) blockBytes = (size_t)v; }      // test hook: small blocks exercise the refill paths
    const bool isStdin = !strcmp(path, "stdin") || !strcmp(path, "-");
    if (isStdin) { fprintf(stderr, "Reading queries from stdin.\n"); fd = 0; ownFd = false; }
    else { fd = ::open(path, O_RDONLY); ownFd = true; }
    if (fd < 0) { err = std::string("Failure to open input file: ") + path; return false; }
    struct stat st;
    if (!isStdin && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) { mapPtr = (const char *)m; mapLen = (size_t)st.st_size; madvise(m, mapLen, MADV_SEQUENTIAL); }
    }
    if (mapPtr) { cur = mapPtr; end = mapPtr + mapLen; atEof = true; }
    else { chunk = std::make_shared<std::vector<char>>(); chunk->reserve(blockBytes + (1u << 20)); cur = end = chunk->data(); atEof = false; fill(); }
    // openQueryFile: the first character selects the format and is consumed
    fastq = (cur < end && *cur == '@');
    if (cur < end) cur++;
    done = false;
    return true;
}
void ReadSplitter::close()
{
    if (mapPtr) { munmap((void *)mapPtr, mapLen); mapPtr = nullptr; mapLen = 0; }
    if (fd >= 0 && ownFd) ::close(fd);
    fd = -1; chunk.reset(); cur = end = nullptr; done = true;
}
// streaming source: appends one block to the current chunk, or -- when spans handed out earlier still point into it -- starts a new chunk with the
// unconsumed tail copied over (the spans keep their chunk alive through the shared_ptr they carry)
void ReadSplitter::fill()
{
    if (mapPtr || atEof) return;
    const size_t tail = (size_t)(end - cur);
    if (chunk.use_count() > 1 || chunk->capacity() < tail + blockBytes) {
        auto nc = std::make_shared<std::vector<char>>(); nc->reserve(std::max(tail * 2, tail + blockBytes) + (1u << 20));
        nc->assign(cur, end); chunk = nc;
    } else if (cur != chunk->data()) { memmove(chunk->data(), cur, tail); chunk->resize(tail); }
    const size_t have = chunk->size(); chunk->resize(have + blockBytes);
    size_t got = 0;
    // blocks until some input has arrived, then takes what is there without waiting again (up to the block size): a fast producer is read in large blocks,
    // a slow one -- a pipe from a sequencer, an interactive test -- sees its records processed as they come instead of after 32 MB have accumulated
    while (got < blockBytes) {
        if (got > 0) { struct pollfd pf = {fd, POLLIN, 0}; if (poll(&pf, 1, 0) <= 0) break; }
        ssize_t r = ::read(fd, chunk->data() + have + got, blockBytes - got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) { atEof = true; break; }
        got += (size_t)r;
    }
    chunk->resize(have + got);
    cur = chunk->data(); end = cur + chunk->size();
}

// Finds `c` at or after *pos (an offset from cur), reading more input as needed.  Returns false at the end of input (*pos = bytes available).
bool ReadSplitter::seek(char c, size_t *pos)
{
    for (;;) {
        const char *h = findChar(cur + *pos, end, c);
        if (h) { *pos = (size_t)(h - cur); return true; }
        const size_t avail = (size_t)(end - cur);
        if (atEof) { *pos = avail; return false; }
        *pos = avail; fill();
    }
}
bool ReadSplitter::seekNlAt(size_t *pos)            // first "\n@" with the '\n' at or after *pos; *pos = offset of the '@'
{
    for (;;) {
        if (!seek('\n', pos)) return false;
        if (cur + *pos + 1 >= end && !atEof) { const size_t keep = *pos; fill(); *pos = keep; }
        if (cur + *pos + 1 >= end) { *pos = (size_t)(end - cur); return false; }
        if (cur[*pos + 1] == '@') { *pos += 1; return true; }
        *pos += 1;
    }
}

// One record: cur is at the first byte of the id line (the '>' / '@' before it is consumed).  On return cur is past the next record's marker.
bool ReadSplitter::nextSpan(Span &s)
{
    if (done) return false;
    if (cur >= end && !atEof) fill();
    size_t p = 0;
    const bool idNl = seek('\n', &p);
    s.idLen = p; size_t seq0 = idNl ? p + 1 : p;
    p = seq0;
    const bool brk = seek(fastq ? '+' : '>', &p);
    size_t seqEnd = p, recEnd = brk ? p + 1 : p, q0 = recEnd, qEnd = recEnd;
    if (fastq && brk) {
        size_t q = p + 1;
        const bool nl = seek('\n', &q);                      // rest of the '+' line
        q0 = nl ? q + 1 : q; qEnd = q0;
        size_t at = q0;
        const bool more = seekNlAt(&at);
        qEnd = more ? at - 1 : at;                           // the '\n' before the '@' is skipped by the quality loop anyway
        recEnd = more ? at + 1 : at;
    }
    s.hold = chunk; s.base = cur; s.seq0 = seq0; s.seqEnd = seqEnd; s.qual0 = q0; s.qualEnd = qEnd;
    // an empty sequence ends the input -- unless the record is skipped for another reason first (FASTQ: a quality string without a sequence)
    bool empty = true;
    for (size_t k = seq0; k < seqEnd; k++) if (cur[k] != '\n') { empty = false; break; }
    if (empty) {
        bool qualEmpty = true;
        for (size_t k = q0; k < qEnd; k++) if (cur[k] != '\n') { qualEmpty = false; break; }
        if (!fastq || qualEmpty) { done = true; return false; }
    }
    cur += recEnd;
    if (cur >= end && atEof) done = true;                    // the next call would see an empty id and an empty sequence
    return true;
}
size_t ReadSplitter::nextSpans(size_t maxSpans, size_t maxBases, std::vector<Span> &out)
{
    out.clear();
    Span s; size_t bases = 0;
    while (out.size() < maxSpans && bases < maxBases) {
        s.hold.reset();                                      // a reference kept from the last record would make every refill start a new chunk
        if (!nextSpan(s)) break;
        bases += s.seqEnd - s.seq0; out.push_back(s);
    }
    return out.size();
}

// ---- one record -> Read (any thread) ------------------------------------------------------------------------------------------
static inline size_t copyNoNl(const char *p, const char *e, std::string &out, size_t cap, bool &over)
{
    // appends the bytes of [p, e) that are not '\n', at most `cap` of them; over = a further byte existed
    out.clear(); over = false;
    while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p)); const char *segEnd = nl ? nl : e;
        size_t len = (size_t)(segEnd - p);
        if (out.size() + len > cap) { out.append(p, cap - out.size()); over = true; return out.size(); }
        out.append(p, len);
        p = nl ? nl + 1 : e;
    }
    return out.size();
}
bool parseSpan(const Span &s, bool fastq, int maxQueryLength, int wordLen, Read &r)
{
    const char *b = s.base;
    const size_t idn = s.idLen;
    r.id.assign(b, std::min<size_t>(idn, 200));
    for (auto &c : r.id) if (c == ' ') c = '_';
    if (idn > 200) fprintf(stderr, "Warning, Query Id length of %d exceeds maximum length %d.  Id will be truncated.\n", (int)idn, 200);
    bool fail = false, over = false;
    const size_t n = copyNoNl(b + s.seq0, b + s.seqEnd, r.fwd, (size_t)maxQueryLength, over);
    if (over) { fprintf(stderr, "Warning.  Query sequence exceeds maximum length of %d.  Query will be skipped.\n", maxQueryLength); fail = true; }
    r.qual.clear();
    if (fastq) {
        const size_t nq = copyNoNl(b + s.qual0, b + s.qualEnd, r.qual, (size_t)maxQueryLength, over);
        if (over) { fprintf(stderr, "Warning.  Quality score sequence exceeds maximum length of %d.  Query will be skipped.\n", maxQueryLength); fail = true; }
        if (n != nq) { fprintf(stderr, "Warning.  Query sequence (%d) and quality score sequence (%d) have different lengths in fastq file.  Query will be skipped.\n", (int)n,
            (int)nq); fail = true; }
    }
    if (n > 0 && (int)n < wordLen) { fprintf(stderr, "Query length must be at least wordlen bases long. Query will be skipped.\n"); fail = true; }
    if (fail || n == 0) return false;
    finishRead(r);
    return true;
}
void finishRead(Read &r)                                             // Query.c:161-167: 4-bit codes, reverse-complement text
{
    const int n = (int)r.fwd.size();
    r.fwdCodes.resize(n); r.rev.resize(n);
    const char *f = r.fwd.data(); uint8_t *fc = r.fwdCodes.data(); char *rv = &r.rev[0];
    for (int k = 0; k < n; k++) { const uint8_t code = map8to4((unsigned char)f[k]); fc[k] = code; rv[n - 1 - k] = kFourBitChars[kFourBitCompCodes[code]]; }
}

// ---- sequential convenience reader (Session API, tests): exactly max_reads accepted reads per call ------------------------------
bool ReadReader::open(const char *path, std::string &err) { if (!split.open(path, err)) return false; fastq = split.fastq; return true; }
void ReadReader::close() { split.close(); }
bool ReadReader::next(Read &r)
{
    Span s;
    while (split.nextSpan(s)) if (parseSpan(s, fastq, maxQueryLength, wordLen, r)) return true;
    return false;
}

void seedFromRead(const Read &r, RandState &rs)                      // generateRandomSeed, QueryState.c:172-187
{
    int q = 0; const int n = r.len();
    for (int i = 0; i < 5; i++) {
        uint32_t w = 0;
        for (int j = 0; j < 16; j++) { w = (w << 2) | (r.fwdCodes[q] & 0x3); q++; if (q >= n) q = 0; }
        rs.s[i] = w;
    }
}
}  // namespace yaha
