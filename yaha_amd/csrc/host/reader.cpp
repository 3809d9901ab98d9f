// reader.cpp -- FASTA/FASTQ query reader with the reference's exact rules (Query.c:63-228): id = header line
// truncated to 200 chars with spaces turned into '_'; multi-line sequences; only '\n' is skipped inside a
// sequence; reads longer than maxQueryLength or shorter than wordLen are skipped with a warning; FASTQ records end
// at an '@' that follows a newline; an empty sequence ends the input.
#include "yaha_host.h"
#include <cstring>

namespace yaha {

bool ReadReader::open(const char *path, std::string &err)
{
    if (!strcmp(path, "stdin") || !strcmp(path, "-")) { fprintf(stderr, "Reading queries from stdin.\n"); f = stdin; ownFile = false; }
    else { f = fopen(path, "r"); ownFile = true; }
    if (!f) { err = std::string("Failure to open input file: ") + path; return false; }
    static thread_local char dummy;  (void)dummy;
    setvbuf(f, nullptr, _IOFBF, 1 << 20);
    fastq = (getc_unlocked(f) == '@');                               // openQueryFile, Query.c:63-74
    return true;
}
void ReadReader::close() { if (f && ownFile) fclose(f); f = nullptr; }

static void readToChar(FILE *in, char fchar, bool precNL)            // Query.c:52-61
{ char prev = 0; for (;;) { int c = getc_unlocked(in); if ((c == fchar && (!precNL || prev == '\n')) || c == EOF) return; prev = (char)c; } }

// The text part of readNextQuery (Query.c:102-228): id, sequence, quality.  Serial by nature (one input stream); the code
// conversion of the sequence (finish) is left to the caller so that it can run outside the reader lock.
bool ReadReader::nextRaw(Read &r)
{
    for (;;) {
        r.id.clear(); int charCount = 0;
        for (;;) {
            int c = getc_unlocked(f);
            if (c == '\n' || c == EOF) break;
            if (charCount < 200) r.id.push_back(c == ' ' ? '_' : (char)c);
            charCount++;
        }
        if (charCount > 200) fprintf(stderr, "Warning, Query Id length of %d exceeds maximum length %d.  Id will be truncated.\n", charCount, 200);
        const char breakChar = fastq ? '+' : '>';
        r.fwd.clear(); r.qual.clear(); bool fail = false;
        for (;;) {
            int c = getc_unlocked(f);
            if (c == breakChar || c == EOF) break;
            if (c == '\n') continue;
            if ((int)r.fwd.size() >= maxQueryLength) {
                fprintf(stderr, "Warning.  Query sequence exceeds maximum length of %d.  Query will be skipped.\n", maxQueryLength);
                readToChar(f, breakChar, false); fail = true; break;
            }
            r.fwd.push_back((char)c);
        }
        const int n = (int)r.fwd.size();
        if (fastq) {
            readToChar(f, '\n', false);
            char prev = 0;
            for (;;) {
                int c = getc_unlocked(f);
                if ((c == '@' && prev == '\n') || c == EOF) break;
                prev = (char)c;
                if (c == '\n') continue;
                if ((int)r.qual.size() >= maxQueryLength) {
                    fprintf(stderr, "Warning.  Quality score sequence exceeds maximum length of %d.  Query will be skipped.\n", maxQueryLength);
                    readToChar(f, '@', true); fail = true; break;
                }
                r.qual.push_back((char)c);
            }
            if (n != (int)r.qual.size()) {
                fprintf(stderr, "Warning.  Query sequence (%d) and quality score sequence (%d) have different lengths in fastq file.  Query will be skipped.\n", n, (int)r.qual.size());
                fail = true;
            }
        }
        if (n > 0 && n < wordLen) { fprintf(stderr, "Query length must be at least wordlen bases long. Query will be skipped.\n"); fail = true; }
        if (fail) continue;
        if (n == 0) return false;
        return true;
    }
}
void ReadReader::finish(Read &r)                                     // Query.c:161-167: 4-bit codes of both strands, reverse-complement text
{
    const int n = (int)r.fwd.size();
    r.fwdCodes.resize(n); r.revCodes.resize(n); r.rev.resize(n);
    for (int k = 0; k < n; k++) {
        uint8_t code = map8to4((unsigned char)r.fwd[k]); r.fwdCodes[k] = code;
        uint8_t rc = kFourBitCompCodes[code]; r.revCodes[n - 1 - k] = rc; r.rev[n - 1 - k] = kFourBitChars[rc];
    }
}
bool ReadReader::next(Read &r) { if (!nextRaw(r)) return false; finish(r); return true; }

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
