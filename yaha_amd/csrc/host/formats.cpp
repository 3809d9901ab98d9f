// formats.cpp -- .nib2 genome codec and hash-index builder/loader.  Both file formats are part of the drop-in
// contract and are reproduced byte for byte, including the Floyd sampling of
// over-represented k-mers with the default-seeded Marsaglia generator.
#include "yaha_host.h"
#include <cstring>
#include <cstdlib>
#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>

namespace yaha {

// ---- tables: reference Math.c:141-157 (data contract: T0 C1 A2 G3 N4 B5 D6 H7 K8 M9 R10 S11 V12 W13 X14 Y15) ----
const uint8_t kFourBitCodes[128] = {
    14,14,14,14,14,14,14,14,14,14,14,14,14,14,14,14, 14,14,14,14,14,14,14,14,14,14,14,14,14,14,14,14,
    14,14,14,14,14,14,14,14,14,14,14,14,14,14,14,14, 14,14,14,14,14,14,14,14,14,14,14,14,14,14,14,14,
    14, 2, 5, 1, 6,14,14, 3, 7,14,14, 8,14, 9, 4,14, 14,14,10,11, 0, 0,12,13,14,15,14,14,14,14,14,14,
    14, 2, 5, 1, 6,14,14, 3, 7,14,14, 8,14, 9, 4,14, 14,14,10,11, 0, 0,12,13,14,15,14,14,14,14,14,14};
const char    kFourBitChars[16]     = {'T','C','A','G','N','B','D','H','K','M','R','S','V','W','X','Y'};
const uint8_t kFourBitCompCodes[16] = {2, 3, 0, 1, 4, 12, 7, 6, 9, 8, 15, 11, 5, 13, 14, 10};

// ---- RNG: reference Math.c:257-343 ------------------------------------------------------------------------
void randInitDefault(RandState &r) { const uint32_t init[5] = {123456789u, 362436069u, 521288629u, 88675123u, 886756453u}; memcpy(r.s, init, sizeof init); }
uint32_t randBits(RandState &r)
{
    uint32_t t = r.s[0] ^ (r.s[0] >> 7);
    r.s[0] = r.s[1]; r.s[1] = r.s[2]; r.s[2] = r.s[3]; r.s[3] = r.s[4];
    r.s[4] = (r.s[4] ^ (r.s[4] << 6)) ^ (t ^ (t << 13));
    return (r.s[1] + r.s[1] + 1) * r.s[4];
}
static uint32_t randUInt(RandState &r, uint32_t start, uint32_t end)
{ double d = (double)randBits(r) / ((double)0xFFFFFFFFu + 1.0); return start + (uint32_t)(d * (end - start)); }
void randSample(RandState &r, const uint32_t *in, int inLen, uint32_t *out, int outLen)   // modified Floyd, Math.c:304-343
{
    std::vector<uint8_t> marked(inLen, 0);
    bool keepMarked = true; int selectNum = outLen;
    if (outLen > inLen / 2) { keepMarked = false; selectNum = inLen - outLen; }
    for (int i = inLen - selectNum; i < inLen; i++) { uint32_t pos = randUInt(r, 0, (uint32_t)i + 1); if (marked[pos]) marked[i] = 1; else marked[pos] = 1; }
    int o = 0; for (int i = 0; i < inLen; i++) if ((marked[i] != 0) == keepMarked) out[o++] = in[i];
}

// ---- files -----------------------------------------------------------------------------------------------
bool MMap::open(const char *path, std::string &err)
{
    fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("File '") + path + "' does not exist."; return false; }
    struct stat st; fstat(fd, &st); size = (size_t)st.st_size;
    ptr = size ? mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    if (size && ptr == MAP_FAILED) { ptr = nullptr; err = std::string("mmap failed for ") + path; ::close(fd); fd = -1; return false; }
    return true;
}
void MMap::close() { if (ptr) munmap(ptr, size); ptr = nullptr; if (fd >= 0) ::close(fd); fd = -1; }

bool writeFile(const char *path, const void *data, size_t size, std::string &err)
{
    int fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, S_IRWXU | S_IRGRP | S_IROTH);
    if (fd < 0) { err = std::string("cannot create ") + path; return false; }
    size_t done = 0; const char *p = (const char *)data;
    while (done < size) { ssize_t w = ::write(fd, p + done, size - done); if (w < 0) { err = "write error"; ::close(fd); return false; } done += (size_t)w; }
    ::close(fd); return true;
}

// ---- .nib2 -----------------------------------------------------------------------------------------------
int Genome::findSeq(uint32_t off) const
{ for (size_t i = 0; i < seqs.size(); i++) if (off >= seqs[i].start && off < seqs[i].start + seqs[i].length) return (int)i; return -1; }

bool compressFasta(const char *fastaPath, std::vector<uint8_t> &img, std::string &err)
{
    MMap in; if (!in.open(fastaPath, err)) return false;
    const char *g = (const char *)in.ptr; const size_t gsize = in.size;
    struct Seq { std::string name; uint32_t startByte, length; };
    std::vector<Seq> seqs; std::vector<uint8_t> out; out.reserve(gsize / 2 + 16);
    int baseCount = 0; bool have = false;
    auto finalize = [&]() {                                           // finalizeBaseSequence, Compress.c:199-218
        if (!have) return;
        seqs.back().length = (uint32_t)baseCount;
        if (baseCount & 1) out.back() = (uint8_t)(out.back() + 14);
        while (out.size() & 3) out.push_back(0xEE);
    };
    size_t off = 0;
    while (off < gsize) {
        char c = g[off];
        if (c >= (char)0 && c <= (char)31) { off++; continue; }
        if (c == '>') {
            finalize(); baseCount = 0; have = true;
            size_t nl = off + 1; while (nl < gsize && g[nl] != '\n') nl++;
            std::string name(g + off + 1, nl - off - 1);
            size_t sp = name.find(' '); if (sp != std::string::npos) name.resize(sp);
            seqs.push_back({name, (uint32_t)out.size(), 0});
            off = nl + 1; continue;
        }
        uint8_t code = map8to4((unsigned char)c); off++;
        baseCount++;
        if (baseCount & 1) out.push_back((uint8_t)(code << 4)); else out.back() = (uint8_t)(out.back() + code);
    }
    finalize();
    // header, outputBaseSequences Compress.c:140-191
    uint32_t totName = 0; for (auto &s : seqs) totName += (uint32_t)s.name.size();
    uint32_t totNameR = (totName + 3) & ~3u, n = (uint32_t)seqs.size(), preamble = 20 + 16 * n + totNameR;
    img.clear(); img.reserve(preamble + out.size());
    auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) img.push_back((uint8_t)(v >> (8 * k))); };
    put32(0x01020304u); put32(2); put32(preamble); put32(n);
    uint32_t nameOff = 0;
    for (auto &s : seqs) { put32(s.startByte); put32(s.length); put32(nameOff); put32((uint32_t)s.name.size()); nameOff += (uint32_t)s.name.size(); }
    put32(0);
    for (auto &s : seqs) img.insert(img.end(), s.name.begin(), s.name.end());
    for (uint32_t k = totName; k < totNameR; k++) img.push_back(0);
    img.insert(img.end(), out.begin(), out.end());
    return true;
}

bool parseNib2(const uint8_t *img, size_t size, Genome &g, std::string &err)
{
    if (size < 20) { err = "Input nib2 file bad header format."; return false; }
    const uint32_t *h = (const uint32_t *)img;
    int version = (int)h[1];
    if (h[0] != 0x01020304u || (version != 1 && version != 2)) { err = "Input nib2 file bad header format."; return false; }
    int blk = version == 2 ? 16 : 12; uint32_t n = h[3];
    g.bases = img + h[2]; g.nBaseBytes = size - h[2];
    const char *nameStart = (const char *)img + 16 + (size_t)blk * n + 4;
    const uint32_t *p = h + 4; g.seqs.clear();
    for (uint32_t i = 0; i < n; i++) {
        BaseSeq s; s.start = p[0] * 2; s.length = p[1];                   // normalizeBaseSequences, BaseSeq.c:115-119
        if (version == 1) { uint32_t ni = p[2]; s.name.assign(nameStart + (uint16_t)(ni >> 16), ni & 0xFFFF); p += 3; }
        else { s.name.assign(nameStart + p[2], p[3]); p += 4; }
        g.seqs.push_back(s);
    }
    g.maxROff = n ? g.seqs.back().start + g.seqs.back().length : 0;       // baseSequencesMaxROff
    return true;
}
bool loadNib2(const char *path, Genome &g, std::string &err)
{ if (!g.map.open(path, err)) return false; return parseNib2((const uint8_t *)g.map.ptr, g.map.size, g, err); }

// ---- index -----------------------------------------------------------------------------------------------
namespace {
// generateMatches4to2Fast, Index.c:32-43
inline int hashAt(const uint8_t *b, uint32_t off, int len, uint32_t *hash)
{ uint32_t r = 0; for (int i = 0; i < len; i++) { uint8_t c = get4(b, off + i); if (c > 3) return (int)(off + i + 1); r = (r << 2) + c; } *hash = r; return 0; }

template <class Visit> void scanKmers(const Genome &g, int wordLen, int skipDist, Visit visit)   // the loop of Index.c:98-128 / 201-242
{
    const uint8_t *b = g.bases; const uint32_t hashMask = 0xFFFFFFFFu >> (32 - 2 * wordLen); const int save = wordLen - skipDist;
    uint32_t hashCode = 0, partial = 0; const uint64_t nBases = g.nBaseBytes * 2;
    for (auto &s : g.seqs) {
        if ((int64_t)s.length < wordLen) continue;                     // (the reference's unsigned endingOffset would wrap; sequences shorter than a k-mer hold no k-mers)
        uint32_t baseOff = s.start, endingOffset = s.start + s.length - wordLen;
        int bad = hashAt(b, baseOff, wordLen, &hashCode);
        for (;;) {
            if (bad != 0) {
                while ((uint64_t)(uint32_t)bad < nBases && get4(b, (uint32_t)bad) > 3) bad++;
                baseOff = (uint32_t)(((bad + (skipDist - 1)) / skipDist) * skipDist);
                if (baseOff > endingOffset) break;
                bad = hashAt(b, baseOff, wordLen, &hashCode); continue;
            }
            visit(hashCode, baseOff);
            baseOff += skipDist; if (baseOff > endingOffset) break;
            bad = hashAt(b, baseOff + save, skipDist, &partial);
            hashCode = ((hashCode << (skipDist * 2)) | partial) & hashMask;
        }
    }
}
}  // namespace

// 4^15 counters are 4.3 GB: back them with transparent huge pages, otherwise first-touch page faults dominate the build
bool IndexImage::alloc(size_t n)
{
    release(); words = n; bytes = (n * 4 + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1);
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) { bytes = 0; words = 0; return false; }
    madvise(m, bytes, MADV_HUGEPAGE); p = (uint32_t *)m; return true;
}
void IndexImage::release() { if (p) munmap(p, bytes); p = nullptr; bytes = 0; words = 0; }

struct HugeU32 {
    uint32_t *p = nullptr; size_t bytes = 0;
    explicit HugeU32(size_t n) { bytes = (n * 4 + (2u << 20) - 1) & ~((size_t)(2u << 20) - 1);
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); if (m != MAP_FAILED) { madvise(m, bytes, MADV_HUGEPAGE); p = (uint32_t *)m; } }
    ~HugeU32() { if (p) munmap(p, bytes); }
    uint32_t &operator[](size_t i) { return p[i]; }
};

bool buildIndex(const Genome &g, int wordLen, int skipDist, int maxHits, IndexImage &image, FILE *log)
{
    const uint64_t HT = 1ull << (2 * wordLen);
    HugeU32 counts(HT);
    if (!counts.p) return false;
    scanKmers(g, wordLen, skipDist, [&](uint32_t h, uint32_t) { counts[h]++; });
    uint32_t total = 0; for (uint64_t i = 0; i < HT; i++) total += counts[i];
    if (!image.alloc(4 + HT + 1 + (uint64_t)total)) return false;
    image[0] = 0xFFFFFFFFu; image[1] = (uint32_t)wordLen; image[2] = (uint32_t)maxHits; image[3] = total;
    uint32_t *SO = image.p + 4, *ROA = SO + HT + 1;
    { uint32_t off = 0; for (uint64_t i = 0; i < HT; i++) { SO[i] = off; off += counts[i]; } SO[HT] = total; }
    memset(counts.p, 0, HT * 4);
    scanKmers(g, wordLen, skipDist, [&](uint32_t h, uint32_t off) { uint32_t lim = SO[h + 1] - SO[h]; if (counts[h] < lim) { ROA[SO[h] + counts[h]] = off; counts[h]++; } });
    // third pass: sample k-mers with more than maxHits occurrences (Index.c:271-315)
    RandState rs; randInitDefault(rs);
    if (log) fprintf(log, "Randomly Sampling hits for %d-mers that occur more than %d times in the reference.\n", wordLen, maxHits);
    std::vector<uint32_t> samples((size_t)maxHits > 0 ? (size_t)maxHits : 1);
    uint32_t newTotal = 0, over = 0;
    for (uint64_t i = 0; i < HT; i++) {
        uint32_t lim = SO[i + 1] - SO[i]; const uint32_t *src = ROA + SO[i];
        if (lim > (uint32_t)maxHits) { over++; randSample(rs, src, (int)lim, samples.data(), maxHits); src = samples.data(); lim = (uint32_t)maxHits; }
        uint32_t *dst = ROA + newTotal;
        for (uint32_t j = 0; j < lim; j++) dst[j] = src[j];
        SO[i] = newTotal; newTotal += lim;
    }
    SO[HT] = newTotal; image[3] = newTotal;
    image.words = 4 + HT + 1 + (uint64_t)newTotal;
    if (log) fprintf(log, "%u %d-mers had more than %d hits.\n", over, wordLen, maxHits);
    return true;
}

bool parseIndex(const uint32_t *img, size_t bytes, IndexFile &ix, std::string &err)
{
    if (bytes < 16 || (int32_t)img[0] != -1) { err = "Index file version is out of date.\nPlease remake index file and try again."; return false; }
    ix.wordLen = (int)img[1]; ix.maxHits = (int)img[2]; ix.totalMatches = img[3];
    uint64_t HT = 1ull << (2 * ix.wordLen);
    if (bytes < 4 * (4 + HT + 1 + (uint64_t)ix.totalMatches)) { err = "Index file is truncated."; return false; }
    ix.SO = img + 4; ix.ROA = ix.SO + HT + 1;
    return true;
}
bool loadIndex(const char *path, IndexFile &ix, std::string &err)
{ if (!ix.map.open(path, err)) return false; return parseIndex((const uint32_t *)ix.map.ptr, ix.map.size, ix, err); }

}  // namespace yaha
