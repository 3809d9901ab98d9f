// sam.cpp -- SAM / modified-Blast8 record writer (reference AlignOutput.c:115-321).  A record is written straight into the batch's text buffer
// (yaha::Text) through a cursor: its size is bounded before the first byte (id, names, 11 characters per number, the query and quality slices, one
// character per reference base an R or D op spans), so nothing in the record checks for space or calls the C library's formatter.
#include "yaha_host.h"
#include <cstring>

namespace yaha {
namespace {
inline char *putU(char *w, uint32_t v)                                   // decimal, no sign
{
    char t[12]; int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *w++ = t[--n];
    return w;
}
inline char *putI(char *w, int v) { if (v < 0) { *w++ = '-'; return putU(w, (uint32_t)(-(int64_t)v)); } return putU(w, (uint32_t)v); }
inline char *putS(char *w, const char *s, size_t n) { memcpy(w, s, n); return w + n; }
inline char *putHex2(char *w, unsigned v) { static const char H[] = "0123456789ABCDEF"; *w++ = H[(v >> 4) & 15]; *w++ = H[v & 15]; return w; }
}  // namespace

void printClump(const Args &a, const Genome &g, const Read &r, const OutClump &oc, int primaryCount, Text &out)
{
    const ygpu_clump &c = oc.c;
    uint32_t seqStart = c.sro, seqEnd = c.sro + c.refLen - 1;
    int si = g.findSeq(seqStart);
    if (si < 0 || seqEnd >= g.seqs[si].start + g.seqs[si].length) return;          // spans two sequences: silently dropped, :129-136
    const BaseSeq &bs = g.seqs[si];
    seqStart -= bs.start; seqEnd -= bs.start;
    const bool reversed = (oc.status & 0x01) != 0;
    const std::string &queryBuf = reversed ? r.rev : r.fwd;
    const int qlen = r.len();
    if (a.outputSAM) {
        // bound: fixed text and tags < 160; CIGAR <= 12 per op + two clips; MD <= 13 per op + one character per reference base of the clump
        char *const w0 = out.room(r.id.size() + bs.name.size() + 2 * (size_t)qlen + 32 * (size_t)c.n_ops + (size_t)c.refLen + 256);
        char *w = w0;
        w = putS(w, r.id.data(), r.id.size()); *w++ = '\t'; w = putU(w, reversed ? 0x10 : 0); *w++ = '\t'; w = putS(w, bs.name.data(), bs.name.size()); *w++ = '\t';
        w = putU(w, seqStart + 1); *w++ = '\t'; w = putU(w, oc.mapQuality); *w++ = '\t';
        const char clipCode = a.hardClip ? 'H' : 'S';
        const int clipBack = qlen - 1 - c.eqo, clipFront = c.sqo;
        // CIGAR :173-190 (M and R merge into M)
        int matches = 0;
        if (clipFront > 0) { w = putI(w, clipFront); *w++ = clipCode; }
        for (uint32_t k = 0; k < c.n_ops; k++) {
            const char code = YGPU_OP_CODE(oc.ops[k]); const int len = (int)YGPU_OP_LEN(oc.ops[k]);
            if (code == 'M' || code == 'R') { matches += len; continue; }
            if (matches > 0) { w = putI(w, matches); *w++ = 'M'; matches = 0; }
            w = putI(w, len); *w++ = code;
        }
        if (clipBack > 0) { if (matches > 0) { w = putI(w, matches); *w++ = 'M'; matches = 0; } w = putI(w, clipBack); *w++ = clipCode; }
        if (matches > 0) { w = putI(w, matches); *w++ = 'M'; }
        w = putS(w, "\t*\t0\t0\t", 7);
        int qstart = 0, qend = qlen - 1;
        if (a.hardClip) { qstart = c.sqo; qend = c.eqo; }
        if (qend >= qstart) w = putS(w, queryBuf.data() + qstart, (size_t)(qend - qstart + 1));
        *w++ = '\t';
        // sic :206-212
        if (a.fastq) { if (reversed) for (int i = qend; i >= qstart; i--) *w++ = r.qual[i];
            else if (qend >= qstart) w = putS(w, r.qual.data() + qstart, (size_t)(qend - qstart + 1)); }
        else *w++ = '*';
        *w++ = '\t';
        w = putS(w, "AS:i:", 5); w = putI(w, c.totScore); w = putS(w, "\tNM:i:", 6); w = putI(w, c.gapBases + c.mismatchedBases); w = putS(w, "\tMD:Z:", 6);
        // MD :225-273 (the clip ops sit in the list as well: they only reset `previous`)
        matches = 0; char previous = clipFront > 0 ? clipCode : 'U'; uint32_t cur = c.sro;
        for (uint32_t k = 0; k < c.n_ops; k++) {
            const char code = YGPU_OP_CODE(oc.ops[k]); const int len = (int)YGPU_OP_LEN(oc.ops[k]);
            if (code == 'M') { matches += len; cur += len; }
            else if (code == 'R') {
                if (matches > 0) { w = putI(w, matches); matches = 0; }
                if (previous == 'D') *w++ = '0';
                for (int i = 0; i < len; i++) *w++ = kFourBitChars[get4(g.bases, cur + i)];
                cur += len;
            } else if (code == 'D') {
                if (matches > 0) { w = putI(w, matches); matches = 0; }
                *w++ = '^';
                for (int i = 0; i < len; i++) *w++ = kFourBitChars[get4(g.bases, cur + i)];
                cur += len;
            }
            previous = code;
        }
        if (matches > 0) w = putI(w, matches);
        w = putS(w, "\tYF:H:", 6); w = putHex2(w, oc.status);
        if (a.OQC) {
            w = putS(w, "\tYI:i:", 6); w = putI(w, oc.matchedPrimary); w = putS(w, "\tYP:i:", 6); w = putI(w, primaryCount);
            if (oc.status & 0x20) { w = putS(w, "\tYS:i:", 6); w = putI(w, oc.numSecondaries); }
        }
        *w++ = '\n';
        out.len += (size_t)(w - w0);
    }
    if (a.outputBlast8) {                                                          // :307-318
        char *const w0 = out.room(r.id.size() + bs.name.size() + 512); char *w = w0;
        w = putS(w, r.id.data(), r.id.size()); *w++ = '\t'; w = putS(w, bs.name.data(), bs.name.size());
        w += snprintf(w, 160, "\t%4.2f\t%d\t%d\t%d", 0.8 * 100, c.totLength, c.mismatchedBases, c.gapBases);
        if (reversed) w += snprintf(w, 160, "\t%d\t%d\t%d\t%d\t%c", qlen - c.eqo, qlen - c.sqo, seqEnd + 1, seqStart + 1, '-');
        else w += snprintf(w, 160, "\t%d\t%d\t%d\t%d\t%c", c.sqo + 1, c.eqo + 1, seqStart + 1, seqEnd + 1, '+');
        w += snprintf(w, 160, "\t%d\t%d\t%4.2f\n", c.totScore, qlen, ((double)c.matchedBases / qlen) * 100);
        out.len += (size_t)(w - w0);
    }
}
}  // namespace yaha
