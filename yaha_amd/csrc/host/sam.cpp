// sam.cpp -- SAM / modified-Blast8 record writer (reference AlignOutput.c:115-321).  Text is appended to a
// std::string so that batches can be formatted by several host threads and written in input order.
#include "yaha_host.h"
#include <cstring>

namespace yaha {

static inline void appendInt(std::string &s, long v) { char b[24]; int n = snprintf(b, sizeof b, "%ld", v); s.append(b, n); }
static inline void appendUInt(std::string &s, unsigned long v) { char b[24]; int n = snprintf(b, sizeof b, "%lu", v); s.append(b, n); }

void printClump(const Args &a, const Genome &g, const Read &r, const OutClump &oc, int primaryCount, std::string &out)
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
        out += r.id; out += '\t'; appendInt(out, reversed ? 0x10 : 0); out += '\t'; out += bs.name; out += '\t';
        appendUInt(out, seqStart + 1); out += '\t'; appendUInt(out, oc.mapQuality); out += '\t';
        const char clipCode = a.hardClip ? 'H' : 'S';
        const int clipBack = qlen - 1 - c.eqo, clipFront = c.sqo;
        // CIGAR :173-190 (M and R merge into M)
        int matches = 0;
        if (clipFront > 0) { appendInt(out, clipFront); out += clipCode; }
        for (uint32_t k = 0; k < c.n_ops; k++) {
            char code = YGPU_OP_CODE(oc.ops[k]); int len = (int)YGPU_OP_LEN(oc.ops[k]);
            if (code == 'M' || code == 'R') { matches += len; continue; }
            if (matches > 0) { appendInt(out, matches); out += 'M'; matches = 0; }
            appendInt(out, len); out += code;
        }
        if (clipBack > 0) { if (matches > 0) { appendInt(out, matches); out += 'M'; matches = 0; } appendInt(out, clipBack); out += clipCode; }
        if (matches > 0) { appendInt(out, matches); out += 'M'; }
        out += "\t*\t0\t0\t";
        int qstart = 0, qend = qlen - 1;
        if (a.hardClip) { qstart = c.sqo; qend = c.eqo; }
        if (qend >= qstart) out.append(queryBuf, qstart, qend - qstart + 1);
        out += '\t';
        if (a.fastq) { if (reversed) for (int i = qend; i >= qstart; i--) out += r.qual[i]; else for (int i = qstart; i <= qend; i++) out += r.qual[i]; }   // sic :206-212
        else out += '*';
        out += '\t';
        out += "AS:i:"; appendInt(out, c.totScore); out += "\tNM:i:"; appendInt(out, c.gapBases + c.mismatchedBases); out += "\tMD:Z:";
        // MD :225-273 (the clip ops sit in the list as well: they only reset `previous`)
        matches = 0; char previous = clipFront > 0 ? clipCode : 'U'; uint32_t cur = c.sro;
        for (uint32_t k = 0; k < c.n_ops; k++) {
            char code = YGPU_OP_CODE(oc.ops[k]); int len = (int)YGPU_OP_LEN(oc.ops[k]);
            if (code == 'M') { matches += len; cur += len; }
            else if (code == 'R') {
                if (matches > 0) { appendInt(out, matches); matches = 0; }
                if (previous == 'D') out += '0';
                for (int i = 0; i < len; i++) out += kFourBitChars[get4(g.bases, cur + i)];
                cur += len;
            } else if (code == 'D') {
                if (matches > 0) { appendInt(out, matches); matches = 0; }
                out += '^';
                for (int i = 0; i < len; i++) out += kFourBitChars[get4(g.bases, cur + i)];
                cur += len;
            }
            previous = code;
        }
        if (matches > 0) appendInt(out, matches);
        char buf[64]; snprintf(buf, sizeof buf, "\tYF:H:%02X", oc.status); out += buf;
        if (a.OQC) {
            out += "\tYI:i:"; appendInt(out, oc.matchedPrimary); out += "\tYP:i:"; appendInt(out, primaryCount);
            if (oc.status & 0x20) { out += "\tYS:i:"; appendInt(out, oc.numSecondaries); }
        }
        out += '\n';
    }
    if (a.outputBlast8) {                                                          // :307-318
        char buf[512];
        out += r.id; out += '\t'; out += bs.name;
        snprintf(buf, sizeof buf, "\t%4.2f\t%d\t%d\t%d", 0.8 * 100, c.totLength, c.mismatchedBases, c.gapBases); out += buf;
        if (reversed) snprintf(buf, sizeof buf, "\t%d\t%d\t%d\t%d\t%c", qlen - c.eqo, qlen - c.sqo, seqEnd + 1, seqStart + 1, '-');
        else snprintf(buf, sizeof buf, "\t%d\t%d\t%d\t%d\t%c", c.sqo + 1, c.eqo + 1, seqStart + 1, seqEnd + 1, '+');
        out += buf;
        snprintf(buf, sizeof buf, "\t%d\t%d\t%4.2f\n", c.totScore, qlen, ((double)c.matchedBases / qlen) * 100); out += buf;
    }
}
}  // namespace yaha
