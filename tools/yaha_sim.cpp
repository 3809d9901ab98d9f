// yaha_sim -- seeded synthetic genome / read simulator (own code, own RNG; deterministic on every box).
//
// hg18 is not available, so every parity fixture and every bench input comes from here.
//
//   yaha_sim genome --seed S --out g.fa --seqs N --len L [--repeat-frac F] [--nrun K] [--lowcomplex K]
//       N sequences whose lengths follow hg18-like ratios (or all equal with --equal), 41 % GC,
//       a fraction F of bases overwritten by diverged copies of repeat families (Alu-like 300 bp,
//       L1-like fragments of a 6 kbp consensus, microsatellites), optional N-runs, odd sequence lengths.
//   yaha_sim reads  --seed S --genome g.fa --out r.fa --n N --len L --div D [--fastq] [--chimeric P]
//                   [--withN P] [--edges] [--cgr K]
//       wgsim-like reads: uniform position and strand, per-base divergence D with substitutions : indels
//       = 3.3 : 1 and geometric indel lengths (p = 0.3); names carry the true locus.
//       --chimeric P : fraction of reads built from two loci (deletion / inversion / distal) for OQC tests.
//       --cgr K      : every read is a contig of 2 .. K+1 rearranged segments (CGR-like, testdata/README.txt:25-29).
//       --edges      : additionally emit reads touching offset 0 and the last base of every sequence.
//   yaha_sim sv     --seed S --genome g.fa --events E.sim --out r.fa [--per N] [--pad 500] [--len 500] [--cov 5] [--div 0.02]
//       Split-read sets after the reference's testdata/README.txt:9-23 (BASELINE config 5): SV event contigs as SVsim writes them, sampled wgsim-style.
//       E.sim has the two line formats of the reference's own inputs: `TYPE min max step` with TYPE in DEL / DUP / INR / INV (RandomSV_Events.sim:1-4) --
//       for every size min, min+step .. max, N events of that size at random loci -- and `INS chr start end strand name` (Alu_Insertions.sim) -- N
//       insertions of the source segment chr:[start,end) at random loci.  An event contig is `pad` bases of flank on either side of: nothing (DEL: the s
//       bases are gone), the segment twice (DUP, tandem), the segment reverse-complemented (INV), a distal segment of s bases (INR), the source segment (INS:
//       a 300-bp repeat-family copy flanked on both sides -- three segments inside one 500-mer).  Reads: `len`-mers at uniform positions of the contig,
//       either strand, cov x contig length / len of them, mutated like `reads` does.
//   yaha_sim genome ... --repeat-bed F  additionally lists the Alu-like copies it placed (chr, start, end, strand, family, divergence): sources for INS lines.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) { uint64_t z = (x += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() { uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? (uint64_t)(uni() * n) : 0; }
};

static const char BASES[4] = {'A', 'C', 'G', 'T'};
static char randBase(Rng &r, double gc) { double u = r.uni(); if (u < gc / 2) return 'C'; if (u < gc) return 'G'; if (u < gc + (1 - gc) / 2) return 'A'; return 'T'; }
static char otherBase(Rng &r, char b) { for (;;) { char c = BASES[r.below(4)]; if (c != b) return c; } }
static char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

// mutate src into a diverged copy (substitutions + short indels)
static std::string mutate(Rng &r, const std::string &src, double div, double indelFrac)
{
    std::string out; out.reserve(src.size() + 16);
    for (size_t i = 0; i < src.size(); i++) {
        if (r.uni() < div) {
            if (r.uni() < indelFrac) {
                int len = 1; while (r.uni() > 0.3 && len < 30) len++;
                if (r.uni() < 0.5) { for (int k = 0; k < len; k++) out.push_back(BASES[r.below(4)]); out.push_back(src[i]); }   // insertion
                else i += len - 1;                                                                                              // deletion
            } else out.push_back(src[i] == 'N' ? 'N' : otherBase(r, src[i]));
        } else out.push_back(src[i]);
    }
    return out;
}

struct Genome { std::vector<std::string> names; std::vector<std::string> seqs; };

static void writeFasta(const char *path, const Genome &g, int width = 60)
{
    FILE *f = fopen(path, "w"); if (!f) { perror(path); exit(1); }
    for (size_t i = 0; i < g.seqs.size(); i++) {
        fprintf(f, ">%s\n", g.names[i].c_str());
        const std::string &s = g.seqs[i];
        for (size_t p = 0; p < s.size(); p += width) { fwrite(s.data() + p, 1, std::min<size_t>(width, s.size() - p), f); fputc('\n', f); }
    }
    fclose(f);
}

static Genome readFasta(const char *path)
{
    Genome g; FILE *f = fopen(path, "r"); if (!f) { perror(path); exit(1); }
    std::vector<char> buf(1 << 20); std::string *cur = nullptr; bool inName = false; std::string name;
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), f)) > 0)
        for (size_t i = 0; i < n; i++) {
            char c = buf[i];
            if (inName) { if (c == '\n') { inName = false; size_t sp = name.find(' '); if (sp != std::string::npos) name.resize(sp); g.names.push_back(name); g.seqs.emplace_back(); cur = &g.seqs.back(); } else name.push_back(c); }
            else if (c == '>') { inName = true; name.clear(); }
            else if (c > 31 && cur) cur->push_back(c);
        }
    fclose(f); return g;
}

static const char *argval(int argc, char **argv, const char *key, const char *def)
{ for (int i = 2; i + 1 < argc; i++) if (!strcmp(argv[i], key)) return argv[i + 1]; return def; }
static bool argflag(int argc, char **argv, const char *key)
{ for (int i = 2; i < argc; i++) if (!strcmp(argv[i], key)) return true; return false; }

static int cmdGenome(int argc, char **argv)
{
    uint64_t seed = strtoull(argval(argc, argv, "--seed", "1"), 0, 10);
    const char *out = argval(argc, argv, "--out", "genome.fa");
    int nseq = atoi(argval(argc, argv, "--seqs", "2"));
    uint64_t total = strtoull(argval(argc, argv, "--len", "2000000"), 0, 10);
    double repFrac = atof(argval(argc, argv, "--repeat-frac", "0.35"));
    int nruns = atoi(argval(argc, argv, "--nrun", "1"));
    int lowc = atoi(argval(argc, argv, "--lowcomplex", "2"));
    bool equal = argflag(argc, argv, "--equal");
    // number of repeat sub-families: copies per family (and so seed hits per read) scale with total / families;
    // the defaults are calibrated so that a 1 kbp read yields ~10 k seed hits at -H 650
    int nAlu = atoi(argval(argc, argv, "--alu-families", "0")), nL1 = atoi(argval(argc, argv, "--l1-families", "0"));
    if (nAlu <= 0) nAlu = (int)std::max<uint64_t>(6, total / 2500000);
    if (nL1 <= 0) nL1 = (int)std::max<uint64_t>(2, total / 8000000);
    Rng r(seed);
    Genome g;
    // hg18-like length ratios (chr1..chrN descending), odd lengths to exercise the X padding
    std::vector<double> w(nseq); double ws = 0;
    for (int i = 0; i < nseq; i++) { w[i] = equal ? 1.0 : 1.0 / (1.0 + 0.12 * i); ws += w[i]; }
    for (int i = 0; i < nseq; i++) {
        uint64_t len = (uint64_t)(total * w[i] / ws); if ((len & 1) == 0) len += 1; if (i % 3 == 2) len += 2;
        char nm[32]; snprintf(nm, sizeof nm, "chr%d", i + 1);
        g.names.push_back(nm); std::string s(len, 'A');
        for (auto &c : s) c = randBase(r, 0.41);
        g.seqs.push_back(std::move(s));
    }
    // repeat families
    std::vector<std::string> alu;   for (int f = 0; f < nAlu; f++) { std::string c(300, 'A'); for (auto &ch : c) ch = randBase(r, 0.55); alu.push_back(c); }
    std::vector<std::string> l1;    for (int f = 0; f < nL1; f++) { std::string c(6000, 'A'); for (auto &ch : c) ch = randBase(r, 0.40); l1.push_back(c); }
    uint64_t target = (uint64_t)(repFrac * total), placed = 0;
    const char *bedPath = argval(argc, argv, "--repeat-bed", nullptr); FILE *bed = bedPath ? fopen(bedPath, "w") : nullptr;      // (written beside the genome; consumes no random numbers)
    while (placed < target) {
        int si = (int)r.below(nseq); std::string &s = g.seqs[si];
        std::string copy; double u = r.uni(); int fam = -1; double famDiv = 0;
        if (u < 0.60) { fam = (int)r.below(alu.size()); const std::string &c = alu[fam]; famDiv = 0.03 + 0.12 * r.uni(); copy = mutate(r, c, famDiv, 0.12); }
        else if (u < 0.92) { const std::string &c = l1[r.below(l1.size())]; size_t fl = 300 + r.below(3000); size_t st = r.below(c.size() - fl); copy = mutate(r, c.substr(st, fl), 0.02 + 0.15 * r.uni(), 0.10); }
        else { int ul = 1 + (int)r.below(5); std::string unit(ul, 'A'); for (auto &ch : unit) ch = BASES[r.below(4)]; size_t reps = 8 + r.below(40); std::string ms; for (size_t k = 0; k < reps; k++) ms += unit; copy = mutate(r, ms, 0.02, 0.2); }
        bool flip = false;
        if (r.uni() < 0.5) { flip = true; std::reverse(copy.begin(), copy.end()); for (auto &ch : copy) ch = comp(ch); }
        if (copy.size() + 2 >= s.size()) continue;
        size_t pos = r.below(s.size() - copy.size());
        memcpy(&s[pos], copy.data(), copy.size()); placed += copy.size();
        if (bed && fam >= 0) fprintf(bed, "%s\t%zu\t%zu\t%c\tAluLike%d\t%.3f\n", g.names[si].c_str(), pos, pos + copy.size(), flip ? '-' : '+', fam, famDiv);
    }
    if (bed) fclose(bed);
    for (int k = 0; k < lowc; k++) { std::string &s = g.seqs[r.below(nseq)]; size_t len = std::min<size_t>(400 + r.below(600), s.size() / 4); size_t pos = r.below(s.size() - len); char a = BASES[r.below(4)], b = BASES[r.below(4)]; for (size_t i = 0; i < len; i++) s[pos + i] = (r.uni() < 0.9) ? a : b; }
    for (int k = 0; k < nruns; k++) { std::string &s = g.seqs[r.below(nseq)]; size_t len = std::min<size_t>(200 + r.below(2000), s.size() / 8); size_t pos = r.below(s.size() - len); for (size_t i = 0; i < len; i++) s[pos + i] = 'N'; }
    writeFasta(out, g);
    return 0;
}

static std::string revcomp(const std::string &s) { std::string o(s.rbegin(), s.rend()); for (auto &c : o) c = comp(c); return o; }

static int cmdReads(int argc, char **argv)
{
    uint64_t seed = strtoull(argval(argc, argv, "--seed", "7"), 0, 10);
    const char *gpath = argval(argc, argv, "--genome", "genome.fa");
    const char *out = argval(argc, argv, "--out", "reads.fa");
    uint64_t n = strtoull(argval(argc, argv, "--n", "1000"), 0, 10);
    int len = atoi(argval(argc, argv, "--len", "1000"));
    double div = atof(argval(argc, argv, "--div", "0.017"));
    double chim = atof(argval(argc, argv, "--chimeric", "0"));
    double withN = atof(argval(argc, argv, "--withN", "0"));
    bool fastq = argflag(argc, argv, "--fastq"), edges = argflag(argc, argv, "--edges");
    int lenJitter = atoi(argval(argc, argv, "--len-jitter", "0"));
    int cgr = atoi(argval(argc, argv, "--cgr", "0"));          // complex-rearrangement contigs: every read is 2 .. cgr+1 segments (deletions, tandem duplications, inversions, distal pieces)
    Genome g = readFasta(gpath);
    Rng r(seed);
    uint64_t total = 0; for (auto &s : g.seqs) total += s.size();
    FILE *f = fopen(out, "w"); if (!f) { perror(out); return 1; }
    auto pick = [&](int L, int &si, size_t &pos) {
        for (;;) { uint64_t t = r.below(total); si = 0; while (t >= g.seqs[si].size()) { t -= g.seqs[si].size(); si++; } if (g.seqs[si].size() < (size_t)L) continue; pos = std::min<size_t>(t, g.seqs[si].size() - L); return; }
    };
    auto emit = [&](const std::string &name, std::string seq) {
        if (withN > 0 && r.uni() < withN) { size_t p = r.below(seq.size()); size_t l = 1 + r.below(3); for (size_t k = p; k < std::min(seq.size(), p + l); k++) seq[k] = 'N'; }
        if (fastq) { std::string q(seq.size(), 'I'); for (auto &c : q) c = (char)(33 + 2 + r.below(39)); fprintf(f, "@%s\n%s\n+\n%s\n", name.c_str(), seq.c_str(), q.c_str()); }
        else fprintf(f, ">%s\n%s\n", name.c_str(), seq.c_str());
    };
    for (uint64_t i = 0; i < n; i++) {
        int L = len + (lenJitter ? (int)r.below(2 * lenJitter + 1) - lenJitter : 0); if (L < 20) L = 20;
        char nm[160]; std::string seq;
        if (cgr > 0) {
            const int nseg = 2 + (int)r.below((uint64_t)cgr); std::vector<int> cut;
            for (int k = 1; k < nseg; k++) cut.push_back(400 + (int)r.below((uint64_t)std::max(1, L - 800)));
            cut.push_back(0); cut.push_back(L); std::sort(cut.begin(), cut.end());
            int s0; size_t p0; pick(L, s0, p0); size_t anchor = p0; std::string all;
            for (size_t k = 0; k + 1 < cut.size(); k++) {
                const int pl = cut[k + 1] - cut[k]; if (pl <= 0) continue;
                const double u = r.uni(); int sj = s0; size_t pj = anchor; bool inv = false;
                const size_t slen = g.seqs[s0].size();
                if (k == 0) pj = p0;
                else if (u < 0.3) pj = std::min(anchor + 200 + r.below(8000), slen - pl);                                  // deletion
                else if (u < 0.5) pj = anchor > (size_t)pl + 200 ? anchor - std::min<size_t>(anchor, 200 + r.below(3000)) : anchor;   // tandem duplication (steps back)
                else if (u < 0.75) { pj = std::min(anchor + r.below(5000), slen - pl); inv = true; }                        // inversion
                else pick(pl, sj, pj);                                                                                   // distal piece
                if (pj + pl > g.seqs[sj].size()) pj = g.seqs[sj].size() - pl;
                std::string piece = g.seqs[sj].substr(pj, pl); if (inv) piece = revcomp(piece);
                all += piece; if (sj == s0) anchor = pj + pl;
            }
            seq = div > 0 ? mutate(r, all, div, 1.0 / 4.3) : all;
            snprintf(nm, sizeof nm, "cgr_%s_%zu_%dseg_%llu", g.names[s0].c_str(), p0, nseg, (unsigned long long)i);
        } else if (chim > 0 && r.uni() < chim) {
            int L1 = L / 4 + (int)r.below(L / 2), L2 = L - L1; int s1, s2; size_t p1, p2; pick(L1, s1, p1);
            double u = r.uni(); std::string a = g.seqs[s1].substr(p1, L1), b;
            if (u < 0.4 && p1 + L1 + 5000 + L2 < g.seqs[s1].size()) { s2 = s1; p2 = p1 + L1 + 100 + r.below(4000); b = g.seqs[s2].substr(p2, L2); }      // deletion
            else if (u < 0.7) { pick(L2, s2, p2); b = revcomp(g.seqs[s2].substr(p2, L2)); }                                                              // inversion / distal
            else { pick(L2, s2, p2); b = g.seqs[s2].substr(p2, L2); }
            seq = mutate(r, a + b, div, 1.0 / 4.3);
            snprintf(nm, sizeof nm, "%s_%zu_%s_%zu_chim_%llu", g.names[s1].c_str(), p1, g.names[s2].c_str(), p2, (unsigned long long)i);
        } else {
            int si; size_t pos; pick(L, si, pos);
            seq = mutate(r, g.seqs[si].substr(pos, L), div, 1.0 / 4.3);
            snprintf(nm, sizeof nm, "%s_%zu_%zu_%llu", g.names[si].c_str(), pos, pos + L, (unsigned long long)i);
        }
        bool rc = r.uni() < 0.5; if (rc) seq = revcomp(seq);
        std::string name = std::string(nm) + (rc ? "_r" : "_f");
        emit(name, seq);
    }
    if (edges) for (size_t si = 0; si < g.seqs.size(); si++) {
        const std::string &s = g.seqs[si]; int L = std::min<size_t>(len, s.size());
        emit(g.names[si] + "_edge_start", mutate(r, s.substr(0, L), div, 1.0 / 4.3));
        emit(g.names[si] + "_edge_end", mutate(r, s.substr(s.size() - L, L), div, 1.0 / 4.3));
        emit(g.names[si] + "_edge_start_rc", revcomp(s.substr(0, L)));
        emit(g.names[si] + "_edge_end_rc", revcomp(s.substr(s.size() - L, L)));
    }
    fclose(f);
    return 0;
}

// Split-read sets: SV event contigs sampled as 500-mers (testdata/README.txt:9-23, RandomSV_Events.sim, Alu_Insertions.sim)
static int cmdSV(int argc, char **argv)
{
    uint64_t seed = strtoull(argval(argc, argv, "--seed", "11"), 0, 10);
    const char *gpath = argval(argc, argv, "--genome", "genome.fa"), *epath = argval(argc, argv, "--events", "events.sim"), *out = argval(argc, argv, "--out", "sv.fa");
    const int per = atoi(argval(argc, argv, "--per", "25")), pad = atoi(argval(argc, argv, "--pad", "500")), len = atoi(argval(argc, argv, "--len", "500"));
    const double cov = atof(argval(argc, argv, "--cov", "5")), div = atof(argval(argc, argv, "--div", "0.02"));
    Genome g = readFasta(gpath); Rng r(seed);
    uint64_t total = 0; for (auto &s : g.seqs) total += s.size();
    FILE *ef = fopen(epath, "r"); if (!ef) { perror(epath); return 1; }
    FILE *f = fopen(out, "w"); if (!f) { perror(out); return 1; }
    // a locus with `need` bases of room and no N inside (SVsim draws from the assembled part of hg18)
    auto pick = [&](size_t need, int &si, size_t &pos) {
        for (int tries = 0; tries < 100000; tries++) {
            uint64_t t = r.below(total); si = 0; while (t >= g.seqs[si].size()) { t -= g.seqs[si].size(); si++; }
            if (g.seqs[si].size() < need + 2) continue;
            pos = std::min<size_t>(t, g.seqs[si].size() - need);
            if (g.seqs[si].find('N', pos) >= pos + need) return true;
        }
        return false;
    };
    auto seqIndex = [&](const char *name) { for (size_t i = 0; i < g.names.size(); i++) if (g.names[i] == name) return (int)i; return -1; };
    unsigned long long ev = 0, nreads = 0;
    auto sample = [&](const std::string &contig, const char *type, int si, size_t p, size_t sz) {
        if ((int)contig.size() < len) return;
        const int n = (int)(cov * contig.size() / len + 0.5);
        for (int k = 0; k < n; k++) {
            const size_t st = r.below(contig.size() - len + 1);
            std::string seq = div > 0 ? mutate(r, contig.substr(st, len), div, 1.0 / 4.3) : contig.substr(st, len);
            const bool rc = r.uni() < 0.5; if (rc) seq = revcomp(seq);
            fprintf(f, ">sv_%s_%s_%zu_%zu_e%llu_at%zu_%llu_%c\n%s\n", type, g.names[si].c_str(), p, sz, ev, st, nreads++, rc ? 'r' : 'f', seq.c_str());
        }
        ev++;
    };
    char line[512];
    while (fgets(line, sizeof line, ef)) {
        char type[16], chr[128], strand[8], name[128]; long a = 0, b = 0, c = 0;
        if (sscanf(line, "%15s", type) != 1 || type[0] == '#') continue;
        if (!strcmp(type, "INS")) {
            if (sscanf(line, "%*s %127s %ld %ld %7s %127s", chr, &a, &b, strand, name) != 5) { fprintf(stderr, "bad INS line: %s", line); return 1; }
            const int sj = seqIndex(chr); if (sj < 0 || a < 0 || b <= a || (size_t)b > g.seqs[sj].size()) { fprintf(stderr, "INS source outside the genome: %s", line); return 1; }
            std::string src = g.seqs[sj].substr(a, b - a); if (strand[0] == '-') src = revcomp(src);
            for (int k = 0; k < per; k++) {
                int si; size_t p; if (!pick(2 * (size_t)pad, si, p)) break; p += pad;                                  // insertion AT p: flank, source, flank
                sample(g.seqs[si].substr(p - pad, pad) + src + g.seqs[si].substr(p, pad), name, si, p, src.size());
            }
            continue;
        }
        if (sscanf(line, "%*s %ld %ld %ld", &a, &b, &c) != 3 || a < 1 || b < a || c < 1) { fprintf(stderr, "bad event line: %s", line); return 1; }
        for (long sz = a; sz <= b; sz += c) for (int k = 0; k < per; k++) {
            int si; size_t p; if (!pick((size_t)sz + 2 * (size_t)pad, si, p)) break; p += pad;                       // the event's segment is [p, p + sz)
            const std::string &s = g.seqs[si]; const std::string L = s.substr(p - pad, pad), M = s.substr(p, sz), R = s.substr(p + sz, pad);
            if (!strcmp(type, "DEL")) sample(L + R, "DEL", si, p, sz);
            else if (!strcmp(type, "DUP")) sample(L + M + M + R, "DUP", si, p, sz);
            else if (!strcmp(type, "INV")) sample(L + revcomp(M) + R, "INV", si, p, sz);
            else if (!strcmp(type, "INR")) { int sj; size_t q; if (!pick((size_t)sz, sj, q)) break; sample(L + g.seqs[sj].substr(q, sz) + s.substr(p, pad), "INR", si, p, sz); }      // distal insertion AT p
            else { fprintf(stderr, "unknown event type %s\n", type); return 1; }
        }
    }
    fclose(ef); fclose(f);
    fprintf(stderr, "yaha_sim sv: %llu events, %llu reads of %d bases\n", ev, nreads, len);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: yaha_sim genome|reads|sv [options]\n"); return 2; }
    if (!strcmp(argv[1], "genome")) return cmdGenome(argc, argv);
    if (!strcmp(argv[1], "reads")) return cmdReads(argc, argv);
    if (!strcmp(argv[1], "sv")) return cmdSV(argc, argv);
    fprintf(stderr, "unknown command %s\n", argv[1]); return 2;
}
