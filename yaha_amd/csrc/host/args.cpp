// args.cpp -- command line, defaults and derived parameters.  Same option names, defaults, value checks and
// file-name derivation as the reference CLI (Main.c:187-565, AlignArgs.c:27-169) so that
// `yaha -g genome.fa` / `yaha -x index -q reads ...` keep working unchanged.  Extra options of this
// implementation: -gpus N (shard batches over N devices), -ctx M (contexts per device), -device D, -batch N (reads per device batch).
#include "yaha_host.h"
#include <cstring>
#include <cstdlib>
#include <algorithm>

namespace yaha {

static void usage(FILE *o)
{
    fputs("Usage (defaults in parentheses):\n\n"
          "Index creation:\n"
          "  yaha -g genome.{fa|fna|fasta|nib2} [-H maxHits (65525)] [-L wordLen (15)] [-S skipDist (1)] [-device D (0)] [-cpuindex]\n"
          "  yaha -g genome.{fa|fna|fasta} -c   (compress to genome.nib2 only)      yaha -g genome.nib2 -u   (back to genome.fasta)\n"
          "       (built on the GPU when one is visible and -S is 1; -cpuindex forces the host builder; the files are identical)\n\n"
          "Query alignment (hot path on MI355X):\n"
          "  yaha -x indexFile [-q queryFile|(stdin)] [-o8|(-osh)|-oss outFile|(stdout)] [-t hostThreads (1)]\n"
          "       [-gpus N (1)] [-ctx contextsPerGpu (3)] [-device D (0)] [-batch readsPerBatch (about 16 M bases)] [-dpf Y|N (Y: post-filter on the device)]\n"
          "  general : [-BW 5] [-G 50] [-H 650] [-M 25] [-MD 50] [-P 0.9] [-X 25]\n"
          "  scoring : [-AGS Y|N] [-GEC 2] [-GOC 5] [-MS 1] [-RC 3]\n"
          "  OQC     : [-OQC Y|N] [-BP 5] [-MGDP 5] [-MNO minMatch]   FBS: [-FBS Y|N] [-PRL 0.9] [-PSS 0.9]\n"
          "  -o8 modified Blast8, -osh SAM hard clipping, -oss SAM soft clipping.\n", o);
}

static bool parseBool(const char *s, const char *key, bool &out)
{
    if (strlen(s) == 1) { if (strchr("YyTt", s[0])) { out = true; return true; } if (strchr("NnFf", s[0])) { out = false; return true; } }
    fprintf(stderr, "%s is not a valid value for parameter %s.\nUse one of 'YyTt' for Yes and 'NnFf' for No.\n\n", s, key); usage(stderr); return false;
}
static bool parseInt(const char *s, const char *key, int &out)
{
    out = atoi(s);
    if (out < 0) { fprintf(stderr, "%s is not a valid value for parameter %s.\nValue must be a positive integer.\n\n", s, key); usage(stderr); return false; }
    return true;
}
static bool parseFloat(const char *s, const char *key, float &out)
{
    out = (float)atof(s);                                           // float on purpose: Main.c:161-171
    if (out <= 0.0 || out > 1.0) { fprintf(stderr, "%s is not a valid value for parameter %s.\nValue must be in the range 0<value<=1.0.\n\n", s, key); usage(stderr); return false;
        }
    return true;
}

int parseArgs(int argc, char **argv, Args &a)
{
    if (argc <= 1) { usage(stderr); return 1; }
    bool query = false, index = true;
    for (int x = 1; x < argc; x++) {
        const char *k = argv[x]; auto is = [&](const char *s) { return strcmp(k, s) == 0; };
        auto val = [&]() -> const char * { x++; return x < argc ? argv[x] : ""; };
        if (is("-h") || is("-?") || is("-xh")) { usage(stderr); return 1; }
        else if (is("-g")) { a.gfileName = val(); a.haveG = true; }
        // deliberate fix of Main.c:173-178, which turns these into "stdout" and then fails to open it
        else if (is("-q")) { const char *v = val(); a.qfileName = (!strcmp(v, "-") || !strcmp(v, "-stdin") || !strcmp(v, "stdin")) ? "stdin" : v; query = true; index = false; }
        else if (is("-o8")) { a.outputBlast8 = true; a.outputSAM = false; const char *v = val(); a.ofileName = (!strcmp(v, "-stdout")) ? "stdout" : v; a.haveO = true; }
        else if (is("-osh")) { a.outputBlast8 = false; a.outputSAM = true; a.hardClip = true; const char *v = val(); a.ofileName = (!strcmp(v, "-stdout")) ? "stdout" : v;
            a.haveO = true; }
        else if (is("-oss")) { a.outputBlast8 = false; a.outputSAM = true; a.hardClip = false; const char *v = val(); a.ofileName = (!strcmp(v, "-stdout")) ? "stdout" : v;
            a.haveO = true; }
        else if (is("-t")) { if (!parseInt(val(), "-t", a.numThreads)) return 2; }
        else if (is("-v")) a.verbose = true;
        else if (is("-x")) { a.xfileName = val(); a.haveX = true; query = true; index = false; }
        else if (is("-H")) { if (!parseInt(val(), "-H", a.maxHits)) return 2; }
        else if (is("-L")) { if (!parseInt(val(), "-L", a.wordLen)) return 2; }
        else if (is("-S")) { if (!parseInt(val(), "-S", a.skipDist)) return 2; }
        else if (is("-BW")) { if (!parseInt(val(), "-BW", a.bandWidth)) return 2; }
        else if (is("-G")) { if (!parseInt(val(), "-G", a.maxGap)) return 2; }
        else if (is("-M")) { if (!parseInt(val(), "-M", a.minMatch)) return 2; }
        else if (is("-MD")) { if (!parseInt(val(), "-MD", a.maxDesert)) return 2; }
        else if (is("-P")) { if (!parseFloat(val(), "-P", a.minIdentity)) return 2; }
        else if (is("-X")) { if (!parseInt(val(), "-X", a.XCutoff)) return 2; }
        else if (is("-AGS")) { if (!parseBool(val(), "-AGS", a.affineGapScoring)) return 2; }
        else if (is("-GEC")) { if (!parseInt(val(), "-GEC", a.GECost)) return 2; }
        else if (is("-GOC")) { if (!parseInt(val(), "-GOC", a.GOCost)) return 2; }
        else if (is("-MS")) { if (!parseInt(val(), "-MS", a.MScore)) return 2; }
        else if (is("-RC")) { if (!parseInt(val(), "-RC", a.RCost)) return 2; }
        else if (is("-OQC")) { if (!parseBool(val(), "-OQC", a.OQC)) return 2; }
        else if (is("-BP")) { if (!parseInt(val(), "-BP", a.BPCost)) return 2; }
        else if (is("-MGDP")) { if (!parseInt(val(), "-MGDP", a.maxBPLog)) return 2; }
        else if (is("-MNO")) { if (!parseInt(val(), "-MNO", a.OQCMinNonOverlap)) return 2; }
        else if (is("-FBS")) { if (!parseBool(val(), "-FBS", a.FBS)) return 2; }
        else if (is("-PRL")) { if (!parseFloat(val(), "-PRL", a.FBS_PSLength)) return 2; }
        else if (is("-PSS")) { if (!parseFloat(val(), "-PSS", a.FBS_PSScore)) return 2; }
        else if (is("-I")) { if (!parseInt(val(), "-I", a.maxIntron)) return 2; }          // experimental builds of the reference, Main.c:418-435
        else if (is("-R")) { if (!parseInt(val(), "-R", a.minRawScore)) return 2; }
        else if (is("-c")) { a.compress = true; index = false; }                                // the two below: Main.c:284-293 (builds of the reference without COMPILE_USER_MODE)
        else if (is("-u")) { a.uncompress = true; index = false; }
        else if (is("-gpus")) { if (!parseInt(val(), "-gpus", a.gpus)) return 2; if (a.gpus < 1) { fprintf(stderr, "-gpus must be at least 1.\n\n"); usage(stderr); return 2; } }
        else if (is("-ctx")) { if (!parseInt(val(), "-ctx", a.ctxPerGpu)) return 2; if (a.ctxPerGpu < 1 || a.ctxPerGpu > 8) { fprintf(stderr, "-ctx must be between 1 and 8.\n\n");
            usage(stderr); return 2; } }
        else if (is("-device")) { if (!parseInt(val(), "-device", a.device)) return 2; }
        else if (is("-cpuindex")) a.cpuIndex = true;
        else if (is("-dpf")) { if (!parseBool(val(), "-dpf", a.devicePostFilter)) return 2; }        // post-filter (OQC / FBS / MAPQ) on the device (default) or on the host
        else if (is("-batch")) { if (!parseInt(val(), "-batch", a.batchReads)) return 2;
                                 if (a.batchReads < 1 || a.batchReads > 65536) { fprintf(stderr, "-batch must be between 1 and 65536 (reads per device batch).\n\n"); usage(stderr);
                                     return 2; } }
        else { fprintf(stderr, "%s is not a valid option.\n\n", k); usage(stderr); return 2; }
    }
    a.query = query; a.index = index && !query;
    if ((a.compress || a.uncompress) && !query) {                                                  // Main.c:472-533: -c wants a FASTA genome, -u a .nib2
        if (!a.haveG) { fprintf(stderr, "Genome file specification (-g) is required for index creation.\n\n"); usage(stderr); return 2; }
        size_t dot = a.gfileName.rfind('.'); const std::string ext = dot == std::string::npos ? "" : a.gfileName.substr(dot);
        const bool fasta = ext == ".fna" || ext == ".fa" || ext == ".fasta";
        if (!fasta && ext != ".nib2") { fprintf(stderr, "Expecting a \".fa\", \".fna\", \".fasta\", or \".nib2\" genome file.\n"); return 2; }
        if (a.compress && !fasta) { fprintf(stderr, "Expecting a \".fa\", \".fna\", or \".fasta\" genome file.\n"); return 2; }
        if (a.uncompress && !a.compress && fasta) { fprintf(stderr, "Expecting a \".nib2\" genome file.\n"); return 2; }
        a.ofileName = a.gfileName.substr(0, dot) + (a.compress ? ".nib2" : ".fasta");
        postProcessArgs(a, false);
        return 0;
    }
    if (a.index) {
        if (!a.haveG) { fprintf(stderr, "Genome file specification (-g) is required for index creation.\n\n"); usage(stderr); return 2; }
        if (a.haveO) { fprintf(stderr, "Output file specification is not allowed during index creation.\n\n"); usage(stderr); return 2; }
    }
    if (query) {
        if (a.haveG) { fprintf(stderr, "Genome file specification (-g) is not allowed for query alignment.\n"); usage(stderr); return 2; }
        if (!a.haveX) { fprintf(stderr, "Index file specification (-x) is required for query alignment.\n"); usage(stderr); return 2; }
        size_t dot = a.xfileName.rfind('.');                             // Main.c:493-501
        if (dot == std::string::npos) { fprintf(stderr, "Specified index filename has improper or missing file extension.  Is it an index file?\n"); return 2; }
        a.gfileName = a.xfileName.substr(0, dot) + ".nib2";
        if (!a.haveO) { a.outputBlast8 = false; a.outputSAM = true; a.hardClip = true; a.ofileName = "stdout"; }
    }
    postProcessArgs(a, query);
    return 0;
}

void postProcessArgs(Args &a, bool query)
{
    if (a.maxIntron == -1) a.maxIntron = a.maxGap;
    if (a.minRawScore == -1) a.minRawScore = a.minMatch;
    if (a.OQCMinNonOverlap == -1) a.OQCMinNonOverlap = a.minMatch;
    if (a.OQCMinNonOverlap <= 0) { fprintf(stderr, "MNO parameter must be >=1.  MNO=1 will be used.\n"); a.OQCMinNonOverlap = 1; }
    if (a.minNonOverlap == -1) a.minNonOverlap = a.OQCMinNonOverlap;
    if (!a.affineGapScoring) { a.MScore = 1; a.RCost = a.GECost = 1; a.GOCost = 0; }
    int len = 1, score = 0, target = std::min(a.RCost, a.GOCost + a.GECost);
    while (score <= target) { score += a.MScore; len += 1; if (a.MScore <= 0) break; }
    a.minExtLength = (uint8_t)len;
    if (a.maxHits == -1) a.maxHits = query ? 650 : 0xFFFF - 10; else a.maxHits = std::min(a.maxHits, 0xFFFF - 10);
    if (a.maxBPLog < 1) { fprintf(stderr, "MGDP parameter must be between 1 and 9 (inclusive). MGDP=1 will be used.\n"); a.maxBPLog = 1; }
    if (a.maxBPLog > 9) { fprintf(stderr, "MGDP parameter must be between 1 and 9 (inclusive). MGDP=9 will be used.\n"); a.maxBPLog = 9; }
}

void paramsFromArgs(const Args &a, ygpu_params &p)
{
    p.wordLen = a.wordLen; p.maxHits = a.maxHits; p.bandWidth = a.bandWidth; p.maxGap = a.maxGap; p.maxIntron = a.maxIntron;
    p.minMatch = a.minMatch; p.maxDesert = a.maxDesert; p.minNonOverlap = a.minNonOverlap; p.minRawScore = a.minRawScore;
    p.minExtLength = a.minExtLength; p.GOCost = a.GOCost; p.GECost = a.GECost; p.RCost = a.RCost; p.MScore = a.MScore;
    p.XCutoff = a.XCutoff; p.minIdentity = a.minIdentity;
}

std::string samHeader(const Args &a, const Genome &g)                   // outputFileHeader, AlignOutput.c:30-111
{
    if (!a.outputSAM) return "";
    std::string h = "@HD\tVN:1.0\n"; char buf[512];
    for (auto &s : g.seqs) { h += "@SQ\tSN:" + s.name; snprintf(buf, sizeof buf, "\tLN:%u\n", s.length); h += buf; }
    h += "@PG\tID:YAHA\tVN:0.1.83\tCL:yaha";
    h += " -q " + a.qfileName + " -x " + a.xfileName; h += a.hardClip ? " -osh " : " -oss "; h += a.ofileName;
    snprintf(buf, sizeof buf, " -t %d -BW %d -G %d -H %d -M %d -MD %d -P %4.2f -X %d", a.numThreads, a.bandWidth, a.maxGap, a.maxHits, a.minMatch, a.maxDesert, a.minIdentity,
        a.XCutoff); h += buf;
    if (a.affineGapScoring) { snprintf(buf, sizeof buf, " -AGS Y -GEC %d -GOC %d -MS %d -RC %d", a.GECost, a.GOCost, a.MScore, a.RCost); h += buf; } else h += " -AGS N";
    if (a.OQC) {
        snprintf(buf, sizeof buf, " -OQC Y -BP %d -MGDP %d -MNO %d", a.BPCost, a.maxBPLog, a.OQCMinNonOverlap); h += buf;
        if (a.FBS) { snprintf(buf, sizeof buf, " -FBS Y -PRL %4.2f -PSS %4.2f", a.FBS_PSLength, a.FBS_PSScore); h += buf; } else h += " -FBS N";
    } else h += " -OQC N";
    h += "\n";
    return h;
}
}  // namespace yaha
