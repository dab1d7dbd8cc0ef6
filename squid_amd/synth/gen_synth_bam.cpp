// Synthetic STAR-style BAM generator for the SQUID hot path (test/bench input only).
//
// Writes <out>.bam (coordinate-sorted concordant BAM) and <out>.chim.bam (STAR
// "SeparateSAMold"-layout chimeric BAM, unsorted) plus <out>.truth.txt listing the planted
// TSV junctions.  Everything is derived from one splitmix64 stream, so a (config, seed) pair
// always produces byte-identical files.  Layout follows SURVEY.md section 8(d) / appendix G:
// gene islands of 3-8 exons, 2x100 bp reads, insert N(300,30), MAPQ 255, NH:i:1, spliced CIGARs
// where a mate crosses an exon junction, 5 % soft clips > 15 bp, planted junctions at exon
// boundaries supported by split-read triplets and discordant pairs.
//
// This file is input tooling: it is NOT part of the product data path and not part of oracle/.

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    int range(int lo, int hi) { return lo + (int)(next() % (uint64_t)(hi - lo + 1)); }  // inclusive
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

// ---------------------------------------------------------------- BGZF writer
class BgzfWriter {
public:
    BgzfWriter(const std::string& path, int level, int threads) : level_(level), threads_(std::max(1, threads)) {
        fp_ = std::fopen(path.c_str(), "wb");
        if (!fp_) { std::perror(path.c_str()); std::exit(1); }
        buf_.reserve(kBatch * kBlock);
    }
    // uncompressed offset of the next byte / file offset of every finished BGZF block: what a virtual file offset is made of
    unsigned long long tell() const { return utotal_; }
    const std::vector<unsigned long long>& block_offsets() const { return block_coff_; }
    static constexpr size_t block_size() { return kBlock; }
    void write(const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        utotal_ += n;
        buf_.insert(buf_.end(), b, b + n);
        if (buf_.size() >= (size_t)kBatch * kBlock) flush_full_blocks(false);
    }
    void close() {
        flush_full_blocks(true);
        static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                        0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::fwrite(eof, 1, 28, fp_);
        std::fclose(fp_);
        fp_ = nullptr;
    }

private:
    static constexpr size_t kBlock = 0xff00;
    static constexpr size_t kBatch = 256;
    static void compress_block(const uint8_t* src, size_t n, int level, std::vector<uint8_t>& out) {
        out.resize(18 + compressBound(n) + 8 + 64);
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = (Bytef*)src;
        zs.avail_in = (uInt)n;
        zs.next_out = out.data() + 18;
        zs.avail_out = (uInt)(out.size() - 18 - 8);
        deflate(&zs, Z_FINISH);
        size_t clen = zs.total_out;
        deflateEnd(&zs);
        static const uint8_t hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
        std::memcpy(out.data(), hdr, 16);
        uint16_t bsize = (uint16_t)(18 + clen + 8 - 1);
        out[16] = bsize & 0xff;
        out[17] = bsize >> 8;
        uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), src, (uInt)n);
        uint32_t isz = (uint32_t)n;
        std::memcpy(out.data() + 18 + clen, &crc, 4);
        std::memcpy(out.data() + 18 + clen + 4, &isz, 4);
        out.resize(18 + clen + 8);
    }
    void flush_full_blocks(bool all) {
        size_t nblocks = buf_.size() / kBlock;
        size_t tail = buf_.size() - nblocks * kBlock;
        if (all && tail) nblocks++;
        if (!nblocks) return;
        std::vector<std::vector<uint8_t>> outs(nblocks);
        auto work = [&](size_t t) {
            for (size_t i = t; i < nblocks; i += threads_) {
                size_t off = i * kBlock;
                size_t n = std::min(kBlock, buf_.size() - off);
                compress_block(buf_.data() + off, n, level_, outs[i]);
            }
        };
        if (threads_ == 1 || nblocks < 4) {
            for (int t = 0; t < threads_; ++t) work(t);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < threads_; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
        }
        for (auto& o : outs) { block_coff_.push_back(cpos_); std::fwrite(o.data(), 1, o.size(), fp_); cpos_ += o.size(); }
        size_t consumed = std::min(buf_.size(), nblocks * kBlock);
        buf_.erase(buf_.begin(), buf_.begin() + consumed);
    }
    FILE* fp_;
    int level_, threads_;
    std::vector<uint8_t> buf_;
    unsigned long long utotal_ = 0, cpos_ = 0;
    std::vector<unsigned long long> block_coff_;
};

// ---------------------------------------------------------------- BAM record builder
struct CigarOp { char op; int len; };

int reg2bin(int beg, int end) {
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (beg >> 26);
    return 0;
}

// --bwa: ONE coordinate-sorted BAM in the shape `bwa mem` writes (what squid --bwa reads, src/SegmentGraph.cpp:833-1205,1698-1930):
// MAPQ 60 / 0 instead of STAR's 255 / 3, multi-mappers carry an XA:Z tag, the pieces of a split read are a soft-clipped primary and
// a hard-clipped SUPPLEMENTARY record (0x800) of the same name and mate, every record knows its mate, and the junction-supporting
// fragments sit in the sorted stream among the concordant ones (no second file).
static bool g_bwa = false;

struct Rec {
    int32_t refid, pos;
    std::vector<uint8_t> bytes;  // full BAM record incl. block_size
};

void put32(std::vector<uint8_t>& v, int32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)((uint32_t)x >> (8 * i))); }
void put16(std::vector<uint8_t>& v, int x) { v.push_back(x & 0xff); v.push_back((x >> 8) & 0xff); }

// seqlen = number of bases stored (hard clips excluded)
Rec make_record(Rng& rng, const std::string& name, int refid, int pos, int mapq, int flag,
                const std::vector<CigarOp>& cigar, int mrefid, int mpos, int tlen, int nh, bool lowqual_run) {
    Rec r;
    r.refid = refid;
    r.pos = pos;
    int seqlen = 0, reflen = 0;
    for (auto& c : cigar) {
        if (c.op == 'M' || c.op == 'I' || c.op == 'S' || c.op == '=' || c.op == 'X') seqlen += c.len;
        if (c.op == 'M' || c.op == 'D' || c.op == 'N' || c.op == '=' || c.op == 'X') reflen += c.len;
    }
    std::vector<uint8_t>& b = r.bytes;
    put32(b, 0);  // block_size placeholder
    put32(b, refid);
    put32(b, pos);
    b.push_back((uint8_t)(name.size() + 1));
    if (g_bwa) mapq = mapq >= 255 ? 60 : 0;
    b.push_back((uint8_t)mapq);
    put16(b, reg2bin(pos, pos + std::max(reflen, 1)));
    put16(b, (int)cigar.size());
    put16(b, flag);
    put32(b, seqlen);
    put32(b, mrefid);
    put32(b, mpos);
    put32(b, tlen);
    b.insert(b.end(), name.begin(), name.end());
    b.push_back(0);
    static const char* ops = "MIDNSHP=X";
    for (auto& c : cigar) {
        int code = (int)(std::strchr(ops, c.op) - ops);
        put32(b, (int32_t)(((uint32_t)c.len << 4) | (uint32_t)code));
    }
    // bases: uniform ACGT -> 4-bit codes 1,2,4,8
    static const uint8_t code4[4] = {1, 2, 4, 8};
    for (int i = 0; i < (seqlen + 1) / 2; ++i) {
        uint64_t x = rng.next();
        uint8_t hi = code4[x & 3], lo = code4[(x >> 2) & 3];
        if (2 * i + 1 >= seqlen) lo = 0;
        b.push_back((uint8_t)(hi << 4 | lo));
    }
    // qualities: uniform Phred 20..40; optionally a run of Phred 2 (length 12..20) to trip -pl
    size_t qoff = b.size();
    for (int i = 0; i < seqlen; ++i) b.push_back((uint8_t)(20 + rng.next() % 21));
    if (lowqual_run && seqlen > 40) {
        int len = rng.range(12, 20), st = rng.range(0, seqlen - len);
        for (int i = 0; i < len; ++i) b[qoff + st + i] = 2;
    }
    if (g_bwa) {  // NM:i:0, and XA:Z:<alternative hit> on a multi-mapper
        b.push_back('N'); b.push_back('M'); b.push_back('C'); b.push_back(0);
        if (nh > 1) { static const char xa[] = "XAZchr1,+12345,100M,0;"; b.insert(b.end(), xa, xa + sizeof xa); }
    } else {
        // tags NH:i:nh HI:i:1 (STAR default attributes), types 'C'
        b.push_back('N'); b.push_back('H'); b.push_back('C'); b.push_back((uint8_t)nh);
        b.push_back('H'); b.push_back('I'); b.push_back('C'); b.push_back(1);
    }
    int32_t bs = (int32_t)b.size() - 4;
    std::memcpy(b.data(), &bs, 4);
    return r;
}

// ---------------------------------------------------------------- genome model
struct Gene {
    int chr;
    std::vector<int> es, ee;   // exon [start,end)
    std::vector<int> cum;      // transcript offset of each exon start; cum.back()=transcript length
    double weight;
    int tlen() const { return cum.back(); }
};

struct Piece { int refpos; std::vector<CigarOp> cig; int reflen; };

// map transcript interval [t0,t1) of gene g to genomic start + M/N cigar
Piece map_interval(const Gene& g, int t0, int t1) {
    Piece p;
    p.reflen = 0;
    size_t e = std::upper_bound(g.cum.begin(), g.cum.end(), t0) - g.cum.begin() - 1;
    int t = t0;
    p.refpos = g.es[e] + (t0 - g.cum[e]);
    while (t < t1) {
        int avail = g.cum[e + 1] - t;
        int take = std::min(avail, t1 - t);
        p.cig.push_back({'M', take});
        p.reflen += take;
        t += take;
        if (t < t1) {
            int gap = g.es[e + 1] - g.ee[e];
            p.cig.push_back({'N', gap});
            p.reflen += gap;
            ++e;
        }
    }
    return p;
}

struct Contig { std::string name; int len; };

const Contig kHg38[] = {{"chr1", 248956422}, {"chr2", 242193529}, {"chr3", 198295559}, {"chr4", 190214555},
                        {"chr5", 181538259}, {"chr6", 170805979}, {"chr7", 159345973}, {"chr8", 145138636},
                        {"chr9", 138394717}, {"chr10", 133797422}, {"chr11", 135086622}, {"chr12", 133275309},
                        {"chr13", 114364328}, {"chr14", 107043718}, {"chr15", 101991189}, {"chr16", 90338345},
                        {"chr17", 83257441}, {"chr18", 80373285}, {"chr19", 58617616}, {"chr20", 64444167},
                        {"chr21", 46709983}, {"chr22", 50818468}, {"chrX", 156040895}, {"chrY", 57227415},
                        {"chrM", 16569}};

struct Tsv {
    int gx, ex, gy, ey;     // gene / exon index of side X and Y
    bool xhead, yhead;      // end type at each side (false = tail: block ends at bp)
    int bpx, bpy;           // genomic breakpoint
    int txx, txy;           // transcript coordinate of the breakpoint
    int nsplit, npair;
};

struct ChimEmit { std::vector<Rec> recs; };

}  // namespace

int main(int argc, char** argv) {
    std::string config = "C1", out = "synth";
    uint64_t seed = 0;
    long records = -1;
    int ntsv = -1, level = 6, threads = 4, genes_override = -1;
    double chim_copy_frac = 0.2;
    int sup_lo = 10, sup_hi = 60;  // chimeric fragments per planted junction (--support lo,hi)
    int n_interleave = 0;          // --interleave K: K PAIRS of junctions that FilterbyInterleaving removes (see the planting loop)
    bool small_cc = false;  // config C5: compact genes, junctions in the last exon (see below)
    double indel_frac = 0.0;  // fraction of concordant pairs whose left read gets an I / D / =X CIGAR variant (off by default: C1..C5 unchanged)
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
        if (a == "--config") config = val();
        else if (a == "--out") out = val();
        else if (a == "--seed") seed = std::strtoull(val().c_str(), nullptr, 10);
        else if (a == "--records") records = std::atol(val().c_str());
        else if (a == "--tsv") ntsv = std::atoi(val().c_str());
        else if (a == "--level") level = std::atoi(val().c_str());
        else if (a == "--threads") threads = std::atoi(val().c_str());
        else if (a == "--support") { const std::string v = val(); const size_t k = v.find(','); sup_lo = std::atoi(v.c_str()); sup_hi = k == std::string::npos ? sup_lo : std::atoi(v.c_str() + k + 1); }
        else if (a == "--genes") genes_override = std::atoi(val().c_str());
        else if (a == "--chim-copy-frac") chim_copy_frac = std::atof(val().c_str());
        else if (a == "--indel-frac") indel_frac = std::atof(val().c_str());
        else if (a == "--bwa") g_bwa = true;
        else if (a == "--interleave") n_interleave = std::atoi(val().c_str());
        else { std::fprintf(stderr, "unknown arg %s\n", a.c_str()); return 2; }
    }
    std::vector<Contig> contigs;
    int cfgno = 1;
    if (config == "C1") { contigs = {{"chr1", 10000000}}; if (records < 0) records = 20000; if (ntsv < 0) ntsv = 5; cfgno = 1; }
    else if (config == "C2") { contigs = {{"chr17", 83257441}}; if (records < 0) records = 1000000; if (ntsv < 0) ntsv = 20; cfgno = 2; }
    else if (config == "C3" || config == "C4" || config == "C5" || config == "C5g") {
        // C5 = BASELINE.json configs[4]: >= 1e5 small components (every junction sits in the LAST exon of both of its genes, so no
        // concordant read crosses from a junction's segments into the next gene: components stay below 20 nodes even with -w 1 -a 50);
        // C5g = the round-2 shape of that config (junctions at inner exons: the segments chain into one giant component)
        const bool giant = config == "C5g";
        if (giant) config = "C5";
        small_cc = config == "C5" && !giant;
        // (10-60 chimeric fragments per junction, as in the other configs, make the chimeric BAM 9 % of the records; `--support 3,12`
        // gives the ~2 % that STAR's Chimeric.out has on real RNA-seq, but then weight-1 noise edges are as strong as the junctions,
        // -w 1 -a 50 keeps them and ~1 % of the components grow beyond 20 nodes: not the shape BASELINE.json names)
        contigs.assign(kHg38, kHg38 + 25);
        cfgno = config[1] - '0';
        if (records < 0) records = config == "C3" ? 50000000 : (config == "C4" ? 200000000 : 100000000);
        if (ntsv < 0) ntsv = config == "C5" ? (giant ? 100000 : 110000) : 200;
    } else if (config == "T2") {  // two small contigs: inter-chromosomal test case
        contigs = {{"chrA", 3000000}, {"chrB", 2000000}, {"chrC", 50000}};
        if (records < 0) records = 30000; if (ntsv < 0) ntsv = 8; cfgno = 7;
    } else { std::fprintf(stderr, "unknown config\n"); return 2; }
    if (!seed) seed = 20180000ull + cfgno;
    Rng rng(seed);
    const int RL = 100;

    // ---- genes: roughly one per 'records/1500', at least 3x the junction count
    long total_len = 0;
    for (auto& c : contigs) total_len += c.len;
    int ngenes = genes_override > 0 ? genes_override : (int)std::max<long>(std::max(12, 3 * ntsv), records / 1500);
    std::vector<Gene> genes;
    for (size_t c = 0; c < contigs.size(); ++c) {
        int n = (int)((double)ngenes * contigs[c].len / total_len + 0.5);
        const int span = small_cc ? 12000 : 45000;  // max island extent incl. margin
        int maxn = contigs[c].len / span - 1;
        if (n > maxn) n = std::max(0, maxn);
        if (contigs[c].len < 2 * span) n = 0;
        // evenly strided slots with jitter -> sorted, non-overlapping islands
        for (int k = 0; k < n; ++k) {
            long slot = (long)(contigs[c].len - span) * k / std::max(1, n);
            long slotw = (long)(contigs[c].len - span) / std::max(1, n);
            int start = (int)(slot + 1000 + (slotw > span ? rng.next() % (uint64_t)(slotw - span + 1) : 0));
            Gene g;
            g.chr = (int)c;
            int nex = small_cc ? rng.range(3, 5) : rng.range(3, 8), p = start;
            g.cum.push_back(0);
            for (int e = 0; e < nex; ++e) {
                int el = small_cc && e == nex - 1 ? rng.range(450, 600) : rng.range(100, 300);
                g.es.push_back(p);
                g.ee.push_back(p + el);
                g.cum.push_back(g.cum.back() + el);
                p += el + (small_cc ? rng.range(500, 2000) : rng.range(500, 5000));
            }
            g.weight = std::exp(rng.normal());
            genes.push_back(g);
        }
    }
    if (genes.size() < 4) { std::fprintf(stderr, "too few genes\n"); return 2; }

    // ---- planted junctions: distinct genes per side, all four head/tail patterns, intra+inter chr
    std::vector<Tsv> tsvs;
    {
        std::vector<int> order(genes.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        for (size_t i = order.size() - 1; i > 0; --i) std::swap(order[i], order[rng.next() % (i + 1)]);
        size_t cur = 0;
        for (int t = 0; t < ntsv && cur + 1 < order.size(); ++t) {
            Tsv v;
            v.gx = order[cur++];
            v.gy = order[cur++];
            int pat = t % 4;
            v.xhead = (pat == 2 || pat == 3);
            v.yhead = (pat == 0 || pat == 2);
            const Gene& gx = genes[v.gx];
            const Gene& gy = genes[v.gy];
            // tail side: breakpoint at an exon end leaving >=1 exon upstream; head: exon start
            int nx = (int)gx.es.size(), ny = (int)gy.es.size();
            if (!v.xhead) { v.ex = small_cc ? nx - 1 : rng.range(1, nx - 2); v.bpx = gx.ee[v.ex]; v.txx = gx.cum[v.ex + 1]; }
            else { v.ex = small_cc ? nx - 1 : rng.range(1, nx - 2); v.bpx = gx.es[v.ex]; v.txx = gx.cum[v.ex]; }
            if (v.yhead) { v.ey = small_cc ? ny - 1 : rng.range(1, ny - 2); v.bpy = gy.es[v.ey]; v.txy = gy.cum[v.ey]; }
            else { v.ey = small_cc ? ny - 1 : rng.range(1, ny - 2); v.bpy = gy.ee[v.ey]; v.txy = gy.cum[v.ey + 1]; }
            int sup = rng.range(sup_lo, sup_hi);
            v.nsplit = sup / 2;
            v.npair = sup - v.nsplit;
            // keep the two partner genes at comparable depth so the coverage-ratio filter passes
            genes[v.gy].weight = genes[v.gx].weight * (0.6 + 0.8 * rng.uni());
            tsvs.push_back(v);
        }
        // --interleave K (off by default: no draw from the random stream, C1..C5 unchanged): K pairs of junctions between the SAME two
        // exons -- exon start with exon start (head-head) and exon end with exon end (tail-tail).  Both breakpoints of an exon become
        // segment boundaries, so the two junctions are two edges between one pair of segments; FilterbyInterleaving puts them into one
        // group whose Ind1 side is reached from the heads AND from the tails of the Ind2 side and vice versa ("Ind1 is in the middle of
        // Ind2, Ind2 is in the middle of Ind1", SegmentGraph.cpp:2260-2273) and drops the whole group: KeepEdge = false.
        for (int t = 0; t < n_interleave && cur + 1 < order.size(); ++t) {
            const int gxi = order[cur++], gyi = order[cur++];
            const Gene& gx = genes[gxi];
            const Gene& gy = genes[gyi];
            const int nx = (int)gx.es.size(), ny = (int)gy.es.size();
            const int ex = small_cc ? nx - 1 : rng.range(1, nx - 2), ey = small_cc ? ny - 1 : rng.range(1, ny - 2);
            genes[gyi].weight = genes[gxi].weight * (0.6 + 0.8 * rng.uni());
            for (int side = 0; side < 2; ++side) {
                Tsv v;
                v.gx = gxi; v.gy = gyi; v.ex = ex; v.ey = ey;
                v.xhead = v.yhead = side == 0;
                v.bpx = side == 0 ? gx.es[ex] : gx.ee[ex]; v.txx = side == 0 ? gx.cum[ex] : gx.cum[ex + 1];
                v.bpy = side == 0 ? gy.es[ey] : gy.ee[ey]; v.txy = side == 0 ? gy.cum[ey] : gy.cum[ey + 1];
                const int sup = rng.range(sup_lo, sup_hi);
                v.nsplit = sup / 2;
                v.npair = sup - v.nsplit;
                tsvs.push_back(v);
            }
        }
    }

    // ---- per-gene concordant fragment counts
    double wsum = 0;
    for (auto& g : genes) wsum += g.weight * g.tlen();
    long nfrag_total = records / 2;

    // ---- chimeric fragments (built first so that selected copies can be merged into the sorted stream)
    std::vector<Rec> chim;
    std::vector<std::vector<Rec>> gene_extra(genes.size());
    long chim_id = 0;
    auto part_piece = [&](const Gene& g, bool head, int tx, int k0, int k1) {
        // bases at junction distance [k0,k1) on a part anchored at transcript coordinate tx
        // tail-type part: transcript [tx-k1, tx-k0); head-type: [tx+k0, tx+k1)
        int t0 = head ? tx + k0 : tx - k1, t1 = head ? tx + k1 : tx - k0;
        t0 = std::max(t0, 0);
        t1 = std::min(t1, g.tlen());
        return std::pair<int, int>(t0, t1);
    };
    for (auto& v : tsvs) {
        const Gene& gx = genes[v.gx];
        const Gene& gy = genes[v.gy];
        // F = [X part reversed-distance ... junction ... Y part]; X holds F offsets < 0.
        int availx = v.xhead ? gx.tlen() - v.txx : v.txx;
        int availy = v.yhead ? gy.tlen() - v.txy : v.txy;
        for (int f = 0; f < v.nsplit + v.npair; ++f) {
            bool split = f < v.nsplit;
            int L = (int)std::lround(300 + 30 * rng.normal());
            L = std::max(L, 2 * RL + 10);
            // fragment covers F offsets [s, s+L) relative to the junction at 0
            int s;
            if (split) {
                int a = rng.range(25, 75);  // bases of the split mate on its first side
                bool m1split = rng.next() & 1;
                s = m1split ? -a : -(L - RL) - a;  // the junction falls inside mate1 or inside mate2
            } else {
                s = -rng.range(RL + 5, L - RL - 5);  // junction strictly between the mates
            }
            if (-s > availx) s = -availx;
            if (s + L > availy) L = availy - s;
            if (L < 2 * RL) continue;
            if (!split && (-s < RL || s + L < RL)) continue;
            std::string name = "chim" + std::to_string(chim_id++);
            bool swapmates = rng.next() & 1;  // which physical mate is flagged first-in-pair
            // each mate: F interval [a,b), read direction fwd (mate A) or revcomp (mate B)
            struct MateSpec { int a, b; bool rc; bool first; };
            MateSpec mates[2] = {{s, s + RL, false, !swapmates}, {s + L - RL, s + L, true, swapmates}};
            std::vector<Rec> recs;
            std::vector<int> rec_mate, rec_primary;  // (--bwa) which mate a record belongs to, and whether it is that mate's first piece
            for (auto& m : mates) {
                // split the mate's F interval at the junction
                struct Sub { bool isx; int k0, k1; };
                std::vector<Sub> subs;
                if (m.a < 0) subs.push_back({true, std::max(0, -m.b), -m.a});  // X: distances
                if (m.b > 0) subs.push_back({false, std::max(0, m.a), m.b});
                int npieces = (int)subs.size();
                for (int si = 0; si < npieces; ++si) {
                    const Sub& sb = subs[si];
                    const Gene& g = sb.isx ? gx : gy;
                    bool head = sb.isx ? v.xhead : v.yhead;
                    int tx = sb.isx ? v.txx : v.txy;
                    auto iv = part_piece(g, head, tx, sb.k0, sb.k1);
                    if (iv.second - iv.first < 20) continue;
                    Piece p = map_interval(g, iv.first, iv.second);
                    // direction of F along the genome inside this part: X part runs toward the junction,
                    // Y part away from it.  tail-type X / head-type Y run genome-forward.
                    bool fwd_on_genome = sb.isx ? !v.xhead : v.yhead;
                    bool reverse = (fwd_on_genome == m.rc);
                    // read bases outside this piece, in F direction; the CIGAR is written in genome order
                    int before = sb.isx ? 0 : std::max(-m.a, 0);
                    int after = sb.isx ? std::max(m.b, 0) : 0;
                    int lead = fwd_on_genome ? before : after, trail = fwd_on_genome ? after : before;
                    std::vector<CigarOp> cg;
                    bool hard = (npieces == 2 && si == 1);    // second piece of a split read is hard clipped
                    if (lead > 0) cg.push_back({hard ? 'H' : 'S', lead});
                    cg.insert(cg.end(), p.cig.begin(), p.cig.end());
                    if (trail > 0) cg.push_back({hard ? 'H' : 'S', trail});
                    int flag = 0x1 | (reverse ? 0x10 : 0) | (m.first ? 0x40 : 0x80);
                    if (npieces == 2 && si == 1) flag |= 0x100;
                    recs.push_back(make_record(rng, name, g.chr, p.refpos, 255, flag, cg, -1, -1, 0, 1, false));
                    rec_mate.push_back((int)(&m - mates));
                    rec_primary.push_back(1);
                    for (size_t q = 0; q + 1 < recs.size(); ++q) if (rec_mate[q] == rec_mate.back()) rec_primary.back() = 0;
                }
            }
            if (recs.size() < 2) continue;
            if (g_bwa) {
                // mate fields from the other mate's primary piece, supplementary flag on the later pieces, then into the sorted
                // stream of the gene island the record lies in
                for (size_t q = 0; q < recs.size(); ++q) {
                    int other = -1;
                    for (size_t o = 0; o < recs.size(); ++o) if (rec_mate[o] != rec_mate[q] && rec_primary[o]) other = (int)o;
                    std::vector<uint8_t>& b = recs[q].bytes;
                    uint16_t fl;
                    std::memcpy(&fl, b.data() + 18, 2);
                    fl &= (uint16_t)~0x100;
                    if (!rec_primary[q]) fl |= 0x800;
                    int32_t mr = -1, mp = -1;
                    if (other >= 0) {
                        uint16_t ofl;
                        std::memcpy(&ofl, recs[(size_t)other].bytes.data() + 18, 2);
                        if (ofl & 0x10) fl |= 0x20;
                        mr = recs[(size_t)other].refid; mp = recs[(size_t)other].pos;
                    } else fl |= 0x8;
                    std::memcpy(b.data() + 18, &fl, 2);
                    std::memcpy(b.data() + 24, &mr, 4);
                    std::memcpy(b.data() + 28, &mp, 4);
                    int gi = -1;
                    for (int cand : {v.gx, v.gy})
                        if (genes[cand].chr == recs[q].refid && recs[q].pos >= genes[cand].es.front() - 10 && recs[q].pos < genes[cand].ee.back()) gi = cand;
                    if (gi >= 0) gene_extra[gi].push_back(recs[q]);
                }
                continue;
            }
            // a copy of the un-split mate sometimes also sits in the concordant BAM (filtered there by name)
            if (rng.uni() < chim_copy_frac) {
                const Rec& r0 = recs.back();
                int gi = -1;
                for (int cand : {v.gx, v.gy})
                    if (genes[cand].chr == r0.refid && r0.pos >= genes[cand].es.front() - 10 && r0.pos < genes[cand].ee.back()) gi = cand;
                if (gi >= 0) gene_extra[gi].push_back(r0);
            }
            for (auto& r : recs) chim.push_back(std::move(r));
        }
    }
    // a few non-chimeric "partial" fragments in the chimeric file (clip positions -> PartAlignPos path)
    for (size_t gi = 0; gi < genes.size() && gi < 40; gi += 3) {
        const Gene& g = genes[gi];
        if (g.tlen() < 400) continue;
        int t0 = rng.range(0, g.tlen() - 320);
        std::string name = "part" + std::to_string(gi);
        Piece p1 = map_interval(g, t0 + 20, t0 + RL);
        std::vector<CigarOp> c1 = {{'S', 20}};
        c1.insert(c1.end(), p1.cig.begin(), p1.cig.end());
        Piece p2 = map_interval(g, t0 + 200, t0 + 300);
        Rec ra = make_record(rng, name, g.chr, p1.refpos, 255, 0x1 | 0x2 | 0x20 | 0x40, c1, g.chr, p2.refpos, 300, 1, false);
        Rec rb = make_record(rng, name, g.chr, p2.refpos, 255, 0x1 | 0x2 | 0x10 | 0x80, p2.cig, g.chr, p1.refpos, -300, 1, false);
        if (g_bwa) { gene_extra[gi].push_back(ra); gene_extra[gi].push_back(rb); continue; }
        chim.push_back(ra);
        chim.push_back(rb);
    }

    // ---- header
    std::string text = "@HD\tVN:1.4\tSO:coordinate\n";
    for (auto& c : contigs) text += "@SQ\tSN:" + c.name + "\tLN:" + std::to_string(c.len) + "\n";
    text += "@PG\tID:gen_synth_bam\tCL:config=" + config + " seed=" + std::to_string(seed) + "\n";
    auto write_header = [&](BgzfWriter& w, const std::string& t) {
        std::vector<uint8_t> h;
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put32(h, (int32_t)t.size());
        h.insert(h.end(), t.begin(), t.end());
        put32(h, (int32_t)contigs.size());
        for (auto& c : contigs) {
            put32(h, (int32_t)c.name.size() + 1);
            h.insert(h.end(), c.name.begin(), c.name.end());
            h.push_back(0);
            put32(h, c.len);
        }
        w.write(h.data(), h.size());
    };

    // ---- concordant stream, gene by gene (islands are disjoint and sorted => globally sorted)
    BgzfWriter bw(out + ".bam", level, threads);
    write_header(bw, text);
    long nrec = 0, nblocks = 0, frag_id = 0;
    std::vector<Rec> recs;
    // for the .bai: per reference the uncompressed stream offsets of its first record / behind its last, the record count, and
    // per 16 kb window the offset of the first record that overlaps it (the linear index)
    struct RefIdx { unsigned long long ubeg = 0, uend = 0; long n = 0; std::vector<unsigned long long> win; };
    std::vector<RefIdx> ridx(contigs.size());
    unsigned long long u_unplaced = 0;
    long n_unplaced = 0;
    for (size_t gi = 0; gi < genes.size(); ++gi) {
        const Gene& g = genes[gi];
        long nf = (long)std::llround((double)nfrag_total * g.weight * g.tlen() / wsum);
        recs.clear();
        int T = g.tlen();
        for (long f = 0; f < nf; ++f) {
            int L = (int)std::lround(300 + 30 * rng.normal());
            L = std::min(std::max(L, RL), T);
            int s = rng.range(0, T - L);
            bool m1left = rng.next() & 1;
            std::string name = "r" + std::to_string(frag_id++);
            uint64_t dice = rng.next();
            bool dup = (dice % 997) == 0;          // flagged PCR duplicate (dropped by every filter)
            bool multi = ((dice >> 10) % 211) == 0; // multi-mapper: NH 3, MAPQ 3
            bool lowq = ((dice >> 20) % 101) == 0;
            bool pcrcopy = ((dice >> 30) % 61) == 0; // identical un-flagged copy of the fragment
            int clipside = -1, cliplen = 0;
            if (((dice >> 40) % 20) == 0) { clipside = (int)((dice >> 50) & 3); cliplen = 16 + (int)((dice >> 52) % 15); }
            Piece pl = map_interval(g, s, s + RL);
            Piece pr = map_interval(g, s + L - RL, s + L);
            auto clip = [&](Piece& p, bool left, int n, const Gene& gg, int t0, int t1) {
                // replace n transcript bases at one end of the read with a soft clip
                Piece q = left ? map_interval(gg, t0 + n, t1) : map_interval(gg, t0, t1 - n);
                std::vector<CigarOp> c;
                if (left) c.push_back({'S', n});
                c.insert(c.end(), q.cig.begin(), q.cig.end());
                if (!left) c.push_back({'S', n});
                p.refpos = q.refpos; p.cig = c; p.reflen = q.reflen;
            };
            if (clipside == 0) clip(pl, true, cliplen, g, s, s + RL);
            if (clipside == 1) clip(pl, false, cliplen, g, s, s + RL);
            if (clipside == 2) clip(pr, true, cliplen, g, s + L - RL, s + L);
            if (clipside == 3) clip(pr, false, cliplen, g, s + L - RL, s + L);
            if (indel_frac > 0 && (double)((dice >> 8) & 0xffff) / 65536.0 < indel_frac) {
                // CIGAR variants of the same alignment: M a, I 2, M b / M a, D 3, M b / = a, X 1, = b on the first long match
                for (size_t ci = 0; ci < pl.cig.size(); ++ci) {
                    if (pl.cig[ci].op != 'M' || pl.cig[ci].len < 30) continue;
                    const int len = pl.cig[ci].len, a = 10 + (int)((dice >> 24) % 8), kind = (int)((dice >> 32) % 3);
                    std::vector<CigarOp> rep;
                    if (kind == 0) { rep = {{'M', a}, {'I', 2}, {'M', len - a - 2}}; pl.reflen -= 2; }
                    else if (kind == 1) { rep = {{'M', a}, {'D', 3}, {'M', len - a}}; pl.reflen += 3; }
                    else rep = {{'=', a}, {'X', 1}, {'=', len - a - 1}};
                    pl.cig.erase(pl.cig.begin() + ci);
                    pl.cig.insert(pl.cig.begin() + ci, rep.begin(), rep.end());
                    break;
                }
            }
            int fl = 0x1 | 0x2 | 0x20 | (m1left ? 0x40 : 0x80) | (dup ? 0x400 : 0);
            int fr = 0x1 | 0x2 | 0x10 | (m1left ? 0x80 : 0x40) | (dup ? 0x400 : 0);
            int mq = multi ? 3 : 255, nh = multi ? 3 : 1;
            int tl = pr.refpos + pr.reflen - pl.refpos;
            int copies = pcrcopy ? 2 : 1;
            for (int c = 0; c < copies; ++c) {
                std::string nm = c ? name + "d" : name;
                recs.push_back(make_record(rng, nm, g.chr, pl.refpos, mq, fl, pl.cig, g.chr, pr.refpos, tl, nh, lowq));
                recs.push_back(make_record(rng, nm, g.chr, pr.refpos, mq, fr, pr.cig, g.chr, pl.refpos, -tl, nh, false));
            }
        }
        for (auto& r : gene_extra[gi]) recs.push_back(r);
        std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.pos < b.pos; });
        for (auto& r : recs) {
            const unsigned long long u0 = bw.tell();
            bw.write(r.bytes.data(), r.bytes.size());
            ++nrec;
            if (r.refid >= 0 && r.refid < (int)ridx.size()) {
                RefIdx& x = ridx[(size_t)r.refid];
                if (!x.n) x.ubeg = u0;
                x.uend = bw.tell();
                ++x.n;
                // reference span of the record from its cigar (ops M, D, N, =, X)
                uint16_t ncig;
                std::memcpy(&ncig, r.bytes.data() + 16, 2);
                const uint8_t* cg = r.bytes.data() + 36 + r.bytes[12];
                int reflen = 0;
                for (int k = 0; k < ncig; ++k) { uint32_t v; std::memcpy(&v, cg + 4 * k, 4); const int op = v & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += (int)(v >> 4); }
                const size_t w0 = (size_t)(r.pos >> 14), w1 = (size_t)((r.pos + std::max(reflen, 1) - 1) >> 14);
                if (x.win.size() <= w1) x.win.resize(w1 + 1, ~0ull);
                for (size_t w = w0; w <= w1; ++w) if (x.win[w] == ~0ull) x.win[w] = u0;
            }
        }
    }
    // unmapped tail (refid -1): ignored by every pass, but must parse
    for (int i = 0; i < 3; ++i) {
        Rec r = make_record(rng, "unm" + std::to_string(i), -1, -1, 0, 0x1 | 0x4 | 0x8 | 0x40, {}, -1, -1, 0, 0, false);
        // give the unmapped record 100 bases so that seq/qual parsing is exercised
        if (!n_unplaced) u_unplaced = bw.tell();
        ++n_unplaced;
        bw.write(r.bytes.data(), r.bytes.size());
        ++nrec;
    }
    bw.close();
    // ---- <out>.bam.bai (SAM spec 5.2).  Coarse but valid: per reference one real bin (bin 0, one chunk over all its records),
    // the metadata pseudo-bin 37450 (first / end virtual offset, mapped / unmapped counts) and the 16 kb linear index.
    {
        const std::vector<unsigned long long>& bc = bw.block_offsets();
        auto voff = [&](unsigned long long u) -> unsigned long long {
            const size_t blk = (size_t)(u / BgzfWriter::block_size());
            // (an offset exactly at the end of the data lies at the start of the EOF marker block = end of the last data block)
            if (blk >= bc.size()) return bc.empty() ? 0 : ((bc.back() << 16) | (unsigned long long)(u - (bc.size() - 1) * BgzfWriter::block_size()));
            return (bc[blk] << 16) | (unsigned long long)(u % BgzfWriter::block_size());
        };
        FILE* bf = std::fopen((out + ".bam.bai").c_str(), "wb");
        auto w32 = [&](uint32_t v) { std::fwrite(&v, 4, 1, bf); };
        auto w64 = [&](unsigned long long v) { std::fwrite(&v, 8, 1, bf); };
        std::fwrite("BAI\1", 1, 4, bf);
        w32((uint32_t)ridx.size());
        for (const RefIdx& x : ridx) {
            if (!x.n) { w32(0); w32(0); continue; }
            w32(2);
            w32(0); w32(1); w64(voff(x.ubeg)); w64(voff(x.uend));
            w32(37450); w32(2); w64(voff(x.ubeg)); w64(voff(x.uend)); w64((unsigned long long)x.n); w64(0);
            w32((uint32_t)x.win.size());
            unsigned long long lastv = voff(x.ubeg);
            for (unsigned long long u : x.win) { if (u != ~0ull) lastv = voff(u); w64(lastv); }
        }
        w64((unsigned long long)n_unplaced);
        std::fclose(bf);
        (void)u_unplaced;
    }

    // ---- chimeric BAM (unsorted; shuffle so that name order != file order)
    if (!g_bwa) {
        // keep the first five records full-length (they fix ReadLen, ReadRec.cpp:347-348,378-379)
        Rng srng(seed ^ 0xC0FFEEull);
        for (size_t i = chim.size(); i > 6; --i) {
            size_t j = 5 + srng.next() % (i - 5);
            std::swap(chim[i - 1], chim[j]);
        }
        std::string ctext = "@HD\tVN:1.4\n";
        for (auto& c : contigs) ctext += "@SQ\tSN:" + c.name + "\tLN:" + std::to_string(c.len) + "\n";
        BgzfWriter cw(out + ".chim.bam", level, threads);
        write_header(cw, ctext);
        for (auto& r : chim) cw.write(r.bytes.data(), r.bytes.size());
        cw.close();
    }
    // ---- truth
    {
        FILE* tf = std::fopen((out + ".truth.txt").c_str(), "w");
        std::fprintf(tf, "# chrom1\tbp1\tend1\tchrom2\tbp2\tend2\tsplit\tpairs\n");
        for (auto& v : tsvs)
            std::fprintf(tf, "%s\t%d\t%c\t%s\t%d\t%c\t%d\t%d\n", contigs[genes[v.gx].chr].name.c_str(), v.bpx, v.xhead ? 'H' : 'T',
                         contigs[genes[v.gy].chr].name.c_str(), v.bpy, v.yhead ? 'H' : 'T', v.nsplit, v.npair);
        std::fclose(tf);
    }
    (void)nblocks;
    std::printf("{\"config\":\"%s\",\"seed\":%llu,\"concordant_records\":%ld,\"chimeric_records\":%zu,\"genes\":%zu,\"tsv\":%zu}\n",
                config.c_str(), (unsigned long long)seed, nrec, chim.size(), genes.size(), tsvs.size());
    return 0;
}
