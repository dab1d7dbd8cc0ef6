// squid_annotate -- labels the rows of a SQUID `_sv.txt` as fusion-gene / non-fusion-gene from a GTF annotation.
// Counterpart of the reference's utils/AnnotateSQUIDOutput.py: same command line, same output
// columns, same matching rules (line numbers below are that script's).  It is a join of a few hundred breakpoints against the
// gene ranges of one GTF: host work, no GPU stage.
//
//   squid_annotate [--geneid <attr>] [--genesymbol <attr>] <GTFfile> <SquidPrediction> <OutputFile>
//
// One documented difference: the script collects the genes at a breakpoint through list(set(...)) (:239), whose order follows
// Python's per-process string hashing, so the ORDER of the pairs in the FusedGenes column is not reproducible there; here the
// genes come in the order the two walks of LocatePosition_generange meet them.  The set of pairs is the same
// (tests/test_annotate.py compares that column as a multiset, every other byte exactly, against outputs of the real script).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Transcript {
    std::string id, gene, name, chr;
    bool strand = true;  // "+"
    long start = 0, end = 0;
    std::vector<std::pair<long, long>> exons;
};

std::string strip(const std::string& s) {  // str.strip(): ASCII whitespace at both ends
    size_t a = 0, b = s.size();
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
    while (a < b && ws(s[a])) ++a;
    while (b > a && ws(s[b - 1])) --b;
    return s.substr(a, b - a);
}
std::vector<std::string> split_tab(const std::string& s) {
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        size_t t = s.find('\t', a);
        if (t == std::string::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, t - a));
        a = t + 1;
    }
    return out;
}
[[noreturn]] void die(const std::string& msg) { std::cout << msg << std::endl; std::exit(0); }  // the script prints and sys.exit()s (status 0)

// GetFeature (:57-60): first occurrence of `key` anywhere in the line, value between the quote behind it and the next ';'
std::string get_feature(const std::string& line, const std::string& key) {
    const size_t s = line.find(key);
    if (s == std::string::npos) die("substring not found");
    const size_t t = line.find(';', s + 1);
    if (t == std::string::npos) die("substring not found");
    const size_t from = s + key.size() + 2;
    return from < t - 1 ? line.substr(from, t - 1 - from) : std::string();
}

// insertion-ordered map of transcripts (a Python dict keeps the position of a key's first insertion)
struct TranscriptTable {
    std::vector<Transcript> rows;
    std::unordered_map<std::string, size_t> at;
    void set(const Transcript& t) {
        auto it = at.find(t.id);
        if (it == at.end()) { at[t.id] = rows.size(); rows.push_back(t); } else rows[it->second] = t;
    }
    Transcript* find(const std::string& id) { auto it = at.find(id); return it == at.end() ? nullptr : &rows[it->second]; }
};

// ReadGTF (:63-127)
void read_gtf(const std::string& path, const std::string& key_gene, const std::string& key_sym, TranscriptTable& T) {
    std::ifstream fp(path);
    if (!fp) { std::cerr << "cannot open " << path << std::endl; std::exit(1); }
    struct Extra { std::string tid; long a, b; std::string gene, name, chr; bool strand; };
    std::vector<Extra> extra;
    std::string line, curname;
    Transcript cur;
    bool have = false;
    while (std::getline(fp, line)) {
        line += "\n";
        if (line[0] == '#') continue;
        const std::vector<std::string> f = split_tab(strip(line));
        if (f.size() < 7) continue;  // (the script would raise on such a line)
        if (f[2] == "transcript") {
            if (!curname.empty() && have) T.set(cur);
            if (line.find("transcript_id") == std::string::npos) die("GTF file attribute column doesn't contain transcript_id: " + line);
            if (line.find(key_gene) == std::string::npos) die("GTF file attribute column doesn't contain " + key_gene + ": " + line);
            if (line.find(key_sym) == std::string::npos) die("GTF file attribute column doesn't contain " + key_sym + ": " + line);
            curname = get_feature(line, "transcript_id");
            cur = Transcript();
            cur.id = curname; cur.gene = get_feature(line, key_gene); cur.name = get_feature(line, key_sym);
            cur.chr = f[0]; cur.strand = f[6] == "+"; cur.start = std::atol(f[3].c_str()) - 1; cur.end = std::atol(f[4].c_str());
            have = true;
        } else if (f[2] == "exon") {
            const std::string tid = get_feature(line, "transcript_id");
            if (tid == curname && have) cur.exons.push_back({std::atol(f[3].c_str()) - 1, std::atol(f[4].c_str())});
            else extra.push_back(Extra{tid, std::atol(f[3].c_str()) - 1, std::atol(f[4].c_str()), get_feature(line, key_gene), get_feature(line, key_sym), f[0], f[6] == "+"});
        }
    }
    if (!curname.empty() && have) T.set(cur);
    // exon rows that did not follow their transcript row (:104-118), by transcript id (stable)
    std::stable_sort(extra.begin(), extra.end(), [](const Extra& x, const Extra& y) { return x.tid < y.tid; });
    for (const Extra& e : extra) {
        if (Transcript* t = T.find(e.tid)) {
            t->exons.push_back({e.a, e.b});
            // (in the script the loop variable IS the stored object of its id: an exon appended through the table reaches it too, and
            // storing it again later -- T.set(cur) below -- must not lose that exon)
            if (have && cur.id == e.tid) cur.exons.push_back({e.a, e.b});
            continue;
        }
        // a transcript known only from exon rows: the script keeps it in its loop variable and stores it when the NEXT unknown id
        // arrives -- the last one is never stored (kept)
        if (have && cur.id != e.tid) {
            T.set(cur);
            cur = Transcript();
            cur.id = e.tid; cur.gene = e.gene; cur.name = e.name; cur.chr = e.chr; cur.strand = e.strand; cur.start = e.a; cur.end = e.b;
        } else if (!have) {
            cur = Transcript();
            cur.id = e.tid; cur.gene = e.gene; cur.name = e.name; cur.chr = e.chr; cur.strand = e.strand; cur.start = e.a; cur.end = e.b;
            have = true;
        }
        cur.exons.push_back({e.a, e.b});
        // (T.set(cur) above stored a COPY, as Python stores a reference: exons appended to `cur` afterwards must reach the stored
        // object too when it is the same transcript)
        if (Transcript* t = T.find(cur.id)) *t = cur;
    }
    for (Transcript& t : T.rows) {
        if (t.exons.empty()) { std::cerr << "transcript " << t.id << " has no exon (the reference script raises here)" << std::endl; std::exit(1); }
        std::stable_sort(t.exons.begin(), t.exons.end(), [](const std::pair<long, long>& x, const std::pair<long, long>& y) { return x.first < y.first; });
        t.start = t.exons[0].first; t.end = t.exons[0].second;
        for (const auto& e : t.exons) { t.start = std::min(t.start, e.first); t.end = std::max(t.end, e.second); }
        if (!t.strand) std::reverse(t.exons.begin(), t.exons.end());
    }
}

struct GeneLocater {  // (:165-254)
    struct Range { std::string chr; long lb, ub; };
    std::vector<Range> ranges;
    std::vector<std::string> names;
    static bool chr_lt(const std::string& a, const std::string& b) { return a < b; }
    void build(TranscriptTable& T, const std::vector<std::string>& gene_order, std::map<std::string, std::vector<std::string>>& gene_trans) {
        std::vector<Range> r;
        std::vector<std::string> nm;
        for (const std::string& g : gene_order) {
            const std::vector<std::string>& v = gene_trans[g];
            Range x{T.find(v[0])->chr, 0, 0};
            bool first = true;
            for (const std::string& t : v) {
                const Transcript* tr = T.find(t);
                if (tr->chr != x.chr) { std::cerr << "AssertionError: gene " << g << " lies on two chromosomes" << std::endl; std::exit(1); }  // assert at :183
                if (first) { x.lb = tr->start; x.ub = tr->end; first = false; } else { x.lb = std::min(x.lb, tr->start); x.ub = std::max(x.ub, tr->end); }
            }
            r.push_back(x); nm.push_back(g);
        }
        std::vector<size_t> idx(r.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t p, size_t q) {  // tuple order (chr, lb, ub)
            if (r[p].chr != r[q].chr) return r[p].chr < r[q].chr;
            if (r[p].lb != r[q].lb) return r[p].lb < r[q].lb;
            return r[p].ub < r[q].ub;
        });
        for (size_t i : idx) { ranges.push_back(r[i]); names.push_back(nm[i]); }
    }
    // LocatePosition_generange (:203-239), quirks included: `high = mid - 1`, the 20-step minimum of both walks
    std::vector<std::string> locate(const std::string& chr, long pos, long window = 100000, long fuzzy = 50) const {
        std::vector<std::string> genes;
        const long n = (long)names.size();
        long low = 0, high = n;
        while (low < high) {
            const long mid = (low + high) / 2;
            const Range& m = ranges[(size_t)mid];
            if (m.chr < chr || (m.chr == chr && m.ub < pos - fuzzy)) low = mid + 1;
            else if (m.chr == chr && m.lb <= pos + fuzzy && m.ub > pos - fuzzy) { low = mid; high = mid; }
            else high = mid - 1;
        }
        auto hit = [&](long k) { const Range& m = ranges[(size_t)k]; return m.chr == chr && m.lb <= pos + fuzzy && m.ub > pos - fuzzy; };
        long count_low = 0, count_high = 0;
        if (low >= 0 && low != n)
            while (low >= 0 && (count_low < 20 || (ranges[(size_t)low].chr == chr && ranges[(size_t)low].ub + window > pos))) {
                ++count_low;
                if (hit(low)) genes.push_back(names[(size_t)low]);
                --low;
            }
        if (high >= 0 && high != n)
            while (high < n && (count_high < 20 || (ranges[(size_t)high].chr == chr && ranges[(size_t)high].lb <= pos + fuzzy))) {
                ++count_high;
                if (hit(high)) genes.push_back(names[(size_t)high]);
                ++high;
            }
        std::vector<std::string> uniq;  // list(set(genes)): each gene once
        for (const std::string& g : genes) if (std::find(uniq.begin(), uniq.end(), g) == uniq.end()) uniq.push_back(g);
        return uniq;
    }
};

}  // namespace

int main(int argc, char* argv[]) {
    if (argc == 1) {
        std::cout << "squid_annotate [options] <GTFfile> <SquidPrediction> <OutputFile>\noptions:\n"
                     "\t--geneid\tstring\tGTF gene ID attribute string, the attribute name in GTF record that corresponds to the gene ID (default: gene_id)\n"
                     "\t--genesymbol\tstring\tGTF gene symbol attribute string, the attribute name in GTF record that corresponds to the gene symbol (default: gene_name)\n";
        return 0;
    }
    std::string key_gene = "gene_id", key_sym = "gene_name", gtf, sv, out;
    for (int i = 1; i < argc;) {  // ParseArgument (:301-333)
        const std::string a = argv[i];
        if (a == "--geneid") { if (i + 1 >= argc || std::string(argv[i + 1]).substr(0, 2) == "--") die("GTF gene ID attribute string is empty!"); key_gene = argv[i + 1]; i += 2; }
        else if (a == "--genesymbol") { if (i + 1 >= argc || std::string(argv[i + 1]).substr(0, 2) == "--") die("GTF gene symbol attribute string is empty!"); key_sym = argv[i + 1]; i += 2; }
        else if (a.substr(0, 2) == "--") die("Unknown argument " + a);
        else {
            if (i + 2 >= argc) die("Missing GTFfile or SquidPrediction or OutputFile");
            gtf = argv[i]; sv = argv[i + 1]; out = argv[i + 2];
            break;
        }
    }
    TranscriptTable T;
    read_gtf(gtf, key_gene, key_sym, T);
    // Map_Gene_Trans (:130-143): gene -> sorted transcript ids, genes in order of first appearance
    std::vector<std::string> gene_order;
    std::map<std::string, std::vector<std::string>> gene_trans;
    for (const Transcript& t : T.rows) { if (!gene_trans.count(t.gene)) gene_order.push_back(t.gene); gene_trans[t.gene].push_back(t.id); }
    for (auto& kv : gene_trans) std::sort(kv.second.begin(), kv.second.end());
    GeneLocater G;
    G.build(T, gene_order, gene_trans);
    // Annotate (:257-298)
    std::ifstream in(sv);
    if (!in) { std::cerr << "cannot open " << sv << std::endl; return 1; }
    std::ofstream o(out);
    if (!o) { std::cerr << "cannot write " << out << std::endl; return 1; }
    auto join10 = [](const std::vector<std::string>& f) { std::string s; for (size_t i = 0; i < f.size() && i < 10; ++i) { if (i) s += "\t"; s += f[i]; } return s; };
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        const std::vector<std::string> f = split_tab(strip(line));
        if (line[0] == '#') { o << join10(f) << "\tType\tFusedGenes\n"; continue; }
        if (f.size() < 10) continue;
        const bool s1 = f[8] == "+", s2 = f[9] == "+";
        const long bp1 = std::atol((s1 ? f[2] : f[1]).c_str()), bp2 = std::atol((s2 ? f[5] : f[4]).c_str());
        const std::vector<std::string> g1 = G.locate(f[0], bp1), g2 = G.locate(f[3], bp2);
        std::vector<std::string> pairs;
        for (const std::string& a : g1)
            for (const std::string& b : g2) {
                const Transcript &ta = *T.find(gene_trans[a][0]), &tb = *T.find(gene_trans[b][0]);
                // a fusion gene: one breakpoint agrees with its gene's strand, the other does not; the 5' gene comes first
                if ((ta.strand == s1) != (tb.strand == s2)) pairs.push_back(ta.strand == s1 ? ta.name + ":" + tb.name : tb.name + ":" + ta.name);
            }
        if (pairs.empty()) o << join10(f) << "\tnon-fusion-gene\t.\n";
        else { o << join10(f) << "\tfusion-gene\t"; for (size_t i = 0; i < pairs.size(); ++i) o << (i ? "," : "") << pairs[i]; o << "\n"; }
    }
    return 0;
}
