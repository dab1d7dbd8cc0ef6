// Junction sequences of the called SVs (second half): the counterpart of utils/JunctionSequence.cpp --
// `_sv.txt` + the chimeric BAM + the genome FASTA -> <prefix>_junc_precise.fa / _junc_relax.fa / _junc_alt.fa.  A consumer of the
// hot path's output, host code throughout (no device work: the chimeric fragments are merged by the same host code the `squid`
// ingest uses, BuildChimericSBamRecord = build_fragments).  Reference, by line of utils/JunctionSequence.cpp: read the calls :89-110;
// what every chimeric fragment says about a junction :112-168; the call it supports :170-221; narrowing a call to the bases split
// reads cover, support counts and alternative junction points :223-396; genome :398-420; the three writers :422-517.
//
// Reproduced as the reference has them: the second end's left-hand extension restarts from the first read breakpoint where the first
// end's continues from the hit (:318-321 vs :288-291); a left first end's "differs" test looks at the read's SECOND end (:358); a name
// missing from the header table means reference 0 (std::map::operator[]); a base the complement table does not know turns into NUL.
#include <algorithm>
#include <climits>
#include <cstring>
#include <fstream>
#include <map>

#include "sq_internal.h"

namespace sq {

namespace {

struct End {  // one side of a junction: a stretch of a chromosome and which of its ends the junction sits at
    int chr = 0, lo = 0, hi = 0;
    bool left = false;
    bool operator<(const End& r) const { return chr != r.chr ? chr < r.chr : lo != r.lo ? lo < r.lo : hi != r.hi ? hi < r.hi : left < r.left; }
    bool operator==(const End& r) const { return chr == r.chr && lo == r.lo && hi == r.hi && left == r.left; }
    int anchor() const { return left ? lo : hi; }  // the coordinate the junction itself is at
};
struct Junction {
    End a, b;  // a <= b
    Junction() {}
    Junction(const End& x, const End& y) { if (x < y) { a = x; b = y; } else { a = y; b = x; } }
    bool operator<(const Junction& r) const { return a == r.a ? b < r.b : a < r.a; }
    bool operator==(const Junction& r) const { return a == r.a && b == r.b; }
};
End end_of(const Blk& b, bool left) { return End{b.refid, b.refpos, b.refpos + b.matchref, left}; }

// what one chimeric fragment says (:112-168): every discordant step between consecutive blocks of a mate; if there is none, the two
// mates' last blocks when the pair is discordant and some read end hangs over by more than 12 good bases
void junctions_of(const Frag& f, std::vector<Junction>& out) {
    out.clear();
    for (const BlkList* m : {&f.a, &f.b})
        for (size_t i = 0; i + 1 < m->size(); ++i) {
            const Blk &x = (*m)[i], &y = (*m)[i + 1];
            const bool order_ref = x.refpos < y.refpos, order_read = x.readpos < y.readpos;
            if (x.refid != y.refid || x.rev != y.rev || (x.rev ? order_ref == order_read : order_ref != order_read)) out.push_back(Junction(end_of(x, x.rev), end_of(y, !y.rev)));
        }
    if (!out.empty() || f.a.empty() || f.b.empty() || !frag_pair_discordant(f, false)) return;
    const bool hang = (!f.alow && (f.a.front().readpos > 12 || f.atot - f.a.back().readpos - f.a.back().matchread > 12)) ||
                      (!f.blow && (f.b.front().readpos > 12 || f.btot - f.b.back().readpos - f.b.back().matchread > 12));
    if (hang) out.push_back(Junction(end_of(f.a.back(), f.a.back().rev), end_of(f.b.back(), f.b.back().rev)));
}

// the call a read junction belongs to (:170-200): same chromosomes and sides, the read's anchor within [-5, +300] of the call's on
// the inner side, smallest summed deviation, first among equals
int nearest_call(const Junction& r, const std::vector<Junction>& calls) {
    int best = -1, best_dev = INT_MAX;
    auto near = [](const End& x, const End& c) {
        if (x.chr != c.chr || x.left != c.left) return false;
        return x.left ? x.lo >= c.lo - 5 && x.lo <= c.lo + 300 : x.hi >= c.hi - 300 && x.hi <= c.hi + 5;
    };
    for (size_t i = 0; i < calls.size(); ++i) {
        if (!near(r.a, calls[i].a) || !near(r.b, calls[i].b)) continue;
        const int dev = std::abs(r.a.anchor() - calls[i].a.anchor()) + std::abs(r.b.anchor() - calls[i].b.anchor());
        if (dev < best_dev) { best_dev = dev; best = (int)i; }
    }
    return best;
}

// read breakpoints within 5 bases of a call's end, on the coordinate the CALL's side names (:262-275)
int hits_at(const End& e, const std::vector<End>& ends) {
    int hits = 0;
    for (const End& x : ends) hits += std::abs(e.anchor() - (e.left ? x.lo : x.hi)) < 5;
    return hits;
}
// narrow one end of a call to the bases its split reads cover (:283-338).  `ends`: that end of every supporting read junction.
// `restart`: the extension runs over the whole list instead of from the first hit on (the second end of the reference's code)
bool extend(End& e, std::vector<End> ends, bool restart) {
    const int thresh = 5;
    if (e.left) {
        std::sort(ends.begin(), ends.end());
        size_t k = 0;
        while (std::abs(ends[k].lo - e.lo) >= thresh) ++k;
        int right = ends[k].hi;
        for (size_t q = restart ? 0 : k; q < ends.size(); ++q) if (ends[q].lo < right) right = std::max(right, ends[q].hi);
        if (e.lo < right) { e.hi = std::min(right, e.hi); return true; }
    } else {
        std::sort(ends.begin(), ends.end(), [](const End& x, const End& y) { return x.chr != y.chr ? x.chr < y.chr : x.hi != y.hi ? x.hi < y.hi : x.lo != y.lo ? x.lo < y.lo : x.left < y.left; });
        size_t k = ends.size();
        while (std::abs(ends[k - 1].hi - e.hi) >= thresh) --k;
        int leftm = ends[k - 1].lo;
        for (size_t q = restart ? ends.size() : k; q > 0; --q) if (ends[q - 1].hi > leftm) leftm = std::min(leftm, ends[q - 1].lo);
        if (leftm < e.hi) { e.lo = std::max(leftm, e.lo); return true; }
    }
    return false;
}

char complement(char ch) {
    static const char from[] = "ACGTRYSWKMBVDHN.-", to[] = "TGCAYRWSMKVBHDN.-";
    const char* p = std::strchr(from, std::toupper((unsigned char)ch));
    return (p && *p) ? to[p - from] : '\0';
}

struct Genome {
    std::vector<std::string> seq;
    std::vector<std::string> name;
    bool write(std::ofstream& o, const std::string& id, const Junction& j, const std::string& tail) const {
        if ((int)seq[(size_t)j.a.chr].size() < j.a.hi || (int)seq[(size_t)j.b.chr].size() < j.b.hi) return false;
        auto piece = [&](const End& e, bool rc) {
            std::string s = seq[(size_t)e.chr].substr((size_t)e.lo, (size_t)(e.hi - e.lo));
            if (rc) { for (char& ch : s) ch = complement(ch); std::reverse(s.begin(), s.end()); }
            return s;
        };
        const std::string s = piece(j.a, j.a.left) + piece(j.b, !j.b.left);
        o << id << " " << name[(size_t)j.a.chr] << ":" << j.a.lo << ":" << j.a.hi << ":" << (j.a.left ? "-" : "+") << " " << name[(size_t)j.b.chr] << ":" << j.b.lo << ":" << j.b.hi << ":" << (j.b.left ? "+" : "-") << tail << std::endl;
        for (size_t at = 0; at < s.size(); at += 80) o << s.substr(at, std::min<size_t>(80, s.size() - at)) << std::endl;
        return true;
    }
};

}  // namespace

int junction_sequences(sq_ctx* c, const std::vector<std::string>& ref_names, const char* bedpe, const char* fasta, const char* out_prefix) {
    std::map<std::string, int> table;
    for (size_t i = 0; i < ref_names.size(); ++i) table[ref_names[i]] = (int)i;
    if (ref_names.empty()) return fail(c, SQ_E_ARG, "the chimeric BAM names no reference");
    // ---- the calls (:89-110): BEDPE rows, mitochondrion and unplaced contigs left out by the first letter of their names
    std::vector<Junction> calls;
    {
        std::ifstream in(bedpe);
        if (!in) return fail(c, SQ_E_IO, std::string("cannot open ") + bedpe);
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::vector<std::string> f(1);
            for (char ch : line) { if (ch == '\t') f.emplace_back(); else f.back().push_back(ch); }
            if (f.size() < 10) continue;
            auto skip = [](const std::string& n) { return !n.empty() && (n[0] == 'M' || n[0] == 'G' || n[0] == 'K'); };
            if (skip(f[0]) || skip(f[3])) continue;
            auto ref = [&](const std::string& n) { auto it = table.find(n); return it == table.end() ? 0 : it->second; };
            calls.push_back(Junction(End{ref(f[0]), std::atoi(f[1].c_str()), std::atoi(f[2].c_str()), f[8] == "-"}, End{ref(f[3]), std::atoi(f[4].c_str()), std::atoi(f[5].c_str()), f[9] == "-"}));
        }
    }
    // ---- read junctions per call (:202-221)
    std::vector<std::vector<Junction>> reads(calls.size());
    {
        std::vector<Junction> js;
        for (const Frag& f : c->frags0) { junctions_of(f, js); for (const Junction& j : js) { const int k = nearest_call(j, calls); if (k >= 0) reads[(size_t)k].push_back(j); } }
    }
    // ---- narrowing, support, alternative junction points (:223-396)
    std::vector<char> exact(calls.size(), 0);
    std::vector<int> support(calls.size(), 0);
    std::vector<std::vector<Junction>> alts(calls.size());
    for (size_t i = 0; i < calls.size(); ++i) {
        if (reads[i].empty()) continue;
        Junction& J = calls[i];
        std::vector<End> as, bs;
        for (const Junction& r : reads[i]) { as.push_back(r.a); bs.push_back(r.b); }
        const int hit_a = hits_at(J.a, as), hit_b = hits_at(J.b, bs);
        if (!hit_a || !hit_b) continue;
        const bool got_a = extend(J.a, as, false), got_b = extend(J.b, bs, true);
        if (got_a && got_b) { exact[i] = 1; support[i] = std::min(hit_a, hit_b); }
        std::vector<Junction> cand;
        for (const Junction& r : reads[i]) {
            Junction alt = J;
            bool has_a = false, has_b = false, dif_a = false, dif_b = false;
            if (J.a.left == r.a.left && std::abs(J.a.anchor() - r.a.anchor()) < 5) {
                has_a = true;
                if (J.a.left) { alt.a.lo = r.a.lo; dif_a = J.a.lo != r.b.lo; }  // (:358: the read's second end)
                else { alt.a.hi = r.a.hi; dif_a = J.a.hi != r.a.hi; }
            }
            if (J.b.left == r.b.left && std::abs(J.b.anchor() - r.b.anchor()) < 5) {
                has_b = true;
                if (J.b.left) { alt.b.lo = r.b.lo; dif_b = J.b.lo != r.b.lo; }
                else { alt.b.hi = r.b.hi; dif_b = J.b.hi != r.b.hi; }
            }
            if (has_a && has_b && (dif_a || dif_b)) cand.push_back(alt);  // (as it is: the reference does not order the two ends again)
        }
        if (!cand.empty() && !exact[i]) return fail(c, SQ_E_ASSERT, "an alternative junction point without an exact junction (the reference asserts, utils/JunctionSequence.cpp:384)");
        std::sort(cand.begin(), cand.end());
        cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
        alts[i].swap(cand);
    }
    // ---- genome (:398-420): a record whose name the table does not know overwrites reference 0, as std::map::operator[] has it
    Genome G;
    G.name = ref_names;
    G.seq.assign(ref_names.size(), std::string());
    {
        std::ifstream in(fasta);
        if (!in) return fail(c, SQ_E_IO, std::string("cannot open ") + fasta);
        std::string line, name, acc;
        bool open = false;
        auto close = [&]() { if (open && !(name.empty())) { auto it = table.find(name); G.seq[(size_t)(it == table.end() ? 0 : it->second)] = acc; } };
        auto close_last = [&]() { auto it = table.find(name); G.seq[(size_t)(it == table.end() ? 0 : it->second)] = acc; };  // (the last record is stored even without a name, :417)
        while (std::getline(in, line)) {
            if (!line.empty() && line[0] == '>') {
                close();
                size_t e = 1;
                while (e < line.size() && line[e] != ' ' && line[e] != '\t') ++e;
                name = line.substr(1, e - 1);
                acc.clear();
                open = true;
            } else acc += line;
        }
        close_last();
    }
    const std::string pre = out_prefix;
    const int beyond = SQ_E_ASSERT;
    {
        std::ofstream o(pre + "_junc_precise.fa");
        for (size_t i = 0; i < calls.size(); ++i)
            if (exact[i] && !G.write(o, ">squid_" + std::to_string(i), calls[i], " " + std::to_string(support[i]))) return fail(c, beyond, "a junction reaches beyond its chromosome's sequence (the reference asserts, utils/JunctionSequence.cpp:431)");
    }
    {
        std::ofstream o(pre + "_junc_relax.fa");
        for (size_t i = 0; i < calls.size(); ++i) {
            Junction j = calls[i];
            if ((int)G.seq[(size_t)j.a.chr].size() < j.a.hi || (int)G.seq[(size_t)j.b.chr].size() < j.b.hi) return fail(c, beyond, "a junction reaches beyond its chromosome's sequence (the reference asserts, utils/JunctionSequence.cpp:459)");
            if (exact[i])
                for (End* e : {&j.a, &j.b}) { if (e->left) e->hi = std::min(e->hi + 1000, (int)G.seq[(size_t)e->chr].size()); else e->lo = std::max(0, e->lo - 1000); }
            if (!G.write(o, ">squid_" + std::to_string(i), j, "")) return fail(c, beyond, "a junction reaches beyond its chromosome's sequence");
        }
    }
    {
        std::ofstream o(pre + "_junc_alt.fa");
        for (size_t i = 0; i < alts.size(); ++i)
            for (size_t k = 0; k < alts[i].size(); ++k)
                if (!G.write(o, ">squid_" + std::to_string(i) + "_alt_" + std::to_string(k + 1), alts[i][k], " " + std::to_string(support[i]))) return fail(c, beyond, "a junction reaches beyond its chromosome's sequence (the reference asserts, utils/JunctionSequence.cpp:492)");
    }
    return SQ_OK;
}

}  // namespace sq
