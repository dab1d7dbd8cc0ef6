// Genome segmentation (BuildNode_STAR of src/SegmentGraph.cpp:192-761).
//
// Split of work: the GPU (k_classify/k_dedup/k_summarise) has already reduced the concordant stream to one
// 20-byte summary per kept record -- the record key, its first aligned block and its classification -- plus the
// further blocks of concordant records.  What remains here is the order-dependent control automaton: it walks
// the kept stream once, keeps two sliding windows (fully aligned / partially aligned concordant first blocks) as
// index ranges into the summary array, and at every discordant cluster decides the segment boundaries.
// The reference's automaton carries ~15 variables; the same state lives in `Seg` below.
#include <algorithm>
#include <cmath>

#include "sq_internal.h"

namespace sq {

namespace {

struct El {  // a window element = first block of a concordant record
    int32_t refid, refpos, matchref, readpos;
    bool rev;
};

struct Seg {
    const sq_ctx* c;
    const SegmentInput& in;
    const int RL;
    static constexpr int T = 3;     // thresh (SegmentGraph.cpp:286)
    static constexpr int NEAR = 60;  // thresh*20
    std::vector<Blk> D;              // sorted discordant blocks + zero sentinel at [nd] (ledger B21)
    int nd = 0;
    std::vector<std::pair<int, int>> part;  // PartAlignPos
    std::vector<int32_t> cw, pw;     // windows: indices into in.recs (ConcordantCluster / PartialAlignCluster)
    int co = 0, po = 0;              // window offsets
    std::vector<std::pair<std::pair<int, int>, int>> rest;  // ConcordRest as a min-heap on (refid,refpos); value = matchref
    std::vector<Node>& out;
    int ds = 0, de = 0, dcur = 0;    // itdisstart / itdisend / itdiscurrent
    size_t ps = 0, pe = 0;
    int disChr = 0, otherChr = 0, nextdisChr = 0, disright = 0, otherright = 0, nextdisright = 0;
    int markStart = -1, markChr = -1;

    Seg(const sq_ctx* c, const SegmentInput& in, std::vector<Node>& out) : c(c), in(in), RL(c->read_len), out(out) {}

    El el(int32_t idx) const {
        const StreamRec& r = in.recs[idx];
        return El{r.refid, r.fb_refpos, r.fb_matchref, (int32_t)r.fb_readpos, (bool)(r.flags & SR_REV)};
    }
    static bool el_less(const El& a, const El& b) { return a.refid != b.refid ? a.refid < b.refid : a.refpos < b.refpos; }
    void push_node(int chr, int pos, int len) { out.push_back(Node{chr, pos, len, 0, 0.0}); }
    bool have_back() const { return !out.empty(); }
    int back_end() const { return out.back().pos + out.back().len; }

    void new_cluster() {  // SegmentGraph.cpp:341-348 / 604-611
        disright = nextdisright; disChr = nextdisChr;
        nextdisright = D[ds].refpos + D[ds].matchref;
        for (de = ds; de != nd && D[de].refid == D[ds].refid && D[de].refpos < nextdisright + RL; ++de) {
            nextdisright = std::max(nextdisright, D[de].refpos + D[de].matchref);
            nextdisChr = D[de].refid;
        }
    }

    // heap helpers (MinHeapComp, SegmentGraph.cpp:15-17): std heap algorithms with the same comparator keep the
    // same array layout as the reference's heap; only membership matters for the coverage count
    static bool heap_cmp(const std::pair<std::pair<int, int>, int>& l, const std::pair<std::pair<int, int>, int>& r) { return !(l.first < r.first); }

    void close_node(int chr, int& curStart, int& curEnd, int lastC, bool& split) {  // :483-493 / :506-515
        split = true;
        if (D[ds].refpos - curStart > NEAR && lastC - D[ds].refpos > NEAR) {
            push_node(chr, curStart, D[ds].refpos - curStart);
            curStart = D[ds].refpos;
        }
        push_node(chr, curStart, lastC - curStart);
        curStart = lastC; curEnd = lastC;
        markStart = lastC; markChr = chr;
    }

    // one discordant cluster has been passed by record (recChr, recPos): SegmentGraph.cpp:354-611
    void process_cluster(int recChr, int recPos) {
        int curEnd = 0, curStart = 0, disStart = -1, disEnd = -1, disCount = -1;
        bool split = false;
        if (markStart != -1 && D[ds].refid != markChr) { markChr = -1; markStart = -1; }
        while ((int)cw.size() != co && el(cw[co]).refid < D[ds].refid) ++co;
        while ((int)pw.size() != po && el(pw[po]).refid < D[ds].refid) ++po;
        if ((int)cw.size() != co) { El b = el(cw.back()); if (D[ds].refpos > b.refpos + b.matchref + RL) co = (int)cw.size(); }
        if ((int)pw.size() != po) { El b = el(pw.back()); if (D[ds].refpos > b.refpos + b.matchref + RL) po = (int)pw.size(); }
        curStart = D[ds].refpos;
        {
            bool hc = (int)cw.size() != co, hp = (int)pw.size() != po;
            El t{};
            if (hc && hp) { El x = el(cw[co]), y = el(pw[po]); t = el_less(x, y) ? x : y; }
            else if (hc) t = el(cw[co]);
            else if (hp) t = el(pw[po]);
            if ((hc || hp) && (t.refid < D[ds].refid || (t.refid == D[ds].refid && t.refpos < D[ds].refpos))) curStart = t.refpos;
        }
        curStart = std::max(curStart, markStart);
        while (!rest.empty() && (rest.front().first.first < D[ds].refid || (rest.front().first.first == D[ds].refid && rest.front().first.second < D[ds].refpos - RL))) {
            std::pop_heap(rest.begin(), rest.end(), heap_cmp);
            rest.pop_back();
        }
        for (; ps != part.size() && (part[ps].first < D[ds].refid || (part[ps].first == D[ds].refid && part[ps].second + RL < D[ds].refpos)); ++ps) {}
        for (pe = ps; pe != part.size() && part[pe].first == D[ds].refid && part[pe].second < nextdisright + RL; ++pe) {}

        std::vector<int> M;  // MarginPositions
        while (ds != de) {
            const int chr = D[ds].refid;
            if (ds != 0 && D[ds].refid != D[ds - 1].refid && (int)cw.size() == co && (int)pw.size() == po) curStart = D[ds].refpos;
            split = false;
            M.clear();
            for (dcur = ds; dcur != de; ++dcur) {
                M.push_back(D[dcur].refpos);
                M.push_back(D[dcur].refpos + D[dcur].matchref);
                curEnd = std::max(curEnd, M.back());
                if (dcur + 1 != de && D[dcur + 1].refpos > D[dcur].refpos + D[dcur].matchref) break;
            }
            disStart = std::max(curStart, D[ds].refpos);
            disEnd = curEnd;
            disCount = dcur - ds;
            if (dcur != de)
                for (++dcur; dcur != de && D[dcur].refpos < curEnd + T; ++dcur) { M.push_back(D[dcur].refpos); M.push_back(D[dcur].refpos + D[dcur].matchref); }
            for (size_t q = ps; q != pe && part[q].second < curEnd + T; ++q) M.push_back(part[q].second);
            for (int i = po; i != (int)pw.size(); ++i) {
                El it = el(pw[i]);
                if (it.refid != chr) continue;
                const int front = M.front(), e = it.refpos + it.matchref;
                auto inwin = [&](int x) { return x > front - T && x < curEnd + T; };
                if (it.readpos > 15 && inwin(it.refpos)) {
                    if (it.rev && inwin(e)) M.push_back(e);
                    else if (!it.rev) M.push_back(it.refpos);
                } else {
                    if (it.rev && inwin(it.refpos)) M.push_back(it.refpos);
                    else if (!it.rev && inwin(e)) M.push_back(e);
                }
            }
            std::sort(M.begin(), M.end());

            int lastC = -1, lastSup = 0;
            for (size_t ib = 0; ib < M.size();) {
                const int brk = M[ib];
                size_t nx = ib;
                while (nx < M.size() && M[nx] == brk) ++nx;
                bool skip = have_back() && out.back().chr == chr && brk - back_end() < NEAR;
                if (!skip) {
                    int sr = 0, pl = 0, pr = 0;
                    for (size_t k = 0; k < M.size() && M[k] < brk + T; ++k) if (std::abs(brk - M[k]) < T) ++sr;
                    for (int d = ds; d != de; ++d) {
                        int e = D[d].refpos + D[d].matchref;
                        if (e < brk && e > brk - RL && !D[d].rev) ++pl;
                        else if (D[d].refpos > brk && D[d].refpos < brk + RL && D[d].rev) ++pr;
                    }
                    if (sr > 3 || sr + pl > 4 || sr + pr > 4) {
                        auto spans = [&](int id, int p, int m) { return id == chr && p + m >= brk + T && p < brk - T; };
                        int cov = 0;
                        for (int i = co; i < (int)cw.size(); ++i) { El b = el(cw[i]); if (spans(b.refid, b.refpos, b.matchref)) ++cov; }
                        for (int d = ds; d != de; ++d) if (spans(D[d].refid, D[d].refpos, D[d].matchref)) ++cov;
                        for (int i = po; i != (int)pw.size(); ++i) { El b = el(pw[i]); if (spans(b.refid, b.refpos, b.matchref)) ++cov; }
                        if (sr > std::max(cov - sr, 0) + 2)
                            for (const auto& h : rest) if (spans(h.first.first, h.first.second, h.second)) ++cov;
                        if (sr > std::max(cov - sr, 0) + 2) {
                            int sup = std::max(sr + pl, sr + pr);
                            if (lastC == -1 && brk - curStart < NEAR) { markStart = curStart; markChr = chr; }
                            else if ((lastC == -1 || brk - lastC < NEAR) && sup > lastSup) { lastC = brk; lastSup = sup; }
                            else if (brk - lastC >= NEAR) { close_node(chr, curStart, curEnd, lastC, split); lastC = brk; }
                        }
                    }
                }
                ib = nx;
            }
            if (lastC != -1 && (!split || back_end() != lastC)) close_node(chr, curStart, curEnd, lastC, split);
            if (disStart != -1 && !split && disCount > std::min(5.0, 4.0 * (disEnd - disStart) / RL)) {  // :518-527 (FP64 as in the reference)
                if (have_back() && out.back().chr == D[de - 1].refid && disEnd - back_end() < NEAR) out.back().len += disEnd - back_end();
                else push_node(D[de - 1].refid, disStart, disEnd - disStart);
                curStart = disEnd; curEnd = disEnd;
                markStart = disEnd; markChr = chr;
            }
            while ((int)cw.size() != co && el(cw[co]).refid < chr) ++co;
            while ((int)pw.size() != po && el(pw[po]).refid < chr) ++po;
            for (dcur = ds; dcur != de && D[dcur].refpos + D[dcur].matchref <= curEnd; ++dcur) {}
            int zero = curStart;  // concord0pos
            auto step1 = [&](std::vector<int32_t>& W, int& off) {
                if ((int)W.size() == off) return false;
                El b = el(W[off]);
                bool f = true;
                if (b.refid > chr) f = false;
                if (dcur != nd && b.refid == D[dcur].refid && b.refpos + b.matchref + RL >= D[dcur].refpos) f = false;
                if (have_back() && (b.refid > out.back().chr || (b.refid == out.back().chr && b.refpos >= back_end()))) f = false;
                if (f) { zero = std::max(zero, b.refpos + b.matchref); ++off; }
                return f;
            };
            do {
                bool f1 = step1(cw, co), f2 = step1(pw, po);
                if (!f1 && !f2) break;
            } while ((int)cw.size() != co || (int)pw.size() != po);
            auto step2 = [&](std::vector<int32_t>& W, int& off) {
                if ((int)W.size() == off) return false;
                El b = el(W[off]);
                bool f = dcur == nd || b.refid < D[dcur].refid || (b.refid == D[dcur].refid && b.refpos + b.matchref + RL < D[dcur].refpos);
                if (f) { zero = std::max(zero, b.refpos + b.matchref); ++off; }
                return f;
            };
            do {
                bool cfree = (int)cw.size() == co, pfree = (int)pw.size() == po;
                if (!cfree) { El b = el(cw[co]); cfree = b.refid != markChr || b.refpos > zero + RL; }
                if (!pfree) { El b = el(pw[po]); pfree = b.refid != markChr || b.refpos > zero; }
                if (markStart != -1 && (recChr > markChr || recPos > zero + RL) && cfree && pfree) {
                    if (zero > markStart && zero < markStart + NEAR && have_back() && out.back().chr == markChr) out.back().len += zero - back_end();
                    else if (zero > markStart) push_node(markChr, markStart, zero - markStart);
                    curStart = zero;
                    markChr = -1; markStart = -1;
                    break;
                }
                bool f1 = step2(cw, co), f2 = step2(pw, po);
                if (!f1 && !f2) break;
            } while ((int)cw.size() != co || (int)pw.size() != po);
            ds = dcur;
        }
        if (de - ds <= 0) new_cluster();
    }
};

}  // namespace

int segment_genome(sq_ctx* c, const SegmentInput& in, std::vector<Node>& seeds, int64_t& n_break, std::vector<Blk>& disc_sorted) {
    seeds.clear();
    Seg S(c, in, seeds);
    const int RL = c->read_len;
    // ---- discordant blocks and clip positions of the chimeric fragments (SegmentGraph.cpp:203-264)
    S.part.assign(c->ref_len.size(), std::make_pair(0, 0));  // ledger B10
    std::vector<Blk>& D = S.D;
    for (const Frag& f : c->frags) {
        if (frag_end_discordant(f, true) || frag_end_discordant(f, false) || frag_single_anchored(f) || frag_pair_discordant(f, true)) {
            D.insert(D.end(), f.a.begin(), f.a.end());
            D.insert(D.end(), f.b.begin(), f.b.end());
            continue;
        }
        bool ains = false, bins = false;
        auto far = [&](const std::vector<Blk>& R, bool& ins) {
            int prev = -1;
            for (int i = 0; i + 1 < (int)R.size(); ++i)
                if (std::abs(R[i].refpos - R[i + 1].refpos) > 750000) {
                    if (prev != i) D.push_back(R[i]);
                    D.push_back(R[i + 1]);
                    prev = i + 1;
                    if (i + 1 == (int)R.size() - 1) ins = true;
                }
        };
        far(f.a, ains);
        far(f.b, bins);
        if (!f.a.empty() && !f.b.empty() && std::abs(f.a.back().refpos - f.b.back().refpos) > 750000) {
            if (!ains) { D.push_back(f.a.back()); ains = true; }
            if (!bins) { D.push_back(f.b.back()); bins = true; }
        }
        if (!ains && !bins) {
            auto clip = [&](const Blk& b, bool leftEnd) { return std::make_pair(b.refid, (leftEnd == b.rev) ? b.refpos + b.matchref : b.refpos); };
            if (!f.a.empty() && f.a.front().readpos > 15 && !f.alow) S.part.push_back(clip(f.a.front(), true));
            if (!f.a.empty() && f.atot - f.a.back().readpos - f.a.back().matchread > 15 && !f.alow) S.part.push_back(clip(f.a.back(), false));
            if (!f.b.empty() && f.b.front().readpos > 15 && !f.blow) S.part.push_back(clip(f.b.front(), true));
            if (!f.b.empty() && f.btot - f.b.back().readpos - f.b.back().matchread > 15 && !f.blow) {
                // ledger B11: compared with bamdiscordant.back() even when that vector is empty (zero block here)
                Blk z{0, 0, 0, 0, 0, false, false};
                const Blk& l = D.empty() ? z : D.back();
                const Blk& s = f.b.back();
                bool same = l.refid == s.refid && l.refpos == s.refpos && l.readpos == s.readpos && l.matchread == s.matchread && l.matchref == s.matchref && l.rev == s.rev && l.first == s.first;
                if (!same) S.part.push_back(clip(s, false));
            }
        }
    }
    std::sort(S.part.begin(), S.part.end());
    std::sort(D.begin(), D.end(), [](const Blk& x, const Blk& y) { return x.refid != y.refid ? x.refid < y.refid : x.refpos < y.refpos; });  // ledger B8
    S.nd = (int)D.size();
    disc_sorted = D;
    D.push_back(Blk{0, 0, 0, 0, 0, false, false});  // ledger B21 sentinel
    const int nd = S.nd;

    // ---- the kept stream
    n_break = in.n;  // all kept records feed the depth pass unless the loop leaves early
    for (int64_t i = 0; i < in.n; ++i) {
        const StreamRec& r = in.recs[i];
        if (S.ds == nd) { n_break = i + 1; break; }  // SegmentGraph.cpp:338-339, ledger B12 (this record is already in ReadsMain)
        if (S.de - S.ds <= 0) S.new_cluster();
        while (S.ds != nd && (D[S.ds].refid < r.refid || (D[S.ds].refid == r.refid && S.nextdisright < r.pos))) S.process_cluster(r.refid, r.pos);
        // zero-coverage test for a pending node end (:616-630)
        const bool disLead = S.disChr > S.otherChr || (S.disChr == S.otherChr && S.disright > S.otherright);
        const int curRight = disLead ? S.disright : S.otherright, curChr = std::max(S.disChr, S.otherChr);
        const Blk& dn = D[S.ds];  // the zero sentinel once every cluster is consumed
        const bool zerocov = (r.refid != curChr || r.pos > curRight + RL) && (curChr < dn.refid || (curChr == dn.refid && curRight + RL < dn.refpos));
        if (zerocov && S.markStart != -1) {
            if (curChr == S.markChr && curRight > S.markStart && curRight - S.markStart < Seg::NEAR && !seeds.empty() && S.markStart == S.back_end()) seeds.back().len += curRight - S.markStart;
            else if (curChr == S.markChr && curRight > S.markStart && curRight - S.markStart >= Seg::NEAR) S.push_node(S.markChr, S.markStart, curRight - S.markStart);
            S.markStart = -1; S.markChr = -1;
        }
        // prune the windows (:633-646)
        if (zerocov && (curChr != dn.refid || curRight + RL < dn.refpos)) { S.co = (int)S.cw.size(); S.po = (int)S.pw.size(); }
        else {
            auto prune = [&](std::vector<int32_t>& W, int& off) {
                while ((int)W.size() > off && S.el(W[off]).refid != r.refid) ++off;
                while ((int)W.size() > off) {
                    El b = S.el(W[off]);
                    if (b.refid < dn.refid || (!seeds.empty() && b.refid == seeds.back().chr && b.refpos < S.back_end())) ++off; else break;
                }
            };
            prune(S.cw, S.co);
            prune(S.pw, S.po);
        }
        // push the record (:649-700)
        if (r.flags & SR_CONC) {
            const int e = r.fb_refpos + r.fb_matchref;
            const bool mate = r.flags & SR_MATE;  // the reference keys these updates on IsFirstMate()/IsSecondMate()
            if (mate) {
                if (S.otherChr == r.refid) S.otherright = std::max(S.otherright, e);
                else { S.otherright = e; S.otherChr = r.refid; }
            }
            if (r.flags & SR_PART) S.pw.push_back((int32_t)i); else S.cw.push_back((int32_t)i);
            if (S.ds != nd && mate)
                for (int k = 0; k < r.nrest; ++k) {
                    int p = in.rest_refpos[r.rest_off + k], m = in.rest_matchref[r.rest_off + k];
                    if (p >= D[S.ds].refpos - RL) { S.rest.push_back(std::make_pair(std::make_pair(r.refid, p), m)); std::push_heap(S.rest.begin(), S.rest.end(), Seg::heap_cmp); }
                }
        }
    }
    return SQ_OK;
}

// NormalizeSeedNodes + sanity checks + whole-genome tiling (SegmentGraph.cpp:19-38,706-761)
int tile_genome(sq_ctx* c, std::vector<Node>& seeds, std::vector<Node>& out) {
    const std::vector<int32_t>& RL = c->ref_len;
    if (seeds.size() >= 2) {
        std::sort(seeds.begin(), seeds.end(), [](const Node& a, const Node& b) {
            if (a.chr != b.chr) return a.chr < b.chr;
            if (a.pos != b.pos) return a.pos < b.pos;
            return a.len < b.len;
        });
        std::vector<Node> norm;
        for (const Node& n : seeds) {
            if (norm.empty() || norm.back().chr != n.chr || norm.back().pos + norm.back().len <= n.pos) norm.push_back(n);
            else norm.back().len = std::max(norm.back().pos + norm.back().len, n.pos + n.len) - norm.back().pos;
        }
        seeds.swap(norm);
    }
    for (size_t i = 0; i < seeds.size(); ++i) {
        bool ok = seeds[i].chr >= 0 && seeds[i].chr < (int)RL.size() && seeds[i].len > 0 && seeds[i].pos + seeds[i].len <= RL[seeds[i].chr];
        if (ok && i + 1 < seeds.size()) ok = seeds[i].chr != seeds[i + 1].chr || seeds[i].pos + seeds[i].len <= seeds[i + 1].pos;
        if (!ok) return fail(c, SQ_E_ASSERT, "seed segment violates the node sanity assert (SegmentGraph.cpp:708-712)");
    }
    out.clear();
    auto whole = [&](int chr) { out.push_back(Node{chr, 0, RL[chr], 0, 0.0}); };
    auto fill_tail = [&]() {  // rest of the current chromosome after the last tiled node
        const Node& b = out.back();
        if (b.pos + b.len != RL[b.chr]) out.push_back(Node{b.chr, b.pos + b.len, RL[b.chr] - b.pos - b.len, 0, 0.0});
    };
    for (Node n : seeds) {
        if (out.empty() || out.back().chr != n.chr) {
            if (!out.empty()) fill_tail();
            for (int k = out.empty() ? 0 : out.back().chr + 1; k != n.chr; ++k) whole(k);
            if (n.pos != 0) {
                if (n.pos > 100) out.push_back(Node{n.chr, 0, n.pos, 0, 0.0});
                else { n.len += n.pos; n.pos = 0; out.push_back(n); continue; }
            }
        }
        // (with nothing tiled yet and a seed at position 0 the reference reads tmpNodes.back() of an empty vector)
        const int end = out.empty() ? n.pos : out.back().pos + out.back().len;
        if (end < n.pos) {
            if (n.pos - end > 100) out.push_back(Node{n.chr, end, n.pos - end, 0, 0.0});
            else { n.len += n.pos - end; n.pos = end; }
        }
        out.push_back(n);
    }
    if (!out.empty()) fill_tail();
    for (int k = out.empty() ? 0 : out.back().chr + 1; k < (int)RL.size(); ++k) whole(k);
    return SQ_OK;
}

}  // namespace sq
