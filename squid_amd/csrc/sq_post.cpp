// Whole-genome order of the segments: what squid does with the per-component orders after Ordering()
// -- src/main.cpp:45-48 -- before it prints them with -TO (`_component.txt`) and spells them out with -RG (`_genome.fa`):
//   SortComponents (src/SegmentGraph.cpp:4010-4041)   by the median segment id; a component that runs mostly backwards is flipped
//   MergeSingleton (:4043-4137)                       single segments and runs of consecutive ones are put back between their
//                                                     neighbours inside the larger components (:4139-4423)
//   MergeComponents (:4425-4504)                      components of one chromosome are strung together
// `_sv.txt` does not depend on any of this, and the placement step is quadratic in the reference's formulation, so the
// library only runs it when asked (sq_total_order).  Host code: a few thousand to a few hundred thousand integers, order-dependent.
//
// Two reads of the reference are undefined behaviour; the library fixes them the way the test oracle does: the orientation flags of a
// placement start as "not reversed" (:4167,4178 read them uninitialised when no neighbour was found on either side and the graph has
// fewer than 25 segments), and the look at Consecutive[idx] with idx one past the end (:4064,4077) is skipped -- its value is unused.
#include <algorithm>
#include <cstdlib>
#include <functional>

#include "sq_internal.h"

namespace sq {

namespace {

typedef std::vector<int> Comp;          // signed 1-based segment ids
typedef std::vector<Comp> Comps;

struct Geo {  // what the stitching looks at of the segments
    const std::vector<Node>& N;
    const std::vector<int32_t>& ref_len;
    int chr(int v) const { return N[(size_t)std::abs(v) - 1].chr; }
    int pos(int v) const { return N[(size_t)std::abs(v) - 1].pos; }
    int end(int v) const { const Node& n = N[(size_t)std::abs(v) - 1]; return n.pos + n.len; }
    int size() const { return (int)N.size(); }
};

int median_id(const Comp& c) {  // lower median of the absolute ids
    std::vector<int> a(c.size());
    for (size_t i = 0; i < c.size(); ++i) a[i] = std::abs(c[i]);
    std::nth_element(a.begin(), a.begin() + (a.size() - 1) / 2, a.end());
    return a[(a.size() - 1) / 2];
}

// :4010-4041
Comps sort_components(const Comps& in) {
    // a later component with the same median replaces an earlier one in the reference's map, and the sorted medians keep their
    // multiplicity: a repeated median brings the same (last) component twice
    std::vector<std::pair<int, int>> med(in.size());  // (median, index)
    for (size_t i = 0; i < in.size(); ++i) med[i] = std::make_pair(median_id(in[i]), (int)i);
    std::vector<std::pair<int, int>> by = med;
    std::stable_sort(by.begin(), by.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    Comps out(in.size());
    for (size_t i = 0; i < by.size(); ++i) {
        size_t last = i;  // the last component that has this median
        while (last + 1 < by.size() && by[last + 1].first == by[i].first) ++last;
        Comp c = in[(size_t)by[last].second];
        if (c.size() == 1 && c[0] < 0) c[0] = -c[0];
        int down = 0;
        for (size_t j = 0; j + 1 < c.size(); ++j) if (std::abs(c[j]) > std::abs(c[j + 1])) ++down;
        const int half = (int)c.size() / 2;
        if (down > half || (down == half && std::abs(c.front()) > std::abs(c.back()))) {
            std::reverse(c.begin(), c.end());
            for (int& v : c) v = -v;
        }
        out[i] = std::move(c);
    }
    return out;
}

// One thing to be placed: a single segment (lo == hi) or a run of consecutive ones [lo .. hi]; `anchor` = the id whose chromosome
// counts (the segment itself / the median of the run).
struct Piece { Comp ids; int lo, hi, anchor; };
struct Place { int at; bool forward; int piece; };  // insert in front of element `at` of the component

// Where a piece goes (:4154-4225 and :4312-4371 are the same search, once with lo == hi): between two neighbouring elements of a
// component whose ids enclose it most tightly (within 50 ids, not both pointing the wrong way), or at one end of the component whose
// median is nearest.  Returns the component (-1: nowhere) and fills `pl`.
int find_place(const Geo& G, const Comps& comps, const std::vector<int>& medians, const Piece& p, bool single, Place& pl) {
    const int NV = G.size(), pchr = G.chr(p.anchor);
    int best_adj = 50, adj_comp = -1, adj_at = 0;  // (signed: negative = the piece goes in reversed)
    int med_d1 = NV, med_d2 = NV, med_comp = -1;
    for (size_t j = 0; j < comps.size(); ++j) {
        const Comp& C = comps[j];
        const int n = (int)C.size();
        for (int k = 0; k + 1 < n; ++k) {
            // ascending around the gap (smaller id in front, larger behind), then descending.  (The two orientation flags live across
            // both tries, as in the reference: the second try sees the first one's flag where it finds no neighbour of its own.)
            bool fsmall = false, flarge = false;
            for (int dir = 0; dir < 2; ++dir) {
                int dsmall = NV, dlarge = NV;
                auto look = [&](int l, bool want_small) {
                    const int a = std::abs(C[(size_t)l]);
                    if (G.chr(a) != pchr) return;
                    if (want_small) { if (a < p.lo && p.lo - a < dsmall) { dsmall = p.lo - a; fsmall = dir == 0 ? C[(size_t)l] < 0 : C[(size_t)l] > 0; } }
                    else if (a > p.hi && a - p.hi < dlarge) { dlarge = a - p.hi; flarge = dir == 0 ? C[(size_t)l] < 0 : C[(size_t)l] > 0; }
                };
                for (int l = std::max(0, k - 1); l <= k; ++l) look(l, dir == 0);
                for (int l = k + 1; l < std::min(n, k + 3); ++l) look(l, dir != 0);
                if (dsmall + dlarge < std::abs(best_adj) && !(fsmall && flarge)) { best_adj = dir == 0 ? dsmall + dlarge : -(dsmall + dlarge); adj_comp = (int)j; adj_at = k; }
            }
        }
        const int ref = single ? p.lo : p.anchor;  // (the single-segment search measures from the segment, the run search from its median)
        if (G.chr(medians[j]) == pchr && std::abs(medians[j] - ref) < med_d1)
            for (int k = 0; k < n; ++k)
                if (std::abs(std::abs(C[(size_t)k]) - ref) < std::abs(med_d2)) { med_d2 = std::abs(C[(size_t)k]) - ref; med_d1 = std::abs(medians[j] - ref); med_comp = (int)j; }
    }
    if (adj_comp != -1 && (adj_comp == med_comp || std::abs(best_adj) < std::abs(med_d2))) { pl.at = adj_at + 1; pl.forward = best_adj > 0; return adj_comp; }
    if (med_comp != -1) {
        if (med_d2 < 0) { pl.at = (int)comps[(size_t)med_comp].size(); pl.forward = true; return med_comp; }
        if (med_d2 > 0 || !single) { pl.at = 0; pl.forward = true; return med_comp; }
        return -2;  // a single segment that IS an element of the nearest component: the reference drops it (:4237-4242)
    }
    return -1;
}

// :4139-4294 / :4296-4423: place every piece, then rebuild the components with the pieces in
void insert_pieces(const Geo& G, const std::vector<Piece>& pieces, bool single, Comps& comps) {
    std::vector<int> medians(comps.size());
    for (size_t j = 0; j < comps.size(); ++j) medians[j] = median_id(comps[j]);
    std::vector<std::vector<Place>> per(comps.size());
    std::vector<int> left_over;
    for (size_t i = 0; i < pieces.size(); ++i) {
        Place pl{0, true, (int)i};
        const int where = find_place(G, comps, medians, pieces[i], single, pl);
        if (where >= 0) per[(size_t)where].push_back(pl);
        else if (where == -1) left_over.push_back((int)i);
    }
    for (size_t j = 0; j < comps.size(); ++j) {
        std::vector<Place>& P = per[j];
        if (P.empty()) continue;
        std::sort(P.begin(), P.end(), [&](const Place& a, const Place& b) { return a.at != b.at ? a.at < b.at : pieces[(size_t)a.piece].ids.front() < pieces[(size_t)b.piece].ids.front(); });
        Comp out;
        size_t q = 0;
        auto emit_group = [&](size_t q_end) {  // the pieces that share one gap
            Comp grp;
            size_t reversed = 0;
            for (; q < q_end; ++q) {
                const Comp& ids = pieces[(size_t)P[q].piece].ids;
                if (P[q].forward) grp.insert(grp.end(), ids.begin(), ids.end());
                else if (single) { grp.push_back(-ids[0]); ++reversed; }
                else { Comp r(ids.rbegin(), ids.rend()); for (int& v : r) v = -v; grp.insert(grp.begin(), r.begin(), r.end()); }  // (a reversed run goes in FRONT of what the gap holds so far)
            }
            if (single && reversed > grp.size() / 2) std::reverse(grp.begin(), grp.end());
            out.insert(out.end(), grp.begin(), grp.end());
        };
        for (int k = 0; k < (int)comps[j].size(); ++k) {
            size_t q_end = q;
            while (q_end < P.size() && P[q_end].at <= k) ++q_end;
            if (q_end > q) emit_group(q_end);
            out.push_back(comps[j][(size_t)k]);
        }
        if (q < P.size()) emit_group(P.size());
        comps[j].swap(out);
    }
    for (int i : left_over) comps.push_back(pieces[(size_t)i].ids);
}

// :4043-4137
Comps merge_singletons(const Geo& G, const Comps& in) {
    auto spans_chromosome = [&](const Comp& c) { return G.pos(c.front()) == 0 && G.end(c.back()) == G.ref_len[(size_t)G.chr(c.front())]; };
    auto consecutive = [&](const Comp& c) {
        for (size_t j = 0; j + 1 < c.size(); ++j) if (c[j + 1] - c[j] != 1 || G.chr(c[j + 1]) != G.chr(c[j])) return false;
        return true;
    };
    auto is_run = [&](const Comp& c) { return consecutive(c) && !spans_chromosome(c); };
    Comps keep, runs;
    for (const Comp& c : in) if (c.size() != 1) (is_run(c) ? runs : keep).push_back(c);
    // the single segments in id order: neighbours form runs of their own or extend an existing run at either end
    std::vector<int> singles;
    Comp cur;
    size_t ri = 0;
    auto close_run = [&]() {
        const int key_chr = G.chr(cur[(cur.size() - 1) / 2]);
        for (; ri < runs.size() && runs[ri].back() + 1 <= cur[0]; ++ri)
            if (runs[ri].back() + 1 >= cur[0] && G.chr(runs[ri][(runs[ri].size() - 1) / 2]) == key_chr) break;
        const bool have = ri < runs.size();
        const bool same_chr = have && G.chr(runs[ri][(runs[ri].size() - 1) / 2]) == key_chr;
        if (same_chr && cur.back() == runs[ri].front() - 1 && (cur.size() > 1 || cur[0] == runs[ri].front() - 1)) runs[ri].insert(runs[ri].begin(), cur.begin(), cur.end());
        else if (same_chr && cur[0] == runs[ri].back() + 1) runs[ri].insert(runs[ri].end(), cur.begin(), cur.end());
        else if (cur.size() == 1) singles.push_back(cur[0]);
        else runs.push_back(cur);
    };
    for (const Comp& c : in) {
        if (c.size() != 1) continue;
        const int v = c[0];
        if (G.pos(v) == 0 && G.end(v) == G.ref_len[(size_t)G.chr(v)]) { keep.push_back(c); continue; }  // a whole chromosome
        if (cur.empty() || (cur.back() + 1 == v && G.chr(cur.back()) == G.chr(v))) { cur.push_back(std::abs(v)); continue; }
        close_run();
        cur.assign(1, std::abs(v));
    }
    if (cur.size() > 1) runs.push_back(cur);
    else if (cur.size() == 1) singles.push_back(cur[0]);
    {
        std::vector<Piece> pieces(singles.size());
        for (size_t i = 0; i < singles.size(); ++i) pieces[i] = Piece{Comp(1, singles[i]), singles[i], singles[i], singles[i]};
        insert_pieces(G, pieces, true, keep);
    }
    // components that have become plain runs by the insertion join the runs (kept in id order), neighbouring runs fuse (:4098-4133)
    Comps runs2, keep2;
    ri = 0;
    for (Comp& c : keep) {
        if (c.size() == 1 || !is_run(c)) { keep2.push_back(std::move(c)); continue; }
        for (; ri < runs.size() && runs[ri].back() < c.front(); ++ri) runs2.push_back(runs[ri]);
        runs2.push_back(std::move(c));
    }
    for (; ri < runs.size(); ++ri) runs2.push_back(runs[ri]);
    Comps fused;
    for (Comp& r : runs2) {
        if (!fused.empty() && fused.back().back() + 1 == r.front() && G.chr(fused.back().back()) == G.chr(r.back())) fused.back().insert(fused.back().end(), r.begin(), r.end());
        else fused.push_back(std::move(r));
    }
    std::vector<Piece> pieces(fused.size());
    for (size_t i = 0; i < fused.size(); ++i) pieces[i] = Piece{fused[i], std::abs(fused[i].front()), std::abs(fused[i].back()), median_id(fused[i])};
    insert_pieces(G, pieces, false, keep2);
    return keep2;
}

// :4425-4504; main leaves the length cut-off at its default of 5 bases (src/SegmentGraph.h:118)
Comps merge_components(const Geo& G, const Comps& in, int len_cutoff = 5) {
    std::vector<int> margins;  // first id (1-based: index + 1 of the last segment) of every chromosome change
    for (int i = 0; i + 1 < G.size(); ++i) if (G.N[(size_t)i].chr != G.N[(size_t)i + 1].chr) margins.push_back(i + 1);
    Comps out;
    std::vector<int> med;  // medians of `out`, kept up to date
    for (size_t i = 0; i < in.size(); ++i) {
        const Comp& c = in[i];
        if (out.empty()) { out.push_back(c); med.push_back(median_id(c)); continue; }
        long len = 0;
        for (int v : c) len += G.N[(size_t)std::abs(v) - 1].len;
        const int cm = median_id(c);
        size_t plus_c = out.size(), minus_c = out.size();
        std::ptrdiff_t plus_at = 0, minus_at = 0;
        int ind = 0, diff = std::abs(cm - med[0]) + 1;
        for (size_t j = 0; j < out.size(); ++j)
            if (std::abs(med[j] - cm) < diff) {
                for (size_t q = 0; q < out[j].size(); ++q) {
                    const int a = std::abs(out[j][q]);
                    if (a == std::abs(c.front()) - 1) { minus_at = (std::ptrdiff_t)q; minus_c = j; }
                    else if (a == std::abs(in[j].back()) + 1) { plus_at = (std::ptrdiff_t)q; plus_c = j; }  // in[j], not in[i]: ledger B20
                }
                diff = std::abs(med[j] - cm);
                ind = (int)j;
            }
        bool other_chr = false;
        for (int m : margins) if ((med[(size_t)ind] <= m && cm > m) || (med[(size_t)ind] > m && cm <= m)) { other_chr = true; break; }
        const bool around = plus_c != out.size() && plus_c == minus_c && len < len_cutoff;
        size_t touched;
        if (other_chr) { out.push_back(c); med.push_back(cm); continue; }
        if (around && minus_at - plus_at == 1 && !(out[plus_c][(size_t)plus_at] > 0 && out[minus_c][(size_t)minus_at] > 0)) {
            Comp r(c.rbegin(), c.rend());
            for (int& v : r) v = -v;
            out[minus_c].insert(out[minus_c].begin() + minus_at, r.begin(), r.end());
            touched = minus_c;
        } else if (around && minus_at - plus_at == -1 && !(out[plus_c][(size_t)plus_at] < 0 && out[minus_c][(size_t)minus_at] < 0)) {
            out[plus_c].insert(out[plus_c].begin() + plus_at, c.begin(), c.end());
            touched = plus_c;
        } else {
            out[(size_t)ind].insert(out[(size_t)ind].end(), c.begin(), c.end());
            touched = (size_t)ind;
        }
        med[touched] = median_id(out[touched]);
    }
    return out;
}

}  // namespace

int total_order(sq_ctx* c) {
    if (!c->ordered) return fail(c, SQ_E_ARG, "sq_total_order before sq_order");
    Comps comps(c->ord_off.size() ? c->ord_off.size() - 1 : 0);
    for (size_t k = 0; k + 1 < c->ord_off.size(); ++k) comps[k].assign(c->ord_nodes.begin() + c->ord_off[k], c->ord_nodes.begin() + c->ord_off[k + 1]);
    const Geo G{c->nodes, c->ref_len};
    comps = sort_components(comps);
    comps = merge_singletons(G, comps);
    comps = sort_components(comps);
    comps = merge_components(G, comps);
    c->tot_off.assign(1, 0);
    c->tot_nodes.clear();
    for (const Comp& k : comps) { c->tot_nodes.insert(c->tot_nodes.end(), k.begin(), k.end()); c->tot_off.push_back((int32_t)c->tot_nodes.size()); }
    return SQ_OK;
}

}  // namespace sq
