// Small-graph stages on the host: after the GPU has
// reduced N_c records to a few thousand unique edges, every stage here touches O(N_nodes + E) items with
// order-dependent sweeps.  Line references are into the reference's src/SegmentGraph.cpp.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>

#include "sq_internal.h"
#include "sq_parsort.h"

namespace sq {

bool edge_discordant(const sq_ctx* c, const std::vector<Node>& N, const Edge& e) {  // :159-190
    if (N[e.a].chr != N[e.b].chr) return true;
    if (N[e.b].pos - N[e.a].pos - N[e.a].len > c->P.concord_dist_pos && e.b - e.a > c->P.concord_dist_idx) return true;
    return e.ha != 0 || e.hb != 1;
}

// ---- LocateRead for chimeric fragments (:1207-1293): sequential hint chain, trims the blocks in place
// The reference walks the node list from the hint, one node at a time, up or down (:1213-1243); with fragments whose
// blocks lie on different chromosomes that is O(nodes) per block (881 ms on the dense-graph config).  The nodes tile
// every chromosome, so the nodes a block fits (+-5 bp) form a range [a, bb] found by two binary searches, and the
// walk ends on the first node of that range it would meet -- or, with an empty range, where it would run out of the
// chromosome.  Same result, same final position of the walk (the next block starts from it).
namespace {
struct ChrRange { int lo, hi; };  // node indices [lo, hi) of a chromosome
inline ChrRange chr_range(const std::vector<Node>& N, int chr) {
    auto lo = std::lower_bound(N.begin(), N.end(), chr, [](const Node& x, int c) { return x.chr < c; });
    auto hi = std::upper_bound(lo, N.end(), chr, [](int c, const Node& x) { return c < x.chr; });
    return ChrRange{(int)(lo - N.begin()), (int)(hi - N.begin())};
}
}  // namespace
int locate_fragment(const std::vector<Node>& N, int hint, Frag& f, std::vector<int>& out) {
    const int n = (int)N.size();
    out.clear();
    int i = hint;
    auto fits = [&](int k, const Blk& b) { return N[k].chr == b.refid && b.refpos >= N[k].pos - 5 && b.refpos + b.matchref <= N[k].pos + N[k].len + 5; };
    auto one = [&](Blk& b) {
        if (i < 0 || i >= n) i = hint;
        if (!fits(i, b)) {
            const ChrRange cr = chr_range(N, b.refid);
            // fitting range: pos_k <= refpos + 5 and end_k >= end - 5 (both monotone inside a chromosome)
            const int a = (int)(std::lower_bound(N.begin() + cr.lo, N.begin() + cr.hi, b.refpos + b.matchref - 5, [](const Node& x, int v) { return x.pos + x.len < v; }) - N.begin());
            const int bb = (int)(std::upper_bound(N.begin() + cr.lo, N.begin() + cr.hi, b.refpos + 5, [](int v, const Node& x) { return v < x.pos; }) - N.begin()) - 1;
            const bool any = a <= bb && a < cr.hi && bb >= cr.lo;
            if (N[i].chr < b.refid || (N[i].chr == b.refid && N[i].pos <= b.refpos)) {
                // upward: the first fitting node at or above i, else the first node of a later chromosome (or n)
                if (any && bb >= i) i = std::max(i, a); else i = std::max(i, cr.hi);
            } else {
                // downward: the first fitting node at or below i, else the last node of an earlier chromosome (or -1)
                if (any && a <= i) i = std::min(i, bb); else i = std::min(i, cr.lo - 1);
            }
        }
        if (i < 0 || i >= n || N[i].chr != b.refid) { out.push_back(-1); return; }
        out.push_back(i);
        const int np = N[i].pos, ne = np + N[i].len;
        if (b.refpos < np) { int d = np - b.refpos; if (!b.rev) b.readpos += d; b.matchref -= d; b.matchread -= d; b.refpos = np; }
        if (b.refpos + b.matchref > ne) { int d = b.refpos + b.matchref - ne; if (b.rev) b.readpos += d; b.matchref -= d; b.matchread -= d; }
    };
    for (Blk& b : f.a) one(b);
    for (Blk& b : f.b) one(b);
    return SQ_OK;
}

static int home_node(const std::vector<Node>& N, int start, const Blk& b) {  // :1408-1409, the two walks as binary searches
    const int n = (int)N.size();
    // up while the node lies in front of the block
    const int j0 = (int)(std::partition_point(N.begin(), N.end(), [&](const Node& x) { return x.chr < b.refid || (x.chr == b.refid && x.pos + x.len < b.refpos); }) - N.begin());
    int i = std::max(start, j0);
    if (i >= n) return -2;
    // down while the node lies behind the block's start
    const int j1 = (int)(std::partition_point(N.begin(), N.end(), [&](const Node& x) { return x.chr < b.refid || (x.chr == b.refid && x.pos <= b.refpos); }) - N.begin()) - 1;
    return std::min(i, j1);
}

static std::pair<int, int> split_breakpoints(const Blk& x, const Blk& y) {  // :1435-1440
    int b1 = x.rev ? x.refpos : x.refpos + x.matchref;
    int b2 = y.rev ? y.refpos + y.matchref : y.refpos;
    bool xgt = x.refid != y.refid ? x.refid > y.refid : x.refpos > y.refpos;
    if (xgt) std::swap(b1, b2);
    return std::make_pair(b1, b2);
}

bool pair_overlap(const Frag& f, const std::vector<int>& rn, int i, int j) {  // :1484-1502
    const int na = (int)f.a.size(), nb = (int)f.b.size();
    bool ov = false;
    for (int k = 0; k < na; ++k) if (j == rn[k]) ov = true;
    for (int k = 0; k < nb; ++k) if (i == rn[na + k]) ov = true;
    if (na > 1) {
        if (frag_end_discordant(f, true)) { if ((rn.front() <= j && rn[na - 1] >= j) || (rn.front() >= j && rn[na - 1] <= j)) ov = true; }
        else if (std::abs(i - j) < 3) ov = true;
    }
    if (nb > 1) {
        if (frag_end_discordant(f, false)) { if ((rn[na] <= i && rn.back() >= i) || (rn[na] >= i && rn.back() <= i)) ov = true; }
        else if (std::abs(i - j) < 3) ov = true;
    }
    return ov;
}

// ---- the fragment loops of RawEdgesChim / ExactBreakpoint on the context's host threads.
// LocateRead carries its position from fragment to fragment (`firstfrontindex`, :1402-1403): the node of a fragment's first block is
// where the search for the next fragment starts.  That matters only when a block fits more than one node; a first block that lies
// deep inside a node (more than 5 bases from both ends) is located there from ANY start.  So the fragments are cut into pieces, and
// the start of every piece is the node of the last such fragment in front of it, provided every fragment between that one and the
// piece has a first block no node can take (then it leaves the position alone).  If a piece boundary cannot be settled that way the
// whole loop runs on one thread -- same results either way.
namespace {
struct FirstFit { bool deep, none; int node; };
inline FirstFit first_block_fit(const std::vector<Node>& N, const Frag& f) {
    const Blk& b = !f.a.empty() ? f.a.front() : f.b.front();
    const ChrRange cr = chr_range(N, b.refid);
    const int end = b.refpos + b.matchref;
    const int a = (int)(std::lower_bound(N.begin() + cr.lo, N.begin() + cr.hi, end - 5, [](const Node& x, int v) { return x.pos + x.len < v; }) - N.begin());
    const int bb = (int)(std::upper_bound(N.begin() + cr.lo, N.begin() + cr.hi, b.refpos + 5, [](int v, const Node& x) { return v < x.pos; }) - N.begin()) - 1;
    const bool any = a <= bb && a < cr.hi && bb >= cr.lo;
    if (!any) return FirstFit{false, true, -1};
    if (a == bb) { const Node& u = N[a]; if (b.refpos >= u.pos && end <= u.pos + u.len && end > u.pos + 5 && b.refpos < u.pos + u.len - 5) return FirstFit{true, false, a}; }
    return FirstFit{false, false, -1};
}
}  // namespace
// (for the record loop of --bwa mode, sq_bwa.cpp: is the fragment's first block located in one node from ANY start of the search?)
bool frag_first_block_pins(const std::vector<Node>& N, const Frag& f, int& node) {
    if (f.a.empty() && f.b.empty()) return false;
    const FirstFit ff = first_block_fit(N, f);
    node = ff.node;
    return ff.deep;
}
namespace {
// piece boundaries over the fragments `skip` does not drop, and the search start of every piece; false: run serially
template <class Skip>
bool plan_pieces(const sq_ctx* c, const std::vector<Node>& N, Skip skip, std::vector<size_t>& cut, std::vector<int>& start) {
    const std::vector<Frag>& F = c->frags;
    const int threads = c->pool ? c->pool->size() + 1 : 1;
    const size_t want = (size_t)std::max(1, 4 * threads), n = F.size();
    cut.assign(1, 0); start.assign(1, 0);
    if (threads <= 1 || n < 20000) { cut.push_back(n); return true; }
    for (size_t k = 1; k < want; ++k) {
        const size_t at = n * k / want;
        if (at <= cut.back()) continue;
        int node = -1;
        bool ok = false;
        for (size_t q = at; q-- > 0 && at - q < 4096;) {  // backwards from the cut to the last fragment that pins the position
            const Frag& f = F[q];
            if (skip(f)) continue;
            const FirstFit ff = first_block_fit(N, f);
            if (ff.deep) { node = ff.node; ok = true; break; }
            if (!ff.none) break;
        }
        if (!ok) return false;
        cut.push_back(at); start.push_back(node);
    }
    cut.push_back(n);
    return true;
}
}  // namespace

// RawEdgesChim (:1394-1555).  Trims c->frags in place, like the reference trims Chimrecord.
int chimeric_edges(sq_ctx* c, std::vector<Edge>& raw) {
    const std::vector<Node>& N = c->nodes;
    const int n = (int)N.size();
    auto skip = [](const Frag& f) { return f.a.empty() && f.b.empty(); };
    std::vector<size_t> cut;
    std::vector<int> start;
    if (!plan_pieces(c, N, skip, cut, start)) { cut = {0, c->frags.size()}; start = {0}; }
    const int np = (int)cut.size() - 1;
    struct Piece { std::vector<Edge> raw; std::vector<uint64_t> disc; bool bad = false; };
    std::vector<Piece> out((size_t)np);
    auto run = [&](int pi) {
        Piece& P = out[(size_t)pi];
        int hint = start[(size_t)pi];
        std::vector<int> rn;
        for (size_t fi = cut[(size_t)pi]; fi < cut[(size_t)pi + 1]; ++fi) {
            Frag& f = c->frags[fi];
            if (skip(f)) continue;
            locate_fragment(N, hint, f, rn);
            if (rn[0] != -1) hint = rn[0];
            const int na = (int)f.a.size();
            for (int k = 0; k < (int)rn.size(); ++k)
                if (rn[k] == -1) {
                    const Blk& b = k < na ? f.a[k] : f.b[k - na];
                    int i = home_node(N, hint, b);
                    if (i < 0 || i + 1 >= n) { P.bad = true; return; }
                    P.raw.push_back(make_edge(i, false, i + 1, true));
                }
            auto split = [&](const BlkList& R, int base) {
                for (int k = 0; k + 1 < (int)R.size(); ++k) {
                    int i = rn[base + k], j = rn[base + k + 1];
                    if (i == j || i == -1 || j == -1) continue;
                    Edge e = make_edge(i, R[k].rev, j, !R[k + 1].rev);
                    if (!edge_discordant(c, N, e)) P.raw.push_back(e);
                    else P.disc.push_back(edge_pack(e));
                }
            };
            split(f.a, 0);
            split(f.b, na);
            if (!f.a.empty() && !f.b.empty() && !frag_end_discordant(f, true) && !frag_end_discordant(f, false)) {
                int i = rn[na - 1], j = rn.back();
                if (i != j && i != -1 && j != -1 && !pair_overlap(f, rn, i, j)) {
                    Edge e = make_edge(i, f.a.back().rev, j, f.b.back().rev);
                    if (!edge_discordant(c, N, e)) P.raw.push_back(e);
                    else if (frag_pair_discordant(f, false)) P.disc.push_back(edge_pack(e));
                }
            }
        }
    };
    if (np > 1 && c->pool) c->pool->parallel_for(np, 1 << 20, [&](int pi) { run(pi); }); else for (int pi = 0; pi < np; ++pi) run(pi);
    std::vector<uint64_t> disc;
    {   // the pieces strung together in order, side by side (millions of entries on a dense sample)
        std::vector<size_t> r_at((size_t)np + 1, raw.size()), d_at((size_t)np + 1, 0);
        for (int pi = 0; pi < np; ++pi) {
            if (out[(size_t)pi].bad) return fail(c, SQ_E_ASSERT, "chimeric block outside the node table (reference: out-of-range edge, SegmentGraph.cpp:1410)");
            r_at[(size_t)pi + 1] = r_at[(size_t)pi] + out[(size_t)pi].raw.size(); d_at[(size_t)pi + 1] = d_at[(size_t)pi] + out[(size_t)pi].disc.size();
        }
        raw.resize(r_at.back()); disc.resize(d_at.back());
        auto put = [&](int pi) { std::copy(out[(size_t)pi].raw.begin(), out[(size_t)pi].raw.end(), raw.begin() + (std::ptrdiff_t)r_at[(size_t)pi]); std::copy(out[(size_t)pi].disc.begin(), out[(size_t)pi].disc.end(), disc.begin() + (std::ptrdiff_t)d_at[(size_t)pi]); };
        if (np > 1 && c->pool) c->pool->parallel_for(np, 1 << 20, put); else for (int pi = 0; pi < np; ++pi) put(pi);
    }
    // discordant edges: one raw edge per key, Weight = number of supporting junctions (:1551); the later sort makes the order irrelevant
    // (equal keys are equal values: any sorting order is THE sorted order -- the threaded introsort, final pass split)
    std_sort_parallel(disc.begin(), disc.end(), std::less<uint64_t>(), c->pool ? std::min(c->pool->size() + 1, 32) : 1, true);
    for (size_t i = 0; i < disc.size();) {
        size_t j = i;
        while (j < disc.size() && disc[j] == disc[i]) ++j;
        Edge e;
        e.a = (int32_t)(disc[i] >> 32); e.b = (int32_t)((disc[i] & 0xffffffffull) >> 2); e.ha = (disc[i] >> 1) & 1; e.hb = disc[i] & 1; e.w = (int32_t)(j - i); e.gw = 0;
        raw.push_back(e);
        i = j;
    }
    return SQ_OK;
}

// sort + sum equal keys + drop non-positive (:1943-1957)
void reduce_edges(std::vector<Edge>& raw, std::vector<Edge>& out, int threads) {
    // (the key is a total order and edges of one key differ in Weight only, which is summed: the tie order cannot be seen)
    std_sort_parallel(raw.begin(), raw.end(), edge_key_less, threads, true);
    out.clear();
    for (const Edge& e : raw) {
        if (out.empty() || !edge_key_eq(e, out.back())) out.push_back(e);
        else out.back().w += e.w;
    }
    out.erase(std::remove_if(out.begin(), out.end(), [](const Edge& e) { return e.w <= 0; }), out.end());
}

// ---------------------------------------------------------------------------------------------- filters
namespace {
struct Adj {  // per-node head/tail edge lists in edge order (UpdateNodeLink, :2894-2909)
    std::vector<std::vector<int>> head, tail;
    void build(int n, const std::vector<Edge>& E) {
        head.assign(n, {});
        tail.assign(n, {});
        for (int i = 0; i < (int)E.size(); ++i) {
            (E[i].ha ? head[E[i].a] : tail[E[i].a]).push_back(i);
            (E[i].hb ? head[E[i].b] : tail[E[i].b]).push_back(i);
        }
    }
};
inline int end1(const std::vector<Node>& N, const Edge& e) { return e.ha ? N[e.a].pos : N[e.a].pos + N[e.a].len; }
inline int end2(const std::vector<Node>& N, const Edge& e) { return e.hb ? N[e.b].pos : N[e.b].pos + N[e.b].len; }
struct Box { int i1lo, i1hi, p1lo, p1hi, i2lo, i2hi, p2lo, p2hi; };
}  // namespace

// FilterbyWeight (:1968-2123); the index slips of ledger B14 are part of the behaviour
void filter_by_weight(sq_ctx* c) {
    std::vector<Edge>& E = c->edges;
    const std::vector<Node>& N = c->nodes;
    const int DI = c->P.concord_dist_idx, DP = c->P.concord_dist_pos, m = (int)E.size();
    std::vector<char> seen(m, 0);
    std::vector<int> grp;
    for (int i = 0; i < m; ++i) {
        if (seen[i]) continue;
        const Edge s = E[i];
        const int chr1 = N[s.a].chr, chr2 = N[s.b].chr;
        grp.assign(1, i);
        seen[i] = 1;
        if (s.ha || !s.hb || chr1 != chr2) {
            Box bx[2];
            bx[0] = Box{s.a, s.a, end1(N, s), end1(N, s), s.b, s.b, end2(N, s), end2(N, s)};
            bx[1] = bx[0];
            bool longgroup = false;
            auto orient = [&](const Edge& e) { return (e.ha == s.ha && e.hb == s.hb) ? 0 : ((e.ha != s.ha && e.hb != s.hb) ? 1 : -1); };
            auto grow2 = [&](Box& b, const Edge& e, int q2) {
                b.i2lo = std::min(b.i2lo, e.b); b.i2hi = std::max(b.i2hi, e.b);
                b.p2lo = std::min(b.p2lo, q2); b.p2hi = std::max(b.p2hi, q2);
                if (b.i1hi >= b.i2lo) longgroup = true;
            };
            for (int j = i - 1; j >= 0 && N[E[j].a].chr == chr1; --j) {
                const Edge& e = E[j];
                const int q1 = end1(N, e), q2 = end2(N, e);
                if (s.a < std::min(bx[0].i1lo, bx[1].i1lo) - DI || q1 < std::min(bx[0].p1lo, bx[1].p1lo) - DP) break;  // s.a (not e.a): B14
                int o = orient(e);
                if (o < 0) continue;
                Box& b = bx[o];
                if (edge_discordant(c, N, e) && e.b >= b.i2lo - DI && s.b <= b.i2hi + DI && q2 >= b.p2lo - DP && q2 <= b.p2hi + DP) {  // s.b: B14
                    grp.push_back(j);
                    b.i1lo = std::min(b.i1lo, e.a); b.p1lo = std::min(b.p1lo, q1);
                    grow2(b, e, q2);
                }
            }
            for (int j = i + 1; j < m && N[E[j].a].chr == chr1; ++j) {
                const Edge& e = E[j];
                const int q1 = end1(N, e), q2 = end2(N, e);
                if (e.a > std::max(bx[0].i1hi, bx[1].i1hi) + DI || q1 > std::max(bx[0].p1hi, bx[1].p1hi) + DP) break;
                int o = orient(e);
                if (o < 0) continue;
                Box& b = bx[o];
                const int upper = o == 0 ? e.b : s.b;  // :2042 vs :2056 (B14)
                if (edge_discordant(c, N, e) && e.b >= b.i2lo - DI && upper <= b.i2hi + DI && q2 >= b.p2lo - DP && q2 <= b.p2hi + DP) {
                    grp.push_back(j);
                    if (o == 0) { b.i1hi = std::max(b.i1hi, e.a); b.p1hi = std::max(b.p1hi, q1); }
                    else { b.i1lo = std::min(b.i1lo, e.a); b.p1lo = std::min(b.p1lo, q1); }  // :2060-2061 updates the lower side
                    grow2(b, e, q2);
                }
            }
            std::sort(grp.begin(), grp.end());
            grp.erase(std::unique(grp.begin(), grp.end()), grp.end());
            int sum = 0;
            for (int k : grp) sum += E[k].w;
            for (int k : grp) { E[k].gw = longgroup ? E[k].w : std::max(E[k].gw, sum); seen[k] = 1; }
        } else {
            const int p1 = end1(N, s), p2 = end2(N, s);
            auto alike = [&](const Edge& e) {
                return s.ha == e.ha && s.hb == e.hb && N[e.a].chr == chr1 && N[e.b].chr == chr2 && std::abs(e.b - s.b) <= DI && std::abs(end1(N, e) - p1) <= DP && std::abs(end2(N, e) - p2) <= DP;
            };
            for (int j = i - 1; j >= 0 && E[j].a >= s.a - DI && N[E[j].a].chr == chr1 && N[E[j].a].pos + N[E[j].a].len >= p1 - DP; --j)
                if (E[j].b > s.a && alike(E[j])) grp.push_back(j);
            for (int j = i + 1; j < m && E[j].a <= s.a + DI && N[E[j].a].chr == chr1 && N[E[j].a].pos <= p1 + DP; ++j)
                if (E[j].a < s.b && alike(E[j])) grp.push_back(j);
            int sum = 0;  // indices are distinct by construction, so the reference's sort+unique is a no-op for the sum
            for (int k : grp) sum += E[k].w;
            E[i].gw = sum;
        }
    }
    const int relaxed = c->P.min_edge_weight - 2;
    E.erase(std::remove_if(E.begin(), E.end(), [&](const Edge& e) { return !(e.gw > relaxed); }), E.end());
}

// FilterbyInterleaving (:2161-2277)
void filter_by_interleaving(sq_ctx* c, std::vector<uint8_t>& keep) {
    const std::vector<Edge>& E = c->edges;
    const std::vector<Node>& N = c->nodes;
    const int DI = c->P.concord_dist_idx, DP = c->P.concord_dist_pos, m = (int)E.size();
    keep.assign(m, 1);
    std::vector<char> seen(m, 0);
    std::vector<int> grp;
    for (int i = 0; i < m; ++i) {
        if (seen[i]) continue;
        const Edge& s = E[i];
        if (s.b - s.a <= DI || (N[s.a].chr == N[s.b].chr && std::abs(N[s.a].pos - N[s.b].pos) <= DP)) { seen[i] = 1; continue; }
        const int chr1 = N[s.a].chr;
        int p1lo = end1(N, s), p1hi = p1lo, i1lo = s.a, i1hi = s.a, p2lo = end2(N, s), p2hi = p2lo, i2lo = s.b, i2hi = s.b;
        bool longgroup = false;
        grp.assign(1, i);
        for (int j = i - 1; j >= 0 && N[E[j].a].chr == chr1; --j) {
            const Edge& e = E[j];
            const int q1 = end1(N, e), q2 = end2(N, e);
            if (s.a < i1lo - DI || q1 < p1lo - DP) break;  // s.a: B14
            if (e.b >= i2lo - DI && s.b <= i2hi + DI && q2 >= p2lo - DP && q2 <= p2hi + DP) {  // s.b: B14
                grp.push_back(j);
                i1lo = std::min(i1lo, e.a); p1lo = std::min(p1lo, q1);
                i2lo = std::min(i2lo, e.b); i2hi = std::max(i2hi, e.b); p2lo = std::min(p2lo, q2); p2hi = std::max(p2hi, q2);
                if (i1hi >= i2lo) { longgroup = true; break; }
            }
        }
        for (int j = i + 1; j < m && N[E[j].a].chr == chr1; ++j) {
            const Edge& e = E[j];
            const int q1 = end1(N, e), q2 = end2(N, e);
            if (e.a > i1hi + DI || q1 > p1hi + DP) break;
            if (e.b >= i2lo - DI && e.b <= i2hi + DI && q2 >= p2lo - DP && q2 <= p2hi + DP) {
                grp.push_back(j);
                i1hi = std::max(i1hi, e.a); p1hi = std::max(p1hi, q1);
                i2lo = std::min(i2lo, e.b); i2hi = std::max(i2hi, e.b); p2lo = std::min(p2lo, q2); p2hi = std::max(p2hi, q2);
                if (i1hi >= i2lo) { longgroup = true; break; }
            }
        }
        if (!longgroup) {
            // partner-index extremes per (group, end); an empty side keeps the value-initialised (0,0) pair
            int lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
            bool has[4] = {false, false, false, false};
            auto upd = [&](int w, int v) { if (!has[w]) { lo[w] = hi[w] = v; has[w] = true; } else { lo[w] = std::min(lo[w], v); hi[w] = std::max(hi[w], v); } };
            for (int k : grp) { upd(E[k].ha ? 0 : 1, E[k].b); upd(E[k].hb ? 2 : 3, E[k].a); }
            bool ov1 = std::min(hi[0], hi[1]) >= std::max(lo[0], lo[1]);  // unconditional: stray ';' at :2265 (B14)
            bool ov2 = has[2] && has[3] && std::min(hi[2], hi[3]) >= std::max(lo[2], lo[3]);
            if (ov1 && ov2) for (int k : grp) keep[k] = 0;
        }
        for (int k : grp) seen[k] = 1;
    }
}

// GroupConnection / GroupSelect / FilterEdges (:2394-2526)
void filter_edges(sq_ctx* c, const std::vector<uint8_t>& keep) {
    std::vector<Edge>& E = c->edges;
    const std::vector<Node>& N = c->nodes;
    const int DP = c->P.concord_dist_pos, MEW = c->P.min_edge_weight, n = (int)N.size();
    Adj adj;
    adj.build(n, E);
    std::vector<int> bad;
    std::vector<Edge> del;
    auto gap = [&](int x, int y) { return N[y].pos - N[x].pos - N[x].len; };
    for (int v = 0; v < n; ++v) {
        const std::vector<int>&H = adj.head[v], &T = adj.tail[v];
        int hw = 0, tw = 0;
        for (int e : H) hw += E[e].w;
        for (int e : T) tw += E[e].w;
        const int sw = hw + tw;
        auto weak = [&](const Edge& e) { return e.gw <= 0.01 * sw && e.gw <= MEW; };
        for (int e : H) if (weak(E[e])) del.push_back(E[e]);
        for (int e : T) if (weak(E[e])) del.push_back(E[e]);
        // neighbour groups on one side (GroupConnection)
        struct Side { std::vector<int> conn, lab; int count = 0; };
        auto connect = [&](const std::vector<int>& L, Side& s) {
            if (L.empty()) return;
            for (int e : L) if (!weak(E[e])) s.conn.push_back(E[e].a != v ? E[e].a : E[e].b);
            std::sort(s.conn.begin(), s.conn.end());
            const int k = (int)s.conn.size();
            s.lab.assign(k, -1);
            int mind = -1, idx = -1;
            for (int i = 0; i < k; ++i) {
                int u = s.conn[i];
                if (N[u].chr == N[v].chr && gap(u, v) <= DP && gap(v, u) <= DP && (mind == -1 || mind > std::abs(v - u))) { mind = std::abs(v - u); idx = i; }
            }
            if (idx != -1) {
                s.lab[idx] = 0;
                for (int i = idx + 1; i < k; ++i) { if (N[s.conn[i]].chr == N[v].chr && gap(s.conn[i - 1], s.conn[i]) <= DP) s.lab[i] = 0; else break; }
                for (int i = idx - 1; i >= 0; --i) { if (N[s.conn[i]].chr == N[v].chr && gap(s.conn[i], s.conn[i + 1]) <= DP) s.lab[i] = 0; else break; }
            }
            if (k) {
                s.count = s.lab[0] == -1 ? 1 : 0;
                if (s.lab[0] == -1) s.lab[0] = 1;
                for (int i = 1; i < k; ++i) {
                    if (s.lab[i] != -1) continue;
                    if (N[s.conn[i]].chr != N[s.conn[i - 1]].chr || gap(s.conn[i - 1], s.conn[i]) > DP) s.count++;
                    s.lab[i] = s.count;
                }
            }
        };
        Side hs, ts;
        connect(H, hs);
        connect(T, ts);
        if (hs.count + ts.count >= c->P.max_allowed_degree) { bad.push_back(v); continue; }
        auto select = [&](const std::vector<int>& L, const Side& s, int sidew) {
            if (s.count > 1) {  // GroupSelect: keep the heaviest neighbour group (and the local group 0)
                std::vector<int> lw(s.count + 1, 0);
                auto labof = [&](const Edge& e) { int u = e.a != v ? e.a : e.b; return s.lab[std::find(s.conn.begin(), s.conn.end(), u) - s.conn.begin()]; };
                for (int e : L) if (!weak(E[e])) lw[labof(E[e])] += E[e].w;
                int best = 1;
                for (int i = 1; i < (int)lw.size(); ++i) if (lw[i] > lw[best]) best = i;
                for (int e : L) if (!weak(E[e])) { int l = labof(E[e]); if (l != best && l != 0) del.push_back(E[e]); }
            } else
                for (int e : L) if (!weak(E[e]) && E[e].gw < 0.01 * sidew) del.push_back(E[e]);
        };
        select(H, hs, hw);
        select(T, ts, tw);
    }
    std::sort(del.begin(), del.end(), edge_key_less);
    std::sort(bad.begin(), bad.end());
    std::vector<Edge> kept;
    for (int i = 0; i < (int)E.size(); ++i) {
        const Edge& e = E[i];
        bool goodends = !std::binary_search(bad.begin(), bad.end(), e.a) && !std::binary_search(bad.begin(), bad.end(), e.b);
        bool nearby = N[e.a].chr == N[e.b].chr && std::abs(N[e.b].pos - N[e.a].pos - N[e.a].len) <= DP;
        bool cond1 = (goodends || nearby) && e.gw > MEW;
        bool cond2 = true;
        if (cond1 && (e.b - e.a > c->P.concord_dist_idx || e.ha != 0 || e.hb != 1)) {
            auto ratio_of = [](double c1, double c2) { return (c1 > c2) ? c1 / c2 : c2 / c1; };  // x/0 = inf deletes, 0/0 = NaN keeps (IEEE, as in the reference)
            auto passes = [&](double ratio) { return !((e.w <= MEW + 2 && ratio > 3) || (e.w > MEW + 2 && ratio > 50)); };
            cond2 = passes(ratio_of(N[e.a].depth, N[e.b].depth));
            const Node &na = N[e.a], &nb = N[e.b];
            if (c->depth_bounds && (na.depth_lo != na.depth_hi || nb.depth_lo != nb.depth_hi)) {
                // the depths are only known up to the tie order of the reference's ReadsOther sort: the decision must be
                // the same over the whole box [lo,hi] x [lo,hi], otherwise the exact (sorted) sweep is required
                // (a depth of exactly 0 makes the ratio inf -- still a well defined decision -- unless BOTH depths can be 0,
                // where 0/0 = NaN flips the outcome)
                bool stable = !(na.depth_lo <= 0 && nb.depth_lo <= 0);
                if (stable) {
                    double sup = std::max(ratio_of(na.depth_hi, nb.depth_lo), ratio_of(na.depth_lo, nb.depth_hi));
                    bool overlap = !(na.depth_lo > nb.depth_hi || nb.depth_lo > na.depth_hi);
                    double inf = overlap ? 1.0 : std::min(ratio_of(na.depth_lo, nb.depth_hi), ratio_of(na.depth_hi, nb.depth_lo));
                    stable = passes(sup) == passes(inf) && passes(sup) == cond2;
                }
                if (!stable && std::getenv("SQUID_DEBUG_DEPTH"))
                    std::fprintf(stderr, "depth-ambiguous edge (%d,%d) w=%d: a=[%g,%g,%g] b=[%g,%g,%g]\n", e.a, e.b, e.w, na.depth_lo, na.depth, na.depth_hi, nb.depth_lo, nb.depth, nb.depth_hi);
                if (!stable) c->depth_ambiguous = true;
            }
        }
        if (keep[i] && cond1 && cond2) kept.push_back(e);
    }
    std::sort(kept.begin(), kept.end(), edge_key_less);
    std::vector<Edge> out(kept.size());
    out.resize(std::set_difference(kept.begin(), kept.end(), del.begin(), del.end(), out.begin(), edge_key_less) - out.begin());
    E.swap(out);
}

// CompressNode (:2528-2604): every maximal run of edge-less nodes on one chromosome becomes one node
int compress_nodes(sq_ctx* c) {
    std::vector<Node>& N = c->nodes;
    std::vector<Edge>& E = c->edges;
    const int n = (int)N.size();
    if (E.empty()) return fail(c, SQ_E_ASSERT, "0 nodes are connected by edges (the reference asserts at SegmentGraph.cpp:2537)");
    std::vector<char> linked(n, 0);
    for (const Edge& e : E) { linked[e.a] = 1; linked[e.b] = 1; }
    std::vector<Node> out;
    std::vector<int> remap(n, -1);
    int i = 0;
    while (i < n) {
        if (linked[i]) { remap[i] = (int)out.size(); out.push_back(N[i]); ++i; continue; }
        int j = i;
        while (j < n && !linked[j] && N[j].chr == N[i].chr) ++j;
        Node t{N[i].chr, N[i].pos, N[j - 1].pos + N[j - 1].len - N[i].pos, 0, 0.0};
        for (int k = i; k < j; ++k) { t.support += N[k].support; t.depth += N[k].depth * N[k].len; }  // same summation order as :2552-2556
        t.depth /= t.len;
        out.push_back(t);
        i = j;
    }
    for (Edge& e : E) { e.a = remap[e.a]; e.b = remap[e.b]; }
    N.swap(out);
    return SQ_OK;
}

// FurtherCompressNode (:2693-2892)
int further_compress(sq_ctx* c) {
    std::vector<Node>& N = c->nodes;
    std::vector<Edge>& E = c->edges;
    const int n = (int)N.size(), DI = c->P.concord_dist_idx;
    Adj adj;
    adj.build(n, E);
    std::vector<int> merge(n, -1);
    int cur = 0, rightmost = 0;
    auto sameheads = [](const Edge& x, const Edge& y) { return x.ha == y.ha && x.hb == y.hb; };
    auto closeidx = [&](const Edge& x, const Edge& y) { return std::abs(x.a - y.a) <= DI && std::abs(x.b - y.b) <= DI; };
    auto samechr = [&](const Edge& x, const Edge& y) { return N[x.a].chr == N[y.a].chr && N[x.b].chr == N[y.b].chr; };
    auto discordant_of = [&](int v, std::vector<Edge>& out, bool track) {
        for (int pass = 0; pass < 2; ++pass)
            for (int e : (pass ? adj.tail[v] : adj.head[v])) {
                if (edge_discordant(c, N, E[e])) out.push_back(E[e]);
                else if (track) rightmost = std::max(rightmost, std::max(E[e].a, E[e].b));
            }
    };
    auto next_with = [&](int i, int limit, std::vector<Edge>& nx) {
        int j = i + 1;
        for (; j < n && j < i + 20 && j < limit && N[i].chr == N[j].chr; ++j) { discordant_of(j, nx, false); if (!nx.empty()) break; }
        return j;
    };
    auto cross = [&](const std::vector<Edge>& A, const std::vector<Edge>& B) {
        std::vector<char> am(A.size(), 0), bm(B.size(), 0);
        for (size_t k = 0; k < A.size(); ++k)
            for (size_t l = 0; l < B.size(); ++l)
                if (A[k].b > B[l].a && B[l].b > A[k].a && samechr(A[k], B[l]) && closeidx(A[k], B[l]) && sameheads(A[k], B[l])) { am[k] = 1; bm[l] = 1; }
        for (char x : am) if (!x) return false;
        for (char x : bm) if (!x) return false;
        return true;
    };
    // drop an edge when it continues the previous one's group; `anchor` >= 0 additionally demands the same end at that node
    auto squeeze = [&](std::vector<Edge>& V, int anchor, bool needchr) {
        std::vector<Edge> t(1, V[0]);
        for (size_t k = 0; k + 1 < V.size(); ++k) {
            const Edge &x = V[k], &y = V[k + 1];
            bool same = closeidx(x, y) && sameheads(x, y) && (!needchr || samechr(x, y));
            if (anchor >= 0) same = same && ((x.a == anchor && y.a == anchor) || (x.b == anchor && y.b == anchor));
            if (!same) t.push_back(y);
        }
        V.swap(t);
    };
    for (int i = 0; i < n; ++i) {
        int limit = i + 20;  // minDisInd2
        std::vector<Edge> mine, nx;
        if (i != 0 && N[i].chr != N[i - 1].chr && cur == merge[i - 1]) ++cur;
        discordant_of(i, mine, true);
        if (!mine.empty()) {
            limit = mine[0].a == i ? mine[0].b : i + 20;
            for (size_t k = 1; k < mine.size(); ++k) limit = std::min(limit, mine[k].a == i ? mine[k].b : i + 20);
            squeeze(mine, i, false);
        }
        if (merge[i] == -1) {
            if (mine.empty() && i < rightmost) merge[i] = cur;
            else if (mine.empty() && i == rightmost) { merge[i] = cur++; ++rightmost; }
            else {
                if (i != 0 && cur == merge[i - 1]) ++cur;
                int j = next_with(i, limit, nx);
                bool eq = !nx.empty();
                if (eq) { squeeze(nx, j, true); eq = cross(mine, nx); }
                if (!eq) merge[i] = cur++;
                else for (int k = i; k <= j; ++k) merge[k] = cur;
                rightmost = i + 1;
            }
        } else if (!mine.empty()) {
            int j = next_with(i, limit, nx);
            bool eq = !nx.empty();
            if (eq) { squeeze(mine, -1, true); squeeze(nx, -1, true); eq = cross(mine, nx); }
            if (!eq) ++cur;
            else for (int k = i; k <= j; ++k) merge[k] = cur;
            rightmost = i + 1;
        }
    }
    for (int i = 0; i + 1 < n; ++i)
        if (!(merge[i] == merge[i + 1] || merge[i] + 1 == merge[i + 1])) return fail(c, SQ_E_ASSERT, "node merge map is not contiguous (the reference asserts at SegmentGraph.cpp:2862)");
    std::vector<Node> out;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && merge[j] == merge[i]) ++j;
        out.push_back(Node{N[i].chr, N[i].pos, N[j - 1].pos + N[j - 1].len - N[i].pos, 0, 0.0});  // Support/AvgDepth reset (ledger B15)
        i = j;
    }
    std::vector<Edge> raw;
    for (const Edge& e : E) if (merge[e.a] != merge[e.b]) raw.push_back(make_edge(merge[e.a], e.ha, merge[e.b], e.hb, e.w));
    std::sort(raw.begin(), raw.end(), edge_key_less);
    E.clear();
    for (const Edge& e : raw) { if (E.empty() || !edge_key_eq(e, E.back())) E.push_back(e); else E.back().w += e.w; }
    N.swap(out);
    return SQ_OK;
}

// Multiply / DeMultiplyDisEdges (:3005-3017), cast order of ledger B16
void multiply_discordant(sq_ctx* c, bool undo) {
    const double r = c->P.discordant_ratio;
    if (r == 1) return;
    for (Edge& e : c->edges)
        if (edge_discordant(c, c->nodes, e)) e.w = undo ? (int)(e.w / r) : (int)r * e.w;
}

// CountTop (:51-102)
static void count_top(const Edge& e, std::vector<std::pair<int, int>>& x) {
    std::sort(x.begin(), x.end());
    std::vector<std::pair<int, int>> y(x);
    y.erase(std::unique(y.begin(), y.end()), y.end());
    std::vector<double> score(y.size(), 0);
    for (size_t i = 0; i < y.size(); ++i)
        for (const auto& p : x) {
            if (p == y[i]) score[i] += 1;
            else if (std::abs(y[i].first - p.first) + std::abs(y[i].second - p.second) < 10) score[i] += 0.5;
        }
    x.clear();
    while (x.size() < 5) {
        size_t k = std::max_element(score.begin(), score.end()) - score.begin();
        if (!(score[k] > 3)) break;
        bool far = true;
        for (const auto& p : x) if (std::abs(p.first - y[k].first) + std::abs(p.second - y[k].second) < 50) far = false;
        if (far) x.push_back(y[k]);
        score[k] = 0;
    }
    if (x.empty()) {
        int lo1 = std::numeric_limits<int>::max(), lo2 = lo1, hi1 = 0, hi2 = 0;
        for (const auto& p : y) { lo1 = std::min(lo1, p.first); hi1 = std::max(hi1, p.first); lo2 = std::min(lo2, p.second); hi2 = std::max(hi2, p.second); }
        x.push_back(std::make_pair(e.ha ? lo1 : hi1, e.hb ? lo2 : hi2));
    }
}

// ExactBreakpoint (:3019-3081) on the final nodes; trims c->frags further, like the reference
int exact_breakpoints(sq_ctx* c, BPMap& bp) {
    bp.clear();
    const std::vector<Node>& N = c->nodes;
    auto skip = [](const Frag& f) { return f.a.size() <= 1 && f.b.size() <= 1; };
    std::vector<size_t> cut;
    std::vector<int> start;
    if (!plan_pieces(c, N, skip, cut, start)) { cut = {0, c->frags.size()}; start = {0}; }
    const int np = (int)cut.size() - 1;
    struct Hit { uint64_t key; int b1, b2; };
    std::vector<std::vector<Hit>> out((size_t)np);
    auto run = [&](int pi) {
        std::vector<Hit>& H = out[(size_t)pi];
        int hint = start[(size_t)pi];
        std::vector<int> rn;
        for (size_t fi = cut[(size_t)pi]; fi < cut[(size_t)pi + 1]; ++fi) {
            Frag& f = c->frags[fi];
            if (skip(f)) continue;
            locate_fragment(N, hint, f, rn);
            if (rn[0] != -1) hint = rn[0];
            auto collect = [&](const BlkList& R, int base) {
                for (int k = 0; k + 1 < (int)R.size(); ++k) {
                    int i = rn[base + k], j = rn[base + k + 1];
                    if (i == j || i == -1 || j == -1) continue;
                    Edge e = make_edge(i, R[k].rev, j, !R[k + 1].rev);
                    if (edge_discordant(c, N, e)) { const std::pair<int, int> q = split_breakpoints(R[k], R[k + 1]); H.push_back(Hit{edge_pack(e), q.first, q.second}); }
                }
            };
            collect(f.a, 0);
            collect(f.b, (int)f.a.size());
        }
    };
    if (np > 1 && c->pool) c->pool->parallel_for(np, 1 << 20, [&](int pi) { run(pi); }); else for (int pi = 0; pi < np; ++pi) run(pi);
    std::vector<Hit> all;
    {
        std::vector<size_t> at((size_t)np + 1, 0);
        for (int pi = 0; pi < np; ++pi) at[(size_t)pi + 1] = at[(size_t)pi] + out[(size_t)pi].size();
        all.resize(at.back());
        auto put = [&](int pi) { std::copy(out[(size_t)pi].begin(), out[(size_t)pi].end(), all.begin() + (std::ptrdiff_t)at[(size_t)pi]); };
        if (np > 1 && c->pool) c->pool->parallel_for(np, 1 << 20, put); else for (int pi = 0; pi < np; ++pi) put(pi);
    }
    // (count_top sorts the pairs anyway; a total order on all three fields: equal elements are equal values, any sort gives this list)
    std_sort_parallel(all.begin(), all.end(), [](const Hit& x, const Hit& y) { return x.key != y.key ? x.key < y.key : (x.b1 != y.b1 ? x.b1 < y.b1 : x.b2 < y.b2); },
                      c->pool ? std::min(c->pool->size() + 1, 32) : 1, true);
    std::vector<size_t> grp;
    for (size_t i = 0; i < all.size(); ++i) if (i == 0 || all[i].key != all[i - 1].key) grp.push_back(i);
    grp.push_back(all.size());
    const int ng = (int)grp.size() - 1;
    static const bool prof = std::getenv("SQUID_BP_PROF") != nullptr;
    if (prof) std::fprintf(stderr, "exact_breakpoints: %zu hits in %d groups, %d pieces\n", all.size(), ng, np);
    std::vector<std::vector<std::pair<int, int>>> lists((size_t)ng);
    auto top = [&](int g) {
        std::vector<std::pair<int, int>>& x = lists[(size_t)g];
        for (size_t i = grp[(size_t)g]; i < grp[(size_t)g + 1]; ++i) x.push_back(std::make_pair(all[i].b1, all[i].b2));
        const uint64_t key = all[grp[(size_t)g]].key;
        Edge e;
        e.a = (int32_t)(key >> 32); e.b = (int32_t)((key & 0xffffffffull) >> 2); e.ha = (key >> 1) & 1; e.hb = key & 1; e.w = 1; e.gw = 0;
        count_top(e, x);
    };
    if (ng > 256 && c->pool) {
        const int pieces = std::min(ng, 8 * (c->pool->size() + 1));
        c->pool->parallel_for(pieces, 1 << 20, [&](int k) { for (int g = (int)((int64_t)ng * k / pieces); g < (int)((int64_t)ng * (k + 1) / pieces); ++g) top(g); });
    } else for (int g = 0; g < ng; ++g) top(g);
    for (int g = 0; g < ng; ++g) bp.emplace_hint(bp.end(), all[grp[(size_t)g]].key, std::move(lists[(size_t)g]));
    return SQ_OK;
}

}  // namespace sq
